#!/usr/bin/env python3
"""Per-kernel resource table (VGPRs, spilled VGPRs, scratch bytes per lane, LDS, occupancy) of the lane-pair units, from hipcc's
-Rpass-analysis=kernel-resource-usage remarks.  `python tools/kernel_usage.py [unit ...]`  (no GPU needed)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "sylow_amd", "csrc")
units = sys.argv[1:] or ["plk_pairing", "plk_multi", "plk_verify", "plk_group", "g1", "hash", "sign", "tower", "runtime"]
for u in units:
    r = subprocess.run(["hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Rpass-analysis=kernel-resource-usage",
                        "-c", os.path.join(CSRC, u + ".hip"), "-o", "/dev/null"], capture_output=True, text=True)
    cur = None
    rows = {}
    for line in r.stderr.splitlines():
        m = re.search(r"remark: (?:\S+: )?Function Name: (\S+)", line)
        if m:
            cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip().split("(")[0]
            rows[cur] = {}
            continue
        m = re.search(r"remark:\s+(VGPRs|VGPR Spill|ScratchSize \[bytes/lane\]|LDS Size \[bytes/block\]|Occupancy \[waves/SIMD\]|AGPRs): (\d+)", line)
        if m and cur:
            rows[cur][m.group(1)] = int(m.group(2))
    print(f"== {u}.hip")
    print(f"{'kernel':58s} {'VGPR':>5s} {'spill':>6s} {'scratch B':>10s} {'LDS B':>7s} {'waves/SIMD':>10s} {'(LDS-limited, 256-thread blocks)':>s}")
    for k, v in rows.items():
        lds = v.get('LDS Size [bytes/block]', 0)
        occ = v.get('Occupancy [waves/SIMD]', 0)
        by_lds = (160 * 1024 // lds) if lds else 99          # blocks per CU the 160 KB of LDS admit; a block of 256 threads is one wave per SIMD
        note = f"{min(occ, by_lds):d}" + ("  <-- LDS caps the occupancy" if by_lds < occ else "")
        print(f"{k[:58]:58s} {v.get('VGPRs', 0):5d} {v.get('VGPR Spill', 0):6d} {v.get('ScratchSize [bytes/lane]', 0):10d} "
              f"{lds:7d} {occ:10d} {note}")
