"""Summarise the rocprofv3 passes of tools/prof_configs.sh: per configuration (the window between two marker launches of
tools/prof_configs.py) the kernels that ran, their durations and instruction-class counts, and the two roofs --

  issue_frac = sum over the window's kernels of (SQ_INSTS_VALU_INT64 x 4 + other VALU x 2 issue cycles)
               / sum of (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)            [both from the SAME --pmc pass: no clock enters]
  hbm        = (2 x FETCH_SIZE + WRITE_SIZE) KiB per launch (the gfx950 correction of MI355X_MICROARCH.md), against the algorithmic bytes

Quarter-rate class calibration: profiles/r02_valu_class_calibration.txt (SQ_INSTS_VALU_INT64 = v_mad_[iu]64_[iu]32, 64-bit shifts / adds:
4 issue cycles per wave64 instruction; everything else 2; v_mul_lo_u32 -- 4 cycles, ~2 % of these streams -- sits in the 2-cycle class,
so the fraction is a slight under-estimate).

    python3 tools/prof_summarize.py <pass dir> <profiles tag for the "source" field>
"""
import collections
import csv
import glob
import hashlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_SIMD, N_XCD = 1024, 8
ALGO_BYTES = {"pairing": 576, "Fp op": 96, "scalar-mul": None, "job": None, "hash": 32 + 64, "signature": None, "verify": 225}


HOST_ONLY_UNITS = ("pipeline.hip",)     # no device code: editing them does not change any kernel


def csrc_hash():
    """Hash of the kernel sources with comments and blank space removed (a comment edit is not a new build): the committed PMC summary
    (profiles/pmc_current.json) is only valid for the build it was measured on."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "sylow_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".hpp")) and name not in HOST_ONLY_UNITS:
            with open(os.path.join(d, name), "r", encoding="utf-8") as f:
                text = f.read()
            text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
            text = re.sub(r"//[^\n]*", " ", text)
            text = " ".join(text.split())
            h.update(name.encode() + b"\0" + text.encode())
    return h.hexdigest()[:16]


def short(name):
    return name.split("(")[0].replace("void ", "").strip()


def windows(rows, mark_base, n_cfg, grid_of):
    """rows: dispatches in order; returns list of lists (window i = rows between marker 2i and 2i + 1)"""
    out = [[] for _ in range(n_cfg)]
    cur = None
    for r in rows:
        g = grid_of(r)
        if "k_fp_unop" in r["Kernel_Name"] and g % 256 == 0 and g // 256 <= 2 * n_cfg + 2 and g <= 256 * 4096:
            m = g // 256 - 1
            if m % 2 == 0 and m // 2 < n_cfg:
                cur = m // 2
            else:
                cur = None
            continue
        if cur is not None:
            out[cur].append(r)
    return out


def main():
    out_dir, tag = sys.argv[1], sys.argv[2]
    man = json.load(open(os.path.join(out_dir, "manifest.json")))
    cfgs = man["configs"]
    n_cfg = len(cfgs)
    # durations from the kernel trace
    trace = sorted(csv.DictReader(open(glob.glob(os.path.join(out_dir, "trace", "**", "p_kernel_trace.csv"), recursive=True)[0])), key=lambda r: int(r["Dispatch_Id"]))
    tw = windows(trace, man["mark_base"], n_cfg, lambda r: int(r["Grid_Size_X"]))
    res = {}
    for c, rows in zip(cfgs, tw):
        k = collections.OrderedDict()
        for r in rows:
            e = k.setdefault(short(r["Kernel_Name"]), {"calls": 0, "ns": 0.0})
            e["calls"] += 1
            e["ns"] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        span = (max(float(r["End_Timestamp"]) for r in rows) - min(float(r["Start_Timestamp"]) for r in rows)) if rows else 0.0
        ksum = sum(e["ns"] for e in k.values())
        # a configuration that forks work onto a side stream (the aggregate verifiers) has overlapping dispatches: its kernel durations add up
        # to more than the time it takes, and the span first-start .. last-end is the honest figure; serial configurations keep the kernel sum
        # (their span additionally holds the launch gaps)
        res[c["name"]] = {"units": c["units"], "unit": c["unit"], "launches": c["reps"], "overlapped_dispatches": span < ksum,
                          "ms_per_launch": min(span, ksum) / c["reps"] / 1e6,
                          "ms_per_launch_kernel_sum": ksum / c["reps"] / 1e6, "ms_per_launch_span": span / c["reps"] / 1e6,
                          "kernels": {name: {"calls_per_launch": e["calls"] / c["reps"], "ms_per_launch": e["ns"] / c["reps"] / 1e6} for name, e in k.items()}}
    # counters
    passes = sorted(d for d in os.listdir(out_dir) if d != "trace" and os.path.isdir(os.path.join(out_dir, d)))     # mix, stall, fetch, write (+ the residue groups of prof_residue.sh)
    for pas in passes:
        files = glob.glob(os.path.join(out_dir, pas, "**", "p_counter_collection.csv"), recursive=True)
        if not files:
            continue
        rows = list(csv.DictReader(open(files[0])))
        by_disp = collections.OrderedDict()
        for r in sorted(rows, key=lambda r: int(r["Dispatch_Id"])):
            d = by_disp.setdefault(int(r["Dispatch_Id"]), {"Kernel_Name": r["Kernel_Name"], "Grid_Size": int(r["Grid_Size"]), "c": {}})
            d["c"][r["Counter_Name"]] = d["c"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        cw = windows(list(by_disp.values()), man["mark_base"], n_cfg, lambda r: r["Grid_Size"])
        for c, ds in zip(cfgs, cw):
            e = res[c["name"]]
            for d in ds:
                ke = e["kernels"].setdefault(short(d["Kernel_Name"]), {})
                pc = ke.setdefault("pmc_per_launch", {})
                for name, v in d["c"].items():
                    pc[name] = pc.get(name, 0.0) + v / c["reps"]
    for name, e in res.items():
        tot = collections.defaultdict(float)
        for ke in e["kernels"].values():
            for cn, v in ke.get("pmc_per_launch", {}).items():
                tot[cn] += v
            pc = ke.get("pmc_per_launch", {})
            if "SQ_INSTS_VALU" in pc and pc.get("GRBM_GUI_ACTIVE"):
                i64 = pc.get("SQ_INSTS_VALU_INT64", 0.0)
                ke["issue_frac"] = (4 * i64 + 2 * (pc["SQ_INSTS_VALU"] - i64)) / (pc["GRBM_GUI_ACTIVE"] / N_XCD * N_SIMD)
        u = e["units"]
        if "SQ_INSTS_VALU" in tot and tot.get("GRBM_GUI_ACTIVE"):
            i64, valu = tot.get("SQ_INSTS_VALU_INT64", 0.0), tot["SQ_INSTS_VALU"]
            ideal, avail = 4 * i64 + 2 * (valu - i64), tot["GRBM_GUI_ACTIVE"] / N_XCD * N_SIMD
            e.update({"valu_instr_per_unit": valu / u, "valu_int64_per_unit": i64 / u, "int64_class_frac": i64 / valu if valu else 0.0,
                      "issue_cycles_ideal_per_unit": ideal / u, "issue_cycles_measured_rates_per_unit": (4.19 * i64 + 2.31 * (valu - i64)) / u,
                      "simd_cycles_per_unit": avail / u, "issue_frac": ideal / avail, "issue_frac_vs_measured_issue_rates": (4.19 * i64 + 2.31 * (valu - i64)) / avail})
        if "SQ_WAVE_CYCLES" in tot and tot["SQ_WAVE_CYCLES"]:
            e["wait_any_frac_of_wave_cycles"] = tot.get("SQ_WAIT_ANY", 0.0) / tot["SQ_WAVE_CYCLES"]
            e["vmem_instr_per_unit"] = (tot.get("SQ_INSTS_VMEM_RD", 0.0) + tot.get("SQ_INSTS_VMEM_WR", 0.0)) / u
        if "FETCH_SIZE" in tot and "WRITE_SIZE" in tot:
            e["hbm_bytes_per_unit"] = (2 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024 / u
            ms = e["ms_per_launch"]
            e["hbm_TBps"] = e["hbm_bytes_per_unit"] * u / (ms * 1e-3) / 1e12 if ms else None
    summary = {"source": tag, "csrc_hash": csrc_hash(),
               "method": "tools/prof_configs.sh: kernel-trace pass for durations, --pmc passes mix / stall / fetch / write; windows between marker launches; per launch = window / reps",
               "configs": res}
    json.dump(summary, open(os.path.join(out_dir, "summary.json"), "w"), indent=1)
    # the file bench.py reads: per configuration only what the JSON line needs
    slim = {"source": tag, "csrc_hash": summary["csrc_hash"], "configs": {}}
    for name, e in res.items():
        slim["configs"][name] = {k: e.get(k) for k in ("units", "unit", "ms_per_launch", "overlapped_dispatches", "ms_per_launch_kernel_sum", "valu_instr_per_unit", "valu_int64_per_unit", "int64_class_frac",
                                                       "issue_cycles_ideal_per_unit", "issue_cycles_measured_rates_per_unit", "simd_cycles_per_unit", "issue_frac",
                                                       "issue_frac_vs_measured_issue_rates", "hbm_bytes_per_unit", "wait_any_frac_of_wave_cycles")}
        slim["configs"][name]["kernels"] = list(e["kernels"])
    json.dump(slim, open(os.path.join(out_dir, "pmc_current.json"), "w"), indent=1)
    for name, e in res.items():
        print("%-36s %9.3f ms  issue %.3f  hbm %s B/unit  valu/unit %s" % (name, e["ms_per_launch"], e.get("issue_frac", float("nan")),
              ("%.0f" % e["hbm_bytes_per_unit"]) if "hbm_bytes_per_unit" in e else "-", ("%.0f" % e["valu_instr_per_unit"]) if "valu_instr_per_unit" in e else "-"))


if __name__ == "__main__":
    main()
