#!/bin/bash
# The 1024 -> 2048-wavefront knee of the one-wavefront-per-element kernels (k_miller_wide_batch, k_final_exp_wide_batch): instruction-cache
# and instruction-fetch counters per dispatch against the grid size.  Separate --pmc passes, no trace domain next to them.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/knee; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o p -- python3 tools/dbg/knee_driver.py > $OUT/trace.log 2>&1
run() { name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -o p -- python3 tools/dbg/knee_driver.py > $OUT/$name.log 2>&1; }
run icache SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_TC_INST_REQ GRBM_GUI_ACTIVE
run ifetch SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU
run lds SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAVES
python3 - <<'PY'
import csv, glob, collections, json, os
out = "gpurun_out/knee"
def rows(path):
    f = glob.glob(path)
    return list(csv.DictReader(open(f[0]))) if f else []
summary = {}
trace = rows(out + "/trace/*kernel_trace.csv")
dur = collections.defaultdict(list)
for r in trace:
    k = r["Kernel_Name"].split("(")[0]
    if "wide_batch" in k:
        dur[(k, int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r["Grid_Size"]))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for (k, g), v in sorted(dur.items()):
    summary.setdefault(k, {}).setdefault(str(g), {})["us"] = sum(v) / len(v)
for name in ("icache", "ifetch", "lds"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows(out + "/" + name + "/*counter_collection.csv"):
        k = r["Kernel_Name"].split("(")[0]
        if "wide_batch" not in k: continue
        acc[(k, int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for (k, g), cs in acc.items():
        for c, v in cs.items():
            summary.setdefault(k, {}).setdefault(str(g), {})[c] = sum(v) / len(v)
json.dump(summary, open(out + "/summary.json", "w"), indent=1, sort_keys=True)
for k, per in summary.items():
    print(k)
    for g in sorted(per, key=int):
        e = per[g]
        w = int(g) // 64
        print("  waves %5d: %8.1f us  icache req/wave %9.0f  miss/wave %8.0f  miss rate %.4f  tc_inst_req/wave %8.0f  wait_inst_any/wave_cycles %.3f  valu/wave %9.0f  lds_wait/wave_cycles %.3f" % (
            w, e.get("us", 0), e.get("SQC_ICACHE_REQ", 0) / w, e.get("SQC_ICACHE_MISSES", 0) / w,
            e.get("SQC_ICACHE_MISSES", 0) / max(e.get("SQC_ICACHE_REQ", 1), 1), e.get("SQC_TC_INST_REQ", 0) / w,
            e.get("SQ_WAIT_INST_ANY", 0) / max(e.get("SQ_WAVE_CYCLES", 1), 1), e.get("SQ_INSTS_VALU", 0) / w,
            e.get("SQ_WAIT_INST_LDS", 0) / max(e.get("SQ_WAVE_CYCLES", 1), 1)))
PY
