#!/bin/bash
# tools/ubench/call_cost under the wave-cycle counters: what does SQ_WAIT_ANY book for a call + return that costs 56-68 cycles by s_memtime?
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/call_cost; mkdir -p gpurun_out/call_cost
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --output-format csv -d gpurun_out/call_cost -o p -- tools/ubench/call_cost > gpurun_out/call_cost/log 2>&1
python3 - <<'PY'
import csv, glob, collections
rows = collections.OrderedDict()
for r in sorted(csv.DictReader(open(glob.glob("gpurun_out/call_cost/**/p_counter_collection.csv", recursive=True)[0])), key=lambda r: int(r["Dispatch_Id"])):
    d = rows.setdefault(int(r["Dispatch_Id"]), {"k": r["Kernel_Name"].split("(")[0], "g": r["Grid_Size"], "c": collections.defaultdict(float)})
    d["c"][r["Counter_Name"]] += float(r["Counter_Value"])
seen = set()
for d in rows.values():
    key = (d["k"], d["g"])
    if key in seen: continue
    seen.add(key)
    c = d["c"]; wc = c["SQ_WAVE_CYCLES"]; waves = c["SQ_WAVES"] or 1
    n = 4096
    print("%-34s grid %7s: per iteration per wavefront: %.1f instructions, wave quad-cycles %.1f, parked %.1f quad-cycles (%.3f), issue-stalled %.3f" % (
        d["k"][:34], d["g"], (c["SQ_INSTS_VALU"] + c["SQ_INSTS_SALU"]) / waves / n, wc / waves / n, c["SQ_WAIT_ANY"] / waves / n, c["SQ_WAIT_ANY"] / wc, c["SQ_WAIT_INST_ANY"] / wc))
PY
