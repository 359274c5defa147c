cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/leaf_loop; mkdir -p gpurun_out/leaf_loop
tools/ubench/leaf_loop
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VALU_INT64 GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/leaf_loop -o p -- tools/ubench/leaf_loop > gpurun_out/leaf_loop/log 2>&1
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in csv.DictReader(open(glob.glob("gpurun_out/leaf_loop/**/p_counter_collection.csv", recursive=True)[0])):
    k = r["Kernel_Name"].split("(")[0]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in acc.items():
    wc = v["SQ_WAVE_CYCLES"]
    i64, valu = v["SQ_INSTS_VALU_INT64"], v["SQ_INSTS_VALU"]
    print(k, "issuing %.4f  issue-stalled %.4f  parked %.4f   VALU per SALU %.1f   INT64 share %.3f   IDEAL-ISSUE FRACTION %.4f" % (v["SQ_ACTIVE_INST_ANY"] / wc, v["SQ_WAIT_INST_ANY"] / wc, v["SQ_WAIT_ANY"] / wc, valu / max(v["SQ_INSTS_SALU"], 1), i64 / valu,
          (4 * i64 + 2 * (valu - i64)) / (v["GRBM_GUI_ACTIVE"] / 8 * 1024)))
PY
