cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/tail_trace; mkdir -p gpurun_out/tail_trace
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tail_trace -o p -- python3 tools/dbg/time_quad.py 49152 > gpurun_out/tail_trace/log 2>&1
python3 - <<'PY'
import csv, glob
rows = sorted(csv.DictReader(open(glob.glob("gpurun_out/tail_trace/**/p_kernel_trace.csv", recursive=True)[0])), key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    if "pairing" in r["Kernel_Name"]:
        print("%-40s grid %8s  start %10.3f ms  end %10.3f ms  queue %s" % (r["Kernel_Name"][:40], r["Grid_Size_X"], (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6, r.get("Queue_Id")))
PY
