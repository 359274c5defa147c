#!/bin/bash
# usage: prof_configs.sh <tag> [--only name,name]
# rocprofv3 evidence for EVERY configuration bench.py prints: one --kernel-trace run for durations, then one --pmc
# run per counter group (a trace domain is never combined with --pmc; FETCH_SIZE and WRITE_SIZE in passes of their own, as
# MI355X_MICROARCH.md prescribes), all over the same workload driver (tools/prof_configs.py, which brackets every configuration
# with marker launches).  tools/prof_summarize.py cuts each pass into per-configuration windows and writes
#   gpurun_out/prof_<tag>/{summary.json, pmc_current.json, kernel_stats.csv, manifest.json}
# (the cycle split of the metric's kernels -- issuing / issue-stalled / parked, instruction cache, per-class cycles -- is tools/prof_residue.sh)
# then, from the SAME build: kernel_usage.txt (per-kernel registers / scratch / LDS / occupancy, tools/kernel_usage.py) and
# bench_full_line.json (bench.py's line with the fresh counter facts attached)   -> copy all to profiles/<tag>/ and profiles/pmc_current.json
TAG=${1:-r03_configs}; shift
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o p -- python3 tools/prof_configs.py --manifest $OUT/manifest.json "$@" > $OUT/trace.log 2>&1
cp $OUT/trace/p_kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
run() { name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -o p -- python3 tools/prof_configs.py $EXTRA > $OUT/$name.log 2>&1; }
EXTRA="$@"
run mix SQ_INSTS_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE
run stall SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES
run fetch FETCH_SIZE
run write WRITE_SIZE
python3 tools/prof_summarize.py $OUT "profiles/$TAG"
python3 tools/kernel_usage.py plk_pairing plk_quad plk_multi plk_verify plk_group g1 hash sign sign_wide tower runtime > $OUT/kernel_usage.txt 2>&1
cp profiles/pmc_current.json $OUT/pmc_previous.json 2>/dev/null
cp $OUT/pmc_current.json profiles/pmc_current.json          # on the GPU box's copy of the tree: bench.py reads it from there
python3 bench.py --steps ${BENCH_STEPS:-10} --warmup 3 > $OUT/bench_full_line.json 2> $OUT/bench.err
