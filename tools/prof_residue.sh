#!/bin/bash
# usage: prof_residue.sh <tag> [configs]
# Where do the SIMD cycles of the metric's kernels go that are NOT ideal VALU issue?  (round-5 review item 2.)  The same workload driver and
# marker windows as prof_configs.sh, restricted to the two kernels of the metric, with the SQ / SQC counter groups that split a wavefront's
# cycles into issuing / stalled-on-issue / parked, the instruction-fetch path, and the per-class instruction cycles.  Counter names that this
# ROCm does not know are dropped from a group (the list rocprofv3 -L prints is saved beside the results).  8 SQ slots per pass.
#   -> gpurun_out/prof_<tag>/{summary.json, avail.txt}; tools/prof_residue_table.py prints the split
TAG=${1:-r06_residue}; ONLY=${2:-pairing_2^20,bls_sign_2^20,bls_verify_2^20}      # the verifier needs the signatures the sign configuration writes
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 -L > $OUT/avail.txt 2>&1
have() { grep -qw "$1" $OUT/avail.txt; }
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o p -- python3 tools/prof_configs.py --manifest $OUT/manifest.json --only "$ONLY" > $OUT/trace.log 2>&1
cp $OUT/trace/p_kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
run() {
  name=$1; shift; list=""
  for c in "$@"; do if have $c; then list="$list $c"; else echo "counter $c not available" >> $OUT/dropped.txt; fi; done
  [ -z "$list" ] && return
  rocprofv3 --pmc $list --output-format csv -d $OUT/$name -o p -- python3 tools/prof_configs.py --only "$ONLY" > $OUT/$name.log 2>&1
}
run mix SQ_INSTS_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE
run cyc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAVES
run cls SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT
run fet SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_INSTS_BRANCH SQ_INSTS_SMEM
run mem SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_WAIT_INST_LDS SQ_INSTS_VSKIPPED SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_SENDMSG
run thr SQ_THREAD_CYCLES_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_ACCUM_PREV
python3 tools/prof_summarize.py $OUT "profiles/$TAG" > $OUT/table.txt 2>&1
python3 tools/prof_residue_table.py $OUT/summary.json >> $OUT/table.txt 2>&1
cat $OUT/table.txt
