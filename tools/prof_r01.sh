#!/bin/bash
# rocprofv3 recipe used for profiles/r01_*: kernel-trace stats, then PMC passes (separate runs).
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_r01
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o pairing -- python3 bench.py --steps 3 --warmup 1 --no-cpu > $OUT/bench_trace.log 2>&1
grep '^{' $OUT/bench_trace.log > $OUT/bench_trace.json
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -o pairing -- python3 bench.py --steps 1 --warmup 0 --no-cpu --no-aux > $OUT/bench_pmc_sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pairing -- python3 bench.py --steps 1 --warmup 0 --no-cpu --no-aux > $OUT/bench_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pairing -- python3 bench.py --steps 1 --warmup 0 --no-cpu --no-aux > $OUT/bench_pmc_write.log 2>&1
find $OUT -type f | head -40
for f in $(find $OUT -name '*kernel_stats.csv'); do echo "== $f"; cat $f; done
