#!/bin/bash
# One box, every stage of round 4: pairing / verify time and VALU count (ab_raw), group-law timings, ecPairing / decode, small-batch latency.
# usage: ab_round.sh  (libraries from tools/ab/)   -> gpurun_out/ab_round/*.log
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/ab_round; rm -rf $O; mkdir -p $O
STAGES="lib_r03 lib_head lib_sign3 lib_iso3 lib_chain2 lib_wideiso"
bash tools/dbg/ab_raw.sh $(for s in $STAGES; do echo tools/ab/$s.so; done) > $O/pairing_verify.log 2>&1
for s in lib_head lib_sign3 lib_iso3 lib_wideiso; do
  echo "== $s"
  for t in time_g2 time_group time_decode time_small; do SYLOW_HIP_LIB=$PWD/tools/ab/$s.so python3 tools/dbg/$t.py 2>&1 | grep -E " ms" ; done
done > $O/group_decode_small.log 2>&1
rm -rf gpurun_out/ab_raw
