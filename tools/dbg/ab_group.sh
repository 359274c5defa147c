#!/bin/bash
# usage: ab_group.sh libA.so libB.so -- group-law timings (tools/dbg/time_g2.py, time_group.py, time_decode.py, time_gtpow.py) of two builds
# alternated on ONE box
cd $GRAFT_REPO_ROOT
for r in 1 2; do for L in "$@"; do echo "== $L"; for s in time_g2 time_group time_decode time_gtpow; do SYLOW_HIP_LIB=$PWD/$L python3 tools/dbg/$s.py 2>&1 | tail -9; done; done; done
