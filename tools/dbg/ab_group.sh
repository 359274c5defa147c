#!/bin/bash
# usage: ab_group.sh libA.so libB.so -- group-law timings (tools/dbg/time_g2.py + time_group.py) of two builds alternated on ONE box
cd $GRAFT_REPO_ROOT
for r in 1 2; do for L in "$@"; do echo "== $L"; SYLOW_HIP_LIB=$PWD/$L python3 tools/dbg/time_g2.py 2>&1 | tail -8; SYLOW_HIP_LIB=$PWD/$L python3 tools/dbg/time_group.py 2>&1 | tail -4; done; done
