#!/bin/bash
# usage: ab_g2.sh libA.so libB.so [rounds] -- tools/dbg/time_g2.py (pairing, G1 / G2 scalar mul, Miller loop, subgroup check at 2^20) under two builds on ONE box
A=$1; B=$2; R=${3:-1}
for r in $(seq $R); do for L in $A $B; do echo "== $(basename $L)"; SYLOW_HIP_LIB=$L python tools/dbg/time_g2.py | tail -5; done; done
