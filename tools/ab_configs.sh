#!/bin/bash
for v in A B; do echo "== $v"; SYLOW_HIP_LIB=$PWD/tools/ab/lib_$v.so python tools/bench_configs.py 2>/dev/null | grep -A1 "pairing_2\|miller\|final_exp" | grep per_s; done
