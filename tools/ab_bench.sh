#!/bin/bash
# A/B two builds of libsylow_hip.so in one session (same box, interleaved) -- clocks drift between boxes
for rep in 1 2 3; do
  for v in A B; do
    echo -n "$v: "; SYLOW_HIP_LIB=$PWD/tools/ab/lib_$v.so python bench.py --no-cpu --no-aux --steps 3 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']))"
  done
done
