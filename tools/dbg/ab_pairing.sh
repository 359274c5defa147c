#!/bin/bash
# usage: ab_pairing.sh libA.so libB.so [rounds]  -- alternate two builds of libsylow_hip.so on ONE box (separate processes), k_pairing at 2^20
A=$1; B=$2; R=${3:-3}
for r in $(seq $R); do
  for L in $A $B; do
    SYLOW_HIP_LIB=$L python - <<PY
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch, sylow_amd
from bench import make_points
eng = sylow_amd.Engine(0)
n = 1 << 20
p, q, ka, kb = make_points(eng, n, 5)
gt = eng.empty((48, n))
def timed(fn, reps=4):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
t = timed(lambda: eng._call("sylow_hip_pairing_batch", p.ptr, None, q.ptr, None, gt.ptr, n))
print("%-40s pairing %.2f ms" % (os.path.basename(os.environ["SYLOW_HIP_LIB"]), t))
PY
  done
done
