"""One pairing_batch call per size through the one-wavefront-per-element route (n <= 2048), for the instruction-cache counters of
tools/dbg/knee.sh.  Each size runs 3 times so that the per-dispatch counter rows can be averaged."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, sylow_amd
from bench import make_points
eng = sylow_amd.Engine(0)
for n in (256, 512, 1024, 1536, 2048):
    p, q, ka, kb = make_points(eng, n, 5)
    gt = eng.empty((48, n))
    for _ in range(3):
        eng._call("sylow_hip_pairing_batch", p.ptr, None, q.ptr, None, gt.ptr, n)
    eng.sync()
