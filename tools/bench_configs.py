#!/usr/bin/env python3
"""Throughput of the other BASELINE.json configs on one GPU (device-resident, HIP-event timed):
C2a Fp mul (HBM-bound), C2b G1 scalar-mul, C3 pairings 2^18, C5 ecPairing jobs, hash-to-G1, sign, verify."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sylow_amd
from bench import limbs_row, make_points, G1, G2

eng = sylow_amd.Engine(0)
rand_scalars_soa = lambda seed, n: eng.xoshiro_fp_soa(seed, n)
def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3
res = {}
# C2a: Fp mul / add on 2^24 elements (1.6 GB of traffic per launch; 2^20 is only 96 MB = ~15 us)
for log2n in (20, 24):
    n = 1 << log2n
    a, b, o = eng.empty((4, n)).upload(rand_scalars_soa(1, n)), eng.empty((4, n)).upload(rand_scalars_soa(2, n)), eng.empty((4, n))
    for name in ("mul", "add"):
        t = timed(lambda: eng._call(f"sylow_hip_fp_{name}_batch", a.ptr, b.ptr, o.ptr, n), 20)
        res[f"fp_{name}_2^{log2n}"] = {"elems_per_s": n / t, "GBps_algorithmic": 96 * n / t / 1e9, "frac_of_8TBps": 96 * n / t / 8e12}
    del a, b, o
# C2b: G1 scalar mul 2^20
n = 1 << 20
p, q, ka, kb = make_points(eng, n, 5)
o, oi = eng.empty((8, n)), eng.empty((n,), np.uint8)
t = timed(lambda: eng._call("sylow_hip_g1_scalar_mul_batch", p.ptr, None, ka.ptr, o.ptr, oi.ptr, n), 2)
res["g1_scalar_mul_2^20"] = {"per_s": n / t}
o2, o2i = eng.empty((16, n)), eng.empty((n,), np.uint8)
t = timed(lambda: eng._call("sylow_hip_g2_scalar_mul_batch", q.ptr, None, kb.ptr, o2.ptr, o2i.ptr, n), 2)
res["g2_scalar_mul_2^20"] = {"per_s": n / t}
# C3: 2^18 pairings
n3 = 1 << 18
gt = eng.empty((48, n))
t = timed(lambda: eng._call("sylow_hip_pairing_batch", p.ptr, None, q.ptr, None, gt.ptr, n), 2)
res["pairing_2^20"] = {"per_s": n / t}
ml = eng.empty((48, n))
t1 = timed(lambda: eng._call("sylow_hip_miller_loop_batch", p.ptr, q.ptr, ml.ptr, n), 2)
t2 = timed(lambda: eng._call("sylow_hip_final_exp_batch", ml.ptr, gt.ptr, n), 2)
res["miller_loop_2^20"] = {"per_s": n / t1}; res["final_exp_2^20"] = {"per_s": n / t2}
# C5: ecPairing 2^16 jobs x k pairs
for k in (2, 4):
    nj = 1 << 16
    off = eng.to_device(np.arange(nj + 1, dtype=np.uint64) * np.uint64(k))
    iso = eng.empty((nj,), np.uint8)
    # pairs laid out with SoA stride = n (>= nj*k): reuse p, q
    t = timed(lambda: eng._call("sylow_hip_multi_pairing_batch", p.ptr, None, q.ptr, None, off.ptr, nj, n, 1, None, iso.ptr), 2)
    res[f"ecpairing_2^16_k{k}"] = {"jobs_per_s": nj / t, "pairs_per_s": nj * k / t}
# Gt * Fr (gt.rs:161-187), 2^18 elements
ng = 1 << 18
gk = eng.empty((4, ng)).upload(rand_scalars_soa(77, ng)); go = eng.empty((48, ng))
gin = eng.empty((48, ng)).upload(np.ascontiguousarray(gt.download()[:, :ng]))
t = timed(lambda: eng._call("sylow_hip_gt_pow_batch", gin.ptr, gk.ptr, go.ptr, ng), 2)
res["gt_pow_2^18"] = {"per_s": ng / t}
# hash / sign / verify 2^18
nv = 1 << 18
rng = np.random.default_rng(3)
dm = eng.to_device(rng.integers(0, 256, size=nv * 32, dtype=np.uint8)); doff = eng.to_device(np.arange(nv + 1, dtype=np.uint64) * np.uint64(32))
h, hi = eng.empty((8, nv)), eng.empty((nv,), np.uint8)
t = timed(lambda: eng._call("sylow_hip_hash_to_g1_batch", dm.ptr, doff.ptr, None, 0, h.ptr, hi.ptr, nv), 2)
res["hash_to_g1_2^18"] = {"per_s": nv / t}
sk = eng.empty((4, nv)).upload(rand_scalars_soa(9, nv))
t = timed(lambda: eng._call("sylow_hip_bls_sign_batch", sk.ptr, dm.ptr, doff.ptr, h.ptr, hi.ptr, nv), 2)
res["bls_sign_2^18"] = {"per_s": nv / t}
g2 = eng.empty((16, nv)).upload(np.repeat(limbs_row(G2).T, nv, axis=1)); pk, pki = eng.empty((16, nv)), eng.empty((nv,), np.uint8)
eng._call("sylow_hip_g2_scalar_mul_batch", g2.ptr, None, sk.ptr, pk.ptr, pki.ptr, nv)
ok = eng.empty((nv,), np.uint8)
t = timed(lambda: eng._call("sylow_hip_bls_verify_batch", pk.ptr, None, dm.ptr, doff.ptr, h.ptr, None, ok.ptr, nv), 2)
res["bls_verify_2^18"] = {"per_s": nv / t, "all_ok": int(ok.download().all())}
# batch-wide glued product (the aggregate batch-verification shape): 2^20 pairs -> one Gt
gt1, is1 = eng.empty((48, 1)), eng.empty((1,), np.uint8)
t = timed(lambda: eng._call("sylow_hip_pairing_product_batch", p.ptr, None, q.ptr, None, n, 0, gt1.ptr, is1.ptr), 2)
res["pairing_product_2^20"] = {"pairs_per_s": n / t}
# C5 end to end: the byte-level ecPairing entry point (decode + curve / subgroup checks + glued pairing), 2^16 jobs of
# two pairs e(a P, Q) e(-a P, Q) == 1; EIP-197 encodings built from device-generated points
nj = 1 << 16
pj, qj = eng.from_device_soa(p)[:nj], eng.from_device_soa(q)[:nj]
def be(col):  # [n,4] LE limbs -> [n,32] big-endian bytes
    return col[:, ::-1].astype(">u8").view(np.uint8).reshape(len(col), 32)
PV = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47
ny = np.array([[(v >> (64 * k)) & ((1 << 64) - 1) for k in range(4)] for v in
               [PV - sum(int(pj[i, 4 + k]) << (64 * k) for k in range(4)) for i in range(nj)]], dtype=np.uint64)
g2b = np.concatenate([be(qj[:, 4:8]), be(qj[:, 0:4]), be(qj[:, 12:16]), be(qj[:, 8:12])], axis=1)       # x.c1 x.c0 y.c1 y.c0
pair_a = np.concatenate([be(pj[:, 0:4]), be(pj[:, 4:8]), g2b], axis=1)
pair_b = np.concatenate([be(pj[:, 0:4]), be(ny), g2b], axis=1)
blob = np.concatenate([pair_a, pair_b], axis=1).reshape(-1)
d_in = eng.to_device(blob); d_off = eng.to_device(np.arange(nj + 1, dtype=np.uint64) * np.uint64(2))
d_res, d_st = eng.empty((nj,), np.uint8), eng.empty((nj,), np.uint8)
t = timed(lambda: eng._call("sylow_hip_evm_ecpairing_batch", d_in.ptr, d_off.ptr, nj, 2 * nj, d_res.ptr, d_st.ptr), 2)
res["evm_ecpairing_bytes_2^16_k2"] = {"jobs_per_s": nj / t, "all_true": int(d_res.download().all()), "all_status_ok": int((d_st.download() == 0).all())}
print(json.dumps(res, indent=1))
