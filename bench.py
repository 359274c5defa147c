#!/usr/bin/env python3
"""bench.py -- BN254 optimal-ate pairings/s (BASELINE.json metric) on N MI355X of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--log2n L]

A "step" is one pass of the hot path (sylow_hip_pairing_batch: Miller loop + final
exponentiation, pairing.rs:870-893) over one batch of 2^L synthetic (P_i, Q_i) pairs PER GPU,
inputs and outputs resident in HBM (nothing crosses PCIe inside the timed region).  Batches
shard trivially (independent pairings), so scaling is weak and the data path has no collective;
the only exchange is the 4-byte MIN(=AND) all-reduce of the aggregate BLS-verify flag, exercised
in the untimed `aux` leg.

One JSON line is printed by rank 0.  `roofline` prices the dominant kernel (k_pairing) against
HBM as the contract asks (576 algorithmic bytes per pairing); because a pairing is ~2x10^4 field
multiplications on 576 bytes the meaningful ceiling is VALU issue rate, reported beside it as
`issue_roofline` (see DESIGN.md "Rooflines").
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0                 # MI355X_MICROARCH.md: HBM3E 8 TB/s
ISSUE_PEAK_GINSTR = 580.0             # measured VOP3 wave-instr/s ceiling, profiles/r01_issue_rate_ubench.txt
PAIRING_BYTES = 576                   # 64 (G1 affine) + 128 (G2 affine) + 384 (Gt)  -- SURVEY.md §8(d)
VERIFY_BYTES = 225                    # pk 128 + sig 64 + 32-byte msg + flag
# rocprofv3 PMC facts about the dominant kernel (plk::k_pairing): bench.py cannot collect PMC counters itself, so the
# per-pairing constants measured by tools/prof_pairing.sh (separate --pmc passes, n = 2^20) are read from the committed summary
# profiles/pmc_current.json and scale with n.  Refresh with tools/prof_pairing.sh whenever the kernel changes.
def _load_pmc():
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_current.json")) as f:
            return json.load(f)
    except OSError:
        return None


PMC = _load_pmc()

G1 = [1, 2]
G2 = [0x1800DEEF121F1E76426A00665E5C4479674322D4F75EDADD46DEBD5CD992F6ED,
      0x198E9393920D483A7260BFB731FB5D25F1AA493335A9E71297E485B7AEF312C2,
      0x12C85EA5DB8C6DEB4AAB71808DCB408FE3D1E7690C43D37B4CE6CC0166FA7DAA,
      0x090689D0585FF075EC9E99AD690C3395BC4B313370B38EF355ACDADCD122975B]
M64 = (1 << 64) - 1


def limbs_row(vals):
    return np.array([[(v >> (64 * k)) & M64 for v in vals for k in range(4)]], dtype=np.uint64)


def rand_scalars_soa(seed, n):
    """n scalars < 2^253 (< p, uniform enough for synthetic points) directly in SoA [4][n]"""
    g = np.random.default_rng(seed)
    a = g.integers(0, 1 << 63, size=(4, n), dtype=np.uint64) * np.uint64(2) + g.integers(0, 2, size=(4, n), dtype=np.uint64)
    a[3] &= np.uint64((1 << 61) - 1)
    return a


def make_points(eng, n, seed):
    """P_i = a_i*G1gen, Q_i = b_i*G2gen generated on the device; returns SoA device arrays."""
    g1 = eng.empty((8, n)).upload(np.repeat(limbs_row(G1).T, n, axis=1))
    g2 = eng.empty((16, n)).upload(np.repeat(limbs_row(G2).T, n, axis=1))
    ka = eng.empty((4, n)).upload(rand_scalars_soa(seed, n))
    kb = eng.empty((4, n)).upload(rand_scalars_soa(seed + 1, n))
    p, pi = eng.empty((8, n)), eng.empty((n,), np.uint8)
    q, qi = eng.empty((16, n)), eng.empty((n,), np.uint8)
    eng._call("sylow_hip_g1_scalar_mul_batch", g1.ptr, None, ka.ptr, p.ptr, pi.ptr, n)
    eng._call("sylow_hip_g2_scalar_mul_batch", g2.ptr, None, kb.ptr, q.ptr, qi.ptr, n)
    eng.sync()
    return p, q, ka, kb


def cpu_baseline(target_seconds=10.0):
    """The C oracle (a port of the reference's pairing(), 256-iteration loops as written) on the
    host cores.  Strictly time-bounded: every worker thread runs small chunks until the deadline."""
    import threading

    from oracle import coracle as C
    C.build()
    C.lib()
    one_p = np.array([[1, 0, 0, 0, 2, 0, 0, 0, 1, 0, 0, 0]], dtype=np.uint64)
    one_q = np.concatenate([limbs_row(G2), np.array([[1, 0, 0, 0, 0, 0, 0, 0]], dtype=np.uint64)], axis=1)
    chunk = 32
    p, q = np.repeat(one_p, chunk, 0), np.repeat(one_q, chunk, 0)
    t0 = time.perf_counter()
    C.pairing(p, q)
    single = chunk / (time.perf_counter() - t0)
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 64))
    counts = [0] * cores
    deadline = time.perf_counter() + target_seconds

    def work(i):
        while time.perf_counter() < deadline:
            C.pairing(p, q)               # ctypes releases the GIL inside the C call
            counts[i] += chunk

    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(i,)) for i in range(cores)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    total = sum(counts)
    return {"value": total / dt, "unit": "pairings/s", "cores": cores, "kind": "port",
            "sample": f"{total} generator pairings e(G1,G2) (benches/pairing.rs shape) in {dt:.1f} s on {cores} threads, "
                      f"C oracle restating sylow pairing() incl. its 256-iteration loops; single-thread {single:.0f}/s; "
                      "the reference's only published figure: 8.183 ms/pairing (hardware unstated)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--log2n", type=int, default=20, help="pairings per GPU per step = 2^L")
    ap.add_argument("--no-aux", action="store_true", help="skip the untimed BLS-verify aux leg")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    args = ap.parse_args()

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    dist = None
    # debugging aid for boxes with fewer GPUs than ranks: SYLOW_BENCH_BACKEND=gloo SYLOW_BENCH_SINGLE_DEVICE=1 runs every
    # rank on cuda:0 with CPU-side collectives, exercising the same barrier / MAX / MIN logic as the RCCL path
    backend = os.environ.get("SYLOW_BENCH_BACKEND", "nccl")
    if os.environ.get("SYLOW_BENCH_SINGLE_DEVICE"):
        local_rank = 0
    coll_dev = "cuda" if backend == "nccl" else "cpu"
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    else:
        torch.cuda.set_device(local_rank)

    import sylow_amd
    stream = torch.cuda.current_stream()
    eng = sylow_amd.Engine(local_rank, stream=stream.cuda_stream or None)

    n = 1 << args.log2n
    p, q, ka, kb = make_points(eng, n, seed=0x53594C4F57 + 3 + 1000 * rank)
    gt = eng.empty((48, n))

    def step():
        eng._call("sylow_hip_pairing_batch", p.ptr, None, q.ptr, None, gt.ptr, n)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for a, b in ev:
        a.record(stream)
        step()
        b.record(stream)
    fence()
    elapsed = time.perf_counter() - t0
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))

    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- untimed aux leg: batched BLS verify + aggregate AND over ranks (RCCL MIN) ----------------
    aux = {}
    if not args.no_aux:
        nv = min(n, 1 << 20)                      # BASELINE.json configs[3]: BLS verifies at batch 2^20
        rng = np.random.default_rng(7 + rank)
        msgs_np = rng.integers(0, 256, size=(nv, 32), dtype=np.uint8)
        off = (np.arange(nv + 1, dtype=np.uint64) * np.uint64(32))
        dm, doff = eng.to_device(msgs_np.reshape(-1)), eng.to_device(off)
        sk = eng.empty((4, nv)).upload(rand_scalars_soa(99 + rank, nv))
        g2 = eng.empty((16, nv)).upload(np.repeat(limbs_row(G2).T, nv, axis=1))
        pk, pki = eng.empty((16, nv)), eng.empty((nv,), np.uint8)
        sig, sigi = eng.empty((8, nv)), eng.empty((nv,), np.uint8)
        ok = eng.empty((nv,), np.uint8)
        eng._call("sylow_hip_g2_scalar_mul_batch", g2.ptr, None, sk.ptr, pk.ptr, pki.ptr, nv)
        eng._call("sylow_hip_bls_sign_batch", sk.ptr, dm.ptr, doff.ptr, sig.ptr, sigi.ptr, nv)
        fence()
        tsg = time.perf_counter()
        eng._call("sylow_hip_bls_sign_batch", sk.ptr, dm.ptr, doff.ptr, sig.ptr, sigi.ptr, nv)
        fence()
        dtsg = time.perf_counter() - tsg
        eng._call("sylow_hip_bls_verify_batch", pk.ptr, None, dm.ptr, doff.ptr, sig.ptr, None, ok.ptr, nv)  # warm
        # torch's first device allocation initialises its allocator (seconds on a cold box): keep it outside the clocks
        flag = torch.ones(1, dtype=torch.int32, device="cuda")
        flag2 = torch.ones(1, dtype=torch.int32, device="cuda")
        fence()
        tv = time.perf_counter()
        eng._call("sylow_hip_bls_verify_batch", pk.ptr, None, dm.ptr, doff.ptr, sig.ptr, None, ok.ptr, nv)
        eng._call("sylow_hip_flags_all", ok.ptr, nv, flag.data_ptr())
        if dist is not None:
            flag = flag.to(coll_dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)      # AND over ranks, 4 bytes over xGMI
        fence()
        dtv = time.perf_counter() - tv
        eng._call("sylow_hip_bls_verify_fused_batch", pk.ptr, None, dm.ptr, doff.ptr, sig.ptr, None, ok.ptr, nv)  # warm (+ builds the G2gen line table)
        fence()
        tf = time.perf_counter()
        eng._call("sylow_hip_bls_verify_fused_batch", pk.ptr, None, dm.ptr, doff.ptr, sig.ptr, None, ok.ptr, nv)
        eng._call("sylow_hip_flags_all", ok.ptr, nv, flag2.data_ptr())
        if dist is not None:
            flag2 = flag2.to(coll_dev)
            dist.all_reduce(flag2, op=dist.ReduceOp.MIN)
        fence()
        dtf = time.perf_counter() - tf
        eng._call("sylow_hip_bls_verify_same_signer_batch", pk.ptr, None, dm.ptr, doff.ptr, sig.ptr, None, ok.ptr, nv)   # warm
        fence()
        ts = time.perf_counter()
        eng._call("sylow_hip_bls_verify_same_signer_batch", pk.ptr, None, dm.ptr, doff.ptr, sig.ptr, None, ok.ptr, nv)
        fence()
        dts = time.perf_counter() - ts
        # aggregate verification (examples/verify_multiple_messages_same_signer.rs:41-60 / threshold_signing.rs:92-121 shape): the 2 na
        # pairs (sig_i, G2gen), (-H(m_i), pk_i) as ONE glued product == identity -> one boolean per rank; hashing is inside the clock
        na = min(nv, 1 << 18)
        sig_h, pk_h = sig.download()[:, :na], pk.download()[:, :na]
        g2_row = limbs_row(G2).T
        qq = eng.empty((16, 2 * na)).upload(np.concatenate([np.repeat(g2_row, na, axis=1), pk_h], axis=1))
        hh, hhi = eng.empty((8, na)), eng.empty((na,), np.uint8)
        dm_a, doff_a = eng.to_device(msgs_np[:na].reshape(-1)), eng.to_device(off[:na + 1])
        gt1, is1 = eng.empty((48, 1)), eng.empty((1,), np.uint8)
        eng._call("sylow_hip_hash_to_g1_batch", dm_a.ptr, doff_a.ptr, None, 0, hh.ptr, hhi.ptr, na)
        hneg = eng.empty((4, na))
        eng._call("sylow_hip_fp_neg_batch", hh.ptr + 4 * na * 8, hneg.ptr, na)          # y rows of the SoA array are contiguous
        pp = eng.empty((8, 2 * na)).upload(np.concatenate([sig_h, np.concatenate([hh.download()[:4], hneg.download()], axis=0)], axis=1))
        eng._call("sylow_hip_pairing_product_batch", pp.ptr, None, qq.ptr, None, 2 * na, 0, gt1.ptr, is1.ptr)  # warm
        fence()
        ta = time.perf_counter()
        eng._call("sylow_hip_hash_to_g1_batch", dm_a.ptr, doff_a.ptr, None, 0, hh.ptr, hhi.ptr, na)
        eng._call("sylow_hip_fp_neg_batch", hh.ptr + 4 * na * 8, hneg.ptr, na)
        eng._call("sylow_hip_pairing_product_batch", pp.ptr, None, qq.ptr, None, 2 * na, 0, gt1.ptr, is1.ptr)
        fence()
        dta = time.perf_counter() - ta
        agg_ok = int(is1.download()[0])
        aux = {"aggregate_verify_sigs_per_s": world * na / dta, "aggregate_all_valid": agg_ok, "aggregate_batch_per_gpu": na,
               "bls_signs_per_s": world * nv / dtsg, "bls_verifies_per_s": world * nv / dtv, "same_signer_shape_checks_per_s": world * nv / dts, "bls_verify_batch_per_gpu": nv,
               "bls_all_valid": int(flag.item()), "bls_verify_algorithmic_GBps": world * nv * VERIFY_BYTES / dtv / 1e9,
               "bls_verifies_per_s_fused": world * nv / dtf, "bls_all_valid_fused": int(flag2.item()),
               "note": "verify = lib.rs:223-236 as written (hash + two full pairings); fused = e(sig,G2gen)*e(-H,pk)==1, one final exponentiation; "
                       "aggregate = all 2n pairs as one glued product == identity (hash + negation + product tree + one final exponentiation), one boolean"}

    if rank == 0:
        total = world * n * args.steps
        value = total / elapsed
        achieved = PAIRING_BYTES * n / (kern_ms * 1e-3) / 1e9
        out = {
            "metric": "BN254 optimal-ate pairings/s", "value": value, "unit": "pairings/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "i32/u32 limbs (9x29-bit carry-free core + 8x32-bit Montgomery, exact integer)", "data": "synthetic",
            "config": {"workload": f"pairing_batch: 2^{args.log2n} independent e(a_i*G1, b_i*G2) per GPU per step "
                                   "(BASELINE.json configs[2] shape at the metric's batch=2^20), affine SoA inputs resident in HBM",
                       "batch_per_gpu": n, "parallelism": f"independent shards x{world}, no data-path collective"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": (PMC["hbm_bytes_per_pairing"] * n) if PMC else None,
                         "traffic_note": ("HBM bytes per launch from rocprofv3 PMC (2*FETCH_SIZE + WRITE_SIZE, " + PMC["source"] + "): "
                                          "what is left beyond the algorithmic bytes is the stack frame of the final exponentiation's "
                                          "straight-line part; the Miller loop and the f^x loops run out of registers") if PMC else None,
                         "kernel": "plk::k_pairing", "kernel_ms": kern_ms, "algorithmic_bytes_per_launch": PAIRING_BYTES * n},
        }
        if PMC:
            ginstr = PMC["valu_instr_per_pairing"] * n / (kern_ms * 1e-3) / 1e9
            out["issue_roofline"] = {"bound": "valu-issue", "achieved": ginstr, "peak": ISSUE_PEAK_GINSTR,
                                     "unit": "G wave-instr/s", "frac": ginstr / ISSUE_PEAK_GINSTR,
                                     "note": "the roof that actually binds a pairing (integer multiply-add chains): SQ_INSTS_VALU per "
                                             "launch / kernel time vs the measured VOP3 issue ceiling (profiles/r01_issue_rate_ubench.txt: "
                                             "~580 G wave-instr/s for VOP3, 477-520 for pure v_mad_*64 streams, ~900 for VOP2 adds)"}
        if aux:
            out["aux"] = aux
        if not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
