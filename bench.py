#!/usr/bin/env python3
"""bench.py -- BN254 optimal-ate pairings/s (BASELINE.json metric) on N MI355X of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--log2n L]

A "step" is one pass of the hot path (sylow_hip_pairing_batch: Miller loop + final exponentiation,
pairing.rs:870-893) over one batch of 2^L synthetic (P_i, Q_i) pairs PER GPU, inputs and outputs
resident in HBM (nothing crosses PCIe inside the timed region).  Batches shard trivially
(independent pairings), so scaling is weak and the data path has no collective; the only exchange
is the 4-byte MIN(=AND) all-reduce of the aggregate BLS-verify flag (sylow_amd/sharding.py),
exercised in the untimed `aux` leg.

The metric has two halves and BOTH run through the same loop (W warm-up steps, then exactly K steps between barrier + synchronize,
HIP events around every step on the launch stream): K steps of 2^L pairings (`value`, `ms_per_step`), then K steps of 2^L BLS
verifications incl. the AND over ranks (`config.bls_verifies_per_s`, `config.bls_verify_ms_per_step`; lib.rs:223-236).

One JSON line is printed by rank 0:
 * `roofline`      the roof that BINDS the dominant kernel, plk::k_pairing: VALU issue (cycle-weighted by instruction class) --
                   `achieved` = ideal issue cycles of one launch / its live HIP-event duration, `peak` = 1024 SIMDs x 2.4 GHz,
                   `frac_clock_free` = the same ratio taken inside one rocprofv3 pass (no clock enters); LIVE in every run (round 6):
                   `sustained_mhz` = the engine clock the K timed launches held (every wavefront adds its s_memtime / s_memrealtime
                   deltas: sylow_hip_clock_probe), `simd_cycles_per_pairing_live` and `frac_at_sustained_clock` from it, and
                   `stagger_gain_*` = plain / skewed launch times of an A/B run through the ABI option; the contract's HBM pricing
                   (576 algorithmic bytes per pairing: ~2x10^4 field multiplications on 576 bytes, tiny by nature) stays beside it as
                   `hbm_*`, `traffic` = HBM bytes per launch from the PMC counters; the verify kernel's figures are `verify_*`;
                   the driver's record keeps a prefix of this object's keys, so what certifies the line comes first;
 * `aux`           the other BLS shapes at batch 2^20 (two-pairing form, same signer, aggregates, strong scaling), the batch-size sweep,
                   the end-to-end (host arrays in, host results out) pipeline, and at N = 1 the other single-GPU
                   configs (C2a Fp mul/add -- the HBM-bound kernels --, C2b G1 scalar-mul, C3 2^18 pairings, C5 byte-level
                   ecPairing) each with its own algorithmic GB/s and fraction of the HBM roof;
 * `cpu_baseline`  the C oracle (a port of the reference) on the host cores -- the all-core loop runs inside the C library on POSIX
                   threads, with nproc / affinity / cgroup quota beside it --, the sign shape, the cargo probe, and an oracle
                   spot check of rows the TIMED launches wrote (`checked` / `mismatches`).
"""
import argparse
import hashlib
import json
import os
import re
import shutil
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0                 # MI355X_MICROARCH.md: HBM3E 8 TB/s
N_SIMD = 1024                         # 256 CUs x 4 SIMDs
PEAK_CLOCK_HZ = 2.4e9                 # MI355X_MICROARCH.md: peak engine clock 2400 MHz
ISSUE_PEAK = N_SIMD * PEAK_CLOCK_HZ   # SIMD issue cycles per second, whole chip
PAIRING_BYTES = 576                   # 64 (G1 affine) + 128 (G2 affine) + 384 (Gt)  -- SURVEY.md §8(d)
VERIFY_BYTES = 225                    # pk 128 + sig 64 + 32-byte msg + flag
FP_OP_BYTES = 96                      # 2 x 32 in + 32 out
G1_MUL_BYTES = 160                    # 64 affine in + 32 scalar + 64 affine out
SEED = 0x53594C4F57                   # "SYLOW" (BASELINE.md §3); + config index

G1 = [1, 2]
G2 = [0x1800DEEF121F1E76426A00665E5C4479674322D4F75EDADD46DEBD5CD992F6ED,
      0x198E9393920D483A7260BFB731FB5D25F1AA493335A9E71297E485B7AEF312C2,
      0x12C85EA5DB8C6DEB4AAB71808DCB408FE3D1E7690C43D37B4CE6CC0166FA7DAA,
      0x090689D0585FF075EC9E99AD690C3395BC4B313370B38EF355ACDADCD122975B]
M64 = (1 << 64) - 1
HOST_ONLY_UNITS = ("pipeline.hip",)     # no device code: editing them does not change any kernel


def csrc_hash():
    """Hash of the kernel sources with comments and blank space removed (a comment edit is not a new build): the committed PMC summary
    (profiles/pmc_current.json) is only valid for the build it was measured on."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "sylow_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".hpp")) and name not in HOST_ONLY_UNITS:
            with open(os.path.join(d, name), "r", encoding="utf-8") as f:
                text = f.read()
            text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
            text = re.sub(r"//[^\n]*", " ", text)
            text = " ".join(text.split())
            h.update(name.encode() + b"\0" + text.encode())
    return h.hexdigest()[:16]


def load_pmc():
    """rocprofv3 PMC facts per configuration (bench.py cannot collect PMC counters itself): per-unit constants measured by
    tools/prof_configs.sh (separate --pmc passes, bench.py's sizes) in profiles/pmc_current.json.  Returns (pmc or None, stale?)."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_current.json")) as f:
            pmc = json.load(f)
    except OSError:
        return None, False
    return pmc, pmc.get("csrc_hash") != csrc_hash()


def issue_fields(pmc_cfg):
    """The issue-roof entry of one configuration: cycle-weighted VALU issue cycles (quarter-rate 64-bit class x 4 + other VALU x 2) over
    the SIMD cycles the GPU spent on the configuration's kernels -- numerator and denominator from the SAME rocprofv3 pass, so no clock
    enters and the fraction is a property of the build, not of the box."""
    if not pmc_cfg or pmc_cfg.get("issue_frac") is None:
        return {}
    return {"issue_frac": pmc_cfg["issue_frac"], "issue_frac_vs_measured_issue_rates": pmc_cfg.get("issue_frac_vs_measured_issue_rates"),
            "valu_instr_per_unit": pmc_cfg.get("valu_instr_per_unit"), "int64_class_frac": pmc_cfg.get("int64_class_frac"),
            "issue_cycles_ideal_per_unit": pmc_cfg.get("issue_cycles_ideal_per_unit"), "simd_cycles_per_unit": pmc_cfg.get("simd_cycles_per_unit"),
            "hbm_bytes_per_unit_measured": pmc_cfg.get("hbm_bytes_per_unit"), "profile_ms_per_launch": pmc_cfg.get("ms_per_launch", pmc_cfg.get("ms_per_launch_kernel_sum"))}


def limbs_row(vals):
    return np.array([[(v >> (64 * k)) & M64 for v in vals for k in range(4)]], dtype=np.uint64)


def make_points(eng, n, seed):
    """P_i = a_i*G1gen, Q_i = b_i*G2gen generated on the device from the xoshiro stream; returns SoA device arrays."""
    ka = eng.empty((4, n)).upload(eng.xoshiro_fp_soa(seed, n))
    kb = eng.empty((4, n)).upload(eng.xoshiro_fp_soa(seed + (1 << 32), n))
    p, pi = eng.empty((8, n)), eng.empty((n,), np.uint8)
    q, qi = eng.empty((16, n)), eng.empty((n,), np.uint8)
    eng._call("sylow_hip_g1_generator_mul_batch", ka.ptr, p.ptr, pi.ptr, n)          # fixed-base tables of the generators
    eng._call("sylow_hip_g2_generator_mul_batch", kb.ptr, q.ptr, qi.ptr, n)
    eng.sync()
    return p, q, ka, kb


# ------------------------------------------------------------------------------------------------- CPU leg (rank 0, N = 1)
def cargo_probe():
    """sylow's own benches can only be timed if the box has cargo AND an offline registry."""
    exe = shutil.which("cargo")
    if not exe:
        return {"cargo": None, "case": "no cargo on this box: sylow's own criterion benches cannot be built; baseline = C port of the reference"}
    try:
        ver = subprocess.run([exe, "--version"], capture_output=True, text=True, timeout=20).stdout.strip()
    except Exception as e:  # noqa: BLE001
        ver = f"error: {e}"
    return {"cargo": ver, "case": "cargo present but the reference sources and its un-vendored crates (crypto-bigint 0.6.0-rc.3, sha3 0.11.0-pre.4) "
                                  "do not travel to this box: baseline = C port of the reference"}


def host_cpu_facts():
    """What bounds the all-core leg on this box: logical CPUs, the affinity mask, the cgroup CPU quota (cpu.max / cfs_quota)."""
    nproc = os.cpu_count() or 1
    try:
        aff = len(os.sched_getaffinity(0))
    except AttributeError:
        aff = nproc
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:                 # cgroup v2: "<quota|max> <period>"
            q, per = f.read().split()
            quota = None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                q, per = float(f.read()), float(g.read())
                quota = None if q <= 0 else q / per
        except (OSError, ValueError):
            quota = None
    usable = aff if quota is None else max(1, min(aff, int(quota + 0.999)))
    return {"nproc": nproc, "affinity_cores": aff, "cgroup_cpu_quota_cores": quota, "usable_cores": max(1, min(usable, 256))}


def cpu_baseline(check=None, target_seconds=10.0):
    """The C oracle (a port of the reference's pairing(), 256-iteration loops as written) on the host cores: single-thread
    generator pairing (benches/pairing.rs:5-10) and sign (benches/sig.rs:10-21) shapes, then all cores for `target_seconds`.
    `check` = (p_xy, q_xy, gt) host AoS rows the TIMED launches produced: recomputed here and compared bit for bit."""
    from oracle import coracle as C
    C.build()
    C.lib()
    one_p = np.array([[1, 0, 0, 0, 2, 0, 0, 0, 1, 0, 0, 0]], dtype=np.uint64)
    one_q = np.concatenate([limbs_row(G2), np.array([[1, 0, 0, 0, 0, 0, 0, 0]], dtype=np.uint64)], axis=1)
    chunk = 32
    p, q = np.repeat(one_p, chunk, 0), np.repeat(one_q, chunk, 0)
    C.pairing(p[:2], q[:2])
    c1, d1 = C.bench_pairing_threads(one_p, one_q, 1, 2.0)            # the same C loop as the all-core leg, one thread
    single = float(c1.sum()) / d1
    # sign shape: sk = first PRNG draw, msg = 20_i32.to_be_bytes() (benches/sig.rs:7), n >= 100
    ns = 128
    from sylow_amd import _lib                                  # host-side PRNG only (no GPU involved)
    first = np.empty((4, 1), dtype=np.uint64)
    _lib.check(_lib.load().sylow_hip_host_xoshiro_fp(SEED + 1, first.ctypes.data, 1, 1), "xoshiro")
    sk = np.repeat(first.T.copy(), ns, 0)
    msgs = [(20).to_bytes(4, "big")] * ns
    C.sign(sk[:2], msgs[:2])
    t0 = time.perf_counter()
    C.sign(sk, msgs)
    sign_single = ns / (time.perf_counter() - t0)
    # verify shape (lib.rs:223-236: hash + two full pairings + compare) and BASELINE.json configs[0] as written, "single BLS
    # sign+verify": keygen outside the clock (benches/sig.rs generates the key pair in its setup), n = 32, one thread
    nv = 32
    g2_proj = np.concatenate([limbs_row(G2), np.array([[1, 0, 0, 0, 0, 0, 0, 0]], dtype=np.uint64)], axis=1)
    pk = C.g2_scalar_mul(np.repeat(g2_proj, nv, 0), sk[:nv])
    sig = C.sign(sk[:nv], msgs[:nv])
    C.verify(pk[:1], msgs[:1], sig[:1])
    t0 = time.perf_counter()
    ok = C.verify(pk, msgs[:nv], sig)
    verify_single = nv / (time.perf_counter() - t0)
    t0 = time.perf_counter()
    for i in range(8):
        C.verify(pk[i:i + 1], msgs[i:i + 1], C.sign(sk[i:i + 1], msgs[i:i + 1]))
    sign_verify_ms = (time.perf_counter() - t0) / 8 * 1e3
    host = host_cpu_facts()
    cores = host["usable_cores"]
    # the all-core loop runs INSIDE the C library (POSIX threads, oracle_bench_pairing_threads): no Python, no GIL hand-over in the loop
    counts, dt = C.bench_pairing_threads(one_p, one_q, cores, target_seconds)
    total = int(counts.sum())
    scaling = (total / dt) / single if single else None
    out = {"value": total / dt, "unit": "pairings/s", "cores": cores, "kind": "port",
           "sample": f"{total} generator pairings e(G1gen, G2gen) (benches/pairing.rs:5-10 shape) on {cores} POSIX threads for {dt:.1f} s "
                     f"(loop in C: oracle_bench_pairing_threads); C restatement of the reference's formulas and loop structure (oracle/sylow_oracle.c), not sylow itself",
           "single_thread_pairings_per_s": single, "all_core_over_single_thread": scaling,
           "nproc": host["nproc"], "affinity_cores": host["affinity_cores"], "cgroup_cpu_quota_cores": host["cgroup_cpu_quota_cores"],
           "per_thread_min_max": [int(counts.min()), int(counts.max())],
           "checked": None, "mismatches": None,                 # filled below (kept among the first keys: the driver's record keeps a prefix)
           "single_thread_signs_per_s": sign_single,
           "sign_sample": f"{ns} x sign(sk, 20_i32.to_be_bytes()) (benches/sig.rs:10-21 shape), one thread",
           "single_thread_verifies_per_s": verify_single, "verify_all_true": int(bool(np.all(ok))),
           "verify_sample": f"{nv} x verify(pk, 20_i32.to_be_bytes(), sig) (lib.rs:223-236: hash + two pairings), one thread",
           "C1_single_sign_plus_verify_ms": sign_verify_ms,
           "C1_sample": "BASELINE.json configs[0]: 8 x (sign then verify) of the bench message, one element at a time, one thread (CPU plumbing only)",
           "reference_published": {"pairing_ms": 8.183, "sign_us": 954, "source": "sylow_devguide.pdf p.62, hardware unstated"},
           "rust_toolchain": cargo_probe()}
    if check is not None:
        p_xy, q_xy, gt = check
        m = p_xy.shape[0]
        one = np.zeros((m, 4), dtype=np.uint64); one[:, 0] = 1
        exp = C.pairing(np.concatenate([p_xy, one], axis=1), np.concatenate([q_xy, one, np.zeros((m, 4), dtype=np.uint64)], axis=1))
        out["checked"] = int(m)
        out["mismatches"] = int(m - int(np.all(exp == gt, axis=1).sum()))
        out["check_note"] = "rows of the Gt array written by the TIMED launches, PRNG-chosen indices, recomputed by the oracle"
    return out


# ------------------------------------------------------------------------------------------------- timing helpers
def hip_timed(torch, stream, fn, reps):
    """Average seconds per call of `fn` (a launch sequence on `stream`), HIP events on that stream."""
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(stream)
    for _ in range(reps):
        fn()
    b.record(stream)
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3


def single_gpu_configs(eng, torch, stream, p, q, ka, n, pmc_cfgs=None):
    """BASELINE.json configs[1], [2], [4] on one GPU, device-resident, HIP-event timed (units/s, algorithmic GB/s, fraction of
    the 8 TB/s HBM roof).  Fp mul / add are the HBM-bound kernels of the path (96 algorithmic bytes per element)."""
    res = {}
    for log2n, reps in ((20, 200), (24, 20)):           # 2^20 = 96 MB per launch (~20 us): 200 back-to-back launches
        m = 1 << log2n
        a = eng.empty((4, m)).upload(eng.xoshiro_fp_soa(SEED + 2, m))
        b = eng.empty((4, m)).upload(eng.xoshiro_fp_soa(SEED + 2 + (1 << 32), m))
        o = eng.empty((4, m))
        for name in ("mul", "add"):
            t = hip_timed(torch, stream, lambda: eng._call(f"sylow_hip_fp_{name}_batch", a.ptr, b.ptr, o.ptr, m), reps)
            gbs = FP_OP_BYTES * m / t / 1e9
            res[f"C2a_fp_{name}_2^{log2n}"] = {"units_per_s": m / t, "algorithmic_GBps": gbs, "frac_of_hbm": gbs / HBM_PEAK_GBS, "launch_us": t * 1e6,
                                              "kernel": f"k_fp_binop<{'2' if name == 'mul' else '0'},0>"}
        del a, b, o
    o, oi = eng.empty((8, n)), eng.empty((n,), np.uint8)
    t = hip_timed(torch, stream, lambda: eng._call("sylow_hip_g1_scalar_mul_batch", p.ptr, None, ka.ptr, o.ptr, oi.ptr, n), 3)
    gbs = G1_MUL_BYTES * n / t / 1e9
    res[f"C2b_g1_scalar_mul_2^{n.bit_length() - 1}"] = {"units_per_s": n / t, "algorithmic_GBps": gbs, "frac_of_hbm": gbs / HBM_PEAK_GBS, "kernel": "k_g1_scalar_mul"}
    del o, oi
    # G2 scalar-mul (352 B per unit -- 128 affine in + 32 scalar + 192 projective out): the keygen shape pk = sk * Q on
    # r-torsion points (4-way endomorphism split), and the product that is exact on the whole twist
    o2, o2i = eng.empty((16, n)), eng.empty((n,), np.uint8)
    t = hip_timed(torch, stream, lambda: eng._call("sylow_hip_g2_scalar_mul_subgroup_batch", q.ptr, None, ka.ptr, o2.ptr, o2i.ptr, n), 2)
    tg = hip_timed(torch, stream, lambda: eng._call("sylow_hip_g2_scalar_mul_batch", q.ptr, None, ka.ptr, o2.ptr, o2i.ptr, n), 1)
    tk = hip_timed(torch, stream, lambda: eng._call("sylow_hip_g2_generator_mul_batch", ka.ptr, o2.ptr, o2i.ptr, n), 2)
    gbs = 352 * n / t / 1e9
    res[f"C2c_g2_scalar_mul_2^{n.bit_length() - 1}"] = {"units_per_s": n / t, "any_twist_point_units_per_s": n / tg, "generator_units_per_s": n / tk,
                                                      "algorithmic_GBps": gbs, "frac_of_hbm": gbs / HBM_PEAK_GBS,
                                                      "kernel": "plk::k_g2_scalar_mul_gls (r-torsion inputs) / plk::k_g2_scalar_mul / plk::k_g2_generator_mul (keygen: fixed-base table)"}
    del o2, o2i
    n3 = min(n, 1 << 18)
    p3 = eng.empty((8, n3)).upload(np.ascontiguousarray(p.download()[:, :n3]))
    q3 = eng.empty((16, n3)).upload(np.ascontiguousarray(q.download()[:, :n3]))
    gt3 = eng.empty((48, n3))
    t = hip_timed(torch, stream, lambda: eng._call("sylow_hip_pairing_batch", p3.ptr, None, q3.ptr, None, gt3.ptr, n3), 3)
    gbs = PAIRING_BYTES * n3 / t / 1e9
    res[f"C3_pairing_2^{n3.bit_length() - 1}"] = {"units_per_s": n3 / t, "algorithmic_GBps": gbs, "frac_of_hbm": gbs / HBM_PEAK_GBS, "kernel": "plk::k_pairing"}
    # C5: byte-level ecPairing (EIP-197 192-byte pairs: decode + curve / subgroup checks + glued pairing), 2^16 jobs x k pairs.
    # job j: e(P, Q) e(-P, Q) [k = 2] or e(P, Q) e(-P, Q) e(P', Q') e(-P', Q') [k = 4] -> true; every other job has its last G1
    # point replaced by a different one -> false.
    nj = min(1 << 16, n // 4)
    npts = 2 * nj
    ny = eng.empty((4, n3))
    eng._call("sylow_hip_fp_neg_batch", p3.ptr + 4 * n3 * 8, ny.ptr, n3)                      # y rows of the SoA array are contiguous
    pneg = eng.empty((8, n3)).upload(np.concatenate([p3.download()[:4], ny.download()], axis=0))
    b1, b1n, b2 = eng.empty((n3 * 64,), np.uint8), eng.empty((n3 * 64,), np.uint8), eng.empty((n3 * 128,), np.uint8)
    eng._call("sylow_hip_g1_to_be_bytes_batch", p3.ptr, None, b1.ptr, n3)
    eng._call("sylow_hip_g1_to_be_bytes_batch", pneg.ptr, None, b1n.ptr, n3)
    eng._call("sylow_hip_g2_to_be_bytes_batch", q3.ptr, None, b2.ptr, n3)
    g1b, g1nb, g2b = (x.download().reshape(n3, -1)[:npts] for x in (b1, b1n, b2))
    pos, neg = np.concatenate([g1b, g2b], axis=1), np.concatenate([g1nb, g2b], axis=1)        # [npts][192] each
    for k in (2, 4):
        njk = nj                               # 2^16 jobs for both shapes (BASELINE.json configs[4]); a k = 4 job uses two base points
        if k == 2:
            jobs = np.concatenate([pos[:njk], neg[:njk]], axis=1)
        else:
            jobs = np.concatenate([pos[0:2 * njk:2], neg[0:2 * njk:2], pos[1:2 * njk:2], neg[1:2 * njk:2]], axis=1)
        jobs = jobs.copy()
        spoil = np.arange(njk) % 2 == 1
        jobs[spoil, (k - 1) * 192:(k - 1) * 192 + 64] = g1b[(np.arange(njk)[spoil] + 7) % npts]
        d_in = eng.to_device(jobs.reshape(-1))
        d_off = eng.to_device(np.arange(njk + 1, dtype=np.uint64) * np.uint64(k))
        d_res, d_st = eng.empty((njk,), np.uint8), eng.empty((njk,), np.uint8)
        t = hip_timed(torch, stream, lambda: eng._call("sylow_hip_evm_ecpairing_batch", d_in.ptr, d_off.ptr, njk, k * njk, d_res.ptr, d_st.ptr), 3)
        gbs = (192 * k + 1) * njk / t / 1e9
        r = d_res.download()
        res[f"C5_ecpairing_bytes_2^{int(np.log2(njk))}_k{k}"] = {
            "units_per_s": njk / t, "pairs_per_s": k * njk / t, "algorithmic_GBps": gbs, "frac_of_hbm": gbs / HBM_PEAK_GBS,
            "pattern_ok": int(np.array_equal(r, (~spoil).astype(np.uint8)) and not d_st.download().any()),
            "kernel": "k_evm_decode_pairs + plk::k_pair_lines + plk::k_glued_from_tables + plk::k_final_exp_jobs"}
    for name, e in res.items():                      # both roofs for every configuration
        e.update(issue_fields((pmc_cfgs or {}).get(name)))
    # C1 (BASELINE.json configs[0], the reference's own bench shapes benches/pairing.rs:5-10, benches/sig.rs:10-21) as SINGLE device calls:
    # latency, not throughput -- one Miller loop / final exponentiation spread over a wavefront (docs/DESIGN_LOG.md R5-8.4 .. R5-8.6)
    p1 = eng.empty((8, 1)).upload(np.ascontiguousarray(p.download()[:, :1]))
    q1 = eng.empty((16, 1)).upload(np.ascontiguousarray(q.download()[:, :1]))
    gt1 = eng.empty((48, 1))
    msg = np.frombuffer((20).to_bytes(4, "big"), dtype=np.uint8).copy()
    dm, doff = eng.to_device(msg), eng.to_device(np.array([0, 4], dtype=np.uint64))
    sk1 = eng.empty((4, 1)).upload(eng.xoshiro_fp_soa(SEED, 1))
    g2 = eng.empty((16, 1)).upload(limbs_row(G2).T.copy())
    pk1, pk1i, sig1, sig1i, ok1 = eng.empty((16, 1)), eng.empty((1,), np.uint8), eng.empty((8, 1)), eng.empty((1,), np.uint8), eng.empty((1,), np.uint8)
    eng._call("sylow_hip_g2_scalar_mul_subgroup_batch", g2.ptr, None, sk1.ptr, pk1.ptr, pk1i.ptr, 1)
    t_pair = hip_timed(torch, stream, lambda: eng._call("sylow_hip_pairing_batch", p1.ptr, None, q1.ptr, None, gt1.ptr, 1), 5)
    t_sign = hip_timed(torch, stream, lambda: eng._call("sylow_hip_bls_sign_batch", sk1.ptr, dm.ptr, doff.ptr, sig1.ptr, sig1i.ptr, 1), 5)
    t_ver = hip_timed(torch, stream, lambda: eng._call("sylow_hip_bls_verify_batch", pk1.ptr, None, dm.ptr, doff.ptr, sig1.ptr, None, ok1.ptr, 1), 5)
    res["C1_single_calls"] = {"pairing_ms": t_pair * 1e3, "sign_ms": t_sign * 1e3, "verify_ms": t_ver * 1e3, "verify_ok": int(ok1.download()[0]),
                              "note": "one element per call, HIP-event timed: latency of the single-wavefront routes"}
    return res

def size_sweep(eng, torch, stream, p_h, q_h, pk_h, sig_h, dm, off, n, sk_h=None):
    """Per-element time against the batch size for the two headline operations (HIP events on the launch stream, device-resident SoA
    inputs re-packed to each size's stride outside the clock): where single ecPairing calls, Groth16-size batches and the metric's
    2^20 sit on the same curve (examples/reth_bn128.rs:156-217)."""
    rows = []
    # 8192 / 16384: the lane-quad route (plk_quad.hip, round 6); 32768: one wavefront per SIMD of lane pairs; 65536: one full round
    for m in (1, 64, 1 << 10, 1 << 11, 1 << 12, 1 << 13, 1 << 14, 1 << 15, 1 << 16, 1 << 18, 1 << 20):
        if m > n:
            break
        reps = 20 if m <= (1 << 12) else 5 if m <= (1 << 16) else 3
        pm, qm = eng.empty((8, m)).upload(np.ascontiguousarray(p_h[:, :m])), eng.empty((16, m)).upload(np.ascontiguousarray(q_h[:, :m]))
        pkm, sgm = eng.empty((16, m)).upload(np.ascontiguousarray(pk_h[:, :m])), eng.empty((8, m)).upload(np.ascontiguousarray(sig_h[:, :m]))
        gtm, okm, offm = eng.empty((48, m)), eng.empty((m,), np.uint8), eng.to_device(off[:m + 1])
        tp = hip_timed(torch, stream, lambda: eng._call("sylow_hip_pairing_batch", pm.ptr, None, qm.ptr, None, gtm.ptr, m), reps)
        tv = hip_timed(torch, stream, lambda: eng._call("sylow_hip_bls_verify_batch", pkm.ptr, None, dm.ptr, offm.ptr, sgm.ptr, None, okm.ptr, m), reps)
        rows.append({"n": m, "pairing_ms": tp * 1e3, "pairing_us_per_element": tp * 1e6 / m, "pairings_per_s": m / tp,
                     "verify_ms": tv * 1e3, "verify_us_per_element": tv * 1e6 / m, "verifies_per_s": m / tv, "verify_all_ok": int(okm.download().all()), "reps": reps})
        if sk_h is not None:                              # sign (lib.rs:179-187): n <= 16384 on eight lanes per signature (sign_wide.hip), one lane above
            skm, so, soi = eng.empty((4, m)).upload(np.ascontiguousarray(sk_h[:, :m])), eng.empty((8, m)), eng.empty((m,), np.uint8)
            ts = hip_timed(torch, stream, lambda: eng._call("sylow_hip_bls_sign_batch", skm.ptr, dm.ptr, offm.ptr, so.ptr, soi.ptr, m), reps)
            rows[-1].update({"sign_ms": ts * 1e3, "sign_us_per_element": ts * 1e6 / m, "signs_per_s": m / ts,
                             "sign_equals_batch_signature": int(np.array_equal(so.download(), sig_h[:, :m]))})
            del skm, so, soi
        del pm, qm, pkm, sgm, gtm, okm, offm
    mono = lambda key: int(all(rows[i + 1][key] <= rows[i][key] * 1.02 for i in range(len(rows) - 1)))
    return {"rows": rows, "pairing_us_per_element_monotone": mono("pairing_us_per_element"), "verify_us_per_element_monotone": mono("verify_us_per_element"),
            **({"sign_us_per_element_monotone": mono("sign_us_per_element")} if sk_h is not None else {}),
            "note": "per-element time must not rise with n (2 % tolerance): one call per row, mean of `reps` back-to-back calls"}


def end_to_end(eng, p_h, q_h, gt_h, pk_h, sig_h, msgs_np, off, dev_pairings_per_s, dev_verifies_per_s):
    """HOST arrays in, HOST results out (the value-typed API of pairing.rs:870-893 / lib.rs:223-236): sylow_hip_pairing_host and
    sylow_hip_bls_verify_host cut the batch into chunks that alternate between two streams so that the copies run beside the kernels.
    Element-major ("array of structs") host arrays, pageable and page-locked; wall clock around the synchronous call; results compared
    with what the device-resident launches wrote."""
    n = p_h.shape[1]
    lib = eng.lib
    from sylow_amd import _lib
    res = {}
    src = {"p": np.ascontiguousarray(p_h.T), "q": np.ascontiguousarray(q_h.T), "pk": np.ascontiguousarray(pk_h.T), "sig": np.ascontiguousarray(sig_h.T)}
    blob = np.ascontiguousarray(msgs_np.reshape(-1))
    gt_ref = np.ascontiguousarray(gt_h.T)
    for kind in ("pageable", "pinned"):
        if kind == "pinned":
            h = {k: eng.pinned_empty(v.shape) for k, v in src.items()}
            for k, v in src.items():
                h[k][:] = v
            gt_out, ok_out = eng.pinned_empty((n, 48)), eng.pinned_empty((n,), np.uint8)
        else:
            h = src
            gt_out, ok_out = np.empty((n, 48), dtype=np.uint64), np.empty((n,), dtype=np.uint8)
        call_p = lambda: _lib.check(lib.sylow_hip_pairing_host(h["p"].ctypes.data, None, h["q"].ctypes.data, None, gt_out.ctypes.data, n, 0), "pairing_host")
        call_v = lambda: _lib.check(lib.sylow_hip_bls_verify_host(h["pk"].ctypes.data, None, blob.ctypes.data, off.ctypes.data, h["sig"].ctypes.data, None,
                                                                   ok_out.ctypes.data, n, 0), "bls_verify_host")
        for name, call in (("pairing", call_p), ("verify", call_v)):
            call()
            ts = []
            for _ in range(3):
                t0 = time.perf_counter()
                call()
                ts.append(time.perf_counter() - t0)
            dev = dev_pairings_per_s if name == "pairing" else dev_verifies_per_s
            res[f"{name}_{kind}"] = {"per_s": n / float(np.mean(ts)), "ms": float(np.mean(ts)) * 1e3, "ms_min": min(ts) * 1e3, "ms_max": max(ts) * 1e3,
                                     "frac_of_device_resident_rate": n / float(np.mean(ts)) / dev}
        res[f"pairing_{kind}"]["bit_equal_to_device_resident"] = int(np.array_equal(gt_out, gt_ref))
        res[f"verify_{kind}"]["all_ok"] = int(ok_out.all())
        del gt_out, ok_out
    res["note"] = ("2^%d elements, chunks of 2^16 on two streams; bytes per pairing 192 up + 384 down, per verify 192 + message up, 1 down; "
                   "device-resident rates are this run's headline figures" % (n.bit_length() - 1))
    return res


def spawn_ranks(n_ranks):
    """`python bench.py --gpus N` without a launcher: start the N ranks (one per GPU) as fresh child processes of
    torch.distributed.run with this very command line, stdout / stderr inherited (rank 0 prints the one JSON line), and return
    the launcher's exit code.  The parent holds no GPU state — nothing that initialises HIP has been imported — and does not exec."""
    import socket
    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = str(s.getsockname()[1])
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n_ranks) // n_ranks)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}",
           "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--log2n", type=int, default=20, help="pairings per GPU per step = 2^L")
    ap.add_argument("--no-aux", action="store_true", help="skip the untimed BLS-verify / other-config aux leg")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg (and the oracle spot check)")
    ap.add_argument("--plant-bad", type=int, default=-1, metavar="RANK", help="corrupt one signature on that rank (aux leg): the global AND must read 0")
    ap.add_argument("--force-dist", action="store_true", help="initialise the process group (and the native ncclComm_t) even for one rank")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # bare `python bench.py --gpus N`: this process has not touched the GPU (no torch import yet) and never will — it starts the
        # N ranks as fresh children through torch.distributed.run, lets rank 0's JSON line through on stdout, and exits with their code
        raise SystemExit(spawn_ranks(args.gpus))

    import torch

    from sylow_amd import sharding

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU (or run the bare command, which spawns them)")
    dist = None
    # debugging aid for boxes with fewer GPUs than ranks: SYLOW_BENCH_BACKEND=gloo SYLOW_BENCH_SINGLE_DEVICE=1 runs every
    # rank on cuda:0 with host-side collectives, exercising the same barrier / MAX / MIN logic as the RCCL path
    backend = os.environ.get("SYLOW_BENCH_BACKEND", "nccl")
    if os.environ.get("SYLOW_BENCH_SINGLE_DEVICE"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        from sylow_amd.rccl import quiet_init_env
        quiet_init_env()                                   # one node: no MSCCL stores to parse, loopback bootstrap (defaults only)
        if world == 1:
            os.environ.setdefault("MASTER_PORT", "29533"); os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    import sylow_amd
    stream = torch.cuda.current_stream()
    eng = sylow_amd.Engine(local_rank, stream=stream.cuda_stream or None)

    n = 1 << args.log2n
    p, q, ka, kb = make_points(eng, n, seed=SEED + 3 + 1000 * rank)          # config index 3 = the pairing batch
    gt = eng.empty((48, n))

    def step():
        eng._call("sylow_hip_pairing_batch", p.ptr, None, q.ptr, None, gt.ptr, n)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # live clock probe (sylow_hip_clock_probe): every wavefront of the metric's kernels adds its shader-clock and constant-rate tick deltas
    clk_acc = eng.empty((256,))
    clk_zero = np.zeros(256, dtype=np.uint64)
    wall_khz = eng.wall_clock_khz()

    def probe_reset():
        clk_acc.upload(clk_zero)
        eng.clock_probe(clk_acc)

    def probe_read():
        """-> {sustained_mhz, wave_ticks, waves, longest_wave_ms} of the launches since probe_reset(); switches the probe off"""
        torch.cuda.synchronize()
        eng.clock_probe(None)
        mhz, ticks, waves, longest = eng.clock_probe_summary(clk_acc.download(), wall_khz)
        return {"sustained_mhz": mhz, "wave_ticks": ticks, "waves": waves, "longest_wave_ms": longest}

    def timed_loop(fn):
        """The contract's loop: W untimed warm-up steps, then EXACTLY K steps bracketed by barrier + synchronize on both sides; wall time is
        the MAX over ranks; every step also sits between two HIP events on the launch stream (this rank's device time per step).  The clock
        probe is armed after the warm-up and before the opening fence: it covers exactly the K timed steps."""
        for _ in range(args.warmup):
            fn()
        probe_reset()
        fence()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
        t0 = time.perf_counter()
        for a, b in ev:
            a.record(stream)
            fn()
            b.record(stream)
        fence()
        dt = time.perf_counter() - t0
        ms = [a.elapsed_time(b) for a, b in ev]
        return sharding.max_over_ranks(dt, dist), float(np.mean(ms)), float(np.min(ms)), float(np.max(ms))

    elapsed, kern_ms, kern_ms_min, kern_ms_max = timed_loop(step)
    probe_p = probe_read()

    # rows of what the timed launches wrote, for the oracle spot check in the CPU leg
    check = None
    if rank == 0 and world == 1 and not args.no_cpu:
        idx = np.sort(np.random.default_rng(SEED).choice(n, size=min(16, n), replace=False))
        check = tuple(np.ascontiguousarray(d.download()[:, idx].T) for d in (p, q, gt))

    pmc, pmc_stale = load_pmc()
    pmc_cfgs = pmc.get("configs", {}) if (pmc is not None and not pmc_stale) else {}

    # ---- native RCCL communicator for the C ABI's aggregate entry points (one per rank; backend nccl only) -------------
    comm, comm_err = None, None
    if dist is not None and backend == "nccl":
        from sylow_amd.rccl import NativeComm
        comm, comm_err = NativeComm.from_process_group(dist)
    rccl_ranks = comm.ranks if comm is not None else None
    comm_ptr = comm.value if comm is not None else None

    spreads = {}

    def timed_ranks(fn, reps=3, name=None):
        """(seconds per call with the slowest rank's clock -- barrier + synchronize on both sides --, this rank's mean HIP-event ms per call);
        every call sits between its own pair of events: [min, max] ms go to aux.kernel_ms_spread under `name`."""
        fn()
        fence()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        t0 = time.perf_counter()
        for a, b in ev:
            a.record(stream)
            fn()
            b.record(stream)
        fence()
        dt = (time.perf_counter() - t0) / reps
        ms = [a.elapsed_time(b) for a, b in ev]
        if name:
            spreads[name] = [float(np.min(ms)), float(np.max(ms)), reps]
        return sharding.max_over_ranks(dt, dist), float(np.mean(ms))

    # ---- the metric's second half: BLS verifies at the same batch, the same W + K loop (BASELINE.json configs[3], lib.rs:223-236) ----
    nv = n
    rng = np.random.default_rng(7 + rank)
    msgs_np = rng.integers(0, 256, size=(nv, 32), dtype=np.uint8)
    off = (np.arange(nv + 1, dtype=np.uint64) * np.uint64(32))
    dm, doff = eng.to_device(msgs_np.reshape(-1)), eng.to_device(off)
    sk = eng.empty((4, nv)).upload(eng.xoshiro_fp_soa(SEED + 4 + 1000 * rank, nv))
    g2 = eng.empty((16, nv)).upload(np.repeat(limbs_row(G2).T, nv, axis=1))
    pk, pki = eng.empty((16, nv)), eng.empty((nv,), np.uint8)
    sig, sigi = eng.empty((8, nv)), eng.empty((nv,), np.uint8)
    ok = eng.empty((nv,), np.uint8)
    eng._call("sylow_hip_g2_scalar_mul_subgroup_batch", g2.ptr, None, sk.ptr, pk.ptr, pki.ptr, nv)
    dtsg, sign_ms = timed_ranks(lambda: eng._call("sylow_hip_bls_sign_batch", sk.ptr, dm.ptr, doff.ptr, sig.ptr, sigi.ptr, nv), 3, "bls_sign")
    if args.plant_bad == rank:                # sig_j <- sig_{j+1}: a valid point, the wrong signature
        s_h = sig.download()
        j = (nv // world) // 3               # inside this rank's block of the strong-scaling leg too
        s_h[:, j] = s_h[:, (j + 1) % nv]
        sig.upload(s_h)
    # torch's first device allocation initialises its allocator (seconds on a cold box): keep it outside the clocks
    flag_dev = torch.ones(1, dtype=torch.int32, device="cuda")          # the device word the native calls write; results are kept apart
    flag = flag2 = flag3 = flag_dev                                      # (a host-side backend returns a host copy)

    def and_over_ranks(okbuf, m, out):
        """AND of this rank's flags AND-ed over all ranks: the native entry point (device-side AND + 4-byte ncclAllReduce(min) on the
        rank's own ncclComm_t) when a communicator exists, else the device-side AND + the process group's MIN."""
        if comm is not None:
            eng._call("sylow_hip_all_valid", okbuf.ptr, m, comm_ptr, out.data_ptr())
            return out
        eng._call("sylow_hip_flags_all", okbuf.ptr, m, out.data_ptr())
        return sharding.and_reduce_(out, dist)

    def verify_weak():
        nonlocal flag
        eng._call("sylow_hip_bls_verify_batch", pk.ptr, None, dm.ptr, doff.ptr, sig.ptr, None, ok.ptr, nv)
        flag = and_over_ranks(ok, nv, flag_dev)      # AND over ranks: 4 bytes over xGMI (RCCL MIN)

    v_elapsed, verify_ms, verify_ms_min, verify_ms_max = timed_loop(verify_weak)
    probe_v = probe_read()
    all_valid_weak = int(flag.item())
    n_bad_fused = int(nv - int(ok.download().sum()))

    # ---- untimed aux leg: the other BLS shapes, strong scaling, aggregates ----------------
    aux = {}
    if not args.no_aux:
        def verify_two():
            nonlocal flag2
            eng._call("sylow_hip_bls_verify_two_pairings_batch", pk.ptr, None, dm.ptr, doff.ptr, sig.ptr, None, ok.ptr, nv)
            flag2 = and_over_ranks(ok, nv, flag_dev)

        dtf, verify2_ms = timed_ranks(verify_two, 3, "bls_verify_two_pairings")
        all_valid_two = int(flag2.item())
        n_bad = int(nv - int(ok.download().sum()))
        dts, same_ms = timed_ranks(lambda: eng._call("sylow_hip_bls_verify_same_signer_batch", pk.ptr, None, dm.ptr, doff.ptr, sig.ptr, None, ok.ptr, nv), 3, "same_signer")

        # ---- STRONG scaling, BASELINE.json configs[3] as written: ONE batch of 2^20 verifies (and of 2^20 pairings) sharded over the
        # ranks as contiguous blocks (sharding.shard_bounds); every element is independent and synthetic, so rank r's block is the first
        # hi - lo elements of its own arrays, re-packed to the block's SoA stride outside the clock
        n_total = nv
        lo, hi = sharding.shard_bounds(n_total, rank, world)
        ms = hi - lo

        def block(d, rows, dtype=np.uint64):
            return d if ms == nv else eng.empty((rows, ms) if rows else (ms,), dtype).upload(np.ascontiguousarray(d.download()[..., :ms]))

        pk_s, sig_s, p_s, q_s = block(pk, 16), block(sig, 8), block(p, 8), block(q, 16)
        dm_s = dm if ms == nv else eng.to_device(msgs_np[:ms].reshape(-1))
        doff_s = doff if ms == nv else eng.to_device(off[:ms + 1])
        ok_s, gt_s = eng.empty((max(ms, 1),), np.uint8), eng.empty((48, max(ms, 1)))

        def verify_strong():
            nonlocal flag3
            if ms:
                eng._call("sylow_hip_bls_verify_batch", pk_s.ptr, None, dm_s.ptr, doff_s.ptr, sig_s.ptr, None, ok_s.ptr, ms)
            flag3 = and_over_ranks(ok_s, ms, flag_dev)

        dtvs, _ = timed_ranks(verify_strong, 3, "strong_verify")
        all_valid_strong = int(flag3.item())
        dtps, _ = timed_ranks(lambda: ms and eng._call("sylow_hip_pairing_batch", p_s.ptr, None, q_s.ptr, None, gt_s.ptr, ms), 3, "strong_pairing")
        strong = {"scaling": "strong", "batch_total": n_total, "shard_this_rank": ms, "bls_verifies_per_s": n_total / dtvs, "pairings_per_s": n_total / dtps,
                  "bls_all_valid": all_valid_strong,
                  "note": "ONE batch of %d sharded over %d rank(s) as contiguous blocks; time = slowest rank between barriers; verify includes the AND over ranks" % (n_total, world)}
        del pk_s, sig_s, p_s, q_s, ok_s, gt_s

        # aggregate verification (examples/verify_multiple_messages_same_signer.rs:41-60 / threshold_signing.rs:92-121 shape): the product
        # of the 2 nv pairs (sig_i, G2gen), (-H(m_i), pk_i) of EVERY rank == identity as ONE boolean: each rank reduces its shard to a raw
        # Miller product (signatures summed in G1 first), the 384-byte partials are all-gathered over the rank's ncclComm_t and every rank
        # finishes product + final exponentiation (sylow_hip_bls_aggregate_verify_batch(comm)); hashing is inside the clock.
        # Without a communicator (one rank, or the gloo debugging backend): per-rank booleans AND-ed through the process group.
        na = nv
        gt1, is1 = eng.empty((48, 1)), eng.empty((1,), np.uint8)
        agg = lambda: eng._call("sylow_hip_bls_aggregate_verify_batch", pk.ptr, None, nv, dm.ptr, doff.ptr, sig.ptr, None, nv, comm_ptr, gt1.ptr, is1.ptr)
        eng.sync(); eng.trim(0)
        free_before = torch.cuda.mem_get_info()[0]
        dta, agg_ms = timed_ranks(agg, 3, "aggregate")
        agg_scratch_peak = int(free_before - torch.cuda.mem_get_info()[0])   # what the library's leased blocks hold right after the call: the line tables of the n-pair product
        eng.sync(); eng.trim(0)                                              # a host that shares the GPU hands them back between batches (sylow_hip_trim)
        agg_scratch = int(free_before - torch.cuda.mem_get_info()[0])        # the steady state between calls
        agg_ok = int(is1.download()[0]) if comm is not None else sharding.all_valid(int(is1.download()[0]), dist)
        # the same with ONE signer for the whole batch: both halves collapse (n hashes, two G1 sums, a two-pair product)
        k1 = eng.xoshiro_fp_soa(SEED + 9, 1)
        sk1, sk1one = eng.empty((4, nv)).upload(np.repeat(k1, nv, axis=1)), eng.empty((4, 1)).upload(k1)
        sig1, sig1i = eng.empty((8, nv)), eng.empty((nv,), np.uint8)
        pk1, pk1i = eng.empty((16, 1)), eng.empty((1,), np.uint8)
        g2one = eng.empty((16, 1)).upload(limbs_row(G2).T.copy())
        eng._call("sylow_hip_bls_sign_batch", sk1.ptr, dm.ptr, doff.ptr, sig1.ptr, sig1i.ptr, nv)
        eng._call("sylow_hip_g2_scalar_mul_subgroup_batch", g2one.ptr, None, sk1one.ptr, pk1.ptr, pk1i.ptr, 1)
        agg1 = lambda: eng._call("sylow_hip_bls_aggregate_verify_batch", pk1.ptr, None, 1, dm.ptr, doff.ptr, sig1.ptr, None, nv, comm_ptr, gt1.ptr, is1.ptr)
        dta1, agg1_ms = timed_ranks(agg1, 3, "aggregate_same_signer")
        agg1_ok = int(is1.download()[0])
        del sk1, sig1, sig1i
        aux = {"aggregate_verify_sigs_per_s": world * na / dta, "aggregate_all_valid": agg_ok, "aggregate_batch_per_gpu": na,
               "aggregate_scratch_bytes": agg_scratch, "aggregate_scratch_bytes_during_calls": agg_scratch_peak,
               "aggregate_scratch_note": "device memory the library holds after aggregate_verify + sylow_hip_trim(0) (steady state) and right after the calls (leased blocks, kept for reuse "
               "until trimmed); the line tables take at most min(12 GB, a quarter of the memory free at the first multi-pair call) unless sylow_hip_set_scratch_limit says otherwise -- DESIGN.md 4.1",
               "aggregate_same_signer_sigs_per_s": world * nv / dta1, "aggregate_same_signer_all_valid": agg1_ok,
               "aggregate_path": ("native: sylow_hip_bls_aggregate_verify_batch over this rank's ncclComm_t (all-gather of %d partial products)" % rccl_ranks) if comm is not None
                                 else "per-rank product, booleans AND-ed through the process group",
               "bls_signs_per_s": world * nv / dtsg, "bls_verifies_per_s": world * nv * args.steps / v_elapsed, "same_signer_shape_checks_per_s": world * nv / dts, "bls_verify_batch_per_gpu": nv,
               "bls_all_valid": all_valid_weak, "bls_verify_algorithmic_GBps": world * nv * VERIFY_BYTES * args.steps / v_elapsed / 1e9,
               "bls_verify_frac_of_hbm": world * nv * VERIFY_BYTES * args.steps / v_elapsed / 1e9 / (HBM_PEAK_GBS * world),
               "bls_verifies_per_s_two_pairings": world * nv / dtf, "bls_all_valid_two_pairings": all_valid_two, "bad_flags_this_rank": n_bad,
               "kernel_ms_spread": spreads,
               "kernel_ms_rank0": {"bls_sign": sign_ms, "bls_verify": verify_ms, "bls_verify_two_pairings": verify2_ms, "same_signer": same_ms,
                                   "aggregate": agg_ms, "aggregate_same_signer": agg1_ms},
               "and_path": "native: sylow_hip_all_valid (device AND + ncclAllReduce(min) on this rank's ncclComm_t)" if comm is not None
                           else "sylow_hip_flags_all + torch.distributed MIN" if dist is not None else "sylow_hip_flags_all (one rank)",
               "strong": strong,
               "issue_roof": {k: issue_fields(pmc_cfgs.get(v)) for k, v in (("bls_sign", "bls_sign_2^20"), ("bls_verify", "bls_verify_2^20"),
                                                                               ("bls_verify_two_pairings", "bls_verify_two_pairings_2^20"),
                                                                               ("same_signer_shape", "bls_verify_same_signer_shape_2^20"),
                                                                               ("aggregate", "aggregate_verify_2^20"), ("aggregate_same_signer", "aggregate_same_signer_2^20"))
                              if nv == 1 << 20 and pmc_cfgs.get(v)},
               "timing": "bls_verifies_per_s: the contract loop (--warmup, --steps); every other figure: mean of 3 calls after a warm call, slowest rank's wall clock between barrier + synchronize; kernel_ms_rank0 = HIP events on the launch stream, kernel_ms_spread = [min, max, calls] of the per-call events",
               "note": "verify = sylow_hip_bls_verify_batch: the boolean of lib.rs:223-236 as e(sig,G2gen)*e(-H,pk)==1 (hash + shared-squaring 2-pair Miller loop + ONE final exponentiation); "
                       "two_pairings = the same boolean evaluated literally (hash + two full pairings + compare); "
                       "aggregate = prod_i e(sig_i,G2gen) e(-H(m_i),pk_i) == identity as one boolean: hash + G1 sum of the signatures + product tree over the n key pairs + one final exponentiation; "
                       "aggregate_same_signer = one key: hash + two G1 sums + a two-pair product"}
        # ---- staggered launch A/B on THIS box, through the ABI's option (no environment variable): 3 launches per mode, alternating
        prev_stagger = eng.get_option("STAGGER")
        stagger_ab = {}
        for lg in (args.log2n, 17):
            m = 1 << lg
            if m > n or ("2^%d" % lg) in stagger_ab:
                continue
            if m == n:
                p_m, q_m, gt_m = p, q, gt
            else:
                p_m = eng.empty((8, m)).upload(np.ascontiguousarray(p.download()[:, :m]))
                q_m = eng.empty((16, m)).upload(np.ascontiguousarray(q.download()[:, :m]))
                gt_m = eng.empty((48, m))
            times = {0: [], 1: []}
            for _ in range(3):
                for mode in (0, 1):
                    eng.set_option("STAGGER", mode)
                    times[mode].append(hip_timed(torch, stream, lambda: eng._call("sylow_hip_pairing_batch", p_m.ptr, None, q_m.ptr, None, gt_m.ptr, m), 1) * 1e3)
            stagger_ab["2^%d" % lg] = {"plain_ms": times[0], "staggered_ms": times[1], "gain": float(np.mean(times[0]) / np.mean(times[1]))}
            del p_m, q_m, gt_m
        eng.set_option("STAGGER", prev_stagger)
        aux["stagger_ab"] = dict(stagger_ab, note="k_pairing launched plain (STAGGER option 0) and skewed (1), alternating, 3 single launches each after a warm launch, HIP events; gain = plain / staggered")
        if world == 1 and not args.force_dist:
            p_h, q_h, pk_h, sig_h = p.download(), q.download(), pk.download(), sig.download()
            aux["size_sweep"] = size_sweep(eng, torch, stream, p_h, q_h, pk_h, sig_h, dm, off, n, sk_h=sk.download())
            aux["e2e"] = end_to_end(eng, p_h, q_h, gt.download(), pk_h, sig_h, msgs_np, off, n * args.steps / elapsed, nv * args.steps / v_elapsed)
            del p_h, q_h, pk_h, sig_h
            del dm, doff, sk, g2, pk, pki, sig, sigi, ok, pk1, pk1i, g2one, sk1one
            aux["configs"] = single_gpu_configs(eng, torch, stream, p, q, ka, n, pmc_cfgs)

    if rank == 0:
        total = world * n * args.steps
        value = total / elapsed
        verifies_per_s = world * nv * args.steps / v_elapsed
        kern_s = kern_ms * 1e-3
        hbm_achieved = PAIRING_BYTES * n / kern_s / 1e9
        main_cfg = pmc_cfgs.get("pairing_2^%d" % args.log2n) or {}
        ver_cfg = pmc_cfgs.get("bls_verify_2^%d" % args.log2n) or {}
        have_issue = main_cfg.get("issue_frac") is not None
        issue_cycles = main_cfg["issue_cycles_ideal_per_unit"] * n if have_issue else None       # ideal VALU issue cycles of one launch
        pmc_note = (pmc["source"] if main_cfg else
                    "profiles/pmc_current.json was measured on another build of the kernels (source hash differs): re-run tools/prof_configs.sh"
                    if pmc is not None else "no profiles/pmc_current.json")
        # ---- live figures of THIS run (clock probe): the engine clock the timed kernels sustained, and the issue fraction against it -------
        sus_hz = probe_p["sustained_mhz"] * 1e6 if probe_p["sustained_mhz"] else None
        sus_hz_v = probe_v["sustained_mhz"] * 1e6 if probe_v["sustained_mhz"] else None
        ideal_p, ideal_v = main_cfg.get("issue_cycles_ideal_per_unit"), ver_cfg.get("issue_cycles_ideal_per_unit")
        simd_live = kern_s * sus_hz * N_SIMD / n if sus_hz else None                     # SIMD cycles the chip spent per pairing, live
        simd_live_v = (verify_ms * 1e-3) * sus_hz_v * N_SIMD / nv if sus_hz_v else None
        sab = aux.get("stagger_ab", {}) if aux else {}
        # the driver's record keeps a PREFIX of this object's scalar keys: what certifies the line comes first
        roofline = {
            "bound": "valu-issue", "unit": "G SIMD issue cycles/s",
            "achieved": issue_cycles / kern_s / 1e9 if have_issue else None, "peak": ISSUE_PEAK / 1e9,
            "frac": issue_cycles / kern_s / ISSUE_PEAK if have_issue else None,
            "traffic": (main_cfg["hbm_bytes_per_unit"] * n) if main_cfg.get("hbm_bytes_per_unit") else None,
            "sustained_mhz": probe_p["sustained_mhz"],
            "frac_at_sustained_clock": (ideal_p / simd_live) if (ideal_p and simd_live) else None,
            "frac_clock_free": main_cfg.get("issue_frac"),
            "simd_cycles_per_pairing_live": simd_live, "simd_cycles_per_pairing": main_cfg.get("simd_cycles_per_unit"),
            "kernel": "plk::k_pairing", "kernel_ms": kern_ms, "kernel_ms_min": kern_ms_min, "kernel_ms_max": kern_ms_max,
            "valu_instr_per_pairing": main_cfg.get("valu_instr_per_unit"), "issue_cycles_ideal_per_pairing": ideal_p,
            "verify_sustained_mhz": probe_v["sustained_mhz"],
            "verify_frac_at_sustained_clock": (ideal_v / simd_live_v) if (ideal_v and simd_live_v) else None,
            "verify_issue_frac_clock_free": ver_cfg.get("issue_frac"),
            "stagger_gain_2^%d" % args.log2n: (sab.get("2^%d" % args.log2n) or {}).get("gain"),
            "stagger_gain_2^17": (sab.get("2^17") or {}).get("gain"),
            "hbm_achieved": hbm_achieved, "hbm_frac": hbm_achieved / HBM_PEAK_GBS,
            # ---- beyond the prefix ----
            "hbm_peak": HBM_PEAK_GBS, "hbm_unit": "GB/s",
            "frac_vs_measured_issue_rates": main_cfg.get("issue_frac_vs_measured_issue_rates"),
            "kernel_ms_profiled": main_cfg.get("ms_per_launch_kernel_sum", main_cfg.get("ms_per_launch")),
            "int64_class_frac": main_cfg.get("int64_class_frac"),
            "wave_ticks_per_pairing_live": probe_p["wave_ticks"] / (n * args.steps) if probe_p["wave_ticks"] else None,
            "probe_wavefronts": probe_p["waves"], "probe_longest_wavefront_ms": probe_p["longest_wave_ms"], "wall_clock_khz": wall_khz,
            "note": "achieved = (SQ_INSTS_VALU_INT64 x 4 + other VALU x 2 issue cycles per pairing, rocprofv3 PMC: a constant of the BUILD, "
                    "guarded by the source hash) x batch / live HIP-event kernel time; peak = 1024 SIMDs x 2.4 GHz; "
                    "sustained_mhz = sum of s_memtime deltas / sum of s_memrealtime deltas over every wavefront of the K timed launches "
                    "(sylow_hip_clock_probe), x the constant rate: the clock THIS box held under THIS kernel; "
                    "frac_at_sustained_clock = ideal issue cycles / (kernel time x sustained clock x 1024 SIMDs) -- the live twin of "
                    "frac_clock_free = issue cycles / (GRBM_GUI_ACTIVE / 8 x 1024) of the profiled pass",
            "pmc_source": pmc_note,
            "hbm_note": "ALGORITHMIC bytes (576 per pairing) / kernel time: HBM cannot bind a pairing; the HBM-bound kernels of the path are aux.configs C2a",
            "algorithmic_bytes_per_launch": PAIRING_BYTES * n,
            "traffic_note": "HBM bytes per launch, rocprofv3 PMC: 2*FETCH_SIZE + WRITE_SIZE (gfx950 correction)",
            "verify_kernel": "k_hash_to_g1 + plk::k_bls_verify_fused", "verify_kernel_ms": verify_ms, "verify_kernel_ms_min": verify_ms_min,
            "verify_kernel_ms_max": verify_ms_max, "verify_per_s": verifies_per_s,
            "verify_valu_instr_per_unit": ver_cfg.get("valu_instr_per_unit"),
            "verify_frac": (ideal_v * nv / (verify_ms * 1e-3) / ISSUE_PEAK) if ideal_v else None,
            "verify_simd_cycles_per_unit_live": simd_live_v, "verify_simd_cycles_per_unit": ver_cfg.get("simd_cycles_per_unit"),
            "verify_probe_note": "the verify probe covers plk::k_bls_verify_fused only; verify_kernel_ms also holds k_hash_to_g1 (~5 %), so verify_frac_at_sustained_clock prices the two kernels' time against the pairing kernel's clock",
            "verify_hbm_bytes_per_unit_measured": ver_cfg.get("hbm_bytes_per_unit")}
        # every value in `config` / `roofline` is a scalar: the driver's record keeps scalars of these two objects and drops nested ones
        out = {
            "metric": "BN254 pairings/s (value) and BLS verifies/s (config.bls_verifies_per_s) at batch=2^%d per GPU" % args.log2n,
            "value": value, "unit": "pairings/s",
            "n_gpus": world, "rccl_ranks": rccl_ranks, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "i32/u32 limbs (9x29-bit carry-free core + 8x32-bit Montgomery, exact integer)", "data": "synthetic",
            "config": {"workload": f"2^{args.log2n} pairings e(a_i*G1, b_i*G2) per GPU per step, then 2^{args.log2n} BLS verifies per GPU per step; SoA inputs resident in HBM",
                       "batch_per_gpu": n, "parallelism": f"independent shards x{world}, no data-path collective",
                       "bls_verifies_per_s": verifies_per_s, "bls_verify_ms_per_step": v_elapsed / args.steps * 1e3,
                       "bls_verify_steps": args.steps, "bls_verify_warmup": args.warmup, "bls_verify_batch_per_gpu": nv,
                       "bls_all_valid": all_valid_weak, "bls_bad_flags_this_rank": n_bad_fused,
                       "bls_verify_note": "sylow_hip_bls_verify_batch + AND over ranks (lib.rs:223-236 as e(sig,G2gen) e(-H(m),pk) == 1), same loop as the pairings",
                       "timed_region_s": elapsed + v_elapsed},
            "roofline": roofline,
        }
        if aux:
            out["aux"] = aux
        if not args.no_cpu and world == 1:
            out["cpu_baseline"] = cpu_baseline(check)
        if dist is not None:
            out["collective_backend"] = backend
            if comm_err:
                out["rccl_native_error"] = comm_err
        print(json.dumps(out), flush=True)
    if comm is not None:
        torch.cuda.synchronize()
        comm.destroy()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
