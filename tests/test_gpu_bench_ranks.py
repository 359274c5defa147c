"""The N > 1 code the driver runs (bench.py under torch.distributed.run), executed before the driver does: two ranks as FRESH
child processes on this box's one GPU (SYLOW_BENCH_BACKEND=gloo SYLOW_BENCH_SINGLE_DEVICE=1: host-side collectives, same
barrier / MAX / MIN(=AND) logic as the RCCL path, which differs only in the backend name).  Checks the JSON line, the
aggregate flag, and that ONE planted bad signature on rank 1 flips the global AND on rank 0."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(extra):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, SYLOW_BENCH_BACKEND="gloo", SYLOW_BENCH_SINGLE_DEVICE="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--log2n", "12", "--steps", "1", "--warmup", "1", "--no-cpu"] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                      # exactly ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_two_rank_bench_all_valid():
    out = run_bench([])
    assert out["n_gpus"] == 2 and out["steps"] == 1 and out["scaling"] == "weak"
    assert out["config"]["batch_per_gpu"] == 4096 and out["value"] > 0
    aux = out["aux"]
    assert aux["bls_all_valid"] == 1 and aux["bls_all_valid_two_pairings"] == 1 and aux["aggregate_all_valid"] == 1
    assert aux["bls_verify_batch_per_gpu"] == 4096 and aux["bad_flags_this_rank"] == 0
    assert out["config"]["bls_all_valid"] == 1 and out["config"]["bls_verifies_per_s"] > 0 and out["config"]["bls_bad_flags_this_rank"] == 0
    # strong scaling: ONE batch of 4096 cut into two contiguous blocks (BASELINE.json configs[3] as written)
    st = aux["strong"]
    assert st["batch_total"] == 4096 and st["shard_this_rank"] == 2048 and st["bls_all_valid"] == 1 and st["bls_verifies_per_s"] > 0 and st["pairings_per_s"] > 0
    assert out["rccl_ranks"] is None and out["collective_backend"] == "gloo"        # two ranks on one GPU: RCCL cannot run here


def test_bare_command_spawns_its_own_ranks():
    """`python bench.py --gpus 2` with NO launcher and no WORLD_SIZE in the environment (the shape of the driver's N = 1 command):
    the parent must start the two ranks itself, relay exactly one JSON line, and exit 0."""
    env = dict(os.environ, SYLOW_BENCH_BACKEND="gloo", SYLOW_BENCH_SINGLE_DEVICE="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--log2n", "12", "--steps", "1", "--warmup", "1", "--no-cpu", "--no-aux"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["value"] > 0 and "rccl_ranks" in out
    assert out["config"]["bls_all_valid"] == 1
    # a failing rank must fail the bare command too
    bad = subprocess.run(cmd + ["--log2n", "99"], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert bad.returncode != 0


def test_two_rank_bench_planted_bad_signature_on_rank_1():
    out = run_bench(["--plant-bad", "1"])
    aux = out["aux"]
    assert aux["bls_all_valid"] == 0 and aux["bls_all_valid_two_pairings"] == 0      # rank 0 sees rank 1's failure through the reduce
    assert aux["aggregate_all_valid"] == 0 and aux["aggregate_same_signer_all_valid"] == 1   # the aggregate check sees it too
    assert aux["bad_flags_this_rank"] == 0                                       # ... although all of rank 0's own flags are set
    assert out["config"]["bls_all_valid"] == 0 and out["config"]["bls_bad_flags_this_rank"] == 0     # the headline verify loop sees it too


def run_bench_nccl_one_rank(extra):
    """backend nccl with ONE rank (plain child process, --force-dist): the process group is RCCL, the rank builds its own native
    ncclComm_t from the broadcast unique id, and the C ABI's aggregate entry points run on it -- the path the driver's N > 1 runs take."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    env.pop("SYLOW_BENCH_BACKEND", None); env.pop("SYLOW_BENCH_SINGLE_DEVICE", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--log2n", "12", "--steps", "1", "--warmup", "1", "--no-cpu"] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])


def test_one_rank_nccl_native_communicator():
    out = run_bench_nccl_one_rank([])
    assert out["rccl_ranks"] == 1 and out["collective_backend"] == "nccl" and "rccl_native_error" not in out
    aux = out["aux"]
    assert aux["and_path"].startswith("native") and aux["aggregate_path"].startswith("native")
    assert aux["bls_all_valid"] == 1 and aux["aggregate_all_valid"] == 1 and aux["aggregate_same_signer_all_valid"] == 1 and aux["strong"]["bls_all_valid"] == 1
    bad = run_bench_nccl_one_rank(["--plant-bad", "0"])["aux"]
    assert bad["bls_all_valid"] == 0 and bad["aggregate_all_valid"] == 0 and bad["strong"]["bls_all_valid"] == 0 and bad["aggregate_same_signer_all_valid"] == 1


def test_single_rank_bench_self_check():
    """The N = 1 line at a small batch: oracle spot check of the timed output, all single-GPU configs, CPU leg."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--log2n", "13", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    cb = out["cpu_baseline"]
    assert cb["checked"] == 16 and cb["mismatches"] == 0 and cb["kind"] == "port" and cb["cores"] >= 1
    rf = out["roofline"]
    assert rf["bound"] == "valu-issue" and 0 < rf["hbm_frac"] < 1 and rf["kernel_ms"] > 0 and rf["verify_kernel_ms"] > 0
    assert rf["frac"] is None or 0 < rf["frac"] <= 1              # None while profiles/pmc_current.json belongs to another build of the kernels
    c = out["config"]                                              # the metric's second half sits where the driver's record keeps it
    assert c["bls_verifies_per_s"] > 0 and c["bls_verify_steps"] == 2 and c["bls_all_valid"] == 1 and c["bls_verify_batch_per_gpu"] == 1 << 13
    assert all(not isinstance(v, (dict, list)) for v in list(c.values()) + list(rf.values()))      # scalars only: nested objects are dropped
    sw = out["aux"]["size_sweep"]
    assert [r["n"] for r in sw["rows"]] == [1, 64, 1 << 10, 1 << 11, 1 << 12, 1 << 13] and all(r["verify_all_ok"] == 1 for r in sw["rows"])
    # round 6: the line certifies its own clock -- the live probe of the timed kernels (2^13 elements run on the quad route, which carries the
    # probe as well: 2 steps x 512 wavefronts of 16 quads)
    assert "frac_at_sustained_clock" in rf and "stagger_gain_2^17" in rf
    assert 500 < rf["sustained_mhz"] < 3500 and 500 < rf["verify_sustained_mhz"] < 3500 and rf["probe_wavefronts"] == 2 * 512
    cbk = list(cb)
    assert cbk.index("cgroup_cpu_quota_cores") < 16 and cbk.index("checked") < 16 and cbk.index("mismatches") < 16 and cb["nproc"] >= cb["cores"]
    e2e = out["aux"]["e2e"]
    assert e2e["pairing_pageable"]["bit_equal_to_device_resident"] == 1 and e2e["pairing_pinned"]["bit_equal_to_device_resident"] == 1
    assert e2e["verify_pageable"]["all_ok"] == 1 and e2e["verify_pinned"]["all_ok"] == 1
    cfg = out["aux"]["configs"]
    assert {"C2a_fp_mul_2^20", "C2a_fp_add_2^24", "C2b_g1_scalar_mul_2^13", "C3_pairing_2^13"} <= set(cfg)
    assert all(v["pattern_ok"] == 1 for k, v in cfg.items() if k.startswith("C5_"))
    assert out["aux"]["bls_all_valid"] == 1 and out["aux"]["aggregate_all_valid"] == 1 and out["aux"]["aggregate_same_signer_all_valid"] == 1
    assert "C2c_g2_scalar_mul_2^13" in cfg
    c1 = cfg["C1_single_calls"]                         # one pairing / sign / verify per call: the single-wavefront latency routes
    assert c1["verify_ok"] == 1 and 0 < c1["pairing_ms"] < 50 and 0 < c1["verify_ms"] < 50
