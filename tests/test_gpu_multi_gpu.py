"""Multi-GPU pre-flight: the day this suite lands on a box with >= 2 MI355X it runs the REAL thing -- one process per GPU under
torch.distributed.run, backend nccl (= RCCL over xGMI), a native ncclComm_t per rank -- with no edits: the C ABI's collective entry
points against the oracle (tests/multi_gpu_rank.py) and the driver's own bench command for N = 2 .. device_count (weak + strong
lines, per-rank kernel times, communicator construction time printed).  On a one-GPU box everything here skips; the N > 1 host
logic is still covered there by the gloo child runs of tests/test_gpu_bench_ranks.py and tests/test_sharding_gloo.py.
Partitioning under test: contiguous blocks, no data-path collective, a 4-byte MIN all-reduce and a 384-byte all-gather
(sylow_amd/sharding.py)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def gpu_count():
    import torch
    return torch.cuda.device_count()          # counting devices does not initialise the GPU runtime on this image


def rank_counts():
    n = gpu_count()
    return sorted({k for k in (2, 4, 8, n) if 2 <= k <= n})


def launch(n, script_args, timeout=1800):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("SYLOW_BENCH_BACKEND", None); env.pop("SYLOW_BENCH_SINGLE_DEVICE", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port)] + script_args
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


needs_two = pytest.mark.skipif(gpu_count() < 2, reason="needs >= 2 GPUs on this node (RCCL refuses two ranks on one device)")


@needs_two
def test_native_collectives_n_ranks_vs_oracle():
    for n in rank_counts():
        out = launch(n, [os.path.join(ROOT, "tests", "multi_gpu_rank.py")])
        print(f"[multi-gpu pre-flight] N={n}: {json.dumps(out)}")
        assert out["world"] == n and out["rccl_ranks"] == n
        assert out["product_all_equals_oracle"] == 1 and out["product_all_equals_one_gpu_product"] == 1 and out["product_all_same_on_every_rank"] == 1
        assert out["all_valid"] == 1 and out["aggregate"] == 1
        assert out["all_valid_planted"] == 0 and out["aggregate_planted"] == 0          # one bad signature on the LAST rank flips every rank's answer


@needs_two
def test_driver_bench_command_n_ranks_rccl():
    for n in rank_counts():
        args = [os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--log2n", "16", "--steps", "2", "--warmup", "1", "--no-cpu"]
        out = launch(n, args)
        print(f"[multi-gpu pre-flight] bench N={n}: value={out['value']:.0f} pairings/s, verifies={out['config']['bls_verifies_per_s']:.0f}/s, "
              f"strong={json.dumps(out['aux']['strong'])}, kernel_ms_rank0={json.dumps(out['aux']['kernel_ms_rank0'])}")
        assert out["n_gpus"] == n and out["rccl_ranks"] == n and out["collective_backend"] == "nccl" and "rccl_native_error" not in out
        assert out["aux"]["and_path"].startswith("native") and out["aux"]["aggregate_path"].startswith("native")
        assert out["config"]["bls_all_valid"] == 1 and out["aux"]["aggregate_all_valid"] == 1 and out["aux"]["strong"]["bls_all_valid"] == 1
        assert out["aux"]["strong"]["batch_total"] == 1 << 16
        bad = launch(n, args + ["--plant-bad", str(n - 1)])
        assert bad["config"]["bls_all_valid"] == 0 and bad["aux"]["aggregate_all_valid"] == 0 and bad["aux"]["strong"]["bls_all_valid"] == 0


@needs_two
def test_bare_bench_command_spawns_n_ranks_rccl():
    """`python bench.py --gpus N` with no launcher (round 5: the parent starts its own ranks through torch.distributed.run): backend nccl, one
    rank per GPU, one JSON line, the native communicator on every rank."""
    for n in rank_counts():
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
        for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR", "SYLOW_BENCH_BACKEND", "SYLOW_BENCH_SINGLE_DEVICE"):
            env.pop(k, None)
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--log2n", "16", "--steps", "2", "--warmup", "1", "--no-cpu", "--no-aux"]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=1800, env=env, cwd=ROOT)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, r.stdout[-2000:]
        out = json.loads(lines[0])
        assert out["n_gpus"] == n and out["rccl_ranks"] == n and out["collective_backend"] == "nccl" and out["config"]["bls_all_valid"] == 1


def test_preflight_is_armed():
    """Runs everywhere: the rank script parses, and the skip condition is the device count, nothing else."""
    import ast
    with open(os.path.join(ROOT, "tests", "multi_gpu_rank.py")) as f:
        ast.parse(f.read())
    assert gpu_count() >= 1
