"""`python bench.py --gpus N` without a launcher (CPU side): the parent builds the torch.distributed.run command the contract names, one
rank per GPU on 127.0.0.1, forwards its own arguments, returns the launcher's exit code -- and does so before anything imports torch."""
import importlib.util
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_spawn_command(monkeypatch):
    bench = load_bench()
    seen = {}

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return subprocess.CompletedProcess(cmd, 7)

    monkeypatch.setattr(bench.subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "20", "--warmup", "5"])
    monkeypatch.delenv("MASTER_PORT", raising=False)
    assert bench.spawn_ranks(8) == 7                               # the launcher's exit code comes back
    cmd = seen["cmd"]
    assert cmd[:4] == [sys.executable, "-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and 0 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    assert cmd[-7:] == [os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "20", "--warmup", "5"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_bare_multi_gpu_command_spawns_before_importing_torch(monkeypatch):
    bench = load_bench()
    calls = []
    monkeypatch.setattr(bench, "spawn_ranks", lambda n: calls.append(n) or 0)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    had_torch = "torch" in sys.modules
    try:
        bench.main()
    except SystemExit as e:
        assert e.code == 0
    assert calls == [4]
    assert had_torch or "torch" not in sys.modules                # the parent never got as far as importing torch
