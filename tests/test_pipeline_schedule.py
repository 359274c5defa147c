"""CPU: the chunk schedule of the host pipelines (sylow_amd/csrc/pipeline_schedule.hpp) compiled with g++ -- every chunk within 32 * base
elements, the cuts cover the batch (tests/cpp/schedule_test.cpp)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_schedule_bounds(tmp_path):
    exe = str(tmp_path / "schedule_test")
    subprocess.run(["g++", "-O1", "-std=c++17", os.path.join(ROOT, "tests", "cpp", "schedule_test.cpp"), "-o", exe], check=True, timeout=300)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.startswith("OK"), r.stdout + r.stderr
