"""Lane-pair kernels (sylow_amd/csrc/plk_*.hip): ragged batch sizes (partial wavefronts, odd element counts), identity flags mixed
inside one wavefront, and the single-lane twin (SYLOW_HIP_SINGLE_LANE=1) replaying the pairing test files in a subprocess."""
import os
import subprocess
import sys

import numpy as np
import pytest

from helpers import SEED, Xoshiro, limbs, pack
from test_gpu_pairing import G1, G2, random_points

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("n", [1, 2, 3, 31, 32, 33, 65, 127])
def test_ragged_sizes_vs_oracle(engine, coracle, n):
    rng = Xoshiro(SEED + 200 + n)
    _, _, p_xy, q_xy = random_points(engine, rng, n)
    f = engine.miller_loop(p_xy, q_xy)
    assert np.array_equal(f, coracle.miller_loop(p_xy, q_xy))
    gt = coracle.final_exponentiation(f)
    assert np.array_equal(engine.final_exp(f), gt)
    assert np.array_equal(engine.pairing(p_xy, q_xy), gt)


def test_identity_flags_inside_a_wavefront(engine, coracle):
    rng = Xoshiro(SEED + 230)
    n = 96
    _, _, p_xy, q_xy = random_points(engine, rng, n)
    p_inf = np.array([(i % 5 == 0) for i in range(n)], np.uint8)
    q_inf = np.array([(i % 7 == 3) for i in range(n)], np.uint8)
    got = engine.pairing(p_xy, q_xy, p_inf=p_inf, q_inf=q_inf)
    exp = coracle.final_exponentiation(coracle.miller_loop(p_xy, q_xy))
    one = np.zeros(48, np.uint64); one[0] = 1
    dead = (p_inf | q_inf).astype(bool)
    exp[dead] = one
    assert np.array_equal(got, exp)


def test_single_lane_twin_passes_the_same_pairing_tests():
    """The one-element-per-lane kernels stay selectable; they must satisfy the same parity tests."""
    env = dict(os.environ, SYLOW_HIP_SINGLE_LANE="1")
    files = ["tests/test_gpu_pairing.py", "tests/test_gpu_multi_pairing.py", "tests/test_gpu_hash_bls.py"]
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider"] + files,
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
