"""Lane-pair kernels (sylow_amd/csrc/plk_*.hip): ragged batch sizes (partial wavefronts, odd element counts), identity flags mixed
inside one wavefront (pair geometry: bn254_pair.hpp -- lanes l and 7 - l of a group of 8)."""
import os

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
