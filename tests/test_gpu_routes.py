"""Every route option the library has (sylow_hip_set_option; the Python layer maps the SYLOW_HIP_<NAME> variables below onto it when it loads
the library -- the library itself reads no environment variable), forced over the pairing / verification / multi-pairing test files in a child run -- the
default build picks by batch size, so no single plain run covers every route for every shape:
  SYLOW_HIP_MULTI_TABLES=0   in-register shared-squaring schedule for every multi-pair job (default: by the batch's average job size)
  SYLOW_HIP_MULTI_TABLES=1   lines-to-HBM + table-driven loop for every job, one-pair jobs included
  SYLOW_HIP_WIDE_TAIL=0      no one-wavefront-per-element kernels: small batches and the single-element tails of the one-boolean
                             shapes run on the lane-pair kernels (k_pairing, k_bls_verify_fused, k_final_exp: by default only batches
                             above 6144 / 4096 elements reach them; SYLOW_HIP_WIDE_MAX / SYLOW_HIP_WIDE_VERIFY_MAX move these two caps for crossover runs), and bls_sign_batch / g1_scalar_mul_batch run on one lane per element (k_bls_sign: by
                             default only batches above 16384 reach it; below, sign_wide.hip's eight lanes per signature)
  SYLOW_HIP_WIDE_PACK=0 / 1  the one-wavefront kernels of small batches with one element per wavefront at every size / two elements per
                             wavefront from two elements on (default: two above one wavefront per compute unit, up to 6144 pairings)
  SYLOW_HIP_QUAD_MAX=0       no lane-quad kernels (plk_quad.hip): batches between the one-wavefront cap and 16384 elements run on one lane pair per element
  SYLOW_HIP_QUAD_MAX=1048576 (with WIDE_TAIL=0) EVERY batch of pairings / Miller loops / final exponentiations / verifications of these files on
                             one lane quad per element, single elements included
  SYLOW_HIP_TAIL_SPLIT=0     batches of whole rounds + a short tail as ONE lane-pair launch (default: the tail on quads on a side stream beside the
                             rounds); tests/test_gpu_quad.py compares the two on sizes that split
  SYLOW_HIP_AGG_FORK=0       the aggregate verifiers without their side stream
  SYLOW_HIP_STAGGER=0        k_pairing / k_bls_verify_fused launched plain (default from 2^17 elements: the launch is skewed by half a
                             period, plk_pairing.hip) -- the full-size C3 test is added to the files for this switch
  SYLOW_HIP_STAGGER=2        the skewed launch with the parking blocks' flags muted: every finishing block waits out its bound and
                             recomputes its chunk (the fallback that makes the skew independent of dispatch order); the every-row
                             test of the skewed launch is added to the files for this switch
(sylow_amd/csrc/plk_multi.hip; besides these there is only SYLOW_HIP_SIGN_WIDE_MAX, the signing threshold, consumed in sign.hip.)"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = ["tests/test_gpu_pairing.py", "tests/test_gpu_hash_bls.py", "tests/test_gpu_multi_pairing.py", "tests/test_gpu_evm.py",
         "tests/test_gpu_aggregate.py", "tests/test_gpu_lane_pair.py", "tests/test_gpu_precomputed.py", "tests/test_gpu_hash_chain.py", "tests/test_gpu_groups.py",
         "tests/test_gpu_fuzz_invariants.py", "tests/test_gpu_fr_threshold.py"]
ROUTES = [{"SYLOW_HIP_QUAD_MAX": "0"}, {"SYLOW_HIP_TAIL_SPLIT": "0"}, {"SYLOW_HIP_QUAD_MAX": "1048576", "SYLOW_HIP_WIDE_TAIL": "0"}, {"SYLOW_HIP_MULTI_TABLES": "0"}, {"SYLOW_HIP_MULTI_TABLES": "1"}, {"SYLOW_HIP_WIDE_TAIL": "0"}, {"SYLOW_HIP_AGG_FORK": "0"}, {"SYLOW_HIP_STAGGER": "0"},
          {"SYLOW_HIP_STAGGER": "2"}, {"SYLOW_HIP_WIDE_PACK": "0"}, {"SYLOW_HIP_WIDE_PACK": "1"},
          {"SYLOW_HIP_MULTI_TABLES": "0", "SYLOW_HIP_WIDE_TAIL": "0", "SYLOW_HIP_AGG_FORK": "0"}]


@pytest.mark.parametrize("route", ROUTES, ids=lambda r: ",".join(f"{k[10:]}={v}" for k, v in r.items()))
def test_forced_route_passes_the_same_tests(route):
    if any(k.startswith("SYLOW_HIP_") and k in ("SYLOW_HIP_MULTI_TABLES", "SYLOW_HIP_WIDE_TAIL", "SYLOW_HIP_WIDE_PACK", "SYLOW_HIP_AGG_FORK", "SYLOW_HIP_STAGGER", "SYLOW_HIP_QUAD_MAX", "SYLOW_HIP_TAIL_SPLIT") for k in os.environ):
        pytest.skip("already inside a forced-route run")
    env = dict(os.environ, **route)
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider", "--deselect",
                        "tests/test_gpu_aggregate.py::test_rccl_entry_points_one_rank_communicator", "--deselect",
                        "tests/test_gpu_aggregate.py::test_fallback_path_and_collectives_in_one_process"] + FILES
                       + (["tests/test_gpu_full_size.py::test_c3_pairings_2_18_bilinearity", "tests/test_gpu_full_size.py::test_staggered_launch_every_row"]
                          if "SYLOW_HIP_STAGGER" in route else [])
                       + (["tests/test_gpu_quad.py"] if ("SYLOW_HIP_TAIL_SPLIT" in route or "SYLOW_HIP_QUAD_MAX" in route) else []),
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2500:] + r.stderr[-2500:]
