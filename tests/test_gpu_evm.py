"""The reference's EVM precompile tests (examples/reth_bn128.rs:229-502: test_alt_bn128_add / _mul / _pair)
replayed through the batched adapter sylow_amd/evm.py -> sylow_hip_evm_*_batch."""
import numpy as np
import pytest

from helpers import P, SEED, Xoshiro
from oracle import pyref as R

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def vec(kats):
    return {e["line"]: bytes.fromhex(e["hex"]) for e in kats["eip_vectors_raw"]["hex_literals"]}


def test_alt_bn128_add(engine, vec):
    from sylow_amd import evm
    E = evm.PrecompileError
    inputs = [vec[230], vec[249], vec[268], b"", vec[294]]
    limits = [500, 500, 499, 500, 500]
    out = evm.run_add(engine, inputs, evm.BYZANTIUM_ADD_GAS_COST, limits)
    assert out[0] == vec[238]                                   # reth_bn128.rs:230-246
    assert out[1] == vec[257]                                   # zero sum test
    assert out[2] == E(E.OUT_OF_GAS)                            # out of gas test
    assert out[3] == vec[283] == bytes(64)                      # no input test (right-padded with zeros)
    assert out[4] == E(E.FAILED_TO_CREATE)                      # point not on curve


def test_alt_bn128_mul(engine, vec):
    from sylow_amd import evm
    E = evm.PrecompileError
    inputs = [vec[312], vec[330], vec[342], b"", vec[372]]
    limits = [40_000, 39_999, 40_000, 40_000, 40_000]
    out = evm.run_mul(engine, inputs, evm.BYZANTIUM_MUL_GAS_COST, limits)
    assert out[0] == vec[319]                                   # reth_bn128.rs:312-327
    assert out[1] == E(E.OUT_OF_GAS)
    assert out[2] == vec[349]                                   # zero multiplication test
    assert out[3] == vec[361] == bytes(64)                      # no input test
    assert out[4] == E(E.FAILED_TO_CREATE)


def test_alt_bn128_pair(engine, vec):
    from sylow_amd import evm
    E = evm.PrecompileError
    inputs = [vec[389], vec[419], b"", vec[460], vec[483]]
    full = 2 * evm.BYZANTIUM_PAIR_PER_POINT + evm.BYZANTIUM_PAIR_BASE
    limits = [260_000, full - 1, 260_000, 260_000, 260_000]
    out = evm.run_pair(engine, inputs, gas_limits=limits)
    assert out[0] == vec[406] == (1).to_bytes(32, "big")        # reth_bn128.rs:389-416
    assert out[1] == E(E.OUT_OF_GAS)
    assert out[2] == vec[447] == (1).to_bytes(32, "big")        # no input test
    assert out[3] == E(E.FAILED_TO_CREATE)                      # point not on curve (0x11.. x 192)
    assert out[4] == E(E.PAIR_LENGTH)                           # invalid input length


def test_evm_edge_semantics(engine, vec):
    """field elements >= p are rejected (Bn128FieldPointNotAMember); identity pairs are skipped (EIP-197);
    out-of-subgroup G2 is rejected; ecMul reduces scalars mod r."""
    from sylow_amd import evm
    E = evm.PrecompileError
    good = vec[389]
    big = (P).to_bytes(32, "big")
    assert evm.run_add(engine, [big + good[32:64] + bytes(64)])[0] == E(E.NOT_A_MEMBER)
    assert evm.run_pair(engine, [good[:32] + big + good[64:]])[0] == E(E.NOT_A_MEMBER)
    # a valid job with an extra (G1, 0) and (0, G2) pair still multiplies to one
    extra1 = good[:64] + bytes(128)
    extra2 = bytes(64) + good[64:192]
    assert evm.run_pair(engine, [good + extra1 + extra2])[0] == (1).to_bytes(32, "big")
    assert evm.run_pair(engine, [good[:192]])[0] == (0).to_bytes(32, "big")       # a single non-trivial pairing is not one
    # G2 point on the twist but outside the r-torsion
    rng = Xoshiro(SEED + 60)
    from test_gpu_groups import fp2_sqrt
    while True:
        x = (rng.fp(), rng.fp())
        y = fp2_sqrt(R.fp2_add(R.fp2_mul(R.fp2_square(x), x), R.TWIST_B))
        if y is not None:
            break
    g2bad = b"".join(v.to_bytes(32, "big") for v in (x[1], x[0], y[1], y[0]))
    assert evm.run_pair(engine, [good[:64] + g2bad])[0] == E(E.FAILED_TO_CREATE)
    # ecMul: k and k + r give the same point; k = r gives the identity (64 zero bytes)
    pt = vec[312][:64]
    k = 0x1234567890ABCDEF
    a = evm.run_mul(engine, [pt + k.to_bytes(32, "big"), pt + (k + R.R_ORDER).to_bytes(32, "big"), pt + R.R_ORDER.to_bytes(32, "big"),
                             pt + ((1 << 256) - 1).to_bytes(32, "big")])
    assert a[0] == a[1] and a[2] == bytes(64)
    exp = R.affine_from_proj(R.F1, R.proj_scalar_mul(R.F1, (int.from_bytes(pt[:32], "big"), int.from_bytes(pt[32:], "big"), 1), ((1 << 256) - 1) % R.R_ORDER))
    assert a[3] == R.g1_to_be_bytes_scrubbed(exp)
    # ecAdd against the oracle's byte-level group law on random points
    pts = [R.affine_from_proj(R.F1, R.proj_scalar_mul(R.F1, (1, 2, 1), rng.fp())) for _ in range(8)]
    ins = [R.g1_to_be_bytes_scrubbed(pts[i]) + R.g1_to_be_bytes_scrubbed(pts[i + 1]) for i in range(7)]
    outs = evm.run_add(engine, ins)
    for i in range(7):
        s = R.affine_from_proj(R.F1, R.proj_add(R.F1, R.proj_from_affine(R.F1, pts[i]), R.proj_from_affine(R.F1, pts[i + 1])))
        assert outs[i] == R.g1_to_be_bytes_scrubbed(s)
