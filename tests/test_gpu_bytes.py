"""GPU parity: G1/G2 wire formats (g1.rs:151-280, g2.rs:319-433; groups/mod.rs byte round-trip tests) vs the
Python restatement oracle/pyref.py."""
import numpy as np
import pytest

from helpers import P, SEED, Xoshiro, ints, limbs, pack
from oracle import pyref as R

pytestmark = pytest.mark.gpu
G1 = [1, 2]
G2 = list(R.G2_GEN_AFF[0]) + list(R.G2_GEN_AFF[1])


def test_g1_bytes_roundtrip_and_rejections(engine):
    rng = Xoshiro(SEED + 70)
    n = 64
    xy, inf = engine.g1_scalar_mul(np.repeat(pack(G1, 8), n, 0), limbs([rng.fp() for _ in range(n)]))
    inf[5] = 1
    blobs = engine.g1_to_be_bytes(xy, inf)
    for i in range(n):
        a = (ints(xy[i:i + 1])[0], ints(xy[i:i + 1])[1], bool(inf[i]))
        assert blobs[i] == R.g1_to_be_bytes(a)
    assert blobs[5][0] == 0x80 and blobs[5][63] == 1                    # identity = (0, 1) + flag
    dxy, dinf, st = engine.g1_from_be_bytes(blobs)
    assert st.tolist() == [0] * n and np.array_equal(dinf, inf)
    keep = inf == 0
    assert np.array_equal(dxy[keep], xy[keep]) and ints(dxy[5:6]) == [0, 1]
    bad = [
        (P).to_bytes(32, "big") + blobs[0][32:],                         # x >= p  (fp.rs:821-826)
        blobs[0][:32] + ((1 << 256) - 1).to_bytes(32, "big"),            # y >= p
        b"\x11" * 64,                                                    # off-curve (reth_bn128.rs:294-307)
        bytes([blobs[0][0] | 0x80]) + blobs[0][1:],                      # flag set on a finite point
        bytes(64),                                                       # (0, 0) without flag: not on the curve
        blobs[0][:63] + bytes([blobs[0][63] ^ 1]),                       # perturbed y
    ]
    _, binf, bst = engine.g1_from_be_bytes(bad)
    assert bst.tolist() == [4, 4, 1, 4, 1, 1] and binf.tolist() == [1] * 6
    for b, s in zip(bad, bst):
        try:
            ref = R.g1_from_be_bytes(b)
        except Exception:
            ref = None
        assert (ref is None) == (s != 0)


def test_g2_bytes_roundtrip_and_rejections(engine):
    rng = Xoshiro(SEED + 71)
    n = 24
    xy, inf = engine.g2_scalar_mul(np.repeat(pack(G2, 16), n, 0), limbs([rng.fp() for _ in range(n)]))
    inf[3] = 1
    blobs = engine.g2_to_be_bytes(xy, inf)
    for i in range(n):
        v = ints(xy[i:i + 1])
        assert blobs[i] == R.g2_to_be_bytes(((v[0], v[1]), (v[2], v[3]), bool(inf[i])))
    dxy, dinf, st = engine.g2_from_be_bytes(blobs)
    assert st.tolist() == [0] * n and np.array_equal(dinf, inf)
    keep = inf == 0
    assert np.array_equal(dxy[keep], xy[keep])
    # twist point outside the r-torsion, coordinate >= p, off-curve
    from test_gpu_groups import fp2_sqrt
    while True:
        x = (rng.fp(), rng.fp())
        y = fp2_sqrt(R.fp2_add(R.fp2_mul(R.fp2_square(x), x), R.TWIST_B))
        if y is not None:
            break
    outside = b"".join(v.to_bytes(32, "big") for v in (x[1], x[0], y[1], y[0]))
    bad = [outside, P.to_bytes(32, "big") + blobs[0][32:], blobs[0][:127] + bytes([blobs[0][127] ^ 1])]
    _, binf, bst = engine.g2_from_be_bytes(bad)
    assert bst.tolist() == [2, 4, 1] and binf.tolist() == [1, 1, 1]
