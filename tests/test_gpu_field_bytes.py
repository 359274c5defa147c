"""GPU parity: Fp::from_be_bytes / Fr::from_be_bytes / to_be_bytes as batch entry points (fp.rs:686-737, 746-778).
The reference returns CtOption::new(Self::new(v), v < modulus): both halves are checked -- the value (v mod modulus)
and the flag (DECODE_ERROR where the CtOption is none), incl. the reference's own rejection case (fp.rs:821-826: the
modulus itself)."""
import numpy as np
import pytest

from helpers import P, SEED, Xoshiro, ints
from oracle import pyref as R

pytestmark = pytest.mark.gpu
DECODE_ERROR = 4


def cases(mod):
    rng = Xoshiro(SEED + 70)
    edge = [0, 1, 2, mod - 2, mod - 1, mod, mod + 1, 2 * mod - 1, 2 * mod, 5 * mod, (1 << 256) - 1, 1 << 255, (1 << 254) - 1, 1 << 64, (1 << 128) - 1,
            P, R.R_ORDER, P - 1, R.R_ORDER - 1]
    return edge + [rng.u256() for _ in range(200)] + [rng.fp() for _ in range(200)]


@pytest.mark.parametrize("field", ["fp", "fr"])
def test_from_be_bytes_value_and_flag(engine, field):
    mod = P if field == "fp" else R.R_ORDER
    vals = cases(mod)
    blobs = [v.to_bytes(32, "big") for v in vals]
    out, st = getattr(engine, f"{field}_from_be_bytes")(blobs)
    assert ints(out) == [v % mod for v in vals]                                      # Self::new(v)
    assert st.tolist() == [0 if v < mod else DECODE_ERROR for v in vals]             # is_some
    # to_be_bytes(from_be_bytes(b)) == b exactly for the accepted encodings, and the canonical residue otherwise
    back = getattr(engine, f"{field}_to_be_bytes")(out)
    assert back == [(v % mod).to_bytes(32, "big") for v in vals]


def test_empty_and_ragged_sizes(engine):
    out, st = engine.fp_from_be_bytes([])
    assert out.shape == (0, 4) and st.shape == (0,)
    for n in (1, 63, 64, 65, 257):
        vals = [(i * 0x9E3779B97F4A7C15 + 1) % (1 << 256) for i in range(n)]
        out, st = engine.fr_from_be_bytes([v.to_bytes(32, "big") for v in vals])
        assert ints(out) == [v % R.R_ORDER for v in vals] and st.tolist() == [0 if v < R.R_ORDER else DECODE_ERROR for v in vals]
