"""Replay of the frozen hash chain (tests/golden/hash_chain.json) through the C ABI on the GPU: hash_to_field, the SvdW map,
hash_to_curve and sign must reproduce the committed values for every message length around the Keccak rate boundary and for short,
255-byte and oversize DSTs (hasher.rs:157-173,201-250, :84-128; svdw.rs:180-262; g1.rs:307-331; lib.rs:179-187)."""
import numpy as np
import pytest

from test_hash_chain import load_chain, words

pytestmark = pytest.mark.gpu


def test_gpu_replays_the_frozen_chain(engine):
    ch = load_chain()
    dsts = {k: bytes.fromhex(v) for k, v in ch["dsts"].items()}
    total = 0
    for name, dst in dsts.items():
        es = [e for e in ch["entries"] if e["dst"] == name]
        msgs = [bytes.fromhex(e["msg"]) for e in es]
        d = None if name == "sylow" else dst
        u = engine.hash_to_field(msgs, d)
        assert np.array_equal(u, words([[e["u0"], e["u1"]] for e in es])), name
        q, st = engine.svdw_map(np.concatenate([u[:, :4], u[:, 4:]], axis=0))
        assert not st.any() and np.array_equal(q, words([e["q0"] for e in es] + [e["q1"] for e in es])), name
        h, inf = engine.hash_to_g1(msgs, d)
        assert not inf.any() and np.array_equal(h, words([e["h"] for e in es])), name
        if name == "sylow":
            s, inf = engine.bls_sign(words([[e["sk"]] for e in es]), msgs)
            assert not inf.any() and np.array_equal(s, words([e["sig"] for e in es]))
            assert np.array_equal(engine.hash_to_g1(msgs, dst)[0], h)            # the library DST passed explicitly
        total += len(es)
    assert total >= 64
