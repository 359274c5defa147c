"""The frozen hash chain (tests/golden/hash_chain.json, written by tests/golden/make_hash_chain.py from oracle/pyref.py):
msg -> b_0, b_1..b_3 -> (u0, u1) -> SvdW points -> H(msg) -> sig.  CPU part: the fixture regenerates byte-identically from the
Python restatement, and the C restatement (oracle/sylow_oracle.c) reproduces every stage it exposes -- so an edit that moves either
restatement shows up here.  (hasher.rs:201-250, :84-128; svdw.rs:180-262; g1.rs:307-331; lib.rs:179-187.)"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def load_chain():
    with open(os.path.join(GOLD, "hash_chain.json")) as f:
        return json.load(f)


def words(hexes):
    return np.array([[(int(h, 16) >> (64 * k)) & ((1 << 64) - 1) for h in hs for k in range(4)] for hs in hexes], dtype=np.uint64)


def test_fixture_regenerates_from_pyref():
    sys.path.insert(0, GOLD)
    import make_hash_chain
    fresh = json.loads(json.dumps(make_hash_chain.build(), sort_keys=True))
    assert fresh == load_chain()
    assert len(fresh["entries"]) >= 64


def test_c_oracle_reproduces_the_chain(coracle):
    ch = load_chain()
    dsts = {k: bytes.fromhex(v) for k, v in ch["dsts"].items()}
    for name, dst in dsts.items():
        es = [e for e in ch["entries"] if e["dst"] == name]
        msgs = [bytes.fromhex(e["msg"]) for e in es]
        for e, m in zip(es, msgs):
            assert coracle.expand_message_xmd_keccak(m, dst, 96).hex() == e["b1"] + e["b2"] + e["b3"]
        us = words([[e["u0"]] for e in es] + [[e["u1"]] for e in es])
        q = coracle.svdw_map(us)
        assert np.array_equal(q, words([e["q0"] for e in es] + [e["q1"] for e in es]))
        h, inf = coracle.g1_to_affine(coracle.hash_to_curve(msgs, dst))
        assert not inf.any() and np.array_equal(h, words([e["h"] for e in es]))
        if name == "sylow":
            s, inf = coracle.g1_to_affine(coracle.sign(words([[e["sk"]] for e in es]), msgs))
            assert not inf.any() and np.array_equal(s, words([e["sig"] for e in es]))
