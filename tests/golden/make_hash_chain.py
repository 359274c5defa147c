#!/usr/bin/env python3
"""Freeze the hash chain: writes tests/golden/hash_chain.json from oracle/pyref.py (the pure-Python restatement).

    python tests/golden/make_hash_chain.py            # rewrites the fixture (deterministic: a rerun must be byte-identical)

The reference holds no literal for Keccak-XMD -> SvdW -> sign (hasher.rs:345-388 pins SHA-256 / SHAKE vectors only), so the chain
msg -> (b_0, b_1..b_3) -> (u0, u1) -> SvdW(u0), SvdW(u1) -> H(msg) -> sig is pinned by two restatements (oracle/pyref.py and
oracle/sylow_oracle.c) + public Keccak KATs + the RFC 9380 SHA-256 vectors run through the same XMD routine.  This fixture freezes
what those restatements compute TODAY: a later edit that moves both the same way (a shared misreading of hasher.rs:201-250's framing
that still round-trips sign / verify) changes these bytes and fails tests/test_hash_chain.py.
Follows: src/hasher.rs:157-173,201-250 (expand_message), :84-128 (hash_to_field), src/svdw.rs:180-262 (map), src/groups/g1.rs:307-331
(hash_to_curve), src/lib.rs:179-187 (sign)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyref as R  # noqa: E402

# message lengths around the Keccak rate (136 bytes): the absorbed string is Z_pad (136) || msg || 2 + 1 + DST' bytes
LENGTHS = [0, 1, 2, 3, 4, 7, 8, 9, 31, 32, 33, 47, 48, 55, 56, 63, 64, 65, 96, 97, 98, 99, 100, 101, 102, 103, 104, 105, 106, 127,
           128, 129, 134, 135, 136, 137, 138, 199, 200, 201, 232, 233, 234, 235, 236, 237, 238, 239, 240, 241, 271, 272, 273, 300,
           407, 408, 409, 500, 543, 544, 545, 1000, 1023, 1024]
DSTS = [("sylow", R.DST), ("short", b"QUUX-V01-CS02-with-BN254G1_XMD:KECCAK-256_SVDW_RO_"), ("len255", b"x" * 255), ("oversize256", b"y" * 256),
        ("oversize400", bytes(range(256)) + bytes(range(144)))]
SK = [0x2545F4914F6CDD1D2545F4914F6CDD1D2545F4914F6CDD1D2545F4914F6CDD1D % R.P, 1, 2, R.R_ORDER - 1, R.P - 1]


def message(i, length):
    return bytes((131 * i + 7 * j + (j * j >> 3)) & 0xFF for j in range(length))


def b0_and_blocks(msg, dst):
    """the intermediate digests of hasher.rs:201-250: b_0 and b_1..b_3 (the 96 output bytes are b_1 || b_2 || b_3)"""
    H = R.keccak256
    if len(dst) > 255:
        dst = H(b"H2C-OVERSIZE-DST-" + dst)
    dp = dst + bytes([len(dst)])
    b0 = H(bytes(136) + msg + (96).to_bytes(2, "big") + b"\x00" + dp)
    out = R.expand_message_xmd(msg, dst if len(dst) <= 255 else dst, 96)
    return b0, [out[0:32], out[32:64], out[64:96]]


def hx(v, nbytes=32):
    return format(v, "0%dx" % (2 * nbytes))


def entry(i, length, dst_name, dst, sk):
    msg = message(i, length)
    b0, bl = b0_and_blocks(msg, dst)
    assert b"".join(bl) == R.expand_message_xmd(msg, dst, 96)
    u0, u1 = R.hash_to_field(msg, dst)
    p0, p1 = R.svdw_map_to_point(u0), R.svdw_map_to_point(u1)
    h = R.affine_from_proj(R.F1, R.hash_to_curve(msg, dst))
    e = {"msg": msg.hex(), "dst": dst_name, "b0": b0.hex(), "b1": bl[0].hex(), "b2": bl[1].hex(), "b3": bl[2].hex(),
         "u0": hx(u0), "u1": hx(u1), "q0": [hx(p0[0]), hx(p0[1])], "q1": [hx(p1[0]), hx(p1[1])], "h": [hx(h[0]), hx(h[1])]}
    if dst_name == "sylow":                       # sign() uses the library DST only (lib.rs:90,179-187)
        s = R.affine_from_proj(R.F1, R.sign(sk, msg))
        e["sk"] = hx(sk)
        e["sig"] = [hx(s[0]), hx(s[1])]
    return e


def build():
    entries = []
    for i, length in enumerate(LENGTHS):
        entries.append(entry(i, length, "sylow", R.DST, SK[i % len(SK)]))
    for j, (name, dst) in enumerate(DSTS[1:]):
        for i, length in enumerate([0, 1, 32, 135, 136, 137, 272, 500]):
            entries.append(entry(100 * (j + 1) + i, length, name, dst, 0))
    return {"about": "frozen outputs of oracle/pyref.py for the chain XMD-Keccak256 -> hash_to_field -> SvdW -> hash_to_curve -> sign; "
                     "regenerate with tests/golden/make_hash_chain.py (must be byte-identical)",
            "dsts": {name: dst.hex() for name, dst in DSTS}, "entries": entries}


if __name__ == "__main__":
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hash_chain.json")
    with open(out, "w") as f:
        json.dump(build(), f, indent=0, sort_keys=True)
        f.write("\n")
    print(f"wrote {out}: {len(build()['entries'])} entries")
