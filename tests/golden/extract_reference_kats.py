#!/usr/bin/env python3
"""Extract the known-answer LITERALS held by sylow's own tests into JSON fixtures.

Run in the authoring container only (reads /root/reference, which does not exist on the GPU
box).  Output: tests/golden/reference_kats.json -- data only (inputs and expected outputs the
reference's tests assert), no reference source text.

    python tests/golden/extract_reference_kats.py

Every entry records the reference file:line range it was lifted from.
"""
import json
import os
import re
import sys

REF = os.environ.get("SYLOW_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_kats.json")

NUM = r"(?:0[xX][0-9a-fA-F_]+|\d[\d_]*)"
WORDS4 = re.compile(r"\[\s*(" + NUM + r")\s*,\s*(" + NUM + r")\s*,\s*(" + NUM + r")\s*,\s*(" + NUM + r")\s*,?\s*\]")


def read(rel):
    with open(os.path.join(REF, rel)) as f:
        return f.read()


def fn_body(src, name, start=0):
    """Return (body_text, first_line, last_line) of `fn name(`."""
    m = re.compile(r"fn\s+" + re.escape(name) + r"\s*\(").search(src, start)
    if not m:
        raise KeyError(name)
    i = src.index("{", m.end())
    depth, j = 0, i
    while True:
        if src[j] == "{":
            depth += 1
        elif src[j] == "}":
            depth -= 1
            if depth == 0:
                break
        j += 1
    return src[i:j + 1], src.count("\n", 0, m.start()) + 1, src.count("\n", 0, j) + 1


def const_body(src, name):
    m = re.compile(r"const\s+" + re.escape(name) + r"\b").search(src)
    i = m.end()
    j = src.index(";\n", i)
    return src[i:j], src.count("\n", 0, m.start()) + 1, src.count("\n", 0, j) + 1


def toint(s):
    return int(s.replace("_", ""), 0)


def words(body):
    """All 4-word little-endian u64 array literals in order -> python ints."""
    out = []
    for m in WORDS4.finditer(body):
        w = [toint(g) for g in m.groups()]
        out.append(sum(x << (64 * i) for i, x in enumerate(w)))
    return out


def hx(v):
    return hex(v)


P = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47


def main():
    kats = {"_about": "literal KATs asserted by warlock-labs/sylow's own tests; see extract_reference_kats.py"}

    # ---- Fp: add / sub / mul literal cases (fields/fp.rs) -----------------------------------
    fp = read("src/fields/fp.rs")
    body, l0, l1 = fn_body(fp, "test_addition_cases")
    w = words(body)
    # a,b,exp | c,d,exp | (MODULUS),f,exp | g,h,exp
    kats["fp_add"] = {
        "src": f"src/fields/fp.rs:{l0}-{l1}",
        "cases": [
            [hx(w[0]), hx(w[1]), hx(w[2])],
            [hx(w[3]), hx(w[4]), hx(w[5])],
            [hx(P), hx(w[6]), hx(w[7])],  # BN254_FP_MODULUS wraps to 0 through Fp::new
            [hx(w[8]), hx(w[9]), hx(w[10])],
        ],
    }
    body, l0, l1 = fn_body(fp, "test_subtraction_cases")
    w = words(body)
    kats["fp_sub"] = {
        "src": f"src/fields/fp.rs:{l0}-{l1}",
        "cases": [
            [hx(w[0]), hx(w[1]), hx(w[2])],
            [hx(w[3]), hx(w[4]), hx(w[5])],
            [hx(w[6]), hx(w[7]), hx(w[8])],
            [hx(P), hx(P), hx(w[9])],
        ],
    }
    body, l0, l1 = fn_body(fp, "test_multiplication_cases")
    w = words(body)
    cases = [[hx(w[0]), hx(w[1]), hx(w[2])], [hx(w[3]), hx(w[4]), hx(w[5])], [hx(w[6]), hx(w[7]), hx(w[8])]]
    body2, m0, m1 = fn_body(fp, "test_multiplication_edge_cases")
    w2 = words(body2)
    # a, zero, one, large, expected(large*large)
    cases.append([hx(w2[3]), hx(w2[3]), hx(w2[4])])
    kats["fp_mul"] = {"src": f"src/fields/fp.rs:{l0}-{l1},{m0}-{m1}", "cases": cases}

    # ---- Fp2 (fields/fp2.rs) ------------------------------------------------------------------
    fp2 = read("src/fields/fp2.rs")
    body, l0, l1 = fn_body(fp2, "test_multiplication_cases")
    w = words(body)
    kats["fp2_mul"] = {
        "src": f"src/fields/fp2.rs:{l0}-{l1}",
        "cases": [
            [[hx(w[0]), hx(w[1])], [hx(w[2]), hx(w[3])], [hx(w[4]), hx(w[5])]],
            [[hx(w[6]), hx(w[7])], [hx(w[8]), hx(w[9])], [hx(w[10]), hx(w[11])]],
        ],
    }
    body, l0, l1 = fn_body(fp2, "test_division_cases")
    w = words(body)
    kats["fp2_div"] = {
        "src": f"src/fields/fp2.rs:{l0}-{l1}",
        "cases": [[[hx(w[0]), hx(w[1])], [hx(w[2]), hx(w[3])], [hx(w[4]), hx(w[5])]]],
    }
    for cname in ("TWO_INV", "P_MINUS_3_OVER_4", "P_MINUS_1_OVER_2", "FP2_TWIST_CURVE_CONSTANT"):
        body, l0, l1 = const_body(fp2, cname)
        kats["const_" + cname] = {"src": f"src/fields/fp2.rs:{l0}-{l1}", "value": [hx(x) for x in words(body)]}

    # ---- Fp6 (fields/fp6.rs) ------------------------------------------------------------------
    fp6 = read("src/fields/fp6.rs")
    body, l0, l1 = fn_body(fp6, "test_multiplication_cases")
    w = words(body)
    assert len(w) == 30, len(w)  # a, b, c=a*b ; d, e=d*d
    kats["fp6_mul"] = {
        "src": f"src/fields/fp6.rs:{l0}-{l1}",
        "cases": [
            [[hx(x) for x in w[0:6]], [hx(x) for x in w[6:12]], [hx(x) for x in w[12:18]]],
            [[hx(x) for x in w[18:24]], [hx(x) for x in w[18:24]], [hx(x) for x in w[24:30]]],
        ],
    }
    body, l0, l1 = fn_body(fp6, "test_division_cases")
    w = words(body)
    kats["fp6_div"] = {
        "src": f"src/fields/fp6.rs:{l0}-{l1}",
        "cases": [[[hx(x) for x in w[0:6]], [hx(x) for x in w[6:12]], [hx(x) for x in w[12:18]]]],
    }
    for cname in ("FROBENIUS_COEFF_FP6_C1", "FROBENIUS_COEFF_FP6_C2"):
        body, l0, l1 = const_body(fp6, cname)
        kats[cname] = {"src": f"src/fields/fp6.rs:{l0}-{l1}", "nonunit_words": [hx(x) for x in words(body)]}

    fp12 = read("src/fields/fp12.rs")
    body, l0, l1 = const_body(fp12, "FROBENIUS_COEFF_FP12_C1")
    kats["FROBENIUS_COEFF_FP12_C1"] = {"src": f"src/fields/fp12.rs:{l0}-{l1}", "nonunit_words": [hx(x) for x in words(body)]}

    # ---- G2 generator, psi constants, BLS_X (groups/g2.rs) -------------------------------------
    g2 = read("src/groups/g2.rs")
    for cname in ("G2_X", "G2_Y", "EPS_EXP0", "EPS_EXP1", "BLS_X"):
        body, l0, l1 = const_body(g2, cname)
        kats["const_" + cname] = {"src": f"src/groups/g2.rs:{l0}-{l1}", "value": [hx(x) for x in words(body)]}

    # ---- Gt generator e(G1,G2) (groups/gt.rs) pinned by pairing.rs test_gt_generator -----------
    gt = read("src/groups/gt.rs")
    body, l0, l1 = const_body(gt, "GT")
    w = words(body)
    assert len(w) == 12
    kats["gt_generator"] = {"src": f"src/groups/gt.rs:{l0}-{l1}", "value": [hx(x) for x in w]}

    # ---- pairing KAT e(a*G1, b*G2) (pairing.rs test_cases) -----------------------------------------
    pr = read("src/pairing.rs")
    body, l0, l1 = fn_body(pr, "test_cases")
    decs = [int(x) for x in re.findall(r'"(\d{20,})"', body)]
    assert len(decs) == 14, len(decs)
    kats["pairing_kat"] = {
        "src": f"src/pairing.rs:{l0}-{l1}",
        "a": hx(decs[0]), "b": hx(decs[1]), "gt": [hx(x) for x in decs[2:]],
    }
    body, l0, l1 = const_body(pr, "ATE_LOOP_COUNT_NAF")
    naf = [int(x) for x in re.findall(r"-?\d+", body.split("=", 1)[1])]
    assert len(naf) == 64
    kats["ate_loop_count_naf"] = {"src": f"src/pairing.rs:{l0}-{l1}", "value": naf}

    # ---- SvdW constants (svdw.rs test_constants) -------------------------------------------------
    sv = read("src/svdw.rs")
    body, l0, l1 = fn_body(sv, "test_constants")
    hexes = re.findall(r'from_be_hex\(\s*"([0-9a-fA-F]+)"', body)
    assert len(hexes) == 3
    kats["svdw_constants"] = {
        "src": f"src/svdw.rs:{l0}-{l1}",
        "z": "0x1", "c1": "0x4", "c2": "0x" + hexes[0], "c3": "0x" + hexes[1], "c4": "0x" + hexes[2],
    }

    # ---- RFC 9380 expand_message_xmd(SHA-256) vectors (hasher.rs tests) -------------------------
    hs = read("src/hasher.rs")
    xmd = {}
    for fname in ("short_xmd_hashmap", "long_xmd_hashmap"):
        body, l0, l1 = fn_body(hs, fname)
        pairs = re.findall(r'm\.insert\(\s*"([^"]*)"\s*,\s*"([0-9a-f]+)"\s*\)', body)
        xmd[fname] = {"src": f"src/hasher.rs:{l0}-{l1}", "pairs": pairs}
    kats["xmd_sha256"] = xmd
    # DSTs and lengths used by those tests
    m = re.search(r"mod xmd \{(.*)", hs, re.S)
    xmd_mod = m.group(1)
    dsts = re.findall(r'let dst = b"([^"]+)"', xmd_mod)
    lens = re.findall(r"let len_in_bytes = (0x[0-9a-fA-F]+|\d+)", xmd_mod)
    kats["xmd_sha256"]["dsts"] = dsts
    kats["xmd_sha256"]["lens"] = [toint(x) for x in lens]

    # ---- DST / message of the signing bench + lib.rs ---------------------------------------------
    lib = read("src/lib.rs")
    kats["dst"] = re.search(r'const DST: &\[u8; \d+\] = b"([^"]+)"', lib).group(1)

    # ---- EIP-196/197 byte vectors (examples/reth_bn128.rs, doc-comment code) ---------------------
    ex = read("examples/reth_bn128.rs")
    lines = ex.split("\n")
    # collect every hex::decode("..."\ ...) string-continuation literal with its line number
    vecs = []
    i = 0
    while i < len(lines):
        if "hex::decode(" in lines[i]:
            start = i + 1
            buf = ""
            j = i
            while True:
                buf += lines[j]
                if buf.count('"') >= 2 and re.search(r'"\s*,?\s*$|"\s*\)', lines[j]) and buf.count('"') % 2 == 0:
                    break
                j += 1
            m2 = re.search(r'"([^"]*)"', re.sub(r"\\\s*(//[!/])?\s*", "", buf.replace("//!", "").replace("///", "")))
            hexstr = re.sub(r"[^0-9a-fA-F]", "", m2.group(1)) if m2 else ""
            vecs.append({"line": start, "hex": hexstr})
            i = j + 1
        else:
            i += 1
    kats["eip_vectors_raw"] = {"src": "examples/reth_bn128.rs:229-502", "hex_literals": vecs}

    with open(OUT, "w") as f:
        json.dump(kats, f, indent=1)
    print("wrote", OUT, "with", len(kats), "entries;", len(vecs), "EIP hex literals")


if __name__ == "__main__":
    sys.exit(main())
