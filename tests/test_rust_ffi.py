"""CPU test of the Rust side of the boundary (rust/sylow-hip, source only -- no Rust toolchain in this image): src/ffi.rs must
declare exactly the entry points of include/sylow_hip.h, with the same arity, the same pointer depth and const-ness, and the
matching scalar types.  Both files are parsed here independently of the generator (tools/gen_rust_ffi.py)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
C2RUST = {"int32_t": "i32", "uint64_t": "u64", "size_t": "usize", "uint8_t": "u8", "char": "c_char", "void": "c_void"}


def parse_header():
    text = re.sub(r"/\*.*?\*/", " ", open(os.path.join(ROOT, "include", "sylow_hip.h")).read(), flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(const\s+char\s*\*|int32_t)\s+(sylow_hip_\w+)\s*\(([^)]*)\)\s*;", text):
        ret = "*const c_char" if "char" in m.group(1) else "i32"
        params = []
        body = " ".join(m.group(3).split())
        if body != "void":
            for prm in body.split(","):
                toks = prm.replace("*", " * ").split()
                name = toks.pop()                                  # parameter name
                depth = toks.count("*")
                const = "const" in toks
                base = [t for t in toks if t not in ("*", "const")]
                assert len(base) == 1, prm
                params.append((C2RUST[base[0]], depth, const, name))
        protos[m.group(2)] = (ret, params)
    return protos


def parse_ffi():
    text = open(os.path.join(ROOT, "rust", "sylow-hip", "src", "ffi.rs")).read()
    assert 'extern "C" {' in text
    protos = {}
    for m in re.finditer(r"pub fn (sylow_hip_\w+)\(([^)]*)\)\s*->\s*([^;]+);", text):
        params = []
        if m.group(2).strip():
            for prm in m.group(2).split(","):
                name, ty = [x.strip() for x in prm.split(":")]
                depth = ty.count("*")
                const = "*const" in ty
                assert not ("*const" in ty and "*mut" in ty), ty
                params.append((ty.replace("*const", "").replace("*mut", "").strip(), depth, const, name))
        protos[m.group(1)] = (m.group(3).strip(), params)
    return protos


def test_ffi_rs_matches_header_exactly():
    h, r = parse_header(), parse_ffi()
    assert len(h) >= 85
    assert set(h) == set(r), (sorted(set(h) - set(r)), sorted(set(r) - set(h)))
    for name, (hret, hparams) in h.items():
        rret, rparams = r[name]
        assert hret == rret, (name, hret, rret)
        assert len(hparams) == len(rparams), (name, len(hparams), len(rparams))
        for (ht, hd, hc, hn), (rt, rd, rc, rn) in zip(hparams, rparams):
            assert ht == rt, (name, hn, ht, rt)
            assert hd == rd, (name, hn, "pointer depth", hd, rd)
            assert hc == rc or hd == 0, (name, hn, "const-ness", hc, rc)


def test_ffi_rs_is_what_the_generator_emits():
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_rust_ffi.py"), "--check"])


def test_wrappers_call_only_declared_entry_points():
    """Every ffi::sylow_hip_* the safe layer calls exists in ffi.rs, with the argument count the declaration has."""
    r = parse_ffi()
    used = 0
    for fn in ("lib.rs", "device.rs", "surface.rs"):
        text = open(os.path.join(ROOT, "rust", "sylow-hip", "src", fn)).read()
        for m in re.finditer(r"ffi::(sylow_hip_\w+)\s*\(", text):
            name = m.group(1)
            assert name in r, (fn, name)
            # argument count of the call (top-level commas up to the matching parenthesis)
            i, depth, commas, any_arg = m.end(), 1, 0, False
            while depth:
                c = text[i]
                depth += c in "([{"
                depth -= c in ")]}"
                commas += (c == "," and depth == 1)
                any_arg |= (not c.isspace()) and depth >= 1 and c != ")"
                i += 1
            argc = (commas + 1) if any_arg else 0
            assert argc == len(r[name][1]), (fn, name, argc, len(r[name][1]))
            used += 1
    assert used >= 85


def test_every_entry_point_is_reached_by_a_wrapper():
    """north_star: the host side stays Rust.  Every entry point of include/sylow_hip.h must be called (or passed as a function item) by
    the safe layer -- lib.rs, device.rs or surface.rs -- so that a new C entry point cannot ship without its typed Rust form."""
    h = parse_header()
    text = "".join(open(os.path.join(ROOT, "rust", "sylow-hip", "src", fn)).read() for fn in ("lib.rs", "device.rs", "surface.rs"))
    reached = set(re.findall(r"ffi::(sylow_hip_\w+)", text))
    missing = sorted(set(h) - reached)
    assert not missing, missing
    assert reached <= set(h), sorted(reached - set(h))
