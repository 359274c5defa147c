"""CPU test of the Rust side of the boundary (rust/sylow-hip, source only -- no Rust toolchain in this image): src/ffi.rs must
declare exactly the entry points of include/sylow_hip.h, with the same arity, the same pointer depth and const-ness, and the
matching scalar types.  Both files are parsed here independently of the generator (tools/gen_rust_ffi.py)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
C2RUST = {"int32_t": "i32", "uint64_t": "u64", "int64_t": "i64", "size_t": "usize", "uint8_t": "u8", "char": "c_char", "void": "c_void"}


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


# ---- buffer sizes: every `dev.alloc::<T>(EXPR)` that reaches an ffi call against the header's @shape lines ---------------------------
def _split_top(s, sep=","):
    """split at top-level separators (ignores those inside (), [], {}, <>)"""
    out, depth, cur = [], 0, ""
    for i, ch in enumerate(s):
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        elif ch == "<" and cur.endswith("::"):
            depth += 1
        elif ch == ">" and depth and "::<" in cur and cur.count("<") > cur.count(">"):
            depth -= 1
        if ch == sep and depth == 0:
            out.append(cur.strip()); cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def _rust_functions(text):
    """(name, body) of every `fn` in a Rust source file (brace matching)"""
    for m in re.finditer(r"\bfn\s+(\w+)[^{;]*\{", text):
        i, depth = m.end(), 1
        while depth and i < len(text):
            depth += {"{": 1, "}": -1}.get(text[i], 0)
            i += 1
        yield m.group(1), text[m.end():i]


def _allocations(body):
    """{binding or binding.field: (T, EXPR)} for `let x = dev.alloc::<T>(E)?`, tuple lets and struct literals"""
    allocs = {}
    alloc_re = r"dev\.alloc::<(\w+)>\(((?:[^()]|\([^()]*\))*)\)\?"
    for m in re.finditer(r"let\s+(?:mut\s+)?(\w+)\s*=\s*" + alloc_re, body):
        allocs[m.group(1)] = (m.group(2), m.group(3))
    for m in re.finditer(r"let\s+\(([^)]*)\)\s*=\s*\((.*?)\);", body, flags=re.S):
        names, vals = [x.strip() for x in m.group(1).split(",")], _split_top(m.group(2))
        if len(names) == len(vals):
            for nm, v in zip(names, vals):
                mm = re.fullmatch(alloc_re, v)
                if mm:
                    allocs[nm] = (mm.group(1), mm.group(2))
    for m in re.finditer(r"let\s+(?:mut\s+)?(\w+)\s*=\s*\w+\s*\{([^{}]*)\}", body):
        for fld in _split_top(m.group(2)):
            mm = re.fullmatch(r"(\w+)\s*:\s*" + alloc_re, fld)
            if mm:
                allocs[f"{m.group(1)}.{mm.group(1)}"] = (mm.group(2), mm.group(3))
    return allocs


def _numeric(expr, env):
    """value of a size expression with every identifier-like atom (`n`, `W`, `q.len()`, `p.n`) replaced by a number from env"""
    expr = re.sub(r"([A-Za-z_][\w.]*?)\.max\((\d+)\)", r"max(\1, \2)", expr)              # `n_jobs.max(1)`: never an empty allocation
    atoms = sorted(set(re.findall(r"[A-Za-z_][\w.]*(?:\(\))?", expr)) - {"max"}, key=len, reverse=True)
    for a in atoms:
        if a not in env:
            env[a] = 1000003 + 7919 * len(env)               # distinct large primes-ish: no accidental equalities
        expr = re.sub(r"(?<![\w.])" + re.escape(a) + r"(?![\w(])", str(env[a]), expr)
    return eval(expr, {"__builtins__": {}}, {"max": max})


def rust_size_mismatches(sources, shapes, protos):
    """[(function, entry point, parameter, rust expr, header expr)] for every allocation passed to an ffi call whose size differs"""
    from sylow_amd._shapes import ITEMSIZE
    rust_size = {"u64": 8, "u8": 1, "i32": 4, "u32": 4}
    bad, checked = [], 0
    for text in sources:
        for fname, body in _rust_functions(text):
            allocs = _allocations(body)
            if not allocs:
                continue
            for m in re.finditer(r"ffi::(sylow_hip_\w+)\(((?:[^()]|\((?:[^()]|\([^()]*\))*\))*)\)", body):
                ent = shapes.get(m.group(1))
                if ent is None:
                    continue
                names, shp = ent
                args = _split_top(" ".join(m.group(2).split()))
                if len(args) != len(names):
                    continue
                actual = {nm: a for nm, a in zip(names, args) if nm not in shp}
                for nm, a in zip(names, args):
                    mm = re.fullmatch(r"([\w.]+)\.as_(?:mut_)?ptr\(\)", a)
                    if nm not in shp or not mm or mm.group(1) not in allocs or shp[nm].expr == "*":
                        continue
                    ty, expr = allocs[mm.group(1)]
                    env = {}
                    header_expr = re.sub(r"\b([A-Za-z_]\w*)\b", lambda t: "(" + actual.get(t.group(1), t.group(1)) + ")", shp[nm].expr)
                    have = _numeric(expr, env) * rust_size[ty]
                    need = _numeric(header_expr, env) * ITEMSIZE[shp[nm].dtype]
                    checked += 1
                    if have < need or (have != need and ".max(" not in expr):
                        bad.append((fname, m.group(1), nm, f"{ty}[{expr}]", f"{shp[nm].dtype}[{shp[nm].expr}]"))
    return bad, checked


def _rust_sources():
    base = os.path.join(ROOT, "rust", "sylow-hip", "src")
    return [open(os.path.join(base, f)).read() for f in ("lib.rs", "surface.rs", "device.rs")]


def test_rust_allocations_match_header_shapes():
    """Every buffer a Rust wrapper allocates and hands to an entry point has exactly the size include/sylow_hip.h's @shape line gives for
    the size arguments of that very call (the unsafe blocks' SAFETY comments, checked by a tool instead of by eye)."""
    sys.path.insert(0, ROOT)
    from sylow_amd import _shapes
    bad, checked = rust_size_mismatches(_rust_sources(), _shapes.parse(), parse_header())
    assert checked >= 70, checked
    assert not bad, bad


def test_rust_allocation_check_catches_a_wrong_size():
    """The same check on a deliberately broken copy of lib.rs: `dev.alloc::<u64>(48 * n)` for pairing_batch's gt shrunk to 24 * n."""
    sys.path.insert(0, ROOT)
    from sylow_amd import _shapes
    src = _rust_sources()
    broken = src[0].replace("let gt = dev.alloc::<u64>(48 * n)?;", "let gt = dev.alloc::<u64>(24 * n)?;", 1)
    assert broken != src[0]
    bad, _ = rust_size_mismatches([broken], _shapes.parse(), parse_header())
    assert any(b[1] == "sylow_hip_pairing_batch" and b[2] == "gt_out" for b in bad), bad
