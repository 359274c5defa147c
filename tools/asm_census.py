"""Instruction census per function of a saved ISA file (make asm UNIT=...): multiply-adds, moves, DPP, other VALU, calls, scratch accesses,
code size, registers.  usage: asm_census.py [unit] [name-filter ...]"""
import collections, re, sys
unit = sys.argv[1] if len(sys.argv) > 1 else "plk_pairing"
flt = sys.argv[2:]
lines = open(f"/tmp/sylow_asm/{unit}-hip-amdgcn-amd-amdhsa-gfx950.s").read().split("\n")
funcs = [(i, m.group(1)) for i, l in enumerate(lines) for m in [re.match(r"\s*\.type\s+(\S+),@function", l)] if m]
funcs.append((len(lines), "END"))
for (a, name), (b, _) in zip(funcs, funcs[1:]):
    if flt and not any(f in name for f in flt):
        continue
    c = collections.Counter()
    for l in lines[a:b]:
        t = l.strip().split()
        if not t or t[0][0] in ".;" or t[0].endswith(":"):
            continue
        op = t[0]
        if op.startswith(("v_mad_i64", "v_mad_u64")): c["mad"] += 1
        elif op.startswith("v_mov_b32_e32"): c["mov"] += 1
        elif "_dpp" in op: c["dpp"] += 1
        elif op.startswith("v_"): c["valu"] += 1
        elif op.startswith("s_swappc"): c["call"] += 1
        elif op.startswith("scratch_"): c["scratch"] += 1
        elif op.startswith("ds_"): c["lds"] += 1
        elif op.startswith("s_"): c["salu"] += 1
        else: c["other"] += 1
    meta = {k: v for l in lines[a:b] for k, v in re.findall(r";\s*(codeLenInByte|NumVgprs|ScratchSize)[ =:]+(\d+)", l)}
    print(name[:70].ljust(70), dict(c), meta)
