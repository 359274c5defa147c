"""The machine-readable array shapes of include/sylow_hip.h (`/* @shape name=dtype[expr] ... */` in front of a prototype; grammar in
tools/gen_shape_annotations.py) and the check the Python layer runs on every call: each pointer argument that is the base address of a
live DeviceArray must be at least as large as the header says for the call's size arguments -- a wrong allocation fails BEFORE the
launch, with the entry point, the parameter and both sizes in the message, instead of a kernel writing past a buffer."""
from __future__ import annotations

import operator
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(_HERE), "include", "sylow_hip.h")
ITEMSIZE = {"u64": 8, "i64": 8, "u8": 1, "i32": 4, "void": 1}
_EXPR_OK = re.compile(r"^[\w\s+*()]+$")


class Shape:
    __slots__ = ("param", "dtype", "expr", "optional")

    def __init__(self, param, dtype, expr, optional):
        self.param, self.dtype, self.expr, self.optional = param, dtype, expr, optional

    def elements(self, values):
        """minimum element count for the call's integer arguments (None = not expressible: `[*]`)"""
        if self.expr == "*":
            return None
        return int(eval(self.expr, {"__builtins__": {}}, values))      # expr is [\w+*() ] only (checked at parse time)

    def nbytes(self, values):
        n = self.elements(values)
        return None if n is None else n * ITEMSIZE[self.dtype]


def parse(path: str = HEADER):
    """-> {entry point: (parameter names in order, {parameter: Shape})} for every annotated prototype"""
    text = open(path).read()
    out = {}
    for m in re.finditer(r"/\* @shape ([^\n]*?) \*/\s*\n\s*int32_t\s+(sylow_hip_\w+)\s*\(([^)]*)\)\s*;", text):
        params = re.sub(r"/\*.*?\*/", " ", m.group(3))                  # inline remarks like `uint8_t* out /*[n][64]*/`
        names = [p.replace("*", " ").split()[-1] for p in " ".join(params.split()).split(",")]
        shapes = {}
        for item in m.group(1).split():
            mm = re.fullmatch(r"(\w+)=(\w+)\[([^\]]+)\](\??)", item)
            if not mm:
                raise ValueError(f"{m.group(2)}: bad @shape item {item!r}")
            param, dtype, expr, opt = mm.groups()
            if param not in names or dtype not in ITEMSIZE or not (expr == "*" or _EXPR_OK.match(expr)):
                raise ValueError(f"{m.group(2)}: bad @shape item {item!r}")
            shapes[param] = Shape(param, dtype, expr, bool(opt))
        out[m.group(2)] = (names, shapes)
    return out


_TABLE = None


def table():
    global _TABLE
    if _TABLE is None:
        try:
            _TABLE = parse()
        except OSError:            # the package deployed without the repo's include/ directory: no table, nothing to check
            _TABLE = {}
    return _TABLE


def check_call(name, args, live):
    """args: the call's arguments WITHOUT the trailing stream; live: {base pointer: nbytes} of the engine's DeviceArrays.  Raises ValueError."""
    ent = table().get(name)
    if ent is None:
        return
    names, shapes = ent
    values = {}
    for n, a in zip(names, args):
        if n in shapes or a is None or isinstance(a, (bytes, bool)):
            continue
        try:
            values[n] = operator.index(a)                  # Python ints and numpy integers alike
        except TypeError:
            pass
    for n, a in zip(names, args):
        sh = shapes.get(n)
        if sh is None:
            continue
        if a is None or a == 0:
            try:
                empty = sh.nbytes(values) == 0                   # NULL is fine where the call needs zero elements (an empty batch)
            except NameError:
                empty = True
            if not sh.optional and not empty:
                raise ValueError(f"{name}: {n} must not be NULL")
            continue
        have = live.get(a) if isinstance(a, int) else None
        if have is None:
            continue                                       # not the base of one of our arrays (an offset pointer, a torch tensor): not checkable
        try:
            need = sh.nbytes(values)
        except NameError:
            continue                                       # a size argument of the expression was not an integer: not checkable
        if need is not None and have < need:
            raise ValueError(f"{name}: {n} holds {have} bytes, the call needs {sh.dtype}[{sh.expr}] = {need} bytes")
