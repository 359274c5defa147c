"""include/sylow_hip.h's machine-readable array shapes (`/* @shape ... */`, grammar in tools/gen_shape_annotations.py) and the check the
Python layer runs on every call (sylow_amd/_shapes.py).  CPU: every array-taking entry point is annotated, every expression only uses
that prototype's own integer parameters, the checker accepts right sizes and rejects wrong ones.  GPU: a deliberately short buffer is
refused before the launch."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sylow_amd import _shapes                                      # noqa: E402
from test_rust_ffi import parse_header                              # noqa: E402

UNANNOTATED_OK = {"sylow_hip_init_devices", "sylow_hip_malloc", "sylow_hip_free", "sylow_hip_memcpy_h2d", "sylow_hip_memcpy_d2h",
                  "sylow_hip_host_malloc", "sylow_hip_host_free", "sylow_hip_stream_sync"}     # raw memory plumbing: sizes are the byte counts


def test_every_array_entry_point_is_annotated():
    protos, table = parse_header(), _shapes.parse()
    assert len(table) >= 100
    for name, (_, params) in protos.items():
        arrays = [p for p in params if p[1] and p[3] != "stream"]
        if not arrays or name in UNANNOTATED_OK:
            continue
        assert name in table, f"{name} takes arrays but has no @shape line"
        names, shapes = table[name]
        assert names == [p[3] for p in params], name
        assert set(shapes) == {p[3] for p in arrays}, (name, sorted(shapes), [p[3] for p in arrays])
        ints = {p[3] for p in params if not p[1]}
        for sh in shapes.values():
            if sh.expr != "*":
                used = set(__import__("re").findall(r"[A-Za-z_]\w*", sh.expr))
                assert used <= ints, (name, sh.param, sh.expr)
                assert sh.nbytes({k: 3 for k in ints}) > 0


def test_header_is_what_the_generator_writes():
    """include/sylow_hip.h carries exactly the @shape lines tools/gen_shape_annotations.py renders (nobody edited one by hand, none is stale)"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_shapes", os.path.join(ROOT, "tools", "gen_shape_annotations.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    text = open(gen.HDR).read()
    again, n = gen.render(text)
    assert again == text and n == len(_shapes.parse())


def test_checker_accepts_and_rejects():
    live = {0x1000: 8 * 8 * 5, 0x2000: 16 * 8 * 5, 0x3000: 48 * 8 * 5, 0x4000: 5}
    ok = [0x1000, 0x4000, 0x2000, None, 0x3000, 5]
    _shapes.check_call("sylow_hip_pairing_batch", ok, live)
    _shapes.check_call("sylow_hip_pairing_batch", [0x1000, None, 0x2000, None, 0x3000, 4], live)        # larger than needed is fine
    _shapes.check_call("sylow_hip_pairing_batch", [0x1008, None, 0x2000, None, 0x3000, 5], live)        # an offset pointer is not checkable
    with pytest.raises(ValueError, match="gt_out holds"):
        _shapes.check_call("sylow_hip_pairing_batch", [0x1000, None, 0x2000, None, 0x3000, 6], {**live, 0x1000: 1 << 20, 0x2000: 1 << 20})
    with pytest.raises(ValueError, match="p_inf holds"):
        _shapes.check_call("sylow_hip_pairing_batch", [0x1000, 0x4000, 0x2000, None, 0x3000, 5], {**live, 0x4000: 4})
    with pytest.raises(ValueError, match="must not be NULL"):
        _shapes.check_call("sylow_hip_pairing_batch", [0x1000, None, None, None, 0x3000, 5], live)
    # multi-parameter expressions: multi_pairing_batch's offsets are n_jobs + 1 words, its points n_pairs
    live2 = {0x10: 8 * 8 * 6, 0x20: 16 * 8 * 6, 0x30: 8 * 4, 0x40: 48 * 8 * 3, 0x50: 3}
    _shapes.check_call("sylow_hip_multi_pairing_batch", [0x10, None, 0x20, None, 0x30, 3, 6, 1, 0x40, 0x50], live2)
    with pytest.raises(ValueError, match="pair_offsets holds"):
        _shapes.check_call("sylow_hip_multi_pairing_batch", [0x10, None, 0x20, None, 0x30, 4, 6, 1, None, 0x50], {**live2, 0x50: 4})


@pytest.mark.gpu
def test_short_buffer_is_refused_before_the_launch(engine):
    import sylow_amd
    n = 64
    p, q = engine.empty((8, n)), engine.empty((16, n))
    gt_short = engine.empty((48, n - 1))
    with pytest.raises(sylow_amd._lib.SylowHipError, match="gt_out holds"):
        engine._call("sylow_hip_pairing_batch", p.ptr, None, q.ptr, None, gt_short.ptr, n)
    sig, sigi = engine.empty((8, n)), engine.empty((n // 2,), np.uint8)
    sk, msgs, off = engine.empty((4, n)), engine.empty((n,), np.uint8), engine.empty((n + 1,))
    with pytest.raises(sylow_amd._lib.SylowHipError, match="sig_inf holds"):
        engine._call("sylow_hip_bls_sign_batch", sk.ptr, msgs.ptr, off.ptr, sig.ptr, sigi.ptr, n)
    freed = engine.empty((48, n))
    ptr = freed.ptr
    freed.free()
    assert ptr not in engine._live
