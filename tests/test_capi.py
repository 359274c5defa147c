"""CPU tests of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/sylow_hip.h declares, and fails loudly (no CPU fallback) when there is no GPU."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared():
    with open(os.path.join(ROOT, "include", "sylow_hip.h")) as f:
        return sorted(set(re.findall(r"\b(sylow_hip_[a-z0-9_]+)\s*\(", f.read())))


def test_library_exports_every_declared_symbol():
    import sylow_amd
    from sylow_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        sylow_amd.build()
    lib = sylow_amd.load()
    names = declared()
    assert len(names) >= 35
    for n in names:
        assert hasattr(lib, n), n
    # and the Python binding table covers the same set
    assert set(names) == set(_lib.SIGNATURES), set(names) ^ set(_lib.SIGNATURES)


def test_product_package_does_not_import_oracle():
    """The product path must never route through the oracle (or any CPU fallback)."""
    pkg = os.path.join(ROOT, "sylow_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hpp", ".hip", ".h")):
                text = open(os.path.join(dirpath, fn)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, re.M), os.path.join(dirpath, fn)
                assert "libsylow_oracle" not in text and "sylow_oracle.c" not in text, os.path.join(dirpath, fn)
                assert "pyref" not in text and "coracle" not in text, os.path.join(dirpath, fn)


def test_no_gpu_fails_loudly():
    import sylow_amd
    lib = sylow_amd.load()
    if lib.sylow_hip_device_count() > 0:
        pytest.skip("a GPU is present")
    assert lib.sylow_hip_init(0) < 0
    assert b"no HIP device" in lib.sylow_hip_last_error()
    with pytest.raises(sylow_amd.SylowHipError):
        sylow_amd.Engine(0)


def test_argument_errors_are_reported_not_thrown():
    import sylow_amd
    lib = sylow_amd.load()
    rc = lib.sylow_hip_fp_mul_batch(None, None, None, 4, None)
    assert rc == -2 and b"bad argument" in lib.sylow_hip_last_error()
    assert lib.sylow_hip_fp_mul_batch(None, None, None, 0, None) == -2


def test_header_is_plain_c(tmp_path):
    """include/sylow_hip.h is the drop-in boundary: it must compile as C99 with no C++ / HIP / torch types."""
    import subprocess
    src = tmp_path / "hdr.c"
    src.write_text('#include "sylow_hip.h"\nint main(void) { return 0; }\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only",
                           "-I", os.path.join(ROOT, "include"), str(src)])


def test_host_xoshiro_matches_the_test_stream():
    """sylow_hip_host_xoshiro_fp (bench inputs) is the generator of tests/helpers.py (oracle inputs): BASELINE.md §3."""
    import numpy as np

    import sylow_amd
    from helpers import SEED, Xoshiro, limbs
    lib = sylow_amd.load()
    for seed, n in ((SEED, 300), (SEED + 3, 1), (0, 17), ((1 << 64) - 1, 64)):
        out = np.zeros((4, n + 2), dtype=np.uint64)
        assert lib.sylow_hip_host_xoshiro_fp(seed, out.ctypes.data, n, n + 2) == 0
        r = Xoshiro(seed)
        assert np.array_equal(out[:, :n].T, limbs([r.fp() for _ in range(n)]))
        assert not out[:, n:].any()                                    # the stride tail is left alone
    assert lib.sylow_hip_host_xoshiro_fp(1, None, 4, 4) == -2


def test_environment_switches_are_the_documented_ones():
    """Every variable the library reads is a route selector named in INTEGRATION.md and forced by tests/test_gpu_routes.py (or a
    threshold named there); nothing undocumented changes what the library computes with."""
    import glob
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    read = set()
    for f in glob.glob(os.path.join(root, "sylow_amd", "csrc", "*.h*")):
        src = open(f).read()
        read |= set(re.findall(r'getenv\("([A-Z0-9_]+)"\)', src)) | set(re.findall(r'env_size\("([A-Z0-9_]+)"\)', src))
    assert read == {"SYLOW_HIP_MULTI_TABLES", "SYLOW_HIP_WIDE_TAIL", "SYLOW_HIP_WIDE_PACK", "SYLOW_HIP_AGG_FORK", "SYLOW_HIP_STAGGER",
                    "SYLOW_HIP_SIGN_WIDE_MAX", "SYLOW_HIP_WIDE_MAX", "SYLOW_HIP_WIDE_VERIFY_MAX"}
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    routes = open(os.path.join(root, "tests", "test_gpu_routes.py")).read()
    for name in read:
        assert name in doc, name
        assert name in routes, name
