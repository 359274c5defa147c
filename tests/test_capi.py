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


def test_library_reads_no_environment_variable_and_options_are_documented():
    """Round 6: the route selectors are sylow_hip_set_option values, not getenv switches behind the ABI.  The C++ / HIP sources read no
    environment variable at all; the Python host layer maps SYLOW_HIP_<NAME> onto the options when it loads the library, and every option is
    named in the header, in INTEGRATION.md and forced (or moved) by tests/test_gpu_routes.py."""
    import glob
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for f in glob.glob(os.path.join(root, "sylow_amd", "csrc", "*.h*")):
        src = re.sub(r"//[^\n]*", " ", open(f).read())
        assert "getenv" not in src and "environ" not in src, f
    from sylow_amd import _lib
    hdr = open(os.path.join(root, "include", "sylow_hip.h")).read()
    defs = dict((m.group(1), int(m.group(2))) for m in re.finditer(r"#define SYLOW_HIP_OPT_([A-Z_]+) (\d+)", hdr))
    count = defs.pop("COUNT")
    assert defs == _lib.OPTIONS and sorted(defs.values()) == list(range(count))
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    routes = open(os.path.join(root, "tests", "test_gpu_routes.py")).read()
    for name in defs:
        assert "SYLOW_HIP_" + name in doc, name
        assert "SYLOW_HIP_" + name in routes, name


def test_options_round_trip_without_a_gpu():
    import ctypes

    import sylow_amd
    from sylow_amd import _lib
    lib = sylow_amd.load()
    v = ctypes.c_int64(7)
    for name, opt in _lib.OPTIONS.items():
        assert lib.sylow_hip_get_option(opt, ctypes.byref(v)) == 0
        before = v.value
        assert lib.sylow_hip_set_option(opt, 12345) == 0
        assert lib.sylow_hip_get_option(opt, ctypes.byref(v)) == 0 and v.value == 12345
        assert lib.sylow_hip_set_option(opt, 0) == 0
        assert lib.sylow_hip_get_option(opt, ctypes.byref(v)) == 0 and v.value == 0
        assert lib.sylow_hip_set_option(opt, before) == 0                      # -1 (default) unless the environment set it
        assert lib.sylow_hip_get_option(opt, ctypes.byref(v)) == 0 and v.value == before
    assert lib.sylow_hip_set_option(len(_lib.OPTIONS), 1) == -2 and lib.sylow_hip_set_option(-1, 1) == -2
    assert lib.sylow_hip_get_option(0, None) == -2
    assert lib.sylow_hip_clock_probe(None) == 0                                # switching the probe off needs no device
