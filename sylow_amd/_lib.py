"""ctypes binding of libsylow_hip.so (the C ABI in include/sylow_hip.h).

Fails loudly: there is no CPU fallback anywhere in this package.  If the shared library is
missing or no gfx950 device is present, every compute entry point raises.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SYLOW_HIP_LIB") or os.path.join(_HERE, "libsylow_hip.so")   # override: A/B builds
_lib = None

c_u64p = ctypes.c_void_p
c_u8p = ctypes.c_void_p
c_sz = ctypes.c_size_t
c_i32 = ctypes.c_int32
c_vp = ctypes.c_void_p

# name -> argtypes (restype is int32 unless listed in _RESTYPE)
SIGNATURES = {
    "sylow_hip_init": [c_i32],
    "sylow_hip_init_devices": [ctypes.POINTER(c_i32), c_i32],
    "sylow_hip_set_device": [c_i32],
    "sylow_hip_shutdown": [],
    "sylow_hip_last_error": [],
    "sylow_hip_device_count": [],
    "sylow_hip_malloc": [ctypes.POINTER(c_vp), c_sz],
    "sylow_hip_free": [c_vp],
    "sylow_hip_memcpy_h2d": [c_vp, c_vp, c_sz, c_vp],
    "sylow_hip_memcpy_d2h": [c_vp, c_vp, c_sz, c_vp],
    "sylow_hip_stream_sync": [c_vp],
    "sylow_hip_host_xoshiro_fp": [ctypes.c_uint64, c_vp, c_sz, c_sz],
    "sylow_hip_aos_to_soa": [c_u64p, c_u64p, c_sz, c_sz, c_vp],
    "sylow_hip_soa_to_aos": [c_u64p, c_u64p, c_sz, c_sz, c_vp],
    "sylow_hip_fp_add_batch": [c_u64p, c_u64p, c_u64p, c_sz, c_vp],
    "sylow_hip_fp_sub_batch": [c_u64p, c_u64p, c_u64p, c_sz, c_vp],
    "sylow_hip_fp_mul_batch": [c_u64p, c_u64p, c_u64p, c_sz, c_vp],
    "sylow_hip_fp_sqr_batch": [c_u64p, c_u64p, c_sz, c_vp],
    "sylow_hip_fp_neg_batch": [c_u64p, c_u64p, c_sz, c_vp],
    "sylow_hip_fp_inv_batch": [c_u64p, c_u64p, c_sz, c_vp],
    "sylow_hip_fp2_mul_batch": [c_u64p, c_u64p, c_u64p, c_sz, c_vp],
    "sylow_hip_fp2_sqr_batch": [c_u64p, c_u64p, c_sz, c_vp],
    "sylow_hip_fp2_inv_batch": [c_u64p, c_u64p, c_sz, c_vp],
    "sylow_hip_fp6_mul_batch": [c_u64p, c_u64p, c_u64p, c_sz, c_vp],
    "sylow_hip_fp6_inv_batch": [c_u64p, c_u64p, c_sz, c_vp],
    "sylow_hip_fp12_mul_batch": [c_u64p, c_u64p, c_u64p, c_sz, c_vp],
    "sylow_hip_fp12_sqr_batch": [c_u64p, c_u64p, c_sz, c_vp],
    "sylow_hip_fp12_inv_batch": [c_u64p, c_u64p, c_sz, c_vp],
    "sylow_hip_fp12_frobenius_batch": [c_u64p, c_i32, c_u64p, c_sz, c_vp],
    "sylow_hip_fp2_residue_mul_batch": [c_u64p, c_u64p, c_sz, c_vp],
    "sylow_hip_fp2_frobenius_batch": [c_u64p, ctypes.c_uint64, c_u64p, c_sz, c_vp],
    "sylow_hip_fp6_sqr_batch": [c_u64p, c_u64p, c_sz, c_vp],
    "sylow_hip_fp6_residue_mul_batch": [c_u64p, c_u64p, c_sz, c_vp],
    "sylow_hip_fp6_frobenius_batch": [c_u64p, ctypes.c_uint64, c_u64p, c_sz, c_vp],
    "sylow_hip_fp12_sparse_mul_batch": [c_u64p, c_u64p, c_u64p, c_sz, c_vp],
    "sylow_hip_g1_scalar_mul_batch": [c_u64p, c_u8p, c_u64p, c_u64p, c_u8p, c_sz, c_vp],
    "sylow_hip_g2_scalar_mul_batch": [c_u64p, c_u8p, c_u64p, c_u64p, c_u8p, c_sz, c_vp],
    "sylow_hip_g2_generator_mul_batch": [c_u64p, c_u64p, c_u8p, c_sz, c_vp],
    "sylow_hip_g1_generator_mul_batch": [c_u64p, c_u64p, c_u8p, c_sz, c_vp],
    "sylow_hip_g2_scalar_mul_subgroup_batch": [c_u64p, c_u8p, c_u64p, c_u64p, c_u8p, c_sz, c_vp],
    "sylow_hip_g1_add_batch": [c_u64p, c_u8p, c_u64p, c_u8p, c_u64p, c_u8p, c_sz, c_vp],
    "sylow_hip_g1_normalize_batch": [c_u64p, c_u64p, c_u8p, c_sz, c_vp],
    "sylow_hip_g2_normalize_batch": [c_u64p, c_u64p, c_u8p, c_sz, c_vp],
    "sylow_hip_g2_subgroup_check_batch": [c_u64p, c_u8p, c_u8p, c_sz, c_vp],
    "sylow_hip_miller_loop_batch": [c_u64p, c_u64p, c_u64p, c_sz, c_vp],
    "sylow_hip_final_exp_batch": [c_u64p, c_u64p, c_sz, c_vp],
    "sylow_hip_pairing_batch": [c_u64p, c_u8p, c_u64p, c_u8p, c_u64p, c_sz, c_vp],
    "sylow_hip_multi_pairing_batch": [c_u64p, c_u8p, c_u64p, c_u8p, c_u64p, c_sz, c_sz, c_i32, c_u64p, c_u8p, c_vp],
    "sylow_hip_fext_add_batch": [c_u64p, c_u64p, c_u64p, ctypes.c_int32, c_sz, c_vp],
    "sylow_hip_fext_sub_batch": [c_u64p, c_u64p, c_u64p, ctypes.c_int32, c_sz, c_vp],
    "sylow_hip_fext_neg_batch": [c_u64p, c_u64p, ctypes.c_int32, c_sz, c_vp],
    "sylow_hip_fext_scale_batch": [c_u64p, c_u64p, c_u64p, ctypes.c_int32, c_sz, c_vp],
    "sylow_hip_svdw_map_batch": [c_u64p, c_u64p, c_u8p, c_sz, c_vp],
    "sylow_hip_fp_compute_naf_batch": [c_u64p, c_u64p, c_u64p, c_sz, c_vp],
    "sylow_hip_hash_to_field_batch": [c_u8p, c_u64p, ctypes.c_char_p, c_sz, c_u64p, c_sz, c_vp],
    "sylow_hip_hash_to_g1_batch": [c_u8p, c_u64p, ctypes.c_char_p, c_sz, c_u64p, c_u8p, c_sz, c_vp],
    "sylow_hip_bls_sign_batch": [c_u64p, c_u8p, c_u64p, c_u64p, c_u8p, c_sz, c_vp],
    "sylow_hip_bls_verify_batch": [c_u64p, c_u8p, c_u8p, c_u64p, c_u64p, c_u8p, c_u8p, c_sz, c_vp],
    "sylow_hip_bls_verify_two_pairings_batch": [c_u64p, c_u8p, c_u8p, c_u64p, c_u64p, c_u8p, c_u8p, c_sz, c_vp],
    "sylow_hip_bls_verify_fused_batch": [c_u64p, c_u8p, c_u8p, c_u64p, c_u64p, c_u8p, c_u8p, c_sz, c_vp],
    "sylow_hip_evm_ecadd_batch": [c_u8p, c_u8p, c_u8p, c_sz, c_vp],
    "sylow_hip_evm_ecmul_batch": [c_u8p, c_u8p, c_u8p, c_sz, c_vp],
    "sylow_hip_evm_ecpairing_batch": [c_u8p, c_u64p, c_sz, c_sz, c_u8p, c_u8p, c_vp],
    "sylow_hip_g1_to_be_bytes_batch": [c_u64p, c_u8p, c_u8p, c_sz, c_vp],
    "sylow_hip_g1_from_be_bytes_batch": [c_u8p, c_u64p, c_u8p, c_u8p, c_sz, c_vp],
    "sylow_hip_g2_to_be_bytes_batch": [c_u64p, c_u8p, c_u8p, c_sz, c_vp],
    "sylow_hip_g2_from_be_bytes_batch": [c_u8p, c_u64p, c_u8p, c_u8p, c_sz, c_vp],
    "sylow_hip_bls_aggregate_partial_batch": [c_u64p, c_u8p, c_sz, c_u8p, c_u64p, c_u64p, c_u8p, c_sz, c_u64p, c_vp],
    "sylow_hip_bls_aggregate_verify_batch": [c_u64p, c_u8p, c_sz, c_u8p, c_u64p, c_u64p, c_u8p, c_sz, c_vp, c_u64p, c_u8p, c_vp],
    "sylow_hip_bls_weighted_partial_batch": [c_u64p, c_u8p, c_sz, c_u8p, c_u64p, c_u64p, c_u8p, c_u64p, c_sz, c_u64p, c_vp],
    "sylow_hip_bls_batch_verify_weighted": [c_u64p, c_u8p, c_sz, c_u8p, c_u64p, c_u64p, c_u8p, c_u64p, c_sz, c_vp, c_u64p, c_u8p, c_vp],
    "sylow_hip_bls_verify_same_signer_batch": [c_u64p, c_u8p, c_u8p, c_u64p, c_u64p, c_u8p, c_u8p, c_sz, c_vp],
    "sylow_hip_g2_precompute_batch": [c_u64p, c_u64p, c_sz, c_vp],
    "sylow_hip_f29_hook_batch": [c_i32, c_u64p, c_u64p, c_u64p, c_sz, c_vp],
    "sylow_hip_fp_pow_batch": [c_u64p, c_u64p, c_u64p, c_sz, c_vp],
    "sylow_hip_fp_sqrt_batch": [c_u64p, c_u64p, c_u8p, c_sz, c_vp],
    "sylow_hip_fp_is_square_batch": [c_u64p, c_u8p, c_sz, c_vp],
    "sylow_hip_fr_add_batch": [c_u64p, c_u64p, c_u64p, c_sz, c_vp],
    "sylow_hip_fr_sub_batch": [c_u64p, c_u64p, c_u64p, c_sz, c_vp],
    "sylow_hip_fr_mul_batch": [c_u64p, c_u64p, c_u64p, c_sz, c_vp],
    "sylow_hip_fr_sqr_batch": [c_u64p, c_u64p, c_sz, c_vp],
    "sylow_hip_fr_neg_batch": [c_u64p, c_u64p, c_sz, c_vp],
    "sylow_hip_fr_inv_batch": [c_u64p, c_u64p, c_sz, c_vp],
    "sylow_hip_g1_lincomb_batch": [c_u64p, c_u8p, c_u64p, c_u64p, c_u8p, c_sz, c_sz, c_vp],
    "sylow_hip_glued_miller_loop_batch": [c_u64p, c_u64p, c_u64p, c_sz, c_sz, c_u64p, c_vp],
    "sylow_hip_pairing_product_batch": [c_u64p, c_u8p, c_u64p, c_u8p, c_sz, c_i32, c_u64p, c_u8p, c_vp],
    "sylow_hip_g1_on_curve_batch": [c_u64p, c_u8p, c_u8p, c_sz, c_vp],
    "sylow_hip_g2_psi_batch": [c_u64p, c_u8p, c_u64p, c_u8p, c_u8p, c_sz, c_vp],
    "sylow_hip_gt_pow_batch": [c_u64p, c_u64p, c_u64p, c_sz, c_vp],
    "sylow_hip_g2_add_batch": [c_u64p, c_u8p, c_u64p, c_u8p, c_u64p, c_u8p, c_sz, c_vp],
    "sylow_hip_g1_sub_batch": [c_u64p, c_u8p, c_u64p, c_u8p, c_u64p, c_u8p, c_sz, c_vp],
    "sylow_hip_g2_sub_batch": [c_u64p, c_u8p, c_u64p, c_u8p, c_u64p, c_u8p, c_sz, c_vp],
    "sylow_hip_g1_projective_new_batch": [c_u64p, c_u8p, c_sz, c_vp],
    "sylow_hip_g2_projective_new_batch": [c_u64p, c_u8p, c_sz, c_vp],
    "sylow_hip_g1_ct_eq_batch": [c_u64p, c_u64p, c_u8p, c_sz, c_vp],
    "sylow_hip_g2_ct_eq_batch": [c_u64p, c_u64p, c_u8p, c_sz, c_vp],
    "sylow_hip_g1_double_batch": [c_u64p, c_u8p, c_u64p, c_u8p, c_sz, c_vp],
    "sylow_hip_g2_double_batch": [c_u64p, c_u8p, c_u64p, c_u8p, c_sz, c_vp],
    "sylow_hip_flags_all": [c_u8p, c_sz, c_vp, c_vp],
    "sylow_hip_fp_from_be_bytes_batch": [c_u8p, c_u64p, c_u8p, c_sz, c_vp],
    "sylow_hip_fr_from_be_bytes_batch": [c_u8p, c_u64p, c_u8p, c_sz, c_vp],
    "sylow_hip_fp_to_be_bytes_batch": [c_u64p, c_u8p, c_sz, c_vp],
    "sylow_hip_fr_to_be_bytes_batch": [c_u64p, c_u8p, c_sz, c_vp],
    "sylow_hip_miller_loop_precomputed_batch": [c_u64p, c_sz, c_u64p, c_u64p, c_u64p, c_sz, c_vp],
    "sylow_hip_glued_miller_loop_precomputed_batch": [c_u64p, c_sz, c_u64p, c_u64p, c_u64p, c_sz, c_sz, c_u64p, c_vp],
    "sylow_hip_pairing_product_partial_batch": [c_u64p, c_u8p, c_u64p, c_u8p, c_sz, c_i32, c_u64p, c_vp],
    "sylow_hip_fp12_product_final_exp": [c_u64p, c_sz, c_u64p, c_u8p, c_vp],
    "sylow_hip_g2_line_table_words": [],
    "sylow_hip_g2_line_table": [c_u64p, c_sz, c_sz, c_vp, c_vp],
    "sylow_hip_bls_verify_line_table_batch": [c_vp, c_u8p, c_u8p, c_u64p, c_u64p, c_u8p, c_u8p, c_sz, c_vp],
    "sylow_hip_all_valid": [c_u8p, c_sz, c_vp, c_vp, c_vp],
    "sylow_hip_pairing_product_all": [c_u64p, c_u8p, c_u64p, c_u8p, c_sz, c_i32, c_vp, c_u64p, c_u8p, c_vp],
    "sylow_hip_fp12_cyclotomic_sqr_batch": [c_u64p, c_u64p, c_sz, c_vp],
    "sylow_hip_fp12_hook_batch": [c_i32, c_u64p, c_u64p, c_u64p, c_sz, c_vp],
    "sylow_hip_trim": [c_sz],
    "sylow_hip_set_scratch_limit": [c_sz],
    "sylow_hip_set_option": [c_i32, ctypes.c_int64],
    "sylow_hip_get_option": [c_i32, ctypes.POINTER(ctypes.c_int64)],
    "sylow_hip_clock_probe": [c_u64p],
    "sylow_hip_wall_clock_khz": [ctypes.POINTER(c_i32)],
    "sylow_hip_g1_sum_batch": [c_u64p, c_u8p, c_sz, c_u64p, c_u8p, c_vp],
    "sylow_hip_pairing_host": [c_u64p, c_u8p, c_u64p, c_u8p, c_u64p, c_sz, c_sz],
    "sylow_hip_bls_verify_host": [c_u64p, c_u8p, c_u8p, c_u64p, c_u64p, c_u8p, c_u8p, c_sz, c_sz],
    "sylow_hip_pairing_host_bytes": [c_u8p, c_u8p, c_u64p, c_u8p, c_u8p, c_sz, c_sz],
    "sylow_hip_bls_verify_host_bytes": [c_u8p, c_u8p, c_u64p, c_u8p, c_u8p, c_u8p, c_u8p, c_sz, c_sz],
    "sylow_hip_host_malloc": [ctypes.POINTER(ctypes.c_void_p), c_sz],
    "sylow_hip_host_free": [c_vp],
}
_RESTYPE = {"sylow_hip_last_error": ctypes.c_char_p}

# SYLOW_HIP_OPT_* of include/sylow_hip.h.  The LIBRARY reads no environment variable; this host layer does, once, when it loads the library:
# SYLOW_HIP_<NAME>=<integer> becomes sylow_hip_set_option(<NAME>, value) -- what tests/test_gpu_routes.py and the A/B scripts under tools/ set
OPTIONS = {"STAGGER": 0, "MULTI_TABLES": 1, "WIDE_TAIL": 2, "WIDE_PACK": 3, "AGG_FORK": 4, "SIGN_WIDE_MAX": 5, "WIDE_MAX": 6, "WIDE_VERIFY_MAX": 7,
           "QUAD_MAX": 8, "TAIL_SPLIT": 9}


class SylowHipError(RuntimeError):
    pass


def build(force: bool = False) -> str:
    """Compile libsylow_hip.so for gfx950 with hipcc (works without a GPU)."""
    csrc = os.path.join(_HERE, "csrc")
    args = ["make", "-C", csrc]
    if force:
        args.append("-B")
    subprocess.check_call(args)
    return LIB_PATH


def _preload_torch_hip_runtime():
    """One HIP / HSA runtime per process.  A PyTorch-ROCm wheel ships its own libamdhip64.so.7 + libhsa-runtime64 (same sonames
    as /opt/rocm's); if libsylow_hip.so pulled in the system copies first and torch (or torch's RCCL, which dlopens
    libhsa-runtime64.so by file name) arrived later, the process would hold two HSA runtimes and the second one sees
    "no ROCm-capable device".  So when torch is installed its runtime is loaded first -- whether or not torch is ever imported --
    and both share it.  Hosts without torch (C, C++, Rust) get the system runtime.  SYLOW_HIP_SYSTEM_RUNTIME=1 skips this."""
    import importlib.util
    import sys
    if "torch" in sys.modules or os.environ.get("SYLOW_HIP_SYSTEM_RUNTIME") == "1":
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.submodule_search_locations:
        return
    path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(path):
        ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)


def load():
    """Load the shared library and declare every prototype.  Does not touch the GPU."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SylowHipError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is no CPU fallback)")
        _preload_torch_hip_runtime()
        lib = ctypes.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the symbol is missing
            fn.argtypes = argtypes
            fn.restype = _RESTYPE.get(name, c_i32)
        for name, opt in OPTIONS.items():
            v = os.environ.get("SYLOW_HIP_" + name)
            if v is not None and v.strip() != "":
                try:
                    value = int(v)
                except ValueError:
                    raise SylowHipError(f"SYLOW_HIP_{name}={v!r}: not an integer") from None
                if lib.sylow_hip_set_option(opt, value) != 0:
                    raise SylowHipError(f"sylow_hip_set_option({name}) failed")
        _lib = lib
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().sylow_hip_last_error()
        raise SylowHipError(f"{what} failed with {rc}: {msg.decode() if msg else ''}")
