"""Writes the machine-readable `@shape` line in front of every array-taking prototype of include/sylow_hip.h (idempotent: existing @shape
lines are replaced).  Grammar, one comment per prototype, placed directly before it:

    /* @shape p_xy=u64[8*n] p_inf=u8[n]? gt_out=u64[48*n] msgs=u8[*] */

name=dtype[expr]: the array holds at least expr elements of dtype; expr is an integer expression over the prototype's own size_t / int32
parameters; `?` = the pointer may be NULL; `[*]` = length not expressible from the arguments (message blobs, opaque tables).  HOST arrays
of the value-typed calls are annotated the same way.  Consumers: sylow_amd/_shapes.py (checked on every Engine._call),
tests/test_shapes.py, tests/test_rust_ffi.py (the Rust wrappers' allocations)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "include", "sylow_hip.h")

W = {"fp": 4, "fr": 4, "fp2": 8, "fp6": 24, "fp12": 48}
MSG = "msgs=u8[*] msg_offsets=u64[n+1]"


def field(prefix, names):                      # a / b / out of width W[prefix]
    return " ".join(f"{x}=u64[{W[prefix]}*n]" for x in names)


def grp(g, proj=False):
    return {"g1": 12 if proj else 8, "g2": 24 if proj else 16}[g]


S = {}
for f in ("fp", "fr"):
    for op in ("add", "sub", "mul"):
        S[f"{f}_{op}_batch"] = field(f, "a b out".split())
    for op in ("sqr", "neg", "inv"):
        S[f"{f}_{op}_batch"] = field(f, "a out".split())
    S[f"{f}_from_be_bytes_batch"] = "in=u8[32*n] out=u64[4*n] status=u8[n]"
    S[f"{f}_to_be_bytes_batch"] = "a=u64[4*n] out=u8[32*n]"
S["fp_pow_batch"] = "a=u64[4*n] e=u64[4*n] out=u64[4*n]"
S["fp_sqrt_batch"] = "a=u64[4*n] out=u64[4*n] is_some=u8[n]"
S["fp_is_square_batch"] = "a=u64[4*n] flags=u8[n]"
S["fp_compute_naf_batch"] = "k=u64[4*n] out_np=u64[4*n] out_nm=u64[4*n]"
for op in ("add", "sub"):
    S[f"fext_{op}_batch"] = "a=u64[4*degree*n] b=u64[4*degree*n] out=u64[4*degree*n]"
S["fext_neg_batch"] = "a=u64[4*degree*n] out=u64[4*degree*n]"
S["fext_scale_batch"] = "a=u64[4*degree*n] k=u64[4*n] out=u64[4*degree*n]"
for f in ("fp2", "fp6", "fp12"):
    S[f"{f}_mul_batch"] = field(f, "a b out".split())
    for op in ("sqr", "inv", "residue_mul", "frobenius"):
        S[f"{f}_{op}_batch"] = field(f, "a out".split())
del S["fp12_residue_mul_batch"]
S["fp12_cyclotomic_sqr_batch"] = field("fp12", "a out".split())
S["fp12_sparse_mul_batch"] = "f=u64[48*n] ell=u64[24*n] out=u64[48*n]"
S["f29_hook_batch"] = "a=u64[*] b=u64[*]? out=u64[*]"
S["fp12_hook_batch"] = "a=u64[48*n] b=u64[*]? out=u64[48*n]"
S["aos_to_soa"] = "aos=u64[words*n] soa=u64[words*n]"
S["soa_to_aos"] = "soa=u64[words*n] aos=u64[words*n]"
S["host_xoshiro_fp"] = "out_host=u64[3*stride+n]"
for g in ("g1", "g2"):
    w, wp = grp(g), grp(g, True)
    S[f"{g}_scalar_mul_batch"] = f"p_xy=u64[{w}*n] p_inf=u8[n]? k=u64[4*n] out_xy=u64[{w}*n] out_inf=u8[n]"
    S[f"{g}_generator_mul_batch"] = f"k=u64[4*n] out_xy=u64[{w}*n] out_inf=u8[n]"
    for op in ("add", "sub"):
        S[f"{g}_{op}_batch"] = f"a_xy=u64[{w}*n] a_inf=u8[n]? b_xy=u64[{w}*n] b_inf=u8[n]? out_xy=u64[{w}*n] out_inf=u8[n]"
    S[f"{g}_double_batch"] = f"a_xy=u64[{w}*n] a_inf=u8[n]? out_xy=u64[{w}*n] out_inf=u8[n]"
    S[f"{g}_projective_new_batch"] = f"p_xyz=u64[{wp}*n] status=u8[n]"
    S[f"{g}_ct_eq_batch"] = f"a_xyz=u64[{wp}*n] b_xyz=u64[{wp}*n] eq=u8[n]"
    S[f"{g}_normalize_batch"] = f"p_xyz=u64[{wp}*n] out_xy=u64[{w}*n] out_inf=u8[n]"
    S[f"{g}_to_be_bytes_batch"] = f"p_xy=u64[{w}*n] p_inf=u8[n]? out=u8[{8 * w}*n]"
    S[f"{g}_from_be_bytes_batch"] = f"in=u8[{8 * w}*n] out_xy=u64[{w}*n] out_inf=u8[n] status=u8[n]"
S["g2_scalar_mul_subgroup_batch"] = S["g2_scalar_mul_batch"]
S["g1_lincomb_batch"] = "p_xy=u64[8*n_jobs*n_terms] p_inf=u8[n_jobs*n_terms]? k=u64[4*n_jobs*n_terms] out_xy=u64[8*n_jobs] out_inf=u8[n_jobs]"
S["g1_sum_batch"] = "p_xy=u64[8*n] p_inf=u8[n]? out_xy=u64[8] out_inf=u8[1]"
S["g1_on_curve_batch"] = "p_xy=u64[8*n] p_inf=u8[n]? status=u8[n]"
S["g2_psi_batch"] = "q_xy=u64[16*n] q_inf=u8[n]? out_xy=u64[16*n] out_inf=u8[n] status=u8[n]"
S["g2_subgroup_check_batch"] = "q_xy=u64[16*n] q_inf=u8[n]? status=u8[n]"
S["gt_pow_batch"] = "gt=u64[48*n] k=u64[4*n] out=u64[48*n]"
S["miller_loop_batch"] = "p_xy=u64[8*n] q_xy=u64[16*n] f_out=u64[48*n]"
S["final_exp_batch"] = "f=u64[48*n] gt_out=u64[48*n]"
S["pairing_batch"] = "p_xy=u64[8*n] p_inf=u8[n]? q_xy=u64[16*n] q_inf=u8[n]? gt_out=u64[48*n]"
PAIRS = "p_xy=u64[8*n_pairs]? p_inf=u8[n_pairs]? q_xy=u64[16*n_pairs]? q_inf=u8[n_pairs]?"
S["multi_pairing_batch"] = PAIRS + " pair_offsets=u64[n_jobs+1] gt_out=u64[48*n_jobs]? is_one=u8[n_jobs]?"
S["glued_miller_loop_batch"] = "p_xy=u64[8*n_pairs]? q_xy=u64[16*n_pairs]? pair_offsets=u64[n_jobs+1] f_out=u64[48*n_jobs]"
S["pairing_product_batch"] = PAIRS + " gt_out=u64[48]? is_one=u8[1]?"
S["pairing_product_partial_batch"] = PAIRS + " f_out=u64[48]"
S["pairing_product_all"] = PAIRS + " comm=void[*]? gt_out=u64[48]? is_one=u8[1]?"
S["fp12_product_final_exp"] = "parts=u64[48*k] gt_out=u64[48]? is_one=u8[1]?"
S["g2_precompute_batch"] = "q_xy=u64[16*n] coeffs=u64[87*24*n]"
S["miller_loop_precomputed_batch"] = "coeffs=u64[87*24*n_tables] table_idx=u64[n]? p_xy=u64[8*n] f_out=u64[48*n]"
S["glued_miller_loop_precomputed_batch"] = "coeffs=u64[87*24*n_tables]? table_idx=u64[n_pairs]? p_xy=u64[8*n_pairs]? pair_offsets=u64[n_jobs+1] f_out=u64[48*n_jobs]"
S["hash_to_field_batch"] = MSG + " dst_host=u8[dst_len]? out_u=u64[8*n]"
S["hash_to_g1_batch"] = MSG + " dst_host=u8[dst_len]? out_xy=u64[8*n] out_inf=u8[n]"
S["svdw_map_batch"] = "u=u64[4*n] out_xy=u64[8*n] status=u8[n]"
S["bls_sign_batch"] = "sk=u64[4*n] " + MSG + " sig_xy=u64[8*n] sig_inf=u8[n]"
VER = "pk_xy=u64[16*n] pk_inf=u8[n]? " + MSG + " sig_xy=u64[8*n] sig_inf=u8[n]? ok=u8[n]"
for v in ("bls_verify_batch", "bls_verify_fused_batch", "bls_verify_two_pairings_batch"):
    S[v] = VER
S["bls_verify_same_signer_batch"] = "pk_xy=u64[16] pk_inf=u8[1]? " + MSG + " sig_xy=u64[8*n] sig_inf=u8[n]? ok=u8[n]"
S["g2_line_table"] = "q_xy=u64[16*n]? table=i32[*]"
S["bls_verify_line_table_batch"] = "pk_table=i32[*] pk_inf=u8[1]? " + MSG + " sig_xy=u64[8*n] sig_inf=u8[n]? ok=u8[n]"
S["evm_ecadd_batch"] = "in=u8[128*n] out=u8[64*n] status=u8[n]"
S["evm_ecmul_batch"] = "in=u8[96*n] out=u8[64*n] status=u8[n]"
S["evm_ecpairing_batch"] = "in=u8[192*n_pairs]? pair_offsets=u64[n_jobs+1] result=u8[n_jobs] status=u8[n_jobs]"
S["get_option"] = "value_host=i64[1]"
S["clock_probe"] = "acc=u64[256]?"
S["wall_clock_khz"] = "khz_host=i32[1]"
S["flags_all"] = "flags=u8[n] out_dev=i32[1]"
S["all_valid"] = "flags=u8[n] comm=void[*]? out_dev=i32[1]"
AGG = "pk_xy=u64[16*n_pk] pk_inf=u8[n_pk]? " + MSG + " sig_xy=u64[8*n] sig_inf=u8[n]?"
S["bls_aggregate_partial_batch"] = AGG + " f_out=u64[48]"
S["bls_aggregate_verify_batch"] = AGG + " comm=void[*]? gt_out=u64[48]? is_one=u8[1]?"
S["bls_weighted_partial_batch"] = AGG + " weights=u64[4*n] f_out=u64[48]"
S["bls_batch_verify_weighted"] = AGG + " weights=u64[4*n] comm=void[*]? gt_out=u64[48]? is_one=u8[1]?"
S["pairing_host"] = "p_aos=u64[8*n] p_inf=u8[n]? q_aos=u64[16*n] q_inf=u8[n]? gt_aos=u64[48*n]"
S["bls_verify_host"] = "pk_aos=u64[16*n] pk_inf=u8[n]? msgs=u8[*]? msg_offsets=u64[n+1] sig_aos=u64[8*n] sig_inf=u8[n]? ok=u8[n]"
S["pairing_host_bytes"] = "p_be=u8[64*n] q_be=u8[128*n] gt_aos=u64[48*n] status_p=u8[n] status_q=u8[n]"
S["bls_verify_host_bytes"] = "pk_be=u8[128*n] msgs=u8[*]? msg_offsets=u64[n+1] sig_be=u8[64*n] ok=u8[n] status_pk=u8[n] status_sig=u8[n]"


def render(text):
    """header text -> header text with exactly one @shape line in front of every prototype listed in S (idempotent)"""
    text = re.sub(r"/\* @shape [^\n]*? \*/\n", "", text)
    done = set()

    def repl(m):
        name = m.group(2)[len("sylow_hip_"):]
        if name in S:
            done.add(name)
            return f"/* @shape {S[name]} */\n" + m.group(0)
        return m.group(0)
    text = re.sub(r"^(int32_t)\s+(sylow_hip_\w+)\s*\(", repl, text, flags=re.M)
    missing = set(S) - done
    assert not missing, missing
    return text, len(done)


def main():
    text, n = render(open(HDR).read())
    open(HDR, "w").write(text)
    print(f"{n} prototypes annotated")


if __name__ == "__main__":
    main()
