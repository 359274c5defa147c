//! The rest of sylow's trait surface as batches (north_star: `GroupTrait`, `FieldExtensionTrait`): every entry point of
//! include/sylow_hip.h that lib.rs does not already wrap, as a thin typed function over `ffi::`.
//!
//! NOT COMPILED IN THE AUTHORING ENVIRONMENT (see lib.rs).  tests/test_rust_ffi.py checks every `ffi::` call below against the
//! generated declarations (name, arity) and fails when a header entry point is reached by no wrapper.
//!
//! Field elements cross as canonical little-endian words (`Fp::value().to_words()`, fields/fp.rs:232-234) and come back through
//! `Fp::new(U256::from_words(..))`; extension-field elements are `[u64; 8 / 24 / 48]` in the reference's nesting order (the types'
//! coefficients are private upstream, so the word arrays are the stable interchange form -- `fp2_words` etc. in lib.rs rebuild
//! sylow values from them where constructors exist).
use crate::device::{self, Device, DeviceBuf};
use crate::{ffi, first_failure, fp_from_words, messages, DeviceG1, DeviceG2, GtOut, HipError};
use crate::{download_g1, download_g2, gt_from_words, upload_g1, upload_g2};
use std::os::raw::c_void;
use std::ptr;
use sylow::{Fp, Fr, G1Affine, G1Projective, G2Affine, G2Projective};

type BinOp = unsafe extern "C" fn(*const u64, *const u64, *mut u64, usize, *mut c_void) -> i32;
type UnOp = unsafe extern "C" fn(*const u64, *mut u64, usize, *mut c_void) -> i32;

fn fp_words(a: &[Fp]) -> Vec<[u64; 4]> {
    a.iter().map(|x| x.value().to_words()).collect()
}
fn fr_words(a: &[Fr]) -> Vec<[u64; 4]> {
    a.iter().map(|x| x.value().to_words()).collect()
}

fn binop<const W: usize>(dev: &Device, f: BinOp, a: &[[u64; W]], b: &[[u64; W]]) -> Result<Vec<[u64; W]>, HipError> {
    assert_eq!(a.len(), b.len());
    let n = a.len();
    let (da, db, out) = (dev.upload_soa::<W>(a)?, dev.upload_soa::<W>(b)?, dev.alloc::<u64>(W * n)?);
    // SAFETY: three SoA arrays of W * n words each.
    device::check(unsafe { f(da.as_ptr(), db.as_ptr(), out.as_mut_ptr(), n, dev.stream) })?;
    Ok(dev.download_aos::<W>(&out, n)?)
}
fn unop<const W: usize>(dev: &Device, f: UnOp, a: &[[u64; W]]) -> Result<Vec<[u64; W]>, HipError> {
    let n = a.len();
    let (da, out) = (dev.upload_soa::<W>(a)?, dev.alloc::<u64>(W * n)?);
    // SAFETY: two SoA arrays of W * n words each.
    device::check(unsafe { f(da.as_ptr(), out.as_mut_ptr(), n, dev.stream) })?;
    Ok(dev.download_aos::<W>(&out, n)?)
}
fn to_fp(w: Vec<[u64; 4]>) -> Vec<Fp> {
    w.iter().map(|x| fp_from_words(x)).collect()
}

// ------------------------------------------------------------------ Fp / Fr: the operator bounds of FieldExtensionTrait (fp.rs:97-131)
/// Batched `Add / Sub / Mul for Fp` (fp.rs:304-347, 414-422), `square` (fp.rs:620-622), `Neg` (fp.rs:442-449), `inv` (fp.rs:418-433, inv(0) = 0).
pub fn fp_add_batch(dev: &Device, a: &[Fp], b: &[Fp]) -> Result<Vec<Fp>, HipError> {
    Ok(to_fp(binop::<4>(dev, ffi::sylow_hip_fp_add_batch, &fp_words(a), &fp_words(b))?))
}
pub fn fp_sub_batch(dev: &Device, a: &[Fp], b: &[Fp]) -> Result<Vec<Fp>, HipError> {
    Ok(to_fp(binop::<4>(dev, ffi::sylow_hip_fp_sub_batch, &fp_words(a), &fp_words(b))?))
}
pub fn fp_mul_batch(dev: &Device, a: &[Fp], b: &[Fp]) -> Result<Vec<Fp>, HipError> {
    Ok(to_fp(binop::<4>(dev, ffi::sylow_hip_fp_mul_batch, &fp_words(a), &fp_words(b))?))
}
pub fn fp_square_batch(dev: &Device, a: &[Fp]) -> Result<Vec<Fp>, HipError> {
    Ok(to_fp(unop::<4>(dev, ffi::sylow_hip_fp_sqr_batch, &fp_words(a))?))
}
pub fn fp_neg_batch(dev: &Device, a: &[Fp]) -> Result<Vec<Fp>, HipError> {
    Ok(to_fp(unop::<4>(dev, ffi::sylow_hip_fp_neg_batch, &fp_words(a))?))
}
pub fn fp_inv_batch(dev: &Device, a: &[Fp]) -> Result<Vec<Fp>, HipError> {
    Ok(to_fp(unop::<4>(dev, ffi::sylow_hip_fp_inv_batch, &fp_words(a))?))
}
/// `Fp::pow(U256)` (fp.rs:451-457); exponents as canonical words.
pub fn fp_pow_batch(dev: &Device, a: &[Fp], e: &[[u64; 4]]) -> Result<Vec<Fp>, HipError> {
    Ok(to_fp(binop::<4>(dev, ffi::sylow_hip_fp_pow_batch, &fp_words(a), e)?))
}
/// `Fp::sqrt` (fp.rs:611-616) as (candidate, is_some) and `Fp::is_square` (fp.rs:625-631).
pub fn fp_sqrt_batch(dev: &Device, a: &[Fp]) -> Result<Vec<Option<Fp>>, HipError> {
    let n = a.len();
    let (da, out, some) = (dev.upload_soa::<4>(&fp_words(a))?, dev.alloc::<u64>(4 * n)?, dev.alloc::<u8>(n)?);
    // SAFETY: 4 * n words in and out, n flags.
    device::check(unsafe { ffi::sylow_hip_fp_sqrt_batch(da.as_ptr(), out.as_mut_ptr(), some.as_mut_ptr(), n, dev.stream) })?;
    let (w, s) = (dev.download_aos::<4>(&out, n)?, dev.download(&some)?);
    Ok(w.iter().zip(s).map(|(x, ok)| if ok != 0 { Some(fp_from_words(x)) } else { None }).collect())
}
pub fn fp_is_square_batch(dev: &Device, a: &[Fp]) -> Result<Vec<bool>, HipError> {
    let n = a.len();
    let (da, flags) = (dev.upload_soa::<4>(&fp_words(a))?, dev.alloc::<u8>(n)?);
    // SAFETY: 4 * n words, n flags.
    device::check(unsafe { ffi::sylow_hip_fp_is_square_batch(da.as_ptr(), flags.as_mut_ptr(), n, dev.stream) })?;
    Ok(dev.download(&flags)?.iter().map(|&f| f != 0).collect())
}
/// `Fp::compute_naf` (fp.rs:653-662): the two 256-bit digit masks (plus, minus) of every scalar.
pub fn fp_compute_naf_batch(dev: &Device, k: &[Fp]) -> Result<Vec<([u64; 4], [u64; 4])>, HipError> {
    let n = k.len();
    let (dk, np, nm) = (dev.upload_soa::<4>(&fp_words(k))?, dev.alloc::<u64>(4 * n)?, dev.alloc::<u64>(4 * n)?);
    // SAFETY: three arrays of 4 * n words.
    device::check(unsafe { ffi::sylow_hip_fp_compute_naf_batch(dk.as_ptr(), np.as_mut_ptr(), nm.as_mut_ptr(), n, dev.stream) })?;
    Ok(dev.download_aos::<4>(&np, n)?.into_iter().zip(dev.download_aos::<4>(&nm, n)?).collect())
}
fn from_be_bytes(dev: &Device, fr: bool, blobs: &[[u8; 32]]) -> Result<Vec<Option<[u64; 4]>>, HipError> {
    let n = blobs.len();
    let flat: Vec<u8> = blobs.iter().flatten().copied().collect();
    let (d_in, out, st) = (dev.upload(&flat)?, dev.alloc::<u64>(4 * n)?, dev.alloc::<u8>(n)?);
    // SAFETY: 32 * n bytes in, 4 * n words and n status bytes out.
    device::check(unsafe {
        if fr {
            ffi::sylow_hip_fr_from_be_bytes_batch(d_in.as_ptr(), out.as_mut_ptr(), st.as_mut_ptr(), n, dev.stream)
        } else {
            ffi::sylow_hip_fp_from_be_bytes_batch(d_in.as_ptr(), out.as_mut_ptr(), st.as_mut_ptr(), n, dev.stream)
        }
    })?;
    let (w, s) = (dev.download_aos::<4>(&out, n)?, dev.download(&st)?);
    Ok(w.into_iter().zip(s).map(|(x, bad)| if bad == 0 { Some(x) } else { None }).collect())
}
/// `Fp::from_be_bytes` / `Fr::from_be_bytes` (fp.rs:686-719, 746-778): `None` where the value is not below the modulus.
pub fn fp_from_be_bytes_batch(dev: &Device, blobs: &[[u8; 32]]) -> Result<Vec<Option<Fp>>, HipError> {
    Ok(from_be_bytes(dev, false, blobs)?.into_iter().map(|o| o.map(|w| fp_from_words(&w))).collect())
}
pub fn fr_from_be_bytes_batch(dev: &Device, blobs: &[[u8; 32]]) -> Result<Vec<Option<[u64; 4]>>, HipError> {
    from_be_bytes(dev, true, blobs)
}
fn to_be_bytes(dev: &Device, fr: bool, words: &[[u64; 4]]) -> Result<Vec<[u8; 32]>, HipError> {
    let n = words.len();
    let (da, out) = (dev.upload_soa::<4>(words)?, dev.alloc::<u8>(32 * n)?);
    // SAFETY: 4 * n words in, 32 * n bytes out.
    device::check(unsafe {
        if fr {
            ffi::sylow_hip_fr_to_be_bytes_batch(da.as_ptr(), out.as_mut_ptr(), n, dev.stream)
        } else {
            ffi::sylow_hip_fp_to_be_bytes_batch(da.as_ptr(), out.as_mut_ptr(), n, dev.stream)
        }
    })?;
    Ok(dev.download(&out)?.chunks_exact(32).map(|c| c.try_into().unwrap()).collect())
}
/// `Fp::to_be_bytes` / `Fr::to_be_bytes` (fp.rs:727-737).
pub fn fp_to_be_bytes_batch(dev: &Device, a: &[Fp]) -> Result<Vec<[u8; 32]>, HipError> {
    to_be_bytes(dev, false, &fp_words(a))
}
pub fn fr_to_be_bytes_batch(dev: &Device, a: &[Fr]) -> Result<Vec<[u8; 32]>, HipError> {
    to_be_bytes(dev, true, &fr_words(a))
}
/// Fr arithmetic (fp.rs:556-565; Lagrange coefficients of examples/threshold_signing.rs:124-155), canonical words out.
pub fn fr_add_batch(dev: &Device, a: &[Fr], b: &[Fr]) -> Result<Vec<[u64; 4]>, HipError> {
    binop::<4>(dev, ffi::sylow_hip_fr_add_batch, &fr_words(a), &fr_words(b))
}
pub fn fr_sub_batch(dev: &Device, a: &[Fr], b: &[Fr]) -> Result<Vec<[u64; 4]>, HipError> {
    binop::<4>(dev, ffi::sylow_hip_fr_sub_batch, &fr_words(a), &fr_words(b))
}
pub fn fr_mul_batch(dev: &Device, a: &[Fr], b: &[Fr]) -> Result<Vec<[u64; 4]>, HipError> {
    binop::<4>(dev, ffi::sylow_hip_fr_mul_batch, &fr_words(a), &fr_words(b))
}
pub fn fr_square_batch(dev: &Device, a: &[Fr]) -> Result<Vec<[u64; 4]>, HipError> {
    unop::<4>(dev, ffi::sylow_hip_fr_sqr_batch, &fr_words(a))
}
pub fn fr_neg_batch(dev: &Device, a: &[Fr]) -> Result<Vec<[u64; 4]>, HipError> {
    unop::<4>(dev, ffi::sylow_hip_fr_neg_batch, &fr_words(a))
}
pub fn fr_inv_batch(dev: &Device, a: &[Fr]) -> Result<Vec<[u64; 4]>, HipError> {
    unop::<4>(dev, ffi::sylow_hip_fr_inv_batch, &fr_words(a))
}

// ------------------------------------------------------------------ FieldExtension<D, N, F> (extensions.rs:41-238) and the tower
/// Component-wise `Add / Sub / Neg` and `scale(Fp)` of `FieldExtension` for degree 2, 6 or 12 (W = 4 * degree words per element).
pub fn fext_add_batch<const W: usize>(dev: &Device, a: &[[u64; W]], b: &[[u64; W]]) -> Result<Vec<[u64; W]>, HipError> {
    assert_eq!(a.len(), b.len());
    let n = a.len();
    let (da, db, out) = (dev.upload_soa::<W>(a)?, dev.upload_soa::<W>(b)?, dev.alloc::<u64>(W * n)?);
    // SAFETY: three SoA arrays of W * n words; degree = W / 4.
    device::check(unsafe { ffi::sylow_hip_fext_add_batch(da.as_ptr(), db.as_ptr(), out.as_mut_ptr(), (W / 4) as i32, n, dev.stream) })?;
    Ok(dev.download_aos::<W>(&out, n)?)
}
pub fn fext_sub_batch<const W: usize>(dev: &Device, a: &[[u64; W]], b: &[[u64; W]]) -> Result<Vec<[u64; W]>, HipError> {
    assert_eq!(a.len(), b.len());
    let n = a.len();
    let (da, db, out) = (dev.upload_soa::<W>(a)?, dev.upload_soa::<W>(b)?, dev.alloc::<u64>(W * n)?);
    // SAFETY: as fext_add_batch.
    device::check(unsafe { ffi::sylow_hip_fext_sub_batch(da.as_ptr(), db.as_ptr(), out.as_mut_ptr(), (W / 4) as i32, n, dev.stream) })?;
    Ok(dev.download_aos::<W>(&out, n)?)
}
pub fn fext_neg_batch<const W: usize>(dev: &Device, a: &[[u64; W]]) -> Result<Vec<[u64; W]>, HipError> {
    let n = a.len();
    let (da, out) = (dev.upload_soa::<W>(a)?, dev.alloc::<u64>(W * n)?);
    // SAFETY: two SoA arrays of W * n words.
    device::check(unsafe { ffi::sylow_hip_fext_neg_batch(da.as_ptr(), out.as_mut_ptr(), (W / 4) as i32, n, dev.stream) })?;
    Ok(dev.download_aos::<W>(&out, n)?)
}
pub fn fext_scale_batch<const W: usize>(dev: &Device, a: &[[u64; W]], k: &[Fp]) -> Result<Vec<[u64; W]>, HipError> {
    assert_eq!(a.len(), k.len());
    let n = a.len();
    let (da, dk, out) = (dev.upload_soa::<W>(a)?, dev.upload_soa::<4>(&fp_words(k))?, dev.alloc::<u64>(W * n)?);
    // SAFETY: W * n words in and out, 4 * n scalar words.
    device::check(unsafe { ffi::sylow_hip_fext_scale_batch(da.as_ptr(), dk.as_ptr(), out.as_mut_ptr(), (W / 4) as i32, n, dev.stream) })?;
    Ok(dev.download_aos::<W>(&out, n)?)
}
/// Fp2 (fp2.rs): `Mul`, `square`, `inv`, `residue_mul` (x (9 + u), :99-107), `frobenius(e)` (:119-133).
pub fn fp2_mul_batch(dev: &Device, a: &[[u64; 8]], b: &[[u64; 8]]) -> Result<Vec<[u64; 8]>, HipError> {
    binop::<8>(dev, ffi::sylow_hip_fp2_mul_batch, a, b)
}
pub fn fp2_square_batch(dev: &Device, a: &[[u64; 8]]) -> Result<Vec<[u64; 8]>, HipError> {
    unop::<8>(dev, ffi::sylow_hip_fp2_sqr_batch, a)
}
pub fn fp2_inv_batch(dev: &Device, a: &[[u64; 8]]) -> Result<Vec<[u64; 8]>, HipError> {
    unop::<8>(dev, ffi::sylow_hip_fp2_inv_batch, a)
}
pub fn fp2_residue_mul_batch(dev: &Device, a: &[[u64; 8]]) -> Result<Vec<[u64; 8]>, HipError> {
    unop::<8>(dev, ffi::sylow_hip_fp2_residue_mul_batch, a)
}
pub fn fp2_frobenius_batch(dev: &Device, a: &[[u64; 8]], exponent: usize) -> Result<Vec<[u64; 8]>, HipError> {
    let n = a.len();
    let (da, out) = (dev.upload_soa::<8>(a)?, dev.alloc::<u64>(8 * n)?);
    // SAFETY: two SoA arrays of 8 * n words.
    device::check(unsafe { ffi::sylow_hip_fp2_frobenius_batch(da.as_ptr(), exponent as u64, out.as_mut_ptr(), n, dev.stream) })?;
    Ok(dev.download_aos::<8>(&out, n)?)
}
/// Fp6 (fp6.rs): `Mul`, `square` (:213-236), `inv`, `residue_mul` (x v, :189-192), `frobenius(e)` (:205-211).
pub fn fp6_mul_batch(dev: &Device, a: &[[u64; 24]], b: &[[u64; 24]]) -> Result<Vec<[u64; 24]>, HipError> {
    binop::<24>(dev, ffi::sylow_hip_fp6_mul_batch, a, b)
}
pub fn fp6_square_batch(dev: &Device, a: &[[u64; 24]]) -> Result<Vec<[u64; 24]>, HipError> {
    unop::<24>(dev, ffi::sylow_hip_fp6_sqr_batch, a)
}
pub fn fp6_inv_batch(dev: &Device, a: &[[u64; 24]]) -> Result<Vec<[u64; 24]>, HipError> {
    unop::<24>(dev, ffi::sylow_hip_fp6_inv_batch, a)
}
pub fn fp6_residue_mul_batch(dev: &Device, a: &[[u64; 24]]) -> Result<Vec<[u64; 24]>, HipError> {
    unop::<24>(dev, ffi::sylow_hip_fp6_residue_mul_batch, a)
}
pub fn fp6_frobenius_batch(dev: &Device, a: &[[u64; 24]], exponent: usize) -> Result<Vec<[u64; 24]>, HipError> {
    let n = a.len();
    let (da, out) = (dev.upload_soa::<24>(a)?, dev.alloc::<u64>(24 * n)?);
    // SAFETY: two SoA arrays of 24 * n words.
    device::check(unsafe { ffi::sylow_hip_fp6_frobenius_batch(da.as_ptr(), exponent as u64, out.as_mut_ptr(), n, dev.stream) })?;
    Ok(dev.download_aos::<24>(&out, n)?)
}
/// Fp12 (fp12.rs): `Mul` (:229-238), `square` (:536-550), `inv` (:281-286), `frobenius(1..=3)` (:240-262), `sparse_mul` (:426-503),
/// and the Granger-Scott `cyclotomic_squared` of pairing.rs:309-350.
pub fn fp12_mul_batch(dev: &Device, a: &[[u64; 48]], b: &[[u64; 48]]) -> Result<Vec<[u64; 48]>, HipError> {
    binop::<48>(dev, ffi::sylow_hip_fp12_mul_batch, a, b)
}
pub fn fp12_square_batch(dev: &Device, a: &[[u64; 48]]) -> Result<Vec<[u64; 48]>, HipError> {
    unop::<48>(dev, ffi::sylow_hip_fp12_sqr_batch, a)
}
pub fn fp12_inv_batch(dev: &Device, a: &[[u64; 48]]) -> Result<Vec<[u64; 48]>, HipError> {
    unop::<48>(dev, ffi::sylow_hip_fp12_inv_batch, a)
}
pub fn fp12_cyclotomic_squared_batch(dev: &Device, a: &[[u64; 48]]) -> Result<Vec<[u64; 48]>, HipError> {
    unop::<48>(dev, ffi::sylow_hip_fp12_cyclotomic_sqr_batch, a)
}
pub fn fp12_frobenius_batch(dev: &Device, a: &[[u64; 48]], exponent: usize) -> Result<Vec<[u64; 48]>, HipError> {
    assert!((1..=3).contains(&exponent));
    let n = a.len();
    let (da, out) = (dev.upload_soa::<48>(a)?, dev.alloc::<u64>(48 * n)?);
    // SAFETY: two SoA arrays of 48 * n words.
    device::check(unsafe { ffi::sylow_hip_fp12_frobenius_batch(da.as_ptr(), exponent as i32, out.as_mut_ptr(), n, dev.stream) })?;
    Ok(dev.download_aos::<48>(&out, n)?)
}
/// `Fp12::sparse_mul(ell_0, ell_vw, ell_vv)`: `ell[i]` = the three Fp2 coefficients as 24 words.
pub fn fp12_sparse_mul_batch(dev: &Device, f: &[[u64; 48]], ell: &[[u64; 24]]) -> Result<Vec<[u64; 48]>, HipError> {
    assert_eq!(f.len(), ell.len());
    let n = f.len();
    let (df, dl, out) = (dev.upload_soa::<48>(f)?, dev.upload_soa::<24>(ell)?, dev.alloc::<u64>(48 * n)?);
    // SAFETY: 48 * n, 24 * n and 48 * n words.
    device::check(unsafe { ffi::sylow_hip_fp12_sparse_mul_batch(df.as_ptr(), dl.as_ptr(), out.as_mut_ptr(), n, dev.stream) })?;
    Ok(dev.download_aos::<48>(&out, n)?)
}
/// Raw selectors of the library's Fp / Fp12 test hooks (parity tests only; see include/sylow_hip.h for the op codes).
pub fn f29_hook_batch(dev: &Device, op: i32, a: &[[u64; 4]], b: &[[u64; 4]]) -> Result<Vec<[u64; 4]>, HipError> {
    assert_eq!(a.len(), b.len());
    let n = a.len();
    let (da, db, out) = (dev.upload_soa::<4>(a)?, dev.upload_soa::<4>(b)?, dev.alloc::<u64>(4 * n)?);
    // SAFETY: three SoA arrays of 4 * n words.
    device::check(unsafe { ffi::sylow_hip_f29_hook_batch(op, da.as_ptr(), db.as_ptr(), out.as_mut_ptr(), n, dev.stream) })?;
    Ok(dev.download_aos::<4>(&out, n)?)
}
pub fn fp12_hook_batch(dev: &Device, op: i32, a: &[[u64; 48]], b: Option<&[[u64; 48]]>) -> Result<Vec<[u64; 48]>, HipError> {
    let n = a.len();
    let da = dev.upload_soa::<48>(a)?;
    let db = match b {
        Some(b) => Some(dev.upload_soa::<48>(b)?),
        None => None,
    };
    let out = dev.alloc::<u64>(48 * n)?;
    // SAFETY: 48 * n words each; the second operand may be absent for unary selectors.
    device::check(unsafe {
        ffi::sylow_hip_fp12_hook_batch(op, da.as_ptr(), db.as_ref().map_or(ptr::null(), |d| d.as_ptr()), out.as_mut_ptr(), n, dev.stream)
    })?;
    Ok(dev.download_aos::<48>(&out, n)?)
}

// ------------------------------------------------------------------ GroupTrait (group.rs:60-164) and the group law
fn g1_pair_op(dev: &Device, which: u8, a: &[G1Affine], b: &[G1Affine]) -> Result<Vec<G1Projective>, HipError> {
    assert_eq!(a.len(), b.len());
    let n = a.len();
    let (da, db) = (upload_g1(dev, a)?, upload_g1(dev, b)?);
    let out = DeviceG1 { xy: dev.alloc::<u64>(8 * n)?, inf: dev.alloc::<u8>(n)?, n };
    // SAFETY: n affine points + flags on each side, n outputs.
    device::check(unsafe {
        if which == 0 {
            ffi::sylow_hip_g1_add_batch(da.xy.as_ptr(), da.inf.as_ptr(), db.xy.as_ptr(), db.inf.as_ptr(), out.xy.as_mut_ptr(), out.inf.as_mut_ptr(), n, dev.stream)
        } else {
            ffi::sylow_hip_g1_sub_batch(da.xy.as_ptr(), da.inf.as_ptr(), db.xy.as_ptr(), db.inf.as_ptr(), out.xy.as_mut_ptr(), out.inf.as_mut_ptr(), n, dev.stream)
        }
    })?;
    download_g1(dev, &out)
}
fn g2_pair_op(dev: &Device, which: u8, a: &[G2Affine], b: &[G2Affine]) -> Result<Vec<G2Projective>, HipError> {
    assert_eq!(a.len(), b.len());
    let n = a.len();
    let (da, db) = (upload_g2(dev, a)?, upload_g2(dev, b)?);
    let out = DeviceG2 { xy: dev.alloc::<u64>(16 * n)?, inf: dev.alloc::<u8>(n)?, n };
    // SAFETY: n affine points + flags on each side, n outputs.
    device::check(unsafe {
        if which == 0 {
            ffi::sylow_hip_g2_add_batch(da.xy.as_ptr(), da.inf.as_ptr(), db.xy.as_ptr(), db.inf.as_ptr(), out.xy.as_mut_ptr(), out.inf.as_mut_ptr(), n, dev.stream)
        } else {
            ffi::sylow_hip_g2_sub_batch(da.xy.as_ptr(), da.inf.as_ptr(), db.xy.as_ptr(), db.inf.as_ptr(), out.xy.as_mut_ptr(), out.inf.as_mut_ptr(), n, dev.stream)
        }
    })?;
    download_g2(dev, &out)
}
/// `Add` / `Sub` for `&G1Projective` and `&G2Projective` (group.rs:528-599, 614-624), `double` (group.rs:339-386).
pub fn g1_add_batch(dev: &Device, a: &[G1Affine], b: &[G1Affine]) -> Result<Vec<G1Projective>, HipError> {
    g1_pair_op(dev, 0, a, b)
}
pub fn g1_sub_batch(dev: &Device, a: &[G1Affine], b: &[G1Affine]) -> Result<Vec<G1Projective>, HipError> {
    g1_pair_op(dev, 1, a, b)
}
pub fn g2_add_batch(dev: &Device, a: &[G2Affine], b: &[G2Affine]) -> Result<Vec<G2Projective>, HipError> {
    g2_pair_op(dev, 0, a, b)
}
pub fn g2_sub_batch(dev: &Device, a: &[G2Affine], b: &[G2Affine]) -> Result<Vec<G2Projective>, HipError> {
    g2_pair_op(dev, 1, a, b)
}
pub fn g1_double_batch(dev: &Device, a: &[G1Affine]) -> Result<Vec<G1Projective>, HipError> {
    let n = a.len();
    let da = upload_g1(dev, a)?;
    let out = DeviceG1 { xy: dev.alloc::<u64>(8 * n)?, inf: dev.alloc::<u8>(n)?, n };
    // SAFETY: n points + flags in, n outputs.
    device::check(unsafe { ffi::sylow_hip_g1_double_batch(da.xy.as_ptr(), da.inf.as_ptr(), out.xy.as_mut_ptr(), out.inf.as_mut_ptr(), n, dev.stream) })?;
    download_g1(dev, &out)
}
pub fn g2_double_batch(dev: &Device, a: &[G2Affine]) -> Result<Vec<G2Projective>, HipError> {
    let n = a.len();
    let da = upload_g2(dev, a)?;
    let out = DeviceG2 { xy: dev.alloc::<u64>(16 * n)?, inf: dev.alloc::<u8>(n)?, n };
    // SAFETY: n points + flags in, n outputs.
    device::check(unsafe { ffi::sylow_hip_g2_double_batch(da.xy.as_ptr(), da.inf.as_ptr(), out.xy.as_mut_ptr(), out.inf.as_mut_ptr(), n, dev.stream) })?;
    download_g2(dev, &out)
}
/// `Mul<&Fp> for &G2Projective` for points that are NOT known to be in the r-torsion (generic window product, exact on the whole twist).
pub fn mul_g2_any_batch(dev: &Device, q: &DeviceG2, k: &[Fp]) -> Result<DeviceG2, HipError> {
    assert_eq!(q.n, k.len());
    let n = q.n;
    let dk = dev.upload_soa::<4>(&fp_words(k))?;
    let out = DeviceG2 { xy: dev.alloc::<u64>(16 * n)?, inf: dev.alloc::<u8>(n)?, n };
    // SAFETY: n points, 4 * n scalar words, n outputs.
    device::check(unsafe { ffi::sylow_hip_g2_scalar_mul_batch(q.xy.as_ptr(), q.inf.as_ptr(), dk.as_ptr(), out.xy.as_mut_ptr(), out.inf.as_mut_ptr(), n, dev.stream) })?;
    Ok(out)
}
/// `GroupTrait::rand` (g1.rs:293-305, g2.rs:227-240): generator * Fr::rand(rng), the scalars drawn on the host from the caller's generator.
pub fn g1_rand_batch<R: crypto_bigint::rand_core::CryptoRngCore>(dev: &Device, n: usize, rng: &mut R) -> Result<Vec<G1Projective>, HipError> {
    let k: Vec<[u64; 4]> = (0..n).map(|_| <Fr as sylow::FieldExtensionTrait<1, 1>>::rand(rng).value().to_words()).collect();
    let dk = dev.upload_soa::<4>(&k)?;
    let out = DeviceG1 { xy: dev.alloc::<u64>(8 * n)?, inf: dev.alloc::<u8>(n)?, n };
    // SAFETY: 4 * n scalar words, n outputs; the fixed-base table of the generator lives in the library.
    device::check(unsafe { ffi::sylow_hip_g1_generator_mul_batch(dk.as_ptr(), out.xy.as_mut_ptr(), out.inf.as_mut_ptr(), n, dev.stream) })?;
    download_g1(dev, &out)
}
pub fn g2_rand_batch<R: crypto_bigint::rand_core::CryptoRngCore>(dev: &Device, n: usize, rng: &mut R) -> Result<Vec<G2Projective>, HipError> {
    let k: Vec<Fp> = (0..n).map(|_| Fp::new(<Fr as sylow::FieldExtensionTrait<1, 1>>::rand(rng).value())).collect();
    crate::public_keys(dev, &k)
}
/// `GroupTrait::hash_to_curve` for G1 (g1.rs:307-331) with XMD-Keccak256 and sylow's DST (`dst = None`) or the caller's.
pub fn hash_to_curve_batch(dev: &Device, msgs: &[&[u8]], dst: Option<&[u8]>) -> Result<Vec<G1Projective>, HipError> {
    let n = msgs.len();
    let (d_msgs, d_off) = messages(dev, msgs)?;
    let out = DeviceG1 { xy: dev.alloc::<u64>(8 * n)?, inf: dev.alloc::<u8>(n)?, n };
    let (dst_ptr, dst_len) = dst.map_or((ptr::null(), 0), |d| (d.as_ptr(), d.len()));
    // SAFETY: n + 1 offsets into d_msgs; dst is a HOST pointer (read during the call); n outputs.
    device::check(unsafe { ffi::sylow_hip_hash_to_g1_batch(d_msgs.as_ptr(), d_off.as_ptr(), dst_ptr, dst_len, out.xy.as_mut_ptr(), out.inf.as_mut_ptr(), n, dev.stream) })?;
    download_g1(dev, &out)
}
/// `GroupTrait::sign_message` for G1 (g1.rs:355-366) = hash_to_curve(msg) * private_key: the batch is `sign_batch`.
pub fn sign_message_batch(dev: &Device, msgs: &[&[u8]], private_keys: &[Fp]) -> Result<Vec<G1Projective>, HipError> {
    crate::sign_batch(dev, private_keys, msgs)
}
/// `Expander::hash_to_field(msg, 2, 48)` (hasher.rs:84-128): two Fp per message.
pub fn hash_to_field_batch(dev: &Device, msgs: &[&[u8]], dst: Option<&[u8]>) -> Result<Vec<[Fp; 2]>, HipError> {
    let n = msgs.len();
    let (d_msgs, d_off) = messages(dev, msgs)?;
    let out = dev.alloc::<u64>(8 * n)?;
    let (dst_ptr, dst_len) = dst.map_or((ptr::null(), 0), |d| (d.as_ptr(), d.len()));
    // SAFETY: as hash_to_curve_batch; out holds 8 * n words.
    device::check(unsafe { ffi::sylow_hip_hash_to_field_batch(d_msgs.as_ptr(), d_off.as_ptr(), dst_ptr, dst_len, out.as_mut_ptr(), n, dev.stream) })?;
    Ok(dev.download_aos::<8>(&out, n)?.iter().map(|w| [fp_from_words(&w[0..4]), fp_from_words(&w[4..8])]).collect())
}
/// `SvdW::unchecked_map_to_point` (svdw.rs:180-262): u -> (x, y) on E(Fp); `None` where the reference returns an error.
pub fn svdw_map_batch(dev: &Device, u: &[Fp]) -> Result<Vec<Option<[Fp; 2]>>, HipError> {
    let n = u.len();
    let (du, out, st) = (dev.upload_soa::<4>(&fp_words(u))?, dev.alloc::<u64>(8 * n)?, dev.alloc::<u8>(n)?);
    // SAFETY: 4 * n words in, 8 * n words and n status bytes out.
    device::check(unsafe { ffi::sylow_hip_svdw_map_batch(du.as_ptr(), out.as_mut_ptr(), st.as_mut_ptr(), n, dev.stream) })?;
    let (w, s) = (dev.download_aos::<8>(&out, n)?, dev.download(&st)?);
    Ok(w.iter().zip(s).map(|(x, bad)| if bad == 0 { Some([fp_from_words(&x[0..4]), fp_from_words(&x[4..8])]) } else { None }).collect())
}
/// `G2Affine::endomorphism` (g2.rs:140-152).  Where the reference panics (the image is off the curve) the element fails with NotOnCurve.
pub fn endomorphism_batch(dev: &Device, q: &DeviceG2) -> Result<DeviceG2, HipError> {
    let n = q.n;
    let out = DeviceG2 { xy: dev.alloc::<u64>(16 * n)?, inf: dev.alloc::<u8>(n)?, n };
    let st = dev.alloc::<u8>(n)?;
    // SAFETY: n points + flags in, n outputs + n status bytes.
    device::check(unsafe { ffi::sylow_hip_g2_psi_batch(q.xy.as_ptr(), q.inf.as_ptr(), out.xy.as_mut_ptr(), out.inf.as_mut_ptr(), st.as_mut_ptr(), n, dev.stream) })?;
    first_failure(&dev.download(&st)?)?;
    Ok(out)
}
/// `G1Affine::new` (g1.rs:111-132) / the r-torsion test of `G2Projective::new` (g2.rs:460-525) on device-resident affine points: status bytes.
pub fn g1_on_curve_batch(dev: &Device, p: &DeviceG1) -> Result<Vec<u8>, HipError> {
    let st = dev.alloc::<u8>(p.n)?;
    // SAFETY: n points + flags, n status bytes.
    device::check(unsafe { ffi::sylow_hip_g1_on_curve_batch(p.xy.as_ptr(), p.inf.as_ptr(), st.as_mut_ptr(), p.n, dev.stream) })?;
    Ok(dev.download(&st)?)
}
pub fn g2_subgroup_check_batch(dev: &Device, q: &DeviceG2) -> Result<Vec<u8>, HipError> {
    let st = dev.alloc::<u8>(q.n)?;
    // SAFETY: n points + flags, n status bytes.
    device::check(unsafe { ffi::sylow_hip_g2_subgroup_check_batch(q.xy.as_ptr(), q.inf.as_ptr(), st.as_mut_ptr(), q.n, dev.stream) })?;
    Ok(dev.download(&st)?)
}
/// `G1Projective::new([x, y, z])` (g1.rs:383-402) / `G2Projective::new([x, y, z])` (g2.rs:460-525) on raw projective words: status bytes.
pub fn g1_projective_new_batch(dev: &Device, xyz: &[[u64; 12]]) -> Result<Vec<u8>, HipError> {
    let n = xyz.len();
    let (d, st) = (dev.upload_soa::<12>(xyz)?, dev.alloc::<u8>(n)?);
    // SAFETY: 12 * n words, n status bytes.
    device::check(unsafe { ffi::sylow_hip_g1_projective_new_batch(d.as_ptr(), st.as_mut_ptr(), n, dev.stream) })?;
    Ok(dev.download(&st)?)
}
pub fn g2_projective_new_batch(dev: &Device, xyz: &[[u64; 24]]) -> Result<Vec<u8>, HipError> {
    let n = xyz.len();
    let (d, st) = (dev.upload_soa::<24>(xyz)?, dev.alloc::<u8>(n)?);
    // SAFETY: 24 * n words, n status bytes.
    device::check(unsafe { ffi::sylow_hip_g2_projective_new_batch(d.as_ptr(), st.as_mut_ptr(), n, dev.stream) })?;
    Ok(dev.download(&st)?)
}
/// `ConstantTimeEq` / `PartialEq` for projective points (group.rs:426-447) on raw projective words.
pub fn g1_ct_eq_batch(dev: &Device, a: &[[u64; 12]], b: &[[u64; 12]]) -> Result<Vec<bool>, HipError> {
    assert_eq!(a.len(), b.len());
    let n = a.len();
    let (da, db, eq) = (dev.upload_soa::<12>(a)?, dev.upload_soa::<12>(b)?, dev.alloc::<u8>(n)?);
    // SAFETY: 12 * n words on each side, n result bytes.
    device::check(unsafe { ffi::sylow_hip_g1_ct_eq_batch(da.as_ptr(), db.as_ptr(), eq.as_mut_ptr(), n, dev.stream) })?;
    Ok(dev.download(&eq)?.iter().map(|&f| f != 0).collect())
}
pub fn g2_ct_eq_batch(dev: &Device, a: &[[u64; 24]], b: &[[u64; 24]]) -> Result<Vec<bool>, HipError> {
    assert_eq!(a.len(), b.len());
    let n = a.len();
    let (da, db, eq) = (dev.upload_soa::<24>(a)?, dev.upload_soa::<24>(b)?, dev.alloc::<u8>(n)?);
    // SAFETY: 24 * n words on each side, n result bytes.
    device::check(unsafe { ffi::sylow_hip_g2_ct_eq_batch(da.as_ptr(), db.as_ptr(), eq.as_mut_ptr(), n, dev.stream) })?;
    Ok(dev.download(&eq)?.iter().map(|&f| f != 0).collect())
}
/// `GroupAffine::from(&GroupProjective)` (group.rs:475-495) on raw projective words: affine words + identity flag.
pub fn g1_normalize_batch(dev: &Device, xyz: &[[u64; 12]]) -> Result<DeviceG1, HipError> {
    let n = xyz.len();
    let d = dev.upload_soa::<12>(xyz)?;
    let out = DeviceG1 { xy: dev.alloc::<u64>(8 * n)?, inf: dev.alloc::<u8>(n)?, n };
    // SAFETY: 12 * n words in, n outputs.
    device::check(unsafe { ffi::sylow_hip_g1_normalize_batch(d.as_ptr(), out.xy.as_mut_ptr(), out.inf.as_mut_ptr(), n, dev.stream) })?;
    Ok(out)
}
pub fn g2_normalize_batch(dev: &Device, xyz: &[[u64; 24]]) -> Result<DeviceG2, HipError> {
    let n = xyz.len();
    let d = dev.upload_soa::<24>(xyz)?;
    let out = DeviceG2 { xy: dev.alloc::<u64>(16 * n)?, inf: dev.alloc::<u8>(n)?, n };
    // SAFETY: 24 * n words in, n outputs.
    device::check(unsafe { ffi::sylow_hip_g2_normalize_batch(d.as_ptr(), out.xy.as_mut_ptr(), out.inf.as_mut_ptr(), n, dev.stream) })?;
    Ok(out)
}
/// `to_be_bytes` of device-resident points (g1.rs:151-180, g2.rs:319-359).
pub fn g1_to_be_bytes_batch(dev: &Device, p: &DeviceG1) -> Result<Vec<[u8; 64]>, HipError> {
    let out = dev.alloc::<u8>(64 * p.n)?;
    // SAFETY: n points + flags, 64 * n bytes out.
    device::check(unsafe { ffi::sylow_hip_g1_to_be_bytes_batch(p.xy.as_ptr(), p.inf.as_ptr(), out.as_mut_ptr(), p.n, dev.stream) })?;
    Ok(dev.download(&out)?.chunks_exact(64).map(|c| c.try_into().unwrap()).collect())
}
pub fn g2_to_be_bytes_batch(dev: &Device, q: &DeviceG2) -> Result<Vec<[u8; 128]>, HipError> {
    let out = dev.alloc::<u8>(128 * q.n)?;
    // SAFETY: n points + flags, 128 * n bytes out.
    device::check(unsafe { ffi::sylow_hip_g2_to_be_bytes_batch(q.xy.as_ptr(), q.inf.as_ptr(), out.as_mut_ptr(), q.n, dev.stream) })?;
    Ok(dev.download(&out)?.chunks_exact(128).map(|c| c.try_into().unwrap()).collect())
}
/// Weighted aggregation sum_i weights[j][i] * points[j][i] (examples/threshold_signing.rs:124-143), term-major input:
/// element (job j, term i) at index i * n_jobs + j.
pub fn lincomb(dev: &Device, points: &[G1Affine], weights: &[Fr], n_jobs: usize, n_terms: usize) -> Result<Vec<G1Projective>, HipError> {
    assert!(points.len() == n_jobs * n_terms && weights.len() == points.len());
    let dp = upload_g1(dev, points)?;
    let dk = dev.upload_soa::<4>(&fr_words(weights))?;
    let out = DeviceG1 { xy: dev.alloc::<u64>(8 * n_jobs)?, inf: dev.alloc::<u8>(n_jobs)?, n: n_jobs };
    // SAFETY: n_jobs * n_terms points and scalars, n_jobs outputs.
    device::check(unsafe {
        ffi::sylow_hip_g1_lincomb_batch(dp.xy.as_ptr(), dp.inf.as_ptr(), dk.as_ptr(), out.xy.as_mut_ptr(), out.inf.as_mut_ptr(), n_jobs, n_terms, dev.stream)
    })?;
    download_g1(dev, &out)
}
/// `Mul<Fr> for Gt` (gt.rs:188-215): gt[i] ^ k[i], the reference's own 256-step signed-digit algorithm.
pub fn gt_pow_batch(dev: &Device, gt: &[[u64; 48]], k: &[Fr]) -> Result<Vec<GtOut>, HipError> {
    assert_eq!(gt.len(), k.len());
    let n = gt.len();
    let (dg, dk, out) = (dev.upload_soa::<48>(gt)?, dev.upload_soa::<4>(&fr_words(k))?, dev.alloc::<u64>(48 * n)?);
    // SAFETY: 48 * n words, 4 * n scalar words, 48 * n words out.
    device::check(unsafe { ffi::sylow_hip_gt_pow_batch(dg.as_ptr(), dk.as_ptr(), out.as_mut_ptr(), n, dev.stream) })?;
    Ok(dev.download_aos::<48>(&out, n)?.iter().map(gt_from_words).collect())
}

// ------------------------------------------------------------------ pairing.rs: the loops as entry points of their own
/// `G2PreComputed::miller_loop` / `glued_miller_loop` from the POINTS (lines computed on the fly): raw Miller values, 48 words each.
pub fn miller_loop_batch(dev: &Device, p: &DeviceG1, q: &DeviceG2) -> Result<DeviceBuf<u64>, HipError> {
    assert_eq!(p.n, q.n);
    let f = dev.alloc::<u64>(48 * p.n)?;
    // SAFETY: n points on each side, 48 * n words out.
    device::check(unsafe { ffi::sylow_hip_miller_loop_batch(p.xy.as_ptr(), q.xy.as_ptr(), f.as_mut_ptr(), p.n, dev.stream) })?;
    Ok(f)
}
/// Job j multiplies pairs [offsets[j], offsets[j+1]) with shared squarings (pairing.rs:970-1022); raw values out.
pub fn glued_miller_loop_batch(dev: &Device, p: &DeviceG1, q: &DeviceG2, offsets: &[u64]) -> Result<DeviceBuf<u64>, HipError> {
    assert!(p.n == q.n && !offsets.is_empty() && *offsets.last().unwrap() as usize <= p.n);
    let n_jobs = offsets.len() - 1;
    let (d_off, f) = (dev.upload(offsets)?, dev.alloc::<u64>(48 * n_jobs.max(1))?);
    // SAFETY: n pairs, n_jobs + 1 offsets, 48 * n_jobs words out.
    device::check(unsafe { ffi::sylow_hip_glued_miller_loop_batch(p.xy.as_ptr(), q.xy.as_ptr(), d_off.as_ptr(), n_jobs, p.n, f.as_mut_ptr(), dev.stream) })?;
    Ok(f)
}
/// `glued_pairing` per job (pairing.rs:1029-1037; the ecPairing / Groth16 shape): Gt values and `== identity` flags.
pub fn glued_pairing_jobs(dev: &Device, p: &DeviceG1, q: &DeviceG2, offsets: &[u64], skip_identity: bool) -> Result<(Vec<GtOut>, Vec<bool>), HipError> {
    assert!(p.n == q.n && !offsets.is_empty() && *offsets.last().unwrap() as usize <= p.n);
    let n_jobs = offsets.len() - 1;
    let (d_off, gt, one) = (dev.upload(offsets)?, dev.alloc::<u64>(48 * n_jobs.max(1))?, dev.alloc::<u8>(n_jobs.max(1))?);
    // SAFETY: n pairs + flags, n_jobs + 1 offsets, 48 * n_jobs words and n_jobs flags out.
    device::check(unsafe {
        ffi::sylow_hip_multi_pairing_batch(p.xy.as_ptr(), p.inf.as_ptr(), q.xy.as_ptr(), q.inf.as_ptr(), d_off.as_ptr(), n_jobs, p.n, skip_identity as i32,
                                           gt.as_mut_ptr(), one.as_mut_ptr(), dev.stream)
    })?;
    let words = dev.download_aos::<48>(&gt, n_jobs)?;
    Ok((words.iter().map(gt_from_words).collect(), dev.download(&one)?[..n_jobs].iter().map(|&f| f != 0).collect()))
}
/// `glued_miller_loop(&[G2PreComputed], &[G1Affine])` against tables `g2_precompute_batch` wrote (`coeffs` = [87 * 24][n_tables]).
pub fn glued_miller_loop_precomputed(dev: &Device, coeffs: &DeviceBuf<u64>, n_tables: usize, table_idx: &[u64], p: &DeviceG1, offsets: &[u64]) -> Result<DeviceBuf<u64>, HipError> {
    assert!(table_idx.len() == p.n && table_idx.iter().all(|&t| (t as usize) < n_tables));
    assert!(!offsets.is_empty() && *offsets.last().unwrap() as usize <= p.n);
    let n_jobs = offsets.len() - 1;
    let (d_idx, d_off, f) = (dev.upload(table_idx)?, dev.upload(offsets)?, dev.alloc::<u64>(48 * n_jobs.max(1))?);
    // SAFETY: n_tables tables, n indices < n_tables, n points, n_jobs + 1 offsets, 48 * n_jobs words out.
    device::check(unsafe {
        ffi::sylow_hip_glued_miller_loop_precomputed_batch(coeffs.as_ptr(), n_tables, d_idx.as_ptr(), p.xy.as_ptr(), d_off.as_ptr(), n_jobs, p.n, f.as_mut_ptr(), dev.stream)
    })?;
    Ok(f)
}
/// The two halves of a product split over GPUs by a host with its own transport: this shard's Miller product up to a factor in Fp*
/// (an opaque intermediate: the factor disappears in the final exponentiation; the reference's raw value comes from `miller_loop_batch`),
/// and product + final exponentiation over gathered partials (`parts` = [48][k] SoA).
pub fn pairing_product_partial(dev: &Device, p: &DeviceG1, q: &DeviceG2, skip_identity: bool) -> Result<DeviceBuf<u64>, HipError> {
    assert_eq!(p.n, q.n);
    let f = dev.alloc::<u64>(48)?;
    // SAFETY: n pairs + flags, 48 words out.
    device::check(unsafe {
        ffi::sylow_hip_pairing_product_partial_batch(p.xy.as_ptr(), p.inf.as_ptr(), q.xy.as_ptr(), q.inf.as_ptr(), p.n, skip_identity as i32, f.as_mut_ptr(), dev.stream)
    })?;
    Ok(f)
}
pub fn fp12_product_final_exp(dev: &Device, parts: &DeviceBuf<u64>, k: usize) -> Result<(GtOut, bool), HipError> {
    assert_eq!(parts.len, 48 * k);
    let (gt, one) = (dev.alloc::<u64>(48)?, dev.alloc::<u8>(1)?);
    // SAFETY: 48 * k words in, 48 words and one flag out.
    device::check(unsafe { ffi::sylow_hip_fp12_product_final_exp(parts.as_ptr(), k, gt.as_mut_ptr(), one.as_mut_ptr(), dev.stream) })?;
    let words = dev.download_aos::<48>(&gt, 1)?;
    Ok((gt_from_words(&words[0]), dev.download(&one)?[0] != 0))
}
/// This shard's Miller product (up to a factor in Fp*) of the aggregate check (signatures summed in G1 first), for a host-side gather.
pub fn aggregate_partial(dev: &Device, pk: &DeviceG2, msgs: &[&[u8]], sig: &DeviceG1) -> Result<DeviceBuf<u64>, HipError> {
    assert!(sig.n == msgs.len() && (pk.n == msgs.len() || pk.n == 1));
    let (d_msgs, d_off) = messages(dev, msgs)?;
    let f = dev.alloc::<u64>(48)?;
    // SAFETY: pk.n keys, n signatures, n + 1 offsets, 48 words out.
    device::check(unsafe {
        ffi::sylow_hip_bls_aggregate_partial_batch(pk.xy.as_ptr(), pk.inf.as_ptr(), pk.n, d_msgs.as_ptr(), d_off.as_ptr(), sig.xy.as_ptr(), sig.inf.as_ptr(), sig.n,
                                                   f.as_mut_ptr(), dev.stream)
    })?;
    Ok(f)
}
/// The SOUND one-boolean batch verification (small-exponent test): prod_i [e(sig_i, G2gen) e(-H(msg_i), pk_i)]^(w_i) == identity, with
/// `weights` drawn by the caller AFTER the signatures are fixed (e.g. 128 random bits each).  True when every signature is valid; a batch
/// with an invalid one passes with probability at most 2^-(bits of the weights).  `pk` one key per message or ONE key; `comm` as `all_valid`.
pub fn batch_verify_weighted(dev: &Device, pk: &DeviceG2, msgs: &[&[u8]], sig: &DeviceG1, weights: &[Fp], comm: *mut c_void) -> Result<(GtOut, bool), HipError> {
    assert!(sig.n == msgs.len() && weights.len() == msgs.len() && (pk.n == msgs.len() || pk.n == 1));
    let (d_msgs, d_off) = messages(dev, msgs)?;
    let dw = dev.upload_soa::<4>(&fp_words(weights))?;
    let (gt, one) = (dev.alloc::<u64>(48)?, dev.alloc::<u8>(1)?);
    // SAFETY: pk.n keys, n signatures, n + 1 offsets, 4 * n weight words; gt 48 words; one 1 byte; comm null or a live communicator.
    device::check(unsafe {
        ffi::sylow_hip_bls_batch_verify_weighted(pk.xy.as_ptr(), pk.inf.as_ptr(), pk.n, d_msgs.as_ptr(), d_off.as_ptr(), sig.xy.as_ptr(), sig.inf.as_ptr(),
                                                 dw.as_ptr(), sig.n, comm, gt.as_mut_ptr(), one.as_mut_ptr(), dev.stream)
    })?;
    let words = dev.download_aos::<48>(&gt, 1)?;
    Ok((gt_from_words(&words[0]), dev.download(&one)?[0] != 0))
}
/// This shard's Miller product (up to a factor in Fp*) of the weighted test, for a host-side gather (`fp12_product_final_exp` finishes it).
pub fn weighted_partial(dev: &Device, pk: &DeviceG2, msgs: &[&[u8]], sig: &DeviceG1, weights: &[Fp]) -> Result<DeviceBuf<u64>, HipError> {
    assert!(sig.n == msgs.len() && weights.len() == msgs.len() && (pk.n == msgs.len() || pk.n == 1));
    let (d_msgs, d_off) = messages(dev, msgs)?;
    let dw = dev.upload_soa::<4>(&fp_words(weights))?;
    let f = dev.alloc::<u64>(48)?;
    // SAFETY: as batch_verify_weighted; 48 words out.
    device::check(unsafe {
        ffi::sylow_hip_bls_weighted_partial_batch(pk.xy.as_ptr(), pk.inf.as_ptr(), pk.n, d_msgs.as_ptr(), d_off.as_ptr(), sig.xy.as_ptr(), sig.inf.as_ptr(),
                                                  dw.as_ptr(), sig.n, f.as_mut_ptr(), dev.stream)
    })?;
    Ok(f)
}
/// AND of a device-resident flag vector (one rank; `all_valid` in lib.rs adds the reduce over ranks).
pub fn flags_all(dev: &Device, flags: &DeviceBuf<u8>) -> Result<bool, HipError> {
    let out = dev.alloc::<i32>(1)?;
    // SAFETY: flags.len bytes, one int32 out.
    device::check(unsafe { ffi::sylow_hip_flags_all(flags.as_ptr(), flags.len, out.as_mut_ptr(), dev.stream) })?;
    Ok(dev.download(&out)?[0] == 1)
}
/// One signer, many messages, the key's line table rebuilt inside the call (examples/verify_multiple_messages_same_signer.rs:41-60);
/// `KeyTable` in lib.rs is the cached form.  `fused = true` names the one-final-exponentiation kernel explicitly.
pub fn verify_same_signer_batch(dev: &Device, pk: &G2Affine, msgs: &[&[u8]], sig: &[G1Affine]) -> Result<Vec<bool>, HipError> {
    assert_eq!(msgs.len(), sig.len());
    let n = msgs.len();
    let (dpk, dsig) = (upload_g2(dev, std::slice::from_ref(pk))?, upload_g1(dev, sig)?);
    let (d_msgs, d_off) = messages(dev, msgs)?;
    let ok = dev.alloc::<u8>(n)?;
    // SAFETY: one key, n signatures, n + 1 offsets, n flags.
    device::check(unsafe {
        ffi::sylow_hip_bls_verify_same_signer_batch(dpk.xy.as_ptr(), dpk.inf.as_ptr(), d_msgs.as_ptr(), d_off.as_ptr(), dsig.xy.as_ptr(), dsig.inf.as_ptr(),
                                                    ok.as_mut_ptr(), n, dev.stream)
    })?;
    Ok(dev.download(&ok)?.iter().map(|&f| f != 0).collect())
}
pub fn verify_fused_batch(dev: &Device, pk: &DeviceG2, msgs: &[&[u8]], sig: &DeviceG1) -> Result<Vec<bool>, HipError> {
    assert!(pk.n == msgs.len() && sig.n == msgs.len());
    let n = msgs.len();
    let (d_msgs, d_off) = messages(dev, msgs)?;
    let ok = dev.alloc::<u8>(n)?;
    // SAFETY: n keys, n signatures, n + 1 offsets, n flags.
    device::check(unsafe {
        ffi::sylow_hip_bls_verify_fused_batch(pk.xy.as_ptr(), pk.inf.as_ptr(), d_msgs.as_ptr(), d_off.as_ptr(), sig.xy.as_ptr(), sig.inf.as_ptr(), ok.as_mut_ptr(), n, dev.stream)
    })?;
    Ok(dev.download(&ok)?.iter().map(|&f| f != 0).collect())
}

// ------------------------------------------------------------------ examples/reth_bn128.rs: the EVM precompile adapters
pub mod evm {
    //! `run_add` / `run_mul` / `run_pair` (examples/reth_bn128.rs:99-217) as batches over the precompiles' byte formats.  Inputs are
    //! padded / truncated by the caller exactly as the reference's `right_pad` does; per-job status bytes mirror `GroupError`.
    use super::*;

    /// n x 128 bytes (two G1 points) -> n x 64 bytes + status.
    pub fn run_add(dev: &Device, input: &[[u8; 128]]) -> Result<(Vec<[u8; 64]>, Vec<u8>), HipError> {
        let n = input.len();
        let flat: Vec<u8> = input.iter().flatten().copied().collect();
        let (d_in, out, st) = (dev.upload(&flat)?, dev.alloc::<u8>(64 * n)?, dev.alloc::<u8>(n)?);
        // SAFETY: 128 * n bytes in, 64 * n bytes and n status bytes out.
        device::check(unsafe { ffi::sylow_hip_evm_ecadd_batch(d_in.as_ptr(), out.as_mut_ptr(), st.as_mut_ptr(), n, dev.stream) })?;
        Ok((dev.download(&out)?.chunks_exact(64).map(|c| c.try_into().unwrap()).collect(), dev.download(&st)?))
    }
    /// n x 96 bytes (a G1 point and a 256-bit scalar, reduced mod r as EIP-196 requires) -> n x 64 bytes + status.
    pub fn run_mul(dev: &Device, input: &[[u8; 96]]) -> Result<(Vec<[u8; 64]>, Vec<u8>), HipError> {
        let n = input.len();
        let flat: Vec<u8> = input.iter().flatten().copied().collect();
        let (d_in, out, st) = (dev.upload(&flat)?, dev.alloc::<u8>(64 * n)?, dev.alloc::<u8>(n)?);
        // SAFETY: 96 * n bytes in, 64 * n bytes and n status bytes out.
        device::check(unsafe { ffi::sylow_hip_evm_ecmul_batch(d_in.as_ptr(), out.as_mut_ptr(), st.as_mut_ptr(), n, dev.stream) })?;
        Ok((dev.download(&out)?.chunks_exact(64).map(|c| c.try_into().unwrap()).collect(), dev.download(&st)?))
    }
    /// Jobs of k_j pairs of 192 bytes each (`jobs[j].len() == 192 * k_j`) -> (pairing check result, status) per job.
    pub fn run_pair(dev: &Device, jobs: &[&[u8]]) -> Result<(Vec<bool>, Vec<u8>), HipError> {
        let n_jobs = jobs.len();
        let mut offsets = Vec::with_capacity(n_jobs + 1);
        let mut flat = Vec::new();
        offsets.push(0u64);
        for j in jobs {
            assert_eq!(j.len() % 192, 0);
            flat.extend_from_slice(j);
            offsets.push((flat.len() / 192) as u64);
        }
        let n_pairs = flat.len() / 192;
        if flat.is_empty() {
            flat.push(0);
        }
        let (d_in, d_off) = (dev.upload(&flat)?, dev.upload(&offsets)?);
        let (res, st) = (dev.alloc::<u8>(n_jobs.max(1))?, dev.alloc::<u8>(n_jobs.max(1))?);
        // SAFETY: 192 * n_pairs bytes, n_jobs + 1 offsets, n_jobs result and status bytes.
        device::check(unsafe { ffi::sylow_hip_evm_ecpairing_batch(d_in.as_ptr(), d_off.as_ptr(), n_jobs, n_pairs, res.as_mut_ptr(), st.as_mut_ptr(), dev.stream) })?;
        Ok((dev.download(&res)?[..n_jobs].iter().map(|&f| f != 0).collect(), dev.download(&st)?[..n_jobs].to_vec()))
    }
}

// ------------------------------------------------------------------ process-level plumbing
/// Number of GPUs the library can see, and `sylow_hip_init_devices` for hosts that drive several GPUs from one process.
pub fn device_count() -> i32 {
    // SAFETY: no arguments.
    unsafe { ffi::sylow_hip_device_count() }
}
pub fn init_devices(ordinals: &[i32]) -> Result<(), device::Error> {
    // SAFETY: a host array of ordinals.len() int32.
    device::check(unsafe { ffi::sylow_hip_init_devices(ordinals.as_ptr(), ordinals.len() as i32) })
}
/// The seeded xoshiro256** stream of BASELINE.md §3 (bench and test inputs; NOT a cryptographic generator): n draws below p.
pub fn xoshiro_fp(seed: u64, n: usize) -> Result<Vec<Fp>, device::Error> {
    let mut soa = vec![0u64; 4 * n];
    // SAFETY: a HOST array in the library's SoA layout [4][stride] with stride = n.
    device::check(unsafe { ffi::sylow_hip_host_xoshiro_fp(seed, soa.as_mut_ptr(), n, n) })?;
    Ok((0..n).map(|i| fp_from_words(&[soa[i], soa[n + i], soa[2 * n + i], soa[3 * n + i]])).collect())
}

// ------------------------------------------------------------------ host-array pipelines on canonical words
/// Page-locked host memory (`sylow_hip_host_malloc`): staging arrays for the `*_host` calls, whose copies then run asynchronously
/// beside the kernels.  Freed on drop.
pub struct PinnedBuf<T: Copy> {
    ptr: *mut T,
    len: usize,
}
impl<T: Copy> PinnedBuf<T> {
    pub fn new(len: usize) -> Result<Self, device::Error> {
        let mut p: *mut c_void = std::ptr::null_mut();
        // SAFETY: `p` is a valid out-pointer.
        device::check(unsafe { ffi::sylow_hip_host_malloc(&mut p, len * std::mem::size_of::<T>()) })?;
        Ok(PinnedBuf { ptr: p as *mut T, len })
    }
    pub fn as_slice(&self) -> &[T] {
        // SAFETY: `len` elements were allocated; T: Copy has no drop glue and every bit pattern written by the library is a valid integer.
        unsafe { std::slice::from_raw_parts(self.ptr, self.len) }
    }
    pub fn as_mut_slice(&mut self) -> &mut [T] {
        // SAFETY: as above; unique borrow.
        unsafe { std::slice::from_raw_parts_mut(self.ptr, self.len) }
    }
}
impl<T: Copy> Drop for PinnedBuf<T> {
    fn drop(&mut self) {
        // SAFETY: the pointer came from sylow_hip_host_malloc.
        unsafe { ffi::sylow_hip_host_free(self.ptr as *mut c_void) };
    }
}

/// `pairing` on canonical words held by the host (x, y per G1 point: `[u64; 8]`; x.c0, x.c1, y.c0, y.c1 per G2 point: `[u64; 16]`):
/// the chunked two-stream pipeline of `sylow_hip_pairing_host`, writing into `gt` (which may live in a `PinnedBuf`).
pub fn pairing_words_host(p: &[[u64; 8]], p_inf: Option<&[u8]>, q: &[[u64; 16]], q_inf: Option<&[u8]>, gt: &mut [[u64; 48]]) -> Result<(), device::Error> {
    let n = p.len();
    assert!(q.len() == n && gt.len() == n && p_inf.map_or(true, |f| f.len() == n) && q_inf.map_or(true, |f| f.len() == n));
    // SAFETY: n elements in every array; host memory; the call returns after the last copy has landed.
    device::check(unsafe {
        ffi::sylow_hip_pairing_host(p.as_ptr() as *const u64, p_inf.map_or(std::ptr::null(), |f| f.as_ptr()), q.as_ptr() as *const u64,
                                    q_inf.map_or(std::ptr::null(), |f| f.as_ptr()), gt.as_mut_ptr() as *mut u64, n, 0)
    })
}

/// `verify` on canonical words held by the host (`sylow_hip_bls_verify_host`): ok[i] = verify(pk[i], msgs[offsets[i]..offsets[i + 1]], sig[i]).
pub fn verify_words_host(pk: &[[u64; 16]], pk_inf: Option<&[u8]>, msgs: &[u8], offsets: &[u64], sig: &[[u64; 8]], sig_inf: Option<&[u8]>,
                         ok: &mut [u8]) -> Result<(), device::Error> {
    let n = pk.len();
    assert!(sig.len() == n && ok.len() == n && offsets.len() == n + 1 && *offsets.last().unwrap() as usize <= msgs.len());
    assert!(pk_inf.map_or(true, |f| f.len() == n) && sig_inf.map_or(true, |f| f.len() == n));
    // SAFETY: n keys / signatures / flags, n + 1 offsets into msgs; host memory; synchronous call.
    device::check(unsafe {
        ffi::sylow_hip_bls_verify_host(pk.as_ptr() as *const u64, pk_inf.map_or(std::ptr::null(), |f| f.as_ptr()), msgs.as_ptr(), offsets.as_ptr(),
                                       sig.as_ptr() as *const u64, sig_inf.map_or(std::ptr::null(), |f| f.as_ptr()), ok.as_mut_ptr(), n, 0)
    })
}

/// sum_i p[i] as one point (`sylow_hip_g1_sum_batch`): the `+` fold of examples/verify_multiple_messages_same_signer.rs:41-60 over a resident batch.
pub fn g1_sum(dev: &Device, p: &DeviceG1) -> Result<G1Projective, HipError> {
    let out = DeviceG1 { xy: dev.alloc::<u64>(8)?, inf: dev.alloc::<u8>(1)?, n: 1 };
    // SAFETY: p holds p.n points and flags; out one point and one flag.
    device::check(unsafe { ffi::sylow_hip_g1_sum_batch(p.xy.as_ptr(), p.inf.as_ptr(), p.n, out.xy.as_mut_ptr(), out.inf.as_mut_ptr(), dev.stream) })?;
    Ok(download_g1(dev, &out)?.remove(0))
}

/// Upper bound for the line tables of the multi-pair routes (`sylow_hip_set_scratch_limit`; 0 = the default of 12 GB): the one scratch
/// user whose size is not proportional to its input.  Process-wide; results do not depend on it.
pub fn set_scratch_limit(bytes: usize) -> Result<(), device::Error> {
    // SAFETY: no pointers.
    device::check(unsafe { ffi::sylow_hip_set_scratch_limit(bytes) })
}

/// The library's route selectors and thresholds (`SYLOW_HIP_OPT_*` of include/sylow_hip.h; the library reads no environment variable).
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
#[repr(i32)]
pub enum RouteOption { Stagger = 0, MultiTables = 1, WideTail = 2, WidePack = 3, AggFork = 4, SignWideMax = 5, WideMax = 6, WideVerifyMax = 7, QuadMax = 8, TailSplit = 9 }

/// `sylow_hip_set_option`: `None` restores the default.  Process-wide; results do not depend on any setting.
pub fn set_option(option: RouteOption, value: Option<u64>) -> Result<(), device::Error> {
    // SAFETY: plain value arguments.
    device::check(unsafe { ffi::sylow_hip_set_option(option as i32, value.map_or(-1, |v| v as i64)) })
}

/// `sylow_hip_get_option`: `None` = the default is in force.
pub fn get_option(option: RouteOption) -> Result<Option<u64>, device::Error> {
    let mut v: i64 = -1;
    // SAFETY: one host word.
    device::check(unsafe { ffi::sylow_hip_get_option(option as i32, &mut v) })?;
    Ok(if v < 0 { None } else { Some(v as u64) })
}

/// Live clock probe of the metric's kernels (`sylow_hip_clock_probe`): `acc` = 256 zeroed device words, or `None` to switch it off.  The
/// buffer must outlive the probe.  Returns the accumulators' meaning in include/sylow_hip.h.
pub fn clock_probe(acc: Option<&DeviceBuf<u64>>) -> Result<(), device::Error> {
    if let Some(a) = acc { assert!(a.len >= 256); }
    // SAFETY: a device buffer of at least 256 words, or NULL.
    device::check(unsafe { ffi::sylow_hip_clock_probe(acc.map_or(ptr::null_mut(), |a| a.as_mut_ptr())) })
}

/// Rate of the constant-rate counter the clock probe reads, in kHz (`sylow_hip_wall_clock_khz`).
pub fn wall_clock_khz() -> Result<i32, device::Error> {
    let mut khz: i32 = 0;
    // SAFETY: one host word.
    device::check(unsafe { ffi::sylow_hip_wall_clock_khz(&mut khz) })?;
    Ok(khz)
}

/// Hand the current device's idle scratch blocks above `keep_bytes` back to the driver (`sylow_hip_trim`): the library keeps the largest
/// block a call has needed for reuse, which after a 2^20-pair product is several GB.
pub fn trim(keep_bytes: usize) -> Result<(), device::Error> {
    // SAFETY: plain value argument.
    device::check(unsafe { ffi::sylow_hip_trim(keep_bytes) })
}
