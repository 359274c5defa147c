//! sylow-hip -- sylow's hot path as batches on one MI355X: `pairing`, `glued_pairing`, `sign`, `verify` with sylow's own types
//! at the surface and libsylow_hip.so (hand-written gfx950 kernels) underneath.
//!
//! NOT COMPILED IN THE AUTHORING ENVIRONMENT (no cargo / rustc there).  The C declarations in `ffi.rs` are generated from
//! include/sylow_hip.h and verified against it by tests/test_rust_ffi.py; the same entry points are exercised on the GPU by the
//! C++ host (include/sylow_hip.hpp) and the ctypes host (sylow_amd/engine.py) with the argument conventions used below.
//!
//! Conventions of the boundary (include/sylow_hip.h): nothing of sylow's memory layout crosses it -- there is no `#[repr(C)]`
//! upstream and `Fp` holds a Montgomery value -- so points travel in sylow's own big-endian wire format
//! (`G1Affine::to_be_bytes`, groups/g1.rs:151-180; `G2Affine::to_be_bytes`, groups/g2.rs:319-359) and are decoded and validated
//! ON THE DEVICE; field elements come back as canonical little-endian words and are rebuilt with `Fp::new(U256::from_words(..))`
//! (fields/fp.rs:199-201).
mod device;
#[allow(dead_code)]
mod ffi;
pub mod surface;

pub use device::{Device, DeviceBuf, Error};

use crypto_bigint::U256;
use std::os::raw::c_void;
use std::ptr;
use sylow::{Fp, Fp12, Fp2, Fp6, G1Affine, G1Projective, G2Affine, G2Projective, GroupError};

#[cfg(feature = "gt-from-fp12")]
pub type GtOut = sylow::Gt;
#[cfg(not(feature = "gt-from-fp12"))]
pub type GtOut = Fp12;

/// Per-element status bytes of the C ABI (mirror sylow's GroupError, groups/group.rs:38-47).
pub const ST_OK: u8 = 0;
pub const ST_NOT_ON_CURVE: u8 = 1;
pub const ST_NOT_IN_SUBGROUP: u8 = 2;
pub const ST_CANNOT_HASH: u8 = 3;
pub const ST_DECODE_ERROR: u8 = 4;

#[derive(Debug)]
pub enum HipError {
    /// the library or the HIP runtime failed
    Runtime(Error),
    /// element `index` was rejected exactly where sylow would return this error
    Group { index: usize, error: GroupError },
}
impl From<Error> for HipError {
    fn from(e: Error) -> Self {
        HipError::Runtime(e)
    }
}

fn group_error(status: u8) -> GroupError {
    match status {
        ST_NOT_ON_CURVE => GroupError::NotOnCurve,
        ST_NOT_IN_SUBGROUP => GroupError::NotInSubgroup,
        ST_CANNOT_HASH => GroupError::CannotHashToGroup,
        _ => GroupError::DecodeError,
    }
}

pub(crate) fn first_failure(status: &[u8]) -> Result<(), HipError> {
    match status.iter().position(|&s| s != ST_OK) {
        None => Ok(()),
        Some(index) => Err(HipError::Group { index, error: group_error(status[index]) }),
    }
}

pub(crate) fn fp_from_words(w: &[u64]) -> Fp {
    Fp::new(U256::from_words([w[0], w[1], w[2], w[3]]))
}
fn fp2_from_words(w: &[u64]) -> Fp2 {
    Fp2::new(&[fp_from_words(&w[0..4]), fp_from_words(&w[4..8])])
}
fn fp6_from_words(w: &[u64]) -> Fp6 {
    Fp6::new(&[fp2_from_words(&w[0..8]), fp2_from_words(&w[8..16]), fp2_from_words(&w[16..24])])
}
/// 48 canonical words in the reference's nesting order (c0 then c1, each an Fp6) -> Fp12 / Gt
pub(crate) fn gt_from_words(w: &[u64; 48]) -> GtOut {
    let f = Fp12::new(&[fp6_from_words(&w[0..24]), fp6_from_words(&w[24..48])]);
    #[cfg(feature = "gt-from-fp12")]
    {
        sylow::Gt::from(f)
    }
    #[cfg(not(feature = "gt-from-fp12"))]
    {
        f
    }
}

/// Points decoded and validated on the device, resident in the engine's layout.
pub struct DeviceG1 {
    pub xy: DeviceBuf<u64>,  // [8][n]
    pub inf: DeviceBuf<u8>,  // [n]
    pub n: usize,
}
pub struct DeviceG2 {
    pub xy: DeviceBuf<u64>,  // [16][n]
    pub inf: DeviceBuf<u8>,
    pub n: usize,
}

/// `G1Affine` batch -> device (wire format up, decode + on-curve check there: G1Affine::from_be_bytes, g1.rs:224-280).
pub fn upload_g1(dev: &Device, pts: &[G1Affine]) -> Result<DeviceG1, HipError> {
    let n = pts.len();
    let mut bytes = Vec::with_capacity(64 * n);
    for p in pts {
        bytes.extend_from_slice(&p.to_be_bytes());
    }
    let d_in = dev.upload(&bytes)?;
    let (xy, inf, st) = (dev.alloc::<u64>(8 * n)?, dev.alloc::<u8>(n)?, dev.alloc::<u8>(n)?);
    // SAFETY: d_in holds n * 64 bytes; xy 8 * n words; inf / st n bytes.
    device::check(unsafe { ffi::sylow_hip_g1_from_be_bytes_batch(d_in.as_ptr(), xy.as_mut_ptr(), inf.as_mut_ptr(), st.as_mut_ptr(), n, dev.stream) })?;
    first_failure(&dev.download(&st)?)?;
    Ok(DeviceG1 { xy, inf, n })
}

/// `G2Affine` batch -> device (decode + twist equation + r-torsion check there: G2Projective::new, g2.rs:460-525).
pub fn upload_g2(dev: &Device, pts: &[G2Affine]) -> Result<DeviceG2, HipError> {
    let n = pts.len();
    let mut bytes = Vec::with_capacity(128 * n);
    for p in pts {
        bytes.extend_from_slice(&p.to_be_bytes());
    }
    let d_in = dev.upload(&bytes)?;
    let (xy, inf, st) = (dev.alloc::<u64>(16 * n)?, dev.alloc::<u8>(n)?, dev.alloc::<u8>(n)?);
    // SAFETY: d_in holds n * 128 bytes; xy 16 * n words; inf / st n bytes.
    device::check(unsafe { ffi::sylow_hip_g2_from_be_bytes_batch(d_in.as_ptr(), xy.as_mut_ptr(), inf.as_mut_ptr(), st.as_mut_ptr(), n, dev.stream) })?;
    first_failure(&dev.download(&st)?)?;
    Ok(DeviceG2 { xy, inf, n })
}

/// Device G1 batch -> `G1Projective` (affine words -> `G1Projective::new([x, y, 1])`, identity -> default).
pub fn download_g1(dev: &Device, pts: &DeviceG1) -> Result<Vec<G1Projective>, HipError> {
    let xy = dev.download_aos::<8>(&pts.xy, pts.n)?;
    let inf = dev.download(&pts.inf)?;
    let mut out = Vec::with_capacity(pts.n);
    for (i, w) in xy.iter().enumerate() {
        if inf[i] != 0 {
            out.push(G1Projective::default());
        } else {
            let p = G1Projective::new([fp_from_words(&w[0..4]), fp_from_words(&w[4..8]), Fp::ONE])
                .map_err(|error| HipError::Group { index: i, error })?;
            out.push(p);
        }
    }
    Ok(out)
}

/// Device G2 batch -> `G2Projective` (affine words -> `G2Projective::new([x, y, 1])`, identity -> default).
pub fn download_g2(dev: &Device, pts: &DeviceG2) -> Result<Vec<G2Projective>, HipError> {
    let xy = dev.download_aos::<16>(&pts.xy, pts.n)?;
    let inf = dev.download(&pts.inf)?;
    let mut out = Vec::with_capacity(pts.n);
    for (i, w) in xy.iter().enumerate() {
        if inf[i] != 0 {
            out.push(G2Projective::default());
        } else {
            let p = G2Projective::new([fp2_from_words(&w[0..8]), fp2_from_words(&w[8..16]), Fp2::new(&[Fp::ONE, Fp::ZERO])])
                .map_err(|error| HipError::Group { index: i, error })?;
            out.push(p);
        }
    }
    Ok(out)
}

/// Batched `Mul<&Fp> for G1Projective` (group.rs:639-667): out[i] = p[i] * k[i] (the scalar is an Fp VALUE, as upstream).
pub fn mul_g1_batch(dev: &Device, p: &[G1Affine], k: &[Fp]) -> Result<Vec<G1Projective>, HipError> {
    assert_eq!(p.len(), k.len());
    let n = p.len();
    let dp = upload_g1(dev, p)?;
    let words: Vec<[u64; 4]> = k.iter().map(|s| s.value().to_words()).collect();
    let dk = dev.upload_soa::<4>(&words)?;
    let out = DeviceG1 { xy: dev.alloc::<u64>(8 * n)?, inf: dev.alloc::<u8>(n)?, n };
    // SAFETY: n points, 4 * n scalar words, n outputs.
    device::check(unsafe { ffi::sylow_hip_g1_scalar_mul_batch(dp.xy.as_ptr(), dp.inf.as_ptr(), dk.as_ptr(), out.xy.as_mut_ptr(), out.inf.as_mut_ptr(), n, dev.stream) })?;
    download_g1(dev, &out)
}

/// Batched `Mul<&Fp> for G2Projective`: out[i] = q[i] * k[i].  `upload_g2` has established that every q[i] is in the r-torsion
/// (as `G2Projective::new` does upstream), so the product takes the endomorphism-split kernel.
pub fn mul_g2_batch(dev: &Device, q: &[G2Affine], k: &[Fp]) -> Result<Vec<G2Projective>, HipError> {
    assert_eq!(q.len(), k.len());
    let n = q.len();
    let dq = upload_g2(dev, q)?;
    let words: Vec<[u64; 4]> = k.iter().map(|s| s.value().to_words()).collect();
    let dk = dev.upload_soa::<4>(&words)?;
    let out = DeviceG2 { xy: dev.alloc::<u64>(16 * n)?, inf: dev.alloc::<u8>(n)?, n };
    // SAFETY: n points in G2 proper, 4 * n scalar words, n outputs.
    device::check(unsafe {
        ffi::sylow_hip_g2_scalar_mul_subgroup_batch(dq.xy.as_ptr(), dq.inf.as_ptr(), dk.as_ptr(), out.xy.as_mut_ptr(), out.inf.as_mut_ptr(), n, dev.stream)
    })?;
    download_g2(dev, &out)
}

/// `KeyPair::generate`'s public half for a batch of secret keys (lib.rs:131-137): pk[i] = G2Projective::generator() * sk[i].
pub fn public_keys(dev: &Device, sk: &[Fp]) -> Result<Vec<G2Projective>, HipError> {
    let n = sk.len();
    let words: Vec<[u64; 4]> = sk.iter().map(|s| s.value().to_words()).collect();
    let dk = dev.upload_soa::<4>(&words)?;
    let out = DeviceG2 { xy: dev.alloc::<u64>(16 * n)?, inf: dev.alloc::<u8>(n)?, n };
    // SAFETY: 4 * n scalar words, n outputs; the fixed-base table of the generator lives in the library.
    device::check(unsafe { ffi::sylow_hip_g2_generator_mul_batch(dk.as_ptr(), out.xy.as_mut_ptr(), out.inf.as_mut_ptr(), n, dev.stream) })?;
    download_g2(dev, &out)
}

pub(crate) fn messages(dev: &Device, msgs: &[&[u8]]) -> Result<(DeviceBuf<u8>, DeviceBuf<u64>), Error> {
    let mut offsets = Vec::with_capacity(msgs.len() + 1);
    let mut blob = Vec::new();
    offsets.push(0u64);
    for m in msgs {
        blob.extend_from_slice(m);
        offsets.push(blob.len() as u64);
    }
    if blob.is_empty() {
        blob.push(0);
    }
    Ok((dev.upload(&blob)?, dev.upload(&offsets)?))
}

/// Batched `sylow::pairing` (pairing.rs:870-893): out[i] = pairing(p[i], q[i]); an identity on either side gives Gt::identity().
/// Host slices in, host vector out: the points go up in sylow's wire format and the whole call runs through the library's chunked,
/// double-buffered pipeline (`sylow_hip_pairing_host_bytes`: decode + validation + Miller loop + final exponentiation of chunk k
/// while the copy engines move chunk k - 1 out and chunk k + 1 in).  The first rejected element is reported as sylow's GroupError.
pub fn pairing_batch(_dev: &Device, p: &[G1Affine], q: &[G2Affine]) -> Result<Vec<GtOut>, HipError> {
    assert_eq!(p.len(), q.len());
    let n = p.len();
    let (mut pb, mut qb) = (Vec::with_capacity(64 * n), Vec::with_capacity(128 * n));
    for x in p {
        pb.extend_from_slice(&x.to_be_bytes());
    }
    for x in q {
        qb.extend_from_slice(&x.to_be_bytes());
    }
    let mut gt = vec![[0u64; 48]; n];
    let (mut st_p, mut st_q) = (vec![0u8; n], vec![0u8; n]);
    // SAFETY: pb holds n * 64 bytes, qb n * 128 bytes, gt n * 48 words, the status vectors n bytes each -- all HOST memory that
    // outlives the (synchronous) call.
    device::check(unsafe {
        ffi::sylow_hip_pairing_host_bytes(pb.as_ptr(), qb.as_ptr(), gt.as_mut_ptr() as *mut u64, st_p.as_mut_ptr(), st_q.as_mut_ptr(), n, 0)
    })?;
    first_failure(&st_p)?;
    first_failure(&st_q)?;
    Ok(gt.iter().map(gt_from_words).collect())
}

/// `pairing_batch` on points that are already resident (upload_g1 / upload_g2): one stream, no host traffic besides the result.
pub fn pairing_batch_resident(dev: &Device, dp: &DeviceG1, dq: &DeviceG2) -> Result<Vec<GtOut>, HipError> {
    assert_eq!(dp.n, dq.n);
    let n = dp.n;
    let gt = dev.alloc::<u64>(48 * n)?;
    // SAFETY: point arrays and flags hold n elements, gt 48 * n words.
    device::check(unsafe {
        ffi::sylow_hip_pairing_batch(dp.xy.as_ptr(), dp.inf.as_ptr(), dq.xy.as_ptr(), dq.inf.as_ptr(), gt.as_mut_ptr(), n, dev.stream)
    })?;
    Ok(dev.download_aos::<48>(&gt, n)?.iter().map(gt_from_words).collect())
}

/// `sylow::glued_pairing` (pairing.rs:1029-1037) over the WHOLE slice: prod_i e(p[i], q[i]) with one final exponentiation,
/// spread over the GPU.  Returns the value and `== Gt::identity()`.  `skip_identity = false` replays the reference exactly
/// (a G2 identity zeroes the product); `true` drops identity pairs (EIP-197).
pub fn glued_pairing(dev: &Device, p: &[G1Affine], q: &[G2Affine], skip_identity: bool) -> Result<(GtOut, bool), HipError> {
    assert_eq!(p.len(), q.len());
    let n = p.len();
    let (dp, dq) = (upload_g1(dev, p)?, upload_g2(dev, q)?);
    let (gt, one) = (dev.alloc::<u64>(48)?, dev.alloc::<u8>(1)?);
    // SAFETY: n pairs; gt 48 words; one 1 byte.
    device::check(unsafe {
        ffi::sylow_hip_pairing_product_batch(dp.xy.as_ptr(), dp.inf.as_ptr(), dq.xy.as_ptr(), dq.inf.as_ptr(), n, skip_identity as i32,
                                             gt.as_mut_ptr(), one.as_mut_ptr(), dev.stream)
    })?;
    let words = dev.download_aos::<48>(&gt, 1)?;
    Ok((gt_from_words(&words[0]), dev.download(&one)?[0] != 0))
}

/// Batched `sylow::sign` (lib.rs:179-187): sig[i] = H(msgs[i]) * sk[i], XMD-Keccak256 + SvdW with sylow's DST.
pub fn sign_batch(dev: &Device, sk: &[Fp], msgs: &[&[u8]]) -> Result<Vec<G1Projective>, HipError> {
    assert_eq!(sk.len(), msgs.len());
    let n = sk.len();
    let words: Vec<[u64; 4]> = sk.iter().map(|k| k.value().to_words()).collect();      // fp.rs:232-234
    let d_sk = dev.upload_soa::<4>(&words)?;
    let (d_msgs, d_off) = messages(dev, msgs)?;
    let sig = DeviceG1 { xy: dev.alloc::<u64>(8 * n)?, inf: dev.alloc::<u8>(n)?, n };
    // SAFETY: sk 4 * n words, offsets n + 1 entries into d_msgs, outputs n elements.
    device::check(unsafe {
        ffi::sylow_hip_bls_sign_batch(d_sk.as_ptr(), d_msgs.as_ptr(), d_off.as_ptr(), sig.xy.as_mut_ptr(), sig.inf.as_mut_ptr(), n, dev.stream)
    })?;
    download_g1(dev, &sig)
}

/// Which verification kernel answers `verify_batch`.
#[derive(Clone, Copy, PartialEq, Eq)]
pub enum VerifyMode {
    /// lib.rs:223-236 evaluated literally: pairing(sig, G2gen) == pairing(H(msg), pk), two pairings (~1.5x the time)
    AsWritten,
    /// e(sig, G2gen) * e(-H(msg), pk) == 1, one shared-squaring Miller loop, one final exponentiation: the same boolean for every
    /// input (the library's default `sylow_hip_bls_verify_batch`; the shape sylow's examples recommend)
    Fused,
}

/// Batched `sylow::verify` (lib.rs:223-236): ok[i] = verify(pk[i], msgs[i], sig[i]).  Leaves the flag vector on the device as
/// well (for `all_valid`).
pub fn verify_batch(dev: &Device, pk: &[G2Affine], msgs: &[&[u8]], sig: &[G1Affine], mode: VerifyMode) -> Result<(Vec<bool>, DeviceBuf<u8>), HipError> {
    assert!(pk.len() == msgs.len() && sig.len() == msgs.len());
    let n = msgs.len();
    let (dpk, dsig) = (upload_g2(dev, pk)?, upload_g1(dev, sig)?);
    let (d_msgs, d_off) = messages(dev, msgs)?;
    let ok = dev.alloc::<u8>(n)?;
    // SAFETY: n keys, n signatures, n + 1 offsets, n flags.
    device::check(unsafe {
        match mode {
            VerifyMode::AsWritten => ffi::sylow_hip_bls_verify_two_pairings_batch(dpk.xy.as_ptr(), dpk.inf.as_ptr(), d_msgs.as_ptr(), d_off.as_ptr(),
                                                                                 dsig.xy.as_ptr(), dsig.inf.as_ptr(), ok.as_mut_ptr(), n, dev.stream),
            VerifyMode::Fused => ffi::sylow_hip_bls_verify_batch(dpk.xy.as_ptr(), dpk.inf.as_ptr(), d_msgs.as_ptr(), d_off.as_ptr(),
                                                                 dsig.xy.as_ptr(), dsig.inf.as_ptr(), ok.as_mut_ptr(), n, dev.stream),
        }
    })?;
    let flags = dev.download(&ok)?;
    Ok((flags.iter().map(|&f| f != 0).collect(), ok))
}

/// Batched `sylow::verify` (lib.rs:223-236) on host slices through the library's chunked pipeline (`sylow_hip_bls_verify_host_bytes`):
/// keys and signatures travel in wire format, decoding / validation / hashing / the pairing check of chunk k overlap the copies of
/// its neighbours.  Same booleans as `verify_batch(.., VerifyMode::Fused)`; nothing stays on the device.
pub fn verify_batch_host(pk: &[G2Affine], msgs: &[&[u8]], sig: &[G1Affine]) -> Result<Vec<bool>, HipError> {
    assert!(pk.len() == msgs.len() && sig.len() == msgs.len());
    let n = msgs.len();
    let (mut kb, mut sb) = (Vec::with_capacity(128 * n), Vec::with_capacity(64 * n));
    for x in pk {
        kb.extend_from_slice(&x.to_be_bytes());
    }
    for x in sig {
        sb.extend_from_slice(&x.to_be_bytes());
    }
    let mut offsets = Vec::with_capacity(n + 1);
    let mut blob = Vec::new();
    offsets.push(0u64);
    for m in msgs {
        blob.extend_from_slice(m);
        offsets.push(blob.len() as u64);
    }
    if blob.is_empty() {
        blob.push(0);
    }
    let (mut ok, mut st_k, mut st_s) = (vec![0u8; n], vec![0u8; n], vec![0u8; n]);
    // SAFETY: kb n * 128 bytes, sb n * 64 bytes, offsets n + 1 entries into blob, ok / status n bytes each; host memory, synchronous call.
    device::check(unsafe {
        ffi::sylow_hip_bls_verify_host_bytes(kb.as_ptr(), blob.as_ptr(), offsets.as_ptr(), sb.as_ptr(), ok.as_mut_ptr(), st_k.as_mut_ptr(),
                                             st_s.as_mut_ptr(), n, 0)
    })?;
    first_failure(&st_k)?;
    first_failure(&st_s)?;
    Ok(ok.iter().map(|&f| f != 0).collect())
}

/// One signer, many messages (examples/verify_multiple_messages_same_signer.rs:41-60) with the key's `G2PreComputed` line table
/// CACHED across calls: build it once with `KeyTable::new`, then verify any number of batches against it.
pub struct KeyTable {
    table: DeviceBuf<i32>,
    inf: DeviceBuf<u8>,      // the key's identity flag (1 byte): pairing(_, identity) = 1 upstream, so pair B must be dead for such a key
}
impl KeyTable {
    pub fn new(dev: &Device, pk: &G2Affine) -> Result<Self, HipError> {
        let dpk = upload_g2(dev, std::slice::from_ref(pk))?;
        // SAFETY: no arguments.
        let words = unsafe { ffi::sylow_hip_g2_line_table_words() } as usize;
        let table = dev.alloc::<i32>(words)?;
        // SAFETY: dpk is a 1-element SoA array; table holds `words` int32.
        device::check(unsafe { ffi::sylow_hip_g2_line_table(dpk.xy.as_ptr(), 1, 0, table.as_mut_ptr(), dev.stream) })?;
        dev.sync()?;
        Ok(KeyTable { table, inf: dpk.inf })
    }

    pub fn verify_batch(&self, dev: &Device, msgs: &[&[u8]], sig: &[G1Affine]) -> Result<Vec<bool>, HipError> {
        assert_eq!(msgs.len(), sig.len());
        let n = msgs.len();
        let dsig = upload_g1(dev, sig)?;
        let (d_msgs, d_off) = messages(dev, msgs)?;
        let ok = dev.alloc::<u8>(n)?;
        // SAFETY: table built by KeyTable::new on this device; n signatures, n + 1 offsets, n flags; pk_inf = the key's 1-byte flag.
        device::check(unsafe {
            ffi::sylow_hip_bls_verify_line_table_batch(self.table.as_ptr(), self.inf.as_ptr(), d_msgs.as_ptr(), d_off.as_ptr(), dsig.xy.as_ptr(),
                                                       dsig.inf.as_ptr(), ok.as_mut_ptr(), n, dev.stream)
        })?;
        Ok(dev.download(&ok)?.iter().map(|&f| f != 0).collect())
    }
}

/// "Are ALL signatures of the sharded batch valid?" -- one boolean from every GPU of the node.  Each rank (one process per GPU)
/// passes the flag vector of ITS shard and the node's RCCL communicator (`ncclComm_t` as a raw pointer; null = single rank):
/// the flags are AND-ed on the device and the 4-byte word is MIN-reduced over xGMI.  Must be called by every rank.
pub fn all_valid(dev: &Device, flags: &DeviceBuf<u8>, comm: *mut c_void) -> Result<bool, HipError> {
    let out = dev.alloc::<i32>(1)?;
    // SAFETY: flags holds flags.len bytes; out one int32; comm is null or a live communicator spanning the calling ranks.
    device::check(unsafe { ffi::sylow_hip_all_valid(flags.as_ptr(), flags.len, comm, out.as_mut_ptr(), dev.stream) })?;
    Ok(dev.download(&out)?[0] == 1)
}

/// `glued_pairing` over the union of all ranks' pairs: every rank contributes the Miller product of its shard (up to a factor in Fp*), the 384-byte
/// partials are all-gathered, each rank multiplies them and runs ONE final exponentiation.  Every rank gets the same answer.
pub fn glued_pairing_all(dev: &Device, p: &[G1Affine], q: &[G2Affine], skip_identity: bool, comm: *mut c_void) -> Result<(GtOut, bool), HipError> {
    assert_eq!(p.len(), q.len());
    let n = p.len();
    let (dp, dq) = (upload_g1(dev, p)?, upload_g2(dev, q)?);
    let (gt, one) = (dev.alloc::<u64>(48)?, dev.alloc::<u8>(1)?);
    // SAFETY: n pairs; gt 48 words; one 1 byte; comm as for all_valid.
    device::check(unsafe {
        ffi::sylow_hip_pairing_product_all(dp.xy.as_ptr(), dp.inf.as_ptr(), dq.xy.as_ptr(), dq.inf.as_ptr(), n, skip_identity as i32, comm,
                                           gt.as_mut_ptr(), one.as_mut_ptr(), dev.stream)
    })?;
    let words = dev.download_aos::<48>(&gt, 1)?;
    Ok((gt_from_words(&words[0]), dev.download(&one)?[0] != 0))
}

/// Aggregate verification -- the batch shape of examples/verify_multiple_messages_same_signer.rs:41-60 and
/// threshold_signing.rs:92-121: is the product of the 2n pairs (sig_i, G2gen), (-H(msg_i), pk_i) the identity?  The signatures are
/// summed in G1 first (prod_i e(sig_i, G2gen) = e(sum_i sig_i, G2gen)); `pk` holds one key per message, or ONE key for the whole
/// batch (then the hashes are summed as well).  Returns the Gt value of the product and the boolean.  `comm`: as for `all_valid`
/// (every rank passes its shard of a batch spread over the GPUs of the node; null = this process alone).
pub fn aggregate_verify(dev: &Device, pk: &[G2Affine], msgs: &[&[u8]], sig: &[G1Affine], comm: *mut c_void) -> Result<(GtOut, bool), HipError> {
    assert!(sig.len() == msgs.len() && (pk.len() == msgs.len() || pk.len() == 1));
    let n = msgs.len();
    let (dpk, dsig) = (upload_g2(dev, pk)?, upload_g1(dev, sig)?);
    let (d_msgs, d_off) = messages(dev, msgs)?;
    let (gt, one) = (dev.alloc::<u64>(48)?, dev.alloc::<u8>(1)?);
    // SAFETY: pk.len() keys, n signatures, n + 1 offsets; gt 48 words; one 1 byte; comm null or a live communicator.
    device::check(unsafe {
        ffi::sylow_hip_bls_aggregate_verify_batch(dpk.xy.as_ptr(), dpk.inf.as_ptr(), pk.len(), d_msgs.as_ptr(), d_off.as_ptr(),
                                                  dsig.xy.as_ptr(), dsig.inf.as_ptr(), n, comm, gt.as_mut_ptr(), one.as_mut_ptr(), dev.stream)
    })?;
    let words = dev.download_aos::<48>(&gt, 1)?;
    Ok((gt_from_words(&words[0]), dev.download(&one)?[0] != 0))
}

/// A batch of cached `G2PreComputed` tables (pairing.rs:556) resident on the device, and the two loops that consume them:
/// `G2PreComputed::miller_loop(&G1Affine)` (pairing.rs:590-619) and `glued_miller_loop` (pairing.rs:970-1022).
pub struct PrecomputedG2 {
    coeffs: DeviceBuf<u64>,   // [87 * 24][m], canonical words
    m: usize,
}
impl PrecomputedG2 {
    /// `G2Affine::precompute` (pairing.rs:676-708) for every point of the slice.
    pub fn new(dev: &Device, q: &[G2Affine]) -> Result<Self, HipError> {
        let dq = upload_g2(dev, q)?;
        let coeffs = dev.alloc::<u64>(87 * 24 * q.len())?;
        // SAFETY: dq holds q.len() points; coeffs 87 * 24 * q.len() words.
        device::check(unsafe { ffi::sylow_hip_g2_precompute_batch(dq.xy.as_ptr(), coeffs.as_mut_ptr(), q.len(), dev.stream) })?;
        dev.sync()?;
        Ok(PrecomputedG2 { coeffs, m: q.len() })
    }

    /// Raw Miller values f[i] = table[table_idx[i]].miller_loop(p[i]) as 48 canonical words each (MillerLoopResult is not
    /// constructible outside sylow either; feed them to `final_exponentiation_batch`).
    pub fn miller_loop(&self, dev: &Device, p: &[G1Affine], table_idx: &[u64]) -> Result<DeviceBuf<u64>, HipError> {
        assert_eq!(p.len(), table_idx.len());
        assert!(table_idx.iter().all(|&t| (t as usize) < self.m));
        let n = p.len();
        let dp = upload_g1(dev, p)?;
        let d_idx = dev.upload(table_idx)?;
        let f = dev.alloc::<u64>(48 * n)?;
        // SAFETY: m tables, n indices < m, n points, f 48 * n words.
        device::check(unsafe {
            ffi::sylow_hip_miller_loop_precomputed_batch(self.coeffs.as_ptr(), self.m, d_idx.as_ptr(), dp.xy.as_ptr(), f.as_mut_ptr(), n, dev.stream)
        })?;
        Ok(f)
    }
}

/// `MillerLoopResult::final_exponentiation` (pairing.rs:245-492) on device-resident raw Miller values.
pub fn final_exponentiation_batch(dev: &Device, f: &DeviceBuf<u64>) -> Result<Vec<GtOut>, HipError> {
    let n = f.len / 48;
    let gt = dev.alloc::<u64>(48 * n)?;
    // SAFETY: f and gt hold 48 * n words.
    device::check(unsafe { ffi::sylow_hip_final_exp_batch(f.as_ptr(), gt.as_mut_ptr(), n, dev.stream) })?;
    Ok(dev.download_aos::<48>(&gt, n)?.iter().map(gt_from_words).collect())
}

/// Release the library's scratch blocks and generator tables on every device (process teardown; the library stays usable).
pub fn shutdown() -> Result<(), Error> {
    // SAFETY: no arguments.
    device::check(unsafe { ffi::sylow_hip_shutdown() })
}
