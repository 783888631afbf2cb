//! The reference's surface (`verifier/src/lib.rs:29-75`) over the MI355X library, plus `verify_batch`.
//!
//! * `Groth16Verifier::verify(proof, vk, public_inputs)` -> `Result<bool, Groth16Error>`: `Ok(true)` / `Ok(false)` / `Err(PrepareInputsFailed)`
//!   (`groth16/verify.rs:54-56,77`); everything the reference turns into a panic through `unwrap` (`lib.rs:45-46`) panics here too.
//! * `PlonkVerifier::verify` -> `Result<bool, PlonkError>`: never `Ok(false)` (`plonk/verify.rs:316`).
//! * `verify_batch`: one status per proof, nothing panics; `PreparedGroth16Vk` / `PreparedPlonkVk` keep the key work that the reference repeats on every call.
//! Public inputs are 32-byte big-endian values used modulo r, as `bn::Fr::from_slice` + `AffineG1 * Fr` use them.
//! NOT COMPILED in the repository's build image (no Rust toolchain there): see ../README.md.
use bn254_verify_amd_sys as sys;
use core::ffi::{c_int, c_void, CStr};

/// `groth16/error.rs:4-15`: the one error `verify_groth16` returns.
#[derive(Debug, Clone, Copy, PartialEq, Eq)]
pub enum Groth16Error { PrepareInputsFailed }
/// `plonk/error.rs`: the errors `verify_plonk` returns.
#[derive(Debug, Clone, Copy, PartialEq, Eq)]
pub enum PlonkError { InvalidWitness, OpeningPolyMismatch, PairingCheckFailed, Bsb22CommitmentMismatch, InverseNotFound }

/// Outcome of one proof of a batch: the reference's `Result`, with its panics as a value.
#[derive(Debug, Clone, Copy, PartialEq, Eq)]
pub enum Status {
    Accept, Reject, NotMember, NotOnCurve, NotInSubgroup, InputLen, Malformed, OpeningMismatch, PairingFailed, Bsb22Mismatch, Inverse, Unknown(u8),
}
impl From<u8> for Status {
    fn from(s: u8) -> Self {
        match s {
            sys::BN254_ACCEPT => Status::Accept, sys::BN254_REJECT => Status::Reject, sys::BN254_ERR_NOT_MEMBER => Status::NotMember,
            sys::BN254_ERR_NOT_ON_CURVE => Status::NotOnCurve, sys::BN254_ERR_NOT_IN_SUBGROUP => Status::NotInSubgroup, sys::BN254_ERR_INPUT_LEN => Status::InputLen,
            sys::BN254_ERR_MALFORMED => Status::Malformed, sys::BN254_ERR_OPENING_MISMATCH => Status::OpeningMismatch, sys::BN254_ERR_PAIRING_FAILED => Status::PairingFailed,
            sys::BN254_ERR_BSB22_MISMATCH => Status::Bsb22Mismatch, sys::BN254_ERR_INVERSE => Status::Inverse, x => Status::Unknown(x),
        }
    }
}
/// Infrastructure failure (no device, HIP error, bad argument): `rc` of the C ABI with `bn254_last_error()`.
#[derive(Debug, Clone)]
pub struct Error { pub code: c_int, pub message: String }
fn check(rc: c_int) -> Result<(), Error> {
    if rc == sys::BN254_OK { return Ok(()); }
    let message = unsafe { CStr::from_ptr(sys::bn254_last_error()) }.to_string_lossy().into_owned();
    Err(Error { code: rc, message })
}

/// Which reading of a compressed G2 point the key loader uses: the reference's literal function, or gnark's.
#[derive(Debug, Clone, Copy, PartialEq, Eq)]
pub enum VkMode { Reference = 0, Gnark = 1 }

pub const FLAG_STRICT_SCALARS: u32 = sys::BN254_FLAG_STRICT_SCALARS as u32;
pub const FLAG_RLC: u32 = sys::BN254_FLAG_RLC as u32;

fn flatten(proofs: &[&[u8]], min_stride: usize) -> (Vec<u8>, usize) {
    let stride = proofs.iter().map(|p| p.len()).max().unwrap_or(min_stride).max(min_stride);
    let mut buf = vec![0u8; stride * proofs.len()];
    for (i, p) in proofs.iter().enumerate() { buf[i * stride..i * stride + p.len()].copy_from_slice(p); }
    (buf, stride)
}

/// `load_groth16_verifying_key_from_bytes` + `pairing(alpha, beta)` (`groth16/converter.rs:28-89`, `groth16/verify.rs:70`), done once.
pub struct PreparedGroth16Vk { h: *mut sys::Bn254G16Pvk }
unsafe impl Send for PreparedGroth16Vk {}
unsafe impl Sync for PreparedGroth16Vk {}   // the handle is immutable; the library serialises batches per (key, device)
impl PreparedGroth16Vk {
    pub fn new(vk: &[u8], mode: VkMode) -> Result<Self, Error> {
        let mut h = core::ptr::null_mut();
        check(unsafe { sys::bn254_groth16_vk_prepare(vk.as_ptr(), vk.len(), mode as u32, &mut h) })?;
        Ok(Self { h })
    }
    pub fn num_public(&self) -> usize { unsafe { sys::bn254_groth16_vk_num_public(self.h) } }
    /// Allocations ahead of the first batch (otherwise the first call makes them).
    pub fn reserve(&self, n: usize, device: i32) -> Result<(), Error> { check(unsafe { sys::bn254_groth16_reserve(self.h, n, device) }) }
    /// `n` proofs, `proof_stride` bytes apart (>= 256: A | B | C as gnark writes them), `n_public` 32-byte inputs each; host buffers in, status bytes out.
    pub fn verify_batch_raw(&self, proofs: &[u8], proof_stride: usize, public_inputs: &[u8], n_public: usize, n: usize, device: i32, flags: u32) -> Result<Vec<Status>, Error> {
        assert!(proof_stride >= 256 && proofs.len() >= n * proof_stride && public_inputs.len() >= n * n_public * 32);
        let mut st = vec![0u8; n];
        check(unsafe { sys::bn254_groth16_verify_batch(self.h, proofs.as_ptr(), proof_stride, public_inputs.as_ptr(), n_public, n, st.as_mut_ptr(), device, flags) })?;
        Ok(st.into_iter().map(Status::from).collect())
    }
    /// The same over the GPUs selected by `device_mask` (contiguous shards, one host thread per device: SURVEY.md section 8(e)).
    pub fn verify_batch_multi_raw(&self, proofs: &[u8], proof_stride: usize, public_inputs: &[u8], n_public: usize, n: usize, device_mask: u64, flags: u32) -> Result<Vec<Status>, Error> {
        assert!(proof_stride >= 256 && proofs.len() >= n * proof_stride && public_inputs.len() >= n * n_public * 32);
        let mut st = vec![0u8; n];
        check(unsafe { sys::bn254_groth16_verify_batch_multi(self.h, proofs.as_ptr(), proof_stride, public_inputs.as_ptr(), n_public, n, st.as_mut_ptr(), device_mask, flags) })?;
        Ok(st.into_iter().map(Status::from).collect())
    }
    /// Proofs and inputs already in device memory; enqueues on `hip_stream` and returns (BN254_FLAG_RLC: after one stream synchronisation).
    /// # Safety
    /// The three device pointers must be valid for the sizes implied by `n`, `proof_stride`, `n_public` until the stream has run the batch.
    pub unsafe fn verify_batch_device(&self, d_proofs: *const c_void, proof_stride: usize, d_inputs: *const c_void, n_public: usize, n: usize, d_status: *mut c_void, device: i32,
                                      hip_stream: *mut c_void, flags: u32) -> Result<(), Error> {
        check(sys::bn254_groth16_verify_batch_device(self.h, d_proofs, proof_stride, d_inputs, n_public, n, d_status, device, hip_stream, flags))
    }
}
impl Drop for PreparedGroth16Vk { fn drop(&mut self) { unsafe { sys::bn254_groth16_vk_free(self.h) } } }

pub struct Groth16Verifier;
impl Groth16Verifier {
    /// `Groth16Verifier::verify` (`lib.rs:44-49`).  Panics where the reference panics (loader errors through `unwrap`, short buffers, flag `0b00`).
    pub fn verify(proof: &[u8], vk: &[u8], public_inputs: &[[u8; 32]]) -> Result<bool, Groth16Error> {
        let inputs: Vec<u8> = public_inputs.iter().flatten().copied().collect();
        let mut st = 0u8;
        let rc = unsafe { sys::bn254_groth16_verify(proof.as_ptr(), proof.len(), vk.as_ptr(), vk.len(), inputs.as_ptr(), public_inputs.len(), sys::BN254_VK_REFERENCE, &mut st) };
        check(rc).expect("bn254 infrastructure error");
        match Status::from(st) {
            Status::Accept => Ok(true),
            Status::Reject => Ok(false),
            Status::InputLen => Err(Groth16Error::PrepareInputsFailed),
            s => panic!("loader error {s:?}"),           // lib.rs:45-46: unwrap() of Field / Group / InvalidPoint errors
        }
    }
    /// New: N proofs against one key, one `Status` each; nothing panics.
    pub fn verify_batch(proofs: &[&[u8]], vk: &[u8], public_inputs: &[&[[u8; 32]]]) -> Result<Vec<Status>, Error> {
        assert_eq!(proofs.len(), public_inputs.len());
        let pvk = PreparedGroth16Vk::new(vk, VkMode::Reference)?;
        let (buf, stride) = flatten(proofs, 256);
        let n_public = public_inputs.first().map_or(0, |x| x.len());
        assert!(public_inputs.iter().all(|x| x.len() == n_public), "one input count per batch (a wrong count is a per-key error: InputLen for every proof)");
        let inputs: Vec<u8> = public_inputs.iter().flat_map(|xs| xs.iter().flatten().copied()).collect();
        let mut st = pvk.verify_batch_raw(&buf, stride, &inputs, n_public, proofs.len(), 0, 0)?;
        for (s, p) in st.iter_mut().zip(proofs) { if p.len() < 256 { *s = Status::Malformed; } }   // a slice-index panic in the reference
        Ok(st)
    }
}

/// `load_plonk_verifying_key_from_bytes` (`plonk/converter.rs:18-119`) + the key-side tables, done once.
pub struct PreparedPlonkVk { h: *mut sys::Bn254PlonkPvk }
unsafe impl Send for PreparedPlonkVk {}
unsafe impl Sync for PreparedPlonkVk {}     // several threads may call verify_batch on one key: the library hands out per-call contexts
impl PreparedPlonkVk {
    pub fn new(vk: &[u8]) -> Result<Self, Error> {
        let mut h = core::ptr::null_mut();
        check(unsafe { sys::bn254_plonk_vk_prepare(vk.as_ptr(), vk.len(), &mut h) })?;
        Ok(Self { h })
    }
    pub fn num_public(&self) -> usize { unsafe { sys::bn254_plonk_vk_num_public(self.h) } }
    /// Allocations ahead of the first batch of up to `n` proofs (`proof_stride` > 0: also the pinned staging of the host-buffer entry); see the header for the footprint.
    pub fn reserve(&self, n: usize, proof_stride: usize, device: i32) -> Result<(), Error> { check(unsafe { sys::bn254_plonk_reserve(self.h, n, proof_stride, device) }) }
    /// Device memory this key's contexts hold on `device` (bytes) and how many contexts hold any.
    pub fn footprint(&self, device: i32) -> Result<(usize, i32), Error> {
        let (mut bytes, mut ctx) = (0usize, 0);
        check(unsafe { sys::bn254_plonk_footprint(self.h, device, &mut bytes, &mut ctx) })?;
        Ok((bytes, ctx))
    }
    pub fn verify_batch_raw(&self, proofs: &[u8], proof_stride: usize, public_inputs: &[u8], n_public: usize, n: usize, device: i32) -> Result<Vec<Status>, Error> {
        self.verify_batch_flags(proofs, proof_stride, public_inputs, n_public, n, device, 0)
    }
    /// `flags`: `sys::BN254_FLAG_RLC` batches the pairing checks of a pass across proofs (one check per 64 proofs, exact fallback on the groups that fail;
    /// honoured from 8192 proofs per pass).  Same status bytes as `verify_batch_raw`.
    pub fn verify_batch_flags(&self, proofs: &[u8], proof_stride: usize, public_inputs: &[u8], n_public: usize, n: usize, device: i32, flags: u32) -> Result<Vec<Status>, Error> {
        assert!(proofs.len() >= n * proof_stride && public_inputs.len() >= n * n_public * 32);
        let mut st = vec![0u8; n];
        check(unsafe { sys::bn254_plonk_verify_batch_flags(self.h, proofs.as_ptr(), proof_stride, public_inputs.as_ptr(), n_public, n, st.as_mut_ptr(), device, flags) })?;
        Ok(st.into_iter().map(Status::from).collect())
    }
    /// The same over the GPUs selected by `device_mask` (contiguous shards, one host thread per device: SURVEY.md section 8(e)).
    pub fn verify_batch_multi_raw(&self, proofs: &[u8], proof_stride: usize, public_inputs: &[u8], n_public: usize, n: usize, device_mask: u64, flags: u32) -> Result<Vec<Status>, Error> {
        assert!(proofs.len() >= n * proof_stride && public_inputs.len() >= n * n_public * 32);
        let mut st = vec![0u8; n];
        check(unsafe { sys::bn254_plonk_verify_batch_multi(self.h, proofs.as_ptr(), proof_stride, public_inputs.as_ptr(), n_public, n, st.as_mut_ptr(), device_mask, flags) })?;
        Ok(st.into_iter().map(Status::from).collect())
    }
    /// Proofs, inputs and status bytes in device memory.  Host-synchronous: waits for `hip_stream`, returns when the status bytes are in `d_status`.
    /// # Safety
    /// The three device pointers must be valid for the sizes implied by `n`, `proof_stride`, `n_public` for the duration of the call.
    pub unsafe fn verify_batch_device(&self, d_proofs: *const c_void, proof_stride: usize, d_inputs: *const c_void, n_public: usize, n: usize, d_status: *mut c_void, device: i32,
                                      hip_stream: *mut c_void, flags: u32) -> Result<(), Error> {
        check(sys::bn254_plonk_verify_batch_device(self.h, d_proofs, proof_stride, d_inputs, n_public, n, d_status, device, hip_stream, flags))
    }
    /// Durations (ms) of the first sub-batch of the batch that finished last on `device` (slots: `bn254_plonk_last_timing` in the header) and the lanes of its two MSM launches.
    pub fn last_timing(&self, device: i32) -> Result<([f32; sys::BN254_PLONK_NUM_TIMINGS], [usize; 2]), Error> {
        let (mut ms, mut lanes) = ([0f32; sys::BN254_PLONK_NUM_TIMINGS], [0usize; 2]);
        check(unsafe { sys::bn254_plonk_last_timing(self.h, device, ms.as_mut_ptr(), lanes.as_mut_ptr()) })?;
        Ok((ms, lanes))
    }
}
impl Drop for PreparedPlonkVk { fn drop(&mut self) { unsafe { sys::bn254_plonk_vk_free(self.h) } } }

/// The library this crate was generated for?  (`bn254_abi_version`: bumped whenever a function changes its arguments, an array its length or a slot its meaning.)
pub fn abi_matches() -> bool { unsafe { sys::bn254_abi_version() == sys::BN254_ABI_VERSION } }

pub struct PlonkVerifier;
impl PlonkVerifier {
    /// `PlonkVerifier::verify` (`lib.rs:69-74`).
    pub fn verify(proof: &[u8], vk: &[u8], public_inputs: &[[u8; 32]]) -> Result<bool, PlonkError> {
        let inputs: Vec<u8> = public_inputs.iter().flatten().copied().collect();
        let mut st = 0u8;
        let rc = unsafe { sys::bn254_plonk_verify(proof.as_ptr(), proof.len(), vk.as_ptr(), vk.len(), inputs.as_ptr(), public_inputs.len(), &mut st) };
        check(rc).expect("bn254 infrastructure error");
        match Status::from(st) {
            Status::Accept => Ok(true),                                       // never Ok(false): plonk/verify.rs:316
            Status::InputLen => Err(PlonkError::InvalidWitness),              // plonk/verify.rs:57-59
            Status::OpeningMismatch => Err(PlonkError::OpeningPolyMismatch),  // plonk/verify.rs:212-214
            Status::PairingFailed => Err(PlonkError::PairingCheckFailed),     // plonk/kzg.rs:185-187
            Status::Bsb22Mismatch => Err(PlonkError::Bsb22CommitmentMismatch),// plonk/verify.rs:52-54
            Status::Inverse => Err(PlonkError::InverseNotFound),              // plonk/verify.rs:106
            s => panic!("loader error {s:?}"),                                // lib.rs:70-71
        }
    }
    pub fn verify_batch(proofs: &[&[u8]], vk: &[u8], public_inputs: &[&[[u8; 32]]]) -> Result<Vec<Status>, Error> {
        assert_eq!(proofs.len(), public_inputs.len());
        let pvk = PreparedPlonkVk::new(vk)?;
        let (buf, stride) = flatten(proofs, 516);
        let n_public = public_inputs.first().map_or(0, |x| x.len());
        assert!(public_inputs.iter().all(|x| x.len() == n_public), "one input count per batch (a wrong count is a per-key error: InputLen for every proof)");
        let inputs: Vec<u8> = public_inputs.iter().flat_map(|xs| xs.iter().flatten().copied()).collect();
        let mut st = pvk.verify_batch_raw(&buf, stride, &inputs, n_public, proofs.len(), 0)?;
        // a proof shorter than its own layout is a slice-index panic in the reference (plonk/converter.rs:121-178); zero-padded to the stride it would parse as something else
        for (s, p) in st.iter_mut().zip(proofs) { if p.len() < plonk_layout_len(p) { *s = Status::Malformed; } }
        Ok(st)
    }
}

/// Bytes `load_plonk_proof_from_bytes` reads of a proof that starts like `p` (`plonk/converter.rs:121-178`): 8 points, u32 count + claimed values, the second opening
/// (point, value, u32 count) and the BSB22 commitments; `usize::MAX` when `p` ends before a count can be read.
fn plonk_layout_len(p: &[u8]) -> usize {
    let be32 = |o: usize| p.get(o..o + 4).map(|b| u32::from_be_bytes([b[0], b[1], b[2], b[3]]) as usize);
    let Some(n_claimed) = be32(512) else { return usize::MAX };
    let off = 516 + 32 * n_claimed;
    let Some(n_bsb) = be32(off + 96) else { return usize::MAX };
    off + 100 + 64 * n_bsb
}

/// The contiguous shard of rank `r` of `world` for a batch of `n` (the rule of `bn254_shard_plan` and of the multi-process job).
pub fn shard_bounds(n: usize, world: usize, r: usize) -> (usize, usize) {
    let (base, rem) = (n / world, n % world);
    (r * base + r.min(rem), base + usize::from(r < rem))
}
/// Multi-process job, one process per GPU: one `ncclAllGather` of the ranks' status bytes (RCCL over xGMI); `comm` is the host's `ncclComm_t`.
/// # Safety
/// Device pointers and the communicator must be valid; `d_full` holds `n` bytes, `d_scratch` `world * ceil(n / world)` when `n % world != 0`.
pub unsafe fn status_all_gather(comm: *mut c_void, world: i32, rank: i32, d_local: *const c_void, n: usize, d_full: *mut c_void, d_scratch: *mut c_void, hip_stream: *mut c_void) -> Result<(), Error> {
    check(sys::bn254_status_all_gather(comm, world, rank, d_local, n, d_full, d_scratch, hip_stream))
}

#[cfg(test)]
mod tests {
    //! Needs an MI355X (the library has no CPU path) and the repository's `tests/golden/` next to `rust/`.
    use super::*;
    use std::{fs, path::PathBuf};
    fn golden(p: &str) -> Vec<u8> { fs::read(PathBuf::from(env!("CARGO_MANIFEST_DIR")).join("../../tests/golden").join(p)).expect(p) }
    fn sp1(name: &str) -> (i32, Vec<u8>, [[u8; 32]; 2]) {
        let b = golden(&format!("sp1/{name}"));
        let (mut variant, mut raw, mut raw_len, mut pi, mut vh) = (0, vec![0u8; 2048], 0usize, [0u8; 64], [0u8; 32]);
        assert_eq!(unsafe { sys::bn254_sp1_fixture_parse(b.as_ptr(), b.len(), &mut variant, raw.as_mut_ptr(), raw.len(), &mut raw_len, pi.as_mut_ptr(), vh.as_mut_ptr()) }, 0);
        raw.truncate(raw_len);
        let mut inputs = [[0u8; 32]; 2];
        inputs[0].copy_from_slice(&pi[..32]); inputs[1].copy_from_slice(&pi[32..]);
        (variant, raw, inputs)
    }
    /// `test_programs` of the reference (`examples/script/src/main.rs:182-245`), PlonK half: the four fixtures verify against the key recovered from the guest ELF.
    #[test]
    fn reference_plonk_fixtures_verify() {
        let vk = golden("plonk_vk.bin");
        for f in ["fibonacci_plonk_proof.bin", "is-prime_plonk_proof.bin", "sha2_plonk_proof.bin", "tendermint_plonk_proof.bin"] {
            let (variant, proof, inputs) = sp1(f);
            assert_eq!(variant, 2);
            assert_eq!(PlonkVerifier::verify(&proof, &vk, &inputs), Ok(true), "{f}");
            let mut bad = inputs; bad[0][31] ^= 1;
            assert_eq!(PlonkVerifier::verify(&proof, &vk, &bad), Err(PlonkError::OpeningPolyMismatch), "{f}");
        }
    }
    /// A synthetic Groth16 batch from the library's generator: statuses as predicted, `verify` and `verify_batch` agree.
    #[test]
    fn groth16_batch_matches_generator() {
        let (n_public, n) = (2usize, 512usize);
        let mut vk = vec![0u8; unsafe { sys::bn254_synth_groth16_vk_len(n_public) }];
        let (mut proofs, mut inputs, mut expected) = (vec![0u8; 256 * n], vec![0u8; 32 * n_public * n], vec![0u8; n]);
        assert_eq!(unsafe { sys::bn254_synth_groth16(0xB2540000, n_public, n, 8, 1, 0, vk.as_mut_ptr(), proofs.as_mut_ptr(), inputs.as_mut_ptr(), expected.as_mut_ptr()) }, 0);
        let pvk = PreparedGroth16Vk::new(&vk, VkMode::Reference).unwrap();
        assert_eq!(pvk.num_public(), n_public);
        let st = pvk.verify_batch_raw(&proofs, 256, &inputs, n_public, n, 0, 0).unwrap();
        assert!(st.iter().zip(&expected).all(|(s, e)| *s == Status::from(*e)));
        let rlc = pvk.verify_batch_raw(&proofs, 256, &inputs, n_public, n, 0, FLAG_RLC).unwrap();
        assert_eq!(st, rlc);
        let one: Vec<[u8; 32]> = inputs[..64].chunks(32).map(|c| c.try_into().unwrap()).collect();
        assert_eq!(Groth16Verifier::verify(&proofs[..256], &vk, &one), Ok(true));
        assert_eq!(Groth16Verifier::verify(&proofs[..256], &vk, &one[..1]), Err(Groth16Error::PrepareInputsFailed));
        assert_eq!(shard_bounds(1 << 20, 8, 3), (3 << 17, 1 << 17));
    }
}
