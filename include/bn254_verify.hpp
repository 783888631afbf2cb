// bn254_verify.hpp -- the host side of the drop-in, in C++: the crate's surface (verifier/src/lib.rs:27-74) over the C ABI of bn254_verify.h.
//
// The reference is a Rust crate; a Rust host binds the C ABI directly.  This header is the same surface for a C++ host and
// the place where the mapping "status byte -> what the reference does" is written down once:
//
//   reference (Rust)                                                        here (C++17, header only)
//   Groth16Verifier::verify(proof, vk, public_inputs) -> Result<bool, _>    snark_bn254_verifier::Groth16Verifier::verify(...) -> Result<bool, Groth16Error>
//   PlonkVerifier::verify(proof, vk, public_inputs) -> Result<bool, _>      snark_bn254_verifier::PlonkVerifier::verify(...)   -> Result<bool, PlonkError>
//   (new) verify_batch(&[proof], &vk, &[[Fr]])                              Groth16Verifier::verify_batch / PlonkVerifier::verify_batch -> status bytes
//   public_inputs: &[bn::Fr] built with Fr::from_slice(32 big-endian bytes) Fr = std::array<uint8_t, 32>, big-endian
//   .unwrap() of a loader error (lib.rs:45-46, 70-71): a panic              throws Panic (carries the status byte that names the loader error)
//   Ok(true) / Ok(false) / Err(e)                                           Result::ok() + value / error()
//
// Groth16 (groth16/verify.rs:53-78): ACCEPT = Ok(true), REJECT = Ok(false), ERR_INPUT_LEN = Err(PrepareInputsFailed); proof coordinates that
// are no field members / not on the curve / B outside G2 / short buffers are the loaders' errors, i.e. panics.
// PlonK (plonk/verify.rs:46-317, kzg.rs:180-187): ACCEPT = Ok(true); the verifier never answers Ok(false): a failed check is
// Err(OpeningPolyMismatch | PairingCheckFailed | Bsb22CommitmentMismatch | InverseNotFound | InvalidWitness).
//
// Everything runs on the GPU through libbn254_verify_amd.so; without a usable device the calls throw InfrastructureError (BN254_E_NO_DEVICE):
// there is no CPU path.
#pragma once
#include <array>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>
#include "bn254_verify.h"

namespace snark_bn254_verifier {

using Fr = std::array<uint8_t, 32>;   // what bn::Fr::from_slice reads: 32 big-endian bytes (any 256-bit value; used modulo r, examples/script/src/main.rs:204-213)
using Bytes = std::vector<uint8_t>;

// groth16/error.rs
enum class Groth16Error { ProofVerificationFailed, ProcessVerifyingKeyFailed, PrepareInputsFailed, UnexpectedIdentity, GeneralError };
// plonk/error.rs: the variants verify_plonk can return on well-formed buffers
enum class PlonkError { Bsb22CommitmentMismatch, InverseNotFound, InvalidWitness, OpeningPolyMismatch, PairingCheckFailed, GeneralError };

inline const char* to_string(Groth16Error e) {
  switch (e) {
    case Groth16Error::ProofVerificationFailed: return "Proof verification failed";
    case Groth16Error::ProcessVerifyingKeyFailed: return "Process verifying key failed";
    case Groth16Error::PrepareInputsFailed: return "Prepare inputs failed";
    case Groth16Error::UnexpectedIdentity: return "Unexpected identity";
    default: return "General error";
  }
}
inline const char* to_string(PlonkError e) {
  switch (e) {
    case PlonkError::Bsb22CommitmentMismatch: return "BSB22 Commitment number mismatch";
    case PlonkError::InverseNotFound: return "Inverse not found";
    case PlonkError::InvalidWitness: return "Invalid witness";
    case PlonkError::OpeningPolyMismatch: return "Opening linear polynomial mismatch";
    case PlonkError::PairingCheckFailed: return "Pairing check failed";
    default: return "General error";
  }
}

// where the reference unwrap()s a loader error: the status byte says which (BN254_ERR_NOT_MEMBER, _NOT_ON_CURVE, _NOT_IN_SUBGROUP, _MALFORMED)
struct Panic : std::runtime_error {
  int status;
  explicit Panic(int st) : std::runtime_error(std::string("called `Result::unwrap()` on an `Err` value: ") + bn254_status_string(st)), status(st) {}
};
// the library could not run (no device, HIP error, bad argument): not an outcome of the proof
struct InfrastructureError : std::runtime_error {
  int code;
  explicit InfrastructureError(int c) : std::runtime_error(std::string("bn254_verify: ") + bn254_last_error()), code(c) {}
};

template <class T, class E>
class Result {   // Result<bool, Groth16Error> / Result<bool, PlonkError>
 public:
  static Result Ok(T v) { Result r; r.ok_ = true; r.value_ = v; return r; }
  static Result Err(E e) { Result r; r.ok_ = false; r.err_ = e; return r; }
  bool ok() const { return ok_; }
  bool is_err() const { return !ok_; }
  T value() const { if (!ok_) throw std::logic_error("Result: value() on Err"); return value_; }
  E error() const { if (ok_) throw std::logic_error("Result: error() on Ok"); return err_; }
  T unwrap() const { if (!ok_) throw std::runtime_error(std::string("called `Result::unwrap()` on an `Err` value: ") + to_string(err_)); return value_; }

 private:
  bool ok_ = false;
  T value_{};
  E err_{};
};

namespace detail {
inline void check(int rc) { if (rc != BN254_OK) throw InfrastructureError(rc); }
inline Bytes flatten(const std::vector<Fr>& v) { Bytes b(32 * v.size()); for (size_t i = 0; i < v.size(); i++) std::copy(v[i].begin(), v[i].end(), b.begin() + 32 * i); return b; }
inline Result<bool, Groth16Error> groth16_outcome(int st) {
  switch (st) {
    case BN254_ACCEPT: return Result<bool, Groth16Error>::Ok(true);
    case BN254_REJECT: return Result<bool, Groth16Error>::Ok(false);
    case BN254_ERR_INPUT_LEN: return Result<bool, Groth16Error>::Err(Groth16Error::PrepareInputsFailed);
    default: throw Panic(st);
  }
}
inline Result<bool, PlonkError> plonk_outcome(int st) {
  switch (st) {
    case BN254_ACCEPT: return Result<bool, PlonkError>::Ok(true);
    case BN254_ERR_OPENING_MISMATCH: return Result<bool, PlonkError>::Err(PlonkError::OpeningPolyMismatch);
    case BN254_ERR_PAIRING_FAILED: return Result<bool, PlonkError>::Err(PlonkError::PairingCheckFailed);
    case BN254_ERR_BSB22_MISMATCH: return Result<bool, PlonkError>::Err(PlonkError::Bsb22CommitmentMismatch);
    case BN254_ERR_INVERSE: return Result<bool, PlonkError>::Err(PlonkError::InverseNotFound);
    case BN254_ERR_INPUT_LEN: return Result<bool, PlonkError>::Err(PlonkError::InvalidWitness);
    default: throw Panic(st);
  }
}
// proofs of different lengths side by side: records of the longest length (>= min_len), shorter ones zero-padded and remembered
inline Bytes pack(const std::vector<Bytes>& proofs, size_t min_len, size_t* stride, std::vector<bool>* is_short) {
  size_t s = min_len;
  for (const auto& p : proofs) s = p.size() > s ? p.size() : s;
  Bytes b(s * proofs.size(), 0);
  is_short->assign(proofs.size(), false);
  for (size_t i = 0; i < proofs.size(); i++) { std::copy(proofs[i].begin(), proofs[i].end(), b.begin() + s * i); (*is_short)[i] = proofs[i].size() < min_len; }
  *stride = s;
  return b;
}
}  // namespace detail

// A verifying key prepared once (decompression, e(alpha, beta), line tables, window tables: the work lib.rs:46 and groth16/verify.rs:70 repeat per call)
class PreparedGroth16Vk {
 public:
  explicit PreparedGroth16Vk(const Bytes& vk, unsigned mode = BN254_VK_REFERENCE) {
    int rc = bn254_groth16_vk_prepare(vk.data(), vk.size(), mode, &h_);
    if (rc == BN254_E_VK) throw Panic(BN254_ERR_MALFORMED);            // load_groth16_verifying_key_from_bytes(vk).unwrap()
    detail::check(rc);
  }
  ~PreparedGroth16Vk() { if (h_) bn254_groth16_vk_free(h_); }
  PreparedGroth16Vk(const PreparedGroth16Vk&) = delete;
  PreparedGroth16Vk& operator=(const PreparedGroth16Vk&) = delete;
  PreparedGroth16Vk(PreparedGroth16Vk&& o) noexcept : h_(o.h_) { o.h_ = nullptr; }
  size_t num_public() const { return bn254_groth16_vk_num_public(h_); }
  const bn254_g16_pvk* handle() const { return h_; }
  // n records of `stride` bytes, n_public x 32 bytes of inputs per proof; one status byte per proof (bn254_verify.h)
  Bytes verify_batch(const uint8_t* proofs, size_t stride, const uint8_t* public_inputs, size_t n_public, size_t n, int device = 0, unsigned flags = 0) const {
    Bytes st(n ? n : 1);
    detail::check(bn254_groth16_verify_batch(h_, proofs, stride, public_inputs, n_public, n, st.data(), device, flags));
    st.resize(n);
    return st;
  }
  Bytes verify_batch_multi(const uint8_t* proofs, size_t stride, const uint8_t* public_inputs, size_t n_public, size_t n, uint64_t device_mask, unsigned flags = 0) const {
    Bytes st(n ? n : 1);
    detail::check(bn254_groth16_verify_batch_multi(h_, proofs, stride, public_inputs, n_public, n, st.data(), device_mask, flags));
    st.resize(n);
    return st;
  }
  // enqueues on hip_stream and returns (BN254_FLAG_RLC: after one stream synchronisation)
  void verify_batch_device(const void* d_proofs, size_t stride, const void* d_public_inputs, size_t n_public, size_t n, void* d_status, int device = 0, void* hip_stream = nullptr,
                           unsigned flags = 0) const {
    detail::check(bn254_groth16_verify_batch_device(h_, d_proofs, stride, d_public_inputs, n_public, n, d_status, device, hip_stream, flags));
  }
  void reserve(size_t n, int device = 0) const { detail::check(bn254_groth16_reserve(h_, n, device)); }

 private:
  bn254_g16_pvk* h_ = nullptr;
};

struct Groth16Verifier {
  // lib.rs:44-49
  static Result<bool, Groth16Error> verify(const Bytes& proof, const Bytes& vk, const std::vector<Fr>& public_inputs, unsigned mode = BN254_VK_REFERENCE) {
    uint8_t st = 0xEE;
    const Bytes in = detail::flatten(public_inputs);
    int rc = bn254_groth16_verify(proof.data(), proof.size(), vk.data(), vk.size(), in.data(), public_inputs.size(), mode, &st);
    if (rc == BN254_E_VK) throw Panic(BN254_ERR_MALFORMED);
    detail::check(rc);
    return detail::groth16_outcome(st);
  }
  // the batch entry: one status byte per proof (BN254_ACCEPT, BN254_REJECT, BN254_ERR_*); outcome() turns a byte into the reference's answer
  static Bytes verify_batch(const std::vector<Bytes>& proofs, const Bytes& vk, const std::vector<std::vector<Fr>>& public_inputs,
                            unsigned mode = BN254_VK_REFERENCE, int device = 0, unsigned flags = 0) {
    if (proofs.size() != public_inputs.size()) throw std::invalid_argument("verify_batch: one input vector per proof");
    PreparedGroth16Vk pvk(vk, mode);
    const size_t n = proofs.size(), n_public = n ? public_inputs[0].size() : 0;
    Bytes in(32 * n_public * n);
    for (size_t i = 0; i < n; i++) {
      if (public_inputs[i].size() != n_public) throw std::invalid_argument("verify_batch: the proofs of a batch share the number of public inputs");
      const Bytes f = detail::flatten(public_inputs[i]);
      std::copy(f.begin(), f.end(), in.begin() + 32 * n_public * i);
    }
    size_t stride; std::vector<bool> is_short;
    const Bytes pb = detail::pack(proofs, 256, &stride, &is_short);
    Bytes st = pvk.verify_batch(pb.data(), stride, in.data(), n_public, n, device, flags);
    for (size_t i = 0; i < n; i++) if (is_short[i]) st[i] = BN254_ERR_MALFORMED;      // a buffer the loader cannot slice: a panic in the reference
    return st;
  }
  static Result<bool, Groth16Error> outcome(uint8_t status) { return detail::groth16_outcome(status); }
};

class PreparedPlonkVk {
 public:
  explicit PreparedPlonkVk(const Bytes& vk) {
    int rc = bn254_plonk_vk_prepare(vk.data(), vk.size(), &h_);
    if (rc == BN254_E_VK) throw Panic(BN254_ERR_MALFORMED);            // load_plonk_verifying_key_from_bytes(vk).unwrap()
    detail::check(rc);
  }
  ~PreparedPlonkVk() { if (h_) bn254_plonk_vk_free(h_); }
  PreparedPlonkVk(const PreparedPlonkVk&) = delete;
  PreparedPlonkVk& operator=(const PreparedPlonkVk&) = delete;
  size_t num_public() const { return bn254_plonk_vk_num_public(h_); }
  // flags: BN254_FLAG_RLC batches the pairing checks of a pass across proofs (same status bytes; honoured from 8192 proofs per pass)
  Bytes verify_batch(const uint8_t* proofs, size_t stride, const uint8_t* public_inputs, size_t n_public, size_t n, int device = 0, unsigned flags = 0) const {
    Bytes st(n ? n : 1);
    detail::check(bn254_plonk_verify_batch_flags(h_, proofs, stride, public_inputs, n_public, n, st.data(), device, flags));
    st.resize(n);
    return st;
  }
  // the same over the GPUs selected by device_mask (contiguous shards, one host thread per device)
  Bytes verify_batch_multi(const uint8_t* proofs, size_t stride, const uint8_t* public_inputs, size_t n_public, size_t n, uint64_t device_mask, unsigned flags = 0) const {
    Bytes st(n ? n : 1);
    detail::check(bn254_plonk_verify_batch_multi(h_, proofs, stride, public_inputs, n_public, n, st.data(), device_mask, flags));
    st.resize(n);
    return st;
  }
  // proofs, inputs and status bytes in device memory; returns when the status bytes are in d_status (bn254_verify.h)
  void verify_batch_device(const void* d_proofs, size_t stride, const void* d_public_inputs, size_t n_public, size_t n, void* d_status, int device = 0, void* hip_stream = nullptr,
                           unsigned flags = 0) const {
    detail::check(bn254_plonk_verify_batch_device(h_, d_proofs, stride, d_public_inputs, n_public, n, d_status, device, hip_stream, flags));
  }
  void reserve(size_t n, size_t proof_stride = 0, int device = 0) const { detail::check(bn254_plonk_reserve(h_, n, proof_stride, device)); }

 private:
  bn254_plonk_pvk* h_ = nullptr;
};

struct PlonkVerifier {
  // lib.rs:69-74
  static Result<bool, PlonkError> verify(const Bytes& proof, const Bytes& vk, const std::vector<Fr>& public_inputs) {
    uint8_t st = 0xEE;
    const Bytes in = detail::flatten(public_inputs);
    int rc = bn254_plonk_verify(proof.data(), proof.size(), vk.data(), vk.size(), in.data(), public_inputs.size(), &st);
    if (rc == BN254_E_VK) throw Panic(BN254_ERR_MALFORMED);
    detail::check(rc);
    return detail::plonk_outcome(st);
  }
  static Bytes verify_batch(const std::vector<Bytes>& proofs, const Bytes& vk, const std::vector<std::vector<Fr>>& public_inputs, int device = 0) {
    if (proofs.size() != public_inputs.size()) throw std::invalid_argument("verify_batch: one input vector per proof");
    PreparedPlonkVk pvk(vk);
    const size_t n = proofs.size(), n_public = n ? public_inputs[0].size() : 0;
    Bytes in(32 * n_public * n);
    for (size_t i = 0; i < n; i++) {
      if (public_inputs[i].size() != n_public) throw std::invalid_argument("verify_batch: the proofs of a batch share the number of public inputs");
      const Bytes f = detail::flatten(public_inputs[i]);
      std::copy(f.begin(), f.end(), in.begin() + 32 * n_public * i);
    }
    size_t stride; std::vector<bool> is_short;
    const Bytes pb = detail::pack(proofs, 516, &stride, &is_short);
    Bytes st = pvk.verify_batch(pb.data(), stride, in.data(), n_public, n, device);
    // a proof shorter than its own layout (plonk/converter.rs:121-178: 8 points, count + claimed values, second opening, count + commitments) is a slice-index panic in
    // the reference; padded to the stride it would parse as something else
    for (size_t i = 0; i < n; i++) {
      const Bytes& p = proofs[i];
      auto be32 = [&p](size_t o) { return (size_t)p[o] << 24 | (size_t)p[o + 1] << 16 | (size_t)p[o + 2] << 8 | (size_t)p[o + 3]; };
      bool bad = p.size() < 516;
      if (!bad) { const size_t off = 516 + 32 * be32(512); bad = p.size() < off + 100 || p.size() < off + 100 + 64 * be32(off + 96); }
      if (bad) st[i] = BN254_ERR_MALFORMED;
    }
    return st;
  }
  static Result<bool, PlonkError> outcome(uint8_t status) { return detail::plonk_outcome(status); }
};

}  // namespace snark_bn254_verifier
