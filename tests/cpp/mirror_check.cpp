// tests/cpp/mirror_check.cpp -- TEST: the C++ host mirror (include/bn254_verify.hpp) of the crate's surface, driven as a C++ host would.
//   mirror_check cpu              mapping of status bytes to the reference's answers, loader panics, "no device" (runs without a GPU)
//   mirror_check gpu <dir>        Groth16 (synthetic gnark-format batch) and PlonK (<dir>/plonk_vk.bin, proof_i.bin, inputs_i.bin) on device 0;
//                                 prints one line per check, tests/test_cpp_mirror.py compares them with the oracle's verdicts
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iterator>
#include <string>
#include "bn254_verify.hpp"
using namespace snark_bn254_verifier;
static int fails = 0;
#define CHECK(c) do { if (!(c)) { printf("FAIL line %d: %s\n", __LINE__, #c); fails++; } } while (0)
template <class F> static int panics_with(F f) { try { f(); } catch (const Panic& p) { return p.status; } catch (const InfrastructureError& e) { return 1000 - e.code; } return -1; }
static Bytes slurp(const std::string& p) { std::ifstream f(p, std::ios::binary); return Bytes((std::istreambuf_iterator<char>(f)), {}); }
static std::vector<Fr> to_fr(const Bytes& b) { std::vector<Fr> v(b.size() / 32); for (size_t i = 0; i < v.size(); i++) std::copy(b.begin() + 32 * i, b.begin() + 32 * i + 32, v[i].begin()); return v; }

static void mapping() {
  CHECK(Groth16Verifier::outcome(BN254_ACCEPT).ok() && Groth16Verifier::outcome(BN254_ACCEPT).value());
  CHECK(Groth16Verifier::outcome(BN254_REJECT).ok() && !Groth16Verifier::outcome(BN254_REJECT).value());
  CHECK(Groth16Verifier::outcome(BN254_ERR_INPUT_LEN).is_err() && Groth16Verifier::outcome(BN254_ERR_INPUT_LEN).error() == Groth16Error::PrepareInputsFailed);
  for (int st : {BN254_ERR_NOT_MEMBER, BN254_ERR_NOT_ON_CURVE, BN254_ERR_NOT_IN_SUBGROUP, BN254_ERR_MALFORMED}) {
    CHECK(panics_with([&] { Groth16Verifier::outcome((uint8_t)st); }) == st);
    CHECK(panics_with([&] { PlonkVerifier::outcome((uint8_t)st); }) == st);
  }
  CHECK(PlonkVerifier::outcome(BN254_ACCEPT).unwrap());
  CHECK(PlonkVerifier::outcome(BN254_ERR_OPENING_MISMATCH).error() == PlonkError::OpeningPolyMismatch);
  CHECK(PlonkVerifier::outcome(BN254_ERR_PAIRING_FAILED).error() == PlonkError::PairingCheckFailed);
  CHECK(PlonkVerifier::outcome(BN254_ERR_BSB22_MISMATCH).error() == PlonkError::Bsb22CommitmentMismatch);
  CHECK(PlonkVerifier::outcome(BN254_ERR_INVERSE).error() == PlonkError::InverseNotFound);
  CHECK(PlonkVerifier::outcome(BN254_ERR_INPUT_LEN).error() == PlonkError::InvalidWitness);
  CHECK(panics_with([&] { PlonkVerifier::outcome(BN254_REJECT); }) == BN254_REJECT);     // PlonK never answers Ok(false)
  CHECK(std::string(to_string(Groth16Error::PrepareInputsFailed)) == "Prepare inputs failed");
  bool threw = false;
  try { Groth16Verifier::outcome(BN254_ERR_INPUT_LEN).unwrap(); } catch (const std::runtime_error&) { threw = true; }
  CHECK(threw);
}

int main(int argc, char** argv) {
  const std::string mode = argc > 1 ? argv[1] : "cpu";
  mapping();
  const size_t n = 64, n_public = 2;
  Bytes vk(bn254_synth_groth16_vk_len(n_public)), proofs(256 * n), inputs(32 * n_public * n), expected(n);
  CHECK(bn254_synth_groth16(0xB2540033, n_public, n, 4, 1, 2, vk.data(), proofs.data(), inputs.data(), expected.data()) == BN254_OK);
  // a verifying key that does not parse is the unwrap() of lib.rs:46 -- decided on the host, no device needed
  CHECK(panics_with([&] { Groth16Verifier::verify(Bytes(proofs.begin(), proofs.begin() + 256), Bytes(40, 0x5a), to_fr(Bytes(inputs.begin(), inputs.begin() + 64))); }) == BN254_ERR_MALFORMED);
  // ... but the proof was loaded first (lib.rs:70 before :71): its loader error is the one that surfaces
  CHECK(panics_with([&] { PlonkVerifier::verify(Bytes(904, 1), Bytes(33, 7), {}); }) == BN254_ERR_NOT_ON_CURVE);
  CHECK(panics_with([&] { PlonkVerifier::verify(Bytes(904, 0xff), Bytes(33, 7), {}); }) == BN254_ERR_NOT_MEMBER);
  CHECK(panics_with([&] { PlonkVerifier::verify(Bytes(400, 1), Bytes(33, 7), {}); }) == BN254_ERR_MALFORMED);
  { Bytes p(proofs.begin(), proofs.begin() + 256); p[63] ^= 1;      // A off the curve, key unparsable: the proof's error (lib.rs:45 before :46)
    CHECK(panics_with([&] { Groth16Verifier::verify(p, Bytes(40, 0x5a), to_fr(Bytes(inputs.begin(), inputs.begin() + 64))); }) == BN254_ERR_NOT_ON_CURVE); }
  CHECK(panics_with([&] { PreparedGroth16Vk bad(Bytes(519, 0)); }) == BN254_ERR_MALFORMED);
  {
    PreparedGroth16Vk pvk(vk);
    CHECK(pvk.num_public() == n_public);
  }
  if (mode == "cpu") {
    // a well-formed call on a machine without a GPU: an infrastructure error, never a verdict
    int r = panics_with([&] { Groth16Verifier::verify(Bytes(proofs.begin(), proofs.begin() + 256), vk, to_fr(Bytes(inputs.begin(), inputs.begin() + 64))); });
    printf("no_device %d\n", r);
    CHECK(r == 1000 - BN254_E_NO_DEVICE || r == -1 /* a GPU is present after all */);
    printf(fails ? "mirror_check: %d failures\n" : "mirror_check cpu ok\n", fails);
    return fails ? 1 : 0;
  }
  // ---- Groth16 on the device: the batch entry against the generator's statuses, the single entry on three kinds of proof
  std::vector<Bytes> pl(n); std::vector<std::vector<Fr>> il(n);
  for (size_t i = 0; i < n; i++) { pl[i] = Bytes(proofs.begin() + 256 * i, proofs.begin() + 256 * (i + 1)); il[i] = to_fr(Bytes(inputs.begin() + 64 * i, inputs.begin() + 64 * (i + 1))); }
  const Bytes st = Groth16Verifier::verify_batch(pl, vk, il);
  CHECK(st == expected);
  const Bytes st_rlc = Groth16Verifier::verify_batch(pl, vk, il, BN254_VK_REFERENCE, 0, BN254_FLAG_RLC);
  CHECK(st_rlc == expected);
  printf("g16_batch"); for (size_t i = 0; i < n; i++) printf(" %d", st[i]); printf("\n");
  int seen[16] = {0};
  for (size_t i = 0; i < n; i++) {
    if (seen[expected[i] & 15]++) continue;     // one proof of every status class through the single-proof entry
    const int e = expected[i];
    if (e == BN254_ACCEPT || e == BN254_REJECT) { auto r = Groth16Verifier::verify(pl[i], vk, il[i]); CHECK(r.ok() && r.value() == (e == BN254_ACCEPT)); printf("g16_single %zu ok %d\n", i, (int)r.value()); }
    else { int p = panics_with([&] { Groth16Verifier::verify(pl[i], vk, il[i]); }); CHECK(p == e); printf("g16_single %zu panic %d\n", i, p); }
  }
  { auto r = Groth16Verifier::verify(pl[0], vk, std::vector<Fr>(3)); CHECK(r.is_err() && r.error() == Groth16Error::PrepareInputsFailed); printf("g16_input_len err\n"); }
  { Bytes shortp(pl[0].begin(), pl[0].begin() + 100); int p = panics_with([&] { Groth16Verifier::verify(shortp, vk, il[0]); }); CHECK(p == BN254_ERR_MALFORMED); }
  // ---- PlonK: the reference's fixtures and mutations of them (files written by the test)
  if (argc > 2) {
    const std::string dir = argv[2];
    const Bytes pvk = slurp(dir + "/plonk_vk.bin");
    std::vector<Bytes> pp; std::vector<std::vector<Fr>> pi;
    for (int i = 0;; i++) {
      Bytes p = slurp(dir + "/proof_" + std::to_string(i) + ".bin");
      if (p.empty()) break;
      pp.push_back(p); pi.push_back(to_fr(slurp(dir + "/inputs_" + std::to_string(i) + ".bin")));
    }
    const Bytes ps = PlonkVerifier::verify_batch(pp, pvk, pi);
    printf("plonk_batch"); for (uint8_t s : ps) printf(" %d", s); printf("\n");
    for (size_t i = 0; i < pp.size(); i++) {
      int code;
      try { auto r = PlonkVerifier::verify(pp[i], pvk, pi[i]); code = r.ok() ? BN254_ACCEPT : -1; if (!r.ok()) { auto again = PlonkVerifier::outcome(ps[i]); CHECK(again.is_err() && again.error() == r.error()); code = ps[i]; } }
      catch (const Panic& p) { code = p.status; }
      CHECK(code == ps[i]);
      printf("plonk_single %zu %d\n", i, code);
    }
    // too few public inputs: Err(InvalidWitness) (plonk/verify.rs:57-59)
    if (!pp.empty() && pi[0].size() > 1) {
      auto r = PlonkVerifier::verify(pp[0], pvk, std::vector<Fr>(pi[0].begin(), pi[0].begin() + 1));
      CHECK(r.is_err() && r.error() == PlonkError::InvalidWitness);
      printf("plonk_input_len err\n");
    }
  }
  printf(fails ? "mirror_check: %d failures\n" : "mirror_check gpu ok\n", fails);
  return fails ? 1 : 0;
}
