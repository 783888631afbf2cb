// tests/hostsan/hostsan_main.cpp -- TEST HARNESS (see hip/hip_runtime.h beside it): the host half of the library, compiled with g++ -fsanitize=address,undefined
// against a host-memory stand-in for the HIP runtime and stand-ins for the kernel launchers, driven through the C ABI:
//   * malformed-bytes fuzz of bn254_groth16_vk_prepare / bn254_plonk_vk_prepare / bn254_sp1_fixture_parse / the point codecs (attacker-shaped lengths and counts:
//     groth16/converter.rs:28-65, plonk/converter.rs:18-119), seeded from valid keys;
//   * the batch entry points on a fake device: reservation and growth of contexts, the pinned ring and thread-pool copies of bn254_groth16_verify_batch, the RLC
//     pass with a fallback, wide keys, PlonK's context pool with calls in flight, allocation failures on every allocation of a call.
// The stand-in "kernels" only mark proofs: what is under test is everything AROUND the launches.  Prints "hostsan ok".
#include "hip/hip_runtime.h"
std::atomic<int> g_fake_device_count{1}, g_fake_fail_device{-1};
std::atomic<size_t> g_fake_live_allocs{0}, g_fake_fail_alloc_after{0}, g_fake_alloc_counter{0};
thread_local int t_fake_current_device = 0;
#include "../../snark-bn254-verifier_amd/csrc/bn254_capi.hip"
#include <cstdio>
#include <random>

// ---- stand-ins for the launchers of bn254_kernels.hip / bn254_k_plonk.hip / bn254_k_msm.hip / bn254_coop12.hip -----------------------------------------------
static std::atomic<long> g_launches{0};
const char* const bn254_kernel_kind_names[KID_COUNT] = {};
// a proof whose first byte is 0xEE is "invalid": REJECT on the exact path, and its RLC group stays pending
hipError_t bn254_launch_g16(const G16LaunchArgs& a, hipStream_t, hipEvent_t* ev, G16Prof* prof) {
  g_launches++;
  for (size_t i = 0; i < a.n; i++) {
    (void)a.proofs[i * a.stride + 255];                                   // the last byte the loader reads: in bounds of the caller's buffer
    for (int k = 0; k < a.n_public; k++) (void)a.inputs[(i * (size_t)a.n_public + (size_t)k) * 32 + 31];
    a.status[i] = a.proofs[i * a.stride] == 0xEE ? BN254_ST_REJECT : BN254_ST_ACCEPT;
  }
  memset(a.ws, 0x5a, a.n * (size_t)G16_WS_BYTES_PER_PROOF);               // the launch owns its part of the workspace: ASan checks the extent
  if (a.msm_part) {
    const size_t chunks = ((size_t)a.n_public + G16_WIDE_MSM_INPUTS_PER_LANE - 1) / G16_WIDE_MSM_INPUTS_PER_LANE;
    memset(a.msm_part, 0x5b, chunks * 27 * a.n * sizeof(int32_t));
    if (a.msm_digits) memset(a.msm_digits, 0x5c, (size_t)G16_COMB_COLS * (size_t)a.n_public * a.n * sizeof(uint16_t));
  }
  if (ev) for (int i = 0; i < 5; i++) hipEventRecord(ev[i], nullptr);
  if (prof && prof->used < prof->cap) { prof->kid[prof->used] = KID_MILLER_RUN; hipEventRecord(prof->ev[2 * prof->used], nullptr); hipEventRecord(prof->ev[2 * prof->used + 1], nullptr); prof->used++; }
  return hipSuccess;
}
hipError_t bn254_launch_g16_rlc(const G16LaunchArgs& a, const RlcLaunchArgs& r, hipStream_t) {
  g_launches++;
  memset(r.grp_status, 0, ((size_t)r.plan.groups + 255) / 256 * 256);     // the region the real launch clears
  for (size_t i = 0; i < a.n; i++) a.status[i] = BN254_ST_ACCEPT;
  for (size_t i = 0; i < a.n; i++)
    if (a.proofs[i * a.stride] == 0xEE) {                                  // every proof of its group stays pending
      const uint32_t g = rlc_group_of((uint32_t)i, r.plan);
      for (size_t j = 0; j < a.n; j++) if (rlc_group_of((uint32_t)j, r.plan) == g) a.status[j] = BN254_ST_PENDING;
    }
  return hipSuccess;
}
hipError_t bn254_launch_gather_rows(uint8_t* dst, const uint8_t* src, size_t src_stride, uint32_t row_bytes, const uint32_t* idx, uint32_t m, hipStream_t) {
  for (uint32_t k = 0; k < m; k++) memcpy(dst + (size_t)k * row_bytes, src + (size_t)idx[k] * src_stride, row_bytes);
  return hipSuccess;
}
hipError_t bn254_launch_scatter_status(uint8_t* status, const uint8_t* fb, const uint32_t* idx, uint32_t m, hipStream_t) { for (uint32_t k = 0; k < m; k++) status[idx[k]] = fb[k]; return hipSuccess; }
size_t bn254_plonk_work_bytes() { return sizeof(PlonkWork); }
size_t bn254_plonk_key_bytes() { return sizeof(PlonkKey); }
hipError_t bn254_plonk_dev_init(int) { return hipSuccess; }
hipError_t bn254_plonk_self_test(const void*, const void*, std::string* why) { why->clear(); return hipSuccess; }
hipError_t bn254_launch_plonk_stage1(const void*, const uint8_t* d_proofs, size_t stride, const uint8_t* d_inputs, size_t n_public, size_t n, const uint32_t*, void* d_work, void* d_terms, uint8_t* d_flags,
                                     int T1, hipStream_t) {
  g_launches++;
  for (size_t i = 0; i < n; i++) { (void)d_proofs[i * stride + stride - 1]; if (n_public) (void)d_inputs[(i * n_public + n_public - 1) * 32 + 31]; }
  memset(d_work, 0, n * sizeof(PlonkWork)); memset(d_terms, 0, n * (size_t)T1 * sizeof(MsmTerm)); memset(d_flags, 0, n * (size_t)T1);
  return hipSuccess;
}
hipError_t bn254_launch_plonk_stage2(const void*, const uint8_t*, size_t, size_t n, void*, const uint32_t* words, const uint8_t* inf, void* d_terms, uint8_t* d_flags, uint8_t* d_status, int TT, int,
                                     const uint32_t* weight_key, hipStream_t) {
  if (weight_key) (void)weight_key[10];
  g_launches++;
  (void)words[n * 16 - 1]; (void)inf[n - 1];
  memset(d_terms, 0, n * (size_t)TT * sizeof(MsmTerm)); memset(d_flags, 0, n * (size_t)TT); memset(d_status, BN254_ST_PENDING, n);
  return hipSuccess;
}
// BN254_FLAG_RLC on the PlonK path: the group stage reads / writes exactly what the real kernels do (ASan checks the extents); a proof whose first status bit pattern
// is "pending" and whose index is a multiple of 1000 makes its group fail, so that the exact fallback runs
hipError_t bn254_launch_plonk_group_sums(int32_t* ws, const uint8_t* status, size_t n, int32_t* grp_ws, uint8_t* grp_status, int, int, int, int, hipStream_t) {
  g_launches++;
  const size_t groups = (n + 63) / 64;
  (void)ws[0]; (void)status[n - 1];
  memset(grp_ws, 0x33, groups * (size_t)G16_WS_BYTES_PER_PROOF);
  for (size_t g = 0; g < groups; g++) grp_status[g] = BN254_ST_PENDING;
  return hipSuccess;
}
hipError_t bn254_launch_plonk_group_scatter(uint8_t* status, size_t n, const uint8_t* grp_status, uint32_t* n_failed, hipStream_t) {
  g_launches++;
  for (size_t i = 0; i < n; i++) {
    const bool group_failed = ((i / 64) % 16) == 3;
    (void)grp_status[i / 64];
    if ((status[i] & BN254_ST_PENDING) && !group_failed) status[i] = BN254_ST_ACCEPT;
    if (i % 64 == 0 && group_failed) (*n_failed)++;
  }
  return hipSuccess;
}
hipError_t bn254_launch_plonk_dbg_zeta(const void*, size_t n, uint8_t* z, uint8_t* s, hipStream_t) { memset(z, 0, 32 * n); memset(s, 1, n); return hipSuccess; }
size_t bn254_g1_msm_scratch_lanes(const MsmPlan& plan, size_t n) { return (size_t)plan.n_var_rows * ((n + 63) & ~(size_t)63); }
hipError_t bn254_launch_g1_msm_rows(const MsmPlan& plan, const int32_t* terms, const uint8_t* flags, size_t n, int n_terms, int32_t* part, int32_t* glv_tab, const int32_t* tabs, hipStream_t) {
  g_launches++;
  (void)terms[n * (size_t)n_terms * MSM_TERM_DWORDS - 1]; (void)flags[n * (size_t)n_terms - 1]; (void)tabs[0];
  memset(part, 0x11, (size_t)plan.n_rows * 27 * n * sizeof(int32_t));
  memset(glv_tab, 0x12, bn254_g1_msm_scratch_lanes(plan, n) * (size_t)G1_GLV_TAB_BYTES_PER_LANE);
  return hipSuccess;
}
hipError_t bn254_launch_g1_sum_rows(const MsmPlan&, const int32_t*, size_t n, uint32_t* out_words, uint8_t* out_inf, int32_t* ws, uint8_t* status, int, int, int, int, hipStream_t) {
  if (out_words) { memset(out_words, 0, n * 16 * 4); memset(out_inf, 0, n); }
  else { memset(ws, 0x13, n * (size_t)G16_WS_BYTES_PER_PROOF); (void)status[n - 1]; }
  return hipSuccess;
}
hipError_t bn254_launch_pairing2_fixed(int32_t*, uint8_t* status, size_t n, const int32_t*, const int32_t*, const int32_t*, int, hipStream_t, hipStream_t, hipEvent_t, hipEvent_t) {
  g_launches++;
  for (size_t i = 0; i < n; i++) if (status[i] & BN254_ST_PENDING) status[i] = BN254_ST_ACCEPT;
  return hipSuccess;
}
size_t bn254_tab_build_teeth(int form) { return form == 0 ? 13 : form == 1 ? 256 : (size_t)MSM_FW_WINDOWS * MSM_FW_BITS; }
size_t bn254_tab_build_entries(int form) { return form == 2 ? ((size_t)MSM_FW_WINDOWS << MSM_FW_BITS) : 8192; }
size_t bn254_tab_build_out_entries(int form) { return form == 0 ? 8192 : form == 1 ? 32 * 255 : (size_t)MSM_FW_WINDOWS * MSM_FW_ENTRIES; }
hipError_t bn254_launch_tab_build(int, const int32_t*, uint32_t, int32_t*, int32_t*, int32_t*, int32_t*, hipStream_t) { g_launches++; return hipSuccess; }
double bn254_measure_valu_peak(int) { return 1.0; }
double bn254_measure_valu_sustained(double) { return 1.0; }
hipError_t bn254_launch_dbg_fp_mul(const uint8_t*, const uint8_t*, uint8_t*, size_t, hipStream_t) { return hipSuccess; }
hipError_t bn254_launch_dbg_fp12_op(int, const uint8_t*, const uint8_t*, uint8_t*, size_t, int32_t*, uint8_t*, hipStream_t) { return hipSuccess; }
hipError_t bn254_launch_dbg_pairing(const uint8_t*, const uint8_t*, uint8_t*, size_t, int32_t*, uint8_t*, hipStream_t) { return hipSuccess; }
hipError_t bn254_launch_dbg_g2_ate(const uint8_t*, const uint8_t*, uint8_t*, size_t, int32_t*, uint8_t*, hipStream_t) { return hipSuccess; }
hipError_t bn254_launch_dbg_g2_subgroup(const uint8_t*, uint8_t*, size_t, hipStream_t) { return hipSuccess; }

#define CHECK(x) do { if (!(x)) { fprintf(stderr, "hostsan: check failed at line %d: %s (%s)\n", __LINE__, #x, bn254_last_error()); exit(1); } } while (0)

static std::vector<uint8_t> read_file(const std::string& p) {
  FILE* f = fopen(p.c_str(), "rb"); std::vector<uint8_t> v;
  if (!f) return v;
  uint8_t buf[65536]; size_t k;
  while ((k = fread(buf, 1, sizeof buf, f)) > 0) v.insert(v.end(), buf, buf + k);
  fclose(f);
  return v;
}

// one mutation of a byte string: flips, truncations, extensions, 32-bit big-endian counts replaced by hostile values
static std::vector<uint8_t> mutate(const std::vector<uint8_t>& base, std::mt19937_64& g, const std::vector<size_t>& count_offsets) {
  std::vector<uint8_t> v = base;
  switch (g() % 8) {
    case 0: if (!v.empty()) v[g() % v.size()] ^= (uint8_t)(1u << (g() % 8)); break;
    case 1: v.resize(g() % (v.size() + 1)); break;
    case 2: { size_t k = g() % 64; for (size_t i = 0; i < k; i++) v.push_back((uint8_t)g()); break; }
    case 3: if (!count_offsets.empty()) {
        size_t o = count_offsets[g() % count_offsets.size()];
        static const uint32_t hostile[] = {0xffffffffu, 0x80000000u, 0x7fffffffu, 0x10000u, 0x01000000u, 0u, 1u, 3u, 1025u};
        uint32_t c = hostile[g() % 9];
        if (o + 4 <= v.size()) { v[o] = (uint8_t)(c >> 24); v[o + 1] = (uint8_t)(c >> 16); v[o + 2] = (uint8_t)(c >> 8); v[o + 3] = (uint8_t)c; }
      } break;
    case 4: for (int k = 0; k < 8 && !v.empty(); k++) v[g() % v.size()] = (uint8_t)g(); break;
    case 5: if (v.size() > 40) { size_t a = g() % (v.size() - 32); memset(v.data() + a, 0xff, 32); } break;
    case 6: if (v.size() > 40) { size_t a = g() % (v.size() - 32); memset(v.data() + a, 0x00, 32); } break;
    default: if (!v.empty()) { v[0] = (uint8_t)((v[0] & 0x3f) | ((g() % 4) << 6)); } break;   // compression flags
  }
  return v;
}

// ---- the code that runs several host threads: bn254_groth16_verify_batch_multi / bn254_plonk_verify_batch_multi over EIGHT fake devices (the `w > 1` branch no
// one-GPU box ever takes: one host thread per device, per-device contexts of one shared key, error collection), concurrent callers on one key and one device, PlonK
// calls in flight on the context pool.  Run by the ASan/UBSan build after everything else and, alone, by the -fsanitize=thread build (argv[3] = "threads").
static void threaded_scenarios(const std::string& golden, bool big) {
  g_fake_device_count = 8;
  const size_t n_public = 2, n_max = big ? ((size_t)1 << 20) + 777 : 20000 + 777;
  std::vector<uint8_t> vk(bn254_synth_groth16_vk_len(n_public)), proofs(256 * n_max), inputs(32 * n_public * n_max), expected(64), status(n_max + 8), single(n_max + 8);
  CHECK(bn254_synth_groth16(0xB2540000, n_public, 64, 8, 1, 2, vk.data(), proofs.data(), inputs.data(), expected.data()) == 0);
  for (size_t i = 64; i < n_max; i++) { memcpy(&proofs[256 * i], &proofs[256 * (i % 64)], 256); memcpy(&inputs[64 * i], &inputs[64 * (i % 64)], 64); }
  for (size_t i = 0; i < n_max; i += 997) proofs[256 * i] = 0xEE;                  // the stand-in's "invalid" mark: REJECT
  bn254_g16_pvk* pvk = nullptr;
  CHECK(bn254_groth16_vk_prepare(vk.data(), vk.size(), 0, &pvk) == 0);
  auto want = [&](size_t i) { return proofs[256 * i] == 0xEE ? BN254_REJECT : BN254_ACCEPT; };
  // coverage and order: every mask and every ragged size gives the status vector of the single-device entry, nothing past n is written
  const uint64_t masks[] = {0x3, 0xFF, 0xA5, 0x80, 0x1};
  const size_t sizes[] = {n_max, 70000 + 777, 7, 0, 8, 4097};      // (sizes above n_max are skipped: the thread-sanitizer run is the small one)
  for (size_t m : sizes) {
    if (m > n_max) continue;
    memset(single.data(), 0xAB, single.size());
    CHECK(bn254_groth16_verify_batch(pvk, proofs.data(), 256, inputs.data(), n_public, m, single.data(), 0, 0) == 0);
    for (uint64_t mask : masks) {
      if (m == n_max && big && mask != 0xFF) continue;                              // the largest size once (5 GB of fake workspace per run)
      memset(status.data(), 0xAB, status.size());
      CHECK(bn254_groth16_verify_batch_multi(pvk, m ? proofs.data() : nullptr, 256, m ? inputs.data() : nullptr, n_public, m, m ? status.data() : nullptr, mask, 0) == 0);
      if (m) CHECK(memcmp(status.data(), single.data(), m + 8) == 0);
      for (size_t i = 0; i < m; i++) CHECK(status[i] == want(i));
      // the plan the entry followed: contiguous, balanced, in device order
      int devs[64], nsh = 0; size_t first[64], cnt[64];
      CHECK(bn254_shard_plan(m, mask, 8, devs, first, cnt, &nsh) == 0);
      size_t at = 0; for (int k = 0; k < nsh; k++) { CHECK(first[k] == at && cnt[k] + 1 >= cnt[0] && cnt[k] <= cnt[0]); at += cnt[k]; }
      CHECK(at == m);
    }
  }
  CHECK(bn254_groth16_verify_batch_multi(pvk, proofs.data(), 256, inputs.data(), n_public, 5000, status.data(), 0x100, 0) != 0);        // device 8 of 8: refused
  CHECK(bn254_groth16_verify_batch_multi(pvk, proofs.data(), 256, inputs.data(), n_public, 5000, status.data(), 0, 0) != 0);            // empty mask
  // RLC through the multi entry (each device forms its own groups, falls back on its own failed ones)
  bn254_set_rlc_params(64, 0, 1);
  memset(status.data(), 0xAB, status.size());
  CHECK(bn254_groth16_verify_batch_multi(pvk, proofs.data(), 256, inputs.data(), n_public, 20000, status.data(), 0x3C, BN254_FLAG_RLC) == 0);
  for (size_t i = 0; i < 20000; i++) CHECK(status[i] == want(i));
  // a device that fails (out of memory on device 5): the call reports it by ordinal, the other shards are complete, the key keeps working -- also on that device
  {
    bn254_g16_pvk* q = nullptr;
    CHECK(bn254_groth16_vk_prepare(vk.data(), vk.size(), 0, &q) == 0);
    memset(status.data(), 0xAB, status.size());
    const size_t nf = big ? 40000 : 16000, per_dev = nf / 8;                          // (within the buffers of the small, thread-sanitizer run)
    g_fake_fail_device = 5;
    const int rc = bn254_groth16_verify_batch_multi(q, proofs.data(), 256, inputs.data(), n_public, nf, status.data(), 0xFF, 0);
    g_fake_fail_device = -1;
    CHECK(rc == BN254_E_HIP && strstr(bn254_last_error(), "device 5") != nullptr);
    for (size_t i = 0; i < nf; i++) CHECK((i / per_dev == 5) ? status[i] == 0xAB : status[i] == want(i));
    CHECK(bn254_groth16_verify_batch_multi(q, proofs.data(), 256, inputs.data(), n_public, nf, status.data(), 0xFF, 0) == 0);
    for (size_t i = 0; i < nf; i++) CHECK(status[i] == want(i));
    bn254_groth16_vk_free(q);
  }
  // concurrent callers on ONE key: the same device (serialised by the library: one workspace per key and device), different devices, the device-pointer entry
  {
    std::vector<std::thread> th;
    std::atomic<int> bad{0};
    for (int t = 0; t < 6; t++)
      th.emplace_back([&, t] {
        std::vector<uint8_t> st(20000 + 8, 0xAB);
        const size_t m = 20000 - 1000 * (size_t)t;
        int rc = t % 3 == 2 ? bn254_groth16_verify_batch_device(pvk, proofs.data(), 256, inputs.data(), n_public, m, st.data(), t % 2, nullptr, 0)
                            : bn254_groth16_verify_batch(pvk, proofs.data(), 256, inputs.data(), n_public, m, st.data(), t % 2, t == 4 ? BN254_FLAG_RLC : 0);
        if (rc) bad++;
        for (size_t i = 0; i < m; i++) if (st[i] != want(i)) bad++;
        if (st[m] != 0xAB) bad++;
      });
    for (auto& x : th) x.join();
    CHECK(bad == 0);
    float ov; int single_stream; CHECK(bn254_groth16_stream_overlap(pvk, 0, &ov, &single_stream) == 0);
    float share; unsigned bypassed; CHECK(bn254_groth16_rlc_state(pvk, 0, &share, &bypassed) == 0);
  }
  bn254_groth16_vk_free(pvk);
  // single-proof entries from several threads: the key caches (exact-byte lookup, eviction) under contention
  {
    std::vector<std::thread> th;
    std::atomic<int> bad{0};
    for (int t = 0; t < 4; t++)
      th.emplace_back([&, t] {
        for (int k = 0; k < 6; k++) {
          std::vector<uint8_t> key = vk; uint8_t st = 0xAB;
          if ((k + t) % 3 == 0) key.push_back(0);                                  // another byte string: another cache entry (trailing bytes are ignored by the loader)
          if (bn254_groth16_verify(proofs.data() + 256, 256, key.data(), key.size(), inputs.data(), n_public, (unsigned)(k & 1), &st) != 0 || st != BN254_ACCEPT) bad++;
        }
      });
    for (auto& x : th) x.join();
    CHECK(bad == 0);
  }
  // PlonK: the context pool with calls in flight from several threads, the multi entry over the eight devices, a failing device, reserve / footprint
  std::vector<uint8_t> pvkb = read_file(golden + "/plonk_vk.bin");
  CHECK(pvkb.size() == 34368);
  {
    bn254_plonk_pvk* pk = nullptr;
    CHECK(bn254_plonk_vk_prepare(pvkb.data(), pvkb.size(), &pk) == 0);
    const size_t pn = big ? 150000 : 14000, pstride = 904;
    std::vector<uint8_t> pp(pstride * pn, 1), pi(64 * pn, 2), ps(pn + 8, 0xAB);
    size_t held = 0; int ctxs = 0;
    CHECK(bn254_plonk_reserve(pk, 12000, pstride, 0) == 0 && bn254_plonk_footprint(pk, 0, &held, &ctxs) == 0 && held > 0 && ctxs >= 1);
    std::vector<std::thread> th;
    std::atomic<int> bad{0};
    for (int t = 0; t < 5; t++)
      th.emplace_back([&, t] {
        std::vector<uint8_t> s2(12000 + 8, 0xAB);
        const size_t m = 12000 - 1500 * (size_t)t;
        const int rc = t == 3 ? bn254_plonk_verify_batch_device(pk, pp.data(), pstride, pi.data(), 2, m, s2.data(), t % 2, nullptr, 0)
                              : bn254_plonk_verify_batch_flags(pk, pp.data(), pstride, pi.data(), 2, m, s2.data(), t % 2, t == 1 ? BN254_FLAG_RLC : 0);
        if (rc) bad++;
        for (size_t i = 0; i < m; i++) if (s2[i] != BN254_ACCEPT) bad++;
        if (s2[m] != 0xAB) bad++;
      });
    for (auto& x : th) x.join();
    CHECK(bad == 0);
    for (uint64_t mask : {(uint64_t)0xFF, (uint64_t)0x3, (uint64_t)0xA5}) {
      for (size_t m : {pn, (size_t)7, (size_t)4097}) {
        memset(ps.data(), 0xAB, ps.size());
        CHECK(bn254_plonk_verify_batch_multi(pk, pp.data(), pstride, pi.data(), 2, m, ps.data(), mask, m == pn ? BN254_FLAG_RLC : 0) == 0);
        for (size_t i = 0; i < m; i++) CHECK(ps[i] == BN254_ACCEPT);
        CHECK(ps[m] == 0xAB);
      }
    }
    CHECK(bn254_plonk_verify_batch_multi(pk, pp.data(), pstride, pi.data(), 2, 0, ps.data(), 0xFF, 0) == 0);
    CHECK(bn254_plonk_verify_batch_multi(pk, pp.data(), pstride, pi.data(), 2, 100, ps.data(), 0x100, 0) != 0);
    CHECK(bn254_plonk_verify_batch_device(pk, pp.data(), pstride, pi.data(), 2, 100, ps.data(), 0, nullptr, 0x80u) != 0);
    bn254_plonk_vk_free(pk);
    bn254_plonk_pvk* q = nullptr;
    CHECK(bn254_plonk_vk_prepare(pvkb.data(), pvkb.size(), &q) == 0);
    g_fake_fail_device = 2;
    const int rc = bn254_plonk_verify_batch_multi(q, pp.data(), pstride, pi.data(), 2, 12000, ps.data(), 0x0F, 0);
    g_fake_fail_device = -1;
    CHECK(rc == BN254_E_HIP && strstr(bn254_last_error(), "device 2") != nullptr);
    CHECK(bn254_plonk_verify_batch_multi(q, pp.data(), pstride, pi.data(), 2, 12000, ps.data(), 0x0F, 0) == 0);
    bn254_plonk_vk_free(q);
  }
  g_fake_device_count = 1;
  printf("hostsan: threaded scenarios ok (8 fake devices%s)\n", big ? ", 2^20 + 777 proofs over all of them" : "");
}

int main(int argc, char** argv) {
  const std::string golden = argc > 1 ? argv[1] : "tests/golden";
  const long fuzz_iters = argc > 2 ? atol(argv[2]) : 300;
  if (argc > 3 && std::string(argv[3]) == "threads") {        // the -fsanitize=thread build runs these alone (the rest is single-threaded code the ASan build covers)
    threaded_scenarios(golden, false);
    printf("hostsan ok\n");
    return 0;
  }
  std::mt19937_64 g(0xB254);
  // ---------------------------------------------------------------- Groth16: a synthetic key and batch from the library's own generator
  const size_t n_public = 2, n = 70000;
  std::vector<uint8_t> vk(bn254_synth_groth16_vk_len(n_public)), proofs(256 * n), inputs(32 * n_public * n), expected(n), status(n + 8, 0xAB);
  CHECK(bn254_synth_groth16(0xB2540000, n_public, 64, 8, 1, 2, vk.data(), proofs.data(), inputs.data(), expected.data()) == 0);
  for (size_t i = 64; i < n; i++) { memcpy(&proofs[256 * i], &proofs[256 * (i % 64)], 256); memcpy(&inputs[64 * i], &inputs[64 * (i % 64)], 64); }
  for (size_t i = 0; i < n; i += 1000) proofs[256 * i] = 0xEE;                      // the stand-in's "invalid" mark
  bn254_g16_pvk* pvk = nullptr;
  CHECK(bn254_groth16_vk_prepare(vk.data(), vk.size(), 0, &pvk) == 0);
  // key fuzz: seeded from the valid key; every outcome but a crash / sanitizer report is fine
  {
    const std::vector<size_t> counts = {288, 292 + 32 * (n_public + 1)};
    long ok = 0;
    for (long it = 0; it < fuzz_iters; it++) {
      std::vector<uint8_t> m = mutate(vk, g, counts);
      if (it % 5 == 0) m = mutate(m, g, counts);
      bn254_g16_pvk* p = nullptr;
      int rc = bn254_groth16_vk_prepare(m.data(), m.size(), (unsigned)(it & 1), &p);
      if (rc == 0) { ok++; bn254_groth16_vk_free(p); } else CHECK(p == nullptr);
    }
    printf("hostsan: groth16 key fuzz: %ld of %ld mutated keys still parse\n", ok, fuzz_iters);
  }
  // batches on the fake device: sizes around every plan boundary, stride 256 and 324, host buffers (pinned ring + pool copies) and "device" pointers
  for (size_t m : {(size_t)1, (size_t)255, (size_t)4096, (size_t)30721, (size_t)65536, (size_t)65537, n}) {
    memset(status.data(), 0xAB, status.size());
    CHECK(bn254_groth16_verify_batch(pvk, proofs.data(), 256, inputs.data(), n_public, m, status.data(), 0, 0) == 0);
    for (size_t i = 0; i < m; i++) CHECK(status[i] == (proofs[256 * i] == 0xEE ? BN254_REJECT : BN254_ACCEPT));
    CHECK(status[m] == 0xAB);
  }
  CHECK(bn254_groth16_verify_batch_device(pvk, proofs.data(), 256, inputs.data(), n_public, n, status.data(), 0, nullptr, 0) == 0);
  CHECK(bn254_groth16_verify_batch_device(pvk, proofs.data(), 256, inputs.data(), n_public - 1, 1000, status.data(), 0, nullptr, BN254_FLAG_STRICT_SCALARS) == 0);
  // RLC with fallback groups (the stand-in leaves the groups of marked proofs pending): gather, exact pass, scatter
  bn254_set_rlc_params(64, 0, 1);
  memset(status.data(), 0xAB, status.size());
  CHECK(bn254_groth16_verify_batch(pvk, proofs.data(), 256, inputs.data(), n_public, n, status.data(), 0, BN254_FLAG_RLC) == 0);
  for (size_t i = 0; i < n; i++) CHECK(status[i] == (proofs[256 * i] == 0xEE ? BN254_REJECT : BN254_ACCEPT));
  CHECK(bn254_groth16_verify_batch_multi(pvk, proofs.data(), 256, inputs.data(), n_public, 5000, status.data(), 1, 0) == 0);
  // allocation failure on every allocation of a fresh context: an error code, no leak, no crash; the key still works afterwards
  for (size_t fail = 1; fail < 40; fail++) {
    bn254_g16_pvk* q = nullptr;
    CHECK(bn254_groth16_vk_prepare(vk.data(), vk.size(), 0, &q) == 0);
    g_fake_alloc_counter = 0; g_fake_fail_alloc_after = fail;
    int rc = bn254_groth16_verify_batch(q, proofs.data(), 256, inputs.data(), n_public, 70000, status.data(), 0, fail % 2 ? BN254_FLAG_RLC : 0);
    g_fake_fail_alloc_after = 0;
    if (rc == 0) { bn254_groth16_vk_free(q); break; }
    CHECK(bn254_groth16_verify_batch(q, proofs.data(), 256, inputs.data(), n_public, 300, status.data(), 0, 0) == 0);
    bn254_groth16_vk_free(q);
  }
  bn254_groth16_vk_free(pvk);
  // a key with many inputs: comb tables, partial-sum and digit buffers of the wide MSM
  {
    const size_t np = 40, m = 70000;
    std::vector<uint8_t> vk2(bn254_synth_groth16_vk_len(np)), pr2(256 * 8), in2(32 * np * m), ex2(8);
    CHECK(bn254_synth_groth16(0xB2540005, np, 8, 0, 1, 2, vk2.data(), pr2.data(), in2.data(), ex2.data()) == 0);
    std::vector<uint8_t> prm(256 * m); for (size_t i = 0; i < m; i++) memcpy(&prm[256 * i], &pr2[256 * (i % 8)], 256);
    bn254_g16_pvk* q = nullptr;
    CHECK(bn254_groth16_vk_prepare(vk2.data(), vk2.size(), 1, &q) == 0);
    CHECK(bn254_groth16_reserve(q, 1000, 0) == 0);
    CHECK(bn254_groth16_verify_batch(q, prm.data(), 256, in2.data(), np, m, status.data(), 0, 0) == 0);      // larger than the reservation: the entry point grows it
    CHECK(bn254_groth16_verify_batch(q, prm.data(), 256, in2.data(), np, 999, status.data(), 0, 0) == 0);
    bn254_groth16_vk_free(q);
  }
  CHECK(bn254_groth16_verify(proofs.data(), 256, vk.data(), vk.size(), inputs.data(), n_public, 0, status.data()) == 0);
  // ---------------------------------------------------------------- PlonK: the reference's key (tests/golden), mutated; batches through the context pool
  std::vector<uint8_t> pvkb = read_file(golden + "/plonk_vk.bin");
  CHECK(pvkb.size() == 34368);
  {
    bn254_plonk_pvk* pk = nullptr;
    CHECK(bn254_plonk_vk_prepare(pvkb.data(), pvkb.size(), &pk) == 0);
    const size_t pn = 12000, pstride = 904;
    std::vector<uint8_t> pp(pstride * pn, 1), pi(64 * pn, 2), ps(pn + 1, 0xAB);
    std::vector<std::thread> th;
    for (int t = 0; t < 3; t++) th.emplace_back([&, t] { std::vector<uint8_t> s2(pn); CHECK(bn254_plonk_verify_batch(pk, pp.data(), pstride, pi.data(), 2, pn - 1000 * t, s2.data(), 0) == 0); });
    for (auto& x : th) x.join();
    CHECK(bn254_plonk_verify_batch(pk, pp.data(), pstride, pi.data(), 2, pn, ps.data(), 0) == 0);
    CHECK(ps[pn] == 0xAB);
    // ... and with the pairing checks batched across proofs (group workspace, failure counter, the exact fallback on the groups the stand-in fails)
    CHECK(bn254_plonk_verify_batch_flags(pk, pp.data(), pstride, pi.data(), 2, pn, ps.data(), 0, BN254_FLAG_RLC) == 0);
    CHECK(ps[pn] == 0xAB);
    CHECK(bn254_plonk_verify_batch_flags(pk, pp.data(), pstride, pi.data(), 2, 100, ps.data(), 0, BN254_FLAG_RLC) == 0);          // below the threshold: the flag is ignored
    CHECK(bn254_plonk_verify_batch_flags(pk, pp.data(), pstride, pi.data(), 2, 100, ps.data(), 0, 0x80u) != 0);                   // an unknown flag is refused
    for (size_t fail = 1; fail < 60; fail++) {
      bn254_plonk_pvk* q = nullptr;
      CHECK(bn254_plonk_vk_prepare(pvkb.data(), pvkb.size(), &q) == 0);
      g_fake_alloc_counter = 0; g_fake_fail_alloc_after = fail;
      int rc = bn254_plonk_verify_batch(q, pp.data(), pstride, pi.data(), 2, 3000, ps.data(), 0);
      g_fake_fail_alloc_after = 0;
      bn254_plonk_vk_free(q);
      if (rc == 0) break;
    }
    bn254_plonk_vk_free(pk);
    const std::vector<size_t> counts = {0, 72, 368, 372 + 32 + 160 + 33788};   // size, nb_public, n_qcp, n_cci
    long ok = 0;
    for (long it = 0; it < fuzz_iters / 4 + 1; it++) {      // (a PlonK key prepares ten window tables: fewer iterations)
      std::vector<uint8_t> m = mutate(pvkb, g, counts);
      bn254_plonk_pvk* p = nullptr;
      int rc = bn254_plonk_vk_prepare(m.data(), m.size(), &p);
      if (rc == 0) { ok++; bn254_plonk_vk_free(p); } else CHECK(p == nullptr);
    }
    printf("hostsan: plonk key fuzz: %ld of %ld mutated keys still parse\n", ok, fuzz_iters / 4 + 1);
  }
  // ---------------------------------------------------------------- PlonK proof bytes: the parser and the layout function the device's chain lane uses instead of it
  {
    // k_plonk_stage1 runs parse_plonk_proof on one lane and, on another, the transcripts from plonk_proof_layout alone: wherever the parser accepts, the layout must
    // accept with the same offsets and counts; wherever the layout refuses, the parser must refuse too.  Mutated fixture proofs, hostile counts included.
    std::vector<uint8_t> fx = read_file(golden + "/sp1/fibonacci_plonk_proof.bin");
    std::vector<uint8_t> raw(2048); size_t raw_len = 0; int variant = 0; uint8_t pin[64], vh[32];
    CHECK(!fx.empty() && bn254_sp1_fixture_parse(fx.data(), fx.size(), &variant, raw.data(), raw.size(), &raw_len, pin, vh) == 0 && variant == 2);
    raw.resize(raw_len);
    const std::vector<size_t> counts = {512, 516 + 7 * 32 + 96};          // n_claimed, n_bsb
    long agree = 0, parsed = 0;
    for (long it = 0; it < fuzz_iters * 8; it++) {
      std::vector<uint8_t> m = it == 0 ? raw : mutate(raw, g, counts);
      if (it % 5 == 1 && m.size() > 516) { const uint32_t v = (uint32_t)(g() % 20); m[512] = 0; m[513] = 0; m[514] = 0; m[515] = (uint8_t)v; }     // plausible claimed-value counts
      bn254host::PlonkProof pr; bn254host::PlonkLayout lay;
      const int st = bn254host::parse_plonk_proof(pr, m.data(), m.size());
      const bool ok = bn254host::plonk_proof_layout(lay, m.data(), m.size());
      if (st == bn254host::PL_OK) {
        parsed++;
        CHECK(ok && lay.off_claimed == pr.off_claimed && lay.off_zs_h == pr.off_zs_h && lay.off_bsb == pr.off_bsb && lay.n_claimed == pr.n_claimed && lay.n_bsb == pr.n_bsb);
      }
      if (!ok) CHECK(st != bn254host::PL_OK);
      if (ok) { CHECK(lay.off_bsb + 64 * (size_t)lay.n_bsb <= m.size() && lay.off_claimed + 32 * (size_t)lay.n_claimed + 100 <= m.size() + 0); agree++; }
    }
    CHECK(parsed >= 1);
    printf("hostsan: plonk proof fuzz: %ld of %ld mutated proofs parse, layout usable for %ld\n", parsed, fuzz_iters * 8, agree);
  }
  // ---------------------------------------------------------------- SP1 fixture reader and the point codecs
  {
    std::vector<uint8_t> fx = read_file(golden + "/sp1/fibonacci_plonk_proof.bin");
    if (fx.empty()) fx = read_file(golden + "/sp1/fibonacci_groth16_proof.bin");
    long ok = 0;
    if (!fx.empty()) {
      const std::vector<size_t> counts = {4, 12};
      for (long it = 0; it < fuzz_iters * 4; it++) {
        std::vector<uint8_t> m = it == 0 ? fx : mutate(fx, g, counts);
        // little-endian u64 lengths: also plant hostile ones
        if (it % 7 == 3 && m.size() > 12) { for (int k = 0; k < 8; k++) m[4 + k] = (uint8_t)(g() % 3 ? 0xff : 0x00); }
        int variant; std::vector<uint8_t> raw(2048); size_t raw_len; uint8_t pi[64], vh[32];
        int rc = bn254_sp1_fixture_parse(m.data(), m.size(), &variant, raw.data(), (size_t)(g() % 2048), &raw_len, pi, vh);
        if (rc == 0) ok++;
      }
    }
    printf("hostsan: sp1 fixture fuzz: %ld parsed\n", ok);
    for (long it = 0; it < fuzz_iters * 4; it++) {
      uint8_t in[64], out[128], st; for (auto& b : in) b = (uint8_t)g();
      if (it % 3 == 0) memset(in, it % 2 ? 0xff : 0, 32);
      in[0] = (uint8_t)((in[0] & 0x3f) | ((it % 4) << 6));
      CHECK(bn254_g1_decompress(in, out, (int)(it & 1), &st) == 0);
      CHECK(bn254_g2_decompress(in, out, (unsigned)((it >> 1) & 1), (int)(it & 1), &st) == 0);
    }
  }
  threaded_scenarios(golden, true);
  printf("hostsan: %ld stand-in launches, %zu allocations still live (key caches of the single-proof entries)\n", g_launches.load(), g_fake_live_allocs.load());
  printf("hostsan ok\n");
  return 0;
}
