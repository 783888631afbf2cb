// bn254_capi.hip -- implementation of the C ABI declared in include/bn254_verify.h.
// Host orchestration only: key preparation (bn254_host.hpp), device buffers, kernel launches (bn254_kernels.hip).
// There is deliberately no CPU implementation of verify here: if HIP is unusable the calls fail (BN254_E_NO_DEVICE).
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <dlfcn.h>
#include <algorithm>
#include <map>
#include <mutex>
#include <string>
#include <vector>
#include "../../include/bn254_verify.h"
#include "bn254_host.hpp"
#include "bn254_plonk.hpp"
#include "bn254_rlc.h"
#include "bn254_g16_plan.h"
#include <sys/random.h>
#include <atomic>
#include <thread>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <chrono>
#include <cstdio>
#include <stdexcept>

static_assert(BN254_REJECT == BN254_ST_REJECT && BN254_ACCEPT == BN254_ST_ACCEPT && BN254_ERR_NOT_MEMBER == BN254_ST_NOT_MEMBER &&
              BN254_ERR_NOT_ON_CURVE == BN254_ST_NOT_ON_CURVE && BN254_ERR_NOT_IN_SUBGROUP == BN254_ST_NOT_IN_SUBGROUP &&
              BN254_ERR_INPUT_LEN == BN254_ST_INPUT_LEN && BN254_ERR_MALFORMED == BN254_ST_MALFORMED, "status codes out of sync");

using namespace bn254host;

// bn254_k_plonk.hip: the PlonK host stages as device kernels (the same bn254_plonk.hpp source, one proof per lane)
size_t bn254_plonk_work_bytes();
size_t bn254_plonk_key_bytes();
hipError_t bn254_plonk_dev_init(int device);
hipError_t bn254_plonk_self_test(const void* key_host, const void* d_key, std::string* why);
hipError_t bn254_launch_plonk_stage1(const void* d_key, const uint8_t* d_proofs, size_t stride, const uint8_t* d_inputs, size_t n_public, size_t n, const uint32_t lam_key[11],
                                     void* d_work, void* d_terms, uint8_t* d_flags, int T1, hipStream_t s);
hipError_t bn254_launch_plonk_stage2(const void* d_key, const uint8_t* d_proofs, size_t stride, size_t n, void* d_work, const uint32_t* d_lin_words, const uint8_t* d_lin_inf,
                                     void* d_terms, uint8_t* d_flags, uint8_t* d_status, int TT, int T2, const uint32_t* weight_key, hipStream_t s);
hipError_t bn254_launch_plonk_group_sums(int32_t* ws, const uint8_t* status, size_t n, int32_t* grp_ws, uint8_t* grp_status, int e_p0, int inf0, int e_p1, int inf1, hipStream_t s);
hipError_t bn254_launch_plonk_group_scatter(uint8_t* status, size_t n, const uint8_t* grp_status, uint32_t* n_failed, hipStream_t s);

hipError_t bn254_launch_plonk_dbg_zeta(const void* d_work, size_t n, uint8_t* d_zeta, uint8_t* d_status, hipStream_t s);

// The HIP runtime multiplexes every stream of the process onto GPU_MAX_HW_QUEUES hardware queues -- four by default -- and streams that share a queue run one
// after the other: the two sub-batch streams of a large Groth16 batch then lose their overlap once a third party (RCCL) has streams too, and eight PlonK chains
// run at 1.20 instead of 1.51 M proofs/s (profiles/r03_batch_sweep_fine.txt).  The runtime reads the variable when it initialises, so it is a DEPLOYMENT setting
// (GPU_MAX_HW_QUEUES=8 in the environment of the process); the library does not touch the environment.  What it does instead: the first batch
// that runs two sub-batch streams brackets them with events, the next call reads the overlap (bn254_groth16_stream_overlap), and a device whose sub-batch streams
// were found to run one after the other gets one sub-batch per launch from then on (same work, fewer launches) and a line in bn254_last_diagnostic().
static thread_local std::string g_diag;
static thread_local std::string g_err;
static std::atomic<int> g_profiling{0};
static std::atomic<unsigned> g_prof_mask{0xffffffffu};
static std::atomic<unsigned> g_prof_epoch{0};   // bumped by the two profiling setters: an accumulating profile (mode 2) starts over at the next batch
static int set_err(int code, const std::string& msg) { g_err = msg; return code; }
// Knobs of the RLC batch mode.  Initial values come from the environment, read ONCE when the library is loaded (getenv racing a host's
// setenv is undefined behaviour); afterwards only bn254_set_rlc_params changes them.
static long env_long(const char* name, long dflt) { const char* e = getenv(name); return e ? atol(e) : dflt; }
#define RLC_MIN_BATCH 64            // below this the mode has no groups to speak of
#define RLC_PAYS_FROM 200000        // the mode is a longer pipeline (~18 ms whatever the size): measured 0.12 x at 4096, 0.45 x at 16384, 0.94 x at 2^17, 2.0 x at 2^20
static std::atomic<long> g_rlc_min_batch{[] { long v = env_long("BN254_RLC_MIN_BATCH", RLC_PAYS_FROM); return v < RLC_MIN_BATCH ? (long)RLC_MIN_BATCH : v; }()};
static std::atomic<int> g_rlc_adaptive{env_long("BN254_RLC_ADAPTIVE", 1) != 0 ? 1 : 0};
static std::atomic<long> g_rlc_share_min_lanes{env_long("BN254_RLC_SHARE_MIN_LANES", 65536)};
#define HIPCK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return set_err(BN254_E_HIP, std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)

// BN254_FLAG_RLC: per (key, device) buffers of the random-linear-combination batch mode (bn254_rlc.h)
struct RlcDev {
  bool ready = false;
  int32_t *btab = nullptr, *tab = nullptr, *one = nullptr;            // key-side tables (uploaded once)
  uint8_t* grp_status = nullptr; size_t grp_cap = 0;
  uint32_t* idx = nullptr; size_t idx_cap = 0;
  uint8_t *fb_proofs = nullptr, *fb_inputs = nullptr, *fb_status = nullptr; size_t fb_cap = 0, fb_in_cap = 0;
  uint8_t* h_status = nullptr; uint32_t* h_idx = nullptr; size_t h_cap = 0;   // pinned
  // adaptive use of the mode: share of the checked proofs the last RLC passes sent to the exact fallback (exponential average) and how many
  // calls have bypassed the mode since the last pass that measured it
  bool have_obs = false; float fb_share = 0.f; unsigned bypassed = 0, bypassed_total = 0;
};
static void rlc_dev_free(RlcDev& r) {
  void* ptrs[] = {r.btab, r.tab, r.one, r.grp_status, r.idx, r.fb_proofs, r.fb_inputs, r.fb_status};
  for (auto q : ptrs) if (q) (void)hipFree(q);
  if (r.h_status) (void)hipHostFree(r.h_status);
  if (r.h_idx) (void)hipHostFree(r.h_idx);
  const RlcDev keep = r;
  r = RlcDev();
  r.have_obs = keep.have_obs; r.fb_share = keep.fb_share; r.bypassed = keep.bypassed; r.bypassed_total = keep.bypassed_total;
}

// Per (key, device) state.  `mu` serialises everything that touches it: uploads, (re)allocation and the enqueue of a batch.  The
// workspace and the staging buffers are shared by all batches against this key on this device, so a batch first waits (on the GPU:
// hipStreamWaitEvent) for `busy_ev`, the completion event of the previous batch, whatever stream that one ran on.
struct DevState {
  std::mutex mu;
  bool ready = false;
  int32_t *k0 = nullptr, *gtab = nullptr, *dtab = nullptr, *target = nullptr, *msm = nullptr;
  int32_t* ws = nullptr; size_t ws_cap = 0;                         // proofs the workspace can hold
  int32_t* msm_part = nullptr; size_t msm_part_cap = 0, msm_chunks = 0;             // wide keys: partial sums of the public-input MSM (proofs it holds)
  uint8_t *st_proofs = nullptr, *st_inputs = nullptr, *st_status = nullptr;  // staging for the host-buffer entry point
  size_t st_proofs_cap = 0, st_inputs_cap = 0, st_status_cap = 0;
  hipStream_t host_stream = nullptr, copy_stream = nullptr;   // host-buffer entry: copy / compute overlap
  uint8_t* pin[3] = {nullptr, nullptr, nullptr}; size_t pin_cap = 0; hipEvent_t pin_ev[3] = {nullptr, nullptr, nullptr};   // ring of pinned pieces (HOST_RING)
  hipEvent_t busy_ev = nullptr; bool busy_valid = false;
  hipEvent_t ev[5]; bool ev_ready = false; bool ev_recorded = false;
  // concurrent sub-batches (see g16_enqueue_exact): part 0 runs on the caller's stream, parts 1..3 on these, created when first needed -- every
  // stream of a process shares the runtime's few hardware queues (four by default), and a copy stream that lands on the queue of a busy compute
  // stream waits behind its kernels (measured: 3 GB/s instead of 55), so no stream is created that is not used
  hipStream_t aux[3] = {nullptr, nullptr, nullptr}; int aux_count = 0; hipEvent_t fork_ev = nullptr, join_ev[4] = {nullptr, nullptr, nullptr, nullptr};
  // per-launch timing of the first sub-batch (bn254_groth16_kernel_profile)
  std::vector<hipEvent_t> prof_ev; std::vector<uint8_t> prof_kid; G16Prof prof{0, nullptr, nullptr, 0, 0}; size_t prof_n = 0; unsigned prof_epoch = 0;
  // the same for the SECOND sub-batch (its launches run on another stream beside the first's): bn254_groth16_kernel_profile_all
  std::vector<hipEvent_t> prof2_ev; std::vector<uint8_t> prof2_kid; G16Prof prof2{0, nullptr, nullptr, 0, 0}; bool prof2_used = false;
  RlcDev rlc;                                                       // BN254_FLAG_RLC buffers (bn254_rlc.hpp)
  // do the sub-batch streams overlap?  ov_ev: start / end of part 0 and of part 1 of the first two-stream batch; ov_state 0: not measured, 1: events recorded,
  // 2: measured (ov_ratio = sum of the two durations / their union: ~2 side by side, ~1 one after the other); single_stream: fall back to one sub-batch per launch
  hipEvent_t ov_ev[4] = {nullptr, nullptr, nullptr, nullptr}; int ov_state = 0; float ov_ratio = -1.f; bool single_stream = false;
  // the decision is not taken from one measurement (another tenant's kernels, a profiler that serialises dispatches): OV_AGREE consecutive measurements must say
  // "serialised" before the plan changes, a measurement that says "side by side" resets the count; once on one sub-batch per launch, every OV_REPROBE-th batch runs two
  // again and is measured, so that a transient cause does not pin the key to the slower plan for its lifetime.  `diag`: the explanation, per (key, device), handed out by
  // bn254_groth16_stream_overlap through bn254_last_diagnostic() of the calling thread
  int ov_serial_votes = 0; unsigned ov_batches = 0; bool ov_probe = false; std::string diag;
};
#define OV_AGREE 3
#define OV_REPROBE 256
struct bn254_g16_pvk {
  G16Prepared host;
  mutable G16PreparedRlc rlc_host;       // built on the first BN254_FLAG_RLC batch (under mu)
  mutable std::mutex mu;                 // protects the map below (lookup / insertion only) and rlc_host
  mutable std::map<int, DevState> dev;
};

static int check_device(int device) {
  int cnt = 0;
  hipError_t e = hipGetDeviceCount(&cnt);
  if (e != hipSuccess || cnt <= 0) return set_err(BN254_E_NO_DEVICE, "no HIP device available (this library has no CPU fallback)");
  if (device < 0 || device >= cnt) return set_err(BN254_E_BAD_ARG, "device ordinal out of range");
  HIPCK(hipSetDevice(device));
  return BN254_OK;
}
// *dst stays null unless the copy is complete: a caller that retries after a failure uploads exactly what is still missing (a sanitizer run of the
// allocation-failure paths found the retry overwriting -- leaking -- the tables an earlier, partly failed attempt had already uploaded)
template <typename T> static int upload(T** dst, const std::vector<T>& src) {
  if (*dst) return BN254_OK;
  size_t bytes = (src.size() ? src.size() : 1) * sizeof(T);
  T* p = nullptr;
  HIPCK(hipMalloc((void**)&p, bytes));
  if (!src.empty()) {
    hipError_t e = hipMemcpy(p, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(p); return set_err(BN254_E_HIP, std::string("hipMemcpy: ") + hipGetErrorString(e)); }
  }
  *dst = p;
  return BN254_OK;
}
// The fixed-base tables of a key, built on the current device from the key's points (bn254_k_comb.hip; form 0: comb tables, 1: byte windows; pts: 18 dwords per point):
// 80 bytes x 8192 (8160) entries per point stay, the construction scratch (27 dwords per entry, passes of 256 points: 226 MB at most) is freed again.  *dst stays null unless
// the table is complete (as upload() above).
static inline int g16_table_form(const G16Prepared& h) { return h.msm_comb ? 0 : h.key_inputs() > (size_t)G16_WIDE_MSM_MIN_INPUTS ? 1 : 2; }
static int build_tables_on_device(int form, const std::vector<int32_t>& pts, int32_t** dst) {
  if (*dst) return BN254_OK;
  const size_t np = pts.size() / (2 * BN_NL);
  const size_t per_point = bn254_tab_build_out_entries(form) * MSM_ENTRY_DWORDS;   // dwords of finished table per point
  const size_t teeth = bn254_tab_build_teeth(form), entries = bn254_tab_build_entries(form);
  const size_t slice_cap = ((size_t)256 << 13) / entries ? ((size_t)256 << 13) / entries : 1;   // points per pass: 2 M construction entries (226 MB of scratch) at most
  const size_t slice = np < slice_cap ? np : slice_cap;
  int32_t *kp = nullptr, *tab = nullptr, *tplane = nullptr, *taff = nullptr, *plane = nullptr;
  auto drop = [&]() { if (kp) (void)hipFree(kp); if (tplane) (void)hipFree(tplane); if (taff) (void)hipFree(taff); if (plane) (void)hipFree(plane); };
  hipError_t e;
  if ((e = hipMalloc((void**)&kp, pts.size() * sizeof(int32_t))) != hipSuccess || (e = hipMalloc((void**)&tab, np * per_point * sizeof(int32_t))) != hipSuccess ||
      (e = hipMalloc((void**)&tplane, slice * teeth * 27 * sizeof(int32_t))) != hipSuccess || (e = hipMalloc((void**)&taff, slice * teeth * 2 * BN_NL * sizeof(int32_t))) != hipSuccess ||
      (e = hipMalloc((void**)&plane, slice * entries * 27 * sizeof(int32_t))) != hipSuccess ||
      (e = hipMemcpy(kp, pts.data(), pts.size() * sizeof(int32_t), hipMemcpyHostToDevice)) != hipSuccess) {
    drop(); if (tab) (void)hipFree(tab);
    return set_err(BN254_E_HIP, std::string("fixed-base tables of the key: ") + hipGetErrorString(e));
  }
  for (size_t i0 = 0; i0 < np && e == hipSuccess; i0 += slice) {
    const size_t m = np - i0 < slice ? np - i0 : slice;            // the passes run one after the other on the null stream and share the scratch
    e = bn254_launch_tab_build(form, kp + i0 * 2 * BN_NL, (uint32_t)m, tab + i0 * per_point, tplane, taff, plane, nullptr);
  }
  if (e == hipSuccess) e = hipDeviceSynchronize();
  drop();
  if (e != hipSuccess) { (void)hipFree(tab); return set_err(BN254_E_HIP, std::string("fixed-base tables of the key: ") + hipGetErrorString(e)); }
  *dst = tab;
  return BN254_OK;
}
static DevState* dev_state(const bn254_g16_pvk* pvk, int device) {
  std::lock_guard<std::mutex> lk(pvk->mu);
  return &pvk->dev[device];   // std::map nodes never move
}
// caller holds d.mu
static int ensure_dev(const bn254_g16_pvk* pvk, DevState& d, int device, size_t n) {
  int rc = check_device(device);
  if (rc) return rc;
  if (!d.ready) {
    if ((rc = upload(&d.k0, pvk->host.k0)) || (rc = upload(&d.gtab, pvk->host.gtab)) || (rc = upload(&d.dtab, pvk->host.dtab)) || (rc = upload(&d.target, pvk->host.target)))
      return rc;
    // comb tables above 16 inputs; 13-bit windows (bn254_fw.h) up to 16; byte windows only for the diagnostic BN254_WIDE_COMB=0 (k_g16_msm_partial)
    if (!pvk->host.kpts.empty() && pvk->host.msm.empty()) { if ((rc = build_tables_on_device(g16_table_form(pvk->host), pvk->host.kpts, &d.msm))) return rc; }
    else if ((rc = upload(&d.msm, pvk->host.msm))) return rc;
    HIPCK(hipEventCreateWithFlags(&d.busy_ev, hipEventDisableTiming));
    d.ready = true;
  }
  // what a reservation of n proofs needs (bn254_g16_plan.h: the same function the plan probe and its property test read)
  const G16Alloc need = g16_alloc_for(n, pvk->host.key_inputs(), pvk->host.msm_comb);
  if (need.ws_proofs > d.ws_cap) {
    if (d.ws) HIPCK(hipFree(d.ws));   // hipFree waits for the device: no batch is still using the old workspace
    d.ws = nullptr; d.ws_cap = 0;
    HIPCK(hipMalloc((void**)&d.ws, need.ws_proofs * (size_t)G16_WS_BYTES_PER_PROOF));
    d.ws_cap = need.ws_proofs;
  }
  // keys with many public inputs: partial sums (and comb digits) of the public-input MSM, for the proofs of one launch.  Sized HERE (reserve /
  // the entry points call ensure_dev before they enqueue), so that the enqueue path itself never allocates or frees
  if (need.msm_part_proofs > d.msm_part_cap) {
    if (d.msm_part) HIPCK(hipFree(d.msm_part));
    d.msm_part = nullptr; d.msm_part_cap = 0;
    HIPCK(hipMalloc((void**)&d.msm_part, need.msm_part_bytes + need.msm_digit_bytes));
    d.msm_part_cap = need.msm_part_proofs; d.msm_chunks = need.msm_chunks;
  }
  if (g_profiling.load() && !d.ev_ready) {
    for (int i = 0; i < 5; i++) HIPCK(hipEventCreate(&d.ev[i]));
    const int cap = 1024;  // launches per sub-batch: ~720
    d.prof_ev.resize(2 * cap); d.prof_kid.resize(cap);
    for (auto& e : d.prof_ev) HIPCK(hipEventCreate(&e));
    d.prof.ev = d.prof_ev.data(); d.prof.kid = d.prof_kid.data(); d.prof.cap = cap;
    d.prof2_ev.resize(2 * cap); d.prof2_kid.resize(cap);
    for (auto& e : d.prof2_ev) HIPCK(hipEventCreate(&e));
    d.prof2.ev = d.prof2_ev.data(); d.prof2.kid = d.prof2_kid.data(); d.prof2.cap = cap;
    d.ev_ready = true;
  }
  return BN254_OK;
}
static int ensure_aux(DevState& d, int count) {
  if (count > 3) count = 3;
  if (!d.fork_ev) {
    HIPCK(hipEventCreateWithFlags(&d.fork_ev, hipEventDisableTiming));
    for (int i = 0; i < 4; i++) HIPCK(hipEventCreateWithFlags(&d.join_ev[i], hipEventDisableTiming));
  }
  while (d.aux_count < count) { HIPCK(hipStreamCreateWithFlags(&d.aux[d.aux_count], hipStreamNonBlocking)); d.aux_count++; }
  return BN254_OK;
}
// part pi of a batch split over concurrent streams: slot pi % 4, slot 0 = the caller's stream, slots 1..3 = the auxiliary streams
static inline hipStream_t part_stream(DevState& d, hipStream_t user, int pi) { const int k = pi % 4; return k == 0 ? user : d.aux[k - 1]; }
static void dev_free(DevState& d) {
  int32_t* ptrs[] = {d.k0, d.gtab, d.dtab, d.target, d.msm, d.ws, d.msm_part};
  for (auto q : ptrs) if (q) (void)hipFree(q);
  uint8_t* bp[] = {d.st_proofs, d.st_inputs, d.st_status};
  for (auto q : bp) if (q) (void)hipFree(q);
  if (d.ev_ready) { for (int i = 0; i < 5; i++) (void)hipEventDestroy(d.ev[i]); for (auto& e : d.prof_ev) (void)hipEventDestroy(e); for (auto& e : d.prof2_ev) (void)hipEventDestroy(e); }
  for (int i = 0; i < d.aux_count; i++) (void)hipStreamDestroy(d.aux[i]);
  if (d.fork_ev) { (void)hipEventDestroy(d.fork_ev); for (int i = 0; i < 4; i++) (void)hipEventDestroy(d.join_ev[i]); }
  if (d.busy_ev) (void)hipEventDestroy(d.busy_ev);
  for (auto& e : d.ov_ev) if (e) (void)hipEventDestroy(e);
  if (d.host_stream) (void)hipStreamDestroy(d.host_stream);
  if (d.copy_stream) (void)hipStreamDestroy(d.copy_stream);
  for (int i = 0; i < 3; i++) { if (d.pin[i]) (void)hipHostFree(d.pin[i]); if (d.pin_ev[i]) (void)hipEventDestroy(d.pin_ev[i]); }
  rlc_dev_free(d.rlc);
}
static int grow(uint8_t** p, size_t* cap, size_t need) {
  if (need <= *cap) return BN254_OK;
  if (*p) HIPCK(hipFree(*p));
  *p = nullptr; *cap = 0;
  HIPCK(hipMalloc((void**)p, need));
  *cap = need;
  return BN254_OK;
}

// ---------------------------------------------------------------- PlonK (BASELINE configs[3])
// One PlonkCtx = one sub-batch in flight: its own stream, device buffers and pinned host staging.  A batch is cut into sub-batches that worker
// threads drive concurrently, so the host stages of one sub-batch (transcripts, Fr arithmetic) overlap the GPU stages of the others; every
// wait is stream-scoped.
#define PLONK_WORKERS 8
// proofs per pass at most.  Until round 4 this was 65 536 -- one wavefront per SIMD for every one-lane-per-proof kernel of a pass, which left the pairing stage of the
// largest passes at 0.39 of the multiply-add peak; a pass of 2^18 proofs gives the same kernels four (the context's buffers for it: 5.4 GB at the SP1 key shape)
#define PLONK_MAX_LAUNCH 262144
#define PLONK_BIG_PIECE_DEFAULT 131072   // proofs per pass of a batch above 65 536 proofs (profiles/r05_plonk_piece_sweep.txt)
struct PlonkCtx {
  size_t cap = 0;                      // proofs the buffers below hold
  hipStream_t stream = nullptr, aux = nullptr; hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  hipEvent_t tk[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // timing: before stage 1 | after it | MSM rows | sum | stage 2 | MSM rows | sums | pairing check
  float last_ms[BN254_PLONK_NUM_TIMINGS] = {0}; size_t last_lanes[2] = {0, 0}; bool last_valid = false;
  std::vector<PlonkWork> work;        // host scratch per proof (kept across calls)
  int32_t *ws = nullptr, *part = nullptr, *glv_tab = nullptr;   // part: the rows of an MSM launch (bn254_msm.h); glv_tab: the window tables of its variable rows
  size_t part_points = 0;              // projective points (rows x items) `part` holds (plonk_part_points of the capacity)
  size_t glv_lanes = 0;                // lanes glv_tab holds (plonk_scratch_lanes of the capacity); a launch checks its need against it before it is enqueued
  MsmTerm* terms = nullptr; uint8_t* flags = nullptr; uint32_t* words = nullptr; uint8_t *inf = nullptr, *status = nullptr;
  // pinned host staging
  MsmTerm* h_terms = nullptr; uint8_t *h_flags = nullptr, *h_status = nullptr, *h_inf = nullptr; uint32_t* h_words = nullptr;
  // device-side stages (bn254_k_plonk.hip): the batch's proofs and inputs in device memory (through a pinned copy), per-proof state between the stages
  uint8_t *d_in = nullptr, *h_in = nullptr; size_t in_cap = 0; void* d_work = nullptr;
  // BN254_FLAG_RLC: the pairing checks of a pass batched over groups of 64 proofs -- the groups' points and status bytes in a workspace of their own, failed groups counted
  int32_t* grp_ws = nullptr; uint8_t* grp_status = nullptr; uint32_t* d_fail = nullptr; uint32_t* h_fail = nullptr;
};
struct PlonkDev {
  bool ready = false;
  int32_t *tab0 = nullptr, *tab1 = nullptr, *one = nullptr;
  int32_t* fixed_tabs = nullptr;       // window tables of the key's G1 points (plonk_num_tables x MSM_FW_WINDOWS x MSM_FW_ENTRIES entries, bn254_fw.h)
  void* d_key = nullptr;               // the parsed key (PlonkKey) for the device-side stages
  PlonkCtx ctx[PLONK_WORKERS];
  // The contexts are handed out to calls: a call takes one per sub-batch (all at once, so two calls cannot wait for each other) and returns them when it
  // is done.  Calls on ONE prepared key from several host threads therefore run side by side, up to PLONK_WORKERS sub-batches in flight; at 4096 proofs a
  // batch is a chain of latency-bound launches that leaves most of the GPU idle, and two batches in flight verify 1.35 x as many proofs per second.
  std::mutex pool_mu; std::condition_variable pool_cv; bool busy[PLONK_WORKERS] = {};
  float last_ms[BN254_PLONK_NUM_TIMINGS] = {0}; size_t last_lanes[2] = {0, 0}; bool last_valid = false;   // first sub-batch of the call that finished last
};
struct PlonkLease {   // the contexts of one call
  PlonkDev* d; int idx[PLONK_WORKERS]; int n = 0;
  PlonkLease(PlonkDev* d_, int want) : d(d_) {
    std::unique_lock<std::mutex> lk(d->pool_mu);
    d->pool_cv.wait(lk, [&] { int f = 0; for (bool b : d->busy) f += b ? 0 : 1; return f >= want; });
    for (int i = 0; i < PLONK_WORKERS && n < want; i++) if (!d->busy[i]) { d->busy[i] = true; idx[n++] = i; }
  }
  PlonkCtx& ctx(int w) const { return d->ctx[idx[w]]; }
  ~PlonkLease() {
    {
      std::lock_guard<std::mutex> lk(d->pool_mu);
      const PlonkCtx& c = d->ctx[idx[0]];
      if (c.last_valid) { for (int i = 0; i < BN254_PLONK_NUM_TIMINGS; i++) d->last_ms[i] = c.last_ms[i]; d->last_lanes[0] = c.last_lanes[0]; d->last_lanes[1] = c.last_lanes[1]; d->last_valid = true; }
      for (int i = 0; i < n; i++) d->busy[idx[i]] = false;
    }
    d->pool_cv.notify_all();
  }
  PlonkLease(const PlonkLease&) = delete; PlonkLease& operator=(const PlonkLease&) = delete;
};
struct bn254_plonk_pvk {
  PlonkKey key;
  std::vector<int32_t> tab0, tab1, one;
  std::vector<int32_t> fixed_pts;      // every key point that enters an MSM (bn254_plonk.hpp::plonk_table_point) as affine digits, 18 dwords each: their window tables
                                       // (MSM_FW_BITS, bn254_fw.h) are built on the device that uses them (bn254_k_comb.hip form 2)
  MsmShape shape1, shape2, shape2_rlc; // term kinds of the two MSM launches (plonk_msm1_shape / plonk_msm2_shape; _rlc: the weighted form of BN254_FLAG_RLC)
  mutable std::mutex mu;               // protects the map below (lookup / insertion / first upload); batches take contexts from the device's pool
  mutable std::map<int, PlonkDev> dev;
};
static void plonk_ctx_free(PlonkCtx& c) {
  void* ptrs[] = {c.ws, c.part, c.glv_tab, c.terms, c.flags, c.words, c.inf, c.status, c.d_in, c.d_work, c.grp_ws, c.grp_status, c.d_fail};
  for (auto q : ptrs) if (q) (void)hipFree(q);
  void* hp[] = {c.h_terms, c.h_flags, c.h_status, c.h_inf, c.h_words, c.h_in, c.h_fail};
  for (auto q : hp) if (q) (void)hipHostFree(q);
  if (c.stream) (void)hipStreamDestroy(c.stream);
  if (c.aux) (void)hipStreamDestroy(c.aux);
  if (c.ev_fork) (void)hipEventDestroy(c.ev_fork);
  if (c.ev_join) (void)hipEventDestroy(c.ev_join);
  for (auto e : c.tk) if (e) (void)hipEventDestroy(e);
  c = PlonkCtx();
}
static void plonk_dev_free(PlonkDev& d) {
  void* ptrs[] = {d.tab0, d.tab1, d.one, d.fixed_tabs, d.d_key};
  for (auto q : ptrs) if (q) (void)hipFree(q);
  for (auto& c : d.ctx) plonk_ctx_free(c);
  d.ready = false; d.tab0 = d.tab1 = d.one = d.fixed_tabs = nullptr; d.d_key = nullptr;
}
static int plonk_ensure_dev(const bn254_plonk_pvk* pvk, int device, PlonkDev** out) {
  int rc = check_device(device);
  if (rc) return rc;
  PlonkDev& d = pvk->dev[device];
  if (!d.ready) {
    if ((rc = upload(&d.tab0, pvk->tab0)) || (rc = upload(&d.tab1, pvk->tab1)) || (rc = upload(&d.one, pvk->one))) return rc;
    if ((rc = build_tables_on_device(2, pvk->fixed_pts, &d.fixed_tabs))) return rc;
    // the key and the field constants for the device-side stages
    if (sizeof(PlonkKey) != bn254_plonk_key_bytes()) return set_err(BN254_E_HIP, "PlonK key layout differs between the translation units");
    HIPCK(bn254_plonk_dev_init(device));
    if (!d.d_key) HIPCK(hipMalloc(&d.d_key, sizeof(PlonkKey)));
    HIPCK(hipMemcpy(d.d_key, &pvk->key, sizeof(PlonkKey), hipMemcpyHostToDevice));
    // known-answer check of the device stages on this GPU before the key is used there (bn254_k_plonk.hip::bn254_plonk_self_test); BN254_PLONK_SELFTEST=0 skips it
    static const bool selftest = [] { const char* e = getenv("BN254_PLONK_SELFTEST"); return !e || atoi(e) != 0; }();
    if (selftest) {
      std::string why;
      HIPCK(bn254_plonk_self_test(&pvk->key, d.d_key, &why));
      if (!why.empty()) return set_err(BN254_E_HIP, why);
    }
    d.ready = true;
  }
  *out = &d;
  return BN254_OK;
}
// Variable terms per JOINT row of an MSM launch over m_pad lanes per row (bn254_msm.h: Straus rows share the doublings of a step between their terms; 0 = one row per
// term).  A launch must still fill the GPU: two wavefronts per SIMD are 131 072 lanes, so joint rows pay from passes of tens of thousands of proofs on.
// BN254_MSM_JOINT=g forces a group size (0: never).
static int plonk_joint_g(size_t m_pad) {
  static const int env = [] { const char* e = getenv("BN254_MSM_JOINT"); return e ? atoi(e) : -1; }();
  if (env >= 0) return env > MSM_MAX_JOINT ? MSM_MAX_JOINT : env;
  return m_pad >= 49152 ? MSM_MAX_JOINT : 0;      // measured (profiles/r04_msm_joint_rows_sweep.txt): all the terms of a sum in one row, from 49 152 proofs per pass
}
// lanes an MSM launch may use at one wavefront per SIMD: the planner splits variable terms over two rows while the launch stays within it (bn254_msm.h)
static size_t msm_lane_budget() { static const size_t v = [] { const char* e = getenv("BN254_MSM_LANE_BUDGET"); long x = e ? atol(e) : 65536; return (size_t)(x < 64 ? 64 : x); }(); return v; }
// Lanes of window-table scratch a context of capacity `need` proofs must hold: the largest launch ANY batch of up to `need` proofs can make with a launch of
// `n_var` variable terms -- split (2 n_var rows) while that stays within the budget, one row per term above.  (Rounds 2-3 sized the scratch from `need`
// itself while the launch form follows the batch's own size, and a 5000-proof batch on a 5120-proof context wrote 15 MB past the end.)
static size_t plonk_scratch_lanes(size_t need, int n_var) {
  const size_t need_pad = (need + 63) & ~(size_t)63, b = msm_lane_budget() / 64 * 64;
  size_t split = 2 * (size_t)n_var * need_pad; if (split > b) split = b;
  const size_t full = (size_t)n_var * need_pad;
  return split > full ? split : full;
}
// Points (rows x items) the row buffer of a context of capacity `need` must hold for launches of `shape`: a split launch (latency form) has at most lane_budget / n_pad rows,
// so rows x items stays within the lane budget; an unsplit one has the rows of its shape's plan without joint rows (joint rows only merge rows), whatever the item count.
static size_t plonk_part_points(size_t need, const MsmShape& shape) {
  MsmPlan big;
  if (!msm_plan_build(big, shape, 64, 0, 0, 0)) return need * (size_t)MSM_MAX_ROWS;
  const size_t need_pad = (need + 63) & ~(size_t)63;
  // (a sum without variable terms, or an empty one, takes one row even when the budget has none left: two sums, two rows beyond the budget at most)
  size_t split = msm_lane_budget() + 2 * need_pad; if (split > need_pad * (size_t)MSM_MAX_ROWS) split = need_pad * (size_t)MSM_MAX_ROWS;
  const size_t full = (size_t)big.n_rows * need_pad;
  return split > full ? split : full;
}
static int shape_var(const MsmShape& sh) { int v = 0; for (int s = 0; s < sh.n_sums; s++) v += sh.n_var[s]; return v; }
// Knobs of the PlonK batch plan (bn254_set_plonk_params; the environment gives their initial values once, at load time):
//   piece      proofs per pass while a batch is a set of latency-bound chains side by side (5040: every launch of a pass is one wavefront generation and the
//              MSM launches keep their split form)
//   workers    sub-batches in flight (contexts), at most PLONK_WORKERS
//   big_from   from this many proofs a batch runs as FEW LARGE passes instead (throughput: one row per variable term, fixed windows packed, the pairing check on
//              the lane kernels with the whole Miller loop in one launch): 65 536 proofs in one pass 2.28 M proofs/s against 1.48 M as eight chains of 5040-proof passes.
//              0 (default): the plan measured on the MI355X (plonk_auto_plan, profiles/r04_plonk_plan_sweep.txt)
//   big_piece  proofs per pass of that form (at most PLONK_MAX_LAUNCH)
static std::atomic<long> g_plonk_piece{[] { long v = env_long("BN254_PLONK_PIECE", 5040); return v < 256 ? 256 : (v > PLONK_MAX_LAUNCH ? (long)PLONK_MAX_LAUNCH : v); }()};
static std::atomic<int> g_plonk_workers{[] { long v = env_long("BN254_PLONK_WORKERS", PLONK_WORKERS); return (int)(v < 1 ? 1 : (v > PLONK_WORKERS ? PLONK_WORKERS : v)); }()};
static std::atomic<long> g_plonk_big_from{[] { long v = env_long("BN254_PLONK_BIG_FROM", 0); return v < 0 ? 0 : v; }()};      // 0: the measured plan of plonk_auto_plan
// BN254_FLAG_RLC on the PlonK entry: honoured from this many proofs per pass (BN254_PLONK_RLC_MIN gives the initial value)
static std::atomic<long> g_plonk_rlc_min{[] { long v = env_long("BN254_PLONK_RLC_MIN", 8192); return v < 64 ? 64 : v; }()};
static std::atomic<long> g_plonk_big_piece{[] { long v = env_long("BN254_PLONK_BIG_PIECE", PLONK_BIG_PIECE_DEFAULT); return v < 256 ? 256 : (v > PLONK_MAX_LAUNCH ? (long)PLONK_MAX_LAUNCH : v); }()};
// the plan of a batch (bn254_plonk_verify_batch): sub-batches side by side, proofs per sub-batch, proofs per pass of a sub-batch
// The default plan by batch size (profiles/r04_plonk_plan_sweep.txt, one MI355X): chains of 5040-proof passes side by side up to ~9000 proofs (8192: 7.06 ms against
// 7.27 ms as one pass); ONE pass of the whole batch up to ~20 000 (16 384: 12.2 against 12.7 ms); TWO passes side by side up to ~40 000 (32 768: 18.3 ms against 20.2 ms
// as one pass and 22.3 ms as chains); one pass again up to 65 536 (49 152: 25.1 ms = 1.96 M proofs/s, 65 536: 28.8 ms = 2.28 M, chains 1.48 M); beyond, passes of up to
// big_piece proofs (bn254_set_plonk_params; default PLONK_BIG_PIECE_DEFAULT) on up to eight contexts (round 4, passes of 65 536: 262 144 proofs at 2.62 M proofs/s).
static void plonk_auto_plan(size_t n, size_t chain_piece, size_t big_piece, int max_workers, size_t* piece, int* workers_cap) {
  if (n <= 9000) { *piece = chain_piece; *workers_cap = max_workers; }
  else if (n <= 20000) { *piece = n; *workers_cap = 1; }
  else if (n <= 40000) { *piece = (n + 1) / 2; *workers_cap = max_workers < 2 ? max_workers : 2; }
  else if (n <= 65536) { *piece = n; *workers_cap = max_workers; }
  else { *piece = n < big_piece ? n : big_piece; *workers_cap = max_workers; }
}
static void plonk_plan(size_t n, size_t piece, int max_workers, int* workers, size_t* per, size_t* pass) {
  int w = (int)((n + piece - 1) / piece); if (w > max_workers) w = max_workers; if (w < 1) w = 1;
  const size_t p = (n + (size_t)w - 1) / (size_t)w;
  const size_t npass = (p + piece - 1) / piece;
  *workers = w; *per = p; *pass = npass ? (p + npass - 1) / npass : p;
}
// n: proofs of the largest pass the context will run; in_bytes: the proof + input bytes of such a pass (device-side stages: staged through pinned memory).  Everything a pass
// needs is sized HERE, before anything is enqueued: the run path itself neither allocates nor frees (a hipFree is a device-wide synchronisation while other contexts are in flight).
static int plonk_ensure_ctx(const bn254_plonk_pvk* pvk, PlonkCtx& c, size_t n, size_t in_bytes) {
  if (!c.stream) {
    HIPCK(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking)); HIPCK(hipStreamCreateWithFlags(&c.aux, hipStreamNonBlocking));
    HIPCK(hipEventCreateWithFlags(&c.ev_fork, hipEventDisableTiming)); HIPCK(hipEventCreateWithFlags(&c.ev_join, hipEventDisableTiming));
    for (auto& e : c.tk) HIPCK(hipEventCreate(&e));
  }
  if (in_bytes > c.in_cap) {
    if (c.d_in) HIPCK(hipFree(c.d_in));
    if (c.h_in) HIPCK(hipHostFree(c.h_in));
    c.d_in = nullptr; c.h_in = nullptr; c.in_cap = 0;
    const size_t cap = (in_bytes + 65535) / 65536 * 65536;
    HIPCK(hipMalloc((void**)&c.d_in, cap));
    HIPCK(hipHostMalloc((void**)&c.h_in, cap, hipHostMallocDefault));
    c.in_cap = cap;
  }
  size_t need = n < PLONK_MAX_LAUNCH ? (n + 255) / 256 * 256 : (size_t)PLONK_MAX_LAUNCH;
  if (need <= c.cap) return BN254_OK;
  // drop the old buffers and forget them BEFORE anything is allocated: if an allocation below fails the context is left empty (cap = 0, every
  // pointer null), never with a stale pointer that a later call or plonk_ctx_free would free a second time
  auto drop = [&c] {
    void** dp[] = {(void**)&c.ws, (void**)&c.part, (void**)&c.glv_tab, (void**)&c.terms, (void**)&c.flags, (void**)&c.words, (void**)&c.inf, (void**)&c.status, (void**)&c.d_work,
                   (void**)&c.grp_ws, (void**)&c.grp_status, (void**)&c.d_fail};
    for (auto q : dp) { if (*q) (void)hipFree(*q); *q = nullptr; }
    void** hp[] = {(void**)&c.h_terms, (void**)&c.h_flags, (void**)&c.h_status, (void**)&c.h_inf, (void**)&c.h_words, (void**)&c.h_fail};
    for (auto q : hp) { if (*q) (void)hipHostFree(*q); *q = nullptr; }
    c.cap = 0; c.glv_lanes = 0; c.part_points = 0;
  };
  drop();
  const int T1 = plonk_stage1_terms(pvk->key), TT = plonk_stage2_terms(pvk->key) + 2;
  const size_t tmax = (size_t)(TT > T1 ? TT : T1);
  // window-table scratch of the variable rows: the bound over every batch size up to `need` and both launches (plonk_scratch_lanes)
  const int v1 = shape_var(pvk->shape1), v2 = shape_var(pvk->shape2_rlc);        // (the weighted form of the second launch has one variable term more)
  const size_t tab_lanes = plonk_scratch_lanes(need, v1 > v2 ? v1 : v2);
  hipError_t e = hipSuccess;
  auto dm = [&e](void** q, size_t bytes) { if (e == hipSuccess) e = hipMalloc(q, bytes ? bytes : 1); };
  auto hm = [&e](void** q, size_t bytes) { if (e == hipSuccess) e = hipHostMalloc(q, bytes ? bytes : 1, hipHostMallocDefault); };
  dm((void**)&c.ws, need * (size_t)G16_WS_BYTES_PER_PROOF);
  size_t pp = plonk_part_points(need, pvk->shape1);                              // one projective point per row and item of a launch's plan
  { const size_t b = plonk_part_points(need, pvk->shape2), c2 = plonk_part_points(need, pvk->shape2_rlc); if (b > pp) pp = b; if (c2 > pp) pp = c2; }
  dm((void**)&c.part, pp * 27 * sizeof(int32_t));
  c.part_points = pp;
  dm((void**)&c.glv_tab, tab_lanes * (size_t)G1_GLV_TAB_BYTES_PER_LANE);       // 65536 lanes = 117 MB for capacities up to 8192 proofs
  c.glv_lanes = tab_lanes;
  dm((void**)&c.terms, need * tmax * sizeof(MsmTerm));
  dm((void**)&c.flags, need * tmax);
  dm((void**)&c.words, need * 16 * sizeof(uint32_t));
  dm((void**)&c.inf, need);
  dm((void**)&c.status, need);
  dm((void**)&c.d_work, need * bn254_plonk_work_bytes());
  const size_t groups = (need / 64 + 255) / 256 * 256;                             // need is a multiple of 256: need / 64 groups, rounded to whole workgroups
  dm((void**)&c.grp_ws, groups * (size_t)G16_WS_BYTES_PER_PROOF);
  dm((void**)&c.grp_status, groups);
  dm((void**)&c.d_fail, sizeof(uint32_t));
  hm((void**)&c.h_terms, need * tmax * sizeof(MsmTerm));
  hm((void**)&c.h_flags, need * tmax);
  hm((void**)&c.h_status, need);
  hm((void**)&c.h_inf, need);
  hm((void**)&c.h_words, need * 16 * sizeof(uint32_t));
  hm((void**)&c.h_fail, sizeof(uint32_t));
  if (e != hipSuccess) { drop(); return set_err(BN254_E_HIP, std::string("PlonK context allocation: ") + hipGetErrorString(e)); }
  c.cap = need;
  return BN254_OK;
}
// Host threads of the PlonK stages: one process-wide pool, started on first use.  (Spawning and joining 16 threads costs ~0.4 ms, and a batch
// has two host stages: 8 % of a 4096-proof batch.)  run(n, fn) executes fn(0) on the caller and fn(1..n-1) on pool threads and returns when all
// are done; jobs of concurrent callers (the sub-batch workers of a large batch, other keys) share the queue.
class HostPool {
 public:
  static HostPool& get() {
    static HostPool pool([] { unsigned hw = std::thread::hardware_concurrency(); if (hw == 0) hw = 1; return hw > 32 ? 32u : hw; }());
    return pool;
  }
  void run(unsigned n, const std::function<void(unsigned)>& fn) {
    if (n <= 1) { fn(0); return; }
    struct Job { std::mutex m; std::condition_variable c; unsigned left; bool failed = false; } job;
    job.left = n - 1;
    {
      std::lock_guard<std::mutex> lk(mu_);
      for (unsigned t = 1; t < n; t++)
        q_.emplace_back([&job, &fn, t] {
          try { fn(t); } catch (...) { std::lock_guard<std::mutex> l(job.m); job.failed = true; }   // a pool thread must not die with the job still counted
          std::lock_guard<std::mutex> l(job.m);
          if (--job.left == 0) job.c.notify_one();
        });
    }
    cv_.notify_all();
    // the queued lambdas reference `job` and `fn` on this frame: whatever fn(0) does -- including throwing -- the frame must outlive them
    struct Wait {
      Job& j;
      ~Wait() { std::unique_lock<std::mutex> lk(j.m); j.c.wait(lk, [this] { return j.left == 0; }); }
    } wait{job};
    fn(0);
    {
      std::unique_lock<std::mutex> lk(job.m);
      job.c.wait(lk, [&] { return job.left == 0; });
      if (job.failed) throw std::runtime_error("a host-pool slice failed");
    }
  }
  ~HostPool() {
    { std::lock_guard<std::mutex> lk(mu_); stop_ = true; }
    cv_.notify_all();
    for (auto& t : th_) t.join();
  }

 private:
  explicit HostPool(unsigned n) { for (unsigned i = 0; i < n; i++) th_.emplace_back([this] { loop(); }); }
  void loop() {
    for (;;) {
      std::function<void()> f;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return stop_ || !q_.empty(); });
        if (q_.empty()) return;
        f = std::move(q_.front()); q_.pop_front();
      }
      f();
    }
  }
  std::mutex mu_; std::condition_variable cv_; std::deque<std::function<void()>> q_; std::vector<std::thread> th_; bool stop_ = false;
};
template <class F> static void plonk_parallel(size_t n, unsigned hw, F&& f) {
  if (hw == 0) hw = 1;
  if (n < 64) hw = 1;
  if (hw == 1) { for (size_t i = 0; i < n; i++) f(i); return; }
  HostPool::get().run(hw, [&](unsigned t) { for (size_t i = t; i < n; i += hw) f(i); });
}


// ---- prepared keys of the single-proof entry points (bn254_groth16_verify, bn254_plonk_verify): the last KEY_CACHE_SLOTS keys by exact bytes.
// Entries are shared_ptrs: an evicted key is freed when its last in-flight call returns.  The cache object itself is never destroyed (keys hold
// device memory; freeing it from a static destructor would race the HIP runtime's own teardown).  BN254_KEY_CACHE=0 switches it off, BN254_KEY_CACHE=N (1 .. 64) sets the
// number of keys kept (a caller that rotates through more keys than slots pays the preparation, ~9 ms of an 11 ms call, on every miss).
#define KEY_CACHE_SLOTS 4
#define KEY_CACHE_MAX_SLOTS 64
template <class T, void (*FREE)(T*)>
class KeyCache {
 public:
  std::shared_ptr<T> find(const uint8_t* vk, size_t len, unsigned mode) {
    if (!slots()) return nullptr;
    std::lock_guard<std::mutex> lk(mu_);
    for (auto& e : e_)
      if (e.h && e.mode == mode && e.bytes.size() == len && memcmp(e.bytes.data(), vk, len) == 0) { e.tick = ++clock_; return e.h; }
    return nullptr;
  }
  static int capacity() { return slots(); }
  std::shared_ptr<T> insert(const uint8_t* vk, size_t len, unsigned mode, T* raw) {
    std::shared_ptr<T> h(raw, [](T* p) { FREE(p); });
    if (!slots()) return h;
    std::lock_guard<std::mutex> lk(mu_);
    Entry* v = &e_[0];
    for (auto& e : e_) { if (!e.h) { v = &e; break; } if (e.tick < v->tick) v = &e; }
    v->bytes.assign(vk, vk + len); v->mode = mode; v->h = h; v->tick = ++clock_;
    return h;
  }

 private:
  // unset: KEY_CACHE_SLOTS; 0: off; N: N slots (at most KEY_CACHE_MAX_SLOTS)
  static int slots() { static const int n = [] { const char* e = getenv("BN254_KEY_CACHE"); long v = e ? atol(e) : KEY_CACHE_SLOTS; return (int)(v < 0 ? 0 : (v > KEY_CACHE_MAX_SLOTS ? KEY_CACHE_MAX_SLOTS : v)); }(); return n; }
  struct Entry { std::vector<uint8_t> bytes; unsigned mode = 0; std::shared_ptr<T> h; uint64_t tick = 0; };
  std::mutex mu_; std::vector<Entry> e_ = std::vector<Entry>((size_t)(slots() > 0 ? slots() : 1)); uint64_t clock_ = 0;
};
static KeyCache<bn254_g16_pvk, bn254_groth16_vk_free>& g16_key_cache() { static auto* c = new KeyCache<bn254_g16_pvk, bn254_groth16_vk_free>(); return *c; }
static KeyCache<bn254_plonk_pvk, bn254_plonk_vk_free>& plonk_key_cache() { static auto* c = new KeyCache<bn254_plonk_pvk, bn254_plonk_vk_free>(); return *c; }


extern "C" {

const char* bn254_last_error(void) { return g_err.c_str(); }
const char* bn254_version(void) { return "bn254-verify-amd 0.5 (gfx950)"; }
int bn254_abi_version(void) { return BN254_ABI_VERSION; }
int bn254_dbg_key_cache_slots(void) { return KeyCache<bn254_g16_pvk, bn254_groth16_vk_free>::capacity(); }
const char* bn254_status_string(int s) {
  switch (s) {
    case BN254_REJECT: return "reject"; case BN254_ACCEPT: return "accept"; case BN254_ERR_NOT_MEMBER: return "coordinate not a field member";
    case BN254_ERR_NOT_ON_CURVE: return "point not on curve"; case BN254_ERR_NOT_IN_SUBGROUP: return "G2 point not in the r-torsion subgroup";
    case BN254_ERR_INPUT_LEN: return "wrong number of public inputs"; case BN254_ERR_MALFORMED: return "malformed input";
    case BN254_ERR_OPENING_MISMATCH: return "opening polynomial mismatch"; case BN254_ERR_PAIRING_FAILED: return "pairing check failed";
    case BN254_ERR_BSB22_MISMATCH: return "BSB22 commitment count mismatch"; case BN254_ERR_INVERSE: return "inverse not found";
    default: return "unknown";
  }
}
void bn254_set_profiling(int enabled) { g_profiling.store(enabled); g_prof_epoch++; }
void bn254_set_profile_kernels(unsigned mask) { g_prof_mask.store(mask); g_prof_epoch++; }
int bn254_groth16_num_kernel_kinds(void) { return KID_COUNT; }
const char* bn254_groth16_kernel_kind_name(int i) {
  if (i == KID_MSM_PARTIAL) { const char* e = getenv("BN254_WIDE_COMB"); if (!(e && atoi(e) == 0)) return "k_g16_msm_partial_comb"; }   // the table form in use
  return (i >= 0 && i < KID_COUNT) ? bn254_kernel_kind_names[i] : "";
}
const char* bn254_groth16_kernel_name(int i) {
  static const char* names[BN254_G16_NUM_KERNELS] = {"phase_prepare", "phase_miller", "phase_subgroup", "phase_finalexp"};
  return (i >= 0 && i < BN254_G16_NUM_KERNELS) ? names[i] : "";
}

int bn254_groth16_vk_prepare(const uint8_t* vk, size_t vk_len, unsigned mode, bn254_g16_pvk** out) {
  if (!vk || !out || mode > 1) return set_err(BN254_E_BAD_ARG, "bad argument");
  *out = nullptr;
  G16Key key;
  if (parse_g16_vk(key, vk, vk_len, (int)mode) != DEC_OK) return set_err(BN254_E_VK, "verifying key does not parse");
  bn254_g16_pvk* p = new (std::nothrow) bn254_g16_pvk();
  if (!p) return set_err(BN254_E_NOMEM, "out of memory");
  if (!prepare_g16(p->host, key, (int)mode)) { delete p; return set_err(BN254_E_VK, "no line table for a G2 element of the key (unreachable for a point on the twist: bn254_host.hpp::prepare_g16)"); }
  *out = p;
  return BN254_OK;
}
void bn254_groth16_vk_free(bn254_g16_pvk* pvk) {
  if (!pvk) return;
  for (auto& kv : pvk->dev) {
    if (hipSetDevice(kv.first) != hipSuccess) continue;
    (void)hipDeviceSynchronize();
    dev_free(kv.second);
  }
  delete pvk;
}
size_t bn254_groth16_vk_num_public(const bn254_g16_pvk* pvk) { return pvk ? (pvk->host.n_k ? pvk->host.n_k - 1 : (size_t)-1) : 0; }

int bn254_groth16_reserve(const bn254_g16_pvk* pvk, size_t n, int device) {
  if (!pvk) return set_err(BN254_E_BAD_ARG, "null key");
  DevState* d = dev_state(pvk, device);
  std::lock_guard<std::mutex> lk(d->mu);
  return ensure_dev(pvk, *d, device, n ? n : 1);
}

}  // extern "C"

// Enqueue the exact pipeline for n proofs on `user`.  Caller holds d->mu and has called ensure_dev.
static int g16_enqueue_exact(const bn254_g16_pvk* pvk, DevState* d, const void* d_proofs, size_t proof_stride, const void* d_inputs,
                             size_t n_public, size_t n, void* d_status, hipStream_t user, unsigned flags) {
  // BN254_STREAMS = 1..4 sub-batches in flight (default 2: +4.5 % over one stream at 2^20, profiles/r01_streams.txt)
  static const int n_streams = [] { const char* e = getenv("BN254_STREAMS"); int v = e ? atoi(e) : 2; return v < 1 ? 1 : (v > 4 ? 4 : v); }();
  // BN254_CHUNK_LOG2 (experiment): proofs per workspace chunk, default 2^20
  static const size_t chunk = [] { const char* e = getenv("BN254_CHUNK_LOG2"); int v = e ? atoi(e) : 20; if (v < 12) v = 12; if (v > 20) v = 20; return (size_t)1 << v; }();
  const int profiling = g_profiling.load();
  // the overlap of the sub-batch streams, measured on an earlier batch: read it once it is there (no waiting)
  if (d->ov_state == 1 && hipEventQuery(d->ov_ev[1]) == hipSuccess && hipEventQuery(d->ov_ev[3]) == hipSuccess) {
    float a0 = 0, a1 = 0, s1 = 0, e1 = 0;
    if (hipEventElapsedTime(&a0, d->ov_ev[0], d->ov_ev[1]) == hipSuccess && hipEventElapsedTime(&a1, d->ov_ev[2], d->ov_ev[3]) == hipSuccess &&
        hipEventElapsedTime(&s1, d->ov_ev[0], d->ov_ev[2]) == hipSuccess && hipEventElapsedTime(&e1, d->ov_ev[0], d->ov_ev[3]) == hipSuccess) {
      const float lo_ = s1 < 0 ? s1 : 0, hi_ = e1 > a0 ? e1 : a0;
      d->ov_ratio = (a0 + a1) / (hi_ - lo_ > 1e-6f ? hi_ - lo_ : 1e-6f);
      static const bool fallback = [] { const char* e = getenv("BN254_STREAM_FALLBACK"); return !e || atoi(e) != 0; }();
      if (d->ov_ratio < 1.15f) {
        d->ov_serial_votes++;
        if (d->ov_serial_votes >= OV_AGREE || d->ov_probe) {
          d->single_stream = fallback;
          d->diag = "the two sub-batch streams of a Groth16 batch ran one after the other on this device (overlap " + std::to_string(d->ov_ratio) + ", " +
                    std::to_string(d->ov_serial_votes) + " measurements in a row): the process's streams share a hardware queue -- set GPU_MAX_HW_QUEUES=8 before the HIP runtime "
                    "initialises (INTEGRATION.md)" + (fallback ? "; using one sub-batch per launch, re-measured every " + std::to_string(OV_REPROBE) + " batches" : "");
        }
      } else {
        d->ov_serial_votes = 0;
        if (d->single_stream) d->diag = "the sub-batch streams overlap again (" + std::to_string(d->ov_ratio) + "): back to two sub-batches side by side";
        d->single_stream = false;
      }
      // keep measuring until the question is settled either way: OV_AGREE agreeing answers
      d->ov_state = (d->ov_serial_votes > 0 && d->ov_serial_votes < OV_AGREE && !d->single_stream) ? 0 : 2;
    } else d->ov_state = 2;
    d->ov_probe = false;
  }
  // on one sub-batch per launch: every OV_REPROBE-th batch tries two streams again and is measured
  bool probe_now = false;
  if (d->single_stream && d->ov_state == 2 && ++d->ov_batches % OV_REPROBE == 0) { probe_now = true; d->ov_probe = true; d->ov_state = 0; }
  for (size_t off = 0; off < n; off += chunk) {
    size_t m = n - off < chunk ? n - off : chunk;
    G16ChunkPlan plan;
    if (!g16_plan_chunk(plan, m, pvk->host.key_inputs(), n_public, n_streams, d->single_stream && !probe_now)) return set_err(BN254_E_BAD_ARG, "batch cannot be planned");
    const bool wide = plan.wide, concurrent = plan.concurrent, split_small = plan.split_small;
    const int parts = plan.parts;
    // the buffers were sized by ensure_dev (bn254_groth16_reserve or the entry point itself): this path only enqueues, after checking the plan against them
    if (m > d->ws_cap) return set_err(BN254_E_BAD_ARG, "workspace smaller than the batch: bn254_groth16_reserve first");
    if (wide) for (int pi = 0; pi < parts; pi++)
      if (plan.part[pi].count > d->msm_part_cap) return set_err(BN254_E_BAD_ARG, "workspace of a key with many public inputs is smaller than the batch: bn254_groth16_reserve first");
    if (concurrent || split_small) { int rc = ensure_aux(*d, concurrent ? parts - 1 : 2); if (rc) return rc; }
    if (concurrent) HIPCK(hipEventRecord(d->fork_ev, user));
    // the first two sub-batches of a batch that runs several, while the question is open -- and only when the two are of (nearly) equal size: a short second part
    // beside a long first one reads as "no overlap" whatever the queues do
    const bool measure_overlap = concurrent && parts >= 2 && d->ov_state == 0 && plan.part[1].count * 10 >= plan.part[0].count * 9;
    if (measure_overlap) for (auto& e : d->ov_ev) if (!e) HIPCK(hipEventCreate(&e));
    for (int pi = 0; pi < parts; pi++) {
      const size_t lo = plan.part[pi].first, hi = lo + plan.part[pi].count;
      hipStream_t st = concurrent ? part_stream(*d, user, pi) : user;
      if (concurrent && pi < 4 && st != user) HIPCK(hipStreamWaitEvent(st, d->fork_ev, 0));
      if (measure_overlap && pi < 2) HIPCK(hipEventRecord(d->ov_ev[2 * pi], st));
      G16LaunchArgs a;
      a.proofs = (const uint8_t*)d_proofs + (off + lo) * proof_stride; a.stride = proof_stride;
      a.inputs = (const uint8_t*)d_inputs + (off + lo) * n_public * 32; a.n_public = (int)n_public; a.n = hi - lo;
      a.ws = d->ws + lo * (size_t)(G16_WS_BYTES_PER_PROOF / 4); a.status = (uint8_t*)d_status + off + lo; a.msm_tab = d->msm; a.k0 = d->k0;
      a.gtab = d->gtab; a.dtab = d->dtab; a.target = d->target;
      a.inputs_match_key = pvk->host.inputs_match(n_public) ? 1 : 0;
      a.strict_scalars = (flags & BN254_FLAG_STRICT_SCALARS) ? 1 : 0;
      a.part_of_larger = parts > 1 ? 1 : 0;
      a.msm_part = wide ? d->msm_part : nullptr;
      a.msm_comb = pvk->host.msm_comb ? 1 : 0;
      a.msm_digits = (wide && pvk->host.msm_comb) ? (uint16_t*)(d->msm_part + d->msm_chunks * 27 * d->msm_part_cap) : nullptr;
      if (split_small && parts == 1) {
        a.split_streams[0] = d->aux[0]; a.split_streams[1] = d->aux[1];
        a.split_ev[0] = d->fork_ev; a.split_ev[1] = d->join_ev[1]; a.split_ev[2] = d->join_ev[2];
      }
      // the events bracket the kernels of the LAST chunk only (one chunk for n <= 2^20)
      const bool prof_this = profiling && d->ev_ready && pi == 0;
      // mode 1: the event pairs of THIS batch; mode 2: the pairs accumulate over the batches enqueued since the last call of a profiling setter (a caller that
      // times many back-to-back batches reads them once at the end instead of synchronising with every batch; a full pool simply stops recording)
      const unsigned epoch = g_prof_epoch.load();
      const bool keep = profiling == 2 && d->prof_epoch == epoch && d->prof.used > 0;
      if (prof_this) {
        d->prof.mask = g_prof_mask.load(); d->prof_n = a.n; d->prof_epoch = epoch;
        if (!keep) { d->prof.used = 0; d->prof2.used = 0; d->prof2_used = false; }
      }
      const bool prof_second = profiling && d->ev_ready && pi == 1;
      if (prof_second) { d->prof2.mask = g_prof_mask.load(); if (!keep) d->prof2.used = 0; d->prof2_used = true; }
      hipError_t e = bn254_launch_g16(a, st, prof_this ? d->ev : nullptr, prof_this ? &d->prof : (prof_second ? &d->prof2 : nullptr));
      if (e != hipSuccess) return set_err(e == hipErrorNoBinaryForGpu || e == hipErrorInvalidDeviceFunction ? BN254_E_NO_DEVICE : BN254_E_HIP,
                                           std::string("kernel launch: ") + hipGetErrorString(e));
      if (measure_overlap && pi < 2) HIPCK(hipEventRecord(d->ov_ev[2 * pi + 1], st));
      if (concurrent && (pi + 4 >= parts) && st != user) { HIPCK(hipEventRecord(d->join_ev[pi % 4], st)); HIPCK(hipStreamWaitEvent(user, d->join_ev[pi % 4], 0)); }
    }
    if (measure_overlap) d->ov_state = 1;
  }
  d->ev_recorded = profiling && d->ev_ready;
  return BN254_OK;
}

// ---- BN254_FLAG_RLC (bn254_rlc.h): first pass in groups, exact second pass over the proofs of groups that failed -----------------------------
static int rlc_ensure(const bn254_g16_pvk* pvk, DevState* d, size_t n, size_t n_public) {
  RlcDev& r = d->rlc;
  if (!r.ready) {
    {
      std::lock_guard<std::mutex> lk(pvk->mu);
      if (!pvk->rlc_host.ready && !prepare_g16_rlc(pvk->rlc_host, pvk->host)) return set_err(BN254_E_VK, "degenerate key element (RLC tables)");
    }
    int rc;
    if ((rc = upload(&r.btab, pvk->rlc_host.btab)) || (rc = upload(&r.one, pvk->rlc_host.one))) return rc;
    if ((rc = build_tables_on_device(2, pvk->rlc_host.pts, &r.tab))) return rc;      // -alpha and K[0]: 13-bit windows like the key's own (vm_rlc_group_points reads both)
    r.ready = true;
  }
  if (n > r.grp_cap) {
    if (r.grp_status) HIPCK(hipFree(r.grp_status));
    if (r.idx) HIPCK(hipFree(r.idx));
    if (r.h_status) HIPCK(hipHostFree(r.h_status));
    if (r.h_idx) HIPCK(hipHostFree(r.h_idx));
    r.grp_status = nullptr; r.idx = nullptr; r.h_status = nullptr; r.h_idx = nullptr; r.grp_cap = r.idx_cap = r.h_cap = 0;
    const size_t cap = g16_rlc_alloc(n);               // group status regions of the launch parts are rounded up to 256 each (bn254_g16_plan.h)
    HIPCK(hipMalloc((void**)&r.grp_status, cap));
    HIPCK(hipMalloc((void**)&r.idx, cap * sizeof(uint32_t)));
    HIPCK(hipHostMalloc((void**)&r.h_status, cap, hipHostMallocDefault));
    HIPCK(hipHostMalloc((void**)&r.h_idx, cap * sizeof(uint32_t), hipHostMallocDefault));
    r.grp_cap = r.idx_cap = r.h_cap = n;
  }
  (void)n_public;
  return BN254_OK;
}
static int g16_enqueue_rlc(const bn254_g16_pvk* pvk, DevState* d, int device, const void* d_proofs, size_t proof_stride, const void* d_inputs,
                           size_t n_public, size_t n, void* d_status, hipStream_t user, unsigned flags) {
  (void)device;
  static const int n_streams = [] { const char* e = getenv("BN254_STREAMS"); int v = e ? atoi(e) : 2; return v < 1 ? 1 : (v > 4 ? 4 : v); }();
  static const int log2_group = [] { const char* e = getenv("BN254_RLC_GROUP_LOG2"); int v = e ? atoi(e) : 5; return v < 1 ? 1 : (v > 16 ? 16 : v); }();
  // proofs per lane in the Miller loop (shared accumulator, one squaring of f per lane and step): 2^BN254_RLC_SHARE_LOG2, at most the group
  static const int log2_share_env = [] { const char* e = getenv("BN254_RLC_SHARE_LOG2"); int v = e ? atoi(e) : 3; return v < 0 ? 0 : (v > 3 ? 3 : v); }();
  uint32_t key[11];
  if (getrandom(key, sizeof key, 0) != (ssize_t)sizeof key) return set_err(BN254_E_HIP, "getrandom failed: no weights for the RLC mode");
  size_t seen_checked = 0, seen_fallback = 0;
  const size_t chunk = G16_MAX_BATCH;
  for (size_t off = 0; off < n; off += chunk) {
    const size_t m = n - off < chunk ? n - off : chunk;
    int rc = rlc_ensure(pvk, d, m, n_public);
    if (rc) return rc;
    RlcDev& r = d->rlc;
    const int parts = g16_rlc_parts(m, n_streams);
    {
      const long ml0 = g_rlc_share_min_lanes.load();
      if (g16_rlc_need(m, n_streams, log2_group, log2_share_env, ml0 < 1 ? 1 : (size_t)ml0) > g16_rlc_alloc(r.grp_cap)) return set_err(BN254_E_HIP, "RLC group buffer smaller than the batch (internal sizing error)");
    }
    const bool concurrent = parts > 1;
    if (concurrent) { rc = ensure_aux(*d, parts - 1); if (rc) return rc; HIPCK(hipEventRecord(d->fork_ev, user)); }
    const size_t per = ((m + parts - 1) / parts + 255) / 256 * 256;
    size_t grp_off = 0;
    for (int pi = 0; pi < parts; pi++) {
      const size_t lo = (size_t)pi * per, hi = lo + per < m ? lo + per : m;
      if (lo >= hi) break;
      hipStream_t st = concurrent ? part_stream(*d, user, pi) : user;
      if (concurrent && pi < 4 && st != user) HIPCK(hipStreamWaitEvent(st, d->fork_ev, 0));
      G16LaunchArgs a;
      a.proofs = (const uint8_t*)d_proofs + (off + lo) * proof_stride; a.stride = proof_stride;
      a.inputs = (const uint8_t*)d_inputs + (off + lo) * n_public * 32; a.n_public = (int)n_public; a.n = hi - lo;
      a.ws = d->ws + lo * (size_t)(G16_WS_BYTES_PER_PROOF / 4); a.status = (uint8_t*)d_status + off + lo; a.msm_tab = d->msm; a.k0 = d->k0;
      a.gtab = d->gtab; a.dtab = d->dtab; a.target = d->target;
      a.inputs_match_key = 1;
      a.strict_scalars = (flags & BN254_FLAG_STRICT_SCALARS) ? 1 : 0;
      a.msm_part = nullptr;
      RlcLaunchArgs ra;
      memcpy(ra.key, key, sizeof key);
      ra.counter_base = (uint32_t)(off + lo);
      // sharing needs enough lanes to fill the GPU; small parts keep one proof per lane
      const long ml = g_rlc_share_min_lanes.load();
      const int log2_share = g16_rlc_share(a.n, log2_group, log2_share_env, ml < 1 ? 1 : (size_t)ml);
      ra.plan = rlc_plan((uint32_t)a.n, log2_group, log2_share);
      ra.grp_status = r.grp_status + grp_off; grp_off += ((size_t)ra.plan.groups + 255) / 256 * 256;
      ra.btab = r.btab; ra.rlc_tab = r.tab; ra.one = r.one;
      hipError_t e = bn254_launch_g16_rlc(a, ra, st);
      if (e != hipSuccess) return set_err(e == hipErrorNoBinaryForGpu || e == hipErrorInvalidDeviceFunction ? BN254_E_NO_DEVICE : BN254_E_HIP,
                                           std::string("kernel launch (rlc): ") + hipGetErrorString(e));
      if (concurrent && (pi + 4 >= parts) && st != user) { HIPCK(hipEventRecord(d->join_ev[pi % 4], st)); HIPCK(hipStreamWaitEvent(user, d->join_ev[pi % 4], 0)); }
    }
    // which proofs are still pending (their group's product was not one)?  One stream synchronisation per chunk.
    HIPCK(hipMemcpyAsync(r.h_status, (const uint8_t*)d_status + off, m, hipMemcpyDeviceToHost, user));
    HIPCK(hipStreamSynchronize(user));
    uint32_t cnt = 0;
    for (size_t i = 0; i < m; i++) {
      if (r.h_status[i] == BN254_ST_PENDING) r.h_idx[cnt++] = (uint32_t)i;
      else if (r.h_status[i] == BN254_ST_ACCEPT) seen_checked++;
    }
    seen_checked += cnt; seen_fallback += cnt;
    if (cnt == 0) continue;
    if (cnt > r.fb_cap || (size_t)cnt * n_public * 32 > r.fb_in_cap) {
      void* ptrs[] = {r.fb_proofs, r.fb_inputs, r.fb_status};
      for (auto q : ptrs) if (q) HIPCK(hipFree(q));
      r.fb_proofs = r.fb_inputs = r.fb_status = nullptr; r.fb_cap = r.fb_in_cap = 0;
      const size_t cap = ((size_t)cnt + 4095) / 4096 * 4096;
      HIPCK(hipMalloc((void**)&r.fb_proofs, cap * 256));
      HIPCK(hipMalloc((void**)&r.fb_inputs, cap * (n_public ? n_public : 1) * 32));
      HIPCK(hipMalloc((void**)&r.fb_status, cap));
      r.fb_cap = cap; r.fb_in_cap = cap * n_public * 32;
    }
    HIPCK(hipMemcpyAsync(r.idx, r.h_idx, (size_t)cnt * sizeof(uint32_t), hipMemcpyHostToDevice, user));
    hipError_t e = bn254_launch_gather_rows(r.fb_proofs, (const uint8_t*)d_proofs + off * proof_stride, proof_stride, 256, r.idx, cnt, user);
    if (e == hipSuccess && n_public) e = bn254_launch_gather_rows(r.fb_inputs, (const uint8_t*)d_inputs + off * n_public * 32, n_public * 32, (uint32_t)(n_public * 32), r.idx, cnt, user);
    if (e != hipSuccess) return set_err(BN254_E_HIP, std::string("gather launch: ") + hipGetErrorString(e));
    rc = g16_enqueue_exact(pvk, d, r.fb_proofs, 256, r.fb_inputs, n_public, cnt, r.fb_status, user, flags);
    if (rc) return rc;
    e = bn254_launch_scatter_status((uint8_t*)d_status + off, r.fb_status, r.idx, cnt, user);
    if (e != hipSuccess) return set_err(BN254_E_HIP, std::string("scatter launch: ") + hipGetErrorString(e));
  }
  if (seen_checked) {
    RlcDev& r = d->rlc;
    const float share = (float)seen_fallback / (float)seen_checked;
    r.fb_share = r.have_obs ? 0.5f * r.fb_share + 0.5f * share : share;
    r.have_obs = true;
  }
  return BN254_OK;
}
// The RLC pass costs about half an exact pass and every proof of a failed group pays the exact pass on top, so the mode loses once about half
// of the proofs fall back (measured: 0.84 x at 1/16 invalid proofs and groups of 32).  While the recent share is above RLC_BYPASS_SHARE the
// batch entry points run the exact path directly (same status bytes by construction) and re-measure with an RLC pass every RLC_PROBE_EVERY calls.
// BN254_RLC_ADAPTIVE=0 switches this off.
#define RLC_BYPASS_SHARE 0.45f
#define RLC_PROBE_EVERY 8
static bool rlc_bypass(RlcDev& r) {
  const bool adaptive = g_rlc_adaptive.load() != 0;   // bn254_set_rlc_params
  if (!adaptive || !r.have_obs || r.fb_share <= RLC_BYPASS_SHARE) { r.bypassed = 0; return false; }
  if (r.bypassed + 1 >= RLC_PROBE_EVERY) { r.bypassed = 0; return false; }
  r.bypassed++; r.bypassed_total++;
  return true;
}
// does a batch of this shape qualify for the RLC mode at all (the adaptive bypass, rlc_bypass, is decided separately, once per call)
static bool rlc_eligible(const bn254_g16_pvk* pvk, size_t n_public, size_t n, unsigned flags) {
  return (flags & BN254_FLAG_RLC) && pvk->host.inputs_match(n_public) && n_public <= (size_t)RLC_MAX_PUBLIC && n >= (size_t)g_rlc_min_batch.load();
}
// one batch on `user`: waits for the previous batch of this (key, device), runs the exact or the RLC pipeline, records busy_ev.
// use_rlc: -1 = decide here; 0 / 1 = the caller (the host-buffer entry, which must know before it cuts the batch into chunks) has decided
static int g16_enqueue(const bn254_g16_pvk* pvk, DevState* d, int device, const void* d_proofs, size_t proof_stride, const void* d_inputs,
                       size_t n_public, size_t n, void* d_status, hipStream_t user, unsigned flags, int use_rlc = -1) {
  if (d->busy_valid) HIPCK(hipStreamWaitEvent(user, d->busy_ev, 0));
  int rc;
  // BN254_FLAG_RLC is honoured where it pays: from RLC_PAYS_FROM proofs (bn254_set_rlc_params / BN254_RLC_MIN_BATCH at load time move the
  // threshold: the tests run the mode on small batches); smaller batches take the exact path -- same status bytes
  const bool rlc = use_rlc >= 0 ? use_rlc != 0 : (rlc_eligible(pvk, n_public, n, flags) && !rlc_bypass(d->rlc));
  if (rlc) rc = g16_enqueue_rlc(pvk, d, device, d_proofs, proof_stride, d_inputs, n_public, n, d_status, user, flags);
  else rc = g16_enqueue_exact(pvk, d, d_proofs, proof_stride, d_inputs, n_public, n, d_status, user, flags);
  if (rc) return rc;
  HIPCK(hipEventRecord(d->busy_ev, user));
  d->busy_valid = true;
  return BN254_OK;
}

extern "C" {

int bn254_groth16_verify_batch_device(const bn254_g16_pvk* pvk, const void* d_proofs, size_t proof_stride, const void* d_inputs,
                                      size_t n_public, size_t n, void* d_status, int device, void* hip_stream, unsigned flags) {
  if (!pvk || (n && (!d_proofs || !d_status)) || proof_stride < 256 || (n && n_public && !d_inputs) || (flags & ~3u)) return set_err(BN254_E_BAD_ARG, "bad argument");
  if (n == 0) return BN254_OK;
  DevState* d = dev_state(pvk, device);
  std::lock_guard<std::mutex> lk(d->mu);
  int rc = ensure_dev(pvk, *d, device, n);
  if (rc) return rc;
  return g16_enqueue(pvk, d, device, d_proofs, proof_stride, d_inputs, n_public, n, d_status, (hipStream_t)hip_stream, flags);
}

void bn254_set_rlc_params(long min_batch, int adaptive, long share_min_lanes) {
  if (min_batch >= 0) g_rlc_min_batch.store(min_batch < RLC_MIN_BATCH ? (long)RLC_MIN_BATCH : min_batch);
  if (adaptive >= 0) g_rlc_adaptive.store(adaptive ? 1 : 0);
  if (share_min_lanes >= 0) g_rlc_share_min_lanes.store(share_min_lanes < 1 ? 1 : share_min_lanes);
}

int bn254_groth16_rlc_state(const bn254_g16_pvk* pvk, int device, float* fallback_share, unsigned* bypassed_calls) {
  if (!pvk) return set_err(BN254_E_BAD_ARG, "bad argument");
  DevState* d = dev_state(pvk, device);
  std::lock_guard<std::mutex> lk(d->mu);
  if (fallback_share) *fallback_share = d->rlc.have_obs ? d->rlc.fb_share : -1.f;
  if (bypassed_calls) *bypassed_calls = d->rlc.bypassed_total;
  return BN254_OK;
}

const char* bn254_last_diagnostic(void) { return g_diag.c_str(); }

// How the two sub-batch streams of this (key, device) ran on the first batch that used two: sum of their durations / their union (about 2: side by side; about
// 1: one after the other, i.e. they share a hardware queue -- see GPU_MAX_HW_QUEUES in INTEGRATION.md); -1 while no such batch has been measured.
int bn254_groth16_stream_overlap(const bn254_g16_pvk* pvk, int device, float* overlap, int* single_stream) {
  if (!pvk || !overlap) return set_err(BN254_E_BAD_ARG, "bad argument");
  DevState* d = dev_state(pvk, device);
  std::lock_guard<std::mutex> lk(d->mu);
  *overlap = d->ov_ratio;            // -1 until the first measurement has been read
  if (single_stream) *single_stream = d->single_stream ? 1 : 0;
  g_diag = d->diag;                  // the explanation belongs to the (key, device); the caller's thread receives it here
  return BN254_OK;
}

// Host-only probe of the Groth16 plan (bn254_g16_plan.h): for a key with key_inputs public inputs (comb: its MSM tables are in comb form), a context RESERVED for
// `reserved` proofs and a batch of n proofs with n_public inputs each -- what the context allocates, and every launch the batch makes: out[] receives, per launch,
// 8 values {chunk, first proof of the chunk, proofs, stream slot (-1: caller's stream, not concurrent), form (0 lanes, 1 cooperative, 2 latency mode), Miller steps
// per launch, workspace bytes it addresses (first byte offset, one past the last)}; alloc[] = {workspace bytes, partial-sum bytes, digit bytes, proofs per wide launch}.
// Returns the number of launches through *n_launches (at most max_launches are written).
int bn254_dbg_g16_plan(size_t key_inputs, int comb, size_t reserved, size_t n, size_t n_public, int n_streams, int single_stream, uint64_t alloc[4], uint64_t* out,
                       int max_launches, int* n_launches) {
  if (!alloc || !out || !n_launches || n_streams < 1 || n_streams > 4) return set_err(BN254_E_BAD_ARG, "bad argument");
  const G16Alloc a = g16_alloc_for(reserved, key_inputs, comb != 0);
  alloc[0] = (uint64_t)a.ws_proofs * G16_WS_BYTES_PER_PROOF; alloc[1] = a.msm_part_bytes; alloc[2] = a.msm_digit_bytes; alloc[3] = a.msm_part_proofs;
  int k = 0;
  const size_t chunk = G16_MAX_BATCH;
  int ci = 0;
  for (size_t off = 0; off < n; off += chunk, ci++) {
    const size_t m = n - off < chunk ? n - off : chunk;
    G16ChunkPlan p;
    if (!g16_plan_chunk(p, m, key_inputs, n_public, n_streams, single_stream != 0)) return set_err(BN254_E_BAD_ARG, "batch cannot be planned");
    for (int pi = 0; pi < p.parts; pi++) {
      const G16Part& q = p.part[pi];
      const G16Form f = g16_launch_form(q.count, n_public, n_public == key_inputs, p.wide, p.parts > 1, p.split_small && p.parts == 1, true, -1);
      if (k < max_launches) {
        uint64_t* o = out + 8 * (size_t)k;
        o[0] = (uint64_t)ci; o[1] = q.first; o[2] = q.count; o[3] = (uint64_t)(int64_t)q.stream_slot; o[4] = (uint64_t)f.form; o[5] = (uint64_t)f.run_steps;
        o[6] = (uint64_t)q.first * G16_WS_BYTES_PER_PROOF; o[7] = (uint64_t)(q.first + q.count) * G16_WS_BYTES_PER_PROOF;
      }
      k++;
    }
  }
  *n_launches = k;
  return BN254_OK;
}

// ... and of the group status bytes of the RLC mode for a chunk of m proofs: what the launch parts address against what a context reserved for `reserved` proofs holds
int bn254_dbg_g16_rlc_plan(size_t reserved, size_t m, int n_streams, int log2_group, int log2_share, size_t min_lanes, uint64_t* need, uint64_t* alloc) {
  if (!need || !alloc || m == 0 || n_streams < 1 || n_streams > 4 || log2_group < 1 || log2_group > 16 || log2_share < 0 || log2_share > 3) return set_err(BN254_E_BAD_ARG, "bad argument");
  *need = g16_rlc_need(m, n_streams, log2_group, log2_share, min_lanes); *alloc = g16_rlc_alloc(reserved);
  return BN254_OK;
}

int bn254_groth16_last_kernel_ms(const bn254_g16_pvk* pvk, int device, float ms[BN254_G16_NUM_KERNELS]) {
  if (!pvk || !ms) return set_err(BN254_E_BAD_ARG, "bad argument");
  DevState* d = dev_state(pvk, device);
  std::lock_guard<std::mutex> lk(d->mu);
  if (!d->ev_recorded) return set_err(BN254_E_BAD_ARG, "no profiled batch on this device");
  HIPCK(hipSetDevice(device));
  HIPCK(hipEventSynchronize(d->ev[4]));
  for (int i = 0; i < BN254_G16_NUM_KERNELS; i++) HIPCK(hipEventElapsedTime(&ms[i], d->ev[i], d->ev[i + 1]));
  return BN254_OK;
}

int bn254_groth16_kernel_profile(const bn254_g16_pvk* pvk, int device, unsigned launches[], float total_ms[], size_t* proofs_per_launch) {
  if (!pvk || !launches || !total_ms) return set_err(BN254_E_BAD_ARG, "bad argument");
  DevState* dp = dev_state(pvk, device);
  std::lock_guard<std::mutex> lk(dp->mu);
  DevState& d = *dp;
  if (!d.ev_recorded) return set_err(BN254_E_BAD_ARG, "no profiled batch on this device");
  HIPCK(hipSetDevice(device));
  for (int k = 0; k < KID_COUNT; k++) { launches[k] = 0; total_ms[k] = 0.f; }
  for (int i = 0; i < d.prof.used; i++) {
    HIPCK(hipEventSynchronize(d.prof.ev[2 * i + 1]));
    float ms = 0.f;
    HIPCK(hipEventElapsedTime(&ms, d.prof.ev[2 * i], d.prof.ev[2 * i + 1]));
    launches[d.prof.kid[i]]++; total_ms[d.prof.kid[i]] += ms;
  }
  if (proofs_per_launch) *proofs_per_launch = d.prof_n;
  return BN254_OK;
}

// Launches, summed durations AND the union of the launch intervals per kernel kind over the first TWO sub-batches of the last profiled batch (they
// run on two streams side by side).  union_ms[k] = length of the union of the intervals [start, end] of every launch of kind k, on a common time
// base (HIP events of both streams against the first sub-batch's first event): for two streams that run the same kernel at the same time it is
// about one launch's duration, for launches that happen to run one after the other it is the sum -- either way "work of all those launches / union"
// is the rate the GPU delivered while that kernel kind was running.
int bn254_groth16_kernel_profile_all(const bn254_g16_pvk* pvk, int device, unsigned launches[], float total_ms[], float union_ms[], size_t* proofs_per_launch) {
  if (!pvk || !launches || !total_ms || !union_ms) return set_err(BN254_E_BAD_ARG, "bad argument");
  DevState* dp = dev_state(pvk, device);
  std::lock_guard<std::mutex> lk(dp->mu);
  DevState& d = *dp;
  if (!d.ev_recorded || d.prof.used == 0) return set_err(BN254_E_BAD_ARG, "no profiled batch on this device");
  HIPCK(hipSetDevice(device));
  std::vector<std::vector<std::pair<float, float>>> iv(KID_COUNT);
  for (int k = 0; k < KID_COUNT; k++) { launches[k] = 0; total_ms[k] = 0.f; union_ms[k] = 0.f; }
  const hipEvent_t ref = d.prof.ev[0];
  const G16Prof* ps[2] = {&d.prof, d.prof2_used ? &d.prof2 : nullptr};
  for (const G16Prof* p : ps) {
    if (!p) continue;
    for (int i = 0; i < p->used; i++) {
      HIPCK(hipEventSynchronize(p->ev[2 * i + 1]));
      float a = 0.f, b = 0.f;
      HIPCK(hipEventElapsedTime(&a, ref, p->ev[2 * i]));
      HIPCK(hipEventElapsedTime(&b, ref, p->ev[2 * i + 1]));
      launches[p->kid[i]]++; total_ms[p->kid[i]] += b - a;
      iv[p->kid[i]].push_back({a, b});
    }
  }
  for (int k = 0; k < KID_COUNT; k++) {
    auto& v = iv[k];
    std::sort(v.begin(), v.end());
    float cur_lo = 0.f, cur_hi = 0.f; bool open = false;
    for (auto& x : v) {
      if (!open) { cur_lo = x.first; cur_hi = x.second; open = true; }
      else if (x.first <= cur_hi) { if (x.second > cur_hi) cur_hi = x.second; }
      else { union_ms[k] += cur_hi - cur_lo; cur_lo = x.first; cur_hi = x.second; }
    }
    if (open) union_ms[k] += cur_hi - cur_lo;
  }
  if (proofs_per_launch) *proofs_per_launch = d.prof_n;
  return BN254_OK;
}

// Host buffers.  The caller's memory is pageable, and a hipMemcpyAsync from pageable memory is neither asynchronous nor fast (the runtime stages
// it through its own bounce buffer while the calling thread waits).  So the library keeps a ring of three PINNED pieces per (key, device): host
// threads copy piece i + 1 of the caller's buffers into the ring while piece i travels to the device (a true asynchronous copy on the copy
// stream) and the previous compute chunk runs; a compute chunk (2^17 proofs first, so that the exposed copy is short, then 2^18) waits on the
// GPU for the event of its last piece.  Only stream-scoped synchronisation, one status copy at the end.  The device lock is held for the whole
// call: the staging buffers belong to this batch until its statuses are back.
#define HOST_RING 3
static int host_ring_ensure(DevState& d, size_t piece_bytes) {
  if (piece_bytes <= d.pin_cap) return BN254_OK;
  for (int i = 0; i < HOST_RING; i++) {
    if (d.pin[i]) HIPCK(hipHostFree(d.pin[i]));
    d.pin[i] = nullptr;
  }
  d.pin_cap = 0;
  for (int i = 0; i < HOST_RING; i++) {
    HIPCK(hipHostMalloc((void**)&d.pin[i], piece_bytes, hipHostMallocDefault));
    if (!d.pin_ev[i]) HIPCK(hipEventCreateWithFlags(&d.pin_ev[i], hipEventDisableTiming));
  }
  d.pin_cap = piece_bytes;
  return BN254_OK;
}
static void parallel_copy(uint8_t* dst, const uint8_t* src, size_t bytes) {
  static const unsigned hw = [] { unsigned v = std::thread::hardware_concurrency(); const char* e = getenv("BN254_HOST_COPY_THREADS"); if (e) v = (unsigned)atoi(e); return v < 1 ? 1u : (v > 8 ? 8u : v); }();
  if (bytes < ((size_t)4 << 20) || hw == 1) { memcpy(dst, src, bytes); return; }
  const size_t per = ((bytes + hw - 1) / hw + 4095) & ~(size_t)4095;
  HostPool::get().run(hw, [&](unsigned t) {
    const size_t lo = (size_t)t * per;
    if (lo < bytes) memcpy(dst + lo, src + lo, bytes - lo < per ? bytes - lo : per);
  });
}
int bn254_groth16_verify_batch(const bn254_g16_pvk* pvk, const uint8_t* proofs, size_t proof_stride, const uint8_t* public_inputs,
                               size_t n_public, size_t n, uint8_t* status, int device, unsigned flags) {
  if (!pvk || (n && (!proofs || !status)) || proof_stride < 256 || (n && n_public && !public_inputs) || (flags & ~3u)) return set_err(BN254_E_BAD_ARG, "bad argument");
  if (n == 0) return BN254_OK;
  DevState* d = dev_state(pvk, device);
  std::lock_guard<std::mutex> lk(d->mu);
  int rc = ensure_dev(pvk, *d, device, n);
  if (rc) return rc;
  const size_t in_row = n_public * 32, row = proof_stride + in_row;
  size_t pb = n * proof_stride, ib = n * in_row;
  if ((rc = grow(&d->st_proofs, &d->st_proofs_cap, pb)) || (rc = grow(&d->st_inputs, &d->st_inputs_cap, ib ? ib : 32)) ||
      (rc = grow(&d->st_status, &d->st_status_cap, n)))
    return rc;
  if (!d->host_stream) { HIPCK(hipStreamCreateWithFlags(&d->host_stream, hipStreamNonBlocking)); HIPCK(hipStreamCreateWithFlags(&d->copy_stream, hipStreamNonBlocking)); }
  // compute chunks: a short first one (its copy is the only exposed one: 2^17 proofs = 42 MB, under a millisecond of DMA), then the rest in chunks
  // as large as the workspace allows -- every chunk boundary drains both sub-batch streams, so fewer chunks is faster
  static const size_t first_chunk = [] { const char* e = getenv("BN254_HOST_FIRST_CHUNK_LOG2"); int v = e ? atoi(e) : 17; if (v < 12) v = 12; if (v > 20) v = 20; return (size_t)1 << v; }();
  const size_t hchunk = (size_t)G16_MAX_BATCH - first_chunk;
  // copy pieces: about 20 MB of the caller's bytes each (65536 proofs at 2 public inputs), a multiple of 256 proofs
  static const size_t piece_bytes_target = [] { const char* e = getenv("BN254_HOST_PIECE_MB"); long v = e ? atol(e) : 20; return (size_t)(v < 1 ? 1 : v) << 20; }();
  size_t piece = piece_bytes_target / row / 256 * 256;
  if (piece < 256) piece = 256;
  if (piece > n) piece = (n + 255) / 256 * 256;
  if ((rc = host_ring_ensure(*d, piece * row))) return rc;
  // the RLC mode forms its groups over the whole batch it is handed: keep it in one piece -- but only when this call really runs the mode
  // (same predicate as g16_enqueue, the adaptive bypass included, decided ONCE here); a flag that will be ignored keeps the chunked
  // copy / compute overlap
  const int use_rlc = (rlc_eligible(pvk, n_public, n, flags) && !rlc_bypass(d->rlc)) ? 1 : 0;
  static const bool timing = getenv("BN254_HOST_TIMING") != nullptr;   // diagnostics on stderr: where the host thread spends the call
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
  double t_copy = 0, t_wait = 0, t_enq = 0;
  const auto t_begin = now();
  size_t copied = 0, computed = 0, slot_uses = 0;
  // the first chunk is short (its copy is the only exposed one) unless the batch is small anyway
  size_t c_end = (use_rlc || n < 2 * first_chunk) ? n : first_chunk;
  while (computed < n) {
    hipEvent_t last = nullptr;
    while (copied < c_end) {
      const size_t m = c_end - copied < piece ? c_end - copied : piece;
      const int slot = (int)(slot_uses % HOST_RING);
      auto ta = now();
      if (slot_uses >= HOST_RING) HIPCK(hipEventSynchronize(d->pin_ev[slot]));     // the piece that used this slot has left for the device
      auto tb = now();
      parallel_copy(d->pin[slot], proofs + copied * proof_stride, m * proof_stride);
      if (in_row) parallel_copy(d->pin[slot] + m * proof_stride, public_inputs + copied * in_row, m * in_row);
      auto tc = now();
      t_wait += ms(ta, tb); t_copy += ms(tb, tc);
      HIPCK(hipMemcpyAsync(d->st_proofs + copied * proof_stride, d->pin[slot], m * proof_stride, hipMemcpyHostToDevice, d->copy_stream));
      if (in_row) HIPCK(hipMemcpyAsync(d->st_inputs + copied * in_row, d->pin[slot] + m * proof_stride, m * in_row, hipMemcpyHostToDevice, d->copy_stream));
      HIPCK(hipEventRecord(d->pin_ev[slot], d->copy_stream));
      last = d->pin_ev[slot];
      slot_uses++; copied += m;
    }
    if (last) HIPCK(hipStreamWaitEvent(d->host_stream, last, 0));
    auto td = now();
    rc = g16_enqueue(pvk, d, device, d->st_proofs + computed * proof_stride, proof_stride, d->st_inputs + computed * in_row, n_public, c_end - computed,
                     d->st_status + computed, d->host_stream, flags, use_rlc);
    if (rc) {   // pieces of the pinned ring and earlier chunks may still be in flight: the ring and the staging buffers must be quiescent when the lock is released
      const std::string keep = g_err;
      (void)hipStreamSynchronize(d->copy_stream); (void)hipStreamSynchronize(d->host_stream);
      g_err = keep;
      return rc;
    }
    t_enq += ms(td, now());
    computed = c_end;
    c_end = n - c_end < hchunk ? n : c_end + hchunk;
  }
  const auto t_enqueued = now();
  HIPCK(hipMemcpyAsync(status, d->st_status, n, hipMemcpyDeviceToHost, d->host_stream));
  HIPCK(hipStreamSynchronize(d->host_stream));
  if (timing) fprintf(stderr, "host-buffer batch %zu: pieces of %zu proofs; host copies %.2f ms, ring waits %.2f ms, kernel enqueue %.2f ms, all enqueued after %.2f ms, done after %.2f ms\n",
                      n, piece, t_copy, t_wait, t_enq, ms(t_begin, t_enqueued), ms(t_begin, now()));
  return BN254_OK;
}

// The shard plan of a multi-device batch: the devices selected by device_mask in ascending order, shard k = the
// contiguous range [first[k], first[k] + count[k]) of the batch on devices[k]; balanced, the first n % w shards one proof longer (the same
// rule as sharding.shard_bounds of the multi-process job).  Host arithmetic only: needs no GPU, device_count is the caller's.
int bn254_shard_plan(size_t n, uint64_t device_mask, int device_count, int devices[64], size_t first[64], size_t count[64], int* n_shards) {
  if (!device_mask || !devices || !first || !count || !n_shards) return set_err(BN254_E_BAD_ARG, "bad argument");
  int w = 0;
  for (int b = 0; b < 64; b++)
    if ((device_mask >> b) & 1) {
      if (b >= device_count) return set_err(BN254_E_BAD_ARG, "device_mask selects a device that does not exist");
      devices[w++] = b;
    }
  const size_t base = n / (size_t)w, rem = n % (size_t)w;
  for (int r = 0; r < w; r++) { first[r] = (size_t)r * base + ((size_t)r < rem ? (size_t)r : rem); count[r] = base + ((size_t)r < rem ? 1 : 0); }
  *n_shards = w;
  return BN254_OK;
}

// ---- the gather of a multi-PROCESS job (one process per GPU): one ncclAllGather of the ranks' status bytes -----------------
// RCCL is not linked: a host that runs such a job already has it in its process (it created the communicator), so the symbol is looked up at the
// first call -- among the objects already loaded, then in librccl.so.
typedef int (*nccl_all_gather_fn)(const void*, void*, size_t, int, void*, hipStream_t);
static nccl_all_gather_fn rccl_all_gather() {
  static nccl_all_gather_fn fn = [] {
    void* sym = dlsym(RTLD_DEFAULT, "ncclAllGather");
    if (!sym) { void* h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL); if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL); if (h) sym = dlsym(h, "ncclAllGather"); }
    return (nccl_all_gather_fn)sym;
  }();
  return fn;
}
int bn254_status_all_gather(void* nccl_comm, int world, int rank, const void* d_local, size_t n, void* d_full, void* d_scratch, void* hip_stream) {
  if (!nccl_comm || world <= 0 || world > 64 || rank < 0 || rank >= world || (n && (!d_local || !d_full))) return set_err(BN254_E_BAD_ARG, "bad argument");
  if (n == 0) return BN254_OK;
  int devs[64], nsh = 0; size_t first[64], cnt[64];
  int rc = bn254_shard_plan(n, world == 64 ? ~0ull : ((1ull << world) - 1), world, devs, first, cnt, &nsh);   // the ranks' contiguous ranges
  if (rc) return rc;
  nccl_all_gather_fn ag = rccl_all_gather();
  if (!ag) return set_err(BN254_E_HIP, "ncclAllGather not found: RCCL is neither loaded in this process nor loadable as librccl.so");
  hipStream_t s = (hipStream_t)hip_stream;
  const size_t cap = (n + (size_t)world - 1) / (size_t)world;
  const int nccl_uint8 = 1;
  if (n % (size_t)world == 0) {
    // equal shards: the gathered blocks ARE the status vector
    int e = ag(d_local, d_full, cap, nccl_uint8, nccl_comm, s);
    return e ? set_err(BN254_E_HIP, "ncclAllGather failed (" + std::to_string(e) + ")") : BN254_OK;
  }
  // ragged: every rank sends a block of `cap` bytes (its shard, padded), in place in the scratch; the blocks are then packed into the vector
  if (!d_scratch) return set_err(BN254_E_BAD_ARG, "n is not a multiple of the world size: the gather needs world * ceil(n / world) bytes of scratch");
  uint8_t* sc = (uint8_t*)d_scratch;
  HIPCK(hipMemsetAsync(sc + (size_t)rank * cap, 0, cap, s));
  HIPCK(hipMemcpyAsync(sc + (size_t)rank * cap, d_local, cnt[rank], hipMemcpyDeviceToDevice, s));
  int e = ag(sc + (size_t)rank * cap, sc, cap, nccl_uint8, nccl_comm, s);
  if (e) return set_err(BN254_E_HIP, "ncclAllGather failed (" + std::to_string(e) + ")");
  for (int r = 0; r < world; r++)
    if (cnt[r]) HIPCK(hipMemcpyAsync((uint8_t*)d_full + first[r], sc + (size_t)r * cap, cnt[r], hipMemcpyDeviceToDevice, s));
  return BN254_OK;
}

int bn254_groth16_verify_batch_multi(const bn254_g16_pvk* pvk, const uint8_t* proofs, size_t proof_stride, const uint8_t* public_inputs,
                                     size_t n_public, size_t n, uint8_t* status, uint64_t device_mask, unsigned flags) {
  if (!pvk || !device_mask) return set_err(BN254_E_BAD_ARG, "bad argument");
  int cnt = 0;
  if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0) return set_err(BN254_E_NO_DEVICE, "no HIP device available (this library has no CPU fallback)");
  int devs[64], nsh = 0; size_t los[64], cnts[64];
  int prc = bn254_shard_plan(n, device_mask, cnt, devs, los, cnts, &nsh);
  if (prc) return prc;
  const size_t w = (size_t)nsh;
  if (w == 1) return bn254_groth16_verify_batch(pvk, proofs, proof_stride, public_inputs, n_public, n, status, devs[0], flags);
  // one host thread per device drives its shard
  std::vector<int> rcs(w, BN254_OK); std::vector<std::string> errs(w);
  std::vector<std::thread> th;
  for (size_t r = 0; r < w; r++) {
    const size_t lo = los[r], cntp = cnts[r];
    th.emplace_back([&, r, lo, cntp]() {
      if (!cntp) return;
      rcs[r] = bn254_groth16_verify_batch(pvk, proofs + lo * proof_stride, proof_stride, public_inputs ? public_inputs + lo * n_public * 32 : nullptr, n_public, cntp,
                                          status + lo, devs[r], flags);
      if (rcs[r]) errs[r] = g_err;   // thread-local in the worker
    });
  }
  for (auto& t : th) t.join();
  for (size_t r = 0; r < w; r++) if (rcs[r]) return set_err(rcs[r], "device " + std::to_string(devs[r]) + ": " + errs[r]);
  return BN254_OK;
}

// load_groth16_proof_from_bytes (groth16/converter.rs:14-26) on the host, for the one case in which no kernel can run: a single proof against key bytes that do not
// load.  A, B, C in this order; per point: every coordinate < p (Field(NotMember)), the curve equation (Group(NotOnCurve)), and for B the r-torsion (Group(NotInSubgroup)).
static uint8_t g16_proof_loader_status(const uint8_t* p /* 256 bytes */) {
  auto g1 = [](const uint8_t* b) -> uint8_t {
    if (!be_lt_p(b) || !be_lt_p(b + 32)) return BN254_ERR_NOT_MEMBER;
    G1Aff a; a.x = fp_from_be(b); a.y = fp_from_be(b + 32);
    return g1_on_curve(a) ? BN254_ACCEPT : BN254_ERR_NOT_ON_CURVE;
  };
  uint8_t st = g1(p);
  if (st != BN254_ACCEPT) return st;
  for (int i = 0; i < 4; i++) if (!be_lt_p(p + 64 + 32 * i)) return BN254_ERR_NOT_MEMBER;
  G2Aff b; b.x.c1 = fp_from_be(p + 64); b.x.c0 = fp_from_be(p + 96); b.y.c1 = fp_from_be(p + 128); b.y.c0 = fp_from_be(p + 160);
  if (!g2_on_curve(b)) return BN254_ERR_NOT_ON_CURVE;
  if (!g2_in_subgroup(b)) return BN254_ERR_NOT_IN_SUBGROUP;
  return g1(p + 192);
}
int bn254_groth16_verify(const uint8_t* proof, size_t proof_len, const uint8_t* vk, size_t vk_len, const uint8_t* public_inputs,
                         size_t n_public, unsigned mode, uint8_t* status) {
  if (!proof || !vk || !status || mode > 1) return set_err(BN254_E_BAD_ARG, "bad argument");
  // reference order: the proof is loaded (and its errors surface) before the key (lib.rs:45-46).  A short proof buffer is a
  // slice-index panic there.
  if (proof_len < 256) { *status = BN254_ERR_MALFORMED; return BN254_OK; }
  // the reference parses the key on every call (lib.rs:46); here the prepared form of the last few keys is kept (exact byte match), so a
  // caller that verifies one proof at a time against the same key pays the preparation (9 ms of an 11 ms call) once
  std::shared_ptr<bn254_g16_pvk> pvk = g16_key_cache().find(vk, vk_len, mode);
  if (!pvk) {
    bn254_g16_pvk* raw = nullptr;
    int rc = bn254_groth16_vk_prepare(vk, vk_len, mode, &raw);
    if (rc == BN254_E_VK) {
      // the key does not load (lib.rs:46 panics) -- but the proof was loaded first (lib.rs:45), so its loader error wins.  Nothing can be launched without a key: the
      // loader's checks (< p, curve equation, r-torsion of B; groth16/converter.rs:14-26) run here on the host, once, for this one proof
      const uint8_t ps = g16_proof_loader_status(proof);
      *status = ps == BN254_ACCEPT ? (uint8_t)BN254_ERR_MALFORMED : ps;
      return BN254_OK;
    }
    if (rc) return rc;
    pvk = g16_key_cache().insert(vk, vk_len, mode, raw);
  }
  return bn254_groth16_verify_batch(pvk.get(), proof, proof_len, public_inputs, n_public, 1, status, 0, 0);
}

int bn254_groth16_proof_write_raw(const uint8_t a[64], const uint8_t b[128], const uint8_t c[64], uint8_t out[BN254_GROTH16_RAW_PROOF_LEN]) {
  if (!a || !b || !c || !out) return set_err(BN254_E_BAD_ARG, "bad argument");
  memcpy(out, a, 64); memcpy(out + 64, b, 128); memcpy(out + 192, c, 64);
  memset(out + 256, 0, BN254_GROTH16_RAW_PROOF_LEN - 256);   // u32 nbCommitments = 0, then the 64-byte commitment PoK (zero)
  return BN254_OK;
}

// ---------------------------------------------------------------- PlonK (BASELINE configs[3]): entry points
int bn254_plonk_vk_prepare(const uint8_t* vk, size_t vk_len, bn254_plonk_pvk** out) {
  if (!vk || !out) return set_err(BN254_E_BAD_ARG, "bad argument");
  *out = nullptr;
  auto* p = new bn254_plonk_pvk();
  if (parse_plonk_vk(p->key, vk, vk_len) != DEC_OK) { delete p; return set_err(BN254_E_VK, "PlonK verifying key does not parse"); }
  // line tables of the two KZG G2 points (kzg.rs:175-187: e(P0, g2[0]) e(P1, g2[1]) == 1); target = 1 in GT
  std::vector<FixedLine> t0(BN_ATE_STEPS), t1(BN_ATE_STEPS);
  if (!fixed_line_table(t0.data(), p->key.kzg_g2[0]) || !fixed_line_table(t1.data(), p->key.kzg_g2[1])) { delete p; return set_err(BN254_E_VK, "no line table for a KZG G2 point (unreachable for a point on the twist: bn254_host.hpp::prepare_g16)"); }
  p->tab0.resize((size_t)BN_ATE_STEPS * FIXED_LINE_DWORDS); p->tab1.resize((size_t)BN_ATE_STEPS * FIXED_LINE_DWORDS);
  for (int s = 0; s < BN_ATE_STEPS; s++) {
    int32_t* a = p->tab0.data() + (size_t)s * FIXED_LINE_DWORDS; int32_t* b = p->tab1.data() + (size_t)s * FIXED_LINE_DWORDS;
    put_fp2(a, t0[s].m); put_fp2(a + 2 * BN_NL, t0[s].c); put_fp2(a + 4 * BN_NL, t0[s].xc);
    put_fp2(b, t1[s].m); put_fp2(b + 2 * BN_NL, t1[s].c); put_fp2(b + 4 * BN_NL, t1[s].xc);
  }
  p->one.resize(12 * BN_NL);
  put_fp12(p->one.data(), fp12_one());
  // byte-window tables of the key's G1 points that enter the MSMs with per-proof scalars (plonk/verify.rs:253-284: ql, qr, qm, qo, qk, s3; plonk/kzg.rs:74-85:
  // s1, s2, qcp[]; kzg.rs:169: the KZG generator): 32 mixed additions per term instead of a 128-step double-and-add chain
  {
    // built on the device that uses them (bn254_k_comb.hip form 2: MSM_FW_WINDOWS windows of MSM_FW_BITS bits, bn254_fw.h): the host keeps the points
    const int nt = plonk_num_tables(p->key);
    p->fixed_pts.resize((size_t)nt * 2 * BN_NL);
    for (int i = 0; i < nt; i++) { const G1Aff& q = plonk_table_point(p->key, i); fp_to_limbs(p->fixed_pts.data() + (size_t)i * 2 * BN_NL, q.x); fp_to_limbs(p->fixed_pts.data() + (size_t)i * 2 * BN_NL + BN_NL, q.y); }
  }
  plonk_msm1_shape(p->key, p->shape1); plonk_msm2_shape(p->key, p->shape2); plonk_msm2_shape(p->key, p->shape2_rlc, true);
  *out = p;
  return BN254_OK;
}
void bn254_plonk_vk_free(bn254_plonk_pvk* pvk) {
  if (!pvk) return;
  for (auto& kv : pvk->dev) { if (hipSetDevice(kv.first) != hipSuccess) continue; (void)hipDeviceSynchronize(); plonk_dev_free(kv.second); }
  delete pvk;
}
size_t bn254_plonk_vk_num_public(const bn254_plonk_pvk* pvk) { return pvk ? (size_t)pvk->key.nb_public : 0; }

// One MSM launch of a sub-batch: plan the rows for this batch size (bn254_msm.h: a pure function of the launch's term kinds, the item count and the lane
// budget), check the plan against what the context holds -- the launch form follows the BATCH, the buffers the context's CAPACITY -- and enqueue rows + sums.
static int plonk_msm(const PlonkDev* d, PlonkCtx& c, const MsmShape& shape, size_t m, int n_terms, bool to_words, size_t* lanes_out, hipEvent_t ev_rows) {
  MsmPlan plan;
  const size_t m_pad = (m + 63) & ~(size_t)63;
  // BN254_MSM_SPLIT_AT (experiments): the bit position at which the variable terms' low and high rows meet, instead of the planner's choice
  static const int force_a = [] { const char* e = getenv("BN254_MSM_SPLIT_AT"); int v = e ? atoi(e) : 0; return (v >= 2 && v <= 126 && !(v & 1)) ? v : 0; }();
  if (!msm_plan_build(plan, shape, m_pad, msm_lane_budget(), force_a, plonk_joint_g(m_pad))) return set_err(BN254_E_BAD_ARG, "PlonK key shape needs more MSM rows than the launch supports");
  if (m > c.cap || bn254_g1_msm_scratch_lanes(plan, m) > c.glv_lanes || (size_t)plan.n_rows * m > c.part_points || (size_t)plan.n_rows > (size_t)MSM_MAX_ROWS)
    return set_err(BN254_E_HIP, "PlonK context smaller than the launch (internal sizing error)");
  hipError_t e = bn254_launch_g1_msm_rows(plan, (const int32_t*)c.terms, c.flags, m, n_terms, c.part, c.glv_tab, d->fixed_tabs, c.stream);
  if (ev_rows) HIPCK(hipEventRecord(ev_rows, c.stream));
  if (e == hipSuccess)
    e = to_words ? bn254_launch_g1_sum_rows(plan, c.part, m, c.words, c.inf, nullptr, nullptr, 0, 0, 0, 0, c.stream)
                 : bn254_launch_g1_sum_rows(plan, c.part, m, nullptr, nullptr, c.ws, c.status, VE_LX_ELEM, BN254_ST_LINF, VE_CX_ELEM, BN254_ST_LINF2, c.stream);
  if (e != hipSuccess) return set_err(BN254_E_HIP, std::string("MSM launch: ") + hipGetErrorString(e));
  if (lanes_out) *lanes_out = (size_t)plan.n_rows * m_pad;
  return BN254_OK;
}
// one sub-batch [0, m) on its context: stage 1 (host) -> digest MSM (GPU) -> stage 2 (host) -> folding MSMs + pairing check (GPU) -> statuses
static int plonk_run(const bn254_plonk_pvk* pvk, const PlonkDev* d, PlonkCtx& c, int device, const uint8_t* proofs, size_t proof_stride, const uint8_t* public_inputs,
                     size_t n_public, size_t m, uint8_t* status, unsigned host_threads) {
  HIPCK(hipSetDevice(device));
  const PlonkKey& key = pvk->key;
  const int T1 = plonk_stage1_terms(key), T2 = plonk_stage2_terms(key);
  static const bool timing = getenv("BN254_PLONK_TIMING") != nullptr;   // stage durations on stderr (diagnostics)
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
  auto t0 = now();
  // per-proof scratch of the context (every field a stage reads is written by an earlier stage of the same call: no clearing needed); the term
  // and flag rows are cleared by the host thread that fills them
  if (c.work.size() < m) c.work.resize(m);
  std::vector<PlonkWork>& work = c.work;
  // The KZG batching scalar of every proof: fresh, uniform and unpredictable to the prover, as the reference draws it
  // (Fr::random(&mut OsRng), plonk/kzg.rs:149-154).  It MUST be secret until the proof is fixed: the two opening quotients are bound by
  // no transcript, so a prover who knows lambda can shift them by (lambda D, -D) and cancel a wrong evaluation
  // (tests/test_oracle_golden.py::test_kzg_batching_scalar_must_be_unpredictable).  A ChaCha20 key and nonce from getrandom(2) per call;
  // proof i takes the 384 bits of blocks 3i .. 3i+2 reduced mod r (the host threads expand them: 192 KB of getrandom per 4096 proofs took 0.7 ms).
  ChaChaKey lam_key;
  {
    uint8_t seed[44];
    for (size_t got = 0; got < sizeof seed;) {
      ssize_t k = getrandom(seed + got, sizeof seed - got, 0);
      if (k <= 0) return set_err(BN254_E_HIP, "getrandom failed: no KZG batching scalars");
      got += (size_t)k;
    }
    memcpy(lam_key.k, seed, 32); memcpy(lam_key.nonce, seed + 32, 12);
  }
  // ---- stage 1 on the host threads.  Every thread runs the first half of the stage for its proofs, inverts the products of their denominators
  // with ONE field inversion (Montgomery's trick across proofs; the inversion is a third of the stage's time per proof) and runs the second half.
  {
    unsigned hw = host_threads ? host_threads : 1; if (m < 64) hw = 1;
    auto slice = [&](unsigned t) {
      const FrCtx& F = fr_ctx();
      const size_t cnt = (m - t + hw - 1) / hw;
      std::vector<PlonkStage1> s1(cnt);
      std::vector<FrM> pre(cnt);
      FrM run = F.one;
      size_t k = 0;
      for (size_t i = t; i < m; i += hw, k++) {
        memset(&c.h_terms[i * T1], 0, (size_t)T1 * sizeof(MsmTerm)); memset(&c.h_flags[i * T1], 0, (size_t)T1);
        {
          uint32_t lw[12];
          for (int j = 0; j < 3; j++) chacha20_block4(lw + 4 * j, lam_key, (uint32_t)(3 * i + j));
          work[i].lambda = F.from_be_reduce((const uint8_t*)lw, 48);
        }
        work[i].status = s1[k].a(key, proofs + i * proof_stride, proof_stride, public_inputs + i * n_public * 32, n_public, work[i]);
        pre[k] = run;
        if (work[i].status == PL_OK) run = F.mul(run, s1[k].acc);      // acc != 0: a product of non-zero denominators
      }
      FrM inv = F.inverse(run);
      for (size_t i = t + (cnt - 1) * hw; k-- > 0; i -= hw) {
        if (work[i].status != PL_OK) continue;
        const FrM ai = F.mul(inv, pre[k]);
        inv = F.mul(inv, s1[k].acc);
        work[i].status = s1[k].b(ai, &c.h_terms[i * T1], &c.h_flags[i * T1]);
      }
    };
    if (hw == 1) slice(0);
    else HostPool::get().run(hw, [&](unsigned t) { slice(t); });
  }
  auto t1_ = now();
  // ---- the linearised-polynomial digest on the GPU, back to the host for the folding transcript
  HIPCK(hipMemcpyAsync(c.terms, c.h_terms, m * T1 * sizeof(MsmTerm), hipMemcpyHostToDevice, c.stream));
  HIPCK(hipMemcpyAsync(c.flags, c.h_flags, m * (size_t)T1, hipMemcpyHostToDevice, c.stream));   // GLV signs (bn254_plonk.hpp::put_term)
  HIPCK(hipEventRecord(c.tk[0], c.stream));
  HIPCK(hipEventRecord(c.tk[1], c.stream));
  int mrc = plonk_msm(d, c, pvk->shape1, m, T1, true, &c.last_lanes[0], c.tk[2]);
  if (mrc) return mrc;
  hipError_t e = hipSuccess;
  HIPCK(hipEventRecord(c.tk[3], c.stream));
  HIPCK(hipMemcpyAsync(c.h_words, c.words, m * 16 * sizeof(uint32_t), hipMemcpyDeviceToHost, c.stream));
  HIPCK(hipMemcpyAsync(c.h_inf, c.inf, m, hipMemcpyDeviceToHost, c.stream));
  HIPCK(hipStreamSynchronize(c.stream));
  auto t2_ = now();
  // ---- stage 2 on the host threads: the terms of P0 (T2 of them) and of P1 (2) side by side, so that ONE launch does all scalar multiplications
  const int TT = T2 + 2;
  plonk_parallel(m, host_threads, [&](size_t i) {
    memset(&c.h_terms[i * TT], 0, (size_t)TT * sizeof(MsmTerm)); memset(&c.h_flags[i * TT], 0, (size_t)TT);
    if (work[i].status == PL_OK) {
      work[i].pr.raw = proofs + i * proof_stride;
      plonk_stage2(key, proofs + i * proof_stride, work[i], &c.h_words[i * 16], c.h_inf[i] != 0, &c.h_terms[i * TT], &c.h_flags[i * TT], &c.h_terms[i * TT + T2]);
      c.h_status[i] = BN254_ST_PENDING;
    } else {
      c.h_status[i] = (uint8_t)work[i].status;
    }
  });
  auto t3_ = now();
  // ---- P0, P1 and the pairing check on the GPU: everything on the context's stream, no host wait in between
  HIPCK(hipMemcpyAsync(c.status, c.h_status, m, hipMemcpyHostToDevice, c.stream));
  HIPCK(hipMemcpyAsync(c.terms, c.h_terms, m * TT * sizeof(MsmTerm), hipMemcpyHostToDevice, c.stream));
  HIPCK(hipMemcpyAsync(c.flags, c.h_flags, m * (size_t)TT, hipMemcpyHostToDevice, c.stream));
  HIPCK(hipEventRecord(c.tk[4], c.stream));
  mrc = plonk_msm(d, c, pvk->shape2, m, TT, false, &c.last_lanes[1], c.tk[5]);
  if (mrc) return mrc;
  HIPCK(hipEventRecord(c.tk[6], c.stream));
  e = bn254_launch_pairing2_fixed(c.ws, c.status, m, d->tab0, d->tab1, d->one, BN254_ERR_PAIRING_FAILED, c.stream, c.aux, c.ev_fork, c.ev_join);
  if (e != hipSuccess) return set_err(BN254_E_HIP, std::string("pairing launch: ") + hipGetErrorString(e));
  HIPCK(hipEventRecord(c.tk[7], c.stream));
  HIPCK(hipMemcpyAsync(c.h_status, c.status, m, hipMemcpyDeviceToHost, c.stream));
  HIPCK(hipStreamSynchronize(c.stream));
  memcpy(status, c.h_status, m);
  {
    // slots as bn254_plonk_last_timing names them; the stages ran on host threads: [0] stage 1, [4] stage 2 are host wall times, [1] is 0
    auto t4_ = now();
    c.last_ms[0] = (float)ms(t0, t1_); c.last_ms[1] = 0.f; c.last_ms[4] = (float)ms(t2_, t3_); c.last_ms[8] = (float)ms(t0, t4_);
    HIPCK(hipEventElapsedTime(&c.last_ms[2], c.tk[1], c.tk[2])); HIPCK(hipEventElapsedTime(&c.last_ms[3], c.tk[2], c.tk[3]));
    HIPCK(hipEventElapsedTime(&c.last_ms[5], c.tk[4], c.tk[5])); HIPCK(hipEventElapsedTime(&c.last_ms[6], c.tk[5], c.tk[6]));
    HIPCK(hipEventElapsedTime(&c.last_ms[7], c.tk[6], c.tk[7]));
    c.last_valid = true;
  }
  if (timing) fprintf(stderr, "plonk sub-batch %zu: stage1 %.2f ms, msm1 %.2f ms, stage2 %.2f ms, msm2+pairing %.2f ms\n", m, ms(t0, t1_), ms(t1_, t2_), ms(t2_, t3_), ms(t3_, now()));
  return BN254_OK;
}

// The same sub-batch with BOTH host stages on the device (bn254_k_plonk.hip): one H2D copy of the proofs and inputs, stage 1 -> digest MSM -> stage 2 ->
// folding MSMs -> pairing check on the context's stream without a host wait in between, one D2H copy of the status bytes.
// resident: proofs / public_inputs / status are DEVICE memory of `device` (bn254_plonk_verify_batch_device): no staging copy, the status bytes leave with a device-to-device copy.
static int plonk_run_device(const bn254_plonk_pvk* pvk, const PlonkDev* d, PlonkCtx& c, int device, const uint8_t* proofs, size_t proof_stride, const uint8_t* public_inputs,
                            size_t n_public, size_t m, uint8_t* status, unsigned flags, bool resident) {
  HIPCK(hipSetDevice(device));
  const PlonkKey& key = pvk->key;
  const int T1 = plonk_stage1_terms(key), T2 = plonk_stage2_terms(key), TT = T2 + 2;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
  auto t0 = now();
  const size_t pb = m * proof_stride, ib = m * n_public * 32, need = pb + ib;
  if (!resident && need > c.in_cap) return set_err(BN254_E_HIP, "PlonK context staging smaller than the pass (internal sizing error)");   // sized by plonk_ensure_ctx
  if (m > c.cap) return set_err(BN254_E_HIP, "PlonK context smaller than the pass (internal sizing error)");
  uint32_t lam_key[11];
  for (size_t got = 0; got < sizeof lam_key;) {   // fresh per call, secret until the proofs are fixed (plonk_run has the reasoning)
    ssize_t k = getrandom((uint8_t*)lam_key + got, sizeof lam_key - got, 0);
    if (k <= 0) return set_err(BN254_E_HIP, "getrandom failed: no KZG batching scalars");
    got += (size_t)k;
  }
  if (!resident) {
    parallel_copy(c.h_in, proofs, pb);
    if (ib) parallel_copy(c.h_in + pb, public_inputs, ib);
  }
  auto t1_ = now();
  if (!resident) HIPCK(hipMemcpyAsync(c.d_in, c.h_in, need, hipMemcpyHostToDevice, c.stream));
  const uint8_t* d_proofs = resident ? proofs : c.d_in; const uint8_t* d_inputs = resident ? public_inputs : c.d_in + pb;
  HIPCK(hipEventRecord(c.tk[0], c.stream));
  hipError_t e = bn254_launch_plonk_stage1(d->d_key, d_proofs, proof_stride, d_inputs, n_public, m, lam_key, c.d_work, c.terms, c.flags, T1, c.stream);
  if (e != hipSuccess) return set_err(BN254_E_HIP, std::string("PlonK stage 1 launch: ") + hipGetErrorString(e));
  HIPCK(hipEventRecord(c.tk[1], c.stream));
  int mrc = plonk_msm(d, c, pvk->shape1, m, T1, true, &c.last_lanes[0], c.tk[2]);
  if (mrc) return mrc;
  HIPCK(hipEventRecord(c.tk[3], c.stream));
  // BN254_FLAG_RLC: the pairing checks of the pass batched over groups of 64 proofs -- honoured from g_plonk_rlc_min proofs per pass (below, the one remaining
  // pairing is the same latency-bound launch as the per-proof checks and nothing is gained)
  const bool rlc = (flags & BN254_FLAG_RLC) != 0 && m >= (size_t)g_plonk_rlc_min.load();
  e = bn254_launch_plonk_stage2(d->d_key, d_proofs, proof_stride, m, c.d_work, c.words, c.inf, c.terms, c.flags, c.status, TT, T2, rlc ? lam_key : nullptr, c.stream);
  HIPCK(hipEventRecord(c.tk[4], c.stream));
  if (e != hipSuccess) return set_err(BN254_E_HIP, std::string("PlonK stage 2 launch: ") + hipGetErrorString(e));
  mrc = plonk_msm(d, c, rlc ? pvk->shape2_rlc : pvk->shape2, m, TT, false, &c.last_lanes[1], c.tk[5]);
  if (mrc) return mrc;
  HIPCK(hipEventRecord(c.tk[6], c.stream));
  bool exact = !rlc;
  if (rlc) {
    // group sums (weighted points of the 64 proofs of a wavefront) -> one pairing check per group -> pending proofs of passed groups accepted; the proofs of a
    // failed group stay pending and the exact check below runs on exactly their wavefronts (every other wavefront of its kernels exits at once)
    const size_t groups = (m + 63) / 64;
    HIPCK(hipMemsetAsync(c.d_fail, 0, sizeof(uint32_t), c.stream));
    e = bn254_launch_plonk_group_sums(c.ws, c.status, m, c.grp_ws, c.grp_status, VE_LX_ELEM, BN254_ST_LINF, VE_CX_ELEM, BN254_ST_LINF2, c.stream);
    if (e == hipSuccess) e = bn254_launch_pairing2_fixed(c.grp_ws, c.grp_status, groups, d->tab0, d->tab1, d->one, BN254_ERR_PAIRING_FAILED, c.stream, c.aux, c.ev_fork, c.ev_join);
    if (e == hipSuccess) e = bn254_launch_plonk_group_scatter(c.status, m, c.grp_status, c.d_fail, c.stream);
    if (e != hipSuccess) return set_err(BN254_E_HIP, std::string("joint pairing launch: ") + hipGetErrorString(e));
    HIPCK(hipMemcpyAsync(c.h_fail, c.d_fail, sizeof(uint32_t), hipMemcpyDeviceToHost, c.stream));
    HIPCK(hipStreamSynchronize(c.stream));
    exact = *c.h_fail != 0;
  }
  if (exact) {
    e = bn254_launch_pairing2_fixed(c.ws, c.status, m, d->tab0, d->tab1, d->one, BN254_ERR_PAIRING_FAILED, c.stream, c.aux, c.ev_fork, c.ev_join);
    if (e != hipSuccess) return set_err(BN254_E_HIP, std::string("pairing launch: ") + hipGetErrorString(e));
  }
  HIPCK(hipEventRecord(c.tk[7], c.stream));
  if (resident) HIPCK(hipMemcpyAsync(status, c.status, m, hipMemcpyDeviceToDevice, c.stream));
  else HIPCK(hipMemcpyAsync(c.h_status, c.status, m, hipMemcpyDeviceToHost, c.stream));
  HIPCK(hipStreamSynchronize(c.stream));
  if (!resident) memcpy(status, c.h_status, m);
  {
    auto t4_ = now();
    // slots as bn254_plonk_last_timing names them; [0] is the host copy into pinned memory, everything else a kernel of the chain
    c.last_ms[0] = (float)ms(t0, t1_); c.last_ms[8] = (float)ms(t0, t4_);
    for (int k = 1; k <= 7; k++) HIPCK(hipEventElapsedTime(&c.last_ms[k], c.tk[k - 1], c.tk[k]));
    c.last_valid = true;
  }
  return BN254_OK;
}

int bn254_plonk_last_timing(const bn254_plonk_pvk* pvk, int device, float ms[BN254_PLONK_NUM_TIMINGS], size_t lanes[2]) {
  if (!pvk || !ms) return set_err(BN254_E_BAD_ARG, "bad argument");
  PlonkDev* d = nullptr;
  {
    std::lock_guard<std::mutex> lk(pvk->mu);
    auto it = pvk->dev.find(device);
    if (it != pvk->dev.end()) d = &it->second;
  }
  if (!d) return set_err(BN254_E_BAD_ARG, "no PlonK batch on this device yet");
  std::lock_guard<std::mutex> lk(d->pool_mu);
  if (!d->last_valid) return set_err(BN254_E_BAD_ARG, "no PlonK batch on this device yet");
  for (int i = 0; i < BN254_PLONK_NUM_TIMINGS; i++) ms[i] = d->last_ms[i];
  if (lanes) { lanes[0] = d->last_lanes[0]; lanes[1] = d->last_lanes[1]; }
  return BN254_OK;
}

}  // extern "C"

// the plan of a batch of n proofs under the current knobs: sub-batches side by side, proofs per sub-batch, proofs per pass
static void plonk_plan_for(size_t n, int* workers, size_t* per, size_t* pass_cap) {
  int max_workers = g_plonk_workers.load();
  size_t piece;
  const long big_from = g_plonk_big_from.load();
  if (big_from == 0) plonk_auto_plan(n, (size_t)g_plonk_piece.load(), (size_t)g_plonk_big_piece.load(), max_workers, &piece, &max_workers);
  else piece = n >= (size_t)big_from ? (size_t)g_plonk_big_piece.load() : (size_t)g_plonk_piece.load();
  plonk_plan(n, piece, max_workers, workers, per, pass_cap);
}
// BN254_PLONK_HOST=1: the transcripts and the Fr arithmetic on host threads (rounds 1-2) instead of the device kernels of bn254_k_plonk.hip
static bool plonk_dev_stages() { static const bool v = [] { const char* e = getenv("BN254_PLONK_HOST"); return !(e && atoi(e) != 0); }(); return v; }

// One batch.  resident = false: proofs / public_inputs / status are the caller's host buffers (each pass stages its share through the context's pinned memory);
// resident = true: they are device memory of `device` and nothing is staged.  Either way the call returns when every status byte is where the caller asked for it.
static int plonk_batch(const bn254_plonk_pvk* pvk, const uint8_t* proofs, size_t proof_stride, const uint8_t* public_inputs, size_t n_public, size_t n, uint8_t* status,
                       int device, unsigned flags, bool resident) {
  PlonkDev* d;
  int rc;
  {
    std::lock_guard<std::mutex> lk(pvk->mu);
    if ((rc = plonk_ensure_dev(pvk, device, &d))) return rc;
  }
  unsigned hw = std::thread::hardware_concurrency(); if (hw == 0) hw = 1; if (hw > 32) hw = 32;
  const bool dev_stages = plonk_dev_stages();
  if (resident && !dev_stages) return set_err(BN254_E_BAD_ARG, "BN254_PLONK_HOST=1 (the diagnostic host-thread stages) reads the proofs on the host: use the host-buffer entry");
  // Plan.  Up to `big_from` proofs the batch is cut into up to PLONK_WORKERS contiguous sub-batches, one context and one host thread each, and every sub-batch runs in
  // balanced passes of at most `piece` = 5040 proofs (a sub-batch of 6144 is two passes of 3072): up to there every launch of a pass is ONE wavefront generation and the
  // MSM launches keep their split form, and several such chains of latency-bound launches side by side fill the GPU where one chain of larger launches does not
  // (round 3: 8192 proofs 9.9 -> 8.3 ms, 16 384 15.7 -> 13.1 ms).  From `big_from` proofs the launches are large enough to be throughput-bound on their own and the
  // batch runs as few passes of up to PLONK_MAX_LAUNCH proofs (rounds 4-5; bn254_set_plonk_params has the numbers).
  int workers; size_t per, pass_cap;                                  // sub-batches, proofs per sub-batch, proofs per (equal-sized) pass of a sub-batch
  plonk_plan_for(n, &workers, &per, &pass_cap);
  PlonkLease lease(d, workers);   // waits until that many contexts are free
  for (int w = 0; w < workers; w++) if ((rc = plonk_ensure_ctx(pvk, lease.ctx(w), pass_cap, (dev_stages && !resident) ? pass_cap * (proof_stride + n_public * 32) : 0))) return rc;
  std::vector<int> rcs(workers, BN254_OK); std::vector<std::string> errs(workers);
  auto body = [&](int w) {
    const size_t lo = (size_t)w * per, hi = lo + per < n ? lo + per : n;
    for (size_t off = lo; off < hi; off += pass_cap) {
      const size_t m = hi - off < pass_cap ? hi - off : pass_cap;
      int r = dev_stages ? plonk_run_device(pvk, d, lease.ctx(w), device, proofs + off * proof_stride, proof_stride, public_inputs + off * n_public * 32, n_public, m, status + off, flags, resident)
                         : plonk_run(pvk, d, lease.ctx(w), device, proofs + off * proof_stride, proof_stride, public_inputs + off * n_public * 32, n_public, m, status + off,
                                     (hw + workers - 1) / workers);
      if (r) {
        // work of this pass may still be enqueued on the context's streams: drain them before the lease hands the context (its staging, its term and status
        // buffers) to the next call
        rcs[w] = r; errs[w] = g_err;
        (void)hipStreamSynchronize(lease.ctx(w).stream); (void)hipStreamSynchronize(lease.ctx(w).aux);
        return;
      }
    }
  };
  if (workers == 1) body(0);
  else {
    std::vector<std::thread> th;
    for (int w = 0; w < workers; w++) th.emplace_back(body, w);
    for (auto& t : th) t.join();
  }
  for (int w = 0; w < workers; w++) if (rcs[w]) return set_err(rcs[w], errs[w]);
  return BN254_OK;
}

extern "C" {

int bn254_plonk_verify_batch(const bn254_plonk_pvk* pvk, const uint8_t* proofs, size_t proof_stride, const uint8_t* public_inputs,
                             size_t n_public, size_t n, uint8_t* status, int device) {
  return bn254_plonk_verify_batch_flags(pvk, proofs, proof_stride, public_inputs, n_public, n, status, device, 0);
}
int bn254_plonk_verify_batch_flags(const bn254_plonk_pvk* pvk, const uint8_t* proofs, size_t proof_stride, const uint8_t* public_inputs,
                                   size_t n_public, size_t n, uint8_t* status, int device, unsigned flags) {
  if (!pvk || (n && (!proofs || !status)) || (n && n_public && !public_inputs)) return set_err(BN254_E_BAD_ARG, "bad argument");
  if (flags & ~(unsigned)BN254_FLAG_RLC) return set_err(BN254_E_BAD_ARG, "unknown flag (the PlonK batch entry knows BN254_FLAG_RLC)");
  if (n == 0) return BN254_OK;
  return plonk_batch(pvk, proofs, proof_stride, public_inputs, n_public, n, status, device, flags, false);
}
// proofs, public inputs and status bytes resident in the memory of `device` (what the bench times: inputs in HBM when the timed region starts)
int bn254_plonk_verify_batch_device(const bn254_plonk_pvk* pvk, const void* d_proofs, size_t proof_stride, const void* d_public_inputs, size_t n_public, size_t n,
                                    void* d_status, int device, void* hip_stream, unsigned flags) {
  if (!pvk || (n && (!d_proofs || !d_status)) || (n && n_public && !d_public_inputs)) return set_err(BN254_E_BAD_ARG, "bad argument");
  if (flags & ~(unsigned)BN254_FLAG_RLC) return set_err(BN254_E_BAD_ARG, "unknown flag (the PlonK batch entry knows BN254_FLAG_RLC)");
  if (n == 0) return BN254_OK;
  int rc = check_device(device);
  if (rc) return rc;
  // the passes run on the key's own context streams: whatever the caller's stream still has to do to the inputs comes first
  HIPCK(hipStreamSynchronize((hipStream_t)hip_stream));
  return plonk_batch(pvk, (const uint8_t*)d_proofs, proof_stride, (const uint8_t*)d_public_inputs, n_public, n, (uint8_t*)d_status, device, flags, true);
}
// several GPUs of the node: contiguous shards (bn254_shard_plan), one host thread per device through the host-buffer entry -- the PlonK twin of bn254_groth16_verify_batch_multi
int bn254_plonk_verify_batch_multi(const bn254_plonk_pvk* pvk, const uint8_t* proofs, size_t proof_stride, const uint8_t* public_inputs, size_t n_public, size_t n,
                                   uint8_t* status, uint64_t device_mask, unsigned flags) {
  if (!pvk || !device_mask || (n && (!proofs || !status)) || (n && n_public && !public_inputs)) return set_err(BN254_E_BAD_ARG, "bad argument");
  if (flags & ~(unsigned)BN254_FLAG_RLC) return set_err(BN254_E_BAD_ARG, "unknown flag (the PlonK batch entry knows BN254_FLAG_RLC)");
  int cnt = 0;
  if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0) return set_err(BN254_E_NO_DEVICE, "no HIP device available (this library has no CPU fallback)");
  int devs[64], nsh = 0; size_t los[64], cnts[64];
  int prc = bn254_shard_plan(n, device_mask, cnt, devs, los, cnts, &nsh);
  if (prc) return prc;
  if (n == 0) return BN254_OK;
  if (nsh == 1) return plonk_batch(pvk, proofs, proof_stride, public_inputs, n_public, n, status, devs[0], flags, false);
  std::vector<int> rcs((size_t)nsh, BN254_OK); std::vector<std::string> errs((size_t)nsh);
  std::vector<std::thread> th;
  for (int r = 0; r < nsh; r++) {
    th.emplace_back([&, r]() {
      if (!cnts[r]) return;
      rcs[r] = plonk_batch(pvk, proofs + los[r] * proof_stride, proof_stride, public_inputs ? public_inputs + los[r] * n_public * 32 : nullptr, n_public, cnts[r], status + los[r], devs[r], flags, false);
      if (rcs[r]) errs[r] = g_err;   // thread-local in the worker
    });
  }
  for (auto& t : th) t.join();
  for (int r = 0; r < nsh; r++) if (rcs[r]) return set_err(rcs[r], "device " + std::to_string(devs[r]) + ": " + errs[r]);
  return BN254_OK;
}
// Everything a batch of up to n proofs needs on `device`, allocated now: the key's tables, the contexts of the plan such a batch runs under (bn254_set_plonk_params) with
// their row, window-table and workspace buffers, and -- proof_stride > 0: the host-buffer entry will be used -- their pinned staging for records of that stride.  A later
// batch of that size then neither allocates nor frees (growing a context frees its old buffers, and hipFree waits for the whole device).
int bn254_plonk_reserve(const bn254_plonk_pvk* pvk, size_t n, size_t proof_stride, int device) {
  if (!pvk) return set_err(BN254_E_BAD_ARG, "null key");
  if (n == 0) n = 1;
  PlonkDev* d;
  int rc;
  {
    std::lock_guard<std::mutex> lk(pvk->mu);
    if ((rc = plonk_ensure_dev(pvk, device, &d))) return rc;
  }
  int workers; size_t per, pass_cap;
  plonk_plan_for(n, &workers, &per, &pass_cap);
  PlonkLease lease(d, workers);
  const size_t in_bytes = (plonk_dev_stages() && proof_stride) ? pass_cap * (proof_stride + (size_t)pvk->key.nb_public * 32) : 0;
  for (int w = 0; w < workers; w++) if ((rc = plonk_ensure_ctx(pvk, lease.ctx(w), pass_cap, in_bytes))) return rc;
  return BN254_OK;
}
// device memory this key holds on `device` right now: its contexts' buffers and the window tables of its points (131 MB for the reference's key); and how many contexts hold any
int bn254_plonk_footprint(const bn254_plonk_pvk* pvk, int device, size_t* bytes, int* contexts) {
  if (!pvk || !bytes) return set_err(BN254_E_BAD_ARG, "bad argument");
  *bytes = 0; if (contexts) *contexts = 0;
  PlonkDev* d = nullptr;
  {
    std::lock_guard<std::mutex> lk(pvk->mu);
    auto it = pvk->dev.find(device);
    if (it != pvk->dev.end()) d = &it->second;
  }
  if (!d) return BN254_OK;
  std::lock_guard<std::mutex> lk(d->pool_mu);
  if (d->fixed_tabs) *bytes += (pvk->fixed_pts.size() / (2 * BN_NL)) * (size_t)MSM_FW_WINDOWS * MSM_FW_ENTRIES * MSM_ENTRY_DWORDS * sizeof(int32_t);
  const int T1 = plonk_stage1_terms(pvk->key), TT = plonk_stage2_terms(pvk->key) + 2;
  const size_t tmax = (size_t)(TT > T1 ? TT : T1);
  for (const PlonkCtx& c : d->ctx) {
    if (!c.cap && !c.in_cap) continue;
    if (contexts) (*contexts)++;
    *bytes += c.in_cap + c.cap * (size_t)G16_WS_BYTES_PER_PROOF + c.part_points * 27 * sizeof(int32_t) + c.glv_lanes * (size_t)G1_GLV_TAB_BYTES_PER_LANE +
              c.cap * tmax * (sizeof(MsmTerm) + 1) + c.cap * (16 * sizeof(uint32_t) + 2) + c.cap * bn254_plonk_work_bytes();
  }
  return BN254_OK;
}

void bn254_set_plonk_params(long piece, int workers, long big_from, long big_piece) {
  if (piece >= 0) g_plonk_piece.store(piece < 256 ? 256 : (piece > PLONK_MAX_LAUNCH ? (long)PLONK_MAX_LAUNCH : piece));
  if (workers >= 0) g_plonk_workers.store(workers < 1 ? 1 : (workers > PLONK_WORKERS ? PLONK_WORKERS : workers));
  if (big_from >= 0) g_plonk_big_from.store(big_from);      // 0: the measured default plan
  if (big_piece >= 0) g_plonk_big_piece.store(big_piece < 256 ? 256 : (big_piece > PLONK_MAX_LAUNCH ? (long)PLONK_MAX_LAUNCH : big_piece));
}

int bn254_plonk_verify(const uint8_t* proof, size_t proof_len, const uint8_t* vk, size_t vk_len, const uint8_t* public_inputs,
                       size_t n_public, uint8_t* status) {
  if (!proof || !vk || !status) return set_err(BN254_E_BAD_ARG, "bad argument");
  std::shared_ptr<bn254_plonk_pvk> pvk = plonk_key_cache().find(vk, vk_len, 0);
  if (!pvk) {
    bn254_plonk_pvk* raw = nullptr;
    int rc = bn254_plonk_vk_prepare(vk, vk_len, &raw);
    if (rc == BN254_E_VK) {
      // the proof is loaded before the key (lib.rs:70 before :71): its loader error (short buffer, coordinate >= p, off the curve; plonk/converter.rs:121-178) wins
      // over the key's.  Host work for this one proof: no kernel can run without a key
      PlonkProof pr;
      const int ps = parse_plonk_proof(pr, proof, proof_len);
      *status = ps == PL_OK ? (uint8_t)BN254_ERR_MALFORMED : (uint8_t)ps;
      return BN254_OK;
    }
    if (rc) return rc;
    pvk = plonk_key_cache().insert(vk, vk_len, 0, raw);
  }
  return bn254_plonk_verify_batch(pvk.get(), proof, proof_len, public_inputs, n_public, 1, status, 0);
}

// ---------------------------------------------------------------- gnark / SP1 formats (host only: byte shuffling and one square root)
int bn254_g1_compress(const uint8_t xy[64], uint8_t out[32]) {
  if (!xy || !out) return set_err(BN254_E_BAD_ARG, "bad argument");
  if (!be_lt_p(xy) || !be_lt_p(xy + 32)) return set_err(BN254_E_BAD_ARG, "coordinate not reduced");
  G1Aff p; p.x = fp_from_be(xy); p.y = fp_from_be(xy + 32);
  enc_g1_compressed(out, p);
  return BN254_OK;
}
int bn254_g2_compress(const uint8_t xy[128], uint8_t out[64]) {
  if (!xy || !out) return set_err(BN254_E_BAD_ARG, "bad argument");
  for (int i = 0; i < 4; i++) if (!be_lt_p(xy + 32 * i)) return set_err(BN254_E_BAD_ARG, "coordinate not reduced");
  G2Aff p; p.x.c1 = fp_from_be(xy); p.x.c0 = fp_from_be(xy + 32); p.y.c1 = fp_from_be(xy + 64); p.y.c0 = fp_from_be(xy + 96);
  enc_g2_compressed(out, p);
  return BN254_OK;
}
int bn254_g1_decompress(const uint8_t in[32], uint8_t out[64], int checked, uint8_t* status) {
  if (!in || !out || !status) return set_err(BN254_E_BAD_ARG, "bad argument");
  G1Aff p;
  if (dec_g1_compressed(p, in) != DEC_OK) { *status = BN254_ERR_MALFORMED; return BN254_OK; }
  // checked (converter.rs:46-60): AffineG1::new = curve equation; G1 has cofactor 1, so there is nothing else to test
  if (checked && !g1_on_curve(p)) { *status = BN254_ERR_NOT_ON_CURVE; return BN254_OK; }
  enc_g1_uncompressed(out, p);
  *status = BN254_ACCEPT;
  return BN254_OK;
}
int bn254_g2_decompress(const uint8_t in[64], uint8_t out[128], unsigned mode, int checked, uint8_t* status) {
  if (!in || !out || !status || mode > 1) return set_err(BN254_E_BAD_ARG, "bad argument");
  G2Aff p;
  if (dec_g2_compressed(p, in, (int)mode) != DEC_OK) { *status = BN254_ERR_MALFORMED; return BN254_OK; }
  if (checked) {  // converter.rs:91-111: AffineG2::new = curve equation, then the r-torsion test
    if (!g2_on_curve(p)) { *status = BN254_ERR_NOT_ON_CURVE; return BN254_OK; }
    if (!g2_in_subgroup(p)) { *status = BN254_ERR_NOT_IN_SUBGROUP; return BN254_OK; }
  }
  enc_g2_uncompressed(out, p);
  *status = BN254_ACCEPT;
  return BN254_OK;
}
// SP1 v2.0.0 `SP1ProofWithPublicValues` as written by bincode (little-endian, u64 lengths): u32 variant (2 PlonK, 3 Groth16),
// String public_inputs[0], String public_inputs[1] (decimal), String encoded_proof (hex), String raw_proof (hex), [u8; 32]
// vkey hash, ...  (examples/script/src/main.rs:115-138 reads the same fields)
static bool sp1_string(const uint8_t* b, size_t len, size_t& off, const uint8_t** s, size_t* n) {
  if (off + 8 > len) return false;
  uint64_t k = 0; for (int i = 7; i >= 0; i--) k = k << 8 | b[off + i];
  off += 8;
  if (k > len - off) return false;
  *s = b + off; *n = (size_t)k; off += (size_t)k;
  return true;
}
static bool dec_to_be32(const uint8_t* s, size_t n, uint8_t out[32]) {
  memset(out, 0, 32);
  if (n == 0) return false;
  for (size_t i = 0; i < n; i++) {
    if (s[i] < '0' || s[i] > '9') return false;
    unsigned carry = s[i] - '0';
    for (int j = 31; j >= 0; j--) { unsigned v = out[j] * 10u + carry; out[j] = (uint8_t)v; carry = v >> 8; }
    if (carry) return false;  // more than 256 bits
  }
  return true;
}
int bn254_sp1_fixture_parse(const uint8_t* buf, size_t len, int* variant, uint8_t* raw_proof, size_t raw_cap, size_t* raw_len,
                            uint8_t public_inputs[64], uint8_t vkey_hash[32]) {
  if (!buf || !variant || !raw_proof || !raw_len || !public_inputs || !vkey_hash) return set_err(BN254_E_BAD_ARG, "bad argument");
  if (len < 4) return set_err(BN254_E_BAD_ARG, "truncated fixture");
  *variant = (int)((uint32_t)buf[0] | (uint32_t)buf[1] << 8 | (uint32_t)buf[2] << 16 | (uint32_t)buf[3] << 24);
  size_t off = 4, n0, n1, ne, nr; const uint8_t *s0, *s1, *se, *sr;
  if (!sp1_string(buf, len, off, &s0, &n0) || !sp1_string(buf, len, off, &s1, &n1) || !sp1_string(buf, len, off, &se, &ne) ||
      !sp1_string(buf, len, off, &sr, &nr) || off + 32 > len)
    return set_err(BN254_E_BAD_ARG, "truncated fixture");
  if (!dec_to_be32(s0, n0, public_inputs) || !dec_to_be32(s1, n1, public_inputs + 32)) return set_err(BN254_E_BAD_ARG, "public input is not a decimal number below 2^256");
  if (nr % 2 || nr / 2 > raw_cap) return set_err(BN254_E_BAD_ARG, "raw proof does not fit");
  for (size_t i = 0; i < nr / 2; i++) {
    int v = 0;
    for (int k = 0; k < 2; k++) {
      uint8_t c = sr[2 * i + k];
      int d = (c >= '0' && c <= '9') ? c - '0' : (c >= 'a' && c <= 'f') ? c - 'a' + 10 : (c >= 'A' && c <= 'F') ? c - 'A' + 10 : -1;
      if (d < 0) return set_err(BN254_E_BAD_ARG, "raw proof is not hexadecimal");
      v = v * 16 + d;
    }
    raw_proof[i] = (uint8_t)v;
  }
  *raw_len = nr / 2;
  memcpy(vkey_hash, buf + off, 32);
  return BN254_OK;
}

// ---------------------------------------------------------------- device-arithmetic probes (tests)
struct DevBuf {   // frees on every exit path
  uint8_t* p = nullptr;
  ~DevBuf() { if (p) (void)hipFree(p); }
};
static int run_probe(size_t in_a, size_t in_b, size_t out_sz, const uint8_t* a, const uint8_t* b, uint8_t* o, size_t n, int device,
                     hipError_t (*launch)(const uint8_t*, const uint8_t*, uint8_t*, size_t)) {
  int rc = check_device(device);
  if (rc) return rc;
  if (n == 0) return BN254_OK;
  DevBuf da, db, dout;
  HIPCK(hipMalloc((void**)&da.p, in_a * n));
  HIPCK(hipMemcpy(da.p, a, in_a * n, hipMemcpyHostToDevice));
  if (in_b && b) { HIPCK(hipMalloc((void**)&db.p, in_b * n)); HIPCK(hipMemcpy(db.p, b, in_b * n, hipMemcpyHostToDevice)); }
  HIPCK(hipMalloc((void**)&dout.p, out_sz * n));
  hipError_t e = launch(da.p, db.p, dout.p, n);
  if (e != hipSuccess) return set_err(BN254_E_HIP, std::string("probe launch: ") + hipGetErrorString(e));
  HIPCK(hipDeviceSynchronize());
  HIPCK(hipMemcpy(o, dout.p, out_sz * n, hipMemcpyDeviceToHost));
  return BN254_OK;
}
// probe (tests): stage 1 of the device path alone -- zeta (32-byte big-endian, canonical; zero where the proof failed before the challenges) and the
// stage-1 status of each proof (BN254_ACCEPT = alive, or the error code the stage decided)
int bn254_dbg_plonk_stage1(const bn254_plonk_pvk* pvk, const uint8_t* proofs, size_t proof_stride, const uint8_t* public_inputs, size_t n_public, size_t n,
                           uint8_t* zeta_out, uint8_t* status_out, int device) {
  if (!pvk || !proofs || !zeta_out || !status_out || n == 0 || n > PLONK_MAX_LAUNCH) return set_err(BN254_E_BAD_ARG, "bad argument");
  PlonkDev* d;
  int rc;
  {
    std::lock_guard<std::mutex> lk(pvk->mu);
    if ((rc = plonk_ensure_dev(pvk, device, &d))) return rc;
  }
  PlonkLease lease(d, 1);
  PlonkCtx& c = lease.ctx(0);
  if ((rc = plonk_ensure_ctx(pvk, c, n, 0))) return rc;
  const size_t pb = n * proof_stride, ib = n * n_public * 32;
  DevBuf in, zo, so;
  HIPCK(hipMalloc((void**)&in.p, pb + ib + 4)); HIPCK(hipMalloc((void**)&zo.p, 32 * n)); HIPCK(hipMalloc((void**)&so.p, n));
  HIPCK(hipMemcpy(in.p, proofs, pb, hipMemcpyHostToDevice));
  if (ib) HIPCK(hipMemcpy(in.p + pb, public_inputs, ib, hipMemcpyHostToDevice));
  uint32_t lam_key[11] = {0};
  hipError_t e = bn254_launch_plonk_stage1(d->d_key, in.p, proof_stride, in.p + pb, n_public, n, lam_key, c.d_work, c.terms, c.flags, plonk_stage1_terms(pvk->key), c.stream);
  if (e == hipSuccess) e = bn254_launch_plonk_dbg_zeta(c.d_work, n, zo.p, so.p, c.stream);
  if (e != hipSuccess) return set_err(BN254_E_HIP, std::string("probe launch: ") + hipGetErrorString(e));
  HIPCK(hipStreamSynchronize(c.stream));
  HIPCK(hipMemcpy(zeta_out, zo.p, 32 * n, hipMemcpyDeviceToHost));
  HIPCK(hipMemcpy(status_out, so.p, n, hipMemcpyDeviceToHost));
  return BN254_OK;
}

// the multiply-add issue rate of THIS device (lane-level v_mad_u64_u32 per second, sixteen independent chains per lane, four wavefronts per SIMD, best of five
// launches of ~0.25 ms): what bench.py divides its VALU rooflines by (the constant of profiles/r01_ubench_valu.txt, 35.1e12, stays as the reference)
int bn254_dbg_valu_peak(int device, double* mads_per_s) {
  if (!mads_per_s) return set_err(BN254_E_BAD_ARG, "bad argument");
  int rc = check_device(device);
  if (rc) return rc;
  *mads_per_s = bn254_measure_valu_peak(12);
  return *mads_per_s > 0 ? BN254_OK : set_err(BN254_E_HIP, "peak measurement failed");
}
int bn254_dbg_valu_peak_sustained(int device, double ms_target, double* mads_per_s) {
  if (!mads_per_s || !(ms_target > 0.0) || ms_target > 2000.0) return set_err(BN254_E_BAD_ARG, "bad argument");
  int rc = check_device(device);
  if (rc) return rc;
  *mads_per_s = bn254_measure_valu_sustained(ms_target);
  return *mads_per_s > 0 ? BN254_OK : set_err(BN254_E_HIP, "peak measurement failed");
}
int bn254_dbg_fp_mul(const uint8_t* a, const uint8_t* b, uint8_t* out, size_t n, int device) {
  return run_probe(32, 32, 32, a, b, out, n, device, [](const uint8_t* x, const uint8_t* y, uint8_t* o, size_t m) { return bn254_launch_dbg_fp_mul(x, y, o, m, nullptr); });
}
static thread_local int g_probe_op = 0;
static thread_local int32_t* g_probe_ws = nullptr;
static thread_local uint8_t* g_probe_kinds = nullptr;
static int probe_ws_alloc(size_t n, int device) {
  int rc = check_device(device);
  if (rc) return rc;
  if (n > G16_MAX_LAUNCH) return set_err(BN254_E_BAD_ARG, "probe batch too large");
  HIPCK(hipMalloc((void**)&g_probe_ws, (n ? n : 1) * (size_t)G16_WS_BYTES_PER_PROOF));
  HIPCK(hipMalloc((void**)&g_probe_kinds, n ? n : 1));  // status bytes of the probe lanes
  return BN254_OK;
}
static void probe_ws_free() { if (g_probe_ws) (void)hipFree(g_probe_ws); if (g_probe_kinds) (void)hipFree(g_probe_kinds); g_probe_ws = nullptr; g_probe_kinds = nullptr; }
int bn254_dbg_fp12_op(int op, const uint8_t* a, const uint8_t* b, uint8_t* out, size_t n, int device) {
  g_probe_op = op;
  int rc = probe_ws_alloc(n, device);
  if (rc) return rc;
  rc = run_probe(384, b ? 384 : 0, 384, a, b, out, n, device, [](const uint8_t* x, const uint8_t* y, uint8_t* o, size_t m) { return bn254_launch_dbg_fp12_op(g_probe_op, x, y, o, m, g_probe_ws, g_probe_kinds, nullptr); });
  probe_ws_free();
  return rc;
}
int bn254_dbg_pairing(const uint8_t* g1, const uint8_t* g2, uint8_t* out_gt, size_t n, int device) {
  int rc = probe_ws_alloc(n, device);
  if (rc) return rc;
  rc = run_probe(64, 128, 384, g1, g2, out_gt, n, device, [](const uint8_t* x, const uint8_t* y, uint8_t* o, size_t m) { return bn254_launch_dbg_pairing(x, y, o, m, g_probe_ws, g_probe_kinds, nullptr); });
  probe_ws_free();
  return rc;
}
int bn254_dbg_g2_subgroup_ate(const uint8_t* g1, const uint8_t* g2, uint8_t* out_flags, size_t n, int device) {
  int rc = probe_ws_alloc(n, device);
  if (rc) return rc;
  rc = run_probe(64, 128, 1, g1, g2, out_flags, n, device, [](const uint8_t* x, const uint8_t* y, uint8_t* o, size_t m) { return bn254_launch_dbg_g2_ate(x, y, o, m, g_probe_ws, g_probe_kinds, nullptr); });
  probe_ws_free();
  return rc;
}
int bn254_dbg_g2_subgroup(const uint8_t* g2, uint8_t* out_flags, size_t n, int device) {
  return run_probe(128, 0, 1, g2, nullptr, out_flags, n, device, [](const uint8_t* x, const uint8_t*, uint8_t* o, size_t m) { return bn254_launch_dbg_g2_subgroup(x, o, m, nullptr); });
}

#if defined(BN254_PLONK_MARKS)
// diagnostics build: stage 1 of ONE proof on the host, and the intermediate values it dumped (bn254_plonk.hpp::PL_DUMP): the reference the device's dump is held to
int bn254_dbg_plonk_dump_host(const bn254_plonk_pvk* pvk, const uint8_t* proof, size_t proof_len, const uint8_t* inputs, size_t n_public, uint8_t out[64 * 32], int* status) {
  if (!pvk || !proof || !out || !status) return set_err(BN254_E_BAD_ARG, "bad argument");
  PlonkWork wk; std::vector<MsmTerm> terms(plonk_stage1_terms(pvk->key)); std::vector<uint8_t> fl(terms.size());
  memset(g_plonk_dump_host, 0, sizeof g_plonk_dump_host);
  g_plonk_sha_n_host = 0;
  wk.lambda = fr_ctx().one;
  *status = plonk_stage1(pvk->key, proof, proof_len, inputs, n_public, wk, terms.data(), fl.data());
  memcpy(out, g_plonk_dump_host, 64 * 32);
  return BN254_OK;
}
int bn254_dbg_plonk_sha_dump_host(uint32_t out[32 * 24], uint32_t* n) { memcpy(out, g_plonk_sha_dump_host, sizeof g_plonk_sha_dump_host); *n = g_plonk_sha_n_host; return BN254_OK; }
#endif
// GLV decomposition probe (host only): k (32 bytes big-endian, any value: reduced mod r) -> |k1|, |k2| (16 bytes big-endian each) and signs
int bn254_dbg_glv_decompose(const uint8_t k32[32], uint8_t k1_16[16], uint8_t k2_16[16], int* neg1, int* neg2) {
  if (!k32 || !k1_16 || !k2_16 || !neg1 || !neg2) return set_err(BN254_E_BAD_ARG, "bad argument");
  const FrCtx& F = fr_ctx();
  Glv g = glv_decompose(F.to_canon(F.from_be32(k32)));
  for (int i = 0; i < 8; i++) { k1_16[i] = (uint8_t)(g.k1[1] >> (56 - 8 * i)); k1_16[8 + i] = (uint8_t)(g.k1[0] >> (56 - 8 * i)); k2_16[i] = (uint8_t)(g.k2[1] >> (56 - 8 * i)); k2_16[8 + i] = (uint8_t)(g.k2[0] >> (56 - 8 * i)); }
  *neg1 = g.neg1 ? 1 : 0; *neg2 = g.neg2 ? 1 : 0;
  return BN254_OK;
}

// host-only probe of the Fr inversion the PlonK stages use (bn254_plonk.hpp::FrCtx::inverse, binary extended GCD; which = 1: the Fermat form it replaced;
// field = 1: the same code instantiated for Fp, as the curve checks of the proof points use it).  in / out: 32-byte big-endian canonical values.
// host-only probes of the PlonK batch plan and of the scratch sizing (tests: every pass of every plan must fit the scratch of a context of its capacity)
int bn254_dbg_plonk_plan(size_t n, size_t piece, int max_workers, int* workers, size_t* per_worker, size_t* per_pass) {
  if (!workers || !per_worker || !per_pass || n == 0 || piece == 0 || max_workers < 1) return set_err(BN254_E_BAD_ARG, "bad argument");
  plonk_plan(n, piece, max_workers, workers, per_worker, per_pass);
  return BN254_OK;
}
size_t bn254_dbg_plonk_scratch_lanes(size_t capacity, int n_var) { return plonk_scratch_lanes(capacity, n_var); }
size_t bn254_dbg_plonk_part_points(size_t capacity, int n_qcp, int stage) {
  if (n_qcp < 0 || n_qcp > PLONK_MAX_QCP || stage < 1 || stage > 3) return 0;
  PlonkKey key; key.n_qcp = (uint32_t)n_qcp;
  MsmShape sh;
  if (stage == 1) plonk_msm1_shape(key, sh); else plonk_msm2_shape(key, sh, stage == 3);
  return plonk_part_points(capacity, sh);
}
// the row plan of one MSM launch of the PlonK path (stage 1: the digest; 2: the KZG check) for a key with n_qcp commitments and a batch of n proofs:
// rows, rows with a window table, scratch lanes the launch needs, the longest row in the planner's cost units, rows per sum; rows_out (optional):
// MSM_MAX_ROWS x 9 ints {variable term, pos_lo, pos_hi, unit term, sum, scratch slot, first fixed window, one past the last, joint-row term mask}
int bn254_dbg_plonk_msm_plan(int n_qcp, int stage, size_t n, size_t lane_budget, int* n_rows, int* n_var_rows, size_t* scratch_lanes, int* chain, int sum_rows[2],
                             int fixed_terms[2], int* rows_out) {
  if (n_qcp < 0 || n_qcp > PLONK_MAX_QCP || (stage != 1 && stage != 2) || n == 0 || !n_rows || !n_var_rows || !scratch_lanes || !chain || !sum_rows || !fixed_terms)
    return set_err(BN254_E_BAD_ARG, "bad argument");
  PlonkKey key; key.n_qcp = (uint32_t)n_qcp;
  MsmShape sh;
  if (stage == 1) plonk_msm1_shape(key, sh); else plonk_msm2_shape(key, sh);
  MsmPlan plan;
  if (!msm_plan_build(plan, sh, (n + 63) & ~(size_t)63, lane_budget ? lane_budget : msm_lane_budget(), 0, plonk_joint_g((n + 63) & ~(size_t)63))) return set_err(BN254_E_BAD_ARG, "shape cannot be planned");
  *n_rows = plan.n_rows; *n_var_rows = plan.n_var_rows; *scratch_lanes = bn254_g1_msm_scratch_lanes(plan, n); *chain = msm_plan_chain(plan);
  for (int k = 0; k < 2; k++) { sum_rows[k] = plan.count[k]; fixed_terms[k] = plan.n_fixed[k]; }
  if (rows_out)
    for (int r = 0; r < plan.n_rows; r++) {
      const MsmRow& w = plan.row[r];
      int* o = rows_out + 9 * r;
      o[0] = w.n_joint ? -1 : w.var_term; o[1] = w.pos_lo; o[2] = w.pos_hi; o[3] = w.unit_term; o[4] = w.sum; o[5] = w.glv_slot; o[6] = w.fw_lo; o[7] = w.fw_hi;
      o[8] = 0;                                     // a joint row: the bit mask of the terms it walks together
      for (int j = 0; j < w.n_joint; j++) o[8] |= 1 << plan.var_list[w.sum][w.var_term + j];
    }
  return BN254_OK;
}
int bn254_dbg_fr_inverse(const uint8_t in32[32], uint8_t out32[32], int which, int field) {
  if (!in32 || !out32) return set_err(BN254_E_BAD_ARG, "bad argument");
  const FrCtx& F = field ? fp64_ctx().F : fr_ctx();
  const FrM a = F.from_be_reduce(in32, 32);
  F.to_be(out32, which == 1 ? F.inverse_fermat(a) : which == 2 ? F.inverse_bgcd(a) : F.inverse(a));
  return BN254_OK;
}
// n products a_i * b_i in the field (operands: any 256-bit values, reduced and converted to Montgomery form first), through the
// product form the DEVICE stages use (form 32: eight 32-bit words) or the host's (form 64: four 64-bit limbs on __int128); out = canonical big-endian
int bn254_dbg_fr_mul(const uint8_t* a, const uint8_t* b, uint8_t* out, size_t n, int form, int field) {
  if ((!a || !b || !out) && n) return set_err(BN254_E_BAD_ARG, "bad argument");
  if (form != 32 && form != 64) return set_err(BN254_E_BAD_ARG, "form is 32 or 64");
  const FrCtx& F = field ? fp64_ctx().F : fr_ctx();
  for (size_t i = 0; i < n; i++) {
    const FrM x = F.from_be_reduce(a + 32 * i, 32), y = F.from_be_reduce(b + 32 * i, 32);
    F.to_be(out + 32 * i, form == 32 ? F.mul_w32(x, y) : F.mul_w64(x, y));
  }
  return BN254_OK;
}

// The fixed-base tables a device built for a key (bn254_k_comb.hip) against the host constructions (build_comb_table / build_window_table): the tables of the first `inputs`
// points are read back and compared entry by entry as field values.  *mismatches = entries that differ (0: identical); needs a device.
static int compare_tables(int form, const std::vector<int32_t>& pts, const int32_t* d_tab, int inputs, size_t* mismatches) {
  const size_t np = pts.size() / (2 * BN_NL), n_entries = form == 0 ? ((size_t)1 << G16_COMB_TEETH) : (size_t)32 * 255, per = n_entries * MSM_ENTRY_DWORDS;
  if ((size_t)inputs > np) inputs = (int)np;
  std::vector<int32_t> dev_tab((size_t)inputs * per), host_tab(per);
  HIPCK(hipMemcpy(dev_tab.data(), d_tab, dev_tab.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
  size_t bad = 0;
  for (int i = 0; i < inputs; i++) {
    G1Aff K; K.x = fp_from_limbs(pts.data() + (size_t)i * 2 * BN_NL); K.y = fp_from_limbs(pts.data() + (size_t)i * 2 * BN_NL + BN_NL);
    if (form == 0) build_comb_table(host_tab.data(), K); else build_window_table(host_tab.data(), K);
    for (size_t e = form == 0 ? 1 : 0; e < n_entries; e++) {
      const int32_t* a = dev_tab.data() + (size_t)i * per + e * MSM_ENTRY_DWORDS; const int32_t* b = host_tab.data() + e * MSM_ENTRY_DWORDS;
      if (!fp_eq(fp_from_limbs(a), fp_from_limbs(b)) || !fp_eq(fp_from_limbs(a + BN_NL), fp_from_limbs(b + BN_NL)) || a[18] != 0 || a[19] != 0) bad++;
    }
  }
  *mismatches = bad;
  return BN254_OK;
}
// window tables of MSM_FW_BITS bits (form 2): MSM_FW_WINDOWS x (2^MSM_FW_BITS - 1) entries per point: every window's first, middle and last entries and a pseudo-random
// sample, each against d 2^(bits w) P by double-and-add
static int compare_window_tables(const std::vector<int32_t>& pts, const int32_t* d_tab, size_t* mismatches) {
  const size_t np = pts.size() / (2 * BN_NL), per = (size_t)MSM_FW_WINDOWS * MSM_FW_ENTRIES;
  size_t bad = 0;
  uint64_t x = 0x9E3779B97F4A7C15ull;
  std::vector<int32_t> e(MSM_ENTRY_DWORDS);
  for (size_t i = 0; i < np; i++) {
    G1Aff P; P.x = fp_from_limbs(pts.data() + i * 2 * BN_NL); P.y = fp_from_limbs(pts.data() + i * 2 * BN_NL + BN_NL);
    G1Proj bw = g1_from_affine(P);
    for (int w = 0; w < MSM_FW_WINDOWS; w++) {
      std::vector<uint32_t> ds = {1, 2, 3, 255 % MSM_FW_ENTRIES + 1, 256 % MSM_FW_ENTRIES + 1, (MSM_FW_ENTRIES >> 1), (MSM_FW_ENTRIES >> 1) + 1, MSM_FW_ENTRIES - 1, MSM_FW_ENTRIES};
      for (int k = 0; k < 14; k++) { x ^= x >> 12; x ^= x << 25; x ^= x >> 27; ds.push_back(1 + (uint32_t)((x * 0x2545F4914F6CDD1Dull) >> 40) % MSM_FW_ENTRIES); }
      for (uint32_t dd : ds) {
        HIPCK(hipMemcpy(e.data(), d_tab + (i * per + (size_t)w * MSM_FW_ENTRIES + dd - 1) * MSM_ENTRY_DWORDS, MSM_ENTRY_DWORDS * sizeof(int32_t), hipMemcpyDeviceToHost));
        G1Proj acc = g1_identity();
        for (int bit = MSM_FW_BITS - 1; bit >= 0; bit--) { acc = g1_dbl(acc); if ((dd >> bit) & 1) acc = g1_add(acc, bw); }
        const G1Aff want = g1_to_affine(acc);
        if (!fp_eq(fp_from_limbs(e.data()), want.x) || !fp_eq(fp_from_limbs(e.data() + BN_NL), want.y) || e[18] != 0 || e[19] != 0) bad++;
      }
      for (int b = 0; b < MSM_FW_BITS; b++) bw = g1_dbl(bw);
    }
  }
  *mismatches = bad;
  return BN254_OK;
}
int bn254_dbg_comb_table_compare(const bn254_g16_pvk* pvk, int device, int inputs, size_t* mismatches) {
  if (!pvk || !mismatches || inputs < 1) return set_err(BN254_E_BAD_ARG, "bad argument");
  if (pvk->host.kpts.empty()) return set_err(BN254_E_BAD_ARG, "the key's tables were not built on the device");
  DevState* d = dev_state(pvk, device);
  std::lock_guard<std::mutex> lk(d->mu);
  int rc = ensure_dev(pvk, *d, device, 1);
  if (rc) return rc;
  const int form = g16_table_form(pvk->host);
  if (form == 2) return compare_window_tables(pvk->host.kpts, d->msm, mismatches);
  return compare_tables(form, pvk->host.kpts, d->msm, inputs, mismatches);
}
int bn254_dbg_plonk_table_compare(const bn254_plonk_pvk* pvk, int device, size_t* mismatches) {
  if (!pvk || !mismatches) return set_err(BN254_E_BAD_ARG, "bad argument");
  PlonkDev* d;
  std::lock_guard<std::mutex> lk(pvk->mu);
  int rc = plonk_ensure_dev(pvk, device, &d);
  if (rc) return rc;
  return compare_window_tables(pvk->fixed_pts, d->fixed_tabs, mismatches);
}
// host-only probe of the comb tables of keys with many public inputs: x * P from build_comb_table(P) and the column digits the kernels use
int bn254_dbg_comb_mul(const uint8_t p64[64], const uint8_t x32[32], uint8_t out64[64]) {
  if (!p64 || !x32 || !out64) return set_err(BN254_E_BAD_ARG, "bad argument");
  G1Aff P; P.x = fp_from_be(p64); P.y = fp_from_be(p64 + 32);
  if (!g1_on_curve(P)) return set_err(BN254_E_BAD_ARG, "not a curve point");
  std::vector<int32_t> tab(((size_t)1 << G16_COMB_TEETH) * MSM_ENTRY_DWORDS);
  build_comb_table(tab.data(), P);
  uint32_t w[8];
  for (int k = 0; k < 8; k++) { const uint8_t* q = x32 + 28 - 4 * k; w[k] = (uint32_t)q[0] << 24 | (uint32_t)q[1] << 16 | (uint32_t)q[2] << 8 | (uint32_t)q[3]; }
  G1Proj acc = g1_identity();
  for (int col = G16_COMB_COLS - 1; col >= 0; col--) {
    acc = g1_dbl(acc);
    const uint32_t idx = g16_comb_digit(w, col);
    if (idx) {
      G1Aff e; e.x = fp_from_limbs(tab.data() + (size_t)idx * MSM_ENTRY_DWORDS); e.y = fp_from_limbs(tab.data() + (size_t)idx * MSM_ENTRY_DWORDS + BN_NL);
      acc = g1_add_mixed(acc, e);
    }
  }
  if (g1_is_identity(acc)) { memset(out64, 0, 64); return BN254_OK; }
  enc_g1_uncompressed(out64, g1_to_affine(acc));
  return BN254_OK;
}

// ---------------------------------------------------------------- synthetic workload generator
size_t bn254_synth_groth16_vk_len(size_t n_public) { return 292 + 32 * (n_public + 1) + 4 + 128; }

int bn254_synth_groth16(uint64_t seed, size_t n_public, size_t n, int invalid_every, int agree, int threads, uint8_t* vk_out,
                        uint8_t* proofs_out, uint8_t* inputs_out, uint8_t* expected) {
  return bn254_synth_groth16_range(seed, n_public, 0, n, invalid_every, agree, threads, vk_out, proofs_out, inputs_out, expected);
}
// proofs [first, first + n) of the stream bn254_synth_groth16 generates for `seed` (proof i is a function of (seed, i) alone), written to
// positions 0 .. n-1 of the output buffers: a rank of a sharded job generates its own contiguous shard only
int bn254_synth_groth16_range(uint64_t seed, size_t n_public, size_t first, size_t n, int invalid_every, int agree, int threads, uint8_t* vk_out,
                              uint8_t* proofs_out, uint8_t* inputs_out, uint8_t* expected) {
  if (!vk_out || (n && (!proofs_out || !expected)) || (n && n_public && !inputs_out)) return set_err(BN254_E_BAD_ARG, "bad argument");
  static GenTables* tabs = nullptr;
  static std::mutex tmu;
  {
    std::lock_guard<std::mutex> lk(tmu);
    if (!tabs) { tabs = new GenTables(); build_gen_tables(*tabs); }
  }
  SplitMix64 rng{seed};
  // trapdoors; beta, gamma, delta rejection-sampled into the mode-agreement set when asked:
  // y(beta G2), y(gamma G2) must have c0 / c1 in DIFFERENT halves of [0,p), y(delta G2) in the SAME half
  auto same_half = [](const G2Aff& q) { return fp_is_large(q.y.c0) == fp_is_large(q.y.c1); };
  U256 alpha = fr_random(rng, true), beta, gamma, delta;
  G2Aff beta2, gamma2, delta2;
  const bool agree_modes = (agree & 1) != 0;     // agree bit 1 (value 2): every proof with index = 3 mod 7 has L = the identity (see the worker)
  for (;;) { beta = fr_random(rng, true); beta2 = g2_mul_gen(*tabs, beta); if (!agree_modes || !same_half(beta2)) break; }
  for (;;) { gamma = fr_random(rng, true); gamma2 = g2_mul_gen(*tabs, gamma); if (!agree_modes || !same_half(gamma2)) break; }
  for (;;) { delta = fr_random(rng, true); delta2 = g2_mul_gen(*tabs, delta); if (!agree_modes || same_half(delta2)) break; }
  std::vector<U256> kk(n_public + 1);
  for (auto& k : kk) k = fr_random(rng, true);
  // gnark-compressed vk: alpha1 | beta1 | beta2 | gamma2 | delta1 | delta2 | nK | K.. | 0 (no commitments) | 2 x G2 infinity
  memset(vk_out, 0, bn254_synth_groth16_vk_len(n_public));
  enc_g1_compressed(vk_out, g1_to_affine(g1_mul_gen(*tabs, alpha)));
  enc_g1_compressed(vk_out + 32, g1_to_affine(g1_mul_gen(*tabs, beta)));
  enc_g2_compressed(vk_out + 64, beta2);
  enc_g2_compressed(vk_out + 128, gamma2);
  enc_g1_compressed(vk_out + 192, g1_to_affine(g1_mul_gen(*tabs, delta)));
  enc_g2_compressed(vk_out + 224, delta2);
  uint32_t nk = (uint32_t)(n_public + 1);
  vk_out[288] = (uint8_t)(nk >> 24); vk_out[289] = (uint8_t)(nk >> 16); vk_out[290] = (uint8_t)(nk >> 8); vk_out[291] = (uint8_t)nk;
  for (size_t i = 0; i <= n_public; i++) enc_g1_compressed(vk_out + 292 + 32 * i, g1_to_affine(g1_mul_gen(*tabs, kk[i])));
  size_t off = 292 + 32 * (n_public + 1) + 4;
  vk_out[off] = 0x40; vk_out[off + 64] = 0x40;
  if (n == 0) return BN254_OK;
  U256 delta_inv = fr_inv(delta), alpha_beta = fr_mul(alpha, beta);
  // a few twist points outside the r-torsion for the NOT_IN_SUBGROUP class
  std::vector<G2Aff> bad_b;
  if (invalid_every > 0) {
    SplitMix64 r2{seed ^ 0xabcdef1234567ull};
    while (bad_b.size() < 4) {
      G2Aff q; U256 t0 = fr_random(r2, false), t1 = fr_random(r2, false);
      uint8_t b0[32], b1[32]; u256_to_be(b0, t0); u256_to_be(b1, t1);
      q.x.c0 = fp_from_be(b0); q.x.c1 = fp_from_be(b1);
      if (!fp2_sqrt(q.y, fp2_add(fp2_mul(fp2_sqr(q.x), q.x), g2_twist_b()))) continue;
      if (g2_in_subgroup(q)) continue;  // probability ~ 1/cofactor
      bad_b.push_back(q);
    }
  }
  if (threads <= 0) { threads = (int)std::thread::hardware_concurrency(); if (threads <= 0) threads = 1; }
  if ((size_t)threads > n) threads = (int)n;
  G1Aff g1gen; g1gen.x = fp_one(); g1gen.y = fp_add(fp_one(), fp_one());
  auto worker = [&](int tid) {
    for (size_t li = tid; li < n; li += threads) {
      const size_t i = first + li;   // global index: seeds the proof and selects its class
      SplitMix64 r{seed * 0x9e3779b97f4a7c15ull + 0x1000 + i};
      U256 a = fr_random(r, true), b = fr_random(r, true);
      U256 ell = kk[0];
      std::vector<U256> xs(n_public);
      for (size_t s = 0; s < n_public; s++) { xs[s] = fr_random(r, false); ell = fr_add(ell, fr_mul(xs[s], kk[s + 1])); }
      if ((agree & 2) && n_public > 0 && i % 7 == 3) {
        // L = K0 + sum x_s K_s = the identity: the last input cancels the rest (valid proofs whose public-input point is the point at infinity --
        // bn::pairing_batch skips such a pair; the kernels replace its line by 1)
        const size_t last = n_public - 1;
        ell = fr_sub(ell, fr_mul(xs[last], kk[last + 1]));
        U256 zero = {{0, 0, 0, 0}};
        xs[last] = fr_mul(fr_sub(zero, ell), fr_inv(kk[last + 1]));
        ell = zero;
      }
      // c = (a b - alpha beta - gamma ell) / delta   =>   e(A,B) = e(alpha,beta) e(L,gamma) e(C,delta)
      U256 c = fr_mul(fr_sub(fr_sub(fr_mul(a, b), alpha_beta), fr_mul(gamma, ell)), delta_inv);
      G1Aff A = g1_to_affine(g1_mul_gen(*tabs, a));
      G2Aff B = g2_mul_gen(*tabs, b);
      G1Proj Cp = g1_mul_gen(*tabs, c);
      uint8_t st = BN254_ACCEPT;
      int cls = -1;
      if (invalid_every > 0 && (i % (size_t)invalid_every) == (size_t)invalid_every - 1) cls = (int)((i / (size_t)invalid_every) % 5);
      if (cls == 1) { Cp = g1_add_mixed(Cp, g1gen); st = BN254_REJECT; }
      if (cls == 3) { B = bad_b[(i / (size_t)invalid_every) % bad_b.size()]; st = BN254_ERR_NOT_IN_SUBGROUP; }
      G1Aff C = g1_is_identity(Cp) ? g1gen : g1_to_affine(Cp);
      uint8_t* p = proofs_out + 256 * li;
      enc_g1_uncompressed(p, A); enc_g2_uncompressed(p + 64, B); enc_g1_uncompressed(p + 192, C);
      for (size_t s = 0; s < n_public; s++) u256_to_be(inputs_out + (li * n_public + s) * 32, xs[s]);
      if (cls == 0 && n_public > 0) {  // x_0 + 1 (as raw integer; stays below 2^256)
        U256 one = {{1, 0, 0, 0}}, t; u256_add(t, xs[0], one); u256_to_be(inputs_out + li * n_public * 32, t); st = BN254_REJECT;
      }
      if (cls == 2) {  // A.y + 1 mod p: off the curve (y+1 = -y only for y = (p-1)/2)
        Fp y1 = fp_add(A.y, fp_one()); fp_to_be(p + 32, y1); st = BN254_ERR_NOT_ON_CURVE;
      }
      if (cls == 4) { memset(p, 0xff, 32); st = BN254_ERR_NOT_MEMBER; }  // A.x = 2^256 - 1 >= p
      expected[li] = st;
    }
  };
  std::vector<std::thread> th;
  for (int t = 0; t < threads; t++) th.emplace_back(worker, t);
  for (auto& x : th) x.join();
  return BN254_OK;
}

}  // extern "C"
