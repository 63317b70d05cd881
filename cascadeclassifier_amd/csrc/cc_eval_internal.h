// Shared between cc_eval.hip (evaluator, bulk operator(), predict) and cc_split.hip (best-split search): device-side
// feature records, the evaluator handle and the helpers both translation units use.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <mutex>
#include <vector>

#include "cc_internal.h"

namespace ccamd {

#define CC_HIP(expr)                                                                                         \
  do {                                                                                                       \
    hipError_t e_ = (expr);                                                                                  \
    if (e_ != hipSuccess) return set_error(CC_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                                           __FILE__, __LINE__);                                              \
  } while (0)

// Haar feature with the reference's fastRect offsets (row stride W+1) — haarfeatures.cpp:266-309.
struct HaarFeatDev {
  int p[3][4];
  float w[3];
  int tilted;
};
struct LbpFeatDev {
  int p[16];
};

template <class T>
struct EBuf {
  T* p = nullptr;
  size_t n = 0;
  ~EBuf() {
    if (p) (void)hipFree(p);
  }
  hipError_t ensure(size_t count) {
    if (count <= n) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&p), count * sizeof(T));
    if (e == hipSuccess) n = count;
    return e;
  }
};


// A blocking copy that stays off the legacy stream. Plain hipMemcpy / hipMemset are refused while ANY thread of the process
// captures a hipGraph (the detector's single-image path captures one per scale plan) and fail that thread's capture with them;
// stream-ordered copies on a non-blocking stream are not.
inline hipError_t copy_sync(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t st) {
  const hipError_t e = hipMemcpyAsync(dst, src, bytes, kind, st);
  return e != hipSuccess ? e : hipStreamSynchronize(st);
}
struct OwnStream {  // for entry points that have no evaluator to borrow a stream from
  hipStream_t s = nullptr;
  hipError_t create() { return hipStreamCreateWithFlags(&s, hipStreamNonBlocking); }
  ~OwnStream() {
    if (s) (void)hipStreamDestroy(s);
  }
};

struct PinnedBuf {
  void* p = nullptr;
  size_t bytes = 0;
  ~PinnedBuf() {
    if (p) (void)hipHostFree(p);
  }
  hipError_t ensure(size_t b) {
    if (b <= bytes) return hipSuccess;
    if (p) (void)hipHostFree(p);
    p = nullptr;
    bytes = 0;
    hipError_t e = hipHostMalloc(&p, b, hipHostMallocDefault);
    if (e == hipSuccess) bytes = b;
    return e;
  }
};

cc_status eval_device(cc_evaluator* e);
// Launches k_eval_batch over `feats` [fb, fe) for ns samples into d_out_ptr (device). Caller holds e->mu.
// Stable sort of every row of [rows][n] with the sample position as the value, one block per row (cc_split.hip); only for
// n <= sort_rows_block_limit(): larger rows take the device-wide segmented sort at the call site.
int sort_rows_block_limit();
hipError_t sort_rows_block(const float* vals, int rows, int n, float* keys_out, int* idx_out, hipStream_t st);

cc_status launch_batch(cc_evaluator* e, bool haar, const void* feats, int fb, int fe, const int32_t* d_idx, int ns,
                       float* d_out_ptr, int normalized, size_t out_pitch /* 0 = n_samples */);
// Sends the images queued by cc_eval_set_image to the device (runs of consecutive sample indices, one launch per run).
// Every entry point that reads stored samples on the device calls it first. Caller holds e->mu.
cc_status flush_pending_images(cc_evaluator* e);

}  // namespace ccamd

struct cc_evaluator {
  using HaarFeature = ccamd::HaarFeature;
  template <class T>
  using EBuf = ccamd::EBuf<T>;
  using HaarFeatDev = ccamd::HaarFeatDev;
  using LbpFeatDev = ccamd::LbpFeatDev;
  int type = 0, mode = 0, W = 0, H = 0, max_samples = 0, device = 0, cols = 0;
  bool use_tilted = false;
  std::vector<HaarFeature> haar;
  std::vector<int32_t> lbp;
  std::vector<float> cls;
  int nfeat = 0;
  hipStream_t stream = nullptr;
  EBuf<int32_t> d_sum, d_tilted;
  EBuf<float> d_nf;
  EBuf<HaarFeatDev> d_haar;
  EBuf<LbpFeatDev> d_lbp;
  // scratch (guarded by mu: the calc entry points may be called concurrently)
  std::mutex mu;
  EBuf<uint8_t> d_imgs;
  EBuf<int32_t> d_idx;
  EBuf<float> d_out;
  EBuf<HaarFeatDev> d_custom;
  EBuf<HaarFeatDev> d_haar_plain;  // catalog with plain fastRect offsets (cc_eval_calc_list), built on first use
  EBuf<LbpFeatDev> d_lbp_plain;
  EBuf<uint8_t> d_pred;
  hipEvent_t ev_a = nullptr, ev_b = nullptr;
  hipEvent_t ev_piece[8] = {};  // categorical split search: one per piece of the sums on its way back (cc_split.hip)
  double last_ms = 0;
  int S = 16;
  // resident tables of the split search (cc_eval_presort): per group of 64 features the sorted values and sample
  // indices, interleaved so that lane = feature reads are coalesced: [group][rank][64]
  EBuf<float> d_sorted_val;
  EBuf<uint16_t> d_sorted_idx16;
  EBuf<int32_t> d_sorted_idx32;
  EBuf<uint8_t> d_codes;  // LBP: [feature][sample] codes
  int cat_sorted_n = 0;          // samples per variable in d_cat_sorted (0: not built)
  EBuf<uint32_t> d_cat_sorted;  // LBP: (sample << 8 | code) in (code, sample) order, [group][rank][64] (cc_split.hip)
  int presort_n = 0;      // samples covered by the tables (0 = none)
  int presort_f0 = 0, presort_f1 = 0;  // variables covered by the tables
  EBuf<double> d_split_tab, d_split_out;
  EBuf<int32_t> d_split_idx;
  ccamd::PinnedBuf pin_in, pin_out;
  // ---- single-image path (cc_eval_set_image / cc_eval_calc / cc_eval_calc_list), see cc_eval.hip ----
  // images set one at a time that the device has not seen yet: pixels (W * H each), sample index, and where a sample's
  // latest image sits in the queue (pend_slot[idx], -1 = not queued); guarded by mu
  std::vector<uint8_t> pend_px;
  std::vector<int32_t> pend_idx;
  std::vector<int32_t> pend_slot;
  std::atomic<int> pend_n{0};
  // host mirror of the sample set LAST by cc_eval_set_image: its integral(s) and norm factor, computed on the host with the
  // device kernels' arithmetic; answers cc_eval_calc / cc_eval_calc_list for that sample without a launch
  int mirror_idx = -1;
  std::vector<int32_t> mirror_sum, mirror_tilted;
  float mirror_nf = 0.f;
  std::once_flag host_catalog_once;
  std::vector<HaarFeatDev> h_haar;  // catalog with plain fastRect offsets (row stride W + 1)
  std::vector<LbpFeatDev> h_lbp;
  ~cc_evaluator() {
    if (ev_a) (void)hipEventDestroy(ev_a);
    if (ev_b) (void)hipEventDestroy(ev_b);
    for (hipEvent_t& ev : ev_piece)
      if (ev) (void)hipEventDestroy(ev);
    if (stream) (void)hipStreamDestroy(stream);
  }
};

