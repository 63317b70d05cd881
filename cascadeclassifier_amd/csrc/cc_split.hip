// Best-split search of a boosted-tree node on gfx950 (section 6 of the C ABI).
// Replaces CvDTree::find_best_split (traincascade/lib/src/o_cvdtree.cpp:313-357) with the per-variable searches of
// CvBoostTree (o_cvboostree.cpp:151-516) over the variable data of CvCascadeBoostTrainData
// (o_cvcascadeboosttraindata.cpp:403-482).
//
// MI355X-first shape: the reference keeps a budgeted slice of sorted indices in host memory and re-evaluates + re-sorts
// every other variable for every node; here the sorted order of EVERY variable stays resident in HBM for the whole
// stage (6 B per (variable, sample): 19.5 GB for 162 336 x 20 000), and one node search is a single streaming pass.
//
// Data layout in HBM (cc_eval_presort): variables in groups of 64; per group the sorted values (f32) and sample indices
// (u16 when n_samples <= 65 536, else i32) are interleaved [group][rank][64 lanes], so that lane = variable reads of
// one rank are one coalesced 256-B / 128-B segment. LBP: category codes u8 [variable][sample].
//
// The running sums of the reference are sequential double additions in sorted order; they are reproduced bit for bit by
// giving each variable to one thread. Parallelism comes from the 10^5 variables, not from the scan.
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <atomic>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <thread>

#include "cc_eval_internal.h"

namespace ccamd {

// per stored sample: weight and (regression) response * weight or (classification) class; w < 0 marks "not in the node"
struct SplitEntry {
  double w, t;
};

// ------------------------------------------------------------------------------------------------
// [rows][n] row-major (one sorted variable per row) -> [group][rank][64]
// ------------------------------------------------------------------------------------------------
template <class TI>
__global__ __launch_bounds__(256) void k_interleave(const float* __restrict__ vals, const int* __restrict__ idx, int rows, int n,
                                                    float* __restrict__ out_val, TI* __restrict__ out_idx, size_t group0) {
  __shared__ float tv[64][65];
  __shared__ int ti[64][65];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r0 = blockIdx.x * 64, g = blockIdx.y;
  for (int j = wave; j < 64; j += 4) {
    const int row = g * 64 + j;
    float v = 0.f;
    int s = 0;
    if (row < rows && r0 + lane < n) {
      v = vals[(size_t)row * n + r0 + lane];
      s = idx[(size_t)row * n + r0 + lane];
    }
    tv[j][lane] = v;
    ti[j][lane] = s;
  }
  __syncthreads();
  for (int rr = wave; rr < 64; rr += 4) {
    if (r0 + rr >= n) break;
    const size_t o = ((group0 + g) * (size_t)n + r0 + rr) * 64 + lane;
    out_val[o] = tv[lane][rr];
    out_idx[o] = (TI)ti[lane][rr];
  }
}

// ------------------------------------------------------------------------------------------------
// Stable sort of every row of [rows][n] (one variable per row) with the sample position as the value: one block of 1024
// threads per row, the whole row in registers + LDS (hipcub::BlockRadixSort: LSD radix, stable -- equal values keep
// increasing sample order, what std::stable_sort over (value, index) gives the reference,
// o_cvcascadeboosttraindata.cpp:582-596). A row of a boosting stage (<= 24 576 samples) fits a CU's LDS, so it is read
// once and written once: 8 B + 8 B per element instead of the ~70 B of a device-wide segmented radix sort's four passes.
// ------------------------------------------------------------------------------------------------
constexpr int SORT_THREADS = 1024;
template <int ITEMS>
struct RowSort {
  using Sort = hipcub::BlockRadixSort<float, SORT_THREADS, ITEMS, unsigned short>;
  static constexpr size_t lds_bytes = sizeof(typename Sort::TempStorage);
};
template <int ITEMS>
__global__ __launch_bounds__(SORT_THREADS) void k_sort_rows_block(const float* __restrict__ vals, int rows, int n, float* __restrict__ keys_out,
                                                                  int* __restrict__ idx_out) {
  extern __shared__ __attribute__((aligned(16))) char sort_lds[];
  using Sort = typename RowSort<ITEMS>::Sort;
  typename Sort::TempStorage& temp = *reinterpret_cast<typename Sort::TempStorage*>(sort_lds);
  const int row = blockIdx.x;
  if (row >= rows) return;
  const float* src = vals + (size_t)row * n;
  float key[ITEMS];
  unsigned short val[ITEMS];
#pragma unroll
  for (int i = 0; i < ITEMS; i++) {  // blocked arrangement: thread t owns positions t * ITEMS ..
    const int pos = (int)threadIdx.x * ITEMS + i;
    // padding = +inf: sorts behind every finite value. Precondition: no NaN among the values (a NaN's radix key sorts behind
    // +inf and would push padding into the first n ranks). The evaluators cannot produce one: Haar values are finite sums
    // divided by a positive norm factor or 0 when it is 0 (haarfeatures.h:108-112), LBP codes are 0..255.
    key[i] = pos < n ? src[pos] : __int_as_float(0x7f800000);
    val[i] = (unsigned short)pos;
  }
  Sort(temp).SortBlockedToStriped(key, val);
#pragma unroll
  for (int i = 0; i < ITEMS; i++) {  // striped: rank = i * 1024 + t, consecutive lanes write consecutive ranks
    const int rank = i * SORT_THREADS + (int)threadIdx.x;
    if (rank < n) {
      keys_out[(size_t)row * n + rank] = key[i];
      idx_out[(size_t)row * n + rank] = (int)val[i];
    }
  }
}
template <int ITEMS>
static hipError_t launch_sort_rows_block(const float* vals, int rows, int n, float* keys_out, int* idx_out, hipStream_t st) {
  constexpr size_t lds = RowSort<ITEMS>::lds_bytes;
  if (lds > 64 * 1024) {
    // The opt-in is per device and cheap: set it on every launch (a process-wide "done" flag, as round 3 kept, leaves the
    // second device of a process without it, and is not thread-safe).
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sort_rows_block<ITEMS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL((k_sort_rows_block<ITEMS>), dim3((unsigned)rows), dim3(SORT_THREADS), lds, st, vals, rows, n, keys_out, idx_out);
  return hipGetLastError();
}

int sort_rows_block_limit() {
  static const bool device_sort_only = std::getenv("CCAMD_PRESORT_DEVICE_SORT") != nullptr;  // A/B: the device-wide sort for every size
  return device_sort_only ? 0 : SORT_THREADS * 24;
}
hipError_t sort_rows_block(const float* vals, int rows, int n, float* keys_out, int* idx_out, hipStream_t st) {
  if (n <= SORT_THREADS * 4) return launch_sort_rows_block<4>(vals, rows, n, keys_out, idx_out, st);
  if (n <= SORT_THREADS * 8) return launch_sort_rows_block<8>(vals, rows, n, keys_out, idx_out, st);
  if (n <= SORT_THREADS * 12) return launch_sort_rows_block<12>(vals, rows, n, keys_out, idx_out, st);
  if (n <= SORT_THREADS * 16) return launch_sort_rows_block<16>(vals, rows, n, keys_out, idx_out, st);
  if (n <= SORT_THREADS * 20) return launch_sort_rows_block<20>(vals, rows, n, keys_out, idx_out, st);
  return launch_sort_rows_block<24>(vals, rows, n, keys_out, idx_out, st);
}

__global__ void k_codes_u8(const float* __restrict__ in, uint8_t* __restrict__ out, size_t total) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < total) out[i] = (uint8_t)(int)in[i];
}

__global__ void k_iota_rows2(int* __restrict__ v, size_t total, int n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < total) v[i] = (int)(i % (size_t)n);
}

// ------------------------------------------------------------------------------------------------
// Ordered variables: one thread per variable walks its sorted samples.
//   MODE 0: find_split_ord_reg   (o_cvboostree.cpp:361-426)
//   MODE 1: find_split_ord_class, GINI      (o_cvboostree.cpp:192-221)
//   MODE 2: find_split_ord_class, MISCLASS  (o_cvboostree.cpp:222-238)
// The reference tests the boundary between sorted positions i and i+1 after adding sample i; here the test happens when
// the NEXT node member arrives (samples outside the node are skipped), which is the same sequence of operations.
// ------------------------------------------------------------------------------------------------
struct SplitOrdArgs {
  const float* sv;
  const void* si;
  const SplitEntry* tab;
  int n_pre;       // samples per variable in the tables
  int n_vars;
  double w_total0, w_total1;  // weights[n], weights[n + 1]
  double rsum0;               // node_value * weights[n]
  double* best_val;
  int* best_i;
  float* best_vl;
  float* best_vr;
  int n_groups;
  int dbg_nogather;  // timing experiment (CCAMD_DEBUG_SPLIT_NOGATHER): every lane reads table entry `lane`
};

// TAB selects where the per-sample table lives: 0 = global memory (16-B entries; any size), 1 = LDS, 16-B entries,
// 2 = LDS, 8-B entries (regression with responses +-1: entry = response * w, w = |entry|; classification: entry = w with
// the class in the sign bit; NaN = not in the node). A wavefront's 64 lanes gather 64 unrelated entries per rank, which
// costs ~64 L2 requests from global memory but a few LDS cycles from a block-resident copy.
#ifndef CC_SPLIT_UNROLL
#define CC_SPLIT_UNROLL 16  // ranks whose loads are in flight per thread before the sequential part consumes them
#endif
template <int MODE, class TI, int TAB>
__global__ __launch_bounds__(1024) void k_split_ord(SplitOrdArgs A) {
  extern __shared__ double l_tab[];
  const int lane = threadIdx.x & 63;
  const int group = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (TAB != 0) {
    const int words = A.n_pre * (TAB == 1 ? 2 : 1);
    const double* src = reinterpret_cast<const double*>(A.tab);
    for (int i = threadIdx.x; i < words; i += blockDim.x) l_tab[i] = src[i];
    __syncthreads();
  }
  if (group >= A.n_groups) return;
  const int f = group * 64 + lane;
  const size_t base = (size_t)group * A.n_pre * 64 + lane;
  const float* sv = A.sv + base;
  const TI* si = reinterpret_cast<const TI*>(A.si) + base;
  const float epsilon = FLT_EPSILON * 2;
  double L = 0, R, lsum = 0, rsum = 0, lcw0 = 0, lcw1 = 0, rcw0 = A.w_total0, rcw1 = A.w_total1;
  if (MODE == 0) {
    R = A.w_total0;
    rsum = A.rsum0;
  } else {
    R = rcw0 + rcw1;
    rsum = rcw0 * rcw0 + rcw1 * rcw1;  // rsum2; lsum plays lsum2
  }
  double best_val = -1.0;
  int best_i = -1, count = 0;
  float prev = 0.f, vl = 0.f, vr = 0.f;
  constexpr int U = CC_SPLIT_UNROLL;
  for (int r0 = 0; r0 < A.n_pre; r0 += U) {
    float v[U];
    SplitEntry e[U];
    unsigned s[U];
#pragma unroll
    for (int k = 0; k < U; k++) {
      const bool in = r0 + k < A.n_pre;
      v[k] = in ? sv[(size_t)(r0 + k) * 64] : 0.f;
      s[k] = in ? (unsigned)si[(size_t)(r0 + k) * 64] : 0u;
    }
#pragma unroll
    for (int k = 0; k < U; k++) {
      const unsigned g = A.dbg_nogather ? (unsigned)lane : s[k];
      if (TAB == 0)
        e[k] = A.tab[g];
      else if (TAB == 1)
        e[k] = reinterpret_cast<const SplitEntry*>(l_tab)[g];
      else {
        const double x = l_tab[g];
        e[k].w = fabs(x);  // NaN stays NaN: fails the membership test below
        e[k].t = MODE == 0 ? x : (__double_as_longlong(x) < 0 ? 1.0 : 0.0);
      }
      if (r0 + k >= A.n_pre) e[k].w = -1.0;
    }
#pragma unroll
    for (int k = 0; k < U; k++) {
      const double w = e[k].w;
      if (!(w >= 0.0)) continue;
      if (count > 0 && prev + epsilon < v[k]) {
        double val;
        bool candidate = true;
        if (MODE == 2) {
          const double a = lcw0 + rcw1, b = lcw1 + rcw0;
          val = a > b ? a : b;
        } else {
          const double num = MODE == 0 ? lsum * lsum * R + rsum * rsum * L : lsum * R + rsum * L;
          const double den = L * R;
          // The quotient can only matter if it exceeds best_val. num < best_val * den * (1 - 2^-50) (two roundings,
          // each within 2^-53) implies num / den < best_val exactly, hence fl(num / den) <= best_val: the division
          // is skipped without changing any result. Anything else (incl. den <= 0, NaN) takes the division.
          candidate = !(den > 0.0 && num < best_val * den * (1.0 - 0x1p-50));
          val = candidate ? num / den : 0.0;
        }
        if (candidate && best_val < val) {
          best_val = val;
          best_i = count - 1;
          vl = prev;
          vr = v[k];
        }
      }
      if (MODE == 0) {
        const double t = e[k].t;
        L += w;
        R -= w;
        lsum += t;
        rsum -= t;
      } else {
        const bool c1 = e[k].t != 0.0;
        if (MODE == 1) {
          const double w2 = w * w;
          L += w;
          R -= w;
          const double lv = c1 ? lcw1 : lcw0, rv = c1 ? rcw1 : rcw0;
          lsum += 2 * lv * w + w2;
          rsum -= 2 * rv * w - w2;
          if (c1) {
            lcw1 = lv + w;
            rcw1 = rv - w;
          } else {
            lcw0 = lv + w;
            rcw0 = rv - w;
          }
        } else {
          if (c1) {
            lcw1 += w;
            rcw1 -= w;
          } else {
            lcw0 += w;
            rcw0 -= w;
          }
        }
      }
      prev = v[k];
      count++;
    }
  }
  if (f < A.n_vars) {
    A.best_val[f] = best_val;
    A.best_i[f] = best_i;
    A.best_vl[f] = vl;
    A.best_vr[f] = vr;
  }
}

// ------------------------------------------------------------------------------------------------
// The same search, re-shaped in round 4 for the case the LDS table of 8-byte entries covers (TAB 2 above: +-1 responses or
// class labels, <= 20 480 samples -- every node of a cascade stage): straight-line code with ONE rare branch per rank.
// Counters of k_split_ord at configs[4] (profiles/r04_training_kernels.txt): the block-per-CU launch gives a SIMD 2.5
// wavefronts, each of which walks 20 000 ranks through four divergent branches per rank (membership, value change,
// candidate, record) -- 6 s_cbranch per rank, 45 vector instructions, one issued every ~6.5 cycles: neither memory (34 %
// of HBM) nor the LDS (10 % busy) limits it, the branch bubbles and the instruction count of a lone wavefront do.
// Here
//  * a sample outside the node adds +0.0 to every running sum (exact: the sums start at +0 / are only ever decreased by
//    +0) and is kept out of `count`, `prev` and the value-change test by its mask bit: no membership branch;
//  * the quality's numerator / denominator and the "cannot beat the record" test are computed for every rank; only a
//    lane that may set a record enters the one branch, which holds the division and the record update;
//  * the record's threshold best * (1 - 2^-50) is kept beside the record instead of being recomputed per rank (one
//    multiplication less; the test stays a sufficient condition: num < fl(bt * den), bt = fl(best (1 - 2^-50)), den > 0,
//    best > 0 imply num / den < best, hence fl(num / den) <= best);
//  * the loads of the next 16 ranks are issued before the current 16 are consumed.
// Operation order of every sum and of the quality expression is the reference's (o_cvboostree.cpp:361-426, :192-238).
// ------------------------------------------------------------------------------------------------
constexpr int SPLIT_LEAN_WAVES = 12;  // wavefronts per block at most: 3 per SIMD, i.e. 168 VGPRs for the two chunks of table reads in flight
constexpr int SPLIT_TABLE_PAD_RANKS = 64;  // the sorted tables are allocated (and zeroed) this many ranks beyond the last group's end:
                                           // the lean kernel reads up to 3 chunks ahead without bounds checks
template <int MODE, class TI>
__global__ __launch_bounds__(64 * SPLIT_LEAN_WAVES) void k_split_ord_lean(SplitOrdArgs A) {
  extern __shared__ double l_tab[];
  const int lane = threadIdx.x & 63;
  const int group = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  {
    const double* src = reinterpret_cast<const double*>(A.tab);
    for (int i = threadIdx.x; i < A.n_pre; i += blockDim.x) l_tab[i] = src[i];
    __syncthreads();
  }
  if (group >= A.n_groups) return;
  const int f = group * 64 + lane;
  const size_t base = (size_t)group * A.n_pre * 64 + lane;
  const float* sv = A.sv + base;
  const TI* si = reinterpret_cast<const TI*>(A.si) + base;
  const float epsilon = FLT_EPSILON * 2;
  constexpr unsigned NOT_IN_NODE_HI = 0x7ff80000u;  // high word of the quiet NaN the host writes for samples outside the node (low word 0)
  double L = 0, R, lsum = 0, rsum = 0, lcw0 = 0, lcw1 = 0, rcw0 = A.w_total0, rcw1 = A.w_total1;
  if (MODE == 0) {
    R = A.w_total0;
    rsum = A.rsum0;
  } else {
    R = rcw0 + rcw1;
    rsum = rcw0 * rcw0 + rcw1 * rcw1;  // rsum2; lsum plays lsum2
  }
  double best_val = -1.0;
  double bt = -HUGE_VAL;  // best_val * (1 - 2^-50) once best_val > 0; -inf before: every value change is a candidate
  int best_i = -1, count = 0;
  float prev = 0.f, vl = 0.f, vr = 0.f;
  constexpr int U = CC_SPLIT_UNROLL;
  static_assert(3 * U <= SPLIT_TABLE_PAD_RANKS, "table padding covers the read-ahead");
  float va[U], vb[U];
  unsigned sa[U], sb[U];
  auto load = [&](int r0, float (&vv)[U], unsigned (&ss)[U]) {  // no bounds checks: see SPLIT_TABLE_PAD_RANKS
#pragma unroll
    for (int k = 0; k < U; k++) {
      vv[k] = sv[(size_t)(r0 + k) * 64];
      ss[k] = (unsigned)si[(size_t)(r0 + k) * 64];
    }
  };
  // A chunk is consumed in blocks of B ranks: the running sums, the quality's numerator / denominator and the "cannot beat
  // the record" test of the B ranks are straight-line code (the B evaluations are independent chains the scheduler can
  // interleave -- a branch per rank, as in k_split_ord, serialises them: the 4-deep dependent double chain plus the branch
  // of one rank then sits in front of the next rank's), and ONE branch per block asks whether any lane has a candidate
  // in any of the B ranks. Inside it the candidates are taken in rank order against the up-to-date record, exactly as the
  // sequential loop does. The test in the straight-line part uses the record as of the block's start: an older (smaller)
  // record only makes the test more conservative (more candidates), never wrong.
  constexpr int B = 4;
  static_assert(U % B == 0, "chunks are whole blocks");
  auto process = [&](int r0, const float (&vv)[U], const unsigned (&ss)[U]) {
    double x[U];
#pragma unroll
    for (int k = 0; k < U; k++) {
      unsigned g = A.dbg_nogather ? (unsigned)lane : ss[k];
      g = g < (unsigned)A.n_pre ? g : 0u;  // ranks behind the last group's end carry padding
      x[k] = l_tab[g];
    }
#pragma unroll
    for (int k0 = 0; k0 < U; k0 += B) {
      double num[B], den[B];  // MODE 2: num = the quality itself
      bool cand[B];
      int cnt[B];
      float pv[B];
#pragma unroll
      for (int j = 0; j < B; j++) {
        const int k = k0 + j;
        const unsigned long long xb = (unsigned long long)__double_as_longlong(x[k]);
        const unsigned lo = (unsigned)xb, hi = (unsigned)(xb >> 32);
        const bool member = (hi != NOT_IN_NODE_HI) & (r0 + k < A.n_pre);
        const unsigned thi = member ? hi : 0u;  // outside the node: +0.0 (the marker's low word is 0)
        const unsigned tlo = member ? lo : 0u;  // (read-ahead ranks may carry any entry)
        const double w = __hiloint2double((int)(thi & 0x7fffffffu), (int)tlo);
        const float val_k = vv[k];
        const bool changed = member & (count > 0) & (prev + epsilon < val_k);
        cnt[j] = count;
        pv[j] = prev;
        if (MODE == 2) {
          const double a = lcw0 + rcw1, b = lcw1 + rcw0;
          num[j] = a > b ? a : b;
          den[j] = 1.0;
          cand[j] = changed & (best_val < num[j]);
        } else {
          num[j] = MODE == 0 ? lsum * lsum * R + rsum * rsum * L : lsum * R + rsum * L;
          den[j] = L * R;
          // anything but "den > 0 and provably below the record" takes the division (incl. den <= 0, NaN), as k_split_ord does
          cand[j] = changed & !((den[j] > 0.0) & (num[j] < bt * den[j]));
        }
        if (MODE == 0) {
          const double t = __hiloint2double((int)thi, (int)tlo);  // response * w, sign included
          L += w;
          R -= w;
          lsum += t;
          rsum -= t;
        } else {
          const bool c1 = (int)thi < 0;  // class 1 carries the sign bit
          if (MODE == 1) {
            const double w2 = w * w;
            L += w;
            R -= w;
            const double lv = c1 ? lcw1 : lcw0, rv = c1 ? rcw1 : rcw0;
            lsum += 2 * lv * w + w2;
            rsum -= 2 * rv * w - w2;
            const double nl = lv + w, nr = rv - w;
            lcw1 = c1 ? nl : lcw1;
            rcw1 = c1 ? nr : rcw1;
            lcw0 = c1 ? lcw0 : nl;
            rcw0 = c1 ? rcw0 : nr;
          } else {
            const double w1 = c1 ? w : 0.0, w0 = c1 ? 0.0 : w;
            lcw1 += w1;
            rcw1 -= w1;
            lcw0 += w0;
            rcw0 -= w0;
          }
        }
        prev = member ? val_k : prev;
        count += member ? 1 : 0;
      }
      bool any = cand[0];
#pragma unroll
      for (int j = 1; j < B; j++) any |= cand[j];
      if (__builtin_expect(any, 0)) {
#pragma unroll
        for (int j = 0; j < B; j++) {
          if (cand[j]) {
            const double val = MODE == 2 ? num[j] : num[j] / den[j];
            if (best_val < val) {
              best_val = val;
              if (MODE != 2) bt = val > 0.0 ? val * (1.0 - 0x1p-50) : -HUGE_VAL;
              best_i = cnt[j] - 1;
              vl = pv[j];
              vr = vv[k0 + j];
            }
          }
        }
      }
    }
  };
  load(0, va, sa);
  for (int r0 = 0; r0 < A.n_pre; r0 += 2 * U) {  // two chunks per trip: the next chunk's table reads fly while one is consumed
    load(r0 + U, vb, sb);
    process(r0, va, sa);
    load(r0 + 2 * U, va, sa);
    process(r0 + U, vb, sb);
  }
  if (f < A.n_vars) {
    A.best_val[f] = best_val;
    A.best_i[f] = best_i;
    A.best_vl[f] = vl;
    A.best_vr[f] = vr;
  }
}

// ------------------------------------------------------------------------------------------------
// Categorical variables (LBP, 256 categories): one block per variable, one thread per category; the node's samples are
// streamed through LDS in node order and every thread adds the samples of its own category, i.e. each category's sums
// are accumulated in the reference's order (o_cvboostree.cpp:456-464 regression, :283-288 classification).
// hist[var][category] = {sum of response*w, sum of w} or {w of class 0, w of class 1}.
// ------------------------------------------------------------------------------------------------
template <bool CLASSIFIER>
__global__ __launch_bounds__(256) void k_split_cat(const uint8_t* __restrict__ codes, int n_pre, const int32_t* __restrict__ node_idx,
                                                   const SplitEntry* __restrict__ node_tab, int n, double* __restrict__ hist) {
  __shared__ int l_code[256];
  __shared__ double l_w[256], l_t[256];
  const int f = blockIdx.x, cat = threadIdx.x;
  double a0 = 0, a1 = 0;
  for (int i0 = 0; i0 < n; i0 += 256) {
    const int i = i0 + threadIdx.x;
    int code = -1;
    double w = 0, t = 0;
    if (i < n) {
      const int g = node_idx ? node_idx[i] : i;
      code = codes[(size_t)f * n_pre + g];
      const SplitEntry e = node_tab[i];
      w = e.w;
      t = e.t;
    }
    l_code[threadIdx.x] = code;
    l_w[threadIdx.x] = w;
    l_t[threadIdx.x] = t;
    __syncthreads();
    const int m = min(256, n - i0);
    for (int j = 0; j < m; j++) {
      if (l_code[j] == cat) {
        if (CLASSIFIER) {
          if (l_t[j] != 0.0)
            a1 += l_w[j];
          else
            a0 += l_w[j];
        } else {
          a0 += l_t[j];
          a1 += l_w[j];
        }
      }
    }
    __syncthreads();
  }
  hist[((size_t)f * 256 + cat) * 2] = a0;
  hist[((size_t)f * 256 + cat) * 2 + 1] = a1;
}

// ------------------------------------------------------------------------------------------------
// Categorical variables, round 4: the same per-category sums from a table sorted once per stage.
// cc_eval_presort sorts every LBP variable's samples by (code, sample) -- the stable row sort the ordered variables use --
// and stores (sample << 8 | code) as [group][rank][64]. A category's samples are then one contiguous run in increasing
// sample order, so ONE thread per variable (lane = variable: coalesced 256-B reads per rank, as k_split_ord) adds each
// category up in the reference's order (o_cvboostree.cpp:456-464, :283-288) PROVIDED the node lists its samples in increasing
// order, which is what a trainer's nodes do (the root is 0..n-1 or the kept samples in order, children are stable
// partitions of their parent); any other node order takes k_split_cat above. Samples outside the node add +0.0 (exact:
// the sums start at +0 and x + 0.0 == x). Work per node: n_pre table entries per variable instead of n * 256 compares.
// ------------------------------------------------------------------------------------------------
struct SplitCatArgs {
  const uint32_t* packed;
  const SplitEntry* tab;
  int n_pre, n_vars, n_groups;
  int waves;     // wavefronts of a block that own a (group, part) (the others only help to copy the table)
  int parts;     // a variable's ranks are cut into this many parts ...
  int part_len;  // ... of this many ranks (a multiple of SPLIT_CAT_DEPTH * SPLIT_CAT_UNROLL)
  double* hist;
};
constexpr int SPLIT_CAT_UNROLL = 16;  // ranks per chunk: four 16-byte loads per lane
constexpr int SPLIT_CAT_DEPTH = 6;    // chunks whose loads are in flight (one wavefront per SIMD: latency is hidden by distance, not by occupancy)
// zeroed ranks behind the table: read-ahead without bounds checks. A trip that starts at rank r0 < n_pre consumes DEPTH chunks and
// requests the DEPTH after them: ranks up to r0 + 2 * DEPTH * UNROLL - 1.
constexpr int SPLIT_CAT_PAD_RANKS = (2 * SPLIT_CAT_DEPTH + 1) * SPLIT_CAT_UNROLL;
// Table layout: [group][rank / 4][64 lanes][4 ranks] -- a lane reads four consecutive ranks of its variable with one 16-byte
// load, a wavefront 1 KB per load instruction. Per-sample table: a sample outside the node is stored as +0.0 (8-byte form:
// response * w, or w with the class in the sign bit) or {0, 0} (16-byte form), so it takes no test at all.
// Parts: 8 464 LBP variables are 133 groups -- one wavefront on every eighth SIMD. A variable's ranks can be cut into parts,
// each walked by its own wavefront, without splitting a sum: the part in which a category's run STARTS owns it, walks on past
// its nominal end until that run is over, and skips a run it found already under way at its first rank. (Helps when runs
// are short; aligned positives give a variable one code for half the samples, and the walk past the end then costs what
// the cut saved.)
template <bool CLASSIFIER, int TAB>
__global__ __launch_bounds__(256) void k_split_cat_sorted(SplitCatArgs A) {
  extern __shared__ double l_tab[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // scalar: rank tests below are wave-uniform
  if (TAB != 0) {
    const int words = A.n_pre * (TAB == 1 ? 2 : 1);
    const double* src = reinterpret_cast<const double*>(A.tab);
    for (int i = threadIdx.x; i < words; i += blockDim.x) l_tab[i] = src[i];
    __syncthreads();
  }
  const int unit = blockIdx.x * A.waves + wave;
  const int group = unit / A.parts, part = unit - group * A.parts;
  if (wave >= A.waves || group >= A.n_groups) return;
  const int f = group * 64 + lane;
  const bool live = f < A.n_vars;
  const int n4 = (A.n_pre + 3) >> 2;
  const uint4* p = reinterpret_cast<const uint4*>(A.packed) + (size_t)group * n4 * 64 + lane;
  double* h = A.hist + (size_t)(live ? f : 0) * 512;
  const int begin = part * A.part_len, end = min(A.n_pre, begin + A.part_len);
  if (begin >= A.n_pre) return;
  // cur: the code of the run the walk is in; own: this part sums that run
  int cur = -1;
  if (begin > 0) cur = (int)(A.packed[(((size_t)group * n4 + ((begin - 1) >> 2)) * 64 + lane) * 4 + ((begin - 1) & 3)] & 255u);
  bool own = false;
  double a0 = 0, a1 = 0;
  constexpr int U = SPLIT_CAT_UNROLL, D = SPLIT_CAT_DEPTH;
  static_assert(2 * D * U <= SPLIT_CAT_PAD_RANKS && U % 4 == 0, "table padding covers the read-ahead");
  uint4 buf[D][U / 4];
  auto load = [&](int r0, uint4(&x)[U / 4]) {
#pragma unroll
    for (int k = 0; k < U / 4; k++) x[k] = p[(size_t)((r0 >> 2) + k) * 64];
  };
  auto process = [&](int r0, const uint4(&x4)[U / 4]) {
    uint32_t x[U];
#pragma unroll
    for (int k = 0; k < U / 4; k++) {
      x[4 * k] = x4[k].x;
      x[4 * k + 1] = x4[k].y;
      x[4 * k + 2] = x4[k].z;
      x[4 * k + 3] = x4[k].w;
    }
    double w[U], t[U];
#pragma unroll
    for (int k = 0; k < U; k++) {
      const unsigned g = x[k] >> 8;
      if (TAB == 2) {
        t[k] = l_tab[g];
        w[k] = 0;
      } else {
        const SplitEntry e = TAB == 1 ? reinterpret_cast<const SplitEntry*>(l_tab)[g] : A.tab[g];
        w[k] = e.w;
        t[k] = e.t;
      }
    }
#pragma unroll
    for (int k = 0; k < U; k++) {
      if (r0 + k >= A.n_pre) break;  // wave-uniform
      const int code = (int)(x[k] & 255u);
      if (code != cur) {
        if (own && live) {
          h[2 * cur] = a0;
          h[2 * cur + 1] = a1;
        }
        a0 = 0;
        a1 = 0;
        cur = code;
        own = r0 + k < end;  // a run that starts behind the nominal end is the next part's
      }
      if (TAB == 2) {
        if (CLASSIFIER) {  // entry = w, class 1 in the sign bit: max(x, +0) is w for class 0 and 0 for class 1
          a0 += fmax(t[k], 0.0);
          a1 += fmax(-t[k], 0.0);
        } else {  // entry = response * w with response +-1
          a0 += t[k];
          a1 += fabs(t[k]);
        }
      } else if (CLASSIFIER) {
        const bool c1 = t[k] != 0.0;
        a0 += c1 ? 0.0 : w[k];
        a1 += c1 ? w[k] : 0.0;
      } else {
        a0 += t[k];
        a1 += w[k];
      }
    }
  };
#pragma unroll
  for (int d = 0; d < D; d++) load(begin + d * U, buf[d]);
  int r0 = begin;
  while (r0 < end || (r0 < A.n_pre && __any(own))) {  // ... || runs that straddle the nominal end
#pragma unroll
    for (int d = 0; d < D; d++) {
      process(r0 + d * U, buf[d]);
      load(r0 + (D + d) * U, buf[d]);
    }
    r0 += D * U;
  }
  if (own && live) {
    h[2 * cur] = a0;
    h[2 * cur + 1] = a1;
  }
}

// [rows][n] sorted codes (as floats) and sample positions -> (sample << 8 | code) as [group][rank / 4][64][4]
__global__ __launch_bounds__(256) void k_interleave_cat(const float* __restrict__ vals, const int* __restrict__ idx, int rows, int n,
                                                        uint32_t* __restrict__ out, size_t group0) {
  __shared__ uint32_t tv[64][65];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r0 = blockIdx.x * 64, g = blockIdx.y;
  for (int j = wave; j < 64; j += 4) {
    const int row = g * 64 + j;
    uint32_t v = 0;
    if (row < rows && r0 + lane < n) v = ((uint32_t)idx[(size_t)row * n + r0 + lane] << 8) | ((uint32_t)(int)vals[(size_t)row * n + r0 + lane] & 255u);
    tv[j][lane] = v;
  }
  __syncthreads();
  const size_t n4 = ((size_t)n + 3) >> 2;
  for (int rr = wave; rr < 64; rr += 4) {  // [group][rank / 4][lane][rank % 4]; r0 is a multiple of 64
    if (r0 + rr >= n) break;
    out[(((group0 + g) * n4 + ((r0 + rr) >> 2)) * 64 + lane) * 4 + (rr & 3)] = tv[lane][rr];
  }
}

}  // namespace ccamd

using namespace ccamd;

// compute units of a device (hipGetDeviceProperties fills a 1.5 KB struct and takes ~1 ms: asked once per device)
static int device_cus(int device) {
  static std::mutex mu;
  static int cached[64] = {0};
  std::lock_guard<std::mutex> lk(mu);
  if (device >= 0 && device < 64 && cached[device] > 0) return cached[device];
  int v = 0;
  if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || v < 1) v = 256;
  if (device >= 0 && device < 64) cached[device] = v;
  return v;
}

extern "C" {

cc_status cc_eval_presort_range(cc_evaluator* e, int fi_begin, int fi_end, int n_samples) {
  if (!e) return set_error(CC_ERR_INVALID_ARG, "cc_eval_presort: null evaluator");
  if (n_samples < 1 || n_samples > e->max_samples)
    return set_error(CC_ERR_OUT_OF_RANGE, "cc_eval_presort: n_samples %d out of range (max_samples %d)", n_samples, e->max_samples);
  if (fi_begin < 0 || fi_end > e->nfeat || fi_begin >= fi_end)
    return set_error(CC_ERR_OUT_OF_RANGE, "cc_eval_presort: features [%d, %d) out of range (%d)", fi_begin, fi_end, e->nfeat);
  cc_status st = eval_device(e);
  if (st != CC_OK) return st;
  std::lock_guard<std::mutex> lk(e->mu);
  st = flush_pending_images(e);
  if (st != CC_OK) return st;
  e->presort_n = 0;
  const bool haar = e->type == CC_FEATURE_HAAR;
  const int F = fi_end - fi_begin, N = n_samples;
  const void* feats = haar ? (const void*)e->d_haar.p : (const void*)e->d_lbp.p;
  // variables per pass: whole groups of 64, at most 2^28 values per scratch array
  int FB = (int)std::min<size_t>((size_t)F, std::max<size_t>(64, (((size_t)1 << 28) / (size_t)N) / 64 * 64));
  const size_t groups = ((size_t)F + 63) / 64;
  size_t free_b = 0, total_b = 0;
  CC_HIP(hipMemGetInfo(&free_b, &total_b));
  // LBP keeps the (code, sample)-sorted table of k_split_cat_sorted beside the codes while sample numbers fit 24 bits
  const bool no_cat_table = std::getenv("CCAMD_SPLIT_CAT_STREAM") != nullptr;  // A/B: round-1 categorical search only
  const bool cat_table = !haar && N < (1 << 24) && !no_cat_table;
  const size_t resident = haar ? groups * 64 * (size_t)N * (4 + (N <= 65536 ? 2 : 4)) : (size_t)F * N + (cat_table ? groups * 64 * ((size_t)N + 3 + SPLIT_CAT_PAD_RANKS) * 4 : 0);
  const size_t have = e->d_sorted_val.n * 4 + e->d_sorted_idx16.n * 2 + e->d_sorted_idx32.n * 4 + e->d_codes.n + e->d_cat_sorted.n * 4 + e->d_out.n * 4;
  const size_t scratch = (size_t)FB * N * (haar || cat_table ? 16 : 4);
  if (resident + scratch > free_b + have)
    return set_error(CC_ERR_UNSUPPORTED, "cc_eval_presort: needs %.1f GB of device memory (%.1f GB free)",
                     (double)(resident + scratch) / 1e9, (double)(free_b + have) / 1e9);
  EBuf<float> keys_out;
  EBuf<int> iota, sorted, offsets;
  EBuf<char> temp;
  const size_t cap = (size_t)FB * N;
  if (haar || cat_table) {
    CC_HIP(e->d_out.ensure(cap));
    CC_HIP(keys_out.ensure(cap));
    CC_HIP(sorted.ensure(cap));
  }
  bool scratch_for_device_sort = false;
  // stable sort of the nf rows of d_out (one variable per row) into keys_out / sorted
  auto sort_rows = [&](int nf) -> cc_status {
    if (N <= sort_rows_block_limit()) {  // a row fits one block: sorted in LDS, read once and written once
      // a part that refuses the ~100 KB LDS request (or the launch) takes the device-wide sort below instead of failing
      if (sort_rows_block(e->d_out.p, nf, N, keys_out.p, sorted.p, e->stream) == hipSuccess) return CC_OK;
      (void)hipGetLastError();
    }
    if (!scratch_for_device_sort) {
      CC_HIP(iota.ensure(cap));
      CC_HIP(offsets.ensure((size_t)FB + 1));
      std::vector<int> off((size_t)FB + 1);
      for (int i = 0; i <= FB; i++) off[(size_t)i] = (int)((size_t)i * N);
      CC_HIP(hipMemcpyAsync(offsets.p, off.data(), off.size() * 4, hipMemcpyHostToDevice, e->stream));
      CC_HIP(hipStreamSynchronize(e->stream));  // `off` is pageable and about to go out of scope
      hipLaunchKernelGGL(k_iota_rows2, dim3((unsigned)((cap + 255) / 256)), dim3(256), 0, e->stream, iota.p, cap, N);
      scratch_for_device_sort = true;
    }
    const size_t total = (size_t)nf * N;
    size_t temp_bytes = 0;
    CC_HIP(hipcub::DeviceSegmentedRadixSort::SortPairs(nullptr, temp_bytes, e->d_out.p, keys_out.p, iota.p, sorted.p, (int)total, nf, offsets.p,
                                                       offsets.p + 1, 0, 32, e->stream));
    CC_HIP(temp.ensure(std::max<size_t>(temp_bytes, 1)));
    // stable: equal values keep increasing sample order
    CC_HIP(hipcub::DeviceSegmentedRadixSort::SortPairs(temp.p, temp_bytes, e->d_out.p, keys_out.p, iota.p, sorted.p, (int)total, nf, offsets.p,
                                                       offsets.p + 1, 0, 32, e->stream));
    return CC_OK;
  };
  if (!haar) {
    CC_HIP(e->d_codes.ensure((size_t)F * N));
    if (cat_table) {
      // ranks padded to a multiple of 4 per group (the tail of a group's last quad is never summed: r >= n_pre); zeroed
      // padding behind the last group for the read-ahead
      const size_t used = groups * 64 * ((((size_t)N + 3) >> 2) << 2), pad = (size_t)SPLIT_CAT_PAD_RANKS * 64;
      CC_HIP(e->d_cat_sorted.ensure(used + pad));
      CC_HIP(hipMemsetAsync(e->d_cat_sorted.p, 0, (used + pad) * sizeof(uint32_t), e->stream));
    }
    e->cat_sorted_n = 0;
    for (int f0 = 0; f0 < F; f0 += FB) {
      const int f1 = std::min(F, f0 + FB), nf = f1 - f0;
      const size_t total = (size_t)nf * N;
      CC_HIP(e->d_out.ensure(total));
      st = launch_batch(e, false, feats, fi_begin + f0, fi_begin + f1, nullptr, N, e->d_out.p, 1, 0);
      if (st != CC_OK) return st;
      hipLaunchKernelGGL(k_codes_u8, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, e->stream, e->d_out.p,
                         e->d_codes.p + (size_t)f0 * N, total);
      if (cat_table) {
        st = sort_rows(nf);
        if (st != CC_OK) return st;
        hipLaunchKernelGGL(k_interleave_cat, dim3((unsigned)((N + 63) / 64), (unsigned)((nf + 63) / 64)), dim3(256), 0, e->stream, keys_out.p,
                           sorted.p, nf, N, e->d_cat_sorted.p, (size_t)f0 / 64);
      }
    }
    CC_HIP(hipGetLastError());
    CC_HIP(hipStreamSynchronize(e->stream));
    e->presort_n = N;
    e->presort_f0 = fi_begin;
    e->presort_f1 = fi_end;
    e->cat_sorted_n = cat_table ? N : 0;
    return CC_OK;
  }
  const bool idx16 = N <= 65536;
  {  // tables + SPLIT_TABLE_PAD_RANKS ranks of zeroed padding behind the last group (read-ahead of k_split_ord_lean)
    const size_t used = groups * 64 * (size_t)N, pad = (size_t)SPLIT_TABLE_PAD_RANKS * 64;
    CC_HIP(e->d_sorted_val.ensure(used + pad));
    CC_HIP(hipMemsetAsync(e->d_sorted_val.p + used, 0, pad * sizeof(float), e->stream));
    if (idx16) {
      CC_HIP(e->d_sorted_idx16.ensure(used + pad));
      CC_HIP(hipMemsetAsync(e->d_sorted_idx16.p + used, 0, pad * sizeof(uint16_t), e->stream));
    } else {
      CC_HIP(e->d_sorted_idx32.ensure(used + pad));
      CC_HIP(hipMemsetAsync(e->d_sorted_idx32.p + used, 0, pad * sizeof(int32_t), e->stream));
    }
  }
  for (int f0 = 0; f0 < F; f0 += FB) {
    const int f1 = std::min(F, f0 + FB), nf = f1 - f0;
    st = launch_batch(e, true, feats, fi_begin + f0, fi_begin + f1, nullptr, N, e->d_out.p, 1, 0);
    if (st != CC_OK) return st;
    st = sort_rows(nf);
    if (st != CC_OK) return st;
    const dim3 grid((unsigned)((N + 63) / 64), (unsigned)((nf + 63) / 64));
    if (idx16)
      hipLaunchKernelGGL((k_interleave<uint16_t>), grid, dim3(256), 0, e->stream, keys_out.p, sorted.p, nf, N, e->d_sorted_val.p,
                         e->d_sorted_idx16.p, (size_t)f0 / 64);
    else
      hipLaunchKernelGGL((k_interleave<int32_t>), grid, dim3(256), 0, e->stream, keys_out.p, sorted.p, nf, N, e->d_sorted_val.p,
                         e->d_sorted_idx32.p, (size_t)f0 / 64);
    CC_HIP(hipGetLastError());
  }
  CC_HIP(hipStreamSynchronize(e->stream));
  e->presort_n = N;
  e->presort_f0 = fi_begin;
  e->presort_f1 = fi_end;
  return CC_OK;
}

cc_status cc_eval_presort(cc_evaluator* e, int n_samples) {
  if (!e) return set_error(CC_ERR_INVALID_ARG, "cc_eval_presort: null evaluator");
  return cc_eval_presort_range(e, 0, e->nfeat, n_samples);
}

cc_status cc_eval_find_best_split(cc_evaluator* e, const int32_t* sample_idx, int n, const double* weights, const float* responses,
                                  const int32_t* class_labels, double node_value, int boost_type, int split_criteria, cc_split* out,
                                  double* per_var_quality, int32_t* per_var_point) {
  if (!e || !weights || !out) return set_error(CC_ERR_INVALID_ARG, "cc_eval_find_best_split: null argument");
  if (boost_type < 0 || boost_type > 3) return set_error(CC_ERR_INVALID_ARG, "cc_eval_find_best_split: unknown boost type %d", boost_type);
  const bool is_classifier = boost_type == 0 || boost_type == 1;
  if (is_classifier ? !class_labels : !responses)
    return set_error(CC_ERR_INVALID_ARG, "cc_eval_find_best_split: %s", is_classifier ? "class_labels required for DISCRETE / REAL boost"
                                                                                        : "responses required for LOGIT / GENTLE boost");
  if (e->presort_n <= 0) return set_error(CC_ERR_INVALID_ARG, "cc_eval_find_best_split: call cc_eval_presort first");
  const int N = e->presort_n;
  if (n < 0 || n > N) return set_error(CC_ERR_OUT_OF_RANGE, "cc_eval_find_best_split: n %d exceeds the %d presorted samples", n, N);
  int criteria = split_criteria;
  if (criteria != 1 && criteria != 3) criteria = boost_type == 0 ? 3 : 1;  // o_cvboostree.cpp:188-190
  const bool gini = criteria == 1;
  cc_status st = eval_device(e);
  if (st != CC_OK) return st;
  std::memset(out, 0, sizeof(*out));
  out->quality = -1.f;
  const int F = e->presort_f1 - e->presort_f0, var0 = e->presort_f0;  // per-variable outputs are indexed from presort's fi_begin
  if (per_var_quality)
    for (int f = 0; f < F; f++) per_var_quality[f] = -1.0;
  if (per_var_point)
    for (int f = 0; f < F; f++) per_var_point[f] = -1;
  if (n <= 1) return CC_OK;  // get_num_valid(vi) <= 1: no variable is searched
  for (int i = 0; i < n; i++) {
    if (is_classifier && (class_labels[i] < 0 || class_labels[i] > 1))
      return set_error(CC_ERR_INVALID_ARG, "cc_eval_find_best_split: class label %d of sample %d is not 0 / 1", class_labels[i], i);
    if (!(weights[i] >= 0.0)) return set_error(CC_ERR_INVALID_ARG, "cc_eval_find_best_split: weight of sample %d is negative or NaN", i);
  }
  std::lock_guard<std::mutex> lk(e->mu);
  const bool haar = e->type == CC_FEATURE_HAAR;
  PinnedBuf& pin_in = e->pin_in;
  PinnedBuf& pin_out = e->pin_out;
  if (haar) {
    // per stored sample {w, t}; not in this node: w = -1 (16-B entries) or NaN (8-B entries)
    const bool idx16 = N <= 65536;
    const int mode = !is_classifier ? 0 : (gini ? 1 : 2);
    bool unit_responses = !is_classifier;
    if (!is_classifier)
      for (int i = 0; i < n && unit_responses; i++) unit_responses = responses[i] == 1.0f || responses[i] == -1.0f;
    const size_t lds_cap = 160 * 1024;
    int tab_kind = 0;
    if ((is_classifier || unit_responses) && (size_t)N * 8 <= lds_cap)
      tab_kind = 2;
    else if ((size_t)N * 16 <= lds_cap)
      tab_kind = 1;
    if (std::getenv("CCAMD_SPLIT_GLOBAL_TABLE")) tab_kind = 0;
    const size_t entry_bytes = tab_kind == 2 ? 8 : 16;
    CC_HIP(pin_in.ensure((size_t)N * entry_bytes));
    std::vector<uint8_t> seen((size_t)N, 0);
    SplitEntry* tab16 = static_cast<SplitEntry*>(pin_in.p);
    double* tab8 = static_cast<double*>(pin_in.p);
    for (int g = 0; g < N; g++) {
      if (tab_kind == 2)
        tab8[g] = std::numeric_limits<double>::quiet_NaN();
      else
        tab16[g] = SplitEntry{-1.0, 0.0};
    }
    for (int i = 0; i < n; i++) {
      const int g = sample_idx ? sample_idx[i] : i;
      if (g < 0 || g >= N) return set_error(CC_ERR_OUT_OF_RANGE, "cc_eval_find_best_split: sample index %d outside the %d presorted samples", g, N);
      if (seen[(size_t)g]) return set_error(CC_ERR_INVALID_ARG, "cc_eval_find_best_split: sample %d occurs twice in the node", g);
      seen[(size_t)g] = 1;
      const double w = weights[i];
      if (tab_kind == 2)
        tab8[g] = is_classifier ? (class_labels[i] ? -w : w) : responses[i] * w;
      else {
        tab16[g].w = w;
        tab16[g].t = is_classifier ? (double)class_labels[i] : responses[i] * w;
      }
    }
    const size_t groups = ((size_t)F + 63) / 64, fpad = groups * 64;
    CC_HIP(e->d_split_tab.ensure((size_t)N * 2));
    CC_HIP(e->d_split_out.ensure(fpad * 3));  // best_val (8 B) + best_i, vl, vr (4 B each) + slack
    CC_HIP(hipMemcpyAsync(e->d_split_tab.p, pin_in.p, (size_t)N * entry_bytes, hipMemcpyHostToDevice, e->stream));
    SplitOrdArgs A;
    A.sv = e->d_sorted_val.p;
    A.si = idx16 ? (const void*)e->d_sorted_idx16.p : (const void*)e->d_sorted_idx32.p;
    A.tab = reinterpret_cast<const SplitEntry*>(e->d_split_tab.p);
    A.n_pre = N;
    A.n_vars = F;
    A.n_groups = (int)groups;
    A.w_total0 = weights[n];
    A.w_total1 = weights[n + 1];
    A.rsum0 = node_value * weights[n];
    A.best_val = e->d_split_out.p;
    A.best_i = reinterpret_cast<int*>(e->d_split_out.p + fpad);
    A.best_vl = reinterpret_cast<float*>(A.best_i + fpad);
    A.best_vr = A.best_vl + fpad;
    A.dbg_nogather = std::getenv("CCAMD_DEBUG_SPLIT_NOGATHER") ? 1 : 0;
    // wavefronts per block: with the table in LDS one block owns a CU, so spread the groups evenly over the CUs
    // (162 336 variables = 2 537 groups -> 254 blocks of 10 wavefronts on 256 CUs); from global memory, one wavefront
    // k_split_ord_lean for the regression / GINI searches (Gentle 6.77 against 7.01 ms, GINI 8.91 against 9.31 at configs[4]);
    // the MISCLASS search has no division and nothing to hoist: the round-1 kernel stays (4.46 against 4.76 ms).
    // CCAMD_SPLIT_BRANCHY=1: the round-1 kernel everywhere (A/B runs).
    static const bool lean = std::getenv("CCAMD_SPLIT_BRANCHY") == nullptr;
    int wpb = 1;
    if (tab_kind != 0) {
      const int cus = device_cus(e->device);
      wpb = (int)std::min<size_t>(16, std::max<size_t>(1, (groups + cus - 1) / cus));
      if (const char* v = std::getenv("CCAMD_SPLIT_WAVES")) wpb = std::max(1, std::min(16, std::atoi(v)));
      if (tab_kind == 2 && lean && mode != 2) wpb = std::min(wpb, SPLIT_LEAN_WAVES);
    }
    const unsigned blocks = (unsigned)((groups + wpb - 1) / wpb);
    const size_t lds = tab_kind == 0 ? 0 : (size_t)N * entry_bytes;
    (void)hipEventRecord(e->ev_a, e->stream);
#define CC_LAUNCH_ORD3(M, TI, T)                                                                                              \
  do {                                                                                                                        \
    if (lds > 64 * 1024)                                                                                                      \
      CC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_split_ord<M, TI, T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    hipLaunchKernelGGL((k_split_ord<M, TI, T>), dim3(blocks), dim3(64 * wpb), lds, e->stream, A);                             \
  } while (0)
#define CC_LAUNCH_LEAN(M, TI)                                                                                                 \
  do {                                                                                                                        \
    if (lds > 64 * 1024)                                                                                                      \
      CC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_split_ord_lean<M, TI>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    hipLaunchKernelGGL((k_split_ord_lean<M, TI>), dim3(blocks), dim3(64 * wpb), lds, e->stream, A);                           \
  } while (0)
#define CC_LAUNCH_ORD2(M, TI)          \
  do {                                 \
    if (tab_kind == 0)                 \
      CC_LAUNCH_ORD3(M, TI, 0);        \
    else if (tab_kind == 1)            \
      CC_LAUNCH_ORD3(M, TI, 1);        \
    else if (lean && M != 2)           \
      CC_LAUNCH_LEAN(M, TI);           \
    else                               \
      CC_LAUNCH_ORD3(M, TI, 2);        \
  } while (0)
#define CC_LAUNCH_ORD(M)               \
  do {                                 \
    if (idx16)                         \
      CC_LAUNCH_ORD2(M, uint16_t);     \
    else                               \
      CC_LAUNCH_ORD2(M, int32_t);      \
  } while (0)
    if (mode == 0)
      CC_LAUNCH_ORD(0);
    else if (mode == 1)
      CC_LAUNCH_ORD(1);
    else
      CC_LAUNCH_ORD(2);
#undef CC_LAUNCH_ORD
#undef CC_LAUNCH_LEAN
#undef CC_LAUNCH_ORD2
#undef CC_LAUNCH_ORD3
    (void)hipEventRecord(e->ev_b, e->stream);
    CC_HIP(hipGetLastError());
    CC_HIP(pin_out.ensure(fpad * 24));
    CC_HIP(hipMemcpyAsync(pin_out.p, e->d_split_out.p, fpad * 20, hipMemcpyDeviceToHost, e->stream));
    CC_HIP(hipStreamSynchronize(e->stream));
    float ms = 0;
    if (hipEventElapsedTime(&ms, e->ev_a, e->ev_b) == hipSuccess) e->last_ms = ms;
    const double* bv = static_cast<const double*>(pin_out.p);
    const int* bi = reinterpret_cast<const int*>(bv + fpad);
    const float* vl = reinterpret_cast<const float*>(bi + fpad);
    const float* vr = vl + fpad;
    // the winner, variable by variable as DTreeBestSplitFinder::operator() does (o_cvdtree.cpp:320-342): a variable
    // reports a split only if it beats the best quality so far (a float), and replaces it only if its own quality,
    // rounded to float, is larger
    float best_q = -1.f;
    int winner = -1;
    for (int f = 0; f < F; f++) {
      if (per_var_quality) per_var_quality[f] = bi[f] >= 0 ? bv[f] : -1.0;
      if (per_var_point) per_var_point[f] = bi[f];
      if (bi[f] < 0 || !((double)best_q < bv[f])) continue;
      const float q = (float)bv[f];
      if (best_q < q) {
        best_q = q;
        winner = f;
      }
    }
    if (winner >= 0 && best_q > 0) {  // o_cvdtree.cpp:351
      out->found = 1;
      out->var_idx = var0 + winner;
      out->quality = best_q;
      out->ord_c = (vl[winner] + vr[winner]) * 0.5f;
      out->split_point = bi[winner];
    }
    return CC_OK;
  }
  // ---- categorical (LBP) ----
  const size_t hist_n = (size_t)F * 256 * 2;
  CC_HIP(e->d_split_out.ensure(hist_n));
  bool ascending = true;  // the node lists its samples in increasing order: what k_split_cat_sorted's exactness needs
  for (int i = 1; i < n && ascending && sample_idx; i++) ascending = sample_idx[i - 1] < sample_idx[i];
  const bool stream_only = std::getenv("CCAMD_SPLIT_CAT_STREAM") != nullptr;  // read per call: tests compare the two paths
  if (ascending && e->cat_sorted_n == N && !stream_only) {
    bool unit_responses = !is_classifier;
    if (!is_classifier)
      for (int i = 0; i < n && unit_responses; i++) unit_responses = responses[i] == 1.0f || responses[i] == -1.0f;
    const size_t lds_cap = 160 * 1024;
    int tab_kind = 0;  // as for the ordered search: 2 = 8-byte entries in LDS, 1 = 16-byte entries in LDS, 0 = global memory
    if ((is_classifier || unit_responses) && (size_t)N * 8 <= lds_cap)
      tab_kind = 2;
    else if ((size_t)N * 16 <= lds_cap)
      tab_kind = 1;
    if (std::getenv("CCAMD_SPLIT_GLOBAL_TABLE")) tab_kind = 0;
    const size_t entry_bytes = tab_kind == 2 ? 8 : 16;
    CC_HIP(pin_in.ensure((size_t)N * entry_bytes));
    SplitEntry* tab16 = static_cast<SplitEntry*>(pin_in.p);
    double* tab8 = static_cast<double*>(pin_in.p);
    std::memset(pin_in.p, 0, (size_t)N * entry_bytes);  // not in the node: +0.0 / {0, 0}, adds nothing to any sum
    for (int i = 0; i < n; i++) {  // strictly increasing: no sample twice
      const int g = sample_idx ? sample_idx[i] : i;
      if (g < 0 || g >= N) return set_error(CC_ERR_OUT_OF_RANGE, "cc_eval_find_best_split: sample index %d outside the %d presorted samples", g, N);
      const double w = weights[i];
      if (tab_kind == 2)
        tab8[g] = is_classifier ? (class_labels[i] ? -w : w) : responses[i] * w;
      else {
        tab16[g].w = w;
        tab16[g].t = is_classifier ? (double)class_labels[i] : responses[i] * w;
      }
    }
    CC_HIP(e->d_split_tab.ensure((size_t)N * 2));
    CC_HIP(hipMemcpyAsync(e->d_split_tab.p, pin_in.p, (size_t)N * entry_bytes, hipMemcpyHostToDevice, e->stream));
    (void)hipEventRecord(e->ev_a, e->stream);
    CC_HIP(hipMemsetAsync(e->d_split_out.p, 0, hist_n * 8, e->stream));  // categories without a sample
    const size_t groups = ((size_t)F + 63) / 64;
    SplitCatArgs A;
    A.packed = e->d_cat_sorted.p;
    A.tab = reinterpret_cast<const SplitEntry*>(e->d_split_tab.p);
    A.n_pre = N;
    A.n_vars = F;
    A.n_groups = (int)groups;
    A.hist = e->d_split_out.p;
    // One wavefront per (group, part); with the table in LDS a block owns a CU (<= 4 wavefronts of it): as many parts as
    // give every CU a full block, no part shorter than 1 024 ranks. CCAMD_SPLIT_CAT_PARTS overrides (1 = round-4 first form).
    const int cus = device_cus(e->device);
    int parts = (int)std::max<size_t>(1, (size_t)cus * 4 / groups);
    parts = std::max(1, std::min(parts, N / 1024));
    if (const char* v = std::getenv("CCAMD_SPLIT_CAT_PARTS")) parts = std::max(1, std::min(64, std::atoi(v)));
    const int chunk = SPLIT_CAT_DEPTH * SPLIT_CAT_UNROLL;
    A.parts = parts;
    A.part_len = ((N + parts - 1) / parts + chunk - 1) / chunk * chunk;
    const size_t units = groups * (size_t)parts;
    A.waves = tab_kind == 0 ? 4 : (int)std::min<size_t>(4, std::max<size_t>(1, (units + cus - 1) / cus));
    const unsigned blocks = (unsigned)((units + A.waves - 1) / A.waves);
    const size_t lds = tab_kind == 0 ? 0 : (size_t)N * entry_bytes;
#define CC_LAUNCH_CAT(C, T)                                                                                                          \
  do {                                                                                                                               \
    if (lds > 64 * 1024)                                                                                                             \
      CC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_split_cat_sorted<C, T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    hipLaunchKernelGGL((k_split_cat_sorted<C, T>), dim3(blocks), dim3(256), lds, e->stream, A);                                      \
  } while (0)
    if (is_classifier) {
      if (tab_kind == 2)
        CC_LAUNCH_CAT(true, 2);
      else if (tab_kind == 1)
        CC_LAUNCH_CAT(true, 1);
      else
        CC_LAUNCH_CAT(true, 0);
    } else {
      if (tab_kind == 2)
        CC_LAUNCH_CAT(false, 2);
      else if (tab_kind == 1)
        CC_LAUNCH_CAT(false, 1);
      else
        CC_LAUNCH_CAT(false, 0);
    }
#undef CC_LAUNCH_CAT
    (void)hipEventRecord(e->ev_b, e->stream);
    CC_HIP(hipGetLastError());
  } else {
    CC_HIP(pin_in.ensure((size_t)n * (sizeof(SplitEntry) + 4)));
    SplitEntry* tab = static_cast<SplitEntry*>(pin_in.p);
    int32_t* idx_host = reinterpret_cast<int32_t*>(tab + n);
    {
      std::vector<uint8_t> seen((size_t)N, 0);
      for (int i = 0; i < n; i++) {
        const int g = sample_idx ? sample_idx[i] : i;
        if (g < 0 || g >= N) return set_error(CC_ERR_OUT_OF_RANGE, "cc_eval_find_best_split: sample index %d outside the %d presorted samples", g, N);
        if (seen[(size_t)g]) return set_error(CC_ERR_INVALID_ARG, "cc_eval_find_best_split: sample %d occurs twice in the node", g);
        seen[(size_t)g] = 1;
        idx_host[i] = g;
        tab[i].w = weights[i];
        tab[i].t = is_classifier ? (double)class_labels[i] : responses[i] * weights[i];
      }
    }
    CC_HIP(e->d_split_tab.ensure((size_t)n * 2));
    CC_HIP(e->d_split_idx.ensure((size_t)n));
    CC_HIP(hipMemcpyAsync(e->d_split_tab.p, tab, (size_t)n * sizeof(SplitEntry), hipMemcpyHostToDevice, e->stream));
    CC_HIP(hipMemcpyAsync(e->d_split_idx.p, idx_host, (size_t)n * 4, hipMemcpyHostToDevice, e->stream));
    (void)hipEventRecord(e->ev_a, e->stream);
    if (is_classifier)
      hipLaunchKernelGGL(k_split_cat<true>, dim3((unsigned)F), dim3(256), 0, e->stream, e->d_codes.p, N, sample_idx ? e->d_split_idx.p : nullptr,
                         reinterpret_cast<const SplitEntry*>(e->d_split_tab.p), n, e->d_split_out.p);
    else
      hipLaunchKernelGGL(k_split_cat<false>, dim3((unsigned)F), dim3(256), 0, e->stream, e->d_codes.p, N, sample_idx ? e->d_split_idx.p : nullptr,
                         reinterpret_cast<const SplitEntry*>(e->d_split_tab.p), n, e->d_split_out.p);
    (void)hipEventRecord(e->ev_b, e->stream);
    CC_HIP(hipGetLastError());
  }
  // The 34.7 MB of sums (8 464 variables) come back in pieces; the host's part -- ordering each variable's categories and
  // scanning them (split_categories) -- starts on a piece as soon as it has landed, on up to 16 threads.
  CC_HIP(pin_out.ensure(hist_n * 8));
  constexpr int kPieces = 8;
  for (int c = 0; c < kPieces; c++)
    if (!e->ev_piece[c]) CC_HIP(hipEventCreateWithFlags(&e->ev_piece[c], hipEventDisableTiming));
  const int per_piece = (F + kPieces - 1) / kPieces;
  double* hist = static_cast<double*>(pin_out.p);
  for (int c = 0; c < kPieces; c++) {
    const int f0 = std::min(F, c * per_piece), f1 = std::min(F, f0 + per_piece);
    if (f1 > f0)
      CC_HIP(hipMemcpyAsync(hist + (size_t)f0 * 512, e->d_split_out.p + (size_t)f0 * 512, (size_t)(f1 - f0) * 512 * 8, hipMemcpyDeviceToHost, e->stream));
    CC_HIP(hipEventRecord(e->ev_piece[c], e->stream));
  }
  std::vector<CatSplit> res((size_t)F);
  std::atomic<int> copy_failed{0};
  const bool trace = std::getenv("CCAMD_TRACE_SPLIT") != nullptr;  // host-side timeline of the call's tail (stderr)
  const auto t_enq = std::chrono::steady_clock::now();
  double landed_ms[kPieces] = {};
  {
    const int nt = std::max(1, std::min<int>({(int)std::thread::hardware_concurrency(), 16, F / 64 + 1}));
    std::vector<std::thread> th;
    for (int t = 0; t < nt; t++)
      th.emplace_back([&, t]() {
        for (int c = 0; c < kPieces; c++) {
          if (hipEventSynchronize(e->ev_piece[c]) != hipSuccess) {
            copy_failed = 1;
            return;
          }
          if (trace && t == 0) landed_ms[c] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_enq).count();
          const int f0 = std::min(F, c * per_piece), f1 = std::min(F, f0 + per_piece);
          for (int f = f0 + t; f < f1; f += nt) split_categories(hist + (size_t)f * 512, 256, is_classifier, gini, res[(size_t)f]);
        }
      });
    for (auto& x : th) x.join();
  }
  if (trace) {
    std::fprintf(stderr, "[ccamd split] after the last enqueue: pieces landed (as seen by worker 0) at");
    for (int c = 0; c < kPieces; c++) std::fprintf(stderr, " %.2f", landed_ms[c]);
    std::fprintf(stderr, " ms; workers done at %.2f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_enq).count());
  }
  CC_HIP(hipStreamSynchronize(e->stream));
  if (copy_failed) return set_error(CC_ERR_HIP, "cc_eval_find_best_split: copying the category sums back failed");
  float ms = 0;
  if (hipEventElapsedTime(&ms, e->ev_a, e->ev_b) == hipSuccess) e->last_ms = ms;
  float best_q = -1.f;
  int winner = -1;
  for (int f = 0; f < F; f++) {
    const CatSplit& r = res[(size_t)f];
    if (per_var_quality) per_var_quality[f] = r.found ? r.quality : -1.0;
    if (per_var_point) per_var_point[f] = r.found ? r.n_left - 1 : -1;
    if (!r.found || !((double)best_q < r.quality)) continue;
    const float q = (float)r.quality;
    if (best_q < q) {
      best_q = q;
      winner = f;
    }
  }
  if (winner >= 0 && best_q > 0) {
    out->found = 1;
    out->var_idx = var0 + winner;
    out->quality = best_q;
    std::memcpy(out->subset, res[(size_t)winner].subset, sizeof(out->subset));
  }
  return CC_OK;
}

}  // extern "C"
