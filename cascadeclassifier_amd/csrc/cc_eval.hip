// Training-side feature evaluator on gfx950: batched setImage (per-sample integral / tilted integral / norm factor),
// bulk operator() over (feature range x samples) and stump-cascade predict over stored samples.
// Replaces CvHaarEvaluator / CvLBPEvaluator (traincascade/lib/include/haarfeatures.h:61-122, lbpfeatures.h:37-83,
// traincascade/lib/src/haarfeatures.cpp:89-114, lbpfeatures.cpp:15-28) behind section 4 of the C ABI.
//
// Data layout in HBM: sum / tilted are [max_samples][(W+1)*(H+1)] int32, one sample per row exactly like the
// reference's `sum` / `tilted` Mats; normfactor is [max_samples] float. The batch kernel stages a tile of samples
// TRANSPOSED into LDS ([integral entry][sample]) so that a wavefront's lanes (samples) hit distinct banks, and writes
// out[(fi - fi_begin) * n_samples + s] with the sample index fastest (coalesced row segments).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>

#include "cc_eval_internal.h"

namespace ccamd {

// ------------------------------------------------------------------------------------------------
// setImage for a batch: one 64-thread block per sample.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_set_images(const uint8_t* __restrict__ imgs, int W, int H, int first_idx,
                                                   int32_t* __restrict__ sum, int32_t* __restrict__ tilted,
                                                   float* __restrict__ normfactor, int want_norm) {
  extern __shared__ int32_t lds[];  // row prefix sums [H][W+1]
  const int cols = (W + 1) * (H + 1), sw = W + 1;
  const int i = blockIdx.x;
  const uint8_t* img = imgs + (size_t)i * W * H;
  int32_t* s = sum + (size_t)(first_idx + i) * cols;
  for (int y = threadIdx.x; y < H; y += 64) {
    int acc = 0;
    lds[y * sw] = 0;
    for (int x = 0; x < W; x++) {
      acc += img[y * W + x];
      lds[y * sw + x + 1] = acc;
    }
  }
  __syncthreads();
  for (int x = threadIdx.x; x <= W; x += 64) {
    int acc = 0;
    s[x] = 0;
    for (int y = 0; y < H; y++) {
      acc += lds[y * sw + x];
      s[(y + 1) * sw + x] = acc;
    }
  }
  if (tilted) {
    int32_t* t = tilted + (size_t)(first_idx + i) * cols;
    for (int e = threadIdx.x; e < cols; e += 64) {
      const int Y = e / sw, X = e - Y * sw;
      int acc = 0;
      for (int y = 0; y < Y; y++) {
        const int half = Y - y - 1;
        const int x0 = max(X - 1 - half, 0), x1 = min(X - 1 + half, W - 1);
        if (x1 >= x0) acc += lds[y * sw + x1 + 1] - lds[y * sw + x0];
      }
      t[e] = acc;
    }
  }
  if (want_norm) {
    // calcNormFactor (features.cpp:13-25): sums over normrect (1,1,W-2,H-2); the reference's double sqsum integral
    // holds exact integers, so the 4-corner difference equals the exact integer sum of squares computed here.
    long long sq = 0;
    int sm = 0;
    for (int e = threadIdx.x; e < (W - 2) * (H - 2); e += 64) {
      const int y = 1 + e / (W - 2), x = 1 + e % (W - 2);
      const int p = img[y * W + x];
      sm += p;
      sq += p * p;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      sm += __shfl_xor(sm, d);
      sq += __shfl_xor(sq, d);
    }
    if (threadIdx.x == 0) {
      const double area = (double)((W - 2) * (H - 2));
      normfactor[first_idx + i] = (float)sqrt((double)(area * (double)sq - (double)sm * (double)sm));
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Bulk operator(): block = BATCH_THREADS threads, a tile of S samples staged transposed into LDS, loops over a feature chunk.
// lane -> (feature sub-index, sample): s = lane % S.
// ------------------------------------------------------------------------------------------------
struct BatchArgs {
  const int32_t* sum;
  const int32_t* tilted;
  const float* normfactor;
  const int32_t* sample_idx;  // optional
  int n_samples;
  int cols;
  int S;                      // samples per tile (power of two, <= 64)
  int feat_begin, feat_end;   // indices into the feature table
  int feats_per_block;
  const void* feats;
  float* out;                 // [feat_end - feat_begin][out_pitch], the first n_samples of a row are written
  size_t out_pitch;           // elements between the rows of two features (>= n_samples)
  int n_tiles, xcd_tiles;     // wide kernel: sample tiles, and whether the grid is laid out XCD by XCD
  int debug_nostore;          // timing experiment: skip the output stores
  int normalized;             // Haar: divide by normfactor (operator()) or not (Feature::calc)
  int use_tilted;
};

constexpr int BATCH_THREADS = 1024;  // 16 wavefronts per block: the LDS tile allows two blocks per CU = full occupancy

// LDS tile [entry p][S samples], XOR-swizzled: sample s of entry p sits at word p * S + (s ^ (p & (S - 1))).
// Staging writes walk consecutive entries of one sample (coalesced global reads): without the swizzle all 64 lanes of a
// store would hit one bank (stride S words); with it they spread over S banks. The evaluation reads one entry for S
// consecutive samples: the swizzle permutes them within the same S words, still conflict-free. The feature tables hold
// byte offsets of the swizzled entry, (p * S + (p & (S - 1))) * 4 (tilted features: plus the tilted tile's base), so that a
// corner read is ONE xor with the lane's sample byte index: address = entry ^ (s * 4).
__device__ __forceinline__ int batch_tile_word(int p, int s, int S) { return p * S + (s ^ (p & (min(S, 32) - 1))); }

// a / b, correctly rounded, with the part that depends on b alone hoisted: operator() divides every feature value of a
// sample by the same norm factor (haarfeatures.h:108-112). The compiler's expansion of an IEEE float division is
//   y0 = rcp(b); y1 = fma(fma(-b, y0, 1), y0, y0); q0 = a * y1; q1 = fma(fma(-b, q0, a), y1, q0); q = fma(fma(-b, q1, a), y1, q1)
// wrapped in v_div_scale / v_div_fmas / v_div_fixup, which rescale operands whose exponents would push an intermediate
// into the denormal or overflow range and patch zero / infinity / NaN results; for finite normal operands far from both
// ends (norm factors are >= 1 and < 2^21, feature sums are 0 or of magnitude >= 2^-8 and < 2^27) those three are the
// identity, so the same five operations give the same bits. A zero dividend keeps its sign. Checked against the division
// operator on the device over 2^32 operand pairs plus the edge values (cc_debug_division_check, tests/test_gpu_eval.py).
__device__ __forceinline__ float refined_rcp(float b) {
  const float y0 = __builtin_amdgcn_rcpf(b);
  return __builtin_fmaf(__builtin_fmaf(-b, y0, 1.0f), y0, y0);
}
__device__ __forceinline__ float div_by_refined(float a, float b, float y1) {
  const float q0 = a * y1;
  const float q1 = __builtin_fmaf(__builtin_fmaf(-b, q0, a), y1, q0);
  const float q = __builtin_fmaf(__builtin_fmaf(-b, q1, a), y1, q1);
  return a == 0.0f ? a : q;
}

// Counts operand pairs for which div_by_refined differs from the division operator (parity instrumentation).
__global__ void k_division_check(unsigned long long seed, int per_thread, unsigned long long* mismatches) {
  unsigned long long x = seed + (unsigned long long)(blockIdx.x * blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull;
  auto next = [&]() {
    x ^= x << 13;
    x ^= x >> 7;
    x ^= x << 17;
    return x;
  };
  unsigned long long bad = 0;
  for (int i = 0; i < per_thread; i++) {
    const unsigned long long r = next();
    // b: norm factors, any float in [1, 2^21); a: any float with magnitude in [2^-8, 2^27), either sign
    const float b = __uint_as_float(0x3F800000u + (unsigned)(r % (21u << 23)));
    const unsigned am = 0x3B800000u + (unsigned)((r >> 32) % (35u << 23));
    const float a = __uint_as_float(am | ((unsigned)(r >> 63) << 31));
    const float y1 = refined_rcp(b);
    if (__float_as_uint(div_by_refined(a, b, y1)) != __float_as_uint(a / b)) bad++;
    // integer-valued dividends (what the catalog's integer weights produce) against integer-valued and sqrt-like divisors
    const float ai = (float)(int)((r >> 8) % 33554432u) - 16777216.0f;
    const float bi = sqrtf((float)(1u + (unsigned)(r >> 40) % 16777215u) * (float)(1u + (unsigned)(r >> 16) % 65535u));
    if (bi >= 1.0f && bi < 2097152.0f && __float_as_uint(div_by_refined(ai, bi, refined_rcp(bi))) != __float_as_uint(ai / bi)) bad++;
  }
  if (bad) atomicAdd(mismatches, bad);
}

template <bool HAAR>
__global__ __launch_bounds__(BATCH_THREADS) void k_eval_batch(BatchArgs A) {
  extern __shared__ int32_t lds[];  // [cols][S] (+ [cols][S] tilted)
  const int S = A.S;
  const int s0 = blockIdx.x * S;
  int32_t* lsum = lds;
  int32_t* ltil = lds + (size_t)A.cols * S;
  // stage: consecutive threads read consecutive entries of one sample (coalesced), write transposed
  for (int e = threadIdx.x; e < A.cols * S; e += BATCH_THREADS) {
    const int s = e / A.cols, p = e - s * A.cols;
    int v = 0, t = 0;
    if (s0 + s < A.n_samples) {
      const int si = A.sample_idx ? A.sample_idx[s0 + s] : s0 + s;
      v = A.sum[(size_t)si * A.cols + p];
      if (HAAR && A.use_tilted) t = A.tilted[(size_t)si * A.cols + p];
    }
    lsum[batch_tile_word(p, s, S)] = v;
    if (HAAR && A.use_tilted) ltil[batch_tile_word(p, s, S)] = t;
  }
  __syncthreads();
  const int s = threadIdx.x % S;
  const int fsub = threadIdx.x / S, fpar = BATCH_THREADS / S;
  const bool valid = s0 + s < A.n_samples;
  float nf = 1.f;
  if (HAAR && A.normalized && valid) nf = A.normfactor[A.sample_idx ? A.sample_idx[s0 + s] : s0 + s];
  const int f0 = A.feat_begin + blockIdx.y * A.feats_per_block;
  const int f1 = min(f0 + A.feats_per_block, A.feat_end);
  const char* tile = reinterpret_cast<const char*>(lds);
  const int s4 = s * 4;
  auto at = [&](int entry) { return *reinterpret_cast<const int32_t*>(tile + (entry ^ s4)); };
  if (HAAR) {
    // software pipeline: the next feature's record (a 64-byte per-lane global load) is in flight while this one is evaluated
    const HaarFeatDev* feats = reinterpret_cast<const HaarFeatDev*>(A.feats);
    const float y1 = refined_rcp(nf);  // the sample's norm factor does not change from feature to feature
    HaarFeatDev F = feats[min(f0 + fsub, f1 - 1)];
#pragma unroll 2
    for (int f = f0 + fsub; f < f1; f += fpar) {
      const HaarFeatDev N = feats[min(f + fpar, f1 - 1)];
      float ret = F.w[0] * (float)(at(F.p[0][0]) - at(F.p[0][1]) - at(F.p[0][2]) + at(F.p[0][3])) +
                  F.w[1] * (float)(at(F.p[1][0]) - at(F.p[1][1]) - at(F.p[1][2]) + at(F.p[1][3]));
      if (F.w[2] != 0.0f) ret += F.w[2] * (float)(at(F.p[2][0]) - at(F.p[2][1]) - at(F.p[2][2]) + at(F.p[2][3]));
      const float val = A.normalized ? (nf == 0.0f ? 0.0f : div_by_refined(ret, nf, y1)) : ret;
      if (valid && (!A.debug_nostore || val == 12345.678f)) A.out[(size_t)(f - A.feat_begin) * A.out_pitch + s0 + s] = val;
      F = N;
    }
  } else {
    const LbpFeatDev* feats = reinterpret_cast<const LbpFeatDev*>(A.feats);
    for (int f = f0 + fsub; f < f1; f += fpar) {
      const LbpFeatDev F = feats[f];
      int p[16];
#pragma unroll
      for (int j = 0; j < 16; j++) p[j] = at(F.p[j]);
      const int c = p[5] - p[6] - p[9] + p[10];
      const int code = (p[0] - p[1] - p[4] + p[5] >= c ? 128 : 0) | (p[1] - p[2] - p[5] + p[6] >= c ? 64 : 0) |
                       (p[2] - p[3] - p[6] + p[7] >= c ? 32 : 0) | (p[6] - p[7] - p[10] + p[11] >= c ? 16 : 0) |
                       (p[10] - p[11] - p[14] + p[15] >= c ? 8 : 0) | (p[9] - p[10] - p[13] + p[14] >= c ? 4 : 0) |
                       (p[8] - p[9] - p[12] + p[13] >= c ? 2 : 0) | (p[4] - p[5] - p[8] + p[9] >= c ? 1 : 0);
      const float val = (float)code;
      if (valid && (!A.debug_nostore || val == 12345.678f)) A.out[(size_t)(f - A.feat_begin) * A.out_pitch + s0 + s] = val;
    }
  }
}

// Wide variant (S == 64, no tilted tile; the 24x24 BASIC / CORE / LBP shapes): a block owns a tile of 64 samples (one
// block per CU, 160 KB of LDS) and a WAVEFRONT evaluates one feature for the 64 samples, so the feature record is
// wave-uniform and travels through scalar loads. In the narrow kernel above every lane fetches its half-wavefront's
// 64-byte record with vector loads: 64 lanes x 52 bytes through the texture-address path cost ~64 cycles per
// iteration and CU, more than the arithmetic (~25) and the LDS reads (~18) together.
__device__ __forceinline__ HaarFeatDev load_feat(const HaarFeatDev __attribute__((address_space(4)))* p) {
  HaarFeatDev o;
  const int __attribute__((address_space(4)))* w = (const int __attribute__((address_space(4)))*)p;
  int* d = reinterpret_cast<int*>(&o);
#pragma unroll
  for (int i = 0; i < 16; i++) d[i] = w[i];
  return o;
}
__device__ __forceinline__ LbpFeatDev load_feat(const LbpFeatDev __attribute__((address_space(4)))* p) {
  LbpFeatDev o;
  const int __attribute__((address_space(4)))* w = (const int __attribute__((address_space(4)))*)p;
#pragma unroll
  for (int i = 0; i < 16; i++) o.p[i] = w[i];
  return o;
}

template <bool HAAR>
__global__ __launch_bounds__(BATCH_THREADS) void k_eval_batch_wide(BatchArgs A) {
  extern __shared__ int32_t lds[];  // [cols][64], swizzled like the narrow tile (S = 64: word p * 64 + (s ^ (p & 31)))
  constexpr int S = 64;
  // Blocks b and b + 8 run on the same XCD (round-robin dispatch): give each XCD a contiguous run of sample tiles, so that
  // the 256-byte pieces its CUs write into a feature row are neighbours in that XCD's L2 (placement affects speed only).
  const int tiles_per_xcd = (A.n_tiles + 7) >> 3;
  const int tile = A.xcd_tiles ? (int)(blockIdx.x & 7) * tiles_per_xcd + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  if (tile >= A.n_tiles) return;
  const int s0 = tile * S;
  {
    // Staging: wavefront w copies samples w, w + 16, ... (4 of the 64); its lanes walk the sample's entries, 8 loads in
    // flight per lane before the first LDS store (the tile is the first thing the block touches: nothing else hides
    // the latency of these loads, and one CU runs one block).
    const int lane = threadIdx.x & 63, w0 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int UNROLL = 8;
    for (int s = w0; s < S; s += BATCH_THREADS / 64) {
      const bool in = s0 + s < A.n_samples;  // wave-uniform
      const int si = in ? (A.sample_idx ? A.sample_idx[s0 + s] : s0 + s) : 0;
      const int32_t* src = A.sum + (size_t)si * A.cols;
      for (int p0 = 0; p0 < A.cols; p0 += 64 * UNROLL) {
        int v[UNROLL];
#pragma unroll
        for (int k = 0; k < UNROLL; k++) {
          const int p = p0 + k * 64 + lane;
          v[k] = 0;
          if (in && p < A.cols) v[k] = src[p];
        }
#pragma unroll
        for (int k = 0; k < UNROLL; k++) {
          const int p = p0 + k * 64 + lane;
          if (p < A.cols) lds[p * S + (s ^ (p & 31))] = v[k];
        }
      }
    }
  }
  __syncthreads();
  const int s = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool valid = s0 + s < A.n_samples;
  float nf = 1.f;
  if (HAAR && A.normalized && valid) nf = A.normfactor[A.sample_idx ? A.sample_idx[s0 + s] : s0 + s];
  const int f0 = A.feat_begin + blockIdx.y * A.feats_per_block;
  const int f1 = min(f0 + A.feats_per_block, A.feat_end);
  // A corner read is ONE xor: LDS address = (tile base + s * 4) ^ entry. That equals base + ((s * 4) ^ entry) because the
  // tile starts at LDS address 0 (this kernel has no static shared memory in front of its dynamic tile; checked below:
  // any base with its low 18 bits clear would do, the tile spans less than 2^18 bytes).
  const unsigned tile_base = (unsigned)(unsigned long long)(const __attribute__((address_space(3))) char*)lds;
  if (tile_base & 0x3FFFFu) __builtin_trap();
  const unsigned s4 = tile_base + (unsigned)s * 4u;
  auto at = [&](int entry) { return *(const __attribute__((address_space(3))) int32_t*)(unsigned long long)(s4 ^ (unsigned)entry); };
  constexpr int WAVES = BATCH_THREADS / 64;
  float* out = A.out + s0 + s;
  if (HAAR) {
    const HaarFeatDev __attribute__((address_space(4)))* feats = (const HaarFeatDev __attribute__((address_space(4)))*)A.feats;
    const float y1 = refined_rcp(nf);
    auto eval = [&](const HaarFeatDev& F, int f) {
      if (A.debug_nostore == 2) {  // timing experiment: the store stream alone
        if (valid) out[(size_t)(f - A.feat_begin) * A.out_pitch] = F.w[0];
        return;
      }
      if (A.debug_nostore == 3) {  // timing experiment: the store stream alone in 512-byte pieces (every second tile writes two)
        if (!(tile & 1)) {
          if (valid) out[(size_t)(f - A.feat_begin) * A.out_pitch] = F.w[0];
          if (s0 + 64 + s < A.n_samples) out[(size_t)(f - A.feat_begin) * A.out_pitch + 64] = F.w[0];
        }
        return;
      }
      float ret = F.w[0] * (float)(at(F.p[0][0]) - at(F.p[0][1]) - at(F.p[0][2]) + at(F.p[0][3])) +
                  F.w[1] * (float)(at(F.p[1][0]) - at(F.p[1][1]) - at(F.p[1][2]) + at(F.p[1][3]));
      if (F.w[2] != 0.0f) ret += F.w[2] * (float)(at(F.p[2][0]) - at(F.p[2][1]) - at(F.p[2][2]) + at(F.p[2][3]));
      const float val = A.normalized ? (nf == 0.0f ? 0.0f : div_by_refined(ret, nf, y1)) : ret;
      if (valid && (A.debug_nostore != 1 || val == 12345.678f)) out[(size_t)(f - A.feat_begin) * A.out_pitch] = val;
    };
    // Two features per trip: their records (scalar loads, wave-uniform) are requested together. Scalar loads and LDS reads
    // share one completion counter and scalar data may return out of order, so a wavefront cannot wait for an LDS read
    // while a scalar load is in flight without waiting for that load too: a record fetch cannot be overlapped with the
    // evaluation of the previous feature, only amortised over more features.
    // (Round 3 tried requesting the NEXT trip's records before this trip's LDS reads, so that their latency runs under the
    // LDS round trip: the arithmetic alone got 10 % faster (5.9 -> 5.3 ms without the stores), the kernel did not
    // (6.57 vs 6.43 ms): the store stream alone takes 5.8 ms and is what bounds it. Not kept.)
    for (int f = f0 + wave; f < f1; f += 2 * WAVES) {
      const bool two = f + WAVES < f1;
      const HaarFeatDev Fa = load_feat(feats + f), Fb = load_feat(feats + (two ? f + WAVES : f));
      eval(Fa, f);
      if (two) eval(Fb, f + WAVES);
    }
  } else {
    const LbpFeatDev __attribute__((address_space(4)))* feats = (const LbpFeatDev __attribute__((address_space(4)))*)A.feats;
    auto eval = [&](const LbpFeatDev& F, int f) {
      int p[16];
#pragma unroll
      for (int j = 0; j < 16; j++) p[j] = at(F.p[j]);
      const int c = p[5] - p[6] - p[9] + p[10];
      const int code = (p[0] - p[1] - p[4] + p[5] >= c ? 128 : 0) | (p[1] - p[2] - p[5] + p[6] >= c ? 64 : 0) |
                       (p[2] - p[3] - p[6] + p[7] >= c ? 32 : 0) | (p[6] - p[7] - p[10] + p[11] >= c ? 16 : 0) |
                       (p[10] - p[11] - p[14] + p[15] >= c ? 8 : 0) | (p[9] - p[10] - p[13] + p[14] >= c ? 4 : 0) |
                       (p[8] - p[9] - p[12] + p[13] >= c ? 2 : 0) | (p[4] - p[5] - p[8] + p[9] >= c ? 1 : 0);
      const float val = (float)code;
      if (valid && (!A.debug_nostore || val == 12345.678f)) out[(size_t)(f - A.feat_begin) * A.out_pitch] = val;
    };
    for (int f = f0 + wave; f < f1; f += 2 * WAVES) {
      const bool two = f + WAVES < f1;
      const LbpFeatDev Fa = load_feat(feats + f), Fb = load_feat(feats + (two ? f + WAVES : f));
      eval(Fa, f);
      if (two) eval(Fb, f + WAVES);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Training-side stump-cascade predict: one thread per sample (boost.cpp:461-477, o_cvcascadeboosttree.cpp:16-39).
// ------------------------------------------------------------------------------------------------
struct PredictArgs {
  const int32_t* sum;
  const int32_t* tilted;
  const float* normfactor;
  const int32_t* sample_idx;
  int n_samples, cols;
  int nstages;
  const int* stage_ntrees;
  const float* stage_thr;  // threshold - CV_THRESHOLD_EPS
  const void* feats;       // per stump feature, fastRect offsets
  const float* stump_thr;
  const float* stump_left;
  const float* stump_right;
  const int* subsets;      // 8 words per stump / node (LBP)
  // general trees (max_nodes_per_tree > 1): feats / stump_thr / subsets are then indexed by NODE
  int trees;
  const int* tree_root;
  const int* tree_leaf0;
  const int* node_left;
  const int* node_right;
  const float* leaves;
  uint8_t* out;
};

template <bool HAAR>
__global__ __launch_bounds__(64) void k_predict(PredictArgs A) {
  const int s = blockIdx.x * 64 + threadIdx.x;
  if (s >= A.n_samples) return;
  const int si = A.sample_idx ? A.sample_idx[s] : s;
  const int32_t* img = A.sum + (size_t)si * A.cols;
  const int32_t* timg = A.tilted ? A.tilted + (size_t)si * A.cols : img;
  const float nf = HAAR ? A.normfactor[si] : 1.f;
  int k = 0;
  uint8_t pass = 1;
  for (int st = 0; st < A.nstages && pass; st++) {
    double acc = 0;
    const int nt = A.stage_ntrees[st];
    for (int i = 0; i < nt; i++, k++) {
      if (A.trees) {  // CvCascadeBoostTree::predict on a general tree: ordered "<= goes left", categorical "bit set goes left"
        int idx = 0;
        const int root = A.tree_root[k];
        do {
          const int n = root + idx;
          bool go_left;
          if (HAAR) {
            const HaarFeatDev F = reinterpret_cast<const HaarFeatDev*>(A.feats)[n];
            const int32_t* b = F.tilted ? timg : img;
            float ret = F.w[0] * (float)(b[F.p[0][0]] - b[F.p[0][1]] - b[F.p[0][2]] + b[F.p[0][3]]) +
                        F.w[1] * (float)(b[F.p[1][0]] - b[F.p[1][1]] - b[F.p[1][2]] + b[F.p[1][3]]);
            if (F.w[2] != 0.0f) ret += F.w[2] * (float)(b[F.p[2][0]] - b[F.p[2][1]] - b[F.p[2][2]] + b[F.p[2][3]]);
            const float val = nf == 0.0f ? 0.0f : ret / nf;
            go_left = val <= A.stump_thr[n];
          } else {
            const LbpFeatDev F = reinterpret_cast<const LbpFeatDev*>(A.feats)[n];
            int p[16];
#pragma unroll
            for (int j = 0; j < 16; j++) p[j] = img[F.p[j]];
            const int c = p[5] - p[6] - p[9] + p[10];
            const int code = (p[0] - p[1] - p[4] + p[5] >= c ? 128 : 0) | (p[1] - p[2] - p[5] + p[6] >= c ? 64 : 0) |
                             (p[2] - p[3] - p[6] + p[7] >= c ? 32 : 0) | (p[6] - p[7] - p[10] + p[11] >= c ? 16 : 0) |
                             (p[10] - p[11] - p[14] + p[15] >= c ? 8 : 0) | (p[9] - p[10] - p[13] + p[14] >= c ? 4 : 0) |
                             (p[8] - p[9] - p[12] + p[13] >= c ? 2 : 0) | (p[4] - p[5] - p[8] + p[9] >= c ? 1 : 0);
            go_left = (A.subsets[(size_t)n * 8 + (code >> 5)] & (1 << (code & 31))) != 0;
          }
          idx = go_left ? A.node_left[n] : A.node_right[n];
        } while (idx > 0);
        acc += (double)A.leaves[A.tree_leaf0[k] - idx];
        continue;
      }
      if (HAAR) {
        const HaarFeatDev F = reinterpret_cast<const HaarFeatDev*>(A.feats)[k];
        const int32_t* b = F.tilted ? timg : img;
        float ret = F.w[0] * (float)(b[F.p[0][0]] - b[F.p[0][1]] - b[F.p[0][2]] + b[F.p[0][3]]) +
                    F.w[1] * (float)(b[F.p[1][0]] - b[F.p[1][1]] - b[F.p[1][2]] + b[F.p[1][3]]);
        if (F.w[2] != 0.0f) ret += F.w[2] * (float)(b[F.p[2][0]] - b[F.p[2][1]] - b[F.p[2][2]] + b[F.p[2][3]]);
        const float val = nf == 0.0f ? 0.0f : ret / nf;
        acc += (double)(val <= A.stump_thr[k] ? A.stump_left[k] : A.stump_right[k]);
      } else {
        const LbpFeatDev F = reinterpret_cast<const LbpFeatDev*>(A.feats)[k];
        int p[16];
#pragma unroll
        for (int j = 0; j < 16; j++) p[j] = img[F.p[j]];
        const int c = p[5] - p[6] - p[9] + p[10];
        const int code = (p[0] - p[1] - p[4] + p[5] >= c ? 128 : 0) | (p[1] - p[2] - p[5] + p[6] >= c ? 64 : 0) |
                         (p[2] - p[3] - p[6] + p[7] >= c ? 32 : 0) | (p[6] - p[7] - p[10] + p[11] >= c ? 16 : 0) |
                         (p[10] - p[11] - p[14] + p[15] >= c ? 8 : 0) | (p[9] - p[10] - p[13] + p[14] >= c ? 4 : 0) |
                         (p[8] - p[9] - p[12] + p[13] >= c ? 2 : 0) | (p[4] - p[5] - p[8] + p[9] >= c ? 1 : 0);
        const int word = A.subsets[(size_t)k * 8 + (code >> 5)];
        acc += (double)((word & (1 << (code & 31))) ? A.stump_left[k] : A.stump_right[k]);
      }
    }
    if (acc < (double)A.stage_thr[st]) pass = 0;
  }
  A.out[s] = pass;
}

// operator()(feature_idx[k], si) for a list of catalog features and one stored sample: one thread per list entry, the
// sample's integral row read straight from global memory (2.5 KB, cache resident after the first touch).
template <bool HAAR>
__global__ __launch_bounds__(256) void k_eval_list(const void* __restrict__ feats, const int32_t* __restrict__ list, int n,
                                                   const int32_t* __restrict__ sum_row, const int32_t* __restrict__ tilted_row,
                                                   const float* __restrict__ nf_of_sample, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (HAAR) {
    const float nf = *nf_of_sample;
    const HaarFeatDev F = reinterpret_cast<const HaarFeatDev*>(feats)[list[i]];
    const int32_t* b = F.tilted ? tilted_row : sum_row;
    float ret = F.w[0] * (float)(b[F.p[0][0]] - b[F.p[0][1]] - b[F.p[0][2]] + b[F.p[0][3]]) +
                F.w[1] * (float)(b[F.p[1][0]] - b[F.p[1][1]] - b[F.p[1][2]] + b[F.p[1][3]]);
    if (F.w[2] != 0.0f) ret += F.w[2] * (float)(b[F.p[2][0]] - b[F.p[2][1]] - b[F.p[2][2]] + b[F.p[2][3]]);
    out[i] = nf == 0.0f ? 0.0f : ret / nf;  // haarfeatures.h:108-112
  } else {
    const LbpFeatDev F = reinterpret_cast<const LbpFeatDev*>(feats)[list[i]];
    int p[16];
#pragma unroll
    for (int j = 0; j < 16; j++) p[j] = sum_row[F.p[j]];
    const int c = p[5] - p[6] - p[9] + p[10];
    const int code = (p[0] - p[1] - p[4] + p[5] >= c ? 128 : 0) | (p[1] - p[2] - p[5] + p[6] >= c ? 64 : 0) |
                     (p[2] - p[3] - p[6] + p[7] >= c ? 32 : 0) | (p[6] - p[7] - p[10] + p[11] >= c ? 16 : 0) |
                     (p[10] - p[11] - p[14] + p[15] >= c ? 8 : 0) | (p[9] - p[10] - p[13] + p[14] >= c ? 4 : 0) |
                     (p[8] - p[9] - p[12] + p[13] >= c ? 2 : 0) | (p[4] - p[5] - p[8] + p[9] >= c ? 1 : 0);
    out[i] = (float)code;
  }
}

// Feature::calc on caller-held integral rows: one thread per (feature, row).
__global__ void k_feature_calc_rows(const HaarFeatDev* __restrict__ feats, int n_feats, const int32_t* __restrict__ sum,
                                    const int32_t* __restrict__ tilted, int n_rows, int row_len, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_feats * n_rows) return;
  const int f = i / n_rows, r = i - f * n_rows;
  const HaarFeatDev F = feats[f];
  const int32_t* b = (F.tilted ? tilted : sum) + (size_t)r * row_len;
  float ret = F.w[0] * (float)(b[F.p[0][0]] - b[F.p[0][1]] - b[F.p[0][2]] + b[F.p[0][3]]) +
              F.w[1] * (float)(b[F.p[1][0]] - b[F.p[1][1]] - b[F.p[1][2]] + b[F.p[1][3]]);
  if (F.w[2] != 0.0f) ret += F.w[2] * (float)(b[F.p[2][0]] - b[F.p[2][1]] - b[F.p[2][2]] + b[F.p[2][3]]);
  out[i] = ret;
}

// Byte offset of integral entry p in the batch kernel's swizzled LDS tile (see k_eval_batch), plus a base in bytes.
static inline int batch_entry(int p, int S, int base_bytes) { return (p * S + (p & (std::min(S, 32) - 1))) * 4 + base_bytes; }

// batch_S > 0: offsets for k_eval_batch (swizzled tile of batch_S samples; a tilted feature reads the tilted tile
// `tilted_base` bytes behind the sum tile); batch_S == 0: plain entry offsets (fastRect, row stride `step`).
static void haar_to_dev(const HaarFeature& f, int step, HaarFeatDev& d, int batch_S = 0, int tilted_base = 0) {
  std::memset(&d, 0, sizeof(d));
  d.tilted = f.tilted;
  for (int j = 0; j < 3; j++) d.w[j] = f.w[j];
  for (int j = 0; j < 3; j++) {
    if (f.w[j] == 0.0f) break;  // offsets stay 0 from the first zero weight on (haarfeatures.cpp:292-308)
    const int x = f.r[j][0], y = f.r[j][1], w = f.r[j][2], h = f.r[j][3];
    if (!f.tilted) {  // CV_SUM_OFFSETS, traincascade_features.h:40-50
      d.p[j][0] = x + step * y;
      d.p[j][1] = x + w + step * y;
      d.p[j][2] = x + step * (y + h);
      d.p[j][3] = x + w + step * (y + h);
    } else {  // CV_TILTED_OFFSETS, traincascade_features.h:54-63
      d.p[j][0] = x + step * y;
      d.p[j][1] = x - h + step * (y + h);
      d.p[j][2] = x + w + step * (y + w);
      d.p[j][3] = x + w - h + step * (y + w + h);
    }
    if (batch_S > 0)
      for (int k = 0; k < 4; k++) d.p[j][k] = batch_entry(d.p[j][k], batch_S, f.tilted ? tilted_base : 0);
  }
  if (batch_S > 0)  // unused rectangles read entry 0 of their tile (value times weight 0 upstream; never added here)
    for (int j = 0; j < 3; j++)
      if (f.w[j] == 0.0f)
        for (int k = 0; k < 4; k++) d.p[j][k] = batch_entry(0, batch_S, f.tilted ? tilted_base : 0);
}

static void lbp_to_dev(const int32_t* r, int step, LbpFeatDev& d, int batch_S = 0) {  // lbpfeatures.cpp:53-63
  for (int rr = 0; rr < 4; rr++)
    for (int cc = 0; cc < 4; cc++) {
      const int p = (r[0] + cc * r[2]) + step * (r[1] + rr * r[3]);
      d.p[4 * rr + cc] = batch_S > 0 ? batch_entry(p, batch_S, 0) : p;
    }
}

}  // namespace ccamd

using namespace ccamd;

namespace ccamd {

cc_status eval_device(cc_evaluator* e) {
  int n = 0;
  hipError_t err = hipGetDeviceCount(&n);
  if (err != hipSuccess || n <= 0)
    return set_error(CC_ERR_NO_DEVICE, "no usable HIP device (%s); this library has no CPU fallback",
                     err != hipSuccess ? hipGetErrorString(err) : "device count is 0");
  if (e->device < 0 || e->device >= n) return set_error(CC_ERR_INVALID_ARG, "device %d out of range (devices: %d)", e->device, n);
  CC_HIP(hipSetDevice(e->device));
  return CC_OK;
}

cc_status launch_batch(cc_evaluator* e, bool haar, const void* feats, int fb, int fe, const int32_t* d_idx, int ns,
                       float* d_out_ptr, int normalized, size_t out_pitch) {
  BatchArgs A;
  A.sum = e->d_sum.p;
  A.tilted = e->use_tilted ? e->d_tilted.p : nullptr;
  A.normfactor = e->d_nf.p;
  A.sample_idx = d_idx;
  A.n_samples = ns;
  A.cols = e->cols;
  A.S = e->S;
  A.feat_begin = fb;
  A.feat_end = fe;
  A.feats = feats;
  A.out = d_out_ptr;
  A.out_pitch = out_pitch ? out_pitch : (size_t)ns;
  A.normalized = normalized;
  A.debug_nostore = std::getenv("CCAMD_DEBUG_EVAL_NOSTORE") ? std::max(1, std::atoi(std::getenv("CCAMD_DEBUG_EVAL_NOSTORE"))) : 0;
  A.use_tilted = e->use_tilted ? 1 : 0;
  const int nfe = fe - fb;
  const int tiles = (ns + e->S - 1) / e->S;
  // Feature chunks: a block stages its sample tile once per chunk and then walks the chunk's features. Enough chunks for
  // about `rounds` rounds of blocks over the chip (balance), each amortising its tile load over >= 2048 features.
  // Measured at 162 336 x 20 000 (ms for the whole matrix): 2 chunks per launch 7.9, 5: 7.2, 16: 6.6, 32: 7.0, 64: 8.0.
  const int per_cu = e->S == 64 ? 1 : 2, rounds = 20;
  int chunks = (256 * per_cu * rounds + tiles - 1) / std::max(tiles, 1);
  chunks = std::max(1, std::min(chunks, (nfe + 2047) / 2048));
  if (const char* v = std::getenv("CCAMD_EVAL_CHUNKS")) chunks = std::max(1, std::min(std::atoi(v), nfe));  // tuning
  A.feats_per_block = (nfe + chunks - 1) / chunks;
  chunks = (nfe + A.feats_per_block - 1) / A.feats_per_block;
  const size_t lds = (size_t)e->cols * e->S * 4 * (haar && e->use_tilted ? 2 : 1);
  (void)hipEventRecord(e->ev_a, e->stream);
  A.n_tiles = tiles;
  A.xcd_tiles = std::getenv("CCAMD_EVAL_NO_XCD_TILES") ? 0 : 1;
  if (e->S == 64) {  // wide tile (cc_eval_create chose it): one feature per wavefront, scalar record loads
    const int gx = A.xcd_tiles ? ((tiles + 7) / 8) * 8 : tiles;
    if (haar)
      hipLaunchKernelGGL(k_eval_batch_wide<true>, dim3(gx, chunks), dim3(BATCH_THREADS), lds, e->stream, A);
    else
      hipLaunchKernelGGL(k_eval_batch_wide<false>, dim3(gx, chunks), dim3(BATCH_THREADS), lds, e->stream, A);
  } else if (haar)
    hipLaunchKernelGGL(k_eval_batch<true>, dim3(tiles, chunks), dim3(BATCH_THREADS), lds, e->stream, A);
  else
    hipLaunchKernelGGL(k_eval_batch<false>, dim3(tiles, chunks), dim3(BATCH_THREADS), lds, e->stream, A);
  (void)hipEventRecord(e->ev_b, e->stream);
  CC_HIP(hipGetLastError());
  return CC_OK;
}

static cc_status upload_indices(cc_evaluator* e, const int32_t* sample_idx, int ns, const int32_t** d_idx) {
  *d_idx = nullptr;
  if (!sample_idx) {
    if (ns > e->max_samples) return set_error(CC_ERR_OUT_OF_RANGE, "n_samples %d exceeds max_samples %d", ns, e->max_samples);
    return CC_OK;
  }
  for (int i = 0; i < ns; i++)
    if (sample_idx[i] < 0 || sample_idx[i] >= e->max_samples)
      return set_error(CC_ERR_OUT_OF_RANGE, "sample index %d out of range (max_samples %d)", sample_idx[i], e->max_samples);
  CC_HIP(e->d_idx.ensure((size_t)ns));
  CC_HIP(hipMemcpyAsync(e->d_idx.p, sample_idx, (size_t)ns * 4, hipMemcpyHostToDevice, e->stream));
  *d_idx = e->d_idx.p;
  return CC_OK;
}

}  // namespace ccamd

namespace ccamd {
cc_status flush_pending_images(cc_evaluator* e) {
  const int n = (int)e->pend_idx.size();
  if (n == 0) return CC_OK;
  const size_t px = (size_t)e->W * e->H;
  // upload in sample order so that consecutive indices become one launch
  std::vector<int> order((size_t)n);
  for (int i = 0; i < n; i++) order[(size_t)i] = i;
  std::sort(order.begin(), order.end(), [&](int a, int b) { return e->pend_idx[(size_t)a] < e->pend_idx[(size_t)b]; });
  CC_HIP(e->pin_in.ensure((size_t)n * px));
  uint8_t* host = static_cast<uint8_t*>(e->pin_in.p);
  for (int i = 0; i < n; i++) std::memcpy(host + (size_t)i * px, e->pend_px.data() + (size_t)order[(size_t)i] * px, px);
  CC_HIP(e->d_imgs.ensure((size_t)n * px));
  CC_HIP(hipMemcpyAsync(e->d_imgs.p, host, (size_t)n * px, hipMemcpyHostToDevice, e->stream));
  const size_t lds = (size_t)e->H * (e->W + 1) * 4;
  for (int i = 0; i < n;) {
    int j = i + 1;
    while (j < n && e->pend_idx[(size_t)order[(size_t)j]] == e->pend_idx[(size_t)order[(size_t)j - 1]] + 1) j++;
    hipLaunchKernelGGL(k_set_images, dim3(j - i), dim3(64), lds, e->stream, e->d_imgs.p + (size_t)i * px, e->W, e->H,
                       e->pend_idx[(size_t)order[(size_t)i]], e->d_sum.p, e->use_tilted ? e->d_tilted.p : nullptr, e->d_nf.p,
                       e->type == CC_FEATURE_HAAR ? 1 : 0);
    i = j;
  }
  CC_HIP(hipGetLastError());
  CC_HIP(hipStreamSynchronize(e->stream));
  for (int32_t idx : e->pend_idx) e->pend_slot[(size_t)idx] = -1;
  e->pend_idx.clear();
  e->pend_px.clear();
  e->pend_n.store(0, std::memory_order_release);
  return CC_OK;
}

// The integral(s) and the norm factor of one W x H window on the host, entry for entry what k_set_images writes
// (haarfeatures.cpp:100-114, features.cpp:13-25, lbpfeatures.cpp:22-28): sum(y + 1, x + 1) = pixels above and to the left
// incl.; tilted(Y, X) = pixels of rows y < Y within |x - (X - 1)| <= Y - y - 1; norm factor from the exact integer sums over
// normrect (1, 1, W - 2, H - 2). Integer arithmetic throughout, one correctly rounded double sqrt at the end.
static void host_window_integrals(const cc_evaluator* e, const uint8_t* px, std::vector<int32_t>& sum, std::vector<int32_t>& tilted, float& nf) {
  const int W = e->W, H = e->H, sw = W + 1;
  sum.assign((size_t)e->cols, 0);
  for (int y = 0; y < H; y++) {
    int acc = 0;
    for (int x = 0; x < W; x++) {
      acc += px[y * W + x];
      sum[(size_t)(y + 1) * sw + x + 1] = sum[(size_t)y * sw + x + 1] + acc;
    }
  }
  if (e->use_tilted) {
    // row prefix sums r[y][x + 1] = px[y][0..x]; then the kernel's row-by-row clipped spans
    std::vector<int32_t> rp((size_t)H * sw, 0);
    for (int y = 0; y < H; y++)
      for (int x = 0; x < W; x++) rp[(size_t)y * sw + x + 1] = rp[(size_t)y * sw + x] + px[y * W + x];
    tilted.assign((size_t)e->cols, 0);
    for (int Y = 0; Y <= H; Y++)
      for (int X = 0; X <= W; X++) {
        int acc = 0;
        for (int y = 0; y < Y; y++) {
          const int half = Y - y - 1;
          const int x0 = std::max(X - 1 - half, 0), x1 = std::min(X - 1 + half, W - 1);
          if (x1 >= x0) acc += rp[(size_t)y * sw + x1 + 1] - rp[(size_t)y * sw + x0];
        }
        tilted[(size_t)Y * sw + X] = acc;
      }
  }
  nf = 0.f;
  if (e->type == CC_FEATURE_HAAR) {
    long long sq = 0;
    int sm = 0;
    for (int y = 1; y < H - 1; y++)
      for (int x = 1; x < W - 1; x++) {
        const int p = px[y * W + x];
        sm += p;
        sq += p * p;
      }
    const double area = (double)((W - 2) * (H - 2));
    nf = (float)std::sqrt((double)(area * (double)sq - (double)sm * (double)sm));
  }
}

// operator()(fi) on the mirrored window: the expression of k_eval_list / k_eval_batch, operation for operation
// (haarfeatures.h:108-122, lbpfeatures.h:70-83). This translation unit is compiled with -ffp-contract=off, and float
// division on the host is the correctly rounded quotient the device kernels produce.
static inline float host_mirror_value(const cc_evaluator* e, int fi) {
  if (e->type == CC_FEATURE_HAAR) {
    const HaarFeatDev& F = e->h_haar[(size_t)fi];
    const int32_t* b = F.tilted ? e->mirror_tilted.data() : e->mirror_sum.data();
    float ret = F.w[0] * (float)(b[F.p[0][0]] - b[F.p[0][1]] - b[F.p[0][2]] + b[F.p[0][3]]) +
                F.w[1] * (float)(b[F.p[1][0]] - b[F.p[1][1]] - b[F.p[1][2]] + b[F.p[1][3]]);
    if (F.w[2] != 0.0f) ret += F.w[2] * (float)(b[F.p[2][0]] - b[F.p[2][1]] - b[F.p[2][2]] + b[F.p[2][3]]);
    const float nf = e->mirror_nf;
    return nf == 0.0f ? 0.0f : ret / nf;
  }
  const int32_t* s = e->mirror_sum.data();
  const LbpFeatDev& F = e->h_lbp[(size_t)fi];
  int p[16];
  for (int j = 0; j < 16; j++) p[j] = s[F.p[j]];
  const int c = p[5] - p[6] - p[9] + p[10];
  const int code = (p[0] - p[1] - p[4] + p[5] >= c ? 128 : 0) | (p[1] - p[2] - p[5] + p[6] >= c ? 64 : 0) |
                   (p[2] - p[3] - p[6] + p[7] >= c ? 32 : 0) | (p[6] - p[7] - p[10] + p[11] >= c ? 16 : 0) |
                   (p[10] - p[11] - p[14] + p[15] >= c ? 8 : 0) | (p[9] - p[10] - p[13] + p[14] >= c ? 4 : 0) |
                   (p[8] - p[9] - p[12] + p[13] >= c ? 2 : 0) | (p[4] - p[5] - p[8] + p[9] >= c ? 1 : 0);
  return (float)code;
}
static void host_catalog(cc_evaluator* e) {
  std::call_once(e->host_catalog_once, [e]() {
    if (e->type == CC_FEATURE_HAAR) {
      e->h_haar.resize(e->haar.size());
      for (size_t i = 0; i < e->h_haar.size(); i++) haar_to_dev(e->haar[i], e->W + 1, e->h_haar[i]);
    } else {
      e->h_lbp.resize((size_t)e->nfeat);
      for (int i = 0; i < e->nfeat; i++) lbp_to_dev(&e->lbp[(size_t)i * 4], e->W + 1, e->h_lbp[(size_t)i]);
    }
  });
}
}  // namespace ccamd

extern "C" {

cc_status cc_debug_division_check(int device, uint64_t n_pairs, uint64_t seed, uint64_t* mismatches) {
  if (!mismatches) return set_error(CC_ERR_INVALID_ARG, "cc_debug_division_check: null output");
  int n = 0;
  hipError_t err = hipGetDeviceCount(&n);
  if (err != hipSuccess || n <= 0) return set_error(CC_ERR_NO_DEVICE, "no usable HIP device; this library has no CPU fallback");
  if (device < 0 || device >= n) return set_error(CC_ERR_INVALID_ARG, "device %d out of range (devices: %d)", device, n);
  CC_HIP(hipSetDevice(device));
  unsigned long long* d = nullptr;
  CC_HIP(hipMalloc(reinterpret_cast<void**>(&d), sizeof(unsigned long long)));
  OwnStream own;  // not the legacy stream: see copy_sync
  if (hipError_t es = own.create(); es != hipSuccess) {
    (void)hipFree(d);
    return set_error(CC_ERR_HIP, "cc_debug_division_check: %s", hipGetErrorString(es));
  }
  (void)hipMemsetAsync(d, 0, sizeof(unsigned long long), own.s);
  const int threads = 256, blocks = 4096;
  const int per_thread = (int)std::max<uint64_t>(1, std::min<uint64_t>((n_pairs + (uint64_t)threads * blocks - 1) / ((uint64_t)threads * blocks), 1u << 20));
  hipLaunchKernelGGL(k_division_check, dim3(blocks), dim3(threads), 0, own.s, (unsigned long long)seed, per_thread, d);
  unsigned long long h = 0;
  const hipError_t e2 = copy_sync(&h, d, sizeof(h), hipMemcpyDeviceToHost, own.s);
  (void)hipFree(d);
  if (e2 != hipSuccess) return set_error(CC_ERR_HIP, "cc_debug_division_check: %s", hipGetErrorString(e2));
  *mismatches = h;
  return CC_OK;
}

cc_status cc_eval_create(int feature_type, int haar_mode, int win_w, int win_h, int max_samples, int device, cc_evaluator** out) {
  if (!out) return set_error(CC_ERR_INVALID_ARG, "cc_eval_create: null output");
  *out = nullptr;
  if (feature_type == CC_FEATURE_HOG) return set_error(CC_ERR_UNSUPPORTED, "cc_eval_create: HOG is outside the accelerated path");
  if (feature_type != CC_FEATURE_HAAR && feature_type != CC_FEATURE_LBP) return set_error(CC_ERR_INVALID_ARG, "cc_eval_create: unknown feature type %d", feature_type);
  if (max_samples <= 0) return set_error(CC_ERR_INVALID_ARG, "cc_eval_create: maxSampleCount must be > 0");  // features.cpp:75
  if (win_w < 3 || win_h < 3 || win_w > 256 || win_h > 256) return set_error(CC_ERR_INVALID_ARG, "cc_eval_create: window %dx%d out of range", win_w, win_h);
  if (feature_type == CC_FEATURE_HAAR && (haar_mode < CC_HAAR_BASIC || haar_mode > CC_HAAR_ALL))
    return set_error(CC_ERR_INVALID_ARG, "cc_eval_create: unknown Haar mode %d", haar_mode);
  std::unique_ptr<cc_evaluator> e(new cc_evaluator());
  e->type = feature_type;
  e->mode = haar_mode;
  e->W = win_w;
  e->H = win_h;
  e->max_samples = max_samples;
  e->device = device;
  e->cols = (win_w + 1) * (win_h + 1);
  e->use_tilted = feature_type == CC_FEATURE_HAAR && haar_mode == CC_HAAR_ALL;
  cc_status st = eval_device(e.get());
  if (st != CC_OK) return st;
  e->cls.assign((size_t)max_samples, 0.f);
  // samples per LDS tile: largest power of two (<= 32) that keeps the tile within 80 KB (two blocks per CU). With 32
  // samples per tile each 32-lane half of a wavefront is one feature over 32 consecutive samples: LDS reads are
  // conflict-free and every output store is a full 128-byte line.
  const size_t per_sample = (size_t)e->cols * 4 * (e->use_tilted ? 2 : 1);
  const size_t budget = 80 * 1024;
  int S = 32;
  while (S > 1 && per_sample * S > budget) S >>= 1;
  if (per_sample * S > budget) return set_error(CC_ERR_UNSUPPORTED, "cc_eval_create: window %dx%d too large for the LDS tile", win_w, win_h);
  // the wide tile (64 samples, the whole 160 KB of a CU, k_eval_batch_wide) where it fits and no tilted tile is needed
  if (!e->use_tilted && per_sample * 64 <= 160 * 1024 && !std::getenv("CCAMD_EVAL_NARROW_TILE")) S = 64;
  e->S = S;
  if (per_sample * S > 64 * 1024) {
    CC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_eval_batch<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(per_sample * S)));
    CC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_eval_batch<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(per_sample * S)));
    CC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_eval_batch_wide<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(per_sample * S)));
    CC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_eval_batch_wide<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(per_sample * S)));
  }
  CC_HIP(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
  CC_HIP(hipEventCreate(&e->ev_a));
  CC_HIP(hipEventCreate(&e->ev_b));
  CC_HIP(e->d_sum.ensure((size_t)max_samples * e->cols));
  CC_HIP(hipMemsetAsync(e->d_sum.p, 0, (size_t)max_samples * e->cols * 4, e->stream));
  if (e->use_tilted) {
    CC_HIP(e->d_tilted.ensure((size_t)max_samples * e->cols));
    CC_HIP(hipMemsetAsync(e->d_tilted.p, 0, (size_t)max_samples * e->cols * 4, e->stream));
  }
  CC_HIP(e->d_nf.ensure((size_t)max_samples));
  CC_HIP(hipMemsetAsync(e->d_nf.p, 0, (size_t)max_samples * 4, e->stream));
  if (feature_type == CC_FEATURE_HAAR) {
    haar_catalog(win_w, win_h, haar_mode, e->haar);
    e->nfeat = (int)e->haar.size();
    std::vector<HaarFeatDev> dev(e->haar.size());
    for (size_t i = 0; i < dev.size(); i++) haar_to_dev(e->haar[i], win_w + 1, dev[i], e->S, e->use_tilted ? e->cols * e->S * 4 : 0);
    CC_HIP(e->d_haar.ensure(std::max<size_t>(dev.size(), 1)));
    CC_HIP(copy_sync(e->d_haar.p, dev.data(), dev.size() * sizeof(HaarFeatDev), hipMemcpyHostToDevice, e->stream));
  } else {
    lbp_catalog(win_w, win_h, e->lbp);
    e->nfeat = (int)(e->lbp.size() / 4);
    std::vector<LbpFeatDev> dev((size_t)e->nfeat);
    for (int i = 0; i < e->nfeat; i++) lbp_to_dev(&e->lbp[(size_t)i * 4], win_w + 1, dev[i], e->S);
    CC_HIP(e->d_lbp.ensure(std::max<size_t>(dev.size(), 1)));
    CC_HIP(copy_sync(e->d_lbp.p, dev.data(), dev.size() * sizeof(LbpFeatDev), hipMemcpyHostToDevice, e->stream));
  }
  CC_HIP(hipStreamSynchronize(e->stream));
  *out = e.release();
  return CC_OK;
}

void cc_eval_destroy(cc_evaluator* e) {
  if (!e) return;
  (void)hipSetDevice(e->device);
  delete e;
}

int cc_eval_num_features(const cc_evaluator* e) { return e ? e->nfeat : 0; }
int cc_eval_max_cat_count(const cc_evaluator* e) { return e && e->type == CC_FEATURE_LBP ? 256 : 0; }
int cc_eval_feature_size(const cc_evaluator* e) { return e ? 1 : 0; }
const float* cc_eval_labels(const cc_evaluator* e) { return e ? e->cls.data() : nullptr; }

cc_status cc_eval_feature_geometry(const cc_evaluator* e, int fi, int32_t* rects, float* weights, int* tilted) {
  if (!e || !rects) return set_error(CC_ERR_INVALID_ARG, "cc_eval_feature_geometry: null argument");
  if (fi < 0 || fi >= e->nfeat) return set_error(CC_ERR_OUT_OF_RANGE, "cc_eval_feature_geometry: feature %d out of range (%d)", fi, e->nfeat);
  if (e->type == CC_FEATURE_HAAR) {
    std::memcpy(rects, e->haar[fi].r, sizeof(int32_t) * 12);
    if (weights) std::memcpy(weights, e->haar[fi].w, sizeof(float) * 3);
    if (tilted) *tilted = e->haar[fi].tilted;
  } else
    std::memcpy(rects, &e->lbp[(size_t)fi * 4], sizeof(int32_t) * 4);
  return CC_OK;
}

cc_status cc_eval_set_images(cc_evaluator* e, const uint8_t* imgs, int n, int first_idx, const uint8_t* labels) {
  if (!e || (!imgs && n > 0)) return set_error(CC_ERR_INVALID_ARG, "cc_eval_set_images: null argument");
  if (n < 0 || first_idx < 0 || first_idx + n > e->max_samples)  // features.cpp:87: idx < cls.rows
    return set_error(CC_ERR_OUT_OF_RANGE, "cc_eval_set_images: samples [%d, %d) exceed max_samples %d", first_idx, first_idx + n, e->max_samples);
  cc_status st = eval_device(e);
  if (st != CC_OK) return st;
  if (n == 0) return CC_OK;
  std::lock_guard<std::mutex> lk(e->mu);
  st = flush_pending_images(e);  // images set earlier, one at a time, must not land on top of these
  if (st != CC_OK) return st;
  if (e->mirror_idx >= first_idx && e->mirror_idx < first_idx + n) e->mirror_idx = -1;
  const size_t bytes = (size_t)n * e->W * e->H;
  CC_HIP(e->d_imgs.ensure(bytes));
  CC_HIP(hipMemcpyAsync(e->d_imgs.p, imgs, bytes, hipMemcpyHostToDevice, e->stream));
  const size_t lds = (size_t)e->H * (e->W + 1) * 4;
  hipLaunchKernelGGL(k_set_images, dim3(n), dim3(64), lds, e->stream, e->d_imgs.p, e->W, e->H, first_idx, e->d_sum.p,
                     e->use_tilted ? e->d_tilted.p : nullptr, e->d_nf.p, e->type == CC_FEATURE_HAAR ? 1 : 0);
  CC_HIP(hipGetLastError());
  CC_HIP(hipStreamSynchronize(e->stream));
  if (labels)
    for (int i = 0; i < n; i++) e->cls[(size_t)first_idx + i] = (float)labels[i];
  return CC_OK;
}

// setImage for ONE window (haarfeatures.cpp:100-114, lbpfeatures.cpp:22-28) -- the call the trainer's negative-mining loop
// makes for every candidate window (cascadeclassifier.cpp:340-347), 10^5..10^6 times per late stage, each followed by a
// few operator() calls for the same sample. A launch per window would cost more than the window's arithmetic
// (round 3: 15.7 k windows/s through two launches per window), so this call does no device work: it queues the pixels
// (a later image for the same sample replaces the queued one; the queue goes to the device, runs of consecutive samples
// per launch, before anything reads stored samples there) and keeps a host mirror of THIS window's integral(s) and norm
// factor, from which cc_eval_calc / cc_eval_calc_list answer for this sample ("scalar, host-mirror fast
// path"). The mirror's values are bit-identical to the device's (tests/test_gpu_eval.py compares every catalog feature).
cc_status cc_eval_set_image(cc_evaluator* e, const uint8_t* img, size_t row_stride, uint8_t cls_label, int idx) {
  if (!e || !img) return set_error(CC_ERR_INVALID_ARG, "cc_eval_set_image: null argument");
  if (row_stride < (size_t)e->W) return set_error(CC_ERR_INVALID_ARG, "cc_eval_set_image: row stride smaller than the window width");
  if (idx < 0 || idx >= e->max_samples) return set_error(CC_ERR_OUT_OF_RANGE, "cc_eval_set_image: idx %d out of range (%d)", idx, e->max_samples);
  const size_t px = (size_t)e->W * e->H;
  constexpr int kMaxQueued = 4096;
  std::lock_guard<std::mutex> lk(e->mu);
  if ((int)e->pend_idx.size() >= kMaxQueued && (e->pend_slot.empty() || e->pend_slot[(size_t)idx] < 0)) {
    cc_status st = eval_device(e);
    if (st == CC_OK) st = flush_pending_images(e);
    if (st != CC_OK) return st;
  }
  if (e->pend_slot.empty()) e->pend_slot.assign((size_t)e->max_samples, -1);
  int slot = e->pend_slot[(size_t)idx];
  if (slot < 0) {
    slot = (int)e->pend_idx.size();
    e->pend_idx.push_back(idx);
    e->pend_px.resize((size_t)(slot + 1) * px);
    e->pend_slot[(size_t)idx] = slot;
    e->pend_n.store(slot + 1, std::memory_order_release);
  }
  uint8_t* dst = e->pend_px.data() + (size_t)slot * px;
  for (int y = 0; y < e->H; y++) std::memcpy(dst + (size_t)y * e->W, img + (size_t)y * row_stride, (size_t)e->W);
  e->mirror_idx = -1;
  host_window_integrals(e, dst, e->mirror_sum, e->mirror_tilted, e->mirror_nf);
  e->mirror_idx = idx;
  e->cls[(size_t)idx] = (float)cls_label;
  return CC_OK;
}

cc_status cc_eval_calc_batch(cc_evaluator* e, int fi_begin, int fi_end, const int32_t* sample_idx, int n_samples, float* out,
                             int out_on_device) {
  if (!e || !out) return set_error(CC_ERR_INVALID_ARG, "cc_eval_calc_batch: null argument");
  if (fi_begin < 0 || fi_end > e->nfeat || fi_begin > fi_end)
    return set_error(CC_ERR_OUT_OF_RANGE, "cc_eval_calc_batch: features [%d, %d) out of range (%d)", fi_begin, fi_end, e->nfeat);
  if (n_samples < 0) return set_error(CC_ERR_INVALID_ARG, "cc_eval_calc_batch: negative sample count");
  cc_status st = eval_device(e);
  if (st != CC_OK) return st;
  if (fi_begin == fi_end || n_samples == 0) return CC_OK;
  std::lock_guard<std::mutex> lk(e->mu);
  if (cc_status fst = flush_pending_images(e); fst != CC_OK) return fst;  // images set one at a time reach the device first
  const int32_t* d_idx = nullptr;
  st = upload_indices(e, sample_idx, n_samples, &d_idx);
  if (st != CC_OK) return st;
  const bool haar = e->type == CC_FEATURE_HAAR;
  const void* feats = haar ? (const void*)e->d_haar.p : (const void*)e->d_lbp.p;
  const size_t total = (size_t)(fi_end - fi_begin) * n_samples;
  float* dst = out;
  if (!out_on_device) {
    CC_HIP(e->d_out.ensure(total));
    dst = e->d_out.p;
  }
  st = launch_batch(e, haar, feats, fi_begin, fi_end, d_idx, n_samples, dst, 1, 0);
  if (st != CC_OK) return st;
  if (!out_on_device) CC_HIP(hipMemcpyAsync(out, dst, total * 4, hipMemcpyDeviceToHost, e->stream));
  CC_HIP(hipStreamSynchronize(e->stream));
  float ms = 0;
  if (hipEventElapsedTime(&ms, e->ev_a, e->ev_b) == hipSuccess) e->last_ms = ms;
  return CC_OK;
}

cc_status cc_eval_calc_batch_device(cc_evaluator* e, int fi_begin, int fi_end, const int32_t* sample_idx, int n_samples,
                                    float* d_out, size_t pitch) {
  if (!e || !d_out) return set_error(CC_ERR_INVALID_ARG, "cc_eval_calc_batch_device: null argument");
  if (fi_begin < 0 || fi_end > e->nfeat || fi_begin > fi_end)
    return set_error(CC_ERR_OUT_OF_RANGE, "cc_eval_calc_batch_device: features [%d, %d) out of range (%d)", fi_begin, fi_end, e->nfeat);
  if (n_samples < 0) return set_error(CC_ERR_INVALID_ARG, "cc_eval_calc_batch_device: negative sample count");
  if (pitch != 0 && pitch < (size_t)n_samples)
    return set_error(CC_ERR_INVALID_ARG, "cc_eval_calc_batch_device: pitch %zu is smaller than the %d samples of a row", pitch, n_samples);
  cc_status st = eval_device(e);
  if (st != CC_OK) return st;
  if (fi_begin == fi_end || n_samples == 0) return CC_OK;
  std::lock_guard<std::mutex> lk(e->mu);
  if (cc_status fst = flush_pending_images(e); fst != CC_OK) return fst;  // images set one at a time reach the device first
  const int32_t* d_idx = nullptr;
  st = upload_indices(e, sample_idx, n_samples, &d_idx);
  if (st != CC_OK) return st;
  const bool haar = e->type == CC_FEATURE_HAAR;
  st = launch_batch(e, haar, haar ? (const void*)e->d_haar.p : (const void*)e->d_lbp.p, fi_begin, fi_end, d_idx, n_samples, d_out, 1, pitch);
  if (st != CC_OK) return st;
  CC_HIP(hipStreamSynchronize(e->stream));
  float ms = 0;
  if (hipEventElapsedTime(&ms, e->ev_a, e->ev_b) == hipSuccess) e->last_ms = ms;
  return CC_OK;
}

namespace ccamd {
__global__ void k_iota_rows(int* __restrict__ v, size_t total, int n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < total) v[i] = (int)(i % (size_t)n);
}
__global__ void k_narrow_u16(const int* __restrict__ in, unsigned short* __restrict__ out, size_t total) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < total) out[i] = (unsigned short)in[i];
}
}  // namespace ccamd

cc_status cc_eval_calc_batch_sorted(cc_evaluator* e, int fi_begin, int fi_end, int n_samples, float* vals, void* idx, int idx_bytes) {
  if (!e || !idx) return set_error(CC_ERR_INVALID_ARG, "cc_eval_calc_batch_sorted: null argument");
  if (fi_begin < 0 || fi_end > e->nfeat || fi_begin > fi_end)
    return set_error(CC_ERR_OUT_OF_RANGE, "cc_eval_calc_batch_sorted: features [%d, %d) out of range (%d)", fi_begin, fi_end, e->nfeat);
  if (n_samples < 0 || n_samples > e->max_samples) return set_error(CC_ERR_OUT_OF_RANGE, "cc_eval_calc_batch_sorted: n_samples %d out of range", n_samples);
  if (idx_bytes != 2 && idx_bytes != 4) return set_error(CC_ERR_INVALID_ARG, "cc_eval_calc_batch_sorted: idx_bytes must be 2 or 4");
  if (idx_bytes == 2 && n_samples > 65536) return set_error(CC_ERR_INVALID_ARG, "cc_eval_calc_batch_sorted: 16-bit indices need n_samples <= 65536");
  cc_status st = eval_device(e);
  if (st != CC_OK) return st;
  const int nf = fi_end - fi_begin;
  if (nf == 0 || n_samples == 0) return CC_OK;
  std::lock_guard<std::mutex> lk(e->mu);
  if (cc_status fst = flush_pending_images(e); fst != CC_OK) return fst;  // images set one at a time reach the device first
  const bool haar = e->type == CC_FEATURE_HAAR;
  const size_t total = (size_t)nf * n_samples;
  if (total > (size_t)INT32_MAX) return set_error(CC_ERR_INVALID_ARG, "cc_eval_calc_batch_sorted: block too large (%zu values); use smaller feature ranges", total);
  EBuf<float> keys_out;
  EBuf<int> iota, sorted, offsets;
  EBuf<unsigned short> narrow;
  EBuf<char> temp;
  CC_HIP(e->d_out.ensure(total));
  CC_HIP(keys_out.ensure(total));
  CC_HIP(iota.ensure(total));
  CC_HIP(sorted.ensure(total));
  CC_HIP(offsets.ensure((size_t)nf + 1));
  st = launch_batch(e, haar, haar ? (const void*)e->d_haar.p : (const void*)e->d_lbp.p, fi_begin, fi_end, nullptr, n_samples, e->d_out.p, 1, 0);
  if (st != CC_OK) return st;
  bool sorted_in_blocks = false;
  if (n_samples <= sort_rows_block_limit()) {  // one block per row, the row in registers + LDS (cc_split.hip)
    if (sort_rows_block(e->d_out.p, nf, n_samples, keys_out.p, sorted.p, e->stream) == hipSuccess)
      sorted_in_blocks = true;
    else
      (void)hipGetLastError();  // refused LDS request / launch: the device-wide segmented sort below
  }
  if (!sorted_in_blocks) {
    std::vector<int> off((size_t)nf + 1);
    for (int i = 0; i <= nf; i++) off[(size_t)i] = i * n_samples;
    CC_HIP(hipMemcpyAsync(offsets.p, off.data(), off.size() * 4, hipMemcpyHostToDevice, e->stream));
    hipLaunchKernelGGL(k_iota_rows, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, e->stream, iota.p, total, n_samples);
    size_t temp_bytes = 0;
    CC_HIP(hipcub::DeviceSegmentedRadixSort::SortPairs(nullptr, temp_bytes, e->d_out.p, keys_out.p, iota.p, sorted.p, (int)total, nf,
                                                       offsets.p, offsets.p + 1, 0, 32, e->stream));
    CC_HIP(temp.ensure(std::max<size_t>(temp_bytes, 1)));
    // radix sort is stable: equal values keep increasing sample order
    CC_HIP(hipcub::DeviceSegmentedRadixSort::SortPairs(temp.p, temp_bytes, e->d_out.p, keys_out.p, iota.p, sorted.p, (int)total, nf,
                                                       offsets.p, offsets.p + 1, 0, 32, e->stream));
  }
  if (idx_bytes == 2) {
    CC_HIP(narrow.ensure(total));
    hipLaunchKernelGGL(k_narrow_u16, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, e->stream, sorted.p, narrow.p, total);
    CC_HIP(hipMemcpyAsync(idx, narrow.p, total * 2, hipMemcpyDeviceToHost, e->stream));
  } else
    CC_HIP(hipMemcpyAsync(idx, sorted.p, total * 4, hipMemcpyDeviceToHost, e->stream));
  if (vals) CC_HIP(hipMemcpyAsync(vals, e->d_out.p, total * 4, hipMemcpyDeviceToHost, e->stream));
  CC_HIP(hipGetLastError());
  CC_HIP(hipStreamSynchronize(e->stream));
  return CC_OK;
}

cc_status cc_eval_calc(cc_evaluator* e, int fi, int si, float* out) {
  if (!e || !out) return set_error(CC_ERR_INVALID_ARG, "cc_eval_calc: null argument");
  if (si < 0 || si >= e->max_samples) return set_error(CC_ERR_OUT_OF_RANGE, "cc_eval_calc: sample %d out of range (%d)", si, e->max_samples);
  if (si == e->mirror_idx) {  // the window set last by cc_eval_set_image: answered from its host mirror, no launch
    if (fi < 0 || fi >= e->nfeat) return set_error(CC_ERR_OUT_OF_RANGE, "cc_eval_calc: feature %d out of range (%d)", fi, e->nfeat);
    host_catalog(e);
    *out = host_mirror_value(e, fi);
    return CC_OK;
  }
  const int32_t idx = si;
  return cc_eval_calc_batch(e, fi, fi + 1, &idx, 1, out, 0);
}

cc_status cc_eval_calc_list(cc_evaluator* e, const int32_t* feature_idx, int n_feats, int si, float* out) {
  if (!e || (n_feats > 0 && (!feature_idx || !out))) return set_error(CC_ERR_INVALID_ARG, "cc_eval_calc_list: null argument");
  if (n_feats < 0) return set_error(CC_ERR_INVALID_ARG, "cc_eval_calc_list: negative count");
  if (si < 0 || si >= e->max_samples) return set_error(CC_ERR_OUT_OF_RANGE, "cc_eval_calc_list: sample %d out of range (%d)", si, e->max_samples);
  for (int i = 0; i < n_feats; i++)
    if (feature_idx[i] < 0 || feature_idx[i] >= e->nfeat)
      return set_error(CC_ERR_OUT_OF_RANGE, "cc_eval_calc_list: feature %d out of range (%d)", feature_idx[i], e->nfeat);
  if (si == e->mirror_idx) {  // see cc_eval_calc
    host_catalog(e);
    for (int i = 0; i < n_feats; i++) out[i] = host_mirror_value(e, feature_idx[i]);
    return CC_OK;
  }
  cc_status st = eval_device(e);
  if (st != CC_OK) return st;
  if (n_feats == 0) return CC_OK;
  const bool haar = e->type == CC_FEATURE_HAAR;
  std::lock_guard<std::mutex> lk(e->mu);
  st = flush_pending_images(e);
  if (st != CC_OK) return st;
  if (haar && !e->d_haar_plain.p) {  // catalog with plain row offsets (row stride W + 1), built once
    std::vector<HaarFeatDev> dev(e->haar.size());
    for (size_t i = 0; i < dev.size(); i++) haar_to_dev(e->haar[i], e->W + 1, dev[i]);
    CC_HIP(e->d_haar_plain.ensure(std::max<size_t>(dev.size(), 1)));
    CC_HIP(copy_sync(e->d_haar_plain.p, dev.data(), dev.size() * sizeof(HaarFeatDev), hipMemcpyHostToDevice, e->stream));
  }
  if (!haar && !e->d_lbp_plain.p) {
    std::vector<LbpFeatDev> dev((size_t)e->nfeat);
    for (int i = 0; i < e->nfeat; i++) lbp_to_dev(&e->lbp[(size_t)i * 4], e->W + 1, dev[i]);
    CC_HIP(e->d_lbp_plain.ensure(std::max<size_t>(dev.size(), 1)));
    CC_HIP(copy_sync(e->d_lbp_plain.p, dev.data(), dev.size() * sizeof(LbpFeatDev), hipMemcpyHostToDevice, e->stream));
  }
  CC_HIP(e->d_idx.ensure((size_t)n_feats));
  CC_HIP(e->d_out.ensure((size_t)n_feats));
  CC_HIP(hipMemcpyAsync(e->d_idx.p, feature_idx, (size_t)n_feats * 4, hipMemcpyHostToDevice, e->stream));
  const int32_t* row = e->d_sum.p + (size_t)si * e->cols;
  const int32_t* trow = e->use_tilted ? e->d_tilted.p + (size_t)si * e->cols : row;
  if (haar)
    hipLaunchKernelGGL(k_eval_list<true>, dim3((n_feats + 255) / 256), dim3(256), 0, e->stream, (const void*)e->d_haar_plain.p, e->d_idx.p,
                       n_feats, row, trow, e->d_nf.p + si, e->d_out.p);
  else
    hipLaunchKernelGGL(k_eval_list<false>, dim3((n_feats + 255) / 256), dim3(256), 0, e->stream, (const void*)e->d_lbp_plain.p, e->d_idx.p,
                       n_feats, row, trow, e->d_nf.p + si, e->d_out.p);
  CC_HIP(hipGetLastError());
  CC_HIP(hipMemcpyAsync(out, e->d_out.p, (size_t)n_feats * sizeof(float), hipMemcpyDeviceToHost, e->stream));
  CC_HIP(hipStreamSynchronize(e->stream));
  return CC_OK;
}

cc_status cc_eval_calc_custom_haar(cc_evaluator* e, const cc_haar_feature* feats, int n_feats, int normalized,
                                   const int32_t* sample_idx, int n_samples, float* out) {
  if (!e || !feats || !out) return set_error(CC_ERR_INVALID_ARG, "cc_eval_calc_custom_haar: null argument");
  if (e->type != CC_FEATURE_HAAR) return set_error(CC_ERR_INVALID_ARG, "cc_eval_calc_custom_haar: evaluator is not HAAR");
  if (n_feats < 0 || n_samples < 0) return set_error(CC_ERR_INVALID_ARG, "cc_eval_calc_custom_haar: negative count");
  cc_status st = eval_device(e);
  if (st != CC_OK) return st;
  if (n_feats == 0 || n_samples == 0) return CC_OK;
  std::vector<HaarFeatDev> dev((size_t)n_feats);
  for (int i = 0; i < n_feats; i++) {
    HaarFeature f;
    std::memcpy(f.r, feats[i].r, sizeof(f.r));
    std::memcpy(f.w, feats[i].w, sizeof(f.w));
    f.tilted = feats[i].tilted != 0;
    if (f.tilted && !e->use_tilted) return set_error(CC_ERR_INVALID_ARG, "cc_eval_calc_custom_haar: tilted feature on an evaluator without tilted integrals (mode != ALL)");
    for (int j = 0; j < 3; j++) {  // every touched integral entry must lie inside the (W+1)x(H+1) window integral
      if (f.w[j] == 0.0f) break;
      const int x = f.r[j][0], y = f.r[j][1], w = f.r[j][2], h = f.r[j][3];
      bool ok = x >= 0 && y >= 0 && w >= 0 && h >= 0;
      ok = ok && (f.tilted ? (x - h >= 0 && x + w <= e->W && y + w + h <= e->H) : (x + w <= e->W && y + h <= e->H));
      if (!ok) return set_error(CC_ERR_OUT_OF_RANGE, "cc_eval_calc_custom_haar: rect %d of feature %d leaves the window", j, i);
    }
    haar_to_dev(f, e->W + 1, dev[i], e->S, e->use_tilted ? e->cols * e->S * 4 : 0);
  }
  std::lock_guard<std::mutex> lk(e->mu);
  if (cc_status fst = flush_pending_images(e); fst != CC_OK) return fst;  // images set one at a time reach the device first
  const int32_t* d_idx = nullptr;
  st = upload_indices(e, sample_idx, n_samples, &d_idx);
  if (st != CC_OK) return st;
  CC_HIP(e->d_custom.ensure((size_t)n_feats));
  CC_HIP(hipMemcpyAsync(e->d_custom.p, dev.data(), dev.size() * sizeof(HaarFeatDev), hipMemcpyHostToDevice, e->stream));
  const size_t total = (size_t)n_feats * n_samples;
  CC_HIP(e->d_out.ensure(total));
  st = launch_batch(e, true, e->d_custom.p, 0, n_feats, d_idx, n_samples, e->d_out.p, normalized ? 1 : 0, 0);
  if (st != CC_OK) return st;
  CC_HIP(hipMemcpyAsync(out, e->d_out.p, total * 4, hipMemcpyDeviceToHost, e->stream));
  CC_HIP(hipStreamSynchronize(e->stream));
  return CC_OK;
}

cc_status cc_haar_feature_calc(int device, const cc_haar_feature* feats, int n_feats, int step, const int32_t* sum,
                               const int32_t* tilted, int n_rows, int row_len, float* out) {
  if (!feats || !out || n_feats < 0 || n_rows < 0 || row_len < 1 || step < 1) return set_error(CC_ERR_INVALID_ARG, "cc_haar_feature_calc: bad argument");
  int ndev = 0;
  hipError_t err = hipGetDeviceCount(&ndev);
  if (err != hipSuccess || ndev <= 0)
    return set_error(CC_ERR_NO_DEVICE, "no usable HIP device (%s); this library has no CPU fallback",
                     err != hipSuccess ? hipGetErrorString(err) : "device count is 0");
  if (device < 0 || device >= ndev) return set_error(CC_ERR_INVALID_ARG, "device %d out of range (devices: %d)", device, ndev);
  CC_HIP(hipSetDevice(device));
  if (n_feats == 0 || n_rows == 0) return CC_OK;
  std::vector<HaarFeatDev> dev((size_t)n_feats);
  bool need_sum = false, need_tilted = false;
  for (int i = 0; i < n_feats; i++) {
    HaarFeature f;
    std::memcpy(f.r, feats[i].r, sizeof(f.r));
    std::memcpy(f.w, feats[i].w, sizeof(f.w));
    f.tilted = feats[i].tilted != 0;
    haar_to_dev(f, step, dev[i]);
    (f.tilted ? need_tilted : need_sum) = true;
    for (int j = 0; j < 3; j++)
      for (int k = 0; k < 4; k++)
        if (dev[i].p[j][k] < 0 || dev[i].p[j][k] >= row_len)
          return set_error(CC_ERR_OUT_OF_RANGE, "cc_haar_feature_calc: feature %d reads offset %d outside the %d-entry integral", i, dev[i].p[j][k], row_len);
  }
  if ((need_sum && !sum) || (need_tilted && !tilted)) return set_error(CC_ERR_INVALID_ARG, "cc_haar_feature_calc: a needed integral image is NULL");
  EBuf<HaarFeatDev> d_f;
  EBuf<int32_t> d_s, d_t;
  EBuf<float> d_o;
  const size_t nint = (size_t)n_rows * row_len, nout = (size_t)n_feats * n_rows;
  OwnStream own;  // not the legacy stream: see copy_sync
  CC_HIP(own.create());
  CC_HIP(d_f.ensure((size_t)n_feats));
  CC_HIP(copy_sync(d_f.p, dev.data(), dev.size() * sizeof(HaarFeatDev), hipMemcpyHostToDevice, own.s));
  if (sum) {
    CC_HIP(d_s.ensure(nint));
    CC_HIP(copy_sync(d_s.p, sum, nint * 4, hipMemcpyHostToDevice, own.s));
  }
  if (tilted) {
    CC_HIP(d_t.ensure(nint));
    CC_HIP(copy_sync(d_t.p, tilted, nint * 4, hipMemcpyHostToDevice, own.s));
  }
  CC_HIP(d_o.ensure(nout));
  hipLaunchKernelGGL(k_feature_calc_rows, dim3((unsigned)((nout + 255) / 256)), dim3(256), 0, own.s, d_f.p, n_feats, d_s.p, d_t.p,
                     n_rows, row_len, d_o.p);
  CC_HIP(hipGetLastError());
  CC_HIP(copy_sync(out, d_o.p, nout * 4, hipMemcpyDeviceToHost, own.s));
  return CC_OK;
}

cc_status cc_eval_get_sample(cc_evaluator* e, int idx, int32_t* sum, int32_t* tilted, float* normfactor) {
  if (!e) return set_error(CC_ERR_INVALID_ARG, "cc_eval_get_sample: null evaluator");
  if (idx < 0 || idx >= e->max_samples) return set_error(CC_ERR_OUT_OF_RANGE, "cc_eval_get_sample: idx %d out of range", idx);
  cc_status st = eval_device(e);
  if (st != CC_OK) return st;
  std::lock_guard<std::mutex> lk(e->mu);
  st = flush_pending_images(e);  // the device's copy is what this call reports, also for a sample set a moment ago
  if (st != CC_OK) return st;
  if (sum) CC_HIP(copy_sync(sum, e->d_sum.p + (size_t)idx * e->cols, (size_t)e->cols * 4, hipMemcpyDeviceToHost, e->stream));
  if (tilted) {
    if (!e->use_tilted) return set_error(CC_ERR_INVALID_ARG, "cc_eval_get_sample: evaluator keeps no tilted integrals (mode != ALL)");
    CC_HIP(copy_sync(tilted, e->d_tilted.p + (size_t)idx * e->cols, (size_t)e->cols * 4, hipMemcpyDeviceToHost, e->stream));
  }
  if (normfactor) {
    if (e->type != CC_FEATURE_HAAR) return set_error(CC_ERR_INVALID_ARG, "cc_eval_get_sample: LBP evaluator has no norm factor");
    CC_HIP(copy_sync(normfactor, e->d_nf.p + idx, 4, hipMemcpyDeviceToHost, e->stream));
  }
  return CC_OK;
}

cc_status cc_eval_predict_cascade(cc_evaluator* e, const cc_cascade* c, const int32_t* sample_idx, int n_samples, uint8_t* out) {
  if (!e || !c || !out) return set_error(CC_ERR_INVALID_ARG, "cc_eval_predict_cascade: null argument");
  const Cascade& m = c->m;
  if (m.feature_type != e->type || m.win_w != e->W || m.win_h != e->H)
    return set_error(CC_ERR_INVALID_ARG, "cc_eval_predict_cascade: cascade (%s %dx%d) does not match the evaluator (%dx%d)",
                     m.feature_type == CC_FEATURE_HAAR ? "HAAR" : "LBP", m.win_w, m.win_h, e->W, e->H);
  if (m.has_tilted && !e->use_tilted) return set_error(CC_ERR_INVALID_ARG, "cc_eval_predict_cascade: cascade has tilted features but the evaluator keeps no tilted integrals");
  if (n_samples < 0) return set_error(CC_ERR_INVALID_ARG, "cc_eval_predict_cascade: negative sample count");
  cc_status st = eval_device(e);
  if (st != CC_OK) return st;
  if (n_samples == 0) return CC_OK;
  std::lock_guard<std::mutex> lk(e->mu);
  if (cc_status fst = flush_pending_images(e); fst != CC_OK) return fst;  // images set one at a time reach the device first
  const int32_t* d_idx = nullptr;
  st = upload_indices(e, sample_idx, n_samples, &d_idx);
  if (st != CC_OK) return st;
  const bool haar = e->type == CC_FEATURE_HAAR;
  const bool trees = m.max_nodes_per_tree > 1;
  // per-record tables are indexed by stump for stump cascades and by node for general trees
  const std::vector<int32_t>& rec_feature = trees ? m.node_feature : m.stump_feature;
  const std::vector<float>& rec_thr = trees ? m.node_threshold : m.stump_threshold;
  const size_t ns = rec_feature.size();
  EBuf<HaarFeatDev> dh;
  EBuf<LbpFeatDev> dl;
  EBuf<int> d_ntrees, d_sub;
  EBuf<float> d_sthr, d_thr, d_left, d_right;
  std::vector<int> ntrees(m.stage_ntrees.begin(), m.stage_ntrees.end());
  if (haar) {
    std::vector<HaarFeatDev> dev(ns);
    for (size_t i = 0; i < ns; i++) {
      HaarFeature f;
      const int fi = rec_feature[i];
      std::memcpy(f.r, &m.haar_rects[(size_t)fi * 12], sizeof(f.r));
      std::memcpy(f.w, &m.haar_weights[(size_t)fi * 3], sizeof(f.w));
      f.tilted = m.haar_tilted[fi];
      haar_to_dev(f, e->W + 1, dev[i]);
    }
    CC_HIP(dh.ensure(ns));
    CC_HIP(copy_sync(dh.p, dev.data(), ns * sizeof(HaarFeatDev), hipMemcpyHostToDevice, e->stream));
  } else {
    std::vector<LbpFeatDev> dev(ns);
    for (size_t i = 0; i < ns; i++) lbp_to_dev(&m.lbp_rects[(size_t)rec_feature[i] * 4], e->W + 1, dev[i]);
    CC_HIP(dl.ensure(ns));
    CC_HIP(copy_sync(dl.p, dev.data(), ns * sizeof(LbpFeatDev), hipMemcpyHostToDevice, e->stream));
    CC_HIP(d_sub.ensure(ns * 8));
    CC_HIP(copy_sync(d_sub.p, m.node_subset.data(), ns * 8 * 4, hipMemcpyHostToDevice, e->stream));
  }
  CC_HIP(d_ntrees.ensure(ntrees.size()));
  CC_HIP(copy_sync(d_ntrees.p, ntrees.data(), ntrees.size() * 4, hipMemcpyHostToDevice, e->stream));
  CC_HIP(d_sthr.ensure(ntrees.size()));
  CC_HIP(copy_sync(d_sthr.p, m.stage_threshold.data(), ntrees.size() * 4, hipMemcpyHostToDevice, e->stream));
  CC_HIP(d_thr.ensure(ns));
  CC_HIP(copy_sync(d_thr.p, rec_thr.data(), ns * 4, hipMemcpyHostToDevice, e->stream));
  EBuf<int> d_root, d_leaf0, d_nl, d_nr;
  EBuf<float> d_leaves;
  if (!trees) {
    CC_HIP(d_left.ensure(ns));
    CC_HIP(copy_sync(d_left.p, m.stump_left.data(), ns * 4, hipMemcpyHostToDevice, e->stream));
    CC_HIP(d_right.ensure(ns));
    CC_HIP(copy_sync(d_right.p, m.stump_right.data(), ns * 4, hipMemcpyHostToDevice, e->stream));
  } else {
    const size_t nt = m.tree_first_node.size();
    CC_HIP(d_root.ensure(nt));
    CC_HIP(copy_sync(d_root.p, m.tree_first_node.data(), nt * 4, hipMemcpyHostToDevice, e->stream));
    CC_HIP(d_leaf0.ensure(nt));
    CC_HIP(copy_sync(d_leaf0.p, m.tree_first_leaf.data(), nt * 4, hipMemcpyHostToDevice, e->stream));
    CC_HIP(d_nl.ensure(ns));
    CC_HIP(copy_sync(d_nl.p, m.node_left.data(), ns * 4, hipMemcpyHostToDevice, e->stream));
    CC_HIP(d_nr.ensure(ns));
    CC_HIP(copy_sync(d_nr.p, m.node_right.data(), ns * 4, hipMemcpyHostToDevice, e->stream));
    CC_HIP(d_leaves.ensure(m.leaves.size()));
    CC_HIP(copy_sync(d_leaves.p, m.leaves.data(), m.leaves.size() * 4, hipMemcpyHostToDevice, e->stream));
  }
  CC_HIP(e->d_pred.ensure((size_t)n_samples));
  PredictArgs A;
  A.sum = e->d_sum.p;
  A.tilted = e->use_tilted ? e->d_tilted.p : nullptr;
  A.normfactor = e->d_nf.p;
  A.sample_idx = d_idx;
  A.n_samples = n_samples;
  A.cols = e->cols;
  A.nstages = (int)ntrees.size();
  A.stage_ntrees = d_ntrees.p;
  A.stage_thr = d_sthr.p;
  A.feats = haar ? (const void*)dh.p : (const void*)dl.p;
  A.stump_thr = d_thr.p;
  A.stump_left = d_left.p;
  A.stump_right = d_right.p;
  A.subsets = d_sub.p;
  A.trees = trees ? 1 : 0;
  A.tree_root = d_root.p;
  A.tree_leaf0 = d_leaf0.p;
  A.node_left = d_nl.p;
  A.node_right = d_nr.p;
  A.leaves = d_leaves.p;
  A.out = e->d_pred.p;
  if (haar)
    hipLaunchKernelGGL(k_predict<true>, dim3((n_samples + 63) / 64), dim3(64), 0, e->stream, A);
  else
    hipLaunchKernelGGL(k_predict<false>, dim3((n_samples + 63) / 64), dim3(64), 0, e->stream, A);
  CC_HIP(hipGetLastError());
  CC_HIP(hipMemcpyAsync(out, e->d_pred.p, (size_t)n_samples, hipMemcpyDeviceToHost, e->stream));
  CC_HIP(hipStreamSynchronize(e->stream));
  return CC_OK;
}

cc_status cc_eval_last_kernel_ms(cc_evaluator* e, double* ms) {
  if (!e || !ms) return set_error(CC_ERR_INVALID_ARG, "cc_eval_last_kernel_ms: null argument");
  *ms = e->last_ms;
  return CC_OK;
}

}  // extern "C"
