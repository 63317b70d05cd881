// Detection hot path on gfx950: scale pyramid (bit-exact fixed-point bilinear), integral images (sum + wrap-around
// sqsum), sliding-window cascade evaluation with early exit, stage-0 skip-rule filter. Host orchestration at the
// bottom (cc_detector). Replaces cv::CascadeClassifier::detectMultiScale as called by the reference's detection tool
// (tools/detection/Cpp/main.cpp:42-45); behaviour follows SURVEY.md Appendix A.
//
// Data layout in HBM (per frame slot f of a pass; all slabs are sized for the largest pass, at most max_batch frames):
//   pyramid  u8   : scale s at pyr + f*pyr_frame_bytes + img_ofs[s], row pitch pitch8[s] (multiple of 4)
//   integral i32  : sum   at integ + (f*nchan + 0)*int_frame_elems + int_ofs[s], (h+1) rows x pitchI[s] (multiple of 4)
//                   sqsum at integ + (f*nchan + 1)*int_frame_elems + int_ofs[s]   (Haar only; u32 wrap-around). With an even
//                   window size, scales scanned with step 2 keep only what the variance test reads: odd rows, and
//                   of those the odd columns packed (column 2c+1 at c)
//   rej0 mask u64 : bit gx&63 of word mask_ofs[s] + gy*nxw[s] + (gx>>6) = window (gx,gy) was rejected AT STAGE 0
//   candidates    : one global list {frame, scale, gx, gy} + counter; the filtered list adds the output rectangle.
//
// Arithmetic is compiled with -ffp-contract=off: Haar feature values and stage sums follow the CPU operation order
// exactly (float multiply/add without fusion, double stage accumulator), which makes decisions bit-identical.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <sstream>
#include <future>
#include <memory>
#include <thread>

#include <dlfcn.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <map>
#include <mutex>
#include <unordered_set>

#include "cc_internal.h"

namespace ccamd {

#include "build/cc_eval_kernel_src.h"  // kEvalKernelSrc: the text of cc_eval_kernel.inc

#define CC_HIP(expr)                                                                                         \
  do {                                                                                                       \
    hipError_t e_ = (expr);                                                                                  \
    if (e_ != hipSuccess) return set_error(CC_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                                           __FILE__, __LINE__);                                              \
  } while (0)

#include "cc_eval_kernel.inc"

// ------------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int find_segment(const int* __restrict__ first, int n, int idx) {
  int s = 0;
  while (s + 1 < n && first[s + 1] <= idx) s++;  // n <= a few hundred, wave-uniform
  return s;
}

// ------------------------------------------------------------------------------------------------
// K1: pyramid. One thread = 4 horizontally adjacent output pixels x RESIZE_ROWS consecutive output rows of one scale.
// INTER_LINEAR_EXACT: horizontal 8.8 taps exact in 16 bits, vertical exact in 32 bits, (v + 2^15) >> 16.
// The column taps are looked up once per thread; walking down the rows, the horizontally interpolated values of a source
// row are reused when the next output row starts on it (the usual case below scale 2), so an output pixel costs about
// one new source row (2 byte loads) instead of two rows and two table lookups.
// ------------------------------------------------------------------------------------------------
// A block is 4 wavefronts = 4 consecutive bands of RESIZE_ROWS output rows x 64 words (256 columns): a wavefront stays
// inside one band, so its row taps are wave-uniform (scalar loads).
constexpr int RESIZE_ROWS = 8;
struct __attribute__((packed, aligned(1))) Bytes16 {  // 16 bytes at any address (the hardware takes unaligned global loads)
  unsigned d[4];
};
__host__ __device__ inline int resize_blocks(int pitch8, int h) {
  return ((pitch8 / 4 + 63) / 64) * (((h + RESIZE_ROWS - 1) / RESIZE_ROWS + 3) / 4);
}
// Appends one scale's column taps, padded with copies of the last tap to a multiple of 4 entries (so does every earlier
// scale: the returned offset is a multiple of 4): a thread fetches the taps of its 4 columns with one 16-byte and one
// 8-byte load, and the columns of the row padding get the last column's taps (their output is masked anyway).
static int append_column_taps(const AxisTaps& t, std::vector<int>& ofs, std::vector<uint16_t>& w1) {
  const int at = (int)ofs.size();
  ofs.insert(ofs.end(), t.ofs.begin(), t.ofs.end());
  w1.insert(w1.end(), t.w1.begin(), t.w1.end());
  while (ofs.size() % 4) {
    ofs.push_back(t.ofs.back());
    w1.push_back(t.w1.back());
  }
  return at;
}

__global__ __launch_bounds__(256) void k_resize(const uint8_t* __restrict__ frames, size_t row_stride, size_t frame_stride,
                                                int src_w, int src_h, uint8_t* __restrict__ pyr, size_t pyr_frame_bytes,
                                                const ScaleDev* __restrict__ sd, int nscales,
                                                const int* __restrict__ blk_first, const int* __restrict__ xofs,
                                                const uint16_t* __restrict__ xw1, const int* __restrict__ yofs,
                                                const uint16_t* __restrict__ yw1) {
  const int s = find_segment(blk_first, nscales, blockIdx.x);
  const ScaleDev S = sd[s];
  const int wpr = S.pitch8 >> 2, nxb = (wpr + 63) >> 6;
  const int bi = blockIdx.x - blk_first[s];
  const int bb = bi / nxb, xb = bi - bb * nxb;
  const int band = bb * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), xw = xb * 64 + (threadIdx.x & 63);
  const int ya = band * RESIZE_ROWS;
  if (ya >= S.h || xw >= wpr) return;
  const uint8_t* src = frames + (size_t)blockIdx.y * frame_stride;
  int x0[4], x1[4];
  unsigned wx0[4], wx1[4];
  {  // taps of columns 4 xw .. 4 xw + 3 (the tables are padded to the row pitch, see append_column_taps)
    const int4 o = *reinterpret_cast<const int4*>(xofs + S.xtab_ofs + xw * 4);
    const uint2 w = *reinterpret_cast<const uint2*>(xw1 + S.xtab_ofs + xw * 4);
    x0[0] = o.x, x0[1] = o.y, x0[2] = o.z, x0[3] = o.w;
    wx1[0] = w.x & 0xFFFFu, wx1[1] = w.x >> 16, wx1[2] = w.y & 0xFFFFu, wx1[3] = w.y >> 16;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      wx0[k] = 256u - wx1[k];
      x1[k] = min(x0[k] + 1, src_w - 1);
    }
  }
  // Up to scale ~4.6 the 8 source bytes a row contributes to the thread's 4 columns lie within 16 bytes: they come in
  // with ONE (unaligned) 16-byte load from `start` and are picked out with byte permutes whose selectors are fixed per
  // thread -- instead of 8 single-byte loads per source row, which is what the kernel's time went into.
  const int start = min(x0[0], src_w - 16);  // x0 / x1 do not decrease with k: x0[0] is the first, x1[3] the last byte
  const bool wide = src_w >= 16 && x1[3] - start <= 15;
  unsigned sel0 = 0, sel1 = 0, low0 = 0, low1 = 0;  // per tap: byte index within its 8-byte half, 0xFF where it is the low half
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int o0 = x0[k] - start, o1 = x1[k] - start;
    sel0 |= (unsigned)(o0 & 7) << (8 * k);
    sel1 |= (unsigned)(o1 & 7) << (8 * k);
    low0 |= (o0 < 8 ? 0xFFu : 0u) << (8 * k);
    low1 |= (o1 < 8 ? 0xFFu : 0u) << (8 * k);
  }
  auto hrow = [&](int yy, unsigned* h) {  // horizontal interpolation of source row yy at the 4 columns
    const uint8_t* r = src + (size_t)yy * row_stride;
    if (wide) {
      const Bytes16 v = *reinterpret_cast<const Bytes16*>(r + start);
      // __builtin_amdgcn_perm(hi, lo, sel): byte j of the result = byte sel[j] (0..7) of the 8 bytes {lo, hi}
      const unsigned a_lo = __builtin_amdgcn_perm(v.d[1], v.d[0], sel0), a_hi = __builtin_amdgcn_perm(v.d[3], v.d[2], sel0);
      const unsigned b_lo = __builtin_amdgcn_perm(v.d[1], v.d[0], sel1), b_hi = __builtin_amdgcn_perm(v.d[3], v.d[2], sel1);
      const unsigned t0 = (a_lo & low0) | (a_hi & ~low0), t1 = (b_lo & low1) | (b_hi & ~low1);
#pragma unroll
      for (int k = 0; k < 4; k++) h[k] = wx0[k] * ((t0 >> (8 * k)) & 255u) + wx1[k] * ((t1 >> (8 * k)) & 255u);
    } else {
#pragma unroll
      for (int k = 0; k < 4; k++) h[k] = wx0[k] * r[x0[k]] + wx1[k] * r[x1[k]];
    }
  };
  unsigned hc[4] = {0, 0, 0, 0};
  int cached = -1;  // source row whose interpolation hc holds
  uint8_t* dst = pyr + (size_t)blockIdx.y * pyr_frame_bytes + S.img_ofs;
  const int yb = min(ya + RESIZE_ROWS, S.h);
  for (int y = ya; y < yb; y++) {
    const int y0 = yofs[S.ytab_ofs + y];
    const unsigned wy1 = yw1[S.ytab_ofs + y], wy0 = 256u - wy1;
    const int y1 = min(y0 + 1, src_h - 1);
    unsigned h0[4], h1[4];
    if (y0 == cached) {
#pragma unroll
      for (int k = 0; k < 4; k++) h0[k] = hc[k];
    } else
      hrow(y0, h0);
    if (y1 == y0) {
#pragma unroll
      for (int k = 0; k < 4; k++) h1[k] = h0[k];
    } else
      hrow(y1, h1);
    unsigned packed = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const unsigned v = (h0[k] * wy0 + h1[k] * wy1 + (1u << 15)) >> 16;
      if (xw * 4 + k < S.w) packed |= v << (8 * k);
      hc[k] = h1[k];
    }
    cached = y1;
    reinterpret_cast<unsigned*>(dst + (size_t)y * S.pitch8)[xw] = packed;
  }
}

// ------------------------------------------------------------------------------------------------
// K2: integral images in one pass over the pixels (plus a tiny carry pass), ~10.5 B/px instead of 25 B/px for a
// row pass + column pass. The image is cut into bands of INT_BAND rows; one wavefront owns one band of one scale and
// walks it left to right in chunks of 256 columns (64 lanes x 4 px): per row an in-register prefix of the lane's 4 px,
// a DPP wave scan of the lane totals and a carry into the next chunk; rows accumulate downwards in registers.
//   k_integral_band<.., false>: only the band's column totals H[b][x] (its local integral's last row) are written;
//   k_integral_carry          : H[b][x] <- sum of H over the bands above b (exclusive scan down the bands, in place);
//   k_integral_band<.., true> : recomputes the band-local integral and writes row + H[b][x] (the finished integral).
// sum and sqsum use u32 wrap-around arithmetic throughout (the detector's CV_32S squared sums).
// ------------------------------------------------------------------------------------------------
constexpr int INT_BAND = 8;

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned dpp_u32(unsigned v) {
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, false);
}
// inclusive prefix sum over the 64 lanes (row_shr 1/2/4/8 inside rows of 16, then row_bcast 15 / 31 across rows)
__device__ __forceinline__ unsigned wave_scan_u32(unsigned v) {
  v += dpp_u32<0x111, 0xF>(v);
  v += dpp_u32<0x112, 0xF>(v);
  v += dpp_u32<0x114, 0xF>(v);
  v += dpp_u32<0x118, 0xF>(v);
  v += dpp_u32<0x142, 0xA>(v);
  v += dpp_u32<0x143, 0xC>(v);
  return v;
}

template <bool SQ, bool FINAL>
__global__ __launch_bounds__(256) void k_integral_band(const uint8_t* __restrict__ pyr, size_t pyr_frame_bytes,
                                                       int32_t* __restrict__ integ, size_t int_frame_elems, int nchan,
                                                       int32_t* __restrict__ hbuf, size_t h_frame_elems,
                                                       const ScaleDev* __restrict__ sd, int nscales,
                                                       const int* __restrict__ band_first, int total_bands, int sq_odd_rows_only) {
  const int lane = threadIdx.x & 63;
  const int gb = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (gb >= total_bands) return;
  const int s = find_segment(band_first, nscales, gb);
  const ScaleDev S = sd[s];
  const int bnd = gb - band_first[s];
  const int r0 = bnd * INT_BAND;
  const int nrows = min(INT_BAND, S.h - r0);
  const size_t f = blockIdx.y;
  const uint8_t* src = pyr + f * pyr_frame_bytes + S.img_ofs + (size_t)r0 * S.pitch8;
  int32_t* osum = integ + (f * nchan + 0) * int_frame_elems + S.int_ofs + (size_t)(r0 + 1) * S.pitchI;
  int32_t* osq = SQ ? integ + (f * nchan + 1) * int_frame_elems + S.int_ofs + (size_t)(r0 + 1) * S.pitchI : nullptr;
  int32_t* hsum = hbuf + (f * nchan + 0) * h_frame_elems + S.h_ofs + (size_t)bnd * S.pitchI;
  int32_t* hsq = SQ ? hbuf + (f * nchan + 1) * h_frame_elems + S.h_ofs + (size_t)bnd * S.pitchI : nullptr;
  unsigned carry_s[INT_BAND], carry_q[INT_BAND];
#pragma unroll
  for (int r = 0; r < INT_BAND; r++) carry_s[r] = carry_q[r] = 0;
  for (int c0 = 0; c0 < S.pitchI; c0 += 256) {
    const int px = c0 + lane * 4;
    const bool col_ok = px < S.pitchI;
    uint4 vs = make_uint4(0, 0, 0, 0), vq = make_uint4(0, 0, 0, 0);  // running vertical sums of the row prefixes
    if (FINAL && col_ok) {  // rows above this band
      const int4 a = *reinterpret_cast<const int4*>(hsum + px);
      vs = make_uint4((unsigned)a.x, (unsigned)a.y, (unsigned)a.z, (unsigned)a.w);
      if (SQ) {
        const int4 q = *reinterpret_cast<const int4*>(hsq + px);
        vq = make_uint4((unsigned)q.x, (unsigned)q.y, (unsigned)q.z, (unsigned)q.w);
      }
      if (bnd == 0) {  // integral row 0 is all zeros
        *reinterpret_cast<int4*>(osum - S.pitchI + px) = make_int4(0, 0, 0, 0);
        if (SQ) *reinterpret_cast<int4*>(osq - S.pitchI + px) = make_int4(0, 0, 0, 0);
      }
    }
    unsigned word[INT_BAND];
#pragma unroll
    for (int r = 0; r < INT_BAND; r++)
      word[r] = (r < nrows && px < S.pitch8) ? *reinterpret_cast<const unsigned*>(src + (size_t)r * S.pitch8 + px) : 0u;
#pragma unroll
    for (int r = 0; r < INT_BAND; r++) {
      unsigned p[4], a[4], q[4];
#pragma unroll
      for (int k = 0; k < 4; k++) p[k] = (px + k < S.w) ? ((word[r] >> (8 * k)) & 0xffu) : 0u;
      a[0] = p[0];
      q[0] = p[0] * p[0];
#pragma unroll
      for (int k = 1; k < 4; k++) {
        a[k] = a[k - 1] + p[k];
        q[k] = q[k - 1] + p[k] * p[k];
      }
      const unsigned ts = wave_scan_u32(a[3]);
      const unsigned base_s = carry_s[r] + ts - a[3];
      const unsigned last_s = base_s + a[3];
      unsigned prev_s = dpp_u32<0x138, 0xF>(last_s);  // wave_shr:1: value of the previous lane
      if (lane == 0) prev_s = carry_s[r];
      carry_s[r] = (unsigned)__builtin_amdgcn_readlane((int)last_s, 63);
      // column c of the integral row holds the sum of pixels < c: {prev lane's last, P0, P1, P2}
      vs.x += prev_s;
      vs.y += base_s + a[0];
      vs.z += base_s + a[1];
      vs.w += base_s + a[2];
      if (SQ) {
        const unsigned tq = wave_scan_u32(q[3]);
        const unsigned base_q = carry_q[r] + tq - q[3];
        const unsigned last_q = base_q + q[3];
        unsigned prev_q = dpp_u32<0x138, 0xF>(last_q);
        if (lane == 0) prev_q = carry_q[r];
        carry_q[r] = (unsigned)__builtin_amdgcn_readlane((int)last_q, 63);
        vq.x += prev_q;
        vq.y += base_q + q[0];
        vq.z += base_q + q[1];
        vq.w += base_q + q[2];
      }
      if (FINAL && col_ok && r < nrows) {
        *reinterpret_cast<int4*>(osum + (size_t)r * S.pitchI + px) = make_int4((int)vs.x, (int)vs.y, (int)vs.z, (int)vs.w);
        // The detector reads the squared sums only at the 4 corners of each window's variance rectangle: with a scan
        // step of 2 and an even window height those are odd integral rows; the even rows are never read, so they are
        // not written
        if (SQ) {
          if (sq_odd_rows_only && S.ystep == 2) {
            // ... and of those rows only the odd columns, which are packed (column 2c+1 at c): 8 bytes per lane
            if ((r0 + 1 + r) & 1) *reinterpret_cast<int2*>(osq + (size_t)r * S.pitchI + (px >> 1)) = make_int2((int)vq.y, (int)vq.w);
          } else
            *reinterpret_cast<int4*>(osq + (size_t)r * S.pitchI + px) = make_int4((int)vq.x, (int)vq.y, (int)vq.z, (int)vq.w);
        }
      }
    }
    if (!FINAL && col_ok) {
      *reinterpret_cast<int4*>(hsum + px) = make_int4((int)vs.x, (int)vs.y, (int)vs.z, (int)vs.w);
      if (SQ) *reinterpret_cast<int4*>(hsq + px) = make_int4((int)vq.x, (int)vq.y, (int)vq.z, (int)vq.w);
    }
  }
}

// Exclusive scan of the band totals down the bands (in place): thread = 4 adjacent columns of one channel of one scale.
__global__ __launch_bounds__(64) void k_integral_carry(int32_t* __restrict__ hbuf, size_t h_frame_elems, int nchan,
                                                       const ScaleDev* __restrict__ sd, int nscales,
                                                       const int* __restrict__ blk_first) {
  const int s = find_segment(blk_first, nscales, blockIdx.x);
  const ScaleDev S = sd[s];
  const int quad = (blockIdx.x - blk_first[s]) * 64 + threadIdx.x;
  if (quad * 4 >= S.pitchI) return;
  int4* p = reinterpret_cast<int4*>(hbuf + ((size_t)blockIdx.y * nchan + blockIdx.z) * h_frame_elems + S.h_ofs) + quad;
  const size_t pitch4 = S.pitchI >> 2;
  uint4 acc = make_uint4(0, 0, 0, 0);
  for (int b = 0; b < S.nbands; b++) {
    const int4 v = p[(size_t)b * pitch4];
    p[(size_t)b * pitch4] = make_int4((int)acc.x, (int)acc.y, (int)acc.z, (int)acc.w);
    acc.x += (unsigned)v.x;
    acc.y += (unsigned)v.y;
    acc.z += (unsigned)v.z;
    acc.w += (unsigned)v.w;
  }
}

// ------------------------------------------------------------------------------------------------
// Tilted (45 degree) integral for whole pyramid levels, needed only by cascades with tilted Haar features.
// With L(y,x) = sum of the pixels on the diagonal going up-left from (y-1, x) and R(y,x) = the same going up-right,
//   tilted(y, x) = tilted(y-1, x) + p(y-1, x-1) + L(y-1, x-2) + R(y-1, x)
// (the new bottom pixel of the triangle plus its two new edges), a plain column recurrence; L and R are prefix sums
// along diagonals: L(y,x) = L(y-1,x-1) + p(y-1,x), R(y,x) = R(y-1,x+1) + p(y-1,x). Pixels outside the image are 0, so
// every recurrence is border-safe.
// All three are running sums along y. Round 2 gave a whole diagonal / column to one thread: 1 080 dependent steps for a
// Full-HD image and only w + h threads per scale. Now the y axis is cut into segments of TSEG rows and each sum runs in
// two passes, like the band integrals: the *_totals kernels add up a segment (thread = one diagonal or column of one
// segment), the second kernel starts from the totals of the segments above it (at most h / TSEG small reads) and writes
// the segment's running sums. ~17x the threads for Full-HD, 64 + 17 dependent steps instead of 1 080.
// ------------------------------------------------------------------------------------------------
constexpr int TSEG = 64;  // rows per segment

// Layout of the segment totals of one frame, per scale s at tseg_ofs[s]: L totals [nseg][w + h - 1], R totals likewise,
// then the column totals of the tilted recurrence [nseg][w + 1].
struct TiltSegs {
  int nseg, ndiag, ncol;
  __host__ __device__ TiltSegs(int w, int h) : nseg((h + TSEG - 1) / TSEG), ndiag(w + h - 1), ncol(w + 1) {}
  __host__ __device__ size_t elems() const { return (size_t)nseg * (2 * (size_t)ndiag + (size_t)ncol); }
  __host__ __device__ size_t diag_at(int dir, int g) const { return ((size_t)dir * nseg + g) * (size_t)ndiag; }
  __host__ __device__ size_t col_at(int g) const { return 2 * (size_t)nseg * ndiag + (size_t)g * ncol; }
};

// A block is TILT_GROUPS wavefronts, each with its own group of 64 adjacent diagonals (or columns). Measured (16 Full-HD
// frames, rocprofv3): 1 group per block 2.41 ms for the four kernels, 4 groups per block 2.64 ms -- making the 256-byte
// pieces of neighbouring groups leave one CU together does not help, fewer and fatter blocks schedule worse.
constexpr int TILT_GROUPS = 1;
// group = 64 threads = 256 adjacent diagonals of one scale (a thread walks 4 of them: their pixels are 4 consecutive bytes
// of a row -- one unaligned 32-bit load -- and their sums 4 consecutive words of the output row -- one 16-byte store);
// blockIdx.y = frame, z = 2 * segment + direction (0: L, x - y constant; 1: R, x + y constant)
struct __attribute__((packed, aligned(1))) Bytes4 {
  unsigned d;
};
struct __attribute__((packed, aligned(4))) Words4 {
  int d[4];
};
template <bool FINAL>
__global__ __launch_bounds__(64 * TILT_GROUPS) void k_diag_sums(const uint8_t* __restrict__ pyr, size_t pyr_frame_bytes, int32_t* __restrict__ diag,
                                                  size_t int_frame_elems, int32_t* __restrict__ tseg, size_t tseg_frame_elems,
                                                  const long long* __restrict__ tseg_ofs, const ScaleDev* __restrict__ sd, int nscales,
                                                  const int* __restrict__ blk_first, int n_groups) {
  const int grp = blockIdx.x * TILT_GROUPS + (threadIdx.x >> 6);
  if (grp >= n_groups) return;
  const int s = find_segment(blk_first, nscales, grp);
  const ScaleDev S = sd[s];
  const TiltSegs T(S.w, S.h);
  const int t = ((grp - blk_first[s]) * 64 + (threadIdx.x & 63)) * 4;  // first of this thread's 4 diagonals
  const int dir = blockIdx.z & 1, g = blockIdx.z >> 1;
  if (t >= T.ndiag || g >= T.nseg) return;
  const uint8_t* img = pyr + (size_t)blockIdx.y * pyr_frame_bytes + S.img_ofs;
  int32_t* tot = tseg + (size_t)blockIdx.y * tseg_frame_elems + tseg_ofs[s];
  const int d = dir ? t : t - (S.h - 1);
  const int y0 = g * TSEG, y1 = min(y0 + TSEG, S.h);
  int acc[4] = {0, 0, 0, 0};
  if (FINAL)
    for (int k = 0; k < g; k++)  // the segments above this one
#pragma unroll
      for (int j = 0; j < 4; j++)
        if (t + j < T.ndiag) acc[j] += tot[T.diag_at(dir, k) + t + j];
  int32_t* out = diag + ((size_t)blockIdx.y * 2 + dir) * int_frame_elems + S.int_ofs;
  for (int y = y0; y < y1; y++) {
    const int x = dir ? d - y : d + y;  // column of the first diagonal; the other three follow
    if (x >= 0 && x + 3 < S.w) {
      const unsigned px = reinterpret_cast<const Bytes4*>(img + (size_t)y * S.pitch8 + x)->d;
#pragma unroll
      for (int j = 0; j < 4; j++) acc[j] += (int)((px >> (8 * j)) & 0xffu);
      if (FINAL) *reinterpret_cast<Words4*>(out + (size_t)(y + 1) * S.pitchI + x) = Words4{{acc[0], acc[1], acc[2], acc[3]}};
    } else if (x + 3 >= 0 && x < S.w) {  // the image border cuts the group
#pragma unroll
      for (int j = 0; j < 4; j++)
        if (x + j >= 0 && x + j < S.w) {
          acc[j] += img[(size_t)y * S.pitch8 + x + j];
          if (FINAL) out[(size_t)(y + 1) * S.pitchI + x + j] = acc[j];
        }
    }
  }
  if (!FINAL)
#pragma unroll
    for (int j = 0; j < 4; j++)
      if (t + j < T.ndiag) tot[T.diag_at(dir, g) + t + j] = acc[j];
}

// group = 64 columns of one scale; blockIdx.y = frame, z = segment (rows y0 + 1 .. y1 of the tilted integral)
template <bool FINAL>
__global__ __launch_bounds__(64 * TILT_GROUPS) void k_tilted_cols(const uint8_t* __restrict__ pyr, size_t pyr_frame_bytes,
                                                    const int32_t* __restrict__ diag, int32_t* __restrict__ integ,
                                                    size_t int_frame_elems, int nchan, int tilt_chan, int32_t* __restrict__ tseg,
                                                    size_t tseg_frame_elems, const long long* __restrict__ tseg_ofs,
                                                    const ScaleDev* __restrict__ sd, int nscales, const int* __restrict__ blk_first,
                                                    int n_groups) {
  const int grp = blockIdx.x * TILT_GROUPS + (threadIdx.x >> 6);
  if (grp >= n_groups) return;
  const int s = find_segment(blk_first, nscales, grp);
  const ScaleDev S = sd[s];
  const TiltSegs Tg(S.w, S.h);
  const int x = (grp - blk_first[s]) * 64 + (threadIdx.x & 63);
  const int g = blockIdx.z;
  if (x > S.w || g >= Tg.nseg) return;
  const uint8_t* img = pyr + (size_t)blockIdx.y * pyr_frame_bytes + S.img_ofs;
  const int32_t* L = diag + ((size_t)blockIdx.y * 2 + 0) * int_frame_elems + S.int_ofs;
  const int32_t* R = diag + ((size_t)blockIdx.y * 2 + 1) * int_frame_elems + S.int_ofs;
  int32_t* T = integ + ((size_t)blockIdx.y * nchan + tilt_chan) * int_frame_elems + S.int_ofs;
  int32_t* tot = tseg + (size_t)blockIdx.y * tseg_frame_elems + tseg_ofs[s];
  const int y0 = g * TSEG + 1, y1 = min(y0 + TSEG - 1, S.h);  // integral rows of this segment
  int acc = 0;
  if (FINAL) {
    for (int k = 0; k < g; k++) acc += tot[Tg.col_at(k) + x];
    if (g == 0) T[x] = 0;  // row 0
  }
  for (int y = y0; y <= y1; y++) {
    int v = x >= 1 ? img[(size_t)(y - 1) * S.pitch8 + (x - 1)] : 0;
    if (y >= 2) {
      if (x >= 2) v += L[(size_t)(y - 1) * S.pitchI + (x - 2)];
      if (x < S.w) v += R[(size_t)(y - 1) * S.pitchI + x];
    }
    acc += v;
    if (FINAL) T[(size_t)y * S.pitchI + x] = acc;
  }
  if (!FINAL) tot[Tg.col_at(g) + x] = acc;
}

// Calibration stream for the FETCH_SIZE counter: same load shape as stage_tile (dword per lane, coalesced).
__global__ __launch_bounds__(256) void k_stream_dwords(const uint32_t* __restrict__ p, size_t n_words, uint32_t* __restrict__ out) {
  uint32_t acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
  if (acc == 0x9e3779b9u) atomicAdd(out, acc);  // keeps the loads alive; practically never taken
  atomicAdd(out + 1 + (threadIdx.x & 15), acc);
}

// ------------------------------------------------------------------------------------------------
// K5: stage-0 skip rule. OpenCV's scan loop skips the next grid position of a row after a window rejected at stage 0
// (SURVEY.md A.5): window i is visited iff the run of consecutive stage-0 rejections immediately before it has even
// length. Walk the rej0 bit mask backwards, a 64-bit word at a time.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool visited_by_scan(const unsigned long long* __restrict__ row_mask, int gx) {
  int run = 0;
  int pos = gx - 1;
  while (pos >= 0) {
    const int b = pos & 63;
    const unsigned long long w = row_mask[pos >> 6] << (63 - b);  // bit `pos` now at bit 63
    const int ones = min(__clzll((long long)~w), b + 1);          // leading ones (clz(0) = 64)
    run += ones;
    if (ones < b + 1) break;
    pos -= b + 1;
  }
  return (run & 1) == 0;
}

__global__ __launch_bounds__(256) void k_filter_candidates(const CandRaw* __restrict__ cands, const int* __restrict__ cand_count,
                                                           int cand_cap, const ScaleDev* __restrict__ sd,
                                                           const unsigned long long* __restrict__ masks,
                                                           size_t mask_frame_words, CandOut* __restrict__ out,
                                                           int* __restrict__ out_count) {
  const int n = min(*cand_count, cand_cap);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const CandRaw c = cands[i];
    const ScaleDev S = sd[c.scale];
    const unsigned long long* row = masks + (size_t)c.frame * mask_frame_words + S.mask_ofs + (size_t)c.gy * S.nxw;
    if (!visited_by_scan(row, c.gx)) continue;
    const int slot = atomicAdd(out_count, 1);
    CandOut o;
    o.frame = c.frame;
    o.scale = c.scale;
    o.gx = c.gx;
    o.gy = c.gy;
    o.x = __float2int_rn((float)(c.gx * S.ystep) * S.scale);
    o.y = __float2int_rn((float)(c.gy * S.ystep) * S.scale);
    o.w = S.win_w;
    o.h = S.win_h;
    o.sum = c.sum;
    out[slot] = o;  // out has cand_cap entries; slot < n <= cand_cap
  }
}

__global__ void k_debug_visited(const ScaleDev* __restrict__ sd, int nscales, const int* __restrict__ rowblk_first,
                                const unsigned long long* __restrict__ masks, uint8_t* __restrict__ visited) {
  // one block per grid row of frame 0
  const int s = find_segment(rowblk_first, nscales, blockIdx.x);
  const ScaleDev S = sd[s];
  const int gy = blockIdx.x - rowblk_first[s];
  const unsigned long long* row = masks + S.mask_ofs + (size_t)gy * S.nxw;
  for (int gx = threadIdx.x; gx < S.nx; gx += blockDim.x)
    visited[(size_t)S.win_ofs + (size_t)gy * S.nx + gx] = visited_by_scan(row, gx) ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------
// Negative mining (training side): one thread per window of the reader's stream; integrals are read from global
// memory (windows sit half a window apart, there is little to share), geometry -> offsets on the fly because every
// ladder level has its own row pitch. Arithmetic is the trainer's: value = calc / normfactor, `<=` goes left.
// ------------------------------------------------------------------------------------------------
struct MineLevel {
  int w, h, pitchI, nx, ny;
  long long int_ofs, img_ofs;
  long long win_first;
  int pitch8;
  int pad;
};
struct MineNode {  // a tree node with its feature's geometry
  int r[3][4];
  float w[3];
  int tilted;
  float thr;
  int left, right;  // child > 0: node index inside the tree; child <= 0: leaf index -child
  int subset[8];
  int pad;
};
struct MineArgs {
  const int32_t* integ;  // channels: 0 sum, 1 sqsum (Haar), 2 tilted (if any)
  size_t chan_elems;
  const MineLevel* levels;
  int n_levels;
  long long n_windows;
  int W0, H0, ox, oy, sx, sy;
  int nstages;
  const int* stage_first;
  const int* stage_ntrees;
  const float* stage_thr;
  const MineNode* nodes;
  const int* tree_root;
  const int* tree_leaf0;
  const float* leaves;
  uint8_t* pass;   // [image][n_windows]
  int nchan;       // channels per image in integ: image f starts at integ + f * nchan * chan_elems (blockIdx.y = image)
};

template <bool HAAR>
__global__ __launch_bounds__(256) void k_negmine_windows(MineArgs A) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= A.n_windows) return;
  int l = 0;
  while (l + 1 < A.n_levels && A.levels[l + 1].win_first <= i) l++;
  const MineLevel L = A.levels[l];
  const int k = (int)(i - L.win_first);
  const int gy = k / L.nx, gx = k - gy * L.nx;
  const int x = A.ox + gx * A.sx, y = A.oy + gy * A.sy;
  const int32_t* integ = A.integ + (size_t)blockIdx.y * A.nchan * A.chan_elems;
  const int32_t* sum = integ + L.int_ofs;
  const int32_t* til = integ + 2 * A.chan_elems + L.int_ofs;
  const int P = L.pitchI;
  const size_t base = (size_t)y * P + x;
  float nf = 1.f;
  if (HAAR) {  // calcNormFactor, features.cpp:13-25 (the 4-corner difference of the wrapped squared sums is exact)
    const unsigned* sq = reinterpret_cast<const unsigned*>(integ + A.chan_elems + L.int_ofs);
    const int nw = A.W0 - 2, nh = A.H0 - 2;
    const size_t q = base + P + 1;
    const int vs = sum[q] - sum[q + nw] - sum[q + (size_t)nh * P] + sum[q + (size_t)nh * P + nw];
    const unsigned vq = sq[q] - sq[q + nw] - sq[q + (size_t)nh * P] + sq[q + (size_t)nh * P + nw];
    const double area = (double)(nw * nh);
    nf = (float)sqrt((double)(area * (double)vq - (double)vs * (double)vs));
  }
  uint8_t pass = 1;
  for (int st = 0; st < A.nstages && pass; st++) {
    double acc = 0;
    const int first = A.stage_first[st], nt = A.stage_ntrees[st];
    for (int t = first; t < first + nt; t++) {
      int idx = 0;
      const int root = A.tree_root[t];
      do {
        const MineNode* n = A.nodes + root + idx;
        bool go_left;
        if (HAAR) {
          const int32_t* b = (n->tilted ? til : sum) + base;
          float ret = 0.f;
#pragma unroll
          for (int j = 0; j < 3; j++) {
            if (j == 2 && n->w[2] == 0.0f) break;
            const int rx = n->r[j][0], ry = n->r[j][1], rw = n->r[j][2], rh = n->r[j][3];
            int p0, p1, p2, p3;
            if (!n->tilted) {  // CV_SUM_OFFSETS
              p0 = rx + P * ry;
              p1 = rx + rw + P * ry;
              p2 = rx + P * (ry + rh);
              p3 = rx + rw + P * (ry + rh);
            } else {  // CV_TILTED_OFFSETS
              p0 = rx + P * ry;
              p1 = rx - rh + P * (ry + rh);
              p2 = rx + rw + P * (ry + rw);
              p3 = rx + rw - rh + P * (ry + rw + rh);
            }
            const float term = n->w[j] * (float)(b[p0] - b[p1] - b[p2] + b[p3]);
            ret = j == 0 ? term : ret + term;
          }
          const float val = nf == 0.0f ? 0.0f : ret / nf;
          go_left = val <= n->thr;
        } else {
          const int32_t* b = sum + base;
          int p[16];
#pragma unroll
          for (int rr = 0; rr < 4; rr++)
#pragma unroll
            for (int cc = 0; cc < 4; cc++) p[4 * rr + cc] = b[(n->r[0][0] + cc * n->r[0][2]) + P * (n->r[0][1] + rr * n->r[0][3])];
          const int c = p[5] - p[6] - p[9] + p[10];
          const int code = (p[0] - p[1] - p[4] + p[5] >= c ? 128 : 0) | (p[1] - p[2] - p[5] + p[6] >= c ? 64 : 0) |
                           (p[2] - p[3] - p[6] + p[7] >= c ? 32 : 0) | (p[6] - p[7] - p[10] + p[11] >= c ? 16 : 0) |
                           (p[10] - p[11] - p[14] + p[15] >= c ? 8 : 0) | (p[9] - p[10] - p[13] + p[14] >= c ? 4 : 0) |
                           (p[8] - p[9] - p[12] + p[13] >= c ? 2 : 0) | (p[4] - p[5] - p[8] + p[9] >= c ? 1 : 0);
          go_left = (n->subset[code >> 5] & (1 << (code & 31))) != 0;
        }
        idx = go_left ? n->left : n->right;
      } while (idx > 0);
      acc += (double)A.leaves[A.tree_leaf0[t] - idx];
    }
    if (acc < (double)A.stage_thr[st]) pass = 0;
  }
  A.pass[(size_t)blockIdx.y * A.n_windows + i] = pass;
}

// Same stream, one WAVEFRONT per window: the 64 lanes take the stumps of a stage (stump t = first + lane, + 64, ...), their
// votes meet in a DPP wave sum. A background image yields only ~10^4 stream windows (13 584 for 1920x1080): one thread per
// window leaves most of the chip idle while a few hundred threads walk every stage serially. Used for stump cascades whose
// stage sums are exact in double whatever the order (stage_sums_order_independent), so the parallel sum equals the
// trainer's sequential one bit for bit; other cascades keep k_negmine_windows.
template <bool HAAR>
__global__ __launch_bounds__(256) void k_negmine_wave(MineArgs A) {
  const int lane = threadIdx.x & 63;
  const long long i = (long long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (i >= A.n_windows) return;  // wave-uniform
  int l = 0;
  while (l + 1 < A.n_levels && A.levels[l + 1].win_first <= i) l++;
  const MineLevel L = A.levels[l];
  const int k = (int)(i - L.win_first);
  const int gy = k / L.nx, gx = k - gy * L.nx;
  const int x = A.ox + gx * A.sx, y = A.oy + gy * A.sy;
  const int32_t* integ = A.integ + (size_t)blockIdx.y * A.nchan * A.chan_elems;
  const int32_t* sum = integ + L.int_ofs;
  const int32_t* til = integ + 2 * A.chan_elems + L.int_ofs;
  const int P = L.pitchI;
  const size_t base = (size_t)y * P + x;
  float nf = 1.f;
  if (HAAR) {
    const unsigned* sq = reinterpret_cast<const unsigned*>(integ + A.chan_elems + L.int_ofs);
    const int nw = A.W0 - 2, nh = A.H0 - 2;
    const size_t q = base + P + 1;
    const int vs = sum[q] - sum[q + nw] - sum[q + (size_t)nh * P] + sum[q + (size_t)nh * P + nw];
    const unsigned vq = sq[q] - sq[q + nw] - sq[q + (size_t)nh * P] + sq[q + (size_t)nh * P + nw];
    const double area = (double)(nw * nh);
    nf = (float)sqrt((double)(area * (double)vq - (double)vs * (double)vs));
  }
  uint8_t pass = 1;
  for (int st = 0; st < A.nstages; st++) {
    const int first = A.stage_first[st], nt = A.stage_ntrees[st];
    double part = 0;
    for (int t = first + lane; t < first + nt; t += 64) {
      const MineNode* n = A.nodes + A.tree_root[t];
      bool go_left;
      if (HAAR) {
        const int32_t* b = (n->tilted ? til : sum) + base;
        float ret = 0.f;
#pragma unroll
        for (int j = 0; j < 3; j++) {
          if (j == 2 && n->w[2] == 0.0f) break;
          const int rx = n->r[j][0], ry = n->r[j][1], rw = n->r[j][2], rh = n->r[j][3];
          int p0, p1, p2, p3;
          if (!n->tilted) {
            p0 = rx + P * ry;
            p1 = rx + rw + P * ry;
            p2 = rx + P * (ry + rh);
            p3 = rx + rw + P * (ry + rh);
          } else {
            p0 = rx + P * ry;
            p1 = rx - rh + P * (ry + rh);
            p2 = rx + rw + P * (ry + rw);
            p3 = rx + rw - rh + P * (ry + rw + rh);
          }
          const float term = n->w[j] * (float)(b[p0] - b[p1] - b[p2] + b[p3]);
          ret = j == 0 ? term : ret + term;
        }
        const float val = nf == 0.0f ? 0.0f : ret / nf;
        go_left = val <= n->thr;
      } else {
        const int32_t* b = sum + base;
        int p[16];
#pragma unroll
        for (int rr = 0; rr < 4; rr++)
#pragma unroll
          for (int cc = 0; cc < 4; cc++) p[4 * rr + cc] = b[(n->r[0][0] + cc * n->r[0][2]) + P * (n->r[0][1] + rr * n->r[0][3])];
        const int c = p[5] - p[6] - p[9] + p[10];
        const int code = (p[0] - p[1] - p[4] + p[5] >= c ? 128 : 0) | (p[1] - p[2] - p[5] + p[6] >= c ? 64 : 0) |
                         (p[2] - p[3] - p[6] + p[7] >= c ? 32 : 0) | (p[6] - p[7] - p[10] + p[11] >= c ? 16 : 0) |
                         (p[10] - p[11] - p[14] + p[15] >= c ? 8 : 0) | (p[9] - p[10] - p[13] + p[14] >= c ? 4 : 0) |
                         (p[8] - p[9] - p[12] + p[13] >= c ? 2 : 0) | (p[4] - p[5] - p[8] + p[9] >= c ? 1 : 0);
        go_left = (n->subset[code >> 5] & (1 << (code & 31))) != 0;
      }
      part += (double)A.leaves[A.tree_leaf0[t] - (go_left ? n->left : n->right)];
    }
    if (wave_sum_f64(part) < (double)A.stage_thr[st]) {
      pass = 0;
      break;
    }
  }
  if (lane == 0) A.pass[(size_t)blockIdx.y * A.n_windows + i] = pass;
}

// copies the pixels of selected stream windows out of the ladder: one block per window
__global__ __launch_bounds__(64) void k_negmine_gather(const uint8_t* __restrict__ pyr, size_t pyr_image_bytes, long long n_windows,
                                                       const MineLevel* __restrict__ levels, int n_levels,
                                                       const long long* __restrict__ keep, int W0, int H0, int ox, int oy, int sx, int sy,
                                                       uint8_t* __restrict__ out) {
  const long long gi = keep[blockIdx.x];  // image * n_windows + stream index
  const long long img = gi / n_windows, i = gi - img * n_windows;
  pyr += (size_t)img * pyr_image_bytes;
  int l = 0;
  while (l + 1 < n_levels && levels[l + 1].win_first <= i) l++;
  const MineLevel L = levels[l];
  const int k = (int)(i - L.win_first);
  const int gy = k / L.nx, gx = k - gy * L.nx;
  const uint8_t* src = pyr + L.img_ofs + (size_t)(oy + gy * sy) * L.pitch8 + (ox + gx * sx);
  for (int e = threadIdx.x; e < W0 * H0; e += 64) {
    const int yy = e / W0, xx = e - yy * W0;
    out[(size_t)blockIdx.x * W0 * H0 + e] = src[(size_t)yy * L.pitch8 + xx];
  }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
template <class T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  ~DevBuf() { release(); }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
  hipError_t ensure(size_t count) {
    if (count <= n) return hipSuccess;
    release();
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&p), count * sizeof(T));
    if (e == hipSuccess) n = count;
    return e;
  }
  hipError_t upload(const std::vector<T>& v, hipStream_t st) {
    hipError_t e = ensure(std::max<size_t>(v.size(), 1));
    if (e != hipSuccess || v.empty()) return e;
    return hipMemcpyAsync(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, st);
  }
};

static inline int align_up(int v, int a) { return (v + a - 1) / a * a; }

// Segment totals of the tilted front end (k_diag_sums / k_tilted_cols): where each scale's totals start, per frame.
struct TiltPlan {
  DevBuf<long long> d_ofs;
  size_t frame_elems = 0;
  int max_nseg = 0;
  hipError_t build(const std::vector<ScaleDev>& sd, hipStream_t st) {
    std::vector<long long> ofs(sd.size() + 1, 0);
    max_nseg = 0;
    for (size_t i = 0; i < sd.size(); i++) {
      const TiltSegs T(sd[i].w, sd[i].h);
      ofs[i + 1] = ofs[i] + (long long)T.elems();
      max_nseg = std::max(max_nseg, T.nseg);
    }
    frame_elems = (size_t)ofs[sd.size()];
    hipError_t e = d_ofs.upload(ofs, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);  // `ofs` goes out of scope
    return e;
  }
};

// Tilted integral of every scale of nf frames into channel tilt_chan of `integ` (diag: 2 x int_frame_elems per frame of
// scratch for the diagonal sums, tseg: tp.frame_elems per frame for the segment totals).
static void launch_tilted(hipStream_t st, const TiltPlan& tp, int32_t* tseg, const uint8_t* pyr, size_t pyr_frame_bytes, int32_t* diag,
                          int32_t* integ, size_t int_frame_elems, int nchan, int tilt_chan, const ScaleDev* sd, int ns,
                          const int* diag_first, int n_diag_blocks, const int* tcol_first, int n_tcol_blocks, int nf) {
  if (tp.max_nseg == 0 || nf == 0) return;
  const dim3 gd((n_diag_blocks + TILT_GROUPS - 1) / TILT_GROUPS, nf, 2 * tp.max_nseg), gc((n_tcol_blocks + TILT_GROUPS - 1) / TILT_GROUPS, nf, tp.max_nseg);
  const dim3 bt(64 * TILT_GROUPS);
  hipLaunchKernelGGL(k_diag_sums<false>, gd, bt, 0, st, pyr, pyr_frame_bytes, diag, int_frame_elems, tseg, tp.frame_elems, tp.d_ofs.p, sd, ns,
                     diag_first, n_diag_blocks);
  hipLaunchKernelGGL(k_diag_sums<true>, gd, bt, 0, st, pyr, pyr_frame_bytes, diag, int_frame_elems, tseg, tp.frame_elems, tp.d_ofs.p, sd, ns,
                     diag_first, n_diag_blocks);
  hipLaunchKernelGGL(k_tilted_cols<false>, gc, bt, 0, st, pyr, pyr_frame_bytes, diag, integ, int_frame_elems, nchan, tilt_chan, tseg,
                     tp.frame_elems, tp.d_ofs.p, sd, ns, tcol_first, n_tcol_blocks);
  hipLaunchKernelGGL(k_tilted_cols<true>, gc, bt, 0, st, pyr, pyr_frame_bytes, diag, integ, int_frame_elems, nchan, tilt_chan, tseg,
                     tp.frame_elems, tp.d_ofs.p, sd, ns, tcol_first, n_tcol_blocks);
}

struct Plan {
  int w = 0, h = 0;
  cc_detect_params p{};
  std::vector<ScaleGeom> geom;
  std::vector<ScaleDev> sd;
  size_t pyr_frame_bytes = 0, int_frame_elems = 0, mask_frame_words = 0;
  long long windows = 0, integral_elems = 0;
  int n_resize_blocks = 0, n_bands = 0, n_col_blocks = 0, n_grid_rows = 0, n_diag_blocks = 0, n_tcol_blocks = 0;
  size_t h_frame_elems = 0;
  int n_tiles = 0;  // tiles of TILE_Y window rows (the ahead-of-time kernels); other heights: tiles_for
  struct TileList {
    int n = 0;
    DevBuf<int4> d;
  };
  std::map<int, std::unique_ptr<TileList>> other_tiles;  // tile lists of the specialised kernel's modules, key = tile_list_key(rows, step)
  DevBuf<ScaleDev> d_sd;
  DevBuf<int> d_resize_first, d_band_first, d_col_first, d_gridrow_first, d_diag_first, d_tcol_first, d_xofs, d_yofs;
  DevBuf<uint16_t> d_xw1, d_yw1;
  DevBuf<int4> d_tiles;
  TiltPlan tilt;
  // Single-image calls (the detection tool's shape) are launch-bound: ~10 launches, memsets and copies for well under a
  // millisecond of device work. After a first ordinary call has sized every buffer, the whole pass (H2D copy of the
  // image, pyramid, integrals, cascade kernel, skip filter, copy-back of the counters) is captured into a hipGraph and
  // replayed with one launch for as long as the buffers and kernels it recorded stay the same (`graph_key`).
  bool graph_warm = false;
  hipGraphExec_t graph_exec = nullptr;
  std::vector<const void*> graph_key;
  ~Plan() {
    if (graph_exec) (void)hipGraphExecDestroy(graph_exec);
  }
};

struct TimingEvent {
  hipEvent_t a, b;
  int kind;
};

// Where the filtered candidates of a batch's passes go (called on the host thread that retires the pass), and the first
// error met while retiring one of its passes from another call (cc_detect_batch_submit of the NEXT batch).
struct BatchSink {
  std::function<void(int f0, int nf, std::vector<CandOut>& cands)> consume;
  cc_status status = CC_OK;
  std::string error;
  // Submitted batches (cc_detect_batch_submit): the host side of a retired pass -- sorting and grouping its candidates,
  // ~1.3 ms for 19 Full-HD frames -- runs on a helper thread, so that the submitting thread goes straight on to launch
  // the next pass / the next batch; cc_detect_batch_collect waits for the helpers. (Doing all of it in collect instead was
  // measured too: 18.5 -> 23.5 ms per step, the device idles while the host groups 64 frames in one go.)
  bool async_consume = false;
  std::vector<std::future<void>> jobs;
  void wait_jobs() {
    for (auto& j : jobs)
      if (j.valid()) j.get();
    jobs.clear();
  }
};

}  // namespace ccamd

using namespace ccamd;

constexpr int kStageSlots = 3;  // staging slots for host frames (run_batch: why three)

// One compiled module of the run-time specialised kernel: the tiles it covers (0 = all, 1 / 2 = the tiles of STEP-1 / STEP-2
// scales) and the tile height it was compiled for.
struct SpecCode {
  std::vector<char> code;
  int only_step = 0;
  int tile_y = TILE_Y;
};

struct cc_detector {
  Cascade m;
  unsigned long long serial = 0;  // unique per created detector (tickets name their owner by it, not by address alone)
  int device = 0;
  int max_batch = 1;
  int pass_capacity = 1;  // frames the per-pass workspace is sized for: the largest pass seen so far (<= max_batch)
  hipStream_t own_stream = nullptr, stream = nullptr;
  // cascade tables on the device (one per tile layout)
  DevBuf<int> d_stage_ntrees, d_stage_first;
  DevBuf<int> d_group_first;  // stage groups of the cascade kernel (EvalArgs::group_first), n_groups + 1 entries
  int n_groups = 0;
  int dense_from = 0x7fffffff;  // first stage group whose queue is a plain list instead of a bank-class table (EvalArgs::dense_from)
  DevBuf<float> d_stage_thr;
  int wave_below = 0;
  int last_call_graph = 0;  // the last single-image call was one hipGraph launch (cc_detector_graph_active)
  int stop_after = -1;
  int split_stumps = 0;
  DevBuf<HaarStumpDev> d_haar1, d_haar2;
  DevBuf<HaarStumpDev> d_haar1w, d_haar2w;  // the same stumps dealt to lanes for the wave phase (bank-aware order)
  DevBuf<HaarStumpDev> d_haar_g;            // corners as window coordinates (GlobalReader; kernels with 16-bit tiles)
  DevBuf<LbpStumpDev> d_lbp_g, d_lbp16;      // LBP: window coordinates (GlobalReader) / 16-bit STEP-2 tile offsets (LBP wave phase)
  int lbp16_all = 0;                        // every LBP cell of the cascade sums below 2^16
  DevBuf<LbpStumpDev> d_lbp1, d_lbp2;
  DevBuf<HaarNodeDev> d_hnode1, d_hnode2;  // cascades with trees deeper than stumps
  DevBuf<LbpNodeDev> d_lnode1, d_lnode2;
  DevBuf<int> d_tree_root, d_tree_leaf0;
  DevBuf<float> d_leaves;
  size_t lds = 0;  // dynamic LDS bytes per tile (larger of the two layouts)
  size_t lds_spec = 0;  // the same for the installed specialised kernel (smaller when its STEP-2 tiles hold 16-bit entries)
  size_t lds_extra = 0; // CCAMD_DEBUG_EXTRA_LDS (occupancy experiments)
  int spec_tmode = 0;  // TILE_32 / TILE_16 / TILE_PAIR16 of the installed specialised kernel
  int spec_tile_y = TILE_Y;  // window rows per tile the installed specialised kernel was compiled for (spec_tile_rows)
  DevBuf<HaarStumpP16> d_haar_p16, d_haar_p16w;  // table-driven stumps of the pair tile: stage order, wave-phase order
  // plans + workspace
  std::vector<std::unique_ptr<Plan>> plans;
  DevBuf<uint8_t> d_frames, d_pyr;
  // The integral images are double-buffered: pyramid + integrals of pass i+1 are built on `front_stream` while the
  // cascade kernel of pass i (LDS/VALU-bound, leaves wave slots and all of HBM idle) runs on `stream`.
  DevBuf<int32_t> d_integ[2], d_hbuf, d_diag, d_tseg;
  hipStream_t front_stream = nullptr;
  hipEvent_t front_done[2] = {nullptr, nullptr}, eval_done[2] = {nullptr, nullptr}, batch_begin = nullptr;
  bool eval_pending[2] = {false, false};
  int overlap_front = 1;
  // run-time specialised cascade kernel (cc_detector_specialize); null = table-driven kernel
  hipModule_t spec_mod = nullptr;   // the module that covers every tile, or the tiles of STEP-2 scales when spec_mod1 exists
  hipFunction_t spec_fn = nullptr;
  // Optional second module for the tiles of STEP-1 scales (spec_modules: Haar kernels with 32-bit tiles compile one module per
  // step, each with its own tile height and LDS request)
  hipModule_t spec_mod1 = nullptr;
  hipFunction_t spec_fn1 = nullptr;
  size_t lds_spec1 = 0;
  int spec_tile_y1 = TILE_Y;
  int last_stamp_tiles = 0;  // CCAMD_DEBUG_STAMPS: tiles of the last pass (all launches)
  int spec_only_step = 0;  // tiles the primary module covers: 0 = all, 2 = those of STEP-2 scales (then spec_fn1 covers STEP 1)
  int spec_stages = 0;
  // background build of the specialised module (cc_detector_specialize_async / CCAMD_AUTO_SPECIALIZE): a host thread
  // generates and compiles; the next detection call on the owning thread loads the module and switches over
  std::thread spec_thread;
  std::atomic<int> spec_bg_state{0};  // 0 idle, 1 building, 2 ready to install, 3 failed
  std::vector<SpecCode> spec_bg_code;
  int spec_bg_stages = 0;
  int spec_bg_tmode = 0;
  std::string spec_bg_error;
  // Per slot, like the results. (Round 3 also ran the cascade kernels of consecutive passes on two streams, so that the next
  // one starts on the CUs the previous one's last blocks leave free: 18.50 -> 19.24 ms per step, two kernels of this size
  // only get in each other's way. Not kept.)
  DevBuf<unsigned long long> d_masks[2];
  DevBuf<CandRaw> d_cands[2];
  // Results of a pass are double-buffered so that the host can fetch and group pass i while the device runs pass i+1.
  DevBuf<CandOut> d_out[2];
  DevBuf<int> d_counts[2];  // [0] raw count, [1] filtered count
  int* h_counts = nullptr;  // pinned, 2 x 2 ints
  uint8_t* h_frame = nullptr;  // pinned staging copy of a single host image (graph path)
  size_t h_frame_bytes = 0;
  uint8_t* h_stage = nullptr;  // pinned staging area for batches of host frames, kStageSlots slots like d_frames (stage_host_frames)
  size_t h_stage_bytes = 0;
  int stage_slot = 0;          // staging slot the next pass of host frames takes (round-robin, also across calls)
  int use_graph = 1;
  int early_skip = 1, full_sqsum = 0, pipeline_passes = 4, pipeline_passes_set = 0, even_passes = 0;  // tuning knobs, read once at creation
  hipStream_t copy_stream = nullptr;
  hipEvent_t pass_done[2] = {nullptr, nullptr};
  int cand_cap = 0;
  // The pass launched last, not yet fetched (run_batch): inside a batch that is what lets the host side of pass i overlap
  // the device side of pass i + 1; across calls (cc_detect_batch_submit) it lets the first pass of the next batch overlap
  // the last pass of this one. `sink` receives the pass's candidates when it is retired.
  struct PendingPass {
    bool active = false;
    Plan* plan = nullptr;
    int f0 = 0, nf = 0, slot = 0;
    const uint8_t* dptr = nullptr;
    size_t rs = 0, fs = 0;
    int cap = 0;       // capacity of the candidate lists the pass was launched with
    unsigned gen = 0;  // generation of the candidate lists it wrote into
    bool debug = false;
    std::shared_ptr<ccamd::BatchSink> sink;
  } pending;
  unsigned list_gen = 0;  // bumped whenever the candidate lists are released and regrown
  int next_slot = 0;      // result / integral slot the next pass uses (alternates, also across calls)
  DevBuf<unsigned long long> d_stamps;  // CCAMD_DEBUG_STAMPS experiments
  DevBuf<int32_t> d_dbg_codes;
  DevBuf<double> d_dbg_sums;
  DevBuf<uint8_t> d_dbg_visited;
  // profiling
  bool profiling = false;
  std::vector<TimingEvent> events;
  cc_detector_timings tm{};

  ~cc_detector() {
    for (auto& e : events) {
      (void)hipEventDestroy(e.a);
      (void)hipEventDestroy(e.b);
    }
    if (own_stream) (void)hipStreamDestroy(own_stream);
    if (copy_stream) (void)hipStreamDestroy(copy_stream);

    if (front_stream) (void)hipStreamDestroy(front_stream);
    if (spec_thread.joinable()) spec_thread.join();
    if (spec_mod) (void)hipModuleUnload(spec_mod);
    if (spec_mod1) (void)hipModuleUnload(spec_mod1);
    for (hipEvent_t e : {pass_done[0], pass_done[1], front_done[0], front_done[1], eval_done[0], eval_done[1], batch_begin})
      if (e) (void)hipEventDestroy(e);
    if (h_counts) (void)hipHostFree(h_counts);
    if (h_frame) (void)hipHostFree(h_frame);
    if (h_stage) (void)hipHostFree(h_stage);
  }
};

struct cc_negminer {
  Cascade m;
  int device = 0;
  hipStream_t stream = nullptr;
  DevBuf<MineNode> d_nodes;
  DevBuf<int> d_stage_first, d_stage_ntrees, d_tree_root, d_tree_leaf0;
  DevBuf<float> d_stage_thr, d_leaves;
  // per-image workspace
  DevBuf<uint8_t> d_src, d_pyr, d_pass, d_pix;
  DevBuf<int32_t> d_integ, d_hbuf, d_diag, d_tseg;
  TiltPlan tilt;
  DevBuf<ScaleDev> d_sd;
  DevBuf<MineLevel> d_levels;
  DevBuf<int> d_resize_first, d_band_first, d_col_first, d_diag_first, d_tcol_first, d_xofs, d_yofs;
  DevBuf<uint16_t> d_xw1, d_yw1;
  DevBuf<long long> d_keep;
  // The tables above depend on (image size, offset) only: consecutive images of a background set share them, so they are
  // built and uploaded when that key changes, not per call.
  struct Plan {
    int width = -1, height = -1, ox = -1, oy = -1;
    int nl = 0, n_resize = 0, n_bands = 0, n_cols = 0, n_diag = 0, n_tcol = 0;
    long long pyr_bytes = 0, chan_elems = 0, h_elems = 0, wins = 0;
  } plan;
  uint8_t* h_src = nullptr;   // pinned: the images of a call, tight rows of align4(width)
  uint8_t* h_pass = nullptr;  // pinned: pass flags on their way back
  size_t h_src_bytes = 0, h_pass_bytes = 0;
  hipStream_t copy_stream = nullptr;      // the images' way to the device, piece by piece, under the kernels of the piece before
  std::vector<hipEvent_t> piece_landed;   // one per piece of a call (grown on demand)
  ~cc_negminer() {
    for (hipEvent_t e : piece_landed) (void)hipEventDestroy(e);
    if (copy_stream) (void)hipStreamDestroy(copy_stream);
    if (stream) (void)hipStreamDestroy(stream);
    if (h_src) (void)hipHostFree(h_src);
    if (h_pass) (void)hipHostFree(h_pass);
  }
};

namespace ccamd {

static cc_status ensure_device(int device) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return set_error(CC_ERR_NO_DEVICE, "no usable HIP device (%s); this library has no CPU fallback",
                     e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
  if (device < 0 || device >= n) return set_error(CC_ERR_INVALID_ARG, "device %d out of range (devices: %d)", device, n);
  CC_HIP(hipSetDevice(device));
  return CC_OK;
}

// True when, for every stage, any partial sum of leaf values is exactly representable in double: all leaves are
// integer multiples of q = 2^(emin-23) (emin = smallest exponent among the stage's nonzero leaves) and the sum of the
// larger leaf magnitudes divided by q stays below 2^53. Then the double accumulation never rounds, so its result
// does not depend on the order of the additions.
static bool stage_sums_order_independent(const Cascade& m, double headroom = 1.0) {
  for (size_t s = 0; s < m.stage_ntrees.size(); s++) {
    int emin = INT32_MAX;
    double mag = 0;
    for (int i = 0; i < m.stage_ntrees[s]; i++) {
      const size_t k = (size_t)m.stage_first[s] + i;
      const float l = m.stump_left[k], r = m.stump_right[k];
      if (!std::isfinite(l) || !std::isfinite(r)) return false;
      mag += std::max(std::fabs((double)l), std::fabs((double)r));
      for (float v : {l, r})
        if (v != 0.0f) {
          int e;
          std::frexp(v, &e);
          emin = std::min(emin, e);
        }
    }
    if (emin == INT32_MAX) continue;
    // v = f * 2^e with f in [0.5, 1) and a 24-bit significand: v is a multiple of 2^(e-24); subnormals only get coarser
    const double q = std::ldexp(1.0, emin - 24);
    if (mag * headroom / q >= 9007199254740992.0) return false;
  }
  return true;
}

// `at(y, x)` maps a corner inside the window to what the record stores: an LDS offset of one of the tile layouts, or
// (y << 16 | x) for the records whose corners are read from global memory (GlobalReader). tilt_shift: distance of the
// tilted tile behind the sum tile.
template <class At>
static void build_haar_stumps_at(const Cascade& m, std::vector<HaarStumpDev>& out, At at, int tilt_shift) {
  out.resize(m.stump_feature.size());
  for (size_t i = 0; i < out.size(); i++) {
    HaarStumpDev& d = out[i];
    std::memset(&d, 0, sizeof(d));
    const int fi = m.stump_feature[i];
    d.nrect = 2;
    for (int j = 0; j < 3; j++) {
      const int32_t* r = &m.haar_rects[(size_t)fi * 12 + j * 4];
      const float wt = m.haar_weights[(size_t)fi * 3 + j];
      d.w[j] = wt;
      // rects after the first zero weight contribute w*0 upstream (offsets stay 0): keep corner offsets equal
      const bool used = j < 2 || wt != 0.0f;
      if (j == 2 && wt != 0.0f) d.nrect = 3;
      const int x = used ? r[0] : 0, y = used ? r[1] : 0, rw = used ? r[2] : 0, rh = used ? r[3] : 0;
      if (!m.haar_tilted[fi]) {
        d.ofs[j][0] = at(y, x);
        d.ofs[j][1] = at(y, x + rw);
        d.ofs[j][2] = at(y + rh, x);
        d.ofs[j][3] = at(y + rh, x + rw);
      } else {  // corners of the 45-degree rectangle (CV_TILTED_OFFSETS), read from the tilted tile behind the sum tile
        d.ofs[j][0] = tilt_shift + at(y, x);
        d.ofs[j][1] = tilt_shift + at(y + rh, x - rh);
        d.ofs[j][2] = tilt_shift + at(y + rw, x + rw);
        d.ofs[j][3] = tilt_shift + at(y + rw + rh, x + rw - rh);
      }
    }
    d.thr = m.stump_threshold[i];
    d.left = m.stump_left[i];
    d.right = m.stump_right[i];
    d.pad = (int)i;  // the stump's index: survives the re-ordering of schedule_for_wave_phase
  }
}
template <int STEP>
static void build_haar_stumps(const Cascade& m, std::vector<HaarStumpDev>& out) {
  const TileGeom<STEP> G(m.win_w, m.win_h);
  build_haar_stumps_at(m, out, [&](int y, int x) { return G.at(y, x); }, tile_words_padded(G.words()));
}
static int window_xy(int y, int x) { return (y << 16) | x; }  // GlobalReader records (upright features only)
static void build_haar_gstumps(const Cascade& m, std::vector<HaarStumpDev>& out) { build_haar_stumps_at(m, out, window_xy, 0); }

// Source text of spec_stage<1|2> (and spec_stage0_x2<1|2>) for the first n_stages stages: every stump becomes
// straight-line code whose LDS offsets, weights, threshold and leaf values are literals (hex floats, exact). The
// expression is the one of stump_vote(), term by term, so results are bit-identical to the table-driven path.
//
// The code is software-pipelined by construction: the LDS reads of stump i + D are issued before stump i is computed,
// and scheduling barriers pin that order (left alone, the compiler emits read, wait, compute per stump and every
// wavefront spends most of its time waiting for the LDS round trip). One copy of a stage serves the whole-stage call and
// the stump-split calls: the stage is cut into SPEC_PARTS contiguous parts, a call evaluates parts [p_lo, p_hi) and only
// its first part runs the prologue that issues the first D stumps' reads (a part's tail prefetches into the next part,
// so consecutive parts run without a pipeline drain).
struct SpecStump {
  std::string loads;    // statements "x = b[..];" (variables are declared by the caller)
  std::string decls;    // declarations of those variables
  std::string compute;  // statement adding the stump's vote to `acc`
  double base = 0.;     // constant part of the vote, added once per part (delta form, see spec_stage_source)
  long long base_q = 0; // the same in units of the stage's quantum (fixed-point form)
};

// The generated stages are called from several places of the kernel (dense group, thread phase, stump-split slices).
// Inlined everywhere, the code of every stage exists once per call site; as a real function it exists once (a third of
// the instructions for the bench cascade) at the price of the call convention's register traffic. CCAMD_SPEC_NOINLINE picks.
static const char* spec_stage_inline_attr() {
  const char* e = std::getenv("CCAMD_SPEC_NOINLINE");
  return (e && std::atoi(e) != 0) ? "__noinline__" : "__forceinline__";
}

static int spec_prefetch_depth(int d = 2) {
  if (const char* e = std::getenv("CCAMD_SPEC_PREFETCH")) d = std::max(0, std::min(4, std::atoi(e)));  // tuning
  return d;
}

// Emits the body of one stage from per-stump pieces (see above). `suffixes` = one accumulator / window per entry.
static void spec_emit_stage(std::string& o, const std::vector<SpecStump>& st, int depth, bool parts, const std::vector<std::string>& accs,
                            bool fixed_point = false) {
  const int nt = (int)st.size();
  static const char* kSB = "      __builtin_amdgcn_sched_barrier(0);\n";
  for (const SpecStump& t : st) o += "      " + t.decls + "\n";
  const int P = parts ? SPEC_PARTS : 1;
  char buf[128];
  int prev_nonempty = -1;  // last part before k that holds stumps
  for (int k = 0; k < P; k++) {
    const int e0 = (int)((long long)k * nt / P), e1 = (int)((long long)(k + 1) * nt / P);
    if (e0 == e1) continue;
    // The prologue (reads of the part's first `depth` stumps) belongs to the call whose range STARTS at this part -- or at
    // one of the empty parts just before it: a stage with fewer stumps than SPEC_PARTS has empty parts, and a call that
    // starts on one (the whole-stage call starts on part 0) must still issue the reads of the first stumps it evaluates.
    if (parts)
      snprintf(buf, sizeof(buf), "      if (p_lo > %d && p_lo <= %d) {\n", prev_nonempty, k);
    else
      snprintf(buf, sizeof(buf), "      {\n");
    prev_nonempty = k;
    o += buf;
    for (int i = e0; i < std::min(e0 + depth, nt); i++) o += "      " + st[(size_t)i].loads + "\n" + kSB;
    o += "      }\n";
    if (parts) {
      snprintf(buf, sizeof(buf), "      if (p_lo <= %d && %d < p_hi) {\n", k, k);
      o += buf;
    } else
      o += "      {\n";
    {  // constant parts of this part's votes (delta form): one exact addition per accumulator
      double base = 0.;
      unsigned base_q = 0;  // modulo 2^32, like the accumulator
      for (int i = e0; i < e1; i++) {
        base += st[(size_t)i].base;
        base_q += (unsigned)st[(size_t)i].base_q;
      }
      char lit[64];
      if (fixed_point)
        snprintf(lit, sizeof(lit), "%uu", base_q);
      else
        snprintf(lit, sizeof(lit), "%a", base);
      if (fixed_point ? base_q != 0 : base != 0.)
        for (const std::string& a : accs) o += "      " + a + " += " + lit + ";\n";
    }
    for (int i = e0; i < e1; i++) {
      if (depth > 0 && i + depth < nt) o += "      " + st[(size_t)(i + depth)].loads + "\n" + kSB;
      if (depth == 0) o += "      " + st[(size_t)i].loads + "\n";
      o += "      " + st[(size_t)i].compute + "\n" + kSB;
    }
    o += "      }\n";
  }
}

static std::string spec_stage_source_lbp(const Cascade& m, int n_stages, bool tile16);
template <int STEP>
static void build_lbp_stumps(const Cascade& m, std::vector<LbpStumpDev>& out);
static void build_lbp_stumps16(const Cascade& m, std::vector<LbpStumpDev>& out);

// Layout of the STEP-2 tiles of a specialised kernel: 32-bit entries in two column planes (TileGeom<2>), 16-bit entries
// (TileGeom16), or packed pairs of 16-bit entries (TileGeomP: two neighbouring windows per LDS read, Haar stumps).
enum { TILE_32 = 0, TILE_16 = 1, TILE_PAIR16 = 2 };

// A rectangle sum read from 16-bit entries is exact when 255 * area < 2^16.
static bool fits16(long long area) { return 255LL * area <= 65535LL; }

// Cuts the rectangle (x, y, w, h) into the fewest strips along its longer side whose sums each fit 16 bits.
static std::vector<std::array<int, 4>> pieces16(int x, int y, int w, int h) {
  std::vector<std::array<int, 4>> out;
  const bool along_x = w >= h;
  const int len = along_x ? w : h;
  for (int k = 1; k <= std::max(len, 1); k++) {
    out.clear();
    bool ok = true;
    for (int i = 0; i < k; i++) {
      const int a = (int)((long long)i * len / k), b = (int)((long long)(i + 1) * len / k);
      if (a == b) continue;
      const std::array<int, 4> pc = along_x ? std::array<int, 4>{x + a, y, b - a, h} : std::array<int, 4>{x, y + a, w, b - a};
      ok = ok && fits16((long long)pc[2] * pc[3]);
      out.push_back(pc);
    }
    if (ok) return out;
  }
  return {};  // a single row or column of the window does not fit: the caller's eligibility test has excluded this
}

// Can the first n_stages stages be generated for STEP-2 tiles with 16-bit entries (TileGeom16)? Upright Haar features
// (any rectangle is cut into strips that fit) or LBP cells that fit; the variance rectangle is read as two halves.
static bool tile16_eligible(const Cascade& m, int n_stages) {
  // Measured in round 3. Haar: the 16-bit tile raises the resident blocks per CU from 5 to 7 and the
  // thread-per-window stages gain 4 %, but the table-driven wave phase then reads its corners from global memory and loses
  // twice that: 15 % slower as a whole -> only on request (CCAMD_SPEC_TILE16=1). LBP with EVERY stage compiled (the stock
  // cascade: 20 stages, 139 stumps) has no table-driven stage and no wave phase, and its short stages are chains of
  // dependent stump latencies that more resident wavefronts do hide: 7.6 -> 6.7 ms per 32 frames -> on by default.
  if (m.max_nodes_per_tree > 1) return false;
  const int total = (int)m.stage_ntrees.size();
  n_stages = std::min<int>(n_stages, total);
  const char* on = std::getenv("CCAMD_SPEC_TILE16");
  if (on ? std::atoi(on) == 0 : !(m.feature_type == CC_FEATURE_LBP && n_stages == total)) return false;
  if (m.feature_type == CC_FEATURE_HAAR) {
    if (m.has_tilted) return false;
    const int nrx = m.win_w - 2, nry = m.win_h - 2;
    if (nrx < 2 || !fits16((long long)(nrx - (nrx >> 1)) * nry)) return false;
    if (!fits16(std::max(m.win_w, m.win_h))) return false;  // strips of one row / column always fit
    return true;
  }
  for (int s = 0; s < n_stages; s++)
    for (int i = 0; i < m.stage_ntrees[(size_t)s]; i++) {
      const int32_t* r = &m.lbp_rects[(size_t)m.stump_feature[(size_t)m.stage_first[(size_t)s] + i] * 4];
      if (!fits16((long long)r[2] * r[3])) return false;
    }
  return true;
}

// Table-driven records of the pair tile: every rectangle as strips whose sums fit 16 bits. False when a stump needs more
// than P16_PIECES pieces or an offset does not fit the record.
static bool build_haar_stumps_p16(const Cascade& m, std::vector<HaarStumpP16>& out) {
  const TileGeomP G(m.win_w, m.win_h);
  out.assign(m.stump_feature.size(), HaarStumpP16{});
  for (size_t i = 0; i < out.size(); i++) {
    HaarStumpP16& d = out[i];
    std::memset(&d, 0, sizeof(d));
    const int fi = m.stump_feature[i];
    if (m.haar_tilted[(size_t)fi]) return false;
    d.nrect = 2;
    int np = 0;
    for (int j = 0; j < 3; j++) {
      const int32_t* r = &m.haar_rects[(size_t)fi * 12 + j * 4];
      const float wt = m.haar_weights[(size_t)fi * 3 + j];
      d.w[j] = wt;
      const bool used = j < 2 || wt != 0.0f;
      if (j == 2 && wt != 0.0f) d.nrect = 3;
      if (!used || r[2] <= 0 || r[3] <= 0) continue;  // an empty rectangle sums to 0: no piece
      const auto pcs = pieces16(r[0], r[1], r[2], r[3]);
      if (pcs.empty()) return false;
      for (const auto& pc : pcs) {
        if (np >= P16_PIECES) return false;
        const int o4[4] = {G.at(pc[1], pc[0]), G.at(pc[1], pc[0] + pc[2]), G.at(pc[1] + pc[3], pc[0]), G.at(pc[1] + pc[3], pc[0] + pc[2])};
        for (int k = 0; k < 4; k++) {
          if (o4[k] < 0 || o4[k] > 65535) return false;
          d.ofs[np][k] = (unsigned short)o4[k];
        }
        d.rect_of[np++] = (unsigned char)j;
      }
    }
    d.npieces = (unsigned char)np;
    d.thr = m.stump_threshold[i];
    d.left = m.stump_left[i];
    d.right = m.stump_right[i];
  }
  return true;
}

// Can the specialised kernel use the pair tile (TileGeomP) for its STEP-2 tiles? Upright Haar stump cascades whose
// variance rectangle splits into two halves that fit 16 bits, whose every stump fits the pair tile's record, and whose
// tile row fits one wavefront of 4-column groups (stage_tile_pair). CCAMD_SPEC_PAIR16=0 / 1 turns it off / on.
static bool pair16_eligible(const Cascade& m) {
  if (m.feature_type != CC_FEATURE_HAAR || m.max_nodes_per_tree > 1 || m.has_tilted) return false;
  const char* on = std::getenv("CCAMD_SPEC_PAIR16");
  if (!(on && std::atoi(on) != 0)) return false;
  if (TILE_Y != 2 * EVAL_WAVES || TILE_X != 64) return false;
  const int nrx = m.win_w - 2, nry = m.win_h - 2;
  if (nrx < 2 || !fits16((long long)(nrx - (nrx >> 1)) * nry)) return false;
  if (!fits16(std::max(m.win_w, m.win_h))) return false;
  const TileGeomP G(m.win_w, m.win_h);
  if ((G.cols + 3) / 4 > 64) return false;
  std::vector<HaarStumpP16> tmp;
  return build_haar_stumps_p16(m, tmp);
}

static std::string spec_stage_source(const Cascade& m, int n_stages, int tmode) {
  const CNumericLocale c_numbers;  // "%a" literals must not follow the host program's LC_NUMERIC
  if (m.feature_type == CC_FEATURE_LBP) return spec_stage_source_lbp(m, n_stages, tmode == TILE_16);
  std::vector<HaarStumpDev> t[2];
  build_haar_stumps<1>(m, t[0]);
  build_haar_stumps<2>(m, t[1]);
  const TileGeom16 G16(m.win_w, m.win_h);
  n_stages = std::min<int>(n_stages, (int)m.stage_ntrees.size());
  const int depth = spec_prefetch_depth();
  // Delta form of a vote: `(v < thr ? left : right)` needs both leaf values in registers (a select takes one literal), and
  // the compiler hoists those ~2 registers per stump out of the stage loop until it spills; `right + (v < thr ? left - right
  // : 0)` selects between ONE literal and zero, and the `right`s of a part add up to one constant. Exact -- hence equal to
  // the sequential sum of the votes -- when every partial sum of leaves and differences is representable
  // (stage_sums_order_independent with headroom for the differences); otherwise the plain form is generated.
  const bool delta_form = stage_sums_order_independent(m, 4.0) && !std::getenv("CCAMD_SPEC_NO_DELTA");
  const bool fixed_point_ok = !std::getenv("CCAMD_SPEC_NO_FIXED");  // tuning / bisecting
  std::string o;
  char buf[512];
  auto hexf = [&](float v) {
    snprintf(buf, sizeof(buf), "%af", (double)v);
    return std::string(buf);
  };
  // One stump. When every weight is a small integer and sum |w_j| * 255 * area_j < 2^24, every intermediate of the
  // float expression w0*(float)r0 + w1*(float)r1 [+ w2*(float)r2] is an exactly representable integer, so the value
  // equals (float) of the same combination computed in int32: corners shared by the rectangles merge, one conversion
  // instead of three, no float multiplies. Otherwise the float expression is emitted term by term.
  // `win` names the window (variables x<stump>_<k><win>, base pointer b<win>, vnf<win>, acc<win>).
  // Fixed-point votes. Where a stage's leaves are all multiples of q = 2^k and the sum of their magnitudes stays below
  // 2^31 q, the stage sum of ANY subset of votes is an int32 multiple of q: the delta-form votes are then accumulated as
  // 32-bit integers (one select + one add per stump instead of two selects and a double add; intermediate wrap-around
  // is harmless modulo 2^32) and converted once, exactly, at the end: (double)(int)acc * q is the same real number the
  // double accumulation produces, so every comparison and reported sum is bit-identical.
  auto stage_quantum = [&](int s, double& q) {
    int emin = INT32_MAX;
    double mag = 0;
    for (int i = 0; i < m.stage_ntrees[(size_t)s]; i++) {
      const size_t k = (size_t)m.stage_first[(size_t)s] + i;
      const float l = m.stump_left[k], r = m.stump_right[k];
      mag += std::max(std::fabs((double)l), std::fabs((double)r));
      for (float v : {l, r})
        if (v != 0.0f) {
          int e;
          std::frexp(v, &e);
          emin = std::min(emin, e);
        }
    }
    if (emin == INT32_MAX) return false;
    q = std::ldexp(1.0, emin - 24);  // every leaf is a multiple of q (see stage_sums_order_independent)
    return mag / q < 2147483647.0;
  };
  std::function<std::string(const HaarStumpDev&, const std::string&, double, SpecStump&)> vote_text;
  const TileGeomP GP(m.win_w, m.win_h);
  // `reuse`: words the stump evaluated just before this one holds in variables (tile offset -> name): a corner both stumps
  // read is not loaded again. `vars_out` receives this stump's own map for the next one.
  auto stump = [&](const HaarStumpDev& d, int stump_index, int local, const std::string& win, double fixed_q, bool h16,
                   const std::map<int, std::string>* reuse = nullptr, std::map<int, std::string>* vars_out = nullptr) {
    const int fi = m.stump_feature[(size_t)stump_index];
    const std::string tile_ptr = (h16 ? "h" : "b") + win;  // h<win>: the same tile base as 16-bit entries
    bool int_ok = true;
    double bound = 0;
    for (int j = 0; j < d.nrect; j++) {
      const float w = d.w[j];
      const int32_t* r = &m.haar_rects[(size_t)fi * 12 + j * 4];
      if (w != std::nearbyint(w) || std::fabs(w) > 64.f) int_ok = false;
      bound += std::fabs((double)w) * 255.0 * (double)r[2] * (double)r[3] * (m.haar_tilted[(size_t)fi] ? 2.0 : 1.0);
    }
    if (bound >= 16777216.0) int_ok = false;
    SpecStump out;
    std::map<int, std::string> var;  // LDS offset -> variable holding that word
    auto var_of = [&](int ofs) {
      auto it = var.find(ofs);
      if (it != var.end()) return it->second;
      if (reuse) {
        auto r = reuse->find(ofs);
        if (r != reuse->end()) return var[ofs] = r->second;
      }
      snprintf(buf, sizeof(buf), "x%d_%d%s", local, (int)var.size(), win.c_str());
      const std::string name = buf;
      var[ofs] = name;
      out.decls += (out.decls.empty() ? "unsigned " : ", ") + name;
      snprintf(buf, sizeof(buf), "%s = (unsigned)%s[%d]; ", name.c_str(), tile_ptr.c_str(), ofs);
      out.loads += buf;
      return name;
    };
    std::string e = "{ float v = ";
    if (h16) {
      // 16-bit tile (TileGeom16). Range of the integer value V = sum_j w_j * S_j over all images: pixel p contributes
      // net(p) * I(p), I in [0, 255]. If [Vmin, Vmax] fits int16, V is the sign-extended low half of the same corner
      // combination computed with the 16-bit entries (the dropped high halves only add multiples of 2^16). Otherwise
      // every rectangle is summed exactly from strips whose sums fit 16 bits, and the strips' sums are combined in 32 bits.
      long long vmin = 0, vmax = 0;
      if (int_ok) {
        std::vector<int> net((size_t)(m.win_w + 1) * (size_t)(m.win_h + 1), 0);
        for (int j = 0; j < d.nrect; j++) {
          const int32_t* r = &m.haar_rects[(size_t)fi * 12 + j * 4];
          for (int yy = r[1]; yy < r[1] + r[3]; yy++)
            for (int xx = r[0]; xx < r[0] + r[2]; xx++) net[(size_t)yy * (size_t)(m.win_w + 1) + (size_t)xx] += (int)d.w[j];
        }
        for (int v : net) (v > 0 ? vmax : vmin) += 255LL * v;
      }
      if (int_ok && vmin >= -32768 && vmax <= 32767) {
        std::map<int, int> coef;  // 16-bit tile offset -> integer coefficient
        static const int sign[4] = {1, -1, -1, 1};
        for (int j = 0; j < d.nrect; j++) {
          const int32_t* r = &m.haar_rects[(size_t)fi * 12 + j * 4];
          const int o4[4] = {G16.at(r[1], r[0]), G16.at(r[1], r[0] + r[2]), G16.at(r[1] + r[3], r[0]), G16.at(r[1] + r[3], r[0] + r[2])};
          for (int k = 0; k < 4; k++) coef[o4[k]] += sign[k] * (int)d.w[j];
        }
        std::map<int, std::vector<int>> by_coef;
        for (auto& kv : coef)
          if (kv.second) by_coef[std::abs(kv.second)].push_back(kv.second > 0 ? kv.first + 1 : -(kv.first + 1));
        std::string tt;
        for (auto& g : by_coef) {
          std::string grp;
          for (int so : g.second) {
            grp += so > 0 ? (grp.empty() ? "" : " + ") : " - ";
            grp += var_of(std::abs(so) - 1);
          }
          if (grp.rfind(" - ", 0) == 0) grp = "0u" + grp;
          snprintf(buf, sizeof(buf), "%s%du * (", tt.empty() ? "" : " + ", g.first);
          tt += buf + grp + ")";
        }
        if (tt.empty()) tt = "0u";
        e += "(float)(int)(short)(" + tt + ")";
      } else {
        std::string terms_int, terms_float;
        for (int j = 0; j < d.nrect; j++) {
          const int32_t* r = &m.haar_rects[(size_t)fi * 12 + j * 4];
          std::string rj;
          for (const auto& pc : pieces16(r[0], r[1], r[2], r[3])) {
            const std::string a = var_of(G16.at(pc[1], pc[0])), b2 = var_of(G16.at(pc[1], pc[0] + pc[2])), c = var_of(G16.at(pc[1] + pc[3], pc[0])),
                              dd = var_of(G16.at(pc[1] + pc[3], pc[0] + pc[2]));
            rj += std::string(rj.empty() ? "" : " + ") + "((" + a + " - " + b2 + " - " + c + " + " + dd + ") & 0xffffu)";
          }
          if (rj.empty()) rj = "0u";
          snprintf(buf, sizeof(buf), "%s%d * (int)(", j ? " + " : "", (int)d.w[j]);
          terms_int += buf + rj + ")";
          terms_float += std::string(j ? " + " : "") + hexf(d.w[j]) + " * (float)(int)(" + rj + ")";
        }
        e += int_ok ? "(float)(" + terms_int + ")" : terms_float;
      }
    } else if (int_ok) {
      std::map<int, int> coef;  // LDS offset -> integer coefficient
      static const int sign[4] = {1, -1, -1, 1};
      for (int j = 0; j < d.nrect; j++)
        for (int k = 0; k < 4; k++) coef[d.ofs[j][k]] += sign[k] * (int)d.w[j];
      std::map<int, std::vector<int>> by_coef;  // |coefficient| -> signed offsets (+ofs+1 / -(ofs+1))
      for (auto& kv : coef)
        if (kv.second) by_coef[std::abs(kv.second)].push_back(kv.second > 0 ? kv.first + 1 : -(kv.first + 1));
      std::string tt;
      for (auto& g : by_coef) {
        std::string grp;
        for (int so : g.second) {
          grp += so > 0 ? (grp.empty() ? "" : " + ") : " - ";
          grp += var_of(std::abs(so) - 1);
        }
        if (grp.rfind(" - ", 0) == 0) grp = "0u" + grp;
        snprintf(buf, sizeof(buf), "%s%du * (", tt.empty() ? "" : " + ", g.first);  // unsigned: wrap-around is defined
        tt += buf + grp + ")";
      }
      if (tt.empty()) tt = "0u";
      e += "(float)(int)(" + tt + ")";
    } else {
      for (int j = 0; j < d.nrect; j++) {
        const std::string a = var_of(d.ofs[j][0]), b2 = var_of(d.ofs[j][1]), c = var_of(d.ofs[j][2]), dd = var_of(d.ofs[j][3]);
        e += std::string(j ? " + " : "") + hexf(d.w[j]) + " * (float)(int)(" + a + " - " + b2 + " - " + c + " + " + dd + ")";
      }
    }
    if (out.decls.empty()) out.decls = "";
    else out.decls += ";";
    if (const char* dbg = std::getenv("CCAMD_DEBUG_SPEC_MODE")) {
      // Sensitivity experiments (tools/sweeps): extra work whose results are thrown away, decisions unchanged.
      // 3 = every LDS read issued twice; 4 = the value arithmetic done twice. Measured on the headline bench:
      // mode 3 costs +68 % kernel time, mode 4 +1 %: the kernel is bound by the LDS pipeline, not by VALU issue.
      const int mode = std::atoi(dbg);
      if (mode == 3) {
        std::string dup;
        int k = 0;
        for (auto& kv : var) {
          snprintf(buf, sizeof(buf), "{ unsigned dz%d = (unsigned)%s[%d]; asm volatile(\"\" :: \"v\"(dz%d)); } ", k, tile_ptr.c_str(), kv.first ^ 1, k);
          dup += buf;
          k++;
        }
        out.compute = dup + " ";
      } else if (mode == 4) {
        // the same operations on operands XOR-ed with a value the compiler cannot see through (an added constant would
        // cancel in a - b - c + d and the copy would be merged with the original)
        std::string e2 = e;
        for (auto& kv : var) {
          size_t pos = 0;
          const std::string from = kv.second, to = "(" + kv.second + " ^ __float_as_uint(vnf" + win + "))";
          while ((pos = e2.find(from, pos)) != std::string::npos) {
            const char next = pos + from.size() < e2.size() ? e2[pos + from.size()] : ' ';
            if (next >= '0' && next <= '9') {
              pos += from.size();
              continue;
            }
            e2.replace(pos, from.size(), to);
            pos += to.size();
          }
        }
        out.compute = e2 + "; v *= vnf" + win + "; asm volatile(\"\" :: \"v\"(v)); } ";
      }
    }
    out.compute += e + vote_text(d, win, fixed_q, out);
    if (vars_out) *vars_out = var;
    return out;
  };
  // Corners shared between the stumps of a stage. A quarter of a late stage's corner reads fetch a word another stump of
  // the stage reads too (25x25 possible corners, 360-650 reads), but almost never the stump next to it. Where the stage sum
  // is exact (fixed-point votes: any order gives the same sum) the stumps are therefore re-ordered greedily -- next comes the
  // stump that shares most corners with the one before it -- and a stump takes those words from its predecessor's variables
  // instead of reading them again: 5-15 % fewer LDS reads in stages 1-7 of the bench cascade for one stump's worth of longer
  // live ranges. Not across the parts of a stage: a stump-split call starts at a part boundary with nothing loaded.
  const bool share_corners = !std::getenv("CCAMD_SPEC_NO_SHARE");
  int share_window = 1;  // a stump may take words from this many stumps before it
  if (const char* e = std::getenv("CCAMD_SPEC_SHARE_WINDOW")) share_window = std::max(1, std::min(8, std::atoi(e)));  // tuning
  auto corner_set = [&](const HaarStumpDev& d) {
    std::map<int, int> coef;
    static const int sign[4] = {1, -1, -1, 1};
    for (int j = 0; j < d.nrect; j++)
      for (int k = 0; k < 4; k++) coef[d.ofs[j][k]] += sign[k];
    std::vector<int> v;
    for (auto& kv : coef) v.push_back(kv.first);
    return v;
  };
  auto sharing_order = [&](int s, int step) {
    const int nt = m.stage_ntrees[(size_t)s], f0 = m.stage_first[(size_t)s];
    std::vector<std::vector<int>> pts((size_t)nt);
    for (int i = 0; i < nt; i++) pts[(size_t)i] = corner_set(t[step - 1][(size_t)f0 + i]);
    auto shared_with = [&](int i, const std::vector<int>& recent) {
      int n = 0;
      for (int o : pts[(size_t)i]) n += std::binary_search(recent.begin(), recent.end(), o) ? 1 : 0;
      return n;
    };
    auto part_start = [&](int n) {
      for (int k = 0; k < SPEC_PARTS; k++)
        if (n == (int)((long long)k * nt / SPEC_PARTS)) return true;
      return false;
    };
    std::vector<int> best_order;
    int best_total = -1;
    for (int start = 0; start < nt; start++) {  // greedy chain from every start; the one that saves most reads wins
      std::vector<int> order{start};
      std::vector<char> used((size_t)nt, 0);
      used[(size_t)start] = 1;
      int total = 0;
      for (int n = 1; n < nt; n++) {
        std::vector<int> recent;  // corners of the last `share_window` stumps
        for (int k = 1; k <= share_window && n - k >= 0; k++) {
          const std::vector<int>& q = pts[(size_t)order[(size_t)(n - k)]];
          recent.insert(recent.end(), q.begin(), q.end());
        }
        std::sort(recent.begin(), recent.end());
        int best = -1, best_shared = -1;
        for (int i = 0; i < nt; i++) {
          if (used[(size_t)i]) continue;
          const int sh = shared_with(i, recent);
          if (sh > best_shared) {
            best_shared = sh;
            best = i;
          }
        }
        used[(size_t)best] = 1;
        order.push_back(best);
        if (!part_start(n)) total += best_shared;  // nothing is carried across a part boundary
      }
      if (total > best_total) {
        best_total = total;
        best_order.swap(order);
      }
    }
    return best_order;
  };
  // Text that follows a stump's value expression "{ float v = ...": normalisation and the vote into the accumulator of
  // window `win` (closes the brace); records the constant part of a delta-form vote in `out`.
  vote_text = [&](const HaarStumpDev& d, const std::string& win, double fixed_q, SpecStump& out) -> std::string {
    if (delta_form && fixed_q > 0.) {
      char delta[64];
      const long long lq = (long long)std::llround((double)d.left / fixed_q), rq = (long long)std::llround((double)d.right / fixed_q);
      // The delta reaches the select through a volatile move (left to itself the compiler hoists hundreds of these
      // constant moves out of the stage loops and spills them), and the accumulator passes through an empty asm after
      // every vote: integer adds are associative, and without it the compiler re-associates the chain of votes into a
      // tree of partial sums that it then has to spill.
      snprintf(delta, sizeof(delta), "0x%08x", (unsigned)(lq - rq));
      char vote[512];
      if (!std::getenv("CCAMD_SPEC_NO_CMPX")) {
        // The vote as TWO vector instructions: v_cmpx narrows EXEC to the lanes with v < thr (threshold as a 32-bit
        // literal operand), the delta is added under that mask (again a literal operand), and a scalar move puts EXEC back.
        // The plain form below costs four (move of the delta into a register, compare, select, add) plus a scalar move of
        // the threshold. `thr > v` is the comparison `v < thr` with the operands swapped: false for NaN either way.
        // (A three-instruction form without EXEC traffic -- compare into VCC, v_cndmask of a literal delta against a zero
        // register, add -- does not assemble: a VOP2 with a literal AND the implicit VCC read exceeds gfx9's constant bus.)
        unsigned thr_bits;
        std::memcpy(&thr_bits, &d.thr, 4);
        snprintf(vote, sizeof(vote),
                 "; v *= vnf%s; { unsigned long long sx; asm volatile(\"s_mov_b64 %%1, exec\\n\\tv_cmpx_gt_f32_e32 0x%08x, %%2\\n\\tv_add_u32_e32 %%0, %s, %%0\\n\\ts_mov_b64 exec, %%1\" "
                 ": \"+v\"(ai%s), \"=&s\"(sx) : \"v\"(v) : \"vcc\"); } }",
                 win.c_str(), thr_bits, delta, win.c_str());
      } else
      snprintf(vote, sizeof(vote), "; v *= vnf%s; { unsigned dq; asm volatile(\"v_mov_b32_e32 %%0, %s\" : \"=v\"(dq)); ai%s += (v < %s ? dq : 0u); asm volatile(\"\" : \"+v\"(ai%s)); } }",
               win.c_str(), delta, win.c_str(), hexf(d.thr).c_str(), win.c_str());
      out.base = (double)d.right;
      out.base_q = rq;
      return vote;
    }
    if (delta_form) {  // vote = right + (v < thr ? left - right : 0): the constant `right` is added once per part
      char delta[64];  // (hexf reuses `buf`)
      snprintf(delta, sizeof(delta), "%a", (double)d.left - (double)d.right);
      out.base = (double)d.right;
      return "; v *= vnf" + win + "; acc" + win + " += (v < " + hexf(d.thr) + " ? " + delta + " : 0.); }";
    }
    return "; v *= vnf" + win + "; acc" + win + " += (double)(v < " + hexf(d.thr) + " ? " + hexf(d.left) + " : " + hexf(d.right) + "); }";
  };
  // One stump for the two windows of a pair-tile slot (TileGeomP): every corner word is read once and holds both windows'
  // entries, the corner combinations are packed 16-bit arithmetic, the two halves then go their own way (conversion,
  // normalisation, vote) into acca / accb. Same cases as the 16-bit tile above.
  auto stump_pair = [&](int stump_index, int local, double fixed_q) {
    const HaarStumpDev& d = t[1][(size_t)stump_index];
    const int fi = m.stump_feature[(size_t)stump_index];
    bool int_ok = true;
    double bound = 0;
    for (int j = 0; j < d.nrect; j++) {
      const float w = d.w[j];
      const int32_t* r = &m.haar_rects[(size_t)fi * 12 + j * 4];
      if (w != std::nearbyint(w) || std::fabs(w) > 64.f) int_ok = false;
      bound += std::fabs((double)w) * 255.0 * (double)r[2] * (double)r[3];
    }
    if (bound >= 16777216.0) int_ok = false;
    SpecStump out;
    std::map<int, std::string> var;
    auto var_of = [&](int ofs) {
      auto it = var.find(ofs);
      if (it != var.end()) return it->second;
      snprintf(buf, sizeof(buf), "x%d_%d", local, (int)var.size());
      const std::string name = buf;
      var[ofs] = name;
      out.decls += (out.decls.empty() ? "cc_us2 " : ", ") + name;
      snprintf(buf, sizeof(buf), "%s = cc_pk(b[%d]); ", name.c_str(), ofs);
      out.loads += buf;
      return name;
    };
    long long vmin = 0, vmax = 0;
    if (int_ok) {
      std::vector<int> net((size_t)(m.win_w + 1) * (size_t)(m.win_h + 1), 0);
      for (int j = 0; j < d.nrect; j++) {
        const int32_t* r = &m.haar_rects[(size_t)fi * 12 + j * 4];
        for (int yy = r[1]; yy < r[1] + r[3]; yy++)
          for (int xx = r[0]; xx < r[0] + r[2]; xx++) net[(size_t)yy * (size_t)(m.win_w + 1) + (size_t)xx] += (int)d.w[j];
      }
      for (int v : net) (v > 0 ? vmax : vmin) += 255LL * v;
    }
    std::string pre, ea, eb;  // packed part, value of the first / second window
    if (int_ok && vmin >= -32768 && vmax <= 32767) {
      std::map<int, int> coef;
      static const int sign[4] = {1, -1, -1, 1};
      for (int j = 0; j < d.nrect; j++) {
        const int32_t* r = &m.haar_rects[(size_t)fi * 12 + j * 4];
        const int o4[4] = {GP.at(r[1], r[0]), GP.at(r[1], r[0] + r[2]), GP.at(r[1] + r[3], r[0]), GP.at(r[1] + r[3], r[0] + r[2])};
        for (int k = 0; k < 4; k++) coef[o4[k]] += sign[k] * (int)d.w[j];
      }
      std::map<int, std::vector<int>> by_coef;
      for (auto& kv : coef)
        if (kv.second) by_coef[std::abs(kv.second)].push_back(kv.second > 0 ? kv.first + 1 : -(kv.first + 1));
      std::string tt;
      for (auto& g : by_coef) {
        std::string grp;
        for (int so : g.second) {
          grp += so > 0 ? (grp.empty() ? "" : " + ") : " - ";
          grp += var_of(std::abs(so) - 1);
        }
        if (grp.rfind(" - ", 0) == 0) grp = "cc_k2(0)" + grp;
        if (g.first == 1)
          tt += (tt.empty() ? "(" : " + (") + grp + ")";
        else {
          snprintf(buf, sizeof(buf), "%scc_k2(%d) * (", tt.empty() ? "" : " + ", g.first);
          tt += buf + grp + ")";
        }
      }
      if (tt.empty()) tt = "cc_k2(0)";
      pre = "const cc_us2 t = " + tt + "; ";
      ea = "(float)(int)(short)t.x";
      eb = "(float)(int)(short)t.y";
    } else {
      std::string ia, ib, fa, fb;
      int np = 0;
      for (int j = 0; j < d.nrect; j++) {
        const int32_t* r = &m.haar_rects[(size_t)fi * 12 + j * 4];
        std::string ra, rb;
        for (const auto& pc : pieces16(r[0], r[1], r[2], r[3])) {
          const std::string a = var_of(GP.at(pc[1], pc[0])), b2 = var_of(GP.at(pc[1], pc[0] + pc[2])), c = var_of(GP.at(pc[1] + pc[3], pc[0])),
                            dd = var_of(GP.at(pc[1] + pc[3], pc[0] + pc[2]));
          snprintf(buf, sizeof(buf), "q%d", np++);
          const std::string q = buf;
          pre += "const cc_us2 " + q + " = " + a + " - " + b2 + " - " + c + " + " + dd + "; ";
          ra += std::string(ra.empty() ? "" : " + ") + "(int)" + q + ".x";
          rb += std::string(rb.empty() ? "" : " + ") + "(int)" + q + ".y";
        }
        if (ra.empty()) ra = rb = "0";
        snprintf(buf, sizeof(buf), "%s%d * (", j ? " + " : "", (int)d.w[j]);
        ia += buf + ra + ")";
        ib += buf + rb + ")";
        fa += std::string(j ? " + " : "") + hexf(d.w[j]) + " * (float)(" + ra + ")";
        fb += std::string(j ? " + " : "") + hexf(d.w[j]) + " * (float)(" + rb + ")";
      }
      ea = int_ok ? "(float)(" + ia + ")" : fa;
      eb = int_ok ? "(float)(" + ib + ")" : fb;
    }
    if (!out.decls.empty()) out.decls += ";";
    SpecStump other;  // both windows vote with the same constants
    out.compute = "{ " + pre + "{ float v = " + ea + vote_text(d, "a", fixed_q, out) + " { float v = " + eb + vote_text(d, "b", fixed_q, other) + " }";
    return out;
  };
  for (int step = 1; step <= 2; step++) {
    if (step == 2 && tmode == TILE_PAIR16) {  // STEP-2 tiles hold window pairs: one function serves the dense and the thread phase
      snprintf(buf, sizeof(buf),
               "template <>\n__device__ %s void spec_stage_pair<2>(int st, int p_lo, int p_hi, const int32_t* b, float vnfa, float vnfb, double& "
               "acc_a, double& acc_b) {\n  double acca = 0., accb = 0.;\n  switch (st) {\n",
               spec_stage_inline_attr());
      o += buf;
      for (int s = 0; s < n_stages; s++) {
        snprintf(buf, sizeof(buf), "    case %d: {\n", s);
        o += buf;
        std::vector<SpecStump> st;
        double q = 0.;
        const bool fixed = delta_form && fixed_point_ok && stage_quantum(s, q);
        for (int i = 0; i < m.stage_ntrees[(size_t)s]; i++) st.push_back(stump_pair(m.stage_first[(size_t)s] + i, i, fixed ? q : 0.));
        if (fixed) {
          o += "      unsigned aia = 0u, aib = 0u;\n";
          spec_emit_stage(o, st, depth, true, {"aia", "aib"}, true);
          snprintf(buf, sizeof(buf), "      acca = (double)(int)aia * %a;\n      accb = (double)(int)aib * %a;\n", q, q);
          o += buf;
        } else
          spec_emit_stage(o, st, depth, true, {"acca", "accb"});
        o += "    } break;\n";
      }
      o += "    default: break;\n  }\n  acc_a = acca;\n  acc_b = accb;\n}\n";
      continue;
    }
    snprintf(buf, sizeof(buf),
             "template <>\n__device__ %s double spec_stage<%d>(int st, int p_lo, int p_hi, const int32_t* b, float vnf) {\n", spec_stage_inline_attr(), step);
    o += buf;
    const bool h16 = tmode == TILE_16 && step == 2;  // STEP-2 tiles hold 16-bit entries
    if (h16) o += "  const unsigned short* h = reinterpret_cast<const unsigned short*>(b);\n";
    o += "  double acc = 0.;\n  switch (st) {\n";
    for (int s = 0; s < n_stages; s++) {
      snprintf(buf, sizeof(buf), "    case %d: {\n", s);
      o += buf;
      std::vector<SpecStump> st;
      double q = 0.;
      const bool fixed = delta_form && fixed_point_ok && stage_quantum(s, q);
      if (fixed && share_corners) {
        const int nt = m.stage_ntrees[(size_t)s];
        const std::vector<int> order = sharing_order(s, step);
        std::vector<std::map<int, std::string>> hist;  // variable maps of the stumps of the current part, newest last
        for (int n = 0; n < nt; n++) {
          // spec_emit_stage cuts the stage into SPEC_PARTS contiguous parts at these positions
          for (int k = 0; k < SPEC_PARTS; k++)
            if (n == (int)((long long)k * nt / SPEC_PARTS)) hist.clear();
          std::map<int, std::string> recent, cur;
          for (int k = 0; k < share_window && k < (int)hist.size(); k++)
            for (auto& kv : hist[hist.size() - 1 - (size_t)k]) recent.insert(kv);
          const int i = order[(size_t)n];
          st.push_back(stump(t[step - 1][(size_t)m.stage_first[(size_t)s] + i], m.stage_first[(size_t)s] + i, i, "", q, h16, &recent, &cur));
          hist.push_back(cur);
        }
      } else
      for (int i = 0; i < m.stage_ntrees[(size_t)s]; i++)
        st.push_back(stump(t[step - 1][(size_t)m.stage_first[(size_t)s] + i], m.stage_first[(size_t)s] + i, i, "", fixed ? q : 0., h16));
      if (fixed) {
        o += "      unsigned ai = 0u;\n";
        spec_emit_stage(o, st, depth, true, {"ai"}, true);
        snprintf(buf, sizeof(buf), "      acc = (double)(int)ai * %a;\n", q);
        o += buf;
      } else
        spec_emit_stage(o, st, depth, true, {"acc"});
      o += "    } break;\n";
    }
    o += "    default: break;\n  }\n  return acc;\n}\n";
    // stage 0 for the two windows a thread owns in the dense phase: both windows' reads of a stump travel together
    snprintf(buf, sizeof(buf),
             "template <>\n__device__ __forceinline__ void spec_stage0_x2<%d>(const int32_t* ba, const int32_t* bb, float vnfa, float vnfb, double& "
             "acc_a, double& acc_b) {\n  double acca = 0., accb = 0.;\n  {\n",
             step);
    o += buf;
    if (h16)
      o += "  const unsigned short* ha = reinterpret_cast<const unsigned short*>(ba);\n  const unsigned short* hb = reinterpret_cast<const unsigned short*>(bb);\n";
    {
      std::vector<SpecStump> st;
      double q = 0.;
      const bool fixed = delta_form && fixed_point_ok && stage_quantum(0, q);
      for (int i = 0; i < m.stage_ntrees[0]; i++) {
        const HaarStumpDev& d = t[step - 1][(size_t)m.stage_first[0] + i];
        SpecStump a = stump(d, m.stage_first[0] + i, i, "a", fixed ? q : 0., h16), b2 = stump(d, m.stage_first[0] + i, i, "b", fixed ? q : 0., h16);
        st.push_back(SpecStump{a.loads + b2.loads, a.decls + " " + b2.decls, a.compute + " " + b2.compute, a.base, a.base_q});
      }
      if (fixed) {
        o += "      unsigned aia = 0u, aib = 0u;\n";
        spec_emit_stage(o, st, depth, false, {"aia", "aib"}, true);
        snprintf(buf, sizeof(buf), "      acca = (double)(int)aia * %a;\n      accb = (double)(int)aib * %a;\n", q, q);
        o += buf;
      } else
        spec_emit_stage(o, st, depth, false, {"acca", "accb"});
    }
    o += "  }\n  acc_a = acca;\n  acc_b = accb;\n}\n";
  }
  return o;
}

// LBP variant: the 16 lattice offsets are immediates; the 256-bit subsets stay a (module-resident) table because the word
// a lane needs depends on its own code. Integer arithmetic throughout, the expression of stump_vote().
static std::string spec_stage_source_lbp(const Cascade& m, int n_stages, bool tile16) {
  std::vector<LbpStumpDev> t[2];
  build_lbp_stumps<1>(m, t[0]);
  if (tile16)
    build_lbp_stumps16(m, t[1]);
  else
    build_lbp_stumps<2>(m, t[1]);
  n_stages = std::min<int>(n_stages, (int)m.stage_ntrees.size());
  const int depth = std::min(spec_prefetch_depth(0), 2);  // 16 independent words per stump already: no explicit pipelining measured best (7.8 ms per 32 frames; one stump ahead 8.2, two 8.9)
  std::string o;
  char buf[1024];
  int n_stumps = 0;
  for (int s = 0; s < n_stages; s++) n_stumps += m.stage_ntrees[(size_t)s];
  const bool lbp_select_words = std::getenv("CCAMD_SPEC_LBP_TABLE") == nullptr;  // tuning: the table form of round 2
  o += "static __device__ const int kSpecSubsets[][8] = {\n";
  for (int i = 0; i < n_stumps; i++) {
    const LbpStumpDev& d = t[0][(size_t)i];
    snprintf(buf, sizeof(buf), "  {%d, %d, %d, %d, %d, %d, %d, %d},\n", d.subset[0], d.subset[1], d.subset[2], d.subset[3], d.subset[4], d.subset[5],
             d.subset[6], d.subset[7]);
    o += buf;
  }
  o += "};\n";
  auto hexf = [&](float v) {
    char b2[64];
    snprintf(b2, sizeof(b2), "%af", (double)v);
    return std::string(b2);
  };
  auto stump = [&](const LbpStumpDev& d, int index, int local, const std::string& win, bool h16) {
    SpecStump out;
    std::string P[16];
    for (int k = 0; k < 16; k++) {
      snprintf(buf, sizeof(buf), "p%d_%d%s", local, k, win.c_str());
      P[k] = buf;
      out.decls += (k ? ", " : "int ") + P[k];
      snprintf(buf, sizeof(buf), "%s = %s%s[%d]; ", P[k].c_str(), h16 ? "h" : "b", win.c_str(), d.ofs[k]);
      out.loads += buf;
    }
    out.decls += ";";
    // 16-bit tile: a cell sum is the low half of the corner combination (exact: 255 * cell area < 2^16, tile16_eligible)
    // Cells from horizontal differences: the 12 differences of neighbouring lattice points of a row, then one subtraction
    // per cell (21 integer operations instead of 27). CCAMD_SPEC_LBP_PLAIN_CELLS: every cell from its four corners.
    static const bool row_diffs = !std::getenv("CCAMD_SPEC_LBP_PLAIN_CELLS");
    std::string diffs;
    if (row_diffs) {
      diffs = "const int ";
      bool firstd = true;
      for (int k = 0; k < 15; k++) {
        if (k % 4 == 3) continue;
        snprintf(buf, sizeof(buf), "%sh%d_%d%s = %s - %s", firstd ? "" : ", ", local, k, win.c_str(), P[k].c_str(), P[k + 1].c_str());
        diffs += buf;
        firstd = false;
      }
      diffs += "; ";
    }
    auto cell = [&](int a, int b2, int c, int dd) {
      std::string v = P[a] + " - " + P[b2] + " - " + P[c] + " + " + P[dd];
      if (row_diffs) {
        char hb[96];
        snprintf(hb, sizeof(hb), "h%d_%d%s - h%d_%d%s", local, a, win.c_str(), local, c, win.c_str());
        v = hb;
      }
      return h16 ? "((" + v + ") & 0xffff)" : v;
    };
    std::string e = "{ " + diffs + "const int c = " + cell(5, 6, 9, 10) + "; const int lbp = (" + cell(0, 1, 4, 5) + " >= c ? 128 : 0) | (" + cell(1, 2, 5, 6) +
                    " >= c ? 64 : 0) | (" + cell(2, 3, 6, 7) + " >= c ? 32 : 0) | (" + cell(6, 7, 10, 11) + " >= c ? 16 : 0) | (" +
                    cell(10, 11, 14, 15) + " >= c ? 8 : 0) | (" + cell(9, 10, 13, 14) + " >= c ? 4 : 0) | (" + cell(8, 9, 12, 13) +
                    " >= c ? 2 : 0) | (" + cell(4, 5, 8, 9) + " >= c ? 1 : 0); ";
    if (lbp_select_words) {
      // The 256-bit subset as eight literals picked by the three top bits of the code -- the results of the first three
      // comparisons -- through seven unconditional selects, instead of a load from a table: the table word depends on the
      // lane's own code, so it is a vector memory load whose latency sits in every stump's dependency chain, and the late
      // stages (a handful of windows per tile) are nothing but that chain.
      const int* w = d.subset;
      std::string t = "{ " + diffs + "const int c = " + cell(5, 6, 9, 10) + "; const bool b7 = " + cell(0, 1, 4, 5) + " >= c, b6 = " + cell(1, 2, 5, 6) +
                      " >= c, b5 = " + cell(2, 3, 6, 7) + " >= c; const int lo = (" + cell(6, 7, 10, 11) + " >= c ? 16 : 0) | (" +
                      cell(10, 11, 14, 15) + " >= c ? 8 : 0) | (" + cell(9, 10, 13, 14) + " >= c ? 4 : 0) | (" + cell(8, 9, 12, 13) +
                      " >= c ? 2 : 0) | (" + cell(4, 5, 8, 9) + " >= c ? 1 : 0); ";
      snprintf(buf, sizeof(buf),
               "const unsigned l0 = b5 ? 0x%08xu : 0x%08xu, l1 = b5 ? 0x%08xu : 0x%08xu, l2 = b5 ? 0x%08xu : 0x%08xu, l3 = b5 ? 0x%08xu : 0x%08xu; "
               "const unsigned m0 = b6 ? l1 : l0, m1 = b6 ? l3 : l2; const unsigned sw = b7 ? m1 : m0; "
               "acc%s += (double)(((sw >> lo) & 1u) ? %s : %s); }",
               (unsigned)w[1], (unsigned)w[0], (unsigned)w[3], (unsigned)w[2], (unsigned)w[5], (unsigned)w[4], (unsigned)w[7], (unsigned)w[6],
               win.c_str(), hexf(d.left).c_str(), hexf(d.right).c_str());
      out.compute = t + buf;
      if (const char* dbg = std::getenv("CCAMD_DEBUG_SPEC_MODE")) {
        // Sensitivity experiments (as for Haar above): 3 = every corner read issued twice, 4 = the stump's arithmetic done
        // twice on operands XOR-ed with a value the compiler cannot fold; results thrown away, decisions unchanged.
        const int mode = std::atoi(dbg);
        if (mode == 3) {
          std::string dup;
          for (int k = 0; k < 16; k++) {
            snprintf(buf, sizeof(buf), "{ unsigned dz%d = (unsigned)%s%s[%d]; asm volatile(\"\" :: \"v\"(dz%d)); } ", k, h16 ? "h" : "b", win.c_str(), d.ofs[k] ^ 1, k);
            dup += buf;
          }
          out.compute = dup + out.compute;
        } else if (mode == 4) {
          std::string dup = "{ const int zz = (int)__float_as_uint(vnf" + win + ") ^ 0x3f800001; double accd = 0.; int ";
          for (int k = 0; k < 16; k++) dup += std::string(k ? ", " : "") + "d" + P[k] + " = " + P[k] + " ^ zz";
          dup += "; ";
          std::string body = out.compute;
          for (int k = 15; k >= 0; k--) {  // p<local>_<k><win> -> dp...; longest names first so that p0_1 does not hit p0_10
            size_t pos = 0;
            while ((pos = body.find(P[k], pos)) != std::string::npos) {
              const char next = pos + P[k].size() < body.size() ? body[pos + P[k].size()] : ' ';
              const bool whole = !(next >= '0' && next <= '9') && (pos == 0 || body[pos - 1] != 'd');
              if (whole) {
                body.insert(pos, "d");
                pos += P[k].size() + 1;
              } else
                pos += P[k].size();
            }
          }
          const std::string accname = "acc" + win + " +=";
          const size_t ap = body.find(accname);
          if (ap != std::string::npos) body.replace(ap, accname.size(), "accd +=");
          dup += body + " asm volatile(\"\" :: \"v\"(accd)); } ";
          out.compute = dup + out.compute;
        }
      }
      return out;
    } else
    snprintf(buf, sizeof(buf), "acc%s += (double)((kSpecSubsets[%d][lbp >> 5] & (1 << (lbp & 31))) ? %s : %s); }", win.c_str(), index,
             hexf(d.left).c_str(), hexf(d.right).c_str());
    out.compute = e + buf;
    return out;
  };
  for (int step = 1; step <= 2; step++) {
    snprintf(buf, sizeof(buf),
             "template <>\n__device__ %s double spec_stage<%d>(int st, int p_lo, int p_hi, const int32_t* b, float vnf) {\n", spec_stage_inline_attr(), step);
    o += buf;
    const bool h16 = tile16 && step == 2;  // STEP-2 tiles hold 16-bit entries
    if (h16) o += "  const unsigned short* h = reinterpret_cast<const unsigned short*>(b);\n";
    o += "  double acc = 0.;\n  switch (st) {\n";
    for (int s = 0; s < n_stages; s++) {
      snprintf(buf, sizeof(buf), "    case %d: {\n", s);
      o += buf;
      std::vector<SpecStump> st;
      for (int i = 0; i < m.stage_ntrees[(size_t)s]; i++) {
        const int idx = m.stage_first[(size_t)s] + i;
        st.push_back(stump(t[step - 1][(size_t)idx], idx, i, "", h16));
      }
      spec_emit_stage(o, st, depth, true, {"acc"});
      o += "    } break;\n";
    }
    o += "    default: break;\n  }\n  return acc;\n}\n";
    snprintf(buf, sizeof(buf),
             "template <>\n__device__ __forceinline__ void spec_stage0_x2<%d>(const int32_t* ba, const int32_t* bb, float vnfa, float vnfb, double& "
             "acc_a, double& acc_b) {\n  double acca = 0., accb = 0.;\n  {\n",
             step);
    o += buf;
    if (h16)
      o += "  const unsigned short* ha = reinterpret_cast<const unsigned short*>(ba);\n  const unsigned short* hb = reinterpret_cast<const unsigned short*>(bb);\n";
    {
      std::vector<SpecStump> st;
      for (int i = 0; i < m.stage_ntrees[0]; i++) {
        const int idx = m.stage_first[0] + i;
        SpecStump a = stump(t[step - 1][(size_t)idx], idx, i, "a", h16), b2 = stump(t[step - 1][(size_t)idx], idx, i, "b", h16);
        st.push_back(SpecStump{a.loads + b2.loads, a.decls + " " + b2.decls, a.compute + " " + b2.compute});
      }
      spec_emit_stage(o, st, std::min(depth, 1), false, {"acca", "accb"});
    }
    o += "  }\n  acc_a = acca;\n  acc_b = accb;\n}\n";
  }
  return o;
}

// Wave phase: lane l of step k evaluates stump (64 k + l) of the stage, so the 32 lanes of a half-wavefront read 32
// unrelated LDS words per corner slot (~3.8-way bank conflicts in file order). The stage sums are order-independent
// when this phase is used, so the stumps of a stage may be dealt to the lanes in any order, and the two '+' and the
// two '-' corners of a rectangle may swap slots: a greedy pass fills one 32-lane group at a time with the stump
// (and corner arrangement) that adds the fewest conflict cycles. Returns a reordered copy of the table.
static std::vector<HaarStumpDev> schedule_for_wave_phase(const Cascade& m, const std::vector<HaarStumpDev>& t) {
  std::vector<HaarStumpDev> out;
  out.reserve(t.size());
  auto variant = [](const HaarStumpDev& s, int v) {  // v: 6 bits, per rect swap of slots (0,3) and of slots (1,2)
    HaarStumpDev r = s;
    for (int j = 0; j < 3; j++) {
      if (v & (1 << (2 * j))) std::swap(r.ofs[j][0], r.ofs[j][3]);
      if (v & (2 << (2 * j))) std::swap(r.ofs[j][1], r.ofs[j][2]);
    }
    return r;
  };
  for (size_t s = 0; s < m.stage_ntrees.size(); s++) {
    const int first = m.stage_first[s], nt = m.stage_ntrees[s];
    std::vector<char> used((size_t)nt, 0);
    int left = nt;
    while (left > 0) {
      // per slot: how many distinct words each bank already holds in this 32-lane group
      std::vector<std::vector<int>> words(12 * 32);
      int mx[12] = {0};
      for (int lane = 0; lane < 32 && left > 0; lane++) {
        int best = -1, best_v = 0, best_cost = 1 << 30;
        int looked = 0;
        for (int i = 0; i < nt && looked < 48; i++) {  // bounded look-ahead keeps detector creation fast
          if (used[(size_t)i]) continue;
          looked++;
          const HaarStumpDev& c = t[(size_t)first + i];
          const int nslots = c.nrect == 3 ? 12 : 8;
          for (int v = 0; v < (c.nrect == 3 ? 64 : 16); v++) {
            const HaarStumpDev r = variant(c, v);
            int cost = 0;
            for (int k = 0; k < nslots; k++) {
              const int o = r.ofs[k >> 2][k & 3];
              const std::vector<int>& w = words[(size_t)k * 32 + (o & 31)];
              const bool present = std::find(w.begin(), w.end(), o) != w.end();
              const int load = (int)w.size() + (present ? 0 : 1);
              if (load > mx[k]) cost += load - mx[k];
            }
            if (cost < best_cost) {
              best_cost = cost;
              best = i;
              best_v = v;
              if (cost == 0) break;
            }
          }
          if (best_cost == 0) break;
        }
        const HaarStumpDev r = variant(t[(size_t)first + best], best_v);
        const int nslots = r.nrect == 3 ? 12 : 8;
        for (int k = 0; k < nslots; k++) {
          const int o = r.ofs[k >> 2][k & 3];
          std::vector<int>& w = words[(size_t)k * 32 + (o & 31)];
          if (std::find(w.begin(), w.end(), o) == w.end()) w.push_back(o);
          mx[k] = std::max(mx[k], (int)w.size());
        }
        out.push_back(r);
        used[(size_t)best] = 1;
        left--;
      }
    }
  }
  return out;
}

template <int STEP>
static void build_haar_nodes(const Cascade& m, std::vector<HaarNodeDev>& out) {
  const TileGeom<STEP> G(m.win_w, m.win_h);
  out.resize(m.node_feature.size());
  for (size_t i = 0; i < out.size(); i++) {
    HaarNodeDev& d = out[i];
    std::memset(&d, 0, sizeof(d));
    const int fi = m.node_feature[i];
    d.nrect = 2;
    for (int j = 0; j < 3; j++) {
      const int32_t* r = &m.haar_rects[(size_t)fi * 12 + j * 4];
      const float wt = m.haar_weights[(size_t)fi * 3 + j];
      d.w[j] = wt;
      const bool used = j < 2 || wt != 0.0f;
      if (j == 2 && wt != 0.0f) d.nrect = 3;
      const int x = used ? r[0] : 0, y = used ? r[1] : 0, rw = used ? r[2] : 0, rh = used ? r[3] : 0;
      if (!m.haar_tilted[fi]) {
        d.ofs[j][0] = G.at(y, x);
        d.ofs[j][1] = G.at(y, x + rw);
        d.ofs[j][2] = G.at(y + rh, x);
        d.ofs[j][3] = G.at(y + rh, x + rw);
      } else {
        const int shift = tile_words_padded(G.words());
        d.ofs[j][0] = shift + G.at(y, x);
        d.ofs[j][1] = shift + G.at(y + rh, x - rh);
        d.ofs[j][2] = shift + G.at(y + rw, x + rw);
        d.ofs[j][3] = shift + G.at(y + rw + rh, x + rw - rh);
      }
    }
    d.thr = m.node_threshold[i];
    d.left = m.node_left[i];
    d.right = m.node_right[i];
  }
}

template <int STEP>
static void build_lbp_nodes(const Cascade& m, std::vector<LbpNodeDev>& out) {
  const TileGeom<STEP> G(m.win_w, m.win_h);
  out.resize(m.node_feature.size());
  for (size_t i = 0; i < out.size(); i++) {
    LbpNodeDev& d = out[i];
    std::memset(&d, 0, sizeof(d));
    const int32_t* r = &m.lbp_rects[(size_t)m.node_feature[i] * 4];
    for (int rr = 0; rr < 4; rr++)
      for (int cc = 0; cc < 4; cc++) d.ofs[4 * rr + cc] = G.at(r[1] + rr * r[3], r[0] + cc * r[2]);
    d.left = m.node_left[i];
    d.right = m.node_right[i];
    for (int j = 0; j < 8; j++) d.subset[j] = m.node_subset[i * 8 + j];
  }
}

template <class At>
static void build_lbp_stumps_at(const Cascade& m, std::vector<LbpStumpDev>& out, At at) {
  out.resize(m.stump_feature.size());
  for (size_t i = 0; i < out.size(); i++) {
    LbpStumpDev& d = out[i];
    std::memset(&d, 0, sizeof(d));
    const int32_t* r = &m.lbp_rects[(size_t)m.stump_feature[i] * 4];
    for (int rr = 0; rr < 4; rr++)
      for (int cc = 0; cc < 4; cc++) d.ofs[4 * rr + cc] = at(r[1] + rr * r[3], r[0] + cc * r[2]);
    d.left = m.stump_left[i];
    d.right = m.stump_right[i];
    for (int j = 0; j < 8; j++) d.subset[j] = m.node_subset[i * 8 + j];
  }
}
template <int STEP>
static void build_lbp_stumps(const Cascade& m, std::vector<LbpStumpDev>& out) {
  const TileGeom<STEP> G(m.win_w, m.win_h);
  build_lbp_stumps_at(m, out, [&](int y, int x) { return G.at(y, x); });
}
static void build_lbp_gstumps(const Cascade& m, std::vector<LbpStumpDev>& out) { build_lbp_stumps_at(m, out, window_xy); }
static void build_lbp_stumps16(const Cascade& m, std::vector<LbpStumpDev>& out) {  // STEP-2 tile with 16-bit entries
  const TileGeom16 G(m.win_w, m.win_h);
  build_lbp_stumps_at(m, out, [&](int y, int x) { return G.at(y, x); });
}

static bool same_params(const cc_detect_params& a, const cc_detect_params& b) {
  return a.scale_factor == b.scale_factor && a.min_w == b.min_w && a.min_h == b.min_h && a.max_w == b.max_w && a.max_h == b.max_h;
}

// Tile list {scale, tx, ty, 0} of a plan for tiles of `tile_y` window rows.
// Workgroups are dealt round-robin over the 8 XCDs (blocks b and b+8 share an L2): the list is permuted so that the tiles
// one XCD receives are neighbours in the image and share their halo rows/columns in that XCD's L2. Placement only changes
// speed, never results.
static int tile_list_key(int tile_y, int only_step) { return tile_y * 4 + only_step; }  // Plan::other_tiles
static int debug_only_step() {  // timing experiments (CCAMD_DEBUG_ONLY_STEP=1|2): only the tiles of STEP-1 / STEP-2 scales are evaluated
  static const int v = []() {
    const char* e = std::getenv("CCAMD_DEBUG_ONLY_STEP");
    return e ? std::atoi(e) : 0;
  }();
  return v;
}
static std::vector<int4> plan_tile_list(const std::vector<ScaleGeom>& geom, int tile_y, int only_step = 0) {
  std::vector<int4> tiles;
  for (size_t i = 0; i < geom.size(); i++) {
    const ScaleGeom& g = geom[i];
    if (debug_only_step() && g.ystep != debug_only_step()) continue;
    if (only_step && g.ystep != only_step) continue;
    const int ntx = (g.nx + TILE_X - 1) / TILE_X, nty = (g.ny + tile_y - 1) / tile_y;
    for (int ty_ = 0; ty_ < nty; ty_++)
      for (int tx_ = 0; tx_ < ntx; tx_++) tiles.push_back(make_int4((int)i, tx_, ty_, 0));
  }
  const size_t n = tiles.size(), per = (n + 7) / 8;
  std::vector<int4> perm;
  perm.reserve(n);
  for (size_t j = 0; j < per; j++)
    for (size_t x = 0; x < 8; x++) {
      const size_t src = x * per + j;
      if (src < n) perm.push_back(tiles[src]);
    }
  return perm;
}

static cc_status build_plan(cc_detector* d, int w, int h, const cc_detect_params& p, Plan** out) {
  for (auto& pl : d->plans)
    if (pl->w == w && pl->h == h && same_params(pl->p, p)) {
      *out = pl.get();
      return CC_OK;
    }
  std::unique_ptr<Plan> P(new Plan());
  P->w = w;
  P->h = h;
  P->p = p;
  scale_plan(d->m.win_w, d->m.win_h, w, h, p, P->geom);
  const int ns = (int)P->geom.size();
  std::vector<int> resize_first(ns + 1, 0), band_first(ns + 1, 0), col_first(ns + 1, 0), gridrow_first(ns + 1, 0);
  std::vector<int> diag_first(ns + 1, 0), tcol_first(ns + 1, 0);
  long long h_ofs = 0;
  std::vector<int> xofs, yofs;
  std::vector<uint16_t> xw1, yw1;
  long long img_ofs = 0, int_ofs = 0, mask_ofs = 0, win_ofs = 0;
  P->sd.resize(ns);
  for (int i = 0; i < ns; i++) {
    const ScaleGeom& g = P->geom[i];
    ScaleDev& S = P->sd[i];
    S.w = g.w;
    S.h = g.h;
    S.pitch8 = align_up(g.w, 4);
    S.pitchI = align_up(g.w + 1, 4);
    S.img_ofs = img_ofs;
    S.int_ofs = int_ofs;
    S.mask_ofs = mask_ofs;
    S.win_ofs = win_ofs;
    S.ystep = g.ystep;
    S.nx = g.nx;
    S.ny = g.ny;
    S.nxw = (g.nx + 63) / 64;
    S.scale = g.scale;
    S.win_w = g.win_w;
    S.win_h = g.win_h;
    S.ytab_ofs = (int)yofs.size();
    AxisTaps tx, ty;
    linear_exact_taps(w, g.w, tx);
    linear_exact_taps(h, g.h, ty);
    S.xtab_ofs = append_column_taps(tx, xofs, xw1);
    yofs.insert(yofs.end(), ty.ofs.begin(), ty.ofs.end());
    yw1.insert(yw1.end(), ty.w1.begin(), ty.w1.end());
    img_ofs += (long long)align_up(S.pitch8 * g.h, 16);
    int_ofs += (long long)S.pitchI * (g.h + 1);
    mask_ofs += (long long)S.nxw * g.ny;
    win_ofs += (long long)g.nx * g.ny;
    P->integral_elems += (long long)(g.w + 1) * (g.h + 1);
    resize_first[i + 1] = resize_first[i] + resize_blocks(S.pitch8, g.h);
    S.nbands = (g.h + INT_BAND - 1) / INT_BAND;
    S.h_ofs = h_ofs;
    h_ofs += (long long)S.nbands * S.pitchI;
    band_first[i + 1] = band_first[i] + S.nbands;
    col_first[i + 1] = col_first[i] + (S.pitchI / 4 + 63) / 64;
    gridrow_first[i + 1] = gridrow_first[i] + g.ny;
    diag_first[i + 1] = diag_first[i] + (g.w + g.h - 1 + 255) / 256;  // k_diag_sums: a thread walks 4 diagonals
    tcol_first[i + 1] = tcol_first[i] + (g.w + 1 + 63) / 64;
  }
  P->pyr_frame_bytes = (size_t)((img_ofs + 15) & ~15LL);
  P->int_frame_elems = (size_t)int_ofs;
  P->mask_frame_words = (size_t)mask_ofs;
  P->windows = win_ofs;
  P->n_resize_blocks = resize_first[ns];
  P->n_bands = band_first[ns];
  P->h_frame_elems = (size_t)h_ofs;
  P->n_col_blocks = col_first[ns];
  P->n_grid_rows = gridrow_first[ns];
  P->n_diag_blocks = diag_first[ns];
  P->n_tcol_blocks = tcol_first[ns];
  std::vector<int4> tiles = plan_tile_list(P->geom, TILE_Y);
  P->n_tiles = (int)tiles.size();
  hipStream_t st = d->stream;
  CC_HIP(P->d_sd.upload(P->sd, st));
  CC_HIP(P->d_resize_first.upload(resize_first, st));
  CC_HIP(P->d_band_first.upload(band_first, st));
  CC_HIP(P->d_col_first.upload(col_first, st));
  CC_HIP(P->d_gridrow_first.upload(gridrow_first, st));
  CC_HIP(P->d_diag_first.upload(diag_first, st));
  CC_HIP(P->d_tcol_first.upload(tcol_first, st));
  CC_HIP(P->d_xofs.upload(xofs, st));
  CC_HIP(P->d_yofs.upload(yofs, st));
  CC_HIP(P->d_xw1.upload(xw1, st));
  CC_HIP(P->d_yw1.upload(yw1, st));
  CC_HIP(P->d_tiles.upload(tiles, st));
  if (d->m.feature_type == CC_FEATURE_HAAR && d->m.has_tilted) CC_HIP(P->tilt.build(P->sd, st));
  CC_HIP(hipStreamSynchronize(st));  // host vectors go out of scope
  *out = P.get();
  if (d->plans.size() >= 8) d->plans.erase(d->plans.begin());
  d->plans.push_back(std::move(P));
  return CC_OK;
}

enum { EV_RESIZE = 0, EV_INTEGRAL = 1, EV_EVAL = 2, EV_FILTER = 3, EV_EVAL_STEP1 = 4 };

struct EvScope {  // records a pair of events around a group of launches when profiling is on
  cc_detector* d;
  int kind;
  hipEvent_t a = nullptr, b = nullptr;
  hipStream_t st;
  EvScope(cc_detector* d_, int kind_, hipStream_t st_) : d(d_), kind(kind_), st(st_) {
    if (!d->profiling || kind < 0) return;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) {
      a = b = nullptr;
      return;
    }
    (void)hipEventRecord(a, st);
  }
  ~EvScope() {
    if (!a) return;
    (void)hipEventRecord(b, st);
    d->events.push_back(TimingEvent{a, b, kind});
  }
};

static void collect_events(cc_detector* d) {
  for (auto& e : d->events) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) {
      switch (e.kind) {
        case EV_RESIZE: d->tm.resize_ms += ms; d->tm.resize_launches++; break;
        case EV_INTEGRAL: d->tm.integral_ms += ms; d->tm.integral_launches++; break;
        case EV_EVAL: d->tm.eval_ms += ms; d->tm.eval_launches++; break;
        case EV_EVAL_STEP1: d->tm.eval_step1_ms += ms; break;
        case EV_FILTER: d->tm.finalize_ms += ms; d->tm.finalize_launches++; break;
      }
    }
    (void)hipEventDestroy(e.a);
    (void)hipEventDestroy(e.b);
  }
  d->events.clear();
}

// Integral images of every scale of nf frames: band totals, carry down the bands, finished integral.
static void launch_integral(hipStream_t st, bool sq, const uint8_t* pyr, size_t pyr_frame_bytes, int32_t* integ,
                            size_t int_frame_elems, int nchan /* channel stride of integ / hbuf */, int32_t* hbuf, size_t h_frame_elems, const ScaleDev* sd, int ns,
                            const int* band_first, int n_bands, const int* col_first, int n_col_blocks, int nf, int sq_odd_rows_only = 0) {
  const dim3 grid((n_bands + 3) / 4, nf);
  if (sq)
    hipLaunchKernelGGL((k_integral_band<true, false>), grid, dim3(256), 0, st, pyr, pyr_frame_bytes, integ, int_frame_elems, nchan,
                       hbuf, h_frame_elems, sd, ns, band_first, n_bands, 0);
  else
    hipLaunchKernelGGL((k_integral_band<false, false>), grid, dim3(256), 0, st, pyr, pyr_frame_bytes, integ, int_frame_elems, nchan,
                       hbuf, h_frame_elems, sd, ns, band_first, n_bands, 0);
  hipLaunchKernelGGL(k_integral_carry, dim3(n_col_blocks, nf, sq ? 2 : 1), dim3(64), 0, st, hbuf, h_frame_elems, nchan, sd, ns, col_first);
  if (sq)
    hipLaunchKernelGGL((k_integral_band<true, true>), grid, dim3(256), 0, st, pyr, pyr_frame_bytes, integ, int_frame_elems, nchan,
                       hbuf, h_frame_elems, sd, ns, band_first, n_bands, sq_odd_rows_only);
  else
    hipLaunchKernelGGL((k_integral_band<false, true>), grid, dim3(256), 0, st, pyr, pyr_frame_bytes, integ, int_frame_elems, nchan,
                       hbuf, h_frame_elems, sd, ns, band_first, n_bands, sq_odd_rows_only);
}

// Device pipeline for up to max_batch frames already resident on the device. Leaves the filtered candidate list
// (d_out[slot], d_counts[slot][1]) on the device; no synchronisation.
static cc_status run_device_pass(cc_detector* d, Plan* P, const uint8_t* dframes, int nf, size_t row_stride,
                                 size_t frame_stride, bool debug, int slot, bool single_stream = false) {
  const int ns = (int)P->sd.size();
  hipStream_t st = d->stream;
  hipStream_t fs = d->overlap_front && !single_stream ? d->front_stream : d->stream;  // pyramid + integrals
  const bool haar = d->m.feature_type == CC_FEATURE_HAAR;
  const bool tilt = haar && d->m.has_tilted;
  const int nchan = haar ? (tilt ? 3 : 2) : 1;  // sum, sqsum, tilted
  CC_HIP(d->d_counts[slot].ensure(2));
  CC_HIP(hipMemsetAsync(d->d_counts[slot].p, 0, 2 * sizeof(int), st));
  if (ns == 0 || nf == 0) return CC_OK;
  // even window sizes: the variance rectangle's corners of step-2 scales sit on odd rows and odd columns only
  const int sq_compact = (haar && d->m.win_w % 2 == 0 && d->m.win_h % 2 == 0 && !d->full_sqsum) ? 1 : 0;
  CC_HIP(d->d_pyr.ensure(P->pyr_frame_bytes * (size_t)d->pass_capacity));
  CC_HIP(d->d_integ[slot].ensure(P->int_frame_elems * (size_t)nchan * (size_t)d->pass_capacity));
  CC_HIP(d->d_hbuf.ensure(std::max<size_t>(P->h_frame_elems * (size_t)nchan * (size_t)d->pass_capacity, 4)));
  if (tilt) {
    CC_HIP(d->d_diag.ensure(P->int_frame_elems * 2 * (size_t)d->pass_capacity));
    CC_HIP(d->d_tseg.ensure(std::max<size_t>(P->tilt.frame_elems * (size_t)d->pass_capacity, 1)));
  }
  CC_HIP(d->d_masks[slot].ensure(std::max<size_t>(P->mask_frame_words * (size_t)d->pass_capacity, 1)));
  if (d->cand_cap == 0) d->cand_cap = 1 << 18;
  CC_HIP(d->d_cands[slot].ensure((size_t)d->cand_cap));
  CC_HIP(d->d_out[slot].ensure((size_t)d->cand_cap));
  if (debug) {
    CC_HIP(d->d_dbg_codes.ensure((size_t)std::max<long long>(P->windows, 1)));
    CC_HIP(d->d_dbg_sums.ensure((size_t)std::max<long long>(P->windows, 1)));
    CC_HIP(d->d_dbg_visited.ensure((size_t)std::max<long long>(P->windows, 1)));
  }
  // this slot's integrals may still be read by the cascade kernel launched two passes ago
  if (fs != st && d->eval_pending[slot]) CC_HIP(hipStreamWaitEvent(fs, d->eval_done[slot], 0));
  {
    EvScope ev(d, EV_RESIZE, fs);
    hipLaunchKernelGGL(k_resize, dim3(P->n_resize_blocks, nf), dim3(256), 0, fs, dframes, row_stride, frame_stride, P->w,
                       P->h, d->d_pyr.p, P->pyr_frame_bytes, P->d_sd.p, ns, P->d_resize_first.p, P->d_xofs.p, P->d_xw1.p,
                       P->d_yofs.p, P->d_yw1.p);
  }
  {
    EvScope ev(d, EV_INTEGRAL, fs);
    launch_integral(fs, haar, d->d_pyr.p, P->pyr_frame_bytes, d->d_integ[slot].p, P->int_frame_elems, nchan, d->d_hbuf.p,
                    P->h_frame_elems, P->d_sd.p, ns, P->d_band_first.p, P->n_bands, P->d_col_first.p, P->n_col_blocks, nf,
                    /*sq_odd_rows_only=*/sq_compact);
    if (tilt) {
      launch_tilted(fs, P->tilt, d->d_tseg.p, d->d_pyr.p, P->pyr_frame_bytes, d->d_diag.p, d->d_integ[slot].p, P->int_frame_elems, nchan, 2,
                    P->d_sd.p, ns, P->d_diag_first.p, P->n_diag_blocks, P->d_tcol_first.p, P->n_tcol_blocks, nf);
    }
  }
  if (fs != st) {
    CC_HIP(hipEventRecord(d->front_done[slot], fs));
    CC_HIP(hipStreamWaitEvent(st, d->front_done[slot], 0));
  }
  {
    EvScope ev(d, EV_EVAL, st);
    EvalArgs A;
    A.integ = d->d_integ[slot].p;
    A.int_frame_elems = P->int_frame_elems;
    A.nchan = nchan;
    A.tilt_chan = tilt ? 2 : -1;
    A.sd = P->d_sd.p;
    A.W0 = d->m.win_w;
    A.H0 = d->m.win_h;
    A.nstages = (int)d->m.stage_ntrees.size();
    A.stage_first = d->d_stage_first.p;
    A.stage_ntrees = d->d_stage_ntrees.p;
    A.group_first = d->d_group_first.p;
    A.ngroups = d->n_groups;
    A.dense_from = d->dense_from;
    A.wave_below = d->wave_below;
    A.stop_after = d->stop_after;
    A.split_stumps = d->split_stumps;
    A.early_skip = d->early_skip;
    A.sq_compact = sq_compact;
    A.stage_thr = d->d_stage_thr.p;
    A.masks = d->d_masks[slot].p;
    A.mask_frame_words = P->mask_frame_words;
    A.cands = d->d_cands[slot].p;
    A.cand_count = d->d_counts[slot].p;
    A.cand_cap = d->cand_cap;
    A.stamps = nullptr;
    // The launches of this pass: one ahead-of-time kernel over the plan's tiles, or the specialised kernel's module(s), each
    // over its own tile list (its tile height; the tiles of one step when there is a module per step). The lists were built
    // by ensure_spec_tiles before the pass (never inside a graph capture).
    struct EvalLaunch {
      hipFunction_t fn;
      size_t lds;
      int n_tiles;
      const int4* tiles;
    };
    EvalLaunch launches[2];
    int n_launches = 0;
    const bool run_spec = d->spec_fn && d->m.max_nodes_per_tree <= 1;
    if (run_spec) {
      const int want[2][2] = {{d->spec_tile_y, d->spec_only_step}, {d->spec_fn1 ? d->spec_tile_y1 : 0, 1}};
      for (int i = 0; i < 2; i++) {
        if (want[i][0] == 0) continue;
        EvalLaunch L{i == 0 ? d->spec_fn : d->spec_fn1, i == 0 ? d->lds_spec : d->lds_spec1, P->n_tiles, P->d_tiles.p};
        if (!(want[i][0] == TILE_Y && want[i][1] == 0)) {
          const auto it = P->other_tiles.find(tile_list_key(want[i][0], want[i][1]));
          if (it == P->other_tiles.end() || !it->second)
            return set_error(CC_ERR_HIP, "internal: no tile list for tiles of %d window rows (step %d)", want[i][0], want[i][1]);
          L.n_tiles = it->second->n;
          L.tiles = it->second->d.p;
        }
        launches[n_launches++] = L;
      }
    } else
      launches[n_launches++] = EvalLaunch{nullptr, d->lds, P->n_tiles, P->d_tiles.p};
    size_t stamp_tiles = 0;
    for (int i = 0; i < n_launches; i++) stamp_tiles += (size_t)launches[i].n_tiles;
    if (std::getenv("CCAMD_DEBUG_STAMPS")) {  // timing experiments: per-block phase stamps of the cascade kernel (launch after launch)
      CC_HIP(d->d_stamps.ensure(std::max<size_t>(stamp_tiles * (size_t)nf * STAMP_SLOTS, 1)));
      CC_HIP(hipMemsetAsync(d->d_stamps.p, 0, stamp_tiles * (size_t)nf * STAMP_SLOTS * sizeof(unsigned long long), st));
      A.stamps = d->d_stamps.p;
    }
    A.dbg_codes = debug ? d->d_dbg_codes.p : nullptr;
    A.dbg_sums = debug ? d->d_dbg_sums.p : nullptr;
    A.stumps1 = haar ? (const void*)d->d_haar1.p : (const void*)d->d_lbp1.p;
    A.stumps2 = haar ? (const void*)d->d_haar2.p : (const void*)d->d_lbp2.p;
    A.wstumps1 = d->d_haar1w.p ? (const void*)d->d_haar1w.p : A.stumps1;
    A.wstumps2 = d->d_haar2w.p ? (const void*)d->d_haar2w.p : A.stumps2;
    A.gstumps = haar ? (const void*)d->d_haar_g.p : (const void*)d->d_lbp_g.p;
    A.lbp16_all = d->lbp16_all;
    if (!haar && d->d_lbp16.p) A.wstumps2 = d->d_lbp16.p;  // LBP wave phase of kernels with 16-bit tiles
    if (d->spec_fn && d->spec_tmode == TILE_PAIR16) {      // pair tile: its own record format for the table-driven stages
      A.gstumps = d->d_haar_p16.p;
      A.wstumps2 = d->d_haar_p16w.p;
    }
    A.trees = d->m.max_nodes_per_tree > 1 ? 1 : 0;
    A.nodes1 = haar ? (const void*)d->d_hnode1.p : (const void*)d->d_lnode1.p;
    A.nodes2 = haar ? (const void*)d->d_hnode2.p : (const void*)d->d_lnode2.p;
    A.tree_root = d->d_tree_root.p;
    A.tree_leaf0 = d->d_tree_leaf0.p;
    A.leaves = d->d_leaves.p;
    for (int i = 0; i < n_launches; i++) {
      const EvalLaunch& L = launches[i];
      if (L.n_tiles == 0) continue;
      A.tiles = L.tiles;
      if (L.fn) {
        void* params[] = {&A};
        // (the pass's EV_EVAL pair spans all cascade-kernel launches; the STEP-1 module's launch is also timed on its own)
        EvScope ev1(d, i == 1 ? EV_EVAL_STEP1 : -1, st);
        CC_HIP(hipModuleLaunchKernel(L.fn, (unsigned)L.n_tiles, (unsigned)nf, 1, EVAL_THREADS, 1, 1, (unsigned)L.lds, st, params, nullptr));
      } else if (haar)
        hipLaunchKernelGGL(k_eval_haar, dim3(L.n_tiles, nf), dim3(EVAL_THREADS), L.lds, st, A);
      else
        hipLaunchKernelGGL(k_eval_lbp, dim3(L.n_tiles, nf), dim3(EVAL_THREADS), L.lds, st, A);
      if (A.stamps) A.stamps += (size_t)L.n_tiles * (size_t)nf * STAMP_SLOTS;
    }
    d->last_stamp_tiles = (int)stamp_tiles;
  }
  if (fs != st) {
    CC_HIP(hipEventRecord(d->eval_done[slot], st));
    d->eval_pending[slot] = true;
  }
  {
    EvScope ev(d, EV_FILTER, st);
    hipLaunchKernelGGL(k_filter_candidates, dim3(64), dim3(256), 0, st, d->d_cands[slot].p, d->d_counts[slot].p, d->cand_cap, P->d_sd.p,
                       d->d_masks[slot].p, P->mask_frame_words, d->d_out[slot].p, d->d_counts[slot].p + 1);
    if (debug && P->n_grid_rows)
      hipLaunchKernelGGL(k_debug_visited, dim3(P->n_grid_rows), dim3(256), 0, st, P->d_sd.p, ns, P->d_gridrow_first.p,
                         d->d_masks[slot].p, d->d_dbg_visited.p);
  }
  CC_HIP(hipGetLastError());
  if (const char* path = std::getenv("CCAMD_DEBUG_STAMPS")) {  // dump [n_tiles * nf][STAMP_SLOTS] u64 (overwritten per pass)
    CC_HIP(hipStreamSynchronize(st));
    const int n_tiles_run = d->last_stamp_tiles;
    std::vector<unsigned long long> h((size_t)n_tiles_run * (size_t)nf * STAMP_SLOTS);
    CC_HIP(hipMemcpyAsync(h.data(), d->d_stamps.p, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, d->stream));
    CC_HIP(hipStreamSynchronize(d->stream));
    if (FILE* f = std::fopen(path, "wb")) {
      const int hdr[4] = {n_tiles_run, nf, STAMP_SLOTS, 0};
      std::fwrite(hdr, sizeof(int), 4, f);
      std::fwrite(h.data(), sizeof(unsigned long long), h.size(), f);
      std::fclose(f);
    }
  }
  d->tm.frames += nf;
  d->tm.grid_windows += P->windows * nf;
  d->tm.integral_elems += P->integral_elems * nf;
  return CC_OK;
}

static cc_status check_frame_args(const cc_detector* d, const uint8_t* frames, int n_frames, int width, int height,
                                  size_t row_stride, const cc_detect_params* p, const char* who) {
  if (!d || !p || (!frames && n_frames > 0)) return set_error(CC_ERR_INVALID_ARG, "%s: null argument", who);
  if (n_frames < 0 || width < 1 || height < 1 || row_stride < (size_t)width)
    return set_error(CC_ERR_INVALID_ARG, "%s: bad frame geometry (%dx%d, stride %zu, n %d)", who, width, height, row_stride, n_frames);
  if (width > 32768 || height > 32768) return set_error(CC_ERR_UNSUPPORTED, "%s: frames larger than 32768 px per side", who);
  if (!(p->scale_factor > 1.0)) return set_error(CC_ERR_INVALID_ARG, "%s: scaleFactor must be > 1", who);
  return CC_OK;
}

// The specialised kernel may use another tile height than the ahead-of-time kernels: its tile list is built on first use,
// before the pass is launched and never from inside a hipGraph capture. The copy goes through the detector's own stream and
// is waited for there: a copy on the legacy stream (plain hipMemcpy) is refused while ANY thread of the process captures a
// graph, and fails that thread's capture with it (two detectors on two host threads, one of them capturing its single-image pass).
static cc_status ensure_spec_tiles(cc_detector* d, Plan* P) {
  if (!d->spec_fn || d->m.max_nodes_per_tree > 1) return CC_OK;
  const int want[2][2] = {{d->spec_tile_y, d->spec_only_step}, {d->spec_fn1 ? d->spec_tile_y1 : 0, 1}};
  for (const auto& w : want) {
    if (w[0] == 0 || (w[0] == TILE_Y && w[1] == 0)) continue;  // no such module / the plan's own list serves it
    std::unique_ptr<Plan::TileList>& tl = P->other_tiles[tile_list_key(w[0], w[1])];
    if (tl) continue;
    std::unique_ptr<Plan::TileList> fresh(new Plan::TileList);
    const std::vector<int4> tv = plan_tile_list(P->geom, w[0], w[1]);
    fresh->n = (int)tv.size();
    CC_HIP(fresh->d.ensure(std::max<size_t>(tv.size(), 1)));
    if (!tv.empty()) {
      CC_HIP(hipMemcpyAsync(fresh->d.p, tv.data(), tv.size() * sizeof(int4), hipMemcpyHostToDevice, d->stream));
      CC_HIP(hipStreamSynchronize(d->stream));  // tv ends here
    }
    tl = std::move(fresh);
  }
  return CC_OK;
}

static cc_status run_device_pass(cc_detector* d, Plan* P, const uint8_t* dframes, int nf, size_t row_stride,
                                 size_t frame_stride, bool debug, int slot, bool single_stream);

// Fetches the results of the pending pass and hands them to its sink. On candidate-list overflow the lists grow and the
// pass is redone synchronously. Growing frees the lists of BOTH slots; a pass is judged against the capacity it was
// launched with and against the generation of the lists it wrote into (stale generation => redone as well).
static cc_status retire_pending(cc_detector* d) {
  if (!d->pending.active) return CC_OK;
  cc_detector::PendingPass ps = d->pending;
  d->pending.active = false;
  d->pending.sink.reset();
  std::vector<CandOut> got;
  for (;;) {
    CC_HIP(hipEventSynchronize(d->pass_done[ps.slot]));
    const int raw = d->h_counts[2 * ps.slot], kept = d->h_counts[2 * ps.slot + 1];
    if (raw > ps.cap || ps.gen != d->list_gen) {
      CC_HIP(hipStreamSynchronize(d->stream));
      if (raw > d->cand_cap) {
        d->cand_cap = raw + raw / 2;
        d->d_cands[0].release();
        d->d_cands[1].release();
        d->d_out[0].release();
        d->d_out[1].release();
        d->list_gen++;
      }
      cc_status st2 = run_device_pass(d, ps.plan, ps.dptr, ps.nf, ps.rs, ps.fs, ps.debug, ps.slot, false);
      if (st2 != CC_OK) return st2;
      ps.cap = d->cand_cap;
      ps.gen = d->list_gen;
      CC_HIP(hipMemcpyAsync(d->h_counts + 2 * ps.slot, d->d_counts[ps.slot].p, 2 * sizeof(int), hipMemcpyDeviceToHost, d->stream));
      CC_HIP(hipEventRecord(d->pass_done[ps.slot], d->stream));
      continue;
    }
    got.resize((size_t)kept);
    if (kept > 0) {  // the copy stream is free to run while the main stream executes the next pass
      CC_HIP(hipMemcpyAsync(got.data(), d->d_out[ps.slot].p, (size_t)kept * sizeof(CandOut), hipMemcpyDeviceToHost, d->copy_stream));
      CC_HIP(hipStreamSynchronize(d->copy_stream));
      for (CandOut& c : got) c.frame += ps.f0;
    }
    if (ps.sink && ps.sink->async_consume) {
      std::shared_ptr<BatchSink> sk = ps.sink;
      const int jf0 = ps.f0, jnf = ps.nf;
      auto cands = std::make_shared<std::vector<CandOut>>(std::move(got));
      sk->jobs.push_back(std::async(std::launch::async, [sk, jf0, jnf, cands]() { sk->consume(jf0, jnf, *cands); }));
    } else if (ps.sink && ps.sink->consume)
      ps.sink->consume(ps.f0, ps.nf, got);
    return CC_OK;
  }
}
// Retires a pass that belongs to another batch than the caller's: its failure is that batch's, reported when it is collected.
static void retire_foreign(cc_detector* d) {
  if (!d->pending.active) return;
  std::shared_ptr<BatchSink> sink = d->pending.sink;
  const cc_status st = retire_pending(d);
  if (st != CC_OK && sink && sink->status == CC_OK) {
    sink->status = st;
    sink->error = cc_last_error();
  }
}

// Host frames -> the device staging area of `slot`, on the front stream. Pageable memory (what a caller of the reference's
// shape hands over, tools/detection/Cpp/main.cpp:27-45) cannot be copied asynchronously: the runtime stages it through
// its own pinned chunks on the calling thread, copy after copy, and the call returns when the last one is on the device
// (round 3: a step of 64 Full-HD frames took 19.9 ms this way against 17.2 with resident frames). So the detector keeps
// its own pinned staging area, kStageSlots slots like the device one: the frames of a pass are copied into it by a few host
// threads (tight rows), then ONE asynchronous copy on the front stream moves the pass to the device while the cascade
// kernels of the pass before run -- and the caller's frames are free again when the call returns. A pinned slot is
// reused kStageSlots passes later; run_batch has retired its pass by then (it stages pass i + 1 only after pass i - 2 is
// retired), so its copy is long complete. Frames that already live in pinned memory (hipHostMalloc / hipHostRegister)
// skip the staging copy: those must stay valid until the call that retires their pass (include/cascadeclassifier_amd.h).
static cc_status stage_host_frames(cc_detector* d, const uint8_t* src, int nf, int width, int height, size_t row_stride,
                                   size_t frame_stride, uint8_t* dev, size_t rs, size_t fs, int slot, hipStream_t front) {
  hipPointerAttribute_t attr;
  bool pinned = false;
  if (hipPointerGetAttributes(&attr, src) == hipSuccess)
    pinned = attr.type == hipMemoryTypeHost;
  else
    (void)hipGetLastError();  // ordinary pageable memory is "invalid value" to this query on some runtimes
  static const bool no_stage = std::getenv("CCAMD_NO_PINNED_STAGING") != nullptr;  // the round-3 path, for A/B runs
  if (pinned || no_stage) {
    for (int f = 0; f < nf; f++)
      CC_HIP(hipMemcpy2DAsync(dev + (size_t)f * fs, rs, src + (size_t)f * frame_stride, row_stride, (size_t)width, (size_t)height,
                              hipMemcpyHostToDevice, front));
    return CC_OK;
  }
  const size_t need = fs * (size_t)d->pass_capacity * kStageSlots;
  if (d->h_stage_bytes < need) {
    if (d->h_stage) {
      CC_HIP(hipStreamSynchronize(front));  // no copy may still be reading the old area
      (void)hipHostFree(d->h_stage);
    }
    d->h_stage = nullptr;
    d->h_stage_bytes = 0;
    CC_HIP(hipHostMalloc(reinterpret_cast<void**>(&d->h_stage), need, hipHostMallocDefault));
    d->h_stage_bytes = need;
  }
  uint8_t* hs = d->h_stage + (size_t)slot * fs * (size_t)d->pass_capacity;
  auto copy_frames = [&](int fa, int fb) {
    for (int f = fa; f < fb; f++) {
      const uint8_t* sf = src + (size_t)f * frame_stride;
      uint8_t* df = hs + (size_t)f * fs;
      if (row_stride == rs)
        std::memcpy(df, sf, (size_t)(height - 1) * rs + (size_t)width);
      else
        for (int y = 0; y < height; y++) std::memcpy(df + (size_t)y * rs, sf + (size_t)y * row_stride, (size_t)width);
    }
  };
  static const int want_threads = []() {
    if (const char* e = std::getenv("CCAMD_STAGE_THREADS")) return std::max(1, std::atoi(e));
    const unsigned hc = std::thread::hardware_concurrency();
    return (int)std::max(1u, std::min(8u, hc / 2));  // 64 Full-HD frames per step: 2 threads 18.2, 4 17.9, 8 17.85 ms (resident frames 17.44)
  }();
  const size_t total = fs * (size_t)nf;
  const int nt = (int)std::min<size_t>((size_t)std::min(want_threads, nf), std::max<size_t>(1, total >> 21));  // >= 2 MB per thread
  if (nt <= 1) {
    copy_frames(0, nf);
  } else {
    std::vector<std::future<void>> jobs;
    try {
      for (int t = 1; t < nt; t++) {
        const int fa = (int)((long long)nf * t / nt), fb = (int)((long long)nf * (t + 1) / nt);
        jobs.push_back(std::async(std::launch::async, copy_frames, fa, fb));
      }
      copy_frames(0, nf / nt);
      for (auto& j : jobs) j.get();
    } catch (const std::exception& e) {
      for (auto& j : jobs)
        if (j.valid()) j.wait();
      return set_error(CC_ERR_HIP, "staging host frames: %s", e.what());
    }
  }
  CC_HIP(hipMemcpyAsync(dev, hs, total, hipMemcpyHostToDevice, front));
  return CC_OK;
}

static void spec_poll(cc_detector* d);  // installs a finished background specialisation

// Runs the batch in passes. `consume` (optional) receives the filtered candidates of each pass (frame indices made
// global) on the calling thread. With two or more frames the batch is cut into at least two passes and the host side
// of pass i (copy-back + consume) overlaps the device side of pass i+1.
// `defer_last`: the batch's last pass stays pending when the call returns (cc_detect_batch_submit); its candidates reach
// `consume` when the next call -- or cc_detect_batch_collect -- retires it.
template <class Consume>
static cc_status run_batch(cc_detector* d, const uint8_t* frames, int on_device, int n_frames, int width, int height,
                           size_t row_stride, size_t frame_stride, const cc_detect_params* p, bool want_results, bool debug,
                           Consume consume_fn, bool defer_last = false, std::shared_ptr<BatchSink> shared_sink = nullptr) {
  cc_status stt = ensure_device(d->device);
  if (stt != CC_OK) return stt;
  std::shared_ptr<BatchSink> sink = shared_sink;
  if (!sink) {
    sink = std::make_shared<BatchSink>();
    sink->consume = consume_fn;
  }
  auto consume = [&](int f0_, int nf_, std::vector<CandOut>& c) { sink->consume(f0_, nf_, c); };
  // A pass of an earlier (submitted) batch may still be pending. It can stay so -- and overlap this call's first pass --
  // only if this call runs ordinary passes on the same plan; everything else fetches it first.
  if (d->pending.active) {
    bool same_plan = false;
    for (auto& pl : d->plans)
      if (pl.get() == d->pending.plan && pl->w == width && pl->h == height && same_params(pl->p, *p)) same_plan = true;
    const bool single_image_graph = n_frames == 1 && !on_device && want_results && !debug && !d->profiling && d->use_graph;
    if (!same_plan || debug || !want_results || single_image_graph || n_frames < 1) retire_foreign(d);  // (a new plan may evict the pending pass's)
  }
  spec_poll(d);
  Plan* P = nullptr;
  stt = build_plan(d, width, height, *p, &P);
  if (stt != CC_OK) return stt;
  stt = ensure_spec_tiles(d, P);
  if (stt != CC_OK) return stt;
  if (!d->copy_stream) {
    CC_HIP(hipStreamCreateWithFlags(&d->copy_stream, hipStreamNonBlocking));
    CC_HIP(hipEventCreateWithFlags(&d->pass_done[0], hipEventDisableTiming));
    CC_HIP(hipEventCreateWithFlags(&d->pass_done[1], hipEventDisableTiming));
    CC_HIP(hipHostMalloc(reinterpret_cast<void**>(&d->h_counts), 4 * sizeof(int), hipHostMallocDefault));
    {  // the pyramid / integral stream only fills what the cascade kernel leaves idle: lowest priority
      int least = 0, greatest = 0;
      (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
      if (std::getenv("CCAMD_FRONT_SAME_PRIORITY")) least = greatest;
      CC_HIP(hipStreamCreateWithPriority(&d->front_stream, hipStreamNonBlocking, least));
    }
    for (hipEvent_t* e : {&d->front_done[0], &d->front_done[1], &d->eval_done[0], &d->eval_done[1], &d->batch_begin})
      CC_HIP(hipEventCreateWithFlags(e, hipEventDisableTiming));
    d->overlap_front = std::getenv("CCAMD_NO_FRONT_OVERLAP") ? 0 : 1;
    d->use_graph = std::getenv("CCAMD_NO_GRAPH") ? 0 : 1;

  }
  hipStream_t front = d->overlap_front ? d->front_stream : d->stream;
  // Frames produced by earlier work on a stream the CALLER gave us (cc_detector_set_stream) must be complete before the
  // pyramid reads them. On the detector's own stream there is only our own earlier work -- a pending pass of the batch
  // submitted before -- and waiting for that would serialise exactly what cc_detect_batch_submit exists to overlap.
  if (front != d->stream && d->stream != d->own_stream) {
    CC_HIP(hipEventRecord(d->batch_begin, d->stream));
    CC_HIP(hipStreamWaitEvent(front, d->batch_begin, 0));
  }
  if (n_frames == 1 && !on_device && want_results && !debug && !d->profiling && d->use_graph) {
    // ---- single host image: one pass on one stream, replayed from a hipGraph once the buffers are sized ----
    const size_t rs = (size_t)align_up(width, 4), fs = rs * (size_t)height;
    if (d->h_frame_bytes < fs) {
      if (d->h_frame) (void)hipHostFree(d->h_frame);
      d->h_frame = nullptr;
      d->h_frame_bytes = 0;
      CC_HIP(hipHostMalloc(reinterpret_cast<void**>(&d->h_frame), fs, hipHostMallocDefault));
      d->h_frame_bytes = fs;
    }
    for (int y = 0; y < height; y++) std::memcpy(d->h_frame + (size_t)y * rs, frames + (size_t)y * row_stride, (size_t)width);
    CC_HIP(d->d_frames.ensure(fs * (size_t)d->pass_capacity * 2));
    auto body = [&]() -> cc_status {
      CC_HIP(hipMemcpyAsync(d->d_frames.p, d->h_frame, fs, hipMemcpyHostToDevice, d->stream));
      cc_status s2 = run_device_pass(d, P, d->d_frames.p, 1, rs, fs, false, 0, true);
      if (s2 != CC_OK) return s2;
      CC_HIP(hipMemcpyAsync(d->h_counts, d->d_counts[0].p, 2 * sizeof(int), hipMemcpyDeviceToHost, d->stream));
      return CC_OK;
    };
    auto key_now = [&]() {
      return std::vector<const void*>{d->d_frames.p, d->h_frame, d->d_pyr.p, d->d_integ[0].p, d->d_hbuf.p, d->d_diag.p, d->d_tseg.p, d->d_masks[0].p, d->d_cands[0].p,
                                      d->d_out[0].p, d->d_counts[0].p, d->h_counts, (const void*)d->spec_fn, (const void*)d->spec_fn1, (const void*)d->stream,
                                      (const void*)(size_t)d->cand_cap, (const void*)(size_t)d->wave_below, (const void*)(size_t)(d->stop_after + 16)};
    };
    if (d->eval_pending[0] || d->eval_pending[1]) {  // a batch call may still be using the buffers on the other stream
      CC_HIP(hipStreamSynchronize(d->front_stream));
      d->eval_pending[0] = d->eval_pending[1] = false;
    }
    bool launched = false;
    d->last_call_graph = 0;
    if (P->graph_exec && P->graph_key == key_now()) {
      CC_HIP(hipGraphLaunch(P->graph_exec, d->stream));
      launched = true;
      d->last_call_graph = 1;
    } else if (P->graph_warm) {
      if (P->graph_exec) (void)hipGraphExecDestroy(P->graph_exec);
      P->graph_exec = nullptr;
      hipGraph_t graph = nullptr;
      // Relaxed mode: this thread's stream-ordered calls are captured, nobody's legacy-stream calls are policed (other
      // host threads may be inside synchronous copies of their own handles; under the stricter modes HIP fails those
      // calls and invalidates this capture). One capture at a time per process; if a capture does not come out, the
      // detector simply stops using graphs.
      static std::mutex capture_mu;
      hipError_t ce = hipSuccess, ie = hipSuccess;
      cc_status s2 = CC_OK;
      {
        std::lock_guard<std::mutex> capture_lock(capture_mu);
        ce = std::getenv("CCAMD_DEBUG_FAIL_CAPTURE") ? hipErrorStreamCaptureUnsupported  // tests: the fallback path
                                                     : hipStreamBeginCapture(d->stream, hipStreamCaptureModeRelaxed);
        if (ce == hipSuccess) {
          s2 = body();
          ce = hipStreamEndCapture(d->stream, &graph);
        }
      }
      if (ce == hipSuccess && s2 == CC_OK && graph) ie = hipGraphInstantiate(&P->graph_exec, graph, nullptr, nullptr, 0);
      if (graph) (void)hipGraphDestroy(graph);
      if (ce != hipSuccess || s2 != CC_OK || ie != hipSuccess || !P->graph_exec) {
        // said once per detector, on stderr: the call still succeeds, only the launch-bound single-image path gets slower
        std::fprintf(stderr, "[ccamd] hipGraph capture of the single-image pass failed (begin/end: %s, body status %d, instantiate: %s): "
                             "this detector uses ordinary launches from now on\n",
                     hipGetErrorString(ce), (int)s2, hipGetErrorString(ie));
        (void)hipGetLastError();
        P->graph_exec = nullptr;
        d->use_graph = 0;  // ordinary launches from now on (this call included)
      } else {
        P->graph_key = key_now();
        CC_HIP(hipGraphLaunch(P->graph_exec, d->stream));
        launched = true;
        d->last_call_graph = 1;
      }
    }
    if (!launched) {
      stt = body();
      if (stt != CC_OK) return stt;
      P->graph_warm = true;  // every buffer now has its size: the next call can be captured
    }
    CC_HIP(hipStreamSynchronize(d->stream));
    const int raw = d->h_counts[0], kept = d->h_counts[1];
    if (raw <= d->cand_cap) {
      std::vector<CandOut> got((size_t)kept);
      if (kept > 0) {
        CC_HIP(hipMemcpyAsync(got.data(), d->d_out[0].p, (size_t)kept * sizeof(CandOut), hipMemcpyDeviceToHost, d->stream));
        CC_HIP(hipStreamSynchronize(d->stream));
      }
      consume(0, 1, got);
      return CC_OK;
    }
    // candidate list overflow: the ordinary path below grows the lists and redoes the pass
  }
  // Pass sizes. With results wanted the batch is cut into a few passes so that the host side of pass i (copy-back +
  // grouping) overlaps the device side of pass i+1 and the pyramid/integrals of pass i+1 overlap the cascade kernel of
  // pass i. Nothing overlaps the LAST pass's host work, so it carries about half the frames of the others (32 frames ->
  // 9, 9, 9, 5; a small first pass, to start the cascade kernel earlier, measured no better). A SUBMITTED batch has its last
  // pass overlapped by the batch after it, so it is cut into two even passes only (bench step 17.51 ms against 17.86 / 18.25
  // for 1 / 4 passes, profiles/r03_kernel_experiments.txt).
  std::vector<int> sizes;
  if (want_results && n_frames >= 2) {
    const bool submitted = defer_last && !d->pipeline_passes_set;
    const int passes = std::min(submitted ? 2 : d->pipeline_passes, n_frames);
    const char* explicit_sizes = std::getenv("CCAMD_PASS_SIZES");  // tuning: comma-separated sizes
    if (explicit_sizes && *explicit_sizes) {
      for (const char* q = explicit_sizes; *q;) {
        const int v = std::atoi(q);
        if (v > 0) sizes.push_back(v);
        while (*q && *q != ',') q++;
        if (*q == ',') q++;
      }
    } else {
      int per = (n_frames + passes - 1) / passes;
      if (passes >= 3 && !d->even_passes && !submitted) {
        const int big = (2 * n_frames + 2 * passes - 2) / (2 * passes - 1);
        if (big >= 2 && big * (passes - 1) < n_frames) per = big;
      }
      for (int f = 0; f < n_frames; f += per) sizes.push_back(std::min(per, n_frames - f));
    }
  } else {
    for (int f = 0; f < n_frames; f += d->max_batch) sizes.push_back(std::min(d->max_batch, n_frames - f));
  }
  {  // normalise: sizes within max_batch, summing to n_frames
    std::vector<int> fixed;
    int left = n_frames;
    for (size_t i = 0; left > 0; i++) {
      int v = i < sizes.size() ? sizes[i] : left;
      v = std::max(1, std::min({v, d->max_batch, left}));
      fixed.push_back(v);
      left -= v;
    }
    sizes.swap(fixed);
  }
  for (int v : sizes) d->pass_capacity = std::max(d->pass_capacity, v);  // the workspace only ever grows
  // Host frames: THREE staging slots (device and pinned), handed out round-robin across passes and calls. The frames of
  // pass i + 1 are staged and their copy issued right after pass i is launched and BEFORE the pass before it is fetched
  // (a blocking wait), so the host copy and the H2D transfer of a pass run a whole pass ahead of the kernels that read
  // them. Two slots are not enough for that: the slot of pass i + 1 would be the one of pass i - 1, which is still
  // unfetched at that point and re-reads its frames if it has to be redone (candidate-list overflow). With three, the slot
  // that is overwritten belongs to pass i - 2, fetched when pass i - 1 was launched.
  const uint8_t* prestaged = nullptr;
  auto stage_pass = [&](int pf0, int pnf, const uint8_t** where) -> cc_status {
    const size_t rs = (size_t)align_up(width, 4), fs = rs * (size_t)height;
    const size_t need = fs * (size_t)d->pass_capacity * kStageSlots;
    if (d->d_frames.n < need && d->pending.active) {
      // Growing the staging area frees it, and the unfetched pass still names its frames there (round-3 advisor finding).
      if (d->pending.sink == sink) {
        const cc_status st2 = retire_pending(d);
        if (st2 != CC_OK) return st2;
      } else
        retire_foreign(d);
    }
    if (d->d_frames.n < need) d->stage_slot = 0;
    CC_HIP(d->d_frames.ensure(need));
    const int sslot = d->stage_slot;
    d->stage_slot = (d->stage_slot + 1) % kStageSlots;
    uint8_t* stage = d->d_frames.p + (size_t)sslot * fs * (size_t)d->pass_capacity;
    const cc_status st2 = stage_host_frames(d, frames + (size_t)pf0 * frame_stride, pnf, width, height, row_stride, frame_stride, stage, rs, fs,
                                            sslot, front);
    if (st2 != CC_OK) return st2;
    *where = stage;
    return CC_OK;
  };
  // spec_poll may have installed another kernel: a pending pass keeps the results it was launched for, nothing to redo.
  int f0 = 0;
  for (size_t pi = 0; pi < sizes.size(); f0 += sizes[pi], pi++) {
    const int slot = d->next_slot;
    cc_detector::PendingPass ps;
    ps.plan = P;
    ps.f0 = f0;
    ps.nf = sizes[pi];
    ps.slot = slot;
    ps.rs = row_stride;
    ps.fs = frame_stride;
    ps.debug = debug;
    ps.sink = sink;
    if (on_device) {
      ps.dptr = frames + (size_t)f0 * frame_stride;
    } else {
      ps.rs = (size_t)align_up(width, 4);
      ps.fs = ps.rs * (size_t)height;
      if (!prestaged) {
        stt = stage_pass(f0, sizes[pi], &prestaged);
        if (stt != CC_OK) return stt;
      }
      ps.dptr = prestaged;
      prestaged = nullptr;
    }
    // the slot's result buffers are free: the pass that used them last was retired when the pass after it was launched
    static const bool trace_host = std::getenv("CCAMD_TRACE_HOST") != nullptr;  // host-side timeline of the pass loop (stderr)
    const auto th0 = std::chrono::steady_clock::now();
    auto th = [&](const char* what) {
      if (trace_host)
        std::fprintf(stderr, "[ccamd host] pass %zu %-18s +%.3f ms\n", pi, what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - th0).count());
    };
    stt = run_device_pass(d, P, ps.dptr, ps.nf, ps.rs, ps.fs, debug, slot);
    if (stt != CC_OK) return stt;
    th("launched");
    ps.cap = d->cand_cap;
    ps.gen = d->list_gen;
    if (!on_device && pi + 1 < sizes.size() && want_results) {  // the next pass's frames travel while this pass runs
      stt = stage_pass(f0 + sizes[pi], sizes[pi + 1], &prestaged);
      if (stt != CC_OK) return stt;
      th("next pass staged");
    }
    if (want_results) {
      CC_HIP(hipMemcpyAsync(d->h_counts + 2 * slot, d->d_counts[slot].p, 2 * sizeof(int), hipMemcpyDeviceToHost, d->stream));
      CC_HIP(hipEventRecord(d->pass_done[slot], d->stream));
      // fetch the pass launched before this one -- of this batch or, for the first pass, of the batch submitted before --
      // while the device runs this one
      if (d->pending.active) {
        if (d->pending.sink == sink) {
          stt = retire_pending(d);
          if (stt != CC_OK) return stt;
        } else
          retire_foreign(d);
      }
      th("previous retired");
      ps.active = true;
      d->pending = ps;
    }
    d->next_slot ^= 1;
  }
  if (d->pending.active && !(defer_last && d->pending.sink == sink)) {
    if (d->pending.sink == sink) {
      stt = retire_pending(d);
      if (stt != CC_OK) return stt;
    } else
      retire_foreign(d);
  }
  // Profiling: the event pairs of this call are read once their kernels are done. A submitted batch must not wait for that
  // here (the synchronisation would undo the overlap with the next batch -- which is how the round-3 bench first measured
  // its own instrumentation instead of the pipeline): its events are read by cc_detector_get_timings or by the next
  // synchronous call.
  if (d->profiling && !defer_last) {
    CC_HIP(hipStreamSynchronize(d->stream));
    if (d->front_stream) CC_HIP(hipStreamSynchronize(d->front_stream));
    collect_events(d);
  }
  return CC_OK;
}

static void sort_candidates(std::vector<CandOut>& v) {
  std::sort(v.begin(), v.end(), [](const CandOut& a, const CandOut& b) {
    if (a.frame != b.frame) return a.frame < b.frame;
    if (a.scale != b.scale) return a.scale < b.scale;
    if (a.gy != b.gy) return a.gy < b.gy;
    return a.gx < b.gx;
  });
}


// Host side of one pass (called while the device already runs the next pass). Per frame: order the candidates (scale, y, x)
// = OpenCV's single-threaded order, then group. Frames are independent, so they are spread over a few host threads.
static void group_pass(int min_neighbors, int f0, int nf, std::vector<CandOut>& cands, std::vector<std::vector<cc_rect>>& grouped) {
  std::vector<std::vector<CandOut>> per_frame((size_t)nf);
  for (const CandOut& c : cands) per_frame[(size_t)(c.frame - f0)].push_back(c);
  auto work = [&](int a0, int a1) {
    for (int f = a0; f < a1; f++) {
      sort_candidates(per_frame[(size_t)f]);
      std::vector<cc_rect>& rects = grouped[(size_t)(f0 + f)];
      rects.reserve(per_frame[(size_t)f].size());
      for (const CandOut& c : per_frame[(size_t)f]) rects.push_back(cc_rect{c.x, c.y, c.w, c.h});
      group_rectangles(rects, min_neighbors, 0.2);  // GROUP_EPS
    }
  };
  const int nthr = std::max(1, std::min({nf, (int)std::thread::hardware_concurrency(), 16}));
  if (nthr <= 1 || cands.size() < 2048)
    work(0, nf);
  else {
    std::vector<std::thread> th;
    for (int t = 0; t < nthr; t++) th.emplace_back(work, (int)((long long)nf * t / nthr), (int)((long long)nf * (t + 1) / nthr));
    for (auto& t : th) t.join();
  }
}

// ------------------------------------------------------------------------------------------------
// Run-time specialisation of the cascade kernel (hiprtc, loaded on demand: the library does not link against it).
// ------------------------------------------------------------------------------------------------
struct HipRtcApi {
  void* lib = nullptr;
  int (*create)(void**, const char*, const char*, int, const char* const*, const char* const*) = nullptr;
  int (*compile)(void*, int, const char* const*) = nullptr;
  int (*log_size)(void*, size_t*) = nullptr;
  int (*log)(void*, char*) = nullptr;
  int (*code_size)(void*, size_t*) = nullptr;
  int (*code)(void*, char*) = nullptr;
  int (*destroy)(void**) = nullptr;
  int (*version)(int*, int*) = nullptr;  // optional
  bool ok() const { return create && compile && log_size && log && code_size && code && destroy; }
};

static const HipRtcApi& hiprtc_api() {
  static HipRtcApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char* name : {"libhiprtc.so.7", "libhiprtc.so", "/opt/rocm/lib/libhiprtc.so"}) {
      api.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (api.lib) break;
    }
    if (!api.lib) return;
    auto sym = [&](const char* n) { return dlsym(api.lib, n); };
    api.create = reinterpret_cast<decltype(api.create)>(sym("hiprtcCreateProgram"));
    api.compile = reinterpret_cast<decltype(api.compile)>(sym("hiprtcCompileProgram"));
    api.log_size = reinterpret_cast<decltype(api.log_size)>(sym("hiprtcGetProgramLogSize"));
    api.log = reinterpret_cast<decltype(api.log)>(sym("hiprtcGetProgramLog"));
    api.code_size = reinterpret_cast<decltype(api.code_size)>(sym("hiprtcGetCodeSize"));
    api.code = reinterpret_cast<decltype(api.code)>(sym("hiprtcGetCode"));
    api.destroy = reinterpret_cast<decltype(api.destroy)>(sym("hiprtcDestroyProgram"));
    api.version = reinterpret_cast<decltype(api.version)>(sym("hiprtcVersion"));
  });
  return api;
}

// hiprtc has no <cstdint>: the fixed-width names the kernel source uses
static const char kSpecPrelude[] =
    "typedef signed char int8_t;\ntypedef unsigned char uint8_t;\ntypedef short int16_t;\ntypedef unsigned short uint16_t;\n"
    "typedef int int32_t;\ntypedef unsigned int uint32_t;\ntypedef long long int64_t;\ntypedef unsigned long long uint64_t;\n";

// Compiles `src` for `arch`; identical (source, options) pairs are served from a per-process cache.
// The modules a run-time specialised kernel is compiled as: which tiles each covers and its tile height (-DCC_TILE_Y).
// * LBP kernels whose STEP-2 tiles hold 16-bit entries: ONE module, 20 window rows per tile (64 x 20 windows per block of 256
//   threads). A tile's halo rows are staged per 20 instead of per 8 window rows, the per-block work (barrier rounds,
//   counters, the wave phase's window collection) is paid once per 1 280 windows, and the late stages find 2.5 x the windows per
//   block to fill their wavefronts with; at 26 KB per block six blocks fit a CU (6 wavefronts per SIMD: 80 VGPRs).
//   Stock LBP cascade, ms per 32 Full-HD frames alone (tools/r4_g.sh): 8 rows 4.90, 12 rows 4.25, 16 rows 3.96, 20 rows 3.78,
//   24 rows 3.85, 32 rows 4.12 (each at its best register budget).
// * Haar kernels with 32-bit tiles: TWO modules, one per step -- 12 rows for the STEP-1 tiles, 8 for the STEP-2 tiles. A
//   STEP-1 tile is a third of a STEP-2 tile (12.7 KB against 24 KB), but one launch requests the larger of the two for every
//   block; in a launch of their own the STEP-1 tiles run at 6 blocks per CU. ms per 32 Full-HD frames alone, one run
//   (tools/r4_h.sh): one module at 8 rows 8.35; two modules at 8 / 8 rows 7.82, 12 / 8 rows 7.32, 12 / 12 rows 7.49,
//   16 / 12 rows 7.77, 12 / 16 rows 8.11 (STEP-1 / STEP-2; a 32-bit STEP-2 tile of 12 rows leaves 4 blocks per CU).
// * everything else (pair tile, Haar with 16-bit tiles): one module at the library's 8 rows.
// CCAMD_SPEC_TILE_Y sets every module's rows, CCAMD_SPEC_TILE_Y1 / _Y2 the STEP-1 / STEP-2 module's, CCAMD_SPEC_ONE_MODULE=1
// forces a single module (tuning).
struct SpecModulePlan {
  int only_step, tile_y;
};
static std::vector<SpecModulePlan> spec_modules(const Cascade& m, int tmode) {
  auto valid = [&](int ty) { return ty >= EVAL_WAVES && ty <= 32 && ty % EVAL_WAVES == 0; };
  auto env_rows = [&](const char* name, int dflt) {
    const char* e = std::getenv(name);
    const int v = e ? std::atoi(e) : dflt;
    return valid(v) ? v : dflt;
  };
  // (cascades with tilted features keep the library's tile height: their records and generated offsets carry the distance
  // between the sum tile and the tilted tile behind it, which depends on the tile's rows)
  if (tmode == TILE_PAIR16 || m.has_tilted) return {{0, TILE_Y}};
  const bool lbp16 = m.feature_type == CC_FEATURE_LBP && tmode == TILE_16;
  const bool haar32 = tmode == TILE_32 && m.feature_type == CC_FEATURE_HAAR;
  const bool rows_given = std::getenv("CCAMD_SPEC_TILE_Y") != nullptr;
  const int all = env_rows("CCAMD_SPEC_TILE_Y", lbp16 ? 20 : TILE_Y);
  const bool two = (haar32 || std::getenv("CCAMD_SPEC_TWO_MODULES")) && !std::getenv("CCAMD_SPEC_ONE_MODULE");
  if (!two) return {{0, all}};
  return {{2, env_rows("CCAMD_SPEC_TILE_Y2", all)}, {1, env_rows("CCAMD_SPEC_TILE_Y1", (haar32 && !rows_given) ? 12 : all)}};
}

static cc_status compile_specialised(const std::string& src, const std::string& arch, int n_stages, bool lbp, int tmode, int tile_y, int only_step, int win_w, int win_h, std::vector<char>& code) {
  const bool tile16 = tmode == TILE_16;
  static std::mutex mu;
  static std::map<std::string, std::vector<char>> cache;
  const std::string o_arch = "--offload-arch=" + arch, o_k = "-DCC_SPEC_STAGES=" + std::to_string(n_stages);
  const std::string o_ty = "-DCC_TILE_Y=" + std::to_string(tile_y), o_th = "-DCC_EVAL_THREADS=" + std::to_string(EVAL_THREADS);
  // same code generation rules as the ahead-of-time build (Makefile): no FMA contraction, no fast-math
  // register budget = the occupancy the LDS footprint allows: 5 blocks per CU with the 32-bit tile, 7-8 with the 16-bit one
  // (16-bit tiles of >= 20 rows: 26 KB per block = 6 blocks per CU)
  std::string o_w = "-DCC_EVAL_MIN_WAVES_PER_EU=" + std::to_string(tile16 ? (tile_y >= 20 ? 6 : 7) : CC_EVAL_MIN_WAVES_PER_EU);
  const std::string o_step = "-DCC_ONLY_STEP=" + std::to_string(only_step);
  if (const char* e = std::getenv("CCAMD_SPEC_WAVES_PER_EU")) o_w = "-DCC_EVAL_MIN_WAVES_PER_EU=" + std::to_string(std::max(1, std::min(8, std::atoi(e))));  // tuning
  const std::string o_w0 = "-DCC_SPEC_W0=" + std::to_string(win_w), o_h0 = "-DCC_SPEC_H0=" + std::to_string(win_h);  // tile geometry folds to constants
  std::vector<const char*> optv = {o_arch.c_str(), "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", o_k.c_str(), o_ty.c_str(), o_th.c_str(), o_w.c_str(), o_w0.c_str(), o_h0.c_str()};
  if (only_step) optv.push_back(o_step.c_str());
  if (lbp) optv.push_back("-DCC_SPEC_LBP");
  if (tile16) optv.push_back("-DCC_SPEC_TILE16");
  if (tmode == TILE_PAIR16) optv.push_back("-DCC_SPEC_PAIR16");
  std::vector<std::string> extra;  // tuning: further compiler options, space-separated
  if (const char* e = std::getenv("CCAMD_SPEC_EXTRA_FLAGS")) {
    std::istringstream is(e);
    for (std::string w; is >> w;) extra.push_back(w);
  }
  for (const std::string& w : extra) optv.push_back(w.c_str());
  const char* const* opts = optv.data();
  const int n_opts = (int)optv.size();
  std::string key;  // everything the code object depends on: compiler version, options, then the source
  {
    int major = 0, minor = 0;
    const HipRtcApi& rtc = hiprtc_api();
    if (rtc.version) (void)rtc.version(&major, &minor);
    key += "hiprtc " + std::to_string(major) + "." + std::to_string(minor) + " ";
  }
  for (int i = 0; i < n_opts; i++) key += std::string(opts[i]) + " ";
  key += "#" + src;
  {
    std::lock_guard<std::mutex> lk(mu);
    auto it = cache.find(key);
    if (it != cache.end()) {
      code = it->second;
      return CC_OK;
    }
  }
  // second level: code objects on disk ($CCAMD_CACHE_DIR, else ~/.cache/cascadeclassifier_amd; CCAMD_CACHE_DIR= disables).
  // The file is named by a 64-bit FNV-1a hash of the key and starts with a header that repeats the key's length and two
  // independent 64-bit hashes of it; the key covers architecture, options, the hiprtc version and the generated source.
  // A file whose header does not match (another toolchain, a collision, a foreign or truncated file) is ignored and
  // rewritten: a wrong code object would carry another cascade's thresholds and give wrong detections silently.
  struct CacheHeader {
    char magic[8];
    unsigned long long key_len, h1, h2;
  };
  auto hash_key = [&](unsigned long long seed, unsigned long long prime) {
    unsigned long long h = seed;
    for (unsigned char ch : key) h = (h ^ ch) * prime;
    return h ^ (h >> 29);
  };
  CacheHeader want;
  std::memcpy(want.magic, "CCAMDSP2", 8);
  want.key_len = key.size();
  want.h1 = hash_key(1469598103934665603ull, 1099511628211ull);
  want.h2 = hash_key(0x9E3779B97F4A7C15ull, 0x100000001B3ull * 31ull + 2ull);
  std::string cache_file;
  {
    const char* dir = std::getenv("CCAMD_CACHE_DIR");
    std::string base;
    if (dir)
      base = dir;
    else if (const char* home = std::getenv("HOME"))
      base = std::string(home) + "/.cache/cascadeclassifier_amd";
    if (!base.empty()) {
      char name[64];
      snprintf(name, sizeof(name), "/spec_%016llx_%zu.hsaco", want.h1, key.size());
      cache_file = base + name;
      if (FILE* f = std::fopen(cache_file.c_str(), "rb")) {
        std::fseek(f, 0, SEEK_END);
        const long n = std::ftell(f);
        std::fseek(f, 0, SEEK_SET);
        CacheHeader got;
        std::vector<char> buf;
        bool ok = n > (long)sizeof(CacheHeader) + 64 && std::fread(&got, sizeof(got), 1, f) == 1 && std::memcmp(&got, &want, sizeof(want)) == 0;
        if (ok) {
          buf.resize((size_t)n - sizeof(CacheHeader));
          ok = std::fread(buf.data(), 1, buf.size(), f) == buf.size() && std::memcmp(buf.data(), "\x7f" "ELF", 4) == 0;
        }
        std::fclose(f);
        if (ok) {
          code = buf;
          std::lock_guard<std::mutex> lk(mu);
          cache[key] = code;
          return CC_OK;
        }
      }
      (void)::mkdir(base.c_str(), 0755);  // one level; a missing parent just means no disk cache
    }
  }
  if (const char* dump = std::getenv("CCAMD_DUMP_SPEC_SOURCE")) {  // for inspection with hipcc -S
    if (FILE* f = std::fopen(dump, "w")) {
      std::fwrite(src.data(), 1, src.size(), f);
      std::fclose(f);
    }
  }
  // One compilation at a time per process: builds are rare and seconds long, the compiler stack underneath hiprtc is not
  // worth trusting with concurrent invocations, and a second thread asking for the same code waits here and then finds it.
  static std::mutex compile_mu;
  std::lock_guard<std::mutex> compile_lock(compile_mu);
  {
    std::lock_guard<std::mutex> lk(mu);
    auto it = cache.find(key);
    if (it != cache.end()) {
      code = it->second;
      return CC_OK;
    }
  }
  const HipRtcApi& rtc = hiprtc_api();
  if (!rtc.ok()) return set_error(CC_ERR_UNSUPPORTED, "cc_detector_specialize: libhiprtc is not available (%s)", rtc.lib ? "missing symbols" : "dlopen failed");
  void* prog = nullptr;
  if (rtc.create(&prog, src.c_str(), "cc_eval_kernel_spec.hip", 0, nullptr, nullptr) != 0)
    return set_error(CC_ERR_HIP, "cc_detector_specialize: hiprtcCreateProgram failed");
  const int rc = rtc.compile(prog, n_opts, const_cast<const char**>(opts));
  if (rc != 0) {
    size_t n = 0;
    rtc.log_size(prog, &n);
    std::string log(n + 1, '\0');
    if (n) rtc.log(prog, &log[0]);
    rtc.destroy(&prog);
    return set_error(CC_ERR_HIP, "cc_detector_specialize: hiprtc compilation failed (%d): %.1500s", rc, log.c_str());
  }
  size_t n = 0;
  rtc.code_size(prog, &n);
  code.resize(n);
  rtc.code(prog, code.data());
  rtc.destroy(&prog);
  if (!cache_file.empty()) {  // write to a private name, then rename: readers never see a partial file
    const std::string tmp = cache_file + "." + std::to_string((long long)::getpid()) + ".tmp";
    if (FILE* f = std::fopen(tmp.c_str(), "wb")) {
      const bool ok = std::fwrite(&want, sizeof(want), 1, f) == 1 && std::fwrite(code.data(), 1, code.size(), f) == code.size();
      std::fclose(f);
      if (!ok || std::rename(tmp.c_str(), cache_file.c_str()) != 0) (void)std::remove(tmp.c_str());
    }
  }
  std::lock_guard<std::mutex> lk(mu);
  cache[key] = code;
  return CC_OK;
}


// Host half of the specialisation: source for the first stages (whole stages within the code-size budget) compiled for
// `arch`. No device calls: safe on a background thread.
static cc_status spec_build(const Cascade& m, int n_stages, const std::string& arch, std::vector<SpecCode>& codes, int& k_out, int& tmode_out) {
  if (m.max_nodes_per_tree > 1) return set_error(CC_ERR_UNSUPPORTED, "cc_detector_specialize: stump cascades only");
  int k = 0, stumps = 0;
  int budget = 320;  // instruction cache: more stages measured no faster, 12 stages slower
  if (const char* e = std::getenv("CCAMD_SPEC_BUDGET")) budget = std::max(1, std::atoi(e));  // tuning
  while (k < (int)m.stage_ntrees.size() && k < n_stages && k < MAX_STAGES && (k == 0 || stumps + m.stage_ntrees[(size_t)k] <= budget))
    stumps += m.stage_ntrees[(size_t)k++];
  std::string src = kSpecPrelude;
  src += "namespace ccamd {\n";
  src += kEvalKernelSrc;
  src += "\n}  // namespace ccamd\n";
  const std::string marker = "//@@CC_SPEC_FUNCTIONS@@";
  const size_t pos = src.find(marker);
  if (pos == std::string::npos) return set_error(CC_ERR_HIP, "cc_detector_specialize: kernel source has no specialisation marker");
  const int tmode = pair16_eligible(m) ? TILE_PAIR16 : tile16_eligible(m, k) ? TILE_16 : TILE_32;
  src.replace(pos, marker.size(), spec_stage_source(m, k, tmode));
  k_out = k;
  tmode_out = tmode;
  codes.clear();
  for (const SpecModulePlan& mp : spec_modules(m, tmode)) {
    SpecCode c;
    c.only_step = mp.only_step;
    c.tile_y = mp.tile_y;
    const cc_status st = compile_specialised(src, arch, k, m.feature_type == CC_FEATURE_LBP, tmode, mp.tile_y, mp.only_step, m.win_w, m.win_h, c.code);
    if (st != CC_OK) return st;
    codes.push_back(std::move(c));
  }
  return CC_OK;
}

// Device half: load the code object and make it the detector's cascade kernel. Owning thread only.
static cc_status spec_install(cc_detector* d, const std::vector<SpecCode>& codes, int k, int tmode) {
  if (codes.empty() || codes.size() > 2) return set_error(CC_ERR_HIP, "cc_detector_specialize: %zu modules", codes.size());
  retire_foreign(d);  // a pass still unfetched was launched with the old kernel (and its tile lists): fetch it before the switch
  const bool haar_k = d->m.feature_type == CC_FEATURE_HAAR;
  struct Loaded {
    hipModule_t mod = nullptr;
    hipFunction_t fn = nullptr;
    size_t lds = 0;
    int tile_y = TILE_Y, only_step = 0;
  };
  std::vector<Loaded> L;
  auto unload_all = [&]() {
    for (Loaded& x : L)
      if (x.mod) (void)hipModuleUnload(x.mod);
  };
  for (const SpecCode& c : codes) {
    Loaded x;
    x.tile_y = c.tile_y;
    x.only_step = c.only_step;
    const char* entry = c.only_step == 1 ? "k_eval_spec_step1" : c.only_step == 2 ? "k_eval_spec_step2" : "k_eval_spec";
    if (hipModuleLoadData(&x.mod, c.code.data()) != hipSuccess || hipModuleGetFunction(&x.fn, x.mod, entry) != hipSuccess) {
      (void)hipGetLastError();
      if (x.mod) (void)hipModuleUnload(x.mod);
      unload_all();
      return set_error(CC_ERR_HIP, "cc_detector_specialize: the compiled module does not load or has no entry point");
    }
    const int ty = c.tile_y;
    // tiles this module stages: of the step(s) it covers (CCAMD_DEBUG_ONLY_STEP narrows a single module's request for timing runs)
    const int step = c.only_step ? c.only_step : debug_only_step();
    x.lds = d->lds;
    if (tmode == TILE_32) {
      const TileGeom<1> G1(d->m.win_w, d->m.win_h, ty);
      const TileGeom<2> G2(d->m.win_w, d->m.win_h, ty);
      const int words = step == 1 ? G1.words() : step == 2 ? G2.words() : std::max(G1.words(), G2.words());
      x.lds = eval_lds_bytes(words, d->m.has_tilted, haar_k, ty) + d->lds_extra;
    } else if (tmode == TILE_16) {  // STEP-1 tile in 32 bits, STEP-2 tile in 16 bits
      const TileGeom<1> G1(d->m.win_w, d->m.win_h, ty);
      const TileGeom16 G2(d->m.win_w, d->m.win_h, ty);
      x.lds = eval_lds_bytes(std::max(G1.words(), G2.words()), false, haar_k, ty) + d->lds_extra;
    } else if (tmode == TILE_PAIR16) {  // STEP-2 tile of window pairs, with partial sums for both windows of a slot
      const TileGeom<1> G1(d->m.win_w, d->m.win_h);
      const TileGeomP G2(d->m.win_w, d->m.win_h);
      x.lds = std::max(eval_lds_bytes(G1.words(), false), eval_lds_bytes_pair(G2.words())) + d->lds_extra;
      if (!d->d_haar_p16.p) {
        std::vector<HaarStumpP16> tp, tw;
        build_haar_stumps_p16(d->m, tp);
        // wave phase: the stage's stumps in the bank-aware order computed for this geometry's offsets (the records carry
        // their stump's index in `pad`)
        std::vector<HaarStumpDev> geo;
        build_haar_stumps_at(d->m, geo, [&](int y, int x2) { return G2.at(y, x2); }, 0);
        const std::vector<HaarStumpDev> order = d->wave_below > 0 && !std::getenv("CCAMD_NO_WAVE_SCHEDULE") ? schedule_for_wave_phase(d->m, geo) : geo;
        tw.reserve(order.size());
        for (const HaarStumpDev& r : order) tw.push_back(tp[(size_t)r.pad]);
        CC_HIP(d->d_haar_p16.upload(tp, d->stream));
        CC_HIP(d->d_haar_p16w.upload(tw, d->stream));
        CC_HIP(hipStreamSynchronize(d->stream));
      }
    }
    if (x.lds > 64 * 1024) {  // same opt-in as the ahead-of-time kernels (cc_detector_create)
      const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(x.fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)x.lds);
      if (e != hipSuccess) {
        (void)hipGetLastError();
        (void)hipModuleUnload(x.mod);
        unload_all();
        return set_error(CC_ERR_UNSUPPORTED, "cc_detector_specialize: %zu bytes of LDS per tile cannot be requested for a run-time module (%s)",
                         x.lds, hipGetErrorString(e));
      }
    }
    if (std::getenv("CCAMD_TRACE_HOST")) {  // what the specialised kernel's footprint allows per CU
      int nb = 0;
      if (hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&nb, x.fn, EVAL_THREADS, x.lds) != hipSuccess) (void)hipGetLastError();
      std::fprintf(stderr, "[ccamd host] specialised kernel: tile mode %d, tiles of step %d (0 = all), %d window rows, %zu bytes of LDS per block, %d resident blocks per CU\n",
                   tmode, c.only_step, ty, x.lds, nb);
    }
    L.push_back(x);
  }
  CC_HIP(hipStreamSynchronize(d->stream));
  if (d->spec_mod) (void)hipModuleUnload(d->spec_mod);
  if (d->spec_mod1) (void)hipModuleUnload(d->spec_mod1);
  d->spec_mod1 = nullptr;
  d->spec_fn1 = nullptr;
  // the module for all tiles or for the STEP-2 tiles is the primary one; a STEP-1 module, if any, the second
  const Loaded* prim = &L[0];
  const Loaded* sec = L.size() > 1 ? &L[1] : nullptr;
  if (sec && prim->only_step == 1) std::swap(prim, sec);
  d->spec_mod = prim->mod;
  d->spec_fn = prim->fn;
  d->lds_spec = prim->lds;
  d->spec_tile_y = prim->tile_y;
  d->spec_only_step = prim->only_step;
  if (sec) {
    d->spec_mod1 = sec->mod;
    d->spec_fn1 = sec->fn;
    d->lds_spec1 = sec->lds;
    d->spec_tile_y1 = sec->tile_y;
  }
  d->spec_stages = k;
  d->spec_tmode = tmode;
  return CC_OK;
}

static cc_status device_arch(int device, std::string& arch) {
  hipDeviceProp_t prop;
  CC_HIP(hipGetDeviceProperties(&prop, device));
  arch = prop.gcnArchName;
  arch = arch.substr(0, arch.find(':'));
  return CC_OK;
}

// Picks up a finished background build (called at the start of every detection call).
static void spec_poll(cc_detector* d) {
  const int st = d->spec_bg_state.load(std::memory_order_acquire);
  if (st != 2 && st != 3) return;
  if (d->spec_thread.joinable()) d->spec_thread.join();
  if (st == 2 && spec_install(d, d->spec_bg_code, d->spec_bg_stages, d->spec_bg_tmode) != CC_OK) d->spec_bg_error = cc_last_error();
  d->spec_bg_code.clear();
  d->spec_bg_state.store(0, std::memory_order_release);
}

static cc_status spec_start_background(cc_detector* d, int n_stages) {
  if (d->m.max_nodes_per_tree > 1) return set_error(CC_ERR_UNSUPPORTED, "cc_detector_specialize: stump cascades only");
  if (d->spec_bg_state.load(std::memory_order_acquire) == 1) return CC_OK;  // a build is already running
  spec_poll(d);
  std::string arch;
  cc_status st = device_arch(d->device, arch);
  if (st != CC_OK) return st;
  d->spec_bg_error.clear();
  d->spec_bg_state.store(1, std::memory_order_release);
  d->spec_thread = std::thread([d, n_stages, arch]() {
    std::vector<SpecCode> code;
    int k = 0;
    int tmode = 0;
    const cc_status s2 = spec_build(d->m, n_stages, arch, code, k, tmode);
    if (s2 == CC_OK) {
      d->spec_bg_code.swap(code);
      d->spec_bg_stages = k;
      d->spec_bg_tmode = tmode;
      d->spec_bg_state.store(2, std::memory_order_release);
    } else {
      d->spec_bg_error = cc_last_error();  // this thread's message
      d->spec_bg_state.store(3, std::memory_order_release);
    }
  });
  return CC_OK;
}

}  // namespace ccamd

// Parity instrumentation of inv_sqrt_as_float (cc_eval_kernel.inc): counts values nf for which it differs from
// (float)(1.0 / sqrt(nf)). nf is drawn as the kernels form it -- area * valsqsum - valsum^2 for a random window size, pixel
// sum and squared sum (an integer-valued double) -- and, every fourth draw, as an arbitrary integer below 2^52.
namespace ccamd {
__global__ void k_vnf_check(unsigned long long seed, int per_thread, unsigned long long* mismatches) {
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
    double nf;
    if ((i & 3) == 3) {
      nf = (double)(1ull + (next() >> 12));
    } else {
      const unsigned w = 1u + (unsigned)(r & 0xffu), h = 1u + (unsigned)((r >> 8) & 0xffu);
      const double area = (double)(w * h);
      const unsigned long long n = (unsigned long long)w * h;
      const unsigned long long sm = (r >> 16) % (255ull * n + 1ull);
      // any squared sum a window with that pixel sum can have: between sm^2 / n and 255 * sm
      const unsigned long long lo_sq = (sm * sm + n - 1) / n, hi_sq = 255ull * sm;
      const unsigned long long sq = lo_sq + (hi_sq > lo_sq ? next() % (hi_sq - lo_sq + 1ull) : 0ull);
      nf = area * (double)(unsigned)sq - (double)(int)sm * (double)(int)sm;  // the kernels' expression (valsqsum wraps at 2^32 like theirs)
    }
    if (!(nf > 0.)) continue;
    if (__float_as_uint(inv_sqrt_as_float(nf)) != __float_as_uint((float)(1. / sqrt(nf)))) bad++;
  }
  if (bad) atomicAdd(mismatches, bad);
}
}  // namespace ccamd

extern "C" {

int cc_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

// Serial numbers of the detectors alive in this process. op 0: new serial (registered); 1: unregister `serial`;
// 2: is `serial` alive (returns 1 / 0).
static unsigned long long detector_registry(int op, unsigned long long serial) {
  static std::mutex mu;
  static std::unordered_set<unsigned long long> live;
  static unsigned long long next = 0;
  std::lock_guard<std::mutex> lk(mu);
  if (op == 0) {
    live.insert(++next);
    return next;
  }
  if (op == 1) {
    live.erase(serial);
    return 0;
  }
  return live.count(serial) ? 1 : 0;
}

cc_status cc_detector_create(const cc_cascade* c, int device, int max_batch, cc_detector** out) {
  if (!c || !out) return set_error(CC_ERR_INVALID_ARG, "cc_detector_create: null argument");
  *out = nullptr;
  if (max_batch < 1 || max_batch > 4096) return set_error(CC_ERR_INVALID_ARG, "cc_detector_create: max_batch %d out of range", max_batch);

  cc_status st = ensure_device(device);
  if (st != CC_OK) return st;
  std::unique_ptr<cc_detector> d(new cc_detector());
  d->m = c->m;
  d->serial = detector_registry(0, 0);
  d->device = device;
  d->max_batch = max_batch;
  CC_HIP(hipStreamCreateWithFlags(&d->own_stream, hipStreamNonBlocking));
  d->stream = d->own_stream;
  const TileGeom<1> G1(d->m.win_w, d->m.win_h);
  const TileGeom<2> G2(d->m.win_w, d->m.win_h);
  d->lds = eval_lds_bytes(std::max(G1.words(), G2.words()), d->m.has_tilted, d->m.feature_type == CC_FEATURE_HAAR);
  if ((int)d->m.stage_ntrees.size() >= MAX_STAGES)
    return set_error(CC_ERR_UNSUPPORTED, "cc_detector_create: cascades with %zu stages are not supported (limit %d)",
                     d->m.stage_ntrees.size(), MAX_STAGES - 1);
  if (d->lds > 160 * 1024 - 256)
    return set_error(CC_ERR_UNSUPPORTED, "cc_detector_create: %dx%d window needs %zu bytes of LDS per tile (limit 160 KiB)",
                     d->m.win_w, d->m.win_h, d->lds);
  const bool haar = d->m.feature_type == CC_FEATURE_HAAR;
  if (d->lds > 64 * 1024) {
    if (haar)
      CC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_eval_haar), hipFuncAttributeMaxDynamicSharedMemorySize, (int)d->lds));
    else
      CC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_eval_lbp), hipFuncAttributeMaxDynamicSharedMemorySize, (int)d->lds));
  }
  std::vector<int> ntrees(d->m.stage_ntrees.begin(), d->m.stage_ntrees.end());
  std::vector<int> sfirst(d->m.stage_first.begin(), d->m.stage_first.end());
  CC_HIP(d->d_stage_ntrees.upload(ntrees, d->stream));
  CC_HIP(d->d_stage_first.upload(sfirst, d->stream));
  // One wavefront per window (parallel reduction of a stage's votes) is only bit-identical to the sequential CPU sum
  // when every partial sum is exact in double; LBP stages are too short for it to pay.
  const bool trees = d->m.max_nodes_per_tree > 1;  // general trees: thread-per-window phases only, sequential sums
  const bool exact = !trees && stage_sums_order_independent(d->m);
  d->wave_below = (haar && exact) ? 24 : 0;
  if (!haar && !trees) {
    // LBP wave phase (lanes = the stumps of several whole stages; sums in stump order, so no exactness condition): needs
    // every stage to fit a wavefront. Threshold measured on the stock cascade (tools/sweeps/r3_l.txt).
    bool fits = true;
    for (int v : d->m.stage_ntrees) fits = fits && v <= 64;
    d->wave_below = fits ? 24 : 0;  // 8 ... 32 within 3 % of each other, 48 +5 %, 64 +11 %, off +23 %
  }
  d->split_stumps = exact ? 1 : 0;
  if (const char* e = std::getenv("CCAMD_SPLIT_STUMPS")) d->split_stumps = d->split_stumps && std::atoi(e) != 0;
  if (const char* e = std::getenv("CCAMD_DEBUG_STOP_AFTER_STAGE")) d->stop_after = std::atoi(e);  // timing experiments
  d->early_skip = std::getenv("CCAMD_NO_EARLY_SKIP") ? 0 : 1;
  d->full_sqsum = std::getenv("CCAMD_FULL_SQSUM") ? 1 : 0;
  if (const char* e = std::getenv("CCAMD_PIPELINE_PASSES")) {
    d->pipeline_passes = std::max(1, std::atoi(e));
    d->pipeline_passes_set = 1;
  }
  if (const char* e = std::getenv("CCAMD_CAND_CAP")) d->cand_cap = std::max(16, std::atoi(e));  // initial candidate-list capacity (tests: forces the overflow path)
  d->even_passes = std::getenv("CCAMD_EVEN_PASSES") ? 1 : 0;
  if (const char* e = std::getenv("CCAMD_DEBUG_EXTRA_LDS")) {  // occupancy experiments: pad the per-block LDS request
    d->lds_extra = (size_t)std::max(0, std::atoi(e));
    d->lds += d->lds_extra;
    if (d->lds > 64 * 1024)
      CC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(haar ? &k_eval_haar : &k_eval_lbp), hipFuncAttributeMaxDynamicSharedMemorySize, (int)d->lds));
  }
  if (const char* e = std::getenv("CCAMD_WAVE_BELOW"))  // tuning knob; only honoured where the wave phase is valid at all
    if (d->wave_below) d->wave_below = std::max(0, std::min(64, std::atoi(e)));  // the wave phase holds one window per lane
  CC_HIP(d->d_stage_thr.upload(d->m.stage_threshold, d->stream));
  {
    // Stage groups (EvalArgs::group_first). Every stage boundary inside the cascade kernel costs the block a barrier, the
    // class counts and a queue rebuild -- about as much as a pass over 25 stumps -- while what it buys is that the windows
    // rejected by the stage stop occupying lanes. For short stages (the stock LBP cascade has 3-10 stumps per stage) the
    // boundary costs more than it saves, so consecutive stages are put into one group while the group stays within
    // `budget` stumps; a stage longer than the budget is a group of its own (the form every stage had before). Grouping
    // never changes a result: a window's exit stage and stage sum are recorded where it fails, whatever the lanes around
    // it do. Cascades with deeper trees keep one stage per group.
    // Measured (round 3, stock LBP cascade, ms per 32 Full-HD frames alone on the device; tools/sweeps/r3_*.txt): grouping
    // from stage 1 on is slower at every budget but 12 stumps (-3 %), because the lanes of windows that died inside a
    // group keep executing; the gain is in the LATE stages, where a handful of windows per tile pay a barrier round per
    // stage. Hence two knobs: groups start at stage `from`, and hold up to `budget` stumps.
    const bool lbp = !haar && !trees;
    // LBP with its wave phase (which takes over below 24 windows), round 3 (tiles of 8 rows, bank-class table): groups of
    // <= 20 stumps from stage 2 on 4.91-5.06, 12 from stage 1: 5.04-5.25, one stage per group 5.29, 30 from stage 2: 5.36.
    // Round 4 (list queue from stage 2; specialised kernel on tiles of 20 rows): <= 14 stumps 3.70, <= 20 3.87, <= 30 4.17;
    // at 8 rows 4.84 / 4.87 / 5.13.
    int budget = lbp ? 14 : 0, from = lbp ? 2 : 1;
    if (const char* e = std::getenv("CCAMD_GROUP_STUMPS")) budget = trees ? 0 : std::max(0, std::atoi(e));  // tuning
    if (const char* e = std::getenv("CCAMD_GROUP_FROM")) from = std::max(1, std::atoi(e));
    std::vector<int> gf;
    const int nst = (int)d->m.stage_ntrees.size();
    for (int s0 = 0; s0 < nst;) {
      gf.push_back(s0);
      const int cap = s0 < from ? 0 : budget;  // stage 0 is the dense phase: always alone
      int s1 = s0 + 1, sum = d->m.stage_ntrees[(size_t)s0];
      while (s1 < nst && sum + d->m.stage_ntrees[(size_t)s1] <= cap) sum += d->m.stage_ntrees[(size_t)s1++];
      s0 = s1;
    }
    gf.push_back(nst);
    if (gf.size() < 2) gf.push_back(nst);  // no stage at all: one empty group, the kernels read group_first[1]
    d->n_groups = (int)gf.size() - 1;
    // Queue form (EvalArgs::dense_from): LBP stump cascades switch from the bank-class table to a plain list at stage 2.
    // There the tile is down to ~100 of its 512 windows: the table needs 6.2 rows for them (its fullest class) where a list
    // needs 3.1, i.e. four wavefront passes over the group's 19 stumps instead of two, and the LBP kernel is bound by the
    // vector ALU (73 % busy; LDS 52 %, profiles/r04_pmc_eval_lbp.json), not by the ~2.5-way bank conflicts the list costs.
    // Haar cascades keep the table everywhere (LDS-bound; CCAMD_DENSE_FROM=<stage> to experiment).
    int dense_stage = lbp ? 2 : -1;
    if (const char* e = std::getenv("CCAMD_DENSE_FROM")) dense_stage = std::atoi(e);
    d->dense_from = 0x7fffffff;
    if (dense_stage >= 1)
      for (int g = (int)gf.size() - 2; g >= 1; g--)
        if (gf[(size_t)g] >= dense_stage) d->dense_from = g;
    CC_HIP(d->d_group_first.upload(gf, d->stream));
    CC_HIP(hipStreamSynchronize(d->stream));
  }
  if (trees) {
    std::vector<int> root(d->m.tree_first_node.begin(), d->m.tree_first_node.end()), leaf0(d->m.tree_first_leaf.begin(), d->m.tree_first_leaf.end());
    CC_HIP(d->d_tree_root.upload(root, d->stream));
    CC_HIP(d->d_tree_leaf0.upload(leaf0, d->stream));
    CC_HIP(d->d_leaves.upload(d->m.leaves, d->stream));
    if (haar) {
      std::vector<HaarNodeDev> n1, n2;
      build_haar_nodes<1>(d->m, n1);
      build_haar_nodes<2>(d->m, n2);
      CC_HIP(d->d_hnode1.upload(n1, d->stream));
      CC_HIP(d->d_hnode2.upload(n2, d->stream));
      CC_HIP(hipStreamSynchronize(d->stream));
    } else {
      std::vector<LbpNodeDev> n1, n2;
      build_lbp_nodes<1>(d->m, n1);
      build_lbp_nodes<2>(d->m, n2);
      CC_HIP(d->d_lnode1.upload(n1, d->stream));
      CC_HIP(d->d_lnode2.upload(n2, d->stream));
      CC_HIP(hipStreamSynchronize(d->stream));
    }
  } else if (haar) {
    std::vector<HaarStumpDev> s1, s2;
    build_haar_stumps<1>(d->m, s1);
    build_haar_stumps<2>(d->m, s2);
    CC_HIP(d->d_haar1.upload(s1, d->stream));
    CC_HIP(d->d_haar2.upload(s2, d->stream));
    std::vector<HaarStumpDev> sg;
    if (!d->m.has_tilted) build_haar_gstumps(d->m, sg);
    CC_HIP(d->d_haar_g.upload(sg, d->stream));
    CC_HIP(hipStreamSynchronize(d->stream));
    if (d->wave_below > 0 && !std::getenv("CCAMD_NO_WAVE_SCHEDULE")) {
      const std::vector<HaarStumpDev> w1 = schedule_for_wave_phase(d->m, s1), w2 = schedule_for_wave_phase(d->m, s2);
      CC_HIP(d->d_haar1w.upload(w1, d->stream));
      CC_HIP(d->d_haar2w.upload(w2, d->stream));
      CC_HIP(hipStreamSynchronize(d->stream));
    }
    CC_HIP(hipStreamSynchronize(d->stream));
  } else {
    std::vector<LbpStumpDev> s1, s2;
    build_lbp_stumps<1>(d->m, s1);
    build_lbp_stumps<2>(d->m, s2);
    CC_HIP(d->d_lbp1.upload(s1, d->stream));
    CC_HIP(d->d_lbp2.upload(s2, d->stream));
    std::vector<LbpStumpDev> sg, s16;
    build_lbp_gstumps(d->m, sg);
    CC_HIP(d->d_lbp_g.upload(sg, d->stream));
    d->lbp16_all = 1;
    for (size_t i = 0; i < d->m.stump_feature.size(); i++) {
      const int32_t* r = &d->m.lbp_rects[(size_t)d->m.stump_feature[i] * 4];
      if (!fits16((long long)r[2] * r[3])) d->lbp16_all = 0;
    }
    if (d->lbp16_all) {
      build_lbp_stumps16(d->m, s16);
      CC_HIP(d->d_lbp16.upload(s16, d->stream));
    }
    CC_HIP(hipStreamSynchronize(d->stream));
  }
  // CCAMD_AUTO_SPECIALIZE=<stages>: build the specialised kernel in the background; detection starts on the table-driven
  // kernel and switches over when the module is ready (no change to the calling code)
  if (const char* e = std::getenv("CCAMD_AUTO_SPECIALIZE")) {
    const int k = std::atoi(e);
    if (k > 0 && d->m.max_nodes_per_tree == 1) (void)spec_start_background(d.get(), k);
  }
  *out = d.release();
  return CC_OK;
}

void cc_detector_destroy(cc_detector* d) {
  if (!d) return;
  detector_registry(1, d->serial);
  (void)hipSetDevice(d->device);
  if (d->pending.active) {  // a submitted batch nobody collected: let the device finish, drop the results
    (void)hipStreamSynchronize(d->stream);
    d->pending.active = false;
    d->pending.sink.reset();
  }
  delete d;
}

cc_status cc_detector_set_stream(cc_detector* d, void* hip_stream) {
  if (!d) return set_error(CC_ERR_INVALID_ARG, "cc_detector_set_stream: null detector");
  d->stream = hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : d->own_stream;
  return CC_OK;
}

cc_status cc_detector_specialize(cc_detector* d, int n_stages) {
  if (!d) return set_error(CC_ERR_INVALID_ARG, "cc_detector_specialize: null detector");
  cc_status st = ensure_device(d->device);
  if (st != CC_OK) return st;
  if (d->spec_thread.joinable()) d->spec_thread.join();  // a background build, if any, is superseded
  d->spec_bg_state.store(0, std::memory_order_release);
  d->spec_bg_code.clear();
  if (n_stages <= 0) {  // back to the table-driven kernel
    CC_HIP(hipStreamSynchronize(d->stream));
    if (d->spec_mod) (void)hipModuleUnload(d->spec_mod);
    if (d->spec_mod1) (void)hipModuleUnload(d->spec_mod1);
    d->spec_mod = d->spec_mod1 = nullptr;
    d->spec_fn = d->spec_fn1 = nullptr;
    d->spec_stages = 0;
    return CC_OK;
  }
  std::string arch;
  st = device_arch(d->device, arch);
  if (st != CC_OK) return st;
  std::vector<SpecCode> code;
  int k = 0;
  int tmode = 0;
  st = spec_build(d->m, n_stages, arch, code, k, tmode);
  if (st != CC_OK) return st;
  return spec_install(d, code, k, tmode);
}

cc_status cc_detector_specialize_async(cc_detector* d, int n_stages) {
  if (!d) return set_error(CC_ERR_INVALID_ARG, "cc_detector_specialize_async: null detector");
  if (n_stages <= 0) return set_error(CC_ERR_INVALID_ARG, "cc_detector_specialize_async: n_stages must be positive");
  cc_status st = ensure_device(d->device);
  if (st != CC_OK) return st;
  return spec_start_background(d, n_stages);
}

int cc_detector_specialized_stages(const cc_detector* d) { return d ? d->spec_stages : 0; }

cc_status cc_cascade_compile_specialized(const cc_cascade* c, int n_stages, const char* arch, size_t* code_bytes) {
  if (!c || !arch || !code_bytes) return set_error(CC_ERR_INVALID_ARG, "cc_cascade_compile_specialized: null argument");
  std::vector<SpecCode> code;
  int k = 0;
  int tmode = 0;
  const cc_status st = spec_build(c->m, std::max(1, n_stages), arch, code, k, tmode);
  if (st != CC_OK) return st;
  *code_bytes = 0;
  for (const SpecCode& m : code) *code_bytes += m.code.size();  // all modules (Haar: one per step)
  return CC_OK;
}

int cc_detector_graph_active(const cc_detector* d) {
  if (!d) return (int)set_error(CC_ERR_INVALID_ARG, "cc_detector_graph_active: null detector");
  return d->last_call_graph;
}

cc_status cc_detector_set_profiling(cc_detector* d, int enabled) {
  if (!d) return set_error(CC_ERR_INVALID_ARG, "cc_detector_set_profiling: null detector");
  d->profiling = enabled != 0;
  return CC_OK;
}

cc_status cc_detector_get_timings(cc_detector* d, cc_detector_timings* t, int reset) {
  if (!d || !t) return set_error(CC_ERR_INVALID_ARG, "cc_detector_get_timings: null argument");
  if (!d->events.empty()) {  // event pairs of submitted batches: their kernels may still be running
    cc_status st = ensure_device(d->device);
    if (st != CC_OK) return st;
    CC_HIP(hipStreamSynchronize(d->stream));
    if (d->front_stream) CC_HIP(hipStreamSynchronize(d->front_stream));
    collect_events(d);
  }
  *t = d->tm;
  if (reset) std::memset(&d->tm, 0, sizeof(d->tm));
  return CC_OK;
}

cc_status cc_detect_batch_device_only(cc_detector* d, const uint8_t* frames, int on_device, int n_frames, int width, int height,
                                      size_t row_stride, size_t frame_stride, const cc_detect_params* p) {
  cc_status st = check_frame_args(d, frames, n_frames, width, height, row_stride, p, "cc_detect_batch_device_only");
  if (st != CC_OK) return st;
  return run_batch(d, frames, on_device, n_frames, width, height, row_stride, frame_stride, p, false, false,
                   [](int, int, std::vector<CandOut>&) {});
}

cc_status cc_detect_batch(cc_detector* d, const uint8_t* frames, int on_device, int n_frames, int width, int height,
                          size_t row_stride, size_t frame_stride, const cc_detect_params* p, cc_rect* out, int cap,
                          int32_t* offsets) {
  cc_status st = check_frame_args(d, frames, n_frames, width, height, row_stride, p, "cc_detect_batch");
  if (st != CC_OK) return st;
  if (!offsets || (cap > 0 && !out) || cap < 0) return set_error(CC_ERR_INVALID_ARG, "cc_detect_batch: bad output buffers");
  const auto t_start = std::chrono::steady_clock::now();
  double group_ms = 0;
  size_t n_cands = 0;
  std::vector<std::vector<cc_rect>> grouped((size_t)n_frames);
  const int min_neighbors = p->min_neighbors;
  auto consume = [&](int f0, int nf, std::vector<CandOut>& cands) {
    const auto t0 = std::chrono::steady_clock::now();
    n_cands += cands.size();
    group_pass(min_neighbors, f0, nf, cands, grouped);
    group_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  };
  st = run_batch(d, frames, on_device, n_frames, width, height, row_stride, frame_stride, p, true, false, consume);
  if (st != CC_OK) return st;
  long long total = 0;
  for (int f = 0; f < n_frames; f++) {
    offsets[f] = (int32_t)total;
    for (const cc_rect& r : grouped[(size_t)f]) {
      if (total < cap) out[total] = r;
      total++;
    }
  }
  offsets[n_frames] = (int32_t)total;
  if (std::getenv("CCAMD_TIMING")) {
    const auto t1 = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[ccamd] detect_batch: total %.3f ms, of which sort+group on the host %.3f ms (overlapped), %zu candidates\n",
                 std::chrono::duration<double, std::milli>(t1 - t_start).count(), group_ms, n_cands);
  }
  if (total > cap) return set_error(CC_ERR_BUFFER_TOO_SMALL, "cc_detect_batch: %lld rectangles, capacity %d", total, cap);
  return CC_OK;
}

struct cc_batch_ticket {
  cc_detector* owner = nullptr;
  unsigned long long owner_serial = 0;  // cc_detector::serial of the owner: an address can be reused by a later detector
  int n_frames = 0;
  std::shared_ptr<BatchSink> sink;
  // Shared with the sink's consume function, NOT owned by the ticket alone: the detector may still hold the sink of a pass
  // it has not fetched when the ticket ends on an error path, and that pass's helper then writes here (round-3 advisor
  // finding: the function used to capture the ticket's address).
  std::shared_ptr<std::vector<std::vector<cc_rect>>> grouped;
  ~cc_batch_ticket() {
    if (sink) {
      try {
        sink->wait_jobs();
      } catch (...) {
      }
    }
  }
};

// The helper threads of a submitted batch run std::sort / std::vector code: what they throw (bad_alloc) surfaces in
// future::get and must not cross the C ABI.
static cc_status wait_sink_jobs(BatchSink& sink, const char* who) {
  try {
    sink.wait_jobs();
  } catch (const std::exception& e) {
    sink.jobs.clear();
    return set_error(CC_ERR_HIP, "%s: grouping a pass failed on the host: %s", who, e.what());
  } catch (...) {
    sink.jobs.clear();
    return set_error(CC_ERR_HIP, "%s: grouping a pass failed on the host", who);
  }
  return CC_OK;
}

cc_status cc_detect_batch_submit(cc_detector* d, const uint8_t* frames, int on_device, int n_frames, int width, int height,
                                 size_t row_stride, size_t frame_stride, const cc_detect_params* p, cc_batch_ticket** ticket) {
  if (!ticket) return set_error(CC_ERR_INVALID_ARG, "cc_detect_batch_submit: null ticket pointer");
  *ticket = nullptr;
  cc_status st = check_frame_args(d, frames, n_frames, width, height, row_stride, p, "cc_detect_batch_submit");
  if (st != CC_OK) return st;
  std::unique_ptr<cc_batch_ticket> t(new cc_batch_ticket);
  t->owner = d;
  t->owner_serial = d->serial;
  t->n_frames = n_frames;
  t->grouped = std::make_shared<std::vector<std::vector<cc_rect>>>((size_t)n_frames);
  t->sink = std::make_shared<BatchSink>();
  std::shared_ptr<std::vector<std::vector<cc_rect>>> grouped = t->grouped;
  const int min_neighbors = p->min_neighbors;
  t->sink->consume = [grouped, min_neighbors](int f0, int nf, std::vector<CandOut>& cands) { group_pass(min_neighbors, f0, nf, cands, *grouped); };
  t->sink->async_consume = true;  // passes of a batch cover disjoint frames: their helpers never touch the same entry of `grouped`
  st = run_batch(d, frames, on_device, n_frames, width, height, row_stride, frame_stride, p, true, false,
                 [](int, int, std::vector<CandOut>&) {}, /*defer_last=*/true, t->sink);
  if (st != CC_OK) {
    if (d->pending.active && d->pending.sink == t->sink) {  // the batch failed: its unfetched pass delivers to nobody
      d->pending.active = false;
      d->pending.sink.reset();
    }
    return st;
  }
  *ticket = t.release();
  return CC_OK;
}

// A ticket is only ever ended by the detector it came from: with any other detector (or one that was destroyed and whose
// address a new detector took over -- `serial` tells them apart) the call fails and the ticket stays valid.
static bool ticket_is_of(const cc_detector* d, const cc_batch_ticket* t) { return d && t->owner == d && t->owner_serial == d->serial; }

cc_status cc_detect_batch_collect(cc_detector* d, cc_batch_ticket* t, cc_rect* out, int cap, int32_t* offsets) {
  if (!t) return set_error(CC_ERR_INVALID_ARG, "cc_detect_batch_collect: null ticket");
  if (!ticket_is_of(d, t)) return set_error(CC_ERR_INVALID_ARG, "cc_detect_batch_collect: the ticket belongs to another detector (it stays valid)");
  std::unique_ptr<cc_batch_ticket> own(t);  // from here on the ticket ends with this call (except CC_ERR_BUFFER_TOO_SMALL)
  if (!offsets || (cap > 0 && !out) || cap < 0) {
    if (d->pending.active && d->pending.sink == t->sink) (void)retire_pending(d);
    return set_error(CC_ERR_INVALID_ARG, "cc_detect_batch_collect: bad output buffers");
  }
  cc_status st = ensure_device(d->device);
  if (st != CC_OK) {
    (void)own.release();  // nothing was fetched: the caller may collect or discard again
    return st;
  }
  if (d->pending.active && d->pending.sink == t->sink) {  // its last pass has not been fetched by a later submit
    st = retire_pending(d);
    if (st != CC_OK) return st;
  }
  if (t->sink->status != CC_OK) return set_error(t->sink->status, "cc_detect_batch_collect: %s", t->sink->error.c_str());
  st = wait_sink_jobs(*t->sink, "cc_detect_batch_collect");  // the helper threads that sort + group what the passes delivered
  if (st != CC_OK) return st;
  const std::vector<std::vector<cc_rect>>& grouped = *t->grouped;
  long long total = 0;
  for (int f = 0; f < t->n_frames; f++) {
    offsets[f] = (int32_t)total;
    for (const cc_rect& r : grouped[(size_t)f]) {
      if (total < cap) out[total] = r;
      total++;
    }
  }
  offsets[t->n_frames] = (int32_t)total;
  if (total > cap) {
    (void)own.release();  // the results stay in the ticket: collect again with room for offsets[n_frames] rectangles
    return set_error(CC_ERR_BUFFER_TOO_SMALL, "cc_detect_batch_collect: %lld rectangles, capacity %d", total, cap);
  }
  return CC_OK;
}

cc_status cc_detect_batch_discard(cc_detector* d, cc_batch_ticket* t) {
  if (!t) return CC_OK;
  if (!detector_registry(2, t->owner_serial)) {  // the owner was destroyed (it waited for the device and dropped the pass)
    delete t;
    return CC_OK;
  }
  if (!ticket_is_of(d, t)) return set_error(CC_ERR_INVALID_ARG, "cc_detect_batch_discard: the ticket belongs to another detector (it stays valid)");
  std::unique_ptr<cc_batch_ticket> own(t);
  if (d->pending.active && d->pending.sink == t->sink) {  // let its last pass finish and drop what it delivers
    cc_status st = ensure_device(d->device);
    if (st != CC_OK) {  // the pass cannot be fetched: cut it loose (its helper, if any, keeps `grouped` alive by itself)
      d->pending.active = false;
      d->pending.sink.reset();
      return st;
    }
    (void)retire_pending(d);
  }
  (void)wait_sink_jobs(*t->sink, "cc_detect_batch_discard");
  return CC_OK;
}

cc_status cc_detect_multiscale(cc_detector* d, const uint8_t* gray, int width, int height, size_t row_stride,
                               const cc_detect_params* p, cc_rect* out, int cap, int* n) {
  if (!n) return set_error(CC_ERR_INVALID_ARG, "cc_detect_multiscale: null count pointer");
  int32_t offsets[2] = {0, 0};
  cc_status st = cc_detect_batch(d, gray, 0, 1, width, height, row_stride, row_stride * (size_t)height, p, out, cap, offsets);
  if (st == CC_OK || st == CC_ERR_BUFFER_TOO_SMALL) *n = offsets[1];
  return st;
}

cc_status cc_detect_multiscale_levels(cc_detector* d, const uint8_t* gray, int width, int height, size_t row_stride,
                                      const cc_detect_params* p, cc_rect* out, int32_t* reject_levels, double* level_weights,
                                      int cap, int* n) {
  cc_status st = check_frame_args(d, gray, 1, width, height, row_stride, p, "cc_detect_multiscale_levels");
  if (st != CC_OK) return st;
  if (!n || cap < 0 || (cap > 0 && (!out || !reject_levels || !level_weights)))
    return set_error(CC_ERR_INVALID_ARG, "cc_detect_multiscale_levels: bad output buffers");
  std::vector<CandOut> cands;
  st = run_batch(d, gray, 0, 1, width, height, row_stride, row_stride * (size_t)height, p, true, false,
                 [&](int, int, std::vector<CandOut>& c) { cands.insert(cands.end(), c.begin(), c.end()); });
  if (st != CC_OK) return st;
  sort_candidates(cands);  // OpenCV's single-threaded order
  std::vector<cc_rect> rects;
  std::vector<int> levels;
  std::vector<double> weights;
  const int nstages = (int)d->m.stage_ntrees.size();
  for (const CandOut& c : cands) {  // only windows that passed every stage are reported: level = number of stages
    rects.push_back(cc_rect{c.x, c.y, c.w, c.h});
    levels.push_back(nstages);
    weights.push_back(c.sum);
  }
  group_rectangles(rects, p->min_neighbors, 0.2, &levels, &weights);
  *n = (int)rects.size();
  for (int i = 0; i < (int)rects.size() && i < cap; i++) {
    out[i] = rects[(size_t)i];
    reject_levels[i] = levels[(size_t)i];
    level_weights[i] = weights[(size_t)i];
  }
  if ((int)rects.size() > cap) return set_error(CC_ERR_BUFFER_TOO_SMALL, "cc_detect_multiscale_levels: %zu rectangles, capacity %d", rects.size(), cap);
  return CC_OK;
}

cc_status cc_detect_raw(cc_detector* d, const uint8_t* gray, int width, int height, size_t row_stride, const cc_detect_params* p,
                        int32_t* cand, int cap, int* n) {
  cc_status st = check_frame_args(d, gray, 1, width, height, row_stride, p, "cc_detect_raw");
  if (st != CC_OK) return st;
  if (!n || (cap > 0 && !cand)) return set_error(CC_ERR_INVALID_ARG, "cc_detect_raw: bad output buffers");
  std::vector<CandOut> cands;
  st = run_batch(d, gray, 0, 1, width, height, row_stride, row_stride * (size_t)height, p, true, false,
                 [&](int, int, std::vector<CandOut>& c) { cands.insert(cands.end(), c.begin(), c.end()); });
  if (st != CC_OK) return st;
  sort_candidates(cands);
  *n = (int)cands.size();
  for (int i = 0; i < (int)cands.size() && i < cap; i++) {
    const CandOut& c = cands[i];
    const int32_t v[7] = {c.scale, c.gx, c.gy, c.x, c.y, c.w, c.h};
    std::memcpy(cand + 7 * (size_t)i, v, sizeof(v));
  }
  if ((int)cands.size() > cap) return set_error(CC_ERR_BUFFER_TOO_SMALL, "cc_detect_raw: %zu candidates, capacity %d", cands.size(), cap);
  return CC_OK;
}

cc_status cc_detect_debug_windows(cc_detector* d, const uint8_t* gray, int width, int height, size_t row_stride,
                                  const cc_detect_params* p, int32_t* codes, double* sums, uint8_t* visited, int64_t cap,
                                  int64_t* n_windows) {
  cc_status st = check_frame_args(d, gray, 1, width, height, row_stride, p, "cc_detect_debug_windows");
  if (st != CC_OK) return st;
  if (!n_windows) return set_error(CC_ERR_INVALID_ARG, "cc_detect_debug_windows: null count pointer");
  std::vector<CandOut> cands;
  st = run_batch(d, gray, 0, 1, width, height, row_stride, row_stride * (size_t)height, p, true, true,
                 [&](int, int, std::vector<CandOut>& c) { cands.insert(cands.end(), c.begin(), c.end()); });
  if (st != CC_OK) return st;
  Plan* P = nullptr;
  st = build_plan(d, width, height, *p, &P);
  if (st != CC_OK) return st;
  *n_windows = P->windows;
  if (P->windows > cap) return set_error(CC_ERR_BUFFER_TOO_SMALL, "cc_detect_debug_windows: %lld windows, capacity %lld", P->windows, (long long)cap);
  const size_t nw = (size_t)P->windows;
  if (nw) {
    // through the detector's stream, not the legacy one: see ensure_spec_tiles
    if (codes) CC_HIP(hipMemcpyAsync(codes, d->d_dbg_codes.p, nw * sizeof(int32_t), hipMemcpyDeviceToHost, d->stream));
    if (sums) CC_HIP(hipMemcpyAsync(sums, d->d_dbg_sums.p, nw * sizeof(double), hipMemcpyDeviceToHost, d->stream));
    if (visited) CC_HIP(hipMemcpyAsync(visited, d->d_dbg_visited.p, nw, hipMemcpyDeviceToHost, d->stream));
    CC_HIP(hipStreamSynchronize(d->stream));
  }
  return CC_OK;
}

// ---- building blocks -------------------------------------------------------------------------------------------
cc_status cc_debug_vnf_check(int device, uint64_t n_values, uint64_t seed, uint64_t* mismatches) {
  if (!mismatches) return set_error(CC_ERR_INVALID_ARG, "cc_debug_vnf_check: null output");
  cc_status st = ensure_device(device);
  if (st != CC_OK) return st;
  unsigned long long* d = nullptr;
  CC_HIP(hipMalloc(reinterpret_cast<void**>(&d), sizeof(unsigned long long)));
  CC_HIP(hipMemset(d, 0, sizeof(unsigned long long)));
  const int threads = 256, blocks = 4096;
  const int per_thread = (int)std::max<uint64_t>(1, std::min<uint64_t>((n_values + (uint64_t)threads * blocks - 1) / ((uint64_t)threads * blocks), 1u << 20));
  hipLaunchKernelGGL(k_vnf_check, dim3(blocks), dim3(threads), 0, 0, (unsigned long long)seed, per_thread, d);
  unsigned long long h = 0;
  const hipError_t e2 = hipMemcpy(&h, d, sizeof(h), hipMemcpyDeviceToHost);
  (void)hipFree(d);
  if (e2 != hipSuccess) return set_error(CC_ERR_HIP, "cc_debug_vnf_check: %s", hipGetErrorString(e2));
  *mismatches = h;
  return CC_OK;
}

cc_status cc_resize_linear_exact_u8(int device, const uint8_t* src, int sw, int sh, size_t sstride, uint8_t* dst, int dw, int dh,
                                    size_t dstride) {
  if (!src || !dst || sw < 1 || sh < 1 || dw < 1 || dh < 1 || sstride < (size_t)sw || dstride < (size_t)dw)
    return set_error(CC_ERR_INVALID_ARG, "cc_resize_linear_exact_u8: bad argument");
  cc_status st = ensure_device(device);
  if (st != CC_OK) return st;
  ScaleDev S;
  std::memset(&S, 0, sizeof(S));
  S.w = dw;
  S.h = dh;
  S.pitch8 = align_up(dw, 4);
  AxisTaps tx, ty;
  linear_exact_taps(sw, dw, tx);
  linear_exact_taps(sh, dh, ty);
  std::vector<int> xofs_pad;
  std::vector<uint16_t> xw1_pad;
  S.xtab_ofs = append_column_taps(tx, xofs_pad, xw1_pad);
  std::vector<ScaleDev> sd{S};
  const int nblk = resize_blocks(S.pitch8, dh);
  std::vector<int> first{0, nblk};
  DevBuf<ScaleDev> d_sd;
  DevBuf<int> d_first, d_xofs, d_yofs;
  DevBuf<uint16_t> d_xw1, d_yw1;
  DevBuf<uint8_t> d_src, d_dst;
  const size_t spitch = (size_t)align_up(sw, 4);
  CC_HIP(d_sd.upload(sd, nullptr));
  CC_HIP(d_first.upload(first, nullptr));
  CC_HIP(d_xofs.upload(xofs_pad, nullptr));
  CC_HIP(d_yofs.upload(ty.ofs, nullptr));
  CC_HIP(d_xw1.upload(xw1_pad, nullptr));
  CC_HIP(d_yw1.upload(ty.w1, nullptr));
  CC_HIP(d_src.ensure(spitch * sh));
  CC_HIP(d_dst.ensure((size_t)S.pitch8 * dh));
  CC_HIP(hipMemcpy2D(d_src.p, spitch, src, sstride, sw, sh, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_resize, dim3(nblk, 1), dim3(256), 0, nullptr, d_src.p, spitch, (size_t)0, sw, sh, d_dst.p, (size_t)0,
                     d_sd.p, 1, d_first.p, d_xofs.p, d_xw1.p, d_yofs.p, d_yw1.p);
  CC_HIP(hipGetLastError());
  CC_HIP(hipMemcpy2D(dst, dstride, d_dst.p, S.pitch8, dw, dh, hipMemcpyDeviceToHost));
  return CC_OK;
}

cc_status cc_debug_stream_dwords(int device, size_t n_bytes, int repeats, uint32_t* checksum) {
  if (n_bytes < 4 || repeats < 1) return set_error(CC_ERR_INVALID_ARG, "cc_debug_stream_dwords: bad argument");
  cc_status st = ensure_device(device);
  if (st != CC_OK) return st;
  DevBuf<uint32_t> buf, out;
  const size_t n_words = n_bytes / 4;
  CC_HIP(buf.ensure(n_words));
  CC_HIP(out.ensure(32));
  CC_HIP(hipMemset(buf.p, 1, n_words * 4));
  CC_HIP(hipMemset(out.p, 0, 32 * 4));
  for (int r = 0; r < repeats; r++)
    hipLaunchKernelGGL(k_stream_dwords, dim3(256 * 8), dim3(256), 0, nullptr, buf.p, n_words, out.p);
  CC_HIP(hipGetLastError());
  uint32_t h[32];
  CC_HIP(hipMemcpy(h, out.p, sizeof(h), hipMemcpyDeviceToHost));
  uint32_t c = 0;
  for (int i = 1; i < 17; i++) c += h[i];
  if (checksum) *checksum = c;
  return CC_OK;
}

cc_status cc_integral_u8(int device, const uint8_t* img, int width, int height, size_t row_stride, int32_t* sum, int32_t* sqsum,
                         int32_t* tilted) {
  if (!img || width < 1 || height < 1 || row_stride < (size_t)width) return set_error(CC_ERR_INVALID_ARG, "cc_integral_u8: bad argument");
  cc_status st = ensure_device(device);
  if (st != CC_OK) return st;
  ScaleDev S;
  std::memset(&S, 0, sizeof(S));
  S.w = width;
  S.h = height;
  S.pitch8 = align_up(width, 4);
  S.pitchI = align_up(width + 1, 4);
  S.nbands = (height + INT_BAND - 1) / INT_BAND;
  S.h_ofs = 0;
  std::vector<ScaleDev> sd{S};
  std::vector<int> band_first{0, S.nbands}, col_first{0, (S.pitchI / 4 + 63) / 64};
  DevBuf<ScaleDev> d_sd;
  DevBuf<int> d_band_first, d_col_first;
  DevBuf<uint8_t> d_img;
  DevBuf<int32_t> d_int, d_h, d_tilt;
  const size_t elems = (size_t)S.pitchI * (height + 1), helems = (size_t)S.pitchI * S.nbands;
  CC_HIP(d_sd.upload(sd, nullptr));
  CC_HIP(d_band_first.upload(band_first, nullptr));
  CC_HIP(d_col_first.upload(col_first, nullptr));
  CC_HIP(d_img.ensure((size_t)S.pitch8 * height));
  CC_HIP(d_int.ensure(elems * 2));
  CC_HIP(d_h.ensure(helems * 2));
  CC_HIP(hipMemcpy2D(d_img.p, S.pitch8, img, row_stride, width, height, hipMemcpyHostToDevice));
  launch_integral(nullptr, true, d_img.p, 0, d_int.p, elems, 2, d_h.p, helems, d_sd.p, 1, d_band_first.p, S.nbands, d_col_first.p,
                  col_first[1], 1);
  CC_HIP(hipGetLastError());
  const size_t opitch = (size_t)(width + 1) * 4;
  if (sum) CC_HIP(hipMemcpy2D(sum, opitch, d_int.p, (size_t)S.pitchI * 4, opitch, height + 1, hipMemcpyDeviceToHost));
  if (sqsum) CC_HIP(hipMemcpy2D(sqsum, opitch, d_int.p + elems, (size_t)S.pitchI * 4, opitch, height + 1, hipMemcpyDeviceToHost));
  if (tilted) {  // same kernels as the detection pipeline
    DevBuf<int32_t> d_diag;
    std::vector<int> diag_first{0, (width + height - 1 + 255) / 256}, tcol_first{0, (width + 1 + 63) / 64};
    DevBuf<int> d_diag_first, d_tcol_first;
    CC_HIP(d_diag_first.upload(diag_first, nullptr));
    CC_HIP(d_tcol_first.upload(tcol_first, nullptr));
    CC_HIP(d_diag.ensure(elems * 2));
    CC_HIP(d_tilt.ensure(elems));
    TiltPlan tp;
    CC_HIP(tp.build(std::vector<ScaleDev>(1, S), nullptr));
    DevBuf<int32_t> d_tseg;
    CC_HIP(d_tseg.ensure(std::max<size_t>(tp.frame_elems, 1)));
    launch_tilted(nullptr, tp, d_tseg.p, d_img.p, (size_t)0, d_diag.p, d_tilt.p, elems, 1, 0, d_sd.p, 1, d_diag_first.p, diag_first[1],
                  d_tcol_first.p, tcol_first[1], 1);
    CC_HIP(hipGetLastError());
    CC_HIP(hipMemcpy2D(tilted, opitch, d_tilt.p, (size_t)S.pitchI * 4, opitch, height + 1, hipMemcpyDeviceToHost));
  }
  return CC_OK;
}

// ---- negative mining ---------------------------------------------------------------------------------------------
namespace {

struct MineGeom {
  int w, h, nx, ny;
};

// The reader's scale ladder and window grid for one image (imagestorage.cpp:57-126), in its float arithmetic.
void mine_ladder(int W0, int H0, int cols, int rows, int ox, int oy, std::vector<MineGeom>& out) {
  out.clear();
  const float scaleFactor = 1.4142135623730950488016887242097F, stepFactor = 0.5F;
  float scale = std::max(((float)W0 + ox) / ((float)cols), ((float)H0 + oy) / ((float)rows));
  int lw = (int)(scale * cols + 0.5F), lh = (int)(scale * rows + 0.5F);
  for (;;) {
    MineGeom g{lw, lh, 0, 0};
    int x = ox;
    g.nx = 1;
    while ((int)(x + (1.0F + stepFactor) * W0) < lw) {
      x += (int)(stepFactor * W0);
      g.nx++;
    }
    int y = oy;
    g.ny = 1;
    while ((int)(y + (1.0F + stepFactor) * H0) < lh) {
      y += (int)(stepFactor * H0);
      g.ny++;
    }
    out.push_back(g);
    scale *= scaleFactor;
    if (!(scale <= 1.0F) || out.size() > 64) break;
    lw = (int)(scale * cols);
    lh = (int)(scale * rows);
  }
}

cc_status mine_check(const cc_negminer* m, int width, int height, int ox, int oy, const char* who) {
  if (!m) return set_error(CC_ERR_INVALID_ARG, "%s: null miner", who);
  if (width < 1 || height < 1 || width > 32768 || height > 32768) return set_error(CC_ERR_INVALID_ARG, "%s: bad image size", who);
  // NegReader::nextImg only accepts offsets with 0 <= ox <= cols - W, 0 <= oy <= rows - H
  if (ox < 0 || oy < 0 || ox > width - m->m.win_w || oy > height - m->m.win_h)
    return set_error(CC_ERR_INVALID_ARG, "%s: offset (%d,%d) does not leave room for a %dx%d window in a %dx%d image", who, ox, oy,
                     m->m.win_w, m->m.win_h, width, height);
  return CC_OK;
}

}  // namespace

cc_status cc_negminer_create(const cc_cascade* c, int device, cc_negminer** out) {
  if (!c || !out) return set_error(CC_ERR_INVALID_ARG, "cc_negminer_create: null argument");
  *out = nullptr;
  cc_status st = ensure_device(device);
  if (st != CC_OK) return st;
  std::unique_ptr<cc_negminer> m(new cc_negminer());
  m->m = c->m;
  m->device = device;
  CC_HIP(hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking));
  CC_HIP(hipStreamCreateWithFlags(&m->copy_stream, hipStreamNonBlocking));
  const Cascade& M = m->m;
  const bool haar = M.feature_type == CC_FEATURE_HAAR;
  std::vector<MineNode> nodes(M.node_feature.size());
  for (size_t i = 0; i < nodes.size(); i++) {
    MineNode& n = nodes[i];
    std::memset(&n, 0, sizeof(n));
    const int fi = M.node_feature[i];
    if (haar) {
      bool used = true;
      for (int j = 0; j < 3; j++) {
        const float wt = M.haar_weights[(size_t)fi * 3 + j];
        if (wt == 0.0f) used = false;  // offsets stay 0 from the first zero weight on (haarfeatures.cpp:292-308)
        if (!used) continue;
        n.w[j] = wt;
        for (int k = 0; k < 4; k++) n.r[j][k] = M.haar_rects[(size_t)fi * 12 + j * 4 + k];
      }
      n.tilted = M.haar_tilted[fi];
      n.thr = M.node_threshold[i];
    } else {
      for (int k = 0; k < 4; k++) n.r[0][k] = M.lbp_rects[(size_t)fi * 4 + k];
      for (int j = 0; j < 8; j++) n.subset[j] = M.node_subset[i * 8 + j];
    }
    n.left = M.node_left[i];
    n.right = M.node_right[i];
  }
  std::vector<int> sfirst(M.stage_first.begin(), M.stage_first.end()), sn(M.stage_ntrees.begin(), M.stage_ntrees.end());
  std::vector<int> root(M.tree_first_node.begin(), M.tree_first_node.end()), leaf0(M.tree_first_leaf.begin(), M.tree_first_leaf.end());
  CC_HIP(m->d_nodes.upload(nodes, m->stream));
  CC_HIP(m->d_stage_first.upload(sfirst, m->stream));
  CC_HIP(m->d_stage_ntrees.upload(sn, m->stream));
  CC_HIP(m->d_stage_thr.upload(M.stage_threshold, m->stream));  // already threshold - 1e-5f (CV_THRESHOLD_EPS)
  CC_HIP(m->d_tree_root.upload(root, m->stream));
  CC_HIP(m->d_tree_leaf0.upload(leaf0, m->stream));
  CC_HIP(m->d_leaves.upload(M.leaves, m->stream));
  CC_HIP(hipStreamSynchronize(m->stream));
  *out = m.release();
  return CC_OK;
}

void cc_negminer_destroy(cc_negminer* m) {
  if (!m) return;
  (void)hipSetDevice(m->device);
  delete m;
}

cc_status cc_negminer_plan(const cc_negminer* m, int width, int height, int ox, int oy, int32_t* lw, int32_t* lh, int32_t* nx,
                           int32_t* ny, int cap, int* n_levels, int64_t* n_windows) {
  cc_status st = mine_check(m, width, height, ox, oy, "cc_negminer_plan");
  if (st != CC_OK) return st;
  if (!n_levels || !n_windows) return set_error(CC_ERR_INVALID_ARG, "cc_negminer_plan: null output");
  std::vector<MineGeom> g;
  mine_ladder(m->m.win_w, m->m.win_h, width, height, ox, oy, g);
  *n_levels = (int)g.size();
  *n_windows = 0;
  for (size_t i = 0; i < g.size(); i++) {
    *n_windows += (int64_t)g[i].nx * g[i].ny;
    if ((int)i < cap) {
      if (lw) lw[i] = g[i].w;
      if (lh) lh[i] = g[i].h;
      if (nx) nx[i] = g[i].nx;
      if (ny) ny[i] = g[i].ny;
    }
  }
  return CC_OK;
}

// Tables of one (image size, offset): ladder geometry, resize taps, kernel block maps. Cached in m->plan.
static cc_status mine_plan(cc_negminer* m, int width, int height, int ox, int oy, const char* who) {
  cc_negminer::Plan& P = m->plan;
  if (P.width == width && P.height == height && P.ox == ox && P.oy == oy) return CC_OK;
  P.width = -1;  // invalid until everything below has succeeded
  const Cascade& M = m->m;
  const int W0 = M.win_w, H0 = M.win_h;
  const bool haar = M.feature_type == CC_FEATURE_HAAR, tilt = haar && M.has_tilted;
  std::vector<MineGeom> g;
  mine_ladder(W0, H0, width, height, ox, oy, g);
  const int nl = (int)g.size();
  std::vector<ScaleDev> sd((size_t)nl);
  std::vector<MineLevel> lv((size_t)nl);
  std::vector<int> resize_first(nl + 1, 0), band_first(nl + 1, 0), col_first(nl + 1, 0), diag_first(nl + 1, 0), tcol_first(nl + 1, 0);
  std::vector<int> xofs, yofs;
  std::vector<uint16_t> xw1, yw1;
  long long img_ofs = 0, int_ofs = 0, h_ofs = 0, wins = 0;
  for (int i = 0; i < nl; i++) {
    ScaleDev& S = sd[(size_t)i];
    std::memset(&S, 0, sizeof(S));
    S.w = g[i].w;
    S.h = g[i].h;
    if (S.w < W0 + ox || S.h < H0 + oy) return set_error(CC_ERR_INVALID_ARG, "%s: ladder level %d (%dx%d) smaller than window + offset", who, i, S.w, S.h);
    S.pitch8 = align_up(S.w, 4);
    S.pitchI = align_up(S.w + 1, 4);
    S.img_ofs = img_ofs;
    S.int_ofs = int_ofs;
    S.h_ofs = h_ofs;
    S.nbands = (S.h + INT_BAND - 1) / INT_BAND;
    S.ytab_ofs = (int)yofs.size();
    AxisTaps tx, ty;
    linear_exact_taps(width, S.w, tx);
    linear_exact_taps(height, S.h, ty);
    S.xtab_ofs = append_column_taps(tx, xofs, xw1);
    yofs.insert(yofs.end(), ty.ofs.begin(), ty.ofs.end());
    yw1.insert(yw1.end(), ty.w1.begin(), ty.w1.end());
    MineLevel& L = lv[(size_t)i];
    L.w = S.w;
    L.h = S.h;
    L.pitchI = S.pitchI;
    L.pitch8 = S.pitch8;
    L.nx = g[i].nx;
    L.ny = g[i].ny;
    L.int_ofs = int_ofs;
    L.img_ofs = img_ofs;
    L.win_first = wins;
    L.pad = 0;
    wins += (long long)g[i].nx * g[i].ny;
    img_ofs += (long long)align_up(S.pitch8 * S.h, 16);
    int_ofs += (long long)S.pitchI * (S.h + 1);
    h_ofs += (long long)S.nbands * S.pitchI;
    resize_first[i + 1] = resize_first[i] + resize_blocks(S.pitch8, S.h);
    band_first[i + 1] = band_first[i] + S.nbands;
    col_first[i + 1] = col_first[i] + (S.pitchI / 4 + 63) / 64;
    diag_first[i + 1] = diag_first[i] + (S.w + S.h - 1 + 255) / 256;
    tcol_first[i + 1] = tcol_first[i] + (S.w + 1 + 63) / 64;
  }
  hipStream_t s = m->stream;
  CC_HIP(m->d_sd.upload(sd, s));
  CC_HIP(m->d_levels.upload(lv, s));
  CC_HIP(m->d_resize_first.upload(resize_first, s));
  CC_HIP(m->d_band_first.upload(band_first, s));
  CC_HIP(m->d_col_first.upload(col_first, s));
  CC_HIP(m->d_diag_first.upload(diag_first, s));
  CC_HIP(m->d_tcol_first.upload(tcol_first, s));
  CC_HIP(m->d_xofs.upload(xofs, s));
  CC_HIP(m->d_yofs.upload(yofs, s));
  CC_HIP(m->d_xw1.upload(xw1, s));
  CC_HIP(m->d_yw1.upload(yw1, s));
  if (tilt) CC_HIP(m->tilt.build(sd, s));
  CC_HIP(hipStreamSynchronize(s));  // the uploads read host vectors that end here
  P.nl = nl;
  P.n_resize = resize_first[nl];
  P.n_bands = band_first[nl];
  P.n_cols = col_first[nl];
  P.n_diag = diag_first[nl];
  P.n_tcol = tcol_first[nl];
  P.pyr_bytes = (img_ofs + 15) & ~15LL;
  P.chan_elems = int_ofs;
  P.h_elems = h_ofs;
  P.wins = wins;
  P.ox = ox;
  P.oy = oy;
  P.height = height;
  P.width = width;
  return CC_OK;
}

static cc_status pinned_ensure(uint8_t** p, size_t* have, size_t need) {
  if (*have >= need) return CC_OK;
  if (*p) (void)hipHostFree(*p);
  *p = nullptr;
  *have = 0;
  CC_HIP(hipHostMalloc(reinterpret_cast<void**>(p), need, hipHostMallocDefault));
  *have = need;
  return CC_OK;
}

// n_images images of one size, consumed with one offset: ONE copy to the device, one launch of every kernel over all of
// them (the front-end kernels and the window kernels take the image as blockIdx.y, like the detector's frames), one copy back.
static cc_status mine_images(cc_negminer* m, const uint8_t* const* images, int n_images, int width, int height, size_t row_stride, int ox,
                             int oy, uint8_t* pass, int64_t cap, int64_t* n_windows, uint8_t* pixels, int64_t* keep_index, int max_keep,
                             int* n_keep, const char* who) {
  cc_status st = mine_check(m, width, height, ox, oy, who);
  if (st != CC_OK) return st;
  if (!images || n_images < 1 || !pass || !n_windows || row_stride < (size_t)width) return set_error(CC_ERR_INVALID_ARG, "%s: bad argument", who);
  for (int k = 0; k < n_images; k++)
    if (!images[k]) return set_error(CC_ERR_INVALID_ARG, "%s: image %d is null", who, k);
  if (pixels && (!keep_index || !n_keep || max_keep < 0)) return set_error(CC_ERR_INVALID_ARG, "%s: bad keep buffers", who);
  st = ensure_device(m->device);
  if (st != CC_OK) return st;
  st = mine_plan(m, width, height, ox, oy, who);
  if (st != CC_OK) return st;
  const cc_negminer::Plan& P = m->plan;
  const Cascade& M = m->m;
  const int W0 = M.win_w, H0 = M.win_h, nl = P.nl, K = n_images;
  const bool haar = M.feature_type == CC_FEATURE_HAAR, tilt = haar && M.has_tilted;
  const long long wins = P.wins;
  *n_windows = wins;
  if (wins * K > cap) return set_error(CC_ERR_BUFFER_TOO_SMALL, "%s: %lld windows (%d images), capacity %lld", who, wins * K, K, (long long)cap);
  hipStream_t s = m->stream;
  const int nchan = haar ? (tilt ? 3 : 2) : 1;
  const size_t chan_elems = (size_t)P.chan_elems, spitch = (size_t)align_up(width, 4), src_bytes = spitch * (size_t)height;
  CC_HIP(m->d_src.ensure(src_bytes * K));
  CC_HIP(m->d_pyr.ensure((size_t)P.pyr_bytes * K));
  CC_HIP(m->d_integ.ensure(chan_elems * (size_t)nchan * K));
  CC_HIP(m->d_hbuf.ensure(std::max<size_t>((size_t)P.h_elems * (size_t)nchan * K, 4)));
  CC_HIP(m->d_pass.ensure((size_t)std::max<long long>(wins * K, 1)));
  st = pinned_ensure(&m->h_src, &m->h_src_bytes, src_bytes * K);
  if (st != CC_OK) return st;
  st = pinned_ensure(&m->h_pass, &m->h_pass_bytes, (size_t)std::max<long long>(wins * K, 1));
  if (st != CC_OK) return st;
  MineArgs A;
  A.integ = m->d_integ.p;
  A.chan_elems = chan_elems;
  A.nchan = nchan;
  A.levels = m->d_levels.p;
  A.n_levels = nl;
  A.n_windows = wins;
  A.W0 = W0;
  A.H0 = H0;
  A.ox = ox;
  A.oy = oy;
  A.sx = (int)(0.5F * W0);
  A.sy = (int)(0.5F * H0);
  A.nstages = (int)M.stage_ntrees.size();
  A.stage_first = m->d_stage_first.p;
  A.stage_ntrees = m->d_stage_ntrees.p;
  A.stage_thr = m->d_stage_thr.p;
  A.nodes = m->d_nodes.p;
  A.tree_root = m->d_tree_root.p;
  A.tree_leaf0 = m->d_tree_leaf0.p;
  A.leaves = m->d_leaves.p;
  A.pass = m->d_pass.p;
  if (tilt) {
    CC_HIP(m->d_diag.ensure(chan_elems * 2 * K));
    CC_HIP(m->d_tseg.ensure(std::max<size_t>(m->tilt.frame_elems * K, 1)));
  }
  // one wavefront per window where the parallel stage sum is exact (stumps, order-independent sums); else one thread per window
  const bool wave_mode = M.max_nodes_per_tree == 1 && stage_sums_order_independent(M) && !std::getenv("CCAMD_NEGMINE_THREAD_PER_WINDOW");
  // Every kernel over the images [k0, k0 + n): the front-end kernels and the window kernels take the image as blockIdx.y.
  auto launch_images = [&](int k0, int n) {
    const uint8_t* src = m->d_src.p + (size_t)k0 * src_bytes;
    uint8_t* pyr = m->d_pyr.p + (size_t)k0 * (size_t)P.pyr_bytes;
    int32_t* integ = m->d_integ.p + (size_t)k0 * nchan * chan_elems;
    hipLaunchKernelGGL(k_resize, dim3(P.n_resize, n), dim3(256), 0, s, src, spitch, src_bytes, width, height, pyr, (size_t)P.pyr_bytes,
                       m->d_sd.p, nl, m->d_resize_first.p, m->d_xofs.p, m->d_xw1.p, m->d_yofs.p, m->d_yw1.p);
    launch_integral(s, haar, pyr, (size_t)P.pyr_bytes, integ, chan_elems, nchan, m->d_hbuf.p + (size_t)k0 * nchan * (size_t)P.h_elems, (size_t)P.h_elems,
                    m->d_sd.p, nl, m->d_band_first.p, P.n_bands, m->d_col_first.p, P.n_cols, n);
    if (tilt)
      launch_tilted(s, m->tilt, m->d_tseg.p + (size_t)k0 * m->tilt.frame_elems, pyr, (size_t)P.pyr_bytes, m->d_diag.p + (size_t)k0 * 2 * chan_elems, integ,
                    chan_elems, nchan, 2, m->d_sd.p, nl, m->d_diag_first.p, P.n_diag, m->d_tcol_first.p, P.n_tcol, n);
    if (wins == 0) return;
    MineArgs B = A;
    B.integ = integ;
    B.pass = m->d_pass.p + (size_t)k0 * (size_t)wins;
    if (wave_mode) {
      const unsigned nb = (unsigned)((wins + 3) / 4);
      if (haar)
        hipLaunchKernelGGL(k_negmine_wave<true>, dim3(nb, n), dim3(256), 0, s, B);
      else
        hipLaunchKernelGGL(k_negmine_wave<false>, dim3(nb, n), dim3(256), 0, s, B);
    } else {
      const unsigned nb = (unsigned)((wins + 255) / 256);
      if (haar)
        hipLaunchKernelGGL(k_negmine_windows<true>, dim3(nb, n), dim3(256), 0, s, B);
      else
        hipLaunchKernelGGL(k_negmine_windows<false>, dim3(nb, n), dim3(256), 0, s, B);
    }
  };
  // Pageable rows -> pinned, tight rows, then asynchronous transfers on the copy stream: the images travel in pieces of >= 8 MB;
  // a piece's copy is issued as soon as it is staged (it runs under the staging of the next piece) and its kernels are queued
  // behind it on the compute stream (they run under the next piece's copy). A Full-HD background is 2 MB for 13 584 windows:
  // the image's way to the device is most of what a call costs. Large pieces are staged by up to 4 threads.
  {
    auto stage = [&](int ka, int kb) {
      for (int k = ka; k < kb; k++) {
        uint8_t* dst = m->h_src + (size_t)k * src_bytes;
        if (row_stride == spitch)
          std::memcpy(dst, images[k], (size_t)(height - 1) * spitch + (size_t)width);
        else
          for (int y = 0; y < height; y++) std::memcpy(dst + (size_t)y * spitch, images[k] + (size_t)y * row_stride, (size_t)width);
      }
    };
    const int per_piece = (int)std::max<size_t>(1, ((size_t)8 << 20) / std::max<size_t>(src_bytes, 1));
    const size_t n_pieces = ((size_t)K + per_piece - 1) / per_piece;
    while (m->piece_landed.size() < n_pieces) {
      hipEvent_t ev = nullptr;
      CC_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
      m->piece_landed.push_back(ev);
    }
    size_t piece = 0;
    for (int k0 = 0; k0 < K; k0 += per_piece, piece++) {
      const int k1 = std::min(K, k0 + per_piece), n = k1 - k0;
      const int nt = std::min({4, n, (int)std::max<size_t>(1, ((size_t)n * src_bytes) >> 21)});
      if (nt <= 1) {
        stage(k0, k1);
      } else {
        std::vector<std::future<void>> jobs;
        try {
          for (int t = 1; t < nt; t++) jobs.push_back(std::async(std::launch::async, stage, k0 + (int)((long long)n * t / nt), k0 + (int)((long long)n * (t + 1) / nt)));
          stage(k0, k0 + n / nt);
          for (auto& j : jobs) j.get();
        } catch (const std::exception& e) {
          for (auto& j : jobs)
            if (j.valid()) j.wait();
          (void)hipStreamSynchronize(m->copy_stream);
          (void)hipStreamSynchronize(s);
          return set_error(CC_ERR_HIP, "%s: staging the images: %s", who, e.what());
        }
      }
      CC_HIP(hipMemcpyAsync(m->d_src.p + (size_t)k0 * src_bytes, m->h_src + (size_t)k0 * src_bytes, (size_t)n * src_bytes, hipMemcpyHostToDevice,
                            m->copy_stream));
      CC_HIP(hipEventRecord(m->piece_landed[piece], m->copy_stream));
      CC_HIP(hipStreamWaitEvent(s, m->piece_landed[piece], 0));
      launch_images(k0, n);
    }
  }
  CC_HIP(hipGetLastError());
  if (wins > 0) CC_HIP(hipMemcpyAsync(m->h_pass, m->d_pass.p, (size_t)(wins * K), hipMemcpyDeviceToHost, s));
  CC_HIP(hipStreamSynchronize(s));
  if (wins > 0) std::memcpy(pass, m->h_pass, (size_t)(wins * K));
  if (pixels) {
    std::vector<long long> keep;
    for (long long i = 0; i < wins * K && (int)keep.size() < max_keep; i++)
      if (pass[i]) keep.push_back(i);
    *n_keep = (int)keep.size();
    if (!keep.empty()) {
      const size_t wsz = (size_t)W0 * H0;
      CC_HIP(m->d_keep.upload(keep, s));
      CC_HIP(m->d_pix.ensure(keep.size() * wsz));
      hipLaunchKernelGGL(k_negmine_gather, dim3((unsigned)keep.size()), dim3(64), 0, s, m->d_pyr.p, (size_t)P.pyr_bytes, wins, m->d_levels.p, nl,
                         m->d_keep.p, W0, H0, ox, oy, A.sx, A.sy, m->d_pix.p);
      CC_HIP(hipGetLastError());
      CC_HIP(hipMemcpyAsync(pixels, m->d_pix.p, keep.size() * wsz, hipMemcpyDeviceToHost, s));
      CC_HIP(hipStreamSynchronize(s));
      for (size_t i = 0; i < keep.size(); i++) keep_index[i] = keep[i];
    }
  }
  return CC_OK;
}

cc_status cc_negminer_run(cc_negminer* m, const uint8_t* gray, int width, int height, size_t row_stride, int ox, int oy, uint8_t* pass,
                          int64_t cap, int64_t* n_windows, uint8_t* pixels, int64_t* keep_index, int max_keep, int* n_keep) {
  if (!m) return set_error(CC_ERR_INVALID_ARG, "cc_negminer_run: null miner");
  return mine_images(m, &gray, 1, width, height, row_stride, ox, oy, pass, cap, n_windows, pixels, keep_index, max_keep, n_keep, "cc_negminer_run");
}

cc_status cc_negminer_run_batch(cc_negminer* m, const uint8_t* const* images, int n_images, int width, int height, size_t row_stride, int ox,
                                int oy, uint8_t* pass, int64_t cap, int64_t* n_windows, uint8_t* pixels, int64_t* keep_index, int max_keep,
                                int* n_keep) {
  if (!m) return set_error(CC_ERR_INVALID_ARG, "cc_negminer_run_batch: null miner");
  if (n_images > 256) return set_error(CC_ERR_INVALID_ARG, "cc_negminer_run_batch: at most 256 images per call (%d given)", n_images);
  return mine_images(m, images, n_images, width, height, row_stride, ox, oy, pass, cap, n_windows, pixels, keep_index, max_keep, n_keep,
                     "cc_negminer_run_batch");
}

}  // extern "C"
