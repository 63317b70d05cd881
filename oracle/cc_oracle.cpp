// cc_oracle.cpp — CPU ORACLE. TEST INFRASTRUCTURE ONLY.
//
// A plain, scalar CPU restatement of the reference's hot path, used ONLY by tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg as the checker / reported baseline.
// Nothing under cascadeclassifier_amd/ (the product) may include, link or call this file.
//
// What it restates, and where it comes from (paths relative to /root/reference):
//   training side (in-tree, PINNED by the reference's own KATs, see tests/test_oracle_kats.py):
//     CV_SUM_OFFSETS / CV_TILTED_OFFSETS   traincascade/lib/include/traincascade_features.h:40-63
//     calcNormFactor                       traincascade/lib/src/features.cpp:13-25
//     CvHaarEvaluator::generateFeatures    traincascade/lib/src/haarfeatures.cpp:127-251
//     CvHaarEvaluator::Feature::calc       traincascade/lib/include/haarfeatures.h:114-122
//     CvHaarEvaluator::operator()          traincascade/lib/include/haarfeatures.h:108-112
//     CvLBPEvaluator::generateFeatures     traincascade/lib/src/lbpfeatures.cpp:35-63
//     CvLBPEvaluator::Feature::calc        traincascade/lib/include/lbpfeatures.h:70-83
//     CvCascadeBoostTree::predict          traincascade/lib/src/o_cvcascadeboosttree.cpp:16-39
//     CvCascadeBoost::predict              traincascade/lib/src/boost.cpp:461-477
//   detection side: the arithmetic lives in OpenCV 4.6.0 (pinned at external/CMakeLists.txt:11), which
//   is NOT vendored in the reference and NOT installed here. Call sites: tools/detection/Cpp/main.cpp:42,45,
//   tools/detection/Python/detect.py:16,22, haarfeatures.cpp:109,112, lbpfeatures.cpp:27. The functions
//   below restate the published algorithm of cv::integral, cv::resize(INTER_LINEAR_EXACT),
//   cv::CascadeClassifier::detectMultiScale and cv::groupRectangles.
//   PARITY UNPINNED for that part: the reference holds no golden vector for any detection result
//   (test_integration.cpp only asserts !empty()); cv::integral is pinned through the calc KATs.
//
// Build: see oracle/Makefile (g++ -O2 -ffp-contract=off).

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#if defined(__GNUC__)
#define ORC_API extern "C" __attribute__((visibility("default")))
#else
#define ORC_API extern "C"
#endif

namespace {

// cvRound(double/float): round half to even (SSE cvtsd2si semantics under the default rounding mode).
inline int cv_round_d(double v) { return (int)std::nearbyint(v); }
inline int cv_round_f(float v) { return (int)std::nearbyintf(v); }

}  // namespace

// ---------------------------------------------------------------------------------------------
// Integral images (cv::integral semantics). Outputs are (h+1) x (w+1), row 0 / col 0 zero.
// sum: int32. sqsum_f64: double (training path, haarfeatures.cpp:109,112 default sdepth for sqsum).
// sqsum_i32: int32 with wrap-around accumulation (detector path). tilted: int32, 45-degree triangle sums.
// ---------------------------------------------------------------------------------------------
ORC_API void orc_integral_u8(const uint8_t* img, int w, int h, int stride, int32_t* sum, double* sqsum_f64,
                             int32_t* sqsum_i32, int32_t* tilted) {
  const int sw = w + 1;
  if (sum) {
    for (int x = 0; x <= w; x++) sum[x] = 0;
    for (int y = 0; y < h; y++) {
      int32_t rs = 0;
      sum[(y + 1) * sw] = 0;
      for (int x = 0; x < w; x++) {
        rs += img[y * stride + x];
        sum[(y + 1) * sw + x + 1] = sum[y * sw + x + 1] + rs;
      }
    }
  }
  if (sqsum_f64) {
    for (int x = 0; x <= w; x++) sqsum_f64[x] = 0;
    for (int y = 0; y < h; y++) {
      double rs = 0;
      sqsum_f64[(y + 1) * sw] = 0;
      for (int x = 0; x < w; x++) {
        double p = img[y * stride + x];
        rs += p * p;
        sqsum_f64[(y + 1) * sw + x + 1] = sqsum_f64[y * sw + x + 1] + rs;
      }
    }
  }
  if (sqsum_i32) {
    uint32_t* q = (uint32_t*)sqsum_i32;  // wrap-around arithmetic, identical bits to int32 overflow
    for (int x = 0; x <= w; x++) q[x] = 0;
    for (int y = 0; y < h; y++) {
      uint32_t rs = 0;
      q[(y + 1) * sw] = 0;
      for (int x = 0; x < w; x++) {
        uint32_t p = img[y * stride + x];
        rs += p * p;
        q[(y + 1) * sw + x + 1] = q[y * sw + x + 1] + rs;
      }
    }
  }
  if (tilted) {
    // tilted(Y,X) = sum over pixels (y,x) with y < Y and |x - X + 1| <= Y - y - 1 (pixels outside the image = 0).
    std::vector<int32_t> rowpre((size_t)h * (w + 1));
    for (int y = 0; y < h; y++) {
      rowpre[(size_t)y * (w + 1)] = 0;
      for (int x = 0; x < w; x++) rowpre[(size_t)y * (w + 1) + x + 1] = rowpre[(size_t)y * (w + 1) + x] + img[y * stride + x];
    }
    for (int Y = 0; Y <= h; Y++)
      for (int X = 0; X <= w; X++) {
        int32_t acc = 0;
        for (int y = 0; y < Y; y++) {
          int half = Y - y - 1;
          int x0 = std::max(X - 1 - half, 0), x1 = std::min(X - 1 + half, w - 1);
          if (x1 >= x0) acc += rowpre[(size_t)y * (w + 1) + x1 + 1] - rowpre[(size_t)y * (w + 1) + x0];
        }
        tilted[Y * sw + X] = acc;
      }
  }
}

// ---------------------------------------------------------------------------------------------
// Feature catalogs. A Haar feature = up to 3 weighted rects (+ tilted flag); LBP = one cell rect.
// ---------------------------------------------------------------------------------------------
struct OrcHaarFeature {
  int32_t tilted;
  int32_t r[3][4];  // x y w h
  float wt[3];
};

namespace {

struct HaarSink {
  OrcHaarFeature* out;
  int cap;
  int n;
  void add(bool tilted, int x0, int y0, int w0, int h0, float wt0, int x1, int y1, int w1, int h1, float wt1, int x2 = 0,
           int y2 = 0, int w2 = 0, int h2 = 0, float wt2 = 0.f) {
    if (out && n < cap) {
      OrcHaarFeature& f = out[n];
      f.tilted = tilted ? 1 : 0;
      int v[3][4] = {{x0, y0, w0, h0}, {x1, y1, w1, h1}, {x2, y2, w2, h2}};
      std::memcpy(f.r, v, sizeof(v));
      f.wt[0] = wt0;
      f.wt[1] = wt1;
      f.wt[2] = wt2;
    }
    n++;
  }
};

}  // namespace

// mode: 0 BASIC, 1 CORE, 2 ALL. Returns catalog size; fills out[0..min(cap,n)) if out != NULL.
// Order and weights exactly as haarfeatures.cpp:127-251 (note x3/y3 use weight +2 there).
ORC_API int orc_haar_catalog(int W, int H, int mode, OrcHaarFeature* out, int cap) {
  HaarSink s{out, cap, 0};
  for (int x = 0; x < W; x++)
    for (int y = 0; y < H; y++)
      for (int dx = 1; dx <= W; dx++)
        for (int dy = 1; dy <= H; dy++) {
          if (x + dx * 2 <= W && y + dy <= H) s.add(false, x, y, dx * 2, dy, -1.f, x + dx, y, dx, dy, +2.f);
          if (x + dx <= W && y + dy * 2 <= H) s.add(false, x, y, dx, dy * 2, -1.f, x, y + dy, dx, dy, +2.f);
          if (x + dx * 3 <= W && y + dy <= H) s.add(false, x, y, dx * 3, dy, -1.f, x + dx, y, dx, dy, +2.f);
          if (x + dx <= W && y + dy * 3 <= H) s.add(false, x, y, dx, dy * 3, -1.f, x, y + dy, dx, dy, +2.f);
          if (mode != 0) {
            if (x + dx * 4 <= W && y + dy <= H) s.add(false, x, y, dx * 4, dy, -1.f, x + dx, y, dx * 2, dy, +2.f);
            if (x + dx <= W && y + dy * 4 <= H) s.add(false, x, y, dx, dy * 4, -1.f, x, y + dy, dx, dy * 2, +2.f);
          }
          if (x + dx * 2 <= W && y + dy * 2 <= H)
            s.add(false, x, y, dx * 2, dy * 2, -1.f, x, y, dx, dy, +2.f, x + dx, y + dy, dx, dy, +2.f);
          if (mode != 0) {
            if (x + dx * 3 <= W && y + dy * 3 <= H) s.add(false, x, y, dx * 3, dy * 3, -1.f, x + dx, y + dy, dx, dy, +9.f);
          }
          if (mode == 2) {
            if (x + 2 * dx <= W && y + 2 * dx + dy <= H && x - dy >= 0) s.add(true, x, y, dx * 2, dy, -1.f, x, y, dx, dy, +2.f);
            if (x + dx <= W && y + dx + 2 * dy <= H && x - 2 * dy >= 0) s.add(true, x, y, dx, 2 * dy, -1.f, x, y, dx, dy, +2.f);
            if (x + 3 * dx <= W && y + 3 * dx + dy <= H && x - dy >= 0)
              s.add(true, x, y, dx * 3, dy, -1.f, x + dx, y + dx, dx, dy, +3.f);
            if (x + dx <= W && y + dx + 3 * dy <= H && x - 3 * dy >= 0)
              s.add(true, x, y, dx, 3 * dy, -1.f, x - dy, y + dy, dx, dy, +3.f);
            if (x + 4 * dx <= W && y + 4 * dx + dy <= H && x - dy >= 0)
              s.add(true, x, y, dx * 4, dy, -1.f, x + dx, y + dx, dx * 2, dy, +2.f);
            if (x + dx <= W && y + dx + 4 * dy <= H && x - 4 * dy >= 0)
              s.add(true, x, y, dx, 4 * dy, -1.f, x - dy, y + dy, dx, 2 * dy, +2.f);
          }
        }
  return s.n;
}

// lbpfeatures.cpp:35-45. out[i] = {x, y, w, h} of one cell of the 3x3 grid.
ORC_API int orc_lbp_catalog(int W, int H, int32_t* out, int cap) {
  int n = 0;
  for (int x = 0; x < W; x++)
    for (int y = 0; y < H; y++)
      for (int w = 1; w <= W / 3; w++)
        for (int h = 1; h <= H / 3; h++)
          if (x + 3 * w <= W && y + 3 * h <= H) {
            if (out && n < cap) {
              out[4 * n + 0] = x;
              out[4 * n + 1] = y;
              out[4 * n + 2] = w;
              out[4 * n + 3] = h;
            }
            n++;
          }
  return n;
}

namespace {

// traincascade_features.h:40-63.
inline void sum_offsets(int p[4], const int32_t r[4], int step) {
  p[0] = r[0] + step * r[1];
  p[1] = r[0] + r[2] + step * r[1];
  p[2] = r[0] + step * (r[1] + r[3]);
  p[3] = r[0] + r[2] + step * (r[1] + r[3]);
}
inline void tilted_offsets(int p[4], const int32_t r[4], int step) {
  p[0] = r[0] + step * r[1];
  p[1] = r[0] - r[3] + step * (r[1] + r[3]);
  p[2] = r[0] + r[2] + step * (r[1] + r[2]);
  p[3] = r[0] + r[2] - r[3] + step * (r[1] + r[2] + r[3]);
}

// haarfeatures.h:114-122 (Feature::calc). `img` is one flattened integral (row stride `step`).
// Offsets are only formed for rects up to the first zero weight (haarfeatures.cpp:292-308): later rects keep
// p0..p3 = 0 and contribute w * (img[0]-img[0]-img[0]+img[0]) = 0.
inline float haar_calc(const OrcHaarFeature& f, const int32_t* img, int step) {
  int p[3][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  for (int j = 0; j < 3; j++) {
    if (f.wt[j] == 0.0f) break;
    if (f.tilted)
      tilted_offsets(p[j], f.r[j], step);
    else
      sum_offsets(p[j], f.r[j], step);
  }
  float ret = f.wt[0] * (img[p[0][0]] - img[p[0][1]] - img[p[0][2]] + img[p[0][3]]) +
              f.wt[1] * (img[p[1][0]] - img[p[1][1]] - img[p[1][2]] + img[p[1][3]]);
  if (f.wt[2] != 0.0f) ret += f.wt[2] * (img[p[2][0]] - img[p[2][1]] - img[p[2][2]] + img[p[2][3]]);
  return ret;
}

// lbpfeatures.h:70-83 + lbpfeatures.cpp:53-63: 16 lattice corners p[4*r+c] = (x + c*w, y + r*h).
inline int lbp_calc(const int32_t rect[4], const int32_t* s, int step) {
  int p[16];
  for (int r = 0; r < 4; r++)
    for (int c = 0; c < 4; c++) p[4 * r + c] = (rect[0] + c * rect[2]) + step * (rect[1] + r * rect[3]);
  int cval = s[p[5]] - s[p[6]] - s[p[9]] + s[p[10]];
  return (s[p[0]] - s[p[1]] - s[p[4]] + s[p[5]] >= cval ? 128 : 0) | (s[p[1]] - s[p[2]] - s[p[5]] + s[p[6]] >= cval ? 64 : 0) |
         (s[p[2]] - s[p[3]] - s[p[6]] + s[p[7]] >= cval ? 32 : 0) | (s[p[6]] - s[p[7]] - s[p[10]] + s[p[11]] >= cval ? 16 : 0) |
         (s[p[10]] - s[p[11]] - s[p[14]] + s[p[15]] >= cval ? 8 : 0) | (s[p[9]] - s[p[10]] - s[p[13]] + s[p[14]] >= cval ? 4 : 0) |
         (s[p[8]] - s[p[9]] - s[p[12]] + s[p[13]] >= cval ? 2 : 0) | (s[p[4]] - s[p[5]] - s[p[8]] + s[p[9]] >= cval ? 1 : 0);
}

}  // namespace

// Feature::calc on an arbitrary flattened integral (the test_features.cpp:462-560 KAT shape).
ORC_API float orc_haar_feature_calc(const OrcHaarFeature* f, const int32_t* integral, int step) {
  return haar_calc(*f, integral, step);
}
ORC_API int orc_lbp_feature_calc(const int32_t* rect, const int32_t* integral, int step) { return lbp_calc(rect, integral, step); }

// ---------------------------------------------------------------------------------------------
// Training-side evaluator state: setImage (haarfeatures.cpp:100-114, lbpfeatures.cpp:22-28) for n samples.
// sum/tilted: n x (W+1)(H+1) int32, one sample per row; normfactor: n floats (Haar only).
// ---------------------------------------------------------------------------------------------
ORC_API void orc_set_images(const uint8_t* imgs, int n, int W, int H, int want_tilted, int32_t* sum, int32_t* tilted,
                            float* normfactor) {
  const int cols = (W + 1) * (H + 1);
  std::vector<double> sq(cols);
  for (int i = 0; i < n; i++) {
    const uint8_t* img = imgs + (size_t)i * W * H;
    int32_t* s = sum + (size_t)i * cols;
    orc_integral_u8(img, W, H, W, s, normfactor ? sq.data() : nullptr, nullptr,
                    (want_tilted && tilted) ? tilted + (size_t)i * cols : nullptr);
    if (normfactor) {
      // features.cpp:13-25: normrect = (1,1,W-2,H-2) on the (W+1)x(H+1) integrals.
      int32_t nr[4] = {1, 1, W - 2, H - 2};
      int p[4];
      sum_offsets(p, nr, W + 1);
      double area = (double)(nr[2] * nr[3]);
      int valSum = s[p[0]] - s[p[1]] - s[p[2]] + s[p[3]];
      double valSqSum = sq[p[0]] - sq[p[1]] - sq[p[2]] + sq[p[3]];
      normfactor[i] = (float)std::sqrt((double)(area * valSqSum - (double)valSum * valSum));
    }
  }
}

// out[(fi - fi0) * ns + s] = evaluator(fi, sample_idx ? sample_idx[s] : s). Haar: haarfeatures.h:108-112.
ORC_API void orc_haar_eval_batch(const OrcHaarFeature* feats, int fi0, int fi1, const int32_t* sum, const int32_t* tilted,
                                 const float* normfactor, int W, int H, const int32_t* sample_idx, int ns, float* out) {
  const int cols = (W + 1) * (H + 1), step = W + 1;
  for (int fi = fi0; fi < fi1; fi++)
    for (int s = 0; s < ns; s++) {
      int si = sample_idx ? sample_idx[s] : s;
      float nf = normfactor[si];
      const OrcHaarFeature& f = feats[fi];
      const int32_t* img = (f.tilted ? tilted : sum) + (size_t)si * cols;
      out[(size_t)(fi - fi0) * ns + s] = !nf ? 0.0f : (haar_calc(f, img, step) / nf);
    }
}

ORC_API void orc_lbp_eval_batch(const int32_t* rects, int fi0, int fi1, const int32_t* sum, int W, int H,
                                const int32_t* sample_idx, int ns, float* out) {
  const int cols = (W + 1) * (H + 1), step = W + 1;
  for (int fi = fi0; fi < fi1; fi++)
    for (int s = 0; s < ns; s++) {
      int si = sample_idx ? sample_idx[s] : s;
      out[(size_t)(fi - fi0) * ns + s] = (float)lbp_calc(rects + 4 * fi, sum + (size_t)si * cols, step);
    }
}

// ---------------------------------------------------------------------------------------------
// cv::resize(INTER_LINEAR_EXACT), 8-bit single channel (OpenCV 4.6.0 imgproc resize.cpp,
// fixed-point "bit-exact" linear path): per axis, source coordinate f = scale*(d+0.5)-0.5 with
// scale = 1/((double)dst/src); taps (i, i+1) with 8.8 fixed-point weights (round-half-even of frac*256);
// clamped to the first / last pixel outside [0, src-1); horizontal pass exact in 16 bits, vertical pass
// exact in 32 bits, result (v + 2^15) >> 16.
// ---------------------------------------------------------------------------------------------
namespace {
struct AxisTab {
  std::vector<int> ofs;        // left tap index
  std::vector<uint16_t> w0, w1;  // 8.8 weights for taps ofs, ofs+1 (tap index clamped on use)
};
void linear_exact_axis(int src, int dst, AxisTab& t) {
  t.ofs.resize(dst);
  t.w0.resize(dst);
  t.w1.resize(dst);
  double inv_scale = (double)dst / (double)src;
  double scale = 1.0 / inv_scale;
  for (int d = 0; d < dst; d++) {
    double f = scale * ((double)d + 0.5) - 0.5;
    int i = (int)std::floor(f);
    if (i >= 0 && src > 1) {
      if (i < src - 1) {
        int c1 = cv_round_d((f - (double)i) * 256.0);
        t.ofs[d] = i;
        t.w1[d] = (uint16_t)c1;
        t.w0[d] = (uint16_t)(256 - c1);
      } else {  // right border: last pixel
        t.ofs[d] = src - 1;
        t.w0[d] = 256;
        t.w1[d] = 0;
      }
    } else {  // left border (or 1-pixel source): first pixel
      t.ofs[d] = 0;
      t.w0[d] = 256;
      t.w1[d] = 0;
    }
  }
}
}  // namespace

ORC_API void orc_resize_linear_exact_u8(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw, int dh,
                                        int dstride) {
  if (sw == dw && sh == dh) {  // cv::resize copies when sizes match
    for (int y = 0; y < dh; y++) std::memcpy(dst + (size_t)y * dstride, src + (size_t)y * sstride, dw);
    return;
  }
  AxisTab tx, ty;
  linear_exact_axis(sw, dw, tx);
  linear_exact_axis(sh, dh, ty);
  for (int y = 0; y < dh; y++) {
    int y0 = ty.ofs[y], y1 = std::min(y0 + 1, sh - 1);
    const uint8_t* r0 = src + (size_t)y0 * sstride;
    const uint8_t* r1 = src + (size_t)y1 * sstride;
    for (int x = 0; x < dw; x++) {
      int x0 = tx.ofs[x], x1 = std::min(x0 + 1, sw - 1);
      uint32_t h0 = (uint32_t)tx.w0[x] * r0[x0] + (uint32_t)tx.w1[x] * r0[x1];  // <= 65280
      uint32_t h1 = (uint32_t)tx.w0[x] * r1[x0] + (uint32_t)tx.w1[x] * r1[x1];
      uint32_t v = h0 * ty.w0[y] + h1 * ty.w1[y];
      dst[(size_t)y * dstride + x] = (uint8_t)((v + (1u << 15)) >> 16);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Detection (cv::CascadeClassifier::detectMultiScale, new-format cascades).
// ---------------------------------------------------------------------------------------------
struct OrcCascade {
  int32_t feature_type;  // 0 HAAR, 1 LBP
  int32_t win_w, win_h;
  int32_t nstages;
  const int32_t* stage_ntrees;     // [nstages]
  const float* stage_threshold;    // [nstages] (float)stageThreshold as parsed; THRESHOLD_EPS applied here
  int32_t nstumps;
  const int32_t* stump_feature;    // [nstumps]
  const float* stump_threshold;    // [nstumps] (Haar)
  const float* stump_left;         // leafValues[0]
  const float* stump_right;        // leafValues[1]
  int32_t subset_size;             // (maxCatCount+31)/32, 0 for Haar
  const int32_t* stump_subset;     // [nstumps*subset_size] (LBP)
  int32_t nfeatures;
  const OrcHaarFeature* haar;      // [nfeatures] (Haar)
  const int32_t* lbp_rect;         // [nfeatures*4] (LBP)
  // General trees (used when max_nodes_per_tree > 1). Weak classifier t has tree_nnodes[t] nodes,
  // stored consecutively; child > 0 = node index inside the tree, child <= 0 = leaf index -child; upstream advances its
  // leaf cursor by nodeCount + 1 per tree.
  int32_t max_nodes_per_tree;
  const int32_t* tree_nnodes;      // [nstumps] (= number of weak classifiers)
  const int32_t* node_left;
  const int32_t* node_right;
  const int32_t* node_feature;
  const float* node_threshold;
  const int32_t* node_subset;      // [nnodes*subset_size]
  const float* leaves;
};

struct OrcScale {
  float scale;
  int32_t w, h;          // resized image size (sz)
  int32_t ystep;
  int32_t nx, ny;        // grid windows along x / y actually enumerated by the scan loops
  int32_t win_w, win_h;  // cvRound(W0*scale), cvRound(H0*scale): size of the emitted rectangle
};

namespace {

void scale_list(int W0, int H0, int imgw, int imgh, double scaleFactor, int minW, int minH, int maxW, int maxH,
                std::vector<float>& scales) {
  scales.clear();
  if (maxW == 0 || maxH == 0) {
    maxW = imgw;
    maxH = imgh;
  }
  if (imgh < H0 || imgw < W0) return;
  std::vector<float> all;
  for (double factor = 1;; factor *= scaleFactor) {
    int ww = cv_round_d(W0 * factor), wh = cv_round_d(H0 * factor);
    if (ww > imgw || wh > imgh) break;
    all.push_back((float)factor);
    if (all.size() > 100000) break;
  }
  for (size_t i = 0; i < all.size(); i++) {
    int ww = cv_round_f(W0 * all[i]), wh = cv_round_f(H0 * all[i]);
    if (ww > maxW || wh > maxH) break;
    if (ww < minW || wh < minH) continue;
    scales.push_back(all[i]);
  }
  if (scales.empty() && !all.empty()) {
    size_t imin = 0;
    double dmin = 0;
    for (size_t v = 0; v < all.size(); v++) {
      int ww = cv_round_f(W0 * all[v]), wh = cv_round_f(H0 * all[v]);
      double d = (double)(minW - ww) * (minW - ww) + (double)(minH - wh) * (minH - wh);
      if (v == 0 || dmin > d) {
        dmin = d;
        imin = v;
      }
    }
    scales.push_back(all[imin]);
  }
}

void scale_data(int W0, int H0, int imgw, int imgh, const std::vector<float>& scales, std::vector<OrcScale>& sd) {
  sd.resize(scales.size());
  int nstripes = 1;
  for (size_t i = 0; i < scales.size(); i++) {
    float sc = scales[i];
    OrcScale& s = sd[i];
    s.scale = sc;
    s.w = cv_round_f(imgw / sc);
    s.h = cv_round_f(imgh / sc);
    s.ystep = sc >= 2 ? 1 : 2;
    int szw_w = std::max(s.w + 1 - W0, 0), szw_h = std::max(s.h + 1 - H0, 0);
    if (i == 0) nstripes = (int)std::ceil(szw_w / 32.);
    // rows are handed out in stripes of stripeSize rows; the union of all stripes is [0, nstripes*stripeSize)
    int stripe = std::max((szw_h / s.ystep + nstripes - 1) / std::max(nstripes, 1), 1) * s.ystep;
    int y_end = std::min(nstripes * stripe, szw_h);
    s.nx = (szw_w + s.ystep - 1) / s.ystep;
    s.ny = (y_end + s.ystep - 1) / s.ystep;
    if (szw_w <= 0 || y_end <= 0) s.nx = s.ny = 0;
    s.win_w = cv_round_f(W0 * sc);
    s.win_h = cv_round_f(H0 * sc);
  }
}

struct ScaleBuffers {
  std::vector<uint8_t> img;
  std::vector<int32_t> sum, sqsum, tilted;
};

// runAt for one window. Returns 1 (all stages passed), -stage (rejected at `stage`; 0 for stage 0) or -1
// (setWindow failed). *last_sum = stage accumulator at exit.
inline int run_at(const OrcCascade& c, const std::vector<float>& stage_thr, const int32_t* sum, const int32_t* sqsum,
                  const int32_t* tilted, int step, int x, int y, double* last_sum) {
  const int W0 = c.win_w, H0 = c.win_h;
  const int32_t* pwin = sum + (size_t)y * step + x;
  const int32_t* ptilt = tilted ? tilted + (size_t)y * step + x : nullptr;
  float vnf = 1.f;
  if (c.feature_type == 0) {
    const int32_t nr[4] = {1, 1, W0 - 2, H0 - 2};
    int n[4];
    sum_offsets(n, nr, step);
    const int32_t* pq = sqsum + (size_t)y * step + x;
    int valsum = pwin[n[0]] - pwin[n[1]] - pwin[n[2]] + pwin[n[3]];
    unsigned valsqsum = (unsigned)pq[n[0]] - (unsigned)pq[n[1]] - (unsigned)pq[n[2]] + (unsigned)pq[n[3]];
    double area = (double)(nr[2] * nr[3]);
    double nf = area * valsqsum - (double)valsum * valsum;
    if (nf > 0.) {
      nf = std::sqrt(nf);
      vnf = (float)(1. / nf);
      if (!(area * vnf < 1e-1)) {
        *last_sum = 0;
        return -1;
      }
    } else {
      *last_sum = 0;
      return -1;
    }
  }
  int si = 0;
  double tmp = 0;
  if (c.max_nodes_per_tree > 1) {  // predictOrdered / predictCategorical: walk each tree from its root
    int nodeOfs = 0, leafOfs = 0;
    for (int st = 0; st < c.nstages; st++) {
      tmp = 0;
      for (int i = 0; i < c.stage_ntrees[st]; i++, si++) {
        int idx = 0;
        const int root = nodeOfs;
        do {
          const int n = root + idx;
          if (c.feature_type == 0) {
            const OrcHaarFeature& f = c.haar[c.node_feature[n]];
            const double val = haar_calc(f, f.tilted ? ptilt : pwin, step) * vnf;
            idx = val < c.node_threshold[n] ? c.node_left[n] : c.node_right[n];
          } else {
            const int code = lbp_calc(c.lbp_rect + 4 * c.node_feature[n], pwin, step);
            const int32_t* subset = c.node_subset + (size_t)n * c.subset_size;
            idx = (subset[code >> 5] & (1 << (code & 31))) ? c.node_left[n] : c.node_right[n];
          }
        } while (idx > 0);
        tmp += c.leaves[leafOfs - idx];
        nodeOfs += c.tree_nnodes[si];
        leafOfs += c.tree_nnodes[si] + 1;
      }
      if (tmp < stage_thr[st]) {
        *last_sum = tmp;
        return -st;
      }
    }
    *last_sum = tmp;
    return 1;
  }
  for (int st = 0; st < c.nstages; st++) {
    tmp = 0;
    int nt = c.stage_ntrees[st];
    for (int i = 0; i < nt; i++, si++) {
      if (c.feature_type == 0) {
        const OrcHaarFeature& f = c.haar[c.stump_feature[si]];
        float v = haar_calc(f, f.tilted ? ptilt : pwin, step) * vnf;
        double value = v;
        tmp += value < c.stump_threshold[si] ? c.stump_left[si] : c.stump_right[si];
      } else {
        int code = lbp_calc(c.lbp_rect + 4 * c.stump_feature[si], pwin, step);
        const int32_t* subset = c.stump_subset + (size_t)si * c.subset_size;
        tmp += (subset[code >> 5] & (1 << (code & 31))) ? c.stump_left[si] : c.stump_right[si];
      }
    }
    if (tmp < stage_thr[st]) {
      *last_sum = tmp;
      return -st;
    }
  }
  *last_sum = tmp;
  return 1;
}

struct Cand {
  int32_t scale_idx, gx, gy, x, y, w, h;
};

}  // namespace

// Scale table only (for tests of the pyramid geometry). Returns number of scales.
ORC_API int orc_scales(int W0, int H0, int imgw, int imgh, double scaleFactor, int minW, int minH, int maxW, int maxH,
                       OrcScale* out, int cap) {
  std::vector<float> scales;
  std::vector<OrcScale> sd;
  scale_list(W0, H0, imgw, imgh, scaleFactor, minW, minH, maxW, maxH, scales);
  scale_data(W0, H0, imgw, imgh, scales, sd);
  for (size_t i = 0; i < sd.size() && (int)i < cap; i++) out[i] = sd[i];
  return (int)sd.size();
}

// Full detection without grouping. Outputs:
//   cand_out[7*k..] = {scale_idx, gx, gy, x, y, w, h} in single-thread OpenCV order (scale, y, x);
//   codes (optional, may be NULL): per grid window result of runAt for every scale, concatenated scale-major then
//     row-major [gy][gx]: 1 pass, -k rejected at stage k (0 = stage 0), -1 setWindow failed. It is computed for
//     every grid window, including those the scan loop never visits because of the stage-0 skip rule;
//   visited (optional, same layout): 1 = the scan loop visits this window;
//   sums (optional, same layout): stage accumulator at exit for each window.
// nthreads > 1 splits work over (scale,row) units; results are identical for any thread count.
// Returns number of candidates (may exceed cand_cap; only cand_cap are written).
ORC_API int orc_detect_raw(const OrcCascade* c, const uint8_t* img, int w, int h, int stride, double scaleFactor, int minW,
                           int minH, int maxW, int maxH, int nthreads, int32_t* cand_out, int cand_cap, int32_t* codes,
                           uint8_t* visited, double* sums, int64_t* n_grid_windows, int64_t* n_visited_windows) {
  std::vector<float> scales;
  std::vector<OrcScale> sd;
  scale_list(c->win_w, c->win_h, w, h, scaleFactor, minW, minH, maxW, maxH, scales);
  scale_data(c->win_w, c->win_h, w, h, scales, sd);
  std::vector<float> stage_thr(c->nstages);
  for (int i = 0; i < c->nstages; i++) stage_thr[i] = c->stage_threshold[i] - 1e-5f;  // THRESHOLD_EPS

  bool has_tilted = false;
  if (c->feature_type == 0)
    for (int i = 0; i < c->nfeatures; i++) has_tilted = has_tilted || c->haar[i].tilted;

  const int ns = (int)sd.size();
  std::vector<ScaleBuffers> bufs(ns);
  std::vector<int64_t> code_ofs(ns + 1, 0);
  for (int i = 0; i < ns; i++) code_ofs[i + 1] = code_ofs[i] + (int64_t)sd[i].nx * sd[i].ny;
  if (n_grid_windows) *n_grid_windows = code_ofs[ns];
  nthreads = std::max(nthreads, 1);

  // phase 1: pyramid + integrals, one scale per work item
  {
    std::atomic<int> next(0);
    auto work = [&]() {
      for (int i; (i = next.fetch_add(1)) < ns;) {
        ScaleBuffers& b = bufs[i];
        const OrcScale& s = sd[i];
        b.img.resize((size_t)s.w * s.h);
        orc_resize_linear_exact_u8(img, w, h, stride, b.img.data(), s.w, s.h, s.w);
        b.sum.resize((size_t)(s.w + 1) * (s.h + 1));
        if (c->feature_type == 0) b.sqsum.resize((size_t)(s.w + 1) * (s.h + 1));
        if (has_tilted) b.tilted.resize((size_t)(s.w + 1) * (s.h + 1));
        orc_integral_u8(b.img.data(), s.w, s.h, s.w, b.sum.data(), nullptr, c->feature_type == 0 ? b.sqsum.data() : nullptr,
                        has_tilted ? b.tilted.data() : nullptr);
      }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nthreads; t++) th.emplace_back(work);
    work();
    for (auto& t : th) t.join();
  }

  // phase 2: scan loops. Work unit = one grid row of one scale (the skip rule is a per-row recurrence).
  struct Unit {
    int si, gy;
  };
  std::vector<Unit> units;
  for (int i = 0; i < ns; i++)
    for (int gy = 0; gy < sd[i].ny; gy++) units.push_back({i, gy});
  std::vector<std::vector<Cand>> unit_cands(units.size());
  std::vector<int64_t> unit_visited(units.size(), 0);
  const bool full = codes || visited || sums;
  {
    std::atomic<size_t> next(0);
    auto work = [&]() {
      for (size_t u; (u = next.fetch_add(1)) < units.size();) {
        const int si = units[u].si, gy = units[u].gy;
        const OrcScale& s = sd[si];
        const ScaleBuffers& b = bufs[si];
        const int step = s.w + 1, y = gy * s.ystep;
        bool skip_next = false;
        for (int gx = 0; gx < s.nx; gx++) {
          const int x = gx * s.ystep;
          const bool vis = !skip_next;
          skip_next = false;
          if (!vis && !full) continue;
          double ls = 0;
          int r = run_at(*c, stage_thr, b.sum.data(), c->feature_type == 0 ? b.sqsum.data() : nullptr,
                         has_tilted ? b.tilted.data() : nullptr, step, x, y, &ls);
          if (full) {
            int64_t o = code_ofs[si] + (int64_t)gy * s.nx + gx;
            if (codes) codes[o] = r;
            if (visited) visited[o] = vis ? 1 : 0;
            if (sums) sums[o] = ls;
          }
          if (!vis) continue;
          unit_visited[u]++;
          if (r > 0) unit_cands[u].push_back({si, gx, gy, cv_round_f(x * s.scale), cv_round_f(y * s.scale), s.win_w, s.win_h});
          if (r == 0) skip_next = true;  // rejected at stage 0: the scan loop does x += ystep once more
        }
      }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nthreads; t++) th.emplace_back(work);
    work();
    for (auto& t : th) t.join();
  }
  int n = 0;
  int64_t nv = 0;
  for (size_t u = 0; u < units.size(); u++) {
    nv += unit_visited[u];
    for (const Cand& k : unit_cands[u]) {
      if (n < cand_cap && cand_out) std::memcpy(cand_out + 7 * (size_t)n, &k, sizeof(Cand));
      n++;
    }
  }
  if (n_visited_windows) *n_visited_windows = nv;
  return n;
}

// cv::groupRectangles(rectList, groupThreshold, eps). rects: n x {x,y,w,h}. Output written to
// out (cap rects), returns count. Output order = class-label order of cv::partition for the given input order.
// levels / level_weights (both NULL or both n long): cv::groupRectangles(rectList, rejectLevels, levelWeights,
// groupThreshold, eps), the grouping of detectMultiScale's outputRejectLevels overload: per class the highest level and,
// among the members at that level, the largest weight; out_levels / out_weights receive them per kept rectangle.
static int group_rectangles_impl(const int32_t* rects, int n, int groupThreshold, double eps, int32_t* out, int cap,
                                 const int32_t* levels, const double* level_weights, int32_t* out_levels, double* out_weights) {
  if (groupThreshold <= 0 || n == 0) {
    for (int i = 0; i < n && i < cap; i++) {
      std::memcpy(out + 4 * i, rects + 4 * i, 16);
      if (levels) {
        out_levels[i] = levels[i];
        out_weights[i] = level_weights[i];
      }
    }
    return n;
  }
  auto similar = [&](const int32_t* a, const int32_t* b) {
    double delta = eps * (std::min(a[2], b[2]) + std::min(a[3], b[3])) * 0.5;
    return std::abs(a[0] - b[0]) <= delta && std::abs(a[1] - b[1]) <= delta &&
           std::abs(a[0] + a[2] - b[0] - b[2]) <= delta && std::abs(a[1] + a[3] - b[1] - b[3]) <= delta;
  };
  // cv::partition: connected components of the `similar` graph; labels numbered by first appearance.
  std::vector<int> parent(n);
  for (int i = 0; i < n; i++) parent[i] = i;
  auto find = [&](int a) {
    while (parent[a] != a) a = parent[a] = parent[parent[a]];
    return a;
  };
  for (int i = 0; i < n; i++)
    for (int j = i + 1; j < n; j++)
      if (similar(rects + 4 * i, rects + 4 * j)) {
        int a = find(i), b = find(j);
        if (a != b) parent[b] = a;
      }
  std::vector<int> label(n), root_label(n, -1);
  int nclasses = 0;
  for (int i = 0; i < n; i++) {
    int r = find(i);
    if (root_label[r] < 0) root_label[r] = nclasses++;
    label[i] = root_label[r];
  }
  std::vector<int32_t> rr((size_t)nclasses * 4, 0);
  std::vector<int> rw(nclasses, 0);
  for (int i = 0; i < n; i++) {
    int cl = label[i];
    for (int k = 0; k < 4; k++) rr[4 * cl + k] += rects[4 * i + k];
    rw[cl]++;
  }
  for (int i = 0; i < nclasses; i++) {
    float s = 1.f / rw[i];
    for (int k = 0; k < 4; k++) rr[4 * i + k] = cv_round_f(rr[4 * i + k] * s);
  }
  std::vector<int> rejectLevels(nclasses, 0);
  std::vector<double> rejectWeights(nclasses, 2.2250738585072014e-308 /* DBL_MIN */);
  if (levels)
    for (int i = 0; i < n; i++) {
      int cls = label[i];
      if (levels[i] > rejectLevels[cls]) {
        rejectLevels[cls] = levels[i];
        rejectWeights[cls] = level_weights[i];
      } else if (levels[i] == rejectLevels[cls] && level_weights[i] > rejectWeights[cls])
        rejectWeights[cls] = level_weights[i];
    }
  int m = 0;
  for (int i = 0; i < nclasses; i++) {
    const int32_t* r1 = &rr[4 * i];
    int n1 = rw[i];
    if (n1 <= groupThreshold) continue;
    int j;
    for (j = 0; j < nclasses; j++) {
      int n2 = rw[j];
      if (j == i || n2 <= groupThreshold) continue;
      const int32_t* r2 = &rr[4 * j];
      int dx = cv_round_d(r2[2] * eps), dy = cv_round_d(r2[3] * eps);
      if (r1[0] >= r2[0] - dx && r1[1] >= r2[1] - dy && r1[0] + r1[2] <= r2[0] + r2[2] + dx &&
          r1[1] + r1[3] <= r2[1] + r2[3] + dy && (n2 > std::max(3, n1) || n1 < 3))
        break;
    }
    if (j == nclasses) {
      if (m < cap) {
        std::memcpy(out + 4 * m, r1, 16);
        if (levels) {
          out_levels[m] = rejectLevels[i];
          out_weights[m] = rejectWeights[i];
        }
      }
      m++;
    }
  }
  return m;
}

ORC_API int orc_group_rectangles(const int32_t* rects, int n, int groupThreshold, double eps, int32_t* out, int cap) {
  return group_rectangles_impl(rects, n, groupThreshold, eps, out, cap, nullptr, nullptr, nullptr, nullptr);
}

ORC_API int orc_group_rectangles_levels(const int32_t* rects, const int32_t* levels, const double* level_weights, int n,
                                        int groupThreshold, double eps, int32_t* out, int32_t* out_levels, double* out_weights,
                                        int cap) {
  return group_rectangles_impl(rects, n, groupThreshold, eps, out, cap, levels, level_weights, out_levels, out_weights);
}

// detectMultiScale = raw candidates + groupRectangles(minNeighbors, eps = 0.2).
ORC_API int orc_detect_multiscale(const OrcCascade* c, const uint8_t* img, int w, int h, int stride, double scaleFactor,
                                  int minNeighbors, int minW, int minH, int maxW, int maxH, int nthreads, int32_t* out,
                                  int cap) {
  std::vector<int32_t> cand(7 * 1024);
  int n = orc_detect_raw(c, img, w, h, stride, scaleFactor, minW, minH, maxW, maxH, nthreads, cand.data(), 1024, nullptr,
                         nullptr, nullptr, nullptr, nullptr);
  if (n > 1024) {
    cand.resize((size_t)7 * n);
    n = orc_detect_raw(c, img, w, h, stride, scaleFactor, minW, minH, maxW, maxH, nthreads, cand.data(), n, nullptr,
                       nullptr, nullptr, nullptr, nullptr);
  }
  std::vector<int32_t> rects((size_t)4 * std::max(n, 1));
  for (int i = 0; i < n; i++) std::memcpy(&rects[4 * i], &cand[7 * i + 3], 16);
  return orc_group_rectangles(rects.data(), n, minNeighbors, 0.2, out, cap);
}

// Training-side stage evaluation for one sample (boost.cpp:461-477 + o_cvcascadeboosttree.cpp:16-39), stumps only:
// ordered split goes LEFT when value <= threshold (note: detector uses `<`), categorical goes left when the
// subset bit is set; stage passes iff sum >= threshold - CV_THRESHOLD_EPS(1e-5f).
ORC_API int orc_train_predict(const OrcCascade* c, const int32_t* sum, const int32_t* tilted, const float* normfactor,
                              int si, int W, int H) {
  const int cols = (W + 1) * (H + 1), step = W + 1;
  const int32_t* img = sum + (size_t)si * cols;
  const int32_t* timg = tilted ? tilted + (size_t)si * cols : nullptr;
  int k = 0;
  if (c->max_nodes_per_tree > 1) {  // CvCascadeBoostTree::predict: walk while the node has children
    int nodeOfs = 0, leafOfs = 0;
    for (int st = 0; st < c->nstages; st++) {
      double acc = 0;
      for (int i = 0; i < c->stage_ntrees[st]; i++, k++) {
        int idx = 0;
        do {
          const int n = nodeOfs + idx;
          if (c->feature_type == 0) {
            float nf = normfactor[si];
            const OrcHaarFeature& f = c->haar[c->node_feature[n]];
            float val = !nf ? 0.0f : haar_calc(f, f.tilted ? timg : img, step) / nf;
            idx = val <= c->node_threshold[n] ? c->node_left[n] : c->node_right[n];
          } else {
            int code = lbp_calc(c->lbp_rect + 4 * c->node_feature[n], img, step);
            const int32_t* subset = c->node_subset + (size_t)n * c->subset_size;
            idx = (subset[code >> 5] & (1 << (code & 31))) ? c->node_left[n] : c->node_right[n];
          }
        } while (idx > 0);
        acc += c->leaves[leafOfs - idx];
        nodeOfs += c->tree_nnodes[k];
        leafOfs += c->tree_nnodes[k] + 1;
      }
      if (acc < c->stage_threshold[st] - 1e-5f) return 0;
    }
    return 1;
  }
  for (int st = 0; st < c->nstages; st++) {
    double acc = 0;
    for (int i = 0; i < c->stage_ntrees[st]; i++, k++) {
      if (c->feature_type == 0) {
        float nf = normfactor[si];
        const OrcHaarFeature& f = c->haar[c->stump_feature[k]];
        float val = !nf ? 0.0f : haar_calc(f, f.tilted ? timg : img, step) / nf;
        acc += val <= c->stump_threshold[k] ? c->stump_left[k] : c->stump_right[k];
      } else {
        int code = lbp_calc(c->lbp_rect + 4 * c->stump_feature[k], img, step);
        const int32_t* subset = c->stump_subset + (size_t)k * c->subset_size;
        acc += (subset[code >> 5] & (1 << (code & 31))) ? c->stump_left[k] : c->stump_right[k];
      }
    }
    if (acc < c->stage_threshold[st] - 1e-5f) return 0;
  }
  return 1;
}

// Negative mining over ONE background image, as the reference does it window by window: NegReader::nextImg has chosen
// the image and the offset (imagestorage.cpp:57-88); then NegReader::get (imagestorage.cpp:90-126) hands out one window
// per call and CvCascadeClassifier::fillPassedSamples (cascadeclassifier.cpp:329-357) runs setImage + predict on it.
// pass[i] = predict result of stream window i; the first max_keep passing windows' pixels / stream indices are returned.
// Returns the stream length for this image (windows until the reader would move on to the next image).
ORC_API int64_t orc_negmine_image(const OrcCascade* c, const uint8_t* src, int cols, int rows, int stride, int ox, int oy,
                                  uint8_t* pass, int64_t cap, uint8_t* pixels, int64_t* keep_index, int max_keep, int* n_keep) {
  const int W = c->win_w, H = c->win_h;
  const float scaleFactor = 1.4142135623730950488016887242097F, stepFactor = 0.5F;
  bool has_tilted = false;
  if (c->feature_type == 0)
    for (int i = 0; i < c->nfeatures; i++) has_tilted = has_tilted || c->haar[i].tilted;
  // nextImg(): point = offset; scale; first resize
  int px = ox, py = oy;
  float scale = std::max(((float)W + px) / ((float)cols), ((float)H + py) / ((float)rows));
  int iw = (int)(scale * cols + 0.5F), ih = (int)(scale * rows + 0.5F);
  std::vector<uint8_t> img((size_t)iw * ih);
  orc_resize_linear_exact_u8(src, cols, rows, stride, img.data(), iw, ih, iw);
  std::vector<uint8_t> win((size_t)W * H);
  const int ncols = (W + 1) * (H + 1);
  std::vector<int32_t> sum(ncols), til(ncols);
  float nf = 0;
  int64_t n = 0;
  int kept = 0;
  for (;;) {
    // get(): copy the window at `point`
    for (int y = 0; y < H; y++) std::memcpy(&win[(size_t)y * W], &img[(size_t)(py + y) * iw + px], W);
    // setImage + predict (haarfeatures.cpp:100-114 / lbpfeatures.cpp:22-28; cascadeclassifier.cpp:297-306)
    orc_set_images(win.data(), 1, W, H, has_tilted ? 1 : 0, sum.data(), til.data(), c->feature_type == 0 ? &nf : nullptr);
    const int ok = orc_train_predict(c, sum.data(), has_tilted ? til.data() : nullptr, &nf, 0, W, H);
    if (n < cap && pass) pass[n] = (uint8_t)ok;
    if (ok && kept < max_keep && pixels) {
      std::memcpy(pixels + (size_t)kept * W * H, win.data(), (size_t)W * H);
      keep_index[kept] = n;
      kept++;
    }
    n++;
    // advance (imagestorage.cpp:105-124)
    if ((int)(px + (1.0F + stepFactor) * W) < iw)
      px += (int)(stepFactor * W);
    else {
      px = ox;
      if ((int)(py + (1.0F + stepFactor) * H) < ih)
        py += (int)(stepFactor * H);
      else {
        py = oy;
        scale *= scaleFactor;
        if (scale <= 1.0F) {
          iw = (int)(scale * cols);
          ih = (int)(scale * rows);
          img.assign((size_t)iw * ih, 0);
          orc_resize_linear_exact_u8(src, cols, rows, stride, img.data(), iw, ih, iw);
        } else
          break;  // the reader would call nextImg() here
      }
    }
  }
  if (n_keep) *n_keep = kept;
  return n;
}

// ---------------------------------------------------------------------------------------------
// Best-split search of one tree node: CvDTree::find_best_split (o_cvdtree.cpp:313-357, whose
// parallel_reduce is the serial stand-in of o_blockedrange.h:41-47, i.e. ONE range over all variables) calling
//   CvBoostTree::find_split_ord_class  o_cvboostree.cpp:151-247   (DISCRETE / REAL boost, Haar)
//   CvBoostTree::find_split_cat_class  o_cvboostree.cpp:249-359   (DISCRETE / REAL boost, LBP)
//   CvBoostTree::find_split_ord_reg    o_cvboostree.cpp:361-426   (LOGIT / GENTLE boost, Haar)
//   CvBoostTree::find_split_cat_reg    o_cvboostree.cpp:428-516   (LOGIT / GENTLE boost, LBP)
// on the variable data CvCascadeBoostTrainData::get_ord_var_data / get_cat_var_data deliver for a node
// (o_cvcascadeboosttraindata.cpp:403-482; no missing values in cascade training, so n1 == n).
//
// vals: [F][n] feature values of the node's samples in node order (the evaluator's operator(), computed by
// orc_*_eval_batch). The reference sorts each ordered row with std::sort + LessThanIdx, which leaves the order of equal
// values unspecified; this restatement (and the product) take equal values in increasing `tie_key` order (the stored
// sample index), one of the orders std::sort may produce. Categories are sorted exactly as the reference does
// (std::sort on pointers with LessThanPtr). weights: n + 2 doubles, the "subtree weights" of calc_node_value
// (o_cvboostree.cpp:657-732): w[i] per node sample, w[n] and w[n+1] the totals.
// boost_type: 0 DISCRETE, 1 REAL, 2 LOGIT, 3 GENTLE; split_criteria: 0 DEFAULT, 1 GINI, 3 MISCLASS, 4 SQERR (boost.h).
// ---------------------------------------------------------------------------------------------
struct OrcSplit {
  int32_t found, var_idx;
  float quality, ord_c;
  int32_t split_point;
  int32_t subset[8];
};

namespace {

struct LessIdxTie {
  const float* v;
  const int32_t* key;
  bool operator()(int a, int b) const { return v[a] < v[b] || (!(v[b] < v[a]) && key[a] < key[b]); }
};
template <typename T>
struct LessThanPtrT {
  bool operator()(T* a, T* b) const { return *a < *b; }
};

// one variable; returns false when no split beats init_quality
bool split_ord_reg(const float* values, const int* indices, int n, const double* weights, const float* responses,
                   double node_value, float init_quality, OrcSplit& sp) {
  const float epsilon = 1.1920929e-07f * 2;  // FLT_EPSILON * 2
  int best_i = -1;
  double L = 0, R = weights[n];
  double best_val = init_quality, lsum = 0, rsum = node_value * R;
  for (int i = 0; i < n - 1; i++) {
    int idx = indices[i];
    double w = weights[idx];
    double t = responses[idx] * w;
    L += w;
    R -= w;
    lsum += t;
    rsum -= t;
    if (values[i] + epsilon < values[i + 1]) {
      double val = (lsum * lsum * R + rsum * rsum * L) / (L * R);
      if (best_val < val) {
        best_val = val;
        best_i = i;
      }
    }
  }
  if (best_i < 0) return false;
  sp.ord_c = (values[best_i] + values[best_i + 1]) * 0.5f;
  sp.split_point = best_i;
  sp.quality = (float)best_val;
  return true;
}

bool split_ord_class(const float* values, const int* indices, int n, const double* weights, const int32_t* responses,
                     int criteria, float init_quality, OrcSplit& sp) {
  const float epsilon = 1.1920929e-07f * 2;
  const double* rcw0 = weights + n;
  double lcw[2] = {0, 0}, rcw[2] = {rcw0[0], rcw0[1]};
  int best_i = -1;
  double best_val = init_quality;
  if (criteria == 1) {  // GINI
    double L = 0, R = rcw[0] + rcw[1];
    double lsum2 = 0, rsum2 = rcw[0] * rcw[0] + rcw[1] * rcw[1];
    for (int i = 0; i < n - 1; i++) {
      int idx = indices[i];
      double w = weights[idx], w2 = w * w;
      idx = responses[idx];
      L += w;
      R -= w;
      double lv = lcw[idx], rv = rcw[idx];
      lsum2 += 2 * lv * w + w2;
      rsum2 -= 2 * rv * w - w2;
      lcw[idx] = lv + w;
      rcw[idx] = rv - w;
      if (values[i] + epsilon < values[i + 1]) {
        double val = (lsum2 * R + rsum2 * L) / (L * R);
        if (best_val < val) {
          best_val = val;
          best_i = i;
        }
      }
    }
  } else {  // MISCLASS
    for (int i = 0; i < n - 1; i++) {
      int idx = indices[i];
      double w = weights[idx];
      idx = responses[idx];
      lcw[idx] += w;
      rcw[idx] -= w;
      if (values[i] + epsilon < values[i + 1]) {
        double val = lcw[0] + rcw[1], val2 = lcw[1] + rcw[0];
        val = val > val2 ? val : val2;
        if (best_val < val) {
          best_val = val;
          best_i = i;
        }
      }
    }
  }
  if (best_i < 0) return false;
  sp.ord_c = (values[best_i] + values[best_i + 1]) * 0.5f;
  sp.split_point = best_i;
  sp.quality = (float)best_val;
  return true;
}

bool split_cat_reg(const int* cat_labels, int n, int mi, const double* weights, const float* responses,
                   float init_quality, OrcSplit& sp) {
  std::vector<double> sum_buf(mi + 1, 0.0), cnt_buf(mi + 1, 0.0);
  double* sum = sum_buf.data() + 1;
  double* counts = cnt_buf.data() + 1;
  std::vector<double*> sum_ptr(mi);
  double L = 0, R = 0, best_val = init_quality, lsum = 0, rsum = 0;
  int best_subset = -1;
  for (int i = 0; i < n; i++) {
    int idx = cat_labels[i];
    double w = weights[i];
    double s = sum[idx] + responses[i] * w;
    double nc = counts[idx] + w;
    sum[idx] = s;
    counts[idx] = nc;
  }
  for (int i = 0; i < mi; i++) {
    R += counts[i];
    rsum += sum[i];
    sum[i] = std::fabs(counts[i]) > 2.2204460492503131e-16 ? sum[i] / counts[i] : 0;
    sum_ptr[i] = sum + i;
  }
  std::sort(sum_ptr.begin(), sum_ptr.end(), LessThanPtrT<double>());
  for (int i = 0; i < mi; i++) sum[i] *= counts[i];
  for (int subset_i = 0; subset_i < mi - 1; subset_i++) {
    int idx = (int)(sum_ptr[subset_i] - sum);
    double ni = counts[idx];
    if (ni > 1.1920929e-07f) {
      double s = sum[idx];
      lsum += s;
      L += ni;
      rsum -= s;
      R -= ni;
      if (L > 1.1920929e-07f && R > 1.1920929e-07f) {
        double val = (lsum * lsum * R + rsum * rsum * L) / (L * R);
        if (best_val < val) {
          best_val = val;
          best_subset = subset_i;
        }
      }
    }
  }
  if (best_subset < 0) return false;
  sp.quality = (float)best_val;
  std::memset(sp.subset, 0, sizeof(sp.subset));
  for (int i = 0; i <= best_subset; i++) {
    int idx = (int)(sum_ptr[i] - sum);
    sp.subset[idx >> 5] |= 1 << (idx & 31);
  }
  return true;
}

bool split_cat_class(const int* cat_labels, int n, int mi, const double* weights, const int32_t* responses, int criteria,
                     float init_quality, OrcSplit& sp) {
  std::vector<double> cjk_buf(2 * mi + 2, 0.0);
  double* cjk = cjk_buf.data() + 2;
  std::vector<double*> dbl_ptr(mi);
  double lcw[2] = {0, 0}, rcw[2] = {0, 0};
  double L = 0, R;
  double best_val = init_quality;
  int best_subset = -1;
  for (int i = 0; i < n; i++) {
    double w = weights[i];
    int j = cat_labels[i];
    int k = responses[i];
    cjk[j * 2 + k] += w;
  }
  for (int j = 0; j < mi; j++) {
    rcw[0] += cjk[j * 2];
    rcw[1] += cjk[j * 2 + 1];
    dbl_ptr[j] = cjk + j * 2 + 1;
  }
  R = rcw[0] + rcw[1];
  std::sort(dbl_ptr.begin(), dbl_ptr.end(), LessThanPtrT<double>());
  for (int subset_i = 0; subset_i < mi - 1; subset_i++) {
    int idx = (int)(dbl_ptr[subset_i] - cjk) / 2;
    const double* crow = cjk + idx * 2;
    double w0 = crow[0], w1 = crow[1];
    double weight = w0 + w1;
    if (weight < 1.1920929e-07f) continue;
    lcw[0] += w0;
    rcw[0] -= w0;
    lcw[1] += w1;
    rcw[1] -= w1;
    if (criteria == 1) {
      double lsum2 = lcw[0] * lcw[0] + lcw[1] * lcw[1];
      double rsum2 = rcw[0] * rcw[0] + rcw[1] * rcw[1];
      L += weight;
      R -= weight;
      if (L > 1.1920929e-07f && R > 1.1920929e-07f) {
        double val = (lsum2 * R + rsum2 * L) / (L * R);
        if (best_val < val) {
          best_val = val;
          best_subset = subset_i;
        }
      }
    } else {
      double val = lcw[0] + rcw[1];
      double val2 = lcw[1] + rcw[0];
      val = val > val2 ? val : val2;
      if (best_val < val) {
        best_val = val;
        best_subset = subset_i;
      }
    }
  }
  if (best_subset < 0) return false;
  sp.quality = (float)best_val;
  std::memset(sp.subset, 0, sizeof(sp.subset));
  for (int i = 0; i <= best_subset; i++) {
    int idx = (int)(dbl_ptr[i] - cjk) >> 1;
    sp.subset[idx >> 5] |= 1 << (idx & 31);
  }
  return true;
}

}  // namespace

// per_feature_quality (nullable, F floats): the quality find_split_* reports for variable vi when called with
// init_quality = -1 on its own (NaN-free marker -1 when it finds none); per_feature_point likewise (split_point, or the
// number of categories sent left - 1).
ORC_API void orc_find_best_split(const float* vals, int F, int n, int categorical, int mi, const int32_t* tie_key,
                                 const double* weights, const float* responses, const int32_t* class_labels,
                                 double node_value, int boost_type, int split_criteria, OrcSplit* out,
                                 float* per_feature_quality, int32_t* per_feature_point) {
  const bool is_classifier = boost_type == 0 || boost_type == 1;
  int criteria = split_criteria;
  if (criteria != 1 && criteria != 3) criteria = boost_type == 0 ? 3 : 1;  // o_cvboostree.cpp:188-190
  OrcSplit best;
  std::memset(&best, 0, sizeof(best));
  best.quality = -1;  // DTreeBestSplitFinder ctor, o_cvdtree.cpp:291
  std::vector<int> order(n), labels(n);
  std::vector<float> sorted(n);
  for (int vi = 0; vi < F; vi++) {
    const float* row = vals + (size_t)vi * n;
    if (n <= 1) continue;  // get_num_valid(vi) <= 1
    OrcSplit sp;
    std::memset(&sp, 0, sizeof(sp));
    OrcSplit alone;
    std::memset(&alone, 0, sizeof(alone));
    bool res, res_alone = false;
    if (!categorical) {
      for (int i = 0; i < n; i++) order[i] = i;
      std::sort(order.begin(), order.end(), LessIdxTie{row, tie_key});
      for (int i = 0; i < n; i++) sorted[i] = row[order[i]];
      if (is_classifier) {
        res = split_ord_class(sorted.data(), order.data(), n, weights, class_labels, criteria, best.quality, sp);
        if (per_feature_quality) res_alone = split_ord_class(sorted.data(), order.data(), n, weights, class_labels, criteria, -1.f, alone);
      } else {
        res = split_ord_reg(sorted.data(), order.data(), n, weights, responses, node_value, best.quality, sp);
        if (per_feature_quality) res_alone = split_ord_reg(sorted.data(), order.data(), n, weights, responses, node_value, -1.f, alone);
      }
    } else {
      for (int i = 0; i < n; i++) labels[i] = (int)row[i];  // get_cat_var_data, o_cvcascadeboosttraindata.cpp:464-482
      if (is_classifier) {
        res = split_cat_class(labels.data(), n, mi, weights, class_labels, criteria, best.quality, sp);
        if (per_feature_quality) res_alone = split_cat_class(labels.data(), n, mi, weights, class_labels, criteria, -1.f, alone);
      } else {
        res = split_cat_reg(labels.data(), n, mi, weights, responses, best.quality, sp);
        if (per_feature_quality) res_alone = split_cat_reg(labels.data(), n, mi, weights, responses, -1.f, alone);
      }
      if (res_alone) {
        int cnt = 0;
        for (int k = 0; k < 8; k++) cnt += __builtin_popcount((unsigned)alone.subset[k]);
        alone.split_point = cnt - 1;
      }
    }
    if (per_feature_quality) {
      per_feature_quality[vi] = res_alone ? alone.quality : -1.f;
      per_feature_point[vi] = res_alone ? alone.split_point : -1;
    }
    if (res && best.quality < sp.quality) {  // o_cvdtree.cpp:340-341
      sp.var_idx = vi;
      sp.found = 1;
      best = sp;
    }
  }
  if (!(best.quality > 0)) {  // o_cvdtree.cpp:351
    std::memset(&best, 0, sizeof(best));
    best.quality = -1;
  }
  *out = best;
}

ORC_API int orc_version() { return 1; }
