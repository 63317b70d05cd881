// Host-side pieces of the hot path that are tiny and serial in the reference too: error plumbing, the scale
// pyramid geometry, the tap tables of the fixed-point bilinear resize, rectangle grouping and the feature catalogs.
// Everything that scales with pixels / windows / samples runs in the HIP kernels (cc_detect.hip, cc_eval.hip).
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <numeric>

#include "cc_internal.h"

namespace ccamd {

static thread_local std::string g_last_error;

cc_status set_error(cc_status code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_last_error = buf;
  return code;
}

// cvRound: round half to even, as the SSE conversion does under the default rounding mode.
static inline int rnd_f(float v) { return (int)std::nearbyintf(v); }
static inline int rnd_d(double v) { return (int)std::nearbyint(v); }

// Scale list and per-scale geometry of detectMultiScale (OpenCV 4.6.0 cascadedetect.cpp
// detectMultiScaleNoGrouping + FeatureEvaluator::updateScaleData + the stripe split of the scan loop).
void scale_plan(int W0, int H0, int imgw, int imgh, const cc_detect_params& p, std::vector<ScaleGeom>& out) {
  out.clear();
  int maxW = p.max_w, maxH = p.max_h;
  if (maxW == 0 || maxH == 0) {
    maxW = imgw;
    maxH = imgh;
  }
  if (imgh < H0 || imgw < W0) return;
  std::vector<float> all, scales;
  for (double factor = 1;; factor *= p.scale_factor) {
    const int ww = rnd_d(W0 * factor), wh = rnd_d(H0 * factor);
    if (ww > imgw || wh > imgh) break;
    all.push_back((float)factor);
    if (all.size() > 100000) break;
  }
  for (float s : all) {
    const int ww = rnd_f(W0 * s), wh = rnd_f(H0 * s);
    if (ww > maxW || wh > maxH) break;
    if (ww < p.min_w || wh < p.min_h) continue;
    scales.push_back(s);
  }
  if (scales.empty() && !all.empty()) {  // minSize == maxSize off the grid: closest scale
    size_t best = 0;
    double dbest = 0;
    for (size_t v = 0; v < all.size(); v++) {
      const int ww = rnd_f(W0 * all[v]), wh = rnd_f(H0 * all[v]);
      const double d = (double)(p.min_w - ww) * (p.min_w - ww) + (double)(p.min_h - wh) * (p.min_h - wh);
      if (v == 0 || dbest > d) {
        dbest = d;
        best = v;
      }
    }
    scales.push_back(all[best]);
  }
  int nstripes = 1;
  for (size_t i = 0; i < scales.size(); i++) {
    ScaleGeom g;
    g.scale = scales[i];
    g.w = rnd_f(imgw / g.scale);
    g.h = rnd_f(imgh / g.scale);
    g.ystep = g.scale >= 2 ? 1 : 2;
    const int szw_w = std::max(g.w + 1 - W0, 0), szw_h = std::max(g.h + 1 - H0, 0);
    if (i == 0) nstripes = std::max((int)std::ceil(szw_w / 32.), 1);
    const int stripe = std::max((szw_h / g.ystep + nstripes - 1) / nstripes, 1) * g.ystep;
    const int y_end = std::min(nstripes * stripe, szw_h);
    g.nx = (szw_w + g.ystep - 1) / g.ystep;
    g.ny = (y_end + g.ystep - 1) / g.ystep;
    if (szw_w <= 0 || y_end <= 0) g.nx = g.ny = 0;
    g.win_w = rnd_f(W0 * g.scale);
    g.win_h = rnd_f(H0 * g.scale);
    out.push_back(g);
  }
}

// INTER_LINEAR_EXACT tap table of one axis: f = scale*(d+0.5)-0.5, scale = 1/((double)dst/src).
void linear_exact_taps(int src, int dst, AxisTaps& t) {
  t.ofs.resize(dst);
  t.w1.resize(dst);
  const double inv_scale = (double)dst / (double)src;
  const double scale = 1.0 / inv_scale;
  for (int d = 0; d < dst; d++) {
    const double f = scale * ((double)d + 0.5) - 0.5;
    const int i = (int)std::floor(f);
    if (i >= 0 && src > 1 && i < src - 1) {
      t.ofs[d] = i;
      t.w1[d] = (uint16_t)rnd_d((f - (double)i) * 256.0);
    } else {  // outside [0, src-1): replicate the border pixel
      t.ofs[d] = (i >= 0 && src > 1) ? src - 1 : 0;
      t.w1[d] = 0;
    }
  }
}

// cv::groupRectangles. Classes are the connected components of the SimilarRects graph, labelled in
// order of first appearance, which is what cv::partition yields.
void group_rectangles(std::vector<cc_rect>& rects, int group_threshold, double eps, std::vector<int>* levels,
                      std::vector<double>* level_weights) {
  const int n = (int)rects.size();
  if (group_threshold <= 0 || n == 0) return;  // (with both vectors given, upstream leaves them untouched here as well)
  std::vector<int> parent(n), rank_(n, 0);
  std::iota(parent.begin(), parent.end(), 0);
  auto find = [&](int a) {
    int r = a;
    while (parent[r] != r) r = parent[r];
    while (parent[a] != r) {
      int nx = parent[a];
      parent[a] = r;
      a = nx;
    }
    return r;
  };
  // SimilarRects needs |a.x - b.x| <= delta <= eps * (a.w + a.h) / 2, so after sorting by x only a short run of
  // followers can be similar to a rectangle: a sweep instead of all n^2 pairs (same components, same labels).
  std::vector<int> order(n);
  std::iota(order.begin(), order.end(), 0);
  std::sort(order.begin(), order.end(), [&](int a, int b) { return rects[a].x < rects[b].x; });
  for (int oi = 0; oi < n; oi++) {
    const int i = order[oi];
    const cc_rect& a = rects[i];
    const double reach = eps * (a.width + a.height) * 0.5;
    for (int oj = oi + 1; oj < n; oj++) {
      const int j = order[oj];
      const cc_rect& b = rects[j];
      if (b.x - a.x > reach) break;
      const double delta = eps * (std::min(a.width, b.width) + std::min(a.height, b.height)) * 0.5;
      if (std::abs(a.x - b.x) <= delta && std::abs(a.y - b.y) <= delta &&
          std::abs(a.x + a.width - b.x - b.width) <= delta && std::abs(a.y + a.height - b.y - b.height) <= delta) {
        int ra = find(i), rb = find(j);
        if (ra == rb) continue;
        if (rank_[ra] < rank_[rb]) std::swap(ra, rb);
        parent[rb] = ra;
        if (rank_[ra] == rank_[rb]) rank_[ra]++;
      }
    }
  }
  std::vector<int> cls_of_root(n, -1), label(n);
  int nclasses = 0;
  for (int i = 0; i < n; i++) {
    const int r = find(i);
    if (cls_of_root[r] < 0) cls_of_root[r] = nclasses++;
    label[i] = cls_of_root[r];
  }
  std::vector<cc_rect> acc(nclasses, cc_rect{0, 0, 0, 0});
  std::vector<int> cnt(nclasses, 0);
  for (int i = 0; i < n; i++) {
    cc_rect& a = acc[label[i]];
    a.x += rects[i].x;
    a.y += rects[i].y;
    a.width += rects[i].width;
    a.height += rects[i].height;
    cnt[label[i]]++;
  }
  for (int i = 0; i < nclasses; i++) {
    const float s = 1.f / cnt[i];
    acc[i] = cc_rect{rnd_f(acc[i].x * s), rnd_f(acc[i].y * s), rnd_f(acc[i].width * s), rnd_f(acc[i].height * s)};
  }
  // outputRejectLevels variant: per class the highest level among its members and, among those, the largest weight
  std::vector<int> cls_level(nclasses, 0), out_levels;
  std::vector<double> cls_weight(nclasses, DBL_MIN), out_weights;
  const bool with_levels = levels && level_weights && !levels->empty() && !level_weights->empty();
  if (with_levels)
    for (int i = 0; i < n; i++) {
      const int c = label[i];
      if ((*levels)[(size_t)i] > cls_level[c]) {
        cls_level[c] = (*levels)[(size_t)i];
        cls_weight[c] = (*level_weights)[(size_t)i];
      } else if ((*levels)[(size_t)i] == cls_level[c] && (*level_weights)[(size_t)i] > cls_weight[c])
        cls_weight[c] = (*level_weights)[(size_t)i];
    }
  std::vector<cc_rect> out;
  for (int i = 0; i < nclasses; i++) {
    const cc_rect r1 = acc[i];
    const int n1 = cnt[i];
    if (n1 <= group_threshold) continue;
    int j;
    for (j = 0; j < nclasses; j++) {
      const int n2 = cnt[j];
      if (j == i || n2 <= group_threshold) continue;
      const cc_rect r2 = acc[j];
      const int dx = rnd_d(r2.width * eps), dy = rnd_d(r2.height * eps);
      if (r1.x >= r2.x - dx && r1.y >= r2.y - dy && r1.x + r1.width <= r2.x + r2.width + dx &&
          r1.y + r1.height <= r2.y + r2.height + dy && (n2 > std::max(3, n1) || n1 < 3))
        break;
    }
    if (j == nclasses) {
      out.push_back(r1);
      out_levels.push_back(with_levels ? cls_level[i] : n1);  // "useDefaultWeights": the class size
      out_weights.push_back(cls_weight[i]);
    }
  }
  rects.swap(out);
  if (levels) levels->swap(out_levels);
  if (level_weights) level_weights->swap(out_weights);
}

// Haar catalog in the reference's order (traincascade/lib/src/haarfeatures.cpp:127-251); the feature index is part of
// the contract because valCache rows / var_idx are catalog indices.
void haar_catalog(int W, int H, int mode, std::vector<HaarFeature>& out) {
  out.clear();
  auto add = [&](bool tilted, int x0, int y0, int w0, int h0, float t0, int x1, int y1, int w1, int h1, float t1,
                 int x2 = 0, int y2 = 0, int w2 = 0, int h2 = 0, float t2 = 0.f) {
    HaarFeature f;
    const int32_t r[3][4] = {{x0, y0, w0, h0}, {x1, y1, w1, h1}, {x2, y2, w2, h2}};
    std::memcpy(f.r, r, sizeof(r));
    f.w[0] = t0;
    f.w[1] = t1;
    f.w[2] = t2;
    f.tilted = tilted ? 1 : 0;
    out.push_back(f);
  };
  const bool core = mode != CC_HAAR_BASIC, all = mode == CC_HAAR_ALL;
  for (int x = 0; x < W; x++)
    for (int y = 0; y < H; y++)
      for (int dx = 1; dx <= W; dx++)
        for (int dy = 1; dy <= H; dy++) {
          const bool fx1 = x + dx <= W, fy1 = y + dy <= H;
          if (x + 2 * dx <= W && fy1) add(false, x, y, 2 * dx, dy, -1, x + dx, y, dx, dy, 2);          // x2
          if (fx1 && y + 2 * dy <= H) add(false, x, y, dx, 2 * dy, -1, x, y + dy, dx, dy, 2);          // y2
          if (x + 3 * dx <= W && fy1) add(false, x, y, 3 * dx, dy, -1, x + dx, y, dx, dy, 2);          // x3 (weight 2, as coded)
          if (fx1 && y + 3 * dy <= H) add(false, x, y, dx, 3 * dy, -1, x, y + dy, dx, dy, 2);          // y3
          if (core && x + 4 * dx <= W && fy1) add(false, x, y, 4 * dx, dy, -1, x + dx, y, 2 * dx, dy, 2);  // x4
          if (core && fx1 && y + 4 * dy <= H) add(false, x, y, dx, 4 * dy, -1, x, y + dy, dx, 2 * dy, 2);  // y4
          if (x + 2 * dx <= W && y + 2 * dy <= H)
            add(false, x, y, 2 * dx, 2 * dy, -1, x, y, dx, dy, 2, x + dx, y + dy, dx, dy, 2);          // x2_y2
          if (core && x + 3 * dx <= W && y + 3 * dy <= H)
            add(false, x, y, 3 * dx, 3 * dy, -1, x + dx, y + dy, dx, dy, 9);                            // centre
          if (all) {
            if (x + 2 * dx <= W && y + 2 * dx + dy <= H && x - dy >= 0) add(true, x, y, 2 * dx, dy, -1, x, y, dx, dy, 2);
            if (fx1 && y + dx + 2 * dy <= H && x - 2 * dy >= 0) add(true, x, y, dx, 2 * dy, -1, x, y, dx, dy, 2);
            if (x + 3 * dx <= W && y + 3 * dx + dy <= H && x - dy >= 0) add(true, x, y, 3 * dx, dy, -1, x + dx, y + dx, dx, dy, 3);
            if (fx1 && y + dx + 3 * dy <= H && x - 3 * dy >= 0) add(true, x, y, dx, 3 * dy, -1, x - dy, y + dy, dx, dy, 3);
            if (x + 4 * dx <= W && y + 4 * dx + dy <= H && x - dy >= 0) add(true, x, y, 4 * dx, dy, -1, x + dx, y + dx, 2 * dx, dy, 2);
            if (fx1 && y + dx + 4 * dy <= H && x - 4 * dy >= 0) add(true, x, y, dx, 4 * dy, -1, x - dy, y + dy, dx, 2 * dy, 2);
          }
        }
}

// LBP catalog (traincascade/lib/src/lbpfeatures.cpp:35-45): one cell (x y w h) of the 3x3 grid per feature.
void lbp_catalog(int W, int H, std::vector<int32_t>& rects) {
  rects.clear();
  for (int x = 0; x < W; x++)
    for (int y = 0; y < H; y++)
      for (int w = 1; w <= W / 3; w++)
        for (int h = 1; h <= H / 3; h++)
          if (x + 3 * w <= W && y + 3 * h <= H) {
            rects.push_back(x);
            rects.push_back(y);
            rects.push_back(w);
            rects.push_back(h);
          }
}

// ------------------------------------------------------------------------------------------------
// Categorical split: category ordering + scan (o_cvboostree.cpp:289-357 classification, :466-515 regression).
// The accumulation over samples happened on the device; this is the O(n_cat log n_cat) remainder. std::sort with a
// plain `<` on the keys is the reference's call (LessThanPtr over pointers compares the same doubles), so categories
// with equal keys come out in the same order as there.
// ------------------------------------------------------------------------------------------------
void split_categories(const double* hist, int n_cat, bool is_classifier, bool gini, CatSplit& out) {
  const double feps = (double)FLT_EPSILON;
  out.found = false;
  out.quality = -1;
  out.n_left = 0;
  std::memset(out.subset, 0, sizeof(out.subset));
  // scratch on the stack for the usual 256 categories (8 464 calls per node search: no allocator traffic)
  constexpr int kStack = 256;
  int order_s[kStack];
  double key_s[kStack], tot_s[kStack], cnt_s[kStack];
  std::vector<int> order_v;
  std::vector<double> key_v, tot_v, cnt_v;
  int* order = order_s;
  double *key = key_s, *tot = tot_s, *cnt = cnt_s;
  if (n_cat > kStack) {
    order_v.resize((size_t)n_cat);
    key_v.resize((size_t)n_cat);
    tot_v.resize((size_t)n_cat);
    cnt_v.resize((size_t)n_cat);
    order = order_v.data();
    key = key_v.data();
    tot = tot_v.data();
    cnt = cnt_v.data();
  }
  for (int c = 0; c < n_cat; c++) order[(size_t)c] = c;
  double best = -1.0;  // init_quality of a search on its own
  int best_pos = -1;
  if (!is_classifier) {
    double right_w = 0, right_s = 0;
    for (int c = 0; c < n_cat; c++) {
      const double s = hist[2 * c], w = hist[2 * c + 1];
      right_w += w;
      right_s += s;
      cnt[(size_t)c] = w;
      key[(size_t)c] = std::fabs(w) > DBL_EPSILON ? s / w : 0;  // average response of the category
    }
    std::sort(order, order + n_cat, [&](int a, int b) { return key[(size_t)a] < key[(size_t)b]; });
    for (int c = 0; c < n_cat; c++) tot[(size_t)c] = key[(size_t)c] * cnt[(size_t)c];  // "revert back to unnormalized sums"
    double left_w = 0, left_s = 0;
    for (int pos = 0; pos < n_cat - 1; pos++) {
      const int c = order[(size_t)pos];
      const double w = cnt[(size_t)c];
      if (!(w > feps)) continue;
      const double s = tot[(size_t)c];
      left_s += s;
      left_w += w;
      right_s -= s;
      right_w -= w;
      if (left_w > feps && right_w > feps) {
        const double val = (left_s * left_s * right_w + right_s * right_s * left_w) / (left_w * right_w);
        if (best < val) {
          best = val;
          best_pos = pos;
        }
      }
    }
  } else {
    double lcw[2] = {0, 0}, rcw[2] = {0, 0};
    for (int c = 0; c < n_cat; c++) {
      rcw[0] += hist[2 * c];
      rcw[1] += hist[2 * c + 1];
      key[(size_t)c] = hist[2 * c + 1];  // weight of the category's class-1 samples
    }
    double left_w = 0, right_w = rcw[0] + rcw[1];
    std::sort(order, order + n_cat, [&](int a, int b) { return key[(size_t)a] < key[(size_t)b]; });
    for (int pos = 0; pos < n_cat - 1; pos++) {
      const int c = order[(size_t)pos];
      const double w0 = hist[2 * c], w1 = hist[2 * c + 1];
      const double weight = w0 + w1;
      if (weight < feps) continue;
      lcw[0] += w0;
      rcw[0] -= w0;
      lcw[1] += w1;
      rcw[1] -= w1;
      if (gini) {
        const double lsum2 = lcw[0] * lcw[0] + lcw[1] * lcw[1];
        const double rsum2 = rcw[0] * rcw[0] + rcw[1] * rcw[1];
        left_w += weight;
        right_w -= weight;
        if (left_w > feps && right_w > feps) {
          const double val = (lsum2 * right_w + rsum2 * left_w) / (left_w * right_w);
          if (best < val) {
            best = val;
            best_pos = pos;
          }
        }
      } else {
        const double a = lcw[0] + rcw[1], b = lcw[1] + rcw[0];
        const double val = a > b ? a : b;
        if (best < val) {
          best = val;
          best_pos = pos;
        }
      }
    }
  }
  if (best_pos < 0) return;
  out.found = true;
  out.quality = best;
  out.n_left = best_pos + 1;
  for (int pos = 0; pos <= best_pos; pos++) {
    const int c = order[(size_t)pos];
    out.subset[c >> 5] |= 1 << (c & 31);
  }
}

}  // namespace ccamd

using namespace ccamd;

extern "C" {

const char* cc_last_error(void) { return g_last_error.c_str(); }
int cc_version(void) { return 100; }

cc_status cc_scale_plan(int win_w, int win_h, int width, int height, const cc_detect_params* p, cc_scale_info* out, int cap,
                        int* n) {
  if (!p || !n) return set_error(CC_ERR_INVALID_ARG, "cc_scale_plan: null argument");
  if (!(p->scale_factor > 1.0)) return set_error(CC_ERR_INVALID_ARG, "cc_scale_plan: scaleFactor must be > 1");
  if (win_w < 3 || win_h < 3 || width < 1 || height < 1) return set_error(CC_ERR_INVALID_ARG, "cc_scale_plan: bad sizes");
  std::vector<ScaleGeom> g;
  scale_plan(win_w, win_h, width, height, *p, g);
  *n = (int)g.size();
  for (int i = 0; i < (int)g.size() && i < cap && out; i++)
    out[i] = cc_scale_info{g[i].scale, g[i].w, g[i].h, g[i].ystep, g[i].nx, g[i].ny, g[i].win_w, g[i].win_h};
  if ((int)g.size() > cap) return set_error(CC_ERR_BUFFER_TOO_SMALL, "cc_scale_plan: %zu scales, capacity %d", g.size(), cap);
  return CC_OK;
}

cc_status cc_group_rectangles(const cc_rect* rects, int n, int group_threshold, double eps, cc_rect* out, int cap,
                              int* n_out) {
  if ((n > 0 && !rects) || !n_out || n < 0) return set_error(CC_ERR_INVALID_ARG, "cc_group_rectangles: bad argument");
  std::vector<cc_rect> v(rects, rects + n);
  group_rectangles(v, group_threshold, eps);
  *n_out = (int)v.size();
  for (int i = 0; i < (int)v.size() && i < cap && out; i++) out[i] = v[i];
  if ((int)v.size() > cap) return set_error(CC_ERR_BUFFER_TOO_SMALL, "cc_group_rectangles: %zu rects, capacity %d", v.size(), cap);
  return CC_OK;
}

}  // extern "C"
