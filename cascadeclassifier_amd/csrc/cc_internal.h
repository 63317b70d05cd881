// Internal declarations shared by the translation units of libcascadeclassifier_amd.so.
#pragma once

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/cascadeclassifier_amd.h"

#include <locale.h>

namespace ccamd {

// Numbers in cascade XML and in generated kernel source are written and parsed with printf / strtod, which follow the
// calling thread's LC_NUMERIC: under a comma-decimal locale (a host program that called setlocale(LC_ALL, "")) "%.8e"
// would print "1,5e+00" and strtod would stop at the '.' of "1.5". Entry points that format or parse numbers hold one
// of these: it switches the calling THREAD to the "C" locale for its lifetime (uselocale; no other thread is affected).
struct CNumericLocale {
  locale_t prev = (locale_t)0;
  CNumericLocale() {
    static locale_t c = newlocale(LC_ALL_MASK, "C", (locale_t)0);
    if (c != (locale_t)0) prev = uselocale(c);
  }
  ~CNumericLocale() {
    if (prev != (locale_t)0) uselocale(prev);
  }
  CNumericLocale(const CNumericLocale&) = delete;
  CNumericLocale& operator=(const CNumericLocale&) = delete;
};

// ---- error plumbing (thread-local message behind cc_last_error) -------------------------------
cc_status set_error(cc_status code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));

// ---- minimal XML DOM (the subset cv::FileStorage writes) ---------------------------------------
struct XmlNode {
  std::string name;
  std::string text;  // concatenated character data (entities decoded)
  std::vector<std::pair<std::string, std::string>> attrs;
  std::vector<XmlNode> children;
  const XmlNode* child(const char* n) const {
    for (const XmlNode& c : children)
      if (c.name == n) return &c;
    return nullptr;
  }
};
// Returns false and fills err on malformed input.
bool xml_parse(const char* text, size_t len, XmlNode& root, std::string& err);

// ---- parsed cascade ------------------------------------------------
struct HaarFeature {
  int32_t r[3][4];
  float w[3];
  int32_t tilted;
};

struct Cascade {
  int feature_type = 0;
  int win_w = 0, win_h = 0;
  int max_cat_count = 0, subset_size = 0;
  int max_nodes_per_tree = 0;
  bool has_tilted = false;
  // stages
  std::vector<int32_t> stage_first, stage_ntrees;
  std::vector<float> stage_threshold;  // (float)stageThreshold - THRESHOLD_EPS
  // trees (general form)
  std::vector<int32_t> tree_first_node, tree_nnodes, tree_first_leaf;
  std::vector<int32_t> node_left, node_right, node_feature;
  std::vector<float> node_threshold;
  std::vector<int32_t> node_subset;  // subset_size words per node
  std::vector<float> leaves;
  // stump view (max_nodes_per_tree == 1)
  std::vector<int32_t> stump_feature;
  std::vector<float> stump_threshold, stump_left, stump_right;
  // features
  std::vector<int32_t> haar_rects;   // [n][3][4]
  std::vector<float> haar_weights;   // [n][3]
  std::vector<int32_t> haar_tilted;  // [n]
  std::vector<int32_t> lbp_rects;    // [n][4]
  int n_features() const { return feature_type == CC_FEATURE_HAAR ? (int)haar_tilted.size() : (int)(lbp_rects.size() / 4); }
};
cc_status cascade_from_xml(const XmlNode& root, Cascade& out);

// ---- pyramid geometry (host) --------------------------------------------------------------------
struct ScaleGeom {
  float scale;
  int w, h, ystep, nx, ny, win_w, win_h;
};
void scale_plan(int W0, int H0, int imgw, int imgh, const cc_detect_params& p, std::vector<ScaleGeom>& out);

// Per-axis tap table of the fixed-point bilinear resize: left tap and the 8.8 weight of the right tap
// (weight of the left tap is 256 - w1). Border samples have w1 = 0.
struct AxisTaps {
  std::vector<int32_t> ofs;
  std::vector<uint16_t> w1;
};
void linear_exact_taps(int src, int dst, AxisTaps& t);

// levels / level_weights (optional, one per rectangle): the outputRejectLevels overload of cv::groupRectangles
void group_rectangles(std::vector<cc_rect>& rects, int group_threshold, double eps, std::vector<int>* levels = nullptr,
                      std::vector<double>* level_weights = nullptr);

// ---- catalogs (training side) -------------------------------------------------------------------
void haar_catalog(int W, int H, int mode, std::vector<HaarFeature>& out);
void lbp_catalog(int W, int H, std::vector<int32_t>& rects);

// ---- split search, categorical variables: the part after the per-category accumulation (host) ------
// hist: n_cat pairs, regression {sum of response*w, sum of w}, classification {w of class 0, w of class 1}, each
// accumulated in node sample order. Orders the categories and scans them as find_split_cat_reg / find_split_cat_class
// do (o_cvboostree.cpp:466-515, 289-357) with init_quality -1.
struct CatSplit {
  bool found;
  double quality;      // best_val (double, before the cast to float)
  int n_left;          // categories sent left
  int32_t subset[8];
};
void split_categories(const double* hist, int n_cat, bool is_classifier, bool gini, CatSplit& out);

}  // namespace ccamd

struct cc_cascade {
  ccamd::Cascade m;
};
