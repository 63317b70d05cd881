// Cascade model: reader of the new-format cascade.xml (the format CvCascadeClassifier::save writes,
// traincascade/lib/src/cascadeclassifier.cpp:439-456 with the tags of cascadeclassifier.h:27-73) and the
// C ABI accessors of section 1 of include/cascadeclassifier_amd.h. Host code only.
#include <cerrno>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>

#include "cc_internal.h"

namespace ccamd {
namespace {

bool split_tokens(const std::string& s, std::vector<std::string>& out) {
  out.clear();
  size_t i = 0, n = s.size();
  while (i < n) {
    while (i < n && std::isspace((unsigned char)s[i])) i++;
    size_t j = i;
    while (j < n && !std::isspace((unsigned char)s[j])) j++;
    if (j > i) out.emplace_back(s, i, j - i);
    i = j;
  }
  return true;
}

bool to_int(const std::string& t, int32_t& v) {
  errno = 0;
  char* e = nullptr;
  long long r = std::strtoll(t.c_str(), &e, 10);
  if (e == t.c_str() || errno) return false;
  if (*e == '.') {  // FileStorage writes some ints as "6." when read through a float path; accept integral reals
    double d = std::strtod(t.c_str(), &e);
    if (*e || d != std::floor(d)) return false;
    r = (long long)d;
  } else if (*e)
    return false;
  if (r < INT32_MIN || r > INT32_MAX) return false;
  v = (int32_t)r;
  return true;
}

bool to_double(const std::string& t, double& v) {
  errno = 0;
  char* e = nullptr;
  v = std::strtod(t.c_str(), &e);
  return e != t.c_str() && *e == 0;
}

bool node_int(const XmlNode* n, int32_t& v) {
  if (!n) return false;
  std::vector<std::string> tk;
  split_tokens(n->text, tk);
  return tk.size() == 1 && to_int(tk[0], v);
}

bool node_double(const XmlNode* n, double& v) {
  if (!n) return false;
  std::vector<std::string> tk;
  split_tokens(n->text, tk);
  return tk.size() == 1 && to_double(tk[0], v);
}

std::string trimmed(const std::string& s) {
  size_t a = 0, b = s.size();
  while (a < b && std::isspace((unsigned char)s[a])) a++;
  while (b > a && std::isspace((unsigned char)s[b - 1])) b--;
  return s.substr(a, b - a);
}

}  // namespace

cc_status cascade_from_xml(const XmlNode& root, Cascade& c) {
  const XmlNode* casc = nullptr;
  if (root.name == "opencv_storage") {
    casc = root.child("cascade");
    if (!casc && !root.children.empty()) casc = &root.children[0];
  } else
    casc = &root;
  if (!casc) return set_error(CC_ERR_PARSE, "cascade XML: <opencv_storage> has no cascade node");
  const XmlNode* st = casc->child("stageType");
  if (!st) {
    if (casc->child("size") || casc->child("stages"))
      return set_error(CC_ERR_UNSUPPORTED,
                       "cascade XML: old-format (opencv-haar-classifier) cascade; only the new format written by "
                       "traincascade is supported");
    return set_error(CC_ERR_PARSE, "cascade XML: missing <stageType>");
  }
  if (trimmed(st->text) != "BOOST") return set_error(CC_ERR_PARSE, "cascade XML: stageType must be BOOST");
  const XmlNode* ft = casc->child("featureType");
  if (!ft) return set_error(CC_ERR_PARSE, "cascade XML: missing <featureType>");
  std::string fts = trimmed(ft->text);
  if (fts == "HAAR")
    c.feature_type = CC_FEATURE_HAAR;
  else if (fts == "LBP")
    c.feature_type = CC_FEATURE_LBP;
  else if (fts == "HOG")
    return set_error(CC_ERR_UNSUPPORTED, "cascade XML: HOG cascades are outside the accelerated path (Haar/LBP only)");
  else
    return set_error(CC_ERR_PARSE, "cascade XML: unknown featureType '%s'", fts.c_str());
  int32_t w = 0, h = 0;
  if (!node_int(casc->child("width"), w) || !node_int(casc->child("height"), h) || w < 3 || h < 3 || w > 4096 || h > 4096)
    return set_error(CC_ERR_PARSE, "cascade XML: bad <width>/<height>");
  c.win_w = w;
  c.win_h = h;
  const XmlNode* fp = casc->child("featureParams");
  if (!fp) fp = casc->child("featuhreParams");  // historical typo carried by some stock files
  int32_t maxcat = c.feature_type == CC_FEATURE_LBP ? 256 : 0;
  if (fp && fp->child("maxCatCount") && !node_int(fp->child("maxCatCount"), maxcat))
    return set_error(CC_ERR_PARSE, "cascade XML: bad <maxCatCount>");
  if (maxcat < 0 || maxcat > 256 || (c.feature_type == CC_FEATURE_LBP && maxcat != 256) ||
      (c.feature_type == CC_FEATURE_HAAR && maxcat != 0))
    return set_error(CC_ERR_PARSE, "cascade XML: maxCatCount %d does not match featureType %s", maxcat, fts.c_str());
  c.max_cat_count = maxcat;
  c.subset_size = maxcat > 0 ? (maxcat + 31) / 32 : 0;
  const int node_step = 3 + (maxcat > 0 ? c.subset_size : 1);

  // ---- features first (stage nodes are validated against their count)
  const XmlNode* feats = casc->child("features");
  if (!feats) return set_error(CC_ERR_PARSE, "cascade XML: missing <features>");
  std::vector<std::string> tk;
  for (const XmlNode& f : feats->children) {
    if (f.name != "_") continue;
    if (c.feature_type == CC_FEATURE_HAAR) {
      const XmlNode* rects = f.child("rects");
      if (!rects) return set_error(CC_ERR_PARSE, "cascade XML: Haar feature without <rects>");
      int32_t r[3][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
      float wt[3] = {0, 0, 0};
      int ri = 0;
      int32_t tilted = 0;
      if (f.child("tilted") && !node_int(f.child("tilted"), tilted)) return set_error(CC_ERR_PARSE, "cascade XML: bad <tilted>");
      tilted = tilted != 0;
      for (const XmlNode& rn : rects->children) {
        if (rn.name != "_") continue;
        if (ri >= 3) return set_error(CC_ERR_PARSE, "cascade XML: more than 3 rects in a Haar feature");
        split_tokens(rn.text, tk);
        double wd = 0;
        if (tk.size() != 5 || !to_int(tk[0], r[ri][0]) || !to_int(tk[1], r[ri][1]) || !to_int(tk[2], r[ri][2]) ||
            !to_int(tk[3], r[ri][3]) || !to_double(tk[4], wd))
          return set_error(CC_ERR_PARSE, "cascade XML: malformed Haar rect '%s'", rn.text.c_str());
        wt[ri] = (float)wd;
        // every integral entry the rect touches must lie inside the (W+1)x(H+1) window integral
        const int x = r[ri][0], y = r[ri][1], rw = r[ri][2], rh = r[ri][3];
        bool ok = rw >= 0 && rh >= 0 && x >= 0 && y >= 0;
        if (!tilted)
          ok = ok && x + rw <= w && y + rh <= h;
        else
          ok = ok && x - rh >= 0 && x + rw <= w && y + rw + rh <= h;
        if (!ok)
          return set_error(CC_ERR_PARSE, "cascade XML: Haar rect (%d %d %d %d%s) leaves the %dx%d window", x, y, rw, rh,
                           tilted ? " tilted" : "", w, h);
        ri++;
      }
      if (ri < 1) return set_error(CC_ERR_PARSE, "cascade XML: Haar feature with no rect");
      for (int j = 0; j < 3; j++) {
        for (int k = 0; k < 4; k++) c.haar_rects.push_back(r[j][k]);
        c.haar_weights.push_back(wt[j]);
      }
      c.haar_tilted.push_back(tilted);
      c.has_tilted = c.has_tilted || tilted;
    } else {
      const XmlNode* rn = f.child("rect");
      int32_t r[4];
      if (!rn) return set_error(CC_ERR_PARSE, "cascade XML: LBP feature without <rect>");
      split_tokens(rn->text, tk);
      if (tk.size() != 4 || !to_int(tk[0], r[0]) || !to_int(tk[1], r[1]) || !to_int(tk[2], r[2]) || !to_int(tk[3], r[3]))
        return set_error(CC_ERR_PARSE, "cascade XML: malformed LBP rect '%s'", rn->text.c_str());
      if (r[0] < 0 || r[1] < 0 || r[2] < 1 || r[3] < 1 || r[0] + 3 * r[2] > w || r[1] + 3 * r[3] > h)
        return set_error(CC_ERR_PARSE, "cascade XML: LBP rect (%d %d %d %d) leaves the %dx%d window", r[0], r[1], r[2], r[3], w, h);
      for (int k = 0; k < 4; k++) c.lbp_rects.push_back(r[k]);
    }
  }
  const int nfeat = c.n_features();
  if (nfeat == 0) return set_error(CC_ERR_PARSE, "cascade XML: empty <features>");

  // ---- stages
  const XmlNode* stages = casc->child("stages");
  if (!stages) return set_error(CC_ERR_PARSE, "cascade XML: missing <stages>");
  c.max_nodes_per_tree = 0;
  for (const XmlNode& sn : stages->children) {
    if (sn.name != "_") continue;
    double thr = 0;
    if (!node_double(sn.child("stageThreshold"), thr)) return set_error(CC_ERR_PARSE, "cascade XML: bad <stageThreshold>");
    const XmlNode* weak = sn.child("weakClassifiers");
    if (!weak) return set_error(CC_ERR_PARSE, "cascade XML: stage without <weakClassifiers>");
    c.stage_first.push_back((int32_t)c.tree_first_node.size());
    c.stage_threshold.push_back((float)thr - 1e-5f);  // THRESHOLD_EPS
    int ntrees = 0;
    for (const XmlNode& wn : weak->children) {
      if (wn.name != "_") continue;
      const XmlNode* in = wn.child("internalNodes");
      const XmlNode* lv = wn.child("leafValues");
      if (!in || !lv) return set_error(CC_ERR_PARSE, "cascade XML: weak classifier without internalNodes/leafValues");
      split_tokens(in->text, tk);
      if (tk.empty() || tk.size() % node_step) return set_error(CC_ERR_PARSE, "cascade XML: internalNodes length %zu is not a multiple of %d", tk.size(), node_step);
      const int nn = (int)tk.size() / node_step;
      std::vector<std::string> lt;
      split_tokens(lv->text, lt);
      const int nl = (int)lt.size();
      if (nl != nn + 1)  // upstream advances its leaf cursor by nodeCount + 1 per tree
        return set_error(CC_ERR_PARSE, "cascade XML: a tree with %d nodes must have %d leaf values (found %d)", nn, nn + 1, nl);
      c.tree_first_node.push_back((int32_t)c.node_left.size());
      c.tree_nnodes.push_back(nn);
      c.tree_first_leaf.push_back((int32_t)c.leaves.size());
      for (int k = 0; k < nn; k++) {
        const std::string* t = &tk[(size_t)k * node_step];
        int32_t l, r, fi;
        if (!to_int(t[0], l) || !to_int(t[1], r) || !to_int(t[2], fi)) return set_error(CC_ERR_PARSE, "cascade XML: malformed internalNodes");
        // child > 0: internal node index inside this tree; child <= 0: leaf index -child
        if (l >= nn || r >= nn || -l >= nl || -r >= nl) return set_error(CC_ERR_PARSE, "cascade XML: tree child index out of range");
        // the writer numbers internal nodes breadth-first, so a child's index exceeds its parent's; insisting on it
        // guarantees that walking the tree terminates (a cyclic file would otherwise hang a kernel)
        if ((l > 0 && l <= k) || (r > 0 && r <= k)) return set_error(CC_ERR_PARSE, "cascade XML: tree child index does not increase (cycle)");
        if (fi < 0 || fi >= nfeat) return set_error(CC_ERR_PARSE, "cascade XML: featureIdx %d out of range (features: %d)", fi, nfeat);
        c.node_left.push_back(l);
        c.node_right.push_back(r);
        c.node_feature.push_back(fi);
        if (c.subset_size > 0) {
          for (int j = 0; j < c.subset_size; j++) {
            int32_t sw;
            if (!to_int(t[3 + j], sw)) return set_error(CC_ERR_PARSE, "cascade XML: malformed category subset");
            c.node_subset.push_back(sw);
          }
          c.node_threshold.push_back(0.f);
        } else {
          double td;
          if (!to_double(t[3], td)) return set_error(CC_ERR_PARSE, "cascade XML: malformed node threshold");
          c.node_threshold.push_back((float)td);
        }
      }
      for (int k = 0; k < nl; k++) {
        double v;
        if (!to_double(lt[k], v)) return set_error(CC_ERR_PARSE, "cascade XML: malformed leaf value");
        c.leaves.push_back((float)v);
      }
      if (nn > c.max_nodes_per_tree) c.max_nodes_per_tree = nn;
      ntrees++;
    }
    if (ntrees == 0) return set_error(CC_ERR_PARSE, "cascade XML: stage with no weak classifier");
    c.stage_ntrees.push_back(ntrees);
  }
  if (c.stage_ntrees.empty()) return set_error(CC_ERR_PARSE, "cascade XML: no stages");

  if (c.max_nodes_per_tree == 1) {
    // Stump view, as upstream builds it: left = leafValues[0], right = leafValues[1] (the writer always emits child
    // refs 0 / -1 for a stump, o_cvcascadeboosttree.cpp:60-77); two leaves per stump are required.
    const size_t nt = c.tree_first_node.size();
    for (size_t t = 0; t < nt; t++) {
      const int n0 = c.tree_first_node[t], l0 = c.tree_first_leaf[t];
      const int nl = (t + 1 < nt ? c.tree_first_leaf[t + 1] : (int)c.leaves.size()) - l0;
      if (nl != 2) return set_error(CC_ERR_PARSE, "cascade XML: stump %zu has %d leaf values (expected 2)", t, nl);
      c.stump_feature.push_back(c.node_feature[n0]);
      c.stump_threshold.push_back(c.node_threshold[n0]);
      c.stump_left.push_back(c.leaves[l0]);
      c.stump_right.push_back(c.leaves[l0 + 1]);
    }
  }
  return CC_OK;
}

}  // namespace ccamd

using namespace ccamd;

extern "C" {

cc_status cc_cascade_load_xml_mem(const char* text, size_t len, cc_cascade** out) {
  if (!text || !out) return set_error(CC_ERR_INVALID_ARG, "cc_cascade_load_xml_mem: null argument");
  *out = nullptr;
  XmlNode root;
  std::string err;
  if (!xml_parse(text, len, root, err)) return set_error(CC_ERR_PARSE, "cascade XML: %s", err.c_str());
  cc_cascade* c = new cc_cascade();
  cc_status st = cascade_from_xml(root, c->m);
  if (st != CC_OK) {
    delete c;
    return st;
  }
  *out = c;
  return CC_OK;
}

cc_status cc_cascade_load_xml(const char* path, cc_cascade** out) {
  if (!path || !out) return set_error(CC_ERR_INVALID_ARG, "cc_cascade_load_xml: null argument");
  *out = nullptr;
  std::ifstream f(path, std::ios::binary);
  if (!f) return set_error(CC_ERR_IO, "cannot open '%s'", path);
  std::stringstream ss;
  ss << f.rdbuf();
  std::string s = ss.str();
  return cc_cascade_load_xml_mem(s.data(), s.size(), out);
}

void cc_cascade_destroy(cc_cascade* c) { delete c; }

cc_status cc_cascade_info_get(const cc_cascade* c, cc_cascade_info* info) {
  if (!c || !info) return set_error(CC_ERR_INVALID_ARG, "cc_cascade_info_get: null argument");
  const Cascade& m = c->m;
  info->feature_type = m.feature_type;
  info->win_w = m.win_w;
  info->win_h = m.win_h;
  info->n_stages = (int32_t)m.stage_ntrees.size();
  info->n_weak = (int32_t)m.tree_first_node.size();
  info->n_nodes = (int32_t)m.node_left.size();
  info->n_leaves = (int32_t)m.leaves.size();
  info->n_features = m.n_features();
  info->max_cat_count = m.max_cat_count;
  info->subset_size = m.subset_size;
  info->max_nodes_per_tree = m.max_nodes_per_tree;
  info->has_tilted = m.has_tilted ? 1 : 0;
  return CC_OK;
}

cc_status cc_cascade_stages(const cc_cascade* c, const int32_t** first_weak, const int32_t** n_weak, const float** threshold) {
  if (!c) return set_error(CC_ERR_INVALID_ARG, "cc_cascade_stages: null cascade");
  if (first_weak) *first_weak = c->m.stage_first.data();
  if (n_weak) *n_weak = c->m.stage_ntrees.data();
  if (threshold) *threshold = c->m.stage_threshold.data();
  return CC_OK;
}

cc_status cc_cascade_stumps(const cc_cascade* c, const int32_t** feature_idx, const float** threshold, const float** left,
                            const float** right, const int32_t** subsets) {
  if (!c) return set_error(CC_ERR_INVALID_ARG, "cc_cascade_stumps: null cascade");
  if (c->m.max_nodes_per_tree != 1) return set_error(CC_ERR_UNSUPPORTED, "cc_cascade_stumps: cascade has trees deeper than stumps");
  if (feature_idx) *feature_idx = c->m.stump_feature.data();
  if (threshold) *threshold = c->m.stump_threshold.data();
  if (left) *left = c->m.stump_left.data();
  if (right) *right = c->m.stump_right.data();
  if (subsets) *subsets = c->m.subset_size ? c->m.node_subset.data() : nullptr;
  return CC_OK;
}

cc_status cc_cascade_features(const cc_cascade* c, const int32_t** rects, const float** weights, const int32_t** tilted) {
  if (!c) return set_error(CC_ERR_INVALID_ARG, "cc_cascade_features: null cascade");
  if (c->m.feature_type == CC_FEATURE_HAAR) {
    if (rects) *rects = c->m.haar_rects.data();
    if (weights) *weights = c->m.haar_weights.data();
    if (tilted) *tilted = c->m.haar_tilted.data();
  } else {
    if (rects) *rects = c->m.lbp_rects.data();
    if (weights) *weights = nullptr;
    if (tilted) *tilted = nullptr;
  }
  return CC_OK;
}

}  // extern "C"
