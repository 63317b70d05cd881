// Cascade model: reader of the new-format cascade.xml (the format CvCascadeClassifier::save writes,
// traincascade/lib/src/cascadeclassifier.cpp:439-456 with the tags of cascadeclassifier.h:27-73) and the
// C ABI accessors of section 1 of include/cascadeclassifier_amd.h. Host code only.
#include <algorithm>
#include <cerrno>
#include <cmath>
#include <cstdio>
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

namespace {
// FileStorage prints float reals with "%.8e" and integral reals as "2." (8 significant digits + exponent round-trip a float)
std::string real_text(float v) {
  char b[64];
  if (v == (float)(long long)v && v > -1e9f && v < 1e9f)
    snprintf(b, sizeof(b), "%lld.", (long long)v);
  else
    snprintf(b, sizeof(b), "%.8e", (double)v);
  return b;
}
// doubles (node values in the legacy layout) go out with "%.16e"
std::string real_text_d(double v) {
  char b[64];
  if (v == (double)(long long)v && v > -1e9 && v < 1e9)
    snprintf(b, sizeof(b), "%lld.", (long long)v);
  else
    snprintf(b, sizeof(b), "%.16e", v);
  return b;
}
// the raw <stageThreshold> whose (float)t - 1e-5f is the stored value; false if no float maps onto it
bool stage_threshold_preimage(float stored, float& t) {
  t = stored + 1e-5f;
  for (int k = 0; k < 64 && (float)(t - 1e-5f) != stored; k++) t = (float)(t - 1e-5f) < stored ? std::nextafterf(t, INFINITY) : std::nextafterf(t, -INFINITY);
  return (float)(t - 1e-5f) == stored;
}
cc_status write_text_file(const char* path, const std::string& text) {
  std::ofstream f(path, std::ios::binary);
  if (!f) return set_error(CC_ERR_IO, "cannot open '%s' for writing", path);
  f.write(text.data(), (std::streamsize)text.size());
  if (!f) return set_error(CC_ERR_IO, "write to '%s' failed", path);
  return CC_OK;
}
}  // namespace

extern "C" {

cc_status cc_cascade_load_xml_mem(const char* text, size_t len, cc_cascade** out) {
  const CNumericLocale c_numbers;  // "%.8e" / strtod must not follow the host program's LC_NUMERIC
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

cc_status cc_cascade_save_xml(const cc_cascade* c, const char* path) {
  const CNumericLocale c_numbers;  // "%.8e" / strtod must not follow the host program's LC_NUMERIC
  if (!c || !path) return set_error(CC_ERR_INVALID_ARG, "cc_cascade_save_xml: null argument");
  const Cascade& m = c->m;
  const bool haar = m.feature_type == CC_FEATURE_HAAR;
  std::ostringstream o;
  int max_weak = 0;
  for (int n : m.stage_ntrees) max_weak = std::max(max_weak, n);
  o << "<?xml version=\"1.0\"?>\n<opencv_storage>\n<cascade>\n";
  o << "  <stageType>BOOST</stageType>\n  <featureType>" << (haar ? "HAAR" : "LBP") << "</featureType>\n";
  o << "  <height>" << m.win_h << "</height>\n  <width>" << m.win_w << "</width>\n";
  o << "  <stageParams>\n    <maxWeakCount>" << max_weak << "</maxWeakCount></stageParams>\n";
  o << "  <featureParams>\n    <maxCatCount>" << m.max_cat_count << "</maxCatCount>\n    <featSize>1</featSize></featureParams>\n";
  o << "  <stageNum>" << m.stage_ntrees.size() << "</stageNum>\n  <stages>\n";
  for (size_t s = 0; s < m.stage_ntrees.size(); s++) {
    // the model keeps (float)stageThreshold - 1e-5f; write back a value that parses to the same float after the
    // reader subtracts the epsilon again: search the neighbourhood of thr + 1e-5f
    float t;
    if (!stage_threshold_preimage(m.stage_threshold[s], t))
      return set_error(CC_ERR_UNSUPPORTED, "cc_cascade_save_xml: stage threshold %zu has no float pre-image", s);
    o << "    <_>\n      <maxWeakCount>" << m.stage_ntrees[s] << "</maxWeakCount>\n      <stageThreshold>" << real_text(t)
      << "</stageThreshold>\n      <weakClassifiers>\n";
    for (int i = 0; i < m.stage_ntrees[s]; i++) {
      const size_t t_i = (size_t)m.stage_first[s] + i;
      const int n0 = m.tree_first_node[t_i], nn = m.tree_nnodes[t_i], l0 = m.tree_first_leaf[t_i];
      o << "        <_>\n          <internalNodes>\n           ";
      for (int k = 0; k < nn; k++) {
        o << " " << m.node_left[n0 + k] << " " << m.node_right[n0 + k] << " " << m.node_feature[n0 + k];
        if (m.subset_size > 0)
          for (int j = 0; j < m.subset_size; j++) o << " " << m.node_subset[(size_t)(n0 + k) * m.subset_size + j];
        else
          o << " " << real_text(m.node_threshold[n0 + k]);
      }
      o << "</internalNodes>\n          <leafValues>\n           ";
      for (int k = 0; k < nn + 1; k++) o << " " << real_text(m.leaves[l0 + k]);
      o << "</leafValues></_>\n";
    }
    o << "      </weakClassifiers></_>\n";
  }
  o << "  </stages>\n  <features>\n";
  const int nf = m.n_features();
  for (int f = 0; f < nf; f++) {
    if (haar) {
      o << "    <_>\n      <rects>\n";
      for (int j = 0; j < 3; j++) {
        const int32_t* r = &m.haar_rects[(size_t)f * 12 + j * 4];
        const float w = m.haar_weights[(size_t)f * 3 + j];
        if (j > 0 && r[2] == 0 && w == 0.0f) break;  // Feature::write stops at the first rect of zero width
        o << "        <_>\n          " << r[0] << " " << r[1] << " " << r[2] << " " << r[3] << " " << real_text(w) << "</_>\n";
      }
      o << "      </rects>\n      <tilted>" << (m.haar_tilted[f] ? 1 : 0) << "</tilted></_>\n";
    } else {
      const int32_t* r = &m.lbp_rects[(size_t)f * 4];
      o << "    <_>\n      <rect>\n        " << r[0] << " " << r[1] << " " << r[2] << " " << r[3] << "</rect></_>\n";
    }
  }
  o << "  </features>\n</cascade>\n</opencv_storage>\n";
  return write_text_file(path, o.str());
}

cc_status cc_cascade_save_xml_legacy(const cc_cascade* c, const char* path) {
  const CNumericLocale c_numbers;  // "%.8e" / strtod must not follow the host program's LC_NUMERIC
  if (!c || !path) return set_error(CC_ERR_INVALID_ARG, "cc_cascade_save_xml_legacy: null argument");
  const Cascade& m = c->m;
  if (m.feature_type != CC_FEATURE_HAAR)
    return set_error(CC_ERR_UNSUPPORTED, "old file format is used for Haar-like features only");
  std::ostringstream o;
  o << "<?xml version=\"1.0\"?>\n<opencv_storage>\n<cascade type_id=\"opencv-haar-classifier\">\n";
  o << "  <size>\n    " << m.win_w << " " << m.win_h << "</size>\n  <stages>\n";
  for (size_t s = 0; s < m.stage_ntrees.size(); s++) {
    float thr;
    if (!stage_threshold_preimage(m.stage_threshold[s], thr))
      return set_error(CC_ERR_UNSUPPORTED, "cc_cascade_save_xml_legacy: stage threshold %zu has no float pre-image", s);
    o << "    <_>\n      <trees>\n";
    for (int i = 0; i < m.stage_ntrees[s]; i++) {
      const size_t t_i = (size_t)m.stage_first[s] + i;
      const int n0 = m.tree_first_node[t_i], l0 = m.tree_first_leaf[t_i];
      o << "        <_>\n";
      // breadth-first walk from the root; an inner child gets the next free number when it is queued
      std::vector<int> queue(1, 0);
      int next_idx = 0;
      for (size_t q = 0; q < queue.size(); q++) {
        const int k = n0 + queue[q];
        const int f = m.node_feature[k];
        o << "          <_>\n            <feature>\n              <rects>\n";
        for (int j = 0; j < 3; j++) {
          const int32_t* r = &m.haar_rects[(size_t)f * 12 + j * 4];
          const float w = m.haar_weights[(size_t)f * 3 + j];
          if (j > 0 && r[2] == 0 && w == 0.0f) break;
          o << "                <_>\n                  " << r[0] << " " << r[1] << " " << r[2] << " " << r[3] << " " << real_text(w) << "</_>\n";
        }
        o << "              </rects>\n              <tilted>" << (m.haar_tilted[f] ? 1 : 0) << "</tilted></feature>\n";
        o << "            <threshold>" << real_text(m.node_threshold[k]) << "</threshold>\n";
        const int child[2] = {m.node_left[k], m.node_right[k]};
        const char* node_tag[2] = {"left_node", "right_node"};
        const char* val_tag[2] = {"left_val", "right_val"};
        for (int side = 0; side < 2; side++) {
          if (child[side] > 0) {
            queue.push_back(child[side]);
            o << "            <" << node_tag[side] << ">" << ++next_idx << "</" << node_tag[side] << ">";
          } else {
            o << "            <" << val_tag[side] << ">" << real_text_d((double)m.leaves[l0 - child[side]]) << "</" << val_tag[side] << ">";
          }
          o << (side == 0 ? "\n" : "</_>\n");
        }
      }
      o << "        </_>\n";
    }
    o << "      </trees>\n      <stage_threshold>" << real_text(thr) << "</stage_threshold>\n      <parent>" << (int)s - 1
      << "</parent>\n      <next>-1</next></_>\n";
  }
  o << "  </stages>\n</cascade>\n</opencv_storage>\n";
  return write_text_file(path, o.str());
}

cc_status cc_vec_read(const char* path, int32_t* count, int32_t* vec_size, uint8_t* pixels, int cap_samples) {
  if (!path || !count || !vec_size) return set_error(CC_ERR_INVALID_ARG, "cc_vec_read: null argument");
  std::ifstream f(path, std::ios::binary);
  if (!f) return set_error(CC_ERR_IO, "cannot open '%s'", path);
  int32_t hdr[2];
  int16_t mm[2];
  f.read(reinterpret_cast<char*>(hdr), 8);
  f.read(reinterpret_cast<char*>(mm), 4);
  if (!f || hdr[0] < 0 || hdr[1] <= 0) return set_error(CC_ERR_PARSE, "wrong file format for %s", path);
  f.seekg(0, std::ios::end);
  const long long fsize = (long long)f.tellg();
  if (12 + (long long)hdr[0] * (1 + 2LL * hdr[1]) > fsize)
    return set_error(CC_ERR_PARSE, "%s: header promises %d samples of %d values but the file has %lld bytes", path, hdr[0], hdr[1], fsize);
  f.seekg(12, std::ios::beg);
  *count = hdr[0];
  *vec_size = hdr[1];
  if (!pixels) return CC_OK;
  std::vector<int16_t> rec((size_t)hdr[1]);
  const int n = std::min(hdr[0], std::max(cap_samples, 0));
  for (int i = 0; i < n; i++) {
    char zero;
    f.read(&zero, 1);
    f.read(reinterpret_cast<char*>(rec.data()), (std::streamsize)rec.size() * 2);
    if (!f) return set_error(CC_ERR_PARSE, "%s: sample %d is truncated (vec-file has incorrect structure)", path, i);
    for (int k = 0; k < hdr[1]; k++) pixels[(size_t)i * hdr[1] + k] = (uint8_t)rec[(size_t)k];
  }
  return CC_OK;
}

cc_status cc_vec_write(const char* path, const uint8_t* pixels, int count, int width, int height) {
  if (!path || (count > 0 && !pixels) || count < 0 || width < 1 || height < 1) return set_error(CC_ERR_INVALID_ARG, "cc_vec_write: bad argument");
  std::ofstream f(path, std::ios::binary);
  if (!f) return set_error(CC_ERR_IO, "cannot open '%s' for writing", path);
  const int32_t hdr[2] = {count, width * height};
  const int16_t mm[2] = {0, 0};
  f.write(reinterpret_cast<const char*>(hdr), 8);
  f.write(reinterpret_cast<const char*>(mm), 4);
  std::vector<int16_t> rec((size_t)width * height);
  for (int i = 0; i < count; i++) {
    const char zero = 0;
    for (size_t k = 0; k < rec.size(); k++) rec[k] = pixels[(size_t)i * rec.size() + k];
    f.write(&zero, 1);
    f.write(reinterpret_cast<const char*>(rec.data()), (std::streamsize)rec.size() * 2);
  }
  if (!f) return set_error(CC_ERR_IO, "write to '%s' failed", path);
  return CC_OK;
}

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
