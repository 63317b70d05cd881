// Implementation of the C++ host adaptor (see ccamd/traincascade_features.hpp). Everything numeric is delegated to the
// C ABI (HIP kernels); status codes become cv::Exception like the reference's CV_Assert / CV_Error failures.
#include <locale.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>

#include "ccamd/traincascade_features.hpp"
#include "ccamd/value_cache_policy.hpp"

namespace {

// printf("%e") / strtod follow the calling thread's LC_NUMERIC; XML numbers must not (same guard as the library's).
struct CNumericLocale {
  locale_t prev = (locale_t)0;
  CNumericLocale() {
    static locale_t c = newlocale(LC_ALL_MASK, "C", (locale_t)0);
    if (c != (locale_t)0) prev = uselocale(c);
  }
  ~CNumericLocale() {
    if (prev != (locale_t)0) uselocale(prev);
  }
};

[[noreturn]] void throw_last(const char* where) {
  throw cv::Exception(-1, std::string(where) + ": " + cc_last_error());
}
inline void check(cc_status st, const char* where) {
  if (st != CC_OK) throw_last(where);
}

// Per-thread caches behind the scalar operator()(fi, si): the bookkeeping (which access hits, what a miss launches) is
// ccamd::ValueCacheIndex (ccamd/value_cache_policy.hpp), the values live here.
struct ValueCache {
  ccamd::ValueCacheIndex ix;
  std::vector<float> row;  // values of feature ix.row_fi for every sample
  std::vector<float> col;  // values of the learned list ix.list for sample ix.list_si
};
thread_local ValueCache g_cache;
std::atomic<unsigned long long> g_next_uid{1};

}  // namespace

// ---------------------------------------------------------------- cv::FileStorage / cv::FileNode (XML)
#ifndef CCAMD_USE_OPENCV
namespace cv {

namespace {
// Minimal XML reader for <opencv_storage> documents: elements, character data, the five predefined entities; comments,
// the XML declaration and attributes are skipped. Returns null on malformed input.
std::shared_ptr<FileNode::Elem> parse_storage(const std::string& t) {
  size_t i = 0;
  std::vector<std::shared_ptr<FileNode::Elem>> stack;
  std::shared_ptr<FileNode::Elem> top_level;
  auto decode = [](const std::string& raw) {
    std::string o;
    for (size_t k = 0; k < raw.size(); k++) {
      if (raw[k] != '&') {
        o += raw[k];
        continue;
      }
      const size_t e = raw.find(';', k);
      if (e == std::string::npos) return raw;
      const std::string ent = raw.substr(k + 1, e - k - 1);
      o += ent == "lt" ? '<' : ent == "gt" ? '>' : ent == "amp" ? '&' : ent == "quot" ? '"' : ent == "apos" ? '\'' : '?';
      k = e;
    }
    return o;
  };
  while (i < t.size()) {
    if (t[i] != '<') {
      const size_t e = t.find('<', i);
      const std::string chunk = t.substr(i, (e == std::string::npos ? t.size() : e) - i);
      if (!stack.empty()) stack.back()->text += decode(chunk);
      i = e == std::string::npos ? t.size() : e;
      continue;
    }
    if (t.compare(i, 4, "<!--") == 0) {
      const size_t e = t.find("-->", i);
      if (e == std::string::npos) return nullptr;
      i = e + 3;
      continue;
    }
    if (t.compare(i, 2, "<?") == 0) {
      const size_t e = t.find("?>", i);
      if (e == std::string::npos) return nullptr;
      i = e + 2;
      continue;
    }
    const size_t e = t.find('>', i);
    if (e == std::string::npos) return nullptr;
    std::string tag = t.substr(i + 1, e - i - 1);
    i = e + 1;
    if (!tag.empty() && tag[0] == '/') {
      if (stack.empty() || stack.back()->name != tag.substr(1)) return nullptr;
      auto done = stack.back();
      stack.pop_back();
      if (stack.empty()) top_level = done;
      continue;
    }
    const bool self_closing = !tag.empty() && tag.back() == '/';
    if (self_closing) tag.pop_back();
    const size_t sp = tag.find_first_of(" \t\r\n");
    auto el = std::make_shared<FileNode::Elem>();
    el->name = tag.substr(0, sp);
    if (el->name.empty()) return nullptr;
    if (!stack.empty()) stack.back()->kids.push_back(el);
    if (!self_closing)
      stack.push_back(el);
    else if (stack.empty())
      top_level = el;
  }
  return stack.empty() ? top_level : nullptr;
}
}  // namespace

std::string FileNode::text() const {
  if (!e_) return std::string();
  const std::string& t = e_->text;
  const size_t a = t.find_first_not_of(" \t\r\n"), b = t.find_last_not_of(" \t\r\n");
  return a == std::string::npos ? std::string() : t.substr(a, b - a + 1);
}
bool FileNode::isInt() const {
  if (!e_ || !e_->kids.empty()) return false;
  const std::string t = text();
  if (t.empty()) return false;
  char* end = nullptr;
  (void)std::strtol(t.c_str(), &end, 10);
  return end && *end == 0;
}
bool FileNode::isReal() const {
  if (!e_ || !e_->kids.empty() || isInt()) return false;
  const std::string t = text();
  if (t.empty()) return false;
  char* end = nullptr;
  const CNumericLocale c_numbers;
  (void)std::strtod(t.c_str(), &end);
  return end && *end == 0;
}
FileNode FileNode::operator[](const std::string& key) const {
  if (e_)
    for (const auto& k : e_->kids)
      if (k->name == key) return FileNode(k);
  return FileNode();
}
FileNode::operator int() const {  // like OpenCV: a missing node reads as 0
  return e_ ? (int)std::strtol(text().c_str(), nullptr, 10) : 0;
}
FileNode::operator double() const {
  const CNumericLocale c_numbers;
  return e_ ? std::strtod(text().c_str(), nullptr) : 0.0;
}

bool FileStorage::open(const std::string& filename, int flags) {
  release();
  root_.reset();
  if (!(flags & WRITE)) {  // READ: parse the whole document (a file, or with MEMORY the text itself)
    std::string text = filename;
    if (!(flags & MEMORY)) {
      std::ifstream f(filename, std::ios::binary);
      if (!f) return false;
      std::ostringstream ss;
      ss << f.rdbuf();
      text = ss.str();
    }
    auto top = parse_storage(text);
    if (!top || top->name != "opencv_storage") return false;
    root_ = top;
    return true;
  }
  filename_ = filename;
  memory_ = (flags & MEMORY) != 0;
  opened_ = true;
  out_.str("");
  out_ << "<?xml version=\"1.0\"?>\n<opencv_storage>\n";
  stack_.clear();
  stack_.push_back(Level{true, false, "opencv_storage"});
  return true;
}

void FileStorage::indent() {
  if (line_open_) return;
  for (size_t i = 1; i < stack_.size(); i++) out_ << "  ";
}

std::string FileStorage::releaseAndGetString() {
  std::string s;
  if (opened_) {
    if (line_open_) out_ << "\n";
    line_open_ = false;
    out_ << "</opencv_storage>\n";
    s = out_.str();
    opened_ = false;
  }
  return s;
}

void FileStorage::release() {
  if (!opened_) return;
  const std::string s = releaseAndGetString();
  if (!memory_ && !filename_.empty()) {
    std::ofstream f(filename_, std::ios::binary);
    f << s;
  }
}

void FileStorage::element_open(const std::string& tag) {
  if (line_open_) {
    out_ << "\n";
    line_open_ = false;
  }
  indent();
  out_ << "<" << tag << ">";
}

FileStorage& FileStorage::putNumber(const std::string& text) {
  if (!opened_) throw Exception(-2, "FileStorage is not opened for writing");
  Level& top = stack_.back();
  if (top.is_map) {
    if (!have_key_) throw Exception(-2, "FileStorage: value without a key inside a map");
    element_open(pending_key_);
    out_ << text << "</" << pending_key_ << ">\n";
    have_key_ = false;
  } else if (top.flow) {  // "[:" sequence: space separated on one line
    out_ << (line_open_ ? " " : "") << text;
    line_open_ = true;
  } else {
    element_open("_");
    out_ << text << "</_>\n";
  }
  return *this;
}

FileStorage& FileStorage::put(const std::string& s) {
  if (!opened_) throw Exception(-2, "FileStorage is not opened for writing");
  if (s == "{" || s == "{:" || s == "[" || s == "[:") {
    const bool is_map = s[0] == '{', flow = s.size() > 1;
    Level& top = stack_.back();
    std::string tag = "_";
    if (top.is_map) {
      if (!have_key_) throw Exception(-2, "FileStorage: structure without a key inside a map");
      tag = pending_key_;
      have_key_ = false;
    }
    element_open(tag);
    if (flow) {
      out_ << "\n";
      stack_.push_back(Level{is_map, flow, tag});
      indent();
      line_open_ = false;
    } else {
      out_ << "\n";
      stack_.push_back(Level{is_map, flow, tag});
    }
    return *this;
  }
  if (s == "}" || s == "]") {
    if (stack_.size() <= 1) throw Exception(-2, "FileStorage: unbalanced structure end");
    const Level lv = stack_.back();
    stack_.pop_back();
    if (lv.flow) {
      out_ << "</" << lv.tag << ">\n";
      line_open_ = false;
    } else {
      indent();
      out_ << "</" << lv.tag << ">\n";
    }
    return *this;
  }
  Level& top = stack_.back();
  if (top.is_map && !have_key_) {
    pending_key_ = s;
    have_key_ = true;
    return *this;
  }
  return putNumber(s);  // string value
}

FileStorage& operator<<(FileStorage& fs, const std::string& s) { return fs.put(s); }
FileStorage& operator<<(FileStorage& fs, const char* s) { return fs.put(std::string(s)); }
FileStorage& operator<<(FileStorage& fs, int v) { return fs.putNumber(std::to_string(v)); }
FileStorage& operator<<(FileStorage& fs, bool v) { return fs.putNumber(v ? "1" : "0"); }
static std::string real_text(double v, const char* fmt) {
  const CNumericLocale c_numbers;
  if (v == (double)(long long)v && v > -1e15 && v < 1e15) {  // OpenCV prints integral reals as "2."
    char b[64];
    snprintf(b, sizeof(b), "%lld.", (long long)v);
    return b;
  }
  char b[64];
  snprintf(b, sizeof(b), fmt, v);
  return b;
}
FileStorage& operator<<(FileStorage& fs, float v) { return fs.putNumber(real_text(v, "%.8e")); }
FileStorage& operator<<(FileStorage& fs, double v) { return fs.putNumber(real_text(v, "%.16e")); }

}  // namespace cv
#endif

// ---------------------------------------------------------------- params
CvParams::CvParams() : name("params") {}  // features.cpp:27
void CvParams::printDefaults() const { std::cout << "--" << name << "--" << std::endl; }  // features.cpp:28-29
void CvParams::printAttrs() const {}                                                      // features.cpp:30
bool CvParams::scanAttr(const std::string, const std::string) { return false; }           // features.cpp:31

CvFeatureParams::CvFeatureParams() : maxCatCount(0), featSize(1) { name = CC_FEATURE_PARAMS; }  // features.cpp:36-39

void CvFeatureParams::init(const CvFeatureParams& fp) {  // features.cpp:41-45
  maxCatCount = fp.maxCatCount;
  featSize = fp.featSize;
}

void CvFeatureParams::write(cv::FileStorage& fs) const {  // features.cpp:47-51
  fs << CC_MAX_CAT_COUNT << maxCatCount;
  fs << CC_FEATURE_SIZE << featSize;
}

bool CvFeatureParams::read(const cv::FileNode& node) {  // features.cpp:53-60
  if (node.empty()) return false;
  maxCatCount = node[CC_MAX_CAT_COUNT];
  featSize = node[CC_FEATURE_SIZE];
  return maxCatCount >= 0 && featSize >= 1;
}

cv::Ptr<CvFeatureParams> CvFeatureParams::create(int featureType) {  // features.cpp:62-68 (HOG: outside this path)
  return featureType == HAAR ? cv::Ptr<CvFeatureParams>(new CvHaarFeatureParams)
         : featureType == LBP ? cv::Ptr<CvFeatureParams>(new CvLBPFeatureParams)
                              : cv::Ptr<CvFeatureParams>();
}

CvHaarFeatureParams::CvHaarFeatureParams() : mode(BASIC) { name = HFP_NAME; }            // haarfeatures.cpp:12-15
CvHaarFeatureParams::CvHaarFeatureParams(int _mode) : mode(_mode) { name = HFP_NAME; }  // haarfeatures.cpp:17-20

void CvHaarFeatureParams::init(const CvFeatureParams& fp) {  // haarfeatures.cpp:22-26
  CvFeatureParams::init(fp);
  mode = dynamic_cast<const CvHaarFeatureParams&>(fp).mode;
}

void CvHaarFeatureParams::write(cv::FileStorage& fs) const {  // haarfeatures.cpp:28-36
  CvFeatureParams::write(fs);
  const std::string modeStr = mode == BASIC ? CC_MODE_BASIC : mode == CORE ? CC_MODE_CORE : mode == ALL ? CC_MODE_ALL : std::string();
  CV_Assert(!modeStr.empty());
  fs << CC_MODE << modeStr;
}

bool CvHaarFeatureParams::read(const cv::FileNode& node) {  // haarfeatures.cpp:38-52
  if (!CvFeatureParams::read(node)) return false;
  cv::FileNode rnode = node[CC_MODE];
  if (!rnode.isString()) return false;
  std::string modeStr;
  rnode >> modeStr;
  mode = !modeStr.compare(CC_MODE_BASIC) ? BASIC : !modeStr.compare(CC_MODE_CORE) ? CORE : !modeStr.compare(CC_MODE_ALL) ? ALL : -1;
  return mode >= 0;
}

void CvHaarFeatureParams::printDefaults() const {  // haarfeatures.cpp:54-59 (the unbalanced bracket is the reference's text)
  CvFeatureParams::printDefaults();
  std::cout << "  [-mode <" CC_MODE_BASIC << "(default) | " << CC_MODE_CORE << " | " << CC_MODE_ALL << std::endl;
}

void CvHaarFeatureParams::printAttrs() const {  // haarfeatures.cpp:61-68
  CvFeatureParams::printAttrs();
  const std::string mode_str = mode == BASIC ? CC_MODE_BASIC : mode == CORE ? CC_MODE_CORE : mode == ALL ? CC_MODE_ALL : "";
  std::cout << "mode: " << mode_str << std::endl;
}

// haarfeatures.cpp:70-85. Kept as the reference has it: "-mode" is parsed (an unknown value leaves mode == -1) but the
// function returns false either way, since the base class knows no attribute (pinned by test_features.cpp:96-106).
bool CvHaarFeatureParams::scanAttr(const std::string prmName, const std::string val) {
  if (!CvFeatureParams::scanAttr(prmName, val)) {
    if (!prmName.compare("-mode")) {
      mode = !val.compare(CC_MODE_CORE) ? CORE : !val.compare(CC_MODE_ALL) ? ALL : !val.compare(CC_MODE_BASIC) ? BASIC : -1;
      if (mode == -1) return false;
    }
    return false;
  }
  return true;
}

CvLBPFeatureParams::CvLBPFeatureParams() {  // lbpfeatures.cpp:9-13
  maxCatCount = 256;
  name = LBPF_NAME;
}

// ---------------------------------------------------------------- evaluator base
CvFeatureEvaluator::CvFeatureEvaluator()
    : npos(0), nneg(0), numFeatures(0), featureParams(nullptr), h(nullptr), maxSampleCount(0), generation(0), uid(0), lastSetIdx(-1), lastSetMirrored(false) {}

CvFeatureEvaluator::~CvFeatureEvaluator() {
  if (h) cc_eval_destroy(h);
}

void CvFeatureEvaluator::init(const CvFeatureParams* _featureParams, int _maxSampleCount, cv::Size _winSize) {
  CV_Assert(_maxSampleCount > 0);  // features.cpp:75
  featureParams = const_cast<CvFeatureParams*>(_featureParams);  // non-owning, as in the reference (features.cpp:76)
  winSize = _winSize;
  numFeatures = 0;
  maxSampleCount = _maxSampleCount;
  if (h) {
    cc_eval_destroy(h);
    h = nullptr;
  }
  lastSetIdx = -1;
  lastSetMirrored = false;
  check(cc_eval_create(featureType(), haarMode(), winSize.width, winSize.height, _maxSampleCount, /*device=*/0, &h),
        "CvFeatureEvaluator::init");
  // `cls` is a header over the library's host label array: CvCascadeBoostTrainData wraps it without copying
  // (o_cvcascadeboosttraindata.cpp:238-239), so it must stay contiguous float in host memory.
  cls = cv::Mat(_maxSampleCount, 1, CV_32FC1, const_cast<float*>(cc_eval_labels(h)));
  generateFeatures();
  generation++;
  uid = g_next_uid.fetch_add(1);
}

void CvFeatureEvaluator::setImage(const cv::Mat& img, uchar clsLabel, int idx) {
  CV_Assert(img.cols == winSize.width);  // features.cpp:85-87
  CV_Assert(img.rows == winSize.height);
  CV_Assert(idx >= 0 && idx < cls.rows);
  CV_Assert(img.type() == CV_8UC1);
  check(cc_eval_set_image(h, img.ptr<uchar>(0), img.rows > 1 ? (size_t)(img.ptr<uchar>(1) - img.ptr<uchar>(0)) : (size_t)img.cols, clsLabel, idx),
        "CvFeatureEvaluator::setImage");
  generation++;  // cls(idx) was written by the library: getCls() answers from host memory at once (features.cpp:88)
  lastSetIdx = idx;
  lastSetMirrored = true;
}

void CvFeatureEvaluator::setImages(const uchar* imgs, int n, int first_idx, const uchar* labels) {
  check(cc_eval_set_images(h, imgs, n, first_idx, labels), "CvFeatureEvaluator::setImages");
  generation++;
  lastSetIdx = first_idx + n - 1;
  lastSetMirrored = false;
}

void CvFeatureEvaluator::calcBatch(int fiBegin, int fiEnd, const int* sampleIdx, int nSamples, float* out) const {
  check(cc_eval_calc_batch(h, fiBegin, fiEnd, sampleIdx, nSamples, out, 0), "CvFeatureEvaluator::calcBatch");
}

void CvFeatureEvaluator::presort(int nSamples) const {
  check(cc_eval_presort(h, nSamples), "CvFeatureEvaluator::presort");
}

cc_split CvFeatureEvaluator::findBestSplit(const int* sampleIdx, int n, const double* subtreeWeights, const float* ordResponses,
                                           const int* classLabels, double nodeValue, int boostType, int splitCriteria) const {
  cc_split sp;
  check(cc_eval_find_best_split(h, sampleIdx, n, subtreeWeights, ordResponses, classLabels, nodeValue, boostType, splitCriteria, &sp,
                                nullptr, nullptr),
        "CvFeatureEvaluator::findBestSplit");
  return sp;
}

void CvFeatureEvaluator::calcBatchSorted(int fiBegin, int fiEnd, int nSamples, float* vals, void* sortedIdx, bool idx16) const {
  check(cc_eval_calc_batch_sorted(h, fiBegin, fiEnd, nSamples, vals, sortedIdx, idx16 ? 2 : 4), "CvFeatureEvaluator::calcBatchSorted");
}

float CvFeatureEvaluator::cachedValue(int featureIdx, int sampleIdx) const {
  CV_Assert(sampleIdx >= 0 && sampleIdx < maxSampleCount);
  CV_Assert(featureIdx >= 0 && featureIdx < numFeatures);
  if (lastSetMirrored && sampleIdx == lastSetIdx) {  // the window set last: the library answers from its host mirror
    float v = 0.f;
    check(cc_eval_calc(h, featureIdx, sampleIdx, &v), "CvFeatureEvaluator::operator()");
    return v;
  }
  ValueCache& c = g_cache;
  const ccamd::ValueCacheIndex::Access a = c.ix.access(featureIdx, sampleIdx, uid, generation, lastSetIdx, numFeatures);
  try {
    switch (a) {
      case ccamd::ValueCacheIndex::HIT_ROW:
        return c.row[(size_t)sampleIdx];
      case ccamd::ValueCacheIndex::HIT_LIST:
        return c.col[(size_t)c.ix.list_slot(featureIdx)];
      case ccamd::ValueCacheIndex::MISS_LIST:
        c.col.resize(c.ix.list.size());
        check(cc_eval_calc_list(h, c.ix.list.data(), (int)c.ix.list.size(), sampleIdx, c.col.data()), "CvFeatureEvaluator::operator()");
        return c.col[(size_t)c.ix.list_slot(featureIdx)];
      case ccamd::ValueCacheIndex::MISS_ROW:
      default:
        c.row.resize((size_t)maxSampleCount);
        check(cc_eval_calc_batch(h, featureIdx, featureIdx + 1, nullptr, maxSampleCount, c.row.data(), 0), "CvFeatureEvaluator::operator()");
        return c.row[(size_t)sampleIdx];
    }
  } catch (...) {
    // access() recorded the row / list as cached before its values existed (round-3 advisor finding): a failed evaluation
    // must not leave that claim behind, or the next access would hit stale or missing values
    c.ix.evaluation_failed();
    throw;
  }
}

cv::Ptr<CvFeatureEvaluator> CvFeatureEvaluator::create(int type) {  // features.cpp:91-97
  return type == CvFeatureParams::HAAR  ? cv::Ptr<CvFeatureEvaluator>(new CvHaarEvaluator)
         : type == CvFeatureParams::LBP ? cv::Ptr<CvFeatureEvaluator>(new CvLBPEvaluator)
                                        : cv::Ptr<CvFeatureEvaluator>();
}

// ---------------------------------------------------------------- Haar
void CvHaarEvaluator::init(const CvFeatureParams* _featureParams, int _maxSampleCount, cv::Size _winSize) {
  CV_Assert(_maxSampleCount > 0);  // haarfeatures.cpp:92
  CvFeatureEvaluator::init(_featureParams, _maxSampleCount, _winSize);
}

int CvHaarEvaluator::haarMode() const { return static_cast<const CvHaarFeatureParams*>(featureParams)->mode; }

void CvHaarEvaluator::generateFeatures() { numFeatures = cc_eval_num_features(h); }  // catalog built by the library

CvHaarEvaluator::Feature CvHaarEvaluator::featureAt(int fi) const {
  int32_t r[12];
  float w[3];
  int tilted = 0;
  check(cc_eval_feature_geometry(h, fi, r, w, &tilted), "CvHaarEvaluator::featureAt");
  return Feature(winSize.width + 1, tilted != 0, r[0], r[1], r[2], r[3], w[0], r[4], r[5], r[6], r[7], w[1], r[8], r[9], r[10],
                 r[11], w[2]);
}

void CvHaarEvaluator::writeFeatures(cv::FileStorage& fs, const cv::Mat& featureMap) const {  // _writeFeatures, h:82-95
  fs << FEATURES << "[";
  for (int fi = 0; fi < featureMap.cols; fi++)
    if (featureMap.at<int>(0, fi) >= 0) {
      fs << "{";
      featureAt(fi).write(fs);
      fs << "}";
    }
  fs << "]";
}

void CvHaarEvaluator::writeFeature(cv::FileStorage& fs, int fi) const {
  CV_Assert(fi < numFeatures);
  featureAt(fi).write(fs);
}

CvHaarEvaluator::Feature::Feature() : tilted(false), offset_(0) {
  for (int j = 0; j < CV_HAAR_FEATURE_MAX; j++) {
    rect[j].r = cv::Rect(0, 0, 0, 0);
    rect[j].weight = 0;
    fastRect[j].p0 = fastRect[j].p1 = fastRect[j].p2 = fastRect[j].p3 = 0;
  }
}

CvHaarEvaluator::Feature::Feature(int offset, bool _tilted, int x0, int y0, int w0, int h0, float wt0, int x1, int y1, int w1,
                                  int h1, float wt1, int x2, int y2, int w2, int h2, float wt2)
    : Feature() {
  tilted = _tilted;
  offset_ = offset;
  const int v[3][4] = {{x0, y0, w0, h0}, {x1, y1, w1, h1}, {x2, y2, w2, h2}};
  const float wt[3] = {wt0, wt1, wt2};
  for (int j = 0; j < CV_HAAR_FEATURE_MAX; j++) {
    rect[j].r = cv::Rect(v[j][0], v[j][1], v[j][2], v[j][3]);
    rect[j].weight = wt[j];
  }
  for (int j = 0; j < CV_HAAR_FEATURE_MAX; j++) {  // haarfeatures.cpp:290-308, offsets via CV_SUM/TILTED_OFFSETS
    if (rect[j].weight == 0.0F) break;
    const cv::Rect& r = rect[j].r;
    if (!tilted) {
      fastRect[j].p0 = r.x + offset * r.y;
      fastRect[j].p1 = r.x + r.width + offset * r.y;
      fastRect[j].p2 = r.x + offset * (r.y + r.height);
      fastRect[j].p3 = r.x + r.width + offset * (r.y + r.height);
    } else {
      fastRect[j].p0 = r.x + offset * r.y;
      fastRect[j].p1 = r.x - r.height + offset * (r.y + r.height);
      fastRect[j].p2 = r.x + r.width + offset * (r.y + r.width);
      fastRect[j].p3 = r.x + r.width - r.height + offset * (r.y + r.width + r.height);
    }
  }
}

float CvHaarEvaluator::Feature::calc(const cv::Mat& _sum, const cv::Mat& _tilted, size_t y) const {  // haarfeatures.h:114-122
  cc_haar_feature f;
  f.tilted = tilted ? 1 : 0;
  for (int j = 0; j < CV_HAAR_FEATURE_MAX; j++) {
    f.r[j][0] = rect[j].r.x;
    f.r[j][1] = rect[j].r.y;
    f.r[j][2] = rect[j].r.width;
    f.r[j][3] = rect[j].r.height;
    f.w[j] = rect[j].weight;
  }
  const cv::Mat& m = tilted ? _tilted : _sum;
  CV_Assert(!m.empty() && m.type() == CV_32SC1 && (int)y < m.rows);
  float out = 0;
  check(cc_haar_feature_calc(0, &f, 1, offset_, tilted ? nullptr : m.ptr<int>((int)y), tilted ? m.ptr<int>((int)y) : nullptr, 1, m.cols,
                             &out),
        "CvHaarEvaluator::Feature::calc");
  return out;
}

void CvHaarEvaluator::Feature::write(cv::FileStorage& fs) const {  // haarfeatures.cpp:311-320
  fs << CC_RECTS << "[";
  for (int ri = 0; ri < CV_HAAR_FEATURE_MAX && rect[ri].r.width != 0; ++ri)
    fs << "[:" << rect[ri].r.x << rect[ri].r.y << rect[ri].r.width << rect[ri].r.height << rect[ri].weight << "]";
  fs << "]" << CC_TILTED << tilted;
}

// ---------------------------------------------------------------- LBP
void CvLBPEvaluator::generateFeatures() { numFeatures = cc_eval_num_features(h); }

void CvLBPEvaluator::writeFeatures(cv::FileStorage& fs, const cv::Mat& featureMap) const {
  fs << FEATURES << "[";
  for (int fi = 0; fi < featureMap.cols; fi++)
    if (featureMap.at<int>(0, fi) >= 0) {
      int32_t r[12];
      check(cc_eval_feature_geometry(h, fi, r, nullptr, nullptr), "CvLBPEvaluator::writeFeatures");
      fs << "{" << CC_RECT << "[:" << r[0] << r[1] << r[2] << r[3] << "]"
         << "}";  // lbpfeatures.cpp:65-68
    }
  fs << "]";
}

// ---------------------------------------------------------------- detector
namespace ccamd {

CascadeClassifier::CascadeClassifier() : c(nullptr), d(nullptr), device(0) {}
CascadeClassifier::CascadeClassifier(const cv::String& filename, int device_) : c(nullptr), d(nullptr), device(device_) { load(filename); }
CascadeClassifier::~CascadeClassifier() {
  if (d) cc_detector_destroy(d);
  if (c) cc_cascade_destroy(c);
}

bool CascadeClassifier::load(const cv::String& filename) {
  if (d) cc_detector_destroy(d);
  if (c) cc_cascade_destroy(c);
  d = nullptr;
  c = nullptr;
  if (cc_cascade_load_xml(filename.c_str(), &c) != CC_OK) {  // like cv: load() reports failure, empty() stays true
    err = cc_last_error();
    c = nullptr;
    return false;
  }
  return true;
}

cv::Size CascadeClassifier::getOriginalWindowSize() const {
  cc_cascade_info info;
  if (!c || cc_cascade_info_get(c, &info) != CC_OK) return cv::Size();
  return cv::Size(info.win_w, info.win_h);
}

int CascadeClassifier::specialize(int nStages) {
  if (empty()) return 0;
  if (!d) check(cc_detector_create(c, device, 1, &d), "CascadeClassifier::specialize");
  if (cc_detector_specialize(d, nStages) != CC_OK) {
    err = cc_last_error();
    return 0;
  }
  return cc_detector_specialized_stages(d);
}

void CascadeClassifier::detectMultiScale(const cv::Mat& image, std::vector<cv::Rect>& objects, double scaleFactor, int minNeighbors,
                                         int /*flags*/, cv::Size minSize, cv::Size maxSize) {
  objects.clear();
  CV_Assert(scaleFactor > 1 && image.type() == CV_8UC1);  // cascadedetect.cpp: CV_Assert(scaleFactor > 1 && depth == CV_8U)
  if (empty()) return;
  if (!d) check(cc_detector_create(c, device, 1, &d), "CascadeClassifier::detectMultiScale");
  cc_detect_params p;
  p.scale_factor = scaleFactor;
  p.min_neighbors = minNeighbors;
  p.min_w = minSize.width;
  p.min_h = minSize.height;
  p.max_w = maxSize.width;
  p.max_h = maxSize.height;
  std::vector<cc_rect> out(1024);
  int n = 0;
  cc_status st = cc_detect_multiscale(d, image.data, image.cols, image.rows, image.step, &p, out.data(), (int)out.size(), &n);
  if (st == CC_ERR_BUFFER_TOO_SMALL) {
    out.resize((size_t)n);
    st = cc_detect_multiscale(d, image.data, image.cols, image.rows, image.step, &p, out.data(), (int)out.size(), &n);
  }
  check(st, "CascadeClassifier::detectMultiScale");
  for (int i = 0; i < n; i++) objects.emplace_back(out[i].x, out[i].y, out[i].width, out[i].height);
}

void CascadeClassifier::detectMultiScale(const cv::Mat& image, std::vector<cv::Rect>& objects, std::vector<int>& rejectLevels,
                                         std::vector<double>& levelWeights, double scaleFactor, int minNeighbors, int flags,
                                         cv::Size minSize, cv::Size maxSize, bool outputRejectLevels) {
  rejectLevels.clear();
  levelWeights.clear();
  if (!outputRejectLevels) {
    detectMultiScale(image, objects, scaleFactor, minNeighbors, flags, minSize, maxSize);
    return;
  }
  objects.clear();
  CV_Assert(scaleFactor > 1 && image.type() == CV_8UC1);
  if (empty()) return;
  if (!d) check(cc_detector_create(c, device, 1, &d), "CascadeClassifier::detectMultiScale");
  cc_detect_params p;
  p.scale_factor = scaleFactor;
  p.min_neighbors = minNeighbors;
  p.min_w = minSize.width;
  p.min_h = minSize.height;
  p.max_w = maxSize.width;
  p.max_h = maxSize.height;
  std::vector<cc_rect> out(1024);
  std::vector<int32_t> lv(1024);
  std::vector<double> wt(1024);
  int n = 0;
  cc_status st = cc_detect_multiscale_levels(d, image.data, image.cols, image.rows, image.step, &p, out.data(), lv.data(), wt.data(), (int)out.size(), &n);
  if (st == CC_ERR_BUFFER_TOO_SMALL) {
    out.resize((size_t)n);
    lv.resize((size_t)n);
    wt.resize((size_t)n);
    st = cc_detect_multiscale_levels(d, image.data, image.cols, image.rows, image.step, &p, out.data(), lv.data(), wt.data(), (int)out.size(), &n);
  }
  check(st, "CascadeClassifier::detectMultiScale");
  for (int i = 0; i < n; i++) {
    objects.emplace_back(out[i].x, out[i].y, out[i].width, out[i].height);
    rejectLevels.push_back(lv[i]);
    levelWeights.push_back(wt[i]);
  }
}

}  // namespace ccamd
