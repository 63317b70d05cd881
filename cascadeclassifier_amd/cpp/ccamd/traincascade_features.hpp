// C++ host adaptor: the reference's feature-evaluator plugin surface (same class names, virtuals, argument meaning and
// error behaviour as traincascade/lib/include/traincascade_features.h:131-188, haarfeatures.h:31-122,
// lbpfeatures.h:22-83) implemented on top of the C ABI of include/cascadeclassifier_amd.h, i.e. on the HIP kernels.
// A trainer built against the reference headers can switch to these by changing the include path: CvCascadeClassifier
// keeps owning a cv::Ptr<CvFeatureEvaluator>, CvCascadeBoostTrainData keeps calling operator()(featureIdx, sampleIdx).
#pragma once

#include <mutex>
#include <string>
#include <vector>

#include "../../../include/cascadeclassifier_amd.h"
#include "cv_compat.hpp"

#define FEATURES "features"
#define CC_MAX_CAT_COUNT "maxCatCount"
#define CC_FEATURE_SIZE "featSize"
#define CC_MODE "mode"
#define CC_MODE_BASIC "BASIC"
#define CC_MODE_CORE "CORE"
#define CC_MODE_ALL "ALL"
#define CC_RECTS "rects"
#define CC_TILTED "tilted"
#define CC_RECT "rect"
#define CV_HAAR_FEATURE_MAX 3

#define CC_FEATURE_PARAMS "featureParams"  // cascadeclassifier.h:58
#define HFP_NAME "haarFeatureParams"       // haarfeatures.h:21
#define LBPF_NAME "lbpFeatureParams"       // lbpfeatures.h:18

// traincascade_features.h:105-122, features.cpp:27-32: the polymorphic base every parameter struct of the trainer derives
// from (the driver iterates them for params.xml and the command line: traincascade.cpp:59-81,141, cascadeclassifier.cpp:200,359-400)
class CvParams {
 public:
  CvParams();
  virtual ~CvParams() {}
  virtual void write(cv::FileStorage& fs) const = 0;
  virtual bool read(const cv::FileNode& node) = 0;
  virtual void printDefaults() const;
  virtual void printAttrs() const;
  virtual bool scanAttr(const std::string prmName, const std::string val);
  std::string name;
};

class CvFeatureParams : public CvParams {  // traincascade_features.h:131-144, features.cpp:37-68
 public:
  enum { HAAR = 0, LBP = 1, HOG = 2 };
  CvFeatureParams();
  virtual void init(const CvFeatureParams& fp);
  void write(cv::FileStorage& fs) const override;
  bool read(const cv::FileNode& node) override;
  static cv::Ptr<CvFeatureParams> create(int featureType);
  int maxCatCount;
  int featSize;
};

class CvHaarFeatureParams : public CvFeatureParams {  // haarfeatures.h:31-51, haarfeatures.cpp:12-85
 public:
  enum { BASIC = 0, CORE = 1, ALL = 2 };
  CvHaarFeatureParams();
  CvHaarFeatureParams(int _mode);
  void init(const CvFeatureParams& fp) override;
  void write(cv::FileStorage& fs) const override;
  bool read(const cv::FileNode& node) override;
  void printDefaults() const override;
  void printAttrs() const override;
  bool scanAttr(const std::string prm, const std::string val) override;
  int mode;
};

struct CvLBPFeatureParams : CvFeatureParams {  // lbpfeatures.h:22-26, lbpfeatures.cpp:9-13
  CvLBPFeatureParams();
};

class CvFeatureEvaluator {  // traincascade_features.h:155-188
 public:
  CvFeatureEvaluator();
  virtual ~CvFeatureEvaluator();
  virtual void init(const CvFeatureParams* _featureParams, int _maxSampleCount, cv::Size _winSize);
  virtual void setImage(const cv::Mat& img, uchar clsLabel, int idx);
  virtual void writeFeatures(cv::FileStorage& fs, const cv::Mat& featureMap) const = 0;
  virtual float operator()(int featureIdx, int sampleIdx) const = 0;
  static cv::Ptr<CvFeatureEvaluator> create(int type);

  int getNumFeatures() const { return numFeatures; }
  int getMaxCatCount() const { return featureParams->maxCatCount; }
  int getFeatureSize() const { return featureParams->featSize; }
  const cv::Mat& getCls() const { return cls; }
  float getCls(int si) const { return cls.at<float>(si, 0); }

  // ---- batched fast paths of the MI355X implementation (what CvCascadeBoostTrainData::precalculate and
  //      fillPassedSamples should call instead of looping over operator() / setImage) ----------------------------
  // n images of winSize, densely packed, stored at first_idx..; labels may be NULL.
  void setImages(const uchar* imgs, int n, int first_idx, const uchar* labels);
  // out[(fi - fiBegin) * nSamples + s] = (*this)(fi, sampleIdx ? sampleIdx[s] : s)
  void calcBatch(int fiBegin, int fiEnd, const int* sampleIdx, int nSamples, float* out) const;
  // values (optional) + per-feature argsort of samples 0..nSamples-1: the rows precalculate() stores in `buf`
  // (unsigned short when sample_count < 65536, else int; o_cvcascadeboosttraindata.cpp:250-251,490-556)
  void calcBatchSorted(int fiBegin, int fiEnd, int nSamples, float* vals, void* sortedIdx, bool idx16) const;
  // Node split search on the device (CvDTree::find_best_split, o_cvdtree.cpp:345-357): presort() once per stage after
  // the samples are set, then findBestSplit() per tree node with the node's sample slots, CvBoostTree::calc_node_value's
  // subtree weights (n + 2 doubles) and either the ordered responses (LOGIT / GENTLE) or the 0/1 class labels.
  void presort(int nSamples) const;
  cc_split findBestSplit(const int* sampleIdx, int n, const double* subtreeWeights, const float* ordResponses,
                         const int* classLabels, double nodeValue, int boostType, int splitCriteria) const;
  cc_evaluator* handle() const { return h; }  // for direct C-ABI calls (the library itself sends queued setImage() images first)

 protected:
  virtual void generateFeatures() = 0;
  virtual int featureType() const = 0;
  virtual int haarMode() const { return 0; }
  float cachedValue(int featureIdx, int sampleIdx) const;

  int npos, nneg;
  int numFeatures;
  cv::Size winSize;
  CvFeatureParams* featureParams;
  cv::Mat cls;
  cc_evaluator* h;
  int maxSampleCount;
  unsigned generation;  // bumped by every setImage: invalidates cached feature values
  unsigned long long uid;  // unique per init(): the per-thread value caches belong to one initialised evaluator
  int lastSetIdx;          // sample index of the most recent setImage (a prediction walk asks for that sample)

  // setImage() is cc_eval_set_image: the library queues the window (it reaches the device in runs of consecutive
  // indices when something first reads stored samples there -- the positives / negatives of a stage are set one by one,
  // cascadeclassifier.cpp:329-357) and mirrors the window set LAST on the host, so that the prediction walk right behind
  // a setImage (cascadeclassifier.cpp:346-347) is answered without a launch. `lastSetMirrored` says the sample set last
  // went through that call (setImages() batches do not leave a mirror).
  bool lastSetMirrored;
};

class CvHaarEvaluator : public CvFeatureEvaluator {  // haarfeatures.h:61-106
 public:
  void init(const CvFeatureParams* _featureParams, int _maxSampleCount, cv::Size _winSize) override;
  float operator()(int featureIdx, int sampleIdx) const override { return cachedValue(featureIdx, sampleIdx); }
  void writeFeatures(cv::FileStorage& fs, const cv::Mat& featureMap) const override;
  void writeFeature(cv::FileStorage& fs, int fi) const;  // for old file format

  class Feature {
   public:
    Feature();
    Feature(int offset, bool _tilted, int x0, int y0, int w0, int h0, float wt0, int x1, int y1, int w1, int h1, float wt1,
            int x2 = 0, int y2 = 0, int w2 = 0, int h2 = 0, float wt2 = 0.0F);
    // Un-normalised response on a flattened integral image held by the caller (row `y` of `sum` / `tilted`), computed
    // on the device: the shape of the reference's Feature::calc known-answer tests (test_features.cpp:462-560).
    float calc(const cv::Mat& sum, const cv::Mat& tilted, size_t y) const;
    void write(cv::FileStorage& fs) const;
    bool tilted;
    struct {
      cv::Rect r;
      float weight;
    } rect[CV_HAAR_FEATURE_MAX];
    struct {
      int p0, p1, p2, p3;
    } fastRect[CV_HAAR_FEATURE_MAX];
    int offset_;
  };

 protected:
  void generateFeatures() override;
  int featureType() const override { return CvFeatureParams::HAAR; }
  int haarMode() const override;
  Feature featureAt(int fi) const;
};

class CvLBPEvaluator : public CvFeatureEvaluator {  // lbpfeatures.h:37-68
 public:
  float operator()(int featureIdx, int sampleIdx) const override { return cachedValue(featureIdx, sampleIdx); }
  void writeFeatures(cv::FileStorage& fs, const cv::Mat& featureMap) const override;

 protected:
  void generateFeatures() override;
  int featureType() const override { return CvFeatureParams::LBP; }
};

namespace ccamd {

// cv::CascadeClassifier-shaped detector (the calls tools/detection/Cpp/main.cpp:42-45 makes).
class CascadeClassifier {
 public:
  CascadeClassifier();
  explicit CascadeClassifier(const cv::String& filename, int device = 0);
  ~CascadeClassifier();
  CascadeClassifier(const CascadeClassifier&) = delete;
  CascadeClassifier& operator=(const CascadeClassifier&) = delete;
  bool load(const cv::String& filename);
  bool empty() const { return c == nullptr; }
  // image: CV_8UC1. Same defaults as cv::CascadeClassifier::detectMultiScale.
  void detectMultiScale(const cv::Mat& image, std::vector<cv::Rect>& objects, double scaleFactor = 1.1, int minNeighbors = 3,
                        int flags = 0, cv::Size minSize = cv::Size(), cv::Size maxSize = cv::Size());
  // the outputRejectLevels overload (levels = number of stages, weights = the last stage's sum of the group's best window)
  void detectMultiScale(const cv::Mat& image, std::vector<cv::Rect>& objects, std::vector<int>& rejectLevels,
                        std::vector<double>& levelWeights, double scaleFactor = 1.1, int minNeighbors = 3, int flags = 0,
                        cv::Size minSize = cv::Size(), cv::Size maxSize = cv::Size(), bool outputRejectLevels = false);
  cv::Size getOriginalWindowSize() const;
  // Optional, once per loaded cascade: compile its first stages into the cascade kernel (cc_detector_specialize).
  // Returns the number of stages in effect; 0 if the cascade or the installation does not support it (lastError()).
  int specialize(int nStages = 7);
  const std::string& lastError() const { return err; }

 private:
  cc_cascade* c;
  cc_detector* d;
  int device;
  std::string err;
};

}  // namespace ccamd
