// C++ parity test of the host adaptor: the hot-path cases of the reference's own unit tests
// (traincascade/test/test_features.cpp:15-70, 150-237, 252-392, 462-560) restated against
// cascadeclassifier_amd/cpp/ccamd/traincascade_features.hpp, i.e. against the HIP kernels through the C ABI,
// plus a detectMultiScale call shaped like tools/detection/Cpp/main.cpp:42-45. Needs a GPU. Exit code 0 = all passed.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>

#include "ccamd/traincascade_features.hpp"

static int g_fail = 0, g_checks = 0;
#define CHECK(cond)                                                        \
  do {                                                                     \
    g_checks++;                                                            \
    if (!(cond)) {                                                         \
      g_fail++;                                                            \
      std::printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond);          \
    }                                                                      \
  } while (0)
#define TEST_CASE(name) std::printf("[ RUN ] %s\n", name);

// cv::integral stand-in for building KAT inputs: the library's own device integral.
static cv::Mat device_integral(const cv::Mat& img, bool want_tilted) {
  cv::Mat sum(img.rows + 1, img.cols + 1, CV_32SC1), tilted(img.rows + 1, img.cols + 1, CV_32SC1);
  if (cc_integral_u8(0, img.data, img.cols, img.rows, img.step, sum.ptr<int>(), nullptr, want_tilted ? tilted.ptr<int>() : nullptr) != CC_OK) {
    std::printf("cc_integral_u8: %s\n", cc_last_error());
    std::exit(2);
  }
  return want_tilted ? tilted : sum;
}
static cv::Mat flatten(const cv::Mat& m) { return cv::Mat(1, m.rows * m.cols, CV_32SC1, const_cast<uchar*>(m.data)); }

int main(int argc, char** argv) {
  if (cc_device_count() <= 0) {
    std::printf("no HIP device: %s\n", cc_last_error());
    return 3;
  }
  TEST_CASE("factories") {
    CHECK(CvFeatureParams::create(CvFeatureParams::HAAR) != nullptr);
    CHECK(CvFeatureParams::create(CvFeatureParams::LBP)->maxCatCount == 256);
    CHECK(CvFeatureParams::create(99) == nullptr);
    CHECK(CvFeatureEvaluator::create(CvFeatureParams::HAAR) != nullptr);
    CHECK(CvFeatureEvaluator::create(CvFeatureParams::LBP) != nullptr);
    CHECK(CvFeatureEvaluator::create(99) == nullptr);
  }
  TEST_CASE("init: catalogs non-empty, ALL > BASIC, maxSampleCount <= 0 throws") {
    CvHaarFeatureParams basic(CvHaarFeatureParams::BASIC), all(CvHaarFeatureParams::ALL);
    CvHaarEvaluator eb, ea;
    eb.init(&basic, 1, cv::Size(24, 24));
    ea.init(&all, 1, cv::Size(24, 24));
    CHECK(eb.getNumFeatures() == 162336);
    CHECK(ea.getNumFeatures() == 261600);
    CHECK(ea.getNumFeatures() > eb.getNumFeatures());
    CHECK(eb.getMaxCatCount() == 0 && eb.getFeatureSize() == 1);
    bool threw = false;
    try {
      CvHaarEvaluator bad;
      bad.init(&basic, 0, cv::Size(24, 24));
    } catch (const cv::Exception&) {
      threw = true;
    }
    CHECK(threw);
  }
  TEST_CASE("CvHaarEvaluator::operator(): returns 0 for every feature on a constant image") {
    CvHaarFeatureParams params(CvHaarFeatureParams::BASIC);
    params.maxCatCount = 0;
    params.featSize = 1;
    CvHaarEvaluator evaluator;
    evaluator.init(&params, 1, cv::Size(24, 24));
    cv::Mat constImg(24, 24, CV_8UC1, cv::Scalar(128));
    evaluator.setImage(constImg, 1, 0);
    bool allZero = true;
    for (int fi = 0; fi < evaluator.getNumFeatures(); fi += 97)
      if (evaluator(fi, 0) != 0.0f) allZero = false;
    std::vector<float> all((size_t)evaluator.getNumFeatures());
    evaluator.calcBatch(0, evaluator.getNumFeatures(), nullptr, 1, all.data());
    for (float v : all)
      if (v != 0.0f) allZero = false;
    CHECK(allZero);
    CHECK(evaluator.getNumFeatures() > 0);
  }
  TEST_CASE("CvHaarEvaluator::operator(): non-zero on a vertical step edge; labels stored") {
    CvHaarFeatureParams params(CvHaarFeatureParams::BASIC);
    CvHaarEvaluator evaluator;
    evaluator.init(&params, 2, cv::Size(24, 24));
    cv::Mat img(24, 24, CV_8UC1, cv::Scalar(0));
    img(cv::Rect(12, 0, 12, 24)).setTo(cv::Scalar(255));
    evaluator.setImage(img, 1, 0);
    evaluator.setImage(img, 0, 1);
    std::vector<float> v((size_t)evaluator.getNumFeatures() * 2);
    evaluator.calcBatch(0, evaluator.getNumFeatures(), nullptr, 2, v.data());
    bool nz = false;
    for (float x : v) nz = nz || x != 0.0f;
    CHECK(nz);
    CHECK(evaluator.getCls(0) == 1.0f && evaluator.getCls(1) == 0.0f);
    CHECK(evaluator.getCls().rows == 2 && evaluator.getCls().at<float>(0, 0) == 1.0f);
    bool threw = false;
    try {
      evaluator.setImage(cv::Mat(24, 25, CV_8UC1, cv::Scalar(0)), 1, 0);
    } catch (const cv::Exception&) {
      threw = true;
    }
    CHECK(threw);
  }
  TEST_CASE("CvHaarEvaluator::setImage: ALL mode also computes the tilted integral") {
    CvHaarFeatureParams params(CvHaarFeatureParams::ALL);
    CvHaarEvaluator evaluator;
    evaluator.init(&params, 1, cv::Size(24, 24));
    evaluator.setImage(cv::Mat(24, 24, CV_8UC1, cv::Scalar(64)), 0, 0);
    CHECK(evaluator(0, 0) == 0.0f);
    CHECK(evaluator(evaluator.getNumFeatures() - 1, 0) == 0.0f);
    CHECK(evaluator.getCls(0) == 0.0f);
  }
  TEST_CASE("CvLBPEvaluator: 255 on a constant image, < 255 on an edge, samples isolated by index") {
    CvLBPFeatureParams params;
    CvLBPEvaluator evaluator;
    evaluator.init(&params, 2, cv::Size(24, 24));
    cv::Mat constImg(24, 24, CV_8UC1, cv::Scalar(80));
    cv::Mat textImg(24, 24, CV_8UC1, cv::Scalar(0));
    textImg(cv::Rect(0, 12, 24, 12)).setTo(cv::Scalar(200));
    evaluator.setImage(constImg, 0, 0);
    evaluator.setImage(textImg, 1, 1);
    CHECK(evaluator.getNumFeatures() == 8464 && evaluator.getMaxCatCount() == 256);
    std::vector<float> v((size_t)evaluator.getNumFeatures() * 2);
    evaluator.calcBatch(0, evaluator.getNumFeatures(), nullptr, 2, v.data());
    bool all255 = true, some_less = false;
    for (int fi = 0; fi < evaluator.getNumFeatures(); fi++) {
      all255 = all255 && v[(size_t)fi * 2] == 255.0f;
      some_less = some_less || v[(size_t)fi * 2 + 1] < 255.0f;
    }
    CHECK(all255);
    CHECK(some_less);
    CHECK(evaluator(0, 0) == 255.0f);
    CHECK(evaluator.getCls(0) == 0.0f && evaluator.getCls(1) == 1.0f);
  }
  TEST_CASE("CvHaarEvaluator::Feature::calc KATs: -3200, 0, -3600, tilted 32") {
    typedef CvHaarEvaluator::Feature HaarFeature;
    {
      cv::Mat img(8, 8, CV_8UC1, cv::Scalar(0));
      img.colRange(4, 8).setTo(100);
      cv::Mat sum = device_integral(img, false);
      cv::Mat unusedTilted;
      HaarFeature feature(sum.cols, false, 0, 0, 4, 8, +1.0F, 4, 0, 4, 8, -1.0F);
      CHECK(feature.calc(flatten(sum), unusedTilted, 0) == -3200.0F);
    }
    {
      cv::Mat sum = device_integral(cv::Mat(8, 8, CV_8UC1, cv::Scalar(42)), false);
      HaarFeature feature(sum.cols, false, 0, 0, 4, 8, +1.0F, 4, 0, 4, 8, -1.0F);
      CHECK(feature.calc(flatten(sum), cv::Mat(), 0) == 0.0F);
    }
    {
      cv::Mat img(3, 9, CV_8UC1, cv::Scalar(0));
      img.colRange(3, 6).setTo(200);
      cv::Mat sum = device_integral(img, false);
      HaarFeature feature(sum.cols, false, 0, 0, 9, 3, +1.0F, 3, 0, 3, 3, -3.0F);
      CHECK(feature.calc(flatten(sum), cv::Mat(), 0) == -3600.0F);
    }
    {
      cv::Mat tilted = device_integral(cv::Mat(16, 16, CV_8UC1, cv::Scalar(1)), true);
      HaarFeature feature(tilted.cols, true, 8, 2, 4, 4, +1.0F, 0, 0, 0, 0, 0.0F);
      CHECK(feature.calc(cv::Mat(), flatten(tilted), 0) == 32.0F);
    }
  }
  TEST_CASE("writeFeatures emits XML the cascade reader accepts") {
    CvHaarFeatureParams params(CvHaarFeatureParams::BASIC);
    CvHaarEvaluator evaluator;
    evaluator.init(&params, 1, cv::Size(24, 24));
    cv::Mat featureMap(1, evaluator.getNumFeatures(), CV_32SC1, cv::Scalar(-1));
    featureMap.at<int>(0, 5) = 0;
    featureMap.at<int>(0, 16) = 1;  // an x2_y2 feature (3 rects) follows soon in the catalog order
    featureMap.at<int>(0, 6) = 2;
    cv::FileStorage fs("unused.xml", cv::FileStorage::WRITE | cv::FileStorage::MEMORY);
    fs << "cascade" << "{" << "stageType" << "BOOST" << "featureType" << "HAAR" << "height" << 24 << "width" << 24;
    fs << "featureParams" << "{";
    params.write(fs);
    fs << "}" << "stageNum" << 1 << "stages" << "[" << "{" << "maxWeakCount" << 1 << "stageThreshold" << -1.5f;
    fs << "weakClassifiers" << "[" << "{" << "internalNodes" << "[:" << 0 << -1 << 2 << 4.5e-03f << "]";
    fs << "leafValues" << "[:" << -1.0f << 1.0f << "]" << "}" << "]" << "}" << "]";
    evaluator.writeFeatures(fs, featureMap);
    fs << "}";
    const std::string xml = fs.releaseAndGetString();
    cc_cascade* c = nullptr;
    const cc_status st = cc_cascade_load_xml_mem(xml.data(), xml.size(), &c);
    if (st != CC_OK) std::printf("%s\n%s\n", cc_last_error(), xml.c_str());
    CHECK(st == CC_OK);
    if (c) {
      cc_cascade_info info;
      cc_cascade_info_get(c, &info);
      CHECK(info.n_features == 3 && info.n_stages == 1 && info.feature_type == CC_FEATURE_HAAR && info.win_w == 24);
      cc_cascade_destroy(c);
    }
  }
  TEST_CASE("presort + findBestSplit: a Gentle-AdaBoost stump separates left-bright from right-bright windows") {
    CvHaarFeatureParams params(CvHaarFeatureParams::BASIC);
    CvHaarEvaluator evaluator;
    const int n = 40;
    evaluator.init(&params, n, cv::Size(24, 24));
    std::vector<float> resp(n);
    std::vector<double> w(n + 2, 0.0);
    for (int i = 0; i < n; i++) {
      cv::Mat img(24, 24, CV_8UC1, cv::Scalar(40));
      const bool positive = i % 2 == 0;
      for (int y = 0; y < 24; y++)
        for (int x = 0; x < 12; x++) img.at<uchar>(y, positive ? x : x + 12) = (uchar)(200 + (i * 7 + y) % 17);
      evaluator.setImage(img, positive ? 1 : 0, i);
      resp[i] = positive ? 1.f : -1.f;
      w[i] = 1.0 / n;
      w[n] += w[i];
    }
    evaluator.presort(n);
    double value = 0;
    for (int i = 0; i < n; i++) value += resp[i] * w[i];
    value *= 1. / w[n];
    const cc_split sp = evaluator.findBestSplit(nullptr, n, w.data(), resp.data(), nullptr, value, /*GENTLE*/ 3, /*DEFAULT*/ 0);
    CHECK(sp.found == 1 && sp.quality > 0 && sp.var_idx >= 0 && sp.var_idx < evaluator.getNumFeatures());
    bool separates = true;
    const bool pos_left = evaluator(sp.var_idx, 0) <= sp.ord_c;
    for (int i = 0; i < n; i++) separates = separates && ((evaluator(sp.var_idx, i) <= sp.ord_c) == (i % 2 == 0 ? pos_left : !pos_left));
    CHECK(separates);
    CHECK(sp.split_point == n / 2 - 1);
  }
  if (argc > 1) {
    TEST_CASE("detection tool call shape: CascadeClassifier(file); detectMultiScale(gray, objects, 4, 50)");
    ccamd::CascadeClassifier cascade((std::string(argv[1])));
    CHECK(!cascade.empty());
    CHECK(cascade.getOriginalWindowSize() == cv::Size(24, 24));
    cv::Mat gray(240, 320, CV_8UC1, cv::Scalar(0));
    for (int y = 0; y < gray.rows; y++)
      for (int x = 0; x < gray.cols; x++) gray.at<uchar>(y, x) = (uchar)((y * 7 + x * 13) & 0xFF);  // test_integration.cpp:59-64
    std::vector<cv::Rect> objects;
    cascade.detectMultiScale(gray, objects, 4, 50);  // tools/detection/Cpp/main.cpp:45
    CHECK(objects.empty());
    cascade.detectMultiScale(gray, objects, 1.1, 0);
    std::printf("        ungrouped candidates on the synthetic pattern: %zu\n", objects.size());
    {
      std::vector<cv::Rect> lobj;
      std::vector<int> levels;
      std::vector<double> weights;
      cascade.detectMultiScale(gray, lobj, levels, weights, 1.1, 0, 0, cv::Size(), cv::Size(), true);
      CHECK(lobj.size() == objects.size() && levels.size() == lobj.size() && weights.size() == lobj.size());
      bool all_last = true;
      for (int l : levels) all_last = all_last && l == 25;  // accepted windows passed all 25 stages
      CHECK(all_last);
    }
    const int spec = cascade.specialize(3);  // 0 where libhiprtc is missing; results must not change either way
    std::vector<cv::Rect> again;
    cascade.detectMultiScale(gray, again, 1.1, 0);
    CHECK(spec >= 0 && again.size() == objects.size());
    bool same = again.size() == objects.size();
    for (size_t i = 0; same && i < again.size(); i++) same = again[i] == objects[i];
    CHECK(same);
    ccamd::CascadeClassifier missing("/nonexistent.xml");
    CHECK(missing.empty());
    bool threw = false;
    try {
      cascade.detectMultiScale(gray, objects, 1.0, 3);
    } catch (const cv::Exception&) {
      threw = true;
    }
    CHECK(threw);
  }
  std::printf("%d checks, %d failed\n", g_checks, g_fail);
  return g_fail ? 1 : 0;
}
