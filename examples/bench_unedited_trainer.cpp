// The trainer's negative-mining loop as the reference writes it (CvCascadeClassifier::fillPassedSamples, negative branch,
// cascadeclassifier.cpp:329-357): ONE window at a time -- setImage(window, 0, 0) followed by predict(0), which walks the
// trained stages asking the evaluator for one feature value after the other (boost.cpp:461-477 ->
// o_cvcascadeboosttree.cpp:16-39 -> CvCascadeBoostTrainData::getVarValue -> operator()(featureIdx, sampleIdx)) --
// run UNCHANGED against the C++ adaptor (ccamd/traincascade_features.hpp), next to the batched replacement
// cc_negminer_run on the same window stream. Prints windows per second for both and checks that every window gets the
// same verdict. Usage: bench_unedited_trainer cascade.xml [n_stages_kept = all] [width height]
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <string>
#include <vector>

#include "ccamd/traincascade_features.hpp"

static void die(const char* what) {
  std::printf("%s: %s\n", what, cc_last_error());
  std::exit(2);
}

int main(int argc, char** argv) {
  if (argc < 2) {
    std::printf("usage: %s cascade.xml [stages] [width height]\n", argv[0]);
    return 2;
  }
  if (cc_device_count() <= 0) die("no HIP device");
  cc_cascade* c = nullptr;
  if (cc_cascade_load_xml(argv[1], &c) != CC_OK) die("cc_cascade_load_xml");
  cc_cascade_info inf;
  cc_cascade_info_get(c, &inf);
  if (inf.feature_type != CC_FEATURE_HAAR || inf.max_nodes_per_tree != 1) {
    std::printf("needs a Haar stump cascade\n");
    return 2;
  }
  const int32_t *first, *nweak, *feat, *rects, *tilted;
  const float *sthr, *thr, *left, *right, *weights;
  cc_cascade_stages(c, &first, &nweak, &sthr);
  cc_cascade_stumps(c, &feat, &thr, &left, &right, nullptr);
  cc_cascade_features(c, &rects, &weights, &tilted);
  const int n_stages = argc > 2 ? std::min(std::atoi(argv[2]), (int)inf.n_stages) : (int)inf.n_stages;
  const int W = argc > 4 ? std::atoi(argv[3]) : 640, H = argc > 4 ? std::atoi(argv[4]) : 480;
  const int W0 = inf.win_w, H0 = inf.win_h;

  // the evaluator the trainer would own, one sample slot like fillPassedSamples uses (idx 0 of the negatives in flight)
  CvHaarFeatureParams params(CvHaarFeatureParams::BASIC);
  cv::Ptr<CvFeatureEvaluator> eval = CvFeatureEvaluator::create(CvFeatureParams::HAAR);
  eval->init(&params, 1, cv::Size(W0, H0));
  // cascade feature -> catalog index (the trainer's weak classifiers hold catalog indices)
  std::map<std::vector<int>, int> catalog;
  for (int fi = 0; fi < eval->getNumFeatures(); fi++) {
    int32_t r[12];
    float w[3];
    int t;
    cc_eval_feature_geometry(eval->handle(), fi, r, w, &t);
    std::vector<int> key(r, r + 12);
    for (int j = 0; j < 3; j++) key.push_back((int)(w[j] * 16));
    key.push_back(t);
    catalog.emplace(key, fi);
  }
  std::vector<int> stump_fi((size_t)inf.n_weak, -1);
  for (int k = 0; k < first[n_stages - 1] + nweak[n_stages - 1]; k++) {
    const int f = feat[k];
    std::vector<int> key(rects + (size_t)f * 12, rects + (size_t)f * 12 + 12);
    for (int j = 0; j < 3; j++) {
      if (weights[(size_t)f * 3 + j] == 0.f)
        for (int q = 0; q < 4; q++) key[(size_t)j * 4 + q] = 0;
      key.push_back((int)(weights[(size_t)f * 3 + j] * 16));
    }
    key.push_back(tilted[f]);
    auto it = catalog.find(key);
    if (it == catalog.end()) {
      std::printf("stump %d uses a feature outside the BASIC catalog\n", k);
      return 2;
    }
    stump_fi[(size_t)k] = it->second;
  }

  // background image: smooth pseudo-random texture
  std::vector<uint8_t> img((size_t)W * H);
  unsigned s = 12345;
  for (int y = 0; y < H; y++)
    for (int x = 0; x < W; x++) {
      s = s * 1664525u + 1013904223u;
      img[(size_t)y * W + x] = (uint8_t)(128 + 60 * std::sin(x * 0.07) * std::cos(y * 0.05) + (int)((s >> 24) & 31) - 16);
    }

  // ---- batched path: the whole stream in one call ------------------------------------------------
  cc_cascade* trained = c;  // cc_negminer evaluates every stage of the cascade it is given: keep n_stages by truncation below
  cc_negminer* miner = nullptr;
  if (cc_negminer_create(trained, 0, &miner) != CC_OK) die("cc_negminer_create");
  int32_t lw[64], lh[64], nx[64], ny[64];
  int n_levels = 0;
  int64_t n_windows = 0;
  if (cc_negminer_plan(miner, W, H, 0, 0, lw, lh, nx, ny, 64, &n_levels, &n_windows) != CC_OK) die("cc_negminer_plan");
  std::vector<uint8_t> pass_batched((size_t)n_windows);
  int64_t nw = 0;
  int nkeep = 0;
  if (n_stages == (int)inf.n_stages) {
    cc_negminer_run(miner, img.data(), W, H, (size_t)W, 0, 0, pass_batched.data(), n_windows, &nw, nullptr, nullptr, 0, &nkeep);  // warm-up
    const auto t0 = std::chrono::steady_clock::now();
    if (cc_negminer_run(miner, img.data(), W, H, (size_t)W, 0, 0, pass_batched.data(), n_windows, &nw, nullptr, nullptr, 0, &nkeep) != CC_OK)
      die("cc_negminer_run");
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::printf("batched  (cc_negminer_run)        : %lld windows in %.3f ms = %.1f kwindows/s\n", (long long)nw, dt * 1e3, nw / dt / 1e3);
  }

  // ---- unedited path: window by window through the plugin surface -----------------------------------
  const int sx = (int)(0.5F * W0), sy = (int)(0.5F * H0);
  std::vector<uint8_t> pass_scalar;
  pass_scalar.reserve((size_t)n_windows);
  double t_scalar = 0;
  // the reader's ladder levels (NegReader::nextImg resizes on the CPU in the reference, imagestorage.cpp:57-88: not part of
  // the evaluator's boundary) are built before the clock starts
  std::vector<std::vector<uint8_t>> levels((size_t)n_levels);
  for (int l = 0; l < n_levels; l++) {
    levels[(size_t)l].resize((size_t)lw[l] * lh[l]);
    if (cc_resize_linear_exact_u8(0, img.data(), W, H, (size_t)W, levels[(size_t)l].data(), lw[l], lh[l], (size_t)lw[l]) != CC_OK) die("cc_resize");
  }
  for (int rep = 0; rep < 3; rep++) {
    pass_scalar.clear();
    const auto t0 = std::chrono::steady_clock::now();
    for (int l = 0; l < n_levels; l++) {
      std::vector<uint8_t>& level = levels[(size_t)l];
      for (int gy = 0; gy < ny[l]; gy++)
        for (int gx = 0; gx < nx[l]; gx++) {
          cv::Mat win(H0, W0, CV_8UC1, level.data() + (size_t)(gy * sy) * lw[l] + gx * sx, (size_t)lw[l]);
          eval->setImage(win, 0, 0);                           // cascadeclassifier.cpp:346
          bool passed = true;                                  // predict(0): cascadeclassifier.cpp:297-306
          for (int st = 0; st < n_stages && passed; st++) {    // boost.cpp:461-477
            double sum = 0;
            for (int k = first[st]; k < first[st] + nweak[st]; k++) {
              const float val = (*eval)(stump_fi[(size_t)k], 0);  // getVarValue
              sum += val <= thr[k] ? left[k] : right[k];          // o_cvcascadeboosttree.cpp:23-31
            }
            passed = !(sum < sthr[st]);
          }
          pass_scalar.push_back(passed ? 1 : 0);
        }
    }
    t_scalar = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::printf("unedited (setImage + operator()) %s: %zu windows in %.2f ms = %.1f kwindows/s\n", rep ? "       " : "1st run", pass_scalar.size(),
                t_scalar * 1e3, pass_scalar.size() / t_scalar / 1e3);
  }
  int diff = 0, accepted = 0;
  if (n_stages == (int)inf.n_stages) {
    if ((int64_t)pass_scalar.size() != nw) diff = -1;
    for (size_t i = 0; diff >= 0 && i < pass_scalar.size(); i++) diff += pass_scalar[i] != pass_batched[i];
    std::printf("verdicts identical to the batched path: %s (%d differ)\n", diff == 0 ? "yes" : "NO", diff);
  }
  for (uint8_t p : pass_scalar) accepted += p;
  std::printf("accepted %d of %zu\n", accepted, pass_scalar.size());
  cc_negminer_destroy(miner);
  cc_cascade_destroy(c);
  return diff == 0 ? 0 : 1;
}
