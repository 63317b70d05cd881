// CPU test of the value-cache policy behind CvFeatureEvaluator::operator() (ccamd/value_cache_policy.hpp): replays the
// trainer's two access shapes -- the row walk of CvCascadeBoostTrainData::precalculate
// (o_cvcascadeboosttraindata.cpp:502-504, 539-541: for fi: for si = 0..n) and the prediction walk of negative mining
// (cascadeclassifier.cpp:340-347: setImage(window, idx), then the cascade's features for sample idx) -- and counts what
// a device would be asked to do. No GPU needed.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "ccamd/value_cache_policy.hpp"

using ccamd::ValueCacheIndex;

static int failures = 0;
#define EXPECT(cond)                                                     \
  do {                                                                   \
    if (!(cond)) {                                                       \
      std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond);      \
      failures++;                                                        \
    }                                                                    \
  } while (0)

struct Counts {
  long row_launches = 0, list_launches = 0, hits = 0;
  size_t longest_list = 0;
};

int main() {
  const int F = 40000, N = 300;  // catalog size, samples
  const unsigned long long uid = 7;
  unsigned gen = 1;
  int last_set = -1;
  ValueCacheIndex ix;
  Counts c;
  auto touch = [&](int fi, int si) {
    switch (ix.access(fi, si, uid, gen, last_set, F)) {
      case ValueCacheIndex::MISS_ROW: c.row_launches++; break;
      case ValueCacheIndex::MISS_LIST:
        c.list_launches++;
        if (ix.list.size() > c.longest_list) c.longest_list = ix.list.size();
        break;
      default: c.hits++; break;
    }
  };
  // 1. the stage's samples are set one by one (fillPassedSamples), then precalculate walks rows
  for (int si = 0; si < N; si++) {
    gen++;
    last_set = si;
  }
  const int rows = 20000;  // more rows than the learned list could hold
  for (int fi = 0; fi < rows; fi++)
    for (int si = 0; si < N; si++) touch(fi, si);
  EXPECT(c.row_launches == rows);   // one launch per feature ...
  EXPECT(c.list_launches == 0);     // ... and not one list launch: the first sample of a row is not a prediction walk
  EXPECT(ix.list.empty());          // row misses teach the list nothing
  EXPECT(c.hits == (long)rows * (N - 1));
  // 2. negative mining: window after window into slot `slot`, each followed by the cascade's weak classifiers
  std::vector<int> cascade;
  for (int k = 0; k < 120; k++) cascade.push_back((k * 331 + 17) % F);
  c = Counts();
  const int slot = 123, windows = 500;
  for (int w = 0; w < windows; w++) {
    gen++;
    last_set = slot;
    for (int fi : cascade) touch(fi, slot);
  }
  // the first window learns the list feature by feature; every later window is ONE list launch
  EXPECT(c.row_launches == 0);
  EXPECT(c.list_launches == (long)cascade.size() + (windows - 1));
  EXPECT(c.longest_list == cascade.size());
  // 3. a new stage was trained: precalculate again (rows), then mining with a longer cascade
  for (int si = 0; si < N; si++) {
    gen++;
    last_set = si;
  }
  c = Counts();
  for (int fi = 100; fi < 140; fi++)
    for (int si = 0; si < N; si++) touch(fi, si);
  EXPECT(c.row_launches == 40 && c.list_launches == 0);
  EXPECT(ix.list.size() == cascade.size());  // the learned list survived the row walk untouched
  for (int k = 0; k < 30; k++) cascade.push_back((k * 977 + 5) % F);
  c = Counts();
  for (int w = 0; w < 50; w++) {
    gen++;
    last_set = slot;
    for (int fi : cascade) touch(fi, slot);
  }
  EXPECT(c.row_launches == 0);
  EXPECT(c.list_launches == 1 + 30 + 49);  // first window: the known list at once, then the 30 new features one by one
  // 4. the list never grows past its bound: a cascade-like walk over more features than it holds starts over
  c = Counts();
  gen++;
  last_set = 5;
  for (int fi = 0; fi < (int)ValueCacheIndex::kMaxLearnedFeatures + 10; fi++) touch(20000 + (fi % 19000), 5);
  EXPECT(ix.list.size() <= ValueCacheIndex::kMaxLearnedFeatures);
  // 5. another evaluator on the same thread forgets everything
  EXPECT(ix.access(1, 0, uid + 1, 1, 3, F) == ValueCacheIndex::MISS_ROW);
  EXPECT(ix.list.empty());
  // 6. a miss whose evaluation fails leaves no cached claim behind (round-3 advisor finding): the next access of the same
  //    feature / sample is a miss again, for rows and for lists, and the learned list keeps its features
  {
    ValueCacheIndex jx;
    EXPECT(jx.access(7, 0, 99, 1, /*last_set=*/-1, F) == ValueCacheIndex::MISS_ROW);
    jx.evaluation_failed();
    EXPECT(jx.access(7, 1, 99, 1, -1, F) == ValueCacheIndex::MISS_ROW);  // without the hook this would be HIT_ROW on an empty row
    EXPECT(jx.access(7, 2, 99, 1, -1, F) == ValueCacheIndex::HIT_ROW);
    EXPECT(jx.access(11, 4, 99, 2, /*last_set=*/4, F) == ValueCacheIndex::MISS_LIST);
    jx.evaluation_failed();
    EXPECT(jx.access(11, 4, 99, 2, 4, F) == ValueCacheIndex::MISS_LIST);  // not HIT_LIST on values that were never produced
    EXPECT(jx.list.size() == 1 && jx.list[0] == 11);
    EXPECT(jx.access(11, 4, 99, 2, 4, F) == ValueCacheIndex::HIT_LIST);
  }
  if (failures) return 1;
  std::printf("test_cache_policy OK\n");
  return 0;
}
