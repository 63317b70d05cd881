// Bookkeeping of the per-thread value caches behind the scalar CvFeatureEvaluator::operator()(featureIdx, sampleIdx)
// (haarfeatures.h:108-112, lbpfeatures.h:44-45 of the reference): which accesses hit, and what a miss launches. No
// device calls and no values in here, so that the policy can be unit-tested on a machine without a GPU
// (tests/cpp/test_cache_policy.cpp). The trainer calls operator() in two shapes:
//  * ROW  : one feature over many samples -- precalculate and the cache-miss paths of get_ord_var_data / get_cat_var_data
//           (o_cvcascadeboosttraindata.cpp:403-458,490-596: `for fi: for si = 0..n`). A miss evaluates the whole row fi
//           on the device (one launch per feature) and the following samples hit.
//  * LIST : many features of ONE freshly set sample -- stage prediction during negative mining
//           (cascadeclassifier.cpp:340-347 -> boost.cpp:461-477 -> getVarValue): setImage(window, idx) and then the
//           cascade's features for sample idx, window after window. The features asked for are learned; a miss evaluates
//           the whole learned list for the sample in ONE launch (cc_eval_calc_list) and the other weak classifiers hit.
// A miss is LIST-shaped only if the sample asked for is the one that was set LAST and either it was set since the
// previous miss (first feature of a prediction walk) or the previous miss was a LIST miss on the same sample (the walk
// reached a feature that is not learned yet). Round 2 decided from "same sample as the previous miss, other feature",
// which is also what the row walk of precalculate looks like at every feature's first sample: each row then paid a list
// launch as well, and the learned list filled up with catalog features.
#ifndef CCAMD_VALUE_CACHE_POLICY_HPP_
#define CCAMD_VALUE_CACHE_POLICY_HPP_
#include <cstddef>
#include <cstdint>
#include <vector>

namespace ccamd {

struct ValueCacheIndex {
  enum Access { HIT_ROW, HIT_LIST, MISS_ROW, MISS_LIST };
  static constexpr size_t kMaxLearnedFeatures = 16384;

  unsigned long long owner = 0;  // uid of the evaluator (one per init()) the cached state belongs to
  unsigned generation = 0;       // sample generation the cached VALUES belong to (the learned list survives new samples)
  int row_fi = -1;               // feature whose row is cached
  int list_si = -1;              // sample whose learned-list values are cached
  std::vector<int32_t> list;     // learned feature list, in first-seen order
  std::vector<int32_t> slot;     // feature index -> position in `list` (-1 = not in it)
  // previous miss
  bool have_miss = false, last_was_list = false;
  unsigned miss_generation = 0;
  int last_si = -1;

  void forget_everything(unsigned long long new_owner, unsigned gen) {
    *this = ValueCacheIndex();
    owner = new_owner;
    generation = gen;
  }
  int list_slot(int fi) const { return slot.empty() ? -1 : slot[(size_t)fi]; }
  // access() answers MISS_* after it has already recorded the row / the list values as cached (the caller fills them next).
  // If that evaluation fails the record must go, or the next access of the same feature / sample would be a HIT on values
  // that were never produced. The learned list itself stays (it holds feature indices, not values).
  void evaluation_failed() {
    row_fi = -1;
    list_si = -1;
    have_miss = false;
  }

  // Classifies the access (fi, si). On MISS_LIST the feature is in `list` afterwards (the caller evaluates the whole list
  // for sample si and then reads position list_slot(fi)); on MISS_ROW the caller evaluates row fi.
  Access access(int fi, int si, unsigned long long uid, unsigned gen, int last_set_idx, int num_features) {
    if (owner != uid) forget_everything(uid, gen);
    if (generation != gen) {  // samples changed: values are stale, the learned list is not
      generation = gen;
      row_fi = -1;
      list_si = -1;
    }
    if (row_fi == fi) return HIT_ROW;
    if (list_si == si && list_slot(fi) >= 0) return HIT_LIST;
    const bool fresh = !have_miss || miss_generation != gen;
    const bool list_shape = si == last_set_idx && (fresh || (last_was_list && last_si == si));
    have_miss = true;
    miss_generation = gen;
    last_si = si;
    last_was_list = list_shape;
    if (!list_shape) {
      row_fi = fi;
      return MISS_ROW;
    }
    if (slot.empty()) slot.assign((size_t)num_features, -1);
    if (slot[(size_t)fi] < 0) {
      if (list.size() >= kMaxLearnedFeatures) {  // full: start over (the cascade in training is what gets re-learned)
        for (int32_t f : list) slot[(size_t)f] = -1;
        list.clear();
      }
      slot[(size_t)fi] = (int32_t)list.size();
      list.push_back(fi);
    }
    list_si = si;
    return MISS_LIST;
  }
};

}  // namespace ccamd
#endif
