"""Python host mirror of the training-side plugin surface CvFeatureEvaluator / CvHaarEvaluator / CvLBPEvaluator
(traincascade/lib/include/traincascade_features.h:155-188) on top of the C ABI: same method names, argument meaning
and error behaviour (CV_Assert failures surface as CascadeError instead of cv::Exception)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L

HAAR, LBP, HOG = L.CC_FEATURE_HAAR, L.CC_FEATURE_LBP, L.CC_FEATURE_HOG
BASIC, CORE, ALL = L.CC_HAAR_BASIC, L.CC_HAAR_CORE, L.CC_HAAR_ALL


BOOST_DISCRETE, BOOST_REAL, BOOST_LOGIT, BOOST_GENTLE = 0, 1, 2, 3  # CvBoost types (boost.h)
SPLIT_DEFAULT, SPLIT_GINI, SPLIT_MISCLASS, SPLIT_SQERR = 0, 1, 3, 4


def _vp(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class CvFeatureParams:
    """CvFeatureParams / CvHaarFeatureParams / CvLBPFeatureParams (haarfeatures.h:31-51, lbpfeatures.h:22-26)."""

    def __init__(self, feature_type=HAAR, mode=BASIC):
        self.feature_type = feature_type
        self.mode = mode
        self.maxCatCount = 256 if feature_type == LBP else 0
        self.featSize = 1

    @staticmethod
    def create(feature_type):
        return CvFeatureParams(feature_type) if feature_type in (HAAR, LBP) else None


class CvFeatureEvaluator:
    def __init__(self, feature_type, device=0):
        self.feature_type = feature_type
        self.device = device
        self._e = C.c_void_p()
        self.winSize = None

    @staticmethod
    def create(feature_type, device=0):
        """features.cpp:91-97: unknown type -> empty Ptr (None here). HOG is outside the accelerated path."""
        return CvFeatureEvaluator(feature_type, device) if feature_type in (HAAR, LBP) else None

    def init(self, featureParams: CvFeatureParams, maxSampleCount: int, winSize):
        self._release()
        w, h = winSize
        L.check(L.lib().cc_eval_create(self.feature_type, featureParams.mode, int(w), int(h), int(maxSampleCount), self.device,
                                       C.byref(self._e)))
        self.winSize = (int(w), int(h))
        self.featureParams = featureParams
        self.maxSampleCount = int(maxSampleCount)

    def setImage(self, img, clsLabel: int, idx: int):
        img = np.ascontiguousarray(img, np.uint8)
        if img.shape != (self.winSize[1], self.winSize[0]):  # features.cpp:85-86
            raise L.CascadeError(L.CC_ERR_INVALID_ARG, f"setImage: image {img.shape[::-1]} != winSize {self.winSize}")
        L.check(L.lib().cc_eval_set_image(self._e, _vp(img), img.shape[1], int(clsLabel), int(idx)))

    def setImages(self, imgs, labels=None, first_idx=0):
        imgs = np.ascontiguousarray(imgs, np.uint8)
        if imgs.ndim != 3 or imgs.shape[1:] != (self.winSize[1], self.winSize[0]):
            raise L.CascadeError(L.CC_ERR_INVALID_ARG, "setImages: images do not match winSize")
        lab = None if labels is None else np.ascontiguousarray(labels, np.uint8)
        L.check(L.lib().cc_eval_set_images(self._e, _vp(imgs), imgs.shape[0], int(first_idx), _vp(lab)))

    def __call__(self, featureIdx: int, sampleIdx: int) -> float:
        out = C.c_float(0)
        L.check(L.lib().cc_eval_calc(self._e, int(featureIdx), int(sampleIdx), C.byref(out)))
        return out.value

    def calc_list(self, feature_idx, sample_idx: int) -> np.ndarray:
        """operator()(fi, sample_idx) for an arbitrary list of features and one stored sample, one launch (cc_eval_calc_list)."""
        fi = np.ascontiguousarray(feature_idx, np.int32)
        out = np.empty(len(fi), np.float32)
        L.check(L.lib().cc_eval_calc_list(self._e, _vp(fi), len(fi), int(sample_idx), _vp(out)))
        return out

    def calc_batch(self, fi_begin, fi_end, sample_idx=None, n_samples=None) -> np.ndarray:
        idx = None if sample_idx is None else np.ascontiguousarray(sample_idx, np.int32)
        ns = len(idx) if idx is not None else (self.maxSampleCount if n_samples is None else n_samples)
        out = np.empty((fi_end - fi_begin, ns), np.float32)
        L.check(L.lib().cc_eval_calc_batch(self._e, int(fi_begin), int(fi_end), _vp(idx), ns, _vp(out), 0))
        return out

    def calc_batch_sorted(self, fi_begin, fi_end, n_samples=None, idx_bytes=None):
        """Values and per-feature argsort of the samples (the sorted-index half of precalculate)."""
        ns = self.maxSampleCount if n_samples is None else n_samples
        if idx_bytes is None:
            idx_bytes = 2 if ns < 65536 else 4  # is_buf_16u, o_cvcascadeboosttraindata.cpp:250-251
        vals = np.empty((fi_end - fi_begin, ns), np.float32)
        idx = np.empty((fi_end - fi_begin, ns), np.uint16 if idx_bytes == 2 else np.int32)
        L.check(L.lib().cc_eval_calc_batch_sorted(self._e, int(fi_begin), int(fi_end), ns, _vp(vals), _vp(idx), idx_bytes))
        return vals, idx

    def calc_batch_device(self, fi_begin, fi_end, out_ptr, sample_idx=None, n_samples=None, pitch=0):
        """Values into device memory at out_ptr, rows `pitch` floats apart (0 = densely packed)."""
        idx = None if sample_idx is None else np.ascontiguousarray(sample_idx, np.int32)
        ns = len(idx) if idx is not None else (self.maxSampleCount if n_samples is None else n_samples)
        L.check(L.lib().cc_eval_calc_batch_device(self._e, int(fi_begin), int(fi_end), _vp(idx), ns, C.c_void_p(out_ptr), int(pitch)))

    def calc_custom_haar(self, feats, normalized=False, sample_idx=None, n_samples=None) -> np.ndarray:
        """feats: list of (tilted, [(x, y, w, h, weight), ...]) — Feature::calc on stored samples."""
        arr = (L.HaarFeatureC * len(feats))()
        for i, (tilted, rects) in enumerate(feats):
            arr[i].tilted = 1 if tilted else 0
            for j, r in enumerate(rects):
                for k in range(4):
                    arr[i].r[j][k] = int(r[k])
                arr[i].w[j] = float(r[4])
        idx = None if sample_idx is None else np.ascontiguousarray(sample_idx, np.int32)
        ns = len(idx) if idx is not None else (self.maxSampleCount if n_samples is None else n_samples)
        out = np.empty((len(feats), ns), np.float32)
        L.check(L.lib().cc_eval_calc_custom_haar(self._e, arr, len(feats), 1 if normalized else 0, _vp(idx), ns, _vp(out)))
        return out

    def predict_cascade(self, cascade, sample_idx=None, n_samples=None) -> np.ndarray:
        idx = None if sample_idx is None else np.ascontiguousarray(sample_idx, np.int32)
        ns = len(idx) if idx is not None else (self.maxSampleCount if n_samples is None else n_samples)
        out = np.empty(ns, np.uint8)
        L.check(L.lib().cc_eval_predict_cascade(self._e, cascade._c, _vp(idx), ns, _vp(out)))
        return out

    # ---- best-split search of a boosted-tree node (CvDTree::find_best_split, o_cvdtree.cpp:313-357)
    def presort(self, n_samples=None, fi_begin=0, fi_end=None):
        """Evaluate features [fi_begin, fi_end) (default: all) on stored samples [0, n_samples) and keep the sorted
        order (Haar) / the category codes (LBP) resident on the device; call again whenever the stored samples change."""
        n = self.maxSampleCount if n_samples is None else int(n_samples)
        self._presorted = (int(fi_begin), self.getNumFeatures() if fi_end is None else int(fi_end))
        L.check(L.lib().cc_eval_presort_range(self._e, self._presorted[0], self._presorted[1], n))

    def find_best_split(self, weights, *, responses=None, class_labels=None, sample_idx=None, node_value=0.0,
                        boost_type=BOOST_GENTLE, split_criteria=0, per_var=False):
        """weights: n + 2 doubles (CvBoostTree::calc_node_value's subtree weights: per sample, then the totals).
        Returns a dict (found, var_idx, quality, ord_c, split_point, subset); with per_var=True also the per-variable
        best qualities (float64) and split points."""
        w = np.ascontiguousarray(weights, np.float64)
        n = len(w) - 2
        idx = None if sample_idx is None else np.ascontiguousarray(sample_idx, np.int32)
        if idx is not None and len(idx) != n:
            raise ValueError("weights must hold len(sample_idx) + 2 values")
        resp = None if responses is None else np.ascontiguousarray(responses, np.float32)
        lab = None if class_labels is None else np.ascontiguousarray(class_labels, np.int32)
        for a in (resp, lab):
            if a is not None and len(a) != n:
                raise ValueError("responses / class_labels must hold one value per node sample")
        sp = L.Split()
        nf = self._presorted[1] - self._presorted[0] if getattr(self, "_presorted", None) else self.getNumFeatures()
        q = np.empty(nf, np.float64) if per_var else None
        pt = np.empty(nf, np.int32) if per_var else None
        L.check(L.lib().cc_eval_find_best_split(self._e, _vp(idx), n, _vp(w), _vp(resp), _vp(lab), float(node_value), int(boost_type),
                                                int(split_criteria), C.byref(sp), _vp(q), _vp(pt)))
        out = {"found": bool(sp.found), "var_idx": sp.var_idx, "quality": np.float32(sp.quality), "ord_c": np.float32(sp.ord_c),
               "split_point": sp.split_point, "subset": np.array(sp.subset[:], np.int32)}
        return (out, q, pt) if per_var else out

    def getNumFeatures(self) -> int:
        return L.lib().cc_eval_num_features(self._e)

    def getMaxCatCount(self) -> int:
        return L.lib().cc_eval_max_cat_count(self._e)

    def getFeatureSize(self) -> int:
        return L.lib().cc_eval_feature_size(self._e)

    def getCls(self, si=None):
        p = L.lib().cc_eval_labels(self._e)
        a = np.ctypeslib.as_array(p, shape=(self.maxSampleCount,))
        return a if si is None else float(a[si])

    def feature_geometry(self, fi):
        rects = np.zeros((3, 4), np.int32)
        w = np.zeros(3, np.float32)
        t = C.c_int(0)
        L.check(L.lib().cc_eval_feature_geometry(self._e, int(fi), _vp(rects), _vp(w), C.byref(t)))
        return (rects[0].copy(),) if self.feature_type == LBP else (rects, w, t.value)

    def get_sample(self, idx):
        cols = (self.winSize[0] + 1) * (self.winSize[1] + 1)
        s = np.empty(cols, np.int32)
        haar = self.feature_type == HAAR
        t = np.empty(cols, np.int32) if haar and self.featureParams.mode == ALL else None
        nf = np.empty(1, np.float32) if haar else None
        L.check(L.lib().cc_eval_get_sample(self._e, int(idx), _vp(s), _vp(t), _vp(nf)))
        return s, t, (float(nf[0]) if nf is not None else None)

    def last_kernel_ms(self) -> float:
        ms = C.c_double(0)
        L.check(L.lib().cc_eval_last_kernel_ms(self._e, C.byref(ms)))
        return ms.value

    def _release(self):
        if getattr(self, "_e", None):
            L.lib().cc_eval_destroy(self._e)
            self._e = C.c_void_p()

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass


class NegativeMiner:
    """Batched form of the negative branch of CvCascadeClassifier::fillPassedSamples (cascadeclassifier.cpp:329-357):
    one call runs the reader's whole window stream of one background image (imagestorage.cpp:57-126) through the
    trained stages on the device."""

    def __init__(self, cascade, device=0):
        self._m = C.c_void_p()
        self._cascade = cascade  # keep the model alive
        L.check(L.lib().cc_negminer_create(cascade._c, device, C.byref(self._m)))
        inf = cascade.info()
        self.win = (inf["win_w"], inf["win_h"])

    def plan(self, width, height, ox=0, oy=0):
        lw, lh, nx, ny = (np.zeros(64, np.int32) for _ in range(4))
        nl, nw = C.c_int(0), C.c_int64(0)
        L.check(L.lib().cc_negminer_plan(self._m, width, height, ox, oy, _vp(lw), _vp(lh), _vp(nx), _vp(ny), 64, C.byref(nl), C.byref(nw)))
        k = nl.value
        return {"levels": list(zip(lw[:k].tolist(), lh[:k].tolist(), nx[:k].tolist(), ny[:k].tolist())), "n_windows": nw.value}

    def run(self, img, ox=0, oy=0, max_keep=64):
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        cap = self.plan(w, h, ox, oy)["n_windows"]
        flags = np.zeros(max(cap, 1), np.uint8)
        pix = np.zeros((max(max_keep, 1), self.win[1], self.win[0]), np.uint8)
        idx = np.zeros(max(max_keep, 1), np.int64)
        nw, nk = C.c_int64(0), C.c_int(0)
        L.check(L.lib().cc_negminer_run(self._m, _vp(img), w, h, w, ox, oy, _vp(flags), cap, C.byref(nw), _vp(pix), _vp(idx),
                                        max_keep, C.byref(nk)))
        return flags[:nw.value].copy(), pix[:nk.value].copy(), idx[:nk.value].copy()

    def run_batch(self, imgs, ox=0, oy=0, max_keep=64):
        """cc_negminer_run_batch: several images of one size, one offset (consecutive images of a background set). Returns
        (flags [n_images][n_windows], kept pixels, kept indices as image * n_windows + stream index)."""
        imgs = [np.ascontiguousarray(im, np.uint8) for im in imgs]
        h, w = imgs[0].shape
        if any(im.shape != (h, w) for im in imgs):
            raise ValueError("run_batch: images of one size only")
        k = len(imgs)
        per = self.plan(w, h, ox, oy)["n_windows"]
        flags = np.zeros(max(per * k, 1), np.uint8)
        pix = np.zeros((max(max_keep, 1), self.win[1], self.win[0]), np.uint8)
        idx = np.zeros(max(max_keep, 1), np.int64)
        ptrs = (C.c_void_p * k)(*[im.ctypes.data for im in imgs])
        nw, nk = C.c_int64(0), C.c_int(0)
        L.check(L.lib().cc_negminer_run_batch(self._m, ptrs, k, w, h, w, ox, oy, _vp(flags), per * k, C.byref(nw), _vp(pix), _vp(idx),
                                              max_keep, C.byref(nk)))
        return flags[:nw.value * k].reshape(k, nw.value).copy(), pix[:nk.value].copy(), idx[:nk.value].copy()

    def __del__(self):
        try:
            if self._m:
                L.lib().cc_negminer_destroy(self._m)
                self._m = C.c_void_p()
        except Exception:
            pass
