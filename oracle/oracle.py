"""ctypes front-end of the CPU oracle (oracle/cc_oracle.cpp). TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
(cascadeclassifier_amd/) never does. See the header of cc_oracle.cpp for what is restated and what is pinned.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import xml.etree.ElementTree as ET
from dataclasses import dataclass, field

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "build", "libcc_oracle.so")


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "cc_oracle.cpp")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


class HaarFeature(C.Structure):
    _fields_ = [("tilted", C.c_int32), ("r", (C.c_int32 * 4) * 3), ("wt", C.c_float * 3)]


HAAR_DTYPE = np.dtype([("tilted", "<i4"), ("r", "<i4", (3, 4)), ("wt", "<f4", (3,))])
assert HAAR_DTYPE.itemsize == C.sizeof(HaarFeature)


class _Cascade(C.Structure):
    _fields_ = [
        ("feature_type", C.c_int32), ("win_w", C.c_int32), ("win_h", C.c_int32), ("nstages", C.c_int32),
        ("stage_ntrees", C.c_void_p), ("stage_threshold", C.c_void_p), ("nstumps", C.c_int32),
        ("stump_feature", C.c_void_p), ("stump_threshold", C.c_void_p), ("stump_left", C.c_void_p),
        ("stump_right", C.c_void_p), ("subset_size", C.c_int32), ("stump_subset", C.c_void_p),
        ("nfeatures", C.c_int32), ("haar", C.c_void_p), ("lbp_rect", C.c_void_p),
        ("max_nodes_per_tree", C.c_int32), ("tree_nnodes", C.c_void_p), ("node_left", C.c_void_p), ("node_right", C.c_void_p),
        ("node_feature", C.c_void_p), ("node_threshold", C.c_void_p), ("node_subset", C.c_void_p), ("leaves", C.c_void_p),
    ]


SCALE_DTYPE = np.dtype([("scale", "<f4"), ("w", "<i4"), ("h", "<i4"), ("ystep", "<i4"), ("nx", "<i4"), ("ny", "<i4"),
                        ("win_w", "<i4"), ("win_h", "<i4")])

_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.orc_haar_feature_calc.restype = C.c_float
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


# ------------------------------------------------------------------ integral / resize
def integral(img: np.ndarray, sqsum_f64=False, sqsum_i32=False, tilted=False):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape
    out = {"sum": np.empty((h + 1, w + 1), np.int32)}
    if sqsum_f64:
        out["sqsum_f64"] = np.empty((h + 1, w + 1), np.float64)
    if sqsum_i32:
        out["sqsum_i32"] = np.empty((h + 1, w + 1), np.int32)
    if tilted:
        out["tilted"] = np.empty((h + 1, w + 1), np.int32)
    lib().orc_integral_u8(_p(img), w, h, w, _p(out["sum"]), _p(out.get("sqsum_f64")), _p(out.get("sqsum_i32")),
                          _p(out.get("tilted")))
    return out


def resize_linear_exact(img: np.ndarray, dw: int, dh: int) -> np.ndarray:
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape
    dst = np.empty((dh, dw), np.uint8)
    lib().orc_resize_linear_exact_u8(_p(img), w, h, w, _p(dst), dw, dh, dw)
    return dst


# ------------------------------------------------------------------ catalogs / training-side evaluator
def haar_catalog(W: int, H: int, mode: int) -> np.ndarray:
    n = lib().orc_haar_catalog(W, H, mode, None, 0)
    out = np.zeros(n, HAAR_DTYPE)
    lib().orc_haar_catalog(W, H, mode, _p(out), n)
    return out


def haar_catalog_size(W: int, H: int, mode: int) -> int:
    return lib().orc_haar_catalog(W, H, mode, None, 0)


def lbp_catalog(W: int, H: int) -> np.ndarray:
    n = lib().orc_lbp_catalog(W, H, None, 0)
    out = np.zeros((n, 4), np.int32)
    lib().orc_lbp_catalog(W, H, _p(out), n)
    return out


def make_haar_feature(tilted, rects) -> np.ndarray:
    """rects: up to 3 of (x, y, w, h, weight)."""
    f = np.zeros(1, HAAR_DTYPE)
    f["tilted"] = 1 if tilted else 0
    for j, r in enumerate(rects):
        f["r"][0, j] = r[:4]
        f["wt"][0, j] = r[4]
    return f


def haar_feature_calc(feature: np.ndarray, integral_flat: np.ndarray, step: int) -> float:
    integral_flat = np.ascontiguousarray(integral_flat, dtype=np.int32)
    return float(lib().orc_haar_feature_calc(_p(feature), _p(integral_flat), step))


def lbp_feature_calc(rect, integral_flat: np.ndarray, step: int) -> int:
    rect = np.ascontiguousarray(rect, dtype=np.int32)
    integral_flat = np.ascontiguousarray(integral_flat, dtype=np.int32)
    return int(lib().orc_lbp_feature_calc(_p(rect), _p(integral_flat), step))


def set_images(imgs: np.ndarray, want_tilted=False, want_norm=True):
    """imgs: n x H x W uint8 -> (sum[n, (W+1)(H+1)], tilted or None, normfactor or None)."""
    imgs = np.ascontiguousarray(imgs, dtype=np.uint8)
    n, H, W = imgs.shape
    cols = (W + 1) * (H + 1)
    s = np.empty((n, cols), np.int32)
    t = np.empty((n, cols), np.int32) if want_tilted else None
    nf = np.empty(n, np.float32) if want_norm else None
    lib().orc_set_images(_p(imgs), n, W, H, 1 if want_tilted else 0, _p(s), _p(t), _p(nf))
    return s, t, nf


def haar_eval_batch(feats, fi0, fi1, s, t, nf, W, H, sample_idx=None) -> np.ndarray:
    ns = len(sample_idx) if sample_idx is not None else s.shape[0]
    idx = None if sample_idx is None else np.ascontiguousarray(sample_idx, dtype=np.int32)
    out = np.empty((fi1 - fi0, ns), np.float32)
    lib().orc_haar_eval_batch(_p(feats), fi0, fi1, _p(s), _p(t), _p(nf), W, H, _p(idx), ns, _p(out))
    return out


def lbp_eval_batch(rects, fi0, fi1, s, W, H, sample_idx=None) -> np.ndarray:
    ns = len(sample_idx) if sample_idx is not None else s.shape[0]
    idx = None if sample_idx is None else np.ascontiguousarray(sample_idx, dtype=np.int32)
    out = np.empty((fi1 - fi0, ns), np.float32)
    lib().orc_lbp_eval_batch(_p(rects), fi0, fi1, _p(s), W, H, _p(idx), ns, _p(out))
    return out


# ------------------------------------------------------------------ cascade (parsed independently of the product)
@dataclass
class Cascade:
    feature_type: int  # 0 HAAR, 1 LBP
    win_w: int
    win_h: int
    stage_ntrees: np.ndarray
    stage_threshold: np.ndarray
    stump_feature: np.ndarray
    stump_threshold: np.ndarray
    stump_left: np.ndarray
    stump_right: np.ndarray
    subset_size: int
    stump_subset: np.ndarray
    haar: np.ndarray
    lbp_rect: np.ndarray
    max_nodes_per_tree: int = 1
    tree_nnodes: np.ndarray = None
    node_left: np.ndarray = None
    node_right: np.ndarray = None
    node_feature: np.ndarray = None
    node_threshold: np.ndarray = None
    node_subset: np.ndarray = None
    leaves: np.ndarray = None
    _keep: list = field(default_factory=list)

    @property
    def nstages(self):
        return len(self.stage_ntrees)

    @property
    def nstumps(self):
        return len(self.stump_feature)

    @property
    def nfeatures(self):
        return len(self.haar) if self.feature_type == 0 else len(self.lbp_rect)

    def c_struct(self) -> _Cascade:
        c = _Cascade(self.feature_type, self.win_w, self.win_h, self.nstages, _p(self.stage_ntrees).value,
                     _p(self.stage_threshold).value, self.nstumps, _p(self.stump_feature).value,
                     _p(self.stump_threshold).value, _p(self.stump_left).value, _p(self.stump_right).value,
                     self.subset_size, _p(self.stump_subset).value, self.nfeatures, _p(self.haar).value,
                     _p(self.lbp_rect).value, self.max_nodes_per_tree, _p(self.tree_nnodes).value, _p(self.node_left).value,
                     _p(self.node_right).value, _p(self.node_feature).value, _p(self.node_threshold).value,
                     _p(self.node_subset).value, _p(self.leaves).value)
        return c


def load_cascade_xml(path: str) -> Cascade:
    """Independent (ElementTree) reader of the new-format cascade XML."""
    root = ET.parse(path).getroot()
    casc = root.find("cascade")
    if casc is None:  # first child of <opencv_storage>
        casc = list(root)[0]
    assert casc.findtext("stageType").strip() == "BOOST"
    ftype = casc.findtext("featureType").strip()
    W, H = int(casc.findtext("width")), int(casc.findtext("height"))
    fp = casc.find("featureParams")
    if fp is None:
        fp = casc.find("featuhreParams")  # historical typo in stock files
    max_cat = int(fp.findtext("maxCatCount")) if fp is not None and fp.findtext("maxCatCount") else 0
    subset = (max_cat + 31) // 32 if max_cat > 0 else 0
    node_step = 3 + (subset if max_cat > 0 else 1)
    ntrees, sthr, sf, st, sl, sr, ss = [], [], [], [], [], [], []
    t_nn, n_l, n_r, n_f, n_t, n_s, lv = [], [], [], [], [], [], []
    max_nodes = 0
    for stage in casc.find("stages"):
        if stage.tag != "_":
            continue
        sthr.append(np.float32(float(stage.findtext("stageThreshold"))))
        weak = [w for w in stage.find("weakClassifiers") if w.tag == "_"]
        ntrees.append(len(weak))
        for w in weak:
            toks = w.findtext("internalNodes").split()
            leaves = [np.float32(float(v)) for v in w.findtext("leafValues").split()]
            nn = len(toks) // node_step
            max_nodes = max(max_nodes, nn)
            assert len(leaves) == nn + 1, "a tree with n nodes has n + 1 leaves"
            t_nn.append(nn)
            lv.extend(leaves)
            for k in range(nn):
                t = toks[k * node_step:(k + 1) * node_step]
                n_l.append(int(t[0]))
                n_r.append(int(t[1]))
                n_f.append(int(t[2]))
                if subset:
                    n_s.extend(int(v) for v in t[3:3 + subset])
                    n_t.append(np.float32(0))
                else:
                    n_t.append(np.float32(float(t[3])))
            # stump view (first node; meaningful when every tree is a stump)
            sf.append(int(toks[2]))
            if subset:
                ss.extend(int(v) for v in toks[3:3 + subset])
                st.append(np.float32(0))
            else:
                st.append(np.float32(float(toks[3])))
            sl.append(leaves[0])
            sr.append(leaves[1])
    feats = [f for f in casc.find("features") if f.tag == "_"]
    haar = np.zeros(len(feats) if ftype == "HAAR" else 0, HAAR_DTYPE)
    lbp = np.zeros((len(feats) if ftype == "LBP" else 0, 4), np.int32)
    for i, f in enumerate(feats):
        if ftype == "HAAR":
            rects = [r for r in f.find("rects") if r.tag == "_"]
            for j, r in enumerate(rects):
                v = r.text.split()
                haar["r"][i, j] = [int(v[0]), int(v[1]), int(v[2]), int(v[3])]
                haar["wt"][i, j] = np.float32(float(v[4]))
            haar["tilted"][i] = int(f.findtext("tilted") or 0) != 0
        else:
            lbp[i] = [int(v) for v in f.findtext("rect").split()]
    return Cascade(0 if ftype == "HAAR" else 1, W, H, np.array(ntrees, np.int32), np.array(sthr, np.float32),
                   np.array(sf, np.int32), np.array(st, np.float32), np.array(sl, np.float32), np.array(sr, np.float32),
                   subset, np.array(ss, np.int32), haar, lbp, max_nodes, np.array(t_nn, np.int32), np.array(n_l, np.int32),
                   np.array(n_r, np.int32), np.array(n_f, np.int32), np.array(n_t, np.float32), np.array(n_s, np.int32),
                   np.array(lv, np.float32))


# ------------------------------------------------------------------ detection
def scales(W0, H0, imgw, imgh, scale_factor=1.1, min_size=(0, 0), max_size=(0, 0)) -> np.ndarray:
    out = np.zeros(4096, SCALE_DTYPE)
    n = lib().orc_scales(W0, H0, imgw, imgh, C.c_double(scale_factor), min_size[0], min_size[1], max_size[0], max_size[1],
                         _p(out), 4096)
    return out[:n].copy()


@dataclass
class RawDetection:
    candidates: np.ndarray  # n x 7: scale_idx, gx, gy, x, y, w, h
    codes: np.ndarray | None
    visited: np.ndarray | None
    sums: np.ndarray | None
    n_grid_windows: int
    n_visited_windows: int


def detect_raw(c: Cascade, img: np.ndarray, scale_factor=1.1, min_size=(0, 0), max_size=(0, 0), nthreads=1,
               full=False) -> RawDetection:
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape
    cs = c.c_struct()
    ngrid = C.c_int64(0)
    nvis = C.c_int64(0)
    codes = vis = sums = None
    if full:
        sc = scales(c.win_w, c.win_h, w, h, scale_factor, min_size, max_size)
        tot = int((sc["nx"].astype(np.int64) * sc["ny"]).sum())
        codes = np.zeros(tot, np.int32)
        vis = np.zeros(tot, np.uint8)
        sums = np.zeros(tot, np.float64)
    cap = 4096
    while True:
        cand = np.zeros((cap, 7), np.int32)
        n = lib().orc_detect_raw(C.byref(cs), _p(img), w, h, w, C.c_double(scale_factor), min_size[0], min_size[1],
                                 max_size[0], max_size[1], nthreads, _p(cand), cap, _p(codes), _p(vis), _p(sums),
                                 C.byref(ngrid), C.byref(nvis))
        if n <= cap:
            break
        cap = n
    return RawDetection(cand[:n].copy(), codes, vis, sums, ngrid.value, nvis.value)


def group_rectangles(rects: np.ndarray, group_threshold: int, eps: float = 0.2) -> np.ndarray:
    rects = np.ascontiguousarray(rects, dtype=np.int32).reshape(-1, 4)
    out = np.zeros((max(len(rects), 1), 4), np.int32)
    n = lib().orc_group_rectangles(_p(rects), len(rects), group_threshold, C.c_double(eps), _p(out), len(out))
    return out[:n].copy()


def detect_multiscale(c: Cascade, img: np.ndarray, scale_factor=1.1, min_neighbors=3, min_size=(0, 0), max_size=(0, 0),
                      nthreads=1) -> np.ndarray:
    raw = detect_raw(c, img, scale_factor, min_size, max_size, nthreads)
    return group_rectangles(raw.candidates[:, 3:7], min_neighbors, 0.2)


def detect_multiscale_levels(c: Cascade, img: np.ndarray, scale_factor=1.1, min_neighbors=3, min_size=(0, 0), max_size=(0, 0), nthreads=1):
    """detectMultiScale(..., outputRejectLevels=true): accepted windows carry level = number of stages and weight = the last
    stage's sum (runAt's gypWeight); grouped with cv::groupRectangles(rects, levels, weights, ...)."""
    raw = detect_raw(c, img, scale_factor, min_size, max_size, nthreads, full=True)
    h, w = np.asarray(img).shape
    sc = scales(c.win_w, c.win_h, w, h, scale_factor, min_size, max_size)
    first = np.concatenate([[0], np.cumsum(sc["nx"].astype(np.int64) * sc["ny"])])
    cand = raw.candidates
    n = len(cand)
    idx = first[cand[:, 0]] + cand[:, 2].astype(np.int64) * sc["nx"][cand[:, 0]] + cand[:, 1] if n else np.zeros(0, np.int64)
    weights = np.ascontiguousarray(raw.sums[idx], np.float64)
    levels = np.full(n, c.nstages, np.int32)
    rects = np.ascontiguousarray(cand[:, 3:7], np.int32)
    out = np.zeros((max(n, 1), 4), np.int32)
    ol = np.zeros(max(n, 1), np.int32)
    ow = np.zeros(max(n, 1), np.float64)
    m = lib().orc_group_rectangles_levels(_p(rects), _p(levels), _p(weights), n, min_neighbors, C.c_double(0.2), _p(out), _p(ol), _p(ow), max(n, 1))
    return out[:m], ol[:m], ow[:m]


def train_predict(c: Cascade, s, t, nf, si, W, H) -> int:
    cs = c.c_struct()
    return int(lib().orc_train_predict(C.byref(cs), _p(s), _p(t), _p(nf), si, W, H))


def negmine_image(c: Cascade, img: np.ndarray, ox=0, oy=0, max_keep=64):
    """Reader stream of one background image through setImage + predict. Returns (pass flags, kept pixels, kept indices)."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape
    cs = c.c_struct()
    cap = 1 << 22
    flags = np.zeros(cap, np.uint8)
    pix = np.zeros((max(max_keep, 1), c.win_h, c.win_w), np.uint8)
    idx = np.zeros(max(max_keep, 1), np.int64)
    nk = C.c_int(0)
    lib().orc_negmine_image.restype = C.c_int64
    n = lib().orc_negmine_image(C.byref(cs), _p(img), w, h, w, int(ox), int(oy), _p(flags), C.c_int64(cap), _p(pix), _p(idx),
                                int(max_keep), C.byref(nk))
    return flags[:n].copy(), pix[:nk.value].copy(), idx[:nk.value].copy()


# ------------------------------------------------------------------ split search
SPLIT_DTYPE = np.dtype([("found", "<i4"), ("var_idx", "<i4"), ("quality", "<f4"), ("ord_c", "<f4"), ("split_point", "<i4"),
                        ("subset", "<i4", (8,))])
BOOST_DISCRETE, BOOST_REAL, BOOST_LOGIT, BOOST_GENTLE = 0, 1, 2, 3


def find_best_split(vals: np.ndarray, weights: np.ndarray, *, categorical=False, responses=None, class_labels=None,
                    node_value=0.0, boost_type=BOOST_GENTLE, split_criteria=0, tie_key=None, per_feature=False):
    """vals: [F][n] float32 values of the node's samples in node order; weights: n + 2 doubles (subtree weights)."""
    vals = np.ascontiguousarray(vals, np.float32)
    F, n = vals.shape
    weights = np.ascontiguousarray(weights, np.float64)
    assert weights.shape == (n + 2,)
    tie = np.arange(n, dtype=np.int32) if tie_key is None else np.ascontiguousarray(tie_key, np.int32)
    resp = None if responses is None else np.ascontiguousarray(responses, np.float32)
    labels = None if class_labels is None else np.ascontiguousarray(class_labels, np.int32)
    out = np.zeros(1, SPLIT_DTYPE)
    q = np.empty(F, np.float32) if per_feature else None
    pt = np.empty(F, np.int32) if per_feature else None
    lib().orc_find_best_split(_p(vals), F, n, 1 if categorical else 0, 256, _p(tie), _p(weights), _p(resp), _p(labels),
                              C.c_double(node_value), boost_type, split_criteria, _p(out), _p(q), _p(pt))
    return (out[0], q, pt) if per_feature else out[0]
