"""Python host mirror of the detection boundary: a CascadeClassifier with the call shape of cv2.CascadeClassifier as
the reference's Python tool uses it (tools/detection/Python/detect.py:16,22), on top of the C ABI."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib as L


def _params(scale_factor, min_neighbors, min_size, max_size) -> L.DetectParams:
    mn = min_size or (0, 0)
    mx = max_size or (0, 0)
    return L.DetectParams(float(scale_factor), int(min_neighbors), int(mn[0]), int(mn[1]), int(mx[0]), int(mx[1]))


def _vp(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _view(ptr, ctype, n):
    if not ptr or n == 0:
        return np.zeros(0, np.dtype(ctype))
    return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), shape=(n,)).copy()


@dataclass
class CascadeModel:
    info: dict
    stage_first: np.ndarray
    stage_ntrees: np.ndarray
    stage_threshold: np.ndarray
    stump_feature: np.ndarray | None
    stump_threshold: np.ndarray | None
    stump_left: np.ndarray | None
    stump_right: np.ndarray | None
    stump_subsets: np.ndarray | None
    rects: np.ndarray
    weights: np.ndarray | None
    tilted: np.ndarray | None


def scale_plan(win_w, win_h, width, height, scale_factor=1.1, min_size=None, max_size=None) -> np.ndarray:
    p = _params(scale_factor, 3, min_size, max_size)
    n = C.c_int(0)
    buf = (L.ScaleInfo * 4096)()
    L.check(L.lib().cc_scale_plan(win_w, win_h, width, height, C.byref(p), buf, 4096, C.byref(n)))
    dt = np.dtype([("scale", "<f4"), ("w", "<i4"), ("h", "<i4"), ("ystep", "<i4"), ("nx", "<i4"), ("ny", "<i4"),
                   ("win_w", "<i4"), ("win_h", "<i4")])
    return np.frombuffer(bytes(buf), dt, n.value).copy()


def group_rectangles(rects, group_threshold, eps=0.2) -> np.ndarray:
    rects = np.ascontiguousarray(rects, np.int32).reshape(-1, 4)
    out = np.zeros((max(len(rects), 1), 4), np.int32)
    n = C.c_int(0)
    L.check(L.lib().cc_group_rectangles(_vp(rects), len(rects), int(group_threshold), float(eps), _vp(out), len(out), C.byref(n)))
    return out[:n.value].copy()


def integral(img, device=0, sqsum=False, tilted=False):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    out = {"sum": np.empty((h + 1, w + 1), np.int32)}
    if sqsum:
        out["sqsum"] = np.empty((h + 1, w + 1), np.int32)
    if tilted:
        out["tilted"] = np.empty((h + 1, w + 1), np.int32)
    L.check(L.lib().cc_integral_u8(device, _vp(img), w, h, w, _vp(out["sum"]), _vp(out.get("sqsum")), _vp(out.get("tilted"))))
    return out


def resize_linear_exact(img, dw, dh, device=0):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    dst = np.empty((dh, dw), np.uint8)
    L.check(L.lib().cc_resize_linear_exact_u8(device, _vp(img), w, h, w, _vp(dst), dw, dh, dw))
    return dst


class CascadeClassifier:
    """cv2.CascadeClassifier-shaped front end. load()/empty()/detectMultiScale() as the reference's tools call them."""

    def __init__(self, filename: str | None = None, device: int = 0, max_batch: int = 1):
        self._c = C.c_void_p()
        self._d = C.c_void_p()
        self.device = device
        self.max_batch = max_batch
        if filename is not None and not self.load(filename):
            pass  # like cv2: constructor does not throw; empty() reports it

    # -- model ----------------------------------------------------------------------------------
    def load(self, filename: str) -> bool:
        self._release()
        st = L.lib().cc_cascade_load_xml(filename.encode(), C.byref(self._c))
        if st != L.CC_OK:
            self.load_error = L.lib().cc_last_error().decode()
            self._c = C.c_void_p()
            return False
        return True

    def load_from_string(self, text: str | bytes) -> bool:
        self._release()
        b = text.encode() if isinstance(text, str) else text
        st = L.lib().cc_cascade_load_xml_mem(b, len(b), C.byref(self._c))
        if st != L.CC_OK:
            self.load_error = L.lib().cc_last_error().decode()
            self._c = C.c_void_p()
            return False
        return True

    def empty(self) -> bool:
        return not self._c

    def save(self, filename: str, baseFormat: bool = False):
        """Write the model back as a cascade.xml: the new-format layout of CvCascadeClassifier::save, or with
        baseFormat=True its legacy "opencv-haar-classifier" layout (cascadeclassifier.cpp:439-531; Haar only)."""
        fn = L.lib().cc_cascade_save_xml_legacy if baseFormat else L.lib().cc_cascade_save_xml
        L.check(fn(self._c, filename.encode()))

    def info(self) -> dict:
        ci = L.CascadeInfo()
        L.check(L.lib().cc_cascade_info_get(self._c, C.byref(ci)))
        return {n: getattr(ci, n) for n, _ in ci._fields_}

    def model(self) -> CascadeModel:
        inf = self.info()
        a, b, c = C.c_void_p(), C.c_void_p(), C.c_void_p()
        L.check(L.lib().cc_cascade_stages(self._c, C.byref(a), C.byref(b), C.byref(c)))
        ns, nw, nf = inf["n_stages"], inf["n_weak"], inf["n_features"]
        sf, sn, sthr = _view(a, C.c_int32, ns), _view(b, C.c_int32, ns), _view(c, C.c_float, ns)
        stump = [None] * 5
        if inf["max_nodes_per_tree"] == 1:
            p = [C.c_void_p() for _ in range(5)]
            L.check(L.lib().cc_cascade_stumps(self._c, *[C.byref(x) for x in p]))
            stump = [_view(p[0], C.c_int32, nw), _view(p[1], C.c_float, nw), _view(p[2], C.c_float, nw),
                     _view(p[3], C.c_float, nw),
                     _view(p[4], C.c_int32, nw * inf["subset_size"]).reshape(nw, -1) if inf["subset_size"] else None]
        r, w, t = C.c_void_p(), C.c_void_p(), C.c_void_p()
        L.check(L.lib().cc_cascade_features(self._c, C.byref(r), C.byref(w), C.byref(t)))
        if inf["feature_type"] == L.CC_FEATURE_HAAR:
            rects = _view(r, C.c_int32, nf * 12).reshape(nf, 3, 4)
            weights = _view(w, C.c_float, nf * 3).reshape(nf, 3)
            tilted = _view(t, C.c_int32, nf)
        else:
            rects, weights, tilted = _view(r, C.c_int32, nf * 4).reshape(nf, 4), None, None
        return CascadeModel(inf, sf, sn, sthr, *stump, rects, weights, tilted)

    # -- detector -------------------------------------------------------------------------------
    def _detector(self):
        if not self._c:
            raise L.CascadeError(L.CC_ERR_INVALID_ARG, "CascadeClassifier is empty")
        if not self._d:
            L.check(L.lib().cc_detector_create(self._c, self.device, self.max_batch, C.byref(self._d)))
        return self._d

    def set_stream(self, hip_stream: int | None):
        L.check(L.lib().cc_detector_set_stream(self._detector(), C.c_void_p(hip_stream or 0)))

    def detectMultiScale(self, image, scaleFactor=1.1, minNeighbors=3, flags=0, minSize=None, maxSize=None) -> np.ndarray:
        """image: HxW uint8 (gray). Returns an (n, 4) int32 array of (x, y, w, h) like cv2 does."""
        image = np.ascontiguousarray(image, np.uint8)
        if image.ndim != 2:
            raise L.CascadeError(L.CC_ERR_INVALID_ARG, "detectMultiScale expects a single-channel 8-bit image")
        h, w = image.shape
        p = _params(scaleFactor, minNeighbors, minSize, maxSize)
        cap = 1024
        while True:
            out = np.zeros((cap, 4), np.int32)
            n = C.c_int(0)
            st = L.lib().cc_detect_multiscale(self._detector(), _vp(image), w, h, w, C.byref(p), _vp(out), cap, C.byref(n))
            if st == L.CC_ERR_BUFFER_TOO_SMALL:
                cap = n.value
                continue
            L.check(st)
            return out[:n.value].copy()

    def detect_batch(self, frames, scaleFactor=1.1, minNeighbors=3, minSize=None, maxSize=None, device_ptr=None,
                     shape=None, row_stride=None, frame_stride=None):
        """frames: (n, H, W) uint8 numpy array in host memory, or device_ptr + shape=(n,H,W) for frames already in HBM.
        Returns a list of (k_i, 4) arrays."""
        p = _params(scaleFactor, minNeighbors, minSize, maxSize)
        if device_ptr is None:
            frames = np.ascontiguousarray(frames, np.uint8)
            n, h, w = frames.shape
            ptr, on_dev, rs, fs = _vp(frames), 0, w, w * h
        else:
            n, h, w = shape
            ptr, on_dev = C.c_void_p(device_ptr), 1
            rs = row_stride or w
            fs = frame_stride or rs * h
        cap = max(256 * n, 1024)
        while True:
            out = np.zeros((cap, 4), np.int32)
            offs = np.zeros(n + 1, np.int32)
            st = L.lib().cc_detect_batch(self._detector(), ptr, on_dev, n, w, h, rs, fs, C.byref(p), _vp(out), cap, _vp(offs))
            if st == L.CC_ERR_BUFFER_TOO_SMALL:
                cap = int(offs[n])
                continue
            L.check(st)
            return [out[offs[i]:offs[i + 1]].copy() for i in range(n)]

    def detect_batch_submit(self, frames, scaleFactor=1.1, minNeighbors=3, minSize=None, maxSize=None, device_ptr=None,
                            shape=None, row_stride=None, frame_stride=None):
        """First half of detect_batch (cc_detect_batch_submit): launches the batch and returns a ticket while its last pass
        still runs. Submit the next batch before collecting this one to overlap them. The frames must stay alive until
        detect_batch_collect(ticket)."""
        p = _params(scaleFactor, minNeighbors, minSize, maxSize)
        keep = None
        if device_ptr is None:
            keep = frames = np.ascontiguousarray(frames, np.uint8)
            n, h, w = frames.shape
            ptr, on_dev, rs, fs = _vp(frames), 0, w, w * h
        else:
            n, h, w = shape
            ptr, on_dev = C.c_void_p(device_ptr), 1
            rs = row_stride or w
            fs = frame_stride or rs * h
        t = C.c_void_p()
        L.check(L.lib().cc_detect_batch_submit(self._detector(), ptr, on_dev, n, w, h, rs, fs, C.byref(p), C.byref(t)))
        return {"ticket": t, "n": n, "frames": keep}

    def detect_batch_collect(self, ticket):
        """Second half: waits for the batch and returns what detect_batch returns (a list of (k_i, 4) arrays)."""
        n = ticket["n"]
        cap = max(256 * n, 1024)
        while True:
            out = np.zeros((cap, 4), np.int32)
            offs = np.zeros(n + 1, np.int32)
            st = L.lib().cc_detect_batch_collect(self._detector(), ticket["ticket"], _vp(out), cap, _vp(offs))
            if st == L.CC_ERR_BUFFER_TOO_SMALL:  # the ticket is still valid
                cap = int(offs[n])
                continue
            ticket["ticket"] = None
            ticket["frames"] = None
            L.check(st)
            return [out[offs[i]:offs[i + 1]].copy() for i in range(n)]

    def detect_batch_discard(self, ticket):
        """Ends a submitted batch whose results are not wanted (cc_detect_batch_discard)."""
        t, ticket["ticket"], ticket["frames"] = ticket["ticket"], None, None
        if t is not None:
            L.check(L.lib().cc_detect_batch_discard(self._detector(), t))

    def run_device_only(self, device_ptr, shape, scaleFactor=1.1, minSize=None, maxSize=None, row_stride=None,
                        frame_stride=None):
        n, h, w = shape
        p = _params(scaleFactor, 3, minSize, maxSize)
        rs = row_stride or w
        fs = frame_stride or rs * h
        L.check(L.lib().cc_detect_batch_device_only(self._detector(), C.c_void_p(device_ptr), 1, n, w, h, rs, fs, C.byref(p)))

    def detectMultiScale3(self, image, scaleFactor=1.1, minNeighbors=3, flags=0, minSize=None, maxSize=None, outputRejectLevels=True):
        """cv2.CascadeClassifier.detectMultiScale3: (rects (n, 4) int32, rejectLevels (n,) int32, levelWeights (n,) float64)."""
        if not outputRejectLevels:
            r = self.detectMultiScale(image, scaleFactor, minNeighbors, flags, minSize, maxSize)
            return r, np.zeros(0, np.int32), np.zeros(0, np.float64)
        image = np.ascontiguousarray(image, np.uint8)
        h, w = image.shape
        p = _params(scaleFactor, minNeighbors, minSize, maxSize)
        cap = 1024
        while True:
            rects = np.zeros((cap, 4), np.int32)
            levels = np.zeros(cap, np.int32)
            weights = np.zeros(cap, np.float64)
            n = C.c_int(0)
            st = L.lib().cc_detect_multiscale_levels(self._detector(), _vp(image), w, h, w, C.byref(p), _vp(rects), _vp(levels), _vp(weights),
                                                     cap, C.byref(n))
            if st == L.CC_ERR_BUFFER_TOO_SMALL:
                cap = n.value
                continue
            L.check(st)
            return rects[:n.value].copy(), levels[:n.value].copy(), weights[:n.value].copy()

    def detect_raw(self, image, scaleFactor=1.1, minSize=None, maxSize=None) -> np.ndarray:
        image = np.ascontiguousarray(image, np.uint8)
        h, w = image.shape
        p = _params(scaleFactor, 0, minSize, maxSize)
        cap = 4096
        while True:
            out = np.zeros((cap, 7), np.int32)
            n = C.c_int(0)
            st = L.lib().cc_detect_raw(self._detector(), _vp(image), w, h, w, C.byref(p), _vp(out), cap, C.byref(n))
            if st == L.CC_ERR_BUFFER_TOO_SMALL:
                cap = n.value
                continue
            L.check(st)
            return out[:n.value].copy()

    def debug_windows(self, image, scaleFactor=1.1, minSize=None, maxSize=None):
        image = np.ascontiguousarray(image, np.uint8)
        h, w = image.shape
        p = _params(scaleFactor, 0, minSize, maxSize)
        inf = self.info()
        sc = scale_plan(inf["win_w"], inf["win_h"], w, h, scaleFactor, minSize, maxSize)
        tot = int((sc["nx"].astype(np.int64) * sc["ny"]).sum())
        codes = np.zeros(max(tot, 1), np.int32)
        sums = np.zeros(max(tot, 1), np.float64)
        vis = np.zeros(max(tot, 1), np.uint8)
        n = C.c_int64(0)
        L.check(L.lib().cc_detect_debug_windows(self._detector(), _vp(image), w, h, w, C.byref(p), _vp(codes), _vp(sums),
                                                _vp(vis), tot, C.byref(n)))
        assert n.value == tot
        return codes[:tot], sums[:tot], vis[:tot]

    def specialize(self, n_stages: int = 4) -> int:
        """Compile the first n_stages stages of this cascade into the cascade kernel (hiprtc, a few seconds; Haar stump
        cascades). Results are unchanged; returns the number of stages in effect. n_stages <= 0 switches back."""
        L.check(L.lib().cc_detector_specialize(self._detector(), int(n_stages)))
        return self.specialized_stages()

    def specialize_async(self, n_stages: int = 4):
        """Start the same build on a background thread; detection keeps using the table-driven kernel until a later call
        finds the module ready (specialized_stages() then becomes non-zero)."""
        L.check(L.lib().cc_detector_specialize_async(self._detector(), int(n_stages)))

    def specialized_stages(self) -> int:
        return L.lib().cc_detector_specialized_stages(self._detector())

    def graph_active(self) -> bool:
        """True if the last single-image detectMultiScale call was one hipGraph launch (cc_detector_graph_active)."""
        return L.lib().cc_detector_graph_active(self._detector()) == 1

    def set_profiling(self, on: bool):
        L.check(L.lib().cc_detector_set_profiling(self._detector(), 1 if on else 0))

    def timings(self, reset=False) -> dict:
        t = L.DetectorTimings()
        L.check(L.lib().cc_detector_get_timings(self._detector(), C.byref(t), 1 if reset else 0))
        return {n: getattr(t, n) for n, _ in t._fields_}

    # -- lifetime -------------------------------------------------------------------------------
    def _release(self):
        if getattr(self, "_d", None):
            L.lib().cc_detector_destroy(self._d)
            self._d = C.c_void_p()
        if getattr(self, "_c", None):
            L.lib().cc_cascade_destroy(self._c)
            self._c = C.c_void_p()

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass


def vec_read(path: str, max_samples: int | None = None) -> np.ndarray:
    """Samples of a .vec file as (n, width*height) uint8 (PosReader semantics, imagestorage.cpp:138-182)."""
    cnt, vs = C.c_int32(0), C.c_int32(0)
    L.check(L.lib().cc_vec_read(path.encode(), C.byref(cnt), C.byref(vs), None, 0))
    n = cnt.value if max_samples is None else min(cnt.value, max_samples)
    out = np.zeros((max(n, 1), vs.value), np.uint8)
    L.check(L.lib().cc_vec_read(path.encode(), C.byref(cnt), C.byref(vs), _vp(out), n))
    return out[:n]


def vec_write(path: str, samples: np.ndarray, width: int, height: int):
    samples = np.ascontiguousarray(samples, np.uint8).reshape(-1, width * height)
    L.check(L.lib().cc_vec_write(path.encode(), _vp(samples), len(samples), width, height))
