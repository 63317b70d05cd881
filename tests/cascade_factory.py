"""Small synthetic cascades for tests (tilted Haar features, trees deeper than stumps): written in the reference's
cascade.xml format, loaded by BOTH the product (cc_cascade_load_xml) and the oracle's independent reader."""
import numpy as np

from oracle import oracle as orc


def _fmt(v):
    return ("%d." % int(v)) if float(v).is_integer() else ("%.8e" % v)


def haar_xml(feats, stages, mode="ALL", W=24, H=24):
    """feats: oracle HAAR_DTYPE array; stages: list of (threshold, [weak]) with weak = (nodes, leaves),
    nodes = list of (left, right, feature_idx, threshold)."""
    L = ['<?xml version="1.0"?>', "<opencv_storage>", "<cascade>", "  <stageType>BOOST</stageType>", "  <featureType>HAAR</featureType>",
         f"  <height>{H}</height>", f"  <width>{W}</width>",
         "  <featureParams><maxCatCount>0</maxCatCount><featSize>1</featSize><mode>%s</mode></featureParams>" % mode,
         f"  <stageNum>{len(stages)}</stageNum>", "  <stages>"]
    for thr, weaks in stages:
        L.append("    <_><maxWeakCount>%d</maxWeakCount><stageThreshold>%.8e</stageThreshold><weakClassifiers>" % (len(weaks), thr))
        for nodes, leaves in weaks:
            flat = " ".join("%d %d %d %.8e" % (l, r, f, t) for (l, r, f, t) in nodes)
            L.append("      <_><internalNodes>%s</internalNodes><leafValues>%s</leafValues></_>" % (flat, " ".join("%.8e" % v for v in leaves)))
        L.append("    </weakClassifiers></_>")
    L.append("  </stages>")
    L.append("  <features>")
    for f in feats:
        rects = "".join("<_>%d %d %d %d %s</_>" % (*f["r"][j], _fmt(f["wt"][j])) for j in range(3) if f["wt"][j] != 0)
        L.append("    <_><rects>%s</rects><tilted>%d</tilted></_>" % (rects, int(f["tilted"])))
    L += ["  </features>", "</cascade>", "</opencv_storage>"]
    return "\n".join(L) + "\n"


def calibration_values(feats, windows):
    """Normalised feature values (training-side evaluator of the oracle) of `feats` on 24x24 `windows`."""
    s, t, nf = orc.set_images(windows, want_tilted=bool(feats["tilted"].any()))
    ok = nf > 0
    return orc.haar_eval_batch(feats, 0, len(feats), s, t, nf, 24, 24)[:, ok]


def tilted_stump_cascade(windows, seed=11, stage_sizes=(6, 10, 14, 20), tilted=True, min_area=16):
    """Stump cascade whose features are drawn from the ALL catalog with every second one tilted (tilted=False: upright
    features only, from rectangles of at least min_area pixels)."""
    rng = np.random.default_rng(seed)
    cat = orc.haar_catalog(24, 24, 2)
    ui = np.nonzero((cat["tilted"] == 0) & (cat["r"][:, 0, 2] * cat["r"][:, 0, 3] >= min_area))[0]
    ti = np.nonzero(cat["tilted"] == 1)[0] if tilted else ui
    n = sum(stage_sizes)
    idx = np.empty(n, np.int64)
    idx[0::2] = rng.choice(ti, len(idx[0::2]), replace=False)
    idx[1::2] = rng.choice(ui, len(idx[1::2]), replace=False)
    feats = cat[idx].copy()
    v = calibration_values(feats, windows)
    thr = np.median(v, axis=1).astype(np.float32)
    a = rng.uniform(0.25, 1.0, n).astype(np.float32)
    sign = rng.choice([-1.0, 1.0], n).astype(np.float32)
    stages, k = [], 0
    alive = np.ones(v.shape[1], bool)
    for nw in stage_sizes:
        votes = np.where(v[k:k + nw] < thr[k:k + nw, None], (a * sign)[k:k + nw, None], (-a * sign)[k:k + nw, None]).astype(np.float64)
        sums = votes.sum(0)
        st = np.float32(np.median(sums[alive]) if alive.sum() > 20 else np.median(sums))
        alive &= sums >= st
        weaks = [([(0, -1, k + i, thr[k + i])], [a[k + i] * sign[k + i], -a[k + i] * sign[k + i]]) for i in range(nw)]
        stages.append((st, weaks))
        k += nw
    return haar_xml(feats, stages, mode="ALL" if tilted else "BASIC")


def lbp_xml(rects, stages, W=24, H=24):
    """rects: (n,4) cell rects; stages: list of (threshold, [weak]) with weak = (nodes, leaves),
    nodes = list of (left, right, feature_idx, [8 subset words])."""
    L = ['<?xml version="1.0"?>', "<opencv_storage>", '<cascade type_id="opencv-cascade-classifier">', "  <stageType>BOOST</stageType>",
         "  <featureType>LBP</featureType>", f"  <height>{H}</height>", f"  <width>{W}</width>",
         "  <featureParams><maxCatCount>256</maxCatCount><featSize>1</featSize></featureParams>", f"  <stageNum>{len(stages)}</stageNum>",
         "  <stages>"]
    for thr, weaks in stages:
        L.append("    <_><maxWeakCount>%d</maxWeakCount><stageThreshold>%.8e</stageThreshold><weakClassifiers>" % (len(weaks), thr))
        for nodes, leaves in weaks:
            flat = " ".join("%d %d %d %s" % (l, r, f, " ".join(str(int(w)) for w in sub)) for (l, r, f, sub) in nodes)
            L.append("      <_><internalNodes>%s</internalNodes><leafValues>%s</leafValues></_>" % (flat, " ".join("%.8e" % v for v in leaves)))
        L.append("    </weakClassifiers></_>")
    L.append("  </stages>")
    L.append("  <features>")
    for r in rects:
        L.append("    <_><rect>%d %d %d %d</rect></_>" % tuple(int(v) for v in r))
    L += ["  </features>", "</cascade>", "</opencv_storage>"]
    return "\n".join(L) + "\n"


# tree shapes in the writer's convention (internal children: positive BFS index; leaves: 0, -1, -2, ... in emission order)
_SHAPES = [
    [(0, -1)],                       # stump
    [(1, 2), (0, -1), (-2, -3)],     # full depth 2
    [(0, 1), (-1, -2)],              # left leaf, right subtree
    [(1, -2), (0, -1)],              # left subtree, right leaf
    [(1, 2), (3, 0), (-1, -2), (-3, -4)],  # depth 3, unbalanced
]


def haar_tree_cascade(windows, seed=21, stage_sizes=(4, 6, 8), with_tilted=False):
    rng = np.random.default_rng(seed)
    cat = orc.haar_catalog(24, 24, 2 if with_tilted else 0)
    pool = np.nonzero(cat["r"][:, 0, 2] * cat["r"][:, 0, 3] >= 16)[0]
    shapes = [_SHAPES[int(rng.integers(0, len(_SHAPES)))] for _ in range(sum(stage_sizes))]
    n_nodes = sum(len(sh) for sh in shapes)
    feats = cat[rng.choice(pool, n_nodes, replace=False)].copy()
    med = np.median(calibration_values(feats, windows), axis=1).astype(np.float32)
    stages, t, fi = [], 0, 0
    for nw in stage_sizes:
        weaks = []
        for _ in range(nw):
            sh = shapes[t]
            nodes = [(l, r, fi + k, med[fi + k]) for k, (l, r) in enumerate(sh)]
            leaves = rng.uniform(-1, 1, len(sh) + 1).astype(np.float32)
            weaks.append((nodes, leaves))
            fi += len(sh)
            t += 1
        stages.append((np.float32(-0.15 * nw), weaks))
    return haar_xml(feats, stages, mode="ALL" if with_tilted else "BASIC")


def lbp_tree_cascade(seed=31, stage_sizes=(3, 4, 5, 6)):
    rng = np.random.default_rng(seed)
    cat = orc.lbp_catalog(24, 24)
    shapes = [_SHAPES[int(rng.integers(0, len(_SHAPES)))] for _ in range(sum(stage_sizes))]
    n_nodes = sum(len(sh) for sh in shapes)
    rects = cat[rng.choice(len(cat), n_nodes, replace=False)]
    stages, t, fi = [], 0, 0
    for nw in stage_sizes:
        weaks = []
        for _ in range(nw):
            sh = shapes[t]
            nodes = [(l, r, fi + k, rng.integers(-2**31, 2**31, 8)) for k, (l, r) in enumerate(sh)]
            leaves = rng.uniform(-1, 1, len(sh) + 1).astype(np.float32)
            weaks.append((nodes, leaves))
            fi += len(sh)
            t += 1
        stages.append((np.float32(-0.1 * nw), weaks))
    return lbp_xml(rects, stages)


def lbp_stump_cascade(W, H, seed=41, stage_sizes=(3, 4, 5)):
    """LBP stump cascade for an arbitrary window size (exercises tile geometry other than 24x24)."""
    rng = np.random.default_rng(seed)
    cat = orc.lbp_catalog(W, H)
    n = sum(stage_sizes)
    rects = cat[rng.choice(len(cat), n, replace=False)]
    stages, fi = [], 0
    for nw in stage_sizes:
        weaks = []
        for _ in range(nw):
            weaks.append(([(0, -1, fi, rng.integers(-2**31, 2**31, 8))], rng.uniform(-1, 1, 2).astype(np.float32)))
            fi += 1
        stages.append((np.float32(-0.2 * nw), weaks))
    return lbp_xml(rects, stages, W=W, H=H)
