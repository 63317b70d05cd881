"""GPU parity tests of the node split search (section 6 of the C ABI) against the oracle's restatement of
CvDTree::find_best_split + CvBoostTree::find_split_{ord,cat}_{reg,class} (o_cvdtree.cpp:313-357,
o_cvboostree.cpp:151-516). Everything is compared exactly: the winner (variable, float quality, threshold / subset,
split point) and, variable by variable, the best quality (the device's double rounded to float, as the reference stores
it) and its split point. Equal feature values are ordered by stored-sample index on both sides (the reference's
std::sort leaves that order open; see the oracle header)."""
import numpy as np
import pytest

import cascadeclassifier_amd as cc
from cascadeclassifier_amd import evaluator as ev
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def _samples(n, win, seed, dup=0):
    """'positives' = template + noise, 'negatives' = uniform noise (BASELINE config 5's recipe); `dup` repeats some
    samples so that whole columns of feature values tie."""
    rng = np.random.default_rng(seed)
    W, H = win
    yy, xx = np.mgrid[0:H, 0:W]
    tmpl = 128 + 60 * np.sin(xx / W * 3.1) * np.cos(yy / H * 2.3)
    npos = n // 2
    pos = np.clip(tmpl[None] + rng.normal(0, 15, (npos, H, W)), 0, 255).astype(np.uint8)
    neg = rng.integers(0, 256, (n - npos, H, W), dtype=np.uint8)
    imgs = np.concatenate([pos, neg])
    labels = np.concatenate([np.ones(npos, np.uint8), np.zeros(n - npos, np.uint8)])
    perm = rng.permutation(n)
    imgs, labels = imgs[perm], labels[perm]
    for k in range(dup):
        imgs[(7 * k + 3) % n] = imgs[(11 * k + 1) % n]
    return imgs, labels


def _weights(n, labels01, seed, classifier):
    rng = np.random.default_rng(seed)
    w = rng.random(n) ** 3 + 1e-3
    w /= w.sum()
    if classifier:  # calc_node_value, o_cvboostree.cpp:671-685: totals per class, accumulated in node order
        r = [0.0, 0.0]
        for i in range(n):
            r[int(labels01[i])] += w[i]
        return np.concatenate([w, r])
    tot = 0.0
    for i in range(n):
        tot += w[i]
    return np.concatenate([w, [tot, 0.0]])


def _node_value(w, resp):  # regression branch of calc_node_value, o_cvboostree.cpp:711-722
    s = 0.0
    r = 0.0
    for i in range(len(resp)):
        r += w[i]
        s += float(resp[i]) * w[i]
    return s * (1.0 / r)


def _setup(ftype, mode, win, n, seed, dup=0):
    imgs, labels = _samples(n, win, seed, dup)
    e = cc.CvFeatureEvaluator.create(ftype)
    e.init(cc.CvFeatureParams(ftype, mode), n, win)
    e.setImages(imgs, labels)
    e.presort()
    s, t, nf = orc.set_images(imgs, want_tilted=(ftype == ev.HAAR and mode == ev.ALL), want_norm=ftype == ev.HAAR)
    return e, imgs, labels, (s, t, nf)


def _oracle_vals(ftype, mode, win, ints, sample_idx):
    s, t, nf = ints
    W, H = win
    if ftype == ev.HAAR:
        cat = orc.haar_catalog(W, H, mode)
        return orc.haar_eval_batch(cat, 0, len(cat), s, t, nf, W, H, sample_idx)
    cat = orc.lbp_catalog(W, H)
    return orc.lbp_eval_batch(cat, 0, len(cat), s, W, H, sample_idx)


def _check(e, ftype, mode, win, ints, labels, *, boost_type, criteria=0, sample_idx=None, seed=0, real_responses=False):
    n_all = len(labels)
    idx = None if sample_idx is None else np.asarray(sample_idx, np.int32)
    n = n_all if idx is None else len(idx)
    lab = labels.astype(np.int32) if idx is None else labels[idx].astype(np.int32)
    classifier = boost_type in (ev.BOOST_DISCRETE, ev.BOOST_REAL)
    w = _weights(n, lab, seed, classifier)
    kw = {}
    node_value = 0.0
    if classifier:
        kw["class_labels"] = lab
    else:
        resp = (lab * 2 - 1).astype(np.float32)
        if real_responses:  # LOGIT boost hands arbitrary working responses to the tree
            resp = (resp * np.random.default_rng(seed + 5).random(n) * 3).astype(np.float32)
        kw["responses"] = resp
        node_value = _node_value(w, resp)
    got, gq, gpt = e.find_best_split(w, sample_idx=idx, node_value=node_value, boost_type=boost_type, split_criteria=criteria,
                                     per_var=True, **kw)
    vals = _oracle_vals(ftype, mode, win, ints, idx)
    want, wq, wpt = orc.find_best_split(vals, w, categorical=ftype == ev.LBP, node_value=node_value, boost_type=boost_type,
                                        split_criteria=criteria, tie_key=idx, per_feature=True, **kw)
    gqf = np.where(gpt >= 0, gq.astype(np.float32), np.float32(-1))
    assert (gpt == wpt).all(), f"split points differ for {int((gpt != wpt).sum())} variables"
    assert (gqf.view(np.uint32) == wq.view(np.uint32)).all(), f"qualities differ for {int((gqf != wq).sum())} variables"
    assert bool(want["found"]) == got["found"]
    if got["found"]:
        assert got["var_idx"] == want["var_idx"] and got["quality"] == want["quality"]
        if ftype == ev.HAAR:
            assert got["ord_c"] == want["ord_c"] and got["split_point"] == want["split_point"]
        else:
            assert (got["subset"] == want["subset"]).all()
    return got


WIN = (12, 10)


@pytest.mark.parametrize("boost_type,criteria", [(ev.BOOST_GENTLE, 0), (ev.BOOST_LOGIT, 0), (ev.BOOST_REAL, 0), (ev.BOOST_DISCRETE, 0),
                                                 (ev.BOOST_REAL, ev.SPLIT_MISCLASS), (ev.BOOST_DISCRETE, ev.SPLIT_GINI)])
def test_haar_root_node(boost_type, criteria):
    e, imgs, labels, ints = _setup(ev.HAAR, ev.BASIC, WIN, 700, 1)
    got = _check(e, ev.HAAR, ev.BASIC, WIN, ints, labels, boost_type=boost_type, criteria=criteria, seed=3,
                 real_responses=boost_type == ev.BOOST_LOGIT)
    assert got["found"]


@pytest.mark.parametrize("n", [5000, 9000, 13000, 17000, 24577])
def test_presort_row_sizes(n):
    """cc_eval_presort sorts a row of up to 24 576 samples with one block (registers + LDS; 4 ... 24 keys per thread are
    separate kernel instances) and larger rows with the device-wide segmented sort: one size per instance and one past
    the limit, with duplicated samples so that whole columns tie (equal values must keep increasing sample order)."""
    e, _, labels, ints = _setup(ev.HAAR, ev.BASIC, (8, 8), n, seed=60 + n % 7, dup=n // 10)
    _check(e, ev.HAAR, ev.BASIC, (8, 8), ints, labels, boost_type=ev.BOOST_GENTLE, seed=3)


def test_haar_table_in_global_memory(monkeypatch):
    """Sample sets too large for the LDS-resident weight table take the global-memory gather path; force it here."""
    monkeypatch.setenv("CCAMD_SPLIT_GLOBAL_TABLE", "1")
    e, imgs, labels, ints = _setup(ev.HAAR, ev.BASIC, WIN, 300, 11, dup=20)
    for bt in (ev.BOOST_GENTLE, ev.BOOST_REAL, ev.BOOST_DISCRETE):
        _check(e, ev.HAAR, ev.BASIC, WIN, ints, labels, boost_type=bt, seed=12, sample_idx=np.arange(0, 300, 2) if bt == ev.BOOST_REAL else None)


def test_haar_all_mode_with_ties_and_node_subsets():
    e, imgs, labels, ints = _setup(ev.HAAR, ev.ALL, WIN, 500, 2, dup=120)
    _check(e, ev.HAAR, ev.ALL, WIN, ints, labels, boost_type=ev.BOOST_GENTLE, seed=1)
    rng = np.random.default_rng(9)
    sub = np.sort(rng.choice(500, 333, replace=False))  # a child node / a weight-trimmed sample set
    _check(e, ev.HAAR, ev.ALL, WIN, ints, labels, boost_type=ev.BOOST_GENTLE, sample_idx=sub, seed=2)
    _check(e, ev.HAAR, ev.ALL, WIN, ints, labels, boost_type=ev.BOOST_REAL, sample_idx=rng.permutation(sub), seed=4)  # any node order
    _check(e, ev.HAAR, ev.ALL, WIN, ints, labels, boost_type=ev.BOOST_DISCRETE, sample_idx=sub[:2], seed=5)


@pytest.mark.parametrize("boost_type,criteria", [(ev.BOOST_GENTLE, 0), (ev.BOOST_LOGIT, 0), (ev.BOOST_REAL, 0), (ev.BOOST_DISCRETE, 0)])
def test_lbp_root_and_subset(boost_type, criteria):
    e, imgs, labels, ints = _setup(ev.LBP, 0, WIN, 600, 3, dup=40)
    got = _check(e, ev.LBP, 0, WIN, ints, labels, boost_type=boost_type, criteria=criteria, seed=6, real_responses=boost_type == ev.BOOST_LOGIT)
    assert got["found"]
    sub = np.random.default_rng(4).permutation(600)[:257]
    _check(e, ev.LBP, 0, WIN, ints, labels, boost_type=boost_type, criteria=criteria, sample_idx=sub, seed=7)


def test_lbp_nodes_in_increasing_order_use_the_sorted_table(monkeypatch):
    """A node that lists its samples in increasing order (every node of a trainer) is searched from the (code, sample)-sorted
    table (k_split_cat_sorted), any other order by streaming the codes (k_split_cat): both must equal the oracle, and each
    other bit for bit, with the per-sample table in LDS (8- and 16-byte entries) and in global memory."""
    e, imgs, labels, ints = _setup(ev.LBP, 0, WIN, 900, 21, dup=70)
    rng = np.random.default_rng(8)
    sub = np.sort(rng.choice(900, 411, replace=False))
    for bt in (ev.BOOST_GENTLE, ev.BOOST_LOGIT, ev.BOOST_REAL, ev.BOOST_DISCRETE):
        _check(e, ev.LBP, 0, WIN, ints, labels, boost_type=bt, sample_idx=sub, seed=9, real_responses=bt == ev.BOOST_LOGIT)
        _check(e, ev.LBP, 0, WIN, ints, labels, boost_type=bt, sample_idx=sub[:2], seed=9)
    lab = labels[sub].astype(np.int32)
    w = _weights(len(sub), lab, 10, False)
    resp = (lab * 2 - 1).astype(np.float32)
    nv = _node_value(w, resp)
    ref = e.find_best_split(w, sample_idx=sub.astype(np.int32), node_value=nv, boost_type=ev.BOOST_GENTLE, responses=resp, per_var=True)
    monkeypatch.setenv("CCAMD_SPLIT_GLOBAL_TABLE", "1")
    glob = e.find_best_split(w, sample_idx=sub.astype(np.int32), node_value=nv, boost_type=ev.BOOST_GENTLE, responses=resp, per_var=True)
    monkeypatch.delenv("CCAMD_SPLIT_GLOBAL_TABLE")
    # a variable's ranks cut into parts (what 20 000 samples get by default): a category belongs to the part its run starts in
    for parts in (2, 5, 28):
        monkeypatch.setenv("CCAMD_SPLIT_CAT_PARTS", str(parts))
        for bt in (ev.BOOST_GENTLE, ev.BOOST_REAL):
            _check(e, ev.LBP, 0, WIN, ints, labels, boost_type=bt, sample_idx=sub, seed=11)
            _check(e, ev.LBP, 0, WIN, ints, labels, boost_type=bt, seed=12)
    monkeypatch.delenv("CCAMD_SPLIT_CAT_PARTS")
    monkeypatch.setenv("CCAMD_SPLIT_CAT_STREAM", "1")
    stream = e.find_best_split(w, sample_idx=sub.astype(np.int32), node_value=nv, boost_type=ev.BOOST_GENTLE, responses=resp, per_var=True)
    for other in (glob, stream):
        assert all(np.array_equal(other[0][k], ref[0][k]) for k in ref[0])
        assert (other[1].view(np.uint64) == ref[1].view(np.uint64)).all() and (other[2] == ref[2]).all()


def test_equal_weights_first_round():
    """Round 0 of boosting: all positives share one weight and all negatives another, so many categories / prefixes tie
    exactly — the case where the reference's own tie handling (std::sort of categories, first-best-wins) decides."""
    for ftype, mode in ((ev.HAAR, ev.BASIC), (ev.LBP, 0)):
        e, imgs, labels, ints = _setup(ftype, mode, WIN, 400, 5, dup=30)
        n = 400
        lab = labels.astype(np.int32)
        w = np.where(lab == 1, 0.5 / lab.sum(), 0.5 / (n - lab.sum()))
        tot = 0.0
        for v in w:
            tot += v
        W = np.concatenate([w, [tot, 0.0]])
        resp = (lab * 2 - 1).astype(np.float32)
        nv = _node_value(W, resp)
        got, gq, gpt = e.find_best_split(W, responses=resp, node_value=nv, per_var=True)
        want, wq, wpt = orc.find_best_split(_oracle_vals(ftype, mode, WIN, ints, None), W, categorical=ftype == ev.LBP, responses=resp,
                                            node_value=nv, per_feature=True)
        assert (gpt == wpt).all() and (np.where(gpt >= 0, gq.astype(np.float32), np.float32(-1)) == wq).all()
        assert got["found"] and got["var_idx"] == want["var_idx"] and got["quality"] == want["quality"]
        assert got["ord_c"] == want["ord_c"] and (got["subset"] == want["subset"]).all()


def test_full_catalog_24x24():
    """The real shape: Haar BASIC on 24x24 = 162 336 variables (several presort passes, 2 537 groups of 64)."""
    win = (24, 24)
    e, imgs, labels, ints = _setup(ev.HAAR, ev.BASIC, win, 300, 7)
    got = _check(e, ev.HAAR, ev.BASIC, win, ints, labels, boost_type=ev.BOOST_GENTLE, seed=8)
    assert got["found"]


def test_degenerate_nodes_and_errors():
    imgs, labels = _samples(64, WIN, 1)
    e = cc.CvFeatureEvaluator.create(ev.HAAR)
    e.init(cc.CvFeatureParams(ev.HAAR, ev.BASIC), 64, WIN)
    e.setImages(imgs, labels)
    w = np.full(66, 1 / 64.0)
    with pytest.raises(cc.CascadeError, match="cc_eval_presort first"):
        e.find_best_split(w, responses=np.ones(64, np.float32))
    e.presort(64)
    assert not e.find_best_split(np.array([1.0, 1.0, 0.0]), responses=np.ones(1, np.float32), sample_idx=[5])["found"]  # n <= 1
    # constant images: every feature value is 0 for every sample, no boundary to split at
    e.setImages(np.full((64, WIN[1], WIN[0]), 77, np.uint8), labels)
    e.presort(64)
    resp = (labels.astype(np.float32) * 2 - 1)
    assert not e.find_best_split(np.concatenate([np.full(64, 1 / 64.0), [1.0, 0.0]]), responses=resp, node_value=float(resp.mean()))["found"]
    with pytest.raises(cc.CascadeError, match="twice"):
        e.find_best_split(np.array([0.5, 0.5, 1.0, 0.0]), responses=np.ones(2, np.float32), sample_idx=[3, 3])
    with pytest.raises(cc.CascadeError, match="class_labels required"):
        e.find_best_split(w, responses=resp, boost_type=ev.BOOST_REAL)
    with pytest.raises(cc.CascadeError):
        e.find_best_split(np.array([0.5, 0.5, 1.0, 0.0]), responses=np.ones(2, np.float32), sample_idx=[3, 64])
    with pytest.raises(cc.CascadeError):
        e.presort(65)


def test_variables_sharded_over_two_devices_worth_of_ranges():
    """each GPU presorts a contiguous catalog range; the shard results combine to the unsharded winner."""
    from cascadeclassifier_amd.distributed import pick_split, shard_range
    for ftype, mode in ((ev.HAAR, ev.BASIC), (ev.LBP, 0)):
        e, imgs, labels, ints = _setup(ftype, mode, WIN, 300, 21, dup=10)
        lab = labels.astype(np.int32)
        w = _weights(300, lab, 3, False)
        resp = (lab * 2 - 1).astype(np.float32)
        nv = _node_value(w, resp)
        whole, q_all, pt_all = e.find_best_split(w, responses=resp, node_value=nv, per_var=True)
        F = e.getNumFeatures()
        parts = []
        for r in range(3):
            lo, hi = shard_range(F, r, 3)
            e.presort(300, lo, hi)
            part, q, pt = e.find_best_split(w, responses=resp, node_value=nv, per_var=True)
            assert (q == q_all[lo:hi]).all() and (pt == pt_all[lo:hi]).all() and (not part["found"] or lo <= part["var_idx"] < hi)
            parts.append(part)
        got = pick_split(parts)
        assert got["found"] and got["var_idx"] == whole["var_idx"] and got["quality"] == whole["quality"]
        assert got["ord_c"] == whole["ord_c"] and got["split_point"] == whole["split_point"] and (got["subset"] == whole["subset"]).all()


def test_gentle_adaboost_rounds_pick_identical_stumps():
    """Ten rounds of Gentle AdaBoost with stumps (CvCascadeBoost::train -> update_weights, boost.cpp:378-398: weak response
    f = weighted mean of y in the leaf, w *= exp(-y f), renormalise), the weights evolving from round to round: every
    round the device search and the oracle choose the same variable, threshold and quality, for Haar and for LBP."""
    for ftype, mode in ((ev.HAAR, ev.CORE), (ev.LBP, 0)):
        n = 360
        e, imgs, labels, ints = _setup(ftype, mode, WIN, n, 31, dup=12)
        vals_all = _oracle_vals(ftype, mode, WIN, ints, None)
        y = (labels.astype(np.float64) * 2 - 1)
        resp = y.astype(np.float32)
        w = np.full(n, 1.0 / n)
        chosen = []
        for rnd in range(10):
            tot = 0.0
            s = 0.0
            for i in range(n):  # calc_node_value of the root, o_cvboostree.cpp:711-722
                tot += w[i]
                s += float(resp[i]) * w[i]
            W = np.concatenate([w, [tot, 0.0]])
            nv = s * (1.0 / tot)
            got = e.find_best_split(W, responses=resp, node_value=nv)
            want = orc.find_best_split(vals_all, W, categorical=ftype == ev.LBP, responses=resp, node_value=nv)
            assert got["found"] and bool(want["found"]), rnd
            assert got["var_idx"] == want["var_idx"] and got["quality"] == want["quality"], rnd
            assert got["ord_c"] == want["ord_c"] and got["split_point"] == want["split_point"] and (got["subset"] == want["subset"]).all(), rnd
            chosen.append(got["var_idx"])
            v = e.calc_batch(got["var_idx"], got["var_idx"] + 1)[0]
            assert (v == vals_all[got["var_idx"]]).all()
            if ftype == ev.HAAR:
                left = v <= got["ord_c"]
            else:
                code = v.astype(np.int32)
                left = (got["subset"][code >> 5] >> (code & 31)) & 1 == 1
            f = np.empty(n)
            for side in (left, ~left):
                sw = 0.0
                sy = 0.0
                for i in np.nonzero(side)[0]:
                    sw += w[i]
                    sy += y[i] * w[i]
                f[side] = sy * (1.0 / sw) if sw > 0 else 0.0
            w = w * np.exp(-y * f)
            w = w * (1.0 / w.sum())
        assert len(set(chosen)) > 3  # boosting moves on to other variables as the weights change
