"""Run-time specialised cascade kernel (cc_detector_specialize, hiprtc): the first stages compiled into straight-line
code must not change a single bit. Per-window result codes, stage sums, visited flags, candidates and grouped rectangles
are compared with the CPU oracle AND with the table-driven kernel, for the synthetic stock-profile cascade (all
specialisation depths incl. the stump-split path), a cascade with tilted features and a non-24x24 window."""
import numpy as np
import pytest

import cascadeclassifier_amd as cc
from cascadeclassifier_amd import _lib as L
from oracle import oracle as orc
from tests import cascade_factory as cf
from tests.util import frame_natural, frame_uniform

pytestmark = pytest.mark.gpu


def _same_as_oracle(p, o, img, sf):
    ref = orc.detect_raw(o, img, sf, nthreads=8, full=True)
    codes, sums, vis = p.debug_windows(img, sf)
    assert (codes == ref.codes).all(), f"{(codes != ref.codes).sum()} window results differ"
    assert (sums == ref.sums).all() and (vis == ref.visited).all()
    raw = p.detect_raw(img, sf)
    assert raw.shape == ref.candidates.shape and (raw == ref.candidates).all()
    a, b = p.detectMultiScale(img, sf, 2), orc.detect_multiscale(o, img, sf, 2, nthreads=8)
    assert a.shape == b.shape and (a == b).all()
    return len(raw)


@pytest.mark.parametrize("k", [1, 4, 7])
def test_specialised_stock_profile_cascade(haar_xml, k):
    o = orc.load_cascade_xml(haar_xml)
    p = cc.CascadeClassifier(haar_xml)
    assert p.specialized_stages() == 0
    got = p.specialize(k)
    assert 1 <= got <= k and p.specialized_stages() == got
    img = frame_natural(640, 360, 11)
    n = _same_as_oracle(p, o, img, 1.1)
    n += _same_as_oracle(p, o, frame_uniform(300, 200, 12), 1.25)
    assert n > 0
    # batch path (several passes, both streams) against the table-driven kernel
    frames = np.stack([frame_natural(480, 270, 20 + i) for i in range(6)])
    spec = p.detect_batch(frames, 1.1, 3)
    p.specialize(0)
    assert p.specialized_stages() == 0
    plain = p.detect_batch(frames, 1.1, 3)
    assert all(a.shape == b.shape and (a == b).all() for a, b in zip(spec, plain))


def test_specialised_mixed_fixed_point_and_double_stages(haar_xml, tmp_path):
    """The generated stages accumulate votes as int32 multiples of the stage's leaf quantum where the sums fit and as
    doubles otherwise. A tiny leaf (quantum 2^-45) in stages 1 and 3 of the stock-profile cascade pushes those two out of
    the integer form while their neighbours keep it: results must still match the oracle bit for bit."""
    import re
    text = open(haar_xml).read()
    sizes = [int(v) for v in re.findall(r"<maxWeakCount>(\d+)</maxWeakCount>", text[text.index("<stages>"):])]
    leaves = list(re.finditer(r"<leafValues>\s*(\S+)\s+(\S+)</leafValues>", text))
    assert len(leaves) == sum(sizes)
    edits = {sum(sizes[:1]) + 2: ("3.1e-07", "-3.1e-07"), sum(sizes[:3]) + 5: ("-2.9e-07", "4.4e-07")}
    out, last = [], 0
    for i, mt in enumerate(leaves):
        if i in edits:
            out.append(text[last:mt.start()] + "<leafValues>%s %s</leafValues>" % edits[i])
            last = mt.end()
    text = "".join(out) + text[last:]
    path = str(tmp_path / "mixed.xml")
    open(path, "w").write(text)
    o = orc.load_cascade_xml(path)
    p = cc.CascadeClassifier(path)
    assert p.specialize(7) == 7
    n = _same_as_oracle(p, o, frame_natural(640, 360, 11), 1.1)
    n += _same_as_oracle(p, o, frame_uniform(300, 200, 12), 1.25)
    assert n > 0


def test_specialised_tilted_and_other_windows(tmp_path):
    img = frame_natural(320, 240, 3)
    cal = np.stack([img[y:y + 24, x:x + 24] for y in range(0, 200, 9) for x in range(0, 280, 11)])
    xml = cf.tilted_stump_cascade(cal)
    path = str(tmp_path / "t.xml")
    open(path, "w").write(xml)
    p = cc.CascadeClassifier(path)
    assert p.specialize(8) >= 1
    _same_as_oracle(p, orc.load_cascade_xml(path), frame_natural(333, 127, 6), 1.2)


@pytest.mark.parametrize("sizes", [(6, 3, 2, 1), (2, 1, 3, 9, 2), (1, 1, 1)])
def test_specialised_stages_shorter_than_the_part_count(tmp_path, sizes, monkeypatch):
    """A generated stage is cut into as many parts as the block has wavefronts; a stage with FEWER stumps than that has
    empty parts, and a call that starts on an empty part must still issue the first stumps' reads (round-2 advisor
    finding: the prologue was tied to `p_lo == k` and skipped, so 1-3-stump stages summed never-loaded words). Both the
    whole-stage call and the stump-split slices are exercised (few survivors => the split path)."""
    img = frame_natural(320, 240, 3)
    cal = np.stack([img[y:y + 24, x:x + 24] for y in range(0, 200, 9) for x in range(0, 280, 11)])
    xml = cf.tilted_stump_cascade(cal, seed=5 + len(sizes), stage_sizes=sizes, tilted=False)
    path = str(tmp_path / "short.xml")
    open(path, "w").write(xml)
    o = orc.load_cascade_xml(path)
    for split in ("1", "0"):
        monkeypatch.setenv("CCAMD_SPLIT_STUMPS", split)
        p = cc.CascadeClassifier(path)
        assert p.specialize(len(sizes)) == len(sizes)
        n = _same_as_oracle(p, o, frame_natural(333, 227, 6), 1.2)
        n += _same_as_oracle(p, o, frame_uniform(200, 150, 8), 1.3)
        assert n > 0


@pytest.mark.parametrize("budget", [(0, 0), (8, 24), (7, 12), (1000, 1000)])
def test_stage_groups_do_not_change_results(lbp_xml, haar_xml, budget, monkeypatch):
    """Stage groups (one compaction + barrier per GROUP of short stages, lanes leave by predication inside a group):
    any cut of the cascade into groups must give the same codes, exit stages, stage sums and rectangles -- table-driven
    and specialised kernels, LBP and Haar, incl. the whole cascade as one dense group."""
    monkeypatch.setenv("CCAMD_DENSE_STUMPS", str(budget[0]))
    monkeypatch.setenv("CCAMD_GROUP_STUMPS", str(budget[1]))
    img, img2 = frame_natural(640, 360, 11), frame_uniform(300, 200, 12)
    for xml, k in ((lbp_xml, 20), (haar_xml, 4)):
        o = orc.load_cascade_xml(xml)
        p = cc.CascadeClassifier(xml)
        n = _same_as_oracle(p, o, img, 1.1)  # table-driven
        assert p.specialize(k) == k
        n += _same_as_oracle(p, o, img, 1.1)
        n += _same_as_oracle(p, o, img2, 1.25)
        assert n > 0


@pytest.mark.parametrize("dense_from,group_stumps,wave_below", [(1, 0, 24), (2, 20, 24), (3, 12, 8), (2, 10, 0), (0, 20, 24)])
def test_queue_form_does_not_change_results(lbp_xml, haar_xml, dense_from, group_stumps, wave_below, monkeypatch):
    """Round 4: from stage group CCAMD_DENSE_FROM on the block's queue is a plain list (wave-aggregated append) instead
    of the bank-class table. Which form a group reads must not change a code, exit stage, stage sum or rectangle:
    table-driven and specialised kernels, LBP (where the list is the default from stage 2) and Haar, together with the
    wave-phase threshold and the stage-group budget (the three knobs meet in the same queue)."""
    monkeypatch.setenv("CCAMD_DENSE_FROM", str(dense_from))
    monkeypatch.setenv("CCAMD_GROUP_STUMPS", str(group_stumps))
    monkeypatch.setenv("CCAMD_WAVE_BELOW", str(wave_below))
    img, img2 = frame_natural(640, 360, 21), frame_uniform(300, 200, 22)
    for xml, k in ((lbp_xml, 20), (haar_xml, 4)):
        o = orc.load_cascade_xml(xml)
        p = cc.CascadeClassifier(xml)
        n = _same_as_oracle(p, o, img, 1.1)  # table-driven
        assert p.specialize(k) == k
        n += _same_as_oracle(p, o, img, 1.1)
        n += _same_as_oracle(p, o, img2, 1.25)
        assert n > 0


def test_noinline_generated_stages_match(lbp_xml, haar_xml, monkeypatch):
    """CCAMD_SPEC_NOINLINE=1: the generated stages as one real function instead of a copy per call site."""
    monkeypatch.setenv("CCAMD_SPEC_NOINLINE", "1")
    monkeypatch.setenv("CCAMD_CACHE_DIR", "")
    img = frame_natural(640, 360, 11)
    for xml, k in ((lbp_xml, 20), (haar_xml, 7)):
        p = cc.CascadeClassifier(xml)
        assert p.specialize(k) == k
        assert _same_as_oracle(p, orc.load_cascade_xml(xml), img, 1.1) > 0


@pytest.mark.timeout(300)
@pytest.mark.parametrize("wave_below,split", [(24, 1), (8, 1), (64, 1), (0, 1), (24, 0)])
def test_every_late_stage_shape_in_one_launch(haar_xml, wave_below, split, monkeypatch):
    """Regression shape for the round-2 stall record ("incident records"): ONE launch in which some tiles
    carry long queues through the late stages (pasted face templates: whole rounds of row groups plus leftover groups
    that are split by stumps, ns > 1, with the extra block barrier of that path), others drop below `wave_below` at
    different stages (wave phase) and others empty out early -- under every setting of the switches that choose
    between those paths. Every barrier of the kernel sits on a block-uniform condition; if one did not, this launch
    would hang (the timeout fails the test) or diverge from the oracle."""
    monkeypatch.setenv("CCAMD_WAVE_BELOW", str(wave_below))
    monkeypatch.setenv("CCAMD_SPLIT_STUMPS", str(split))
    import os
    from tests.util import upscale
    tm = np.load(os.path.join(os.path.dirname(haar_xml), "face_template_24x24.npy"))
    img = frame_natural(704, 396, 5)
    img[200:, :352] = frame_uniform(352, 196, 6)           # tiles that empty out in stage 0-1
    img[:60, 400:] = 128                                    # flat: every window fails the variance test
    rng = np.random.default_rng(3)
    for k in (1.0, 1.0, 1.1, 1.21, 1.6, 2.0, 2.7, 4.0):     # faces: queues that stay long into the late stages
        s_ = int(24 * k)
        y, x = int(rng.integers(0, 396 - s_)), int(rng.integers(0, 704 - s_))
        img[y:y + s_, x:x + s_] = upscale(tm, s_)
    o = orc.load_cascade_xml(haar_xml)
    p = cc.CascadeClassifier(haar_xml)
    n = _same_as_oracle(p, o, img, 1.1)                     # table-driven kernel
    assert p.specialize(7) == 7
    n += _same_as_oracle(p, o, img, 1.1)
    assert n > 8


@pytest.mark.timeout(300)
@pytest.mark.parametrize("wave_below", [0, 8, 24, 64])
def test_lbp_wave_phase_thresholds(lbp_xml, tmp_path, wave_below, monkeypatch):
    """LBP wave phase (below `wave_below` windows a wavefront takes a window and its lanes the stumps of several whole
    stages; per-stage sums in stump order): every threshold, specialised (16-bit tile) and table-driven kernel, must give
    the oracle's codes, exit stages and stage sums. A cascade with a stage of more than 64 stumps cannot use it (a stage
    must fit a wavefront) and must still be right."""
    monkeypatch.setenv("CCAMD_WAVE_BELOW", str(wave_below))
    img, img2 = frame_natural(640, 360, 11), frame_uniform(300, 200, 12)
    o = orc.load_cascade_xml(lbp_xml)
    p = cc.CascadeClassifier(lbp_xml)
    n = _same_as_oracle(p, o, img, 1.1)
    assert p.specialize(20) == 20
    n += _same_as_oracle(p, o, img, 1.1) + _same_as_oracle(p, o, img2, 1.25)
    assert n > 0
    long_stage = cf.lbp_stump_cascade(24, 24, seed=9, stage_sizes=(3, 70, 5, 4))
    path = str(tmp_path / "long.xml")
    open(path, "w").write(long_stage)
    q = cc.CascadeClassifier(path)
    assert q.specialize(4) == 4
    assert _same_as_oracle(q, orc.load_cascade_xml(path), img2, 1.2) >= 0


def test_tile16_kernels_match(lbp_xml, haar_xml, tmp_path, monkeypatch):
    """CCAMD_SPEC_TILE16=1: STEP-2 tiles hold the low 16 bits of the integral (half the LDS bytes, 7-8 resident blocks
    per CU). Rectangle sums are exact modulo 2^16 while 255 * area < 2^16; larger rectangles are generated as strips that
    satisfy the bound, the variance rectangle as two halves, and the table-driven stages read the 32-bit integral from
    global memory. Everything must stay bit-identical: the stock-profile cascade (partly specialised, so that the
    global-memory records run too), the stock LBP cascade, and a cascade built ONLY from rectangles of >= 258 pixels
    (every stump takes the strip form)."""
    monkeypatch.setenv("CCAMD_SPEC_TILE16", "1")
    img, img2 = frame_natural(640, 360, 11), frame_uniform(300, 200, 12)
    for xml, k in ((haar_xml, 7), (haar_xml, 2), (lbp_xml, 20), (lbp_xml, 6)):
        p = cc.CascadeClassifier(xml)
        assert p.specialize(k) == k
        o = orc.load_cascade_xml(xml)
        n = _same_as_oracle(p, o, img, 1.1) + _same_as_oracle(p, o, img2, 1.25)
        assert n > 0
    cal = np.stack([img[y:y + 24, x:x + 24] for y in range(0, 300, 9) for x in range(0, 600, 11)])
    big = cf.tilted_stump_cascade(cal, seed=17, stage_sizes=(5, 8, 11, 14), tilted=False, min_area=258)
    path = str(tmp_path / "big.xml")
    open(path, "w").write(big)
    o = orc.load_cascade_xml(path)
    assert (o.haar["r"][:, 0, 2] * o.haar["r"][:, 0, 3]).min() >= 258  # every first rectangle exceeds the 16-bit bound
    for k in (4, 2):
        p = cc.CascadeClassifier(path)
        assert p.specialize(k) == k
        assert _same_as_oracle(p, o, img, 1.1) + _same_as_oracle(p, o, img2, 1.25) > 0
    frames = np.stack([frame_natural(480, 270, 20 + i) for i in range(6)])
    p = cc.CascadeClassifier(haar_xml)
    p.specialize(7)
    spec = p.detect_batch(frames, 1.1, 3)
    p.specialize(0)
    plain = p.detect_batch(frames, 1.1, 3)
    assert all(a.shape == b.shape and (a == b).all() for a, b in zip(spec, plain))


def test_pair_tile_kernels_match(haar_xml, tmp_path, monkeypatch):
    """CCAMD_SPEC_PAIR16=1: STEP-2 tiles hold PAIRS of 16-bit entries (entry i and its neighbour in the plane), so two
    windows two pixels apart share every LDS read and the corner arithmetic runs packed; the dense and the thread phase
    work on slots of one or two windows. Codes, exit stages, stage sums and rectangles must stay bit-identical: the
    stock-profile cascade fully / partly specialised (the table-driven stages of the thread phase and the wave phase use
    the strip records), every setting of the wave-phase and stump-split switches, a cascade built only from rectangles of
    >= 258 pixels (every stump in the strip form), and another window size."""
    monkeypatch.setenv("CCAMD_SPEC_PAIR16", "1")
    img, img2 = frame_natural(640, 360, 11), frame_uniform(300, 200, 12)
    o = orc.load_cascade_xml(haar_xml)
    for k in (7, 2, 1):
        p = cc.CascadeClassifier(haar_xml)
        assert p.specialize(k) == k
        assert _same_as_oracle(p, o, img, 1.1) + _same_as_oracle(p, o, img2, 1.25) > 0
    for env in ({"CCAMD_WAVE_BELOW": "0"}, {"CCAMD_WAVE_BELOW": "64"}, {"CCAMD_SPLIT_STUMPS": "0"}, {"CCAMD_WAVE_BELOW": "4", "CCAMD_SPEC_BUDGET": "60"}):
        for kk, v in env.items():
            monkeypatch.setenv(kk, v)
        p = cc.CascadeClassifier(haar_xml)
        assert p.specialize(7) >= 1
        assert _same_as_oracle(p, o, img, 1.1) > 0
        for kk in env:
            monkeypatch.delenv(kk)
    cal = np.stack([img[y:y + 24, x:x + 24] for y in range(0, 300, 9) for x in range(0, 600, 11)])
    big = cf.tilted_stump_cascade(cal, seed=17, stage_sizes=(5, 8, 11, 14), tilted=False, min_area=258)
    path = str(tmp_path / "big.xml")
    open(path, "w").write(big)
    ob = orc.load_cascade_xml(path)
    for k in (4, 2):
        p = cc.CascadeClassifier(path)
        assert p.specialize(k) == k
        assert _same_as_oracle(p, ob, img, 1.1) + _same_as_oracle(p, ob, img2, 1.25) > 0
    frames = np.stack([frame_natural(480, 270, 20 + i) for i in range(6)])
    p = cc.CascadeClassifier(haar_xml)
    p.specialize(7)
    spec = p.detect_batch(frames, 1.1, 3)
    p.specialize(0)
    plain = p.detect_batch(frames, 1.1, 3)
    assert all(a.shape == b.shape and (a == b).all() for a, b in zip(spec, plain))


def test_shared_corners_between_stumps_do_not_change_results(haar_xml, monkeypatch):
    """The generator re-orders the stumps of a stage (where the stage sum is exact) so that neighbours share rectangle
    corners, and a stump takes the shared words from its predecessor's registers instead of reading them again
    (CCAMD_SPEC_NO_SHARE=1: original order, every corner read; CCAMD_SPEC_SHARE_WINDOW: how many predecessors). Every variant
    must give the oracle's codes, exit stages, stage sums and rectangles -- also through the stump-split path, whose calls
    start in the middle of a stage."""
    img = frame_natural(640, 360, 11)
    o = orc.load_cascade_xml(haar_xml)
    for env in ({}, {"CCAMD_SPEC_NO_SHARE": "1"}, {"CCAMD_SPEC_SHARE_WINDOW": "3"}, {"CCAMD_WAVE_BELOW": "0"}, {"CCAMD_SPEC_TILE16": "1"}):
        for kk, v in env.items():
            monkeypatch.setenv(kk, v)
        p = cc.CascadeClassifier(haar_xml)
        assert p.specialize(7) == 7
        assert _same_as_oracle(p, o, img, 1.1) > 0
        for kk in env:
            monkeypatch.delenv(kk)


def test_specialised_lbp_cascade(lbp_xml):
    """The stock LBP cascade (20 stages, 139 stumps) compiled whole: bit-exact like the table-driven kernel."""
    o = orc.load_cascade_xml(lbp_xml)
    p = cc.CascadeClassifier(lbp_xml)
    assert p.specialize(20) == 20
    n = _same_as_oracle(p, o, frame_natural(640, 360, 11), 1.1)
    n += _same_as_oracle(p, o, frame_uniform(300, 200, 12), 1.25)
    frames = np.stack([frame_natural(480, 270, 40 + i) for i in range(5)])
    spec = p.detect_batch(frames, 1.1, 2)
    p.specialize(0)
    plain = p.detect_batch(frames, 1.1, 2)
    assert all(a.shape == b.shape and (a == b).all() for a, b in zip(spec, plain))


def test_specialise_refuses_what_it_does_not_cover(tmp_path):
    img = frame_natural(320, 240, 3)
    cal = np.stack([img[y:y + 24, x:x + 24] for y in range(0, 200, 9) for x in range(0, 280, 11)])
    trees = cc.CascadeClassifier()
    assert trees.load_from_string(cf.haar_tree_cascade(cal, with_tilted=False))
    with pytest.raises(cc.CascadeError, match="stump cascades only"):
        trees.specialize(2)


def test_background_specialisation_switches_over(haar_xml, monkeypatch):
    """cc_detector_specialize_async / CCAMD_AUTO_SPECIALIZE: detection runs on the table-driven kernel while a host thread
    compiles, and a later call picks the module up; results identical before and after."""
    import time
    img = frame_natural(400, 300, 91)
    want = orc.detect_multiscale(orc.load_cascade_xml(haar_xml), img, 1.1, 2, nthreads=8)
    p = cc.CascadeClassifier(haar_xml)
    p.specialize_async(2)
    seen_plain = p.specialized_stages() == 0
    deadline = time.time() + 120
    while p.specialized_stages() == 0 and time.time() < deadline:
        got = p.detectMultiScale(img, 1.1, 2)  # keeps working during the build
        assert got.shape == want.shape and (got == want).all()
        time.sleep(0.05)
    assert seen_plain and p.specialized_stages() == 2
    got = p.detectMultiScale(img, 1.1, 2)
    assert got.shape == want.shape and (got == want).all()
    # the environment switch does the same for a detector created without any call
    monkeypatch.setenv("CCAMD_AUTO_SPECIALIZE", "2")
    q = cc.CascadeClassifier(haar_xml)
    deadline = time.time() + 120
    while q.specialized_stages() == 0 and time.time() < deadline:
        got = q.detectMultiScale(img, 1.1, 2)
        assert got.shape == want.shape and (got == want).all()
        time.sleep(0.02)
    assert q.specialized_stages() == 2
    del q  # destroying a detector joins its build thread
    r = cc.CascadeClassifier(haar_xml)
    r.specialize_async(3)
    del r  # ... also while the build is still running


def test_fast_variance_norm_factor_is_the_two_rounded_operations():
    """Round 4: the cascade kernels take varianceNormFactor = (float)(1.0 / sqrt(nf)) from v_rsq_f64 + two Newton steps,
    with the two correctly rounded operations as the fallback near a float rounding boundary. Over 2^32 values of nf drawn
    as the kernels form them (area * valsqsum - valsum^2, every window size up to 256 x 256) plus arbitrary integers below
    2^52: no value may differ."""
    import ctypes as C
    bad = C.c_uint64(123)
    L.check(L.lib().cc_debug_vnf_check(0, 1 << 32, 20261005, C.byref(bad)))
    assert bad.value == 0
    L.check(L.lib().cc_debug_vnf_check(0, 1 << 28, 7, C.byref(bad)))
    assert bad.value == 0


@pytest.mark.parametrize("env", [
    {"CCAMD_SPEC_ONE_MODULE": "1"},                                   # Haar: one module for both steps, 8 rows
    {"CCAMD_SPEC_TILE_Y1": "8", "CCAMD_SPEC_TILE_Y2": "8"},           # two modules, library height
    {"CCAMD_SPEC_TILE_Y1": "16", "CCAMD_SPEC_TILE_Y2": "12"},         # QUEUE_ROWS = 32 / 24 (not a power of two)
    {"CCAMD_SPEC_TILE_Y": "20"},                                      # both modules 20 rows: five window rows per thread
    {"CCAMD_SPEC_TWO_MODULES": "1", "CCAMD_SPEC_TILE_Y1": "12"},      # LBP: a module per step, 12 / 20 rows
    {"CCAMD_SPEC_TILE_Y": "8"},                                       # LBP at the library's height (two rows per thread: spec_stage0_x2)
    {"CCAMD_SPEC_TILE_Y": "32", "CCAMD_WAVE_BELOW": "64"},            # LBP: 2 048 windows per block, the widest wave phase
])
def test_tile_heights_and_modules_do_not_change_results(lbp_xml, haar_xml, env, monkeypatch):
    """Round 4: the specialised kernel is compiled with its own tile height, as one module or one per step (spec_modules).
    Whatever the split and the heights -- incl. heights whose queue-row count is not a power of two, the rolled dense loop of
    tiles with more than two window rows per thread, and a second module for LBP -- every window's code, exit stage, stage
    sum, visited flag and every rectangle equal the CPU oracle's; frames larger and smaller than a tile, both steps."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    monkeypatch.setenv("CCAMD_CACHE_DIR", "")
    img, img2 = frame_natural(640, 360, 31), frame_uniform(150, 100, 32)
    for xml, k in ((lbp_xml, 20), (haar_xml, 4)):
        o = orc.load_cascade_xml(xml)
        p = cc.CascadeClassifier(xml)
        assert p.specialize(k) == k
        n = _same_as_oracle(p, o, img, 1.1)
        n += _same_as_oracle(p, o, img2, 1.25)
        assert n > 0
