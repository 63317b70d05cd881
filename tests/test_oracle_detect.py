"""Self-consistency of the detection side of the oracle. PARITY UNPINNED against OpenCV (absent here; the reference
holds no golden detection output): these tests pin the restated rules of SURVEY.md Appendix A to hand-computed cases and
to properties that must hold whatever the inputs."""
import numpy as np

from oracle import oracle as orc
from tests.util import frame_natural, frame_uniform


def test_resize_identity_and_constant():
    img = frame_uniform(37, 21, 1)
    assert (orc.resize_linear_exact(img, 37, 21) == img).all()
    flat = np.full((50, 70), 93, np.uint8)
    assert (orc.resize_linear_exact(flat, 33, 17) == 93).all()


def test_resize_half_scale_is_2x2_mean_rounded():
    # dst = src/2 exactly: f = 2(d+0.5)-0.5 = 2d+0.5 -> taps (2d, 2d+1) with weights 128/128 on both axes:
    # v = (128*(a+b)*128 + 128*(c+d)*128 + 2^15) >> 16 = (a+b+c+d+2) >> 2
    img = frame_uniform(64, 48, 2).astype(np.int64)
    want = (img[0::2, 0::2] + img[0::2, 1::2] + img[1::2, 0::2] + img[1::2, 1::2] + 2) >> 2
    assert (orc.resize_linear_exact(img.astype(np.uint8), 32, 24) == want).all()


def test_resize_three_to_two_hand_case():
    # src 3 -> dst 2: scale 1.5; d=0: f=0.25 -> taps (0,1), w1 = round(0.25*256)=64; d=1: f=1.75 -> taps (1,2), w1=192
    row = np.array([[10, 110, 250]], np.uint8)
    got = orc.resize_linear_exact(np.repeat(row, 3, 0), 2, 3)[0]
    h0 = 192 * 10 + 64 * 110
    h1 = 64 * 110 + 192 * 250
    assert got.tolist() == [(h0 * 256 + 32768) >> 16, (h1 * 256 + 32768) >> 16]


def test_scan_loop_skip_rule_recurrence(lbp_xml, haar_xml):
    for xml in (lbp_xml, haar_xml):
        c = orc.load_cascade_xml(xml)
        r = orc.detect_raw(c, frame_natural(200, 150, 4), 1.2, full=True)
        sc = orc.scales(24, 24, 200, 150, 1.2)
        o = 0
        n_vis = 0
        for s in sc:
            codes = r.codes[o:o + s["nx"] * s["ny"]].reshape(s["ny"], s["nx"])
            vis = r.visited[o:o + s["nx"] * s["ny"]].reshape(s["ny"], s["nx"])
            o += s["nx"] * s["ny"]
            assert vis[:, 0].all()
            # visited(i+1) = not (visited(i) and rejected_at_stage0(i))
            want = ~(vis[:, :-1].astype(bool) & (codes[:, :-1] == 0))
            assert (vis[:, 1:].astype(bool) == want).all()
            n_vis += int(vis.sum())
        assert n_vis == r.n_visited_windows
        # candidates = visited windows that pass, in (scale, y, x) order
        assert len(r.candidates) == int(((r.codes == 1) & (r.visited == 1)).sum())
        key = r.candidates[:, 0].astype(np.int64) << 40 | r.candidates[:, 2].astype(np.int64) << 20 | r.candidates[:, 1]
        assert (np.diff(key) > 0).all()


def test_thread_count_does_not_change_results(haar_xml):
    c = orc.load_cascade_xml(haar_xml)
    img = frame_natural(320, 240, 8)
    a = orc.detect_raw(c, img, 1.1, nthreads=1, full=True)
    b = orc.detect_raw(c, img, 1.1, nthreads=7, full=True)
    assert (a.codes == b.codes).all() and (a.sums == b.sums).all() and (a.candidates == b.candidates).all()


def test_candidate_rectangle_geometry(lbp_xml):
    c = orc.load_cascade_xml(lbp_xml)
    r = orc.detect_raw(c, frame_natural(320, 240, 9), 1.1)
    sc = orc.scales(24, 24, 320, 240, 1.1)
    for s_idx, gx, gy, x, y, w, h in r.candidates:
        s = sc[s_idx]
        assert x == int(np.rint(np.float32(gx * s["ystep"]) * s["scale"])) and w == s["win_w"] and h == s["win_h"]
        assert 0 <= gx < s["nx"] and 0 <= gy < s["ny"]


def test_low_variance_windows_are_rejected_before_stage0(haar_xml):
    c = orc.load_cascade_xml(haar_xml)
    r = orc.detect_raw(c, np.full((60, 80), 200, np.uint8), 1.1, full=True)
    assert (r.codes == -1).all() and (r.visited == 1).all() and (r.sums == 0).all()
