"""Multi-rank layer (frames shard across ranks, gather of detections only) on CPU: world_size-2 gloo processes.
The sharding / gather code is the product's (cascadeclassifier_amd/distributed.py); per-frame detections are synthetic
rectangle lists here, because the detection kernels need a GPU (they are covered by tests/test_gpu_detect.py)."""
import os
import socket
import sys

import numpy as np
import pytest

from cascadeclassifier_amd.distributed import gather_detections, shard_range

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partitions_everything():
    for n in (0, 1, 7, 64, 512, 513):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    assert [shard_range(512, r, 8) for r in range(8)] == [(64 * r, 64 * r + 64) for r in range(8)]  # config 4


def _fake_detections(frame_idx):
    rng = np.random.default_rng(1000 + frame_idx)
    k = int(rng.integers(0, 6))
    return rng.integers(0, 1900, (k, 4)).astype(np.int32)


def _worker(rank, world, port, n_frames, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_range(n_frames, rank, world)
    mine = [_fake_detections(f) for f in range(lo, hi)]
    allr = gather_detections(mine)
    q.put((rank, [a.tolist() for a in allr]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_frames", [7, 2, 1])
def test_gather_detections_gloo_world2(n_frames):
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_frames, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = [_fake_detections(f).tolist() for f in range(n_frames)]
    assert got[0] == want and got[1] == want  # every rank holds all frames' detections, in global frame order


def test_gather_is_identity_without_process_group():
    rects = [np.array([[1, 2, 3, 4]], np.int32), np.zeros((0, 4), np.int32)]
    out = gather_detections(rects)
    assert len(out) == 2 and out[0].tolist() == [[1, 2, 3, 4]] and out[1].shape == (0, 4)


# ---- training side: split search with variables sharded over ranks ------------------------------------------------
class _FakeShard:
    """Stands in for an evaluator that presorted one catalog range: returns that range's best split from a table of
    per-variable float qualities (the device search itself is covered by tests/test_gpu_split.py)."""

    def __init__(self, quality, lo, hi):
        self.q, self.lo, self.hi = quality, lo, hi

    def find_best_split(self, weights, **kw):
        q = self.q[self.lo:self.hi]
        ok = len(q) > 0 and q.max() > 0
        k = int(np.argmax(q)) if ok else 0  # first occurrence of the maximum
        return {"found": bool(ok), "var_idx": self.lo + k, "quality": np.float32(q[k]) if ok else np.float32(-1),
                "ord_c": np.float32(0.25 * (self.lo + k)), "split_point": 7 + self.lo + k, "subset": np.arange(8, dtype=np.int32) * (self.lo + k)}


def _qualities(case):
    rng = np.random.default_rng(case)
    q = rng.random(101).astype(np.float32)
    if case == 1:
        q[[13, 77]] = 2.0  # the same best quality in both shards: the lower variable index must win
    if case == 2:
        q[:] = -1.0  # no variable has a split
    return q


def _split_worker(rank, world, port, case, q):
    import torch.distributed as dist

    from cascadeclassifier_amd.distributed import find_best_split_sharded
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    qual = _qualities(case)
    got = find_best_split_sharded(_FakeShard(qual, *shard_range(len(qual), rank, world)), None)
    q.put((rank, got["found"], got["var_idx"], float(got["quality"]), float(got["ord_c"]), got["split_point"], got["subset"].tolist()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("case", [0, 1, 2])
def test_sharded_split_search_gloo_world2(case):
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_split_worker, args=(r, 2, port, case, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict((r[0], r[1:]) for r in (q.get(timeout=120) for _ in range(2)))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    qual = _qualities(case)
    whole = _FakeShard(qual, 0, len(qual)).find_best_split(None)  # the unsharded scan
    want = (whole["found"], whole["var_idx"], float(whole["quality"]), float(whole["ord_c"]), whole["split_point"], whole["subset"].tolist())
    if not whole["found"]:
        assert not got[0][0] and not got[1][0]
    else:
        assert got[0] == want and got[1] == want


# ---- section 7 of the C ABI: cc_shard_range / cc_comm_* / cc_gather_detections ------------------------------------
def test_c_abi_shard_range_matches_python():
    import ctypes as C

    from cascadeclassifier_amd import _lib as L
    for n in (0, 1, 7, 64, 512, 513):
        for world in (1, 2, 3, 8):
            for r in range(world):
                lo, hi = C.c_int(-1), C.c_int(-1)
                L.lib().cc_shard_range(n, r, world, C.byref(lo), C.byref(hi))
                assert (lo.value, hi.value) == shard_range(n, r, world)


def test_c_abi_gather_single_rank_needs_no_device_and_keeps_the_result():
    """world == 1: the gather is a copy (no RCCL, no HIP call), so the protocol's packing / unpacking and the
    BUFFER_TOO_SMALL -> cc_gather_fetch rule are testable here."""
    import ctypes as C

    from cascadeclassifier_amd import _lib as L
    from cascadeclassifier_amd.distributed import Comm
    frames = [_fake_detections(f) for f in range(9)] + [np.zeros((0, 4), np.int32)]
    comm = Comm(0, 0, 1)
    got = gather_detections(frames, comm=comm)
    assert len(got) == len(frames) and all(a.shape == b.shape and (a == b).all() for a, b in zip(got, frames))
    assert gather_detections([], comm=comm) == []
    # too-small buffers: status, totals reported, result kept for cc_gather_fetch
    counts = np.array([len(r) for r in frames], np.int32)
    off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    rects = np.ascontiguousarray(np.concatenate(frames), np.int32)
    nf, nr = C.c_int(0), C.c_int(0)
    small = np.empty((1, 4), np.int32)
    so = np.empty(2, np.int32)
    st = L.lib().cc_gather_detections(comm._c, rects.ctypes.data_as(C.c_void_p), off.ctypes.data_as(C.c_void_p), len(frames),
                                      small.ctypes.data_as(C.c_void_p), 1, so.ctypes.data_as(C.c_void_p), 1, C.byref(nf), C.byref(nr))
    assert st == L.CC_ERR_BUFFER_TOO_SMALL and nf.value == len(frames) and nr.value == len(rects)
    out = np.empty((nr.value, 4), np.int32)
    oo = np.empty(nf.value + 1, np.int32)
    L.check(L.lib().cc_gather_fetch(comm._c, out.ctypes.data_as(C.c_void_p), nr.value, oo.ctypes.data_as(C.c_void_p), nf.value))
    assert (oo == off).all() and (out == rects).all()
    # argument checks
    bad = off.copy()
    bad[3] = bad[2] - 1
    assert L.lib().cc_gather_detections(comm._c, rects.ctypes.data_as(C.c_void_p), bad.ctypes.data_as(C.c_void_p), len(frames), None, 0, None, 0,
                                        C.byref(nf), C.byref(nr)) == L.CC_ERR_INVALID_ARG
    comm.close()
    c = C.c_void_p()
    assert L.lib().cc_comm_create(0, 0, 2, None, C.byref(c)) == L.CC_ERR_INVALID_ARG  # more than one rank needs the unique id
    assert L.lib().cc_comm_create(0, 3, 2, None, C.byref(c)) == L.CC_ERR_INVALID_ARG


def _comm_worker(rank, world, port, q):
    import torch.distributed as dist

    from cascadeclassifier_amd import _lib as L
    from cascadeclassifier_amd.distributed import Comm
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        comm = Comm.from_torch(0)  # the id travels over gloo; both ranks then ask RCCL for a communicator on device 0
    except L.CascadeError as e:
        q.put((rank, "refused", str(e)))
    else:
        lo, hi = shard_range(7, rank, world)
        allr = gather_detections([_fake_detections(f) for f in range(lo, hi)], comm=comm)
        q.put((rank, "ok", [a.tolist() for a in allr]))
        comm.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_c_abi_comm_two_ranks_on_one_gpu_either_works_or_fails_loudly():
    """Two ranks on the box's single GPU: RCCL either builds the communicator (then the gathered lists must be right)
    or refuses duplicate devices -- in which case cc_comm_create must return an error on BOTH ranks (bench.py then falls
    back to the torch collective), not hang. Bounded by a timeout."""
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_comm_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        got = [q.get(timeout=180) for _ in range(2)]
    finally:
        for p in procs:
            p.join(timeout=30)
            if p.is_alive():
                p.kill()
    # a rank that aborts at exit (round 2: "double free or corruption" inside librccl after an unused id) must fail the test
    assert [p.exitcode for p in procs] == [0, 0], [p.exitcode for p in procs]
    kinds = {g[1] for g in got}
    assert len(kinds) == 1, got  # both ranks agree
    if kinds == {"ok"}:
        want = [_fake_detections(f).tolist() for f in range(7)]
        assert all(g[2] == want for g in got)
    else:
        print("RCCL refused two ranks on one device:", got[0][2])


# ---- the C ABI's world > 1 gather, executed: CPU processes over the tcp transport (CCAMD_COMM_TRANSPORT=tcp) --------
def _tcp_rank(rank, world, n_frames, case, id_q, out_q):
    """One rank of a cc_comm_* job. The 128 id bytes travel through a multiprocessing queue (any channel will do)."""
    import ctypes as C
    os.environ["CCAMD_COMM_TRANSPORT"] = "tcp"
    os.environ["CCAMD_COMM_TIMEOUT_S"] = "60"
    from cascadeclassifier_amd import _lib as L
    from cascadeclassifier_amd.distributed import Comm
    if rank == 0:
        uid = Comm.unique_id()
        assert Comm.unique_id() == uid  # an id nobody has used yet is handed out again, not replaced
        for _ in range(world - 1):
            id_q.put(uid)
    else:
        uid = id_q.get(timeout=60)
    comm = Comm(0, rank, world, uid)
    assert (L.lib().cc_comm_rank(comm._c), L.lib().cc_comm_world(comm._c)) == (rank, world)
    lo, hi = shard_range(n_frames, rank, world)
    mine = [_fake_detections(f) for f in range(lo, hi)]
    if (case == "empty_rank" and rank == 1) or (case == "configs3" and rank == 5):
        mine = [np.zeros((0, 4), np.int32) for _ in mine]  # frames, but not one rectangle
    res = {}
    if case == "bad_args" and rank == world - 1:
        # one rank calls with decreasing offsets: it still takes part in the header exchange, and EVERY rank returns an error
        nf, nr = C.c_int(0), C.c_int(0)
        off = np.array([0, 5, 3], np.int32)
        rects = np.zeros((5, 4), np.int32)
        st = L.lib().cc_gather_detections(comm._c, rects.ctypes.data_as(C.c_void_p), off.ctypes.data_as(C.c_void_p), 2, None, 0, None, 0,
                                          C.byref(nf), C.byref(nr))
        res["status"] = st
    elif case == "bad_args":
        try:
            gather_detections(mine, comm=comm)
            res["status"] = 0
        except L.CascadeError as e:
            res["status"] = e.status
            res["msg"] = str(e)
    elif case in ("small_on_one", "configs3") and rank == 0:
        # rank 0 alone passes buffers that are too small: BUFFER_TOO_SMALL after the collectives, then cc_gather_fetch
        counts = np.array([len(r) for r in mine], np.int32)
        off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
        rects = np.ascontiguousarray(np.concatenate(mine) if mine else np.zeros((0, 4), np.int32), np.int32)
        nf, nr = C.c_int(0), C.c_int(0)
        so = np.empty(2, np.int32)
        st = L.lib().cc_gather_detections(comm._c, rects.ctypes.data_as(C.c_void_p), off.ctypes.data_as(C.c_void_p), len(mine), None, 0,
                                          so.ctypes.data_as(C.c_void_p), 1, C.byref(nf), C.byref(nr))
        assert st == L.CC_ERR_BUFFER_TOO_SMALL, st
        out = np.empty((max(nr.value, 1), 4), np.int32)
        oo = np.empty(nf.value + 1, np.int32)
        L.check(L.lib().cc_gather_fetch(comm._c, out.ctypes.data_as(C.c_void_p), len(out), oo.ctypes.data_as(C.c_void_p), nf.value))
        res["all"] = [out[oo[f]:oo[f + 1]].tolist() for f in range(nf.value)]
    else:
        res["all"] = [a.tolist() for a in gather_detections(mine, comm=comm)]
        if case == "twice":  # the communicator is reusable; a second gather with other data
            res["again"] = [a.tolist() for a in gather_detections(mine[::-1], comm=comm)]
    comm.close()
    out_q.put((rank, res))


def _run_tcp_job(world, n_frames, case):
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    id_q, out_q = ctx.Queue(), ctx.Queue()
    procs = [ctx.Process(target=_tcp_rank, args=(r, world, n_frames, case, id_q, out_q)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        got = dict(out_q.get(timeout=120) for _ in range(world))
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()
    assert [p.exitcode for p in procs] == [0] * world, [p.exitcode for p in procs]
    return got


@pytest.mark.parametrize("world,n_frames", [(2, 7), (3, 7), (3, 2), (2, 0)])
def test_c_abi_gather_world_gt_1_over_the_tcp_transport(world, n_frames):
    """cc_comm_create / cc_gather_detections with world 2 and 3, run by spawned CPU processes: unequal frame counts
    (7 over 3, 2 over 3 = a rank without frames), padding to the longest payload, unpacking in rank order."""
    got = _run_tcp_job(world, n_frames, "plain")
    want = [_fake_detections(f).tolist() for f in range(n_frames)]
    assert all(got[r]["all"] == want for r in range(world))


def test_c_abi_gather_rank_without_rectangles_and_reuse():
    got = _run_tcp_job(3, 8, "empty_rank")
    lo, hi = shard_range(8, 1, 3)
    want = [([] if lo <= f < hi else _fake_detections(f).tolist()) for f in range(8)]
    assert all(got[r]["all"] == want for r in range(3))
    got = _run_tcp_job(2, 5, "twice")
    want = [_fake_detections(f).tolist() for f in range(5)]
    spans = [shard_range(5, r, 2) for r in range(2)]
    again = [_fake_detections(f).tolist() for (lo, hi) in spans for f in reversed(range(lo, hi))]
    assert all(got[r]["all"] == want and got[r]["again"] == again for r in range(2))


def test_c_abi_gather_buffer_too_small_on_one_rank_only():
    """Rank 0 alone has too little room: it gets CC_ERR_BUFFER_TOO_SMALL only after both collectives (the other ranks
    complete normally) and reads the kept result with cc_gather_fetch, without communicating again."""
    got = _run_tcp_job(3, 7, "small_on_one")
    want = [_fake_detections(f).tolist() for f in range(7)]
    assert all(got[r]["all"] == want for r in range(3))


def test_c_abi_gather_at_the_shape_of_baseline_configs3():
    """BASELINE configs[3] as the 8-GPU run will see it, rehearsed on CPU over the tcp transport: world 8, 512 frames =
    64 per rank, one rank whose frames hold no rectangle, rank 0 with too little room (CC_ERR_BUFFER_TOO_SMALL after the
    collectives, then cc_gather_fetch). Every rank ends up with all 512 frames' rectangles in frame order and exits 0.
    (ncclCommInitRank / ncclAllGather themselves with world > 1 stay unverified until a multi-GPU node runs them.)"""
    got = _run_tcp_job(8, 512, "configs3")
    lo, hi = shard_range(512, 5, 8)
    assert (lo, hi) == (320, 384)
    want = [([] if lo <= f < hi else _fake_detections(f).tolist()) for f in range(512)]
    assert all(got[r]["all"] == want for r in range(8))


def test_c_abi_gather_bad_arguments_on_one_rank_fail_everywhere_without_a_hang():
    """Round-2 advisor finding: a rank whose argument check failed returned before the first all-gather and left its peers
    blocked in it. Now it contributes an error marker and all ranks return CC_ERR_INVALID_ARG together."""
    from cascadeclassifier_amd import _lib as L
    got = _run_tcp_job(3, 6, "bad_args")
    assert [got[r]["status"] for r in range(3)] == [L.CC_ERR_INVALID_ARG] * 3, got
    assert "rank 2 reported invalid arguments" in got[0]["msg"]


def test_comm_getters_and_transport_mismatch(monkeypatch):
    import ctypes as C

    from cascadeclassifier_amd import _lib as L
    assert L.lib().cc_comm_rank(None) == -1 and L.lib().cc_comm_world(None) == -1  # not a status code that looks like a rank
    monkeypatch.setenv("CCAMD_COMM_TRANSPORT", "tcp")
    buf = C.create_string_buffer(128)
    L.check(L.lib().cc_comm_unique_id(buf))
    monkeypatch.delenv("CCAMD_COMM_TRANSPORT")
    c = C.c_void_p()
    # an id of the test transport is refused unless the environment selects that transport
    assert L.lib().cc_comm_create(0, 1, 2, buf, C.byref(c)) == L.CC_ERR_INVALID_ARG
