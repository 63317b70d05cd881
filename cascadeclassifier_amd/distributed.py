"""Multi-GPU layer of the detection path: frames shard across ranks (one process per GPU), no data-path collective;
the only exchange is a gather of the detections (a few KB per frame) to every rank / rank 0 over torch.distributed
(backend "nccl" = RCCL over xGMI on the GPU node, "gloo" in the CPU tests). SURVEY.md §8e."""
from __future__ import annotations

import numpy as np


def shard_range(n_items: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous block partition [lo, hi) of n_items over `world` ranks; the first n_items % world ranks get one more."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class Comm:
    """The C ABI's RCCL communicator (cc_comm_*, include/cascadeclassifier_amd.h section 7): what a C++ host program uses
    for the gather of detections. from_torch() bootstraps it inside a torch.distributed job: rank 0 creates the RCCL
    unique id and the existing process group broadcasts its 128 bytes."""

    def __init__(self, device: int, rank: int, world: int, unique_id: bytes | None = None):
        import ctypes as C

        from . import _lib as L
        self._c = C.c_void_p()
        buf = None if unique_id is None else C.create_string_buffer(bytes(unique_id), 128)
        L.check(L.lib().cc_comm_create(int(device), int(rank), int(world), buf, C.byref(self._c)))
        self.rank, self.world = rank, world

    @staticmethod
    def unique_id() -> bytes:
        import ctypes as C

        from . import _lib as L
        buf = C.create_string_buffer(128)
        L.check(L.lib().cc_comm_unique_id(buf))
        return buf.raw

    @classmethod
    def from_torch(cls, device_index: int, group=None):
        """Collective over `group`. Rank 0 makes the id; the broadcast carries a flag byte in front of it, so that a
        failure on rank 0 (librccl missing, ...) reaches every rank through the SAME broadcast and all of them raise --
        a rank 0 that raised before the broadcast would leave the others waiting in it."""
        import torch
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        dev = torch.device("cuda", device_index) if dist.get_backend(group) == "nccl" else torch.device("cpu")
        msg = torch.zeros(1 + 128, dtype=torch.uint8, device=dev)
        err = None
        if rank == 0:
            try:
                uid = cls.unique_id()
                msg[0] = 1
                msg[1:].copy_(torch.frombuffer(bytearray(uid), dtype=torch.uint8))
            except Exception as e:  # noqa: BLE001 -- reported below, after the broadcast every rank takes part in
                err = e
        dist.broadcast(msg, src=0, group=group)
        host = msg.cpu().numpy()
        if int(host[0]) != 1:
            if err is not None:
                raise err
            from . import _lib as L
            raise L.CascadeError(L.CC_ERR_UNSUPPORTED, "Comm.from_torch: rank 0 could not create the communicator id")
        return cls(device_index, rank, world, host[1:].tobytes())

    def gather_all(self, per_frame: list[np.ndarray]) -> list[np.ndarray]:
        """cc_gather_detections: this rank's per-frame rectangle lists in, every rank's (global frame order) out."""
        import ctypes as C

        from . import _lib as L
        counts = np.array([len(r) for r in per_frame], np.int32)
        offsets = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
        rects = (np.concatenate([np.asarray(r, np.int32).reshape(-1, 4) for r in per_frame]) if len(per_frame) else np.zeros((0, 4), np.int32))
        rects = np.ascontiguousarray(rects, np.int32)
        nf, nr = C.c_int(0), C.c_int(0)
        cap_f, cap_r = max(len(per_frame) * self.world, 1), max(len(rects) * self.world, 16)
        out = np.empty((cap_r, 4), np.int32)
        off = np.empty(cap_f + 1, np.int32)
        st = L.lib().cc_gather_detections(self._c, rects.ctypes.data_as(C.c_void_p), offsets.ctypes.data_as(C.c_void_p), len(per_frame),
                                          out.ctypes.data_as(C.c_void_p), cap_r, off.ctypes.data_as(C.c_void_p), cap_f, C.byref(nf), C.byref(nr))
        if st == L.CC_ERR_BUFFER_TOO_SMALL:  # the collectives are done: fetch the kept result, do not gather again
            out = np.empty((max(nr.value, 1), 4), np.int32)
            off = np.empty(nf.value + 1, np.int32)
            st = L.lib().cc_gather_fetch(self._c, out.ctypes.data_as(C.c_void_p), len(out), off.ctypes.data_as(C.c_void_p), nf.value)
        L.check(st)
        return [out[off[f]:off[f + 1]].copy() for f in range(nf.value)]

    def close(self):
        from . import _lib as L
        if getattr(self, "_c", None):
            L.lib().cc_comm_destroy(self._c)
            self._c = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


def gather_detections(per_frame: list[np.ndarray], device=None, group=None, comm: Comm | None = None) -> list[np.ndarray]:
    """per_frame: this rank's list of (k_i, 4) int32 rectangle arrays, frames in global order within the rank's shard.
    Returns the concatenation over ranks (rank order = global frame order under shard_range), on every rank.
    Two collectives: all_gather of [n_frames, n_rects] headers, then one padded all_gather of a flat int32 payload
    (per-frame counts followed by rectangles). Payloads are KB-sized: latency-bound, so one message per rank.
    With `comm` (a Comm) the exchange is the C ABI's cc_gather_detections on RCCL; without, the same protocol on
    torch.distributed (any backend: the CPU tests use gloo)."""
    if comm is not None:
        return comm.gather_all(per_frame)
    import torch
    import torch.distributed as dist

    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return [np.asarray(r, np.int32).reshape(-1, 4) for r in per_frame]
    world = dist.get_world_size(group)
    dev = device if device is not None else torch.device("cpu")
    counts = np.array([len(r) for r in per_frame], np.int32)
    flat = np.concatenate([counts] + [np.asarray(r, np.int32).reshape(-1) for r in per_frame]) if len(per_frame) else np.zeros(0, np.int32)
    header = torch.tensor([len(per_frame), int(counts.sum())], dtype=torch.int32, device=dev)
    headers = [torch.empty_like(header) for _ in range(world)]
    dist.all_gather(headers, header, group=group)
    headers = [h.cpu().numpy() for h in headers]
    max_len = max(int(h[0]) + 4 * int(h[1]) for h in headers)
    payload = torch.zeros(max(max_len, 1), dtype=torch.int32, device=dev)
    if len(flat):
        payload[: len(flat)] = torch.from_numpy(flat).to(dev)
    payloads = [torch.empty_like(payload) for _ in range(world)]
    dist.all_gather(payloads, payload, group=group)
    out: list[np.ndarray] = []
    for h, p in zip(headers, payloads):
        nf = int(h[0])
        p = p.cpu().numpy()
        cnt = p[:nf]
        o = nf
        for c in cnt:
            out.append(p[o:o + 4 * int(c)].reshape(-1, 4).copy())
            o += 4 * int(c)
    return out


# ---- training side: variables shard across ranks ------------------------------------------------
_SPLIT_WORDS = 13  # found, var_idx, quality bits, ord_c bits, split_point, subset[8]


def pick_split(per_shard: list[dict]) -> dict:
    """The reference scans variables in catalog order and keeps the first one whose float quality is strictly larger than
    the best so far (o_cvdtree.cpp:320-342), i.e. the first variable with the largest float quality. Shards are
    contiguous catalog ranges in rank order, so the global winner is the shard result with the largest quality, the
    lowest rank winning ties."""
    best = None
    for s in per_shard:
        if s["found"] and (best is None or s["quality"] > best["quality"]):
            best = s
    return best if best is not None else dict(per_shard[0], found=False)


def find_best_split_sharded(evaluator, weights, device=None, group=None, **kw) -> dict:
    """evaluator has presorted this rank's variable range (CvFeatureEvaluator.presort(n, *shard_range(F, rank, world)));
    every rank passes the same node. One all_gather of 52 bytes per rank (the arg-max exchange of SURVEY §8e)."""
    import torch
    import torch.distributed as dist

    local = evaluator.find_best_split(weights, **kw)
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    dev = device if device is not None else torch.device("cpu")
    words = np.zeros(_SPLIT_WORDS, np.int32)
    words[0], words[1], words[4] = int(local["found"]), local["var_idx"], local["split_point"]
    words[2] = np.float32(local["quality"]).view(np.int32)
    words[3] = np.float32(local["ord_c"]).view(np.int32)
    words[5:] = local["subset"]
    mine = torch.from_numpy(words).to(dev)
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine, group=group)
    shards = []
    for p in parts:
        p = p.cpu().numpy()
        shards.append({"found": bool(p[0]), "var_idx": int(p[1]), "quality": p[2:3].view(np.float32)[0], "ord_c": p[3:4].view(np.float32)[0],
                       "split_point": int(p[4]), "subset": p[5:].copy()})
    return pick_split(shards)
