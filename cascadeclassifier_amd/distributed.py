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


def gather_detections(per_frame: list[np.ndarray], device=None, group=None) -> list[np.ndarray]:
    """per_frame: this rank's list of (k_i, 4) int32 rectangle arrays, frames in global order within the rank's shard.
    Returns the concatenation over ranks (rank order = global frame order under shard_range), on every rank.
    Two collectives: all_gather of [n_frames, n_rects] headers, then one padded all_gather of a flat int32 payload
    (per-frame counts followed by rectangles). Payloads are KB-sized: latency-bound, so one message per rank."""
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
