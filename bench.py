#!/usr/bin/env python3
"""Headline benchmark: detection Mwindows/s on synthetic 1920x1080 frames with the (synthetic, stock-profile) Haar
frontal-face cascade, full scale pyramid (scaleFactor 1.1, minNeighbors 3) — BASELINE.json configs[1].

A step = one pass of the whole detection path over a batch of frames that is already resident in HBM: pyramid, integral
images, cascade evaluation, skip-rule filter, copy-back of the candidates, rectangle grouping, and (N > 1) the gather of
the detections over RCCL. One process per GPU; frames shard across ranks (weak scaling: --frames per GPU per step).

Prints ONE JSON line on rank 0 (see the driver contract); `roofline` prices the cascade-evaluation kernel against the
HBM roofline with the algorithmic byte count of SURVEY.md §8d; `cpu_baseline` times the CPU oracle (a restatement of
the reference path; OpenCV itself is not installed) on the host cores, rank 0 at N = 1 only.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def make_frames(n, w, h, seed0, content="natural"):
    from tests.util import frame_natural, frame_uniform, upscale
    tm = np.load(os.path.join(ROOT, "data", "face_template_24x24.npy"))
    frames = np.empty((n, h, w), np.uint8)
    for i in range(n):
        img = frame_natural(w, h, seed0 + i) if content == "natural" else frame_uniform(w, h, seed0 + i)
        rng = np.random.default_rng(10_000 + seed0 + i)
        for k in (1.0, 1.6, 2.7, 4.5, 8.0):
            s = int(24 * k)
            y, x = int(rng.integers(0, h - s)), int(rng.integers(0, w - s))
            img[y:y + s, x:x + s] = upscale(tm, s)
        frames[i] = img
    return frames


def usable_cores():
    """Host cores this process may really use: the affinity mask, capped by the cgroup CPU quota if there is one."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def kernel_source_sha16():
    """Identifies the cascade-kernel sources a counter profile belongs to (first 16 hex digits of their sha256)."""
    import hashlib
    h = hashlib.sha256()
    for name in ("cc_eval_kernel.inc", "cc_detect.hip"):
        with open(os.path.join(ROOT, "cascadeclassifier_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def corner_reads_frame0(clf, frame, scale_factor):
    """Integral-image corner reads of the reference algorithm on one frame: for every window the scan visits, 4 for the
    variance rectangle (Haar) plus, for every stump of every stage the window reaches, 4 per rectangle (LBP: 16)."""
    m = clf.model()
    codes, sums, vis = clf.debug_windows(frame, scale_factor)
    ns = len(m.stage_first)
    if m.info["feature_type"] == 0:
        nrect = 2 + (m.weights[:, 2] != 0).astype(np.int64)
        per_weak = 4 * nrect[m.stump_feature]
        base = 4
    else:
        per_weak = np.full(len(m.stump_feature), 16, np.int64)
        base = 0
    per_stage = np.array([per_weak[m.stage_first[s]:m.stage_first[s] + m.stage_ntrees[s]].sum() for s in range(ns)], np.int64)
    cum = np.concatenate([[0], np.cumsum(per_stage)])  # cum[k] = reads of stages 0 .. k-1
    # result codes: 1 = passed every stage, 0 = rejected by stage 0, -k = rejected by stage k (k >= 1); a Haar window that
    # fails the variance test is reported as -1 with stage sum exactly 0 and reaches no stage
    stages_run = np.where(codes == 1, ns, np.where(codes == 0, 1, -codes + 1)).astype(np.int64)
    if m.info["feature_type"] == 0:
        stages_run[(codes == -1) & (sums == 0.0)] = 0
    reads = base + cum[np.clip(stages_run, 0, ns)]
    return int(reads[vis != 0].sum())


def cascade_kernel_launches(tm, spec, feature_type, plan, chan_bytes):
    """The cascade kernel's launches of one pass as a kernel trace lists them, with each one's average duration (HIP events
    of this run) and algorithmic bytes per frame (every integral entry of the scales it covers, once). A run-time
    specialised Haar kernel is one module per step: k_eval_spec_step2 over the tiles of STEP-2 scales, then k_eval_spec_step1."""
    passes = max(tm["eval_launches"], 1)
    total = tm["eval_ms"] / passes
    px = (plan["w"] + 1).astype(np.int64) * (plan["h"] + 1)
    b1, b2 = int(px[plan["ystep"] == 1].sum()) * chan_bytes, int(px[plan["ystep"] == 2].sum()) * chan_bytes
    s1 = tm.get("eval_step1_ms", 0.0) / passes
    if spec and s1 > 0:
        return "k_eval_spec_step2+k_eval_spec_step1", [("k_eval_spec_step2", total - s1, b2), ("k_eval_spec_step1", s1, b1)]
    name = "k_eval_spec" if spec else ("k_eval_haar" if feature_type == 0 else "k_eval_lbp")
    return name, [(name, total, b1 + b2)]


PROFILE_ROUND = "r04"  # profiles/<round>_traffic_k_eval*.json, profiles/<round>_pmc_eval*.json


def replay_counters(suffix, kernel_name, cascade_file, frames_per_launch, full_hd):
    """Counter-based figures (HBM traffic, LDS / VALU busy, wait shares) need their own rocprofv3 passes (tools/profile_bench.sh,
    tools/pmc_eval.sh): a bench run can only REPLAY the committed measurement, and does so only when that profile was taken
    from the very kernel sources this run executes (sha of the kernel files recorded next to the numbers), for the same
    kernel and cascade. Returns (traffic bytes per launch or None, note, secondary dict or None)."""
    src_sha = kernel_source_sha16()
    traffic, note, secondary = None, None, None
    tname = f"profiles/{PROFILE_ROUND}_traffic_k_eval{suffix}.json"
    tfile = os.path.join(ROOT, tname)
    if os.path.exists(tfile) and full_hd:
        tj = json.load(open(tfile))
        if tj.get("kernel_src_sha16") == src_sha and tj.get("kernel") == kernel_name and tj.get("cascade", cascade_file) == cascade_file:
            traffic = round(tj["hbm_bytes_per_frame"] * frames_per_launch)
            note = {"replayed_from": tname, "profile_kernel_src_sha16": src_sha,
                    "how": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, own passes, FETCH x2 (gfx950), scaled to this run's frames per launch"}
        else:
            note = {"refused": f"{tname} was measured on other kernel sources, another kernel or another cascade "
                               f"({tj.get('kernel')}, {tj.get('cascade')}, {tj.get('kernel_src_sha16')} vs {kernel_name}, {cascade_file}, {src_sha}): "
                               "re-run tools/profile_bench.sh"}
    pname = f"profiles/{PROFILE_ROUND}_pmc_eval{suffix}.json"
    pfile = os.path.join(ROOT, pname)
    if os.path.exists(pfile) and full_hd:
        pj = json.load(open(pfile))
        if pj.get("kernel_src_sha16") == src_sha and pj.get("kernel") == kernel_name and pj.get("cascade", cascade_file) == cascade_file:
            secondary = {k: pj.get(k) for k in ("lds_pipeline_busy", "lds_bank_conflict_share", "valu_busy", "valu_active_counter_share",
                                                "wave_cycles_waiting_on_counter_or_barrier", "wave_cycles_ready_not_issued", "wave_cycles_issuing")}
            secondary["replayed_from"] = pname + " (tools/pmc_eval.sh)"
            secondary["profile_kernel_src_sha16"] = src_sha
    return traffic, note, secondary


def measure_extra_workload(cc, torch, dev, dev_index, cascade, specialize, frames_host, args, label):
    """One of the post-headline workloads (rank 0, N = 1; outside the headline's timed region): the same step as the
    headline -- resident frames, whole detection path incl. copy-back + grouping -- for another cascade / frame content.
    Returns value, kernel time, HBM and LDS roofline fractions, and whether the CPU oracle's rectangles agree on frame 0."""
    import numpy as np
    B, H, W = frames_host.shape
    frames = torch.from_numpy(frames_host).to(dev)
    clf = cc.CascadeClassifier(cascade, device=dev_index, max_batch=B)
    inf = clf.info()
    spec = 0
    if specialize > 0 and inf["max_nodes_per_tree"] == 1:
        try:
            spec = clf.specialize(specialize)
        except cc.CascadeError as e:
            print(f"[bench] {label}: specialisation unavailable: {e}", file=sys.stderr)
    plan = cc.scale_plan(inf["win_w"], inf["win_h"], W, H, args.scale_factor)
    wpf = int((plan["nx"].astype(np.int64) * plan["ny"]).sum())
    ipx = int(((plan["w"] + 1).astype(np.int64) * (plan["h"] + 1)).sum())
    bytes_frame = (8 if inf["feature_type"] == 0 else 4) * ipx

    def run(k):  # the headline's step form: pipelined submit / collect unless --sync-steps
        out, prev = None, None
        for _ in range(k):
            if args.sync_steps:
                out = clf.detect_batch(None, args.scale_factor, args.min_neighbors, device_ptr=frames.data_ptr(), shape=(B, H, W))
                continue
            t = clf.detect_batch_submit(None, args.scale_factor, args.min_neighbors, device_ptr=frames.data_ptr(), shape=(B, H, W))
            if prev is not None:
                out = clf.detect_batch_collect(prev)
            prev = t
        return clf.detect_batch_collect(prev) if prev is not None else out
    run(1)
    clf.set_profiling(True)
    clf.timings(reset=True)
    torch.cuda.synchronize()
    steps = max(2, min(args.steps, 6))
    t0 = time.perf_counter()
    last = run(steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tm = clf.timings(reset=True)
    clf.set_profiling(False)
    eval_ms = tm["eval_ms"] / max(tm["eval_launches"], 1)
    fpl = tm["frames"] / max(tm["eval_launches"], 1)
    ach = bytes_frame * fpl / (eval_ms * 1e-3) / 1e9 if eval_ms > 0 else 0.0
    reads0 = corner_reads_frame0(clf, frames_host[0], args.scale_factor)
    lds_ach = reads0 * 4 * fpl / (eval_ms * 1e-3) / 1e9 if eval_ms > 0 else 0.0
    from oracle import oracle as orc
    want = orc.detect_multiscale(orc.load_cascade_xml(cascade), frames_host[0], args.scale_factor, args.min_neighbors, nthreads=usable_cores())
    same = want.shape == last[0].shape and bool((want == last[0]).all())
    kernel_name, _ = cascade_kernel_launches(tm, spec, inf["feature_type"], plan, 8 if inf["feature_type"] == 0 else 4)
    full_hd = (W, H) == (1920, 1080) and abs(args.scale_factor - 1.1) < 1e-12 and args.content == "natural" and "uniform" not in label
    traffic, traffic_note, secondary = replay_counters("" if inf["feature_type"] == 0 else "_lbp", kernel_name, os.path.basename(cascade), fpl, full_hd)
    del clf
    return {"workload": label, "value": round(wpf * B * steps / dt / 1e6, 3), "unit": "Mwindows/s", "ms_per_step": round(dt / steps * 1e3, 4),
            "steps": steps, "frames_per_step": B, "kernel_specialized_stages": spec, "cascade_kernel_ms_per_launch": round(eval_ms, 4),
            "frames_per_launch": fpl, "roofline_frac_hbm": round(ach / HBM_PEAK_GBS, 5), "achieved_GBps": round(ach, 2),
            "lds_frac": round(lds_ach / (256 * 128 * 2.4), 4), "traffic": traffic, "traffic_source": traffic_note, "secondary": secondary,
            "rectangles_identical_to_cpu_oracle_frame0": same}


def measure_training_eval(cc, torch):
    """BASELINE configs[4] beside the headline (rank 0, N = 1): the trainer's bulk feature evaluation -- the Haar BASIC catalog
    of a 24x24 window (162 336 features) over 10 000 positives + 10 000 negatives, written as the out[feature][sample] matrix
    CvCascadeBoostTrainData::precalculate fills (o_cvcascadeboosttraindata.cpp:582-596), in row blocks into one resident
    buffer. Kernel time by HIP events; one row block is compared bit for bit with the CPU oracle."""
    import numpy as np
    from cascadeclassifier_amd import evaluator as ev
    rng = np.random.default_rng(7)
    n = 10000
    tmpl = rng.integers(0, 256, (24, 24)).astype(np.float64)
    pos = np.clip(np.rint(tmpl + rng.normal(0, 15, (n, 24, 24))), 0, 255).astype(np.uint8)
    neg = rng.integers(0, 256, (n, 24, 24), dtype=np.uint8)
    imgs, labels = np.concatenate([pos, neg]), np.concatenate([np.ones(n, np.uint8), np.zeros(n, np.uint8)])
    N = len(imgs)
    e = cc.CvFeatureEvaluator.create(ev.HAAR)
    e.init(cc.CvFeatureParams(ev.HAAR, ev.BASIC), N, (24, 24))
    nfeat = e.getNumFeatures()
    e.setImages(imgs, labels)
    block = 32768
    out = torch.empty((block, N), dtype=torch.float32, device="cuda")
    e.calc_batch_device(0, block, out.data_ptr(), n_samples=N)  # warm-up
    torch.cuda.synchronize()
    kernel_ms, launches = 0.0, 0
    t0 = time.perf_counter()
    for f0 in range(0, nfeat, block):
        e.calc_batch_device(f0, min(f0 + block, nfeat), out.data_ptr(), n_samples=N)
        kernel_ms += e.last_kernel_ms()
        launches += 1
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    f0, f1 = 100000, 100000 + 512
    e.calc_batch_device(f0, f1, out.data_ptr(), n_samples=N)
    torch.cuda.synchronize()
    got = out[: f1 - f0].cpu().numpy()
    from oracle import oracle as orc
    s, t, nf = orc.set_images(imgs, want_tilted=False)
    want = orc.haar_eval_batch(orc.haar_catalog(24, 24, ev.BASIC), f0, f1, s, t, nf, 24, 24)
    evals = nfeat * N
    return {"workload": f"CvHaarEvaluator bulk evaluation for one boosting stage: Haar BASIC {nfeat} features x {N} samples of 24x24 -- BASELINE configs[4]",
            "value": round(evals / (kernel_ms * 1e-3) / 1e9, 2), "unit": "G feature evaluations/s (kernel time)", "kernel_ms": round(kernel_ms, 3),
            "launches": launches, "wall_ms": round(wall * 1e3, 3), "hbm_write_GBps": round(evals * 4 / (kernel_ms * 1e-3) / 1e9, 1),
            "roofline_frac_hbm": round(evals * 4 / (kernel_ms * 1e-3) / (HBM_PEAK_GBS * 1e9), 4),
            "values_identical_to_cpu_oracle_rows": [f0, f1], "values_identical_to_cpu_oracle": bool((got.view(np.uint32) == want.view(np.uint32)).all())}


def measure_split_search(cc, kind):
    """SURVEY 8f-2 beside the headline (rank 0, N = 1): one node's best-split search over every variable of a 24x24 catalog
    (Haar BASIC: 162 336 ordered variables; LBP: 8 464 categorical ones) x 20 000 samples, after one presort -- the call that
    replaces CvDTree::find_best_split (o_cvdtree.cpp:313-357). Gentle AdaBoost's regression search; the winner is compared with
    the CPU oracle's search over the 256 variables around it."""
    import numpy as np
    from cascadeclassifier_amd import evaluator as ev
    from oracle import oracle as orc
    rng = np.random.default_rng(7)
    n = 10000
    tmpl = rng.integers(0, 256, (24, 24)).astype(np.float64)
    pos = np.clip(np.rint(tmpl + rng.normal(0, 40, (n, 24, 24))), 0, 255).astype(np.uint8)
    neg = np.clip(np.rint(0.5 * tmpl + 0.5 * rng.integers(0, 256, (24, 24)) + rng.normal(0, 40, (n, 24, 24))), 0, 255).astype(np.uint8)
    imgs, labels = np.concatenate([pos, neg]), np.concatenate([np.ones(n, np.uint8), np.zeros(n, np.uint8)])
    N = len(imgs)
    ftype = ev.HAAR if kind == "HAAR" else ev.LBP
    e = cc.CvFeatureEvaluator.create(ftype)
    e.init(cc.CvFeatureParams(ftype, ev.BASIC), N, (24, 24))
    e.setImages(imgs, labels)
    t0 = time.perf_counter()
    e.presort()
    presort_s = time.perf_counter() - t0
    resp = (labels.astype(np.float32) * 2 - 1)
    w = rng.random(N) + 0.05
    w /= w.sum()
    tot = float(np.cumsum(w)[-1])
    W = np.concatenate([w, [tot, 0.0]])
    nv = float(np.cumsum(resp * w)[-1] * (1.0 / tot))
    e.find_best_split(W, responses=resp, node_value=nv)  # warm-up
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        got = e.find_best_split(W, responses=resp, node_value=nv)
    wall = (time.perf_counter() - t0) / reps
    F = e.getNumFeatures()
    lo = max(0, min(got["var_idx"] - 128, F - 256))
    s, t, nf = orc.set_images(imgs, want_tilted=False, want_norm=ftype == ev.HAAR)
    if ftype == ev.HAAR:
        vals = orc.haar_eval_batch(orc.haar_catalog(24, 24, ev.BASIC), lo, lo + 256, s, t, nf, 24, 24)
    else:
        vals = orc.lbp_eval_batch(orc.lbp_catalog(24, 24), lo, lo + 256, s, 24, 24)
    want = orc.find_best_split(vals, W, categorical=ftype == ev.LBP, responses=resp, node_value=nv)
    same = bool(bool(want["found"]) and int(want["var_idx"]) + lo == int(got["var_idx"]) and float(want["quality"]) == float(got["quality"]))
    return {"workload": f"best-split search of one tree node (SURVEY 8f-2): {'Haar BASIC' if ftype == ev.HAAR else 'LBP'} 24x24, {F} variables x {N} samples, Gentle AdaBoost",
            "value": round(wall * 1e3, 3), "unit": "ms per node (whole cc_eval_find_best_split call)", "kernel_ms": round(e.last_kernel_ms(), 3),
            "presort_s_per_stage": round(presort_s, 3), "winner_identical_to_cpu_oracle_over_256_variables_around_it": same,
            "values_identical_to_cpu_oracle": same}


def measure_negative_mining(cc):
    """SURVEY 8f-1 beside the headline (rank 0, N = 1): the reader's window stream of 32 background images per call
    (cc_negminer_run_batch) through the first 10 stages of the headline cascade; image 0's flags are compared with the CPU
    oracle's window-by-window loop."""
    import subprocess
    import tempfile
    import numpy as np
    from cascadeclassifier_amd import evaluator as ev
    from oracle import oracle as orc
    from tests.util import frame_natural
    xml = os.path.join(tempfile.mkdtemp(), "trunc.xml")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "truncate_cascade.py"),
                           os.path.join(ROOT, "data", "haarcascade_frontalface_synthetic.xml"), "10", xml])
    m = ev.NegativeMiner(cc.CascadeClassifier(xml))
    res = {}
    same = True
    for (w, h) in ((640, 480), (1920, 1080)):
        imgs = [frame_natural(w, h, 100 + k) for k in range(32)]
        per = m.plan(w, h)["n_windows"]
        flags = m.run_batch(imgs, max_keep=256)[0]  # warm-up
        reps = 5
        t0 = time.perf_counter()
        for _ in range(reps):
            flags = m.run_batch(imgs, max_keep=256)[0]
        dt = (time.perf_counter() - t0) / reps
        if (w, h) == (640, 480):
            same = bool((flags[0] == orc.negmine_image(orc.load_cascade_xml(xml), imgs[0], 0, 0, max_keep=1)[0]).all())
        res[f"{w}x{h}"] = {"ms_per_image": round(dt / 32 * 1e3, 4), "mwindows_per_s": round(per * 32 / dt / 1e6, 1), "windows_per_image": per}
    return {"workload": "batched negative mining (SURVEY 8f-1): 32 background images per call, 10 trained stages of the headline cascade",
            "value": res["1920x1080"]["mwindows_per_s"], "unit": "M stream windows/s (1920x1080 backgrounds, wall time incl. the images' way to the device)",
            "by_image_size": res, "values_identical_to_cpu_oracle": same}


def spawn_ranks(n):
    """Starts n copies of this script (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* set, 127.0.0.1 rendezvous on a free port) and
    waits for them. Returns the first non-zero exit status, or 0. If one rank dies the others are terminated, so a
    failed rendezvous cannot hang the caller."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    status = 0
    alive = set(range(n))
    while alive:
        for r in sorted(alive):
            rc = procs[r].poll()
            if rc is None:
                continue
            alive.discard(r)
            if rc != 0 and status == 0:
                status = rc if rc > 0 else 1
                for q in alive:
                    procs[q].terminate()
        time.sleep(0.05)
    print("[bench] rank exit codes: " + json.dumps([p.returncode for p in procs]), file=sys.stderr)
    return status


def setup_gather(args, rank, world, dev_index, comm_dev):
    """N > 1: the gather of detections runs on the C ABI's own RCCL communicator (cc_comm_* / cc_gather_detections, what a
    C++ host uses); torch.distributed only carries the 128-byte unique id to the other ranks. The torch collective with
    the same protocol stays as the fallback (and is what the gloo rehearsal on a one-GPU box uses). Returns (comm or None,
    description). Touches no GPU by itself (the RCCL transport does, the loopback TCP one does not)."""
    import torch
    import torch.distributed as dist

    from cascadeclassifier_amd.distributed import gather_detections
    comm = None
    gather_kind = "none (single rank)"
    if world > 1:
        gather_kind = f"torch.distributed all_gather ({args.backend})"
        # (CCAMD_COMM_TRANSPORT=tcp: the same C-ABI calls over the library's loopback transport -- the rehearsal of this leg
        # on a one-GPU box, where RCCL refuses two ranks on one device)
        tcp = os.environ.get("CCAMD_COMM_TRANSPORT") == "tcp"
        if (args.backend == "nccl" or tcp) and not os.environ.get("CCAMD_BENCH_TORCH_GATHER"):
            from cascadeclassifier_amd.distributed import Comm
            # ncclCommInitRank / ncclAllGather with more than one rank have never run on hardware (no multi-GPU box so far): both
            # steps run under a watchdog, so that a rank stuck inside them makes EVERY rank fall back to the torch collective
            # (the verdict is an all_reduce over the ranks) instead of hanging the measurement.
            limit = float(os.environ.get("CCAMD_BENCH_COMM_TIMEOUT_S", "90"))

            def guarded(fn):
                import threading
                box = {}

                def body():
                    try:
                        box["value"] = fn()
                    except Exception as e:  # noqa: BLE001
                        box["error"] = e
                th = threading.Thread(target=body, daemon=True)
                th.start()
                th.join(limit)
                if th.is_alive():
                    return None, TimeoutError(f"no answer within {limit:.0f} s")
                return box.get("value"), box.get("error")
            ok = torch.zeros(1, dtype=torch.int32, device=comm_dev)
            comm, err = guarded(lambda: Comm.from_torch(dev_index))
            if err is None and comm is not None:
                ok += 1
            else:
                print(f"[bench] rank {rank}: cc_comm_create failed ({err}); gathering with torch.distributed", file=sys.stderr)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)  # all ranks or none
            if int(ok.item()) == 1:
                # one trial gather before anything is timed; if it fails on any rank, every rank falls back to the torch collective
                ok.fill_(0)
                trial, err = guarded(lambda: gather_detections([np.array([[rank, 0, 1, 1]], np.int32)], device=comm_dev, comm=comm))
                if err is None and trial is not None and len(trial) == world and all(len(t) == 1 and int(t[0][0]) == r for r, t in enumerate(trial)):
                    ok += 1
                else:
                    print(f"[bench] rank {rank}: cc_gather_detections failed ({err}); gathering with torch.distributed", file=sys.stderr)
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 1:
                gather_kind = "cc_gather_detections (C ABI, " + ("loopback TCP transport" if tcp else "RCCL ncclAllGather x2") + ")"
            else:
                if comm is not None:
                    guarded(comm.close)
                comm = None
    return comm, gather_kind


def rehearse_gather(args):
    """`--rehearse-gather`: the N > 1 leg of this script without a detector and without a GPU -- what can be run of BASELINE
    configs[3] (8 ranks x 64 frames) where there is no 8-GPU node. Every rank makes up the rectangles of its frames (frame f
    gets f % 7 rectangles derived from f; one rank has none), the ranks rendezvous, build the communicator, run the trial
    gather and `--steps` timed gathers with the barriers and the max-over-ranks reduction of the real run, and every rank
    checks that it holds all frames' rectangles in frame order. Rank 0 prints ONE line; it carries no `metric` / `value`."""
    import torch
    import torch.distributed as dist

    from cascadeclassifier_amd.distributed import gather_detections
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE {world}"
    assert args.backend != "nccl", "--rehearse-gather runs without GPUs: use --backend gloo"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(args.backend)
    comm_dev = torch.device("cpu")
    comm, gather_kind = setup_gather(args, rank, world, 0, comm_dev)
    B = args.frames

    def fake(f, owner):
        k = 0 if owner == world - 3 else f % 7
        return np.array([[f, j, 24 + j, 24 + f % 5] for j in range(k)], np.int32).reshape(-1, 4)
    mine = [fake(rank * B + i, rank) for i in range(B)]
    want = [fake(r * B + i, r) for r in range(world) for i in range(B)]

    def sync():
        if world > 1:
            dist.barrier()
    ok = True
    for _ in range(args.warmup):
        gather_detections(mine, device=comm_dev, comm=comm)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        got = gather_detections(mine, device=comm_dev, comm=comm)
        ok = ok and len(got) == len(want) and all(a.shape == b.shape and (a == b).all() for a, b in zip(got, want))
    sync()
    dt = time.perf_counter() - t0
    flag = torch.tensor([dt, 0.0 if ok else 1.0], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
    if comm is not None:
        comm.close()
    if rank == 0:
        print(json.dumps({"rehearsal": "gather of detections only: no detector, no GPU, made-up rectangles", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "frames_per_rank_per_step": B, "frames_per_step": B * world, "gather": gather_kind,
                          "gather_ms_per_step": round(float(flag[0]) / max(args.steps, 1) * 1e3, 4),
                          "every_rank_holds_all_frames_in_order": bool(flag[1] == 0.0)}))
    if world > 1:
        dist.destroy_process_group()
    return 0 if float(flag[1]) == 0.0 else 4


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=64, help="frames per GPU per step (64 = BASELINE configs[3]: 512 frames over 8 GPUs)")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--cascade", default=os.path.join(ROOT, "data", "haarcascade_frontalface_synthetic.xml"))
    ap.add_argument("--scale-factor", type=float, default=1.1)
    ap.add_argument("--min-neighbors", type=int, default=3)
    ap.add_argument("--content", choices=["natural", "uniform"], default="natural",
                    help="frame distribution of SURVEY 8d config 2: (ii) natural-like 1/f noise (default, the headline) or (i) "
                         "i.i.d. uniform noise (almost every window dies in stage 0-1)")
    ap.add_argument("--cpu-frames", type=int, default=4, help="frames of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--specialize", type=int, default=7, help="stages compiled into the cascade kernel at load time (hiprtc; 0 = "
                                                              "table-driven kernel only)")
    ap.add_argument("--device-only", action="store_true", help="time the device pipeline only (no copy-back/grouping)")
    ap.add_argument("--sync-steps", action="store_true", help="one synchronous cc_detect_batch call per step instead of pipelined "
                                                            "submit / collect (the headline of rounds 1-2)")
    ap.add_argument("--no-extra", action="store_true", help="skip the legs reported beside the headline (host split, host-frame "
                                                          "call shape, LBP cascade, uniform-noise frames); rank 0 at N = 1 only")
    ap.add_argument("--cpu-reps", type=int, default=5, help="repetitions of the CPU-baseline sample (median is reported)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse "
                                                      "the multi-rank path on a box with fewer GPUs than ranks)")
    ap.add_argument("--rehearse-gather", action="store_true", help="no detector, no GPU: every rank makes up the rectangles of its "
                                                                   "--frames frames and the ranks run the N > 1 leg only (rendezvous, communicator, "
                                                                   "trial gather, gathers, max-over-ranks timing); prints a line marked as a rehearsal")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` by itself: this process never touches the GPU; it starts one rank per GPU as child
        # processes (the env contract of torch.distributed.run) and exits with their status. Rank 0 prints the JSON line.
        sys.exit(spawn_ranks(args.gpus))
    if args.rehearse_gather:
        sys.exit(rehearse_gather(args))

    import torch
    import torch.distributed as dist

    import cascadeclassifier_amd as cc
    from cascadeclassifier_amd.distributed import gather_detections

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dev_index = local_rank if args.backend == "nccl" else local_rank % max(torch.cuda.device_count(), 1)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(dev_index)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(args.backend)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE {world}"
    dev = torch.device("cuda", dev_index)
    comm_dev = dev if args.backend == "nccl" else torch.device("cpu")

    W, H, B = args.width, args.height, args.frames
    frames_host = make_frames(B, W, H, seed0=rank * B, content=args.content)
    frames = torch.from_numpy(frames_host).to(dev)  # resident in HBM before the timed region
    clf = cc.CascadeClassifier(args.cascade, device=dev_index, max_batch=B)
    assert not clf.empty(), getattr(clf, "load_error", "")
    inf = clf.info()
    spec_stages = 0
    if args.specialize > 0 and inf["max_nodes_per_tree"] == 1:
        try:  # load-time work, outside the timed region; without hiprtc the table-driven kernel stays in use
            spec_stages = clf.specialize(args.specialize)
        except cc.CascadeError as e:
            print(f"[bench] specialisation unavailable: {e}", file=sys.stderr)
    plan = cc.scale_plan(inf["win_w"], inf["win_h"], W, H, args.scale_factor)
    windows_per_frame = int((plan["nx"].astype(np.int64) * plan["ny"]).sum())
    integral_px = int(((plan["w"] + 1).astype(np.int64) * (plan["h"] + 1)).sum())
    chan_bytes = 8 if inf["feature_type"] == 0 else 4
    eval_bytes_per_frame = chan_bytes * integral_px  # every integral entry read exactly once

    comm, gather_kind = setup_gather(args, rank, world, dev_index, comm_dev)

    def step():
        if args.device_only:
            clf.run_device_only(frames.data_ptr(), (B, H, W), args.scale_factor)
            return None
        rects = clf.detect_batch(None, args.scale_factor, args.min_neighbors, device_ptr=frames.data_ptr(), shape=(B, H, W))
        if world > 1:
            rects = gather_detections(rects, device=comm_dev, comm=comm)
        return rects

    def run_steps(k):
        """k steps. Default: the steps are PIPELINED through cc_detect_batch_submit / _collect -- step i + 1 is submitted
        before step i is collected, so its pyramid / integral work runs under the cascade kernel of step i and step i's
        copy-back and grouping under the device side of step i + 1 (the C ABI's form for streams of batches). Every step is
        submitted, collected and (N > 1) gathered inside the call: nothing is left in flight when it returns.
        --sync-steps: one synchronous cc_detect_batch per step."""
        if args.device_only or args.sync_steps:
            out = None
            for _ in range(k):
                out = step()
            return out
        out, prev = None, None
        for _ in range(k):
            t = clf.detect_batch_submit(None, args.scale_factor, args.min_neighbors, device_ptr=frames.data_ptr(), shape=(B, H, W))
            if prev is not None:
                out = clf.detect_batch_collect(prev)
                if world > 1:
                    out = gather_detections(out, device=comm_dev, comm=comm)
            prev = t
        if prev is not None:
            out = clf.detect_batch_collect(prev)
            if world > 1:
                out = gather_detections(out, device=comm_dev, comm=comm)
        return out

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    run_steps(args.warmup)
    clf.set_profiling(True)
    clf.timings(reset=True)
    sync()
    t0 = time.perf_counter()
    last = run_steps(args.steps)
    sync()
    dt = time.perf_counter() - t0
    tm = clf.timings(reset=True)
    clf.set_profiling(False)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # Where the step time goes that is not kernel time: the same step without copy-back and grouping (device pipeline
    # only), and the step fed from HOST frames (the reference's call shape, tools/detection/Cpp/main.cpp:27-45: the image is
    # in host memory; H2D copies included). Both outside the headline's timed region; rank 0, N = 1.
    extra_legs = rank == 0 and world == 1 and not args.device_only and not args.no_extra
    host_split = None
    host_frames_leg = None
    if extra_legs:
        k = max(2, min(args.steps, 5))
        clf.run_device_only(frames.data_ptr(), (B, H, W), args.scale_factor)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(k):
            clf.run_device_only(frames.data_ptr(), (B, H, W), args.scale_factor)
        torch.cuda.synchronize()
        dev_ms = (time.perf_counter() - t1) / k * 1e3
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(k):
            step()
        torch.cuda.synchronize()
        sync_ms = (time.perf_counter() - t1) / k * 1e3
        host_split = {"device_pipeline_only_ms_per_step": round(dev_ms, 4),
                      "synchronous_call_ms_per_step": round(sync_ms, 4),
                      "synchronous_call_value": round(windows_per_frame * B / (sync_ms * 1e-3) / 1e6, 3),
                      "host_ms_per_synchronous_step": round(sync_ms - dev_ms, 4),
                      "hidden_by_pipelined_steps_ms": round(sync_ms - dt / args.steps * 1e3, 4),
                      "what": "one synchronous cc_detect_batch per step, and the same step without candidate copy-back, host grouping and "
                              f"Python list building (device pipeline only), {k} steps each after the timed region. Their difference is what a "
                              "synchronous call spends on the host with the device idle; the headline's steps are submitted one ahead "
                              "(config.steps_are), which hides that and the first pass's unoverlapped pyramid / integral work. The device legs "
                              "of the passes overlap, so kernel_ms_per_step does not add up to either"}
        # Host frames, in the headline's own step form: step i + 1 is submitted (its frames copied into the detector's pinned
        # staging area and sent to the device on the front stream) before step i is collected.
        def host_steps(n_steps):
            out, prev = None, None
            for _ in range(n_steps):
                t = clf.detect_batch_submit(frames_host, args.scale_factor, args.min_neighbors)
                if prev is not None:
                    out = clf.detect_batch_collect(prev)
                prev = t
            return clf.detect_batch_collect(prev)
        host_last = host_steps(2)
        kh = max(k, args.steps)  # as many steps as the headline: the pipeline's ramp is amortised the same way
        t1 = time.perf_counter()
        host_last = host_steps(kh)
        hdt = (time.perf_counter() - t1) / kh
        clf.detect_batch(frames_host, args.scale_factor, args.min_neighbors)
        t1 = time.perf_counter()
        for _ in range(k):
            clf.detect_batch(frames_host, args.scale_factor, args.min_neighbors)
        hsync = (time.perf_counter() - t1) / k
        host_same = last is not None and len(host_last) == len(last) and all(a.shape == b.shape and (a == b).all() for a, b in zip(host_last, last))
        host_frames_leg = {"ms_per_step": round(hdt * 1e3, 4), "value": round(windows_per_frame * B / hdt / 1e6, 3), "unit": "Mwindows/s",
                           "synchronous_call_ms_per_step": round(hsync * 1e3, 4),
                           "synchronous_call_value": round(windows_per_frame * B / hsync / 1e6, 3),
                           "rectangles_identical_to_resident_frames": bool(host_same),
                           "what": f"the same steps with the {B} frames in pageable host memory (the reference's call shape, "
                                   "tools/detection/Cpp/main.cpp:27-45), pipelined through cc_detect_batch_submit / _collect like the headline: "
                                   "submit copies a pass's frames into the detector's pinned staging area (host threads) and issues ONE "
                                   f"asynchronous H2D copy per pass on the front stream ({kh} steps); synchronous_call_*: one cc_detect_batch per "
                                   "step from the same host frames; never the headline `value`"}
        if not host_same:
            print("bench.py: rectangles from host frames differ from the resident-frame run", file=sys.stderr)
            sys.exit(3)

    # windows the scan actually visits (grid windows minus the positions the stage-0 skip rule jumps over), frame 0,
    # outside the timed region (SURVEY.md §8d asks for it next to the grid count)
    visited0 = None
    if rank == 0 and not args.device_only and not os.environ.get("CCAMD_BENCH_NO_VISITED"):  # (one extra 1-frame launch)
        visited0 = int(clf.debug_windows(frames_host[0], args.scale_factor)[2].sum())
    total_windows = windows_per_frame * B * world * args.steps
    value = total_windows / dt / 1e6
    # average duration of one launch of the cascade kernel (HIP events on the detector's stream) and the frames one
    # launch covers (a step is cut into a few pipelined passes, one launch each)
    eval_ms = tm["eval_ms"] / max(tm["eval_launches"], 1)
    frames_per_launch = tm["frames"] / max(tm["eval_launches"], 1)
    ach = eval_bytes_per_frame * frames_per_launch / (eval_ms * 1e-3) / 1e9 if eval_ms > 0 else 0.0
    src_sha = kernel_source_sha16()
    kernel_name, kernel_launches = cascade_kernel_launches(tm, spec_stages, inf["feature_type"], plan, chan_bytes)
    full_hd = (W, H) == (1920, 1080) and abs(args.scale_factor - 1.1) < 1e-12
    lbp_suffix = "" if inf["feature_type"] == 0 else "_lbp"
    traffic, traffic_note, secondary = replay_counters(lbp_suffix, kernel_name, os.path.basename(args.cascade), frames_per_launch, full_hd)
    # Secondary limiter measured IN THIS RUN: the rectangle-corner gathers the reference algorithm performs on the windows
    # its scan visits (frame 0: 4 corners of the variance rectangle + 4 per rectangle of every stump the window reaches;
    # LBP: 16 per stump), priced against the LDS gather peak (ds_read_b32: 128 B/clk/CU x 256 CUs x 2.4 GHz).
    lds = None
    if rank == 0 and not args.device_only and not os.environ.get("CCAMD_BENCH_NO_VISITED") and eval_ms > 0:
        reads0 = corner_reads_frame0(clf, frames_host[0], args.scale_factor)
        lds_peak = 256 * 128 * 2.4  # GB/s
        lds_ach = reads0 * 4 * frames_per_launch / (eval_ms * 1e-3) / 1e9
        lds = {"corner_reads_frame0": reads0, "achieved": round(lds_ach, 1), "peak": round(lds_peak, 1), "unit": "GB/s",
               "frac": round(lds_ach / lds_peak, 4), "measured": "in this run (frame 0's result codes x per-stage corner counts, same launch time as `achieved`)"}
    out = {
        "metric": "detection Mwindows/sec (1080p, haarcascade_frontalface) + achieved HBM GB/s",
        "value": round(value, 3),
        # the same metric for one synchronous cc_detect_batch call per step -- what rounds 1-2 reported as `value` (round 3 and
        # later report pipelined steps, config.steps_are): use this one for trends across rounds and against the synchronous CPU
        # baseline. Measured after the timed region (rank 0, N = 1); null where that leg does not run.
        "value_synchronous": (round(value, 3) if args.sync_steps else (host_split["synchronous_call_value"] if host_split else None)),
        "unit": "Mwindows/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "i32+f32 (f64 stage sums)" if inf["feature_type"] == 0 else "i32 (f64 stage sums)",
        "data": "synthetic",
        "config": {
            "workload": f"{W}x{H} {'Haar' if inf['feature_type'] == 0 else 'LBP'} frontalface detection, full scale pyramid (scaleFactor {args.scale_factor}, minNeighbors "
                        f"{args.min_neighbors}), {len(plan)} scales, {windows_per_frame} grid windows/frame",
            "cascade": os.path.basename(args.cascade) + f" ({inf['n_stages']} stages, {inf['n_weak']} weak classifiers"
                       + ("; synthetic, stock 25-stage/2913-stump profile)" if "synthetic" in os.path.basename(args.cascade) else ")"),
            "visited_windows_frame0": visited0,
            "kernel_specialized_stages": spec_stages,
            "frames_per_gpu_per_step": B,
            "frame_content": ("1/f noise (sigma 40)" if args.content == "natural" else "i.i.d. uniform noise") + " + 5 pasted face templates",
            "parallelism": f"frames sharded over {world} GPU(s); RCCL gather of detections only",
            "gather": gather_kind,
            "steps_are": "device pipeline only" if args.device_only else ("synchronous cc_detect_batch calls" if args.sync_steps else
                         "pipelined: cc_detect_batch_submit(step i + 1) before cc_detect_batch_collect(step i); every step submitted, "
                         "collected and gathered inside the timed region"),
            "timed_region": "device pipeline only" if args.device_only else
                            "pyramid+integral+cascade eval+skip filter+candidate copy-back+host grouping" + ("+RCCL gather" if world > 1 else ""),
        },
        "frames_per_s": round(B * world * args.steps / dt, 2),
        "kernel_ms_per_step": {k: round(tm[k] / args.steps, 4) for k in ("resize_ms", "integral_ms", "eval_ms", "finalize_ms")},
        "host_split": host_split,
        "host_frames": host_frames_leg,
        "roofline": {
            "kernel": kernel_name,
            # one entry per cascade-kernel launch of a pass, named as a kernel trace names it (rocprofv3 --stats lists them
            # separately: its averages are these): the object's own achieved / avg_launch_ms are the pass's, i.e. their sum
            "kernels": [{"kernel": n, "avg_launch_ms": round(ms, 4), "algorithmic_bytes_per_launch": round(b * frames_per_launch),
                         "achieved": round(b * frames_per_launch / (ms * 1e-3) / 1e9, 2) if ms > 0 else None,
                         "frac": round(b * frames_per_launch / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if ms > 0 else None}
                        for n, ms, b in kernel_launches],
            "bound": "hbm",
            "achieved": round(ach, 2),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(ach / HBM_PEAK_GBS, 5),
            "traffic": traffic,
            "traffic_source": traffic_note,
            "algorithmic_bytes_per_launch": round(eval_bytes_per_frame * frames_per_launch),
            "frames_per_launch": frames_per_launch,
            "avg_launch_ms": round(eval_ms, 4),
            "kernel_src_sha16": src_sha,
            "bound_is": "the roofline SURVEY.md 8d prices this kernel against (HBM bytes: every integral entry once); it is NOT what "
                        "limits the kernel",
            "practical_limiter": "VALU issue and LDS corner gathers (about 50 stump evaluations per window): see lds, secondary and DESIGN.md 4.4",
            "lds": lds,
            "secondary": secondary,
        },
    }
    if rank == 0 and world == 1 and args.cpu_frames > 0:
        from oracle import oracle as orc
        o = orc.load_cascade_xml(args.cascade)
        cores = usable_cores()
        nfr = min(args.cpu_frames, B)
        orc.detect_multiscale(o, frames_host[0][:270, :480], args.scale_factor, args.min_neighbors, nthreads=cores)  # warm-up
        ok = True
        times = []
        for rep in range(max(1, args.cpu_reps)):  # median of >= 5 repetitions after a warm-up
            t0 = time.perf_counter()
            for i in range(nfr):
                r = orc.detect_multiscale(o, frames_host[i], args.scale_factor, args.min_neighbors, nthreads=cores)
                if rep == 0 and last is not None:
                    ok = ok and r.shape == last[i].shape and bool((r == last[i]).all())
            times.append(time.perf_counter() - t0)
        cdt = float(np.median(times))
        t1 = time.perf_counter()
        orc.detect_multiscale(o, frames_host[0], args.scale_factor, args.min_neighbors, nthreads=1)
        one = time.perf_counter() - t1
        out["cpu_baseline"] = {
            "value": round(windows_per_frame * nfr / cdt / 1e6, 3),
            "unit": "Mwindows/s",
            "cores": cores,
            "kind": "port",
            "sample": f"{nfr} of the same {W}x{H} frames, full detectMultiScale, CPU oracle (restatement of the reference path; "
                      f"OpenCV not installed), {cores} threads over grid rows; median of {len(times)} repetitions after a warm-up call",
            "seconds": round(cdt, 2),
            "seconds_all_repetitions": [round(t, 2) for t in times],
            "value_1_thread": round(windows_per_frame / one / 1e6, 3),
            "value_1_thread_sample": "frame 0 once",
            "rectangles_identical_to_gpu": ok if last is not None else None,
        }
    # The workloads SURVEY.md 8d lists beside the headline, each as a short run (rank 0, N = 1): the stock LBP cascade
    # (BASELINE configs[2]), the headline cascade on i.i.d. uniform noise (distribution (i)), the trainer's bulk feature
    # evaluation (BASELINE configs[4]), and the two "next" rows of SURVEY 8f that have device kernels of their own: the node
    # split search (both feature types) and batched negative mining.
    if extra_legs:
        del clf
        extras = []
        lbp = os.path.join(ROOT, "data", "lbpcascade_frontalface.xml")
        try:
            if os.path.abspath(args.cascade) != lbp and os.path.exists(lbp):
                extras.append(measure_extra_workload(cc, torch, dev, dev_index, lbp, 20, frames_host, args,
                                                     f"{W}x{H} LBP frontalface (stock lbpcascade_frontalface.xml), same frames -- BASELINE configs[2]"))
            if args.content != "uniform":
                extras.append(measure_extra_workload(cc, torch, dev, dev_index, args.cascade, args.specialize,
                                                     make_frames(B, W, H, seed0=0, content="uniform"), args,
                                                     f"{W}x{H} headline cascade on i.i.d. uniform noise (SURVEY 8d distribution (i))"))
            extras.append(measure_training_eval(cc, torch))
            extras.append(measure_split_search(cc, "HAAR"))
            extras.append(measure_split_search(cc, "LBP"))
            extras.append(measure_negative_mining(cc))
        except Exception as e:  # noqa: BLE001 -- the headline line must still be printed
            extras.append({"error": str(e)})
        out["extra_workloads"] = extras
    if rank == 0:
        print(json.dumps(out))
    if comm is not None:
        comm.close()
    if world > 1:
        dist.destroy_process_group()
    if out.get("cpu_baseline", {}).get("rectangles_identical_to_gpu") is False:
        sys.exit("bench.py: GPU rectangles differ from the CPU oracle on the baseline sample: the number above is invalid")
    if any(e.get("rectangles_identical_to_cpu_oracle_frame0") is False or e.get("values_identical_to_cpu_oracle") is False
           for e in out.get("extra_workloads", [])):
        sys.exit("bench.py: an extra workload's GPU results differ from the CPU oracle")


if __name__ == "__main__":
    main()
