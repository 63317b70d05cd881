"""bench.py as the driver calls it: `python bench.py --gpus N` must start its own ranks (no launcher in front of it)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, timeout):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)


def test_multi_rank_launch_without_gpu_fails_loudly_and_does_not_hang():
    """No GPU here: every rank must die with an error and the parent must return non-zero promptly (it terminates the
    surviving ranks instead of waiting for a rendezvous that cannot complete)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a box without a GPU")
    r = _run(["--gpus", "2", "--backend", "gloo", "--frames", "1", "--steps", "1", "--warmup", "0", "--width", "320", "--height", "240",
              "--cpu-frames", "0"], timeout=300)
    assert r.returncode != 0
    assert not r.stdout.strip().startswith("{")  # no headline number without a device


@pytest.mark.parametrize("transport", ["tcp", "torch"])
def test_configs3_gather_leg_rehearsed_with_eight_ranks_on_cpu(transport, monkeypatch):
    """BASELINE configs[3] (512 frames over 8 ranks) has no 8-GPU node to run on here: `bench.py --gpus 8 --backend gloo
    --rehearse-gather` runs what does not need one -- the script's own rank spawning, rendezvous on 127.0.0.1, the C ABI
    communicator (loopback TCP transport) with its trial gather, or the torch collective, the per-step gathers of 8 x 64
    frames (one rank without a rectangle) between the real run's barriers, the max-over-ranks reduction, one line from rank 0
    -- and every child must exit 0."""
    if transport == "tcp":
        monkeypatch.setenv("CCAMD_COMM_TRANSPORT", "tcp")
    else:
        monkeypatch.delenv("CCAMD_COMM_TRANSPORT", raising=False)
    r = _run(["--gpus", "8", "--backend", "gloo", "--rehearse-gather", "--frames", "64", "--steps", "3", "--warmup", "1"], timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    codes = [ln for ln in r.stderr.splitlines() if ln.startswith("[bench] rank exit codes:")]
    assert len(codes) == 1 and json.loads(codes[0].split(":", 1)[1]) == [0] * 8, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert "value" not in out and "metric" not in out  # a rehearsal is not a measurement
    assert out["n_gpus"] == 8 and out["frames_per_step"] == 512 and out["every_rank_holds_all_frames_in_order"] is True
    assert out["gather"].startswith("cc_gather_detections (C ABI, loopback TCP" if transport == "tcp" else "torch.distributed all_gather (gloo)")


def test_eight_ranks_without_gpus_fail_loudly_with_their_exit_codes(monkeypatch):
    """The real `--gpus 8` command on a box without GPUs: no line, a non-zero status, and the children's exit codes on
    stderr (the first rank to die fails the job, the others are terminated: nobody waits in a rendezvous)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a box without a GPU")
    monkeypatch.setenv("CCAMD_COMM_TRANSPORT", "tcp")
    r = _run(["--gpus", "8", "--backend", "gloo", "--frames", "1", "--steps", "1", "--warmup", "0", "--width", "320", "--height", "240",
              "--cpu-frames", "0"], timeout=600)
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    codes = [ln for ln in r.stderr.splitlines() if ln.startswith("[bench] rank exit codes:")]
    assert len(codes) == 1
    rc = json.loads(codes[0].split(":", 1)[1])
    assert len(rc) == 8 and all(c != 0 for c in rc), rc


@pytest.mark.gpu
def test_bench_two_ranks_on_one_gpu_gloo():
    """The N > 1 leg end to end on the one-GPU box: two ranks share the card, detections are gathered over gloo
    (the RCCL path needs one GPU per rank; the driver runs that)."""
    r = _run(["--gpus", "2", "--backend", "gloo", "--frames", "2", "--steps", "1", "--warmup", "1", "--width", "640", "--height", "480",
              "--cpu-frames", "0"], timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout  # exactly one JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak"
    assert out["config"]["frames_per_gpu_per_step"] == 2
    assert out["value"] > 0 and "roofline" in out


@pytest.mark.gpu
def test_bench_single_rank_line_has_roofline_and_cpu_baseline():
    r = _run(["--frames", "2", "--steps", "1", "--warmup", "1", "--width", "640", "--height", "480", "--cpu-frames", "1"], timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 1
    assert out["roofline"]["bound"] in ("hbm", "mfma") and out["roofline"]["achieved"] > 0
    assert out["cpu_baseline"]["rectangles_identical_to_gpu"] is True


@pytest.mark.gpu
def test_bench_two_ranks_gather_through_the_c_abi_communicator(monkeypatch):
    """The same two ranks with the gather of detections on the C ABI's communicator (cc_comm_* / cc_gather_detections) over
    its loopback TCP transport -- the code the multi-GPU run executes with RCCL underneath, including the trial gather that
    decides, on all ranks together, whether to fall back to the torch collective."""
    monkeypatch.setenv("CCAMD_COMM_TRANSPORT", "tcp")
    r = _run(["--gpus", "2", "--backend", "gloo", "--frames", "2", "--steps", "2", "--warmup", "1", "--width", "640", "--height", "480",
              "--cpu-frames", "0"], timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["value"] > 0
    assert out["config"]["gather"].startswith("cc_gather_detections (C ABI, loopback TCP")
