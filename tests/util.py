"""Shared helpers for the tests: fixture readers and seeded synthetic inputs."""
import numpy as np


def read_vec(path, w=75, h=32):
    """.vec reader (format: tools/createsamples/utility.cpp:128-152 in the reference):
    header int32 count, int32 vecSize, int16 0, int16 0; record = 1 zero byte + vecSize int16."""
    raw = open(path, "rb").read()
    count, vec_size = np.frombuffer(raw, "<i4", 2, 0)
    assert vec_size == w * h
    rec = 1 + 2 * vec_size
    out = np.empty((count, h, w), np.uint8)
    for i in range(count):
        o = 12 + i * rec + 1
        out[i] = np.frombuffer(raw, "<i2", vec_size, o).astype(np.uint8).reshape(h, w)
    return out


def frame_uniform(w, h, seed):
    """Distribution (i): i.i.d. uniform [0,255]."""
    return np.random.default_rng(seed).integers(0, 256, (h, w), dtype=np.uint8)


def frame_natural(w, h, seed, sigma=40.0, mean=128.0):
    """Distribution (ii): natural-like 1/f noise, sigma ~40, mean 128."""
    rng = np.random.default_rng(seed)
    spec = np.fft.rfft2(rng.standard_normal((h, w)))
    fy = np.fft.fftfreq(h)[:, None]
    fx = np.fft.rfftfreq(w)[None, :]
    f = np.sqrt(fx * fx + fy * fy)
    f[0, 0] = 1.0
    img = np.fft.irfft2(spec / f, s=(h, w))
    img = (img - img.mean()) / img.std() * sigma + mean
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def paste_patches(img, patches, seed):
    """Paste k (patch, scale-size) blobs at seeded positions (config 1: 'a few pasted face-like patches')."""
    rng = np.random.default_rng(seed)
    out = img.copy()
    h, w = out.shape
    for p in patches:
        ph, pw = p.shape
        y = int(rng.integers(0, h - ph))
        x = int(rng.integers(0, w - pw))
        out[y:y + ph, x:x + pw] = p
    return out


def upscale(patch, size):
    """Plain numpy bilinear up/down-scale of a square patch to size x size (input synthesis only)."""
    n = patch.shape[0]
    c = (np.arange(size) + 0.5) * (n / size) - 0.5
    i0 = np.clip(np.floor(c).astype(int), 0, n - 1)
    i1 = np.clip(i0 + 1, 0, n - 1)
    f = np.clip(c - i0, 0.0, 1.0)
    p = patch.astype(np.float64)
    rows = p[i0] * (1 - f)[:, None] + p[i1] * f[:, None]
    out = rows[:, i0] * (1 - f)[None, :] + rows[:, i1] * f[None, :]
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)
