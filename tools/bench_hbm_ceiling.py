#!/usr/bin/env python3
"""Plain streaming rates of the device for context next to the kernels' achieved bandwidths: fill (write only), copy
(read + write) and a reduction (read only) over a 12 GiB float tensor, HIP-event timed. Prints one JSON line."""
import json

import torch

n = 3 * 1024 ** 3  # floats = 12 GiB
x = torch.empty(n, dtype=torch.float32, device="cuda")
y = torch.empty_like(x)


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3


t_fill = timed(lambda: x.fill_(1.0))
t_copy = timed(lambda: y.copy_(x))
t_sum = timed(lambda: x.sum())
gb = n * 4 / 1e9
print(json.dumps({"tensor_GB": round(gb, 2), "fill_write_GBps": round(gb / t_fill, 1), "copy_read_plus_write_GBps": round(2 * gb / t_copy, 1),
                  "sum_read_GBps": round(gb / t_sum, 1)}))
