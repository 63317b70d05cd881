#!/usr/bin/env python3
"""Streams a known number of bytes with the cascade kernel's load shape; run under `rocprofv3 --pmc FETCH_SIZE` and
compare FETCH_SIZE * 1024 of k_stream_dwords with the printed byte count (MI355X_MICROARCH.md, HBM section)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cascadeclassifier_amd import _lib as L  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 30
c = C.c_uint32(0)
L.check(L.lib().cc_debug_stream_dwords(0, n, 3, C.byref(c)))
print("streamed_bytes_per_launch", n, "checksum", c.value)
