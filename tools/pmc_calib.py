"""Calibration of the PMC byte model on gfx950: ONE known stream per process for `rocprofv3 --pmc FETCH_SIZE` /
`--pmc WRITE_SIZE` passes (MI355X_MICROARCH.md, HBM: FETCH_SIZE reads half the bytes of a 16 B/lane streaming read;
other widths are uncalibrated - the products of this library load 8 B per lane).

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/calib_r8 -- python3 tools/pmc_calib.py 8 load

Streams a 1 GiB buffer (2^27 doubles; larger than the 256 MiB Infinity Cache) 4 times with k_calib_stream<W, STORE>.
The counter value per dispatch / 2^30 is the factor to apply to that counter for that access width.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pgdrome_amd import _lib

width = int(sys.argv[1]) if len(sys.argv) > 1 else 8
store = (sys.argv[2] if len(sys.argv) > 2 else "load") == "store"
ctx = _lib.Context(0)
n = 1 << 27
v = ctx.vec_alloc(n)
ctx.vec_fill(v, 0.5)
ctx.sync()
for _ in range(4):
    ctx.calib_stream(v, width, store)
ctx.sync()
ctx.timer_start()
for _ in range(10):
    ctx.calib_stream(v, width, store)
t = ctx.timer_stop() / 10
print("calib width %d %s: %.1f us per GiB pass = %.0f GB/s" % (width, "store" if store else "load", t * 1e6, 8.0 * n / t / 1e9))
