#!/usr/bin/env python3
"""What does the 4-byte alignment of a pixel's disparities (D odd: 257 floats = 1028 bytes per pixel) cost the marching ZNCC kernel?
The same 4096^2 pair, ZNCC 11x11, at D = 255, 256, 257, 264: time per cell.  (D = 256 and 264: every 128-byte run of a workgroup
is made of whole 32-byte sectors; 255 and 257: every run starts and ends inside a sector it shares with the neighbouring block.)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from pandora_amd.engine import Engine  # noqa: E402

eng = Engine(0)
L, R = bench.synthetic_pair(4096, 4096, 0, 264)
eng.set_images(L, R, 1)
for D in [int(a) for a in sys.argv[1:]] or (255, 256, 257, 264, 257, 256):
    cv = eng.alloc_cv(D, 0)
    eng.zncc(cv, 11)
    eng.sync()
    t0 = time.perf_counter()
    for _ in range(5):
        eng.zncc(cv, 11)
    eng.sync()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    print(f"D {D}: {ms:.3f} ms, {ms / D * 257:.3f} ms per 257 disparities", flush=True)
    cv.free()
eng.close()
