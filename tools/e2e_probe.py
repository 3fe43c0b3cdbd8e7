#!/usr/bin/env python3
"""Host-pointer pipeline probe: times dega_hip_encode_job_host on one shape, pinned and pageable; DEGA_PIPELINE_TRACE=1 and
DEGA_PIPELINE_CHUNKS=n in the environment show / steer the chunking.  tools/e2e_probe.py C T [pinned|pageable] [decode]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from __graft_entry__ import load_package
dca = load_package()
ctx = dca.Context(0)
C_, T = int(sys.argv[1]), int(sys.argv[2])
mode = sys.argv[3] if len(sys.argv) > 3 else "pinned"
x = ctx.synth(C_, T).cpu().numpy()
if mode == "pinned":
    pin = dca.PinnedArray((T, C_), np.int32); pin.array[:] = x; src = pin.array
    dst = dca.PinnedArray((C_ * (2 * T + 64),), np.uint8).array
else:
    src, dst = x, np.zeros(C_ * (2 * T + 64), dtype=np.uint8)
os.environ.pop("DEGA_PIPELINE_TRACE_OFF", None)
ctx.encode_job(src, packed=dst)
for i in range(3):
    time.sleep(0.03)  # (an idle gap between the calls: tools/e2e_trace_summary.py finds the last call by it)
    t0 = time.perf_counter(); r = ctx.encode_job(src, packed=dst); dt = time.perf_counter() - t0
    print("%s C %d T %d: %.1f ms  %.2f Gsamples/s  (%.1f GB/s of samples)" % (mode, C_, T, dt * 1e3, C_ * T / dt / 1e9, 4 * C_ * T / dt / 1e9), flush=True)
if len(sys.argv) > 4 and sys.argv[4] == "decode":
    packed, offsets, bits, err = r
    if mode == "pinned":
        pk = dca.PinnedArray((len(packed),), np.uint8); pk.array[:] = packed; packed = pk.array
        back = dca.PinnedArray((T, C_), np.int32).array
    else:
        back = np.zeros((T, C_), dtype=np.int32)
    ctx.decode_job(packed, offsets, bits, T, out=back)
    for i in range(3):
        time.sleep(0.03)
        t0 = time.perf_counter(); ctx.decode_job(packed, offsets, bits, T, out=back); dt = time.perf_counter() - t0
        print("decode %s C %d T %d: %.1f ms  %.2f Gsamples/s  (%.1f GB/s of samples)  ok %s" % (mode, C_, T, dt * 1e3, C_ * T / dt / 1e9, 4 * C_ * T / dt / 1e9, bool((back == x).all())), flush=True)
