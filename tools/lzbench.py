#!/usr/bin/env python3
"""Quick LZMH timing probe on the GPU: tools/lzbench.py [C T]  (synthetic channels rendered as ASCII, cfg 4 in small)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402

C_ = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
T = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
dca = load_package()
ctx = dca.Context(0)
x = ctx.synth(C_, T, seed=1234, c0=0, S=50)
stride = 16 * ((T * 9 + 15) // 16)
text, lens, rerr = ctx.lzmh_render(x, stride)
del x
cap = 16 * ((stride * 3 // 4 + 63) // 16)
out = torch.zeros((C_, cap), dtype=torch.uint8, device="cuda")
bits = torch.zeros(C_, dtype=torch.int64, device="cuda")
err = torch.zeros(C_, dtype=torch.int32, device="cuda")
torch.cuda.synchronize()
ctx.lzmh_encode(text, lens, cap=cap, out=out, bits=bits, err=err)
torch.cuda.synchronize()
ctx.profile(True)
t0 = time.perf_counter()
for _ in range(3):
    ctx.lzmh_encode(text, lens, cap=cap, out=out, bits=bits, err=err)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
ctx.profile(False)
n, ms = ctx.profile_read(2)
nbytes = int(lens.sum().item())
print("encode: %d channels, %.1f MB text, kernel %.3f ms (wall %.3f ms) -> %.2f GB/s, %.3f bits/byte, errors %d" % (
    C_, nbytes / 1e6, ms, dt * 1e3, nbytes / ms / 1e6, float(bits.sum().item()) / nbytes, int((err != 0).sum().item())), flush=True)
if "diag32" in dca.LIB_PATH:
    b = bits.cpu().numpy().reshape(-1, 64)
    names = ["reload_check", "phase1", "phase2", "code_match", "literal", "output", "reload", "loop_top"]
    print({names[k]: [int(b[:, k].mean()), int(b[:, 8 + k].mean())] for k in range(8)}, "total", int(b[:, :8].sum(axis=1).mean()))
    sys.exit(0)
back = torch.zeros((C_, stride), dtype=torch.uint8, device="cuda")
ctx.profile(True)
back, blens, derr = ctx.lzmh_decode(out, bits, stride, out=back)
torch.cuda.synchronize()
ctx.profile(False)
n, dms = ctx.profile_read(3)
if "diag256" in dca.LIB_PATH:
    b = blens.cpu().numpy().reshape(-1, 64)
    names = ["loop_top", "top_up", "token", "publish", "-", "-", "-", "sleep"]
    print("decode: kernel %.3f ms; reading wave" % dms, {names[k]: [int(b[:, k].mean()), int(b[:, 8 + k].mean())] for k in (0, 1, 2, 3, 7)}, "total", int(b[:, :8].sum(axis=1).mean()))
    sys.exit(0)
idx = torch.arange(stride, device="cuda")[None, :] < lens[:, None]
ok = bool((blens == lens).all().item()) and bool(((back == text) | ~idx).all().item()) and int((derr != 0).sum().item()) == 0
print("decode: kernel %.3f ms -> %.2f GB/s, round trip %s" % (dms, nbytes / dms / 1e6, "bit-exact" if ok else "MISMATCH"), flush=True)
