#!/usr/bin/env python3
"""4 vs 8 pairs of waves per workgroup (DEGA_WAVES_PER_WORKGROUP=4|8) of the DEGA kernels on one batch: tools/wgbench.py C T  (runs itself once per shape)."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import torch
    from __graft_entry__ import load_package
    dca = load_package()
    ctx = dca.Context(0)
    C_, T, S = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    x = ctx.synth(C_, T, S=S)
    cap = 4 * ((T * 4 + 67) // 4)
    out = torch.zeros((C_, cap), dtype=torch.uint8, device="cuda"); bits = torch.zeros(C_, dtype=torch.int64, device="cuda"); err = torch.zeros(C_, dtype=torch.int32, device="cuda")
    ctx.encode(x, cap=cap, out=out, bits=bits, err=err); torch.cuda.synchronize()
    ctx.profile(True)
    for _ in range(3):
        ctx.encode(x, cap=cap, out=out, bits=bits, err=err)
    torch.cuda.synchronize()
    _, ems = ctx.profile_read(0)
    y = torch.zeros((T, C_), dtype=torch.int32, device="cuda"); derr = torch.zeros(C_, dtype=torch.int32, device="cuda")
    ctx.decode(out, bits, T, x_tc=y, err=derr); torch.cuda.synchronize()
    ctx.profile(True)
    ctx.decode(out, bits, T, x_tc=y, err=derr); torch.cuda.synchronize()
    _, dms = ctx.profile_read(1)
    print(json.dumps({"waves_per_workgroup": os.environ.get("DEGA_WAVES_PER_WORKGROUP", "auto"), "C": C_, "T": T, "encode_ms": round(ems, 3),
                      "encode_gsamples_s": round(C_ * T / ems / 1e6, 2), "decode_ms": round(dms, 3), "decode_gsamples_s": round(C_ * T / dms / 1e6, 2),
                      "round_trip_ok": bool((y == x).all()), "errors": int((err != 0).sum()) + int((derr != 0).sum())}), flush=True)
else:
    C_, T = sys.argv[1], sys.argv[2]
    S = sys.argv[3] if len(sys.argv) > 3 else "50"
    for w in ("4", "8"):
        subprocess.run([sys.executable, os.path.abspath(__file__), "child", C_, T, S], env=dict(os.environ, DEGA_WAVES_PER_WORKGROUP=w))
