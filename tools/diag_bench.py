#!/usr/bin/env python3
"""Ablation timing of the encode kernel: times libdega_hip.so and the DEGA_DIAG builds (csrc/Makefile `diag`) on the
same synthetic batch in one process each; the DEGA_DIAG=32 build reports cycles per section of the CODING wave.
Diagnostic builds produce wrong streams by construction."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import torch
    from __graft_entry__ import load_package
    dca = load_package()
    ctx = dca.Context(0)
    C_, T = int(sys.argv[2]), int(sys.argv[3])
    x = ctx.synth(C_, T)
    cap = 4 * ((T * 4 + 67) // 4)
    out = torch.zeros((C_, cap), dtype=torch.uint8, device="cuda"); bits = torch.zeros(C_, dtype=torch.int64, device="cuda"); err = torch.zeros(C_, dtype=torch.int32, device="cuda")
    ctx.encode(x, cap=cap, out=out, bits=bits, err=err); torch.cuda.synchronize()
    ctx.profile(True)
    for _ in range(3):
        ctx.encode(x, cap=cap, out=out, bits=bits, err=err)
    torch.cuda.synchronize()
    n, ms = ctx.profile_read(0)
    res = {"lib": os.path.basename(dca.LIB_PATH), "kernel_ms": round(ms, 4), "bits_per_sample": round(float(bits.sum()) / (C_ * T), 3)}
    if "diag32" in dca.LIB_PATH:
        b = bits.cpu().numpy().reshape(-1, 64)
        names = ["steady_steps", "steady_masked_steps", "other_word_steps", "waits"]
        res["coder_passes_per_wave"] = {names[k]: {"count": int(b[:, k].mean()), "cycles": int(b[:, 4 + k].mean())} for k in range(4)}
        res["total_cycles"] = int(b[:, 4:8].sum(axis=1).mean())
    print(json.dumps(res))
    if len(sys.argv) > 4 and sys.argv[4] == "decode":
        y = torch.zeros((T, C_), dtype=torch.int32, device="cuda"); derr = torch.zeros(C_, dtype=torch.int32, device="cuda")
        b2 = bits.clone()
        ctx.profile(True)
        ctx.decode(out, b2, T, x_tc=y, err=derr)
        torch.cuda.synchronize()
        n, ms = ctx.profile_read(1)
        r2 = {"lib": os.path.basename(dca.LIB_PATH), "decode_kernel_ms": round(ms, 4)}
        if "diag128" in dca.LIB_PATH:  # the decode kernel's pass counters: the coder's over in_bits, the parser's over err
            bb = b2.cpu().numpy().reshape(-1, 64)
            ee = derr.cpu().numpy().reshape(-1, 64)
            cn = ["steady", "steady_masked", "general", "idle"]
            pn = ["steady", "general", "cheap_polls", "sleeps_in_general"]
            r2["coder_passes_per_wave"] = {cn[k]: {"count": int(bb[:, k].mean()), "cycles": int(bb[:, 4 + k].mean())} for k in range(4)}
            r2["parser_passes_per_wave"] = {pn[k]: {"count": int(ee[:, k].mean()), "cycles": int(ee[:, 4 + k].mean()) * 1024} for k in range(4)}
        if "diag" not in dca.LIB_PATH:
            r2["round_trip_ok"] = bool((y == x).all())
        print(json.dumps(r2))
else:
    C_, T = (sys.argv[1], sys.argv[2]) if len(sys.argv) > 2 else ("65536", "8640")
    libs = [os.path.join(ROOT, "data-compressor_amd", "libdega_hip.so")] + sorted(
        os.path.join(ROOT, "tools", "diag", f) for f in os.listdir(os.path.join(ROOT, "tools", "diag")) if f.endswith(".so"))
    for lib in libs:
        env = dict(os.environ, DEGA_HIP_LIB=lib)
        subprocess.run([sys.executable, os.path.abspath(__file__), "child", C_, T] + sys.argv[3:], env=env)
