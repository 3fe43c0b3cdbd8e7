#!/usr/bin/env python3
"""Compile-time knobs of the paired-wave kernels (DG_ENC_* / DG_DEC_* in dega_kernels.hpp), timed side by side.
  tools/tunebench.py build name=-DDG_X=1,-DDG_Y=2 ...   (here: hipcc cross-compiles tools/diag/libdega_hip_tune_<name>.so)
  tools/tunebench.py run C T                             (on the GPU box: every built variant + the shipped library)
Measurement helper; nothing in the product loads these builds."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DIAG = os.path.join(ROOT, "tools", "diag")
CSRC = os.path.join(ROOT, "data-compressor_amd", "csrc")
if sys.argv[1] == "build":
    os.makedirs(DIAG, exist_ok=True)
    procs = []
    for spec in sys.argv[2:]:
        name, _, flags = spec.partition("=")
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-Wno-unused-parameter"]
        cmd += [f for f in flags.split(",") if f] + [os.path.join(CSRC, "dega_hip.hip"), "-o", os.path.join(DIAG, "libdega_hip_tune_%s.so" % name), "-Wl,-rpath,/opt/rocm/lib"]
        procs.append((name, subprocess.Popen(cmd, stderr=subprocess.PIPE, text=True)))
        if len(procs) % 4 == 0:
            for n, p in procs[-4:]:
                e = p.communicate()[1]
                print(n, "rc", p.returncode, e[-300:] if p.returncode else "")
    for n, p in procs[len(procs) // 4 * 4:]:
        e = p.communicate()[1]
        print(n, "rc", p.returncode, e[-300:] if p.returncode else "")
elif sys.argv[1] == "child":
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
    _, ems = ctx.profile_read(0)
    y = torch.zeros((T, C_), dtype=torch.int32, device="cuda"); derr = torch.zeros(C_, dtype=torch.int32, device="cuda")
    ctx.decode(out, bits, T, x_tc=y, err=derr); torch.cuda.synchronize()
    ctx.profile(True)
    for _ in range(3):
        ctx.decode(out, bits, T, x_tc=y, err=derr)
    torch.cuda.synchronize()
    _, dms = ctx.profile_read(1)
    res = {"lib": os.path.basename(dca.LIB_PATH), "C": C_, "T": T, "encode_ms": round(ems, 3), "decode_ms": round(dms, 3),
           "round_trip_ok": bool((y == x).all()), "errors": int((err != 0).sum()) + int((derr != 0).sum())}
    if len(sys.argv) > 4 and sys.argv[4] == "lzmh":
        TL = min(T, 4000)
        stride = 16 * ((TL * 9 + 15) // 16)
        text, lens, rerr = ctx.lzmh_render(x[:TL].contiguous(), stride)
        lcap = 16 * ((stride * 3 // 4 + 63) // 16)
        lout = torch.zeros((C_, lcap), dtype=torch.uint8, device="cuda"); lbits = torch.zeros(C_, dtype=torch.int64, device="cuda"); lerr = torch.zeros(C_, dtype=torch.int32, device="cuda")
        ctx.lzmh_encode(text, lens, cap=lcap, out=lout, bits=lbits, err=lerr); torch.cuda.synchronize()
        ctx.profile(True)
        for _ in range(3):
            ctx.lzmh_encode(text, lens, cap=lcap, out=lout, bits=lbits, err=lerr)
        torch.cuda.synchronize()
        _, lems = ctx.profile_read(2)
        back = torch.zeros((C_, stride), dtype=torch.uint8, device="cuda")
        ctx.lzmh_decode(lout, lbits, stride, out=back); torch.cuda.synchronize()
        ctx.profile(True)
        for _ in range(3):
            ctx.lzmh_decode(lout, lbits, stride, out=back)
        torch.cuda.synchronize()
        _, ldms = ctx.profile_read(3)
        res.update({"lzmh_encode_ms": round(lems, 3), "lzmh_decode_ms": round(ldms, 3)})
    print(json.dumps(res), flush=True)
else:
    C_, T = sys.argv[2], sys.argv[3]
    libs = [os.path.join(ROOT, "data-compressor_amd", "libdega_hip.so")] + sorted(
        os.path.join(DIAG, f) for f in os.listdir(DIAG) if f.startswith("libdega_hip_tune_") and f.endswith(".so"))
    for lib in libs:
        subprocess.run([sys.executable, os.path.abspath(__file__), "child", C_, T] + sys.argv[4:], env=dict(os.environ, DEGA_HIP_LIB=lib))
