#!/usr/bin/env python3
"""One process, a few launches of the DEGA encode (and decode) kernel on synthetic data: the target of rocprofv3 --pmc passes.
tools/encprof.py C T [decode]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from __graft_entry__ import load_package
dca = load_package()
ctx = dca.Context(0)
C_, T = int(sys.argv[1]), int(sys.argv[2])
x = ctx.synth(C_, T)
cap = 4 * ((T * 4 + 67) // 4)
out = torch.zeros((C_, cap), dtype=torch.uint8, device="cuda"); bits = torch.zeros(C_, dtype=torch.int64, device="cuda"); err = torch.zeros(C_, dtype=torch.int32, device="cuda")
for _ in range(3):
    ctx.encode(x, cap=cap, out=out, bits=bits, err=err)
torch.cuda.synchronize()
if len(sys.argv) > 3:
    y = torch.zeros((T, C_), dtype=torch.int32, device="cuda"); derr = torch.zeros(C_, dtype=torch.int32, device="cuda")
    for _ in range(3):
        ctx.decode(out, bits, T, x_tc=y, err=derr)
    torch.cuda.synchronize()
print("ok", float(bits.sum()) / (C_ * T))
