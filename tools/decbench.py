import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from __graft_entry__ import load_package
dca = load_package(); ctx = dca.Context(0)
C_, T = 65536, 8640
x = ctx.synth(C_, T); cap = 4 * ((T * 4 + 67) // 4)
out, bits, err = ctx.encode(x, cap=cap)
y, derr = ctx.decode(out, bits, T)
y, derr = ctx.decode(out, bits, T)
torch.cuda.synchronize(); print("ok", bool((y == x).all()))
