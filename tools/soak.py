#!/usr/bin/env python3
"""Randomised differential soak on the GPU: the HIP kernels against the oracle on random shapes, value sizes, models,
workgroup shapes and damaged streams.  tools/soak.py [seconds] [seed]   (not a pytest: a longer hunt for rare cases)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402
from oracle import orc  # noqa: E402
from test_valuesize import pack_be, unpack_be  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
WIDE = len(sys.argv) > 3 and sys.argv[3] == "wide"  # batches of more than 64 Ki channels: the wide workgroup shapes (8 pairs of waves)
dca = load_package()
ctx = dca.Context(0)
t_end = time.time() + budget
rounds = checked = bad = 0


def chain(x_col, vs, ad):
    d, n = pack_be(x_col, vs)
    for name in ("diff", "seg", "bac"):
        r, d, n = orc.stage(name, True, d, n, valuesize=vs, adaptive=ad)
        if r:
            return r, b"", 0
    return 0, d, n


def unchain(d, n, vs, ad):
    for name in ("bac", "seg", "diff"):
        r, d, n = orc.stage(name, False, d, n, valuesize=vs, adaptive=ad)
        if r:
            return r, None
    return 0, unpack_be(d, n, vs)


while time.time() < t_end:
    rounds += 1
    vs = int(rng.choice([32, 32, 32, 16, 8, 24, 13, 31, 5]))
    ad = int(rng.integers(0, 2))
    Cn = int(rng.choice([65537, 70001, 131072 + 77])) if WIDE else int(rng.choice([1, 3, 64, 65, 200, 257, 700]))
    T = int(rng.choice([1, 9, 40, 130])) if WIDE else int(rng.choice([0, 1, 2, 7, 33, 100, 257, 900]))
    top = (1 << vs) - 1
    kind = rng.integers(0, 4, Cn)
    x = np.zeros((T, Cn), dtype=np.int64)
    if WIDE and T > 0:  # vectorised generation for the large batches: walks, with every 16th channel jumping around
        x = np.cumsum(rng.integers(-60, 61, (T, Cn)), axis=0) + top // 4
        x[:, ::16] = rng.integers(0, top + 1, (T, (Cn + 15) // 16))
        x = np.clip(x, 0, top)
    for c in range(0 if not WIDE else Cn, Cn):
        if T == 0:
            break
        if kind[c] == 0:
            col = np.cumsum(rng.integers(-60, 61, T)) + top // 4
        elif kind[c] == 1:
            col = rng.integers(0, top + 1, T)
        elif kind[c] == 2:
            col = np.cumsum(rng.integers(-(top // 6 + 1), top // 6 + 2, T)) + top // 2
        else:
            col = np.where(rng.random(T) < 0.05, rng.integers(0, top + 1, T), top // 3)
        x[:, c] = np.clip(col, 0, top)
    xin = np.ascontiguousarray(x.astype(np.uint32).view(np.int32))
    out, bits, err = ctx.encode_host(xin, adaptive=ad, valuesize=vs)
    sel = rng.choice(Cn, size=min(Cn, 24), replace=False)
    for c in sel:
        r, d, n = chain(x[:, c], vs, ad)
        checked += 1
        if r != err[c] or (r == 0 and (n != int(bits[c]) or out[c, : (n + 7) // 8].tobytes() != d[: (n + 7) // 8])):
            bad += 1
            print("ENCODE MISMATCH", dict(vs=vs, ad=ad, C=Cn, T=T, c=int(c), want=(r, n), got=(int(err[c]), int(bits[c]))), flush=True)
    ok = err == 0
    if ok.any() and T > 0:
        dmg = out.copy()
        dbits = np.where(ok, bits, 0).astype(np.uint64)
        hurt = rng.random(Cn) < 0.3
        for c in np.nonzero(hurt & ok)[0]:
            nb = int(dbits[c])
            if nb > 0:
                k = int(rng.integers(0, nb))
                dmg[c, k // 8] ^= 0x80 >> (k % 8)
        room = 4 * T + 64
        y, counts, derr = ctx.decode_var_host(dmg, dbits, room, adaptive=ad, valuesize=vs)
        for c in sel:
            if not ok[c]:
                continue
            nb = int(dbits[c])
            r, want = unchain(dmg[c, : (nb + 7) // 8].tobytes(), nb, vs, ad)
            checked += 1
            if r != 0 and derr[c] == dca.ERROR_MEMORY:
                continue  # ran out of room before the stage-wise chain's failure point
            if r == 0 and len(want) > room:
                if derr[c] != dca.ERROR_MEMORY:
                    bad += 1
                    print("DECODE ROOM MISMATCH", dict(vs=vs, ad=ad, c=int(c)), flush=True)
                continue
            if derr[c] != r or (r == 0 and (int(counts[c]) != len(want) or (y[: len(want), c].view(np.uint32) != want).any())):
                bad += 1
                print("DECODE MISMATCH", dict(vs=vs, ad=ad, C=Cn, T=T, c=int(c), hurt=bool(hurt[c]), want=r, got=int(derr[c])), flush=True)
    # valuesize 33..64 (int64 containers), one round in six
    if not WIDE and rounds % 6 == 0:
        vs64 = int(rng.choice([33, 40, 48, 63, 64]))
        T6, C6 = int(rng.choice([1, 9, 60, 200])), int(rng.choice([2, 64, 130]))
        top6 = (1 << vs64) - 1
        x6 = np.zeros((T6, C6), dtype=np.uint64)
        for c in range(C6):
            k6 = c % 3
            if k6 == 0:
                col = [int(v) for v in np.clip(np.cumsum(rng.integers(-5000, 5001, T6)) + 10**7, 0, None)]
            elif k6 == 1:
                col = [int(rng.integers(0, 2**62)) * 4 % (top6 + 1) for _ in range(T6)]
            else:
                col = [(top6 // 5) + int(rng.integers(0, 2**20)) for _ in range(T6)]
            x6[:, c] = np.array(col, dtype=np.uint64)
        o6, b6, e6 = ctx.encode64_host(np.ascontiguousarray(x6.view(np.int64)), vs64, adaptive=ad)
        y6, n6, d6 = ctx.decode64_var_host(o6, np.where(e6 == 0, b6, 0).astype(np.uint64), T6 + 2, vs64, adaptive=ad)
        for c in range(min(C6, 12)):
            dd, nn = pack_be(x6[:, c], vs64)
            r = 0
            for name in ("diff", "seg", "bac"):
                r, dd, nn = orc.stage(name, True, dd, nn, valuesize=vs64, adaptive=ad)
                if r:
                    break
            checked += 1
            if r != e6[c] or (r == 0 and (nn != int(b6[c]) or o6[c, : (nn + 7) // 8].tobytes() != dd[: (nn + 7) // 8])):
                bad += 1
                print("ENCODE64 MISMATCH", dict(vs=vs64, ad=ad, c=c, T=T6), flush=True)
            elif r == 0:
                r2, back, bn = 0, dd, nn
                for name in ("bac", "seg", "diff"):
                    r2, back, bn = orc.stage(name, False, back, bn, valuesize=vs64, adaptive=ad)
                    if r2:
                        break
                if r2 != d6[c]:
                    bad += 1
                    print("DECODE64 STATUS MISMATCH", dict(vs=vs64, ad=ad, c=c, want=r2, got=int(d6[c])), flush=True)
                elif r2 == 0:
                    bits_ = np.unpackbits(np.frombuffer(back, dtype=np.uint8))[:bn].reshape(-1, vs64).astype(np.uint64)
                    want = np.zeros(bits_.shape[0], dtype=np.uint64)
                    for k in range(vs64):
                        want |= bits_[:, k] << np.uint64(vs64 - 1 - k)
                    if int(n6[c]) != len(want) or (y6[: len(want), c].view(np.uint64) != want).any():
                        bad += 1
                        print("DECODE64 MISMATCH", dict(vs=vs64, ad=ad, c=c), flush=True)
    # LZMH on random strings
    strings = []
    for i in range(0 if WIDE else int(rng.choice([1, 5, 70]))):
        n = int(rng.choice([0, 1, 3, 402, 403, 404, 700, 2500]))
        k = rng.integers(0, 4)
        if k == 0:
            s = bytes(rng.integers(0, 256, n, dtype=np.uint8))
        elif k == 1:
            s = "".join("%.2f\n" % v for v in 230 + np.cumsum(rng.normal(0, 0.3, n // 6 + 1))).encode()[:n]
        elif k == 2:
            s = bytes(rng.integers(0, 3, n, dtype=np.uint8))
        else:
            s = (bytes(rng.integers(97, 100, 9, dtype=np.uint8)) * (n // 9 + 1))[:n]
        strings.append(s)
    if not strings:
        if rounds % 5 == 0:
            print("rounds", rounds, "checked", checked, "bad", bad, flush=True)
        continue
    lout, lbits, lerr = ctx.lzmh_encode_host(strings)
    for i, s in enumerate(strings):
        r, b, n = orc.stage("lzmh", True, s, 8 * len(s))
        checked += 1
        if lerr[i] != 0 or int(lbits[i]) != n or lout[i, : (n + 7) // 8].tobytes() != b[: (n + 7) // 8]:
            bad += 1
            print("LZMH ENCODE MISMATCH", i, len(s), flush=True)
    ldmg = lout.copy()
    for i in range(len(strings)):
        nb = int(lbits[i])
        if nb > 8 and rng.random() < 0.3:
            k = int(rng.integers(0, nb))
            ldmg[i, k // 8] ^= 0x80 >> (k % 8)
    dec, lens, derr = ctx.lzmh_decode_host(ldmg, lbits, 32768)
    for i in range(len(strings)):
        nb = int(lbits[i])
        r, d, dn = orc.stage("lzmh", False, ldmg[i, : (nb + 7) // 8].tobytes(), nb)
        checked += 1
        if dn // 8 > 32768:
            continue
        if derr[i] != r or int(lens[i]) != dn // 8 or dec[i, : dn // 8].tobytes() != d[: dn // 8]:
            bad += 1
            print("LZMH DECODE MISMATCH", i, len(strings[i]), int(derr[i]), r, int(lens[i]), dn // 8, flush=True)
    if rounds % 20 == 0:
        print("rounds", rounds, "checked", checked, "bad", bad, flush=True)
print("SOAK DONE rounds", rounds, "checked", checked, "bad", bad)
ctx.close()
sys.exit(1 if bad else 0)
