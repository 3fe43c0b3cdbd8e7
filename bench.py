#!/usr/bin/env python3
"""bench.py -- DEGA encode throughput on MI355X (BASELINE.json metric: Msamples/s DEGA encode (int32)).

    python bench.py --gpus N --steps K --warmup W

N > 1: either launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` (one rank per
GPU; RANK / LOCAL_RANK / WORLD_SIZE come from the environment and WORLD_SIZE must equal N), or started plainly, in
which case this process spawns the N ranks itself BEFORE anything touches a GPU and exits with their status.  Fewer than
N visible GPUs is an error, never a silent 1-GPU measurement.

A "step" is one pass of the hot path (diff -> seg -> bac adaptive, fused in one HIP kernel) over one batch of
synthetic meter channels that is already resident in HBM: BASELINE.json configs[1], 65 536 channels x 86 400 int32
samples per GPU, [T][C] layout (22.6 GB), generated on the device.  Channels are independent, so N GPUs
= N disjoint channel ranges, no data-path collective ("scaling": "weak": per-GPU work is fixed).

One JSON line on rank 0: throughput, the roofline of the encode kernel (HBM; algorithmic bytes = 4 B read per sample +
the stream bytes written, over the kernel's hipEvent time on its own stream; beside it the instruction-issue bound the
kernel actually runs against), the CPU baseline (the reference itself, oracle/_ref, when it was built -- else our C
port of it -- timed single threaded on a bounded sample of the same channels, whose GPU streams are also compared bit
for bit), the PCIe-inclusive host-pointer path (`end_to_end`), and -- after the timed region, N = 1 only -- `extra`:
the other BASELINE configs at full size (cfg3 short series, cfg4 LZMH, one GPU's share of cfg5)."""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md (6.29 TB/s measured copy)
# The model behind roofline.issue_bound_ms: what bounds a channel is the instruction stream of ITS coding wave --
# one wave issues an instruction every 4.09 (4-byte encodings) to 4.56 (8-byte) cycles whatever shares the SIMD with it
# (profiles/r03_ubench2_issue_cost.txt).  Instructions of a steady 32-symbol step of that wave from the ISA ledger
# (profiles/r03_isa_ledger.md): 698 VALU + 19 LDS + 26 SALU.
ISSUE_INSTR_PER_SYMBOL = 743.0 / 32.0
ISSUE_CYCLES_PER_INSTR = 4.3   # the mix of 4- and 8-byte encodings of the word path
SHADER_CLOCK_HZ = 2.25e9


def _cpu_encode_columns(job):
    """Pool worker (forked before the GPU is touched, never touches it): encodes its channels `repeat` times with the CPU chain."""
    from oracle import orc
    cols, repeat = job
    use_ref = orc.have_ref()
    n = 0
    for _ in range(repeat):
        for col in cols:
            if use_ref:
                orc.ref_encode_i32(col, 1)
            else:
                orc.encode_i32(col, 1)
            n += col.size
    return n


def cpu_all_cores(pool, ncores, x_sample):
    """The same CPU chain, process-parallel over all host cores (CPU baseline (ii)); channels strided."""
    import numpy as np
    from oracle import orc
    T, n = x_sample.shape
    per_worker = 4  # channels a worker holds; it codes them `repeat` times so that it works for about two seconds
    cols = [np.ascontiguousarray(x_sample[:, c % n]) for c in range(ncores * per_worker)]
    repeat = max(1, int(2.0 * 1.9e6 / max(1, T * per_worker)))
    jobs = [(cols[k * per_worker:(k + 1) * per_worker], repeat) for k in range(ncores)]
    pool.map(_cpu_encode_columns, [(j[0][:1], 1) for j in jobs])  # warm the workers up (library load) outside the timing
    t0 = time.perf_counter()
    done = sum(pool.map(_cpu_encode_columns, jobs))
    dt = time.perf_counter() - t0
    return {"value": round(done / dt / 1e6, 3), "unit": "Msamples/s", "cores": ncores, "kind": "reference" if orc.have_ref() else "port",
            "sample": "%d processes x %d channels x %d samples x %d repeats of the same workload, %.1f s" % (ncores, per_worker, T, repeat, dt)}


def cpu_baseline(x_sample, gpu_out, gpu_bits, adaptive=1):
    """Time the CPU chain on the sample channels (single thread) and check the GPU streams against it."""
    import numpy as np
    from oracle import orc  # the checker / baseline, never the measured product path

    T, n = x_sample.shape
    use_ref = orc.have_ref()
    t0 = time.perf_counter()
    mismatches = 0
    for c in range(n):
        col = np.ascontiguousarray(x_sample[:, c])
        if use_ref:
            ret, stream, nbits, _ = orc.ref_encode_i32(col, adaptive)
        else:
            ret, stream, nbits = orc.encode_i32(col, adaptive)
        if ret != 0 or nbits != int(gpu_bits[c]) or gpu_out[c, : (nbits + 7) // 8].tobytes() != stream:
            mismatches += 1
    dt = time.perf_counter() - t0
    return {
        "value": round(T * n / dt / 1e6, 4), "unit": "Msamples/s", "cores": 1, "kind": "reference" if use_ref else "port",
        "sample": "%d channels x %d samples of the same workload, diff+seg+bac adaptive per channel, %.1f s" % (n, T, dt),
        "gpu_streams_bit_exact": mismatches == 0,
    }


def seg_symbols_per_sample(x_sample):
    """Coded symbols (= seg bits) per sample of a few channels: what the coder's serial chain is made of."""
    import numpy as np
    from oracle import orc
    tot = 0
    n = min(8, x_sample.shape[1])
    for c in range(n):
        col = np.ascontiguousarray(x_sample[:, c]).astype(">i4").tobytes()
        r, d, nb = orc.stage("diff", True, col, 32 * x_sample.shape[0])
        r, d, nb = orc.stage("seg", True, d, nb)
        tot += nb
    return tot / float(n * x_sample.shape[0])


class Env:
    """Rank, device and library handles of this process."""

    def __init__(self, args):
        import torch
        import torch.distributed as dist
        from __graft_entry__ import load_package
        self.torch, self.dist = torch, dist
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != args.gpus:
            sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d: refusing to report a number for the wrong GPU count" % (args.gpus, self.world))
        if torch.cuda.device_count() <= self.local_rank:
            sys.exit("bench.py: rank %d wants GPU %d but only %d are visible" % (self.rank, self.local_rank, torch.cuda.device_count()))
        if self.world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("nccl", rank=self.rank, world_size=self.world, device_id=torch.device("cuda", self.local_rank))
        torch.cuda.set_device(self.local_rank)
        self.dev = torch.device("cuda", self.local_rank)
        self.dca = load_package()
        self.ctx = self.dca.Context(self.local_rank)  # raises (ERROR_LIBRARY_INIT) without a GPU: no fallback

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def reduce_max(self, values):
        t = self.torch.tensor(values, dtype=self.torch.float64, device=self.dev)
        if self.world > 1:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return [float(v) for v in t.tolist()]

    def close(self):
        if self.world > 1:
            self.dist.barrier()
            self.dist.destroy_process_group()
        self.ctx.close()


def spawn_ranks_if_needed(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks here, before this process touches a GPU."""
    if "WORLD_SIZE" in os.environ or args.gpus == 1:
        return
    import socket
    import torch
    have = torch.cuda.device_count()  # does not initialise the GPU
    if have < args.gpus:
        sys.exit("bench.py: --gpus %d but only %d GPU(s) are visible" % (args.gpus, have))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    codes = [p.wait() for p in procs]
    sys.exit(max(abs(c) for c in codes))


def end_to_end(env, args, x, bits_dev):
    """The host-pointer entry points of the C ABI (what a reference caller gets through the plugin): upload, kernels and
    download pipelined over chunks of channels.  Never the headline value.  Two shapes of the same 2.8 GB of samples: the
    headline's channel length (8 192 channels x 86 400: no chunk can finish before a whole channel's serial chain has
    run, ~80 ms, so this shape is bounded by the kernel) and the same bytes as more, shorter channels (65 536 x 10 800:
    bounded by the PCIe link)."""
    import numpy as np
    torch, ctx, dca = env.torch, env.ctx, env.dca
    T = x.shape[0]
    n = min(args.end_to_end_channels, x.shape[1])
    res = {"unit": "Msamples/s", "what": "dega_hip_encode_job_host: chunks of channels on their own streams, H2D + encode + pack + D2H of the stream bytes overlapped; "
                                         "buffers owned by the context (the first call grows them and is not timed)"}
    # the link itself, for scale: one pinned 1 GiB copy each way
    pin = torch.empty(1 << 30, dtype=torch.uint8).pin_memory()
    dbuf = torch.empty(1 << 30, dtype=torch.uint8, device=env.dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    dbuf.copy_(pin, non_blocking=True)
    torch.cuda.synchronize()
    h2d = (1 << 30) / (time.perf_counter() - t0) / 1e9
    t0 = time.perf_counter()
    pin.copy_(dbuf, non_blocking=True)
    torch.cuda.synchronize()
    d2h = (1 << 30) / (time.perf_counter() - t0) / 1e9
    res["pcie_pinned_h2d_gbs"], res["pcie_pinned_d2h_gbs"] = round(h2d, 1), round(d2h, 1)
    del pin, dbuf

    def measure(xh_pageable, tag, check_bits):
        Tn, Cn = xh_pageable.shape
        out = {}
        pinned = dca.PinnedArray((Tn, Cn), np.int32)
        pinned.array[:] = xh_pageable
        packed_buf = dca.PinnedArray((Cn * (Tn * 2 + 64),), np.uint8)
        page_buf = np.zeros(Cn * (Tn * 2 + 64), dtype=np.uint8)  # touched: no first-use page faults inside the timing
        for name, src, dst in (("pageable", xh_pageable, page_buf), ("pinned", pinned.array, packed_buf.array)):
            ctx.encode_job(src, adaptive=1, packed=dst)  # grows the context's buffers
            best = None
            for _ in range(2):
                t0 = time.perf_counter()
                pk, off, b, e = ctx.encode_job(src, adaptive=1, packed=dst)
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            out[name + "_value"] = round(Cn * Tn / best / 1e6, 2)
            out[name + "_seconds"] = round(best, 4)
            out[name + "_h2d_gbs_equivalent"] = round(4.0 * Cn * Tn / best / 1e9, 1)
            if check_bits is not None:
                out["streams_equal_device_resident"] = bool(out.get("streams_equal_device_resident", True) and (b.astype(np.int64) == check_bits).all() and (e == 0).all())
        # and back: packed streams in, samples out
        t0 = time.perf_counter()
        back, derr = ctx.decode_job(pk, off, b, Tn, adaptive=1, out=pinned.array)
        t0 = time.perf_counter()
        back, derr = ctx.decode_job(pk, off, b, Tn, adaptive=1, out=pinned.array)
        dt = time.perf_counter() - t0
        out["decode_pinned_value"] = round(Cn * Tn / dt / 1e6, 2)
        out["decode_round_trip_ok"] = bool((derr == 0).all() and (back == xh_pageable).all())
        pinned.free()
        packed_buf.free()
        res[tag] = out

    xh = np.ascontiguousarray(x[:, :n].cpu().numpy())
    measure(xh, "channels_%d_x_%d" % (n, T), bits_dev[:n].cpu().numpy())
    # the same bytes as 8x the channels of 1/8 the length (a different, equally valid batch of the workload)
    if T % 8 == 0 and n * 8 <= 65536:
        xs = np.ascontiguousarray(xh.reshape(8, T // 8, n).transpose(1, 0, 2).reshape(T // 8, 8 * n))
        measure(xs, "channels_%d_x_%d" % (8 * n, T // 8), None)
    first = res["channels_%d_x_%d" % (n, T)]
    res["value"], res["packed_value"], res["channels"] = first["pageable_value"], first["pinned_value"], n
    return res


def group_end_to_end(env, args, xh):
    """The product's multi-GPU path: dega_hip_group_encode / _decode called from ONE process over all
    --gpus N devices -- contiguous channel ranges per device, a host thread and a pipeline each, host-side concatenate, no
    collective -- on N x (one device's share) host samples, pinned and pageable, with the single-device figure beside it.
    At N = 1 also the group [0, 0] against [0]: what the threads and the concatenate cost.  Never the headline value."""
    import numpy as np
    dca = env.dca
    T, n = xh.shape
    world = env.world
    res = {"unit": "Msamples/s", "what": "dega_hip_group_encode/_decode from one process: channel ranges per device, host-side concatenate, no collective",
           "channels_per_device": n, "samples_per_channel": T}

    def measure(devices, xs):
        out = {}
        Tn, Cn = xs.shape
        g = dca.Group(devices)
        try:
            pinned = dca.PinnedArray((Tn, Cn), np.int32)
            pinned.array[:] = xs
            packed_pin = dca.PinnedArray((Cn * (Tn * 2 + 64),), np.uint8)
            page_buf = np.zeros(Cn * (Tn * 2 + 64), dtype=np.uint8)
            for name, src, dst in (("pageable", xs, page_buf), ("pinned", pinned.array, packed_pin.array)):
                g.encode_job(src, adaptive=1, packed=dst)  # grows the contexts' buffers
                best = None
                for _ in range(2):
                    t0 = time.perf_counter()
                    pk, off, b, e = g.encode_job(src, adaptive=1, packed=dst)
                    dt = time.perf_counter() - t0
                    best = dt if best is None else min(best, dt)
                out["encode_%s_value" % name] = round(Cn * Tn / best / 1e6, 2)
                out["encode_%s_seconds" % name] = round(best, 4)
            back, derr = g.decode_job(pk, off, b, Tn, adaptive=1, out=pinned.array)
            t0 = time.perf_counter()
            back, derr = g.decode_job(pk, off, b, Tn, adaptive=1, out=pinned.array)
            dt = time.perf_counter() - t0
            out["decode_pinned_value"] = round(Cn * Tn / dt / 1e6, 2)
            out["round_trip_ok"] = bool((derr == 0).all() and (e == 0).all() and (back == xs).all())
            out["bits"] = b.astype(np.int64)
            pinned.free()
            packed_pin.free()
        finally:
            g.close()
        return out

    one = measure([0], xh)
    bits_one = one.pop("bits")
    res["devices_1"] = one
    if world == 1:
        two = measure([0, 0], xh)  # two members on the one device: the partition, the threads and the concatenate are the same code
        res["members_2_on_one_device"] = dict(two, streams_equal_single=bool((two.pop("bits") == bits_one).all()))
    else:
        xs = np.ascontiguousarray(np.tile(xh, (1, world)))
        many = measure(list(range(world)), xs)
        res["devices_%d" % world] = dict(many, streams_equal_single=bool((many.pop("bits").reshape(world, -1) == bits_one[None, :]).all()))
    return res


def run_dega(env, args):
    import numpy as np
    torch, ctx = env.torch, env.ctx
    world, rank, dev = env.world, env.rank, env.dev
    C_, T = args.channels, args.samples
    cap = 4 * int((T * args.cap_bytes_per_sample + 67) // 4)

    x = torch.empty((T, C_), dtype=torch.int32, device=dev)
    ctx.synth(C_, T, seed=1234, c0=rank * C_, S=args.step_size, out=x)  # rank r owns channels [r*C, (r+1)*C)
    out = torch.zeros((C_, cap), dtype=torch.uint8, device=dev)
    bits = torch.zeros(C_, dtype=torch.int64, device=dev)
    err = torch.zeros(C_, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()

    for _ in range(args.warmup):
        ctx.encode(x, adaptive=1, cap=cap, out=out, bits=bits, err=err)
    env.barrier()
    ctx.profile(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ctx.encode(x, adaptive=1, cap=cap, out=out, bits=bits, err=err)
    env.barrier()
    elapsed = time.perf_counter() - t0
    ctx.profile(False)
    n_launch, kernel_ms = ctx.profile_read(0)
    elapsed = env.reduce_max([elapsed])[0]

    # bit-exact round trip at full size (the size-independent property of the metric): decode every stream on the
    # device and compare with the input on the device
    round_trip = None
    if not args.no_round_trip:
        y = torch.zeros((T, C_), dtype=torch.int32, device=dev)
        derr = torch.zeros(C_, dtype=torch.int32, device=dev)
        ctx.profile(True)
        ctx.decode(out, bits, T, adaptive=1, x_tc=y, err=derr)
        torch.cuda.synchronize()
        ctx.profile(False)
        _, dec_ms = ctx.profile_read(1)
        ok = bool((y == x).all().item()) and int((derr != 0).sum().item()) == 0
        bad = env.reduce_max([0.0 if ok else 1.0])[0]
        round_trip = {"bit_exact": bad == 0.0, "decode_kernel_ms": round(dec_ms, 3),
                      "decode_msamples_per_s_per_gpu": round(C_ * T / (dec_ms * 1e-3) / 1e6, 1) if dec_ms > 0 else None}
        del y

    n_err = int((err != 0).sum().item())
    out_bytes = int(((bits + 7) // 8).sum().item())
    algo_bytes = 4.0 * C_ * T + out_bytes  # 4 B read per sample + compressed bytes written, per launch
    achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
    res = None
    if rank == 0:
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")  # HBM bytes per launch from rocprofv3 --pmc runs, if recorded
        if os.path.exists(tpath):
            with open(tpath) as f:
                tj = json.load(f)
            if tj.get("channels") == C_ and tj.get("samples") == T:
                traffic = tj.get("hbm_bytes_per_launch")
        res = {
            "metric": "Msamples/s DEGA encode (int32)",
            "value": round(C_ * T * args.steps * world / elapsed / 1e6, 2),
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int32",
            "data": "synthetic",
            "config": {
                "workload": "DEGA encode (diff+seg+bac adaptive), %d channels x %d int32 samples per GPU, [T][C] resident in HBM" % (C_, T),
                "channels_per_gpu": C_, "samples_per_channel": T, "random_walk_step": args.step_size, "seed": 1234,
                "slab_bytes_per_channel": cap, "bits_per_sample_out": round(out_bytes * 8.0 / (C_ * T), 4),
                "channels_in_error": n_err, "partitioning": "channel ranges per GPU, no collective",
            },
            "roofline": {
                "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                "kernel": "dega_encode_kernel<true>", "kernel_ms": round(kernel_ms, 4), "launches_timed": n_launch,
                "algorithmic_bytes_per_launch": int(algo_bytes),
            },
        }
        if round_trip is not None:
            res["round_trip"] = round_trip
        if world == 1 and args.cpu_channels > 0:
            n = min(args.cpu_channels, C_)
            xs = x[:, :n].cpu().numpy()
            res["cpu_baseline"] = cpu_baseline(xs, out[:n].cpu().numpy(), bits[:n].cpu().numpy(), 1)
            res["gpu_over_cpu"] = round(res["value"] / res["cpu_baseline"]["value"], 1)
            # what the kernel really runs against: every channel is a serial chain of coded symbols, 64 Ki channels are one
            # coding wave per SIMD, and a wave issues one instruction per ~4.3 cycles: a launch cannot finish before the
            # longest chain has been issued
            sym = seg_symbols_per_sample(xs)
            issue_ms = T * sym * ISSUE_INSTR_PER_SYMBOL * ISSUE_CYCLES_PER_INSTR / SHADER_CLOCK_HZ * 1e3
            res["roofline"].update({"issue_bound_ms": round(issue_ms, 2), "issue_bound_frac": round(issue_ms / kernel_ms, 4) if kernel_ms > 0 else None,
                                    "issue_bound_model": "%d samples x %.2f coded symbols x %.1f instructions of the coding wave x %.1f cycles / %.2f GHz, one coding wave per SIMD"
                                                         % (T, sym, ISSUE_INSTR_PER_SYMBOL, ISSUE_CYCLES_PER_INSTR, SHADER_CLOCK_HZ / 1e9)})
            if env.pool is not None:
                res["cpu_all_cores"] = cpu_all_cores(env.pool, env.ncores, xs)
        if world == 1 and args.end_to_end_channels > 0:
            res["end_to_end"] = end_to_end(env, args, x, bits)
        if args.end_to_end_channels > 0:
            import numpy as np
            n = min(args.end_to_end_channels, C_)
            res.setdefault("end_to_end", {})["group"] = group_end_to_end(env, args, np.ascontiguousarray(x[:, :n].cpu().numpy()))
    del x, out, bits, err
    torch.cuda.empty_cache()
    return res


def run_roundtrip(env, args):
    """One GPU's share of BASELINE configs[4] (8 Mi channels x 86 400 samples over 8 GPUs = 1 Mi channels per GPU, 362 GB
    of samples: more than HBM holds): the rank's channel range streamed in batches -- generate on the device, encode,
    decode, compare with the input -- one step = the whole range.  No data leaves the GPU; no collective."""
    torch, ctx = env.torch, env.ctx
    world, rank, dev = env.world, env.rank, env.dev
    C_, T, B = args.channels, args.samples, min(args.batch_channels, args.channels)
    cap = 4 * int((T * args.cap_bytes_per_sample + 67) // 4)
    x = torch.empty((T, B), dtype=torch.int32, device=dev)
    y = torch.empty((T, B), dtype=torch.int32, device=dev)
    out = torch.zeros((B, cap), dtype=torch.uint8, device=dev)
    bits = torch.zeros(B, dtype=torch.int64, device=dev)
    err = torch.zeros(B, dtype=torch.int32, device=dev)
    derr = torch.zeros(B, dtype=torch.int32, device=dev)
    nbatch = (C_ + B - 1) // B

    def one_pass(check):
        ok, enc_ms, dec_ms, out_bytes = True, 0.0, 0.0, 0
        for b in range(nbatch):
            c0, n = b * B, min(B, C_ - b * B)
            xs, ys = x[:, :n], y[:, :n]
            if n != B:  # a ragged last batch: contiguous views of its own
                xs, ys = torch.empty((T, n), dtype=torch.int32, device=dev), torch.empty((T, n), dtype=torch.int32, device=dev)
            ctx.synth(n, T, seed=1234, c0=rank * C_ + c0, S=args.step_size, out=xs)
            torch.cuda.synchronize()
            ctx.profile(True)
            ctx.encode(xs, adaptive=1, cap=cap, out=out[:n], bits=bits[:n], err=err[:n])
            ctx.decode(out[:n], bits[:n], T, adaptive=1, x_tc=ys, err=derr[:n])
            torch.cuda.synchronize()
            ctx.profile(False)
            enc_ms += ctx.profile_read(0)[1]
            dec_ms += ctx.profile_read(1)[1]
            if check:
                ok = ok and bool((ys == xs).all().item()) and int((err[:n] != 0).sum().item()) == 0 and int((derr[:n] != 0).sum().item()) == 0
                out_bytes += int(((bits[:n] + 7) // 8).sum().item())
        return ok, enc_ms, dec_ms, out_bytes

    for _ in range(args.warmup):
        one_pass(False)
    env.barrier()
    t0 = time.perf_counter()
    ok, enc_ms, dec_ms, out_bytes = True, 0.0, 0.0, 0
    for _ in range(args.steps):
        o, e, d, ob = one_pass(True)
        ok, enc_ms, dec_ms, out_bytes = ok and o, enc_ms + e, dec_ms + d, ob
    env.barrier()
    elapsed = time.perf_counter() - t0
    elapsed, enc_ms, dec_ms, bad = env.reduce_max([elapsed, enc_ms, dec_ms, 0.0 if ok else 1.0])
    res = None
    if rank == 0:
        kernel_s = (enc_ms + dec_ms) * 1e-3
        algo = 2.0 * (4.0 * C_ * T + out_bytes)  # encode: samples in + stream out; decode: the reverse
        res = {
            "metric": "Msamples/s DEGA encode+decode round trip (int32)", "value": round(C_ * T * args.steps * world / kernel_s / 1e6, 2), "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(kernel_s / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": "DEGA encode + decode + compare, %d channels x %d int32 samples per GPU streamed in %d batches of %d channels (generated on the device)" % (C_, T, nbatch, B),
                       "channels_per_gpu": C_, "samples_per_channel": T, "batch_channels": B, "random_walk_step": args.step_size, "seed": 1234,
                       "bits_per_sample_out": round(out_bytes * 8.0 / (C_ * T), 4), "partitioning": "channel ranges per GPU, batches per range, no collective",
                       "value_counts": "the encode and decode kernels' time (hipEvents, summed over the batches); generation and comparison are outside it",
                       "wall_ms_per_step_with_generation_and_compare": round(elapsed / args.steps * 1e3, 3)},
            "roofline": {"bound": "hbm", "achieved": round(algo * args.steps / kernel_s / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(algo * args.steps / kernel_s / 1e9 / HBM_PEAK_GBS, 5), "traffic": None, "kernel": "dega_encode_kernel + dega_decode_kernel",
                         "encode_ms_per_step": round(enc_ms / args.steps, 3), "decode_ms_per_step": round(dec_ms / args.steps, 3)},
            "round_trip": {"bit_exact": bad == 0.0},
        }
    del x, y, out, bits, err, derr
    torch.cuda.empty_cache()
    return res


def lzmh_cpu_baseline(texts, gpu_out, gpu_bits):
    """The reference's `encode lzmh` (oracle/_ref; else our C port) on the sample channels, single thread."""
    from oracle import orc  # the checker / baseline, never the measured product path

    use_ref = orc.have_ref()
    t0 = time.perf_counter()
    mismatches = 0
    nbytes = 0
    for c, s in enumerate(texts):
        if use_ref:
            ret, b, nb, _ = orc.ref_run_chain(s, 8 * len(s), ["encode lzmh"])
        else:
            ret, b, nb = orc.stage("lzmh", True, s, 8 * len(s))
        nbytes += len(s)
        if ret != 0 or nb != int(gpu_bits[c]) or gpu_out[c, : (nb + 7) // 8].tobytes() != b[: (nb + 7) // 8]:
            mismatches += 1
    dt = time.perf_counter() - t0
    return {"value": round(nbytes / dt / 1e6, 3), "unit": "MB/s", "cores": 1, "kind": "reference" if use_ref else "port",
            "sample": "%d channels (%d bytes) of the same workload, encode lzmh per channel, %.1f s" % (len(texts), nbytes, dt),
            "gpu_streams_bit_exact": mismatches == 0}


def lzmh_end_to_end(env, text, lens, bits_dev, n):
    """The host-pointer path of the second codec (dega_hip_group_lzmh_encode / _decode: chunks of channels on their own
    streams, upload + kernel + pack + download overlapped; packed streams out): host text in, packed streams out, and back.
    A channel's serial chain takes the same kernel time however few channels there are, so this is the kernel's time plus
    what the copies of the first and last chunk add.  Never the headline value."""
    import numpy as np
    dca = env.dca
    th = np.ascontiguousarray(text[:n].cpu().numpy())
    lh = lens[:n].cpu().numpy().astype(np.uint64)
    nbytes = int(lh.sum())
    res = {"unit": "MB/s", "channels": n, "text_bytes": nbytes,
           "what": "dega_hip_group_lzmh_encode/_decode on one device: host text in, packed streams out (and back), pipeline over chunks of channels"}
    g = dca.Group([0])
    try:
        pinned = dca.PinnedArray(th.shape, np.uint8)
        pinned.array[:] = th
        packed_pin = dca.PinnedArray((nbytes + nbytes // 4 + 64 * n + 64,), np.uint8)
        page_buf = np.zeros(nbytes + nbytes // 4 + 64 * n + 64, dtype=np.uint8)
        for name, src, dst in (("pageable", th, page_buf), ("pinned", pinned.array, packed_pin.array)):
            g.lzmh_encode_job(src, lh, packed=dst)
            t0 = time.perf_counter()
            pk, off, b, e = g.lzmh_encode_job(src, lh, packed=dst)
            dt = time.perf_counter() - t0
            res["encode_%s_value" % name] = round(nbytes / dt / 1e6, 1)
            res["encode_%s_seconds" % name] = round(dt, 4)
        res["streams_equal_device_resident"] = bool((b.astype(np.int64) == bits_dev[:n].cpu().numpy()).all() and (e == 0).all())
        back, blens, berr = g.lzmh_decode_job(pk, off, b, th.shape[1], out=pinned.array)
        t0 = time.perf_counter()
        back, blens, berr = g.lzmh_decode_job(pk, off, b, th.shape[1], out=pinned.array)
        dt = time.perf_counter() - t0
        res["decode_pinned_value"] = round(nbytes / dt / 1e6, 1)
        idx = np.arange(th.shape[1])[None, :] < lh[:, None].astype(np.int64)
        res["decode_round_trip_ok"] = bool((berr == 0).all() and (blens == lh).all() and ((back == th) | ~idx).all())
        pinned.free()
        packed_pin.free()
    finally:
        g.close()
    return res


def run_lzmh(env, args):
    """BASELINE configs[3]: LZMH encode of the cfg2 channels rendered as ASCII "%d.%02d\\n" lines, one GPU
    lane per channel.  Same contract as the DEGA line; the unit is bytes of text."""
    torch, ctx = env.torch, env.ctx
    world, rank, dev = env.world, env.rank, env.dev
    C_, T = args.channels, args.samples
    x = torch.empty((T, C_), dtype=torch.int32, device=dev)
    ctx.synth(C_, T, seed=1234, c0=rank * C_, S=args.step_size, out=x)
    if args.lzmh_input == "ascii":
        stride = 16 * ((T * 8 + 64 + 15) // 16)
        text, lens, rerr = ctx.lzmh_render(x, stride)
        del x
        assert int((rerr != 0).sum().item()) == 0, "a channel's text does not fit its row"
        cap = 16 * ((stride * 3 // 4 + 63) // 16)
        what = "rendered as ASCII '%d.%02d\\n' lines"
    else:
        # the channel's samples as the bytes `encode normalize` would hand on: big-endian int32 (cfg 4)
        stride = 16 * ((T * 4 + 15) // 16)
        text = torch.zeros((C_, stride), dtype=torch.uint8, device=dev)
        for c0 in range(0, C_, 4096):  # transpose [T][C] -> [C][T] in slices, then swap to big-endian
            blk = x[:, c0:c0 + 4096].t().contiguous().view(torch.uint8).view(-1, T, 4).flip(2).reshape(-1, 4 * T)
            text[c0:c0 + blk.shape[0], : 4 * T] = blk
        del x
        lens = torch.full((C_,), 4 * T, dtype=torch.int64, device=dev)
        cap = 16 * ((stride * 5 // 4 + 63 + 15) // 16)
        what = "as raw big-endian int32 bytes"
    out = torch.zeros((C_, cap), dtype=torch.uint8, device=dev)
    bits = torch.zeros(C_, dtype=torch.int64, device=dev)
    err = torch.zeros(C_, dtype=torch.int32, device=dev)

    for _ in range(args.warmup):
        ctx.lzmh_encode(text, lens, cap=cap, out=out, bits=bits, err=err)
    env.barrier()
    ctx.profile(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ctx.lzmh_encode(text, lens, cap=cap, out=out, bits=bits, err=err)
    env.barrier()
    elapsed = time.perf_counter() - t0
    ctx.profile(False)
    n_launch, kernel_ms = ctx.profile_read(2)
    elapsed = env.reduce_max([elapsed])[0]
    in_bytes = int(lens.sum().item())
    tot = torch.tensor([in_bytes], dtype=torch.int64, device=dev)
    if world > 1:
        env.dist.all_reduce(tot, op=env.dist.ReduceOp.SUM)
    total_bytes = int(tot.item())

    round_trip = None
    if not args.no_round_trip:
        back = torch.zeros((C_, stride), dtype=torch.uint8, device=dev)
        ctx.profile(True)
        back, blens, derr = ctx.lzmh_decode(out, bits, stride, out=back)
        torch.cuda.synchronize()
        ctx.profile(False)
        _, dec_ms = ctx.profile_read(3)
        ok = bool((blens == lens).all().item()) and int((derr != 0).sum().item()) == 0
        for c0 in range(0, C_, 8192):  # compare in slices: the mask of a whole 45 GB batch would not fit beside it
            sl = slice(c0, min(C_, c0 + 8192))
            idx = torch.arange(stride, device=dev)[None, :] < lens[sl, None]
            ok = ok and bool(((back[sl] == text[sl]) | ~idx).all().item())
        bad = env.reduce_max([0.0 if ok else 1.0])[0]
        round_trip = {"bit_exact": bad == 0.0, "decode_kernel_ms": round(dec_ms, 3),
                      "decode_mb_per_s_per_gpu": round(in_bytes / (dec_ms * 1e-3) / 1e6, 1) if dec_ms > 0 else None}
        del back

    out_bytes = int(((bits + 7) // 8).sum().item())
    algo_bytes = float(in_bytes + out_bytes)  # every text byte read once + the stream bytes written
    achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
    res = None
    if rank == 0:
        res = {
            "metric": "MB/s LZMH encode (%s)" % ("ASCII lines" if args.lzmh_input == "ascii" else "raw int32 bytes"), "value": round(total_bytes * args.steps / elapsed / 1e6, 2), "unit": "MB/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {
                "workload": "LZMH encode, %d channels x %d samples per GPU %s (%.2f GB), resident in HBM" % (C_, T, what, in_bytes / 1e9),
                "channels_per_gpu": C_, "samples_per_channel": T, "random_walk_step": args.step_size, "seed": 1234,
                "text_bytes_per_gpu": in_bytes, "row_bytes_per_channel": stride, "slab_bytes_per_channel": cap,
                "bits_per_byte_out": round(out_bytes * 8.0 / max(1, in_bytes), 4), "channels_in_error": int((err != 0).sum().item()),
                "partitioning": "channel ranges per GPU, no collective",
            },
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                         "traffic": None, "kernel": "lzmh_encode_kernel", "kernel_ms": round(kernel_ms, 4), "launches_timed": n_launch,
                         "algorithmic_bytes_per_launch": int(algo_bytes)},
        }
        if round_trip is not None:
            res["round_trip"] = round_trip
        if world == 1 and getattr(args, "lzmh_end_to_end_channels", 0) > 0:
            res["end_to_end"] = lzmh_end_to_end(env, text, lens, bits, min(args.lzmh_end_to_end_channels, C_))
        if world == 1 and args.cpu_channels > 0:
            # a bounded sample: the reference codes ~7 MB/s, so 128 channels of 600 kB are ~10 s of CPU work
            n = min(max(1, args.cpu_channels // 2), C_)
            lc = lens[:n].cpu().numpy()
            tc = text[:n].cpu().numpy()
            texts = [tc[c, : int(lc[c])].tobytes() for c in range(n)]
            res["cpu_baseline"] = lzmh_cpu_baseline(texts, out[:n].cpu().numpy(), bits[:n].cpu().numpy())
            res["gpu_over_cpu"] = round(res["value"] / res["cpu_baseline"]["value"], 1)
    del text, lens, out, bits, err
    torch.cuda.empty_cache()
    return res


def extras(env, args):
    """The other BASELINE configs at full size, after the timed region (N = 1): sub-records with their own roofline."""
    out = {}
    a = argparse.Namespace(**vars(args))
    # configs[2]: 1 Mi channels x 96 samples (15-min data, S = 300): encode with its round trip
    a.channels, a.samples, a.step_size, a.steps, a.warmup, a.cpu_channels, a.end_to_end_channels, a.no_round_trip = 1048576, 96, 300, 5, 1, 0, 0, False
    r = run_dega(env, a)
    out["cfg3"] = {k: r[k] for k in ("metric", "value", "unit", "ms_per_step", "config", "roofline", "round_trip")}
    # configs[3]: LZMH of the cfg2 channels as ASCII lines, 39.4 GB of text, with the reference's `encode lzmh` beside it
    a = argparse.Namespace(**vars(args))
    a.channels, a.samples, a.step_size, a.steps, a.warmup, a.cpu_channels, a.lzmh_input, a.no_round_trip = 65536, 86400, 50, 2, 1, 128, "ascii", False
    a.lzmh_end_to_end_channels = 4096
    r = run_lzmh(env, a)
    out["cfg4_lzmh"] = {k: r[k] for k in ("metric", "value", "unit", "ms_per_step", "config", "roofline", "round_trip", "cpu_baseline", "gpu_over_cpu", "end_to_end") if k in r}
    # the same channels as the raw big-endian int32 bytes `encode normalize` hands on ("also report")
    a = argparse.Namespace(**vars(args))
    a.channels, a.samples, a.step_size, a.steps, a.warmup, a.cpu_channels, a.lzmh_input, a.no_round_trip = 65536, 86400, 50, 1, 1, 0, "raw", False
    a.lzmh_end_to_end_channels = 0
    r = run_lzmh(env, a)
    out["cfg4_lzmh_raw_int32"] = {k: r[k] for k in ("metric", "value", "unit", "ms_per_step", "config", "roofline", "round_trip") if k in r}
    # configs[4]: one GPU's share (1 Mi channels x 86 400), streamed in batches, encode + decode + compare on the device
    a = argparse.Namespace(**vars(args))
    a.channels, a.samples, a.step_size, a.steps, a.warmup, a.batch_channels = 1048576, 86400, 50, 1, 0, 131072
    r = run_roundtrip(env, a)
    out["cfg5_share"] = {k: r[k] for k in ("metric", "value", "unit", "ms_per_step", "config", "roofline", "round_trip")}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--channels", type=int, default=65536, help="channels per GPU")
    ap.add_argument("--samples", type=int, default=86400, help="samples per channel")
    ap.add_argument("--step-size", type=int, default=50, help="S of the synthetic random walk")
    ap.add_argument("--cap-bytes-per-sample", type=float, default=4.0, help="slab bytes per sample per channel")
    ap.add_argument("--cpu-channels", type=int, default=256, help="channels of the CPU baseline sample (0 = skip)")
    ap.add_argument("--no-round-trip", action="store_true", help="skip the decode + compare after the timed region")
    ap.add_argument("--no-all-cores", action="store_true", help="skip the process-parallel CPU baseline")
    ap.add_argument("--no-extras", action="store_true", help="skip the cfg3 / cfg4 / cfg5-share sub-records after the timed region")
    ap.add_argument("--end-to-end-channels", type=int, default=8192,
                    help="channels of the host-pointer (PCIe-inclusive) measurement after the timed region (0 = skip)")
    ap.add_argument("--lzmh-input", choices=("ascii", "raw"), default="ascii",
                    help="lzmh workload: the channels as ASCII '%%d.%%02d\\n' lines (the codec's domain) or as raw big-endian int32 bytes")
    ap.add_argument("--workload", choices=("dega", "lzmh", "roundtrip"), default="dega",
                    help="dega = BASELINE configs[1] (the headline metric); lzmh = configs[3], the same channels as ASCII lines through LZMH; "
                         "roundtrip = one GPU's share of configs[4]: --channels streamed in batches of --batch-channels, encode + decode + compare")
    ap.add_argument("--lzmh-end-to-end-channels", type=int, default=0, help="lzmh workload: channels of the host-pointer (PCIe-inclusive) measurement (0 = skip)")
    ap.add_argument("--batch-channels", type=int, default=131072, help="roundtrip workload: channels per batch (x + slabs + decoded samples must fit HBM)")
    args = ap.parse_args()
    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be at least 1")
    spawn_ranks_if_needed(args)  # does not return in the launching process

    pool, ncores = None, 0
    if int(os.environ.get("WORLD_SIZE", "1")) == 1 and args.workload == "dega" and args.cpu_channels > 0 and not args.no_all_cores:
        # worker processes for the all-core CPU baseline are forked here, before anything touches the GPU
        import multiprocessing as mp
        ncores = min(len(os.sched_getaffinity(0)), 16)  # this GPU's share of the host
        pool = mp.get_context("fork").Pool(ncores)
    env = Env(args)
    env.pool, env.ncores = pool, ncores
    if args.workload == "lzmh":
        res = run_lzmh(env, args)
    elif args.workload == "roundtrip":
        res = run_roundtrip(env, args)
    else:
        res = run_dega(env, args)
        default_shape = (args.channels, args.samples, args.step_size) == (65536, 86400, 50)
        if res is not None and env.world == 1 and default_shape and not args.no_extras:
            res["extra"] = extras(env, args)
    if env.rank == 0:
        assert res["n_gpus"] == args.gpus
        print(json.dumps(res), flush=True)
    if pool is not None:
        pool.close()
        pool.join()
    env.close()


if __name__ == "__main__":
    main()
