#!/usr/bin/env python3
"""bench.py -- DEGA encode throughput on MI355X (BASELINE.json metric: Msamples/s DEGA encode (int32)).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`)

A "step" is one pass of the hot path (diff -> seg -> bac adaptive, fused in one HIP kernel) over one batch of
synthetic meter channels that is already resident in HBM: BASELINE.json configs[1], 65 536 channels x 86 400 int32
samples per GPU, [T][C] layout (22.6 GB), generated on the device.  Channels are independent, so N GPUs
= N disjoint channel ranges, no data-path collective ("scaling": "weak": per-GPU work is fixed).

One JSON line on rank 0: throughput, the roofline of the encode kernel (HBM; algorithmic bytes = 4 B read per sample +
the stream bytes written, over the kernel's hipEvent time on its own stream) and the CPU baseline (the reference
itself, oracle/_ref, when it was built -- else our C port of it -- timed single threaded on a bounded sample of the
same channels, whose GPU streams are also compared bit for bit).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md (6.29 TB/s measured copy)


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
    ap.add_argument("--end-to-end-channels", type=int, default=8192,
                    help="channels of the host-pointer (PCIe-inclusive) measurement after the timed region (0 = skip)")
    ap.add_argument("--lzmh-input", choices=("ascii", "raw"), default="ascii",
                    help="lzmh workload: the channels as ASCII '%%d.%%02d\\n' lines (the codec's domain) or as raw big-endian int32 bytes")
    ap.add_argument("--workload", choices=("dega", "lzmh", "roundtrip"), default="dega",
                    help="dega = BASELINE configs[1] (the headline metric); lzmh = configs[3], the same channels as ASCII lines through LZMH; "
                         "roundtrip = one GPU's share of configs[4]: --channels streamed in batches of --batch-channels, encode + decode + compare")
    ap.add_argument("--batch-channels", type=int, default=131072, help="roundtrip workload: channels per batch (x + slabs + decoded samples must fit HBM)")
    args = ap.parse_args()
    if args.workload == "lzmh":
        return main_lzmh(args)
    if args.workload == "roundtrip":
        return main_roundtrip(args)

    import numpy as np
    import torch
    import torch.distributed as dist
    from __graft_entry__ import load_package

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    pool, ncores = None, 0
    if world == 1 and args.cpu_channels > 0 and not args.no_all_cores:
        # worker processes for the all-core CPU baseline are forked here, before anything touches the GPU
        import multiprocessing as mp
        ncores = min(len(os.sched_getaffinity(0)), 16)  # this GPU's share of the host
        pool = mp.get_context("fork").Pool(ncores)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    dca = load_package()
    ctx = dca.Context(local_rank)  # raises (ERROR_LIBRARY_INIT) without a GPU: no fallback
    C_, T = args.channels, args.samples
    cap = 4 * int((T * args.cap_bytes_per_sample + 67) // 4)

    x = torch.empty((T, C_), dtype=torch.int32, device=dev)
    ctx.synth(C_, T, seed=1234, c0=rank * C_, S=args.step_size, out=x)  # rank r owns channels [r*C, (r+1)*C)
    out = torch.zeros((C_, cap), dtype=torch.uint8, device=dev)
    bits = torch.zeros(C_, dtype=torch.int64, device=dev)
    err = torch.zeros(C_, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        ctx.encode(x, adaptive=1, cap=cap, out=out, bits=bits, err=err)
    barrier()
    ctx.profile(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ctx.encode(x, adaptive=1, cap=cap, out=out, bits=bits, err=err)
    barrier()
    elapsed = time.perf_counter() - t0
    ctx.profile(False)
    n_launch, kernel_ms = ctx.profile_read(0)

    t_all = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t_all, op=dist.ReduceOp.MAX)
    elapsed = float(t_all.item())

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
        flags = torch.tensor([1 if ok else 0], dtype=torch.int64, device=dev)
        if world > 1:
            dist.all_reduce(flags, op=dist.ReduceOp.MIN)
        round_trip = {"bit_exact": bool(flags.item()), "decode_kernel_ms": round(dec_ms, 3),
                      "decode_msamples_per_s_per_gpu": round(C_ * T / (dec_ms * 1e-3) / 1e6, 1) if dec_ms > 0 else None}
        del y

    n_err = int((err != 0).sum().item())
    out_bytes = int(((bits + 7) // 8).sum().item())
    algo_bytes = 4.0 * C_ * T + out_bytes  # 4 B read per sample + compressed bytes written, per launch
    achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0

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
            if pool is not None:
                res["cpu_all_cores"] = cpu_all_cores(pool, ncores, xs)
        if world == 1 and args.end_to_end_channels > 0:
            # the host-pointer entry point of the C ABI: allocation + H2D + kernel + D2H, never the headline value
            n = min(args.end_to_end_channels, C_)
            xh = np.ascontiguousarray(x[:, :n].cpu().numpy())
            t0 = time.perf_counter()
            ho, hb, he = ctx.encode_host(xh, adaptive=1, cap=cap)
            dt = time.perf_counter() - t0
            t0 = time.perf_counter()
            pk, poff, pb, pe = ctx.encode_packed_host(xh, adaptive=1)
            dtp = time.perf_counter() - t0
            res["end_to_end"] = {"value": round(n * T / dt / 1e6, 2), "unit": "Msamples/s", "channels": n, "seconds": round(dt, 3),
                                 "what": "dega_hip_encode_host: device alloc + H2D of pageable memory + kernel + D2H of the slabs",
                                 "packed_value": round(n * T / dtp / 1e6, 2), "packed_seconds": round(dtp, 3),
                                 "packed_what": "dega_hip_encode_packed_host: the same with the streams compacted on the device before D2H",
                                 "streams_equal_device_resident": bool((hb.astype(np.int64) == bits[:n].cpu().numpy()).all()
                                                                       and (pb.astype(np.int64) == bits[:n].cpu().numpy()).all())}
        print(json.dumps(res), flush=True)
    if pool is not None:
        pool.close()
        pool.join()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


def main_roundtrip(args):
    """One GPU's share of BASELINE configs[4] (8 Mi channels x 86 400 samples over 8 GPUs = 1 Mi channels per GPU, 362 GB
    of samples: more than HBM holds): the rank's channel range streamed in batches -- generate on the device, encode,
    decode, compare with the input -- one step = the whole range.  No data leaves the GPU; no collective."""
    import torch
    import torch.distributed as dist
    from __graft_entry__ import load_package

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dca = load_package()
    ctx = dca.Context(local_rank)
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

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_pass(False)
    barrier()
    t0 = time.perf_counter()
    ok, enc_ms, dec_ms, out_bytes = True, 0.0, 0.0, 0
    for _ in range(args.steps):
        o, e, d, ob = one_pass(True)
        ok, enc_ms, dec_ms, out_bytes = ok and o, enc_ms + e, dec_ms + d, ob
    barrier()
    elapsed = time.perf_counter() - t0
    vals = torch.tensor([elapsed, enc_ms, dec_ms, 0.0 if ok else 1.0], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(vals, op=dist.ReduceOp.MAX)
    elapsed, enc_ms, dec_ms, bad = [float(v) for v in vals.tolist()]
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
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


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


def main_lzmh(args):
    """BASELINE configs[3]: LZMH encode of the cfg2 channels rendered as ASCII "%d.%02d\\n" lines, one GPU
    lane per channel.  Same contract as the DEGA line; the unit is bytes of text."""
    import torch
    import torch.distributed as dist
    from __graft_entry__ import load_package

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dca = load_package()
    ctx = dca.Context(local_rank)
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

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        ctx.lzmh_encode(text, lens, cap=cap, out=out, bits=bits, err=err)
    barrier()
    ctx.profile(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ctx.lzmh_encode(text, lens, cap=cap, out=out, bits=bits, err=err)
    barrier()
    elapsed = time.perf_counter() - t0
    ctx.profile(False)
    n_launch, kernel_ms = ctx.profile_read(2)
    t_all = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t_all, op=dist.ReduceOp.MAX)
    elapsed = float(t_all.item())
    in_bytes = int(lens.sum().item())
    tot = torch.tensor([in_bytes], dtype=torch.int64, device=dev)
    if world > 1:
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
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
        flags = torch.tensor([1 if ok else 0], dtype=torch.int64, device=dev)
        if world > 1:
            dist.all_reduce(flags, op=dist.ReduceOp.MIN)
        round_trip = {"bit_exact": bool(flags.item()), "decode_kernel_ms": round(dec_ms, 3),
                      "decode_mb_per_s_per_gpu": round(in_bytes / (dec_ms * 1e-3) / 1e6, 1) if dec_ms > 0 else None}
        del back

    out_bytes = int(((bits + 7) // 8).sum().item())
    algo_bytes = float(in_bytes + out_bytes)  # every text byte read once + the stream bytes written
    achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
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
        if world == 1 and args.cpu_channels > 0:
            # a bounded sample: the reference codes ~7 MB/s, so 128 channels of 600 kB are ~10 s of CPU work
            n = min(max(1, args.cpu_channels // 2), C_)
            lc = lens[:n].cpu().numpy()
            tc = text[:n].cpu().numpy()
            texts = [tc[c, : int(lc[c])].tobytes() for c in range(n)]
            res["cpu_baseline"] = lzmh_cpu_baseline(texts, out[:n].cpu().numpy(), bits[:n].cpu().numpy())
            res["gpu_over_cpu"] = round(res["value"] / res["cpu_baseline"]["value"], 1)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
