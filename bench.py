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
    args = ap.parse_args()

    import numpy as np
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
            res["cpu_baseline"] = cpu_baseline(x[:, :n].cpu().numpy(), out[:n].cpu().numpy(), bits[:n].cpu().numpy(), 1)
            res["gpu_over_cpu"] = round(res["value"] / res["cpu_baseline"]["value"], 1)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
