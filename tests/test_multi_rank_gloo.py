"""N > 1 path on CPU: world size 2, gloo.  The channel partition, the rank -> channel-id mapping bench.py uses, and the
host-side gather/concatenate of the ranks' packed streams.  There is no GPU here, so each rank produces its streams
with the oracle standing in for the encode kernel (test code only); rank 0 checks the concatenation against the oracle
run over the whole batch."""
import os
import socket
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, %(root)r)
from __graft_entry__ import load_package
from oracle import orc
dca = load_package()
import importlib.util
spec = importlib.util.spec_from_file_location("dca_shard", os.path.join(%(root)r, "data-compressor_amd", "shard.py"))
shard = importlib.util.module_from_spec(spec); spec.loader.exec_module(shard)

dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
rank, world = dist.get_rank(), dist.get_world_size()
C_total, T = 37, 120
c0, c1 = shard.channel_range(rank, world, C_total)
x = dca.synth_reference(c1 - c0, T, seed=1234, c0=c0, S=50)          # rank r generates its own channel ids
out, bits, err = orc.encode_batch_tc(x, 1)                            # stand-in for ctx.encode on this rank's GPU
assert (err == 0).all()
packed, sizes = shard.pack_streams(out, bits)
pa, ba, off = shard.gather_streams(torch.from_numpy(packed), torch.from_numpy(bits.astype(np.int64)))
t = torch.tensor([0.5 + rank], dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)                              # bench.py's max-over-ranks timing
assert float(t) == 0.5 + (world - 1)
if rank == 0:
    xa = dca.synth_reference(C_total, T, seed=1234, c0=0, S=50)       # the whole batch in one piece
    oa, bita, ea = orc.encode_batch_tc(xa, 1)
    assert ba.numpy().astype(np.uint64).tolist() == bita.tolist()
    for c in range(C_total):
        nb = (int(bita[c]) + 7) // 8
        assert pa[int(off[c]): int(off[c + 1])].numpy().tobytes() == oa[c, :nb].tobytes(), c
    assert int(off[-1]) == pa.numel()
    print("GLOO_OK", C_total, world)
dist.barrier()
dist.destroy_process_group()
'''


def test_channel_range_partition():
    import importlib.util
    spec = importlib.util.spec_from_file_location("dca_shard", os.path.join(ROOT, "data-compressor_amd", "shard.py"))
    shard = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(shard)
    for total in (0, 1, 7, 64, 65536, 8388608 + 3):
        for world in (1, 2, 3, 8):
            ranges = [shard.channel_range(r, world, total) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == total
            assert all(ranges[i][1] == ranges[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in ranges]
            assert max(sizes) - min(sizes) <= 1


def test_two_ranks_gloo_gather_equals_single_batch(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-2000:]
    assert "GLOO_OK 37 2" in outs[0][0]
