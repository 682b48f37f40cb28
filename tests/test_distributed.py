"""N>1 host logic on CPU (gloo, world size 2): contiguous particle shards and the single
all-reduce of the cropped [volume | weights] buffer."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partitions_exactly():
    from xmipp3_amd.api import shard_range
    for n in (0, 1, 7, 1000, 1_000_003):
        for world in (1, 2, 3, 8):
            r = [shard_range(n, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[k][1] == r[k + 1][0] for k in range(world - 1))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1


WORKER = textwrap.dedent('''
    import os, sys
    sys.path.insert(0, %r)
    import numpy as np, torch, torch.distributed as dist
    from xmipp3_amd.api import allreduce_reconstruction, shard_range
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()

    class FakeRF:   # what RecFourier exposes to the exchange step: the cropped [volume | weights] view
        cropped = True
        def __init__(self, t): self.t = t
        def cropped_view(self): return self.t

    # every rank "grids" its contiguous shard of 101 particles into its own buffer
    n, size = 101, 3 * 9 * 9 * 5
    lo, hi = shard_range(n, rank, world)
    rng = np.random.default_rng(0)
    contrib = rng.standard_normal((n, size)).astype(np.float32)
    mine = torch.from_numpy(contrib[lo:hi].sum(0))
    rf = FakeRF(mine.clone())
    allreduce_reconstruction(rf)
    expect = contrib.sum(0)
    err = float(np.abs(rf.t.numpy() - expect).max())
    assert err < 1e-4, err
    # all ranks hold the same reduced buffer (the finaliser may run anywhere)
    gathered = [torch.zeros_like(rf.t) for _ in range(world)]
    dist.all_gather(gathered, rf.t)
    assert all(torch.equal(g, gathered[0]) for g in gathered)
    dist.destroy_process_group()
    print("rank", rank, "ok", hi - lo)
''') % ROOT


def test_allreduce_of_sharded_reconstruction_gloo_world2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=240)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
    assert "ok 50" in outs[0] + outs[1] and "ok 51" in outs[0] + outs[1]


STUB = textwrap.dedent('''
    import os, sys
    import torch, torch.distributed as dist
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    assert world == int(sys.argv[1]) and rank == int(os.environ["LOCAL_RANK"])
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t)
    if len(sys.argv) > 2 and sys.argv[2] == "fail" and rank == 1:
        sys.exit(3)
    dist.barrier()
    if rank == 0:
        print("sum", t.item())
    dist.destroy_process_group()
''')


def test_bench_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus N` without a launcher: bench.spawn_ranks starts N fresh ranks (here a gloo stub instead of
    the GPU body), relays rank 0's output and fails when any rank fails."""
    sys.path.insert(0, ROOT)
    import bench
    stub = tmp_path / "stub.py"
    stub.write_text(STUB)
    out = subprocess.run([sys.executable, "-c",
                          f"import sys; sys.path.insert(0, {ROOT!r}); import bench; sys.exit(bench.spawn_ranks({str(stub)!r}, ['2'], 2, timeout=120))"],
                         capture_output=True, text=True, timeout=180)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "sum 3.0" in out.stdout
    rc = bench.spawn_ranks(str(stub), ["2", "fail"], 2, timeout=120)
    assert rc == 3


def test_bench_refuses_a_world_size_that_differs_from_gpus():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode != 0 and "WORLD_SIZE=1" in (out.stderr + out.stdout)
