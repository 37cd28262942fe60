"""N > 1 path of bench.py is 'replicas only': no data-path collective, only a gloo barrier + MAX of one scalar.
Exercise exactly that aggregation with world_size 2 on CPU."""
import os
import socket
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, time, torch, torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dist.barrier()
    dt = 0.1 * (rank + 1)          # rank-dependent "time of K steps"
    t = torch.tensor([dt], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    steps = 10
    value = world * steps / float(t.item())   # replicas: units of all ranks / max time
    if rank == 0:
        print("VALUE", value, float(t.item()))
    dist.barrier()
    dist.destroy_process_group()
""")


def test_world_size_2_gloo_aggregation(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / "w.py"
    script.write_text(WORKER)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=120) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    line = [l for l in outs[0][0].splitlines() if l.startswith("VALUE")][0].split()
    assert abs(float(line[2]) - 0.2) < 1e-12          # max over ranks
    assert abs(float(line[1]) - 2 * 10 / 0.2) < 1e-9  # whole-job aggregate


def test_bench_flop_model():
    sys.path.insert(0, ROOT)
    import importlib
    bench = importlib.import_module("bench")
    assert bench.fwd_flops(8, 16, 4096, 64, False) == 4.0 * 8 * 16 * 64 * 4096 * 4096  # 549.8 GFLOP
    assert abs(bench.fwd_flops(8, 16, 4096, 64, True) / 274.9e9 - 1) < 1e-3
