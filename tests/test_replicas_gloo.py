"""N > 1 path of bench.py is 'replicas only': no data-path collective, only a gloo barrier + MAX of one scalar.
Exercise exactly that aggregation with world_size 2 on CPU."""
import json
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


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_world_size_2_gloo_aggregation(tmp_path):
    port = _free_port()
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


def _bench_line(cmd, env=None):
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
    return r, [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]


def test_bench_py_launches_its_own_two_replicas():
    """bench.py's REAL N=2 path (self-spawn, gloo rendezvous on 127.0.0.1, barrier, MAX over ranks, whole-job value) with the
    GPU step replaced by a rank-dependent sleep: rank r sleeps (r+1)*10 ms per step."""
    r, lines = _bench_line([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1", "--stub-step-ms", "10"],
                           env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
    assert r.returncode == 0, r.stderr
    assert len(lines) == 1                      # rank 0 prints ONE line
    line = lines[0]
    assert line["n_gpus"] == 2 and line["steps"] == 5 and line["warmup"] == 1 and line["scaling"] == "weak"
    t = line["ms_per_step"]                     # max over ranks = rank 1's 20 ms per step
    assert 19.0 <= t <= 40.0
    assert abs(line["value"] - 2 * 1e3 / t) < 1e-6 * line["value"]   # value = N * K / max-over-ranks(time of K steps)
    # the timed region is whole blocks of K steps and at least 50 ms long (here one block: 5 steps of 20 ms); every rank's own time is reported
    assert line["timed_steps"] % 5 == 0 and line["timed_steps"] * t >= 50.0
    pr = line["per_rank"]
    assert [x["rank"] for x in pr] == [0, 1] and 9.0 <= pr[0]["ms_per_step"] <= 20.0 and abs(pr[1]["ms_per_step"] - t) < 1e-6


def test_bench_py_counter_pass_mode_runs_exactly_k_steps():
    """--min-timed-ms 0 (the counter passes of tools/prof_round4.sh: hardware counters serialise the launches) switches the 50 ms rule off"""
    r, lines = _bench_line([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--stub-step-ms", "1",
                            "--min-timed-ms", "0"], env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
    assert r.returncode == 0, r.stderr
    assert lines[0]["steps"] == 3 and lines[0]["timed_steps"] == 3


def test_bench_py_times_at_least_50_ms():
    """K steps of 1 ms would be a 3 ms timed region: the region is repeated in whole blocks of K until it lasts >= 50 ms, `steps` stays K"""
    r, lines = _bench_line([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--stub-step-ms", "1"],
                           env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
    assert r.returncode == 0, r.stderr
    line = lines[0]
    assert line["steps"] == 3 and line["timed_steps"] % 3 == 0 and line["timed_steps"] >= 30
    assert line["timed_steps"] * line["ms_per_step"] >= 45.0 and "per_rank" not in line


def test_bench_py_joins_a_launcher_world_and_refuses_a_mismatch():
    """started the way the driver starts N > 1 (torch.distributed.run sets RANK / WORLD_SIZE): joins that world; and
    `--gpus 4` under WORLD_SIZE=2 fails instead of printing a line for fewer GPUs than asked"""
    port = _free_port()
    procs = []
    for rk in range(2):
        env = dict(os.environ, RANK=str(rk), LOCAL_RANK=str(rk), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "0", "--stub-step-ms", "5"],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT))
    outs = [p.communicate(timeout=300) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    lines = [json.loads(l) for l in outs[0][0].splitlines() if l.startswith("{")]
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2 and not [l for l in outs[1][0].splitlines() if l.startswith("{")]
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    r, lines = _bench_line([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--stub-step-ms", "1"], env=env)
    assert r.returncode != 0 and not lines and "refusing" in r.stderr


def test_bench_flop_model():
    sys.path.insert(0, ROOT)
    import importlib
    bench = importlib.import_module("bench")
    assert bench.fwd_flops(8, 16, 4096, 64, False) == 4.0 * 8 * 16 * 64 * 4096 * 4096  # 549.8 GFLOP
    assert abs(bench.fwd_flops(8, 16, 4096, 64, True) / 274.9e9 - 1) < 1e-3
    alg, exe = bench.pass_flops("bwd", 8, 16, 4096, 64, False)   # 5 GEMM-equivalents in the textbook backward, 7 executed
    assert alg == 2.5 * 549755813888.0 and exe == 3.5 * 549755813888.0


def test_gpu_sensors_without_a_gpu_returns_none():
    """bench.py's clock / power probe (roofline.under_load) must never raise: on a box without amdgpu hwmon files (this container)
    it reports (None, None) and the bench line simply omits the object."""
    sys.path.insert(0, ROOT)
    import bench
    mhz, watts = bench.gpu_sensors(0)
    assert mhz is None or mhz > 0
    assert watts is None or watts >= 0
