#!/usr/bin/env python
"""bench.py — headline benchmark of the hot path (BASELINE.json): attn-ops/s of flash_attention_n at
(B=8, H=16, S=4096, D=64) bf16, n=1, non-causal, on N replicated GPUs (no sharding, no RCCL on the data path).

  python bench.py --gpus N --steps K --warmup W [--workload m0|c1|c2|c3|c4|c5|d256|t32] [--pass fwd|bwd|fwdbwd]

N > 1: bench.py launches its own N replica processes (one per GPU, gloo control plane over 127.0.0.1) when it is started
without WORLD_SIZE; started under `python -m torch.distributed.run --nproc-per-node N ...` it joins that world instead.
Either way WORLD_SIZE must equal --gpus and N GPUs must be visible, or the run FAILS (it never reports fewer GPUs than asked).

A step = one pass of the hot path over the whole batch, inputs resident in HBM:
  --pass fwd     one flash_attention_n forward (Python front end -> ctypes -> fasn_fwd)                     [default, the metric]
  --pass fwdbwd  forward + backward through autograd (fasn_fwd, then fasn_bwd = [delta +] dQ + dK/dV kernels)
  --pass bwd     the backward alone: one fasn_bwd call on saved (o, lse) with preallocated gradients
Rank 0 prints ONE JSON line; `value` = whole-job steps/s = N * K / max-over-ranks(wall time of K steps).
In the line: `roofline` (dominant kernel(s) vs the dense MFMA peak, duration from events on the launch stream, HBM traffic
from the committed PMC pass of THIS libfasn.so build or null), at N=1 `cpu_baseline` (the oracle's restatement of
slow_attention_n on the host cores, bounded sample) and `max_abs_err` against that oracle; the default forward run also
carries `backward` / `fwdbwd` objects (same K and W, measured after the headline's timed region).
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (B, H, S, D, dtype, n, causal)
    "c1": (2, 2, 128, 32, "f32", 1.0, False),      # BASELINE config 1: the reference's CPU-runnable case (slow_attention_n fp32), timed IN FULL on the host
    "m0": (8, 16, 4096, 64, "bf16", 1.0, False),   # the shape BASELINE.json's metric is quoted on
    "c2": (8, 16, 1024, 64, "bf16", 1.0, False),
    "c3": (8, 16, 4096, 64, "f16", 1.0, True),
    "c5": (64, 16, 4096, 64, "bf16", 1.0, True),
    "t32": (32, 16, 1024, 32, "f16", 1.0, False),    # the shape of the reference's Triton test grid (tests/gpu/core/test_flash_attn_triton.py:21-23): head dim 32, fp16
    "d256": (4, 16, 4096, 256, "bf16", 1.0, False),   # head dim 256 (not a BASELINE config: the reference API's "any E" served natively)
    "c4": (4, 32, 8192, 128, "bf16", 0.5, False),   # + dense ALiBi bias [H,L,S] and key-padding mask [B,1,1,S]
}
PEAK_TFLOPS = 2500.0  # dense bf16/fp16 MFMA peak, MI355X (MI355X_MICROARCH.md)
PEAK_TFLOPS_F32 = 157.3   # fp32-input MFMA (v_mfma_f32_32x32x2_f32) = the fp32 vector rate, 1/16 of bf16 (same guide)
# GEMM-equivalents (one = 2*B*H*Sq*Sk*D flops): forward 2 (QK^T, PV); backward 5 in the textbook algorithm (S, dP, dV, dK, dQ),
# 7 executed by the deterministic two-kernel split (S and dP are recomputed by both the dQ and the dK/dV kernel)
GEMMS = {"fwd": (2, 2), "bwd": (5, 7), "fwdbwd": (7, 9)}
# D = 256 (round 4): the two-wave kernels of fasn_fwd_ws256.h / fasn_bwd_ws256.h execute what D <= 128 executes (round 3's feature
# halves: 3 for 2 forward, 9 for 5 backward)
GEMMS256 = {"fwd": (2, 2), "bwd": (5, 7), "fwdbwd": (7, 9)}


def fwd_flops(B, H, S, D, causal):
    return 4.0 * B * H * D * (S * (S + 1) / 2 if causal else S * S)


def pass_flops(which, B, H, S, D, causal, visible_key_fraction=1.0):
    """(algorithmic, executed) flops of one step of `which`. `visible_key_fraction` < 1: a key-padding mask whose fully padded
    64-key tiles the kernels do not walk (C4) - the algorithmic count keeps SURVEY 8(d)'s definition (every score of the
    [S x S] grid), the executed count follows the tiles that actually run (checked against SQ_INSTS_MFMA in profiles/)."""
    g = fwd_flops(B, H, S, D, causal) / 2.0
    gemms = GEMMS256 if D > 128 else GEMMS
    return gemms[which][0] * g, gemms[which][1] * g * visible_key_fraction


def visible_tile_fraction(B, S, tile=64):
    """share of the (batch, 64-key tile) pairs that hold at least one visible key under synth.keypad_mask's lengths"""
    fr = [1.0, 0.875, 0.75, 0.5]
    return sum(-(-int(S * fr[b % 4]) // tile) for b in range(B)) * tile / float(B * S)


def visible_key_fraction(B, S):
    """share of the (batch, key) pairs that are visible under synth.keypad_mask's lengths: the scores a padded batch actually needs"""
    fr = [1.0, 0.875, 0.75, 0.5]
    return sum(int(S * fr[b % 4]) for b in range(B)) / float(B * S)


def lib_sha256():
    path = os.path.join(ROOT, "flash-attention-softmax-n_amd", "libfasn.so")
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def pmc_traffic(workload, which):
    """HBM bytes per launch from the committed rocprofv3 --pmc pass — only if it was taken with THIS build of libfasn.so."""
    prof = os.path.join(ROOT, "profiles", "pmc_latest.json")
    try:
        d = json.load(open(prof))
        if d.get("libfasn_sha256") != lib_sha256():
            return None
        if which == "fwdbwd":   # one forward launch + one fasn_bwd: the sum of the two passes' counters
            parts = [d.get(f"{workload}:{w}", {}).get("hbm_bytes_per_launch") for w in ("fwd", "bwd")]
            return None if None in parts else parts[0] + parts[1]
        return d.get(f"{workload}:{which}", {}).get("hbm_bytes_per_launch")
    except Exception:
        return None


def gpu_sensors(index):
    """Shader clock (MHz) and socket power (W) of GPU `index` from the amdgpu hwmon files (what rocm-smi prints), read while the
    roofline loop's launches are still queued; None where the box does not expose them. The kernels here hold the socket at its
    power limit, so the clock under load - not the nominal 2.4 GHz behind PEAK_TFLOPS - is what the matrix pipes run at."""
    import glob
    try:
        import torch
        want = None
        try:
            pr = torch.cuda.get_device_properties(index)
            want = "%04x:%02x:%02x." % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        except Exception:
            pass
        cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device"))
        cards = [c for c in cards if glob.glob(c + "/hwmon/hwmon*/freq1_input")]
        if want:   # this process's GPU only (a host may expose other tenants' cards too)
            cards = [c for c in cards if want in os.path.realpath(c)]
        elif index < len(cards):
            cards = [cards[index]]
        if not cards:
            return None, None
        h = glob.glob(cards[0] + "/hwmon/hwmon*")[0]
        rd = lambda f: int(open(os.path.join(h, f)).read().strip())
        mhz = rd("freq1_input") / 1e6
        watts = None
        for f in ("power1_average", "power1_input"):
            try:
                watts = rd(f) / 1e6
                break
            except Exception:
                continue
        return mhz, watts
    except Exception:
        return None, None


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_replicas(args):
    """--gpus N without a launcher: start N copies of this script, one per GPU, and relay rank 0's line."""
    port = free_port()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out)
    sys.stdout.flush()
    if any(rcs):
        sys.exit(f"bench.py: replica exit codes {rcs}")


class Control:
    """control plane of the replicas: a barrier and the MAX of one scalar (gloo over loopback); the data path has no collective"""

    def __init__(self, rank, world):
        self.rank, self.world = rank, world
        if world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo", rank=rank, world_size=world)
            self.dist = dist

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()

    def max(self, x):
        if self.world == 1:
            return x
        import torch
        t = torch.tensor([x], dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def gather(self, obj):
        """every rank's `obj` on rank 0 (a list in rank order; None elsewhere)"""
        if self.world == 1:
            return [obj]
        out = [None] * self.world if self.rank == 0 else None
        self.dist.gather_object(obj, out, dst=0)
        return out

    def close(self):
        if self.world > 1:
            self.dist.destroy_process_group()


MIN_TIMED_S = 0.05   # a timed region shorter than this is mostly launch / sync jitter (20 steps of the M0 forward are 10 ms)


def timed(ctl, step, sync, steps, warmup, info=None):
    """W untimed warm-up steps, then the timed region between barrier + device sync on both sides; MAX over ranks. The region is K
    steps, repeated in whole blocks of K until it lasts at least MIN_TIMED_S (the block count comes from one untimed pilot block and
    is agreed across ranks); the returned time is that of ONE block of K steps, `info` gets the block count and this rank's own time."""
    for _ in range(warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):   # pilot block (untimed in the result): how long do K steps take here?
        step()
    sync()
    pilot = max(time.perf_counter() - t0, 1e-6)
    blocks = int(ctl.max(float(max(1, -(-1.25 * MIN_TIMED_S // pilot)))))   # 25 % margin: the pilot block is often the slowest one
    if MIN_TIMED_S <= 0.0:
        blocks = 1
    blocks = min(blocks, 10000)
    ctl.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(blocks * steps):
        step()
    sync()
    dt = (time.perf_counter() - t0) / blocks
    ctl.barrier()
    if info is not None:
        info["timed_steps"] = blocks * steps
        info["local_ms_per_step"] = dt / steps * 1e3
    return ctl.max(dt)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="m0", choices=sorted(WORKLOADS))
    ap.add_argument("--pass", dest="which", default="fwd", choices=["fwd", "bwd", "fwdbwd"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--roofline-launches", type=int, default=0,
                    help="back-to-back launches of the roofline loop (default: 200 forward / 60 backward; counter passes of the profile scripts use fewer)")
    ap.add_argument("--no-extra-passes", action="store_true", help="default forward run: skip the backward / fwdbwd objects")
    ap.add_argument("--min-timed-ms", type=float, default=50.0,
                    help="shortest timed region (default 50 ms; the counter passes of the profile scripts pass 0: exactly K timed steps, counters serialise the launches)")
    ap.add_argument("--stub-step-ms", type=float, default=None,
                    help="testing only: replace the GPU step by a sleep of this many ms (exercises launch + aggregation on CPU)")
    args = ap.parse_args()
    globals()["MIN_TIMED_S"] = args.min_timed_ms * 1e-3
    if args.gpus < 1:
        sys.exit("--gpus must be >= 1")

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return spawn_replicas(args)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: WORLD_SIZE={world} but --gpus {args.gpus}: refusing to report a different GPU count than asked")

    B, H, S, D, dname, n, causal = WORKLOADS[args.workload]
    vis = visible_tile_fraction(B, S) if args.workload == "c4" else 1.0
    visk = visible_key_fraction(B, S) if args.workload == "c4" else 1.0
    peak = PEAK_TFLOPS_F32 if dname == "f32" else PEAK_TFLOPS
    stub = args.stub_step_ms is not None
    ctl = Control(rank, world)
    extra = {}
    kernel_ms = None
    if stub:
        def step():
            time.sleep(args.stub_step_ms * 1e-3 * (1 + rank))   # rank-dependent: the MAX over ranks is rank N-1's time
        tinfo = {}
        dt = timed(ctl, step, lambda: None, args.steps, args.warmup, tinfo)
    else:
        import ctypes  # noqa: F401
        import torch
        assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no CPU fallback)"
        if torch.cuda.device_count() < world:
            sys.exit(f"bench.py: --gpus {world} but only {torch.cuda.device_count()} GPU(s) visible")
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        dtype = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[dname]

        import flash_attention_softmax_n_amd as pkg
        from flash_attention_softmax_n_amd import synth
        lib, fa = pkg._lib.load(), pkg.flash_attn
        q, k, v = (synth.counter_normal((B, H, S, D), seed, dtype=dtype, device=dev) for seed in (101, 102, 103))
        do = synth.counter_normal((B, H, S, D), 104, std=1.0, dtype=dtype, device=dev)
        bias = mask = None
        if args.workload == "c4":
            bias = synth.alibi_bias(H, S, S, dtype, device=dev)
            mask = synth.keypad_mask(B, S, device=dev)
        sync = torch.cuda.synchronize
        stream = torch.cuda.current_stream().cuda_stream
        out_holder = {}

        def step_fwd():
            with torch.no_grad():
                out_holder["o"] = pkg.flash_attention_n(q, k, v, softmax_n_param=n, is_causal=causal, attn_mask=mask, attn_bias=bias)

        qg, kg, vg = (t.detach().clone().requires_grad_() for t in (q, k, v))

        def step_fwdbwd():
            qg.grad = kg.grad = vg.grad = None
            o = pkg.flash_attention_n(qg, kg, vg, softmax_n_param=n, is_causal=causal, attn_mask=mask, attn_bias=bias)
            o.backward(do)

        # the backward alone: one fasn_bwd ([delta +] dQ + dK/dV) on the saved forward state, gradients preallocated
        o_s = torch.empty_like(q)
        lse = torch.empty(B, H, S, dtype=torch.float32, device=dev)
        m8 = None if mask is None else mask.expand(B, H, S, S).view(torch.uint8)
        b4 = None if bias is None else bias.unsqueeze(0).expand(B, H, S, S)
        fargs = pkg._lib.FwdArgs()
        fa._fill_fwd(fargs, q, k, v, o_s, lse, m8, b4, n, 1.0 / D ** 0.5, causal)
        pkg._lib.check(lib.fasn_fwd(fargs, stream), "fasn_fwd")
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        delta = torch.empty_like(lse)
        bargs = pkg._lib.BwdArgs()
        fa._fill_fwd(bargs.fwd, q, k, v, o_s, lse, m8, b4, n, 1.0 / D ** 0.5, causal)
        bargs.dout, bargs.dq, bargs.dk, bargs.dv = (fa._view4(t) for t in (do, dq, dk, dv))
        bargs.delta = delta.data_ptr()
        bargs.flags = 0
        ws_bytes = lib.fasn_bwd_workspace_bytes(bargs)   # (0 with libfasn.so: the backward needs no scratch beyond delta)
        if ws_bytes:
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
            bargs.workspace, bargs.workspace_bytes = ws.data_ptr(), ws_bytes

        def step_bwd():
            pkg._lib.check(lib.fasn_bwd(bargs, stream), "fasn_bwd")

        steps_of = {"fwd": step_fwd, "bwd": step_bwd, "fwdbwd": step_fwdbwd}

        sensors = []

        def kernel_time(fn, iters):
            """average duration of back-to-back launches: events on the launch stream (torch's current stream IS the stream
            the ctypes call launches on), >= 200 launches so the clocks settle"""
            for _ in range(10):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                fn()
            e1.record()
            e1.synchronize()
            return e0.elapsed_time(e1) / iters

        # The roofline measurement (dominant kernels, back to back) runs FIRST: it is part of what this script reports anyway,
        # and it leaves the device at its sustained clocks, so that the W warm-up + K timed steps below measure the steady
        # state a long-running job sees rather than the power ramp of a cold GPU (+5 % on the first ~20 launches).
        # (the forward goes through fasn_fwd_ws with the workspace it asks for, like the front end does: split-K partials or the
        # work counters of the persistent bias launches)
        fws_bytes = lib.fasn_fwd_workspace_bytes(fargs)
        fws = torch.empty(max(fws_bytes, 16), dtype=torch.uint8, device=dev)
        raw = {"fwd": (lambda: lib.fasn_fwd_ws(fargs, fws.data_ptr(), fws_bytes, stream)) if fws_bytes else (lambda: lib.fasn_fwd(fargs, stream)),
               "bwd": lambda: lib.fasn_bwd(bargs, stream)}
        if args.which in raw:
            kernel_ms = kernel_time(raw[args.which], args.roofline_launches or max(200 if args.which == "fwd" else 60, args.steps))
        else:
            kernel_ms = kernel_time(raw["fwd"], 100) + kernel_time(raw["bwd"], 60)

        tinfo = {}
        dt = timed(ctl, steps_of[args.which], sync, args.steps, args.warmup, tinfo)

        def sample_under_load(fn, ms):
            """Clock / power under THIS load, AFTER everything that is timed: about 0.4 s of back-to-back launches, the amdgpu
            hwmon sensors polled while they run (the last clock before completion and the highest power reading are kept - the
            power sensor lags the load). Its own loop on purpose: polled inside the roofline loop, the sensor reads (SMU
            queries) made the timed steps that followed 11 % slower."""
            n = int(min(2000, max(50, 400.0 / max(ms, 1e-3))))
            e1 = torch.cuda.Event()
            for _ in range(n):
                fn()
            e1.record()
            last, pmax = None, None
            while not e1.query():
                mhz, watts = gpu_sensors(dev.index or 0)
                if mhz is None:
                    break
                last = mhz
                pmax = watts if pmax is None or (watts is not None and watts > pmax) else pmax
                time.sleep(0.01)
            e1.synchronize()
            if last is not None:
                sensors.append((last, pmax))

        if args.which == "fwd" and args.workload == "m0" and world == 1 and not args.no_extra_passes:
            # driver-visible backward numbers next to the headline (same K and W, measured after the headline's timed region)
            for w in ("bwd", "fwdbwd"):
                winfo = {}
                dtw = timed(ctl, steps_of[w], sync, args.steps, args.warmup, winfo)
                kms = kernel_time(raw["bwd"], 60) if w == "bwd" else None
                alg, exe = pass_flops(w, B, H, S, D, causal, vis)
                extra["backward" if w == "bwd" else "fwdbwd"] = {
                    "steps_per_s": args.steps / dtw, "ms_per_step": dtw / args.steps * 1e3, "steps": args.steps, "warmup": args.warmup,
                    "timed_steps": winfo["timed_steps"],
                    "algorithmic_tflops": alg / (dtw / args.steps) / 1e12, "executed_tflops": exe / (dtw / args.steps) / 1e12,
                    "frac_of_peak_algorithmic": alg / (dtw / args.steps) / 1e12 / PEAK_TFLOPS,
                    **({"kernels_ms": kms, "kernels_frac_of_peak_executed": exe / (kms * 1e-3) / 1e12 / PEAK_TFLOPS} if kms else {})}

        # every rank samples ITS GPU while all of them run the same loop: a bent 1 -> N curve can then be attributed from this one
        # record (a shared power budget shows as lower clocks on all ranks, host launch contention as equal clocks and longer steps)
        ctl.barrier()
        sample_under_load(raw[args.which] if args.which in raw else raw["bwd"], kernel_ms)

    mine = {"rank": rank, "ms_per_step": tinfo.get("local_ms_per_step")}
    if not stub and sensors and sensors[0][0]:
        mine["sclk_mhz"], mine["socket_power_w"] = sensors[0]
    per_rank = ctl.gather(mine)
    if rank != 0:
        ctl.close()
        return

    ops_per_s = world * args.steps / dt
    alg, exe = pass_flops(args.which, B, H, S, D, causal, vis)
    names = {"fwd": "forward", "bwd": "backward", "fwdbwd": "forward+backward"}
    line = {
        "metric": f"attn-ops/sec (flash_attention_n {names[args.which]})", "value": ops_per_s, "unit": "attn-ops/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "timed_steps": tinfo.get("timed_steps", args.steps),   # the timed region: whole blocks of `steps` steps, at least 50 ms of work
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dname, "data": "synthetic",
        "config": {"workload": f"{args.workload}: flash_attention_n {args.which} (B={B},H={H},S={S},D={D}) n={n} causal={causal}"
                               + (" + ALiBi bias [H,L,S] + key-padding mask [B,1,1,S]" if args.workload == "c4" else ""),
                   "parallelism": f"replicas x{world} (no sharding, no collective)",
                   "output_elements_per_s": ops_per_s * B * H * S * D,
                   "score_elements_per_s": ops_per_s * B * H * S * S * (0.5 if causal else 1.0)},
    }
    if world > 1:
        line["per_rank"] = per_rank   # each rank's own time per step and its GPU's clock / power under the common load
    if stub:
        line["data"] = "stub (no GPU work: launch + aggregation test)"
    else:
        # the kernels behind the timed call, as the library's own launch tables name them (fasn_launch_plan: the host side of the call run
        # with recording launch sites - the names are those of the code objects a rocprofv3 kernel trace of this command shows)
        plans = {"fwd": [pkg._lib.FASN_PLAN_FWD_WS if fws_bytes else pkg._lib.FASN_PLAN_FWD], "bwd": [pkg._lib.FASN_PLAN_BWD]}
        plans["fwdbwd"] = plans["fwd"] + plans["bwd"]
        kernels = [f"{nm} grid={g} block={b}" for code in plans[args.which] for nm, g, b, _ in pkg._lib.launch_plan_described(bargs, code)]
        kernels_raw = [nm for code in plans[args.which] for nm, _, _, _ in pkg._lib.launch_plan(bargs, code)]   # (the code objects' own names: what a rocprofv3 trace shows)
        dense = alg / (kernel_ms * 1e-3) / 1e12
        # C4: SURVEY 8(d) counts every score of the [S x S] grid, padded keys included. Nobody needs those scores and the kernels skip
        # their tiles, so `achieved` / `frac` - the numbers a summary quotes - are taken on the VISIBLE keys; the dense-score figures
        # stay next to them (`achieved_dense_scores`, `frac_dense_scores`), and `frac_executed` follows the 64-key tiles that run.
        achieved = dense * visk
        line["roofline"] = {
            "bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
            **({"frac_on_visible_keys": achieved / peak, "visible_key_fraction": visk, "visible_key_tile_fraction": vis,
                "achieved_dense_scores": dense, "frac_dense_scores": dense / peak} if vis < 1.0 else {}),
            "traffic": pmc_traffic(args.workload, args.which), "kernel_ms": kernel_ms,
            "kernels": kernels, "kernels_raw": kernels_raw,
            "algorithmic_flops_per_launch": alg * visk, "executed_flops_per_launch": exe,
            "gemm_equivalents": {"algorithmic": (GEMMS256 if D > 128 else GEMMS)[args.which][0], "executed": (GEMMS256 if D > 128 else GEMMS)[args.which][1]},
            "frac_executed": exe / (kernel_ms * 1e-3) / 1e12 / peak}
        sm = [x for x in sensors if x[0]]
        if sm:   # nominal peak scaled to the clock the part held during the roofline loop (informative; `frac` stays against the nominal peak)
            mhz, watts = sm[0]
            line["roofline"]["under_load"] = {"sclk_mhz": mhz, "socket_power_w": watts, "nominal_mhz": 2400.0,
                                              "frac_of_clock_adjusted_peak": achieved / (peak * mhz / 2400.0)}
        line.update(extra)

    if not stub and world == 1 and not args.no_cpu_baseline:
        # CPU baseline + accuracy on a bounded sample: batch 0, a few heads, through the oracle's eager restatement of
        # slow_attention_n in the native dtype (exactly what the reference runs on CPU), all host threads.
        import platform
        import torch
        from oracle.ref_attention import ref_attention_n
        full = args.workload == "c1"          # BASELINE config 1 is timed in full: all of (2,2,128,32) fp32, many repeats
        hs = H if S <= 4096 else 2
        nb = B if full else 1
        qc, kc, vc = (t[0:nb, :hs].cpu() for t in (q, k, v))
        threads = torch.get_num_threads()
        bc = None if bias is None else bias[:hs].cpu()
        mc = None if mask is None else mask[0:1].cpu()
        reps = 200 if full else 1
        ref = ref_attention_n(qc, kc, vc, softmax_n_param=n, is_causal=causal, attn_bias=bc, attn_mask=mc) if full else None   # warm-up
        t1 = time.perf_counter()
        for _ in range(reps):
            ref = ref_attention_n(qc, kc, vc, softmax_n_param=n, is_causal=causal, attn_bias=bc, attn_mask=mc)
        cpu_dt = (time.perf_counter() - t1) / reps
        frac = (nb * hs) / (B * H)
        # the sample predicts the whole workload: when that is within the budget of a bounded baseline (<= 25 s), run ALL of it - the remaining
        # (batch, head) slices through the same call - and report the measured time of one whole forward op instead of an extrapolation
        whole = False
        if not full and mask is None and cpu_dt / frac <= 25.0:
            qh, kh, vh = q.cpu(), k.cpu(), v.cpu()   # (outside the timed region: the baseline's inputs are resident in host memory, as the GPU's are in HBM)
            bh_ = None if bias is None else bias.cpu()
            t1 = time.perf_counter()
            for b0 in range(B):
                for h0 in range(0, H, hs):
                    if b0 == 0 and h0 == 0:
                        continue
                    ref_attention_n(qh[b0:b0 + 1, h0:h0 + hs], kh[b0:b0 + 1, h0:h0 + hs], vh[b0:b0 + 1, h0:h0 + hs], softmax_n_param=n,
                                    is_causal=causal, attn_bias=None if bh_ is None else bh_[h0:h0 + hs])
            cpu_dt += time.perf_counter() - t1
            frac, whole = 1.0, True
            del qh, kh, vh
        cpu_model = platform.processor() or "unknown"
        try:
            cpu_model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
        except Exception:
            pass
        line["cpu_baseline"] = {"value": frac / cpu_dt, "unit": "attn-ops/s (forward)", "cores": threads, "kind": "port", "extrapolated": not (full or whole),
                                "cpu": cpu_model,
                                "sample": (f"IN FULL: the whole (B={B},H={H},S={S},D={D}) {dname} forward, mean of {reps} runs = {cpu_dt * 1e3:.3f} ms; "
                                           f"oracle/ref_attention.py (eager, as slow_attention_n)") if full else
                                          (f"IN FULL: every (batch, head) slice of the (B={B},H={H},S={S},D={D}) {dname} forward, {hs} heads per call, {cpu_dt:.2f} s in all "
                                           f"(inputs resident in host memory); oracle/ref_attention.py (eager {dname}, as slow_attention_n)") if whole else
                                          (f"EXTRAPOLATED: batch 0, heads 0..{hs - 1} of the same inputs ({hs}/{B * H} of one forward op) took {cpu_dt:.2f} s, "
                                           f"scaled linearly x{B * H // hs}; oracle/ref_attention.py (eager {dname}, as slow_attention_n)")}
        out = out_holder.get("o")
        if out is None:
            step_fwd()
            out = out_holder["o"]
        line["max_abs_err"] = float((out[0:nb, :hs].float().cpu() - ref.float()).abs().max())
        ref32 = ref_attention_n(qc[:, :2].float(), kc[:, :2].float(), vc[:, :2].float(), softmax_n_param=n, is_causal=causal,
                                attn_bias=None if bc is None else bc[:2].float(), attn_mask=mc)
        line["max_abs_err_vs_fp32_oracle"] = float((out[0:nb, :2].float().cpu() - ref32).abs().max())
    print(json.dumps(line), flush=True)
    ctl.close()


if __name__ == "__main__":
    main()
