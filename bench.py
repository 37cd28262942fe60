#!/usr/bin/env python
"""bench.py — headline benchmark of the hot path (BASELINE.json): attn-ops/s of flash_attention_n forward at
(B=8, H=16, S=4096, D=64) bf16, n=1, non-causal, on N replicated GPUs (no sharding, no RCCL on the data path).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...)

A step = one forward pass of the fused kernel over the whole (8,16,4096,64) batch, inputs resident in HBM.
Rank 0 prints ONE JSON line; `value` = total attn-ops/s over all replicas = N * K / max-over-ranks(wall time of K steps).
Also in the line: `roofline` (dominant kernel vs the dense bf16 MFMA peak, duration from HIP events on the launch stream)
and, at N=1, `cpu_baseline` (the oracle's eager restatement of slow_attention_n timed on the host cores on a bounded
sample) plus `max_abs_err` of the GPU result against that same oracle output.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (B, H, S, D, dtype, n, causal)
    "m0": (8, 16, 4096, 64, torch.bfloat16, 1.0, False),   # the shape BASELINE.json's metric is quoted on
    "c2": (8, 16, 1024, 64, torch.bfloat16, 1.0, False),
    "c3": (8, 16, 4096, 64, torch.float16, 1.0, True),
    "c5": (64, 16, 4096, 64, torch.bfloat16, 1.0, True),
    "c4": (4, 32, 8192, 128, torch.bfloat16, 0.5, False),   # + dense ALiBi bias [H,L,S] and key-padding mask [B,1,1,S]
}
PEAK_TFLOPS = 2500.0  # dense bf16/fp16 MFMA peak, MI355X (MI355X_MICROARCH.md)


def fwd_flops(B, H, S, D, causal):
    return 4.0 * B * H * D * (S * (S + 1) / 2 if causal else S * S)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="m0", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus or world == 1, f"WORLD_SIZE={world} but --gpus {args.gpus}"
    import torch.distributed as dist
    if world > 1:
        # control plane only (barrier + max of one float): gloo over loopback; the data path has no collective
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no CPU fallback)"
    torch.cuda.set_device(local_rank % torch.cuda.device_count())
    dev = torch.device("cuda", torch.cuda.current_device())

    import flash_attention_softmax_n_amd as pkg
    from flash_attention_softmax_n_amd import synth

    B, H, S, D, dtype, n, causal = WORKLOADS[args.workload]
    q, k, v = (synth.counter_normal((B, H, S, D), seed, dtype=dtype, device=dev) for seed in (101, 102, 103))

    bias = mask = None
    if args.workload == "c4":
        bias = synth.alibi_bias(H, S, S, dtype, device=dev)
        mask = synth.keypad_mask(B, S, device=dev)

    def step():
        return pkg.flash_attention_n(q, k, v, softmax_n_param=n, is_causal=causal, attn_mask=mask, attn_bias=bias)

    with torch.no_grad():
        for _ in range(args.warmup):
            out = step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if world > 1:
            dist.barrier()
            t = torch.tensor([dt], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())

    # dominant-kernel duration: HIP events on the launch stream around back-to-back launches of the same kernel
    fa = pkg.flash_attn
    o2 = torch.empty_like(q)
    lse = torch.empty(B, H, S, dtype=torch.float32, device=dev)
    a = pkg._lib.FwdArgs()
    fa._fill_fwd(a, q, k, v, o2, lse, None if mask is None else mask.expand(B, H, S, S).view(torch.uint8),
                 None if bias is None else bias.unsqueeze(0).expand(B, H, S, S), n, 1.0 / D ** 0.5, causal)
    ms = ctypes.c_float(0.0)
    stream = torch.cuda.current_stream().cuda_stream
    pkg._lib.check(pkg._lib.load().fasn_time_fwd(a, stream, 10, max(200, args.steps), ctypes.byref(ms)), "fasn_time_fwd")
    kernel_ms = float(ms.value)
    flops = fwd_flops(B, H, S, D, causal)
    achieved = flops / (kernel_ms * 1e-3) / 1e12

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    ops_per_s = world * args.steps / dt
    traffic = None
    prof = os.path.join(ROOT, "profiles", "pmc_latest.json")  # per-launch HBM bytes from a committed rocprofv3 --pmc run
    if os.path.exists(prof):
        try:
            traffic = json.load(open(prof)).get(args.workload, {}).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    line = {
        "metric": "attn-ops/sec (flash_attention_n forward)", "value": ops_per_s, "unit": "attn-ops/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": {torch.bfloat16: "bf16", torch.float16: "f16"}[dtype], "data": "synthetic",
        "config": {"workload": f"{args.workload}: flash_attention_n fwd (B={B},H={H},S={S},D={D}) n={n} causal={causal}",
                   "parallelism": f"replicas x{world} (no sharding, no collective)",
                   "output_elements_per_s": ops_per_s * B * H * S * D,
                   "score_elements_per_s": ops_per_s * B * H * S * S * (0.5 if causal else 1.0)},
        "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_TFLOPS,
                     "traffic": traffic, "kernel_ms": kernel_ms, "algorithmic_flops_per_launch": flops},
    }

    if world == 1 and not args.no_cpu_baseline:
        # CPU baseline + accuracy on a bounded sample: batch 0, all heads, through the oracle's eager restatement of
        # slow_attention_n in the native dtype (exactly what the reference runs on CPU), all host threads.
        from oracle.ref_attention import ref_attention_n
        hs = H if S <= 4096 else 2
        qc, kc, vc = (t[0:1, :hs].cpu() for t in (q, k, v))
        threads = torch.get_num_threads()
        t1 = time.perf_counter()
        bc = None if bias is None else bias[:hs].cpu()
        mc = None if mask is None else mask[0:1].cpu()
        ref = ref_attention_n(qc, kc, vc, softmax_n_param=n, is_causal=causal, attn_bias=bc, attn_mask=mc)
        cpu_dt = time.perf_counter() - t1
        frac = hs / (B * H)
        line["cpu_baseline"] = {"value": frac / cpu_dt, "unit": "attn-ops/s", "cores": threads, "kind": "port",
                                "sample": f"batch 0, heads 0..{hs - 1} of the same inputs ({hs}/{B * H} of one op), {cpu_dt:.2f} s, "
                                          f"scaled linearly; oracle/ref_attention.py (eager {line['dtype']}, as slow_attention_n)"}
        line["max_abs_err"] = float((out[0:1, :hs].float().cpu() - ref.float()).abs().max())
        ref32 = ref_attention_n(qc[:, :2].float(), kc[:, :2].float(), vc[:, :2].float(), softmax_n_param=n, is_causal=causal,
                                attn_bias=None if bc is None else bc[:2].float(), attn_mask=mc)
        line["max_abs_err_vs_fp32_oracle"] = float((out[0:1, :2].float().cpu() - ref32).abs().max())
    print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
