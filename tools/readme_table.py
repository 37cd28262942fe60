#!/usr/bin/env python3
"""The README's per-workload table from the bench lines of a tools/prof_round6.sh call:  tools/readme_table.py gpurun_out/TAG
(forward | backward with the executed fraction in brackets | forward + backward | HBM bytes per launch from the PMC passes)."""
import json
import sys

NAMES = {"m0": "**M0** (8,16,4096,64) bf16", "c2": "C2 (8,16,1024,64) bf16", "c3": "C3 (8,16,4096,64) fp16 causal", "c5": "C5 (64,16,4096,64) bf16 causal",
         "c4": "**C4** (4,32,8192,128) bf16, n = 0.5, ALiBi bias + key padding", "d256": "d256 (4,16,4096,256) bf16",
         "t32": "t32 (32,16,1024,32) fp16 (the reference's Triton test shape)", "c1": "C1 (2,2,128,32) fp32 (the reference's CPU case)"}


def nbytes(x):
    return "-" if x is None else (f"{x / 1e9:.2f} GB" if x >= 1e9 else f"{x / 1e6:.0f} MB")


def ms(x):
    return f"{x:.3f} ms" if x < 2 else f"{x:.2f} ms"


def main():
    rows = {}
    for line in open(sys.argv[1] + "/bench_all.jsonl"):
        d = json.loads(line)
        w = d["config"]["workload"].split(":")[0]
        p = d["config"]["workload"].split("flash_attention_n ")[1].split(" ")[0]
        rows.setdefault(w, {})[p] = d
    print("| workload (`bench.py --workload`) | forward | backward (executed) | forward + backward | HBM bytes per launch, forward / backward (PMC) |")
    print("|---|---|---|---|---|")
    for w in ("m0", "c2", "c3", "c5", "c4", "d256", "t32", "c1"):
        r = rows.get(w, {})
        f, b, fb = r.get("fwd"), r.get("bwd"), r.get("fwdbwd")
        if w == "c1":
            print(f"| {NAMES[w]} | {ms(f['ms_per_step'])} (launch-bound) | | | |")
            continue
        fr, br = f["roofline"], b["roofline"]
        vis = " of visible keys" if w == "c4" else ""
        print(f"| {NAMES[w]} | {ms(f['ms_per_step'])}, {fr['frac']:.3f}{vis} | {ms(b['ms_per_step'])}, {br['frac']:.3f} ({br['frac_executed']:.3f}) | "
              f"{ms(fb['ms_per_step'])} | {nbytes(fr['traffic'])} / {nbytes(br['traffic'])} |")
    d = json.load(open(sys.argv[1] + "/bench_default.json"))
    print(f"\ndefault line: {d['value']:.0f} attn-ops/s, {d['ms_per_step']:.4f} ms, roofline.frac {d['roofline']['frac']:.3f}, kernel {d['roofline']['kernel_ms']:.3f} ms, "
          f"backward {d['backward']['ms_per_step']:.3f} ms ({d['backward']['frac_of_peak_algorithmic']:.3f}), forward + backward {d['fwdbwd']['ms_per_step']:.3f} ms, "
          f"cpu {d['cpu_baseline']['value']:.3f}")


if __name__ == "__main__":
    main()
