"""Spill gate (CPU): no kernel instantiation that a BASELINE.json config reaches may spill vector registers, and no other kernel may
spill MORE than the recorded allowance (tests/golden/spill_allowance.json: the round-4 state of the non-BASELINE instantiations - head
dim 256, fp32 at head dim 128, dropout / grouped-K/V variants - which can only shrink). A scratch reload inside a tile loop is a
`s_waitcnt vmcnt(0)` that also drains the K/V prefetch in flight, so a spill in a hot kernel is a performance bug, not a detail.
The table comes from the code objects inside libfasn.so (tools/spill_map.py: offload bundles -> llvm-readelf metadata notes)."""
import json
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
LIB = os.path.join(ROOT, "flash-attention-softmax-n_amd", "libfasn.so")
ALLOW = os.path.join(ROOT, "tests", "golden", "spill_allowance.json")

# (config, pass) -> regular expression over the mangled kernel names it launches (fasn_fwd_path + the launch tables of
# csrc/fasn_launch.h, fasn_bwd_launch.h, fasn_bwd_d64.hip, fasn_f32.hip)
BASELINE_KERNELS = {
    "c1 fp32 (2,2,128,32): forward / delta / dQ / dK,dV": r"fasn_f32_(fwd|dq|dkdv)_kernelILi32ELi0E|fasn_f32_delta_kernelILi32E",
    "m0 / c2 bf16 D=64 plain forward": r"fasn_fwd_kernelINS_8bf16_tagELi64ELi2ELi0E",
    "c3 f16 D=64 causal forward": r"fasn_fwd_kernelINS_7f16_tagELi64ELi1ELi1E",
    "c5 bf16 D=64 causal forward": r"fasn_fwd_kernelINS_8bf16_tagELi64ELi1ELi1E",
    "c4 bf16 D=128 ALiBi + key padding forward": r"fasn_fwd_kernelINS_8bf16_tagELi128ELi1ELi7E",
    "m0 / c2 / c3 / c5 backward (delta, pipelined dQ, pipelined dK/dV)": r"fasn_bwd_delta_kernelINS_(8bf16|7f16)_tagELi64E|fasn_bwd_dq_pipe_kernelINS_\w+_tagELi[01]ELi0E|fasn_bwd_dkdv_pipe_kernelINS_\w+_tagELi[01]ELi0E",
    "c4 backward (delta, two-wave dQ, two-wave dK/dV)": r"fasn_bwd_delta_kernelINS_8bf16_tagELi128E|fasn_bwd_dq_ws_kernelINS_8bf16_tagELi128ELi7ELi0E|fasn_bwd_dkdv_ws_kernelINS_8bf16_tagELi128ELi7ELi0ELi0E",
}


@pytest.fixture(scope="module")
def table():
    import spill_map
    if not os.path.exists(spill_map.READELF):
        pytest.skip("llvm-readelf not available")
    t = spill_map.kernel_table(LIB)
    assert len(t) > 300, "could not read the kernel metadata of libfasn.so"
    return t


def test_baseline_kernels_do_not_spill(table):
    rows = []
    for what, pat in BASELINE_KERNELS.items():
        hit = {n: v for n, v in table.items() if re.search(pat, n)}
        assert hit, f"{what}: no kernel matches {pat} (launch tables changed? update BASELINE_KERNELS)"
        for n, v in sorted(hit.items()):
            rows.append((what, n, v))
    print()
    for what, n, v in rows:
        print(f"{v.get('vgpr', 0):4d} regs  spill {v.get('spill', 0):3d}  scratch {v.get('scratch', 0):4d} B  {what}: {n[9:110]}")
    bad = [(what, n, v["spill"], v.get("scratch", 0)) for what, n, v in rows if v.get("spill", 0) or v.get("scratch", 0)]
    assert not bad, f"kernels reachable from a BASELINE config spill: {bad}"


def test_no_kernel_spills_more_than_recorded(table):
    allow = json.load(open(ALLOW))
    worse = {n: (v.get("spill", 0), allow.get(n, 0)) for n, v in table.items() if v.get("spill", 0) > allow.get(n, 0)}
    spilling = sorted(((v.get("spill", 0), n) for n, v in table.items() if v.get("spill", 0)), reverse=True)
    print(f"\n{len(table)} kernels, {len(spilling)} with spilled VGPRs (allowance file: {len(allow)})")
    for s, n in spilling:
        print(f"  spill {s:4d}  {n[9:120]}")
    assert not worse, f"spill regressions (kernel: (now, allowed)): {worse}"
