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

sys.path.insert(0, os.path.join(ROOT, "tests"))


@pytest.fixture(scope="module")
def table():
    import spill_map
    if not os.path.exists(spill_map.READELF):
        pytest.skip("llvm-readelf not available")
    if not os.path.exists(LIB):   # *.so is git-ignored: a checkout without a build has nothing to inspect
        pytest.skip("libfasn.so not built (run __graft_entry__.build() or make -C flash-attention-softmax-n_amd/csrc)")
    t = spill_map.kernel_table(LIB)
    assert len(t) > 300, "could not read the kernel metadata of libfasn.so"
    return t


def _demangled(names):
    import subprocess
    import spill_map
    out = subprocess.run([spill_map.CXXFILT], input="\n".join(names), capture_output=True, text=True, check=True).stdout.splitlines()
    return dict(zip(out, names))


def test_baseline_kernels_do_not_spill(table, pkg):
    """Which kernels a BASELINE config launches is asked of the library itself (fasn_launch_plan: the launch tables run with recording
    launch sites, no GPU needed) - not kept as a list of mangled-name patterns that has to follow every change of a template's arity."""
    from baseline_plans import CONFIGS, kernels
    by_pretty = _demangled(sorted(table))
    rows, missing = [], []
    for cfg in CONFIGS:
        for which in ("fwd", "bwd"):
            for name, grid, block, lds in kernels(pkg, cfg, which):
                hit = [m for d, m in by_pretty.items() if d.startswith("void fasn::" + name + "(")]
                if len(hit) != 1:
                    missing.append((cfg, which, name, hit))
                    continue
                rows.append((f"{cfg} {which}", name, table[hit[0]]))
    assert not missing, f"launch-plan names without exactly one code object: {missing}"
    assert len(rows) >= 3 * len(CONFIGS)   # forward + dQ + dK/dV per config (+ delta where the dQ kernel does not compute it itself)
    print()
    for what, n, v in rows:
        print(f"{v.get('vgpr', 0):4d} regs  spill {v.get('spill', 0):3d}  scratch {v.get('scratch', 0):4d} B  {what}: {n[:110]}")
    bad = [(what, n, v["spill"], v.get("scratch", 0)) for what, n, v in rows if v.get("spill", 0) or v.get("scratch", 0)]
    assert not bad, f"kernels reachable from a BASELINE config spill: {bad}"


def _to_memory(v):
    """Spilled vector registers that reach scratch MEMORY. A kernel that runs one wave per SIMD has 256 accumulation registers next to its 256
    vector registers and hipcc parks values there (`v_accvgpr_write` / `_read`, one instruction, no memory, no wait): the code object's metadata
    counts those as spilled registers too, with a scratch size of zero (round 6: fasn_fwd_kernel<*, 256, ...general> and the plain / causal
    fasn_f32_dkdv_kernel<128> are such kernels). The gate is about the `s_waitcnt vmcnt(0)` of a scratch reload, so it counts those as 0."""
    return v.get("spill", 0) if v.get("scratch", 0) > 0 else 0


def test_no_kernel_spills_more_than_recorded(table):
    allow = json.load(open(ALLOW))
    worse = {n: (_to_memory(v), allow.get(n, 0)) for n, v in table.items() if _to_memory(v) > allow.get(n, 0)}
    spilling = sorted(((_to_memory(v), n) for n, v in table.items() if _to_memory(v)), reverse=True)
    print(f"\n{len(table)} kernels, {len(spilling)} with vector registers spilled to scratch memory (allowance file: {len(allow)})")
    for s, n in spilling:
        print(f"  spill {s:4d}  {n[9:120]}")
    assert not worse, f"spill regressions (kernel: (now, allowed)): {worse}"
