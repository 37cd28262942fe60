#!/usr/bin/env python3
"""Register / spill table of compiled kernels.

  spill_map.py libfasn.so [substring]     per kernel of the library: VGPRs, AGPRs, spilled VGPRs, scratch bytes, LDS bytes
                                          (reads the gfx950 code objects out of the fat binary, metadata notes via llvm-readelf)
  spill_map.py file.s kernel-substring    where does ONE kernel of a `hipcc -S` listing touch scratch: per basic block line count,
                                          MFMA count, scratch stores / loads
tests/test_spill_gate.py uses kernel_table() to keep every instantiation a BASELINE config reaches free of spills."""
import os
import re
import struct
import subprocess
import sys
import tempfile

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
CXXFILT = "/opt/rocm/lib/llvm/bin/llvm-cxxfilt" if os.path.exists("/opt/rocm/lib/llvm/bin/llvm-cxxfilt") else "c++filt"


def code_objects(path):
    """the gfx950 ELF images inside a HIP fat binary (clang offload bundles)"""
    data = open(path, "rb").read()
    out = []
    for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", data):
        off = m.start()
        p = off + 24
        (cnt,) = struct.unpack_from("<Q", data, p)
        p += 8
        for _ in range(cnt):
            eo, es, ts = struct.unpack_from("<QQQ", data, p)
            p += 24
            triple = data[p:p + ts].decode(errors="replace")
            p += ts
            if "gfx950" in triple and es > 0:
                out.append(data[off + eo:off + eo + es])
    return out


def kernel_table(path):
    """{mangled kernel name: {"vgpr", "agpr", "spill", "scratch", "lds", "sgpr"}} for every kernel of the library"""
    table = {}
    keys = {".vgpr_count": "vgpr", ".agpr_count": "agpr", ".vgpr_spill_count": "spill", ".private_segment_fixed_size": "scratch",
            ".group_segment_fixed_size": "lds", ".sgpr_count": "sgpr", ".sgpr_spill_count": "sgpr_spill"}
    with tempfile.TemporaryDirectory() as td:
        for i, co in enumerate(code_objects(path)):
            f = os.path.join(td, f"co{i}.elf")
            open(f, "wb").write(co)
            notes = subprocess.run([READELF, "--notes", f], capture_output=True, text=True).stdout
            cur = None
            for line in notes.splitlines():
                line = line.strip()
                if line.startswith("- .") or line.startswith("-   ."):   # a new list element (kernel or argument)
                    line = line.lstrip("- ").strip()
                    if cur is not None and "name" in cur and cur["name"].endswith(".kd") is False and "vgpr" in cur:
                        table[cur["name"]] = cur
                    if cur is None or "vgpr" in cur:
                        cur = {}
                m = re.match(r"(\.[a-z_]+):\s+(.*)", line)
                if not m or cur is None:
                    continue
                k, v = m.group(1), m.group(2).strip()
                if k == ".name" and "name" not in cur and (v.startswith("_Z") or v.startswith("'_Z")):
                    cur["name"] = v.strip("'")
                elif k in keys and keys[k] not in cur:
                    cur[keys[k]] = int(v)
            if cur is not None and "name" in cur and "vgpr" in cur:
                table[cur["name"]] = cur
    return table


def demangle(names):
    try:
        r = subprocess.run([CXXFILT], input="\n".join(names), capture_output=True, text=True)
        return dict(zip(names, r.stdout.splitlines()))
    except Exception:
        return {n: n for n in names}


def main():
    if sys.argv[1].endswith(".so"):
        t = kernel_table(sys.argv[1])
        sub = sys.argv[2] if len(sys.argv) > 2 else ""
        dm = demangle(list(t))
        rows = sorted(((dm[n], v) for n, v in t.items() if sub in dm[n] or sub in n), key=lambda x: (-x[1].get("spill", 0), x[0]))
        print(f"{len(rows)} kernels, {sum(1 for _, v in rows if v.get('spill', 0))} with spilled VGPRs")
        for n, v in rows:
            n = re.sub(r"\(fasn::\w+\)$", "", n).replace("void fasn::", "").replace("fasn::", "")
            print(f"{v.get('vgpr', 0):4d} v {v.get('agpr', 0):4d} a  spill {v.get('spill', 0):4d}  scratch {v.get('scratch', 0):5d} B  lds {v.get('lds', 0):6d} B  {n[:150]}")
        return
    lines = open(sys.argv[1]).read().split("\n")
    pat = sys.argv[2]
    start = end = None
    for i, l in enumerate(lines):
        if start is None and l.endswith(":") and pat in l and not l.startswith("."):
            start = i
        elif start is not None and l.startswith(".Lfunc_end"):
            end = i
            break
    body = lines[start:end]
    labels = [i for i, l in enumerate(body) if re.match(r"\.LBB\d+_\d+:", l)] + [len(body)]
    tot_st = tot_ld = 0
    for a, b in zip([0] + labels[:-1], labels):
        blk = body[a:b]
        st = sum("scratch_store" in l for l in blk)
        ld = sum("scratch_load" in l for l in blk)
        tot_st += st
        tot_ld += ld
        if st or ld:
            print("%-60s %4d lines mfma %3d  scratch st %3d ld %3d" % (body[a][:60], b - a, sum("v_mfma" in l for l in blk), st, ld))
    print("total scratch stores %d loads %d" % (tot_st, tot_ld))


if __name__ == "__main__":
    main()
