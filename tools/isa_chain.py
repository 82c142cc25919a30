#!/usr/bin/env python
"""Compressed memory / wait trace of a kernel from hipcc's device assembly: the order of global loads, stores, LDS operations, DMA, waits,
barriers, matrix instructions and branches - for finding dependent round-trip chains (every `s_waitcnt vmcnt` behind a load that sits inside a
loop or a row of them in a prologue is one L2 / HBM latency, ~1 us under load on MI355X).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S mmearth-train_amd/csrc/capi_rs.hip -o /tmp/capi_rs.s
    python tools/isa_chain.py /tmp/capi_rs.s 'rsc_wide_kernelILi40ELi0'
"""
import re
import sys


def classify(ins, ops):
    if ins.startswith("global_load_lds") or (ins.startswith("buffer_load") and " lds" in ops):
        return "DMA"
    if ins.startswith(("global_load", "buffer_load", "flat_load", "scratch_load")):
        return "GL" if not ins.startswith("scratch") else "SCR_L"
    if ins.startswith(("global_store", "buffer_store", "flat_store")):
        return "GS"
    if ins.startswith("scratch_store"):
        return "SCR_S"
    if ins.startswith(("global_atomic", "buffer_atomic", "flat_atomic")):
        return "GA"
    if ins.startswith(("ds_read", "ds_load")):
        return "LR"
    if ins.startswith(("ds_write", "ds_store")):
        return "LW"
    if ins.startswith("ds_bpermute") or ins.startswith("ds_permute") or ins.startswith("ds_swizzle"):
        return "PERM"
    if ins.startswith("ds_"):
        return "LA"
    if ins.startswith("v_mfma") or ins.startswith("v_smfma"):
        return "MFMA"
    if ins.startswith("s_waitcnt"):
        m = re.search(r"vmcnt\((\d+)\)", ops)
        l = re.search(r"lgkmcnt\((\d+)\)", ops)
        return "W(" + ",".join(([f"vm{m.group(1)}"] if m else []) + ([f"lgkm{l.group(1)}"] if l else [])) + ")"
    if ins.startswith("s_barrier"):
        return "BAR"
    if ins.startswith(("s_cbranch", "s_branch")):
        return "BR->" + ops.strip()
    if ins.startswith("s_load") or ins.startswith("s_buffer_load"):
        return "SL"
    if ins.startswith("v_"):
        return "V"
    return None


def main():
    path, pat = sys.argv[1], sys.argv[2]
    show_valu = len(sys.argv) > 3
    cur, out, nv = None, [], 0
    for line in open(path):
        m = re.match(r"^(_Z\S+):", line)
        if m:
            if cur and out:
                flush(cur, out)
            cur = m.group(1) if re.search(pat, m.group(1)) else None
            out = []
            continue
        if cur is None:
            continue
        lm = re.match(r"^(\.LBB\S+):", line)
        if lm:
            out.append("[" + lm.group(1) + "]")
            continue
        t = line.strip()
        if not t or t.startswith((";", ".")):
            continue
        parts = t.split(None, 1)
        c = classify(parts[0], parts[1] if len(parts) > 1 else "")
        if c is None or (c == "V" and not show_valu) or c == "W(lgkm0)" and False:
            continue
        out.append(c)
    if cur and out:
        flush(cur, out)


def flush(name, seq):
    print("==", name)
    res, i = [], 0
    while i < len(seq):
        j = i
        while j < len(seq) and seq[j] == seq[i]:
            j += 1
        res.append(seq[i] + (f"x{j - i}" if j - i > 1 else ""))
        i = j
    line = ""
    for r in res:
        if len(line) + len(r) > 150:
            print("  " + line)
            line = ""
        line += r + " "
    print("  " + line)


if __name__ == "__main__":
    main()
