#!/usr/bin/env python
"""MFMA utilisation per kernel and per kernel FAMILY from one rocprofv3 --pmc pass
(SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE) over an eager bench run.

  mfma_util      = sum SQ_VALU_MFMA_BUSY_CYCLES / (kernel duration x 2.4 GHz x 1024 SIMDs)   (rocprofv3's derived MfmaUtil divides by GRBM_GUI_ACTIVE,
                   which under counter collection is ~20x the kernel's own time on this stack - a 7 us fill reads 307 k cycles -, so the
                   dispatch's Start / End timestamps of the same CSV are used, at the peak clock: a LOWER bound when the chip clocks below 2.4 GHz)
  mfma_flop_frac = SQ_INSTS_VALU_MFMA_MOPS_BF16 x 512 / (kernel duration x 2.5 PFLOP/s)      (issued bf16 matrix flops over the dense bf16 peak)

usage: python tools/mfma_util.py <counter_collection.csv> <out.json> [bench args recorded in meta]"""
import collections
import csv
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from families import FAMILIES  # noqa: E402

SIMDS, FLOP_PER_CLK_SIMD = 1024, 1017.0


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    disp = collections.defaultdict(lambda: dict(name=None, c=collections.defaultdict(list)))
    for r in rows:
        d = disp[r["Dispatch_Id"]]
        d["name"] = r["Kernel_Name"]
        d["ns"] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        d["c"][r["Counter_Name"]].append(float(r["Counter_Value"]))
    per = collections.defaultdict(lambda: dict(n=0, busy=0.0, mops=0.0, gui=0.0, sqbusy=0.0))
    CLK = 2.4e9
    for d in disp.values():
        c = d["c"]
        k = per[d["name"]]
        k["n"] += 1
        k["busy"] += sum(c.get("SQ_VALU_MFMA_BUSY_CYCLES", [0.0]))
        k["mops"] += sum(c.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", [0.0]))
        k["sqbusy"] += sum(c.get("SQ_BUSY_CYCLES", [0.0]))
        k["gui"] += d["ns"] * 1e-9 * CLK            # "cycles" of the kernel's own duration at the peak clock
    kern, fam = {}, collections.defaultdict(lambda: dict(busy=0.0, mops=0.0, gui=0.0, n=0))
    for name, k in per.items():
        if k["gui"] <= 0:
            continue
        kern[name[:90]] = dict(launches_sampled=k["n"], us_per_launch=round(k["gui"] / k["n"] / CLK * 1e6, 2),
                               mfma_util=round(k["busy"] / (k["gui"] * SIMDS), 4),
                               mfma_flop_frac=round(k["mops"] * 512 / (k["gui"] * SIMDS * FLOP_PER_CLK_SIMD), 4),
                               mfma_gflop_per_launch=round(k["mops"] * 512 / k["n"] / 1e9, 3))
        f = next((ff for ff, rx in FAMILIES.items() if re.search(rx, name)), "other")
        for q in ("busy", "mops", "gui", "n"):
            fam[f][q] += k[q]
    fams = {f: dict(mfma_util=round(v["busy"] / (v["gui"] * SIMDS), 4), mfma_flop_frac=round(v["mops"] * 512 / (v["gui"] * SIMDS * FLOP_PER_CLK_SIMD), 4),
                    kernel_us_sampled=round(v["gui"] / CLK * 1e6, 1), launches_sampled=v["n"]) for f, v in fam.items() if v["gui"] > 0}
    out = dict(note="MFMA utilisation (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE, eager driver: "
                    "one kernel at a time, stand-alone utilisation): mfma_util = busy cycles / (kernel duration x 2.4 GHz x 1024 SIMDs); mfma_flop_frac = "
                    "issued bf16 MFMA flops / (kernel duration x 2.5 PFLOP/s)",
               meta=dict(commit=os.environ.get("MPMAE_COMMIT", "n/a"), bench_args=" ".join(sys.argv[3:])), families=fams, kernels=kern)
    json.dump(out, open(sys.argv[2], "w"), indent=1)
    print(f"{'family':18s} {'mfma_util':>10s} {'flop_frac':>10s}")
    for f, v in sorted(fams.items(), key=lambda kv: -kv[1]["kernel_us_sampled"]):
        print(f"{f:18s} {v['mfma_util']:10.4f} {v['mfma_flop_frac']:10.4f}")
    print()
    for name, v in sorted(kern.items(), key=lambda kv: -kv[1]["us_per_launch"] * kv[1]["launches_sampled"])[:45]:
        print(f"{name[:70]:70s} {v['us_per_launch']:7.1f} us  util {v['mfma_util']:7.4f}  flop_frac {v['mfma_flop_frac']:7.4f}  {v['mfma_gflop_per_launch']:8.2f} GF/launch")


if __name__ == "__main__":
    main()
