"""Kernel FAMILIES of one step from a rocprofv3 --kernel-trace CSV -> JSON (bench.py reads the committed copy under profiles/rNN/ to
choose and price the dominant family from IN-STEP numbers: both lanes live, second-stage folds charged to their producers).

usage: python tools/families.py <kernel_trace.csv> <out.json> [bench args recorded in meta]
A step = the launches between two hp_fetch_kernel launches (as tools/kstats.py); only steps with the most common launch count."""
import collections
import csv
import json
import os
import re
import sys

# family -> (regex over the kernel symbol, the bench "kinds" whose ops launch it). Folds: reduce_partials<3> / <1 with strides> behind the
# weight-gradient GEMMs and wgrad_group_fold belong to "wgrad"; reduce_partials<2> / group2 to "dwconv7_wgrad"; reduce_partials<0> (GRN
# column statistics) and <1> (LayerNorm gamma / beta partials) are launched by the fused pointwise entry points: "rs".
FAMILIES = collections.OrderedDict([
    ("rs", r"rsc_wide_kernel|rsc_wide1_kernel|rsp_wide_kernel|rsc_narrow_kernel|rsp_narrow_kernel|rsn3_bwd_kernel|rst_kernel|rs_kernel|reduce_partials_kernel<0>|reduce_partials_kernel<1>"),
    ("wgrad", r"gemm_tn2_kernel|gemm_tn3_kernel|gemm_tng_kernel|gemm_tng48_kernel|gemm_tn_bf16_kernel|wgrad_kernel|wgrad_group_fold_kernel|wg_fold_kernel|reduce_partials_kernel<3>"),
    ("dwconv7", r"dwconv7_mfma_kernel|dwconv7_v6_kernel|dwconv7_v6s1_kernel|dwconv7_v5_kernel"),
    ("dwconv7_wgrad", r"dwconv7_wgrad|reduce_partials_kernel<2>|reduce_partials_group2_kernel"),
    ("ps_fwd", r"ps_fwd_kernel"),
    ("gemm_nt", r"gemm_nt_bf16_kernel|gemm_nt_ring_kernel|gemm_nt3_kernel|gemm_kernel"),
    ("loss", r"loss_"),
    ("ln", r"ln_fwd|ln_bwd"),
    ("stem", r"stem_front_kernel|stem_tail|im2col3_kernel|dwstride2_"),
    ("grn_elementwise", r"grn_|colstats"),
    ("adamw", r"adamw_kernel|hp_fetch_kernel|prep_tiled_kernel"),
])


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
    marks = [i for i, e in enumerate(ev) if e[2].startswith("hp_fetch_kernel")]
    spans = [(marks[i], marks[i + 1]) for i in range(len(marks) - 1)]
    common = collections.Counter(hi - lo for lo, hi in spans).most_common(1)[0][0]
    steps = [(lo, hi) for lo, hi in spans if hi - lo == common]
    n = len(steps)
    fam = {k: dict(us_per_step=0.0, launches_per_step=0.0, kernels={}) for k in list(FAMILIES) + ["other"]}
    total = 0.0
    for lo, hi in steps:
        for s, e, name in ev[lo:hi]:
            k = next((f for f, rx in FAMILIES.items() if re.search(rx, name)), "other")
            d = (e - s) / 1e3 / n
            fam[k]["us_per_step"] += d
            fam[k]["launches_per_step"] += 1.0 / n
            kk = fam[k]["kernels"].setdefault(name[:60], [0.0, 0.0])
            kk[0] += d
            kk[1] += 1.0 / n
            total += d
    wall = sum(ev[hi][0] - ev[lo][0] for lo, hi in steps) / n / 1e3
    for d in fam.values():
        d["us_per_step"] = round(d["us_per_step"], 1)
        d["launches_per_step"] = round(d["launches_per_step"], 2)
        d["share_of_kernel_time"] = round(d["us_per_step"] / total, 4)
        d["kernels"] = {k: [round(v[0], 1), round(v[1], 2)] for k, v in sorted(d["kernels"].items(), key=lambda kv: -kv[1][0])}
    mf = os.environ.get("MPMAE_MFMA_JSON")
    if mf and os.path.exists(mf):      # MFMA utilisation per family (tools/mfma_util.py, a separate --pmc pass of the same workload)
        md = json.load(open(mf)).get("families", {})
        for k, d in fam.items():
            if k in md:
                d["mfma_util"] = md[k]["mfma_util"]
                d["mfma_flop_frac"] = md[k]["mfma_flop_frac"]
    out = dict(note="in-step kernel time per family (rocprofv3 --kernel-trace, both lanes live; folds charged to their producers)",
               meta=dict(commit=os.environ.get("MPMAE_COMMIT", "n/a"), bench_args=" ".join(sys.argv[3:]), steps=n, launches_per_step=common,
                         step_wall_us=round(wall, 1), kernel_time_us=round(total, 1)),
               families=fam)
    json.dump(out, open(sys.argv[2], "w"), indent=1)
    for k, d in sorted(fam.items(), key=lambda kv: -kv[1]["us_per_step"]):
        print(f"{k:18s} {d['us_per_step']:8.1f} us/step  {d['launches_per_step']:6.1f} launches  {100 * d['share_of_kernel_time']:5.1f} %")


if __name__ == "__main__":
    main()
