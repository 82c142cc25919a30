"""Per-step kernel table from a rocprofv3 --kernel-trace CSV: every optimizer step starts with one hp_fetch_kernel launch, so the
launches between two consecutive hp_fetch launches are exactly one step (the bench's warm-up, its eager per-op pass and the input staging
of other passes are left out by keeping only intervals whose launch count equals the most common one).

usage: python tools/kstats.py <kernel_trace.csv> [rows]"""
import csv, sys, collections

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
marks = [i for i, e in enumerate(ev) if e[2].startswith("hp_fetch_kernel")]
spans = [(marks[i], marks[i + 1]) for i in range(len(marks) - 1)]
common = collections.Counter(hi - lo for lo, hi in spans).most_common(1)[0][0]
steps = [(lo, hi) for lo, hi in spans if hi - lo == common]
tot, cnt = collections.Counter(), collections.Counter()
for lo, hi in steps:
    for s, e, n in ev[lo:hi]:
        tot[n] += e - s
        cnt[n] += 1
n = len(steps)
wall = sum(ev[hi][0] - ev[lo][0] for lo, hi in steps) / n / 1e3
allk = sum(tot.values()) / n / 1e3
print(f"{n} steps of {common} launches; step wall (hp_fetch to hp_fetch, under the profiler) {wall:.1f} us; kernel time summed over both lanes {allk:.1f} us")
for name, t in tot.most_common(int(sys.argv[2]) if len(sys.argv) > 2 else 60):
    print(f"{name[:72]:72s} {cnt[name] / n:5.1f}/step {t / n / 1e3:9.1f} us/step  avg {t / cnt[name] / 1e3:7.1f} us  {100 * t / sum(tot.values()):5.2f}%")
