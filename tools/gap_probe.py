"""Main-lane gaps of one timed step against what the side lane was doing: for every gap > T us on the main queue, the side-queue kernels
that overlap it and whether the delayed kernel starts right when one of them ends.  usage: gap_probe.py <kernel_trace.csv> [T]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
T = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"], r.get("VGPR_Count", ""), r.get("LDS_Block_Size", ""),
              r.get("Workgroup_Size_X", ""), r.get("Grid_Size_X", "")) for r in rows))
marks = [i for i, e in enumerate(ev) if e[3].startswith("hp_fetch_kernel")]
spans = [(a, b) for a, b in zip(marks[:-1], marks[1:])]
common = collections.Counter(b - a for a, b in spans).most_common(1)[0][0]
a, b = [s for s in spans if s[1] - s[0] == common][len(spans) // 2]
step = ev[a:b]
t0 = step[0][0]
byq = collections.defaultdict(list)
for e in step:
    byq[e[2]].append(e)
mq = max(byq, key=lambda q: len(byq[q]))
main, side = byq[mq], [e for q, l in byq.items() if q != mq for e in l]
for i in range(len(main) - 1):
    g0, g1 = main[i][1], main[i + 1][0]
    if (g1 - g0) / 1e3 < T:
        continue
    nxt = main[i + 1]
    print(f"gap {(g1 - g0) / 1e3:6.1f} us at t={(g0 - t0) / 1e3:7.1f}: after {main[i][3][:34]} -> {nxt[3][:40]} (vgpr {nxt[4]} lds {nxt[5]} wg {nxt[6]} grid {nxt[7]})")
    for s in side:
        if s[0] < g1 and s[1] > g0:
            tag = "  <- ends where the gap ends" if abs(s[1] - g1) < 3000 else ""
            print(f"      side {s[3][:44]:44s} {(s[0] - t0) / 1e3:7.1f}..{(s[1] - t0) / 1e3:7.1f} vgpr {s[4]} lds {s[5]} wg {s[6]} grid {s[7]}{tag}")
