"""Largest main-queue gaps of several consecutive timed steps (is a stall systematic, and where).  usage: gap_steps.py <kernel_trace.csv> [T us] [steps]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
T = float(sys.argv[2]) if len(sys.argv) > 2 else 25.0
NS = int(sys.argv[3]) if len(sys.argv) > 3 else 6
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"]) for r in rows))
marks = [i for i, e in enumerate(ev) if e[3].startswith("hp_fetch_kernel")]
spans = [(a, b) for a, b in zip(marks[:-1], marks[1:])]
common = collections.Counter(b - a for a, b in spans).most_common(1)[0][0]
good = [s for s in spans if s[1] - s[0] == common]
for a, b in good[len(good) // 2: len(good) // 2 + NS]:
    step = ev[a:b]; t0 = step[0][0]
    byq = collections.defaultdict(list)
    for e in step: byq[e[2]].append(e)
    mq = max(byq, key=lambda q: len(byq[q])); main = byq[mq]
    side = [e for q, l in byq.items() if q != mq for e in l]
    gaps = [((main[i + 1][0] - main[i][1]) / 1e3, i) for i in range(len(main) - 1)]
    tot = sum(g for g, _ in gaps if g > 0)
    print(f"step wall {(ev[b][0] - t0) / 1e3:.0f} us, main busy {sum(e[1] - e[0] for e in main) / 1e3:.0f}, gaps {tot:.0f} us; side busy {sum(e[1] - e[0] for e in side) / 1e3:.0f}, side ends {(max(e[1] for e in side) - t0) / 1e3:.0f}, main ends {(main[-1][1] - t0) / 1e3:.0f}")
    for g, i in sorted(gaps, reverse=True):
        if g < T: break
        idx = i + 1
        print(f"    {g:6.1f} us at {(main[i][1] - t0) / 1e3:7.1f} before main kernel #{idx} {main[idx][3][:50]}")
