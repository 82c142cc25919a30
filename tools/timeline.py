"""Per-queue timeline of ONE step from a rocprofv3 --kernel-trace CSV: busy time, gaps, largest gaps.

usage: python tools/timeline.py <kernel_trace.csv> [step_index_from_end]
A step is delimited by consecutive hp_fetch_kernel launches (first kernel of every step).
"""
import csv, sys, collections

rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"]) for r in rows))
marks = [i for i, e in enumerate(ev) if e[3].startswith("hp_fetch_kernel")]
if len(marks) < back + 1:
    sys.exit("not enough steps in the trace")
lo, hi = marks[-back - 1], marks[-back]
step = ev[lo:hi]
t0, t1 = step[0][0], ev[hi][0]
print(f"step wall (hp_fetch to hp_fetch): {(t1 - t0) / 1e3:.1f} us, {len(step)} kernels")
byq = collections.defaultdict(list)
for e in step:
    byq[e[2]].append(e)
for q, lst in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    busy = sum(e[1] - e[0] for e in lst)
    gaps = [(lst[i + 1][0] - lst[i][1], lst[i][3][:50], lst[i + 1][3][:50]) for i in range(len(lst) - 1)]
    pos = [g for g in gaps if g[0] > 0]
    print(f"queue {q}: {len(lst)} kernels, busy {busy / 1e3:.1f} us, span {(lst[-1][1] - lst[0][0]) / 1e3:.1f} us, "
          f"gaps {sum(g[0] for g in pos) / 1e3:.1f} us (median {sorted(g[0] for g in pos)[len(pos) // 2] / 1e3 if pos else 0:.2f} us)")
    for g in sorted(gaps, key=lambda g: -g[0])[:8]:
        print(f"    gap {g[0] / 1e3:7.1f} us  after {g[1]}  before {g[2]}")
# overlap: time with >= 1 kernel running on every queue vs any queue
pts = []
for e in step:
    pts.append((e[0], 1)); pts.append((e[1], -1))
pts.sort()
act, last, any_busy, multi = 0, pts[0][0], 0, 0
for t, d in pts:
    if act >= 1: any_busy += t - last
    if act >= 2: multi += t - last
    act += d; last = t
print(f"GPU busy (>= 1 kernel) {any_busy / 1e3:.1f} us, >= 2 kernels {multi / 1e3:.1f} us, idle {((t1 - t0) - any_busy) / 1e3:.1f} us")
