"""Ordered kernel list of ONE step (hp_fetch to hp_fetch) from a rocprofv3 --kernel-trace CSV, every queue side by side:
   python tools/lane_dump.py <kernel_trace.csv> [step_index_from_end] [t_from_us] [t_to_us]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
t_from = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
t_to = float(sys.argv[4]) if len(sys.argv) > 4 else 1e12
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"]) for r in rows))
marks = [i for i, e in enumerate(ev) if e[3].startswith("hp_fetch_kernel")]
lo, hi = marks[-back - 1], marks[-back]
step = ev[lo:hi]
t0 = step[0][0]
qs = [q for q, _ in collections.Counter(e[2] for e in step).most_common()]
prev = {}
print(f"step {(ev[hi][0] - t0) / 1e3:.1f} us; queues by kernel count: {qs}")
for e in step:
    t = (e[0] - t0) / 1e3
    if t < t_from or t > t_to:
        prev[e[2]] = e[1]; continue
    gap = (e[0] - prev[e[2]]) / 1e3 if e[2] in prev else 0.0
    print(f"{t:8.1f} q{qs.index(e[2])} gap {gap:6.1f} dur {(e[1] - e[0]) / 1e3:7.1f}  {'    ' * qs.index(e[2])}{e[3][:64]}")
    prev[e[2]] = e[1]
