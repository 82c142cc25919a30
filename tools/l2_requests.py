"""Per-kernel L2 (TCC) request volume from one rocprofv3 --pmc pass (TCC_READ_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum), with the kernels'
durations from the same run's kernel trace: reads x 128 B is an upper bound of the bytes the kernel pulled through the L2 -> CU fabric
(a request is at most one 128-byte line; 64-byte requests count as one as well). usage: l2_requests.py <counter csv> <kernel trace csv> <steps run>"""
import csv, sys, collections
cnt = collections.defaultdict(lambda: collections.Counter())
n = collections.Counter()
seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    cnt[k][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (r.get("Dispatch_Id"), k)
    if key not in seen:
        seen.add(key); n[k] += 1
dur = collections.Counter(); nd = collections.Counter()
for r in csv.DictReader(open(sys.argv[2])):
    dur[r["Kernel_Name"]] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    nd[r["Kernel_Name"]] += 1
steps = int(sys.argv[3])
rows = []
for k in cnt:
    rd, rq, hit, miss = (cnt[k][c] for c in ("TCC_READ_sum", "TCC_REQ_sum", "TCC_HIT_sum", "TCC_MISS_sum"))
    per_step = rd * 128 / steps
    us = dur[k] / max(nd[k], 1)
    rows.append((per_step, k, n[k] / steps, rd / max(n[k], 1) * 128, rq / max(n[k], 1) * 128, hit / max(hit + miss, 1), us))
tot = sum(r[0] for r in rows)
print(f"L2 read requests x 128 B per step, all kernels: {tot / 1e9:.2f} GB (under counter collection the kernels run serialised)")
print(f"{'kernel':72s} {'/step':>6s} {'read MB/launch':>15s} {'req MB/launch':>14s} {'hit':>5s} {'us':>7s} {'read TB/s':>9s} {'MB/step':>9s}")
for per_step, k, ps, rdl, rql, hr, us in sorted(rows, reverse=True)[:60]:
    print(f"{k[:72]:72s} {ps:6.1f} {rdl / 1e6:15.1f} {rql / 1e6:14.1f} {hr:5.2f} {us:7.1f} {rdl / 1e6 / max(us, 1e-9):9.2f} {per_step / 1e6:9.1f}")
