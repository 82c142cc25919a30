"""Aggregate a rocprofv3 counter_collection.csv by kernel name."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in rows:
    k = r["Kernel_Name"][:60]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[(k, r["Counter_Name"])] += 1
names = sorted({r["Counter_Name"] for r in rows})
print("kernel".ljust(60), " ".join(n[-16:].rjust(16) for n in names))
key = names[0]
for k, d in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", kv[1].get(key, 0)))[:int(sys.argv[2]) if len(sys.argv) > 2 else 25]:
    n = max(1, cnt[(k, key)])
    print(k.ljust(60), " ".join(f"{d.get(nm, 0) / n:16.3g}" for nm in names))
