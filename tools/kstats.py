"""Summarise a rocprofv3 kernel_stats.csv (per step)."""
import csv, sys
path, steps = sys.argv[1], int(sys.argv[2])
rows = list(csv.DictReader(open(path)))
tot = sum(float(r["TotalDurationNs"]) for r in rows) / steps / 1e3
print(f"total kernel time per step: {tot:.1f} us over {sum(int(r['Calls']) for r in rows) // steps} launches")
for r in rows[: int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
    name = r["Name"][:72]
    print(f"{name:72s} {int(r['Calls']) // steps:4d}/step {float(r['TotalDurationNs']) / steps / 1e3:9.1f} us/step  avg {float(r['AverageNs']) / 1e3:7.1f} us  {r['Percentage']:>6s}%")
