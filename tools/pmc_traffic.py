"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs).

usage: pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json>
FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE counts 128-B requests at 64 B
(MI355X_MICROARCH.md, HBM section), so it is doubled. Values are averages per launch.
"""
import csv, sys, json, collections, subprocess, os


def per_kernel(path, counter):
    tot, cnt = collections.Counter(), collections.Counter()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        tot[r["Kernel_Name"]] += float(r["Counter_Value"])
        cnt[r["Kernel_Name"]] += 1
    return {k: (tot[k] / cnt[k], cnt[k]) for k in tot}


fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
write = per_kernel(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(set(fetch) | set(write)):
    f, nf = fetch.get(k, (0.0, 0))
    w, nw = write.get(k, (0.0, 0))
    out[k] = dict(fetch_bytes=round(f * 1024 * 2), write_bytes=round(w * 1024), launches_sampled=max(nf, nw),
                  traffic_bytes=round(f * 1024 * 2 + w * 1024))
try:
    commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True,
                            cwd=os.path.dirname(os.path.abspath(__file__))).stdout.strip() or os.environ.get("MPMAE_COMMIT", "n/a")
except Exception:
    commit = os.environ.get("MPMAE_COMMIT", "n/a")
json.dump(dict(note="avg per launch; fetch doubled per the gfx950 FETCH_SIZE correction; separate --pmc passes",
               meta=dict(commit=commit, bench_args=os.environ.get("BENCH_ARGS", "")), kernels=out), open(sys.argv[3], "w"), indent=1)
for k, d in sorted(out.items(), key=lambda kv: -kv[1]["traffic_bytes"] * kv[1]["launches_sampled"])[:30]:
    print(f"{k[:70]:70s} {d['launches_sampled']:5d} launches  fetch {d['fetch_bytes'] / 1e6:8.1f} MB  write {d['write_bytes'] / 1e6:8.1f} MB")
