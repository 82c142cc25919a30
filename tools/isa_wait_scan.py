"""Scan the device assembly of libmpmae_hip for 'load, then s_waitcnt vmcnt(0) within a few instructions' sites - exposed memory round trips -
and rank the kernels of one step by them.  usage: hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S -Iinclude -o capi.s
mmearth-train_amd/csrc/capi.hip; python tools/isa_wait_scan.py capi.s profiles/r02/kernel_time_per_step.txt"""
import re, subprocess, sys
txt = open(sys.argv[1]).read().split("\n")
per = {}
for l in open(sys.argv[2]).read().split("\n")[1:]:
    m = re.match(r"(.*?)\s+([\d.]+)/step\s+([\d.]+) us/step\s+avg\s+([\d.]+) us", l)
    if m:
        per[m.group(1).strip()[:60]] = (float(m.group(2)), float(m.group(3)), float(m.group(4)))
funcs, cur = {}, None
for l in txt:
    m = re.match(r"^(_Z\w+):\s", l)
    if m:
        cur = m.group(1); funcs[cur] = []
    elif cur is not None:
        funcs[cur].append(l)
        if "s_endpgm" in l:
            cur = None
names = list(funcs)
dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.split("\n")
rows = []
for n, d in zip(names, dem):
    body = [x.strip() for x in funcs[n] if x.strip() and not x.strip().startswith((";", ".", "//"))]
    exposed, last_load, inloop = 0, -100, 0
    for i, x in enumerate(body):
        if x.startswith(("global_load", "buffer_load")) and "lds" not in x.split()[0]:
            last_load = i
        if x.startswith("s_waitcnt") and "vmcnt(0)" in x and i - last_load <= 10:
            exposed += 1; last_load = -100
    key = d[:60]
    if key in per:
        rows.append((per[key][1], key, exposed, per[key][0], per[key][2]))
rows.sort(reverse=True)
print(f"{'kernel':60s} {'us/step':>8s} {'avg us':>7s} {'n/step':>6s} exposed load->wait(0) sites")
for us, key, ex, n, avg in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 45]:
    print(f"{key:60s} {us:8.1f} {avg:7.1f} {n:6.1f} {ex}")
