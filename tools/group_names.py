import sys, re, collections
agg = collections.defaultdict(float)
for line in sys.stdin:
    m = re.match(r"^(\S+)\s+([\d.]+) us", line)
    if not m: continue
    name, us = m.group(1), float(m.group(2))
    if name.startswith("encoder.stages."): key = "stage" + name.split(".")[2]
    elif name.startswith("decoder_dict"): key = "decoder"
    elif name.startswith("encoder.downsample"): key = "downsample"
    elif name.startswith("head"): key = "heads"
    elif name.startswith("loss") or name.startswith("dloss"): key = "loss"
    elif name.startswith("stem"): key = "stem"
    else: key = "misc:" + name
    agg[key] += us
tot = sum(agg.values())
for k, v in sorted(agg.items(), key=lambda kv: -kv[1]): print(f"{k:28s} {v:9.1f} us  {100*v/tot:5.1f}%")
print("total", tot)
