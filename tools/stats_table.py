#!/usr/bin/env python3
"""Per-step table from a rocprofv3 kernel_stats.csv: tools/stats_table.py file.csv STEPS [other.csv STEPS]"""
import csv, re, sys

def load(path, steps):
    rows = {}
    for r in csv.DictReader(open(path)):
        name = re.sub(r"\(anonymous namespace\)::|unet_conv::|void ", "", r["Name"])
        name = re.sub(r"\((?:[^()]|\([^()]*\))*\)$", "", name)[:70]
        d = rows.setdefault(name, [0, 0.0])
        d[0] += int(r["Calls"]); d[1] += float(r["TotalDurationNs"])
    return {k: (v[0] / steps, v[1] / steps / 1e6) for k, v in rows.items()}

a = load(sys.argv[1], float(sys.argv[2]))
b = load(sys.argv[3], float(sys.argv[4])) if len(sys.argv) > 4 else None
keys = sorted(set(a) | set(b or {}), key=lambda k: -(a.get(k, (0, 0))[1]))
ta = sum(v[1] for v in a.values()); tb = sum(v[1] for v in (b or {}).values())
print(f"{'kernel':70s} {'calls':>6s} {'ms/step':>8s}" + (f" | {'calls':>6s} {'ms/step':>8s} {'delta':>7s}" if b else ""))
for k in keys:
    ca, ma = a.get(k, (0, 0.0))
    line = f"{k:70s} {ca:6.1f} {ma:8.3f}"
    if b is not None:
        cb, mb = b.get(k, (0, 0.0))
        line += f" | {cb:6.1f} {mb:8.3f} {ma - mb:+7.3f}"
    print(line)
print(f"{'TOTAL':70s} {'':6s} {ta:8.3f}" + (f" | {'':6s} {tb:8.3f} {ta - tb:+7.3f}" if b else ""))
