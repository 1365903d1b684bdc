#!/usr/bin/env python3
"""ms per step and launches per step of every kernel in a rocprofv3 kernel_stats.csv of a
bench.py run (steps = launches of sgd_nesterov_kernel).  usage: tools/stats_ms.py FILE [rows]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = [int(r["Calls"]) for r in rows if "sgd_nesterov" in r["Name"]][0]
out, tot = [], 0.0
for r in rows:
    n, t = int(r["Calls"]), int(r["TotalDurationNs"]) / 1e6 / steps
    name = re.sub(r"\(anonymous namespace\)::|unet_conv::|void ", "", r["Name"])
    name = re.sub(r"_ZN(9unet_conv)?12_GLOBAL__N_1\d+", "", name)
    name = re.sub(r"\((unet_conv|\(anon|Igemm|Wgrad|Wino|float|__bf16|unsigned|int|const|Reduce|HIP_).*", "", name)
    out.append((t, n / steps, name))
    tot += t
print(f"steps {steps}  kernel time {tot:.3f} ms/step  launches {sum(o[1] for o in out):.1f}/step")
for t, n, name in sorted(out, reverse=True)[: int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    print(f"{t:7.3f} ms {n:5.1f} x {t / n * 1e3 if n else 0:7.1f} us  {name[:96]}")
