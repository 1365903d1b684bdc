#!/usr/bin/env python3
"""Turns two rocprofv3 PMC passes over bench.py into profiles/<out>.json: HBM bytes per launch
of the 3x3 forward / data-gradient kernel group.

  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_f -o f -- python3 bench.py ...
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_w -o w -- python3 bench.py ...
  python tools/pmc_traffic.py gpurun_out/pmc_f/f_counter_collection.csv \
         gpurun_out/pmc_w/w_counter_collection.csv profiles/r01_igemm_hbm_traffic.json

FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for
gfx950 (128-B requests of 16-B/lane coalesced reads are tallied at 64 B)."""
import csv
import json
import re
import sys

GROUPS = {
    "fp32": r"conv_wino_kernel|conv_patch_f32_kernel|conv_patch_up_kernel|conv_patch_s2_kernel|conv_c32_kernel|conv_wino32q_kernel|conv_wino_up32_kernel|conv_igemm_kernel|conv_igemm_rf_kernel|conv_dgrad_s2_kernel|conv_dgrad_s2_patch_kernel",
    # BASELINE config 4 (`bench.py --matmul bf16`): the same launch group on bf16 tensors
    "bf16": r"conv_patch_b16_kernel|conv_igemm_bf16_kernel|conv_dgrad_s2_patch_b16_kernel",
}
MODE = sys.argv[4] if len(sys.argv) > 4 else "fp32"
GROUP = re.compile(GROUPS[MODE])


def per_launch(path, counter):
    tot, n = 0.0, 0
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and GROUP.search(r["Kernel_Name"]):
            tot += float(r["Counter_Value"])
            n += 1
    return tot * 1024.0 / max(n, 1), n


fetch, nf = per_launch(sys.argv[1], "FETCH_SIZE")
write, nw = per_launch(sys.argv[2], "WRITE_SIZE")
out = {"kernel": GROUP.pattern.replace("|", " | ") + " (3x3 forward + data gradient)",
       "launches_profiled": nf,
       "fetch_bytes_per_launch_corrected": 2.0 * fetch,
       "write_bytes_per_launch": write,
       "hbm_bytes_per_launch": 2.0 * fetch + write,
       "mode": MODE,
       "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over "
                 "`bench.py --steps 2 --warmup 1 --no-alt --no-graph --no-cpu-baseline --no-kernel-timer"
                 + ("" if MODE == "fp32" else " --matmul " + MODE) + "`; "
                 "FETCH_SIZE (KiB) doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests "
                 "as 64 B for 16-B/lane coalesced reads); WRITE_SIZE (KiB) taken as is"}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
