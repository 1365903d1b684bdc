#!/usr/bin/env python3
"""Per-kernel table from a rocprofv3 --pmc counter_collection.csv (tools/pmc_busy.sh):
clock = GRBM_GUI_ACTIVE / 8 XCDs / duration; mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (cycles x
1024 SIMDs); waits are fractions of wave-cycles (SQ_WAVE_CYCLES counts quad-cycles like them)."""
import csv, re, sys
from collections import defaultdict

rows = defaultdict(lambda: defaultdict(float))
dur = {}
names = {}
for r in csv.DictReader(open(sys.argv[1])):
    d = r["Dispatch_Id"]
    rows[d][r["Counter_Name"]] += float(r["Counter_Value"])
    dur[d] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    n = re.sub(r"\(anonymous namespace\)::|unet_conv::|void ", "", r["Kernel_Name"])
    names[d] = re.sub(r"\((?:[^()]|\([^()]*\))*\)$", "", n)[:62]
agg = defaultdict(lambda: defaultdict(float))
for d, c in rows.items():
    a = agg[names[d]]
    a["n"] += 1
    a["ns"] += dur[d]
    for k, v in c.items():
        a[k] += v
print("rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES "
      "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE over `bench.py --steps 2 --warmup 1` (fp32 default), per kernel name, all launches.")
print("clk = GRBM_GUI_ACTIVE/8 XCDs / duration; mfma_busy = MFMA_BUSY_CYCLES / (cycles x 1024 SIMDs); waits are fractions of wave-cycles.")
for name, a in sorted(agg.items(), key=lambda kv: -kv[1]["ns"]):
    if a["ns"] / a["n"] < 20000 and a["SQ_VALU_MFMA_BUSY_CYCLES"] == 0:
        continue
    cyc = a["GRBM_GUI_ACTIVE"] / 8.0
    clk = cyc / a["ns"] if a["ns"] else 0
    busy = a["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024) if cyc else 0
    wc = a["SQ_WAVE_CYCLES"] or 1
    print(f"{name:62s} n={int(a['n']):3d} avg={a['ns'] / a['n'] / 1e3:7.1f}us clk={clk:.2f}GHz mfma_busy={busy:.2f} "
          f"wait_any={a['SQ_WAIT_ANY'] / wc:.2f} wait_inst={a['SQ_WAIT_INST_ANY'] / wc:.2f} "
          f"lds_act={a['SQ_LDS_IDX_ACTIVE'] / wc:.2f} conflict/act={a['SQ_LDS_BANK_CONFLICT'] / max(a['SQ_LDS_IDX_ACTIVE'], 1):.2f}")
