#!/bin/bash
# One rocprofv3 PMC pass with the instruction-class activity counters over a short bench.py run;
# prints per kernel the fractions of wave-cycles spent issuing each class.
# usage (on the GPU box): tools/pmc_mix.sh NAME
set -e
name=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY --output-format csv -d $out/pmcm_$name -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-alt --no-graph --no-cpu-baseline --no-kernel-timer "$@" > $out/pmcm_$name.log 2>&1
f=$(find $out/pmcm_$name -name "*counter_collection.csv" | head -1)
python3 - "$f" > $out/${name}_pmc_mix.txt <<'PY'
import csv, re, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    k = re.sub(r"\(anonymous namespace\)::|unet_conv::|void ", "", r["Kernel_Name"])
    k = re.sub(r"\((?:[^()]|\([^()]*\))*\)$", "", k)[:64]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
rows = sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))
print("fractions of SQ_WAVE_CYCLES (all launches of a kernel name)")
for k, c in rows[:32]:
    w = c.get("SQ_WAVE_CYCLES", 1) or 1
    print(f"{k:64s} valu={c.get('SQ_ACTIVE_INST_VALU',0)/w:.3f} lds={c.get('SQ_ACTIVE_INST_LDS',0)/w:.3f} vmem={c.get('SQ_ACTIVE_INST_VMEM',0)/w:.3f} misc={c.get('SQ_ACTIVE_INST_MISC',0)/w:.3f} sca={c.get('SQ_ACTIVE_INST_SCA',0)/w:.3f} any={c.get('SQ_ACTIVE_INST_ANY',0)/w:.3f} wait_lds={c.get('SQ_WAIT_INST_LDS',0)/w:.3f}")
PY
rm -rf $out/pmcm_$name
head -24 $out/${name}_pmc_mix.txt
