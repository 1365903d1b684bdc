#!/bin/bash
# One rocprofv3 PMC pass (matrix-core busy cycles, clock, wait and LDS counters) over a short
# bench.py run; tools/pmc_busy.py turns the CSV into a per-kernel table.
# usage (on the GPU box): tools/pmc_busy.sh NAME [bench args]
set -e
name=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $out/pmcb_$name -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-alt --no-graph --no-cpu-baseline --no-kernel-timer "$@" > $out/pmcb_$name.log 2>&1
f=$(find $out/pmcb_$name -name "*counter_collection.csv" | head -1)
python3 $GRAFT_REPO_ROOT/tools/pmc_busy.py "$f" > $out/${name}_pmc_mfma_busy.txt
rm -rf $out/pmcb_$name
head -30 $out/${name}_pmc_mfma_busy.txt
