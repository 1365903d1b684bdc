#!/bin/bash
# Two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE: they do not fit one pass) over a short bench.py
# run, for tools/pmc_traffic.py.   usage (on the GPU box): tools/pmc_passes.sh NAME [fp32 | bf16]
set -e
name=$1
mode=${2:-fp32}
extra=""
pat="conv_wino_kernel|conv_patch_f32_kernel|conv_patch_up_kernel|conv_patch_s2_kernel|conv_c32_kernel|conv_wino32q_kernel|conv_wino_up32_kernel|conv_igemm_kernel|conv_igemm_rf_kernel|conv_dgrad_s2_kernel|conv_dgrad_s2_patch_kernel"
if [ "$mode" = "bf16" ]; then
  extra="--matmul bf16"
  pat="conv_patch_b16_kernel|conv_igemm_bf16_kernel|conv_dgrad_s2_patch_b16_kernel"
fi
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc_${name}_$c -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-alt --no-graph --no-cpu-baseline --no-kernel-timer $extra > $out/pmc_${name}_$c.log 2>&1
  f=$(find $out/pmc_${name}_$c -name "*counter_collection.csv" | head -1)
  # keep only the rows of the convolution group (the full CSV is large)
  head -1 "$f" > $out/${name}_$c.csv
  grep -E "$pat" "$f" >> $out/${name}_$c.csv
  rm -rf $out/pmc_${name}_$c
done
python3 $GRAFT_REPO_ROOT/tools/pmc_traffic.py $out/${name}_FETCH_SIZE.csv $out/${name}_WRITE_SIZE.csv $out/${name}_conv_hbm_traffic.json $mode
