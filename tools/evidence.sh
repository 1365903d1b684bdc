set -x
R=$1
python bench.py > gpurun_out/${R}_bench_default.json 2> gpurun_out/${R}_bench_default.err; echo bench rc=$?
python bench.py --clip --no-alt --no-cpu-baseline > gpurun_out/${R}_bench_clip.json 2> gpurun_out/${R}_bench_clip.err; echo clip rc=$?
tools/prof_stats.sh ${R}_fp32 > gpurun_out/${R}_prof1.log 2>&1
tools/prof_stats.sh ${R}_bf16 --matmul bf16 > gpurun_out/${R}_prof2.log 2>&1
tools/prof_stats.sh ${R}_bf16x3 --matmul bf16x3 > gpurun_out/${R}_prof3.log 2>&1
tools/pmc_busy.sh ${R} > gpurun_out/${R}_pmc.log 2>&1
tools/pmc_mix.sh ${R} > gpurun_out/${R}_mix.log 2>&1
tools/pmc_passes.sh ${R} > gpurun_out/${R}_pmcpass.log 2>&1
(for m in fp32 bf16 bf16x3; do python tools/cpu_overhead.py $m 2>/dev/null | tail -1; done) > gpurun_out/${R}_cpu_overhead.txt
timeout -k 5 120 tools/micro/mfma_valu_overlap > gpurun_out/${R}_mfma_valu_overlap.txt 2>&1
timeout -k 5 120 tools/micro/mfma_piece_cost > gpurun_out/${R}_mfma_piece_cost.txt 2>&1
timeout -k 5 120 tools/micro/mfma_valu_overlap_bf16 > gpurun_out/${R}_mfma_valu_overlap_bf16.txt 2>&1
python tools/bench_wino.py 10 > gpurun_out/${R}_wino.txt 2>&1
python tools/bench_wgrad_act.py 10 > gpurun_out/${R}_wgrad.txt 2>&1
echo done
