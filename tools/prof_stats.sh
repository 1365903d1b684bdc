#!/bin/bash
# rocprofv3 kernel-trace statistics of one bench.py run (fp32 headline only), summary copied to
# gpurun_out/<name>_kernel_stats.csv.   usage (on the GPU box): tools/prof_stats.sh NAME [bench args]
set -e
name=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$name -o $name -- python3 $GRAFT_REPO_ROOT/bench.py --no-alt --no-graph --no-cpu-baseline --steps 10 --warmup 3 "$@" > $out/${name}_bench.json 2> $out/${name}_bench.err
f=$(find $out/prof_$name -name "*kernel_stats.csv" | head -1)
cp "$f" $out/${name}_kernel_stats.csv
rm -rf $out/prof_$name
tail -1 $out/${name}_bench.json | cut -c1-300
