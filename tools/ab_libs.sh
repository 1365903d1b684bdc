#!/bin/bash
# A/B of the train step on ONE box (boxes differ by +-1 %): bench.py with each library named on
# the command line, interleaved twice.  Build the variants beside the product library:
#   make -C unet-implementations_amd/csrc BUILD=build_a OUT=../libunet_a.so [EXTRA=-D...]
# usage (on the GPU box): tools/ab_libs.sh a b hip [-- bench args]
libs=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do libs+=("$1"); shift; done; [ "$1" = "--" ] && shift
for rep in 1 2; do for l in "${libs[@]}"; do echo -n "$l: "
  UNET_HIP_LIB=$PWD/unet-implementations_amd/libunet_$l.so timeout -k 10 300 python bench.py --no-alt --no-cpu-baseline --steps 20 "$@" 2>/dev/null |
    python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('graph', round(d['graph_replay']['value'],1), 'eager', round((d.get('eager') or {'value':d['value']})['value'],1))"
done; done
