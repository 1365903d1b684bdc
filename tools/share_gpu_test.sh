#!/bin/bash
# two independent train-step loops sharing one GPU (time-sliced queues): usage share_gpu_test.sh DIR [bench args]
d=$1; shift
cd $d
(timeout -k 10 200 python bench.py --steps 6 --warmup 1 --no-alt --no-cpu-baseline --no-graph "$@" > /tmp/a.json 2> /tmp/a.err; echo A rc=$?) &
(timeout -k 10 200 python bench.py --steps 6 --warmup 1 --no-alt --no-cpu-baseline --no-graph "$@" > /tmp/b.json 2> /tmp/b.err; echo B rc=$?) &
wait
grep -ih "fault\|core dump" /tmp/a.err /tmp/b.err | head -4
cut -c1-110 /tmp/a.json /tmp/b.json
