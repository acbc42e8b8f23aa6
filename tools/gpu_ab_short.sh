#!/bin/bash
# usage: tools/gpu_ab_short.sh <variant .so under csrc/variants> : as tools/gpu_ab.sh, the heightfield workloads of the pipelined layouts only
V=wiki-grx-gym_amd/csrc/variants/$1
one() { python bench.py --no-cpu-baseline --train-iters 0 $* 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(round(j['value']/1e6,2), 'M', round(j['roofline']['kernel_ms']*1e3,2), 'us', end='  ')"; }
for args in "--steps 8000 --warmup 800" "--envs-per-gpu 8192 --steps 4000 --warmup 400" "--robot gr1t2 --steps 8000 --warmup 800"; do
    echo "== $args"
    for rep in 1 2; do echo -n "A: "; one $args; echo -n " | B: "; GRX_HIP_LIB=$V one $args; echo; done
done
