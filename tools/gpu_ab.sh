#!/bin/bash
# usage: tools/gpu_ab.sh <variant .so under csrc/variants> : alternating runs of the product library (A) and the variant (B, GRX_HIP_LIB)
V=wiki-grx-gym_amd/csrc/variants/$1
one() { python bench.py --no-cpu-baseline --train-iters 0 $* 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(round(j['value']/1e6,2), 'M', round(j['roofline']['kernel_ms']*1e3,2), 'us', end='  ')"; }
for args in "--steps 8000 --warmup 800" "--envs-per-gpu 8192 --steps 4000 --warmup 400" "--envs-per-gpu 32768 --steps 1500 --warmup 150" "--robot full_body --envs-per-gpu 4096 --steps 1500 --warmup 150" "--terrain flat --steps 8000 --warmup 800"; do
    echo "== $args"
    for rep in 1 2; do echo -n "A: "; one $args; echo -n " | B: "; GRX_HIP_LIB=$V one $args; echo; done
done
