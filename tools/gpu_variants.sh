#!/bin/bash
# usage: tools/gpu_variants.sh "<bench args>" <variant .so> ... : the product library, then each variant (GRX_HIP_LIB), twice round
one() { python bench.py --no-cpu-baseline --train-iters 0 $ARGS 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(round(j['value']/1e6,2), 'M', round(j['roofline']['kernel_ms']*1e3,2), 'us')"; }
ARGS=$1; shift
for rep in 1 2; do
    echo -n "product: "; one
    for v in "$@"; do echo -n "$v: "; GRX_HIP_LIB=wiki-grx-gym_amd/csrc/variants/$v one; done
done
