#!/bin/bash
# usage: mkvar.sh name "extra flags"   -> variants/libgrx_<name>.so
cd /root/repo/wiki-grx-gym_amd/csrc
FP="-fno-hip-fp32-correctly-rounded-divide-sqrt -ffinite-math-only -fno-signed-zeros -fno-trapping-math -fassociative-math -fno-slp-vectorize -DGRX_REG_CONSTS=2 -mllvm -amdgpu-sched-strategy=iterative-ilp"
HF="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-unused-value"
mkdir -p variants /tmp/var_$1
hipcc $HF $FP $2 -c -o /tmp/var_$1/k.o grx_kernels.hip 2>/tmp/var_$1/k.log &
hipcc $HF $FP $2 -c -o /tmp/var_$1/q.o grx_quad.hip 2>/tmp/var_$1/q.log &
hipcc $HF $FP $2 -c -o /tmp/var_$1/t.o grx_tree16.hip 2>/tmp/var_$1/t.log &
wait
[ -f grx_capi.o ] || hipcc $HF -c -o grx_capi.o grx_capi.cpp
hipcc $HF -shared -o variants/libgrx_$1.so /tmp/var_$1/k.o /tmp/var_$1/q.o /tmp/var_$1/t.o grx_capi.o && echo built $1
