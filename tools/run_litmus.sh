#!/bin/bash
# LDS hand-over litmus (tools/micro/lds_handover.hip) on one MI355X: the product's unfenced flag_set and the fenced variant, >= 1e9 hand-overs
# each.  Usage (through gpurun): bash tools/run_litmus.sh r04   ->  gpurun_out/<tag>_lds_handover_litmus.json (copy to profiles/).
TAG=${1:-r04}; mkdir -p gpurun_out
cd tools/micro
for v in "" "-DGRX_FLAG_FENCED"; do
  out=lds_handover$( [ -n "$v" ] && echo _fenced )
  [ -x $out ] || hipcc --offload-arch=gfx950 -O3 -std=c++17 $v -I../../wiki-grx-gym_amd/csrc -o $out lds_handover.hip || exit 1
done
cd ../..
{ echo '{"runs": ['; tools/micro/lds_handover 300000 1024 4; echo ','; tools/micro/lds_handover_fenced 300000 1024 4; echo ']}'; } > gpurun_out/${TAG}_lds_handover_litmus.json
cat gpurun_out/${TAG}_lds_handover_litmus.json
