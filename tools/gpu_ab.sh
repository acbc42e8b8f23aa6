#!/bin/bash
# A/B of step-kernel builds on one MI355X: tools/gpu_ab.sh "<lib or - for the in-tree one> ..." "<envs> ..." [extra bench.py args]
# Prints one line per (lib, envs): env-steps/s, kernel ms.  Used through gpurun; results land in gpurun_out/ab.jsonl.
LIBS="$1"; SIZES="$2"; shift 2
mkdir -p gpurun_out
for n in $SIZES; do
  steps=$(( 40000000 / n )); [ $steps -gt 4000 ] && steps=4000; [ $steps -lt 200 ] && steps=200
  for lib in $LIBS; do
    if [ "$lib" = "-" ]; then unset GRX_HIP_LIB; else export GRX_HIP_LIB="$PWD/$lib"; fi
    python bench.py --envs-per-gpu $n --steps $steps --warmup $(( steps / 5 )) --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$lib', $n, 'W='+'${GRX_WAVES_PER_BLOCK:-auto}', '%.1f M env-steps/s' % (j['value']/1e6), 'kernel %.1f us' % (j['roofline']['kernel_ms']*1e3), 'wall %.1f us' % (j['ms_per_step']*1e3), flush=True)
        j['lib']='$lib'; open('gpurun_out/ab.jsonl','a').write(json.dumps(j)+'\n')
"
  done
done
