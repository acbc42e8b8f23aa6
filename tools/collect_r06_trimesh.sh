#!/bin/bash
# the trimesh workloads of tools/collect_r06.sh alone (bench lines + rocprofv3 kernel stats; the counter summaries of profiles/ are read as they are)
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
tag=r06; out=gpurun_out/$tag; mkdir -p $out
timeout 300 python bench.py --terrain trimesh --no-cpu-baseline --train-iters 0 2>/dev/null | tail -1 > $out/bench_trimesh.json
timeout 300 python bench.py --terrain trimesh --envs-per-gpu 8192 --steps 4000 --warmup 400 --no-cpu-baseline --train-iters 0 2>/dev/null | tail -1 > $out/bench_trimesh8192.json
stats() {   # name, bench args
    local d=$out/stats_$1; shift
    (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$d -o b -- bash -c "cd $OLDPWD && python bench.py $* --no-cpu-baseline --train-iters 0 > /dev/null" > $OLDPWD/$d.log 2>&1)
    find $d -name "*kernel_trace.csv" -delete
}
stats trimesh4096 --terrain trimesh
stats trimesh8192 --terrain trimesh --envs-per-gpu 8192 --steps 4000 --warmup 400
python -c "
import json
for f in ('bench_trimesh','bench_trimesh8192'):
    j=json.load(open('$out/'+f+'.json')); print(f, round(j['value']/1e6,2), 'M', round(j['roofline']['kernel_ms']*1e3,2), 'us', j['config']['layout']['kernel'])"
