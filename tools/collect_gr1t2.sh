#!/bin/bash
# usage (on the GPU box): tools/collect_gr1t2.sh -- the GR1T2 robot (BASELINE.json's fourth configuration, a rank's 4096-env shard) through the same
# passes as tools/collect_r05.sh: counters, kernel stats, then the bench line (which reads the counter summary just made).
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
tag=r06; wl=gr1t2_rough4096; out=gpurun_out/$tag; mkdir -p $out
bash tools/collect_pmc.sh $tag $wl --robot gr1t2 --steps 400 --warmup 50
python tools/summarise_pmc.py $tag $wl 10141696
find $out/pmc/$wl -name '*counter_collection.csv' -delete
d=$out/stats_$wl
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$d -o b -- bash -c "cd $OLDPWD && python bench.py --robot gr1t2 --no-cpu-baseline --train-iters 0 > /dev/null" > $OLDPWD/$d.log 2>&1)
find $d -name "*kernel_trace.csv" -delete
timeout 600 python bench.py --robot gr1t2 2> $out/gr1t2.err | tail -1 > $out/bench_$wl.json
cp profiles/${tag}_pmc_*_$wl.json $out/
python -c "
import json; j=json.load(open('$out/bench_$wl.json')); print(j['metric'], round(j['value']/1e6,2), 'M', round(j['roofline']['kernel_ms']*1e3,2), 'us', j['roofline']['traffic'], j['roofline']['valu_issue_frac'], (j.get('full_iteration') or {}).get('env_steps_per_s'), j['cpu_baseline']['value'])"
