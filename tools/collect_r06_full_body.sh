#!/bin/bash
# the full-body rows of tools/collect_r06.sh again (tree kernels changed after the first collection): counters, bench lines, kernel stats
export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"; tag=r06; out=gpurun_out/$tag; mkdir -p $out
bash tools/collect_pmc.sh $tag full_body_rough4096 --robot full_body --envs-per-gpu 4096 --steps 150 --warmup 20
bash tools/collect_pmc.sh $tag full_body_rough16384 --robot full_body --envs-per-gpu 16384 --steps 80 --warmup 10
for w in full_body_rough4096:16990208 full_body_rough16384:67960832; do python tools/summarise_pmc.py $tag ${w%%:*} ${w##*:} > /dev/null; done
mkdir -p $out/pmc_json; cp profiles/${tag}_pmc_*full_body*.json $out/pmc_json/
for n in 4096 16384; do timeout 600 python bench.py --robot full_body --envs-per-gpu $n --no-cpu-baseline --train-iters 0 2>> $out/full_body.err | tail -1 > $out/bench_full_body_rough$n.json; done
stats() { local d=$out/stats_$1; shift; rm -rf $d; (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$d -o b -- bash -c "cd $OLDPWD && python bench.py $* --no-cpu-baseline --train-iters 0 > /dev/null" > $OLDPWD/$d.log 2>&1); find $d -name "*kernel_trace.csv" -delete; }
stats full_body_rough4096 --robot full_body --envs-per-gpu 4096 --steps 600 --warmup 60
stats full_body_rough16384 --robot full_body --envs-per-gpu 16384 --steps 300 --warmup 30
python - <<'P'
import json,csv
for n in (4096,16384):
    j=json.load(open(f'gpurun_out/r06/bench_full_body_rough{n}.json')); r=j['roofline']
    k=[x for x in csv.DictReader(open(f'gpurun_out/r06/stats_full_body_rough{n}/b_kernel_stats.csv')) if 'grx_step' in x['Name']][0]
    print(n, round(j['value']/1e6,2),'M', round(r['kernel_ms']*1e3,1),'us live', round(float(k['AverageNs'])/1e3,1),'us rocprof', k['Calls'], 'traffic', r['traffic'], 'valu', r['valu_issue_frac'])
P
