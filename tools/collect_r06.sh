#!/bin/bash
# Round 6: every BASELINE.json configuration's step kernel profiled: bench line + rocprofv3 kernel stats + PMC passes.
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
tag=r06; out=gpurun_out/$tag; mkdir -p $out
# counters
bash tools/collect_pmc.sh $tag rough4096 --steps 400 --warmup 50
bash tools/collect_pmc.sh $tag rough8192 --envs-per-gpu 8192 --steps 300 --warmup 40
bash tools/collect_pmc.sh $tag rough16384 --envs-per-gpu 16384 --steps 200 --warmup 30
bash tools/collect_pmc.sh $tag rough32768 --envs-per-gpu 32768 --steps 150 --warmup 20
bash tools/collect_pmc.sh $tag full_body_rough4096 --robot full_body --envs-per-gpu 4096 --steps 150 --warmup 20
bash tools/collect_pmc.sh $tag full_body_rough16384 --robot full_body --envs-per-gpu 16384 --steps 80 --warmup 10
bash tools/collect_pmc.sh $tag trimesh4096 --terrain trimesh --steps 400 --warmup 50      # the same raster as the reference's corrected triangle mesh
bash tools/collect_pmc.sh $tag trimesh8192 --terrain trimesh --envs-per-gpu 8192 --steps 300 --warmup 40
for w in rough4096:10141696 rough8192:20283392 rough16384:40566784 rough32768:81133568 full_body_rough4096:16990208 full_body_rough16384:67960832 trimesh4096:10141696 trimesh8192:20283392; do
    python tools/summarise_pmc.py $tag ${w%%:*} ${w##*:} > /dev/null   # profiles/r05_pmc_{hbm,sq}_<workload>.json: what the bench lines below read
done
mkdir -p $out/pmc_json; cp profiles/${tag}_pmc_*.json $out/pmc_json/
find $out/pmc -name '*counter_collection.csv' -delete   # (the raw per-dispatch rows: summarised above; gpurun merges at most 64 MiB back)
# headline: three runs of the driver's default command, flat, the layouts, the sweep
timeout 600 python bench.py 2> $out/bench_rough.err | tail -1 > $out/bench_rough.json
cp $out/bench_rough.json $out/bench_rough_runs.jsonl
for i in 2 3; do timeout 300 python bench.py --no-cpu-baseline --train-iters 0 2>/dev/null | tail -1 >> $out/bench_rough_runs.jsonl; done
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --train-iters 0 2>/dev/null | tail -1 > $out/bench_driver_window.json
timeout 300 python bench.py --terrain flat --no-cpu-baseline --train-iters 0 2>/dev/null | tail -1 > $out/bench_flat.json
GRX_BENCH_RBS=1 timeout 300 python bench.py --no-cpu-baseline --train-iters 0 --steps 8000 --warmup 800 2>/dev/null | tail -1 > $out/bench_rough_every_step.json
: > $out/sweep.jsonl
for n in 8192 16384 32768 65536 131072; do
    timeout 300 python bench.py --envs-per-gpu $n --steps $((n <= 32768 ? 4000 : 1500)) --warmup 400 --no-cpu-baseline --train-iters 0 2>> $out/sweep.err | tail -1 >> $out/sweep.jsonl
done
for n in 4096 16384; do
    timeout 600 python bench.py --robot full_body --envs-per-gpu $n --no-cpu-baseline --train-iters 0 2>> $out/full_body.err | tail -1 > $out/bench_full_body_rough$n.json
done
timeout 300 python bench.py --terrain trimesh --no-cpu-baseline --train-iters 0 2>/dev/null | tail -1 > $out/bench_trimesh.json
timeout 300 python bench.py --terrain trimesh --envs-per-gpu 8192 --steps 4000 --warmup 400 --no-cpu-baseline --train-iters 0 2>/dev/null | tail -1 > $out/bench_trimesh8192.json
# kernel stats (rocprofv3 --kernel-trace --stats) of the same commands as the bench lines
stats() {   # name, bench args
    local d=$out/stats_$1; shift
    (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$d -o b -- bash -c "cd $OLDPWD && python bench.py $* --no-cpu-baseline --train-iters 0 > /dev/null" > $OLDPWD/$d.log 2>&1)
    find $d -name "*kernel_trace.csv" -delete
}
stats rough4096
stats rough8192 --envs-per-gpu 8192 --steps 4000 --warmup 400
stats rough16384 --envs-per-gpu 16384 --steps 2000 --warmup 200
stats rough32768 --envs-per-gpu 32768 --steps 1500 --warmup 150
stats full_body_rough4096 --robot full_body --envs-per-gpu 4096 --steps 600 --warmup 60
stats full_body_rough16384 --robot full_body --envs-per-gpu 16384 --steps 300 --warmup 30
stats trimesh4096 --terrain trimesh
stats trimesh8192 --terrain trimesh --envs-per-gpu 8192 --steps 4000 --warmup 400
python -c "
import json
for f in ('bench_rough','bench_flat','bench_driver_window','bench_rough_every_step','bench_full_body_rough4096','bench_full_body_rough16384','bench_trimesh','bench_trimesh8192'):
    try:
        j=json.load(open('$out/'+f+'.json')); print(f, round(j['value']/1e6,2), 'M', round(j['roofline']['kernel_ms']*1e3,2), 'us', j['config']['layout']['kernel'], (j.get('full_iteration') or {}).get('env_steps_per_s'))
    except Exception as e: print(f, 'bad', e)
for l in open('$out/sweep.jsonl'):
    try:
        j=json.loads(l); print(j['config']['envs_per_gpu'], round(j['value']/1e6,2), 'M', round(j['roofline']['kernel_ms']*1e3,1), 'us', j['config']['layout']['kernel'])
    except Exception as e: print('bad sweep line', e)
"
