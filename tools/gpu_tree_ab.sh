export TMPDIR=/tmp
out=gpurun_out/r05d; mkdir -p $out
timeout 600 python -m pytest tests/test_generic_gpu.py tests/test_hip_golden.py -m gpu -q -k "tree or full_body or generic" 2>&1 | grep -E "^FAILED|passed|failed" | head -20
for n in 4096 16384; do timeout 300 python bench.py --robot full_body --envs-per-gpu $n --no-cpu-baseline --train-iters 0 2>/dev/null | tail -1 > $out/bench_full_body_rough$n.json; python -c "
import json; j=json.load(open('$out/bench_full_body_rough$n.json')); print($n, round(j['value']/1e6,2), 'M', round(j['roofline']['kernel_ms']*1e3,1), 'us', j['config']['layout']['kernel'])"; done
FB="python bench.py --robot full_body --envs-per-gpu 4096 --steps 200 --warmup 20 --no-cpu-baseline --train-iters 0"
d=$out/fb_pmc_lds
(cd /tmp && rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OLDPWD/$d -o b -- bash -c "cd $OLDPWD && $FB" > $OLDPWD/$d.log 2>&1)
python - <<'P'
import csv,glob,collections
for f in glob.glob('gpurun_out/r05d/fb_pmc_lds/**/*counter_collection.csv', recursive=True):
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'][:40]; acc[k][r['Counter_Name']]+=float(r['Counter_Value'])
    for k,v in acc.items():
        if 'tree' in k: print(k, {c: round(x/220) for c,x in v.items()}, 'conflict frac', v['SQ_LDS_BANK_CONFLICT']/max(v['SQ_LDS_IDX_ACTIVE'],1))
P
find $out/fb_pmc_lds -name "*kernel_trace.csv" -delete
