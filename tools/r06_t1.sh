export TMPDIR=/tmp; out=gpurun_out/r06; mkdir -p $out
timeout 900 python -m pytest tests/test_generic_gpu.py tests/test_hip_golden.py -m gpu -q -x -k "tree or full_body or generic" 2>&1 | tail -15 > $out/t1_tests.txt
python tools/gpu_tree_sections.py > $out/sections_tree16_a.txt 2>&1
for n in 4096 16384; do python bench.py --robot full_body --envs-per-gpu $n --no-cpu-baseline --train-iters 0 2>/dev/null | tail -1 > $out/bench_fb_a_$n.json; done
cat $out/t1_tests.txt; grep -A9 heightfield $out/sections_tree16_a.txt; python - <<'P'
import json
for n in (4096,16384):
    for t in ('base','a'):
        j=json.load(open(f'gpurun_out/r06/bench_fb_{t}_{n}.json')); print(n,t,round(j['value']/1e6,2),'M',round(j['roofline']['kernel_ms']*1e3,1),'us',j['config']['layout']['kernel'])
P
