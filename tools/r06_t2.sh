# sections + bench of the tree kernel (full body), tag = $1
export TMPDIR=/tmp; out=gpurun_out/r06; mkdir -p $out; tag=${1:-x}
python tools/gpu_tree_sections.py > $out/sections_tree16_$tag.txt 2>&1
for n in 4096 16384; do python bench.py --robot full_body --envs-per-gpu $n --no-cpu-baseline --train-iters 0 2>/dev/null | tail -1 > $out/bench_fb_${tag}_$n.json; done
grep -A9 heightfield $out/sections_tree16_$tag.txt; python - <<P
import json
for n in (4096,16384):
    j=json.load(open(f'gpurun_out/r06/bench_fb_${tag}_{n}.json')); print(n,'$tag',round(j['value']/1e6,2),'M',round(j['roofline']['kernel_ms']*1e3,1),'us',j['config']['layout']['kernel'])
P
