#!/bin/bash
# usage (through gpurun): tools/bench_repeat.sh <tag> [n]   -- n consecutive headline runs + the batch sweep, for the median
# (the boxes are shared: DESIGN.md section 5)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
tag=${1:-r01}; n=${2:-4}; out=gpurun_out/$tag; mkdir -p $out
: > $out/bench_rough_repeat.jsonl
for i in $(seq $n); do timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 >> $out/bench_rough_repeat.jsonl; done
: > $out/bench_flat_repeat.jsonl
for i in 1 2 3; do timeout 300 python bench.py --terrain flat --no-cpu-baseline 2>/dev/null | tail -1 >> $out/bench_flat_repeat.jsonl; done
: > $out/sweep.jsonl
for e in 8192 16384 32768 65536 131072; do
    for i in 1 2; do timeout 300 python bench.py --envs-per-gpu $e --steps $((e <= 32768 ? 4000 : 1500)) --warmup 400 --no-cpu-baseline 2>/dev/null | tail -1 >> $out/sweep.jsonl; done
done
python - <<PY
import json
for f in ("bench_rough_repeat", "bench_flat_repeat", "sweep"):
    print(f, [(j["config"]["envs_per_gpu"], round(j["value"] / 1e6, 1), round(j["roofline"]["kernel_ms"] * 1e3, 1)) for j in map(json.loads, open("$out/" + f + ".jsonl"))])
PY
