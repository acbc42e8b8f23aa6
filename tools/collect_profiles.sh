#!/bin/bash
# usage (on the GPU box, through gpurun): tools/collect_profiles.sh <tag> [skip-tests]
# Regenerates every measurement artefact the docs quote, for the kernel as built in the tree:
#   gpurun_out/<tag>/pytest_gpu.log         python -m pytest tests -m gpu
#   gpurun_out/<tag>/bench_rough.json       python bench.py                       (the headline line)
#   gpurun_out/<tag>/bench_flat.json        python bench.py --terrain flat
#   gpurun_out/<tag>/sweep.jsonl            rough terrain, envs/GPU in {8192, 16384, 32768, 65536, 131072}
#   gpurun_out/<tag>/stats/                 rocprofv3 --kernel-trace --stats of `python bench.py --no-cpu-baseline`
#   gpurun_out/<tag>/pmc_fetch|pmc_write/   rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes
# Copy what is to be judged into profiles/ afterwards (tools/summarise_profiles.py does that).
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
tag=${1:-r01}
out=gpurun_out/$tag
mkdir -p $out
if [ "$2" != "skip-tests" ]; then
    timeout 1500 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1
    tail -3 $out/pytest_gpu.log
fi
# the boxes are shared: a co-tenant's kernels delay this workload's full-CU blocks by 10-80 ms now and then,
# so the headline line is taken three times; bench_rough.json is the FIRST run, bench_rough_runs.jsonl holds all of them
timeout 600 python bench.py 2> $out/bench_rough.err | tail -1 > $out/bench_rough.json
cp $out/bench_rough.json $out/bench_rough_runs.jsonl
for i in 2 3; do timeout 300 python bench.py --no-cpu-baseline 2>> $out/bench_rough.err | tail -1 >> $out/bench_rough_runs.jsonl; done
timeout 300 python bench.py --terrain flat --no-cpu-baseline 2> $out/bench_flat.err | tail -1 > $out/bench_flat.json
# the layouts side by side at the headline size, and the surcharge of GRX_T_RIGID_BODY_STATES (off in the headline)
: > $out/layouts.jsonl
# (lane pair per env, four waves; lane quad, four waves; lane quad, eight waves = the default at this size)
GRX_LANES_PER_ENV=2 timeout 300 python bench.py --no-cpu-baseline --steps 8000 --warmup 800 2>/dev/null | tail -1 >> $out/layouts.jsonl
GRX_LANES_PER_ENV=4 GRX_QUAD_WAVES=4 timeout 300 python bench.py --no-cpu-baseline --steps 8000 --warmup 800 2>/dev/null | tail -1 >> $out/layouts.jsonl
timeout 300 python bench.py --no-cpu-baseline --steps 8000 --warmup 800 2>/dev/null | tail -1 >> $out/layouts.jsonl
GRX_BENCH_RBS=1 timeout 300 python bench.py --no-cpu-baseline --steps 8000 --warmup 800 2>/dev/null | tail -1 >> $out/layouts.jsonl
: > $out/sweep.jsonl
for n in 8192 16384 32768 65536 131072; do
    timeout 300 python bench.py --envs-per-gpu $n --steps $((n <= 32768 ? 4000 : 1500)) --warmup 400 --no-cpu-baseline 2>> $out/sweep.err | tail -1 >> $out/sweep.jsonl
done
BENCH="python bench.py --steps 1000 --warmup 100 --no-cpu-baseline"
# the stats pass profiles the SAME command as the headline line (default step count), so that its average kernel duration
# and bench.py's HIP-event figure describe the same launches; the kernel trace itself (44k rows) is dropped, the stats kept
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$out/stats -o b -- bash -c "cd $OLDPWD && python bench.py --no-cpu-baseline > $out/bench_under_rocprof.json" > $OLDPWD/$out/stats.log 2>&1)
find $out/stats -name "*kernel_trace.csv" -delete
for c in FETCH_SIZE WRITE_SIZE; do
    d=$out/pmc_$(echo $c | tr 'A-Z' 'a-z' | cut -d_ -f1)
    (cd /tmp && rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OLDPWD/$d -o b -- bash -c "cd $OLDPWD && $BENCH" > $OLDPWD/$d.log 2>&1)
done
# SQ counters (instruction mix, issue / wait cycles): three more passes, own runs, kernel-trace only
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"; do
    i=$((i + 1)); d=$out/pmc_sq$i
    (cd /tmp && rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OLDPWD/$d -o b -- bash -c "cd $OLDPWD && python bench.py --steps 300 --warmup 50 --no-cpu-baseline" > $OLDPWD/$d.log 2>&1)
done
# config 5 (full body, generic-tree kernel): bench lines + kernel stats
for n in 4096 16384; do
    timeout 600 python bench.py --robot full_body --envs-per-gpu $n --no-cpu-baseline 2>> $out/full_body.err | tail -1 > $out/bench_full_body_rough$n.json
done
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$out/stats_full_body -o b -- bash -c "cd $OLDPWD && python bench.py --robot full_body --envs-per-gpu 16384 --steps 300 --warmup 30 --no-cpu-baseline > /dev/null" > $OLDPWD/$out/stats_full_body.log 2>&1)
find $out/stats_full_body -name "*kernel_trace.csv" -delete
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$out/stats_full_body4096 -o b -- bash -c "cd $OLDPWD && python bench.py --robot full_body --envs-per-gpu 4096 --steps 600 --warmup 60 --no-cpu-baseline > /dev/null" > $OLDPWD/$out/stats_full_body4096.log 2>&1)
find $out/stats_full_body4096 -name "*kernel_trace.csv" -delete
# config 5 at its per-GPU size: HBM and SQ counter passes of the tree kernel (separate --pmc runs, kernel-trace only)
FB="python bench.py --robot full_body --envs-per-gpu 4096 --steps 200 --warmup 20 --no-cpu-baseline"
for c in FETCH_SIZE WRITE_SIZE; do
    d=$out/fb_pmc_$(echo $c | tr 'A-Z' 'a-z' | cut -d_ -f1)
    (cd /tmp && rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OLDPWD/$d -o b -- bash -c "cd $OLDPWD && $FB" > $OLDPWD/$d.log 2>&1)
done
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_IFETCH"; do
    i=$((i + 1)); d=$out/fb_pmc_sq$i
    (cd /tmp && rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OLDPWD/$d -o b -- bash -c "cd $OLDPWD && $FB" > $OLDPWD/$d.log 2>&1)
done
# config 3's size (8192 envs, lane pairs, eight waves): kernel stats
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$out/stats_8192 -o b -- bash -c "cd $OLDPWD && python bench.py --envs-per-gpu 8192 --steps 4000 --warmup 400 --no-cpu-baseline > $out/bench_rough8192.json" > $OLDPWD/$out/stats_8192.log 2>&1)
find $out/stats_8192 -name "*kernel_trace.csv" -delete
cut -c1-400 $out/bench_rough.json; cut -c1-200 $out/bench_flat.json; cut -c1-200 $out/bench_full_body_rough16384.json
python -c "import json; print('rough runs ms/step:', [round(json.loads(l)['ms_per_step'], 4) for l in open('$out/bench_rough_runs.jsonl')])"
python - <<EOF
import json
for l in open("$out/sweep.jsonl"):
    try:
        j = json.loads(l); print(j["config"]["envs_per_gpu"], round(j["value"] / 1e6, 2), "M env-steps/s", round(j["roofline"]["kernel_ms"] * 1e3, 1), "us/launch")
    except Exception as e:
        print("bad sweep line", e)
EOF
find $out -name "*stats*.csv" | head; find $out -name "*counter_collection.csv" | head
