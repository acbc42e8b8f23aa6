# instruction-fetch counters of the full-body tree kernel (4096 envs): is the unrolled sub-step loop (~70 KB of code) bound by the instruction cache?
export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"; out=gpurun_out/r06/pmc_ifetch; mkdir -p $out
CMD="python bench.py --robot full_body --envs-per-gpu 4096 --steps 100 --warmup 10 --no-cpu-baseline --train-iters 0"
run() { local d=$out/$1; shift; (cd /tmp && timeout 600 rocprofv3 --pmc $* --kernel-trace --output-format csv -d $OLDPWD/$d -o b -- bash -c "cd $OLDPWD && $CMD" > $OLDPWD/$d.log 2>&1); find $d -name "*kernel_trace.csv" -delete; }
run ic1 SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE
run ic2 SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY
run ic3 SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU
run ic4 SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC
python - <<'P'
import csv,glob,collections
for d in ('ic1','ic2','ic3','ic4'):
    for f in glob.glob(f'gpurun_out/r06/pmc_ifetch/{d}/**/*counter_collection.csv', recursive=True):
        acc=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'tree' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
        for c,v in acc.items(): print(c, len(v), sum(v)/len(v))
P
