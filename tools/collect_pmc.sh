#!/bin/bash
# usage (on the GPU box): tools/collect_pmc.sh <tag> <workload name> <bench.py arguments ...>
# rocprofv3 counter passes of ONE bench.py workload, each in a run of its own (--pmc with --kernel-trace only, MI355X_MICROARCH.md):
#   gpurun_out/<tag>/pmc/<workload>/{fetch,write,sq1,sq2,sq3,lds}/ ...counter_collection.csv
# tools/summarise_pmc.py <tag> <workload> turns them into profiles/<tag>_pmc_{hbm,sq}_<workload>.json (what bench.py's roofline object reads).
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
tag=$1; wl=$2; shift 2
out=gpurun_out/$tag/pmc/$wl
mkdir -p $out
CMD="python bench.py $* --no-cpu-baseline --train-iters 0"
run() {   # name, counters
    local d=$out/$1; shift
    (cd /tmp && timeout 600 rocprofv3 --pmc $* --kernel-trace --output-format csv -d $OLDPWD/$d -o b -- bash -c "cd $OLDPWD && $CMD" > $OLDPWD/$d.log 2>&1)
    find $d -name "*kernel_trace.csv" -delete
}
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS
run sq2 SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES
run sq3 SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_IFETCH
echo "$CMD" > $out/command.txt
ls $out
