#!/bin/bash
# usage: tools/pmc_pass.sh <tag> <terrain> <counter> [<counter> ...]
# One rocprofv3 --pmc pass (own run, kernel-trace only) over a short bench.py run; prints the
# per-launch average of every counter for grx_step_kernel and leaves the CSVs under gpurun_out/r01/.
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
tag=$1; terrain=$2; shift 2
d=gpurun_out/r01/pmc_$tag
mkdir -p $d
rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $d -o b -- \
    python bench.py --steps 60 --warmup 10 --no-cpu-baseline --terrain $terrain > $d/run.log 2>&1
tail -1 $d/run.log | cut -c1-160
python3 - $d <<'EOF'
import csv, sys, collections, glob
fs = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
if not fs:
    print("no counter file"); sys.exit(0)
agg = collections.defaultdict(list)
for r in csv.DictReader(open(fs[0])):
    if "grx_step_kernel" in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(k, round(sum(v) / len(v)))
EOF
