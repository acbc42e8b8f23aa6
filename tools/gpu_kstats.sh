#!/bin/bash
# usage (through gpurun): tools/gpu_kstats.sh <name> <command...>   -- rocprofv3 --kernel-trace --stats of a command run from the
# repo root; prints the top kernels by total time and leaves gpurun_out/kstats_<name>.csv (the stats table only).
export TMPDIR=/tmp
root="${GRAFT_REPO_ROOT:-/root/repo}"
name=$1; shift
d=$root/gpurun_out/kstats_$name
rm -rf $d; mkdir -p $d
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $d -o b -- bash -c "cd $root && $*" > $d/run.log 2>&1)
f=$(find $d -name "*kernel_stats.csv" | head -1)
[ -z "$f" ] && { echo "no stats file"; tail -5 $d/run.log; exit 1; }
cp $f $root/gpurun_out/kstats_$name.csv
find $d -name "*kernel_trace.csv" -delete
python3 - $f <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:24]:
    print("%5.1f%% %7d x %8.1f us  %s" % (100 * float(r["TotalDurationNs"]) / tot, int(r["Calls"]), float(r["AverageNs"]) / 1e3, r["Name"][:100]))
PY
