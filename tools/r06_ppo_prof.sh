# per-kernel time of PPO.update() (GR1T1 train shape, captured minibatch step) -> gpurun_out/r06/ppo_stats.txt
export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"; out=gpurun_out/r06/ppo_prof; mkdir -p $out
ONLY_GRAPH=1 UPDATES=4 python tools/gpu_ppo_time.py 2>&1 | tail -2
(cd /tmp && ONLY_GRAPH=1 UPDATES=3 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$out -o p -- bash -c "cd $OLDPWD && python tools/gpu_ppo_time.py" > $OLDPWD/$out.log 2>&1)
python - <<'P'
import csv,glob
f=glob.glob('gpurun_out/r06/ppo_prof/**/*kernel_stats.csv', recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel ms', tot/1e6)
for r in sorted(rows, key=lambda r:-float(r['TotalDurationNs']))[:32]:
    print(f"{r['Name'][:100]:100s} {int(r['Calls']):6d} {float(r['TotalDurationNs'])/1e6:9.2f} ms {float(r['AverageNs'])/1e3:8.1f} us {100*float(r['TotalDurationNs'])/tot:5.1f}%")
P
find $out -name "*kernel_trace.csv" -delete
