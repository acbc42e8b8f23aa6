# learning curves on the final build of round 6: GR1T1 flat (3 seeds x 1500 iterations), the 32-DOF task (1 seed), rough (2 seeds) -- gpurun_out/learning_curve_*.json
export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
what=${1:-all}
if [ $what = all ] || [ $what = flat ]; then timeout 1500 python tools/train_seeds.py 1500 3 4096 plane GR1T1 2>&1 | tail -3; fi
if [ $what = all ] || [ $what = full ]; then timeout 1500 python tools/train_seeds.py 1500 1 4096 heightfield GR1T1_full_body 2>&1 | tail -3; fi
if [ $what = all ] || [ $what = rough ]; then timeout 1500 python tools/train_seeds.py 1500 2 4096 heightfield GR1T1 2>&1 | tail -3; fi
ls -la gpurun_out/learning_curve_*.json
