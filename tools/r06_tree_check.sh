# the tree kernels after a change: parity (tree 8 / 16 / generic cases), then the full-body bench lines at 4096 and 16384 envs
export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python -m pytest tests/test_generic_gpu.py tests/test_hip_golden.py -m gpu -q -x -k "tree or full_body or generic" 2>&1 | tail -3
for n in 4096 16384; do for r in 1 2; do python bench.py --robot full_body --envs-per-gpu $n --no-cpu-baseline --train-iters 0 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print($n, round(j['value']/1e6,2), 'M', round(j['roofline']['kernel_ms']*1e3,1), 'us', j['config']['layout']['kernel'])"; done; done
