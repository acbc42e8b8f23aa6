# A/B of variant libraries against the product library at 8192 / 16384 / 32768 envs (the lane-pair kernels), rough terrain: kernel us by HIP events, alternating runs
export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
one() { python bench.py --no-cpu-baseline --train-iters 0 $* 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(round(j['value']/1e6,2), 'M', round(j['roofline']['kernel_ms']*1e3,2), 'us', end='  ')"; }
for args in "--envs-per-gpu 8192 --steps 4000 --warmup 400" "--envs-per-gpu 16384 --steps 3000 --warmup 300" "--envs-per-gpu 32768 --steps 1500 --warmup 150"; do
    echo "== $args"
    for rep in 1 2; do echo -n "product: "; one $args; for v in "$@"; do echo -n " | $v: "; GRX_HIP_LIB=wiki-grx-gym_amd/csrc/variants/libgrx_$v.so one $args; done; echo; done
done
