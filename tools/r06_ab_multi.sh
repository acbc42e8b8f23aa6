# A/B of several variant libraries against the product library at 4096 and 8192 envs, rough terrain: kernel us by HIP events, alternating runs
export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
one() { python bench.py --no-cpu-baseline --train-iters 0 $* 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(round(j['value']/1e6,2), 'M', round(j['roofline']['kernel_ms']*1e3,2), 'us', end='  ')"; }
for args in "--steps 8000 --warmup 800" "--envs-per-gpu 8192 --steps 4000 --warmup 400"; do
    echo "== $args"
    for rep in 1 2; do echo -n "product: "; one $args; for v in "$@"; do echo -n " | $v: "; GRX_HIP_LIB=wiki-grx-gym_amd/csrc/variants/libgrx_$v.so one $args; done; echo; done
done
