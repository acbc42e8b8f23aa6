export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
V=wiki-grx-gym_amd/csrc/variants/libgrx_pre_trimesh.so
one() { python bench.py --no-cpu-baseline --train-iters 0 $* 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(round(j['value']/1e6,2), 'M', round(j['roofline']['kernel_ms']*1e3,2), 'us', end='  ')"; }
for args in "--steps 8000 --warmup 800" "--envs-per-gpu 8192 --steps 4000 --warmup 400"; do
    echo "== $args"
    for rep in 1 2 3; do echo -n "before: "; GRX_HIP_LIB=$V one $args; echo -n " | after: "; one $args; echo; done
done
