# round 6, trimesh = the reference's corrected mesh: (a) the heightfield launches before / after the change (variants/libgrx_pre_trimesh.so = the library
# of the commit before), (b) the trimesh launches before (quarter-cell ramp) / after (planes per triangle half + vertical faces).  Kernel us by HIP events.
export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
V=wiki-grx-gym_amd/csrc/variants/libgrx_pre_trimesh.so
one() { python bench.py --no-cpu-baseline --train-iters 0 $* 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(round(j['value']/1e6,2), 'M', round(j['roofline']['kernel_ms']*1e3,2), 'us', end='  ')"; }
for args in "--steps 8000 --warmup 800" "--envs-per-gpu 8192 --steps 4000 --warmup 400" "--robot full_body --steps 1500 --warmup 150" "--terrain trimesh --steps 8000 --warmup 800" "--terrain trimesh --envs-per-gpu 8192 --steps 4000 --warmup 400" "--terrain trimesh --robot full_body --steps 1500 --warmup 150"; do
    echo "== $args"
    for rep in 1 2; do echo -n "before: "; GRX_HIP_LIB=$V one $args; echo -n " | after: "; one $args; echo; done
done
