export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
one() { python bench.py --no-cpu-baseline --train-iters 0 $* 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(round(j['value']/1e6,2), 'M', round(j['roofline']['kernel_ms']*1e3,2), 'us', j['config']['layout']['kernel'])"; }
for args in "--steps 8000 --warmup 800" "--envs-per-gpu 8192 --steps 4000 --warmup 400" "--terrain trimesh --steps 8000 --warmup 800" "--terrain trimesh --envs-per-gpu 8192 --steps 4000 --warmup 400" "--terrain trimesh --robot full_body --steps 1500 --warmup 150"; do echo -n "$args: "; one $args; done
python -m pytest tests/test_terrain_golden.py tests/test_hip_parity.py tests/test_generic_gpu.py -m gpu -q -k "trimesh or vertical_face" 2>&1 | tail -2
