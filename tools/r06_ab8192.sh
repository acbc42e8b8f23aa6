# A/B of a variant library against the product library at 8192 / 16384 envs (config 3's launch), rough terrain: kernel us by HIP events, alternating runs
V=wiki-grx-gym_amd/csrc/variants/$1
one() { python bench.py --no-cpu-baseline --train-iters 0 $* 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(round(j['value']/1e6,2), 'M', round(j['roofline']['kernel_ms']*1e3,2), 'us', j['config']['layout']['kernel'], end='  ')"; }
for args in "--envs-per-gpu 8192 --steps 4000 --warmup 400" "--envs-per-gpu 16384 --steps 3000 --warmup 300"; do
    echo "== $args"
    for rep in 1 2; do echo -n "A(product): "; one $args; echo -n " | B($1): "; GRX_HIP_LIB=$V one $args; echo; done
done
