# A/B of a variant library against the product library over the batch sizes of BASELINE.json's configs (rough terrain): kernel us by HIP events
V=wiki-grx-gym_amd/csrc/variants/$1
one() { python bench.py --no-cpu-baseline --train-iters 0 $* 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(round(j['value']/1e6,2), 'M', round(j['roofline']['kernel_ms']*1e3,2), 'us', end='  ')"; }
for args in "--steps 8000 --warmup 800" "--envs-per-gpu 8192 --steps 4000 --warmup 400" "--envs-per-gpu 32768 --steps 1500 --warmup 150"; do
    echo "== $args"
    for rep in 1 2; do echo -n "A(product): "; one $args; echo -n " | B($1): "; GRX_HIP_LIB=$V one $args; echo; done
done
