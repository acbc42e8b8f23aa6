# A/B of variant libraries against the product library on the headline workload (4096 envs, rough): kernel us by HIP events, alternating runs
one() { python bench.py --no-cpu-baseline --train-iters 0 --steps 8000 --warmup 800 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(round(j['value']/1e6,2), 'M', round(j['roofline']['kernel_ms']*1e3,2), 'us', end='  ')"; }
for rep in 1 2; do
  echo -n "product: "; one; for v in "$@"; do echo -n " | $v: "; GRX_HIP_LIB=wiki-grx-gym_amd/csrc/variants/libgrx_$v.so one; done; echo
done
