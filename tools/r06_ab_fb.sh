# A/B of variant libraries against the product library on the full-body workloads (tree kernels): kernel us by HIP events
one() { python bench.py --robot full_body --envs-per-gpu $1 --no-cpu-baseline --train-iters 0 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(round(j['value']/1e6,2), 'M', round(j['roofline']['kernel_ms']*1e3,1), 'us', end='  ')"; }
n=$1; shift
for rep in 1 2; do
  echo -n "$n product: "; one $n; for v in "$@"; do echo -n " | $v: "; GRX_HIP_LIB=wiki-grx-gym_amd/csrc/variants/libgrx_$v.so one $n; done; echo
done
