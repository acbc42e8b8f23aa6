# PPO update (4096 x 64, GR1T1 train shape, HIP-graph path): the BLAS library behind torch's fp32 GEMMs
export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
for v in default hipblaslt hipblas default hipblaslt; do
  echo -n "$v: "
  if [ $v = default ]; then ONLY_GRAPH=1 UPDATES=5 python tools/gpu_ppo_time.py 2>&1 | tail -1;
  else TORCH_BLAS_PREFER_HIPBLASLT=$([ $v = hipblaslt ] && echo 1 || echo 0) GRX_PPO_BLAS=$v ONLY_GRAPH=1 UPDATES=5 python tools/gpu_ppo_time.py 2>&1 | tail -1; fi
done
