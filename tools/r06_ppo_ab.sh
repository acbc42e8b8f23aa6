export TMPDIR=/tmp
for v in 1 0 1 0; do echo -n "GRX_PPO_FUSED_TAIL=$v: "; GRX_PPO_FUSED_TAIL=$v ONLY_GRAPH=1 UPDATES=5 python tools/gpu_ppo_time.py 2>&1 | tail -1; done
