#!/bin/bash
# A/B short training runs (150 iterations, GR1T1 flat, 4096 envs): which switch changes the learning curve?
cd "${GRAFT_REPO_ROOT:-/root/repo}"
run() { echo "== $1"; env $1 timeout 600 python tools/train_curve.py 150 4096 plane 2>&1 | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read())
print('wall', round(j['wall_s'], 1), 'reward', j['mean_reward'], 'len', j['mean_episode_length'], 'std', j['noise_std'][-1])"; rm -rf gpurun_out/train_plane; }
for v in "$@"; do run "$v"; done
