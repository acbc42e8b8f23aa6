run() { timeout 900 python tools/train_seeds.py ${2:-1000} ${3:-1} 4096 trimesh GR1T1_full_body >/dev/null 2>&1; python -c "
import json,glob; f=sorted(glob.glob('gpurun_out/learning_curve_full_body_trimesh_4096*.json'), key=__import__('os').path.getmtime)[-1]; j=json.load(open(f))
for r in j['runs']: print('$1 seed', r['seed'], r['reward_every_50'], r['episode_length_every_50'])"; }
export GRX_TRAIN_ACTOR_GAIN=0.01
run actor_gain_0.01 1000 2
