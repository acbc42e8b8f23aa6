run() { GRX_TRAIN_SET="$2" timeout 900 python tools/train_seeds.py ${3:-600} 1 4096 trimesh GR1T1_full_body >/dev/null 2>&1; python -c "
import json,glob; f=sorted(glob.glob('gpurun_out/learning_curve_full_body_trimesh_4096*.json'), key=__import__('os').path.getmtime)[-1]; j=json.load(open(f)); r=j['runs'][0]; print('$1', r['reward_every_50'], r['episode_length_every_50'], r['noise_std_every_100'])"; }
run registered_task_1500 "" 1500
export GRX_TRAIN_INIT_NOISE=0.2,0.2,0.2,0.2,0.2,0.2,0.2,0.2,0.2,0.2,0.2,0.2,0.02,0.02,0.02,0.02,0.02,0.02,0.02,0.02,0.02,0.02,0.02,0.02,0.02,0.02,0.02,0.02,0.02,0.02,0.02,0.02
run upper_noise_0.02_1500 "" 1500
