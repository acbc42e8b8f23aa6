"""Time PPO.update() (4096 envs x 64 steps, the GR1T1 train config) eager-device path vs HIP-graph path."""
import sys, os, time; sys.path.insert(0, ".")
import torch
if os.environ.get("GRX_PPO_BLAS"):   # 'hipblaslt' / 'hipblas' (tools/r06_ppo_blas.sh)
    torch.backends.cuda.preferred_blas_library(os.environ["GRX_PPO_BLAS"])
from wiki_grx_gym_amd.rl.modules import ActorCriticMLP
from wiki_grx_gym_amd.rl.ppo import PPO
if os.environ.get("GRX_PPO_FORCE_BUCKET") == "1":   # the multi-rank update on a one-rank RCCL group
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29641", rank=0, world_size=1, device_id=torch.device("cuda", 0))
def run(graph):
    os.environ["GRX_PPO_GRAPH"] = str(graph)
    torch.manual_seed(0)
    ac = ActorCriticMLP(39, 168, 10, actor_hidden_dims=[512, 256, 128], critic_hidden_dims=[512, 256, 128], activation="elu", init_noise_std=0.2)
    alg = PPO(ac, num_learning_epochs=8, num_mini_batches=25, clip_param=0.2, gamma=0.99, lam=0.95, value_loss_coef=1.0, entropy_coef=0.01,
              learning_rate=1e-4, learning_rate_min=1e-5, learning_rate_max=1e-3, max_grad_norm=1.0, use_clipped_value_loss=True,
              schedule="adaptive", desired_kl=0.03, device="cuda:0")
    N, T = 4096, 64
    alg.init_storage(N, T)
    st = alg.storage
    g = torch.Generator(device="cuda:0").manual_seed(1)
    for x in (st.observations, st.pri_observations, st.actions, st.rewards, st.values, st.returns, st.advantages, st.actions_log_prob, st.mu):
        x.copy_(torch.randn(x.shape, device="cuda:0", generator=g) * 0.3)
    st.sigma.fill_(0.2); st.actions_log_prob.fill_(-1.0)
    st.step = T
    ts = []
    for it in range(int(os.environ.get("UPDATES", "4"))):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = alg.update()
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    w = torch.cat([p.detach().flatten() for p in ac.parameters()])
    global LAST
    LAST = {k: v.detach().clone() for k, v in ac.state_dict().items()}
    return ts, out, w, alg.learning_rate
if os.environ.get('ONLY_EAGER'):
    t0, o0, w0, lr0 = run(0)
    print('eager-device update s:', [round(t, 3) for t in t0], o0, lr0)
    sys.exit(0)
if os.environ.get('ONLY_GRAPH'):
    t1, o1, w1, lr1 = run(1)
    print('hip-graph   update s:', [round(t, 3) for t in t1], o1, lr1)
    sys.exit(0)
t0, o0, w0, lr0 = run(0)
L0 = LAST
t1, o1, w1, lr1 = run(1)
print({k: round((L0[k] - LAST[k]).abs().max().item(), 5) for k in L0})
t2, o2, w2, lr2 = run(2)
print("static-eager update s:", [round(t, 3) for t in t2], o2, lr2, "max |dw| vs eager", (w0 - w2).abs().max().item())
print("eager-device update s:", [round(t, 3) for t in t0], o0, lr0)
print("hip-graph   update s:", [round(t, 3) for t in t1], o1, lr1)
print("max |dw| eager vs graph after 4 updates:", (w0 - w1).abs().max().item(), "rel", ((w0 - w1).abs().max() / w0.abs().max()).item())
