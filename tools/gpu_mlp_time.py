"""Per-layer time of the rollout MLP layers: libgrx_ppo.so's grx_mlp_layer / policy head against torch (addmm + ELU)."""
import sys, time; sys.path.insert(0, ".")
import torch
from wiki_grx_gym_amd.rl.fused_loss import load_ppo_library, _layer, policy_act, mlp_forward
from wiki_grx_gym_amd.rl.modules import ActorCriticMLP
lib = load_ppo_library()
st = torch.cuda.current_stream().cuda_stream
def t(fn, n=50):
    """GPU time per call: n calls captured in one HIP graph, replayed"""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3): fn()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    global st
    with torch.cuda.graph(g, stream=side):
        st = torch.cuda.current_stream().cuda_stream
        for _ in range(n): fn()
    st = torch.cuda.current_stream().cuda_stream
    g.replay(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): g.replay()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / (5 * n) * 1e6
M = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for K, N in ((39, 512), (168, 512), (512, 256), (256, 128), (128, 10), (128, 1)):
    x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.05; b = torch.randn(N, device="cuda")
    a = t(lambda: _layer(lib, x, w, b, N >= 32, torch.cuda.current_stream().cuda_stream))
    r = t(lambda: torch.nn.functional.elu(torch.addmm(b, x, w.t())) if N >= 32 else torch.addmm(b, x, w.t()))
    print(f"M {M} K {K:4d} N {N:4d}: grx {a:7.1f} us   torch {r:7.1f} us")
ac = ActorCriticMLP(39, 168, 10, actor_hidden_dims=[512, 256, 128], critic_hidden_dims=[512, 256, 128], activation="elu", init_noise_std=0.2).cuda()
obs, pri, eps = torch.randn(M, 39, device="cuda"), torch.randn(M, 168, device="cuda"), torch.randn(M, 10, device="cuda")
with torch.no_grad():
    print("actor (3 layers + head/sample/logp) grx:", round(t(lambda: policy_act(ac.actor, ac.std.detach(), obs, eps)), 1), "us; critic grx:", round(t(lambda: mlp_forward(ac.critic, pri)), 1),
          "us; torch actor mean:", round(t(lambda: ac.actor(obs)), 1), "us, critic:", round(t(lambda: ac.critic(pri)), 1), "us")
