"""Diagnostic for the opt-in HIP-graph PPO update (GRX_PPO_GRAPH=1): eager device path vs captured graph, four updates
with a synthetic rollout (alg.act / process_env_step / compute_returns) between them, plus an fp64 CPU reference of the
clipped gradients of the first minibatch step of the second update.  Finding (round 1): with eager GPU work between the
updates the replayed step's gradient of critic.model.4.bias is wrong (error = its magnitude) while every other tensor
matches the reference to 1e-7 -- hence the graph path is off by default.  MODE=alloc|act|rng|full selects the activity
between updates; GRX_PPO_FUSED_LOSS / GRX_PPO_GRAD_NONE / GRX_PPO_BLAS select the variant (a functional-gradient variant -- autograd.grad
plus copies instead of backward() -- was tried as well) -- all of them showed it with GRX_PPO_LINEAR=torch (torch's own autograd for nn.Linear), so
neither AccumulateGrad nor the BLAS library nor the fused loss is the cause; with the default GRX_PPO_LINEAR=colsum (bias
gradients through libgrx_ppo.so's column sum instead of at::native's reduce kernel) both paths agree bit for bit."""
import sys, os; sys.path.insert(0, ".")
import torch
from wiki_grx_gym_amd.rl.modules import ActorCriticMLP
from wiki_grx_gym_amd.rl.ppo import PPO
N, T, MB, EP = 512, 64, 25, 2
INF = os.environ.get("INFERENCE", "1") == "1"
res = {}
for graph in ("0", "1"):
    os.environ["GRX_PPO_GRAPH"] = graph
    torch.manual_seed(0)
    ac = ActorCriticMLP(39, 168, 10, actor_hidden_dims=[512, 256, 128], critic_hidden_dims=[512, 256, 128], activation="elu", init_noise_std=0.2)
    alg = PPO(ac, num_learning_epochs=EP, num_mini_batches=MB, clip_param=0.2, gamma=0.99, lam=0.95, value_loss_coef=1.0, entropy_coef=0.01,
              learning_rate=1e-4, learning_rate_min=1e-5, learning_rate_max=1e-3, max_grad_norm=1.0, use_clipped_value_loss=True,
              schedule="adaptive", desired_kl=0.03, device="cuda:0")
    alg.init_storage(N, T)
    g = torch.Generator(device="cuda:0").manual_seed(1)
    snaps = []
    for u in range(4):
        torch.manual_seed(50 + u)
        ctx = torch.inference_mode() if INF else torch.no_grad()
        MODE = os.environ.get("MODE", "full")
        if MODE != "full":
            st = alg.storage
            with ctx:
                for t in range(T):
                    o = torch.randn(N, 39, device="cuda:0", generator=g); c = torch.randn(N, 168, device="cuda:0", generator=g)
                    if MODE == "alloc": junk = [torch.randn(1000, 1000, device="cuda:0") for _ in range(4)]; del junk
                    if MODE == "act": ac.act(o); ac.evaluate(c)
                    if MODE == "act_nosample": ac.update_distribution(o); ac.evaluate(c)
                    if MODE == "rng": torch.randn(N, 10, device="cuda:0")
                for nm in ["observations", "pri_observations", "actions", "values", "advantages", "returns", "mu"]:
                    x = getattr(st, nm); x.copy_(torch.randn(x.shape, device="cuda:0", generator=g) * 0.3)
                st.sigma.fill_(0.2); st.actions_log_prob.fill_(-1.0); st.step = T
        with ctx:
            for t in range(T if MODE == "full" else 0):
                o = torch.randn(N, 39, device="cuda:0", generator=g); c = torch.randn(N, 168, device="cuda:0", generator=g)
                a = alg.act(o, c)
                r = torch.randn(N, device="cuda:0", generator=g) * 0.1
                d = (torch.rand(N, device="cuda:0", generator=g) < 0.02)
                alg.process_env_step(r, d, {"time_outs": torch.zeros(N, device="cuda:0", dtype=torch.bool)})
            if MODE == 'full': alg.compute_returns(c)
        # per-minibatch trace: (value loss, surrogate, kl) of the first steps of this update
        trace = []
        if graph == "0":
            inner = alg._losses
            def wrapped(*a, _i=inner, **k):
                if u == 1 and len(trace) == 1:
                    res["after1_0"] = {n: p.detach().clone() for n, p in ac.named_parameters()}
                    res["grad_0"] = {n: p.grad.detach().clone() for n, p in ac.named_parameters()}
                if u == 1 and len(trace) == 0: res["batch"] = [x.detach().clone() for x in a]
                if u == 1 and len(trace) == 0: res["before_0"] = {n: p.detach().clone() for n, p in ac.named_parameters()}
                r = _i(*a, **k)
                if len(trace) < 6: trace.append((r[1].item(), r[0].item(), r[3].item()))
                return r
            alg._losses = wrapped
            out = alg.update()
            alg._losses = inner
        else:
            if alg._graph is None:
                out = alg.update()   # builds the graph; no trace for this one
            else:
                rp = alg._graph.replay
                prev = [None]
                class _G:
                    def replay(self_):
                        if u == 1 and len(trace) == 0: res["before_1"] = {n: p.detach().clone() for n, p in ac.named_parameters()}
                        rp()
                        if u == 1 and len(trace) == 0:
                            res["after1_1"] = {n: p.detach().clone() for n, p in ac.named_parameters()}
                            res["grad_1"] = {n: p.grad.detach().clone() for n, p in ac.named_parameters()}
                            res["static_1"] = [x.detach().clone() for x in alg._static]
                        if len(trace) < 6:
                            cur = alg._sums.tolist()
                            p = prev[0] or [0.0, 0.0, 0.0]
                            trace.append((cur[0] - p[0], cur[1] - p[1], cur[2]))
                            prev[0] = cur
                g_real = alg._graph; alg._graph = _G()
                out = alg.update()
                alg._graph = g_real
        print("  trace", graph, u, [tuple(round(x, 6) for x in t) for t in trace])
        alg.clear_storage()
        snaps.append(torch.cat([p.detach().flatten() for p in ac.parameters()]).clone())
        if u == 0:
            res["opt" + graph] = [{k: v.detach().clone().float().flatten() for k, v in stt.items() if torch.is_tensor(v)} for stt in alg.optimizer.state.values()]
            res["lr" + graph] = alg._lr_t.item()
        print("graph", graph, "update", u, out, alg.learning_rate)
    res[graph] = snaps
for i, (sa, sb) in enumerate(zip(res["opt0"], res["opt1"])):
    for k in sa:
        dmax = (sa[k] - sb[k]).abs().max().item()
        if dmax > 0 or i == 0: print("opt state after update 0: param", i, k, "max diff", dmax, "value", sa[k][:2].tolist(), sb[k][:2].tolist())
print("lr after update 0:", res["lr0"], res["lr1"])
# fp64 CPU reference of that step's clipped gradients
import copy
acd = copy.deepcopy(ac).double().cpu()
with torch.no_grad():
    for (n, p) in acd.named_parameters(): p.copy_(res["before_0"][n].double().cpu())
obs_, cobs_, act_, tv_, adv_, ret_, olp_, omu_, osg_ = [x.double().cpu() for x in res["batch"]]
print("static == eager batch:", [bool((a.cpu() == b.cpu()).all()) for a, b in zip(res["batch"], res["static_1"])])
mu_ = acd.actor(obs_); sg_ = acd.std.expand_as(mu_)
dist_ = torch.distributions.Normal(mu_, sg_)
logp_ = dist_.log_prob(act_).sum(-1); val_ = acd.critic(cobs_)
ratio_ = torch.exp(logp_ - olp_.squeeze()); a_ = adv_.squeeze()
sl_ = torch.max(-a_ * ratio_, -a_ * ratio_.clamp(0.8, 1.2)).mean()
vc_ = tv_ + (val_ - tv_).clamp(-0.2, 0.2)
vl_ = torch.max((val_ - ret_).pow(2), (vc_ - ret_).pow(2)).mean()
loss_ = sl_ + 1.0 * vl_ - 0.01 * dist_.entropy().sum(-1).mean()
loss_.backward()
tn = torch.sqrt(sum((p.grad ** 2).sum() for p in acd.parameters()))
coef = min(1.0, 1.0 / (tn.item() + 1e-6))
for n, p in acd.named_parameters():
    gref = (p.grad * coef).float()
    g0, g1 = res["grad_0"][n].cpu(), res["grad_1"][n].cpu()
    print(f"grad {n:24s} |ref| {gref.abs().max().item():.3e}  eager-ref {(g0 - gref).abs().max().item():.2e}  graph-ref {(g1 - gref).abs().max().item():.2e}")
for n in []:
    b0, b1, a0, a1 = res["before_0"][n], res["before_1"][n], res["after1_0"][n], res["after1_1"][n]
    print(f"step0 of update 1: {n:24s} before diff {(b0-b1).abs().max().item():.2e}  moved eager {(a0-b0).abs().max().item():.3e} graph {(a1-b1).abs().max().item():.3e}  after diff {(a0-a1).abs().max().item():.2e}")
for u in range(4):
    a, b = res["0"][u], res["1"][u]
    print("update", u, "max |eager-graph|", (a - b).abs().max().item(), "n>1e-4:", ((a - b).abs() > 1e-4).sum().item(), "of", a.numel())
