import sys, time; sys.path.insert(0,'.')
import torch
from tests.helpers import *
def explore(label, cfg, N=64, steps=40, noise=False, delay=5.0, scale=0.3, **kw):
    hip, ora = make_sims(cfg, N, **kw)
    gen = torch.Generator().manual_seed(0)
    hip.reset_all(); ora.reset_all()
    print("==", label)
    for s in range(steps):
        a = random_actions(cfg, N, gen, scale)
        nz = torch.rand(N, 39, generator=gen).contiguous() if noise else None
        ora.step(a, delay, s+1, nz); hip.step(a.cuda(), delay, s+1, nz.cuda() if noise else None)
        torch.cuda.synchronize()
        if s in (0,1,2,4,9,14,19,29,39,59,79):
            worst = {}
            for name in ("DOF_POS","DOF_VEL","ROOT_STATES","FEET_CONTACT_FORCE","REW","OBS","PRI_OBS","FEET_AIR_TIME", "MEASURED_HEIGHTS"):
                worst[name] = tensor_diff(hip.tensor(name), ora.tensor(name))
            ex = {n: int((hip.tensor(n).cpu().to(torch.int64) != ora.tensor(n).to(torch.int64)).sum()) for n in CMP_EXACT}
            print(s+1, " ".join(f"{k}:{v[0]:.2e}/{v[1]:.3f}" for k,v in worst.items()), ex, "contactF", float(ora.tensor("FEET_CONTACT_FORCE")[:,:,2].mean()), "resets", int(ora.tensor("RESET").sum()))
    hip.close()
explore("flat nodr", make_cfg(), steps=40)
explore("flat dr+noise+push", make_cfg(noise=True, dr=True, push=True), steps=40, noise=True)
explore("heightfield", make_cfg(terrain="heightfield"), steps=40)
explore("GR1T2 flat", make_cfg(task="GR1T2"), steps=20)
explore("flat zero-delay big actions", make_cfg(), steps=20, delay=0.0, scale=1.0)
