import sys; sys.path.insert(0,'.')
import torch
from tests.helpers import *
cfg = make_cfg()
hip, ora = make_sims(cfg, 256)
hip.reset_all(); ora.reset_all()
gen = torch.Generator().manual_seed(0)
for s in range(30):
    if s>0: sync_state(hip, ora)
    # verify sync
    if s in (10,):
        for n in STATE_TENSORS:
            print("sync", n, tensor_diff(hip.tensor(n), ora.tensor(n))[0])
    a = random_actions(cfg, 256, gen, 0.5)
    ora.step(a, 5.0, s+1); hip.step(a.cuda(), 5.0, s+1); torch.cuda.synchronize()
    d = (hip.tensor("DOF_VEL").cpu()-ora.tensor("DOF_VEL")).abs()
    e = int(d.max(dim=1)[0].argmax())
    print(s, "max dqd %.3e env %d"%(d.max(), e), "Fz", ora.tensor("FEET_CONTACT_FORCE")[e,:,2].tolist(), "hipFz", hip.tensor("FEET_CONTACT_FORCE")[e,:,2].cpu().tolist(), "reset", int(ora.tensor("RESET")[e]), "anch", ora.tensor("ANCHORS")[e,:,2].tolist(), "dpos %.2e"%(hip.tensor("DOF_POS").cpu()-ora.tensor("DOF_POS")).abs().max(), "frac>1e-3 %.3f"%(d>1e-3).float().mean())
