import sys; sys.path.insert(0, ".")
import torch
from tests.helpers import make_cfg, make_sims, random_actions, sync_state
from tests.test_hip_parity import physics_lockstep
cfg = make_cfg("GR1T1Full", noise=False, dr=False)
hip, ora = make_sims(cfg, 64)
hip.reset_all(); ora.reset_all()
gen = torch.Generator().manual_seed(0)
for s in range(4):
    if s > 0: sync_state(hip, ora)
    a = random_actions(cfg, 64, gen, 0.3)
    ora.step(a, 5.0, s + 1); hip.step(a.cuda(), 5.0, s + 1); torch.cuda.synchronize()
    dq = (hip.tensor("DOF_POS").cpu() - ora.tensor("DOF_POS")).abs()
    dv = (hip.tensor("DOF_VEL").cpu() - ora.tensor("DOF_VEL")).abs()
    dr = (hip.tensor("ROOT_STATES").cpu() - ora.tensor("ROOT_STATES")).abs()
    print("step", s, "dq max per dof:", [round(float(x), 5) for x in dq.max(0).values], "\n   dv max", round(float(dv.max()), 4), "root", [round(float(x), 5) for x in dr.max(0).values])
    print("   rew diff", float((hip.tensor("REW").cpu() - ora.tensor("REW")).abs().max()), "torque diff", float((hip.tensor("TORQUES").cpu() - ora.tensor("TORQUES")).abs().max()))
