"""Registered tasks (reference: legged_gym/envs/__init__.py:30-55): "GR1T1" and "GR1T2" are the
LOWER-LIMB configs, exactly as the reference registers them.  "GR1T1_full_body" is this build's addition: the
unfixed-upper-body robot of BASELINE.json config 5 (32 DOF, generic-tree kernel, build-defined observations)."""
from .config import (GR1T1Cfg, GR1T1CfgPPO, GR1T1FullBodyCfg, GR1T1FullBodyCfgPPO, GR1T1FullCfgPPO, GR1T2Cfg, GR1T2CfgPPO, LeggedRobotCfg,  # noqa: F401
                     LeggedRobotCfgPPO, LeggedRobotFFTAICfg, LeggedRobotFFTAICfgPPO)
from .grx_env import GR1T1, GR1T2, GRxEnv, LeggedRobot, LeggedRobotFFTAI  # noqa: F401


def _register():
    from ..utils.task_registry import task_registry
    task_registry.register("GR1T1", GR1T1, GR1T1Cfg(), GR1T1CfgPPO())
    task_registry.register("GR1T2", GR1T2, GR1T2Cfg(), GR1T2CfgPPO())
    task_registry.register("GR1T1_full_body", GR1T1, GR1T1FullBodyCfg(), GR1T1FullBodyCfgPPO())


_register()
