"""Registered tasks (reference: legged_gym/envs/__init__.py:30-55): "GR1T1" and "GR1T2" are the
LOWER-LIMB configs, exactly as the reference registers them."""
from .config import (GR1T1Cfg, GR1T1CfgPPO, GR1T2Cfg, GR1T2CfgPPO, LeggedRobotCfg, LeggedRobotCfgPPO,  # noqa: F401
                     LeggedRobotFFTAICfg, LeggedRobotFFTAICfgPPO)
from .grx_env import GR1T1, GR1T2, GRxEnv, LeggedRobot, LeggedRobotFFTAI  # noqa: F401


def _register():
    from ..utils.task_registry import task_registry
    task_registry.register("GR1T1", GR1T1, GR1T1Cfg(), GR1T1CfgPPO())
    task_registry.register("GR1T2", GR1T2, GR1T2Cfg(), GR1T2CfgPPO())


_register()
