"""Resolved config values == the reference's class_to_dict dump (SURVEY G-9)."""
import json
import os

import numpy as np
import pytest

from wiki_grx_gym_amd.envs import config

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
OWN_KEYS = {"grx", "model"}  # additions of this build


def _cmp(path, mine, ref):
    if isinstance(ref, dict):
        assert isinstance(mine, dict), path
        missing = set(ref) - set(mine) - {"init_member_classes"}  # staticmethod of the reference BaseConfig
        assert not missing, f"{path}: missing keys {missing}"
        extra = set(mine) - set(ref) - OWN_KEYS
        assert not extra, f"{path}: unexpected keys {extra}"
        for k in set(ref) - {"init_member_classes"}:
            _cmp(path + "." + k, mine[k], ref[k])
    elif isinstance(ref, list):
        mine = mine.tolist() if isinstance(mine, np.ndarray) else mine
        assert len(mine) == len(ref), path
        for i, (a, b) in enumerate(zip(mine, ref)):
            _cmp(f"{path}[{i}]", a, b)
    elif isinstance(ref, float) or isinstance(mine, float):
        assert mine == pytest.approx(ref, rel=1e-12, abs=1e-15), path
    else:
        assert mine == ref, path


@pytest.mark.parametrize("name,cls", [("GR1T1", config.GR1T1Cfg), ("GR1T1PPO", config.GR1T1CfgPPO),
                                      ("GR1T2", config.GR1T2Cfg), ("GR1T2PPO", config.GR1T2CfgPPO)])
def test_config_dump(name, cls):
    ref = json.load(open(os.path.join(G, "config_dump.json")))[name]
    mine = config.class_to_dict(cls())
    _cmp(name, mine, ref)


def test_class_to_dict_is_alphabetical():
    d = config.class_to_dict(config.GR1T1Cfg().rewards.scales)
    assert list(d) == sorted(d)
