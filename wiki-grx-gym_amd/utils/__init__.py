from .helpers import (class_to_dict, export_policy_as_jit, get_args, get_load_path, parse_sim_params, set_seed,  # noqa: F401
                      update_cfg_from_args)
from .task_registry import task_registry  # noqa: F401
from .logger import Logger  # noqa: F401
