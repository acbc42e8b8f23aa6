from .modules import ActorCriticMLP, MLP, get_activation  # noqa: F401
from .ppo import PPO  # noqa: F401
from .runner import OnPolicyRunner  # noqa: F401
from .storage import RolloutStorage  # noqa: F401
