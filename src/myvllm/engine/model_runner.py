from minivllm_amd.engine.model_runner import *  # noqa: F401,F403
from minivllm_amd.engine import model_runner as _m
globals().update({k: v for k, v in vars(_m).items() if not k.startswith("__")})
