from minivllm_amd.engine.scheduler import *  # noqa: F401,F403
from minivllm_amd.engine import scheduler as _m
globals().update({k: v for k, v in vars(_m).items() if not k.startswith("__")})
