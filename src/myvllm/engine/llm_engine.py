from minivllm_amd.engine.llm_engine import *  # noqa: F401,F403
from minivllm_amd.engine import llm_engine as _m
globals().update({k: v for k, v in vars(_m).items() if not k.startswith("__")})
