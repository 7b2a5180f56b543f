from minivllm_amd.models import qwen3 as _m
globals().update({k: v for k, v in vars(_m).items() if not k.startswith("__")})
