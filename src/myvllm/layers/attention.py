from minivllm_amd.layers import attention as _m
globals().update({k: v for k, v in vars(_m).items() if not k.startswith("__")})
