from minivllm_amd.layers import activation as _m
globals().update({k: v for k, v in vars(_m).items() if not k.startswith("__")})
