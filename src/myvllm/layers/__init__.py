from minivllm_amd.layers import *  # noqa: F401,F403
