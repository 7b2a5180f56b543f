from minivllm_amd.utils.context import *  # noqa: F401,F403
