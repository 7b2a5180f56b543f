from minivllm_amd.sampling_parameters import SamplingParams  # noqa: F401
