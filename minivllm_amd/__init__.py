"""MI355X-native minimal LLM inference engine (paged KV cache, continuous batching,
Qwen3, tensor parallelism) with hand-written gfx950 HIP kernels."""
from .config import EngineConfig, ModelConfig, model_preset
from .sampling_parameters import SamplingParams

__version__ = "0.1.0"


def __getattr__(name):          # heavy imports stay lazy
    if name == "LLMEngine":
        from .engine.llm_engine import LLMEngine
        return LLMEngine
    raise AttributeError(name)
