"""Online serving on top of ``LLMEngine``: a background engine loop that accepts requests at any
time and streams tokens back (``async_engine``), and an OpenAI-style HTTP front end
(``api_server``).  The reference is an offline library (``generate`` on a fixed prompt list,
reference: src/myvllm/engine/llm_engine.py:75-88); this package is what a deployment adds."""
from .async_engine import AsyncLLMEngine, RequestHandle  # noqa: F401


def __getattr__(name):          # spawns processes: imported on demand
    if name == "DataParallelAsyncEngine":
        from .dp_router import DataParallelAsyncEngine
        return DataParallelAsyncEngine
    raise AttributeError(name)
