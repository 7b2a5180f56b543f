"""Import-path compatibility with the reference package name: ``myvllm.*`` re-exports
``minivllm_amd.*`` (reference layout: src/myvllm/{engine,layers,models,utils})."""
import os as _os
import sys as _sys

_root = _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
if _root not in _sys.path:
    _sys.path.insert(0, _root)
