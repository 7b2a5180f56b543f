"""bench.py's JSON assembly for N > 1 (pure Python): whatever the phases return - numbers,
errors, nothing - rank 0 must end with one well-formed JSON object."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

ARGS = argparse.Namespace(batch=256, second_model="qwen3-0.6b", model="qwen3-8b")
BASE = {"unit": "tokens/s", "n_gpus": 8}
CFG = {"model": "qwen3-8b", "seq_len": 1024}
REQUIRED = {"metric", "value", "unit", "n_gpus", "ms_per_step", "scaling", "config"}


def _res(dec=18000.0, pre=94000.0):
    m = {"init_s": 2.0, "kv_blocks": 100, "prefill_tok_s": pre, "prefill_ms_per_step": 170.0,
         "prefill_tokens_per_step": 16384, "decode_tok_s": dec, "decode_ms_per_step": 14.0}
    return {"main": m, "second": dict(m, decode_tok_s=35000.0), "phase_wall_s": 60.0}


def _assemble(dp, tp, tpc):
    out, good = bench.assemble_multi_gpu(ARGS, 8, dp, tp, tpc, "decode tok/s", BASE, CFG)
    json.loads(json.dumps(out))                       # serialisable
    assert REQUIRED <= set(out)
    return out, good


def test_data_parallel_headline_scales_per_replica_numbers():
    out, good = _assemble(_res(), _res(dec=60000.0), {"error": "tpc phase exceeded 600s"})
    assert good and out["value"] == 8 * 18000.0 and out["scaling"] == "weak"
    assert out["config"]["parallelism"] == "dp8" and out["config"]["global_batch"] == 2048
    assert out["prefill_tok_s"] == 8 * 94000.0 and out["secondary"]["decode_tok_s"] == 8 * 35000.0
    assert out["tensor_parallel"]["decode_tok_s"] == 60000.0
    assert out["tensor_parallel"]["scaling"] == "strong"
    assert "error" in out["tensor_parallel_p2p_allreduce"]


def test_tensor_parallel_becomes_headline_when_data_parallel_failed():
    out, good = _assemble({"error": "dp phase child exited with code 1"}, _res(dec=60000.0), None)
    assert good and out["value"] == 60000.0 and out["scaling"] == "strong"
    assert out["config"]["parallelism"] == "tp8" and "error" in out["data_parallel_replicas"]


def test_all_phases_failed_still_prints_an_object():
    out, good = _assemble({"error": "a"}, {"error": "b"}, None)
    assert not good and out["value"] == 0.0 and out["tensor_parallel"] == {"error": "b"}
    out, good = _assemble(None, None, None)           # --no-dp --no-tp
    assert not good


def test_prefill_only_mode_has_a_headline():
    r = _res()
    for k in ("decode_tok_s", "decode_ms_per_step"):
        r["main"].pop(k)
    out, good = _assemble(r, None, None)
    assert good and out["value"] == 8 * 94000.0 and out["ms_per_step"] == 170.0
