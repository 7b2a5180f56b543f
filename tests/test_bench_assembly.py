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


def _assemble(dp, tp, tpc, long_ctx=None):
    out, good = bench.assemble_multi_gpu(ARGS, 8, dp, tp, tpc, "decode tok/s", BASE, CFG, long_ctx)
    json.loads(json.dumps(out))                       # serialisable
    assert REQUIRED <= set(out)
    return out, good


def test_tensor_parallel_is_the_headline_at_the_one_gpu_batch():
    long_ctx = dict(_res(dec=900.0, pre=120000.0), config={"model": "qwen3-32b", "seq_len": 32768})
    out, good = _assemble(_res(), _res(dec=60000.0), {"error": "tpc phase exceeded 360s"}, long_ctx)
    assert good and out["value"] == 60000.0 and out["scaling"] == "strong"
    assert out["config"]["parallelism"] == "tp8" and out["config"]["global_batch"] == 256
    assert out["prefill_tok_s"] == 94000.0 and out["secondary"]["decode_tok_s"] == 35000.0
    side = out["data_parallel_replicas"]
    assert side["decode_tok_s"] == 8 * 18000.0 and side["scaling"] == "weak"
    assert side["global_batch"] == 2048 and "not the reference" in side["note"]
    assert "error" in out["tensor_parallel_p2p_allreduce"]
    assert out["qwen3_32b_tp8_32k_prefill"]["seq_len"] == 32768
    assert out["qwen3_32b_tp8_32k_prefill"]["prefill_tok_s"] == 120000.0


def test_the_faster_all_reduce_plane_is_the_headline_and_named():
    out, good = _assemble(_res(), _res(dec=60000.0), _res(dec=75000.0))
    assert good and out["value"] == 75000.0 and out["config"]["all_reduce"] == "p2p"
    assert out["config"]["parallelism"] == "tp8" and out["scaling"] == "strong"
    assert out["tensor_parallel_rccl_allreduce"]["decode_tok_s"] == 60000.0
    assert "tensor_parallel_p2p_allreduce" not in out
    out, good = _assemble(_res(), _res(dec=60000.0), _res(dec=50000.0))
    assert out["value"] == 60000.0 and out["config"]["all_reduce"] == "rccl"
    assert out["tensor_parallel_p2p_allreduce"]["decode_tok_s"] == 50000.0
    # RCCL phase failed, peer-to-peer phase delivered: still a TP headline
    out, good = _assemble(_res(), {"error": "tp phase exceeded 600s"}, _res(dec=70000.0))
    assert good and out["value"] == 70000.0 and out["config"]["all_reduce"] == "p2p"
    assert "error" in out["tensor_parallel_rccl_allreduce"] and "headline_fallback" not in out


def test_data_parallel_stands_in_only_as_a_marked_fallback():
    out, good = _assemble(_res(), {"error": "tp phase child exited with code 1"}, None)
    assert good and out["value"] == 8 * 18000.0 and out["scaling"] == "weak"
    assert out["config"]["parallelism"] == "dp8" and "headline_fallback" in out
    assert "error" in out["tensor_parallel"]


def test_all_phases_failed_still_prints_an_object():
    out, good = _assemble({"error": "a"}, {"error": "b"}, None)
    assert not good and out["value"] == 0.0 and out["tensor_parallel"] == {"error": "b"}
    assert out["config"]["parallelism"] == "tp8"
    out, good = _assemble(None, None, None)           # --no-dp --no-tp
    assert not good


def test_prefill_only_mode_has_a_headline():
    r = _res()
    for k in ("decode_tok_s", "decode_ms_per_step"):
        r["main"].pop(k)
    out, good = _assemble(None, r, None)
    assert good and out["value"] == 94000.0 and out["ms_per_step"] == 170.0


def test_overall_time_budget_of_a_multi_gpu_run(monkeypatch):
    """Rank 0 decides what a side phase may still take; all ranks read the same number; a phase
    with too little left is skipped; without a store every phase keeps its own limit."""
    import argparse
    import time

    class FakeStore(dict):
        def set(self, k, v):
            self[k] = v.encode()

        def get(self, k):
            return self[k]

    args = argparse.Namespace(phase_timeout=600, total_timeout=1000, port=1)

    def clock(rank, store, started_ago):
        c = bench.JobClock.__new__(bench.JobClock)
        c.args, c.rank, c.store, c.t0 = args, rank, store, time.time() - started_ago
        return c

    store = FakeStore()
    assert clock(0, store, 100).limit_for("l32k") == 600         # plenty left: the phase's own limit
    assert clock(1, store, 500).limit_for("l32k") == 600         # rank 1 takes rank 0's number
    assert clock(0, store, 700).limit_for("tpc") in (279, 280)   # what is left, minus the reserve
    assert clock(0, store, 900).limit_for("dp") == 0             # not worth starting
    assert clock(1, store, 0).limit_for("dp") == 0
    assert clock(0, None, 990).limit_for("dp") == 360            # no store: no overall limit
    assert clock(0, store, 0).agree("tp_retry", "1") == "1" and clock(1, store, 0).agree("tp_retry", "0") == "1"
