"""Tensor parallelism without a cluster: shard math for tp in {1,2,4,8} with no backend,
the shared-memory control channel across processes, and the full engine at world_size 2
over gloo on CPU (TP=2 tokens == TP=1 tokens)."""
import multiprocessing as mp
import os
import time

import numpy as np
import pytest
import torch

from minivllm_amd.config import EngineConfig, ModelConfig
from minivllm_amd.engine.rpc import ShmBroadcastChannel, decode_message, encode_message
from minivllm_amd.parallel import group as pg
from minivllm_amd.sampling_parameters import SamplingParams


# ------------------------------------------------------------ shard math
@pytest.mark.parametrize("tp", [1, 2, 4, 8])
def test_linear_shards_reassemble(tp):
    from minivllm_amd.layers.linear import (ColumnParallelLinear, MergedColumnParallelLinear,
                                            QKVColumnParallelLinear, RowParallelLinear)
    torch.manual_seed(0)
    hid, inter, H, HKV, D = 64, 96 * 2, 16, 8, 8
    x = torch.randn(5, hid)
    w_col, w_gate, w_up = torch.randn(inter, hid), torch.randn(inter, hid), torch.randn(inter, hid)
    wq, wk, wv = torch.randn(H * D, hid), torch.randn(HKV * D, hid), torch.randn(HKV * D, hid)
    w_row = torch.randn(hid, inter)
    col_out, gate_out, up_out, q_out, k_out, v_out, row_out = [], [], [], [], [], [], 0
    for r in range(tp):
        kw = dict(tp_rank=r, tp_size=tp)
        c = ColumnParallelLinear(hid, inter, **kw); c.weight.weight_loader(c.weight, w_col)
        col_out.append(c(x))
        m = MergedColumnParallelLinear(hid, [inter, inter], **kw)
        m.weight.weight_loader(m.weight, w_gate, 0); m.weight.weight_loader(m.weight, w_up, 1)
        g, u = m(x).chunk(2, -1)
        gate_out.append(g); up_out.append(u)
        qkv = QKVColumnParallelLinear(hid, D, H, HKV, **kw)
        for t, sid in ((wq, "q"), (wk, "k"), (wv, "v")):
            qkv.weight.weight_loader(qkv.weight, t, sid)
        q, k, v = qkv(x).split([H * D // tp, HKV * D // tp, HKV * D // tp], -1)
        q_out.append(q); k_out.append(k); v_out.append(v)
        row = RowParallelLinear(inter, hid, reduce_results=False, **kw)
        row.weight.weight_loader(row.weight, w_row)
        xs = torch.randn(5, inter, generator=torch.Generator().manual_seed(1))
        row_out = row_out + row(xs[:, r * inter // tp:(r + 1) * inter // tp])
    xs = torch.randn(5, inter, generator=torch.Generator().manual_seed(1))
    tol = dict(atol=1e-4, rtol=1e-4)
    torch.testing.assert_close(torch.cat(col_out, -1), x @ w_col.T, **tol)
    torch.testing.assert_close(torch.cat(gate_out, -1), x @ w_gate.T, **tol)
    torch.testing.assert_close(torch.cat(up_out, -1), x @ w_up.T, **tol)
    torch.testing.assert_close(torch.cat(q_out, -1), x @ wq.T, **tol)
    torch.testing.assert_close(torch.cat(k_out, -1), x @ wk.T, **tol)
    torch.testing.assert_close(torch.cat(v_out, -1), x @ wv.T, **tol)
    torch.testing.assert_close(row_out, xs @ w_row.T, **tol)


@pytest.mark.parametrize("tp", [2, 8])
def test_vocab_parallel_embedding_pads_odd_vocab(tp):
    from minivllm_amd.layers.embedding_head import VocabParallelEmbedding
    V, hid = 151643 // 1000, 16            # odd vocabulary like the reference demo's
    full = torch.randn(V, hid)
    ids = torch.tensor([0, 1, V - 1, V // 2])
    total = 0
    for r in range(tp):
        e = VocabParallelEmbedding(V, hid, tp_rank=r, tp_size=tp)
        assert e.num_embeddings_padded % tp == 0 and e.num_embeddings_padded >= V
        e.weight.weight_loader(e.weight, full)
        mask = (ids >= e.vocab_start) & (ids < e.vocab_end)
        local = (ids - e.vocab_start) * mask
        total = total + torch.nn.functional.embedding(local, e.weight) * mask[:, None]
    torch.testing.assert_close(total, full[ids])


# -------------------------------------------------------- control channel
def test_message_codec_roundtrip():
    arrays = {"a": np.arange(7, dtype=np.int32), "b": np.ones((2, 3), np.float32),
              "empty": np.zeros((0,), np.int32)}
    m, s, a = decode_message(encode_message("run_batch", {"x": 1, "flag": True}, arrays))
    assert m == "run_batch" and s == {"x": 1, "flag": True}
    for k in arrays:
        assert a[k].dtype == arrays[k].dtype and np.array_equal(a[k], arrays[k])


def _reader(name, idx, n_msgs, q):
    ch = ShmBroadcastChannel(name, 2, create=False)
    got = []
    for _ in range(n_msgs):
        m, s, a = ch.recv(idx, timeout_s=30)
        got.append((m, s["i"], int(a["v"].sum())))
    ch.close()
    q.put((idx, got))


def test_shm_channel_broadcasts_in_order_without_overwrite():
    name = f"mvl_test_{os.getpid()}"
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    writer = ShmBroadcastChannel(name, 2, create=True, size=1 << 16)
    procs = [ctx.Process(target=_reader, args=(name, i, 50, q)) for i in range(2)]
    for p in procs:
        p.start()
    for i in range(50):                   # back-to-back sends: acks must gate the reuse
        writer.send("step", {"i": i}, {"v": np.full(100, i, np.int64)})
    res = dict(q.get(timeout=60) for _ in procs)
    for p in procs:
        p.join(timeout=30)
    writer.close()
    want = [("step", i, 100 * i) for i in range(50)]
    assert res[0] == want and res[1] == want
    with pytest.raises(ValueError):
        big = ShmBroadcastChannel(name + "b", 1, create=True, size=1024)
        try:
            big.send("x", {}, {"v": np.zeros(4096, np.int64)})
        finally:
            big.close()


# ------------------------------------------------------- engine, TP=2 on CPU
def _cfg(world, port):
    model = ModelConfig(vocab_size=331, hidden_size=64, num_heads=4, num_kv_heads=2, head_dim=32,
                        intermediate_size=96, num_layers=2, max_position=256, base=10000.0,
                        tie_word_embeddings=True)
    return EngineConfig(model=model, max_num_sequences=4, max_num_batched_tokens=64, block_size=8,
                        max_model_length=96, eos=-1, device="cpu", dtype=torch.float32, seed=11,
                        world_size=world, master_port=port, max_cached_blocks=24)


def _generate(world, port, **overrides):
    import dataclasses
    from minivllm_amd.engine.llm_engine import LLMEngine
    eng = LLMEngine(dataclasses.replace(_cfg(world, port), **overrides))
    rng = np.random.default_rng(3)
    prompts = [rng.integers(0, 331, size=n).tolist() for n in (5, 21, 9, 16, 30)]
    out = eng.generate(prompts, SamplingParams(temperature=0, max_tokens=10, ignore_eos=True))
    stats = (eng.scheduler.num_preemptions, eng.stats.decode_steps, eng.scheduler.num_mixed_steps)
    eng.exit()
    return out["token_ids"], stats


def test_tp2_gloo_matches_tp1():
    port = 29600 + os.getpid() % 300
    single, _ = _generate(1, port)
    double, stats = _generate(2, port + 1)          # spawns one gloo worker process
    assert double == single
    assert not pg.is_initialized()                  # exit() tore the group down
    # mixed prefill + decode steps cross the control channel like any other batch
    mixed, stats = _generate(2, port + 2, enable_mixed_batches=True, max_num_batched_tokens=40)
    assert mixed == single and stats[2] > 0
    # fp8 KV cache: quantisation is per element, so sharding the KV heads changes nothing
    q1, _ = _generate(1, port + 3, kv_cache_dtype="fp8")
    q2, _ = _generate(2, port + 4, kv_cache_dtype="fp8")
    assert q2 == q1


def test_tp4_gloo_matches_tp1():
    """Four ranks: every head count, the vocabulary padding (331 -> multiple of 256) and the
    four-way (value, index) sampling exchange differ from the two-rank case."""
    import dataclasses
    from minivllm_amd.engine.llm_engine import LLMEngine
    port = 29000 + os.getpid() % 250

    def run(world, p):
        base = _cfg(world, p)
        cfg = dataclasses.replace(base, model=dataclasses.replace(base.model, num_kv_heads=4,
                                                                  intermediate_size=128))
        eng = LLMEngine(cfg)
        rng = np.random.default_rng(4)
        prompts = [rng.integers(0, 331, size=n).tolist() for n in (7, 19, 3, 26)]
        out = eng.generate(prompts, SamplingParams(temperature=0, max_tokens=8, ignore_eos=True))
        eng.exit()
        return out["token_ids"]

    assert run(4, port + 4) == run(1, port)
    assert not pg.is_initialized()


def test_tp8_gloo_matches_tp1():
    """Eight ranks, one query/KV head each: the sharding of a whole MI355X node (control channel
    with seven readers, eight-way vocabulary split and sampling exchange)."""
    import dataclasses
    from minivllm_amd.engine.llm_engine import LLMEngine
    port = 29100 + os.getpid() % 250

    def run(world, p):
        base = _cfg(world, p)
        cfg = dataclasses.replace(base, model=dataclasses.replace(
            base.model, num_heads=8, num_kv_heads=8, head_dim=16, intermediate_size=128))
        eng = LLMEngine(cfg)
        rng = np.random.default_rng(6)
        prompts = [rng.integers(0, 331, size=n).tolist() for n in (9, 14, 4)]
        out = eng.generate(prompts, SamplingParams(temperature=0, max_tokens=6, ignore_eos=True))
        eng.exit()
        return out["token_ids"]

    before = os.environ.get("OMP_NUM_THREADS")
    os.environ["OMP_NUM_THREADS"] = "1"              # eight CPU ranks on a small machine
    try:
        assert run(8, port + 8) == run(1, port)
    finally:
        if before is None:
            os.environ.pop("OMP_NUM_THREADS", None)
        else:
            os.environ["OMP_NUM_THREADS"] = before
    assert not pg.is_initialized()


def test_forced_single_rank_group_runs_the_sharded_path():
    """world_size 1 with force_tp_group: collectives go through a real (gloo) group."""
    import dataclasses
    from minivllm_amd.engine.llm_engine import LLMEngine
    port = 29300 + os.getpid() % 300
    single, _ = _generate(1, port)
    eng = LLMEngine(dataclasses.replace(_cfg(1, port + 2), force_tp_group=True))
    assert pg.collectives_active()
    rng = np.random.default_rng(3)
    prompts = [rng.integers(0, 331, size=n).tolist() for n in (5, 21, 9, 16, 30)]
    out = eng.generate(prompts, SamplingParams(temperature=0, max_tokens=10, ignore_eos=True))
    eng.exit()
    assert out["token_ids"] == single and not pg.collectives_active()


# ------------------------------------------- chunk-pipelined row-parallel GEMM
def _overlap_worker(rank, world, port, q):
    from minivllm_amd.parallel import group as g
    from minivllm_amd.parallel.overlap import plan_chunks, row_parallel_linear
    g.init_tensor_parallel(rank, world, "gloo", "127.0.0.1", port)
    gen = torch.Generator().manual_seed(0)
    x_full = torch.randn(300, 64, generator=gen)
    w_full = torch.randn(48, 64, generator=gen)
    bias = torch.randn(48, generator=gen)
    xs, ws = x_full[:, rank * 32:(rank + 1) * 32], w_full[:, rank * 32:(rank + 1) * 32]
    res = {}
    for n in (1, 3, 4):
        res[n] = row_parallel_linear(xs.contiguous(), ws.contiguous(), bias if rank == 0 else None, n)
    want = x_full @ w_full.T + bias
    ok = all(torch.allclose(v, want, atol=1e-4, rtol=1e-4) for v in res.values())
    q.put((rank, ok, plan_chunks(100), plan_chunks(16384), plan_chunks(8192)))
    g.destroy_tensor_parallel()


def test_chunked_row_parallel_linear_matches_dense():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29900 + os.getpid() % 90
    procs = [ctx.Process(target=_overlap_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=30)
    assert all(o[1] for o in out)
    assert out[0][2:] == (1, 4, 2)


def test_chunk_bounds_use_bucket_sizes_and_cover_all_rows():
    from minivllm_amd.engine.model_runner import prefill_token_bucket
    from minivllm_amd.parallel import overlap
    for T in (4096, 5120, 16384, 8192 + 512, 4160):
        for n in (2, 3, 4):
            b = overlap.chunk_bounds(T, n)
            assert b[0] == 0 and b[-1] == T and all(x < y for x, y in zip(b, b[1:]))
            sizes = [y - x for x, y in zip(b, b[1:])]
            assert len(sizes) <= n
            assert all(s == prefill_token_bucket(s) for s in sizes[:-1])
    before = (overlap.MIN_CHUNK_TOKENS, overlap.MAX_CHUNKS)
    try:
        overlap.configure(min_chunk_tokens=512, max_chunks=8)
        assert overlap.plan_chunks(4096) == 8 and overlap.plan_chunks(1000) == 1
        overlap.configure(max_chunks=1)
        assert overlap.plan_chunks(1 << 20) == 1
    finally:
        overlap.MIN_CHUNK_TOKENS, overlap.MAX_CHUNKS = before


_DEAD_WORKER_SCRIPT = r"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, {root!r})
from minivllm_amd.config import EngineConfig, ModelConfig
from minivllm_amd.engine.llm_engine import LLMEngine
from minivllm_amd.sampling_parameters import SamplingParams
if __name__ == "__main__":
    model = ModelConfig(vocab_size=331, hidden_size=64, num_heads=4, num_kv_heads=2, head_dim=32,
                        intermediate_size=96, num_layers=2, max_position=256, base=10000.0)
    cfg = EngineConfig(model=model, max_num_sequences=4, max_num_batched_tokens=64, block_size=8,
                       max_model_length=96, eos=-1, device="cpu", dtype=torch.float32, seed=1,
                       world_size=2, master_port={port}, max_cached_blocks=24)
    eng = LLMEngine(cfg)
    eng.add_prompt(list(range(1, 20)), SamplingParams(temperature=0, max_tokens=40, ignore_eos=True))
    eng.step(); eng.step()
    eng.workers[0].kill()                       # this engine's own worker, by handle
    eng.workers[0].join(10)
    print("KILLED", flush=True)
    try:
        eng.step()
    except RuntimeError as e:
        print("RAISED", e, flush=True)
        raise
"""


def test_a_dead_worker_is_reported_and_teardown_does_not_stall():
    """Killing the TP worker between steps: the next step raises, and interpreter shutdown
    (atexit -> exit) neither waits for an acknowledgement nor for a barrier with the dead
    rank."""
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = _DEAD_WORKER_SCRIPT.format(root=root, port=29200 + os.getpid() % 300)
    t0 = time.monotonic()
    proc = subprocess.Popen([sys.executable, "-c", script], stdout=subprocess.PIPE,
                            stderr=subprocess.STDOUT, text=True)
    try:
        out, _ = proc.communicate(timeout=150)
    except subprocess.TimeoutExpired:
        proc.kill()
        out, _ = proc.communicate()
        pytest.fail("engine with a dead worker did not exit:\n" + out[-2000:])
    assert "KILLED" in out and "RAISED" in out and "died" in out, out[-2000:]
    assert proc.returncode not in (0, None)
    # generous bound for a loaded CI machine; the stall this guards against was 15 minutes
    assert time.monotonic() - t0 < 45, out[-2000:]


def _stepwise(world, port, ahead, lens=(12, 30, 7, 30)):
    """Drive the engine step by step; returns (tokens per request, decode steps launched ahead)."""
    import dataclasses
    from minivllm_amd.engine.llm_engine import LLMEngine
    eng = LLMEngine(dataclasses.replace(_cfg(world, port), decode_prelaunch=ahead, block_size=16,
                                        max_cached_blocks=32))
    rng = np.random.default_rng(5)
    seqs = [eng.add_prompt(rng.integers(0, 331, size=n).tolist(),
                           SamplingParams(temperature=0.7 if i % 2 else 0, max_tokens=m, ignore_eos=True))
            for i, (n, m) in enumerate(zip((5, 21, 9, 16), lens))]
    done, launched, steps = {}, 0, 0
    while not eng.is_finished():
        finished, _, _ = eng.step()
        done.update(finished)
        launched += eng.model_runner.has_prelaunched()
        steps += 1
        assert steps < 300
    eng.exit()
    return [done[s.seq_id] for s in seqs], launched


def test_tp2_launch_ahead_decode_matches_plain_loop():
    """Tensor-parallel engines launch decode step n + 1 before reading step n's tokens too: the
    workers derive it from step n on their own (``Batch.prelaunch`` / "prelaunch_next").  Same
    tokens as the plain loop and as one rank, with sampled (temperature > 0) rows in the batch,
    sequences that finish at different steps and KV-block edges in between."""
    port = 29150 + os.getpid() % 250
    plain, n0 = _stepwise(2, port, ahead=False)
    ahead, n1 = _stepwise(2, port + 1, ahead=True)
    single, _ = _stepwise(1, port + 2, ahead=True)
    assert n0 == 0 and n1 >= 8, (n0, n1)
    assert ahead == plain == single


def test_shard_rehearsal_builds_one_ranks_shapes_and_runs():
    """``tp_shard=N``: one process holds rank r's shard of an N-way engine (per-rank weight
    shapes and KV heads) and issues every collective through a single-member group.  Tokens
    are meaningless; shapes, the collective call sites and the engine loop are what is checked."""
    import dataclasses
    from minivllm_amd.engine.llm_engine import LLMEngine
    from minivllm_amd.parallel import group as pg
    port = 29420 + os.getpid() % 250
    base = _cfg(1, port)
    cfg = dataclasses.replace(base, tp_shard=2, tp_shard_rank=1,
                              model=dataclasses.replace(base.model, num_kv_heads=2))
    eng = LLMEngine(cfg)
    try:
        r = eng.model_runner
        assert pg.is_rehearsal() and pg.tp_size() == 2 and pg.tp_rank() == 1
        layer = r.model.model.layers[0]
        assert tuple(layer.self_attn.qkv_proj.weight.shape) == ((2 + 2 * 1) * 32, 64)
        assert tuple(layer.self_attn.o_proj.weight.shape) == (64, 2 * 32)
        assert tuple(layer.mlp.down_proj.weight.shape) == (64, 48)
        assert r.kv_cache.shape[3] == 1                        # one KV head per rank
        assert r.model.lm_head.weight.shape[0] == 384 // 2     # 331 padded to 2 x 64 multiples
        out = eng.generate([[1, 2, 3, 4, 5], list(range(20))],
                           SamplingParams(temperature=0, max_tokens=6, ignore_eos=True))
        assert [len(t) for t in out["token_ids"]] == [6, 6]
    finally:
        eng.exit()
    assert not pg.is_rehearsal() and pg.tp_size() == 1
    with pytest.raises(ValueError):
        dataclasses.replace(_cfg(2, port), tp_shard=2).validate()


@pytest.mark.parametrize("world,chunks", [(2, None), (4, 2)])
def test_prefill_reduce_scatter_norm_all_gather_matches_tp1(world, chunks):
    """Prefill-sized row-parallel outputs are reduce-scattered, normalised on the rank's own rows
    (row-sharded residual stream) and the normalised rows all-gathered; with ``chunks`` the GEMM
    is split over token chunks and each rank owns a share of every chunk.  Greedy tokens must
    equal one rank's (fp32 on the CPU: the sums are exact enough)."""
    import dataclasses
    from minivllm_amd.engine.llm_engine import LLMEngine
    from minivllm_amd.parallel import group as pg
    port = 28800 + os.getpid() % 250 + 3 * world

    def run(w, p):
        base = _cfg(w, p)
        cfg = dataclasses.replace(
            base, scatter_norm_min_rows=8, overlap_max_chunks=chunks,
            overlap_min_chunk_tokens=64 if chunks else None, max_num_batched_tokens=256,
            max_cached_blocks=64,
            model=dataclasses.replace(base.model, num_kv_heads=4, intermediate_size=128))
        eng = LLMEngine(cfg)
        rng = np.random.default_rng(6)
        # 256 rows in one step: two chunks of 128 when chunking is on
        lens = (64, 70, 60, 62) if chunks else (12, 20, 9, 23)
        prompts = [rng.integers(0, 331, size=n).tolist() for n in lens]
        before = pg.SCATTER_NORM_CALLS
        out = eng.generate(prompts, SamplingParams(temperature=0, max_tokens=6, ignore_eos=True))
        used = pg.SCATTER_NORM_CALLS - before
        eng.exit()
        pg.configure_scatter_norm(512)
        return out["token_ids"], used

    many, used = run(world, port)
    one, used1 = run(1, port + 1)
    assert used >= 4 and used1 == 0, (used, used1)        # 2 layers: o, down, o, down (into the final norm)
    assert many == one


def _writer_that_dies(name, q):
    from minivllm_amd.engine.rpc import ShmBroadcastChannel
    ch = ShmBroadcastChannel(name, 1, create=True)
    ch.send("hello", {"x": 1})
    q.put("sent")
    q.close()
    q.join_thread()                               # the message has left this process
    os._exit(0)                                   # no goodbye, the segment stays behind


def test_an_idle_reader_notices_that_the_writer_is_gone():
    """A worker waiting for its next request must not wait forever for a rank 0 that crashed:
    the channel header carries the writer's pid and an idle reader checks it once a second."""
    import multiprocessing as mp
    from minivllm_amd.engine.rpc import ChannelClosed, ShmBroadcastChannel
    name = f"mvl_test_dead_writer_{os.getpid()}"
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_writer_that_dies, args=(name, q))
    p.start()
    assert q.get(timeout=60) == "sent"
    reader = ShmBroadcastChannel(name, 1, create=False)
    try:
        method, scalars, _ = reader.recv(0, timeout_s=30)
        assert method == "hello" and scalars == {"x": 1}
        p.join(timeout=30)
        t0 = time.monotonic()
        with pytest.raises(ChannelClosed):
            reader.recv(0, timeout_s=30)
        assert time.monotonic() - t0 < 10
    finally:
        reader.close()
        from multiprocessing import shared_memory
        try:
            seg = shared_memory.SharedMemory(name=name)
            seg.close()
            seg.unlink()
        except FileNotFoundError:
            pass
