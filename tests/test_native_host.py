"""C++ host core (csrc/host/block_manager.cpp) against the Python implementation: same hash,
same allocation decisions on random workloads, same engine outputs."""
import numpy as np
import pytest
import xxhash

from minivllm_amd.build import build_host
from minivllm_amd.engine.block_manager import (BlockManager, NativeBlockManager, _host_lib,
                                               make_block_manager)
from minivllm_amd.engine.sequence import Sequence
from minivllm_amd.sampling_parameters import SamplingParams

BS = 4


@pytest.fixture(scope="module", autouse=True)
def built():
    build_host(verbose=False)
    assert _host_lib() is not None


def seq(tokens):
    return Sequence(list(tokens), SamplingParams(temperature=1.0, max_tokens=64), block_size=BS)


@pytest.mark.parametrize("n", [0, 1, 3, 4, 7, 8, 31, 32, 33, 63, 64, 100, 257, 1024])
def test_xxh64_matches_the_library(n):
    import ctypes
    rng = np.random.default_rng(n)
    data = rng.integers(0, 256, size=n, dtype=np.uint8).tobytes()
    lib = _host_lib()
    lib.mvl_xxh64.restype = ctypes.c_uint64
    lib.mvl_xxh64.argtypes = [ctypes.c_char_p, ctypes.c_size_t]
    assert lib.mvl_xxh64(data, n) == xxhash.xxh64(data).intdigest()


def test_chain_hash_matches_python():
    rng = np.random.default_rng(0)
    for n in (4, 16, 256):
        toks = rng.integers(0, 150000, size=n).tolist()
        h0 = BlockManager.compute_hash(toks)
        assert NativeBlockManager.compute_hash(toks) == h0
        assert NativeBlockManager.compute_hash(toks, h0) == BlockManager.compute_hash(toks, h0)


def test_same_decisions_as_python_on_a_random_workload():
    rng = np.random.default_rng(42)
    py, cc = BlockManager(48, BS), NativeBlockManager(48, BS)
    prefixes = [rng.integers(0, 50, size=int(n)).tolist() for n in (8, 12, 4, 16)]
    live = []                                             # pairs (python seq, native seq)
    ahead = 0
    for step in range(3000):
        op = rng.integers(0, 10)
        if op < 3:                                        # admit a prompt, often with a shared prefix
            toks = (prefixes[rng.integers(0, 4)] if rng.random() < 0.7 else []) + \
                rng.integers(0, 50, size=int(rng.integers(1, 14))).tolist()
            a, b = seq(toks), seq(toks)
            assert py.can_allocate(a) == cc.can_allocate(b)
            if py.can_allocate(a):
                py.allocate(a), cc.allocate(b)
                live.append((a, b))
        elif op < 8 and live:                             # one decode step of a random sequence
            a, b = live[rng.integers(0, len(live))]
            t = int(rng.integers(0, 50))
            a.append_token(t), b.append_token(t)
            assert py.can_append(a) == cc.can_append(b)
            if py.can_append(a):
                py.append(a), cc.append(b)
                # now and then the step after this one is launched ahead: the block its token
                # starts is given out before that token exists
                if rng.random() < 0.3 and py.num_free_blocks >= 1:
                    py.append_ahead(a), cc.append_ahead(b)
                    ahead += len(a.block_table) > a.num_blocks
            else:                                         # preempt: free and forget
                py.deallocate(a), cc.deallocate(b)
                live.remove((a, b))
        elif live:                                        # finish
            a, b = live.pop(int(rng.integers(0, len(live))))
            py.deallocate(a), cc.deallocate(b)
        for a, b in live:
            assert a.block_table == b.block_table and a.num_cached_tokens == b.num_cached_tokens
        assert py.num_free_blocks == cc.num_free_blocks
    ps, cs = py.stats(), cc.stats()
    assert (ps["free"], ps["used"], ps["cached_hashes"]) == (cs["free"], cs["used"], cs["cached_hashes"])
    assert (py.num_prefix_hits, py.num_prefix_queries) == (cc.num_prefix_hits, cc.num_prefix_queries)
    assert py.num_prefix_hits > 50                        # the workload did exercise sharing
    assert ahead > 20                                     # ... and blocks given out ahead
    for a, b in live:
        assert all(py.blocks[i].ref_count == cc.ref_count(i) for i in a.block_table)


def test_factory_prefers_native_and_can_be_forced(monkeypatch):
    assert isinstance(make_block_manager(4, BS), NativeBlockManager)
    assert isinstance(make_block_manager(4, BS, native=False), BlockManager)
    monkeypatch.setenv("MINIVLLM_PY_HOST", "1")
    assert isinstance(make_block_manager(4, BS), BlockManager)
    with pytest.raises(ValueError):
        make_block_manager(0, BS)


def test_engine_outputs_do_not_depend_on_the_allocator(monkeypatch):
    import torch
    from minivllm_amd.config import EngineConfig, ModelConfig
    from minivllm_amd.engine.llm_engine import LLMEngine

    def run():
        model = ModelConfig(vocab_size=331, hidden_size=64, num_heads=4, num_kv_heads=2, head_dim=32,
                            intermediate_size=96, num_layers=2, max_position=256, base=10000.0)
        eng = LLMEngine(EngineConfig(model=model, max_num_sequences=4, max_num_batched_tokens=64,
                                     block_size=8, max_model_length=96, eos=-1, device="cpu",
                                     dtype=torch.float32, seed=11, max_cached_blocks=14))
        rng = np.random.default_rng(3)
        shared = rng.integers(0, 331, size=16).tolist()
        prompts = [shared + rng.integers(0, 331, size=n).tolist() for n in (5, 21, 9, 16, 30)]
        out = eng.generate(prompts, SamplingParams(temperature=0, max_tokens=12, ignore_eos=True))
        kind = type(eng.scheduler.block_manager).__name__
        stats = (eng.scheduler.num_preemptions, eng.scheduler.block_manager.num_prefix_hits)
        eng.exit()
        return out["token_ids"], kind, stats

    native = run()
    monkeypatch.setenv("MINIVLLM_PY_HOST", "1")
    python = run()
    assert (native[1], python[1]) == ("NativeBlockManager", "BlockManager")
    assert native[0] == python[0] and native[2] == python[2]


def test_block_manager_under_address_and_ub_sanitizers(tmp_path):
    """Host code only: the C++ allocator and a randomised scheduler-like workload
    (tests/native/block_manager_stress.cpp) built with ASan + UBSan, runtimes linked statically."""
    import os
    import shutil
    import subprocess
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("no g++")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "bm_stress")
    build = subprocess.run(
        [gxx, "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined",
         "-fno-sanitize-recover=all", "-static-libasan", "-static-libubsan",
         os.path.join(root, "tests", "native", "block_manager_stress.cpp"),
         os.path.join(root, "minivllm_amd", "csrc", "host", "block_manager.cpp"), "-o", exe],
        capture_output=True, text=True, timeout=300)
    if build.returncode != 0:
        pytest.skip("sanitizer runtimes not available: " + build.stderr[-300:])
    run = subprocess.run([exe, "8000"], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stdout + run.stderr
    assert run.stdout.startswith("ok:")
