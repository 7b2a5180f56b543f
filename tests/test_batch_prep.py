"""ModelRunner.prepare_prefill builds its arrays in one vectorised pass; this checks them row by
row against the definition (reference: src/myvllm/engine/model_runner.py:229-270, which loops
over sequences and tokens)."""
import numpy as np

from minivllm_amd.config import as_engine_config
from minivllm_amd.engine.model_runner import ModelRunner
from minivllm_amd.engine.sequence import Sequence
from minivllm_amd.sampling_parameters import SamplingParams


def _runner(bs):
    cfg = dict(model_preset="qwen3-tiny", num_layers=1, max_num_sequences=8, max_num_batched_tokens=512,
               block_size=bs, max_model_length=256, eos=-1, max_cached_blocks=64, dtype="fp32",
               device="cpu")
    return ModelRunner(as_engine_config(cfg), 0)


def _inside(seqs, num_decode):
    """Order of the sequences inside a batch: the riders (the caller's last ``num_decode``) first."""
    first_decode = len(seqs) - num_decode
    return list(seqs[first_decode:]) + list(seqs[:first_decode])


def _expected(seqs, num_decode, bs):
    ids, pos, slots, cu_q, cu_k = [], [], [], [0], [0]
    for i, s in enumerate(_inside(seqs, num_decode)):
        if i < num_decode:
            start, end = s.num_tokens - 1, s.num_tokens
        else:
            start = s.num_cached_tokens
            end = s.num_tokens if s.prefill_end is None else s.prefill_end
        for p in range(start, end):
            ids.append(s.token_ids[p])
            pos.append(p)
            slots.append(s.block_table[p // bs] * bs + p % bs)
        cu_q.append(cu_q[-1] + end - start)
        cu_k.append(cu_k[-1] + end)
    return ids, pos, slots, cu_q, cu_k


def test_prepare_prefill_arrays_match_the_definition():
    rng = np.random.default_rng(3)
    for bs in (16, 32):
        r = _runner(bs)
        seqs = []
        next_block = 5
        for n, cached, cut in ((37, 0, None), (70, 32, None), (90, 0, 50), (1, 0, None),
                               (55, 0, None), (41, 0, None)):
            s = Sequence(rng.integers(0, 1000, size=n).tolist(), SamplingParams(temperature=0.7, max_tokens=4),
                         block_size=bs)
            nb = (n + bs - 1) // bs
            s.block_table = list(range(next_block, next_block + nb))[::-1]      # not monotonic
            next_block += nb + 2
            s.num_cached_tokens = cached
            s.prefill_end = cut
            seqs.append(s)
        for num_decode in (0, 2):           # the last two ride along with one decode row each
            b = r.prepare_prefill(seqs, num_decode=num_decode)
            ids, pos, slots, cu_q, cu_k = _expected(seqs, num_decode, bs)
            a = b.arrays
            n = len(ids)
            assert a["input_ids"][:n].tolist() == ids and a["input_ids"].dtype == np.int32
            assert a["positions"][:n].tolist() == pos and a["positions"].dtype == np.int32
            assert a["slot_mapping"][:n].tolist() == slots and a["slot_mapping"].dtype == np.int32
            assert a["cu_seqlens_q"].tolist() == cu_q and a["cu_seqlens_k"].tolist() == cu_k
            assert b.num_prefill_tokens == n - num_decode and b.num_seqs == len(seqs)
            assert b.max_seqlen_k == max(cu_k[i + 1] - cu_k[i] for i in range(len(seqs)))
            first_decode = len(seqs) - num_decode
            assert b.max_seqlen_q == max(cu_q[i + 1] - cu_q[i] for i in range(num_decode, len(seqs)))
            assert b.has_prefix            # one sequence has cached tokens
            for i, s in enumerate(_inside(seqs, num_decode)):
                assert a["block_tables"][i, :len(s.block_table)].tolist() == s.block_table
                assert (a["block_tables"][i, len(s.block_table):] == -1).all()
            if num_decode:
                assert a["context_lens"].tolist() == [s.num_tokens for s in seqs[first_decode:]]
