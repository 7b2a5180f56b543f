#!/usr/bin/env python3
"""End-to-end demo with the reference's call pattern (reference: main.py:15-71): build
``LLMEngine(config=dict)``, format 60 chat prompts, ``generate`` with T=0.6 / 256 new tokens,
print per-step and aggregate prefill / decode tokens per second.

    python main.py                     # Qwen3-0.6B shapes, random weights, GPU if present
    python main.py --model qwen3-8b --world-size 8
    python main.py --device cpu --tiny  # plumbing check without a GPU

Differences from the reference demo: the model dimensions are the real Qwen3 ones (the
reference mixes head_dim 64, an odd vocabulary and ``max_position`` 128 < prompt + 256 new
tokens), budgets are sized for one MI355X, and when no Qwen3 tokenizer is available locally a
byte-level tokenizer is used instead of downloading one.
"""
import argparse
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).parent))
sys.path.insert(0, str(Path(__file__).parent / "src"))

from myvllm.engine.llm_engine import LLMEngine as LLM          # noqa: E402
from myvllm.models.qwen3 import Qwen3ForCausalLM               # noqa: E402,F401
from myvllm.sampling_parameters import SamplingParams          # noqa: E402

config = {
    'max_num_sequences': 64,
    'max_num_batched_tokens': 16384,
    'block_size': 256,
    'world_size': 1,
    'model_name_or_path': 'Qwen/Qwen3-0.6B',
    'enforce_eager': False,
    'model_preset': 'qwen3-0.6b',
    'max_model_length': 4096,
    'gpu_memory_utilization': 0.9,
    'eos': 151645,
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="qwen3-0.6b")
    ap.add_argument("--world-size", type=int, default=1)
    ap.add_argument("--device", default="auto")
    ap.add_argument("--tiny", action="store_true", help="tiny model + short generation (CPU check)")
    ap.add_argument("--max-tokens", type=int, default=256)
    ap.add_argument("--checkpoint", default=None, help="HF Qwen3 safetensors directory")
    ap.add_argument("--show", type=int, default=0, help="print the first N completions")
    a = ap.parse_args()
    cfg = dict(config, model_preset=a.model, world_size=a.world_size, device=a.device,
               log_stats=True)
    if a.checkpoint:
        cfg["checkpoint_path"] = a.checkpoint
        cfg["model_name_or_path"] = a.checkpoint
    if a.tiny:
        cfg.update(model_preset="qwen3-tiny", max_position=2048, max_model_length=1024,
                   max_num_batched_tokens=2048,
                   max_num_sequences=16, block_size=16, eos=257, dtype="fp32")
        a.max_tokens = min(a.max_tokens, 16)
    llm = LLM(config=cfg)
    tokenizer = llm.tokenizer
    sampling_params = SamplingParams(temperature=0.6, max_tokens=a.max_tokens)
    prompts = ["introduce yourself" * 15, "list all prime numbers within 100" * 15] * 30
    prompts = [tokenizer.apply_chat_template([{"role": "user", "content": p}], tokenize=False,
                                             add_generation_prompt=True) for p in prompts]
    vocab = llm.cfg.model.vocab_size
    ids = [[t % vocab for t in tokenizer.encode(p)] for p in prompts]
    outputs = llm.generate(ids, sampling_params)
    for prompt, text in list(zip(prompts, outputs["text"]))[:a.show]:
        print(f"\nPrompt: {prompt!r}\nCompletion: {text!r}")
    st = llm.stats.as_dict()
    print(f"\nprefill: {st['prefill_tokens']} tokens in {st['prefill_steps']} steps, "
          f"{st['prefill_tok_s']:.0f} tok/s")
    print(f"decode : {st['decode_tokens']} tokens in {st['decode_steps']} steps, "
          f"{st['decode_tok_s']:.0f} tok/s")
    llm.exit()


if __name__ == "__main__":
    main()
