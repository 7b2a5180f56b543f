"""OpenAI-style HTTP front end for ``AsyncLLMEngine``.

    python -m minivllm_amd.serve.api_server --model-preset qwen3-8b --port 8000 \\
           [--checkpoint-path /path/to/Qwen3-8B] [--data-parallel-size 8] [--world-size 1] \\
           [--max-num-sequences 1024] [--mixed-batches]

Endpoints: ``POST /v1/completions`` and ``POST /v1/chat/completions`` (JSON in, JSON or
server-sent events out with ``"stream": true``), ``GET /v1/models``, ``GET /health``,
``GET /metrics`` (Prometheus text format).  A client that disconnects mid-stream aborts its
request, which returns its KV blocks to the pool; so does a match of one of the request's
``stop`` strings (the text ends before it, ``finish_reason`` is ``"stop"``).  Prompts may be text or token-id lists (random
weights and the byte-level fallback tokenizer make token ids the practical choice offline).

The reference has no server (its entry point is the offline demo, reference: main.py:43-71); the
request fields follow the OpenAI completions API so existing clients work unchanged.
"""
from __future__ import annotations

import argparse
import asyncio
import json
import time
import uuid
from typing import List, Optional, Union

from fastapi import FastAPI, HTTPException, Request
from fastapi.responses import JSONResponse, Response, StreamingResponse

from minivllm_amd.sampling_parameters import SamplingParams
from minivllm_amd.serve.async_engine import AsyncLLMEngine, RequestHandle

MAX_SEQUENCES_PER_REQUEST = 256          # prompts x n of one /v1/completions call


class _TextStream:
    """Incremental detokenisation: text is released only up to the last complete character
    (a multi-byte character may span tokens)."""

    def __init__(self, tokenizer):
        self.tok, self.ids, self.sent = tokenizer, [], 0

    def push(self, tokens: List[int], final: bool = False) -> str:
        self.ids.extend(tokens)
        text = self.tok.decode(self.ids, skip_special_tokens=True)
        if not final:
            while text.endswith("�"):
                text = text[:-1]
        delta, self.sent = text[self.sent:], max(self.sent, len(text))
        return delta


class _StopMatcher:
    """OpenAI ``stop`` strings over a text stream: text is released up to the earliest stop
    string (which is not part of the output), and while none has matched, the longest tail
    that could still grow into one is held back."""

    def __init__(self, stops: List[str]):
        self.stops, self.buf, self.hit = stops, "", False

    def feed(self, piece: str, final: bool = False) -> str:
        if self.hit:
            return ""
        self.buf += piece
        cut = min((i for i in (self.buf.find(st) for st in self.stops) if i >= 0), default=-1)
        if cut >= 0:
            out, self.buf, self.hit = self.buf[:cut], "", True
            return out
        hold = 0
        if not final:
            for st in self.stops:
                for n in range(min(len(st) - 1, len(self.buf)), hold, -1):
                    if self.buf.endswith(st[:n]):
                        hold = n
                        break
        out, self.buf = self.buf[:len(self.buf) - hold], self.buf[len(self.buf) - hold:]
        return out


def _metrics():
    from prometheus_client import CollectorRegistry, Counter, Gauge, Histogram
    reg = CollectorRegistry()
    lat = (0.005, 0.01, 0.02, 0.05, 0.1, 0.2, 0.5, 1, 2, 5, 10, 30, 60)
    return reg, {
        "requests": Counter("minivllm_requests", "requests by outcome", ["finish_reason"], registry=reg),
        "prompt_tokens": Counter("minivllm_prompt_tokens", "prompt tokens received", registry=reg),
        "generated_tokens": Counter("minivllm_generated_tokens", "tokens generated", registry=reg),
        "ttft": Histogram("minivllm_time_to_first_token_seconds", "submit -> first token",
                          buckets=lat, registry=reg),
        "e2e": Histogram("minivllm_request_seconds", "submit -> finished", buckets=lat, registry=reg),
        "running": Gauge("minivllm_running_requests", "sequences holding KV blocks", registry=reg),
        "waiting": Gauge("minivllm_waiting_requests", "sequences queued for prefill", registry=reg),
        "kv_free": Gauge("minivllm_kv_blocks_free", "free KV cache blocks", registry=reg),
        "prefill_tok_s": Gauge("minivllm_prefill_tokens_per_second", "engine average", registry=reg),
        "decode_tok_s": Gauge("minivllm_decode_tokens_per_second", "engine average", registry=reg),
    }


def create_app(engine: AsyncLLMEngine, model_name: str = "minivllm"):
    app = FastAPI(title="minivllm_amd")
    registry, m = _metrics()
    max_len = engine.cfg.max_model_length

    def _account(h: RequestHandle) -> None:
        m["requests"].labels(_reason(h) or "unknown").inc()
        m["prompt_tokens"].inc(h.num_prompt_tokens)
        m["generated_tokens"].inc(len(h.token_ids))
        if h.first_token_at is not None:
            m["ttft"].observe(h.first_token_at - h.submitted_at)
        if h.finished_at is not None:
            m["e2e"].observe(h.finished_at - h.submitted_at)

    def _reason(h: RequestHandle) -> Optional[str]:
        return "stop" if h.stop_text is not None else h.finish_reason

    def _text(h: RequestHandle) -> str:
        cut = h.stop_text
        return cut if cut is not None else engine.tokenizer.decode(h.token_ids,
                                                                   skip_special_tokens=True)

    def _stops(body: dict) -> List[str]:
        st = body.get("stop")
        st = [] if st is None else [st] if isinstance(st, str) else st
        if not isinstance(st, list) or len(st) > 8 or not all(isinstance(x, str) and x for x in st):
            raise HTTPException(400, "stop must be a non-empty string or a list of up to 8 of them")
        return st

    def _n(body: dict) -> int:
        """``n`` completions per prompt: the copies share the prompt's KV blocks through the
        prefix cache and differ by their sampling noise (keyed by the batch row)."""
        try:
            n = int(body.get("n") or 1)
        except (TypeError, ValueError):
            n = 0
        if not 1 <= n <= 16:
            raise HTTPException(400, "n must be an integer between 1 and 16")
        return n

    def _params(body: dict) -> SamplingParams:
        max_tokens = body.get("max_tokens")
        if max_tokens is None:
            max_tokens = body.get("max_completion_tokens")
        try:
            return SamplingParams(temperature=float(body.get("temperature", 1.0)),
                                  max_tokens=16 if max_tokens is None else int(max_tokens),
                                  ignore_eos=bool(body.get("ignore_eos", False)),
                                  stop_token_ids=[int(t) for t in body.get("stop_token_ids") or []],
                                  top_k=int(body.get("top_k") or 0),
                                  top_p=float(1.0 if body.get("top_p") is None else body["top_p"]))
        except (TypeError, ValueError) as e:
            raise HTTPException(400, str(e))

    def _prompts(p) -> List[Union[str, List[int]]]:
        if isinstance(p, str):
            return [p]
        if isinstance(p, list) and p and all(isinstance(t, int) for t in p):
            return [p]
        if isinstance(p, list) and p and all(
                isinstance(x, str) or (isinstance(x, list) and x and all(isinstance(t, int) for t in x))
                for x in p):
            return list(p)
        raise HTTPException(400, "prompt must be a string, a token-id list, or a list of those")

    def _usage(hs: List[RequestHandle]) -> dict:
        pt, ct = sum(h.num_prompt_tokens for h in hs), sum(len(h.token_ids) for h in hs)
        return {"prompt_tokens": pt, "completion_tokens": ct, "total_tokens": pt + ct}

    async def _collect(h: RequestHandle, stops: List[str]) -> None:
        ts, sm, text = _TextStream(engine.tokenizer), _StopMatcher(stops), []
        try:
            async for toks in h:
                if stops and not sm.hit:
                    text.append(sm.feed(ts.push(toks)))
                    if sm.hit:                              # the rest of the generation is not wanted
                        h.stop_text = "".join(text)
                        h.cancel()
            if stops and not sm.hit:
                text.append(sm.feed(ts.push([], final=True), final=True))
                if sm.hit:
                    h.stop_text = "".join(text)
        finally:
            if not h.finished:
                h.cancel()

    async def _run_all(hs: List[RequestHandle], stops: List[str] = ()) -> None:
        try:
            await asyncio.gather(*(_collect(h, list(stops)) for h in hs))
        except ValueError as e:                             # rejected by the scheduler
            raise HTTPException(400, str(e))
        finally:
            for h in hs:
                if h.finished:
                    _account(h)
                else:
                    h.cancel()

    async def _body(request: Request) -> dict:
        try:
            body = await request.json()
        except Exception:
            raise HTTPException(400, "request body is not valid JSON")
        if not isinstance(body, dict):
            raise HTTPException(400, "request body must be a JSON object")
        return body

    def _sse(obj) -> bytes:
        return f"data: {json.dumps(obj)}\n\n".encode()

    async def _stream(hs: List[RequestHandle], make_chunk, request: Request,
                      stops: List[str] = ()):
        """Interleave the token streams of ``hs`` as server-sent events."""
        out: asyncio.Queue = asyncio.Queue()

        async def pump(i: int, h: RequestHandle):
            ts, sm = _TextStream(engine.tokenizer), _StopMatcher(list(stops))
            try:
                async for toks in h:
                    if sm.hit:
                        continue                            # draining after the cancel
                    piece = ts.push(toks)
                    if stops:
                        piece = sm.feed(piece)
                        if sm.hit:
                            h.stop_text = ""                # marks the outcome for the metrics
                            h.cancel()
                    await out.put(make_chunk(i, piece, toks, None))
                tail = ""
                if not sm.hit:
                    tail = ts.push([], final=True)
                    if stops:
                        tail = sm.feed(tail, final=True)
                        if sm.hit:
                            h.stop_text = ""
                await out.put(make_chunk(i, tail, [], _reason(h)))
            except Exception as e:
                await out.put({"error": {"message": str(e), "type": type(e).__name__}})
            await out.put(None)

        tasks = [asyncio.ensure_future(pump(i, h)) for i, h in enumerate(hs)]
        live = len(tasks)
        try:
            while live:
                item = await out.get()
                if item is None:
                    live -= 1
                    continue
                yield _sse(item)
            yield b"data: [DONE]\n\n"
        finally:                                            # also runs when the client went away
            for t in tasks:
                t.cancel()
            for h in hs:
                if not h.finished:
                    h.cancel()
                else:
                    _account(h)

    @app.get("/health")
    async def health():
        return {"status": "ok"}

    @app.get("/v1/models")
    async def models():
        return {"object": "list", "data": [{"id": model_name, "object": "model",
                                            "owned_by": "minivllm_amd", "max_model_len": max_len}]}

    @app.get("/metrics")
    async def metrics():
        from prometheus_client import generate_latest
        # a data-parallel front end asks its replica processes and waits for the answers:
        # off the event loop
        st = await asyncio.get_running_loop().run_in_executor(None, engine.stats)
        m["running"].set(st["running"]); m["waiting"].set(st["waiting"])
        m["kv_free"].set(st["kv_blocks_free"])
        m["prefill_tok_s"].set(st["prefill_tok_s"]); m["decode_tok_s"].set(st["decode_tok_s"])
        return Response(generate_latest(registry), media_type="text/plain; version=0.0.4")

    @app.post("/v1/completions")
    async def completions(request: Request):
        body = await _body(request)
        sp, prompts, stops = _params(body), _prompts(body.get("prompt")), _stops(body)
        loop = asyncio.get_running_loop()
        rid, created = f"cmpl-{uuid.uuid4().hex}", int(time.time())
        n = _n(body)
        if len(prompts) * n > MAX_SEQUENCES_PER_REQUEST:
            raise HTTPException(400, f"a request may ask for at most {MAX_SEQUENCES_PER_REQUEST} "
                                     f"sequences (prompts x n), got {len(prompts) * n}")
        hs = [engine.submit(p, sp, loop=loop) for p in prompts for _ in range(n)]
        if body.get("stream"):
            def chunk(i, text, toks, reason):
                return {"id": rid, "object": "text_completion", "created": created,
                        "model": model_name,
                        "choices": [{"index": i, "text": text, "token_ids": toks,
                                     "finish_reason": reason}]}
            return StreamingResponse(_stream(hs, chunk, request, stops),
                                     media_type="text/event-stream")
        await _run_all(hs, stops)
        return JSONResponse({
            "id": rid, "object": "text_completion", "created": created, "model": model_name,
            "choices": [{"index": i, "text": _text(h),
                         "token_ids": h.token_ids, "finish_reason": _reason(h)}
                        for i, h in enumerate(hs)],
            "usage": _usage(hs)})

    @app.post("/v1/chat/completions")
    async def chat(request: Request):
        body = await _body(request)
        msgs = body.get("messages")
        if not isinstance(msgs, list) or not msgs:
            raise HTTPException(400, "messages must be a non-empty list")
        tok = engine.tokenizer
        try:
            text = tok.apply_chat_template(msgs, tokenize=False, add_generation_prompt=True)
        except Exception:                                   # tokenizer without a template
            text = "".join(f"<|im_start|>{x['role']}\n{x['content']}<|im_end|>\n" for x in msgs)
            text += "<|im_start|>assistant\n"
        sp, stops, n = _params(body), _stops(body), _n(body)
        rid, created = f"chatcmpl-{uuid.uuid4().hex}", int(time.time())
        loop = asyncio.get_running_loop()
        hs = [engine.submit(text, sp, loop=loop) for _ in range(n)]
        if body.get("stream"):
            first = [True] * n

            def chunk(i, piece, toks, reason):
                delta = {"content": piece}
                if first[i]:
                    delta["role"], first[i] = "assistant", False
                return {"id": rid, "object": "chat.completion.chunk", "created": created,
                        "model": model_name,
                        "choices": [{"index": i, "delta": delta, "finish_reason": reason}]}
            return StreamingResponse(_stream(hs, chunk, request, stops),
                                     media_type="text/event-stream")
        await _run_all(hs, stops)
        return JSONResponse({
            "id": rid, "object": "chat.completion", "created": created, "model": model_name,
            "choices": [{"index": i, "finish_reason": _reason(h),
                         "message": {"role": "assistant", "content": _text(h)}}
                        for i, h in enumerate(hs)],
            "usage": _usage(hs)})

    return app


def main(argv: Optional[List[str]] = None) -> None:
    ap = argparse.ArgumentParser(description="OpenAI-style server on minivllm_amd")
    ap.add_argument("--model-preset", default="qwen3-0.6b")
    ap.add_argument("--checkpoint-path", default=None)
    ap.add_argument("--tokenizer", default=None, help="local HF tokenizer directory")
    ap.add_argument("--host", default="127.0.0.1")
    ap.add_argument("--port", type=int, default=8000)
    ap.add_argument("--world-size", type=int, default=1, help="tensor-parallel degree")
    ap.add_argument("--data-parallel-size", type=int, default=1,
                    help="engine replicas (each on its own --world-size GPUs) behind this server")
    ap.add_argument("--max-num-sequences", type=int, default=256)
    ap.add_argument("--max-num-batched-tokens", type=int, default=16384)
    ap.add_argument("--max-model-length", type=int, default=8192)
    ap.add_argument("--min-prefill-tokens", type=int, default=0)
    ap.add_argument("--max-prefill-delay-ms", type=float, default=0.0)
    ap.add_argument("--mixed-batches", action="store_true")
    ap.add_argument("--kv-cache-dtype", choices=["auto", "fp8"], default="auto")
    ap.add_argument("--device", default="cuda")
    ap.add_argument("--enforce-eager", action="store_true")
    a = ap.parse_args(argv)
    cfg = {"model_preset": a.model_preset, "world_size": a.world_size,
           "max_num_sequences": a.max_num_sequences,
           "max_num_batched_tokens": a.max_num_batched_tokens,
           "max_model_length": a.max_model_length, "min_prefill_tokens": a.min_prefill_tokens,
           "max_prefill_delay_ms": a.max_prefill_delay_ms,
           "enable_mixed_batches": a.mixed_batches, "kv_cache_dtype": a.kv_cache_dtype,
           "device": a.device, "enforce_eager": a.enforce_eager or a.device != "cuda",
           "log_stats": False}
    if a.checkpoint_path:
        cfg["checkpoint_path"] = a.checkpoint_path
    if a.tokenizer or a.checkpoint_path:
        cfg["model_name_or_path"] = a.tokenizer or a.checkpoint_path
    if a.device != "cuda":
        cfg["dtype"] = "float32"
    import uvicorn
    if a.data_parallel_size > 1:
        from minivllm_amd.serve.dp_router import DataParallelAsyncEngine
        engine = DataParallelAsyncEngine(cfg, a.data_parallel_size)
    else:
        engine = AsyncLLMEngine(cfg)
    try:
        uvicorn.run(create_app(engine, a.model_preset), host=a.host, port=a.port, log_level="info")
    finally:
        engine.shutdown(drain=False)


if __name__ == "__main__":
    main()
