"""End-to-end check of the HTTP server on a real socket: starts
``python -m minivllm_amd.serve.api_server`` as a child process, waits for /health, sends a plain
and a streaming completion plus a small concurrent burst, prints the timings and a few /metrics
lines, and stops the child.  Usage: python tools/serve_smoke.py [model preset] [port]"""
import json, os, subprocess, sys, threading, time, urllib.request

preset = sys.argv[1] if len(sys.argv) > 1 else "qwen3-0.6b"
port = int(sys.argv[2]) if len(sys.argv) > 2 else 8123
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
base = f"http://127.0.0.1:{port}"
child = subprocess.Popen([sys.executable, "-m", "minivllm_amd.serve.api_server", "--model-preset",
                          preset, "--port", str(port), "--mixed-batches"], cwd=root)


def post(path, body, timeout=120):
    req = urllib.request.Request(base + path, json.dumps(body).encode(),
                                 {"Content-Type": "application/json"})
    return urllib.request.urlopen(req, timeout=timeout)


try:
    t0 = time.time()
    while True:
        if child.poll() is not None:
            raise SystemExit(f"server exited with {child.returncode}")
        try:
            urllib.request.urlopen(base + "/health", timeout=2).read()
            break
        except Exception:
            if time.time() - t0 > 240:
                raise SystemExit("server did not come up")
            time.sleep(1.0)
    print(f"server up after {time.time() - t0:.1f} s", flush=True)
    prompt = list(range(1000, 1128))
    t = time.perf_counter()
    r = json.load(post("/v1/completions", {"prompt": prompt, "max_tokens": 64, "temperature": 0.7,
                                           "ignore_eos": True}))
    dt = time.perf_counter() - t
    print(f"completion: {r['usage']} finish={r['choices'][0]['finish_reason']} in {dt*1e3:.1f} ms "
          f"({dt / 64 * 1e3:.2f} ms/token incl. HTTP)", flush=True)
    t = time.perf_counter()
    first, n = None, 0
    with post("/v1/completions", {"prompt": prompt, "max_tokens": 64, "temperature": 0.7,
                                  "ignore_eos": True, "stream": True}) as resp:
        for line in resp:
            line = line.decode().strip()
            if line.startswith("data: ") and line != "data: [DONE]":
                n += len(json.loads(line[6:])["choices"][0]["token_ids"])
                first = first or time.perf_counter() - t
    print(f"stream: {n} tokens, first chunk after {first*1e3:.1f} ms, all after "
          f"{(time.perf_counter() - t)*1e3:.1f} ms", flush=True)
    lat = []

    def one(i):
        t = time.perf_counter()
        json.load(post("/v1/completions", {"prompt": [(7 * i + j) % 5000 + 10 for j in range(200)],
                                           "max_tokens": 64, "temperature": 0.7, "ignore_eos": True}))
        lat.append(time.perf_counter() - t)
    t = time.perf_counter()
    ths = [threading.Thread(target=one, args=(i,)) for i in range(64)]
    [th.start() for th in ths]
    [th.join(180) for th in ths]
    dt = time.perf_counter() - t
    print(f"64 concurrent requests (200 in / 64 out): {dt:.2f} s wall, {64 * 64 / dt:.0f} generated "
          f"tok/s, request latency median {sorted(lat)[len(lat) // 2]*1e3:.0f} ms", flush=True)
    r = json.load(post("/v1/chat/completions", {"messages": [{"role": "user", "content": "Hello!"}],
                                                "max_tokens": 8}))
    print("chat:", r["choices"][0]["finish_reason"], r["usage"], flush=True)
    for line in urllib.request.urlopen(base + "/metrics", timeout=10).read().decode().splitlines():
        if line.startswith(("minivllm_requests_total", "minivllm_generated_tokens_total",
                            "minivllm_time_to_first_token_seconds_sum", "minivllm_kv_blocks_free")):
            print("metrics:", line, flush=True)
finally:
    child.terminate()
    try:
        child.wait(30)
    except subprocess.TimeoutExpired:
        child.kill()
