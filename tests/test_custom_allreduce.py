"""Peer-to-peer all-reduce kernel: two ranks share the one GPU of the test box and map each
other's buffers through HIP IPC handles (on a node the peers are other GPUs over xGMI; the flag
protocol, the one-/two-shot data paths and graph replay are the same)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, q):
    import torch.distributed as dist
    from minivllm_amd.parallel.custom_allreduce import ONE_SHOT_MAX_BYTES, CustomAllReduce
    try:
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank,
                                world_size=world)
        car = CustomAllReduce(dist.group.WORLD, rank, world, torch.device("cuda", 0),
                              max_bytes=4 << 20)
        assert car.self_test()
        worst = 0.0
        sizes = [16 * world, 4096, ONE_SHOT_MAX_BYTES // 2, ONE_SHOT_MAX_BYTES // 2 + 8 * world,
                 256 * 4096, 2 << 20,                      # elements; both sides of the switch
                 5 * 512, 448 * 512, 333 * 1024, 24 * 5120]   # [tokens, hidden] of real steps
        for dtype in (torch.bfloat16, torch.float16):
            for n in sizes:
                assert n % (8 * world) == 0
                g = torch.Generator(device="cuda").manual_seed(1000 * rank + n % 977)
                x = torch.randn(n, device="cuda", dtype=dtype, generator=g)
                assert car.fits(x)
                mine = x.float().cpu()
                parts = [torch.empty_like(mine) for _ in range(world)]
                dist.all_gather(parts, mine)
                want = sum(parts)
                got = car.all_reduce(x).float().cpu()
                worst = max(worst, float((got - want).abs().max() / (want.abs().max() + 1e-6)))
        # all-reduce fused with residual add + RMSNorm against the two kernels it replaces
        # (same bits) and against fp32 math
        from minivllm_amd.ops import hip
        fused_exact, fused_err = True, 0.0
        for dtype in (torch.bfloat16, torch.float16):
            for rows, hidden in ((1, 4096), (7, 1024), (32, 4096), (16, 5120), (3, 12288), (64, 64)):
                g = torch.Generator(device="cuda").manual_seed(77 * rank + rows + hidden)
                x = torch.randn(rows, hidden, device="cuda", dtype=dtype, generator=g)
                gs = torch.Generator(device="cuda").manual_seed(rows + hidden)      # same on both ranks
                res = torch.randn(rows, hidden, device="cuda", dtype=dtype, generator=gs)
                gamma = (torch.rand(hidden, device="cuda", generator=gs) + 0.5).to(dtype)
                assert car.fits_norm(x, res, gamma)
                mine = x.float().cpu()
                parts = [torch.empty_like(mine) for _ in range(world)]
                dist.all_gather(parts, mine)
                x1, r1 = x.clone(), res.clone()
                car.all_reduce(x1)
                h1, r1 = hip.add_rms_norm(x1, r1, gamma, 1e-6)
                h2, r2 = car.all_reduce_add_rms_norm(x.clone(), res.clone(), gamma, 1e-6)
                fused_exact &= bool(torch.equal(h1, h2)) and bool(torch.equal(r1, r2))
                r32 = sum(parts).cuda() + res.float()
                h32 = r32 * torch.rsqrt(r32.pow(2).mean(-1, keepdim=True) + 1e-6) * gamma.float()
                fused_err = max(fused_err, float((h2.float() - h32).abs().max()),
                                float((r2.float() - r32).abs().max() / 4))
        big = torch.zeros(64, 4096, device="cuda", dtype=torch.bfloat16)
        assert not car.fits_norm(big, big.clone(), torch.ones(4096, device="cuda", dtype=torch.bfloat16))
        assert car.norm_mode(big, big.clone(), torch.ones(4096, device="cuda", dtype=torch.bfloat16)) == "two"
        assert fused_exact, "fused all-reduce + norm differs from all-reduce then norm"
        assert fused_err < 0.08, fused_err
        # two-shot fused form: rank r reduces + normalises rows [r R, (r + 1) R) only, keeps its
        # slice of the residual stream (ResidualShard) and gathers the peers' normalised rows.
        # Two norms in a row, as in a decoder layer (the second consumes the shard of the first);
        # the K-slice partials of a split GEMM as input; against the two-kernel route (same bits).
        from minivllm_amd.ops.partials import ResidualShard, SplitKPartials
        two_exact = True
        for dtype in (torch.bfloat16, torch.float16):
            for rows, hidden in ((8 * world, 4096), (256, 4096), (16 * world, 5120), (world, 64),
                                 (40 * world, 1024)):
                R = rows // world
                mine = slice(rank * R, (rank + 1) * R)
                gs = torch.Generator(device="cuda").manual_seed(5 * rows + hidden)
                res = torch.randn(rows, hidden, device="cuda", dtype=dtype, generator=gs)
                gamma = (torch.rand(hidden, device="cuda", generator=gs) + 0.5).to(dtype)
                r_ref, shard = res.clone(), res.clone()
                for layer in range(2):
                    g = torch.Generator(device="cuda").manual_seed(31 * rank + rows + hidden + layer)
                    x = torch.randn(rows, hidden, device="cuda", dtype=dtype, generator=g)
                    assert car.norm_mode(x, shard, gamma) == "two" or rows * hidden * 2 <= car.one_shot_max_bytes
                    x1 = x.clone()
                    car.all_reduce(x1)
                    h1, r_ref = hip.add_rms_norm(x1, r_ref, gamma, 1e-6)
                    h2, shard = car.all_reduce_add_rms_norm(x.clone(), shard, gamma, 1e-6, two_shot=True)
                    assert isinstance(shard, ResidualShard) and tuple(shard.rows.shape) == (R, hidden)
                    two_exact &= bool(torch.equal(h1, h2)) and bool(torch.equal(r_ref[mine], shard.rows))
                # K-slice partials: three fp32 slices that sum to this rank's contribution
                g = torch.Generator(device="cuda").manual_seed(17 * rank + rows)
                ws = torch.randn(3, rows, hidden, device="cuda", dtype=torch.float32, generator=g)
                part = SplitKPartials(ws, None, dtype)
                x1 = part.materialize()
                car.all_reduce(x1)
                h1, r1 = hip.add_rms_norm(x1, res.clone(), gamma, 1e-6)
                mode = car.norm_mode(part, res, gamma)
                assert mode in ("one", "two")
                h2, r2 = car.all_reduce_add_rms_norm(part, res.clone(), gamma, 1e-6, two_shot=mode == "two")
                r2 = r2.rows if isinstance(r2, ResidualShard) else r2[mine]
                two_exact &= bool(torch.equal(h1, h2)) and bool(torch.equal(r1[mine], r2))
        assert two_exact, "two-shot fused all-reduce + norm differs from all-reduce then norm"
        # graph replay: three calls of different sizes in one graph, replayed on fresh inputs
        bufs = [torch.zeros(n, device="cuda", dtype=torch.bfloat16) for n in (4096, 256 * 4096, 64)]
        fx = torch.zeros(8, 256, device="cuda", dtype=torch.bfloat16)
        fr = torch.zeros(8, 256, device="cuda", dtype=torch.bfloat16)
        ones = torch.ones(256, device="cuda", dtype=torch.bfloat16)
        tx = torch.zeros(2 * world, 256, device="cuda", dtype=torch.bfloat16)
        tr = torch.zeros(2 * world, 256, device="cuda", dtype=torch.bfloat16)
        for b in bufs:
            car.all_reduce(b)                                # warm-up outside capture
        car.all_reduce_add_rms_norm(fx, fr, ones, 1e-6)
        car.all_reduce_add_rms_norm(tx, tr, ones, 1e-6, two_shot=True)
        torch.cuda.synchronize()
        dist.barrier()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            for b in bufs:
                car.all_reduce(b)
            car.all_reduce_add_rms_norm(fx, fr, ones, 1e-6)
            car.all_reduce_add_rms_norm(tx, tr, ones, 1e-6, two_shot=True)
        graph_ok = True
        for it in range(3):
            for b in bufs:
                b.fill_(float(rank + 1 + it))
            fx.fill_(float(rank + 1 + it))
            fr.fill_(1.0)
            tx.fill_(float(rank + 1 + it))
            tr.fill_(1.0)
            torch.cuda.synchronize()
            dist.barrier()
            g.replay()
            torch.cuda.synchronize()
            expect = float(sum(r + 1 + it for r in range(world)))
            graph_ok &= all(bool((b.float() == expect).all()) for b in bufs)
            # constant rows: the residual becomes sum + 1 and the normalised row is all ones
            graph_ok &= bool((fr.float() == expect + 1.0).all())
            graph_ok &= bool((fx.float() - 1.0).abs().max() < 1e-2)
            # two-shot: only this rank's rows of the residual are updated; every row is normalised
            graph_ok &= bool((tr[2 * rank:2 * rank + 2].float() == expect + 1.0).all())
            graph_ok &= bool((tx.float() - 1.0).abs().max() < 1e-2)
            dist.barrier()
        healthy = car.healthy()
        assert not car.fits(torch.zeros(24, device="cuda", dtype=torch.float32))
        car.close()
        dist.destroy_process_group()
        q.put((rank, worst, graph_ok, healthy, None))
    except Exception as e:                                   # report instead of hanging the peer
        import traceback
        q.put((rank, None, False, False, traceback.format_exc()))


@pytest.mark.parametrize("world", [2, 4, 8])
def test_ranks_sharing_one_gpu_match_the_sum(world):
    """2, 4 and 8 ranks (the kernel is instantiated per group size) all on the one GPU."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29400 + os.getpid() % 300 + 7 * world
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=170) for _ in procs]
    for p in procs:
        p.join(timeout=30)
        if p.is_alive():
            p.terminate()
    for rank, worst, graph_ok, healthy, err in res:
        assert err is None, err
        assert worst < 1e-2, (rank, worst)               # one bf16 rounding of the fp32 sum
        assert graph_ok and healthy


def _late_peer_worker(rank, world, port, q):
    import torch.distributed as dist
    from minivllm_amd.parallel.custom_allreduce import CustomAllReduce, PeerAllReduceError
    try:
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank,
                                world_size=world)
        car = CustomAllReduce(dist.group.WORLD, rank, world, torch.device("cuda", 0),
                              max_bytes=1 << 20, timeout_ms=300)
        x = torch.full((4096,), float(rank + 1), device="cuda", dtype=torch.bfloat16)
        car.all_reduce(x)                                   # a normal call first
        ok_first = bool((x.float() == 3.0).all()) and car.healthy()
        dist.barrier()
        verdict = None
        if rank == 0:
            # the peer never makes this call: the kernel must give up after ~0.3 s, leave NaN
            # instead of a sum over stale buffers, and the status check must raise
            y = torch.ones(4096, device="cuda", dtype=torch.bfloat16)
            h = torch.ones(4, 1024, device="cuda", dtype=torch.bfloat16)
            r = torch.ones(4, 1024, device="cuda", dtype=torch.bfloat16)
            gamma = torch.ones(1024, device="cuda", dtype=torch.bfloat16)
            car.all_reduce(y)
            car.all_reduce_add_rms_norm(h, r, gamma, 1e-6)
            torch.cuda.synchronize()
            raised = False
            try:
                car.check()
            except PeerAllReduceError:
                raised = True
            verdict = (bool(torch.isnan(y.float()).all()), bool(torch.isnan(h.float()).all()),
                       not car.healthy(), raised)
        dist.barrier()
        car._release()                                      # no collective close: state is broken
        dist.destroy_process_group()
        q.put((rank, ok_first, verdict, None))
    except Exception:
        import traceback
        q.put((rank, False, None, traceback.format_exc()))


def test_a_peer_that_never_arrives_poisons_the_output_and_raises():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29750 + os.getpid() % 200
    procs = [ctx.Process(target=_late_peer_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {r[0]: r for r in (q.get(timeout=120) for _ in procs)}
    for p in procs:
        p.join(timeout=30)
        if p.is_alive():
            p.terminate()
    for r in res.values():
        assert r[3] is None, r[3]
        assert r[1]
    assert res[0][2] == (True, True, True, True)


def test_loopback_rehearsal_of_one_rank():
    """One process standing in for a rank of an 8-way group: every barrier lane raises the flag
    its peer would raise, all pointer tables are this rank's own buffers.  All four kernel forms
    must complete (eagerly and as graph replays) and stay healthy; a one-shot sum over 8 copies
    of the same buffer is 8 x the input."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import torch.distributed as dist
    from minivllm_amd.ops.partials import ResidualShard
    from minivllm_amd.parallel.custom_allreduce import CustomAllReduce
    torch.cuda.set_device(0)
    port = 29333 + os.getpid() % 200
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    try:
        car = CustomAllReduce(dist.group.WORLD, 3, 8, torch.device("cuda", 0), max_bytes=4 << 20,
                              loopback=True)
        assert car.self_test()
        x = torch.full((4096,), 1.5, device="cuda", dtype=torch.bfloat16)
        assert bool((car.all_reduce(x).float() == 12.0).all())
        big = torch.ones(256 * 4096, device="cuda", dtype=torch.bfloat16)
        h = torch.ones(256, 4096, device="cuda", dtype=torch.bfloat16)
        r = torch.ones(256, 4096, device="cuda", dtype=torch.bfloat16)
        gamma = torch.ones(4096, device="cuda", dtype=torch.bfloat16)
        small = torch.ones(16, 4096, device="cuda", dtype=torch.bfloat16)
        car.all_reduce(big)
        car.all_reduce_add_rms_norm(small, small.clone(), gamma, 1e-6)
        _, shard = car.all_reduce_add_rms_norm(h, r, gamma, 1e-6, two_shot=True)
        assert isinstance(shard, ResidualShard) and tuple(shard.rows.shape) == (32, 4096)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            for _ in range(4):
                car.all_reduce(big)
                _, shard = car.all_reduce_add_rms_norm(h, shard, gamma, 1e-6, two_shot=True)
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        assert car.healthy() and bool(torch.isfinite(h.float()).all())
        car.close()
    finally:
        dist.destroy_process_group()
