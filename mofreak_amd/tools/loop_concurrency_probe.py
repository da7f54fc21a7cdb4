"""How much the frame loop (detector + descriptors, mofreak_compute_stream) gains when several of them share the GPU:
N contexts (own stream, own workspaces), a host thread each, every thread runs the loop on its own resident stack.
usage: python mofreak_amd/tools/loop_concurrency_probe.py [pairs_per_call=128] [calls=6]"""
import sys, os, time, threading, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from mofreak_amd import api, synth

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 128
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 6
W, H = 1920, 1080
fr = synth.moving_objects_stack(9, W, H)
T = pairs + 5
stack_h = np.stack([fr[t % len(fr)] for t in range(T)])


def worker(ctx, stack, rows, out, i, barrier):
    ctx.compute_stream(stack, T, W, H, rows, capacity=rows.numel() // 32)
    ctx.synchronize()
    barrier.wait()
    t0 = time.perf_counter()
    for _ in range(calls):
        n = ctx.compute_stream(stack, T, W, H, rows, capacity=rows.numel() // 32)[0]
    ctx.synchronize()
    out[i] = (time.perf_counter() - t0, n)


for n_ctx in (1, 2, 3):
    ctxs = [api.Context() for _ in range(n_ctx)]
    stacks = [torch.from_numpy(stack_h).cuda() for _ in range(n_ctx)]
    rows = [torch.empty(pairs * 12000 * 32, dtype=torch.uint8, device="cuda") for _ in range(n_ctx)]
    out = [None] * n_ctx
    barrier = threading.Barrier(n_ctx)
    th = [threading.Thread(target=worker, args=(ctxs[i], stacks[i], rows[i], out, i, barrier)) for i in range(n_ctx)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    wall = max(o[0] for o in out)
    print(f"contexts={n_ctx} pairs/call={pairs} rows/call={out[0][1]} aggregate {n_ctx * pairs * calls / wall:.0f} pairs/s ({1e3 * wall / calls:.2f} ms per round of {n_ctx} calls)", flush=True)
    for c in ctxs:
        c.close()
    del stacks, rows
    torch.cuda.empty_cache()
