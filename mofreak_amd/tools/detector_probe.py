"""Times the keypoint detector (and, with `describe`, the descriptors behind it) on 1920x1080 moving-object frame pairs.

usage: python mofreak_amd/tools/detector_probe.py [pairs_per_call=32] [calls=6] [describe|loop]   (PROBE_NOISE=n: noisy frames)
       describe: mofreak_detect_pairs + mofreak_extract_pairs per call; loop: the whole frame loop, mofreak_compute_stream, per call
The command the detector profiles under profiles/ are taken on (mofreak_amd/tools/profile_detector.sh).
"""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from mofreak_amd import api, synth
pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
describe = len(sys.argv) > 3 and sys.argv[3] == "describe"
loop = len(sys.argv) > 3 and sys.argv[3] == "loop"
W, H = 1920, 1080
ctx = api.Context()
distinct = 4
fr = synth.moving_objects_stack(5 + distinct, W, H)
noise = int(os.environ.get("PROBE_NOISE", "0"))  # +-noise grey levels on every pixel of every frame: no exactly-still background
if noise:
    rng = np.random.default_rng(1)
    fr = np.clip(fr.astype(np.int16) + rng.integers(-noise, noise + 1, fr.shape, dtype=np.int16), 0, 255).astype(np.uint8)
cur = torch.from_numpy(np.stack([fr[5 + (p % distinct)] for p in range(pairs)])).cuda()
prev = torch.from_numpy(np.stack([fr[p % distinct] for p in range(pairs)])).cuda()
cap = 32768 * pairs
kps = torch.empty((cap, 3), dtype=torch.float32, device="cuda")
offs = torch.empty(pairs + 1, dtype=torch.int64, device="cuda")
desc = torch.empty((cap, 16), dtype=torch.uint8, device="cuda")
valid = torch.empty(cap, dtype=torch.uint8, device="cuda")
if loop:
    T = pairs + 5
    stack = torch.from_numpy(np.stack([fr[t % len(fr)] for t in range(T)])).cuda()
    rows = torch.empty(pairs * 12000 * 32, dtype=torch.uint8, device="cuda")


def call():
    if loop:
        return ctx.compute_stream(stack, T, W, H, rows, capacity=rows.numel() // 32)[1]
    n = ctx.detect_pairs(cur, prev, W, H, pairs, kps, offs, capacity=cap)
    if describe:
        ctx.extract_pairs(cur, prev, W, H, pairs, kps, desc, valid, kp_offsets=offs, n_kp=n)
    return n


n = call()
ctx.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    n = call()
ctx.synchronize()
t1 = time.perf_counter()
print(f"pairs={pairs} kp/pair={n/pairs:.0f} {'frame loop' if loop else 'detect+describe' if describe else 'detect'} {pairs*steps/(t1-t0):.0f} pairs/s  {1e3*(t1-t0)/steps:.3f} ms/call")
if loop:
    sys.exit(0)
sz = kps[:n, 2].cpu().numpy()
print("size quantiles", np.quantile(sz, [0, .1, .25, .5, .75, .9, 1]).round(1), "frac>=12.56", (sz >= 12.56).mean())
