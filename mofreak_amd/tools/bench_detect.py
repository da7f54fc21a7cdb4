#!/usr/bin/env python3
"""Micro-benchmark of the keypoint detector (mofreak_detect_pairs) on 1920x1080 frame pairs of moving objects, and of
detector + descriptors back to back (what one frame of computeMoFREAKFromFile costs, MoFREAKUtilities.cpp:413-470)."""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mofreak_amd as M  # noqa: E402
from mofreak_amd import synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=64)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    a = ap.parse_args()
    W, H = a.width, a.height
    distinct = 4
    fr = synth.moving_objects_stack(5 + distinct, W, H)
    cur = np.stack([fr[5 + (p % distinct)] for p in range(a.pairs)])
    prev = np.stack([fr[p % distinct] for p in range(a.pairs)])
    ctx = M.Context(0)
    d_cur, d_prev = torch.from_numpy(cur).cuda(), torch.from_numpy(prev).cuda()
    cap = 20000 * a.pairs
    kps = torch.empty((cap, 3), dtype=torch.float32, device="cuda")
    offs = torch.empty(a.pairs + 1, dtype=torch.int64, device="cuda")
    desc = torch.empty((cap, 16), dtype=torch.uint8, device="cuda")
    valid = torch.empty(cap, dtype=torch.uint8, device="cuda")
    s = torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(s):
        ctx.set_stream(s.cuda_stream)
        n = ctx.detect_pairs(d_cur, d_prev, W, H, a.pairs, kps, offs, capacity=cap)
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record()
        for _ in range(a.steps):
            n = ctx.detect_pairs(d_cur, d_prev, W, H, a.pairs, kps, offs, capacity=cap)
        e[1].record()
        for _ in range(a.steps):
            n = ctx.detect_pairs(d_cur, d_prev, W, H, a.pairs, kps, offs, capacity=cap)
            ctx.extract_pairs(d_cur, d_prev, W, H, a.pairs, kps, desc, valid, kp_offsets=offs, n_kp=n)
        e[2].record()
        e[2].synchronize()
    det_ms = e[0].elapsed_time(e[1]) / a.steps
    both_ms = e[1].elapsed_time(e[2]) / a.steps
    out = {"workload": f"{W}x{H} moving objects, {a.pairs} pairs/call", "keypoints_per_pair": n / a.pairs,
           "detect_ms_per_call": det_ms, "detect_pairs_per_s": a.pairs / det_ms * 1e3,
           "detect_pixels_per_s": a.pairs * W * H / det_ms * 1e3,
           "detect_and_describe_ms_per_call": both_ms, "frames_per_s_detect_and_describe": a.pairs / both_ms * 1e3,
           "valid_descriptors_per_pair": float(valid[:n].sum().item()) / a.pairs}
    print(json.dumps(out))
    ctx.set_stream(None)
    ctx.close()


if __name__ == "__main__":
    main()
