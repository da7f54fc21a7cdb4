#!/usr/bin/env python3
"""Micro-benchmark of the frame-preparation kernel (BGR -> gray): an HBM-bound streaming kernel, 4 bytes moved per pixel."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mofreak_amd as M  # noqa: E402


def main(W=1920, H=1080, n_frames=256, steps=20):
    ctx = M.Context(0)
    bgr = torch.randint(0, 256, (n_frames, H, W, 3), dtype=torch.uint8, device="cuda")
    gray = torch.empty((n_frames, H, W), dtype=torch.uint8, device="cuda")
    s = torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(s):
        ctx.set_stream(s.cuda_stream)
        for _ in range(3):
            ctx.bgr_to_gray(bgr, W, H, n_frames, gray)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            ctx.bgr_to_gray(bgr, W, H, n_frames, gray)
        e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    nbytes = 4 * W * H * n_frames
    print(json.dumps({"kernel": "bgr2gray_kernel", "frames": n_frames, "W": W, "H": H, "ms": ms, "GBps": nbytes / ms / 1e6,
                      "frac_of_8TBps": nbytes / ms / 1e6 / 8000, "frames_per_s": n_frames / ms * 1e3}))
    ctx.set_stream(None)
    ctx.close()


if __name__ == "__main__":
    main()
