#!/usr/bin/env python3
"""The whole frame loop (mofreak_compute_stream: BRISK keypoints on |frame - frame[-5]|, their descriptors, rows) on a
device-resident 1920x1080 moving-object stack, with and without the software pipelining of detector and descriptors.
usage: loop_probe.py [PAIRS ...]   (default 32 64 128; LOOP_W / LOOP_H: another frame size)"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

import mofreak_amd as M
from mofreak_amd import synth

W, H = int(os.environ.get("LOOP_W", "1920")), int(os.environ.get("LOOP_H", "1080"))
sizes = [int(a) for a in sys.argv[1:]] or [32, 64, 128]
distinct = 9
base = synth.moving_objects_stack(distinct, W, H)
with M.Context(0) as ctx:
    for pairs in sizes:
        T = pairs + 5
        fr = torch.from_numpy(np.stack([base[t % distinct] for t in range(T)])).cuda()
        rows = torch.empty(pairs * 12000 * 32, dtype=torch.uint8, device="cuda")
        out = {"pairs": pairs}
        digest = {}
        for mode in (True, False, True, False):
            ctx.set_loop_pipelining(mode)
            n_rows, n_kp = ctx.compute_stream(fr, T, W, H, rows, capacity=rows.numel() // 32)  # warm-up (buffers)
            ctx.synchronize()
            steps = max(2, 256 // pairs)
            t0 = time.perf_counter()
            for _ in range(steps):
                n_rows, n_kp = ctx.compute_stream(fr, T, W, H, rows, capacity=rows.numel() // 32)
            dt = (time.perf_counter() - t0) / steps
            key = "pipelined" if mode else "sequential"
            out[key + "_pairs_per_s"] = max(out.get(key + "_pairs_per_s", 0), pairs / dt)
            digest[key] = (n_rows, n_kp, int(rows[: n_rows * 32].to(torch.int64).sum().item()))
        assert digest["pipelined"] == digest["sequential"], digest
        out["rows"], out["keypoints_per_pair"] = digest["pipelined"][0], digest["pipelined"][1] / pairs
        out["speedup"] = out["pipelined_pairs_per_s"] / out["sequential_pairs_per_s"]
        print(json.dumps(out), flush=True)
