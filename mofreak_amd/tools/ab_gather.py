#!/usr/bin/env python3
"""A/B timing of gather-path builds on detector keypoints (1920x1080 moving-object pairs: most of the keypoints are too large
for the tile kernel).  Kernel experiments only.

  ab_gather.py LIB [LIB ...]   each library (a build of the same ABI) in a process of its own: the keypoints are detected once,
                               then the descriptor call is timed: the library's HIP-event times of the binning pass, the tile
                               kernel and the gather path (integral + describe) per call, and a checksum of all rows
  AB_PAIRS / AB_STEPS / AB_PATH=gather (everything through the gather path)
"""
from __future__ import annotations

import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def one(lib, pairs, steps):
    os.environ["MOFREAK_HIP_LIBRARY"] = lib
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch

    from mofreak_amd import api, synth

    W, H = 1920, 1080
    distinct = 4
    fr = synth.moving_objects_stack(5 + distinct, W, H)
    with api.Context(0) as ctx:
        cur = torch.from_numpy(np.stack([fr[5 + (p % distinct)] for p in range(pairs)])).cuda()
        prev = torch.from_numpy(np.stack([fr[p % distinct] for p in range(pairs)])).cuda()
        cap = 32768 * pairs
        kps = torch.empty((cap, 3), dtype=torch.float32, device="cuda")
        offs = torch.empty(pairs + 1, dtype=torch.int64, device="cuda")
        desc = torch.zeros((cap, 16), dtype=torch.uint8, device="cuda")
        valid = torch.zeros(cap, dtype=torch.uint8, device="cuda")
        n = ctx.detect_pairs(cur, prev, W, H, pairs, kps, offs, capacity=cap)
        band = int(os.environ.get("AB_SORT_BAND", "0"))
        if band:  # locality experiment: every pair's keypoints in bands of `band` rows (rows change order: compare times only)
            o = offs.cpu().numpy()
            pair = torch.from_numpy(np.repeat(np.arange(pairs), np.diff(o))).cuda()
            key = pair * 100000 + (kps[:n, 1] / band).floor().long() * 100 + (kps[:n, 0] / 64).floor().long()
            kps[:n] = kps[:n][torch.argsort(key, stable=True)]
        if os.environ.get("AB_PATH") == "gather":
            ctx.set_path(api.PATH_GATHER)
        for _ in range(2):
            ctx.extract_pairs(cur, prev, W, H, pairs, kps, desc, valid, kp_offsets=offs, n_kp=n)
        ctx.synchronize()
        ctx.set_profiling(True)
        ctx.get_profile(reset=True)
        for _ in range(steps):
            ctx.extract_pairs(cur, prev, W, H, pairs, kps, desc, valid, kp_offsets=offs, n_kp=n)
        prof = ctx.get_profile(reset=True)
        ctx.check_status()
        digest = hashlib.sha256(desc[:n].cpu().numpy().tobytes() + valid[:n].cpu().numpy().tobytes()).hexdigest()[:16]
        large = float((kps[:n, 2] >= 12.56).float().mean())
    c = prof["calls"]
    print(json.dumps({"lib": os.path.basename(lib), "pairs": pairs, "keypoints": int(n), "large": round(large, 3), "bin_ms": prof["bin_ms"] / c,
                      "tile_ms": prof["tile_ms"] / c, "gather_ms": prof["gather_ms"] / c, "sha": digest}), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--one":
        one(sys.argv[2], int(os.environ.get("AB_PAIRS", "32")), int(os.environ.get("AB_STEPS", "10")))
    else:
        for lib in sys.argv[1:]:
            for rep in range(int(os.environ.get("AB_REPS", "2"))):
                subprocess.run([sys.executable, os.path.abspath(__file__), "--one", os.path.abspath(lib)], check=False)
