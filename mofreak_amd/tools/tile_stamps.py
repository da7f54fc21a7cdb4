#!/usr/bin/env python3
"""Per-phase share of the tile kernel's time on the benchmark workload (diagnostic build: MOFREAK_TILE_STAMPS=1).

Thread 0 of every workgroup stamps s_memtime between phases; the sums over all workgroups are printed as shares.
Not part of the metric: the diagnostic instantiation is only launched when the context was created with the variable set.
"""
from __future__ import annotations

import argparse
import json
import os
import sys

os.environ["MOFREAK_TILE_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

import numpy as np  # noqa: E402

PHASES = ["stage0 gray tiles", "stage1 MIP", "stage2 row pass", "stage2 column sums", "stage2 column carries", "stage3 FREAK setup + pass A (wave 0)",
          "stage3 thetas", "stage3 pass B + waiting for the other waves"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C3")
    ap.add_argument("--pairs", type=int, default=64)
    args = ap.parse_args()
    import torch

    import mofreak_amd as M
    from mofreak_amd import synth

    cfg = synth.CONFIGS[args.config]
    W, H = cfg["W"], cfg["H"]
    kps = synth.config_grid(args.config)
    T = args.pairs + 5
    frames = np.stack([synth.synth_frame(t, W, H) for t in range(T)])
    with M.Context(0) as ctx:
        d_frames = torch.from_numpy(frames).cuda()
        d_kps = torch.from_numpy(kps).cuda()
        n = args.pairs * len(kps)
        desc = torch.empty((n, 16), dtype=torch.uint8, device="cuda")
        valid = torch.empty(n, dtype=torch.uint8, device="cuda")
        for _ in range(2):
            ctx.extract_pairs(d_frames[5:], d_frames[:args.pairs], W, H, args.pairs, d_kps, desc, valid)
        ctx.synchronize()
        ctx.get_tile_stamps(reset=True)
        ctx.extract_pairs(d_frames[5:], d_frames[:args.pairs], W, H, args.pairs, d_kps, desc, valid)
        ctx.synchronize()
        st = ctx.get_tile_stamps(reset=True).astype(np.float64)[:len(PHASES)]
    tot = st.sum()
    print(json.dumps({"config": args.config, "pairs": args.pairs, "ticks_per_descriptor": tot / n,
                      "shares": {p: round(float(v / tot), 4) for p, v in zip(PHASES, st)}}))


if __name__ == "__main__":
    main()
