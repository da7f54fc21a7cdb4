#!/usr/bin/env python3
"""A/B timing of tile-kernel builds on the metric's workload (C3: 1080p, 8-px grid).  Kernel experiments only -- the
metric is bench.py's, on the in-tree library.

  ab_tile.py LIB [LIB ...]      each library (a build of the same ABI) in a process of its own: average tile_kernel launch
                                time from the library's HIP events, and a checksum of all descriptors of a step (variants
                                must agree bit for bit)
  ab_tile.py --build NAME -DX   compile mofreak_amd/_exp/libvar_NAME.so from the tree's sources with extra flags
  ab_tile.py --ablate MASK      the stage ablation builds of profiles/: libvar_ablMASK.so from a COPY of tile_kernel.hip in which
                                the blocks marked <stage 1> (MIP, both forms), <stage 2> (integral), <stage 3> (FREAK) are disabled for the
                                bits set in MASK (1, 2, 4).  The product source has no switch for this; results are wrong by
                                construction (only the time is of interest)
"""
from __future__ import annotations

import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
EXP = os.path.join(ROOT, "mofreak_amd", "_exp")


def build(name, flags, patch=None):
    """patch: {source file name: function(text) -> text}; patched copies are compiled from _exp/src_NAME/."""
    sys.path.insert(0, ROOT)
    from mofreak_amd import build as B
    os.makedirs(EXP, exist_ok=True)
    out = os.path.join(EXP, f"libvar_{name}.so")
    srcs = [os.path.join(B.CSRC, s) for s in B.SOURCES]
    if patch:
        d = os.path.join(EXP, f"src_{name}")
        os.makedirs(d, exist_ok=True)
        for i, src in enumerate(srcs):
            base = os.path.basename(src)
            if base in patch:
                with open(src) as f:
                    text = patch[base](f.read())
                srcs[i] = os.path.join(d, base)
                with open(srcs[i], "w") as f:
                    f.write(text)
        flags = [*flags, "-I", B.CSRC]
    cmd = [B.hipcc(), *B.FLAGS, *flags, "-o", out, *srcs]
    subprocess.check_call(cmd)
    print(out)


def ablate(mask):
    def patch(text):
        for bit, tag in ((1, "<stage 1>"), (2, "<stage 2>"), (4, "<stage 3>")):
            if mask & bit:
                if bit == 1:  # the MIP: the lane-per-keypoint form beside stage 0 and the wave-per-keypoint stage
                    call = "const uint2 mv = mip_lane_keypoint<L, decltype(CM)::value>(cur, prev, roi, a.f.row_stride, mip_theta);"
                    assert text.count(call) == 1 and text.count("if (!mip_lane) {  // " + tag) == 1
                    text = text.replace(call, "const uint2 mv = make_uint2(roi, 0u);")
                    text = text.replace("if (!mip_lane) {  // " + tag, "if (false) {  // " + tag)
                    continue
                assert text.count("{  // " + tag) == 1, tag
                text = text.replace("{  // " + tag, "if (false) {  // " + tag)
        return text
    build(f"abl{mask}", [], {"tile_kernel.hip": patch})


def one(lib, pairs, steps, config):
    os.environ["MOFREAK_HIP_LIBRARY"] = lib
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch

    import mofreak_amd as M
    from mofreak_amd import synth

    cfg = synth.CONFIGS[config]
    W, H = cfg["W"], cfg["H"]
    kps = synth.config_grid(config)
    frames = np.stack([synth.synth_frame(t, W, H) for t in range(pairs + 5)])
    with M.Context(0) as ctx:
        d_fr, d_kps = torch.from_numpy(frames).cuda(), torch.from_numpy(kps).cuda()
        n = pairs * len(kps)
        desc = torch.empty((n, 16), dtype=torch.uint8, device="cuda")
        valid = torch.empty(n, dtype=torch.uint8, device="cuda")
        ctx.reserve(W, H)
        for _ in range(3):
            ctx.extract_pairs(d_fr[5:], d_fr[:pairs], W, H, pairs, d_kps, desc, valid)
        ctx.synchronize()
        ctx.set_profiling(True)
        ctx.get_profile(reset=True)
        for _ in range(steps):
            ctx.extract_pairs(d_fr[5:], d_fr[:pairs], W, H, pairs, d_kps, desc, valid)
        prof = ctx.get_profile(reset=True)
        ctx.check_status()
        digest = hashlib.sha256(desc.cpu().numpy().tobytes() + valid.cpu().numpy().tobytes()).hexdigest()[:16]
    print(json.dumps({"lib": os.path.basename(lib), "tile_ms": prof["tile_ms"] / prof["calls"], "bin_ms": prof["bin_ms"] / prof["calls"],
                      "pairs": pairs, "Gdesc_per_s": n / (prof["tile_ms"] / prof["calls"]) / 1e6, "sha": digest}), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--build":
        build(sys.argv[2], sys.argv[3:])
    elif len(sys.argv) > 2 and sys.argv[1] == "--ablate":
        ablate(int(sys.argv[2]))
    elif len(sys.argv) > 2 and sys.argv[1] == "--one":
        one(sys.argv[2], int(os.environ.get("AB_PAIRS", "256")), int(os.environ.get("AB_STEPS", "20")), os.environ.get("AB_CONFIG", "C3"))
    else:
        for lib in sys.argv[1:]:
            for rep in range(int(os.environ.get("AB_REPS", "2"))):  # interleaved repeats: the clock drifts with load
                subprocess.run([sys.executable, os.path.abspath(__file__), "--one", os.path.abspath(lib)], check=False)
