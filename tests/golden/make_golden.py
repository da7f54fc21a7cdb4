#!/usr/bin/env python3
"""Regenerates the committed golden vectors from the CPU oracle.

The reference cannot run here (it needs OpenCV 2.4.2 + Boost 1.51) and ships no fixtures of its own, so these
vectors are outputs of oracle/mofreak_oracle.c, which is itself pinned by tests/test_oracle_kat.py.  They
freeze today's results: a later change to the oracle, the synthetic generator or the kernels that alters a
single byte shows up against them.

  golden_pair.npz     one 288x272 frame pair, 500 keypoints of mixed sizes spread over the whole frame
                      -> 16-byte descriptors + validity flags, in the three FREAK bit layouts
  golden_stream.npz   a 9-frame 176x144 stack with per-frame keypoint lists -> .mofreak rows (binary + text)
  golden_detector.npz a 160x120 moving-object frame pair and a tie-heavy 96x72 image -> BRISK keypoints (x, y, size,
                      response, layer) from oracle/brisk_oracle.c, and the rows of detector + descriptors on a 7-frame clip

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle_lib as O  # noqa: E402
from mofreak_amd import synth  # noqa: E402


def main():
    rng = np.random.default_rng(20261003)
    # ---- one pair, mixed sizes
    W, H = 288, 272
    fr = synth.synth_stack(6, W, H, t0=40)
    cur, prev = fr[5], fr[0]
    kps = synth.random_keypoints(rng, 500, W, H)
    kps[:8, 2] = np.float32([0.0, 1e-8, 6.5, 7.0, 9.99, 14.4, 36.0, 300.0])
    out = {"cur": cur, "prev": prev, "kps": kps}
    for name, mode in [("sse", O.BITS_SSE), ("natural", O.BITS_NATURAL), ("sse_signed", O.BITS_SSE_SIGNED)]:
        d, v = O.Freak(bit_mode=mode).extract_pair(cur, prev, kps)
        out[f"desc_{name}"] = d
        out[f"valid_{name}"] = v
    assert 0.25 < out["valid_sse"].mean() < 0.9
    np.savez_compressed(os.path.join(HERE, "golden_pair.npz"), **out)

    # ---- a stream with ragged keypoint lists
    W, H, T = 176, 144, 9
    fr = synth.synth_stack(T, W, H, t0=7)
    counts = [120, 0, 75, 200]
    offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    kps = synth.random_keypoints(rng, int(offs[-1]), W, H, sizes=(7.0, 8.4, 10.5, 12.0))
    kps[:, :2] = np.round(kps[:, :2] * 4) / 4  # quarter-pixel coordinates: exercises the %g text and the float adds
    rows = O.Freak().extract_stream(fr, kps, offs)
    text = O.format_rows(rows)
    np.savez_compressed(os.path.join(HERE, "golden_stream.npz"), frames=fr, kps=kps, kp_offsets=offs,
                        rows=rows.view(np.uint8).reshape(-1, 32), text=np.frombuffer(text, np.uint8))
    # ---- the detector
    clip = synth.moving_objects_stack(7, 160, 120, seed=77)
    diff = O.absdiff(clip[5], clip[0])
    kp_pair = O.brisk_detect(diff, 30, 3)
    small = rng.integers(0, 4, (24, 32)) * 85
    ties = np.kron(small, np.ones((3, 3), np.int64)).astype(np.uint8)  # 96 x 72, blocky four-level noise: many equal scores
    kp_ties = O.brisk_detect(ties, 30, 3)
    lists = [O.brisk_detect(O.absdiff(clip[t], clip[t - 5])) for t in (5, 6)]
    kk = np.concatenate([np.stack([k["x"], k["y"], k["size"]], 1) for k in lists]).astype(np.float32)
    clip_rows = O.Freak().extract_stream(clip, kk, np.int64([0, len(lists[0]), len(kk)]))
    assert len(kp_pair) > 50 and len(kp_ties) > 50 and len(clip_rows) > 5
    np.savez_compressed(os.path.join(HERE, "golden_detector.npz"), clip=clip, kp_pair=kp_pair.view(np.uint8).reshape(-1, 20),
                        ties=ties, kp_ties=kp_ties.view(np.uint8).reshape(-1, 20), clip_rows=clip_rows.view(np.uint8).reshape(-1, 32))
    print(f"golden_pair: {int(out['valid_sse'].sum())}/{len(out['kps'])} valid; golden_stream: {len(rows)} rows, {len(text)} text bytes; "
          f"golden_detector: {len(kp_pair)} + {len(kp_ties)} keypoints, {len(clip_rows)} rows")
    for f in ("golden_pair.npz", "golden_stream.npz", "golden_detector.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")


if __name__ == "__main__":
    main()
