#!/usr/bin/env python3
"""Randomised parity run, GPU against the oracle (a tool, not part of the test suites: `python tests/fuzz_parity_gpu.py
[seconds] [seed] [detector]` on a GPU box; `detector`: detector rounds only, several pairs a call; a mismatch prints the
frame size, octaves and frame seed that replay it: `python tests/fuzz_parity_gpu.py replay W H octaves frame_seed`).  Every round draws a frame size, a keypoint population (counts, sizes, integer or
fractional coordinates, shared list or one list per pair), random byte frames or the synthetic ones, and compares
descriptors and validity flags byte for byte; every few rounds also the detector's keypoints on a moving-object pair,
a set of clips of random lengths through mofreak_extract_clips (against one call per clip), and the whole frame loop
in one call (mofreak_compute_stream) against the detector and the descriptors in two; the clip sets' rows as text made on
the device against the host formatter; the frame loop's stack cut into clips through mofreak_compute_clips and pushed in
chunks into a detector stream.
A new context every 20 s, a third of them with random FREAK parameters (bit mode, orientation / scale normalisation)."""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

import mofreak_amd as M  # noqa: E402
import oracle_lib as O  # noqa: E402
from mofreak_amd import synth  # noqa: E402


def detector_case(ctx, Wd, Hd, octaves, frame_seed, n_pairs=1):
    """The detector on n_pairs moving-object pairs against the oracle; None, or what differs."""
    fr = synth.moving_objects_stack(5 + n_pairs, Wd, Hd, seed=frame_seed)
    k, offs, resp, layer = ctx.detect_pairs_host(fr[5:], fr[:n_pairs], 30, octaves)
    for p in range(n_pairs):
        want = O.brisk_detect(O.absdiff(fr[5 + p], fr[p]), 30, octaves)
        wk = np.stack([want["x"], want["y"], want["size"]], 1).astype(np.float32).reshape(-1, 3)
        kp, rp = k[offs[p]:offs[p + 1]], resp[offs[p]:offs[p + 1]]
        if not (len(kp) == len(wk) and kp.tobytes() == wk.tobytes() and rp.tobytes() == want["response"].tobytes()):
            return f"{Wd}x{Hd} octaves {octaves} frame seed {frame_seed} pairs {n_pairs}, pair {p}: {len(kp)} vs {len(wk)} keypoints"
    return None


def detector_only(budget, seed):
    """Detector calls only, as the full tool makes them: a new context every 150 calls (fresh device buffers), other calls of
    other frame sizes in between (the staging buffers move), one to three pairs a call."""
    rng = np.random.default_rng(seed)
    t_end, runs, t_say = time.time() + budget, 0, time.time()
    while time.time() < t_end:
        with M.Context(0) as ctx:
            for _ in range(150):
                if time.time() >= t_end:
                    break
                if rng.integers(0, 2):
                    W, H = int(rng.choice([97, 160, 320, 640, 1024])), int(rng.choice([80, 121, 240, 480]))
                    frames = rng.integers(0, 256, (6, H, W), dtype=np.uint8)
                    ctx.extract_pairs_host(frames[5:], frames[:1], synth.random_keypoints(rng, 60, W, H, sizes=(8.4, 12.0, 27.0)))
                Wd, Hd = int(rng.choice([160, 320, 481, 640])), int(rng.choice([120, 240, 360]))
                case = (Wd, Hd, int(rng.integers(0, 4)), int(rng.integers(0, 1 << 30)), int(rng.integers(1, 4)))
                bad = detector_case(ctx, *case)
                if bad:
                    print(f"DETECTOR MISMATCH run {runs} seed {seed}: {bad}")
                    again = [detector_case(ctx, *case) is not None for _ in range(5)]
                    with M.Context(0) as fresh:
                        anew = [detector_case(fresh, *case) is not None for _ in range(5)]
                    print(f"  replayed: same context {again}, fresh context {anew} (True = differs again)")
                    sys.exit(1)
                runs += 1
                if time.time() - t_say > 30:
                    t_say = time.time()
                    print(f"... {runs} detector runs", flush=True)
    print(f"fuzz ok: {runs} detector runs, seed {seed}")


def main():
    if len(sys.argv) > 5 and sys.argv[1] == "replay":
        with M.Context(0) as ctx:
            for n_pairs in (1, 2, 3):
                print(n_pairs, detector_case(ctx, int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), n_pairs) or "equal")
        return
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    if len(sys.argv) > 3 and sys.argv[3] == "detector":
        return detector_only(budget, seed)
    rng = np.random.default_rng(seed)
    t_end = time.time() + budget
    rounds = descriptors = detector_rounds = clip_rounds = loop_rounds = 0
    while time.time() < t_end:
        # a context (and an oracle) per parameter set: the default one most of the time
        if rng.integers(0, 3) == 0:
            par = dict(freak_bit_mode=int(rng.integers(0, 3)), freak_orientation_normalized=int(rng.integers(0, 2)),
                       freak_scale_normalized=int(rng.integers(0, 2)))
        else:
            par = {}
        f = O.Freak(orientation_normalized=bool(par.get("freak_orientation_normalized", 1)),
                    scale_normalized=bool(par.get("freak_scale_normalized", 1)), bit_mode=par.get("freak_bit_mode", O.BITS_SSE))
        t_ctx = min(t_end, time.time() + 20)
        print(f"... {rounds} rounds, {int(t_end - time.time())} s to go", flush=True)  # (a run that says nothing for minutes looks hung)
        with M.Context(0, **par) as ctx:
          while time.time() < t_ctx:
              W = int(rng.choice([97, 160, 203, 320, 417, 640, 731, 1024, 1283]))
              H = int(rng.choice([80, 121, 240, 301, 480, 577]))
              n_pairs = int(rng.integers(1, 4))
              n_kp = int(rng.choice([1, 7, 60, 400, 2500, 9000]))
              sizes = [(7.0, 9.0, 12.0), (8.4, 12.0, 18.0, 27.0, 40.5), (12.0,), (12.5, 12.6, 13.3, 14.9, 15.2, 22.7, 31.0, 55.5, 71.9), (6.9, 100.0)][int(rng.integers(0, 5))]
              integer_xy = bool(rng.integers(0, 2))
              if rng.integers(0, 2):
                  frames = rng.integers(0, 256, (n_pairs + 5, H, W), dtype=np.uint8)
              else:
                  frames = synth.synth_stack(n_pairs + 5, W, H)
              kps = synth.random_keypoints(rng, n_kp, W, H, sizes=sizes, integer_xy=integer_xy)
              desc, valid = ctx.extract_pairs_host(frames[5:], frames[:n_pairs], kps)
              ctx.check_status()  # (the bounds-checking library reports an access outside its limits here)
              for p in range(n_pairs):
                  d, v = f.extract_pair(frames[5 + p], frames[p], kps)
                  got_d, got_v = desc[p * n_kp:(p + 1) * n_kp], valid[p * n_kp:(p + 1) * n_kp]
                  if not (np.array_equal(got_v, v) and np.array_equal(got_d, d)):
                      bad = np.nonzero((got_d != d).any(1) | (got_v != v))[0]
                      print(f"MISMATCH round {rounds} seed {seed}: {W}x{H} pairs {n_pairs} kp {n_kp} sizes {sizes} integer {integer_xy}: "
                            f"{len(bad)} keypoints, first {bad[:5]}, kp {kps[bad[0]]}")
                      sys.exit(1)
              rounds += 1
              descriptors += n_pairs * n_kp
              if rounds % 5 == 0:
                  Wd, Hd = int(rng.choice([160, 320, 481, 640])), int(rng.choice([120, 240, 360]))
                  d_oct, d_seed = int(rng.integers(0, 4)), int(rng.integers(0, 1 << 30))
                  bad = detector_case(ctx, Wd, Hd, d_oct, d_seed)
                  if bad:
                      print(f"DETECTOR MISMATCH round {rounds} seed {seed}: {bad}")
                      # the same case again: in this context, then in a fresh one (is it the input, or the moment?)
                      again = [detector_case(ctx, Wd, Hd, d_oct, d_seed) is not None for _ in range(5)]
                      with M.Context(0) as fresh:
                          anew = [detector_case(fresh, Wd, Hd, d_oct, d_seed) is not None for _ in range(5)]
                      print(f"  replayed: same context {again}, fresh context {anew} (True = differs again)")
                      sys.exit(1)
                  detector_rounds += 1
              if rounds % 7 == 0 and not par:  # many clips in one pipelined call == one call per clip
                  Wc, Hc = int(rng.choice([96, 160, 320])), int(rng.choice([80, 120, 240]))
                  lengths = [int(x) for x in rng.choice([0, 1, 5, 6, 7, 9, 14, 23, 40], int(rng.integers(1, 9)))]
                  pool = rng.integers(0, 256, (max(lengths) + 3, Hc, Wc), dtype=np.uint8)
                  clips = [np.ascontiguousarray(pool[i % 3: i % 3 + t]) for i, t in enumerate(lengths)]
                  ck = synth.random_keypoints(rng, int(rng.choice([3, 150, 700])), Wc, Hc, sizes=(7.0, 8.4, 12.0))
                  rows_c, offs_c = ctx.extract_clips(clips, ck, chunk_frames=int(rng.choice([0, 6, 11, 30])))
                  want_c = [ctx.extract_stream_host(c, ck) if len(c) else np.zeros(0, M.api.ROW_DTYPE) for c in clips]
                  if rows_c.tobytes() != (np.concatenate(want_c).tobytes() if want_c else b"") or \
                          offs_c.tolist() != np.concatenate([[0], np.cumsum([len(w) for w in want_c])]).tolist():
                      print(f"CLIPS MISMATCH round {rounds} seed {seed}: {Wc}x{Hc} lengths {lengths}")
                      sys.exit(1)
                  if len(rows_c):  # the rows' text made on the device == the host formatter's, with a segment per clip
                      import torch
                      d_rows = torch.from_numpy(rows_c.view(np.uint8).reshape(-1).copy()).cuda()
                      text, offs_t = ctx.format_rows_device(d_rows, len(rows_c), row_starts=offs_c[:-1])
                      if text[:offs_t[-1]].tobytes() != M.format_rows(rows_c) or any(offs_t[i] != len(M.format_rows(rows_c[:offs_c[i]])) for i in range(len(offs_c) - 1)):
                          print(f"TEXT MISMATCH round {rounds} seed {seed}: {Wc}x{Hc} lengths {lengths}")
                          sys.exit(1)
                  clip_rounds += 1
              if rounds % 11 == 0 and not par:  # the frame loop in one call == detector, then descriptors
                  Wl, Hl, Tl = int(rng.choice([160, 320, 400])), int(rng.choice([120, 240])), int(rng.choice([21, 26, 38, 70]))
                  frl = synth.moving_objects_stack(Tl, Wl, Hl, seed=int(rng.integers(0, 1 << 30)))
                  two = ctx.compute_stream_host(frl)
                  kl, ol, _, _ = ctx.detect_pairs_host(frl[5:], frl[:-5])
                  one = ctx.extract_stream_host(frl, kl, kp_offsets=ol)
                  if two.tobytes() != one.tobytes():
                      print(f"FRAME LOOP MISMATCH round {rounds} seed {seed}: {Wl}x{Hl} T {Tl}: {len(two)} vs {len(one)} rows")
                      sys.exit(1)
                  # the detector inside the pipelined routes: the stack cut into clips through mofreak_compute_clips == one frame loop
                  # per clip; the whole stack pushed in chunks into a detector stream == the frame loop
                  cuts = sorted(set(int(x) for x in rng.integers(0, Tl + 1, 3)) | {0, Tl})
                  pieces = [np.ascontiguousarray(frl[a:b]) for a, b in zip(cuts[:-1], cuts[1:])]
                  rows_p, offs_p, _ = ctx.compute_clips(pieces, chunk_frames=int(rng.choice([0, 7, 12])))
                  want_p = [ctx.compute_stream_host(c) for c in pieces]
                  if rows_p.tobytes() != np.concatenate(want_p).tobytes() or offs_p.tolist() != np.concatenate([[0], np.cumsum([len(w) for w in want_p])]).tolist():
                      print(f"DETECTOR CLIPS MISMATCH round {rounds} seed {seed}: {Wl}x{Hl} T {Tl} cuts {cuts}")
                      sys.exit(1)
                  with ctx.open_stream(Wl, Hl, use_detector=True) as st:
                      parts = [st.push_frames(frl[a:b], None, chunk_frames=int(rng.choice([0, 9]))) for a, b in zip(cuts[:-1], cuts[1:]) if b > a]
                  if np.concatenate(parts).tobytes() != two.tobytes():
                      print(f"DETECTOR STREAM MISMATCH round {rounds} seed {seed}: {Wl}x{Hl} T {Tl} cuts {cuts}")
                      sys.exit(1)
                  loop_rounds += 1
    print(f"fuzz ok: {rounds} rounds, {descriptors} descriptors, {detector_rounds} detector rounds, {clip_rounds} clip-set rounds, "
          f"{loop_rounds} frame-loop rounds, seed {seed}")


if __name__ == "__main__":
    main()
