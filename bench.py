#!/usr/bin/env python3
"""bench.py -- MoFREAK descriptors/sec on dense 1080p frames (BASELINE.json config 3) on N MI355X.

A "step" is one pass of the hot path over the resident batch: --pairs frame pairs of a synthetic
1920x1080 stack (pair i = frames i+5 and i), the dense 8-px grid of 29 106 size-12 keypoints in each pair,
inputs already in HBM, through the C ABI (mofreak_extract_pairs, device pointers).  N > 1: one process per
GPU, every rank its own stack (weak scaling), no data-path collective; the final row gather over RCCL is
timed separately and reported as gather_ms.

Prints ONE JSON line (rank 0).  See DESIGN.md section "Measurement" for how roofline.achieved is defined.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md


def make_stack(T, W, H, t0, workers):
    from mofreak_amd import synth
    out = np.empty((T, H, W), np.uint8)

    def one(t):
        out[t] = synth.synth_frame(t0 + t, W, H)

    with ThreadPoolExecutor(workers) as ex:  # numpy releases the GIL in the heavy ufuncs
        list(ex.map(one, range(T)))
    return out


def cpu_baseline(frames, kps, n_pairs, cores):
    """The CPU oracle (a port, not the reference binary) on a bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    f = oracle_lib.Freak()
    f.extract_pair(frames[5], frames[0], kps[:64])  # warm the library

    def one(p):
        d, v = f.extract_pair(frames[p + 5], frames[p], kps)  # ctypes releases the GIL
        return int(v.sum())

    t = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        n = sum(ex.map(one, range(n_pairs)))
    dt = time.perf_counter() - t
    return {"value": n / dt, "unit": "descriptors/s", "cores": cores, "kind": "port",
            "sample": f"{n_pairs} of the workload's 1920x1080 pairs x {len(kps)} keypoints, oracle/mofreak_oracle.c "
                      f"(gcc -O2, strict FP), {cores} threads over pairs, {dt:.1f} s wall"}


def detector_figures(ctx, torch, synth, W, H, pairs=32, steps=3, with_cpu=True):
    """BRISK keypoints on |cur - prev| of moving-object frames, and detector + descriptors back to back, per frame pair."""
    distinct = 4
    fr = synth.moving_objects_stack(5 + distinct, W, H)
    cur = torch.from_numpy(np.stack([fr[5 + (p % distinct)] for p in range(pairs)])).cuda()
    prev = torch.from_numpy(np.stack([fr[p % distinct] for p in range(pairs)])).cuda()
    cap = 32768 * pairs
    kps = torch.empty((cap, 3), dtype=torch.float32, device="cuda")
    offs = torch.empty(pairs + 1, dtype=torch.int64, device="cuda")
    desc = torch.empty((cap, 16), dtype=torch.uint8, device="cuda")
    valid = torch.empty(cap, dtype=torch.uint8, device="cuda")
    n = ctx.detect_pairs(cur, prev, W, H, pairs, kps, offs, capacity=cap)
    ctx.extract_pairs(cur, prev, W, H, pairs, kps, desc, valid, kp_offsets=offs, n_kp=n)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        n = ctx.detect_pairs(cur, prev, W, H, pairs, kps, offs, capacity=cap)  # synchronises
    t1 = time.perf_counter()
    for _ in range(steps):
        n = ctx.detect_pairs(cur, prev, W, H, pairs, kps, offs, capacity=cap)
        ctx.extract_pairs(cur, prev, W, H, pairs, kps, desc, valid, kp_offsets=offs, n_kp=n)
    ctx.synchronize()
    t2 = time.perf_counter()
    out = {"workload": f"{W}x{H} moving-object frame pairs, BriskFeatureDetector(30, 3 octaves) on |cur - prev|, {pairs} pairs/call",
           "keypoints_per_pair": n / pairs, "pairs_per_s": pairs * steps / (t1 - t0),
           "detect_and_describe_pairs_per_s": pairs * steps / (t2 - t1)}
    if with_cpu:  # part of the cpu_baseline leg: the oracle as a reported baseline, never as the product
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib
        a, b = fr[5], fr[0]
        t0 = time.perf_counter()
        k = oracle_lib.brisk_detect(oracle_lib.absdiff(a, b))
        out["cpu_baseline"] = {"value": 1.0 / (time.perf_counter() - t0), "unit": "pairs/s", "cores": 1, "kind": "port",
                               "sample": f"1 of the pairs, oracle/brisk_oracle.c, {len(k)} keypoints"}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--pairs", type=int, default=256, help="resident frame pairs per GPU")
    ap.add_argument("--config", default="C3", help="synthetic config (C3 = the metric's: 1080p, 8-px grid)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-pairs", type=int, default=192, help="pairs of the workload the CPU oracle is timed on (about 25 core-seconds)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL); 'gloo' + "
                    "--share-device rehearses the N > 1 control flow on a one-GPU box")
    ap.add_argument("--share-device", action="store_true", help="all ranks use cuda:0 (rehearsal only)")
    ap.add_argument("--no-detector", action="store_true", help="skip the (untimed-region) keypoint detector figures")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import mofreak_amd as M
    from mofreak_amd import harness, synth

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world} (launch with torch.distributed.run)"
    if args.share_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    on_device = args.backend == "nccl"  # gloo moves its (small) control tensors and the gathered rows through the host
    if world > 1:
        if on_device:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)

    cfg = synth.CONFIGS[args.config]
    W, H = cfg["W"], cfg["H"]
    kps = synth.config_grid(args.config)
    n_kp, n_pairs, gap = len(kps), args.pairs, synth.GAP_FOR_FRAME_DIFFERENCE
    T = n_pairs + gap
    ncpu = len(os.sched_getaffinity(0))
    workers = max(1, min(16, ncpu // max(1, world)))
    frames = make_stack(T, W, H, t0=1000 * rank, workers=workers)

    ctx = M.Context(local_rank)
    d_frames = torch.from_numpy(frames).cuda()
    d_kps = torch.from_numpy(kps).cuda()
    n_desc = n_pairs * n_kp
    desc = torch.empty((n_desc, 16), dtype=torch.uint8, device="cuda")
    valid = torch.empty(n_desc, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    stream = torch.cuda.Stream()  # a real (non-null) stream shared by torch and the library
    torch.cuda.set_stream(stream)
    ctx.set_stream(stream.cuda_stream)
    ctx.reserve(W, H)

    def step():
        ctx.extract_pairs(d_frames[gap:], d_frames[:n_pairs], W, H, n_pairs, d_kps, desc, valid)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    ctx.set_profiling(True)  # HIP events around every kernel group, on the stream the kernels run on
    ctx.get_profile(reset=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    prof = ctx.get_profile(reset=True)
    ctx.set_profiling(False)
    ctx.check_status()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if on_device else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    n_valid = int(valid.sum().item())
    if os.environ.get("MOFREAK_BENCH_ABLATION") != "1":  # ablation builds of the kernel skip stages on purpose
        assert n_valid == n_desc, f"{n_desc - n_valid} keypoints were erased: the grid is supposed to be border-safe"

    # the path's one exchange step: gather the compacted 32-byte rows to rank 0 (not part of a step)
    rows = torch.empty(n_desc * 32, dtype=torch.uint8, device="cuda")
    n_rows = ctx.compact_rows(d_kps, n_pairs, gap - 1, desc, valid, rows)
    gather_ms = None
    if world > 1:
        fence()
        tg = time.perf_counter()
        allrows, counts = harness.gather_rows(rows if on_device else rows.cpu(), n_rows, dst=0)
        fence()
        gather_ms = (time.perf_counter() - tg) * 1e3
        if rank == 0:
            assert sum(counts) == world * n_rows and allrows.numel() == sum(counts) * 32

    if rank == 0:
        total_desc = world * n_desc * args.steps
        value = total_desc / elapsed
        b_alg_pair = 2 * W * H + 28 * n_kp  # frames read once + keypoints in + descriptors out
        launches = max(prof["calls"], 1)  # one tile_kernel launch per extract call (n_pairs <= 32768)
        tile_ms_avg = prof["tile_ms"] / launches
        pairs_per_launch = prof["pairs"] / launches
        achieved = b_alg_pair * pairs_per_launch / (tile_ms_avg * 1e-3) / 1e9
        all_ms = prof["tile_ms"] + prof["bin_ms"] + prof["gather_ms"]
        pipeline_gbs = b_alg_pair * prof["pairs"] / (all_ms * 1e-3) / 1e9
        traffic = None
        valu = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            traffic = tj.get("tile_kernel_hbm_bytes_per_launch")
            if tj.get("valu_wave_instr_per_descriptor"):
                # what actually bounds the kernel: vector instruction issue.  Instructions per descriptor from the PMC pass
                # (profiles/), issue rate of a wave64 VALU instruction measured on this part; duration measured live.
                n_cus = torch.cuda.get_device_properties(local_rank).multi_processor_count
                clk = 2.4e9
                issued = tj["valu_wave_instr_per_descriptor"] * n_desc / (tile_ms_avg * 1e-3)
                peak_issue = n_cus * 4 * clk / tj.get("valu_cycles_per_wave_instr", 2.0)
                valu = {"achieved_wave_instr_per_s": issued, "peak_wave_instr_per_s": peak_issue, "frac": issued / peak_issue,
                        "wave_instr_per_descriptor": tj["valu_wave_instr_per_descriptor"], "clock_hz_assumed": clk}
        out = {
            "metric": "MoFREAK descriptors/sec on dense 1080p frames; achieved HBM GB/s vs peak",
            "value": value, "unit": "descriptors/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"{args.config}: {n_pairs} resident {W}x{H} frame pairs per GPU, dense {cfg['step']}-px grid, "
                                   f"{n_kp} keypoints/pair of size {cfg['size']}, 16-byte descriptors",
                       "descriptors_per_step_per_gpu": n_desc, "bit_mode": "SSE", "parallelism": f"one stack per GPU x{world}"},
            "roofline": {"bound": "hbm", "kernel": "tile_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_pair": b_alg_pair, "pairs_per_launch": pairs_per_launch,
                         "avg_launch_ms": tile_ms_avg, "launches_timed": prof["calls"],
                         "binning_avg_ms": prof["bin_ms"] / launches, "gather_path_avg_ms": prof["gather_ms"] / launches,
                         "pipeline_achieved_GBs": pipeline_gbs, "pipeline_frac": pipeline_gbs / HBM_PEAK_GBS,
                         "valu_issue": valu},
            "gather_ms": gather_ms,
        }
        if world == 1 and not args.no_cpu_baseline:
            cp = min(args.cpu_pairs, n_pairs)
            out["cpu_baseline"] = cpu_baseline(frames, kps, cp, cores=min(ncpu, 16))
        if world == 1 and not args.no_detector:
            # Outside the metric and its timed region: the row in front of the path, for the record.
            try:
                out["detector"] = detector_figures(ctx, torch, synth, W, H, with_cpu=not args.no_cpu_baseline)
            except Exception as e:  # never let the side figure take the metric line down
                out["detector"] = {"error": repr(e)}
        print(json.dumps(out), flush=True)
    ctx.set_stream(None)
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
