#!/usr/bin/env python3
"""bench.py -- MoFREAK descriptors/sec on dense 1080p frames (BASELINE.json config 3) on N MI355X.

A "step" is one pass of the hot path over the resident batch: --pairs frame pairs of a synthetic
1920x1080 stack (pair i = frames i+5 and i), the dense 8-px grid of 29 106 size-12 keypoints in each pair,
inputs already in HBM, through the C ABI (mofreak_extract_pairs, device pointers).  N > 1: one process per
GPU, every rank its own stack (weak scaling), no data-path collective; the final row gather over RCCL is
timed separately and reported as gather_ms.

Prints ONE JSON line (rank 0).  See DESIGN.md section "Measurement" for how roofline.achieved is defined.

Other workloads of BASELINE.json (each prints its own single JSON line; the metric's line is the default one):
  --config C2                 640x480, 16-px grid, 1000 resident pairs (the bit-exactness configuration), same line shape
  --config C4 [--gpus N]      HMDB51-shaped batch: clips of uneven length, one video per GPU at a time, rows gathered to
                              rank 0 (harness.run_dataset): clips/s, descriptors/s, gather_ms
  --config C5 [--frames N]    TRECVID-shaped 720x576 stream from page-locked host memory through the pipelined frame
                              loop: sustained descriptors/s INCLUDING the host-to-device copies and the rows' way back;
                              --frames 90000 (the hour-long stream): pushed in chunks of 2048 frames, bounded memory
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import threading
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from mofreak_amd import launch  # noqa: E402  (no GPU call, no torch: safe in the parent of the rank processes)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
PCIE_GBS = 63.0        # host link, same guide
METRIC = "MoFREAK descriptors/sec on dense 1080p frames; achieved HBM GB/s vs peak"


_JSON_FD = None  # the process's real standard output (see quiet_stdout)


def quiet_stdout():
    """The job's standard output carries ONE JSON line.  Native libraries print there too (RCCL's version banner at
    communicator creation, for one): from here on file descriptor 1 is standard error, and emit() writes the line to the
    descriptor that was standard output when the process started."""
    global _JSON_FD
    if _JSON_FD is None:
        sys.stdout.flush()
        _JSON_FD = os.dup(1)
        os.dup2(2, 1)


def emit(obj):
    line = (json.dumps(obj) + "\n").encode()
    if _JSON_FD is None:
        sys.stdout.write(line.decode())
        sys.stdout.flush()
    else:
        os.write(_JSON_FD, line)


def make_stack(T, W, H, t0, workers):
    from mofreak_amd import synth
    out = np.empty((T, H, W), np.uint8)

    def one(t):
        out[t] = synth.synth_frame(t0 + t, W, H)

    with ThreadPoolExecutor(workers) as ex:  # numpy releases the GIL in the heavy ufuncs
        list(ex.map(one, range(T)))
    return out


class Verifier:
    """Checks the descriptors the timed steps left in HBM against the CPU oracle, pair by pair: the bench line proves
    its bytes (`verified`), not only its flags.  The oracle is the checker here, never the thing measured."""

    def __init__(self, desc, valid, n_kp):
        self.desc, self.valid, self.n_kp = desc, valid, n_kp  # device tensors of the last timed step
        self.pairs = self.descriptors = self.mismatches = 0
        self.host = {}
        self.lock = threading.Lock()  # check() is called from the baseline's worker threads

    def fetch(self, pairs):
        """One device-to-host copy per pair that will be compared."""
        for p in pairs:
            sl = slice(p * self.n_kp, (p + 1) * self.n_kp)
            self.host[p] = (self.desc[sl].cpu().numpy(), self.valid[sl].cpu().numpy())

    def check(self, p, want_desc, want_valid):
        got_d, got_v = self.host.pop(p)
        bad = int((got_v != want_valid).sum()) + int((got_d != want_desc).any(axis=1).sum())
        with self.lock:
            self.pairs += 1
            self.descriptors += len(want_valid)
            self.mismatches += bad

    def report(self):
        return {"pairs": self.pairs, "descriptors": self.descriptors, "mismatches": self.mismatches,
                "against": "oracle/mofreak_oracle.c on the same frames and keypoints, 16 bytes + validity flag per keypoint"}


def verify_sample(ver, frames, kps, pairs):
    """No cpu_baseline leg (N > 1, --no-cpu-baseline): the oracle on a few pairs of this rank's stack."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    f = oracle_lib.Freak()
    ver.fetch(pairs)
    for p in pairs:
        d, v = f.extract_pair(frames[p + 5], frames[p], kps)
        ver.check(p, d, v)


def cpu_baselines(frames, kps, n_pairs, cores, shape, ver=None):
    """The CPU oracle (a port, not the reference binary) on bounded samples of the same workload: all host cores, one
    thread, and one thread with the FREAK pattern tables rebuilt for every frame pair as the reference does by
    constructing cv::FREAK inside its frame loop (MoFREAKUtilities.cpp:427).  The descriptors of the first leg are
    not thrown away: `ver` compares each pair's with what the GPU wrote for it."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    f = oracle_lib.Freak()
    f.extract_pair(frames[5], frames[0], kps[:64])  # warm the library
    if ver is not None:
        ver.fetch(range(n_pairs))
    first_leg = [True]

    def one(p):
        d, v = f.extract_pair(frames[p + 5], frames[p], kps)  # ctypes releases the GIL
        if ver is not None and first_leg[0]:
            ver.check(p, d, v)
        return int(v.sum())

    def one_ref(p):
        g = oracle_lib.Freak()  # pattern LUT, orientation weights, pair tables: rebuilt per pair
        d, v = g.extract_pair(frames[p + 5], frames[p], kps)
        return int(v.sum())

    what = f"the workload's {shape} pairs x {len(kps)} keypoints, oracle/mofreak_oracle.c (gcc -O2, strict FP)"
    t = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        n = sum(ex.map(one, range(n_pairs)))
    dt = time.perf_counter() - t
    out = {"cpu_baseline": {"value": n / dt, "unit": "descriptors/s", "cores": cores, "kind": "port",
                            "sample": f"{n_pairs} of {what}, {cores} threads over pairs, {dt:.1f} s wall"}}
    first_leg[0] = False
    # one thread: a sample sized from the rate just measured, about 4 s each
    per_pair_1t = dt * cores / max(n_pairs, 1)
    n1 = int(min(n_pairs, max(2, round(4.0 / max(per_pair_1t, 1e-3)))))
    for key, fn, note in (("cpu_baseline_1t", one, "tables built once"),
                          ("cpu_baseline_ref_faithful", one_ref, "cv::FREAK tables rebuilt for every pair (MoFREAKUtilities.cpp:427)")):
        t = time.perf_counter()
        n = sum(fn(p) for p in range(n1))
        dt1 = time.perf_counter() - t
        out[key] = {"value": n / dt1, "unit": "descriptors/s", "cores": 1, "kind": "port",
                    "sample": f"{n1} of {what}, 1 thread, {note}, {dt1:.1f} s wall"}
    return out


def copy_ceiling_gbs(torch, mib=2048, reps=5):
    """Device-to-device copy rate (read + write bytes per second) measured in this run: what a streaming kernel reaches."""
    a = torch.empty(mib << 20, dtype=torch.uint8, device="cuda")
    b = torch.empty_like(a)
    b.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    return 2.0 * a.numel() * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9


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
    # algorithmic bytes per pair (SURVEY.md 8(d) reading for this row): the two frames in + 12 bytes per keypoint out for
    # the detector; the two frames + 28 bytes per keypoint for the descriptors behind it
    kpp = n / pairs
    det_bytes, desc_bytes = 2 * W * H + 12 * kpp, 2 * W * H + 28 * kpp
    det_s, both_s = (t1 - t0) / (pairs * steps), (t2 - t1) / (pairs * steps)
    out = {"workload": f"{W}x{H} moving-object frame pairs, BriskFeatureDetector(30, 3 octaves) on |cur - prev|, {pairs} pairs/call",
           "keypoints_per_pair": kpp, "pairs_per_s": 1.0 / det_s, "detect_and_describe_pairs_per_s": 1.0 / both_s,
           "roofline": {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS,
                        "detect_algorithmic_bytes_per_pair": det_bytes, "detect_achieved": det_bytes / det_s / 1e9,
                        "detect_frac": det_bytes / det_s / 1e9 / HBM_PEAK_GBS,
                        "detect_and_describe_algorithmic_bytes_per_pair": det_bytes + desc_bytes,
                        "detect_and_describe_achieved": (det_bytes + desc_bytes) / both_s / 1e9,
                        "detect_and_describe_frac": (det_bytes + desc_bytes) / both_s / 1e9 / HBM_PEAK_GBS,
                        "note": "host-timed calls (launch gaps included), wall clock around synchronising calls"}}
    # the reference's whole frame loop as ONE call (mofreak_compute_stream: detector -> descriptors -> rows) on a stack of
    # pairs + 5 frames
    T = pairs + 5
    stack = torch.from_numpy(np.stack([fr[t % len(fr)] for t in range(T)])).cuda()
    rows = torch.empty(pairs * 12000 * 32, dtype=torch.uint8, device="cuda")
    ctx.compute_stream(stack, T, W, H, rows, capacity=rows.numel() // 32)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        n_rows, n_loop_kp = ctx.compute_stream(stack, T, W, H, rows, capacity=rows.numel() // 32)
    out["frame_loop"] = {"pairs_per_s": pairs * steps / (time.perf_counter() - t0), "rows_per_pair": n_rows / pairs,
                         "note": "mofreak_compute_stream on a device-resident stack, rows compacted on the device"}
    del stack, rows
    if with_cpu:  # part of the cpu_baseline leg: the oracle as a reported baseline, never as the product
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib
        a, b = fr[5], fr[0]
        t0 = time.perf_counter()
        k = oracle_lib.brisk_detect(oracle_lib.absdiff(a, b))
        out["cpu_baseline"] = {"value": 1.0 / (time.perf_counter() - t0), "unit": "pairs/s", "cores": 1, "kind": "port",
                               "sample": f"1 of the pairs, oracle/brisk_oracle.c, {len(k)} keypoints"}
    return out


def dist_setup(args):
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:  # main() starts the ranks itself when nobody else has; this is a launcher of someone else's
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's rank count and --gpus disagree")
    if args.share_device:
        local_rank = 0
    elif local_rank >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: local rank {local_rank} but only {torch.cuda.device_count()} GPU(s) visible "
                         "(one process per GPU; --share-device rehearses N > 1 on one)")
    torch.cuda.set_device(local_rank)
    on_device = args.backend == "nccl"  # gloo moves its (small) control tensors and the gathered rows through the host
    if world > 1 or args.force_dist:
        # --force-dist: a one-rank process group, so that every collective and the device branch of the row gather run
        # exactly as they do at N > 1 (the same code path, RCCL included, on a one-GPU box)
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(launch.free_port()))
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if on_device:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
        assert dist.get_world_size() == world
    return torch, dist, rank, local_rank, world, on_device


def library_id(M):
    """Which build produced the line: the in-tree library only (no MOFREAK_HIP_LIBRARY override in a bench run)."""
    import hashlib
    path = M.api.LIB_PATH
    want = os.path.join(ROOT, "mofreak_amd", "libmofreak_hip.so")
    if os.path.realpath(path) != os.path.realpath(want):
        raise SystemExit(f"bench.py measures the in-tree build {want}, not {path} (unset MOFREAK_HIP_LIBRARY)")
    with open(path, "rb") as f:
        return {"path": os.path.relpath(path, ROOT), "sha256_16": hashlib.sha256(f.read()).hexdigest()[:16],
                "build_flags": M.load().mofreak_build_flags()}


def grouped(dist):
    """A process group exists (N > 1, or N = 1 with --force-dist)."""
    return dist.is_available() and dist.is_initialized()


def ranks_seen(dist, world):
    """What the process group itself says (not the flag): goes into every line."""
    return dist.get_world_size() if grouped(dist) else 1


def dist_info(dist, args):
    if not grouped(dist):
        return {"process_group": None}
    return {"process_group": {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "forced_at_n1": bool(args.force_dist)}}


def fence(torch, dist, world):
    torch.cuda.synchronize()
    if grouped(dist):
        dist.barrier()
    torch.cuda.synchronize()


def max_over_ranks(torch, dist, world, on_device, seconds):
    if grouped(dist):
        t = torch.tensor([seconds], dtype=torch.float64, device="cuda" if on_device else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())
    return seconds


# ------------------------------------------------------------------------------------------------ C3 (the metric), C2
def bench_resident(args):
    torch, dist, rank, local_rank, world, on_device = dist_setup(args)
    import mofreak_amd as M
    from mofreak_amd import harness, synth

    cfg = synth.CONFIGS[args.config]
    W, H = cfg["W"], cfg["H"]
    kps = synth.config_grid(args.config)
    n_pairs = args.pairs if args.pairs else (1000 if args.config == "C2" else 256)
    n_kp, gap = len(kps), synth.GAP_FOR_FRAME_DIFFERENCE
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
    copy_gbs = copy_ceiling_gbs(torch) if rank == 0 else None
    stream = torch.cuda.Stream()  # a real (non-null) stream shared by torch and the library
    torch.cuda.set_stream(stream)
    ctx.set_stream(stream.cuda_stream)
    ctx.reserve(W, H)

    def step():
        ctx.extract_pairs(d_frames[gap:], d_frames[:n_pairs], W, H, n_pairs, d_kps, desc, valid)

    for _ in range(args.warmup):
        step()
    fence(torch, dist, world)
    steps = args.steps
    if steps is None:  # no --steps: enough of them for a timed region of about 3 s (an external sampler sees the run)
        t0 = time.perf_counter()
        for _ in range(3):
            step()
        fence(torch, dist, world)
        per = max_over_ranks(torch, dist, world, on_device, (time.perf_counter() - t0) / 3)
        steps = int(min(5000, max(10, math.ceil(3.0 / per))))
    ctx.set_profiling(True)  # HIP events around every kernel group, on the stream the kernels run on
    ctx.get_profile(reset=True)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    fence(torch, dist, world)
    elapsed = time.perf_counter() - t0
    prof = ctx.get_profile(reset=True)
    ctx.set_profiling(False)
    ctx.check_status()
    elapsed = max_over_ranks(torch, dist, world, on_device, elapsed)
    # K steps of a few milliseconds are over before an external sampler of GPU activity has looked once: when the timed region
    # was shorter than two seconds, the same step keeps running for about two more (its own clock, reported as `sustained`,
    # never `value`), so the device is visibly busy and the timed figure has a longer run beside it
    sustained = None
    if elapsed < 2.0 and not args.no_sustain:
        extra = int(min(5000, max(steps, math.ceil(2.0 / (elapsed / steps)))))
        fence(torch, dist, world)
        ts = time.perf_counter()
        for _ in range(extra):
            step()
        fence(torch, dist, world)
        sus = max_over_ranks(torch, dist, world, on_device, time.perf_counter() - ts)
        sustained = {"steps": extra, "seconds": sus, "value": world * n_desc * extra / sus, "ms_per_step": sus / extra * 1e3,
                     "note": "the same step repeated after the timed region (untimed for `value`)"}
        ctx.check_status()
    n_valid = int(valid.sum().item())
    assert n_valid == n_desc, f"{n_desc - n_valid} keypoints were erased: the grid is supposed to be border-safe"
    # every rank proves the bytes of its last timed step against the oracle: rank 0 at N = 1 on all the pairs of the
    # cpu_baseline leg (below), otherwise on a few pairs spread over the stack
    ver = Verifier(desc, valid, n_kp)
    full_leg = world == 1 and not args.no_cpu_baseline
    if not full_leg:
        verify_sample(ver, frames, kps, sorted({0, n_pairs // 3, (2 * n_pairs) // 3, n_pairs - 1})[: max(1, args.verify_pairs)])
    ver_all = [ver.pairs, ver.descriptors, ver.mismatches]
    if grouped(dist):  # the line carries the sum over ranks
        t = torch.tensor(ver_all, dtype=torch.int64, device="cuda" if on_device else "cpu")
        dist.all_reduce(t)
        ver_all = [int(x) for x in t.tolist()]

    # the path's one exchange step: gather the compacted 32-byte rows to rank 0 (not part of a step)
    rows = torch.empty(n_desc * 32, dtype=torch.uint8, device="cuda")
    n_rows = ctx.compact_rows(d_kps, n_pairs, gap - 1, desc, valid, rows)
    gather_ms = None
    if grouped(dist):
        fence(torch, dist, world)
        tg = time.perf_counter()
        allrows, counts = harness.gather_rows(rows if on_device else rows.cpu(), n_rows, dst=0)
        fence(torch, dist, world)
        gather_ms = (time.perf_counter() - tg) * 1e3
        if rank == 0:
            assert sum(counts) == world * n_rows and allrows.numel() == sum(counts) * 32

    if rank == 0:
        total_desc = world * n_desc * steps
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
        counters_note = None
        lib = library_id(M)
        tpath = os.path.join(ROOT, "profiles", "traffic.json" if args.config == "C3" else f"traffic_{args.config}.json")
        if os.path.exists(tpath) and (args.config == "C3" or n_pairs == 1000):  # (the counters are per launch of the default number of pairs)
            tj = json.load(open(tpath))
            if tj.get("library_sha256_16") != lib["sha256_16"]:
                # counters of another build say nothing about this one: the line carries none rather than stale ones
                counters_note = (f"profiles/traffic.json was collected on build {tj.get('library_sha256_16')}, this is {lib['sha256_16']}: "
                                 "traffic and valu_issue withheld (rerun mofreak_amd/tools/profile_tile.sh)")
                tj = {}
            traffic = tj.get("tile_kernel_hbm_bytes_per_launch")
            if tj.get("valu_wave_instr_per_descriptor"):
                # What bounds the kernel is vector instruction issue and LDS cycles, not bytes: instructions per descriptor
                # from the PMC passes (profiles/), the issue cost of the instruction classes from the micro-benchmark
                # recorded there, the shader clock from the device, the launch duration measured live.
                props = torch.cuda.get_device_properties(local_rank)
                n_cus = props.multi_processor_count
                clk = float(getattr(props, "clock_rate", 2400000)) * 1e3  # kHz -> Hz
                issued = tj["valu_wave_instr_per_descriptor"] * n_desc / (tile_ms_avg * 1e-3)
                peak_issue = n_cus * 4 * clk / tj.get("valu_cycles_per_wave_instr", 2.0)
                pd = tj.get("per_descriptor", {})
                cu_cycles = n_cus * clk * tile_ms_avg * 1e-3 / n_desc  # CU cycles per descriptor in this run
                valu = {"achieved_wave_instr_per_s": issued, "peak_wave_instr_per_s": peak_issue, "frac": issued / peak_issue,
                        "wave_instr_per_descriptor": tj["valu_wave_instr_per_descriptor"], "clock_hz": clk,
                        "cu_cycles_per_descriptor": cu_cycles,
                        # SQ_ACTIVE_INST_VALU counts quad-cycles over the four SIMDs of a CU: cycles a SIMD's vector pipe is held
                        "valu_busy_frac": pd.get("SQ_ACTIVE_INST_VALU", 0.0) / cu_cycles if cu_cycles else None,
                        "lds_busy_frac": pd.get("SQ_LDS_IDX_ACTIVE", 0.0) / cu_cycles if cu_cycles else None,
                        "lds_bank_conflict_share_of_lds": (pd.get("SQ_LDS_BANK_CONFLICT", 0.0) / pd["SQ_LDS_IDX_ACTIVE"]) if pd.get("SQ_LDS_IDX_ACTIVE") else None}
        out = {
            "metric": METRIC,
            "value": value, "unit": "descriptors/s", "n_gpus": world, "steps": steps, "warmup": args.warmup,
            "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"{args.config}: {n_pairs} resident {W}x{H} frame pairs per GPU, dense {cfg['step']}-px grid, "
                                   f"{n_kp} keypoints/pair of size {cfg['size']}, 16-byte descriptors",
                       "descriptors_per_step_per_gpu": n_desc, "bit_mode": "SSE", "parallelism": f"one stack per GPU x{world}"},
            "timed_region_s": elapsed, "sustained": sustained,
            "roofline": {"bound": "hbm", "bound_observed": "valu (vector instruction issue, the LDS array at about half: valu_issue below; the HBM fraction is the metric's figure, "
                                                           "not what limits this kernel)",
                         "kernel": "tile_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "copy_ceiling_GBs": copy_gbs, "frac_of_copy_ceiling": achieved / copy_gbs if copy_gbs else None,
                         "algorithmic_bytes_per_pair": b_alg_pair, "pairs_per_launch": pairs_per_launch,
                         "avg_launch_ms": tile_ms_avg, "launches_timed": prof["calls"],
                         "binning_avg_ms": prof["bin_ms"] / launches, "gather_path_avg_ms": prof["gather_ms"] / launches,
                         "pipeline_achieved_GBs": pipeline_gbs, "pipeline_frac": pipeline_gbs / HBM_PEAK_GBS,
                         "valu_issue": valu, "counters_note": counters_note},
            "gather_ms": gather_ms, "ranks_seen": ranks_seen(dist, world), "library": lib,
        }
        out.update(dist_info(dist, args))
        if full_leg:
            cp = min(args.cpu_pairs, n_pairs)
            out.update(cpu_baselines(frames, kps, cp, cores=min(ncpu, 16), shape=f"{W}x{H}", ver=ver))
            ver_all = [ver.pairs, ver.descriptors, ver.mismatches]
        out["verified"] = dict(ver.report(), pairs=ver_all[0], descriptors=ver_all[1], mismatches=ver_all[2])
        failed = ver_all[2] != 0
        if world == 1 and not args.no_detector and args.config == "C3":
            # Outside the metric and its timed region: the row in front of the path, for the record.
            try:
                out["detector"] = detector_figures(ctx, torch, synth, W, H, with_cpu=not args.no_cpu_baseline)
                # the detector's tie rounds are chains of short launches per call: more pairs per call share them
                for big_pairs in (128, 256):
                    big = detector_figures(ctx, torch, synth, W, H, pairs=big_pairs, steps=2, with_cpu=False)
                    out["detector"][f"at_{big_pairs}_pairs_per_call"] = {k: big[k] for k in ("pairs_per_s", "detect_and_describe_pairs_per_s", "frame_loop")}
            except Exception as e:  # never let the side figure take the metric line down
                out["detector"] = {"error": repr(e)}
        emit(out)
    ctx.set_stream(None)
    ctx.close()
    if grouped(dist):
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0 and failed:
        raise SystemExit(f"bench: {ver_all[2]} descriptors differ from the oracle")


# ------------------------------------------------------------------------------------------------ C4: a dataset of clips
def shard_stats(harness, lengths, world, n_kp):
    """How even the longest-processing-time-first shard is, and what the root would take in if the rows were gathered to it."""
    shards = harness.shard_videos([int(x) for x in lengths], world)
    loads = [int(sum(int(lengths[i]) for i in s)) for s in shards]
    rows = [int(sum(max(int(lengths[i]) - 5, 0) for i in s)) * n_kp for s in shards]
    return {"frames_per_rank_max_over_mean": max(loads) / (sum(loads) / len(loads)), "clips_per_rank": [len(s) for s in shards],
            "root_ingest_MB_if_rows_are_gathered": sum(rows[1:]) * 32 / 1e6, "rows_per_rank": rows}


def bench_dataset(args):
    torch, dist, rank, local_rank, world, on_device = dist_setup(args)
    from mofreak_amd import harness, synth

    cfg = synth.CONFIGS["C4"]
    W, H = cfg["W"], cfg["H"]
    n_clips = args.clips
    lengths = synth.clip_lengths(n_clips)  # the same seeded lengths on every rank
    brisk = args.keypoints == "brisk"  # the reference's own keypoint source: the detector inside the pipelined pass (mofreak_compute_clips)
    mo = harness.MoFREAKUtilities(harness.HMDB51, device=local_rank,
                                  keypoint_provider="brisk" if brisk else harness.dense_grid_provider(cfg["step"], cfg["size"], cfg["lo"]))
    if brisk:  # frames with something to detect: objects moving over a textured background
        pool = [synth.moving_objects_stack(int(lengths.max()), W, H, seed=900 + k) for k in range(8)]
    else:
        pool = synth.clip_pool(8, int(lengths.max()), W, H)
    if not args.pageable:  # the decoder's output buffers: page-locked, so that a clip's frames go down by DMA at link speed
        pinned = [mo._ctx.host_alloc(p.shape) for p in pool]
        for dst, src in zip(pinned, pool):
            dst[:] = src
        pool = pinned
    clips = [pool[i % len(pool)][: lengths[i]] for i in range(n_clips)]
    names = [f"clip{i:05d}.avi" for i in range(n_clips)]
    batched = not args.per_clip_calls
    # --write: the path's product, <name>.mofreak TEXT files, inside the timed region: every rank formats its own videos' rows
    # on the device and writes their files (tmp + fsync + rename) into one directory on a memory file system
    out_dir = None
    kw = {}
    if args.write:
        import shutil
        base = "/dev/shm" if os.path.isdir("/dev/shm") else None
        out_dir = os.path.join(base or ".", f"mofreak_bench_c4_{os.environ.get('MASTER_PORT', '0')}_{os.getppid() if world > 1 else os.getpid()}")
        args.write_threads = args.write_threads or max(2, min(16, len(os.sched_getaffinity(0)) // max(1, world)))
        kw = {"write": "ranks", "keep_rows": False, "write_threads": args.write_threads}
    # warm-up: a few clips -- with the detector all of them once, because what the detector finds sizes the buffers (rows, keypoints,
    # page-locked memory), and growing them is not part of a step
    n_warm = n_clips if brisk else 64 * world
    harness.run_dataset(clips[:n_warm], names[:n_warm], out_dir, mo, rank, world, on_device=on_device,
                        workers=args.workers, batched=batched, **kw)
    steps = args.steps or 1
    fence(torch, dist, world)
    t0 = time.perf_counter()
    for _ in range(steps):
        res = harness.run_dataset(clips, names, out_dir, mo, rank, world, on_device=on_device, workers=args.workers, batched=batched, **kw)
    fence(torch, dist, world)
    elapsed = max_over_ranks(torch, dist, world, on_device, time.perf_counter() - t0)
    gather_s = max_over_ranks(torch, dist, world, on_device, res["gather_s"])
    written = None
    if args.write:
        write_s = max_over_ranks(torch, dist, world, on_device, res["write_s"])
        if rank == 0:  # what is on disk: every file there, a few of them compared byte for byte with the host formatter on a fresh extraction
            files = [os.path.join(out_dir, n + ".mofreak") for n in names]
            sizes = [os.path.getsize(f) for f in files]
            checked = 0
            for i in sorted({0, n_clips // 3, n_clips - 1}):
                rows = (mo._ctx.compute_stream_host(np.ascontiguousarray(clips[i])) if brisk else
                        mo._ctx.extract_stream_host(np.ascontiguousarray(clips[i]), harness.dense_grid_provider(cfg["step"], cfg["size"], cfg["lo"])(5, W, H)))
                import mofreak_amd as M
                assert open(files[i], "rb").read() == M.format_rows(rows), f"{files[i]} differs from the host formatter's text"
                checked += 1
            written = {"files": len(files), "text_MB": sum(sizes) / 1e6, "write_s_max_over_ranks": write_s, "files_checked_against_host_formatter": checked,
                       "directory": out_dir, "how": f"text made on the device per round (mofreak_format_rows_device), files written by each rank's {args.write_threads} threads, tmp + fsync + rename"}
        fence(torch, dist, world)
        if rank == 0:
            shutil.rmtree(out_dir, ignore_errors=True)
    if rank == 0:
        n_kp = len(synth.config_grid("C4"))
        n_desc = int(((lengths - 5).clip(min=0) * n_kp).sum())
        if brisk:
            n_desc = int(res["total_rows"])
            n_kp = n_desc / max(1, int((lengths - 5).clip(min=0).sum()))
        assert res["total_rows"] == n_desc
        emit(({
            "metric": "MoFREAK clips/sec on an HMDB51-shaped batch, one video per GPU, rows gathered to rank 0 (BASELINE config 4)",
            "value": n_clips * steps / elapsed, "unit": "clips/s", "descriptors_per_s": n_desc * steps / elapsed,
            "n_gpus": world, "steps": steps, "warmup": 1, "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "u8", "data": "synthetic (8 distinct clips cut to the seeded lengths" + (", objects moving over a textured background)" if brisk else ")"),
            "config": {"workload": f"C4: {n_clips} clips {W}x{H}, seeded log-normal lengths {int(lengths.min())}..{int(lengths.max())} "
                                   f"frames (median {int(np.median(lengths))}), " + (f"BRISK detector (threshold 30, 3 octaves) on every pair, {n_kp:.0f} rows/pair on average"
                                                                                     if brisk else f"dense {cfg['step']}-px grid, {n_kp} keypoints/pair"),
                       "descriptors_per_step": n_desc,
                       "parallelism": f"LPT shard of whole clips over {world} rank(s); " + (
                           "every rank's clips in ONE pipelined mofreak_extract_clips call, rows gathered device to device" if res["batched"]
                           else f"{args.workers} host thread(s) with a context each per rank, one synchronous C-ABI call per clip")},
            "frames_in_GBs": float(lengths.sum() * W * H * steps / elapsed / 1e9),
            "gather_ms": gather_s * 1e3, "ranks_seen": ranks_seen(dist, world), **dist_info(dist, args), "rounds": res["rounds"], "written": written,
            "shard": shard_stats(harness, lengths, world, n_kp),
            "rank0_seconds_last_step": {"extract": res["compute_s"], "exchange_and_copy_to_host": res["gather_s"]}, "frames_in_MB_per_step": float(lengths.sum() * W * H / 1e6),
            "note": (f"host frames in ({'pageable' if args.pageable else 'page-locked'} memory) -> " + (
                         "one .mofreak text file per clip, written by the rank that extracted it; extraction, formatting and the files are all inside the timed region"
                         if args.write else "rows on rank 0's host; compute, gather and the root's device-to-host copy are all inside the timed region"))}))
    mo.close()
    if grouped(dist):
        dist.barrier()
        dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------ C5: a long stream
def bench_stream_brisk(args):
    """C5's shape with the reference's own keypoint source: a 720x576 stream pushed chunk by chunk into a detector stream
    (mofreak_stream_push_frames: detector, descriptors and rows window by window on the device, the next window's frames copied
    down meanwhile), bounded memory as in bench_stream."""
    torch, dist, rank, local_rank, world, on_device = dist_setup(args)
    import mofreak_amd as M
    from mofreak_amd import synth

    cfg = synth.CONFIGS["C5"]
    W, H = cfg["W"], cfg["H"]
    T = min(args.frames, 20000)
    distinct = min(T, 128)
    base = synth.moving_objects_stack(distinct, W, H, seed=333 + rank)
    ctx = M.Context(local_rank)
    chunk = min(args.push_frames, 256)
    bufs = [ctx.host_alloc((chunk, H, W)) for _ in range(2)]
    rows = [ctx.host_alloc((chunk * 8192,), M.api.ROW_DTYPE) for _ in range(2)]

    from concurrent.futures import ThreadPoolExecutor
    filler = ThreadPoolExecutor(max_workers=1)  # the decoder's stand-in: the next chunk is put together while this one is pushed

    def fill(b, t0, n):
        for t in range(n):  # (the stream repeats its distinct frames; a forward-backward sweep keeps the motion continuous)
            q = (t0 + t) % (2 * distinct - 2)
            bufs[b][t] = base[q if q < distinct else 2 * distinct - 2 - q]
        return n

    def one_pass(n_frames):
        total = 0
        starts = list(range(0, n_frames, chunk))
        with ctx.open_stream(W, H, use_detector=True) as st:
            nxt = filler.submit(fill, 0, 0, min(chunk, n_frames))
            for k, t0 in enumerate(starts):
                n, b = nxt.result(), k & 1
                if k + 1 < len(starts):
                    nxt = filler.submit(fill, b ^ 1, starts[k + 1], min(chunk, n_frames - starts[k + 1]))
                total += len(st.push_frames(bufs[b][:n], None, chunk_frames=args.chunk if args.chunk < chunk else 0, rows_out=rows[b]))
        return total

    one_pass(min(T, 2 * chunk))
    steps = args.steps or 1
    fence(torch, dist, world)
    t0 = time.perf_counter()
    for _ in range(steps):
        total = one_pass(T)
    fence(torch, dist, world)
    elapsed = max_over_ranks(torch, dist, world, on_device, time.perf_counter() - t0)
    filler.shutdown()
    for b in bufs + rows:
        ctx.host_free(b)
    if rank == 0:
        emit({"metric": "MoFREAK sustained descriptors/sec on a TRECVID-shaped stream with the BRISK detector as the keypoint source, host frames in and rows out included",
              "value": world * total * steps / elapsed, "unit": "descriptors/s", "n_gpus": world, "steps": steps, "warmup": 1, "ms_per_step": elapsed / steps * 1e3,
              "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": f"synthetic ({distinct} distinct frames of moving objects, swept forward and backward)",
              "config": {"workload": f"C5 with detector keypoints: one {W}x{H} stream of {T} frames per GPU pushed in chunks of {chunk} frames into a detector stream "
                                     "(mofreak_stream_push_frames), BRISK threshold 30, 3 octaves", "rows_per_pair": total / max(1, T - 5)},
              "frame_pairs_per_s": world * (T - 5) * steps / elapsed, "ranks_seen": ranks_seen(dist, world), **dist_info(dist, args)})
    ctx.close()
    if grouped(dist):
        dist.barrier()
        dist.destroy_process_group()


def bench_stream(args):
    """BASELINE config 5: an hour-long 720x576 stream (90 000 frames at 25 fps) through mofreak_stream_push_frames in bounded
    memory: two page-locked chunk buffers of --push-frames frames, refilled by host threads (the decoder's stand-in) while the
    other one is being pushed; the device keeps the stream's last gap frames, the rows of a chunk come back into a page-locked
    buffer per slot.  --frames below 2 * --push-frames: the whole stack in one mofreak_extract_stream_pipelined call."""
    torch, dist, rank, local_rank, world, on_device = dist_setup(args)
    import queue

    import mofreak_amd as M
    from mofreak_amd import synth

    cfg = synth.CONFIGS["C5"]
    W, H = cfg["W"], cfg["H"]
    kps = synth.config_grid("C5")
    T = args.frames
    ncpu = len(os.sched_getaffinity(0))
    distinct = min(T, 256)  # distinct synthetic frames, repeated to the stream's length (host generation is not the subject)
    base = make_stack(distinct, W, H, t0=5000 * rank, workers=max(1, min(16, ncpu // max(1, world))))
    ctx = M.Context(local_rank)
    # resident rate of the same shape for reference (frames already in HBM, no rows back)
    d_fr = torch.from_numpy(base).cuda()
    d_kps = torch.from_numpy(kps).cuda()
    np_res = distinct - 5
    desc = torch.empty((np_res * len(kps), 16), dtype=torch.uint8, device="cuda")
    valid = torch.empty(np_res * len(kps), dtype=torch.uint8, device="cuda")
    for _ in range(2):
        ctx.extract_pairs(d_fr[5:], d_fr[:np_res], W, H, np_res, d_kps, desc, valid)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        ctx.extract_pairs(d_fr[5:], d_fr[:np_res], W, H, np_res, d_kps, desc, valid)
    ctx.synchronize()
    resident = 5 * np_res * len(kps) / (time.perf_counter() - t0)
    del d_fr, desc, valid
    steps = args.steps or (1 if T > 20000 else 3)
    chunk = args.push_frames
    pushed = T >= 2 * chunk
    n_rows_max = (T - 5) * len(kps)
    extra = {}
    if not pushed:
        frames = ctx.host_alloc((T, H, W))  # the decoder's output buffer: page-locked
        for t0 in range(0, T, distinct):
            n = min(distinct, T - t0)
            frames[t0:t0 + n] = base[:n]
        rows = ctx.host_alloc((n_rows_max,), M.api.ROW_DTYPE)
        got = ctx.extract_stream_pipelined_host(frames[: min(T, 600)], kps, chunk_frames=args.chunk, rows_out=rows)  # warm-up
        assert len(got) == (min(T, 600) - 5) * len(kps)
        fence(torch, dist, world)
        t0 = time.perf_counter()
        for _ in range(steps):
            got = ctx.extract_stream_pipelined_host(frames, kps, chunk_frames=args.chunk, rows_out=rows)
        fence(torch, dist, world)
        elapsed = max_over_ranks(torch, dist, world, on_device, time.perf_counter() - t0)
        assert len(got) == n_rows_max
        pinned_bytes = frames.nbytes + rows.nbytes
        ctx.host_free(frames)
        ctx.host_free(rows)
        how = f"one {W}x{H} stream of {T} frames per GPU, whole stack in page-locked host memory, one pipelined call, windows of {args.chunk} frames"
    else:
        bufs = [ctx.host_alloc((chunk, H, W)) for _ in range(2)]
        rows = [ctx.host_alloc((chunk * len(kps),), M.api.ROW_DTYPE) for _ in range(2)]
        pinned_bytes = sum(b.nbytes for b in bufs) + sum(r.nbytes for r in rows)
        fillers = max(1, min(args.fill_threads, ncpu // max(1, world)))
        pool = ThreadPoolExecutor(fillers)

        def fill(buf, t0, n):
            """The decoder's stand-in: frames t0 .. t0 + n of the stream into the chunk buffer, on `fillers` host threads."""
            def part(lo, hi):
                for t in range(lo, hi):  # (frame by frame: the stream repeats its `distinct` frames)
                    buf[t - t0] = base[t % distinct]
            step = (n + fillers - 1) // fillers
            return [pool.submit(part, t0 + i * step, min(t0 + n, t0 + (i + 1) * step)) for i in range(fillers) if i * step < n]

        def one_pass():
            filled = queue.Queue()
            free = queue.Queue()
            for b in range(2):
                free.put(b)

            stop = threading.Event()

            def producer():
                try:
                    for t0 in range(0, T, chunk):
                        b = free.get()
                        if b is None or stop.is_set():  # the consumer failed: nobody will hand a buffer back
                            return
                        n = min(chunk, T - t0)
                        for f in fill(bufs[b], t0, n):
                            f.result()
                        filled.put((b, n))
                finally:
                    filled.put(None)

            th = threading.Thread(target=producer, daemon=True)  # (a failure in the consumer must end the process, not hang it)
            th.start()
            total = waited = 0
            checks = []
            try:
                with ctx.open_stream(W, H, use_detector=False) as st:
                    while True:
                        tw = time.perf_counter()
                        item = filled.get()
                        waited += time.perf_counter() - tw
                        if item is None:
                            break
                        b, n = item
                        got = st.push_frames(bufs[b][:n], kps, chunk_frames=args.chunk, rows_out=rows[b])
                        total += len(got)
                        if len(got):
                            checks.append((int(got["frame_number"][0]), int(got["frame_number"][-1])))
                        free.put(b)
            finally:
                stop.set()
                free.put(None)  # wakes a producer that waits for a buffer
                th.join(timeout=30)
            return total, waited, checks

        warm_T, T_keep = min(T, 2 * chunk + 7), T
        try:
            T = warm_T
            total, _, _ = one_pass()  # warm-up on the first chunks
            assert total == (warm_T - 5) * len(kps)
            T = T_keep
            fence(torch, dist, world)
            t0 = time.perf_counter()
            for _ in range(steps):
                total, waited, checks = one_pass()
            fence(torch, dist, world)
            elapsed = max_over_ranks(torch, dist, world, on_device, time.perf_counter() - t0)
            assert total == n_rows_max and checks[0][0] == 4 and checks[-1][1] == T - 2, (total, checks[:1], checks[-1:])
        finally:
            pool.shutdown(wait=False, cancel_futures=True)
        for b in bufs + rows:
            ctx.host_free(b)
        extra = {"push_frames_per_chunk": chunk, "fill_threads": fillers, "seconds_waiting_for_the_filler_last_step": waited}
        how = (f"one {W}x{H} stream of {T} frames per GPU through mofreak_stream_push_frames: two page-locked buffers of {chunk} frames refilled by "
               f"{fillers} host threads, windows of {args.chunk} frames inside a push")
    if rank == 0:
        n_desc = n_rows_max
        value = world * n_desc * steps / elapsed
        # the link is full duplex: one new frame in per processed frame on the way down, 32-byte rows on the way up
        pcie_bound = PCIE_GBS * 1e9 / max(W * H / len(kps), 32.0)
        emit(({
            "metric": "MoFREAK sustained descriptors/sec on a TRECVID-shaped stream, host frames in and rows out included (BASELINE config 5)",
            "value": value, "unit": "descriptors/s", "n_gpus": world, "steps": steps, "warmup": 1,
            "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8",
            "data": f"synthetic ({distinct} distinct frames repeated)",
            "config": {"workload": f"C5: {how}, dense {cfg['step']}-px grid, {len(kps)} keypoints/frame", "descriptors_per_step_per_gpu": n_desc,
                       "parallelism": f"one stream per GPU x{world}"},
            "ranks_seen": ranks_seen(dist, world), **dist_info(dist, args), "frames_per_s": world * (T - 5) * steps / elapsed, "resident_descriptors_per_s": resident,
            "pcie_bound_descriptors_per_s": pcie_bound, "frac_of_min_bound": value / world / min(pcie_bound, resident),
            "h2d_GBs": (T * W * H * steps / elapsed) / 1e9, "d2h_GBs": (n_desc * 32 * steps / elapsed) / 1e9,
            "peak_page_locked_MB": pinned_bytes / 1e6, **extra}))
    ctx.close()
    if grouped(dist):
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default: as many as make the timed region about 3 s)")
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--pairs", type=int, default=0, help="resident frame pairs per GPU (default 256; C2: 1000)")
    ap.add_argument("--config", default="C3", help="C3 = the metric's (1080p, 8-px grid); C2, C4, C5: see the module text")
    ap.add_argument("--stream", action="store_true", help="with --config C5: the pipelined host-to-device stream (its only mode)")
    ap.add_argument("--frames", type=int, default=2005, help="C5: frames in the stream")
    ap.add_argument("--chunk", type=int, default=256, help="C5: frames per window of the copy/compute pipeline")
    ap.add_argument("--push-frames", type=int, default=2048, help="C5: frames per page-locked chunk buffer (streams of at least two such chunks are pushed "
                    "chunk by chunk in bounded memory; BASELINE's hour-long stream: --frames 90000)")
    ap.add_argument("--fill-threads", type=int, default=12, help="C5: host threads that refill a chunk buffer (the decoder's stand-in)")
    ap.add_argument("--clips", type=int, default=512, help="C4: clips in the batch (HMDB51 has 6766)")
    ap.add_argument("--workers", type=int, default=4, help="C4: host threads per rank, each with a context of its own, taking the rank's clips in turn")
    ap.add_argument("--per-clip-calls", action="store_true", help="C4: one synchronous mofreak_extract_stream call per clip (round 2's path) instead of "
                    "one mofreak_extract_clips call per rank")
    ap.add_argument("--pageable", action="store_true", help="C4: clips in ordinary (pageable) host memory instead of page-locked buffers")
    ap.add_argument("--keypoints", default="grid", choices=["grid", "brisk"], help="C4 / C5: the configuration's dense grid, or the reference's own source, the BRISK detector on every pair's difference image")
    ap.add_argument("--write", action="store_true", help="C4: write every clip's .mofreak text file inside the timed region (device formatter, each rank its own files)")
    ap.add_argument("--write-threads", type=int, default=0, help="C4 --write: threads per rank that write the files (default: the rank's share of the cores, at most 16)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sustain", action="store_true", help="do not keep the step running for two seconds after a short timed region (profiler runs: every launch is traced)")
    ap.add_argument("--cpu-pairs", type=int, default=256, help="pairs of the workload the CPU oracle is timed on and the GPU output is verified on (default: all 256 of a step, about 35 core-seconds)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL); 'gloo' + "
                    "--share-device rehearses the N > 1 control flow on a one-GPU box")
    ap.add_argument("--share-device", action="store_true", help="all ranks use cuda:0 (rehearsal only)")
    ap.add_argument("--force-dist", action="store_true", help="--gpus 1: create a one-rank process group of --backend anyway, so that the collectives and "
                    "the device branch of the row gather (RCCL with the default backend) run as they do at N > 1")
    ap.add_argument("--no-detector", action="store_true", help="skip the (untimed-region) keypoint detector figures")
    ap.add_argument("--verify-pairs", type=int, default=4, help="pairs per rank checked against the oracle when there is no "
                    "cpu_baseline leg (N > 1, --no-cpu-baseline); with the leg, all of its pairs are checked")
    ap.add_argument("--launch-timeout", type=float, default=3000.0, help="N > 1 started by this script: seconds before the ranks are stopped")
    args = ap.parse_args()
    if args.gpus > 1 and not launch.launched_by_a_launcher():
        # `python bench.py --gpus N` by itself: N fresh rank processes, started before this one has made any GPU call
        # (it never makes one); rank 0's line is the job's, a failing rank fails the job.
        raise SystemExit(launch.self_launch(args.gpus, os.path.abspath(__file__), sys.argv[1:], timeout_s=args.launch_timeout))
    quiet_stdout()
    if args.config in ("C2", "C3"):
        bench_resident(args)
    elif args.config == "C4":
        bench_dataset(args)
    elif args.config == "C5":
        (bench_stream_brisk if args.keypoints == "brisk" else bench_stream)(args)
    else:
        raise SystemExit(f"unknown --config {args.config}")


if __name__ == "__main__":
    main()
