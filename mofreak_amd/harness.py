"""Host-side mirror of the reference's interface for the path, in Python, over the C ABI.

  MoFREAKUtilities      the reference class (src/MoFREAK/MoFREAKUtilities.h:55-104): same method names,
                        argument meaning and outputs; "videos" are raw gray frame stacks (.npy, T x H x W u8)
                        because neither box has a video decoder; keypoints come from the BRISK detector on the
                        difference image like the reference's (keypoint_provider="brisk", MoFREAKUtilities.cpp:420-423)
                        or from a provider callback (dense grid by default, the benchmark configuration)
  compute_mofreak_files computeMoFREAKFiles() (src/MoFREAK/main.cpp:854-924): walk a directory of videos, one
                        .mofreak file each -- sharded one-video-per-GPU across ranks (no data-path collective)
  gather_rows           the only exchange step: variable-length gather of 32-byte rows to rank 0
  run_dataset           the whole of BASELINE config 4: shard -> per-rank extraction -> gather -> rank 0 writes every
                        video's .mofreak text in (video, frame, keypoint) order, byte-identical to a 1-rank run

PyTorch is plumbing here (device tensors, torch.distributed); the compute is libmofreak_hip.so.
"""
from __future__ import annotations

import os
from collections import deque
from concurrent.futures import ThreadPoolExecutor
from typing import Callable, Sequence

import numpy as np

from . import api, synth

# enum datasets {KTH, TRECVID, HOLLYWOOD, UTI1, UTI2, HMDB51, UCF101}  (MoFREAKUtilities.h:103)
KTH, TRECVID, HOLLYWOOD, UTI1, UTI2, HMDB51, UCF101 = range(7)

KeypointProvider = Callable[[int, int, int], np.ndarray]  # (frame_index, W, H) -> (n, 3) float32


def dense_grid_provider(step: int = 16, size: float = 12.0, lo: int = 38) -> KeypointProvider:
    cache = {}

    def provider(_frame: int, W: int, H: int) -> np.ndarray:
        if (W, H) not in cache:
            cache[(W, H)] = synth.dense_grid(W, H, step, size, lo)
        return cache[(W, H)]

    provider.shared = True  # same list for every frame -> the kp_offsets == NULL fast path
    return provider


class MoFREAKUtilities:
    """Drop-in for the reference class for the extraction path.

    Reference semantics kept: computeMoFREAKFromFile APPENDS to the internal feature list, ALWAYS writes
    mofreak_filename with everything accumulated so far, and optionally clears afterwards
    (MoFREAKUtilities.cpp:374-498); readMoFREAKFeatures appends the file's rows in REVERSE order
    (:1206-1210); getMoFREAKFeatures returns a copy.
    """

    NUMBER_OF_BYTES_FOR_APPEARANCE = 8  # MoFREAKUtilities.h:72
    NUMBER_OF_BYTES_FOR_MOTION = 8      # MoFREAKUtilities.h:73

    BRISK_THRESHOLD = 30  # cv::BriskFeatureDetector(30) (MoFREAKUtilities.cpp:420-421)
    BRISK_OCTAVES = 3     # its default (brisk.h:293)

    def __init__(self, dset: int, device: int = 0, keypoint_provider: KeypointProvider | str | None = None, **params):
        self.dataset = dset
        self.current_action = 0
        self.actions: dict[str, int] = {}
        self.features: deque = deque()  # of structured rows (api.ROW_DTYPE arrays)
        self._labels: deque = deque()
        self._device, self._params = device, dict(params)
        self._ctx = api.Context(device, **params)
        self.keypoint_provider = keypoint_provider or dense_grid_provider()

    def clone(self) -> "MoFREAKUtilities":
        """Another instance with the same parameters and keypoint source and a context (stream, workspace) of its own:
        what a second host thread extracting other videos on the same device uses (one context per thread)."""
        return MoFREAKUtilities(self.dataset, self._device, self.keypoint_provider, **self._params)

    def workers(self, n: int) -> list:
        """This instance and n - 1 clones of it (kept for later calls, closed with it)."""
        self._clones = getattr(self, "_clones", [])
        while len(self._clones) < n - 1:
            self._clones.append(self.clone())
        return [self] + self._clones[: n - 1]

    # ---- the hot path
    def computeMoFREAKFromFile(self, video_filename: str, mofreak_filename: str,
                               clear_features_after_computation: bool) -> None:
        frames = np.load(video_filename, mmap_mode="r")
        if frames.ndim != 3 or frames.dtype != np.uint8:
            raise ValueError(f"{video_filename}: expected a (T, H, W) uint8 gray frame stack")
        rows = self.extract_rows(np.ascontiguousarray(frames))
        if len(rows):
            self.features.append(rows)
        print(f"Writing this mofreak file: {mofreak_filename}")
        self.writeMoFREAKFeaturesToFile(mofreak_filename)
        if clear_features_after_computation:
            self.clearFeatures()

    # north-star spelling (BASELINE.json names these; the reference snapshot does not have them)
    computeMoFREAKFeatures = computeMoFREAKFromFile

    def extract_rows(self, frames: np.ndarray, chunk_frames: int = 0) -> np.ndarray:
        """Rows of one gray frame stack, in the reference's order (frames ascending, keypoints in input order).

        chunk_frames > gap: a long stream is walked in chunks of that many frames, consecutive chunks overlapping
        by `gap` frames (the reference keeps only a `gap`-deep frame queue, MoFREAKUtilities.cpp:391-399, 485-487),
        so that device memory is bounded by the chunk, not by the stream."""
        T, H, W = frames.shape
        gap = self._ctx.params.gap_for_frame_difference
        if T <= gap:
            return np.zeros(0, api.ROW_DTYPE)
        if chunk_frames and chunk_frames > gap and T > chunk_frames:
            parts = []
            for t0 in range(0, T - gap, chunk_frames - gap):
                sub = frames[t0:t0 + chunk_frames]
                rows = self._extract_rows_one(np.ascontiguousarray(sub), frame_offset=t0)
                parts.append(rows)
            return np.concatenate(parts) if parts else np.zeros(0, api.ROW_DTYPE)
        return self._extract_rows_one(frames, 0)

    def _extract_rows_one(self, frames: np.ndarray, frame_offset: int) -> np.ndarray:
        T, H, W = frames.shape
        gap = self._ctx.params.gap_for_frame_difference
        prov = self.keypoint_provider
        if isinstance(prov, str):
            if prov != "brisk":
                raise ValueError(f"unknown keypoint source {prov!r}")
            rows = self._ctx.compute_stream_host(frames, self.BRISK_THRESHOLD, self.BRISK_OCTAVES)
        elif getattr(prov, "shared", False):
            rows = self._ctx.extract_stream_host(frames, prov(gap, W, H))
        else:
            lists = [np.ascontiguousarray(prov(frame_offset + t, W, H), np.float32).reshape(-1, 3) for t in range(gap, T)]
            offs = np.concatenate([[0], np.cumsum([len(k) for k in lists])]).astype(np.int64)
            kps = np.concatenate(lists) if offs[-1] else np.zeros((0, 3), np.float32)
            rows = self._ctx.extract_stream_host(frames, kps, kp_offsets=offs)
        if frame_offset:
            rows["frame_number"] += frame_offset
        return rows

    def buildMoFREAKFeature(self, cur: np.ndarray, prev: np.ndarray, x: float, y: float, size: float):
        """One keypoint of one frame pair -> (appearance[8], motion[8]) or None if FREAK erases it."""
        d, v = self._ctx.extract_pairs_host(cur[None], prev[None], np.float32([[x, y, size]]))
        return (d[0, :8].copy(), d[0, 8:].copy()) if v[0] else None

    # ---- feature list / files
    def getMoFREAKFeatures(self) -> np.ndarray:
        return np.concatenate(list(self.features)) if self.features else np.zeros(0, api.ROW_DTYPE)

    def clearFeatures(self) -> None:
        self.features.clear()

    def writeMoFREAKFeaturesToFile(self, output_file: str) -> None:
        write_atomic(output_file, (api.format_rows(chunk) for chunk in self.features))

    def readMoFREAKFeatures(self, filename: str, num_to_sample: int = 0) -> None:
        with open(filename, "rb") as f:
            rows = api.parse_rows(f.read())
        if num_to_sample and len(rows) > num_to_sample:
            # the reference random_shuffles and takes num_to_sample from the back (:1194-1203); nondeterministic there
            rng = np.random.default_rng()
            rows = rows[rng.permutation(len(rows))][-num_to_sample:]
        if len(rows):
            self.features.append(rows[::-1].copy())

    def setAllFeaturesToLabel(self, label: int) -> None:
        self._label = label  # labels are side metadata here; rows carry no action field

    def setCurrentAction(self, folder_name: str) -> None:
        if folder_name not in self.actions:  # the UCF101 branch (:1049-1059), "the right way to do this"
            self.actions[folder_name] = len(self.actions)
        self.current_action = self.actions[folder_name]

    def close(self):
        for c in getattr(self, "_clones", []):
            c.close()
        self._clones = []
        self._ctx.close()


def write_atomic(path: str, chunks) -> None:
    """Write `chunks` (bytes objects) to `path` through `<path>.tmp` + fsync + rename: the target either does not exist
    or is complete.  The per-video .mofreak file is the pipeline's checkpoint (compute_mofreak_files(skip_existing=True)),
    and an empty file is a legitimate result (a clip no longer than the frame gap), so size cannot tell a finished file
    from one whose writer was killed -- existence has to mean "complete"."""
    tmp = path + ".tmp"
    with open(tmp, "wb") as f:
        for c in chunks:
            f.write(c)
        f.flush()
        os.fsync(f.fileno())
    os.replace(tmp, path)


# ------------------------------------------------------------------ sharding
def shard_videos(costs: Sequence[float], world_size: int) -> list[list[int]]:
    """Longest-processing-time-first assignment of videos to ranks; returns per-rank index lists (ascending)."""
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    load = [0.0] * world_size
    out: list[list[int]] = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda k: (load[k], k))
        out[r].append(i)
        load[r] += costs[i]
    return [sorted(x) for x in out]


def gather_rows(rows, n_rows: int, dst: int = 0, group=None):
    """Variable-length gather of 32-byte rows to rank `dst`: one all_gather of the counts, then every peer sends
    its rows straight to the root (point-to-point, one xGMI link per peer -- not a ring).

    rows: uint8 tensor (>= n_rows*32 bytes) on the rank's device (cuda for nccl/RCCL, cpu for gloo).
    Returns on dst: (uint8 tensor of all rows in rank order, list of per-rank counts); elsewhere (None, counts).
    """
    import torch
    import torch.distributed as dist

    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = rows.device
    cnt = torch.tensor([n_rows], dtype=torch.int64, device=dev)
    all_counts = torch.zeros(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(all_counts, cnt, group=group)
    counts = [int(c) for c in all_counts.cpu().tolist()]  # ONE host copy for all ranks' counts
    mine = rows.reshape(-1)[: n_rows * 32]
    if rank == dst:
        out = torch.empty(sum(counts) * 32, dtype=torch.uint8, device=dev)
        offs = np.concatenate([[0], np.cumsum(counts)]) * 32
        ops = []
        for r in range(world):
            seg = out[offs[r]: offs[r + 1]]
            if r == rank:
                seg.copy_(mine)
            elif counts[r]:
                ops.append(dist.P2POp(dist.irecv, seg, r, group))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        return out, counts
    if n_rows:
        for req in dist.batch_isend_irecv([dist.P2POp(dist.isend, mine, dst, group)]):
            req.wait()
    return None, counts


def compute_mofreak_files(video_paths: Sequence[str], out_dir: str, mofreak: MoFREAKUtilities, rank: int = 0,
                          world_size: int = 1, costs: Sequence[float] | None = None, skip_existing: bool = False) -> list[str]:
    """computeMoFREAKFiles (main.cpp:854-924) for this rank's shard: <video> -> <out_dir>/<video>.mofreak.

    skip_existing: a video whose .mofreak file is already there is not computed again -- the per-video file is the
    pipeline's checkpoint; the reference itself always recomputes."""
    costs = costs if costs is not None else [os.path.getsize(p) for p in video_paths]
    mine = shard_videos(costs, world_size)[rank]
    os.makedirs(out_dir, exist_ok=True)
    written = []
    for i in mine:
        out = os.path.join(out_dir, os.path.basename(video_paths[i]) + ".mofreak")
        if skip_existing and os.path.exists(out):
            continue
        mofreak.computeMoFREAKFromFile(video_paths[i], out, True)
        written.append(out)
    return written


def group_exists() -> bool:
    """A torch.distributed process group has been initialised in this process (any world size, 1 included)."""
    try:
        import torch.distributed as dist
    except ImportError:
        return False
    return dist.is_available() and dist.is_initialized()


def _pinned_rows(mofreak, n_rows: int) -> np.ndarray:
    """A page-locked row buffer of at least n_rows rows, kept on the instance (page-locking memory is slow).  When it has
    to grow, the old buffer is only let go of: its pages are released with the last view of it (api.host_alloc), so rows
    a caller still holds from an earlier call stay readable."""
    buf = getattr(mofreak, "_row_buf", None)
    if buf is None or len(buf) < n_rows:
        buf = mofreak._row_buf = mofreak._ctx.host_alloc((max(n_rows, 1),), api.ROW_DTYPE)
    return buf


def _pinned_bytes(mofreak, n: int):
    import torch
    buf = getattr(mofreak, "_root_buf", None)
    if buf is None or buf.numel() < n:
        buf = mofreak._root_buf = torch.empty(max(n, 1), dtype=torch.uint8, pin_memory=True)
    return buf[:n]


def plan_batches(shard: Sequence[int], nbytes: Sequence[int], batch_bytes: int) -> list[list[int]]:
    """A rank's videos (ascending) cut into consecutive batches of at most batch_bytes of frames (at least one video
    each): what is loaded, extracted, gathered, written and let go of together.  Host memory, HBM and the amount of work
    a crash can lose are bounded by the batch, not by the dataset."""
    out: list[list[int]] = []
    cur: list[int] = []
    acc = 0
    for i in shard:
        if cur and acc + nbytes[i] > batch_bytes:
            out.append(cur)
            cur, acc = [], 0
        cur.append(i)
        acc += nbytes[i]
    if cur:
        out.append(cur)
    return out


def _extract_batch(mofreak, stacks, use_batched: bool, to_device, workers: int, dest=None):
    """Rows of one batch of clips, clip after clip.  Returns (rows, n_rows, per-clip counts): rows is a uint8 CUDA tensor
    when to_device is a device (batched mode only), else a ROW_DTYPE array -- a view of `dest` (a page-locked ROW_DTYPE
    array with room for the batch: the rows are written where the caller keeps them) when one is given.  Clips of different
    frame sizes are handed to mofreak_extract_clips size by size (it takes one W x H per call) and their rows put back in
    the batch's clip order."""
    import torch

    if not stacks:
        return (torch.empty(0, dtype=torch.uint8, device=to_device) if to_device is not None else np.zeros(0, api.ROW_DTYPE)), 0, []
    if not use_batched:
        if workers > 1 and len(stacks) > 1:
            import queue
            from concurrent.futures import ThreadPoolExecutor

            pool = queue.SimpleQueue()
            for m in mofreak.workers(min(workers, len(stacks))):
                pool.put(m)

            def one(st):
                m = pool.get()  # an instance nobody else is using
                try:
                    return m.extract_rows(st)
                finally:
                    pool.put(m)

            with ThreadPoolExecutor(max_workers=min(workers, len(stacks))) as ex:
                parts = list(ex.map(one, stacks))  # in clip order, whichever thread did what
        else:
            parts = [mofreak.extract_rows(st) for st in stacks]
        rows = np.concatenate(parts) if parts else np.zeros(0, api.ROW_DTYPE)
        return rows, len(rows), [len(p) for p in parts]
    # batched: one pipelined call per frame size
    gap = mofreak._ctx.params.gap_for_frame_difference
    prov = mofreak.keypoint_provider
    sizes: dict[tuple[int, int], list[int]] = {}
    for j, st in enumerate(stacks):
        sizes.setdefault((st.shape[1], st.shape[2]), []).append(j)
    counts = [0] * len(stacks)
    pieces = []  # (rows of the group, offsets, member indices)
    for (H, W), members in sizes.items():
        group = [stacks[j] for j in members]
        if prov == "brisk":
            # the reference's own keypoint source: the detector runs window by window inside the same pipelined pass
            # (mofreak_compute_clips); the number of rows is only known afterwards
            thr, octs = mofreak.BRISK_THRESHOLD, mofreak.BRISK_OCTAVES
            if to_device is not None:
                n_pairs = int(sum(max(s.shape[0] - gap, 0) for s in group))
                per_pair = getattr(mofreak, "_rows_per_pair", 8192)
                for _ in range(2):
                    buf = torch.empty(max(n_pairs * per_pair, 1) * 32, dtype=torch.uint8, device=to_device)
                    try:
                        n_rows, offs, _ = mofreak._ctx.compute_clips(group, thr, octs, rows_out=buf)
                        break
                    except api.MoFREAKError as e:
                        if e.code != api.ERR_CAPACITY or _ == 1:
                            raise
                        per_pair = mofreak._rows_per_pair = per_pair * 4
            elif len(sizes) > 1:  # (several frame sizes in one batch: a buffer of its own per size)
                buf, offs, _ = mofreak._ctx.compute_clips(group, thr, octs, rows_per_pair=max(64, (W * H) // 64))
            else:
                # rows wanted on this host: into page-locked memory kept across calls (they travel back window by window under
                # the pipeline's kernels, no staging copy) -- sized by a guess of the rows a pair yields, grown when a call says
                # it was too small (a numpy array for the worst case would be gigabytes of pages to touch)
                n_pairs = int(sum(max(s.shape[0] - gap, 0) for s in group))
                per_pair = getattr(mofreak, "_rows_per_pair_host", max(64, (W * H) // 256))
                for _ in range(4):
                    try:
                        buf, offs, _n_kp = mofreak._ctx.compute_clips(group, thr, octs, rows_out=_pinned_rows(mofreak, max(n_pairs * per_pair, 1)))
                        break
                    except api.MoFREAKError as e:
                        if e.code != api.ERR_CAPACITY or _ == 3:
                            raise
                        per_pair = mofreak._rows_per_pair_host = per_pair * 4
            for k, j in enumerate(members):
                counts[j] = int(offs[k + 1] - offs[k])
            pieces.append((buf, offs, members))
            continue
        kps = prov(gap, W, H)
        cap = int(sum(max(s.shape[0] - gap, 0) for s in group)) * len(kps)
        if to_device is not None:
            buf = torch.empty(max(cap, 1) * 32, dtype=torch.uint8, device=to_device)
            n_rows, offs = mofreak._ctx.extract_clips(group, kps, rows_out=buf)
        elif len(sizes) == 1:
            # the rows are wanted on this host: they travel back window by window under the pipeline's kernels, into
            # page-locked memory (the caller's, or a buffer kept across calls)
            buf, offs = mofreak._ctx.extract_clips(group, kps, rows_out=dest if dest is not None else _pinned_rows(mofreak, cap))
        else:
            buf, offs = mofreak._ctx.extract_clips(group, kps)
        for k, j in enumerate(members):
            counts[j] = int(offs[k + 1] - offs[k])
        pieces.append((buf, offs, members))
    total = int(sum(counts))
    if len(pieces) == 1:
        return pieces[0][0], total, counts
    seg = {}
    for buf, offs, members in pieces:
        for k, j in enumerate(members):
            seg[j] = buf[int(offs[k]) * 32: int(offs[k + 1]) * 32] if to_device is not None else buf[int(offs[k]): int(offs[k + 1])]
    ordered = [seg[j] for j in range(len(stacks))]
    if to_device is not None:
        return (torch.cat(ordered) if total else torch.empty(0, dtype=torch.uint8, device=to_device)), total, counts
    if dest is not None:
        np.concatenate(ordered, out=dest[:total])
        return dest[:total], total, counts
    return np.concatenate(ordered), total, counts


def run_dataset(videos: Sequence, names: Sequence[str], out_dir: str | None, mofreak: MoFREAKUtilities, rank: int = 0,
                world_size: int = 1, costs: Sequence[float] | None = None, group=None, on_device: bool = False,
                workers: int = 1, batched: bool = True, keep_rows: bool = True, batch_bytes: int = 4 << 30,
                write: str = "root", write_threads: int = 8) -> dict:
    """BASELINE config 4 end to end (main.cpp:854-924 over a whole dataset).

    videos[i]: a (T, H, W) uint8 gray stack or the path of a .npy file holding one; names[i]: its output stem.
    1. shard: longest-processing-time-first over `costs` (default: frame counts), one video per GPU at a time; a rank's
       shard is cut into batches of at most `batch_bytes` of frames (plan_batches), and the ranks walk their batches in
       step, one round per batch -- load, extract, gather, write, let go;
    2. every rank extracts the rows of its batch (no collective on the data path).  batched (a shared keypoint list, i.e.
       a dense grid): all clips of the batch in one mofreak_extract_clips call per frame size -- the clips share
       launches and the three-stream copy/compute pipeline, and with a process group the rows stay in HBM.  Otherwise
       (detector keypoints, per-frame providers): one C-ABI call per clip, `workers` host threads with a context each;
    3. the one exchange, whenever a process group exists (world size 1 included: the same code runs): counts per video
       (all_reduce) + gather_rows of the 32-byte rows to rank 0 -- device to device over RCCL when on_device, then ONE
       device-to-host copy on the root;
    4. rank 0 writes <out_dir>/<name>.mofreak for the round's videos, rows in (video, frame, keypoint) order -- the
       bytes a 1-rank run writes; the files of a finished round are on disk before the next round starts (out_dir None:
       nothing is written).  keep_rows=False: rows are not kept after their round (with out_dir they are still
       written); without out_dir and without keep_rows nothing but the counts comes to the host.
       write="ranks" (one node = one file system): every rank writes the files of ITS OWN videos -- the text is made on the
       device from the rows the extraction left in HBM (mofreak_format_rows_device, one call per round with a segment per
       video), lands in page-locked host memory and goes to the files from `write_threads` threads (tmp + fsync + rename each)
       -- the same bytes in the same files, a round's files being written while the next round is extracted (two text buffers);
       with keep_rows=False no row travels to the root at all (the exchange is the counts), so neither the root's link nor
       its single formatter thread stands between eight GPUs and the disk.  write_s: what the rank waited for formatting and
       for writes that had not finished when their buffer was needed again (and at the end).
    Returns timings and, on rank 0 with keep_rows, `rows_per_video`: views into a page-locked buffer that belongs to
    `mofreak` and is reused by its next run_dataset call (copy what has to outlive that); with per-frame keypoint sources
    (no capacity known up front) and several rounds: copies.
    """
    import time

    import torch
    import torch.distributed as dist

    def stack_of(v):
        return np.load(v, mmap_mode="r") if isinstance(v, str) else v

    n = len(videos)
    shapes = [tuple(stack_of(v).shape) for v in videos]  # a .npy header each; the same on every rank
    if costs is None:
        costs = [s[0] for s in shapes]
    nbytes = [int(np.prod(s)) for s in shapes]
    shards = shard_videos(costs, world_size)
    plans = [plan_batches(s, nbytes, batch_bytes) for s in shards]
    n_rounds = max((len(p) for p in plans), default=0)
    distributed = world_size > 1 or group_exists()
    prov = getattr(mofreak, "keypoint_provider", None)
    use_batched = batched and (prov == "brisk" or (not isinstance(prov, str) and getattr(prov, "shared", False)))
    cuda = torch.cuda.is_available()
    by_ranks = write == "ranks" and out_dir is not None
    if write not in ("root", "ranks"):
        raise ValueError("write must be 'root' or 'ranks'")
    rows_dev = torch.device("cuda", mofreak._device) if (use_batched and cuda and (distributed or by_ranks)) else None
    want_rows = keep_rows or (out_dir is not None and not by_ranks)
    if out_dir is not None and by_ranks:
        os.makedirs(out_dir, exist_ok=True)
    text_bufs = [None, None]  # page-locked text buffers: a round's files are written while the next round is extracted
    pending = [[], []]        # the file writes still reading from each of them
    text_bytes = 0
    writers = ThreadPoolExecutor(max_workers=max(1, write_threads)) if by_ranks else None
    if rank == 0 and out_dir is not None:
        os.makedirs(out_dir, exist_ok=True)

    counts = np.zeros(n, np.int64)
    rows_per_video: dict[int, np.ndarray] = {}
    t_compute = t_gather = t_write = 0.0
    rows_here = total_rows = 0
    # Rows that are kept are written where they stay: one page-locked buffer for the whole run (its size is known up front
    # for a shared keypoint list: pairs x keypoints), filled round after round -- no copy per video.
    keep_buf, keep_at = None, 0
    if rank == 0 and keep_rows and use_batched and prov != "brisk":
        gap = mofreak._ctx.params.gap_for_frame_difference
        keep_buf = _pinned_rows(mofreak, int(sum(max(sh[0] - gap, 0) * len(prov(gap, sh[2], sh[1])) for sh in shapes)))
    for r in range(n_rounds):
        mine = plans[rank][r] if r < len(plans[rank]) else []
        round_ids = [i for p in plans if r < len(p) for i in p[r]]  # rank order, ascending inside a rank: the gathered order
        t0 = time.perf_counter()
        stacks = [np.ascontiguousarray(stack_of(videos[i])) for i in mine]
        dest = keep_buf[keep_at:] if (keep_buf is not None and not distributed) else None
        local, n_local, local_counts = _extract_batch(mofreak, stacks, use_batched, rows_dev, workers, dest)
        del stacks
        rows_here += n_local
        t_compute += time.perf_counter() - t0
        if by_ranks and mine:
            # this rank's files of the round: text on the device, files from a few threads
            t2 = time.perf_counter()
            starts = np.concatenate([[0], np.cumsum(local_counts)]).astype(np.int64)
            texts = None
            slot = r & 1
            for fut in pending[slot]:  # the files of two rounds ago read from this buffer
                fut.result()
            pending[slot] = []
            if rows_dev is not None:
                try:
                    text_bufs[slot], offs = mofreak._ctx.format_rows_device(local, n_local, row_starts=starts[:-1], out=text_bufs[slot])
                    texts = [text_bufs[slot][int(offs[k]): int(offs[k + 1])] for k in range(len(mine))]
                except api.MoFREAKError as e:  # rows the device leaves to the host formatter (see include/mofreak_hip.h)
                    if e.code != api.ERR_UNSUPPORTED:
                        raise
            if texts is None:
                host_rows = local[: n_local * 32].cpu().numpy().view(api.ROW_DTYPE).reshape(-1) if rows_dev is not None else local
                texts = list(writers.map(lambda k: api.format_rows(host_rows[starts[k]: starts[k + 1]]), range(len(mine))))
            text_bytes += int(sum(len(t) for t in texts))
            pending[slot] = [writers.submit(write_atomic, os.path.join(out_dir, names[mine[k]] + ".mofreak"), [t]) for k, t in enumerate(texts)]
            del texts
            t_write += time.perf_counter() - t2

        t1 = time.perf_counter()
        round_counts = np.zeros(len(round_ids), np.int64)
        pos = {i: k for k, i in enumerate(round_ids)}
        for i, c in zip(mine, local_counts):
            round_counts[pos[i]] = c
        all_rows = None
        if distributed and not want_rows:  # files written by their ranks, rows not kept: the exchange is the counts
            dev = torch.device("cuda", torch.cuda.current_device()) if on_device else torch.device("cpu")
            c = torch.from_numpy(round_counts).to(dev)
            dist.all_reduce(c, group=group)
            round_counts = c.cpu().numpy()
            n_all = torch.tensor([n_local], dtype=torch.int64, device=dev)
            dist.all_reduce(n_all, group=group)
            per_rank = [int(n_all.item())]
        elif distributed:
            dev = torch.device("cuda", torch.cuda.current_device()) if on_device else torch.device("cpu")
            c = torch.from_numpy(round_counts).to(dev)
            dist.all_reduce(c, group=group)  # every video belongs to exactly one rank: the sum is the per-video count
            round_counts = c.cpu().numpy()
            if rows_dev is not None:
                buf = local if on_device else local[: n_local * 32].cpu()
            else:
                buf = torch.from_numpy(np.ascontiguousarray(local).view(np.uint8).reshape(-1).copy()).to(dev)
            gathered, per_rank = gather_rows(buf, n_local, dst=0, group=group)
            if on_device:
                torch.cuda.synchronize()
            if rank == 0 and want_rows:
                if keep_buf is not None:  # the root's one copy of the round's rows, to where they stay
                    all_rows = keep_buf[keep_at: keep_at + gathered.numel() // 32]
                    torch.from_numpy(all_rows.view(np.uint8).reshape(-1)).copy_(gathered, non_blocking=True)
                    if gathered.is_cuda:
                        torch.cuda.synchronize()
                elif gathered.is_cuda:  # ... or into page-locked memory kept across calls
                    host = _pinned_bytes(mofreak, gathered.numel())
                    host.copy_(gathered, non_blocking=True)
                    torch.cuda.synchronize()
                    all_rows = host.numpy().view(api.ROW_DTYPE).reshape(-1)
                else:
                    all_rows = gathered.numpy().view(api.ROW_DTYPE).reshape(-1)
            del gathered, buf
        else:
            per_rank = [n_local]
            if want_rows:
                all_rows = local if rows_dev is None else local[: n_local * 32].cpu().numpy().view(api.ROW_DTYPE).reshape(-1)
        del local
        t_gather += time.perf_counter() - t1
        for i, c in zip(round_ids, round_counts):
            counts[i] = c
        if rank == 0:
            total_rows += int(sum(per_rank))
            assert int(sum(per_rank)) == int(round_counts.sum())
        if rank == 0 and want_rows:
            assert len(all_rows) == int(round_counts.sum())
            t2 = time.perf_counter()
            off = 0
            for i, c in zip(round_ids, round_counts):
                seg = all_rows[off: off + int(c)]
                off += int(c)
                if out_dir is not None and not by_ranks:
                    write_atomic(os.path.join(out_dir, names[i] + ".mofreak"), [api.format_rows(seg)])
                if keep_rows:
                    rows_per_video[i] = seg if (n_rounds == 1 or keep_buf is not None) else seg.copy()
            keep_at += len(all_rows) if keep_buf is not None else 0
            t_write += time.perf_counter() - t2

    if writers is not None:
        t2 = time.perf_counter()
        for slot in (0, 1):
            for fut in pending[slot]:
                fut.result()  # (a failed write raises here)
        writers.shutdown()
        t_write += time.perf_counter() - t2
    for b in text_bufs:
        if b is not None:
            mofreak._ctx.host_free(b)
    out = {"compute_s": t_compute, "gather_s": t_gather, "videos_here": len(shards[rank]), "rows_here": int(rows_here),
           "batched": bool(use_batched), "rounds": n_rounds, "distributed": bool(distributed)}
    if by_ranks:
        out["write_s"] = t_write
        out["text_bytes_here"] = int(text_bytes)
        out["write"] = "ranks"
    if rank == 0:
        out["total_rows"] = int(total_rows)
        out["rows_per_video_counts"] = counts
        if want_rows and not by_ranks:
            out["write_s"] = t_write
        if keep_rows:
            out["rows_per_video"] = {i: rows_per_video.get(i, np.zeros(0, api.ROW_DTYPE)) for i in range(n)}
    return out


def split_stream(T: int, world_size: int, gap: int = synth.GAP_FOR_FRAME_DIFFERENCE) -> list[tuple[int, int]]:
    """One long stream over `world_size` ranks (the optional variant of config 5): contiguous pieces of
    the T - gap processed frames, as even as possible.  Rank r gets (f0, f1): it loads frames [f0, f1) -- the first
    `gap` of them are its halo, the frames its first pairs look back to (the reference's frame queue,
    MoFREAKUtilities.cpp:391-399, 485-487), loaded twice across ranks, never exchanged -- and produces the rows of
    frames f0 + gap .. f1 - 1.  A rank with nothing to do gets (0, 0)."""
    n_pairs = max(T - gap, 0)
    out = []
    for r in range(world_size):
        p0, p1 = n_pairs * r // world_size, n_pairs * (r + 1) // world_size
        out.append((p0, p1 + gap) if p1 > p0 else (0, 0))
    return out


def run_stream_sharded(frames, mofreak: MoFREAKUtilities, rank: int = 0, world_size: int = 1, group=None, on_device: bool = False,
                       chunk_frames: int = 256) -> dict:
    """ONE long gray stream (T, H, W) split over the ranks with a gap-frame halo; rows gathered to rank 0 in rank order,
    i.e. in frame order: byte-identical to a one-rank run over the whole stream.  frames: an array every rank can index
    (a memory-mapped file, or the rank's own decode of its piece's frame range); only [f0, f1) is touched here.
    Dense-grid (shared) keypoint providers only: the pipelined frame loop takes one list for every frame."""
    import time

    import torch

    T, H, W = frames.shape
    gap = mofreak._ctx.params.gap_for_frame_difference
    prov = mofreak.keypoint_provider
    if isinstance(prov, str) or not getattr(prov, "shared", False):
        raise ValueError("run_stream_sharded needs a shared keypoint list (dense grid)")
    f0, f1 = split_stream(T, world_size, gap)[rank]
    distributed = world_size > 1 or group_exists()
    # with a device-side gather the piece's rows never visit the host on their way: frames down, rows straight into the
    # tensor the gather sends from (the one-clip case of mofreak_extract_clips with MOFREAK_ROWS_DEVICE)
    rows_dev = torch.device("cuda", mofreak._device) if (distributed and on_device and torch.cuda.is_available()) else None
    kps = prov(gap, W, H)
    t0 = time.perf_counter()
    rows, d_rows, n_rows = np.zeros(0, api.ROW_DTYPE), None, 0
    if f1 > f0:
        piece = np.ascontiguousarray(frames[f0:f1])
        if rows_dev is not None:
            d_rows = torch.empty(max((f1 - f0 - gap) * len(kps), 1) * 32, dtype=torch.uint8, device=rows_dev)
            n_rows, _ = mofreak._ctx.extract_clips([piece], kps, chunk_frames=chunk_frames, rows_out=d_rows)
            if f0 and n_rows:  # labels run on across the pieces (:401, :488): frame_number is the third 32-bit field of a row
                d_rows.view(torch.int32).view(-1, 8)[:n_rows, 2] += f0
        else:
            rows = mofreak._ctx.extract_stream_pipelined_host(piece, kps, chunk_frames=chunk_frames)
            rows["frame_number"] += f0
            n_rows = len(rows)
    elif rows_dev is not None:
        d_rows = torch.empty(32, dtype=torch.uint8, device=rows_dev)
    t_compute = time.perf_counter() - t0
    t1 = time.perf_counter()
    if distributed:
        if d_rows is not None:
            buf = d_rows
        else:
            buf = torch.from_numpy(rows.view(np.uint8).reshape(-1).copy())
            if on_device:
                buf = buf.to(torch.device("cuda", torch.cuda.current_device()))
        gathered, per_rank = gather_rows(buf, n_rows, dst=0, group=group)
        all_rows = gathered.cpu().numpy().view(api.ROW_DTYPE).reshape(-1) if rank == 0 else None  # the root's one copy to the host
    else:
        all_rows, per_rank = rows, [n_rows]
    out = {"compute_s": t_compute, "gather_s": time.perf_counter() - t1, "frames_here": (f0, f1), "rows_here": int(n_rows),
           "rows_per_rank": per_rank, "rows_in_hbm": d_rows is not None, "distributed": bool(distributed)}
    if rank == 0:
        out["rows"] = all_rows
    return out


# ------------------------------------------------------------------ .mofreak file utilities
CHOP_LINES = 40000  # src/merge_mofreak_files.py:96


def chop_mofreak_file(path: str, out_dir: str, lines_per_file: int = CHOP_LINES) -> list[str]:
    """chop() of the reference's merge_mofreak_files.py:89-112 for one file: a .mofreak file with more than
    `lines_per_file` rows is cut into `<stem>.<k>.mofreak` pieces, k = 0, 1, ...; shorter files are left alone."""
    with open(path, "rb") as f:
        lines = f.read().splitlines(keepends=True)
    if len(lines) <= lines_per_file:
        return []
    os.makedirs(out_dir, exist_ok=True)
    stem = ".".join(os.path.basename(path).split(".")[:-1])
    out = []
    for k, i in enumerate(range(0, len(lines), lines_per_file)):
        name = os.path.join(out_dir, f"{stem}.{k}.mofreak")
        with open(name, "wb") as f:
            f.writelines(lines[i:i + lines_per_file])
        out.append(name)
    return out


def merge_mofreak_files(in_dir: str, out_dir: str, action: str = "TestSequence", first_id: int = 1) -> list[str]:
    """merge() of merge_mofreak_files.py:115-160: files `<group>.<...>.<id>.mofreak` are concatenated per group in id
    order, starting at id `first_id` (the script starts at 1) and stopping at the first missing id, into
    `<group>.mpeg.<action>.mofreak`."""
    groups: dict[str, list[list[str]]] = {}
    for name in sorted(os.listdir(in_dir)):
        if not os.path.isfile(os.path.join(in_dir, name)):
            continue
        parts = name.split(".")
        groups.setdefault(parts[0], []).append(parts)
    os.makedirs(out_dir, exist_ok=True)
    written = []
    for group, files in groups.items():
        out_name = os.path.join(out_dir, f"{group}.mpeg.{action}.mofreak")
        with open(out_name, "wb") as out:
            file_id = first_id
            while True:
                hit = next((p for p in files if len(p) >= 2 and p[-2].isdigit() and int(p[-2]) == file_id), None)
                if hit is None:
                    break
                with open(os.path.join(in_dir, ".".join(hit)), "rb") as f:
                    out.write(f.read())
                file_id += 1
        written.append(out_name)
    return written


def write_rows_binary(path: str, rows: np.ndarray) -> None:
    """Binary sidecar of a .mofreak file: the 32-byte rows as they leave the device (api.ROW_DTYPE), no text round trip."""
    np.save(path, np.ascontiguousarray(rows, api.ROW_DTYPE), allow_pickle=False)


def read_rows_binary(path: str) -> np.ndarray:
    rows = np.load(path, allow_pickle=False)
    if rows.dtype != api.ROW_DTYPE:
        raise ValueError(f"{path}: not a MoFREAK row file")
    return rows
