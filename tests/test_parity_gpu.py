"""GPU parity: the HIP path, called through the C ABI, against the CPU oracle -- bit-exact, row by row of
SURVEY.md section 8(a).  Every test needs a real MI355X and fails (never skips or falls back) without one."""
import os

import numpy as np
import pytest

import mofreak_amd as M
from mofreak_amd import synth

pytestmark = pytest.mark.gpu

CENTERS = [(5, 5), (5, 9), (5, 13), (9, 5), (9, 13), (13, 5), (13, 9), (13, 13)]
OFFSETS = [(-4, 0), (-3, 3), (0, 4), (3, 3), (4, 0), (3, -3), (0, -4), (-3, -3)]


@pytest.fixture(params=["auto", "gather"])
def ctx_path(request, gpu_ctx):
    """The shared context in both path modes: tile kernel (+ gather path for large keypoints) and gather path only."""
    gpu_ctx.set_path(M.api.PATH_GATHER if request.param == "gather" else M.api.PATH_AUTO)
    yield gpu_ctx
    gpu_ctx.set_path(M.api.PATH_AUTO)


def oracle_pairs(oracle, cur, prev, kps, offsets=None, **kw):
    """Oracle descriptors for a (n_pairs,H,W) batch, shared keypoint list or CSR."""
    f = oracle.Freak(**kw)
    descs, valids = [], []
    for p in range(cur.shape[0]):
        kp = kps if offsets is None else kps[offsets[p]:offsets[p + 1]]
        d, v = f.extract_pair(cur[p], prev[p], kp)
        descs.append(d)
        valids.append(v)
    return np.concatenate(descs), np.concatenate(valids)


# ------------------------------------------------------------------ R2: absdiff + integral
@pytest.mark.parametrize("W,H", [(37, 53), (64, 8), (641, 479), (1920, 1080), (5, 3), (1023, 17)])
def test_diff_integral_matches_oracle(gpu_ctx, oracle, W, H):
    rng = np.random.default_rng(W * 1000 + H)
    cur = rng.integers(0, 256, (2, H, W), dtype=np.uint8)
    prev = rng.integers(0, 256, (2, H, W), dtype=np.uint8)
    cur[1], prev[1] = 255, 0  # worst case for the int32 range
    got = gpu_ctx.diff_integral_host(cur, prev)
    for p in range(2):
        assert np.array_equal(got[p], oracle.integral(oracle.absdiff(cur[p], prev[p])))


# ------------------------------------------------------------------ R3: motionInterchangePattern
def test_mip19_known_answers_on_gpu(gpu_ctx, oracle):
    cur, prev, want = [], [], []

    def add(c, p):
        cur.append(np.asarray(c, np.uint8).reshape(361))
        prev.append(np.asarray(p, np.uint8).reshape(361))
        want.append([oracle.mip(cur[-1], prev[-1], x, y) for (x, y) in CENTERS])

    add(np.full(361, 77), np.full(361, 77))       # KAT 1
    add(np.zeros(361), np.full(361, 255))          # KAT 2
    for ci, (x, y) in enumerate(CENTERS):          # KAT 3, 4 at every centre and offset
        for i, (dx, dy) in enumerate(OFFSETS):
            b = (y + dy - 1) * 19 + (x + dx - 1)
            p = np.zeros(361, np.uint8); p[b] = 12; p[b + 1] = 12
            add(np.zeros(361), p)
            assert (want[-1][ci] >> i) & 1 == 0
            p = np.zeros(361, np.uint8); p[b] = 17
            add(np.zeros(361), p)
            assert (want[-1][ci] >> i) & 1 == 1
            p = np.zeros(361, np.uint8); p[b + 6] = 255       # byte 6 of the contiguous strip counts
            add(np.zeros(361), p)
            assert (want[-1][ci] >> i) & 1 == 1
            p = np.zeros(361, np.uint8); p[(y + 1 + dy) * 19 + (x + dx)] = 255  # 3rd row of a true 3x3 does not
            add(np.zeros(361), p)
            assert (want[-1][ci] >> i) & 1 == 0
    got = gpu_ctx.mip19_host(np.stack(cur), np.stack(prev))
    assert np.array_equal(got, np.asarray(want, np.uint8))
    assert got[0].tolist() == [0] * 8 and got[1].tolist() == [255] * 8


def test_mip19_random_buffers(gpu_ctx, oracle):
    rng = np.random.default_rng(21)
    n = 3000
    yy, xx = np.mgrid[0:19, 0:19]
    gx, gy = rng.uniform(-3, 3, (2, n, 1, 1))  # per-buffer ramps: SSDs land on both sides of THETA
    base = rng.integers(60, 200, (n, 1, 1)) + gx * xx + gy * yy
    cur = (base + rng.integers(-2, 3, (n, 19, 19))).clip(0, 255).astype(np.uint8).reshape(n, 361)
    prev = (base + rng.integers(-3, 4, (n, 19, 19))).clip(0, 255).astype(np.uint8).reshape(n, 361)
    prev[: n // 3] = np.roll(cur[: n // 3], 3, axis=1)
    got = gpu_ctx.mip19_host(cur, prev)
    want = np.array([[oracle.mip(cur[k], prev[k], x, y) for (x, y) in CENTERS] for k in range(n)], np.uint8)
    assert np.array_equal(got, want)
    assert 0.05 < np.unpackbits(got).mean() < 0.95


# ------------------------------------------------------------------ R4: ROI + cv::resize
def test_roi19_matches_oracle_resize_for_every_small_side(gpu_ctx, oracle):
    W, H = 700, 500
    fr = synth.synth_stack(6, W, H)
    rng = np.random.default_rng(8)
    cur = (fr[5].astype(np.int64) + rng.integers(-40, 41, (H, W))).clip(0, 255).astype(np.uint8)
    prev = fr[0]
    sizes = np.concatenate([np.arange(7, 61, dtype=np.float32), np.float32([7.5, 11.01, 12.99, 31.2, 32.0, 32.5, 33.0, 47.3, 64.9, 70.0])])
    kps = np.stack([np.full(len(sizes), 350.6, np.float32), np.full(len(sizes), 251.3, np.float32), sizes], 1)
    got = gpu_ctx.roi19_host(cur, prev, kps)
    for k, s in enumerate(sizes):
        x, y = int(kps[k, 0]), int(kps[k, 1])
        tl_x, tl_y, L = x - int(s) // 2, y - int(s) // 2, int(np.ceil(s))
        assert np.array_equal(got[k, 0], oracle.resize_linear(cur[tl_y:tl_y + L, tl_x:tl_x + L])), s
        assert np.array_equal(got[k, 1], oracle.resize_linear(prev[tl_y:tl_y + L, tl_x:tl_x + L])), s


# ------------------------------------------------------------------ R6: FREAK internals
def test_freak_scale_theta_directions_match_oracle(ctx_path, oracle):
    W, H = 640, 480
    fr = synth.synth_stack(6, W, H)
    rng = np.random.default_rng(17)
    kps = synth.random_keypoints(rng, 4000, W, H)
    kps[:500, 2] = rng.uniform(6.0, 60.0, 500).astype(np.float32)
    info = ctx_path.freak_info_host(fr[5], fr[0], kps)
    f = oracle.Freak()
    valid, _, theta, dirs = f.compute(oracle.absdiff(fr[5], fr[0]), kps)
    assert 0.2 < valid.mean() < 0.95
    want_idx = np.array([f.scale_index(s) for s in kps[:, 2]])
    assert np.array_equal(info[:, 0], want_idx)
    assert np.array_equal(info[:, 1], theta)                  # -1 for erased keypoints on both sides
    assert np.array_equal(info[valid == 1, 2:], dirs[valid == 1])
    assert len(np.unique(theta[valid == 1])) > 100


def test_theta_index_device_matches_oracle(gpu_ctx, oracle):
    rng = np.random.default_rng(33)
    d = np.concatenate([
        rng.integers(-6500, 6501, (400000, 2)),
        rng.integers(-40, 41, (100000, 2)),
        np.array([[0, 0], [1, 0], [-1, 0], [0, 1], [0, -1], [1, 1], [-1, -1], [-1, 1], [1, -1], [6000, -1], [-6000, -1], [-6000, 1]]),
        np.stack(np.meshgrid(np.arange(-60, 61), np.arange(-60, 61)), -1).reshape(-1, 2)]).astype(np.int32)
    got = gpu_ctx.theta_index_host(d)
    want = np.array([oracle.theta_index(int(a), int(b)) for a, b in d], np.int32)
    assert np.array_equal(got, want)
    assert got.min() == 0 and got.max() == 255


# ------------------------------------------------------------------ the composed path (R2+R4+R5+R6)
def test_c2_dense_grid_bit_exact(ctx_path, oracle):
    """BASELINE config 2 shape (640x480, 16-px grid, size 12), 12 pairs of a stack."""
    c = synth.CONFIGS["C2"]
    fr = synth.synth_stack(17, c["W"], c["H"])
    kps = synth.config_grid("C2")
    assert len(kps) == 875
    cur, prev = fr[5:], fr[:-5]
    desc, valid = ctx_path.extract_pairs_host(cur, prev, kps)
    want_d, want_v = oracle_pairs(oracle, cur, prev, kps)
    assert valid.all() and np.array_equal(valid, want_v)
    assert np.array_equal(desc, want_d)
    ones_app, ones_mot = np.unpackbits(desc[:, :8]).mean(), np.unpackbits(desc[:, 8:]).mean()
    assert 0.3 < ones_app < 0.7 and 0.3 < ones_mot < 0.8  # non-degenerate data


def test_c2_all_1000_pairs_bit_exact(gpu_ctx, oracle):
    """BASELINE.md's gate as it is written: byte-identical on C2 -- ALL 1000 pairs of the 1005-frame 640x480 stack, the
    875-keypoint 16-px grid (875 000 descriptors), and the mixed-size variant of the same grid (sizes 8.4 .. 40.5 cycling
    over the grid: tile path and gather path, some keypoints erased by FREAK's border filter) on every 8th pair."""
    import torch
    from concurrent.futures import ThreadPoolExecutor
    c = synth.CONFIGS["C2"]
    W, H, n_pairs = c["W"], c["H"], 1000
    fr = synth.synth_stack(n_pairs + 5, W, H)
    kps = synth.config_grid("C2")
    assert len(kps) == 875
    d_fr, d_kps = torch.from_numpy(fr).cuda(), torch.from_numpy(kps).cuda()
    desc = torch.empty((n_pairs * len(kps), 16), dtype=torch.uint8, device="cuda")
    valid = torch.empty(n_pairs * len(kps), dtype=torch.uint8, device="cuda")
    gpu_ctx.extract_pairs(d_fr[5:], d_fr[:n_pairs], W, H, n_pairs, d_kps, desc, valid)
    gpu_ctx.synchronize()
    gpu_ctx.check_status()
    got_d, got_v = desc.cpu().numpy().reshape(n_pairs, len(kps), 16), valid.cpu().numpy().reshape(n_pairs, len(kps))
    f = oracle.Freak()
    workers = max(1, min(16, len(os.sched_getaffinity(0))))

    def one(p):
        d, v = f.extract_pair(fr[p + 5], fr[p], kps)  # the C oracle releases the GIL
        return int((got_v[p] != v).sum()) + int((got_d[p] != d).any(axis=1).sum())

    with ThreadPoolExecutor(workers) as ex:
        bad = sum(ex.map(one, range(n_pairs)))
    assert bad == 0, f"{bad} of {n_pairs * len(kps)} descriptors differ from the oracle"
    assert got_v.all()
    # the mixed-size variant
    mixed = kps.copy()
    mixed[:, 2] = np.float32([8.4, 12.0, 18.0, 27.0, 40.5])[np.arange(len(kps)) % 5]
    sel = np.arange(0, n_pairs, 8)
    d2, v2 = gpu_ctx.extract_pairs_host(fr[sel + 5], fr[sel], mixed)
    d2, v2 = d2.reshape(len(sel), len(kps), 16), v2.reshape(len(sel), len(kps))

    def one_mixed(i):
        d, v = f.extract_pair(fr[sel[i] + 5], fr[sel[i]], mixed)
        return int((v2[i] != v).sum()) + int((d2[i] != d).any(axis=1).sum())

    with ThreadPoolExecutor(workers) as ex:
        bad = sum(ex.map(one_mixed, range(len(sel))))
    assert bad == 0 and 0.5 < v2.mean() < 1.0


def test_mixed_sizes_random_positions_bit_exact(ctx_path, oracle):
    """Mixed sizes {8.4,12,18,27,40.5}, fractional coordinates, keypoints over the whole frame (some erased)."""
    W, H = 640, 480
    fr = synth.synth_stack(8, W, H)
    rng = np.random.default_rng(5)
    kps = synth.random_keypoints(rng, 6000, W, H)
    kps[:50, 2] = np.float32([0.0, 1e-8, 5.0, 6.9, 7.0, 100.0, 107.0, 150.0, 2.5e5, np.inf] * 5)
    kps[50:60, 0] = np.float32(np.nan)
    cur, prev = fr[5:], fr[:3]
    desc, valid = ctx_path.extract_pairs_host(cur, prev, kps)
    want_d, want_v = oracle_pairs(oracle, cur, prev, kps)
    assert np.array_equal(valid, want_v)
    assert np.array_equal(desc, want_d)
    assert 0.3 < valid.mean() < 0.9
    assert (desc[valid == 0] == 0).all()
    ctx_path.check_status()  # no ROI ever leaves the image for keypoints that survive FREAK's border filter


def test_gather_path_in_chunks_of_pairs_behind_the_tile_kernel(native_lib, oracle):
    """Ragged keypoint lists of mixed sizes over 7 pairs with the gather path's integral workspace limited to 3 pairs: every chunk
    takes its own piece of the band-sorted list (pairs 0-2, 3-5, 6), the tile kernel the dense small keypoints."""
    W, H = 320, 240
    fr = synth.synth_stack(12, W, H)
    rng = np.random.default_rng(17)
    counts = [900, 0, 1500, 40, 700, 1, 1200]
    kps = np.concatenate([synth.random_keypoints(rng, n, W, H, sizes=(7.0, 9.5, 12.0, 16.0, 27.0, 40.0)) for n in counts] + [np.zeros((0, 3), np.float32)])
    offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    cur, prev = fr[5:12], fr[0:7]
    want_d, want_v = oracle_pairs(oracle, cur, prev, kps, offsets=offs)
    with M.Context(0) as ctx:
        ctx.reserve(W, H, 3)
        desc, valid = ctx.extract_pairs_host(cur, prev, kps, kp_offsets=offs)
        ctx.check_status()
    assert np.array_equal(valid, want_v) and np.array_equal(desc, want_d)
    assert 0.2 < valid.mean() < 0.95


@pytest.mark.parametrize("mode", [M.BITS_NATURAL, M.BITS_SSE_SIGNED])
def test_other_bit_modes(native_lib, oracle, mode):
    W, H = 320, 240
    fr = synth.synth_stack(6, W, H)
    kps = synth.random_keypoints(np.random.default_rng(1), 1500, W, H, sizes=(7.0, 9.0, 12.0))
    with M.Context(0, freak_bit_mode=mode) as ctx:
        desc, valid = ctx.extract_pairs_host(fr[5:6], fr[0:1], kps)
    want_d, want_v = oracle_pairs(oracle, fr[5:6], fr[0:1], kps, bit_mode=mode)
    assert np.array_equal(valid, want_v) and np.array_equal(desc, want_d)
    base_d, _ = oracle_pairs(oracle, fr[5:6], fr[0:1], kps)
    assert not np.array_equal(base_d[:, :8], want_d[:, :8]) and np.array_equal(base_d[:, 8:], want_d[:, 8:])


def test_orientation_and_scale_normalisation_flags(native_lib, oracle):
    W, H = 320, 240
    fr = synth.synth_stack(6, W, H)
    kps = synth.random_keypoints(np.random.default_rng(2), 800, W, H, sizes=(7.0, 12.0, 20.0))
    for on, sn in [(0, 1), (1, 0), (0, 0)]:
        with M.Context(0, freak_orientation_normalized=on, freak_scale_normalized=sn) as ctx:
            desc, valid = ctx.extract_pairs_host(fr[5:6], fr[0:1], kps)
        want_d, want_v = oracle_pairs(oracle, fr[5:6], fr[0:1], kps, orientation_normalized=bool(on), scale_normalized=bool(sn))
        assert np.array_equal(valid, want_v) and np.array_equal(desc, want_d), (on, sn)


@pytest.mark.parametrize("counts", [[0, 37, 1, 0, 250, 64], [0, 900, 1, 0, 1500, 700]])
def test_csr_ragged_keypoint_lists(ctx_path, oracle, counts):
    """One keypoint list per pair.  The first set is sparse (fewer than 24 keypoints per tile on average: the whole call goes
    to the gather path, as a detector's output does), the second dense enough for the tile kernel."""
    W, H = 320, 240
    fr = synth.synth_stack(11, W, H)
    rng = np.random.default_rng(12)
    offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    kps = synth.random_keypoints(rng, int(offs[-1]), W, H, sizes=(7.0, 8.4, 12.0))
    cur, prev = fr[5:], fr[:6]
    desc, valid = ctx_path.extract_pairs_host(cur, prev, kps, kp_offsets=offs)
    want_d, want_v = oracle_pairs(oracle, cur, prev, kps, offs)
    assert np.array_equal(valid, want_v) and np.array_equal(desc, want_d)


def test_empty_inputs(gpu_ctx):
    z = np.zeros((2, 64, 64), np.uint8)
    d, v = gpu_ctx.extract_pairs_host(z, z, np.zeros((0, 3), np.float32))
    assert d.shape == (0, 16) and v.shape == (0,)
    d, v = gpu_ctx.extract_pairs_host(z[:0], z[:0], np.float32([[32, 32, 7]]))
    assert d.shape == (0, 16)
    rows = gpu_ctx.extract_stream_host(z, np.float32([[32, 32, 7]]))  # T=2 <= gap
    assert len(rows) == 0


def test_bad_arguments_are_reported(gpu_ctx):
    z = np.zeros((1, 64, 64), np.uint8)
    kp = np.float32([[32, 32, 7]])
    out_d, out_v = np.zeros((1, 16), np.uint8), np.zeros(1, np.uint8)
    with pytest.raises(M.MoFREAKError) as e:
        gpu_ctx.extract_pairs(z, z, 64, 64, 1, kp, out_d, out_v, row_stride=32)
    assert e.value.code == -1 and "row_stride" in str(e.value)
    with pytest.raises(M.MoFREAKError):
        gpu_ctx.extract_pairs_host(z, z, kp, kp_offsets=np.array([0, 5], np.int64))  # offsets[n_pairs] != n_kp
    with pytest.raises(M.MoFREAKError):
        gpu_ctx.extract_pairs(z, z, 0, 64, 1, kp, out_d, out_v)
    with pytest.raises(M.MoFREAKError) as e:  # the gather path's 32-bit byte offsets: rows of 2^23 bytes and more are refused
        gpu_ctx.extract_pairs(z, z, 64, 64, 1, kp, out_d, out_v, row_stride=1 << 23)
    assert e.value.code == -4 and "row_stride" in str(e.value)


def test_a_roi_that_leaves_the_image_is_reported_where_the_reference_throws(gpu_ctx, oracle):
    """size 1100 at the centre of a 1080p pair: FREAK keeps the keypoint (pattern size 339 < 540), its MIP ROI -- side 1100
    from y - 550 -- leaves the image.  The reference throws cv::Exception out of cv::Mat::operator()(Rect)
    (MoFREAKUtilities.cpp:296-297; the oracle returns -1, mofreak_oracle.c:433): here the keypoint is invalid with a zero
    descriptor, its neighbours in the list are untouched, and mofreak_check_status says MOFREAK_ERR_ROI -- once."""
    fr = synth.synth_stack(6, 1920, 1080)
    cur, prev = fr[5:6], fr[0:1]
    kps = np.float32([[960, 540, 12], [960, 540, 1100], [700, 400, 12]])
    gpu_ctx.check_status()  # nothing pending from earlier tests
    d, v = gpu_ctx.extract_pairs_host(cur, prev, kps)
    wd, wv = oracle_pairs(oracle, cur, prev, kps)
    assert wv.tolist() == [1, 0, 1] and v.tolist() == [1, 0, 1]
    assert d.tobytes() == wd.tobytes() and not d[1].any()
    with pytest.raises(M.MoFREAKError) as e:
        gpu_ctx.check_status()
    assert e.value.code == M.api.ERR_ROI and "ROI" in str(e.value)
    gpu_ctx.check_status()  # reading the status clears it


def test_a_roi_wider_than_the_resize_tables_is_refused(gpu_ctx, oracle):
    """ceil(size) = 2049 > 2048, the largest ROI side the host-built cv::resize tap tables cover: MOFREAK_ERR_UNSUPPORTED
    (status bit 2), the keypoint invalid -- as in the oracle, whose ROI test fails for it too."""
    fr = synth.synth_stack(6, 1920, 1080)
    cur, prev = fr[5:6], fr[0:1]
    kps = np.float32([[960, 540, 2049], [960, 540, 12]])
    gpu_ctx.check_status()
    d, v = gpu_ctx.extract_pairs_host(cur, prev, kps)
    wd, wv = oracle_pairs(oracle, cur, prev, kps)
    assert wv.tolist() == [0, 1] and v.tolist() == [0, 1] and d.tobytes() == wd.tobytes()
    with pytest.raises(M.MoFREAKError) as e:
        gpu_ctx.check_status()
    assert e.value.code == M.api.ERR_UNSUPPORTED
    gpu_ctx.check_status()


def test_strides_and_unaligned_frames(ctx_path, oracle):
    """row_stride > W, pair_stride with padding, and a frame base that is not 4-byte aligned."""
    W, H, n = 203, 150, 3
    fr = synth.synth_stack(n + 5, 256, 160)
    rs, ps = 256 + 3, (256 + 3) * 160 + 7
    buf_c = np.zeros(ps * n + 1, np.uint8)
    buf_p = np.zeros(ps * n + 1, np.uint8)
    cur = np.zeros((n, H, W), np.uint8)
    prev = np.zeros((n, H, W), np.uint8)
    for p in range(n):
        cur[p], prev[p] = fr[p + 5, :H, :W], fr[p, :H, :W]
        for y in range(H):
            buf_c[1 + p * ps + y * rs: 1 + p * ps + y * rs + W] = cur[p, y]
            buf_p[1 + p * ps + y * rs: 1 + p * ps + y * rs + W] = prev[p, y]
    kps = synth.random_keypoints(np.random.default_rng(3), 900, W, H, sizes=(7.0, 8.4, 12.0))
    import torch
    dc, dp = torch.from_numpy(buf_c).cuda(), torch.from_numpy(buf_p).cuda()
    dk = torch.from_numpy(kps).cuda()
    out_d = torch.zeros((n * len(kps), 16), dtype=torch.uint8, device="cuda")
    out_v = torch.zeros(n * len(kps), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()  # the context runs on its own stream: torch's fills above must have landed
    ctx_path.extract_pairs(dc[1:], dp[1:], W, H, n, dk, out_d, out_v, row_stride=rs, pair_stride=ps)
    ctx_path.synchronize()
    want_d, want_v = oracle_pairs(oracle, cur, prev, kps)
    assert np.array_equal(out_v.cpu().numpy(), want_v) and np.array_equal(out_d.cpu().numpy(), want_d)


def test_device_pointers_on_torch_stream(native_lib, oracle):
    """The zero-copy path: torch tensors in HBM, work queued on torch's current stream."""
    import torch
    W, H = 640, 480
    fr = synth.synth_stack(9, W, H)
    kps = synth.config_grid("C2")
    with M.Context(0) as ctx:
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            ctx.use_torch_stream()
            frames = torch.from_numpy(fr).cuda()
            dk = torch.from_numpy(kps).cuda()
            n_pairs = 4
            out_d = torch.empty((n_pairs * len(kps), 16), dtype=torch.uint8, device="cuda")
            out_v = torch.empty(n_pairs * len(kps), dtype=torch.uint8, device="cuda")
            ctx.extract_pairs(frames[5:], frames[:4], W, H, n_pairs, dk, out_d, out_v)
            rows = torch.zeros(n_pairs * len(kps) * 32, dtype=torch.uint8, device="cuda")
            n = ctx.compact_rows(dk, n_pairs, 4, out_d, out_v, rows)
        s.synchronize()
        ctx.set_stream(None)
    want_d, want_v = oracle_pairs(oracle, fr[5:], fr[:4], kps)
    assert np.array_equal(out_d.cpu().numpy(), want_d) and np.array_equal(out_v.cpu().numpy(), want_v)
    assert n == int(want_v.sum())
    r = rows.cpu().numpy().view(M.ROW_DTYPE)[:n]
    assert np.array_equal(np.concatenate([r["appearance"], r["motion"]], 1), want_d[want_v == 1])
    assert sorted(set(r["frame_number"].tolist())) == [4, 5, 6, 7]


# ------------------------------------------------------------------ R1 + R7: streams and rows
def test_stream_rows_and_text_match_oracle(ctx_path, oracle):
    """computeMoFREAKFromFile's frame loop on a 12-frame stack: pairing, frame labels, erase order, text."""
    W, H, T = 320, 240, 12
    fr = synth.synth_stack(T, W, H)
    rng = np.random.default_rng(44)
    counts = rng.integers(0, 300, T - 5)
    offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    kps = synth.random_keypoints(rng, int(offs[-1]), W, H, sizes=(7.0, 8.4, 12.0, 18.0))
    rows = ctx_path.extract_stream_host(fr, kps, kp_offsets=offs)
    want = oracle.Freak().extract_stream(fr, kps, offs)
    assert len(rows) == len(want) > 100
    assert rows.tobytes() == want.tobytes()
    assert M.format_rows(rows) == oracle.format_rows(want)
    assert rows["frame_number"].min() == 4 and rows["frame_number"].max() == T - 2
    assert (np.diff(rows["frame_number"]) >= 0).all()
    # shared keypoint list form
    grid = synth.dense_grid(W, H, 16, 7.0, 23)
    rows2 = ctx_path.extract_stream_host(fr, grid)
    offs2 = np.arange(T - 5 + 1, dtype=np.int64) * len(grid)
    want2 = oracle.Freak().extract_stream(fr, np.tile(grid, (T - 5, 1)), offs2)
    assert rows2.tobytes() == want2.tobytes() and len(rows2) == (T - 5) * len(grid)


def test_compact_rows_capacity_error(gpu_ctx):
    W, H = 320, 240
    fr = synth.synth_stack(7, W, H)
    grid = synth.dense_grid(W, H, 16, 7.0, 23)
    rows = np.zeros(10, M.ROW_DTYPE)
    with pytest.raises(M.MoFREAKError) as e:
        gpu_ctx.extract_stream(fr, 7, W, H, grid, rows)
    assert e.value.code == -7


# ------------------------------------------------------------------ full-size properties (BASELINE config 3)
def test_c3_full_resolution_pairs_bit_exact_and_chunk_independent(ctx_path, oracle):
    """1920x1080, 8-px grid (29 106 keypoints/pair): bit-exact vs the oracle on every pair of a 12-pair batch
    (spans two integral chunks), and a pair described alone gives the same bytes as inside the batch."""
    c = synth.CONFIGS["C3"]
    W, H = c["W"], c["H"]
    n_pairs = 12
    fr = synth.synth_stack(n_pairs + 5, W, H)
    kps = synth.config_grid("C3")
    assert len(kps) == 29106
    cur, prev = fr[5:], fr[:n_pairs]
    desc, valid = ctx_path.extract_pairs_host(cur, prev, kps)
    assert valid.all()
    want_d, want_v = oracle_pairs(oracle, cur, prev, kps)
    assert np.array_equal(desc, want_d) and np.array_equal(valid, want_v)
    d1, v1 = ctx_path.extract_pairs_host(cur[9:10], prev[9:10], kps)
    assert np.array_equal(d1, desc[9 * len(kps):10 * len(kps)])
    # idempotence: same call again, same bytes
    d2, _ = ctx_path.extract_pairs_host(cur, prev, kps)
    assert np.array_equal(d2, desc)


def test_identical_frames_give_zero_motion_and_ff_appearance(ctx_path):
    """cur == prev: the difference image is 0 -> every FREAK box mean is equal -> mode S bytes 0xFF (KAT 7),
    theta 0; every MIP SSD is between a strip and shifted copies of the same frame."""
    W, H = 640, 480
    fr = np.full((1, H, W), 91, np.uint8)
    kps = synth.config_grid("C2")
    desc, valid = ctx_path.extract_pairs_host(fr, fr, kps)
    assert valid.all() and (desc[:, :8] == 0xFF).all() and (desc[:, 8:] == 0).all()


# ------------------------------------------------------------------ tile-path specifics
def test_crowded_tiles_and_tile_borders(gpu_ctx, oracle):
    """More keypoints in one 128x64 tile than one batch (128): a 3-px grid puts ~900 in a tile, several batches; mixed
    ROI sides inside a batch; sizes right below and above the tile path's limit; frame not a multiple of the tile."""
    W, H = 331, 275
    fr = synth.synth_stack(7, W, H)
    rng = np.random.default_rng(77)
    xs, ys = np.meshgrid(np.arange(30, W - 30, 3), np.arange(30, H - 30, 3))
    n = xs.size
    kps = np.stack([xs.ravel() + rng.choice([0, 0.25, 0.5, 0.999], n), ys.ravel() + rng.choice([0, 0.5], n),
                    rng.choice(np.float32([7.0, 7.5, 9.0, 11.3, 12.0, 12.5, 12.56, 12.6, 13.99, 14.8, 14.9, 15.0, 16.0, 17.5]), n)], 1).astype(np.float32)
    rng.shuffle(kps)
    cur, prev = fr[5:], fr[:2]
    desc, valid = gpu_ctx.extract_pairs_host(cur, prev, kps)
    want_d, want_v = oracle_pairs(oracle, cur, prev, kps)
    assert np.array_equal(valid, want_v) and np.array_equal(desc, want_d)
    assert valid.sum() > 5000
    gpu_ctx.check_status()
    # keypoints sitting exactly on tile borders (x = 128k and, the earlier tile width, 96k; y = 64k) and one pixel either side
    W, H = 640, 480
    fr = synth.synth_stack(6, W, H)
    pts = [(x + dx, y + dy) for x in (96, 128, 192, 256, 288, 384) for y in (64, 128, 192, 256, 320) for dx in (-1, -0.001, 0, 0.5, 1) for dy in (-1, 0, 0.75)]
    kps = np.float32([[x, y, s] for (x, y) in pts for s in (7.0, 12.0, 12.5, 14.5)])
    desc, valid = gpu_ctx.extract_pairs_host(fr[5:6], fr[0:1], kps)
    want_d, want_v = oracle_pairs(oracle, fr[5:6], fr[0:1], kps)
    assert valid.all() and np.array_equal(valid, want_v) and np.array_equal(desc, want_d)


@pytest.mark.parametrize("sizes", [(7.0,), (7.0, 7.7), (9.0, 9.9), (12.0,), (7.0, 9.0, 12.0, 14.5)])
def test_integer_keypoints_every_halo_and_scale(gpu_ctx, oracle, sizes):
    """Keypoints at integer coordinates take the tile kernel's fixed-offset boxes (BoxInt) -- one scale per tile or
    several, every halo size the pattern sizes select (24, 32, 40; size 14.5 is the gather path's), tiles cut by the image border, groups of
    four that do not fill up, a frame width that is not a multiple of 8 (byte-wise staging) -- and must give the
    float expressions' bits, including the pattern points that sit too close to a rounding boundary."""
    for W, H in ((417, 301), (640, 200)):
        fr = synth.synth_stack(7, W, H)
        rng = np.random.default_rng(len(sizes) * 1000 + W)
        kps = synth.random_keypoints(rng, 3001, W, H, sizes=sizes, integer_xy=True)
        cur, prev = fr[5:], fr[:2]
        desc, valid = gpu_ctx.extract_pairs_host(cur, prev, kps)
        want_d, want_v = oracle_pairs(oracle, cur, prev, kps)
        assert np.array_equal(valid, want_v) and np.array_equal(desc, want_d)
        assert valid.sum() > 2000
    gpu_ctx.check_status()


def test_integer_keypoints_on_a_frame_wider_than_2048(gpu_ctx, oracle):
    """The rounding margin of the fixed-offset boxes depends on the coordinate range (one more bit of float error
    per binade): a 4200-px-wide frame uses the wider margin, a dense integer grid over all of it stays bit-exact."""
    W, H = 4200, 150
    fr = synth.synth_stack(6, W, H)
    xs, ys = np.meshgrid(np.arange(40, W - 40, 7), np.arange(40, H - 40, 9))
    kps = np.stack([xs.ravel(), ys.ravel(), np.full(xs.size, 12.0)], 1).astype(np.float32)
    desc, valid = gpu_ctx.extract_pairs_host(fr[5:6], fr[0:1], kps)
    want_d, want_v = oracle_pairs(oracle, fr[5:6], fr[0:1], kps)
    assert valid.all() and np.array_equal(valid, want_v) and np.array_equal(desc, want_d)


def test_bounds_checking_build_of_the_tile_kernel(oracle):
    """SURVEY.md section 5 (bounds asserts in debug kernels): libmofreak_hip_debug.so is the same source with every
    LDS access of the tile kernel checked against the workgroup's allocation and every descriptor store against the
    output's extent.  It must report nothing on inputs that reach every code path -- integer and fractional keypoints,
    every halo, crowded and border tiles, a width that is not a multiple of 8 -- and still give the oracle's bytes."""
    import os
    import subprocess
    import sys
    from mofreak_amd import build
    build.build_native(debug=True)  # rebuilt whenever a source is newer than it (same check as the product build)
    code = r"""
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import mofreak_amd as M
from mofreak_amd import synth
import oracle_lib
f = oracle_lib.Freak()
assert M.api.load().mofreak_build_flags() == 1, "not the bounds-checking build"
with M.Context(0) as ctx:
    for W, H, integer in ((417, 301, True), (640, 480, False), (331, 275, False)):
        fr = synth.synth_stack(7, W, H)
        rng = np.random.default_rng(W)
        kps = synth.random_keypoints(rng, 6000, W, H, sizes=(7.0, 9.0, 12.0, 14.5, 18.0), integer_xy=integer)
        desc, valid = ctx.extract_pairs_host(fr[5:], fr[:2], kps)
        ctx.check_status()
        for p in range(2):
            d, v = f.extract_pair(fr[5 + p], fr[p], kps)
            assert np.array_equal(desc[p * len(kps):(p + 1) * len(kps)], d) and np.array_equal(valid[p * len(kps):(p + 1) * len(kps)], v)
print("debug build ok")
""" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MOFREAK_HIP_LIBRARY=build.DEBUG_LIB_PATH)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "debug build ok" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


# ------------------------------------------------------------------ pipelined long streams (BASELINE config 5)
@pytest.mark.parametrize("pinned", [False, True])
def test_pipelined_stream_rows_equal_the_whole_stack_call(gpu_ctx, oracle, pinned):
    """TRECVID-shaped (720x576) stream walked in chunks with copy/compute overlap: rows byte-identical to
    mofreak_extract_stream on the whole stack (and so to the oracle), for chunk sizes that do and do not divide the
    stream, page-locked or ordinary host memory, and a stream shorter than one chunk."""
    c = synth.CONFIGS["C5"]
    W, H, T = c["W"], c["H"], 47
    kps = synth.config_grid("C5")
    assert len(kps) == 5103
    src = synth.synth_stack(T, W, H)
    frames = gpu_ctx.host_alloc((T, H, W)) if pinned else src
    if pinned:
        frames[:] = src
    want = gpu_ctx.extract_stream_host(src, kps)
    assert len(want) == (T - 5) * len(kps)
    for chunk in (12, 16, 47, 256):
        rows_buf = gpu_ctx.host_alloc(((T - 5) * len(kps),), M.api.ROW_DTYPE) if pinned else None
        got = gpu_ctx.extract_stream_pipelined_host(frames, kps, chunk_frames=chunk, rows_out=rows_buf)
        assert got.tobytes() == want.tobytes(), f"chunk {chunk}"
        if pinned:
            gpu_ctx.host_free(rows_buf)
    if pinned:
        gpu_ctx.host_free(frames)
    f = oracle.Freak()
    n_pairs = 6
    offs = np.arange(n_pairs + 1, dtype=np.int64) * len(kps)
    assert want[: n_pairs * len(kps)].tobytes() == f.extract_stream(src[: n_pairs + 5], np.tile(kps, (n_pairs, 1)), offs).tobytes()
    # too small a rows buffer is reported, not overrun
    small = np.zeros(1000, M.api.ROW_DTYPE)
    with pytest.raises(M.api.MoFREAKError) as e:
        gpu_ctx.extract_stream_pipelined_host(src, kps, chunk_frames=16, rows_out=small)
    assert e.value.code == M.api.ERR_CAPACITY
    assert str((T - 5) * len(kps)) in str(e.value)  # the error names the size a retry needs


# ------------------------------------------------------------------ many clips in one call (BASELINE config 4)
def _clip_set(rng, W, H, lengths):
    pool = synth.synth_stack(max(lengths) + 7, W, H)
    return [np.ascontiguousarray(pool[(3 * i) % 7: (3 * i) % 7 + t]) for i, t in enumerate(lengths)]


@pytest.mark.parametrize("pinned", [False, True])
def test_chunk_pushes_into_a_stream_equal_the_whole_stack_call(gpu_ctx, pinned):
    """mofreak_stream_push_frames: a stream fed in chunks of any length (shorter than the gap, one frame, hundreds; windows
    inside a chunk that cut it in the middle; single-frame pushes in between) keeps only its last gap frames on the device
    and yields, piece by piece, the rows of mofreak_extract_stream over the whole stack -- frame numbers running on."""
    W, H, T = 320, 240, 131
    frames = synth.synth_stack(T, W, H)
    kps = synth.dense_grid(W, H, 16, 12.0, 38)
    want = gpu_ctx.extract_stream_host(frames, kps)
    assert len(want) == (T - 5) * len(kps)
    src = frames
    if pinned:
        src = gpu_ctx.host_alloc(frames.shape)
        src[:] = frames
    rows_buf = gpu_ctx.host_alloc((60 * len(kps),), M.api.ROW_DTYPE) if pinned else None
    got = []
    with gpu_ctx.open_stream(W, H, use_detector=False) as st:
        t = 0
        for n, window in [(3, 0), (1, 0), (4, 0), (40, 0), (1, -1), (57, 11), (2, 0), (1, -1), (22, 7)]:
            if window < 0:  # a frame-at-a-time push in between
                got.append(st.push(frames[t], kps))
            else:
                got.append(st.push_frames(src[t:t + n], kps, chunk_frames=window, rows_out=rows_buf).copy())
            t += n
            assert st.frames == t
        assert t == T
        # a rows buffer that is too small: reported, the frames consumed all the same
        with pytest.raises(M.MoFREAKError) as e:
            st.push_frames(np.ascontiguousarray(frames[:9]), kps, rows_out=np.zeros(5, M.api.ROW_DTYPE))
        assert e.value.code == M.api.ERR_CAPACITY and st.frames == T + 9
    got = np.concatenate(got)
    assert got.tobytes() == want.tobytes()
    assert [len(g) for g in (got[got["frame_number"] == 4], got[got["frame_number"] == T - 2])] == [len(kps), len(kps)]


@pytest.mark.parametrize("chunk", [0, 9, 23, 64])
def test_clips_in_one_call_equal_one_call_per_clip(gpu_ctx, chunk):
    """mofreak_extract_clips: the dataset loop's body (main.cpp:862-921) for many videos at once.  Rows and per-clip
    offsets equal those of mofreak_extract_stream clip by clip -- pairs never cross a clip boundary, frame numbers restart
    in every clip, clips no longer than the gap (and empty ones) yield nothing -- whatever the window size (windows
    that cut clips in the middle, windows that hold several clips), page-locked and ordinary clips mixed."""
    W, H = 320, 240
    rng = np.random.default_rng(chunk)
    lengths = [20, 6, 5, 0, 31, 1, 12, 7, 40, 3, 9]
    clips = _clip_set(rng, W, H, lengths)
    pinned = []
    for i in (0, 4, 7):  # some of them in page-locked memory
        buf = gpu_ctx.host_alloc(clips[i].shape)
        buf[:] = clips[i]
        clips[i] = buf
        pinned.append(buf)
    kps = synth.random_keypoints(rng, 300, W, H, sizes=(7.0, 8.4, 12.0))  # some of them erased near the border
    want = [gpu_ctx.extract_stream_host(c, kps) if len(c) else np.zeros(0, M.api.ROW_DTYPE) for c in clips]
    rows, offs = gpu_ctx.extract_clips(clips, kps, chunk_frames=chunk)
    assert offs[0] == 0 and offs[-1] == len(rows) == sum(len(w) for w in want)
    for i, w in enumerate(want):
        assert rows[offs[i]:offs[i + 1]].tobytes() == w.tobytes(), f"clip {i}"
        if len(w):
            assert w["frame_number"].min() == 4 and w["frame_number"].max() == lengths[i] - 2
    assert len(want[1]) > 0 and len(want[2]) == 0 and len(want[5]) == 0
    # rows into a device tensor (what the RCCL gather takes): same bytes
    import torch
    d_rows = torch.zeros(len(rows) * 32 + 64, dtype=torch.uint8, device="cuda")
    n, offs2 = gpu_ctx.extract_clips(clips, kps, chunk_frames=chunk, rows_out=d_rows)
    assert n == len(rows) and np.array_equal(offs, offs2)
    assert d_rows[: n * 32].cpu().numpy().tobytes() == rows.tobytes() and int(d_rows[n * 32:].sum()) == 0
    # a rows buffer that is too small: reported with the size needed, nothing written past the capacity
    small = np.zeros(len(rows) // 2 + 1, M.api.ROW_DTYPE)
    guard = small.copy()
    with pytest.raises(M.api.MoFREAKError) as e:
        gpu_ctx.extract_clips(clips, kps, chunk_frames=chunk, rows_out=small[:-1])
    assert e.value.code == M.api.ERR_CAPACITY and str(len(rows)) in str(e.value)
    assert small[-1:].tobytes() == guard[-1:].tobytes()
    for buf in pinned:
        gpu_ctx.host_free(buf)


def test_clips_call_edge_cases(gpu_ctx):
    W, H = 64, 48
    kps = synth.dense_grid(W, H, 8, 7.0, 24)
    rows, offs = gpu_ctx.extract_clips([], kps)
    assert len(rows) == 0 and offs.tolist() == [0]
    short = [np.zeros((t, H, W), np.uint8) for t in (5, 0, 3)]  # nothing longer than the gap: no pair at all
    rows, offs = gpu_ctx.extract_clips(short, kps)
    assert len(rows) == 0 and offs.tolist() == [0, 0, 0, 0]
    one = synth.synth_stack(11, W, H)
    rows, offs = gpu_ctx.extract_clips([one], kps)
    assert rows.tobytes() == gpu_ctx.extract_stream_host(one, kps).tobytes() and offs.tolist() == [0, len(rows)]


# ------------------------------------------------------------------ frame preparation
@pytest.mark.parametrize("W,H", [(320, 240), (1920, 1080), (37, 5), (4099, 3), (641, 2)])
def test_bgr_to_gray_matches_oracle(gpu_ctx, oracle, W, H):
    rng = np.random.default_rng(W + H)
    bgr = rng.integers(0, 256, (3, H, W, 3), dtype=np.uint8)
    bgr[0, 0, :8] = [[255, 255, 255], [0, 0, 0], [255, 0, 0], [0, 255, 0], [0, 0, 255], [1, 1, 1], [254, 255, 253], [128, 127, 129]][: min(8, W)] if W >= 8 else bgr[0, 0, :8]
    got = gpu_ctx.bgr_to_gray_host(bgr)
    for f in range(3):
        assert np.array_equal(got[f], oracle.bgr2gray(bgr[f]))
    if W >= 8:
        assert got[0, 0, 0] == 255 and got[0, 0, 1] == 0  # the three weights sum to 16384: white stays white


def test_bgr_stream_to_rows(gpu_ctx, oracle):
    """Colour frames in, .mofreak rows out: cvtColor on the device, then the frame loop (:391-489)."""
    import torch
    W, H, T = 320, 240, 9
    gray = synth.synth_stack(T, W, H)
    rng = np.random.default_rng(3)
    bgr = np.stack([gray + rng.integers(-20, 21, gray.shape), gray, gray + rng.integers(-20, 21, gray.shape)], -1).clip(0, 255).astype(np.uint8)
    d_bgr = torch.from_numpy(bgr).cuda()
    d_gray = torch.empty((T, H, W), dtype=torch.uint8, device="cuda")
    kps = synth.dense_grid(W, H, 16, 7.0, 23)
    d_kps = torch.from_numpy(kps).cuda()
    rows = torch.zeros((T - 5) * len(kps) * 32, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    gpu_ctx.bgr_to_gray(d_bgr, W, H, T, d_gray)
    n = gpu_ctx.extract_stream(d_gray, T, W, H, d_kps, rows)
    want_gray = np.stack([oracle.bgr2gray(b) for b in bgr])
    assert np.array_equal(d_gray.cpu().numpy(), want_gray)
    offs = np.arange(T - 5 + 1, dtype=np.int64) * len(kps)
    want = oracle.Freak().extract_stream(want_gray, np.tile(kps, (T - 5, 1)), offs)
    assert n == len(want) and rows.cpu().numpy()[: n * 32].tobytes() == want.tobytes()
