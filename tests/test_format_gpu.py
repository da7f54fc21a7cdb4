"""The device formatter (mofreak_format_rows_device, csrc/format_kernel.hip) against the host formatter and the oracle's:
writeMoFREAKFeaturesToFile (MoFREAKUtilities.cpp:691-719) byte for byte, `ostream << float` (printf "%g") included."""
import numpy as np
import pytest

import mofreak_amd as M
from mofreak_amd import api

pytestmark = pytest.mark.gpu


def _rows(n, rng, kind):
    rows = np.zeros(n, api.ROW_DTYPE)
    if kind == "grid":  # the benchmark's rows: integral coordinates, size 12
        rows["x"] = rng.integers(0, 1920, n)
        rows["y"] = rng.integers(0, 1080, n)
        rows["scale"] = 12.0
    elif kind == "detector":  # BRISK keypoints: fractional coordinates and sizes
        rows["x"] = rng.uniform(0, 1920, n).astype(np.float32)
        rows["y"] = rng.uniform(0, 1080, n).astype(np.float32)
        rows["scale"] = (12.0 * rng.choice([1, 1.5, 2, 3, 4, 6], n) * rng.uniform(0.7, 1.4, n)).astype(np.float32)
    else:  # every magnitude of the fixed-notation range, values at and next to its decade and rounding boundaries
        mant = rng.uniform(1, 10, n)
        expo = rng.integers(-4, 6, n)
        v = (mant * 10.0 ** expo).astype(np.float32)
        v = np.minimum(np.maximum(v, np.float32(1e-4)), np.float32(999999.4))
        rows["x"] = v
        scale10 = 10.0 ** rng.integers(0, 6, n)  # values with few decimals: short texts, trailing zeros to drop
        rows["y"] = np.where(rng.random(n) < 0.5, (np.round(v.astype(np.float64) * scale10) / scale10).astype(np.float32), v)
        rows["y"] = np.minimum(np.maximum(rows["y"], np.float32(1e-4)), np.float32(999999.4))
        rows["scale"] = rng.choice(np.float32([14.4, 123.457, 0.5, 999999.4, 99999.95, 0.000123456, 9.999995, 1.0000005, 100000, 123456, 0.1, 0.25, 1e-4, 7, 12]), n)
    rows["frame_number"] = rng.integers(-3, 90000, n)
    rows["appearance"] = rng.integers(0, 256, (n, 8))
    rows["motion"] = rng.integers(0, 256, (n, 8))
    return rows


@pytest.mark.parametrize("kind,n", [("grid", 70001), ("detector", 50000), ("edge", 60000), ("grid", 1), ("grid", 256), ("detector", 257)])
def test_device_text_equals_host_text(gpu_ctx, oracle, kind, n):
    import torch
    rng = np.random.default_rng(hash(kind) % 1000 + n)
    rows = _rows(n, rng, kind)
    want = M.format_rows(rows)
    if n <= 50000:
        assert want == oracle.format_rows(rows)
    d_rows = torch.from_numpy(rows.view(np.uint8).reshape(-1)).cuda()
    text, total = gpu_ctx.format_rows_device(d_rows, n)
    assert total == len(want)
    assert text[:total].tobytes() == want
    # per-video segments: one call, many files
    starts = np.unique(np.concatenate([[0], rng.integers(0, n + 1, 7), [n]])).astype(np.int64)
    text2, offs = gpu_ctx.format_rows_device(d_rows, n, row_starts=starts)
    assert offs[-1] == len(want)
    for i, s in enumerate(starts):
        assert offs[i] == len(M.format_rows(rows[:s]))
    assert text2[:offs[-1]].tobytes() == want
    # into device memory as well
    d_text = torch.empty(total + 64, dtype=torch.uint8, device="cuda")
    need = api.C.c_size_t(0)
    rc = gpu_ctx._lib.mofreak_format_rows_device(gpu_ctx._h, api.C.c_void_p(d_rows.data_ptr()), n, api.C.c_void_p(d_text.data_ptr() + 3), total + 32,
                                                 api.C.byref(need), None, 0, None)
    assert rc == 0 and need.value == total
    assert d_text[3:3 + total].cpu().numpy().tobytes() == want


def test_known_rows_and_the_rows_left_to_the_host(gpu_ctx):
    import torch
    rows = np.zeros(4, api.ROW_DTYPE)
    rows["x"] = [14.4, 123.457, 0.0, 320]
    rows["y"] = [7.5, 99999.5, 1e-4, 239]
    rows["scale"] = [12, 8.4, 18.3, 40.5]
    rows["frame_number"] = [4, 5, 6, 7]
    rows["appearance"][1] = [0, 9, 10, 99, 100, 199, 200, 255]
    text, total = gpu_ctx.format_rows_device(torch.from_numpy(rows.view(np.uint8).reshape(-1)).cuda(), 4)
    got = text[:total].tobytes()
    assert got == M.format_rows(rows)
    assert got.startswith(b"14.4 7.5 4 12 0 0 0 0 0 0 0 0 0 0 0 0 0 0 0 0 0 0 \n123.457 99999.5 5 8.4 0 0 0 9 10 99 100 199 200 255 ")
    for bad in (1e-5, 1.5e6, -1.0, float("inf"), float("nan"), -0.0):
        r = rows.copy()
        r["x"][2] = bad
        with pytest.raises(M.MoFREAKError) as e:
            gpu_ctx.format_rows_device(torch.from_numpy(r.view(np.uint8).reshape(-1)).cuda(), 4)
        assert e.value.code == api.ERR_UNSUPPORTED
    # empty input
    text, total = gpu_ctx.format_rows_device(0, 0)
    assert total == 0


def test_text_of_extracted_rows_without_a_host_round_trip(gpu_ctx, oracle):
    """rows left in HBM by the extraction -> text on the device: the file a video's rows make, identical to the oracle's."""
    import torch
    from mofreak_amd import synth
    W, H, T = 320, 240, 9
    frames = synth.synth_stack(T, W, H)
    kps = synth.config_grid("C1")
    rows = gpu_ctx.extract_stream_host(frames, kps)
    offs = np.arange(T - 5 + 1, dtype=np.int64) * len(kps)
    want = oracle.format_rows(oracle.Freak().extract_stream(frames, np.tile(kps, (T - 5, 1)), offs))
    d_rows = torch.from_numpy(rows.view(np.uint8).reshape(-1)).cuda()
    text, total = gpu_ctx.format_rows_device(d_rows, len(rows))
    assert text[:total].tobytes() == want
