"""Committed golden vectors (tests/golden/, made by make_golden.py from the oracle): the oracle must still
reproduce them on CPU, and the HIP path must reproduce them on the GPU."""
import os

import numpy as np
import pytest

import mofreak_amd as M

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MODES = [("sse", 0), ("natural", 1), ("sse_signed", 2)]


def test_oracle_reproduces_golden_pair(oracle):
    g = np.load(os.path.join(GOLD, "golden_pair.npz"))
    for name, mode in MODES:
        d, v = oracle.Freak(bit_mode=mode).extract_pair(g["cur"], g["prev"], g["kps"])
        assert np.array_equal(v, g[f"valid_{name}"]) and np.array_equal(d, g[f"desc_{name}"])
    # the layouts really differ, and only in the appearance half
    assert not np.array_equal(g["desc_sse"][:, :8], g["desc_natural"][:, :8])
    assert np.array_equal(g["desc_sse"][:, 8:], g["desc_natural"][:, 8:])


def test_oracle_reproduces_golden_stream(oracle):
    g = np.load(os.path.join(GOLD, "golden_stream.npz"))
    rows = oracle.Freak().extract_stream(g["frames"], g["kps"], g["kp_offsets"])
    assert rows.view(np.uint8).reshape(-1, 32).tobytes() == g["rows"].tobytes()
    assert oracle.format_rows(rows) == g["text"].tobytes()


def test_golden_text_round_trips_through_the_product_parser(native_lib):
    g = np.load(os.path.join(GOLD, "golden_stream.npz"))
    rows = g["rows"].copy().view(M.ROW_DTYPE).reshape(-1)
    text = g["text"].tobytes()
    assert M.format_rows(rows) == text                      # the product writer = the golden text
    back = M.parse_rows(text)
    assert len(back) == len(rows)
    assert np.array_equal(back["appearance"], rows["appearance"]) and np.array_equal(back["motion"], rows["motion"])
    assert np.array_equal(back["frame_number"], rows["frame_number"])
    assert np.array_equal(back["x"], rows["x"]) and np.array_equal(back["scale"], rows["scale"])  # quarter-pixel values survive %g


@pytest.mark.gpu
def test_hip_reproduces_golden_pair(native_lib):
    g = np.load(os.path.join(GOLD, "golden_pair.npz"))
    for name, mode in MODES:
        with M.Context(0, freak_bit_mode=mode) as ctx:
            d, v = ctx.extract_pairs_host(g["cur"], g["prev"], g["kps"])
        assert np.array_equal(v, g[f"valid_{name}"]), name
        assert np.array_equal(d, g[f"desc_{name}"]), name


@pytest.mark.gpu
def test_hip_reproduces_golden_stream(gpu_ctx):
    g = np.load(os.path.join(GOLD, "golden_stream.npz"))
    rows = gpu_ctx.extract_stream_host(g["frames"], g["kps"], kp_offsets=g["kp_offsets"])
    assert rows.view(np.uint8).reshape(-1, 32).tobytes() == g["rows"].tobytes()
    assert M.format_rows(rows) == g["text"].tobytes()


@pytest.mark.gpu
def test_device_formatter_reproduces_the_golden_text(gpu_ctx):
    """The committed stream's rows, put into HBM, come back from mofreak_format_rows_device as the committed text."""
    import torch
    g = np.load(os.path.join(GOLD, "golden_stream.npz"))
    rows = np.ascontiguousarray(g["rows"]).view(np.uint8).reshape(-1)
    text, total = gpu_ctx.format_rows_device(torch.from_numpy(rows.copy()).cuda(), len(rows) // 32)
    assert text[:total].tobytes() == g["text"].tobytes()


def _kp(a):
    import oracle_lib
    return a.copy().view(oracle_lib.KEYPOINT_DTYPE).reshape(-1)


def test_oracle_reproduces_golden_detector(oracle):
    g = np.load(os.path.join(GOLD, "golden_detector.npz"))
    clip = g["clip"]
    assert oracle.brisk_detect(oracle.absdiff(clip[5], clip[0])).tobytes() == _kp(g["kp_pair"]).tobytes()
    assert oracle.brisk_detect(g["ties"]).tobytes() == _kp(g["kp_ties"]).tobytes()
    assert len(np.unique(_kp(g["kp_ties"])["layer"])) >= 4


@pytest.mark.gpu
def test_device_reproduces_golden_detector(native_lib):
    g = np.load(os.path.join(GOLD, "golden_detector.npz"))
    clip = g["clip"]
    with M.Context(0) as ctx:
        for img_cur, img_prev, want in [(clip[5], clip[0], _kp(g["kp_pair"])), (g["ties"], None, _kp(g["kp_ties"]))]:
            kps, offs, resp, layer = ctx.detect_pairs_host(img_cur, img_prev)
            assert kps.tobytes() == np.stack([want["x"], want["y"], want["size"]], 1).astype(np.float32).tobytes()
            assert resp.tobytes() == want["response"].tobytes() and np.array_equal(layer, want["layer"])
        rows = ctx.compute_stream_host(clip)
        assert rows.view(np.uint8).reshape(-1, 32).tobytes() == g["clip_rows"].tobytes()
