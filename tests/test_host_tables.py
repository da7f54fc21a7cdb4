"""Host logic of the product checked against the oracle on CPU: the tables libmofreak_hip.so builds at
context creation (tables-only context, no GPU), and the .mofreak text writer/reader."""
import numpy as np
import pytest

import mofreak_amd as M


@pytest.fixture(scope="module")
def tctx(native_lib):
    ctx = M.Context(M.TABLES_ONLY)
    yield ctx
    ctx.close()


def test_pattern_sizes_match_oracle(tctx, oracle):
    assert tctx.pattern_sizes().tolist() == oracle.Freak().pattern_sizes()


def test_pattern_lut_matches_oracle_bitwise(tctx, oracle):
    f = oracle.Freak()
    rng = np.random.default_rng(0)
    cells = [(0, 0), (12, 0), (63, 255), (12, 77)] + [(int(s), int(r)) for s, r in zip(rng.integers(0, 64, 40), rng.integers(0, 256, 40))]
    for s, r in cells:
        a, b = tctx.table_pattern(s, r), f.pattern(s, r)
        assert a.tobytes() == b.tobytes(), (s, r)


def test_orientation_weights_match_oracle(tctx, oracle):
    assert np.array_equal(tctx.table_orientation(), oracle.Freak().orientation_pairs())


@pytest.mark.parametrize("mode", [M.BITS_SSE, M.BITS_NATURAL, M.BITS_SSE_SIGNED])
def test_bit_pairs_match_oracle_layout(native_lib, oracle, mode):
    """The 64 (i, j) pairs behind descriptor bytes 0..7 in each layout."""
    ctx = M.Context(M.TABLES_ONLY, freak_bit_mode=mode)
    got = ctx.table_bit_pairs()
    pairs = oracle.Freak().description_pairs()
    for B in range(8):
        for b in range(8):
            m = 8 * B + b if mode == M.BITS_NATURAL else 16 * b + (15 - B)
            assert tuple(got[8 * B + b]) == tuple(pairs[m])
    ctx.close()


def test_scale_index_thresholds_match_oracle_chain(tctx, oracle):
    f = oracle.Freak()
    rng = np.random.default_rng(3)
    sizes = np.concatenate([
        np.float32([1.2e-7, 1e-3, 0.5, 6.99, 7, 8.4, 12, 18, 27, 36, 40.5, 107, 1e4, 3e38]),
        rng.uniform(0.1, 500, 20000).astype(np.float32),
        np.exp(rng.uniform(np.log(1e-6), np.log(1e9), 5000)).astype(np.float32)])
    # floats right at every bin edge: 7 * 2^((k - 0.5)/16) and its neighbours
    edges = (7.0 * 2.0 ** ((np.arange(1, 64) - 0.5) / 16.0)).astype(np.float32)
    for d in range(-3, 4):
        e = edges.copy()
        for _ in range(abs(d)):
            e = np.nextafter(e, np.float32(np.inf if d > 0 else 0), dtype=np.float32)
        sizes = np.concatenate([sizes, e])
    for s in sizes:
        assert tctx.scale_index(s) == f.scale_index(s), float(s)


@pytest.mark.parametrize("L", [1, 2, 3, 7, 9, 12, 13, 18, 19, 20, 27, 28, 38, 41, 57, 108, 339, 2048])
def test_resize_taps_match_oracle_tables(tctx, oracle, L):
    taps = tctx.table_resize(L)
    xofs, xcoef, xmax = oracle.resize_axis_table(L, 19, True)
    yofs, ycoef, _ = oracle.resize_axis_table(L, 19, False)
    for d in range(19):
        ofs, ofs1, c0, c1 = taps[0, d]
        if d < xmax:
            assert (ofs, ofs1, c0, c1) == (xofs[d], xofs[d] + 1, xcoef[d][0], xcoef[d][1])
        else:
            assert (ofs, ofs1, c0, c1) == (xofs[d], xofs[d], 2048, 0)
        ofs, ofs1, c0, c1 = taps[1, d]
        clip = lambda v: min(max(v, 0), L - 1)
        assert (ofs, ofs1, c0, c1) == (clip(yofs[d]), clip(yofs[d] + 1), ycoef[d][0], ycoef[d][1])
    assert taps[:, :, :2].min() >= 0 and taps[:, :, :2].max() <= L - 1


def test_unsupported_pattern_scale_is_refused(native_lib):
    # sigma < 0.5 would need FREAK::meanIntensity's bilinear branch: refuse loudly instead of approximating
    with pytest.raises(M.MoFREAKError) as e:
        M.Context(M.TABLES_ONLY, freak_pattern_scale=5.0)
    assert e.value.code == -4
    with pytest.raises(M.MoFREAKError):
        M.Context(M.TABLES_ONLY, freak_bit_mode=7)


def test_tables_only_context_refuses_compute(tctx):
    z = np.zeros((1, 64, 64), np.uint8)
    with pytest.raises(M.MoFREAKError) as e:
        tctx.extract_pairs_host(z, z, np.float32([[32, 32, 7]]))
    assert e.value.code == -5
    with pytest.raises(M.MoFREAKError):
        tctx.synchronize()


# ------------------------------------------------------------------ .mofreak text
def test_format_rows_matches_oracle_and_reference_layout(native_lib, oracle):
    rng = np.random.default_rng(9)
    n = 500
    rows = np.zeros(n, M.ROW_DTYPE)
    rows["x"] = rng.uniform(0, 2000, n).astype(np.float32)
    rows["y"] = rng.uniform(0, 1100, n).astype(np.float32)
    rows["x"][:100] = np.floor(rows["x"][:100])
    rows["y"][:100] = np.floor(rows["y"][:100])
    rows["x"][100:110] = np.float32([0, 0.5, 1e-5, 123456.0, 1234567.0, 99999.0, 100000.0, 3.14159274, 1e7, 0.1])
    rows["frame_number"] = rng.integers(0, 100000, n)
    rows["scale"] = rng.choice(np.float32([12, 8.4, 14.4, 40.5, 7, 18.000002]), n)
    rows["appearance"] = rng.integers(0, 256, (n, 8))
    rows["motion"] = rng.integers(0, 256, (n, 8))
    txt = M.format_rows(rows)
    assert txt == oracle.format_rows(rows)
    first = txt.split(b"\n")[0]
    assert first.endswith(b" ") and len(first.split()) == 6 + 16  # x y frame scale mx my + 8 + 8, trailing space
    assert M.format_rows(rows[:0]) == b""


def test_format_known_rows(native_lib):
    rows = np.zeros(2, M.ROW_DTYPE)
    rows[0] = (48, 48, 4, 12, list(range(10, 18)), list(range(200, 208)))
    rows[1] = (123.4567, 0.5, 17, 14.4, [0] * 8, [255] * 8)
    assert M.format_rows(rows) == (b"48 48 4 12 0 0 10 11 12 13 14 15 16 17 200 201 202 203 204 205 206 207 \n"
                                   b"123.457 0.5 17 14.4 0 0 0 0 0 0 0 0 0 0 255 255 255 255 255 255 255 255 \n")


def test_parse_rows_round_trip(native_lib):
    rng = np.random.default_rng(4)
    n = 300
    rows = np.zeros(n, M.ROW_DTYPE)
    rows["x"] = rng.integers(0, 2000, n)          # integral coordinates survive the 6-digit text exactly
    rows["y"] = rng.integers(0, 1100, n)
    rows["frame_number"] = rng.integers(4, 5000, n)
    rows["scale"] = rng.choice(np.float32([12, 8.5, 14.25, 40.5, 7]), n)
    rows["appearance"] = rng.integers(0, 256, (n, 8))
    rows["motion"] = rng.integers(0, 256, (n, 8))
    back = M.parse_rows(M.format_rows(rows))
    assert back.tobytes() == rows.tobytes()
    assert len(M.parse_rows(b"")) == 0 and len(M.parse_rows(b"  \n")) == 0
    with pytest.raises(M.MoFREAKError):
        M.parse_rows(b"1 2 3 4 0 0 1 2 3\n")  # truncated row


# ------------------------------------------------------------------ the tile kernel's MIP sampling order
MIP_CENTRES = [(5, 5), (5, 9), (5, 13), (9, 5), (9, 13), (13, 5), (13, 9), (13, 13)]
MIP_OFFSETS = [(-4, 0), (-3, 3), (0, 4), (3, 3), (4, 0), (3, -3), (0, -4), (-3, -3)]


def _mip_needed_dwords():
    need = set()
    for x, y in MIP_CENTRES:
        need |= {(y - 1) * 19 + (x - 1) + k for k in range(9)}                                        # current strip
        for dx, dy in MIP_OFFSETS:
            need |= {368 + (y - 1 + dy) * 19 + (x - 1 + dx) + k for k in range(9)}                    # previous strips
    return sorted({p // 4 for p in need})


def _mip_lds_cycles(tctx, L, positions, pitch=464, rw=224):  # kTileStagePitch, kTileRW (tables.h)
    """The model of mofreak_amd/tools/mip_lane_order.py, restated: LDS cycles of a keypoint's 20 sampling reads (five
    passes x two source rows x two taps), one cycle per distinct dword on the busiest of the 32 banks per 32-lane group,
    summed over the four byte alignments of the ROI's first pixel."""
    taps = tctx.table_resize(L).astype(int)
    tx, ty = taps[0], taps[1]
    passes = [positions[64 * u:64 * u + 64] for u in range(4)]
    tail = list(positions[256:])
    passes.append(tail + [tail[-1]] * (64 - len(tail)))
    total = 0
    for al in range(4):
        for pl in passes:
            for row in (0, 1):
                for tap in (0, 1):
                    for g in (0, 1):
                        banks = {}
                        for p in pl[32 * g:32 * g + 32]:
                            fr, i = int(p) // 368, min(int(p) % 368, 360)
                            addr = fr * rw + ty[i // 19][row] * pitch + tx[i % 19][0] + tap + al
                            banks.setdefault((addr // 4) % 32, set()).add(addr // 4)
                        total += max(len(v) for v in banks.values())
    return total


@pytest.mark.parametrize("L", range(1, 17))
def test_mip_sampling_order_covers_the_needed_bytes_and_spreads_over_the_banks(tctx, L):
    pos = tctx.table_mip_positions(L)
    dwords = _mip_needed_dwords()
    assert len(dwords) == 75 and len(pos) == 300
    lane_dwords = pos[:64] // 4
    for u in range(4):  # pass u = byte u of the lane's dword: the lane packs four results into one 32-bit LDS store
        assert np.array_equal(pos[64 * u:64 * u + 64], 4 * lane_dwords + u)
    tail_dwords = pos[256::4] // 4
    assert np.array_equal(pos[256:], (4 * tail_dwords[:, None] + np.arange(4)).reshape(-1))
    assert sorted(lane_dwords.tolist() + tail_dwords.tolist()) == dwords  # every needed dword exactly once
    cycles = _mip_lds_cycles(tctx, L, pos)
    in_table_order = _mip_lds_cycles(tctx, L, np.array([4 * d + u for u in range(4) for d in dwords[:64]] + [4 * d + b for d in dwords[64:] for b in range(4)]))
    assert cycles <= in_table_order and cycles <= (160 if L <= 12 else 232), (L, cycles, in_table_order)  # floor: 4 alignments x 20 reads x 2 groups = 160
    if L == 12:
        assert in_table_order == 224 and cycles == 160
