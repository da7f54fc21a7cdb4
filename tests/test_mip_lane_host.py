"""The tile kernel's lane-per-keypoint MIP (mofreak_amd/csrc/mip_lane.h) against the oracle, on the CPU.

The header is compiled for the host with the handful of gfx950 instructions it is written in replaced by plain C++
(MOFREAK_MIP_LANE_HOST), so every compile-time tap table, byte selector, strip position and bit position of every ROI side
the kernel is instantiated for is exercised without a GPU: 168 000 keypoints (7 sides x integer and fractional sizes x
3 frame kinds, ROIs at every byte alignment and against all four image borders) against mo_mip_descriptor
(MoFREAKUtilities.cpp:288-325 -> :46-99).  The GPU parity tests then check the same code as the device runs it."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_lane_per_keypoint_mip_matches_oracle_on_the_host(tmp_path, oracle):
    exe = tmp_path / "mip_lane_host"
    cmd = ["g++", "-O1", "-std=c++17", "-ffp-contract=off", "-o", str(exe), os.path.join(ROOT, "tests", "helpers", "mip_lane_host.cpp"),
           "-x", "c", os.path.join(ROOT, "oracle", "mofreak_oracle.c"), "-lm"]
    subprocess.run(cmd, check=True, capture_output=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout[-2000:]
    assert out.stdout.startswith("ok "), out.stdout[-2000:]
    assert int(out.stdout.split()[1]) > 150000


def test_instantiated_sides_match_the_kernel():
    """The host harness runs the sides the kernel dispatches on (kMipLaneMinL .. kMipLaneMaxL)."""
    import re
    kern = open(os.path.join(ROOT, "mofreak_amd", "csrc", "tile_kernel.hip")).read()
    lo, hi = map(int, re.search(r"kMipLaneMinL = (\d+), kMipLaneMaxL = (\d+)", kern).groups())
    cases = sorted(int(m) for m in re.findall(r"run_half\(std::integral_constant<int, (\d+)>\{\}\)", kern))
    assert cases == list(range(lo, hi + 1))
    host = open(os.path.join(ROOT, "tests", "helpers", "mip_lane_host.cpp")).read()
    assert sorted(set(int(m) for m in re.findall(r"run_side<(\d+), kMipMaskA>", host))) == cases
    assert sorted(set(int(m) for m in re.findall(r"run_side<(\d+), kMipMaskB>", host))) == cases
