#!/usr/bin/env python3
"""A/B timing of detector builds (kernel experiments only; the figures that count are bench.py's, on the in-tree library).

  ab_detect.py --build NAME [-DX ...]   compile mofreak_amd/_exp/libdet_NAME.so from the tree's sources with extra flags
  ab_detect.py --ablate NAME            an ablation build from a patched COPY of detect_kernel.hip (ABLATIONS below); the
                                        product source has no switch for this; results are wrong by construction, only the
                                        time is of interest
  ab_detect.py --run [PAIRS] LIB ...    on the GPU box: every library in a process of its own under
                                        rocprofv3 --kernel-trace --stats (detector_probe.py PAIRS 6), one line per det_* kernel
                                        (microseconds per call) and the probe's own line
"""
from __future__ import annotations

import csv
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
EXP = os.path.join(ROOT, "mofreak_amd", "_exp")


def sub1(text, old, new):
    assert text.count(old) == 1, old
    return text.replace(old, new)


ABLATIONS = {
    # refinement: the cells of a window taken from the image byte under them instead of scored (17 extractions + 80 operations less per cell)
    "noscore": lambda t: sub1(sub1(t, "? patch_ring_score(q, iy + 3, ix + 3) : 0;", "? patch_byte(q, iy + 3, ix + 3) : 0;"),
                              "? patch_ring_score(q, 5 + dy, 5 + dx) : 0;", "? patch_byte(q, 5 + dy, 5 + dx) : 0;"),
    # refinement: no image patch loads (registers filled from the coordinates)
    "noload": lambda t: sub1(t, "__builtin_memcpy(&v, img + L.off + (int64_t)yc * L.w + xs, 16);", "v = make_uint4(xs, yc, r, 0);"),
}


def _tie_stamps(t):
    """det_tie_kernel with wall-clock stamps (100 MHz) around its phases; workgroup 0 prints them (device printf)."""
    t = sub1(t, "    // ---- prologue\n", "    // ---- prologue\n    __shared__ long long st_fs[8], st_ch[8]; __shared__ int st_pass[8], st_nw[8];\n"
                "    if (threadIdx.x < 8) { st_fs[threadIdx.x] = 0; st_ch[threadIdx.x] = 0; st_pass[threadIdx.x] = 0; st_nw[threadIdx.x] = 0; }\n    const long long t_start = wall_clock64();\n")
    t = sub1(t, "    // ---- layer after layer\n", "    const long long t_pro = wall_clock64();\n    // ---- layer after layer\n")
    t = sub1(t, "        const DetLayer L = geom_s.L[layer];\n        // (1) first sight", "        const DetLayer L = geom_s.L[layer];\n        const long long t_l0 = wall_clock64();\n        // (1) first sight")
    t = sub1(t, "        __syncthreads();  // the waiting list is complete; what first sight published is visible\n",
             "        __syncthreads();  // the waiting list is complete; what first sight published is visible\n        const long long t_l1 = wall_clock64();\n")
    t = sub1(t, "                if (!waits) break;\n                if (pass > n_items) {", "                if (!waits) { atomicMax(&st_pass[layer], pass + 1); break; }\n                if (pass > n_items) {")
    t = sub1(t, "        __syncthreads();  // the layer is through (its threads' last decisions included); the waiting list is free again\n",
             "        __syncthreads();  // the layer is through (its threads' last decisions included); the waiting list is free again\n"
             "        if (threadIdx.x == 0) { st_fs[layer] = t_l1 - t_l0; st_ch[layer] = wall_clock64() - t_l1; st_nw[layer] = n_wait; }\n")
    t = sub1(t, "// ------------------------------------------------------------------ ordered emission",
             "// ------------------------------------------------------------------ ordered emission (stamps build)")
    # the kernel's closing brace: the last '}' before the emission banner
    i = t.index("// ------------------------------------------------------------------ ordered emission")
    j = t.rindex("}", 0, i)
    tail = ("    __syncthreads();\n    if (blockIdx.x == 0 && threadIdx.x == 0) {\n        const long long t_end = wall_clock64();\n"
            "        printf(\"tie stamps (us): total %.1f prologue %.1f ties %d |\", (t_end - t_start) / 100.0, (t_pro - t_start) / 100.0, n_ties);\n"
            "        for (int l = 0; l < n_layers; ++l) printf(\" L%d n=%d first %.1f chain %.1f passes %d waiting %d |\", l, tie_lo_s[l + 1] - tie_lo_s[l], st_fs[l] / 100.0, st_ch[l] / 100.0, st_pass[l], st_nw[l]);\n"
            "        printf(\"\\n\");\n    }\n")
    return t[:j] + tail + t[j:]


ABLATIONS["tiestamps"] = _tie_stamps


def build(name, flags, patch=None):
    sys.path.insert(0, ROOT)
    from mofreak_amd import build as B
    os.makedirs(EXP, exist_ok=True)
    out = os.path.join(EXP, f"libdet_{name}.so")
    srcs = [os.path.join(B.CSRC, s) for s in B.SOURCES]
    if patch:
        d = os.path.join(EXP, f"src_{name}")
        os.makedirs(d, exist_ok=True)
        i = [os.path.basename(s) for s in srcs].index("detect_kernel.hip")
        with open(srcs[i]) as f:
            text = patch(f.read())
        srcs[i] = os.path.join(d, "detect_kernel.hip")
        with open(srcs[i], "w") as f:
            f.write(text)
        flags = [*flags, "-I", B.CSRC]
    subprocess.check_call([B.hipcc(), *B.FLAGS, *flags, "-o", out, *srcs])
    print(out)


def run(libs, pairs, describe=""):
    os.chdir(ROOT)
    env0 = dict(os.environ, TMPDIR="/tmp")
    for lib in libs:
        name = os.path.basename(lib)
        out = os.path.join(ROOT, "gpurun_out", "ab_detect", name)
        subprocess.call(["rm", "-rf", out])
        os.makedirs(out, exist_ok=True)
        env = dict(env0, MOFREAK_HIP_LIBRARY=os.path.abspath(lib))
        log = subprocess.run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", out, "--", "python3",
                              "mofreak_amd/tools/detector_probe.py", str(pairs), "6", *([describe] if describe else [])], env=env, capture_output=True,
                             text=True, timeout=300)
        probe = [ln for ln in log.stdout.splitlines() if ln.startswith("pairs=")]
        stats = glob.glob(os.path.join(out, "**", "*kernel_stats.csv"), recursive=True)
        per = {}
        if stats:
            for r in csv.DictReader(open(stats[0])):
                m = re.search(r"(det_\w+|describe_kernel|tile_kernel|band_\w+|bin_\w+|compact_\w+)", r["Name"])
                if m:
                    per[m.group(1)] = per.get(m.group(1), 0.0) + float(r["TotalDurationNs"]) / 7e3  # 1 warm-up + 6 calls; microseconds per call
        total = sum(per.values())
        print(f"{name}: {' '.join(probe) or log.stderr[-300:]}")
        print("   " + "  ".join(f"{k.replace('det_', '').replace('_kernel', '')} {v:.0f}" for k, v in sorted(per.items(), key=lambda kv: -kv[1])) + f"  | sum {total:.0f} us/call", flush=True)
        subprocess.call(["rm", "-rf", out])


if __name__ == "__main__":
    a = sys.argv[1:]
    if a[:1] == ["--build"]:
        build(a[1], a[2:])
    elif a[:1] == ["--ablate"]:
        build(a[1], [], ABLATIONS[a[1]])
    elif a[:1] == ["--run"]:
        pairs = 32
        if len(a) > 1 and a[1].isdigit():
            pairs = int(a[1])
            a = a[1:]
        describe = a[1] if len(a) > 1 and a[1] in ("describe", "loop") else ""  # the descriptors behind the detector / the whole frame loop
        run(a[2 if describe else 1:], pairs, describe)
    else:
        raise SystemExit(__doc__)
