#!/usr/bin/env python3
"""Micro-benchmark of the bag-of-words assignment kernel at the reference's codebook sizes (main.cpp:71,91,110,127,144)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mofreak_amd as M  # noqa: E402


def main(n=7451136):
    ctx = M.Context(0)
    desc = torch.randint(0, 256, (n, 16), dtype=torch.uint8, device="cuda")
    idx = torch.empty(n, dtype=torch.int32, device="cuda")
    s = torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(s):
        ctx.set_stream(s.cuda_stream)
        for K in (600, 1000, 7000, 10100):
            cb = torch.randint(0, 256, (K, 16), dtype=torch.uint8, device="cuda")
            ctx.bow_assign(desc, cb, idx)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                ctx.bow_assign(desc, cb, idx)
            e1.record()
            e1.synchronize()
            ms = e0.elapsed_time(e1) / 3
            print(json.dumps({"kernel": "bow_assign_kernel", "descriptors": n, "codewords": K, "ms": ms,
                              "descriptors_per_s": n / ms * 1e3, "pair_compares_per_s": n * K / ms * 1e3}))
    ctx.set_stream(None)
    ctx.close()


if __name__ == "__main__":
    main()
