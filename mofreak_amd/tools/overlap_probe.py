#!/usr/bin/env python3
"""Does detector work overlap with descriptor work?  The detector's score kernel is bound by vector issue, the gather
path of the descriptors by memory latency.  N host threads, each with a context (stream, workspace) of its own, run
detect + describe on their own batch of full-HD pairs; the aggregate rate against one thread's says what a
software-pipelined frame loop could gain.  usage: overlap_probe.py PAIRS THREADS [STEPS]"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

import mofreak_amd as M
from mofreak_amd import synth

pairs, threads = int(sys.argv[1]), int(sys.argv[2])
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
mode = sys.argv[4] if len(sys.argv) > 4 else "both"   # both | detect | describe
W, H = 1920, 1080
fr = synth.moving_objects_stack(9, W, H)
cur = torch.from_numpy(np.stack([fr[5 + (p % 4)] for p in range(pairs)])).cuda()
prev = torch.from_numpy(np.stack([fr[p % 4] for p in range(pairs)])).cuda()
cap = 32768 * pairs


class Lane:
    def __init__(self):
        self.ctx = M.Context(0)
        self.kps = torch.empty((cap, 3), dtype=torch.float32, device="cuda")
        self.offs = torch.empty(pairs + 1, dtype=torch.int64, device="cuda")
        self.desc = torch.empty((cap, 16), dtype=torch.uint8, device="cuda")
        self.valid = torch.empty(cap, dtype=torch.uint8, device="cuda")

    def step(self):
        if mode != "describe" or not hasattr(self, "n"):
            self.n = self.ctx.detect_pairs(cur, prev, W, H, pairs, self.kps, self.offs, capacity=cap)
        if mode != "detect":
            self.ctx.extract_pairs(cur, prev, W, H, pairs, self.kps, self.desc, self.valid, kp_offsets=self.offs, n_kp=self.n)
            if mode == "describe":
                self.ctx.synchronize()
        return self.n

    def run(self, k):
        for _ in range(k):
            self.step()
        self.ctx.synchronize()


lanes = [Lane() for _ in range(threads)]
for l in lanes:
    l.run(2)
t0 = time.perf_counter()
ts = [threading.Thread(target=l.run, args=(steps,)) for l in lanes]
for t in ts:
    t.start()
for t in ts:
    t.join()
dt = time.perf_counter() - t0
print(f"pairs={pairs} threads={threads} steps={steps} mode={mode}: {pairs * threads * steps / dt:.0f} pairs/s", flush=True)
