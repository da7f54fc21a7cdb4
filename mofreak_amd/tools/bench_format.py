#!/usr/bin/env python3
""".mofreak text throughput: the device formatter (mofreak_format_rows_device) into device memory and into page-locked host
memory, and the host formatter (mofreak_format_rows, one thread) on the same rows.  Rows of the benchmark's kind (grid) and of a
detector's (fractional coordinates).  Prints one JSON line per case."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402


def main():
    import torch

    import mofreak_amd as M
    from mofreak_amd import api
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
    rng = np.random.default_rng(1)
    with M.Context(0) as ctx:
        for kind in ("grid", "detector"):
            rows = np.zeros(n, api.ROW_DTYPE)
            if kind == "grid":
                rows["x"], rows["y"], rows["scale"] = rng.integers(40, 1880, n), rng.integers(40, 1040, n), 12.0
            else:
                rows["x"], rows["y"] = rng.uniform(40, 1880, n).astype(np.float32), rng.uniform(40, 1040, n).astype(np.float32)
                rows["scale"] = rng.uniform(8.4, 72, n).astype(np.float32)
            rows["frame_number"] = rng.integers(4, 90000, n)
            rows["appearance"], rows["motion"] = rng.integers(0, 256, (n, 8)), rng.integers(0, 256, (n, 8))
            d_rows = torch.from_numpy(rows.view(np.uint8).reshape(-1)).cuda()
            out = ctx.host_alloc(n * 96)
            text, total = ctx.format_rows_device(d_rows, n, out=out)  # warm-up, sizes the workspace
            t0 = time.perf_counter()
            reps = 5
            for _ in range(reps):
                text, total = ctx.format_rows_device(d_rows, n, out=out)
            t_pinned = (time.perf_counter() - t0) / reps
            d_text = torch.empty(total + 64, dtype=torch.uint8, device="cuda")
            need = api.C.c_size_t(0)

            def to_device():
                rc = ctx._lib.mofreak_format_rows_device(ctx._h, api.C.c_void_p(d_rows.data_ptr()), n, api.C.c_void_p(d_text.data_ptr()), total + 64, api.C.byref(need), None, 0, None)
                assert rc == 0
            to_device()
            t0 = time.perf_counter()
            for _ in range(reps):
                to_device()
            t_dev = (time.perf_counter() - t0) / reps
            m = min(n, 400_000)
            t0 = time.perf_counter()
            host = M.format_rows(rows[:m])
            t_host = (time.perf_counter() - t0) * n / m
            assert text[:len(host)].tobytes() == host
            print(json.dumps({"rows": n, "kind": kind, "text_bytes": total, "device_to_hbm_rows_per_s": n / t_dev, "device_to_hbm_GBs_text": total / t_dev / 1e9,
                              "device_to_pinned_host_rows_per_s": n / t_pinned, "device_to_pinned_host_GBs_text": total / t_pinned / 1e9,
                              "host_one_thread_rows_per_s": n / t_host}), flush=True)


if __name__ == "__main__":
    main()
