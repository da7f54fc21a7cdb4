#!/usr/bin/env python3
"""Freeze the keypoints of the sequential Python restatement (tests/helpers/brisk_sequential.py) on the inputs of
tests/test_brisk_sequential.py into tests/golden/brisk_sequential.npz.  The inputs are regenerated from seeds; only the
keypoints (x, y, size, response, layer) are stored, for both readings of a float expression (x87: the reference as built; sse)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import test_brisk_sequential as T  # noqa: E402

if __name__ == "__main__":
    out = {f"{model}/{name}": T.sequential_keypoints(img, thr, octaves, model)[0] for model in T.MODELS for name, img, thr, octaves in T.cases()}
    np.savez_compressed(os.path.join(HERE, "brisk_sequential.npz"), **out)
    print({k: len(v) for k, v in out.items()})
