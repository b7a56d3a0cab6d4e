"""Shared helpers for the parity tests."""
import os
from types import SimpleNamespace

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return {k: z[k] for k in z.files}


def make_opt(network="r2plus1d_18", fixconvs=False, nopretrained=False):
    return SimpleNamespace(network=network, fixconvs=fixconvs, nopretrained=nopretrained)


def rel_err(a, ref):
    a = np.asarray(a, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    return float(np.abs(a - ref).max() / (np.abs(ref).max() + 1e-30))


def rel_l2(a, ref):
    a = np.asarray(a, dtype=np.float64).ravel()
    ref = np.asarray(ref, dtype=np.float64).ravel()
    return float(np.linalg.norm(a - ref) / (np.linalg.norm(ref) + 1e-30))


def sample_idx(numel, k=64):
    return np.unique(np.linspace(0, numel - 1, num=min(k, numel)).astype(np.int64))


def case_inputs(g):
    """Regenerate the synthetic inputs of a golden case from its metadata."""
    from zeroshotvideoclassification_amd import synthetic as S
    n, frames, size = int(g["meta_n"]), int(g["meta_frames"]), int(g["meta_size"])
    x = S.synthetic_clips(n, frames, size)
    _, z = S.synthetic_targets(n)
    return x, z
