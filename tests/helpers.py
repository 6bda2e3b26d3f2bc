"""Shared test helpers: rebuild the synthetic state dicts the golden fixtures were made with."""
import os

import numpy as np
import torch

from disprcnn_amd.utils import synth
from disprcnn_amd.modeling.psmnet.keys import psmnet_state_template

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_npz(name="psmnet_golden.npz"):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def state_for(case):
    """case in {'A','At','B'} -> state dict identical to the one loaded into the reference."""
    sd = synth.synth_state_dict(psmnet_state_template(), tempered=(case == "At"))
    return synth.load_bn_stats(sd, os.path.join(GOLDEN, f"bn_stats_{case}.npz"))


def check_samples(z, tag, name, tensor, atol, rtol=0.0):
    flat = tensor.detach().cpu().reshape(-1).double()
    idx = torch.from_numpy(z[f"{tag}_{name}_idx"])
    ref = torch.from_numpy(z[f"{tag}_{name}_val"]).double()
    got = flat[idx]
    err = (got - ref).abs()
    tol = atol + rtol * ref.abs()
    assert bool((err <= tol).all()), f"{tag}/{name}: max err {err.max().item():.3e} (tol {atol}+{rtol}*|ref|)"
    abssum = float(z[f"{tag}_{name}_abssum"])
    got_abssum = flat.abs().sum().item()
    assert abs(got_abssum - abssum) <= 1e-4 * max(abssum, 1.0) + atol * flat.numel() * 0.05, \
        f"{tag}/{name}: abssum {got_abssum} vs {abssum}"
