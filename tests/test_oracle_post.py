"""Pins oracle/post_oracle.py to the maps recorded from the reference's DisparityMapProcessor (tests/golden/make_golden_post.py)."""
import os

import numpy as np
import pytest
import torch

from disprcnn_amd.utils import synth
from oracle import post_oracle as O

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "post_golden.npz"))
CASES = sorted({k.split(":")[0] for k in G.files})
H, W, S = 96, 320, 224


def case_inputs(name):
    lb, rb = torch.from_numpy(G[f"{name}:left"]), torch.from_numpy(G[f"{name}:right"])
    disp = synth.hash_uniform(f"post:{name}", (len(lb), S, S), -48.0, 48.0)
    return lb, rb, disp


@pytest.mark.parametrize("name", CASES)
def test_disparity_map_matches_reference(name):
    lb, rb, disp = case_inputs(name)
    got = O.disparity_map(lb, rb, disp, H, W)
    ref = torch.from_numpy(G[f"{name}:map"])
    assert got.shape == ref.shape
    assert torch.equal(got, ref), f"max diff {(got - ref).abs().max().item()}"


def test_single_roi_keeps_negative_values_and_many_clamp_at_zero():
    """max over the stacked per-ROI maps: with one ROI its negative disparities survive, with two every pixel is >= 0 outside
    the overlap (the other map's zero wins) -- a property of the reference worth a known-answer test."""
    lb, rb, disp = case_inputs("single_negative")
    one = O.disparity_map(lb, rb, disp, H, W)
    assert one.min().item() < 0
    lb2 = torch.cat([lb, torch.tensor([[0.0, 0.0, 4.0, 4.0]])]); rb2 = torch.cat([rb, torch.tensor([[0.0, 0.0, 4.0, 4.0]])])
    two = O.disparity_map(lb2, rb2, torch.cat([disp, torch.zeros(1, S, S)]), H, W)
    assert two.min().item() >= 0


def test_depth_maps_are_reciprocal_of_the_patch():
    lb, rb, disp = case_inputs("right_wider")
    depth = O.roi_depth_maps(lb, rb, disp, H, W, 389.0)
    dm0 = O.disparity_map(lb[:1], rb[:1], disp[:1], H, W)          # one ROI: the map is its patch
    x1, y1, x2, y2 = O.expand_box_to_integer(lb[0].tolist())
    ref = 389.0 / (dm0[y1:y2, x1:x2] + 1e-6)
    assert torch.allclose(depth[0, y1:y2, x1:x2], ref)
    assert depth[0, :y1].abs().max().item() == 0
