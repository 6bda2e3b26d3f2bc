"""Pin the CPU oracle (oracle/psmnet_oracle.py) to fixtures produced by the imported reference
(tests/golden/make_golden.py).  CPU only."""
import hashlib
import os

import numpy as np
import pytest
import torch

from oracle import psmnet_oracle as O
from disprcnn_amd.utils import synth
from tests.helpers import GOLDEN, golden_npz, state_for, check_samples

CV_CASES = [(48, -48, (2, 4, 6, 20)), (8, 0, (1, 3, 5, 12)), (8, -8, (1, 3, 5, 12)), (0, -8, (2, 2, 3, 9)),
            (12, -4, (1, 2, 4, 10)), (48, 0, (1, 32, 28, 28)), (24, -24, (1, 32, 28, 28)), (48, -48, (1, 32, 56, 56))]


def _sha(t):
    return hashlib.sha256(t.contiguous().numpy().tobytes()).hexdigest()


@pytest.mark.parametrize("mx,mn,shp", CV_CASES)
def test_cost_volume_bit_exact(mx, mn, shp):
    z = golden_npz("cost_volume.npz")
    fl, fr = synth.synth_features(*shp, tag=f"cv{mx}_{mn}")
    key = f"cv_{mx}_{mn}_{'x'.join(map(str, shp))}"
    c = O.cost_volume(fl, fr, mx, mn)
    assert tuple(c.shape) == tuple(z[key + "_shape"])
    assert _sha(c) == str(z[key + "_sha"])
    c2 = O.cost_volume_numpy(fl.numpy(), fr.numpy(), mx, mn)
    assert np.array_equal(c2, c.numpy())
    if key + "_full" in z.files:
        assert np.array_equal(z[key + "_full"], c.numpy())


@pytest.mark.parametrize("case,mx,mn", [("A", 48, 0), ("At", 48, 0), ("A2", 24, -24)])
def test_config_a_features_to_disparity(case, mx, mn):
    """Oracle vs reference, entering at the feature boundary.  Same torch-CPU primitives in the same
    order => equal to float rounding; tolerance 1e-4 px (the softmax is sharp on the untrained net)."""
    z = golden_npz()
    sd = state_for("At" if case == "At" else "A")
    fl, fr = synth.synth_features(2, 32, 28, 28, tag="caseA")
    cost = O.cost_volume(fl, fr, mx, mn)
    assert _sha(cost) == str(z[f"{case}_cost_sha"])
    inter = O.regressor3d(sd, cost, want_intermediates=True)
    for name in ("out1", "out2", "out3", "pre1", "post1", "post2", "cost1", "cost2", "cost3"):
        check_samples(z, case, name, inter[name], atol=2e-4, rtol=1e-5)
    pred = O.upsample_softargmin(inter["cost3"], mx, mn, 112, 112)
    ref = torch.from_numpy(z[f"{case}_pred"])
    err = (pred - ref).abs()
    assert err.mean().item() < 1e-4 and err.max().item() < 5e-3, (err.mean().item(), err.max().item())
    # the oracle's own separable lerp (used to reason about the HIP kernel) stays within the a7 tolerance
    pred_sep = O.upsample_softargmin(inter["cost3"], mx, mn, 112, 112, separable=True)
    err = (pred_sep - ref).abs()
    assert err.mean().item() < 1e-3 and err.max().item() < 2e-2


def test_config_b_images_to_disparity():
    z = golden_npz()
    sd = state_for("B")
    left, right = synth.synth_images(2, 224, 224, tag="caseB")
    fl = O.feature_extraction(sd, left)
    fr = O.feature_extraction(sd, right)
    check_samples(z, "B", "featL", fl, atol=1e-4, rtol=1e-5)
    check_samples(z, "B", "featR", fr, atol=1e-4, rtol=1e-5)
    pred = O.psmnet_from_features(sd, fl, fr, 48, -48, 224, 224)
    ref = torch.from_numpy(z["B_pred"])
    err = (pred - ref).abs()
    assert err.mean().item() < 1e-3 and err.max().item() < 2e-2, (err.mean().item(), err.max().item())


def test_fp64_noise_floor_config_a():
    """fp32 reference output vs this oracle in fp64: documents the tolerance floor used for the HIP parity tests."""
    z = golden_npz()
    sd = O.to_dtype(state_for("At"), torch.float64)
    fl, fr = synth.synth_features(2, 32, 28, 28, tag="caseA")
    p64 = O.psmnet_from_features(sd, fl.double(), fr.double(), 48, 0, 112, 112)
    err = (p64.float() - torch.from_numpy(z["At_pred"])).abs()
    assert err.mean().item() < 1e-4 and err.max().item() < 2e-3, (err.mean().item(), err.max().item())


def test_train_mode_loss_matches_reference():
    """Train-mode forward (batch-stat BN, 3 heads) + PSMLoss value vs the reference (loss_utils.py:9-32)."""
    z = golden_npz()
    sd = state_for("B")
    left, right = synth.synth_images(2, 224, 224, tag="caseBtrain")
    target = synth.hash_uniform("tgt", (2, 224, 224), -48.0, 48.0)
    mask = (synth.hash_uniform("mask", (2, 224, 224), 0.0, 1.0) > 0.5).to(torch.uint8)
    with torch.no_grad():
        preds = O.psmnet_forward(sd, left, right, 48, -48, training=True)
    for i, p in enumerate(preds):
        ref = torch.from_numpy(z[f"Bt_pred{i + 1}_s4"])
        err = (p[:, ::4, ::4] - ref).abs()
        assert err.mean().item() < 2e-3, (i, err.mean().item(), err.max().item())
    loss = O.psm_loss(preds, target, mask)
    assert abs(loss.item() - float(z["Bt_loss"])) < 1e-3 * float(z["Bt_loss"])


def test_psm_loss_edge_cases():
    t = torch.zeros(1, 4, 4)
    m0 = torch.zeros(1, 4, 4, dtype=torch.uint8)
    assert float(O.psm_loss(t + 1, t, m0)) == 0.0                      # eval, empty mask -> 0
    l = O.psm_loss((t + 0.5, t + 2.0, t + 1.0), t, m0 + 1)             # smooth-l1: .125, 1.5, .5
    assert abs(float(l) - (0.5 * 0.125 + 0.7 * 1.5 + 0.5)) < 1e-6
