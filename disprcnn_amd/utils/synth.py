"""Closed-form synthetic tensors (weights / inputs) for parity tests and benchmarks.

There is no network for checkpoints or datasets, and the reference's Python
never travels to the GPU box, so every test/bench tensor is produced by an
index-based integer hash (splitmix64) that is bit-reproducible on any host:
no libm, no RNG state.  The golden fixtures under ``tests/golden`` were made by
loading exactly these tensors into the reference ``PSMNet``
(reference: disprcnn/modeling/psmnet/stackhourglass.py:55-104).

Initialisation statistics follow the reference init loop
(stackhourglass.py:90-104): Conv2d/Conv3d weights have std sqrt(2/(k*Cout));
ConvTranspose3d keeps the torch default (kaiming-uniform, bound 1/sqrt(fan_in),
fan_in = weight.size(1)*k^3) because it is not an ``nn.Conv3d`` instance.
"""
import math
import zlib

import numpy as np
import torch

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _MASK
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK
    return z ^ (z >> np.uint64(31))


def hash_uniform(key, shape, lo=-1.0, hi=1.0):
    """Deterministic uniform[lo,hi) fp32 tensor addressed by (key, flat index)."""
    n = int(np.prod(shape)) if len(shape) else 1
    seed = np.uint64(zlib.crc32(key.encode()) & 0xFFFFFFFF) << np.uint64(32)
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) + seed
        bits = _splitmix64(idx) >> np.uint64(40)  # top 24 bits
    u = bits.astype(np.float64) / float(1 << 24)
    out = (lo + (hi - lo) * u).astype(np.float32).reshape(shape)
    return torch.from_numpy(out)


def synth_state_dict(template, tag="w", tempered=False):
    """Fill every entry of ``template`` (name -> tensor, e.g. a state_dict) in closed form.

    Conv weights: uniform with the reference's init std; BN gamma in [0.8,1.2],
    beta in [-0.1,0.1]; running stats left at (0,1) unless a calibrated fixture
    is loaded on top (see ``load_bn_stats``).
    """
    out = {}
    for name, t in template.items():
        shape = tuple(t.shape)
        key = f"{tag}:{name}"
        if name.endswith("num_batches_tracked"):
            out[name] = torch.zeros((), dtype=torch.long)
        elif name.endswith("running_mean"):
            out[name] = torch.zeros(shape)
        elif name.endswith("running_var"):
            out[name] = torch.ones(shape)
        elif len(shape) == 1 and name.endswith(".weight"):
            out[name] = hash_uniform(key, shape, 0.8, 1.2)
        elif len(shape) == 1 and name.endswith(".bias"):
            out[name] = hash_uniform(key, shape, -0.1, 0.1)
        elif len(shape) in (4, 5):
            k = int(np.prod(shape[2:]))
            is_deconv = (".conv5.0." in name) or (".conv6.0." in name)
            if is_deconv:  # ConvTranspose3d: weight [Cin, Cout, k,k,k], torch default init
                bound = 1.0 / math.sqrt(shape[1] * k)
            else:          # std sqrt(2/(k*Cout)) -> uniform bound sqrt(3)*std
                bound = math.sqrt(3.0) * math.sqrt(2.0 / (k * shape[0]))
            w = hash_uniform(key, shape, -bound, bound)
            if tempered and name.startswith("classif") and name.endswith(".2.weight"):
                w = w * 0.1
            out[name] = w
        else:
            raise ValueError(f"unhandled state entry {name} {shape}")
    return out


def load_bn_stats(state, npz_path):
    """Overlay calibrated BN running statistics (fixture) onto a synth state dict."""
    z = np.load(npz_path)
    for k in z.files:
        state[k] = torch.from_numpy(z[k].copy())
    return state


def synth_features(n, c, h, w, tag="feat"):
    """Feature-map-like input ~ uniform, unit-ish variance."""
    a = math.sqrt(3.0)
    return hash_uniform(f"{tag}:L", (n, c, h, w), -a, a), hash_uniform(f"{tag}:R", (n, c, h, w), -a, a)


def synth_images(n, h, w, tag="img"):
    """ImageNet-normalised-like crops: (U[0,1) - mean) / std per channel."""
    mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    left = hash_uniform(f"{tag}:L", (n, 3, h, w), 0.0, 1.0)
    # right view = left shifted by a few pixels + noise so matching is non-trivial
    right = torch.roll(left, shifts=-6, dims=3) * 0.9 + 0.1 * hash_uniform(f"{tag}:R", (n, 3, h, w), 0.0, 1.0)
    return (left - mean) / std, (right - mean) / std


def synth_backbone_state(template):
    """Closed-form ResNet-FPN weights: kaiming_uniform(a=1) bounds for convs (reference resnet.py:262-263, make_layers.py:51),
    BN affine near identity, small FPN biases; running statistics at (0,1) until a calibrated fixture is overlaid."""
    out = {}
    for k, v in template.items():
        shape = tuple(v.shape)
        if k.endswith("num_batches_tracked"):
            out[k] = torch.zeros((), dtype=torch.long)
        elif k.endswith("running_mean"):
            out[k] = torch.zeros(shape)
        elif k.endswith("running_var"):
            out[k] = torch.ones(shape)
        elif len(shape) == 4:
            fan_in = shape[1] * shape[2] * shape[3]
            b = (3.0 / fan_in) ** 0.5
            out[k] = hash_uniform("bb:" + k, shape, -b, b)
        elif ".bn" in k or "downsample.1" in k:
            out[k] = hash_uniform("bb:" + k, shape, 0.8, 1.2) if k.endswith("weight") else hash_uniform("bb:" + k, shape, -0.1, 0.1)
        else:                                   # fpn conv bias
            out[k] = hash_uniform("bb:" + k, shape, -0.05, 0.05)
    return out


# per-key multipliers on the fan-in bound: spread the objectness / class scores and keep the box deltas moderate, so that the
# synthetic heads produce proposals and detections with real NMS chains (used by tests/golden/make_golden_det.py and the tests)
DET_GAIN = {"head.cls_logits.weight": 0.5, "head.bbox_pred.weight": 0.6, "box.predictor.cls_score.weight": 3.0,
            "box.predictor.bbox_pred.weight": 2.0, "mask.predictor.mask_fcn_logits.weight": 4.0}


def synth_det_state(template, gain=None):
    """Closed-form weights for the 2D-stage heads (Stereo RPN, stereo box head, mask head): uniform with a fan-in bound so that
    activations stay O(1) through the heads; buffers (cell anchors) are kept.  `gain` overrides the per-key multiplier."""
    gain = gain or {}
    out = {}
    for k, v in template.items():
        shp = tuple(v.shape)
        if "cell_anchors" in k:
            out[k] = v.clone()
        elif k.endswith("bias"):
            out[k] = hash_uniform("det:" + k, shp, -0.1, 0.1)
        else:
            fan_in = int(np.prod(shp[1:])) if len(shp) > 1 else shp[0]
            if k.endswith("conv5_mask.weight"):                # ConvTranspose2d [Cin, Cout, 2, 2]: one tap per output pixel
                fan_in = shp[0]
            a = gain.get(k, 1.0) * math.sqrt(3.0 / fan_in)
            out[k] = hash_uniform("det:" + k, shp, -a, a)
    return out


def synth_pyramid(n, h, w, c=256, tag="pyr"):
    """Left/right FPN-like feature pyramids for an h x w image: 5 levels at strides 4..64 (ceil division like the backbone)."""
    left, right = [], []
    hh, ww = -(-h // 4), -(-w // 4)
    for lvl in range(5):
        a = math.sqrt(3.0)
        left.append(hash_uniform(f"{tag}:L{lvl}", (n, c, hh, ww), -a, a))
        right.append(torch.roll(left[-1], shifts=-max(1, 8 >> lvl), dims=3) * 0.8 + 0.2 * hash_uniform(f"{tag}:R{lvl}", (n, c, hh, ww), -a, a))
        hh, ww = -(-hh // 2), -(-ww // 2)
    return left, right
