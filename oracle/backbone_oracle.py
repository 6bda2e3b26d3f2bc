"""CPU ORACLE (test infrastructure) for the ResNet-FPN backbone: a functional torch-CPU restatement of
reference modeling/backbone/resnet.py:137-146 (ResNet.forward), :274-316 (Bottleneck, BaseStem), fpn.py:44-82 (FPN,
LastLevelMaxPool).  Pinned by tests/golden/backbone_golden.npz (recorded from the imported reference by
tests/golden/make_golden_backbone.py)."""
import torch
import torch.nn.functional as F

BLOCKS = {"R-50": (3, 4, 6, 3), "R-101": (3, 4, 23, 3), "R-152": (3, 8, 36, 3)}


def _bn(sd, p, x, eps=1e-5):
    s = [1, -1, 1, 1]
    return (x - sd[p + ".running_mean"].view(s)) / torch.sqrt(sd[p + ".running_var"].view(s) + eps) * sd[p + ".weight"].view(s) + sd[p + ".bias"].view(s)


def bottleneck(sd, p, x, stride):
    """STRIDE_IN_1X1: the stride sits in conv1 (and in the projection shortcut)."""
    idt = x
    o = F.relu(_bn(sd, p + ".bn1", F.conv2d(x, sd[p + ".conv1.weight"], None, stride)))
    o = F.relu(_bn(sd, p + ".bn2", F.conv2d(o, sd[p + ".conv2.weight"], None, 1, 1)))
    o = _bn(sd, p + ".bn3", F.conv2d(o, sd[p + ".conv3.weight"]))
    if (p + ".downsample.0.weight") in sd:
        idt = _bn(sd, p + ".downsample.1", F.conv2d(x, sd[p + ".downsample.0.weight"], None, stride))
    return F.relu(o + idt)


def resnet(sd, x, arch="R-50", prefix="body"):
    o = F.relu(_bn(sd, prefix + ".stem.bn1", F.conv2d(x, sd[prefix + ".stem.conv1.weight"], None, 2, 3)))
    o = F.max_pool2d(o, 3, 2, 0, ceil_mode=True)
    outs = []
    for i, nblk in enumerate(BLOCKS[arch]):
        for b in range(nblk):
            o = bottleneck(sd, f"{prefix}.layer{i + 1}.{b}", o, 2 if (b == 0 and i > 0) else 1)
        outs.append(o)
    return outs


def fpn(sd, feats, prefix="fpn"):
    conv = lambda name, t, pad: F.conv2d(t, sd[f"{prefix}.{name}.weight"], sd[f"{prefix}.{name}.bias"], 1, pad)
    last = conv("fpn_inner4", feats[3], 0)
    results = [last]                                     # top level without its layer block (fpn.py:49-50)
    for i in (3, 2, 1):
        lateral = conv(f"fpn_inner{i}", feats[i - 1], 0)
        top_down = F.interpolate(last, size=lateral.shape[-2:], mode="bilinear", align_corners=False)
        last = conv(f"fpn_layer{i}", lateral + top_down, 1)
        results.insert(0, last)
    results.append(F.max_pool2d(results[-1], 1, 2, 0))
    return tuple(results)


def backbone(sd, x, arch="R-50"):
    return fpn(sd, resnet(sd, x, arch))
