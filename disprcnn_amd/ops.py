"""Tensor-level operator wrappers over the C ABI (dense reference layouts in, dense out).

These are the functions a maintainer of the reference would call from
``PSMNet.forward`` (stackhourglass.py:115-128,169-173).  GPU only; errors surface as RuntimeError.
"""
import torch

from . import _lib
from . import engine as E


def cost_volume(left, right, maxdisp, mindisp):
    """[N,C,H',W'] x2 -> [N,2C,(maxdisp-mindisp)//4,H',W'] (bit-exact with stackhourglass.py:115-128)."""
    E.require_gpu(left, "cost_volume"); E.require_gpu(right, "cost_volume")
    if left.shape != right.shape or left.dim() != 4:
        raise ValueError("cost_volume expects two [N,C,H,W] tensors of equal shape")
    left, right = left.contiguous(), right.contiguous()
    n, c, h, w = left.shape
    dp = (maxdisp - mindisp) // 4
    cost = torch.empty(n, 2 * c, dp, h, w, dtype=torch.float32, device=left.device)
    st = _lib.lib().drc_cost_volume_fwd(E._ptr(left), E._ptr(right), E._ptr(cost), n, c, dp, h, w, mindisp // 4, maxdisp // 4,
                                        E._stream_ptr(left.device))
    _lib.check(st, "drc_cost_volume_fwd")
    return cost


def cost_volume_backward(gcost, maxdisp, mindisp):
    E.require_gpu(gcost, "cost_volume_backward")
    gcost = gcost.contiguous()
    n, c2, dp, h, w = gcost.shape
    c = c2 // 2
    gl = torch.empty(n, c, h, w, dtype=torch.float32, device=gcost.device)
    gr = torch.empty_like(gl)
    st = _lib.lib().drc_cost_volume_bwd(E._ptr(gcost), E._ptr(gl), E._ptr(gr), n, c, dp, h, w, mindisp // 4, maxdisp // 4,
                                        E._stream_ptr(gcost.device))
    _lib.check(st, "drc_cost_volume_bwd")
    return gl, gr


def upsample_softargmin(cost, maxdisp, mindisp, H, W):
    """[N,1,D',H',W'] or [N,D',H',W'] -> disparity [N,H,W] (stackhourglass.py:169-173 fused)."""
    E.require_gpu(cost, "upsample_softargmin")
    if cost.dim() == 5:
        cost = cost[:, 0]
    cost = cost.contiguous()
    disp = torch.empty(cost.shape[0], H, W, dtype=torch.float32, device=cost.device)
    E.upsample_softargmin(cost, disp, maxdisp, mindisp)
    return disp


def conv3d_bn(x, weight, scale, shift, stride=1, relu=False, residual=None, transposed=False):
    """Dense-layout convenience wrapper around the tap-conv engine (used by the layer-level parity tests):
    x [N,Cin,D,H,W] -> [N,Cout,D',H',W'];  y = act(scale*conv(x)+shift (+residual))."""
    E.require_gpu(x, "conv3d_bn")
    n, cin, d, h, w = x.shape
    cout = weight.shape[1] if transposed else weight.shape[0]
    dev = x.device
    xb = E.Blocked(n, cin, d, h, w, 1, 1, 1, dev).from_dense(x)
    if transposed:
        od, oh, ow = 2 * d, 2 * h, 2 * w
    else:
        od, oh, ow = (-(-d // stride), -(-h // stride), -(-w // stride))
    yb = E.Blocked(n, cout, od, oh, ow, 1, 1, 1, dev)
    rb = E.Blocked(n, cout, od, oh, ow, 1, 1, 1, dev).from_dense(residual) if residual is not None else None
    plan = E.plan_deconv3d(xb, yb, cout, relu) if transposed else E.plan_conv3d(xb, yb, stride, cout, relu)
    wp = E.pack_weight(weight.to(dev).float(), transposed)
    cp = wp.shape[3]
    sc = torch.ones(cp, device=dev); sh = torch.zeros(cp, device=dev)
    sc[:cout] = scale; sh[:cout] = shift
    plan.run(xb, wp, sc, sh, yb, rb, w16=plan.pack16(weight.to(dev).float(), transposed))
    return yb.to_dense()


def conv2d_bn(x, weight, scale, shift, stride=1, pad=1, dilation=1, relu=False, residual=None, in_halo=None):
    """x [N,Cin,H,W] -> [N,Cout,H',W'] through the same engine (D=1)."""
    E.require_gpu(x, "conv2d_bn")
    n, cin, h, w = x.shape
    cout, _, k, _ = weight.shape
    dev = x.device
    halo = in_halo if in_halo is not None else max(pad, 1)
    xb = E.Blocked(n, cin, 1, h, w, 0, halo, halo, dev).from_dense(x)
    oh = (h + 2 * pad - dilation * (k - 1) - 1) // stride + 1
    ow = (w + 2 * pad - dilation * (k - 1) - 1) // stride + 1
    yb = E.Blocked(n, cout, 1, oh, ow, 0, 1, 1, dev)
    rb = E.Blocked(n, cout, 1, oh, ow, 0, 2, 2, dev).from_dense(residual) if residual is not None else None
    plan = E.plan_conv2d(xb, yb, k, stride, pad, dilation, cout, relu)
    wp = E.pack_conv_weight(weight.to(dev).float())
    cp = E.cout_pad_of(cout)
    sc = torch.ones(cp, device=dev); sh = torch.zeros(cp, device=dev)
    sc[:cout] = scale; sh[:cout] = shift
    plan.run(xb, wp, sc, sh, yb, rb, w16=plan.pack16(weight.to(dev).float()))
    return yb.to_dense()[:, :, 0]


def integer_roi_boxes(left_bbox, right_bbox):
    """[R,4] xyxy float boxes (left, right) -> [R,6] int32 (x1, y1, x2, y2, x1p, x2p) after expand_box_to_integer
    (utils/stereo_utils.py:219-229), computed on the device (the reference goes through .tolist())."""
    lb, rb = left_bbox.float(), right_bbox.float()
    return torch.stack([lb[:, 0].floor(), lb[:, 1].floor(), lb[:, 2].ceil(), lb[:, 3].ceil(), rb[:, 0].floor(), rb[:, 2].ceil()],
                       dim=1).to(torch.int32).contiguous()


def disparity_paste(disp, boxes6, rois_per_image, height, width, clamp0=False, masks=None):
    """Per-ROI disparities [R,S,S] + integer boxes [R,6] -> full-image disparity maps [B,height,width]
    (DisparityMapProcessor, modeling/psmnet/inference.py:18-47; clamp0 / masks: roi_disp_postprocess, disprcnn3d.py:161-190)."""
    E.require_gpu(disp, "disparity_paste")
    dev = disp.device
    B = len(rois_per_image)
    R = int(sum(rois_per_image))
    if disp.dim() != 3 or disp.shape[0] != R or disp.shape[1] != disp.shape[2] or tuple(boxes6.shape) != (R, 6):
        raise ValueError("disparity_paste expects disp [R,S,S] and boxes [R,6] with R = sum(rois_per_image)")
    offs = torch.tensor([0] + list(rois_per_image), dtype=torch.int64).cumsum(0).to(torch.int32).to(dev)
    disp = disp.contiguous().float()
    boxes6 = boxes6.to(device=dev, dtype=torch.int32).contiguous()
    if masks is not None:
        masks = masks.to(device=dev, dtype=torch.float32).contiguous()
        if tuple(masks.shape) != (R, height, width):
            raise ValueError("masks must be [R,height,width]")
    out = torch.empty(B, height, width, dtype=torch.float32, device=dev)
    st = _lib.lib().drc_disparity_paste_fwd(E._ptr(disp), disp.shape[1], E._ptr(boxes6), E._ptr(offs), B, height, width, int(bool(clamp0)),
                                            E._ptr(masks), E._ptr(out), E._stream_ptr(dev))
    _lib.check(st, "drc_disparity_paste_fwd")
    return out


def roi_depth_maps(disp, boxes6, fuxb, height, width):
    """Per-ROI depth maps [R,height,width] = fuxb / (disparity + 1e-6) inside each ROI's box, zero elsewhere
    (PointRCNN.process_input, pointnet_module/point_rcnn/lib/net/point_rcnn.py:121-133)."""
    E.require_gpu(disp, "roi_depth_maps")
    dev = disp.device
    R = disp.shape[0]
    if disp.dim() != 3 or disp.shape[1] != disp.shape[2] or tuple(boxes6.shape) != (R, 6):
        raise ValueError("roi_depth_maps expects disp [R,S,S] and boxes [R,6]")
    fuxb = torch.as_tensor(fuxb, dtype=torch.float32, device=dev).expand(R).contiguous()
    disp = disp.contiguous().float()
    boxes6 = boxes6.to(device=dev, dtype=torch.int32).contiguous()
    out = torch.empty(R, height, width, dtype=torch.float32, device=dev)
    st = _lib.lib().drc_roi_depth_maps_fwd(E._ptr(disp), disp.shape[1], E._ptr(boxes6), E._ptr(fuxb), R, height, width, E._ptr(out),
                                           E._stream_ptr(dev))
    _lib.check(st, "drc_roi_depth_maps_fwd")
    return out
