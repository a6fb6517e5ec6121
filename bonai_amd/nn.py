"""Autograd bindings: torch.autograd.Function wrappers around the C-ABI kernels.

torch.autograd is used as bookkeeping only (which saved tensor feeds which backward launch); every
forward and backward body is a hand-written gfx950 kernel from bonai_amd.kernels.  Activations are
bf16 NHWC, master weights fp32 in the reference's [Cout,Cin,R,S] layout (so reference checkpoints
load unchanged), accumulation fp32.
"""
import torch

from . import kernels as K


def to_nhwc(t):
    return t.contiguous(memory_format=torch.channels_last)


class _ConvFn(torch.autograd.Function):
    """y = act(conv(x, w) + b + residual); w [G?,Cout,Cin,R,S] fp32, x bf16 NHWC.

    Backward: relu mask (loft_relu_bwd_bf16) -> dgrad (loft_conv_tap_bf16 with the transposed packing)
    -> wgrad (loft_conv_wgrad_bf16) -> bias grad (loft_colsum_bf16)."""

    @staticmethod
    def forward(ctx, x, w, b, residual, stride, pad, relu, groups, out_f32):
        grouped = w.dim() == 5
        wg = w if grouped else w[None]
        G, Cout, Cin, R, S = wg.shape
        assert G == groups
        wp = torch.stack([K.pack_w_fwd(wg[i]) for i in range(G)]) if G > 1 else K.pack_w_fwd(wg[0])[None]
        bias = None if b is None else b.float().contiguous()
        y = K.conv2d_fwd(x, wp, bias, R, S, stride, pad, relu=relu, residual=residual,
                         out_dtype=torch.float32 if out_f32 else torch.bfloat16, groups=G)
        ctx.cfg = (stride, pad, relu, G, R, S, grouped, tuple(x.shape[2:]), b is not None, residual is not None)
        ctx.save_for_backward(x, wg, y if relu else None)
        return y

    @staticmethod
    def backward(ctx, g):
        stride, pad, relu, G, R, S, grouped, in_hw, has_b, has_res = ctx.cfg
        x, wg, y = ctx.saved_tensors
        g = to_nhwc(g)
        if g.dtype != torch.bfloat16:
            g = g.to(torch.bfloat16)
        if relu:
            g = K.relu_bwd(g, y)
        Cout, Cin = wg.shape[1], wg.shape[2]
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            wpt = torch.stack([K.pack_w_dgrad(wg[i]) for i in range(G)]) if G > 1 else K.pack_w_dgrad(wg[0])[None]
            gx = K.conv2d_dgrad(g, wpt, in_hw, R, S, stride, pad, groups=G)
        want_b = has_b and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            if want_b:   # bias gradient rides along in the wgrad kernel (ones-operand MFMA)
                dwp, gb = K.conv2d_wgrad(g, x, R, S, stride, pad, groups=G, with_bias=True)
                gb = gb if grouped else gb[0]
                want_b = False
            else:
                dwp = K.conv2d_wgrad(g, x, R, S, stride, pad, groups=G)      # [G, R*S, Cout, Cin] fp32
            gw = dwp.view(G, R, S, Cout, Cin).permute(0, 3, 4, 1, 2)
            gw = gw if grouped else gw[0]
        if want_b:
            if G == 1:
                gb = K.colsum(g, Cout)
            else:
                n = g.shape[0] // G
                gb = torch.stack([K.colsum(g[i * n:(i + 1) * n], Cout) for i in range(G)])
        gres = g if (has_res and ctx.needs_input_grad[3]) else None
        return gx, gw, gb, gres, None, None, None, None, None


def conv2d(x, w, b=None, stride=1, pad=0, relu=False, residual=None, groups=1, out_f32=False):
    return _ConvFn.apply(x, w, b, residual, stride, pad, relu, groups, out_f32)


def linear(x2d, w, b=None, relu=False, out_f32=False):
    """x [N,K] bf16 (row-major), w [O,K] fp32 -> [N,O]."""
    N, Kd = x2d.shape
    y = conv2d(x2d.reshape(N, Kd, 1, 1).contiguous(memory_format=torch.channels_last), w.view(w.shape[0], Kd, 1, 1), b,
               relu=relu, out_f32=out_f32)
    return y.reshape(N, w.shape[0])


class _NarrowHeadFn(torch.autograd.Function):
    """1x1 conv / linear with a handful of outputs (RPN cls+reg, fc_cls+fc_reg, fc_offset, mask logits),
    fp32 output [.., Cout4] (Cout rounded up to a multiple of 4).  The backward zero-pads the output
    gradient to 128 channels so the same MFMA dgrad / wgrad kernels apply."""
    PADW = 128

    @staticmethod
    def forward(ctx, x, w, b):
        Cout, Cin = w.shape[0], w.shape[1]
        c4 = (Cout + 3) // 4 * 4
        wpad = torch.zeros(c4, Cin, 1, 1, dtype=w.dtype, device=w.device)
        wpad[:Cout] = w.view(Cout, Cin, 1, 1)
        bpad = torch.zeros(c4, dtype=torch.float32, device=w.device)
        if b is not None:
            bpad[:Cout] = b
        K.ALGO_SCALE = Cout / c4
        y = K.conv2d_fwd(x, K.pack_w_fwd(wpad)[None], bpad, 1, 1, out_dtype=torch.float32)
        K.ALGO_SCALE = 1.0
        ctx.save_for_backward(x, w)
        ctx.has_b = b is not None
        return y

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        Cout, Cin = w.shape[0], w.shape[1]
        P = _NarrowHeadFn.PADW
        N, c4, H, W = g.shape
        gp = torch.zeros(N, P, H, W, dtype=torch.bfloat16, device=g.device).contiguous(memory_format=torch.channels_last)
        gp[:, :Cout] = g[:, :Cout]
        gx = gw = gb = None
        K.ALGO_SCALE = Cout / P
        if ctx.needs_input_grad[0]:
            wt = torch.zeros(1, Cin, P, dtype=torch.bfloat16, device=w.device)
            wt[0, :, :Cout] = w.view(Cout, Cin).t()
            gx = K.conv2d_dgrad(gp, wt[None], (H, W), 1, 1)
        if ctx.needs_input_grad[1]:
            dwp = K.conv2d_wgrad(gp, x, 1, 1)
            gw = dwp[0, 0, :Cout].reshape(w.shape)
        K.ALGO_SCALE = 1.0
        if ctx.has_b and ctx.needs_input_grad[2]:
            gb = g[:, :Cout].float().sum(dim=(0, 2, 3))
        return gx, gw, gb


def narrow_head(x, w, b=None):
    """x bf16 NHWC [N,Cin,H,W]; w [Cout,Cin(,1,1)] with small Cout -> fp32 [N,ceil4(Cout),H,W]."""
    return _NarrowHeadFn.apply(x, w, b)


class _DeconvFn(torch.autograd.Function):
    """ConvTranspose2d(k=2, s=2) + bias + ReLU (mmdet/models/roi_heads/mask_heads/fcn_mask_head.py:121-124)."""

    @staticmethod
    def forward(ctx, x, w, b):
        N, Cin, H, W = x.shape
        Cout = w.shape[1]
        wp = w.permute(2, 3, 1, 0).reshape(4, Cout, Cin).to(torch.bfloat16).contiguous()
        y = K.empty_nhwc(N, Cout, 2 * H, 2 * W, torch.bfloat16, x.device)
        bias = b.float().contiguous()
        for py in range(2):
            for px in range(2):
                K.conv_tap(x, wp, y, N, H, W, Cin, Cout, H, W, 2 * H, 2 * W, [(0, 0, py * 2 + px)], ss=1, os=2,
                           oo=(py, px), bias=bias, relu=True)
        ctx.save_for_backward(x, w, y)
        return y

    @staticmethod
    def backward(ctx, g):
        x, w, y = ctx.saved_tensors
        N, Cin, H, W = x.shape
        Cout = w.shape[1]
        g = K.relu_bwd(to_nhwc(g), y)
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            wt = w.permute(2, 3, 0, 1).reshape(4, Cin, Cout).to(torch.bfloat16).contiguous()
            gx = K.empty_nhwc(N, Cin, H, W, torch.bfloat16, x.device)
            taps = [(py, px, py * 2 + px) for py in range(2) for px in range(2)]
            K.conv_tap(g, wt, gx, N, 2 * H, 2 * W, Cout, Cin, H, W, H, W, taps, ss=2)
        if ctx.needs_input_grad[1]:
            taps = [(py, px, 0, 0, py * 2 + px) for py in range(2) for px in range(2)]
            db = torch.zeros(1, Cout, dtype=torch.float32, device=g.device) if ctx.needs_input_grad[2] else None
            dwp = K.conv_wgrad(g, x, N, 2 * H, 2 * W, Cout, H, W, Cin, H, W, taps, 4, gos=2, ss=1, db=db, db_tap=-2)
            gw = dwp[0].view(2, 2, Cout, Cin).permute(3, 2, 0, 1)
            gb = db[0] if db is not None else None
        elif ctx.needs_input_grad[2]:
            gb = K.colsum(g, Cout)
        return gx, gw, gb


def deconv2x2_relu(x, w, b):
    return _DeconvFn.apply(x, w, b)


class _RoIAlignFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rois, P, strides, finest_scale, n_rot, *feats):
        ctx.save_for_backward(rois)
        ctx.meta = (P, tuple(strides), finest_scale, n_rot, [tuple(f.shape) for f in feats], feats[0].dtype)
        return K.roi_align_fwd(list(feats), rois, P, strides, finest_scale, n_rot)

    @staticmethod
    def backward(ctx, g):
        (rois,) = ctx.saved_tensors
        P, strides, fs, n_rot, shapes, dt = ctx.meta
        grads = K.roi_align_bwd(to_nhwc(g), rois, shapes, P, strides, fs, n_rot, rois_sorted=True)  # bbox2roi order
        return (None, None, None, None, None) + tuple(K.cast_bf16(x) if dt == torch.bfloat16 else x for x in grads)


def roi_align(feats, rois, P, strides, finest_scale=56, n_rot=1):
    return _RoIAlignFn.apply(rois, P, tuple(strides), finest_scale, n_rot, *feats)


class _FpnTopDownFn(torch.autograd.Function):
    """laterals[i-1] += nearest_x2(laterals[i]), coarsest to finest, in place (necks/fpn.py:176-181)."""

    @staticmethod
    def forward(ctx, *lats):
        lats = list(lats)
        for i in range(len(lats) - 1, 0, -1):
            K.upsample2x_add_(lats[i - 1], lats[i])
        ctx.mark_dirty(*lats[:-1])
        return tuple(lats)

    @staticmethod
    def backward(ctx, *gs):
        gs = [to_nhwc(g).clone() for g in gs]
        for i in range(1, len(gs)):
            K.downsum2x_add_(gs[i], gs[i - 1])
        return tuple(gs)


def fpn_top_down(lats):
    return _FpnTopDownFn.apply(*lats)


class _Subsample2Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.shape = tuple(x.shape)
        return K.subsample2(x)

    @staticmethod
    def backward(ctx, g):
        n, c, h, w = ctx.shape
        big = K.zeros_nhwc(n, c, h, w, torch.bfloat16, g.device)
        return K.subsample2_adjoint_add_(big, to_nhwc(g))


def subsample2(x):
    return _Subsample2Fn.apply(x)
