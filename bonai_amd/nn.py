"""Autograd bindings: torch.autograd.Function wrappers around the C-ABI kernels.

torch.autograd is used as bookkeeping only (which saved tensor feeds which backward launch); every
forward and backward body is a hand-written gfx950 kernel from bonai_amd.kernels.  Activations are
bf16 NHWC, master weights fp32 in the reference's [Cout,Cin,R,S] layout (so reference checkpoints
load unchanged), accumulation fp32.

fp32 parity mode: when the activation handed to conv2d / narrow_head / deconv2x2_relu is fp32, the same layers run
forward-only on the fp32 MFMA kernel (loft_conv_tap_f32) with fp32 operand packings -- used to compare inference results
with the fp32 reference at 1e-3; the backward of these functions refuses fp32 activations.
"""
import os as _os
import weakref as _weakref

import contextlib
import torch

from . import kernels as K
from .debug import DBG


def to_nhwc(t):
    return t.contiguous(memory_format=torch.channels_last)


# Set by bonai_amd.engine.Trainer: callable(param) invoked after a kernel has ACCUMULATED a parameter's gradient straight
# into ``param.grad`` (a slot of the flat gradient arena).  The autograd Function then returns None for that parameter, so
# no per-parameter accumulation launch (267 tiny adds per step) happens; the callback stands in for the
# post-accumulate-grad hook that drives the bucketed all-reduce.  None (default): gradients are returned to autograd.
GRAD_SINK = None


def _direct_slot(p):
    """The arena slot of a leaf parameter, if the trainer bound one."""
    if GRAD_SINK is None or not isinstance(p, torch.nn.Parameter) or p.grad is None:
        return None
    g = p.grad
    return g if (g.dtype == torch.float32 and g.is_contiguous() and g.shape == p.shape) else None


def _mark_sunk(*ps):
    """These parameters' gradients of this step reach the arena through the kernels (now or at the next queue flush), not
    through autograd: the trainer's reducer must not treat autograd's own post-accumulate callback as 'gradient ready'."""
    for p in ps:
        if p is not None:
            p._loft_sunk = True


def _queue_param_grads(jobs):
    """jobs: [(weight Parameter, dW fp32 [Cout, Cin(P)] (a view is fine), bias Parameter | None, db fp32 [Cout] | None, flat_chw)].
    When the trainer's unpack queue is open and every parameter has an arena slot, the gradients are deposited there by the next
    batched unpack launch (no per-parameter accumulation launch of autograd) and True is returned: the caller hands autograd None.
    flat_chw = (C, H, W): the weight is [O, C, H, W] and dW is [O, (H W) C] (tap-major K order)."""
    if UNPACK_Q is None or DBG.no_leaf_sink:
        return False
    slots = []
    for pw, dw, pb, db, flat in jobs:
        sw = _direct_slot(pw)
        sb = _direct_slot(pb) if pb is not None else None
        if sw is None or (pb is not None and (sb is None or db is None)):
            return False
        slots.append((sw, sb))
    for (pw, dw, pb, db, flat), (sw, sb) in zip(jobs, slots):
        sinks = [pw] + ([pb] if pb is not None else [])
        _mark_sunk(*sinks)
        UNPACK_Q.add(dw, db if pb is not None else None, pw, None, 1e-5, (sw, None, sb), [(lambda q=q: _sink_done(q)) for q in sinks],
                     flat_chw=flat, params=sinks)
    return True


def _count_uses(*ps):
    """One more use of these leaf parameters in this step's graph (the sink fires after the last use has deposited)."""
    for p in ps:
        if isinstance(p, torch.nn.Parameter) and p.requires_grad:
            p._loft_pending = getattr(p, '_loft_pending', 0) + 1


def _sink_done(p):
    """One use of ``p`` has deposited its gradient in the arena; tell the trainer once the last use of this step has."""
    p._loft_pending = getattr(p, '_loft_pending', 1) - 1
    if p._loft_pending <= 0:
        p._loft_pending = 0
        GRAD_SINK(p)


# ---- ReLU-backward masks folded into a consumer's data-gradient epilogue ("pre-masked" gradients) -------------------------
# A consumer whose input x is a ReLU output can apply that ReLU's backward mask in its own dgrad epilogue and tag the gradient
# it returns; the producer of x then skips its relu_bwd pass.  That is only valid when the tagged gradient is the WHOLE gradient
# of x: with a second consumer autograd sums the contributions (in place into whichever arrives first), and an un-masked
# contribution would ride along under the tag.  So every Function that consumes activations registers its use of them in
# forward (_note_use), producers start the count of their output (_begin_uses), and a tag is honoured only when exactly ONE use
# was registered (_premasked).  The mask is idempotent, so the fallback -- the producer masks the accumulated gradient itself --
# is always correct.  Counts are keyed by storage address: a producer's output is alive (saved for backward) until the
# producer's own backward has consulted its count; the trainer clears the table every step.
_USES = {}


def _note_use(*tensors):
    if len(_USES) > 65536:          # forward-only loops without a trainer: stale addresses only make the rule more conservative
        _USES.clear()
    for t in tensors:
        if isinstance(t, torch.Tensor):
            k = t.data_ptr()
            _USES[k] = _USES.get(k, 0) + 1


def _begin_uses(y):
    _USES[y.data_ptr()] = 0
    return y


def _premasked(g, y):
    """Has the (accumulated) gradient g of the ReLU output y already been masked by y's ONLY consumer?"""
    return getattr(g, '_loft_premasked', None) == y.data_ptr() and _USES.get(y.data_ptr(), 2) == 1


WGRAD_STREAM = None   # the running Trainer's second stream for backbone weight-gradient launches (None = same stream)
UNPACK_Q = None    # the running Trainer's kernels.UnpackQueue: weight-gradient unpacking of many convs in one launch
PREPACK = None     # the running Trainer's kernels.PrepackRegistry: all trainable convs' packings in one launch per step
# Pair fusion (loft_bneck_pair_bf16): the data gradient of block k's second conv output, computed by block k+1's backward launch
# together with ITS data gradient; keyed by the address of the gradient tensor block k+1 returns (which the entry keeps alive).
PAIR_G = {}


# ---- packed operands of FROZEN (requires_grad=False) convs are reused from step to step.  The cache lives ON the weight
# Parameter object (attribute ``_loft_packs``), so it dies with the model that owns it, and every entry carries the identity
# (weak reference), version counter, address and geometry of each tensor it was computed from.  Round 2 kept a process-global
# dict keyed by (data_ptr, _version, shape): a model built after another one had been freed received the freed model's
# addresses from the caching allocator and silently ran on the PREVIOUS model's stem / layer1 weights
# (tests/test_lifetime_gpu.py).

def _base_of(t):
    return t._base if t._base is not None else t


def _pack_sig(tensors):
    sig = []
    for t in tensors:
        if t is None:
            sig.append(None)
        else:
            b = _base_of(t)
            sig.append((_weakref.ref(b), b._version, t.data_ptr(), tuple(t.shape), tuple(t.stride())))
    return sig


def _pack_sig_valid(sig, tensors):
    if len(sig) != len(tensors):
        return False
    for s, t in zip(sig, tensors):
        if s is None or t is None:
            if s is not None or t is not None:
                return False
            continue
        b = _base_of(t)
        if s[0]() is not b or s[1] != b._version or s[2] != t.data_ptr() or s[3] != tuple(t.shape) or s[4] != tuple(t.stride()):
            return False
    return True


# A hipGraph that was captured while these packings were in use has their device ADDRESSES baked in (bonai_amd/graphs.py): while
# PACK_TRACE is a list (set by FeatureGraphs.capture), every packing handed out is recorded there with the tensors it was
# computed from and their signature; the graphs keep the packings alive and re-validate the signatures before every replay.
PACK_TRACE = None


class _PackCache(dict):
    """The per-parameter cache.  Its entries hold weak references (not picklable): pickling / deep-copying a Parameter that has
    run a forward -- torch.save(model), a model handed to mp.spawn -- must not fail on them, so the cache
    reduces to an EMPTY cache; it is a cache."""

    def __reduce__(self):
        return (_PackCache, ())

    def __deepcopy__(self, memo):
        return _PackCache()


def _pack_cache_get(sub, tensors):
    """Cached packing for the frozen tensors ``tensors`` (tensors[0] = the first weight = the owner) under sub-key ``sub``."""
    d = _base_of(tensors[0]).__dict__.get('_loft_packs')
    e = d.get(sub) if d else None
    if e is not None and _pack_sig_valid(e[0], tensors):
        if PACK_TRACE is not None:
            PACK_TRACE.append((e[0], list(tensors), e[1]))
        return e[1]
    return None


def _pack_cache_put(sub, tensors, value):
    owner = _base_of(tensors[0])
    d = owner.__dict__.get('_loft_packs')
    if d is None:
        d = owner._loft_packs = _PackCache()
    if len(d) >= 8 and sub not in d:     # (dtype, padding, dgrad) variants of ONE conv: bounded; stale variants go first
        d.clear()                        # (a live graph keeps its own references to the packings it replays: see PACK_TRACE)
    sig = _pack_sig(tensors)
    d[sub] = (sig, value)
    if PACK_TRACE is not None:
        PACK_TRACE.append((sig, list(tensors), value))


def clear_pack_cache(module):
    """Drop every cached packing of ``module``'s parameters (frees their device memory; nothing needs this for correctness:
    a captured graph holds its own references and notices at its next replay that the cache no longer vouches for them)."""
    for p in module.parameters():
        p.__dict__.pop('_loft_packs', None)


class _ConvFn(torch.autograd.Function):
    """y = act(conv(x, fold(w, bn)) + bias + residual), bf16 NHWC activations, fp32 master weights [Cout,Cin,R,S].

    forward : loft_fold_pack (BN fold + both operand packings, one launch) -> loft_conv_tap_bf16
    backward: loft_relu_bwd_bf16 -> loft_conv_tap_bf16 (dgrad) -> loft_conv_wgrad_bf16 (+ fused bias gradient)
              -> loft_fold_unpack_bwd (dW in the reference layout, dgamma, dbeta, one launch)
    Tensor arguments after the meta tuple: for each of the G groups (w, b|None), then (gamma, beta) when BN is folded."""

    @staticmethod
    def forward(ctx, x, residual, meta, *tensors):
        _note_use(x, residual)
        stride, pad, relu, G, out_f32, has_b, bn_stats, frozen, input_relu, cout_pad = meta
        ws = tensors[0:2 * G:2]
        bs = tensors[1:2 * G:2]
        gamma, beta = (tensors[2 * G], tensors[2 * G + 1]) if bn_stats is not None else (None, None)
        Cout, Cin, R, S = ws[0].shape
        T = R * S
        if x.shape[1] != Cin or cout_pad:           # channel-padded activations: pad the packings with zeros to match
            Cin = x.shape[1]
            Cout = cout_pad or Cout
        dev = x.device
        pdt = torch.float32 if x.dtype == torch.float32 else K.L.act16()
        key = cached = None
        if frozen:
            key = ('conv', pdt, Cin, Cout, bool(ctx.needs_input_grad[0]))
            ctens = tuple(tensors) + (tuple(bn_stats[:2]) if bn_stats is not None else ())
            cached = _pack_cache_get(key, ctens)
        if cached is not None:
            wp, wpt, bias = cached
        elif PREPACK is not None and key is None and pdt == K.L.act16() and Cout % 2 == 0 and Cin % 2 == 0 and \
                all(isinstance(w, torch.nn.Parameter) for w in ws):
            bn = (gamma, beta, bn_stats[0], bn_stats[1]) if bn_stats is not None else None
            wp, wpt, bias = PREPACK.request(ws, bs, bn, bn_stats[2] if bn_stats is not None else 1e-5, Cout, Cin,
                                            ctx.needs_input_grad[0])
        elif PREPACK is not None and key is None and pdt == torch.float32 and all(isinstance(w, torch.nn.Parameter) and w.dim() == 4
                                                                                   for w in ws):
            bn = (gamma, beta, bn_stats[0], bn_stats[1]) if bn_stats is not None else None
            wp, wpt, bias = PREPACK.request_f32(ws, bs, bn, bn_stats[2] if bn_stats is not None else 1e-5, Cout, Cin,
                                                ctx.needs_input_grad[0])
        else:
            wp = torch.empty(G, T, Cout, Cin, dtype=pdt, device=dev)
            need_dgrad = ctx.needs_input_grad[0]
            wpt = torch.empty(G, T, Cin, Cout, dtype=pdt, device=dev) if need_dgrad else None
            bias = torch.empty(G, Cout, dtype=torch.float32, device=dev)
            bn = (gamma, beta, bn_stats[0], bn_stats[1]) if bn_stats is not None else None
            eps = bn_stats[2] if bn_stats is not None else 1e-5
            for g in range(G):
                K.fold_pack(ws[g], bs[g], bn, eps, out_fwd=wp[g], out_dgrad=None if wpt is None else wpt[g], out_bias=bias[g],
                            want_dgrad=need_dgrad, dtype=pdt, cout_pad=Cout, cin_pad=Cin)
            if key is not None:
                _pack_cache_put(key, ctens, (wp, wpt, bias))
        use_bias = has_b or bn_stats is not None
        global _HEAD_REQ, _HEAD_OUT
        head, _HEAD_REQ = _HEAD_REQ, None           # (conv2d_with_head: a narrow 1x1 head on this conv's output, same launch)
        y = K.conv2d_fwd(x, wp, bias if use_bias else None, R, S, stride, pad, relu=relu, residual=residual,
                         out_dtype=torch.float32 if (out_f32 or pdt == torch.float32) else K.L.act16(), groups=G,
                         planes_cache=any(ctx.needs_input_grad[3:]), head=head)
        if head is not None:
            y, _HEAD_OUT = y
        ctx.meta = meta
        ctx.params = tensors            # the Parameter objects themselves (their .grad may be an arena slot)
        for k, t in enumerate(tensors):  # uses per step of each parameter: the gradient sink fires after the last one
            if t is not None and ctx.needs_input_grad[3 + k] and isinstance(t, torch.nn.Parameter):
                t._loft_pending = getattr(t, '_loft_pending', 0) + 1
        ctx.in_hw = tuple(x.shape[2:])
        ctx.has_res = residual is not None
        ctx.save_for_backward(x, y if relu else None, wpt, *tensors)
        return _begin_uses(y) if relu else y

    @staticmethod
    @K.planes_scoped
    def backward(ctx, g):
        stride, pad, relu, G, out_f32, has_b, bn_stats, frozen, input_relu, cout_pad = ctx.meta
        x, y, wpt = ctx.saved_tensors[:3]
        tensors = ctx.saved_tensors[3:]
        ws = tensors[0:2 * G:2]
        Cout, Cin, R, S = ws[0].shape
        g = to_nhwc(g)
        if g.dtype != x.dtype:          # bf16 training path; fp32 parity mode keeps everything fp32 (parity_f32.hip kernels)
            g = g.to(x.dtype)
        if relu and not _premasked(g, y):
            g = K.relu_bwd(g, y)        # (skipped when the consumer's dgrad epilogue already applied this mask)
        gx = None
        if ctx.needs_input_grad[0]:
            # input_relu: x is a ReLU output, so d/d(pre-activation) = gx * (x > 0): folded into the dgrad epilogue
            gx = K.conv2d_dgrad(g, wpt, ctx.in_hw, R, S, stride, pad, groups=G, mask=x if input_relu else None, out_dtype=x.dtype)
            if input_relu and stride == 1:
                gx._loft_premasked = x.data_ptr()
        join_deposit = False
        if gx is not None and JOIN is not None and G == 1 and x.data_ptr() in JOIN and JOIN[x.data_ptr()] is None \
                and gx.dtype == K.L.act16() and gx.shape == x.shape:
            JOIN[x.data_ptr()] = gx          # (the residual block that also consumes x adds it in its own data-gradient epilogue)
            join_deposit = True
            # autograd now sees ONE contribution for x -- the block's, which masks the sum with x's ReLU in that epilogue and tags
            # it: count this consumer out so that the producer honours the tag and skips its own relu_bwd pass over the map
            if _USES.get(x.data_ptr(), 0) > 1:
                _USES[x.data_ptr()] -= 1
        ngrads = [None] * len(tensors)
        need_w = any(ctx.needs_input_grad[3 + 2 * i] for i in range(G))
        need_b = (has_b and any(ctx.needs_input_grad[3 + 2 * i + 1] for i in range(G))) or \
            (bn_stats is not None and (ctx.needs_input_grad[3 + 2 * G] or ctx.needs_input_grad[3 + 2 * G + 1]))
        if need_w or need_b:
            # every group's gradient goes to the batched unpack (which can sum split-K slots while it reads) when all of its
            # parameters have arena slots; else the atomically combined form every other consumer expects
            queued = UNPACK_Q is not None and need_w and all(
                _direct_slot(ctx.params[2 * i]) is not None for i in range(G)) and (
                bn_stats is None or (_direct_slot(ctx.params[2 * G]) is not None and _direct_slot(ctx.params[2 * G + 1]) is not None))
            if need_b:
                dwp, db = K.conv2d_wgrad(g, x, R, S, stride, pad, groups=G, with_bias=True, slots_ok=queued)
            else:
                dwp, db = K.conv2d_wgrad(g, x, R, S, stride, pad, groups=G, slots_ok=queued), None
            nsp = _nsplit(dwp)
            bn = None
            if bn_stats is not None:
                bn = (tensors[2 * G], tensors[2 * G + 1], bn_stats[0], bn_stats[1])
            for i in range(G):
                pw = ctx.params[2 * i]
                slot_w = _direct_slot(pw) if need_w else None
                slot_g = slot_b = None
                if bn is not None and slot_w is not None:
                    slot_g, slot_b = _direct_slot(ctx.params[2 * G]), _direct_slot(ctx.params[2 * G + 1])
                if slot_w is not None and (bn is None or (slot_g is not None and slot_b is not None)):
                    # accumulate straight into the flat gradient arena; autograd gets None for these inputs
                    eps = bn_stats[2] if bn_stats is not None else 1e-5
                    if UNPACK_Q is not None:
                        sinks = [pw]
                        pb = ctx.params[2 * i + 1] if (bn is None and has_b and db is not None) else None
                        if pb is not None and ctx.needs_input_grad[3 + 2 * i + 1]:
                            slot_b = _direct_slot(pb)           # the conv's own bias gradient: db rides in the same launch
                            if slot_b is not None:
                                sinks.append(pb)
                        if bn is not None and i == G - 1:
                            sinks += [ctx.params[2 * G], ctx.params[2 * G + 1]]
                        _mark_sunk(*sinks)
                        UNPACK_Q.add(dwp[i], None if db is None else db[i], ws[i], bn, eps, (slot_w, slot_g, slot_b),
                                     [(lambda q=q: _sink_done(q)) for q in sinks], nsplit=nsp, params=sinks)
                        if has_b and db is not None and bn is None and slot_b is None:
                            ngrads[2 * i + 1] = db[i][:ws[i].shape[0]]
                        continue
                    _mark_sunk(pw, *((ctx.params[2 * G], ctx.params[2 * G + 1]) if bn is not None else ()))
                    K.fold_unpack_bwd(dwp[i], None if db is None else db[i], ws[i], bn, eps, out=(slot_w, slot_g, slot_b))
                    _sink_done(pw)
                    if bn is not None and i == G - 1:
                        _sink_done(ctx.params[2 * G])
                        _sink_done(ctx.params[2 * G + 1])
                    if has_b and db is not None:
                        ngrads[2 * i + 1] = db[i][:ws[i].shape[0]]
                    continue
                dw, dg, dbeta = K.fold_unpack_bwd(dwp[i], None if db is None else db[i], ws[i], bn,
                                                  bn_stats[2] if bn_stats is not None else 1e-5, need_dw=need_w)
                ngrads[2 * i] = dw
                if has_b and db is not None:
                    ngrads[2 * i + 1] = db[i][:ws[i].shape[0]]
                if bn is not None:
                    ngrads[2 * G], ngrads[2 * G + 1] = dg, dbeta
        gres = g if (ctx.has_res and ctx.needs_input_grad[1]) else None
        return (None if join_deposit else gx, gres, None) + tuple(ngrads)


def _cacheable(t):
    if t is None:
        return True
    base = t._base if t._base is not None else t
    return isinstance(base, torch.nn.Parameter) and not base.requires_grad and not t.requires_grad


def conv2d(x, w, b=None, stride=1, pad=0, relu=False, residual=None, groups=1, out_f32=False, bn=None, input_relu=False,
           cout_pad=None):
    """w: [Cout,Cin,R,S] parameter, or a list of `groups` such parameters (independent branches, one launch);
    b likewise (or None); bn: a FrozenStatBN-like module folded into the conv (its shift becomes the bias).
    x may carry more channels than Cin and cout_pad may exceed Cout: the packed weights are zero-padded to match, so the
    extra input channels are ignored and the extra output channels are exactly zero."""
    ws = list(w) if isinstance(w, (list, tuple)) else [w]
    bs = list(b) if isinstance(b, (list, tuple)) else [b] * len(ws)
    assert len(ws) == groups
    tensors = []
    for wi, bi in zip(ws, bs):
        tensors += [wi, bi]
    bn_stats = None
    if bn is not None:
        tensors += [bn.weight, bn.bias]
        bn_stats = (bn.running_mean, bn.running_var, bn.eps)
    # cache packed operands only for parameters that can never change (requires_grad=False: frozen stem/layer1);
    # the fused SGD kernel updates trainable parameters through raw pointers without bumping tensor versions.
    # Temporaries (e.g. a permuted weight) are never cached: their storage address can be recycled.
    frozen = all(_cacheable(t) for t in tensors)
    meta = (stride, pad, relu, groups, out_f32, b is not None, bn_stats, frozen, input_relu, cout_pad)
    return _ConvFn.apply(x, residual, meta, *tensors)




_HEAD_REQ = _HEAD_OUT = None


def conv2d_with_head(x, w, b, head_pre, stride=1, pad=0, relu=False):
    """conv2d(x, w, b) and, from the SAME launch where the library serves it (256-cout stream tiles), the narrow 1x1 head
    `head_pre` = narrow_head_prepack(w_head, b_head, x.dtype) applied to its output: -> (y, o fp32 NHWC [B,c4,OH,OW] | None).
    o carries NO autograd history (it is computed inside the conv's launch): for no-grad callers -- the RPN's dense forward in the
    sparse-backward training path and at inference; None = not served, the caller runs narrow_head itself."""
    global _HEAD_REQ, _HEAD_OUT
    assert not torch.is_grad_enabled()
    _HEAD_REQ, _HEAD_OUT = head_pre, None
    try:
        y = conv2d(x, w, b, stride=stride, pad=pad, relu=relu)
    finally:
        _HEAD_REQ = None
    o, _HEAD_OUT = _HEAD_OUT, None
    return y, o


class _LinearFn(torch.autograd.Function):
    """y = act(x W^T + b) for an nn.Linear whose weight is passed AS THE PARAMETER ([O, K], never a view), so that under the
    trainer its bf16 packings come from the step's batched launch and its gradient goes straight into the arena.
    x: [N, K] row-major, or -- with flat_chw = (C, H, W) -- the NHWC map [N, C, H, W] that the reference flattens in (c, h, w)
    order before the layer (convfc_bbox_head.py:159-166, offset_head_expand_feature.py:150-156): the weight's columns are
    permuted to (h, w, c) inside the packing kernel instead of permuting activations or copying the weight."""

    @staticmethod
    def forward(ctx, x, w, b, relu, input_relu, flat_chw):
        _note_use(x)
        O = w.shape[0]
        N = x.shape[0]
        Kd = w.shape[1]
        x4 = x.permute(0, 2, 3, 1).reshape(N, Kd, 1, 1) if flat_chw is not None else x.reshape(N, Kd, 1, 1)
        x4 = x4.contiguous(memory_format=torch.channels_last)            # (a view: NHWC memory is already (h, w, c)-flat)
        need_dgrad = ctx.needs_input_grad[0]
        if PREPACK is not None and O % 2 == 0 and Kd % 2 == 0 and isinstance(w, torch.nn.Parameter):
            wp, wpt, bias = PREPACK.request((w,), (b,), None, 1e-5, O, Kd, need_dgrad, flat_chw=flat_chw)
        else:
            weff = w if flat_chw is None else w.view(O, *flat_chw).permute(0, 2, 3, 1).reshape(O, Kd)
            wp, wpt, bias = K.fold_pack(weff.reshape(O, Kd, 1, 1), b, None, 1e-5, want_dgrad=need_dgrad)
            wp, wpt, bias = wp[None], None if wpt is None else wpt[None], bias[None]
        y = K.conv2d_fwd(x4, wp, bias if b is not None else None, 1, 1, 1, 0, relu=relu, out_dtype=K.L.act16(), planes_cache=True)
        ctx.cfg = (relu, input_relu, flat_chw, tuple(x.shape))
        ctx.params = (w, b)
        for k, t in enumerate((w, b)):
            if t is not None and ctx.needs_input_grad[1 + k] and isinstance(t, torch.nn.Parameter):
                t._loft_pending = getattr(t, '_loft_pending', 0) + 1
        ctx.save_for_backward(x4, y if relu else None, wpt, w)
        if relu:
            _begin_uses(y)
        return y.reshape(N, O)

    @staticmethod
    @K.planes_scoped
    def backward(ctx, g):
        relu, input_relu, flat_chw, xshape = ctx.cfg
        x4, y, wpt, w = ctx.saved_tensors
        pw, pb = ctx.params
        N, O = g.shape
        Kd = w.shape[1]
        g4 = g.reshape(N, O, 1, 1).contiguous(memory_format=torch.channels_last)
        if g4.dtype != K.L.act16():
            g4 = g4.to(K.L.act16())
        if relu and not _premasked(g, y):
            g4 = K.relu_bwd(g4, y)
        gx = None
        if ctx.needs_input_grad[0]:
            gx = K.conv2d_dgrad(g4, wpt, (1, 1), 1, 1, 1, 0, mask=x4 if input_relu else None)
            if flat_chw is not None:
                C, H, W = flat_chw
                gx = gx.reshape(N, H, W, C).permute(0, 3, 1, 2)          # NHWC memory, NCHW-shaped: what the producer expects
            else:
                gx = gx.reshape(xshape)
            if input_relu:
                gx._loft_premasked = x4.data_ptr()
        gw = gb = None
        need_w, need_b = ctx.needs_input_grad[1], pb is not None and ctx.needs_input_grad[2]
        if need_w or need_b:
            slot_w = _direct_slot(pw) if need_w else None
            slot_b = _direct_slot(pb) if need_b else None
            queued = UNPACK_Q is not None and slot_w is not None and (not need_b or slot_b is not None)
            dwp, db = K.conv2d_wgrad(g4, x4, 1, 1, 1, 0, with_bias=True, slots_ok=queued)
            if queued:
                sinks = [pw] + ([pb] if need_b else [])
                _mark_sunk(*sinks)
                UNPACK_Q.add(dwp[0, :, 0] if dwp.dim() == 5 else dwp[0, 0], db[0], w, None, 1e-5, (slot_w, None, slot_b),
                             [(lambda q=q: _sink_done(q)) for q in sinks], flat_chw=flat_chw, nsplit=_nsplit(dwp), params=sinks)
            else:
                gw = dwp[0, 0, :O, :Kd]
                if flat_chw is not None:
                    C, H, W = flat_chw
                    gw = gw.reshape(O, H * W, C).permute(0, 2, 1).reshape(O, Kd)
                gb = db[0, :O] if need_b else None
        return gx, gw, gb, None, None, None


def linear(x2d, w, b=None, relu=False, out_f32=False, input_relu=False):
    """x [N,K] bf16 (row-major), w [O,K] fp32 -> [N,O]."""
    N, Kd = x2d.shape
    if isinstance(w, torch.nn.Parameter) and w.dim() == 2 and not out_f32 and x2d.dtype == K.L.act16() and not DBG.no_linear_fn:
        return _LinearFn.apply(x2d, w, b, relu, input_relu, None)
    y = conv2d(x2d.reshape(N, Kd, 1, 1).contiguous(memory_format=torch.channels_last), w.view(w.shape[0], Kd, 1, 1), b,
               relu=relu, out_f32=out_f32, input_relu=input_relu)
    return y.reshape(N, w.shape[0])


def linear_after_flatten(x, w, b=None, relu=True, input_relu=False):
    """``x.flatten(1)`` of an NCHW-shaped map followed by nn.Linear, on NHWC memory (x bf16 [N,C,H,W] channels_last)."""
    N, C, H, W = x.shape
    if isinstance(w, torch.nn.Parameter) and x.dtype == K.L.act16() and not DBG.no_linear_fn:
        return _LinearFn.apply(x, w, b, relu, input_relu, (C, H, W))
    wperm = w.view(-1, C, H, W).permute(0, 2, 3, 1).reshape(w.shape[0], -1)
    return linear(x.permute(0, 2, 3, 1).reshape(N, -1), wperm, b, relu=relu, input_relu=input_relu)


class _NarrowHeadFn(torch.autograd.Function):
    """Conv / linear with a handful of outputs (RPN cls+reg, fc_cls+fc_reg, fc_offset, mask logits, DCNv2 conv_offset),
    fp32 output [.., Cout4] (Cout rounded up to a multiple of 4).  The backward zero-pads the output
    gradient to 128 channels so the same MFMA dgrad / wgrad kernels apply."""
    PADW = 128

    @staticmethod
    def forward(ctx, x, w, b, stride, pad, input_relu=False, prepacked=None, leaves=None, precomputed=None):
        _note_use(x)
        Cout, Cin, R, S = w.shape
        ctx.input_relu = input_relu
        # leaves: [(weight Parameter, bias Parameter | None, first row, end row)] -- the leaf parameters `w` / `b` are made of (w itself,
        # or the pieces of a concatenation): with them the one-pass backward deposits dW / db straight in their arena slots
        ctx.leaves = leaves if (leaves and R == 1 and S == 1 and ctx.needs_input_grad[1] and
                                all(isinstance(l[0], torch.nn.Parameter) for l in leaves)) else None
        if ctx.leaves is not None:
            _count_uses(*[p for l in ctx.leaves for p in l[:2] if p is not None])
        c4 = (Cout + 3) // 4 * 4
        if precomputed is not None:                 # (the producer of x computed this head in its own epilogue: _DeconvFn)
            y = precomputed
        else:
            wp, bpad = prepacked if prepacked is not None else narrow_head_prepack(w, b, x.dtype)
            K.ALGO_SCALE = Cout / c4
            y = K.conv2d_fwd(x, wp, bpad, R, S, stride, pad, out_dtype=torch.float32)
            K.ALGO_SCALE = 1.0
        ctx.save_for_backward(x, w)
        ctx.has_b = b is not None
        ctx.sp = (stride, pad)
        return y

    @staticmethod
    @K.planes_scoped
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        stride, pad = ctx.sp
        Cout, Cin, R, S = w.shape
        if R == 1 and S == 1 and stride == 1 and pad == 0 and Cout <= 8 and Cin % 4 == 0 and Cin <= 1024 and \
                x.dtype in (K.L.act16(), torch.float32) and not DBG.narrow_mfma_bwd:
            # one pass over x: gx (with the producer's ReLU mask when x is a ReLU output), dW and db together
            want_b = ctx.has_b and ctx.needs_input_grad[2]
            gx, dw, db = K.narrow_head_bwd(g, x, w, relu_in=ctx.input_relu, need_gx=ctx.needs_input_grad[0],
                                           need_dw=ctx.needs_input_grad[1], need_db=want_b)
            if gx is not None and ctx.input_relu:
                gx._loft_premasked = x.data_ptr()
            if ctx.leaves is not None and dw is not None and (db is not None or not ctx.has_b) and _queue_param_grads(
                    [(pw, dw[lo:hi], pb, None if (pb is None or db is None) else db[lo:hi], None) for pw, pb, lo, hi in ctx.leaves]):
                return gx, None, None, None, None, None, None, None, None
            if ctx.leaves is not None:      # (no queue / no slots: autograd accumulates; the uses counted in forward are not sunk)
                for l in ctx.leaves:
                    for p_ in l[:2]:
                        if p_ is not None and getattr(p_, '_loft_pending', 0) > 0:
                            p_._loft_pending -= 1
            return gx, (dw.view(w.shape) if dw is not None else None), db, None, None, None, None, None, None
        P = _NarrowHeadFn.PADW
        N, c4, H, W = g.shape
        gp = torch.zeros(N, P, H, W, dtype=x.dtype, device=g.device).contiguous(memory_format=torch.channels_last)
        gp[:, :Cout] = g[:, :Cout]
        gx = gw = gb = None
        K.ALGO_SCALE = Cout / P
        if ctx.needs_input_grad[0]:
            wt = torch.zeros(R * S, Cin, P, dtype=x.dtype, device=w.device)
            wt[:, :, :Cout] = w.permute(2, 3, 1, 0).reshape(R * S, Cin, Cout)
            gx = K.conv2d_dgrad(gp, wt[None], tuple(x.shape[2:]), R, S, stride, pad, out_dtype=x.dtype)
        want_b = ctx.has_b and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            if want_b:   # bias gradient from the same pass (ones-operand MFMA)
                dwp, db = K.conv2d_wgrad(gp, x, R, S, stride, pad, with_bias=True)
                gb = db[0, :Cout]
                want_b = False
            else:
                dwp = K.conv2d_wgrad(gp, x, R, S, stride, pad)
            gw = dwp[0, :, :Cout].permute(1, 2, 0).reshape(w.shape)
        K.ALGO_SCALE = 1.0
        if want_b:
            gb = g[:, :Cout].float().sum(dim=(0, 2, 3))
        if ctx.leaves is not None:
            for l in ctx.leaves:
                for p_ in l[:2]:
                    if p_ is not None and getattr(p_, '_loft_pending', 0) > 0:
                        p_._loft_pending -= 1
        return gx, gw, gb, None, None, None, None, None, None


def narrow_head_prepack(w, b, dtype):
    """(packed weight [1,T,c4,Cin], bias [c4]) of a narrow head: Cout zero-padded to a multiple of 4 by the packing kernel itself
    (one launch; callers that apply the same head to several maps -- the RPN's five levels -- pack once)."""
    if w.dim() == 2:
        w = w.view(w.shape[0], w.shape[1], 1, 1)
    c4 = (w.shape[0] + 3) // 4 * 4
    pdt = torch.float32 if dtype == torch.float32 else K.L.act16()
    with torch.no_grad():
        wp, _, bias = K.fold_pack(w, b, None, 1e-5, want_dgrad=False, dtype=pdt, cout_pad=c4)
    return wp[None], bias


def narrow_head(x, w, b=None, stride=1, pad=0, input_relu=False, prepacked=None, leaves=None, precomputed=None):
    """x bf16 NHWC [N,Cin,H,W]; w [Cout,Cin(,R,S)] with small Cout -> fp32 [N,ceil4(Cout),OH,OW].
    input_relu: x is the output of a ReLU -- the backward folds that ReLU's mask into the data gradient it produces.
    leaves: when w / b are concatenations, [(weight Parameter, bias Parameter | None, first row, end row)] of their pieces (a
    Parameter passed as w is its own leaf): lets the backward deposit the gradients in the trainer's arena directly.
    precomputed: this head's output as the producer of x already computed it (deconv2x2_relu(head=...)): no forward launch here,
    the autograd node and its backward are unchanged."""
    if leaves is None and isinstance(w, torch.nn.Parameter) and (b is None or isinstance(b, torch.nn.Parameter)):
        leaves = [(w, b, 0, int(w.shape[0]))]
    if w.dim() == 2:
        w = w.view(w.shape[0], w.shape[1], 1, 1)
    return _NarrowHeadFn.apply(x, w, b, stride, pad, input_relu, prepacked, leaves, precomputed)


class _MdcnSampleFn(torch.autograd.Function):
    """DCNv2 sampling: (x, raw conv_offset output) -> modulated, bilinearly sampled columns [B, K*C, OH, OW]
    (loft_mdcn_sample_fwd / _bwd).  The contraction with the weight is an ordinary 1x1 conv2d over K*C channels."""

    @staticmethod
    def forward(ctx, x, om, meta):
        _note_use(x, om)
        kh, kw, stride, pad, dil, dg = meta
        ctx.meta = meta
        ctx.save_for_backward(x, om)
        return K.mdcn_sample_fwd(x, om, kh, kw, stride, pad, dil, dg)

    @staticmethod
    def backward(ctx, g):
        x, om = ctx.saved_tensors
        kh, kw, stride, pad, dil, dg = ctx.meta
        g = to_nhwc(g)
        if g.dtype != x.dtype:
            g = g.to(x.dtype)
        dx, dom = K.mdcn_sample_bwd(x, om, g, kh, kw, stride, pad, dil, dg)
        return (K.cast_bf16(dx) if x.dtype == K.L.act16() else dx), dom, None


def mdcn_sample(x, om, kh, kw, stride=1, pad=0, dil=1, deform_groups=1):
    return _MdcnSampleFn.apply(x, om, (kh, kw, stride, pad, dil, deform_groups))


def modulated_deform_conv2d(x, w, b, w_off, b_off, stride=1, pad=0, dil=1, deform_groups=1, bn=None, relu=False,
                            residual=None, input_relu=False):
    """mmcv ModulatedDeformConv2dPack.forward [mmcv==1.0.5] (used at resnet.py:171-194, fpn.py:116-132):
    conv_offset (a plain conv, fp32 output) -> sampling -> 1x1 contraction with w viewed as [Cout, K*Cin] in (tap, c) order
    (+ the folded frozen BN / bias / residual / ReLU epilogue of conv2d)."""
    if dil != 1:
        raise NotImplementedError('conv_offset runs on the dilation-1 tap kernel')
    Cout, Cin, kh, kw = w.shape
    om = narrow_head(x, w_off, b_off, stride=stride, pad=pad)
    col = mdcn_sample(x, om, kh, kw, stride, pad, dil, deform_groups)
    w2 = w.permute(0, 2, 3, 1).reshape(Cout, kh * kw * Cin, 1, 1)
    return conv2d(col, w2, b, relu=relu, residual=residual, bn=bn)


class _DeconvFn(torch.autograd.Function):
    """ConvTranspose2d(k=2, s=2) + bias + ReLU (mmdet/models/roi_heads/mask_heads/fcn_mask_head.py:121-124)."""

    @staticmethod
    def forward(ctx, x, w, b, input_relu=False, head=None):
        _note_use(x)
        N, Cin, H, W = x.shape
        Cout = w.shape[1]
        ctx.input_relu = input_relu
        wp = w.permute(2, 3, 1, 0).reshape(4, Cout, Cin).to(x.dtype).contiguous()
        y = K.empty_nhwc(N, Cout, 2 * H, 2 * W, x.dtype, x.device)
        bias = b.float().contiguous()
        # head = narrow_head_prepack(...) of the 1x1 conv that follows (the mask logits): every parity launch computes it for its
        # own output positions in its epilogue -- the 2N x 2N map is not read back by a launch of its own.  All four launches make
        # the same kernel choice: the first one that is not served switches the request off.
        hd = None
        if head is not None and x.dtype == K.L.act16():
            c4 = int(head[0].shape[-2])
            hd = dict(w=head[0].reshape(c4, -1), b=head[1], out=K.empty_nhwc(N, c4, 2 * H, 2 * W, torch.float32, x.device))
        # ONE launch whose four channel tiles are the four taps (the input tile is read from HBM once) where the library serves
        # it; else one launch per output parity
        if not K.deconv2x2_fwd(x, wp, bias, y, relu=True, head=hd):
            nf = 0
            for py in range(2):
                for px in range(2):
                    K.conv_tap(x, wp, y, N, H, W, Cin, Cout, H, W, 2 * H, 2 * W, [(0, 0, py * 2 + px)], ss=1, os=2,
                               oo=(py, px), bias=bias, relu=True, head=hd)
                    if hd is not None:
                        if hd['fused']:
                            nf += 1
                        else:
                            if nf:
                                raise K.L.LoftHipError('deconv2x2_relu: the head epilogue served some parity launches only')
                            hd = None
        ctx.save_for_backward(x, w, y)
        _DeconvFn.head_out = hd['out'] if hd is not None else None     # (picked up by deconv2x2_relu; no autograd history)
        return _begin_uses(y)

    @staticmethod
    @K.planes_scoped
    def backward(ctx, g):
        x, w, y = ctx.saved_tensors
        N, Cin, H, W = x.shape
        Cout = w.shape[1]
        g = to_nhwc(g)
        if g.dtype != x.dtype:
            g = g.to(x.dtype)
        if not _premasked(g, y):   # (else the 1x1 logits head already applied this ReLU's mask)
            g = K.relu_bwd(g, y)
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            wt = w.permute(2, 3, 0, 1).reshape(4, Cin, Cout).to(x.dtype).contiguous()
            gx = K.empty_nhwc(N, Cin, H, W, x.dtype, x.device)
            taps = [(py, px, py * 2 + px) for py in range(2) for px in range(2)]
            # x is the ReLU output of the last mask conv (fcn_mask_head.py:117-124): its mask rides in this epilogue
            K.conv_tap(g, wt, gx, N, 2 * H, 2 * W, Cout, Cin, H, W, H, W, taps, ss=2, mask=x if ctx.input_relu else None)
            if ctx.input_relu:
                gx._loft_premasked = x.data_ptr()
        if ctx.needs_input_grad[1]:
            taps = [(py, px, 0, 0, py * 2 + px) for py in range(2) for px in range(2)]
            db = torch.zeros(1, Cout, dtype=torch.float32, device=g.device) if ctx.needs_input_grad[2] else None
            dwp = K.conv_wgrad(g, x, N, 2 * H, 2 * W, Cout, H, W, Cin, H, W, taps, 4, gos=2, ss=1, db=db, db_tap=-2)
            gw = dwp[0].view(2, 2, Cout, Cin).permute(3, 2, 0, 1)
            gb = db[0] if db is not None else None
        elif ctx.needs_input_grad[2]:
            gb = K.colsum(g, Cout)
        return gx, gw, gb, None, None


def deconv2x2_relu(x, w, b, input_relu=False, head=None):
    """input_relu: x is a ReLU output -- the backward folds that mask into the data gradient (no separate pass).
    head = narrow_head_prepack(w_head, b_head, x.dtype): -> (y, o | None), o = that 1x1 head applied to y, fp32 NHWC [N,c4,2H,2W],
    computed by the deconvolution's own launches when the library serves it; pass it to narrow_head(y, ..., precomputed=o), which
    keeps the head's autograd node."""
    if head is None:
        return _DeconvFn.apply(x, w, b, input_relu)
    _DeconvFn.head_out = None
    y = _DeconvFn.apply(x, w, b, input_relu, head)
    o, _DeconvFn.head_out = _DeconvFn.head_out, None
    return y, o


# ------------------------------------------------------------------ feature-gradient hub
# The FPN maps feed four consumers (RPN head, bbox / mask / offset RoI extractors).  Plain autograd materialises one gradient
# map per consumer and sums them pairwise: 12 full-map adds + 5 memsets per step (0.75 ms at batch 8 x 1024^2).  Under the
# trainer, feat_hub() hands every consumer its own alias of each map (so autograd never sums behind the kernels' back) and the
# consumers' backward kernels ACCUMULATE into one shared gradient map per level: the first to run creates it (RoIAlign backward
# writes every pixel; the sparse RPN backward starts from zeros), later ones add in place and return None for that input.  The
# hub node's backward then just forwards the shared map (plus any gradient of a consumer that did not take part).
# ---- gradient join: a backbone stage output with TWO consumers (C3 / C4: the next stage's first block and the FPN lateral) ---------
# Plain autograd hands the producer the SUM of the two data gradients, formed by an elementwise add over the whole map (402 + 201 MB
# through HBM per step for C3 / C4).  Under the trainer the residual block that consumes x registers it in forward (JOIN[x] = None);
# the other consumer -- a plain conv node, which autograd runs first because it was created later -- DEPOSITS its data gradient
# there and returns None, and the block's backward feeds the deposit to its first data-gradient launch as the epilogue's `residual`
# (fp32 add before the rounding).  Order-safe: a consumer that finds no open entry (the block already ran, or never registered)
# returns its gradient to autograd as before.
JOIN = None            # {data_ptr of a block input: None (open) | deposited gradient} for the step being built / differentiated

HUB_ENABLED = False    # set by the Trainer for the duration of a train_step
HUB = None             # {feature data_ptr: shared gradient map | None} of the step being differentiated


class FeatFork(tuple):
    """The feature pyramid as a tuple (the first consumer's aliases) + ``branches``: one alias tuple per consumer."""
    branches = ()


class _FeatHubFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, n, *feats):
        _note_use(*feats)
        ctx.n, ctx.nf = n, len(feats)
        ctx.set_materialize_grads(False)      # consumers that accumulated into the shared map return None: keep it None
        return tuple(f.view_as(f) for _ in range(n) for f in feats)

    @staticmethod
    def backward(ctx, *gs):
        _hub_flush()                          # (lists of extractors whose siblings never ran a backward)
        out = []
        for l in range(ctx.nf):
            parts = [gs[c * ctx.nf + l] for c in range(ctx.n) if gs[c * ctx.nf + l] is not None]
            if not parts:
                out.append(None)
                continue
            tot = parts[0]
            for extra in parts[1:]:
                tot = tot + extra.to(tot.dtype)
            out.append(tot)
        return (None,) + tuple(out)


def feat_hub(feats, n):
    """-> FeatFork of ``n`` alias sets of ``feats`` whose consumers share one gradient map per level (see above)."""
    global HUB
    outs = _FeatHubFn.apply(n, *feats)
    nf = len(feats)
    fork = FeatFork(outs[:nf])
    fork.branches = tuple(tuple(outs[c * nf:(c + 1) * nf]) for c in range(n))
    HUB = {f.data_ptr(): None for f in feats}
    HUB.update(_pending=[], _fresh=set(), _expected=0, _seen=0,
               _stream=torch.cuda.current_stream() if feats[0].is_cuda else None)   # the stream the shared maps are updated on
    return fork


ROI_BWD_FUSED = True   # hub-managed RoIAlign backward calls wait for each other and run as ONE pass per level


def _hub_flush():
    """Launch the RoIAlign backward lists the hub has collected: one pass per pyramid level over all of them, every pixel
    of the shared maps written once (kernels.roi_align_bwd_multi).  Called when the last expected list has arrived, and
    before anything else touches the maps (the sparse RPN backward, the hub node's own backward)."""
    if HUB is None or not HUB.get('_pending'):
        return
    pend, HUB['_pending'] = HUB['_pending'], []
    while pend:
        keys, shapes, strides, fs = pend[0][:4]
        group = [p for p in pend if p[:4] == pend[0][:4]][:3]
        pend = [p for p in pend if not any(p is q for q in group)]
        maps = [HUB[k] for k in keys]
        fresh = [k in HUB['_fresh'] for k in keys]
        if not all(fresh):
            for m, f in zip(maps, fresh):
                if f:
                    m.zero_()
        K.roi_align_bwd_multi([p[4:] + (True,) for p in group], shapes, strides, fs,
                              grad_feats=None if all(fresh) else maps, out_dtype=K.L.act16(), out=maps)
        HUB['_fresh'].difference_update(keys)


def _hub_slots(tensors):
    """Per tensor: None (not hub-managed), or its key in HUB."""
    if HUB is None:
        return [None] * len(tensors)
    return [t.data_ptr() if t.data_ptr() in HUB else None for t in tensors]


class _RoIAlignFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rois, P, strides, finest_scale, n_rot, *feats):
        _note_use(*feats)
        ctx.save_for_backward(rois)
        ctx.meta = (P, tuple(strides), finest_scale, n_rot, [tuple(f.shape) for f in feats], feats[0].dtype)
        ctx.hub_keys = _hub_slots(feats)
        if HUB is not None and all(k is not None for k in ctx.hub_keys):
            HUB['_expected'] += 1
        return K.roi_align_fwd(list(feats), rois, P, strides, finest_scale, n_rot)

    @staticmethod
    def backward(ctx, g):
        (rois,) = ctx.saved_tensors
        P, strides, fs, n_rot, shapes, dt = ctx.meta
        g = to_nhwc(g)
        direct = dt == K.L.act16() and g.dtype == K.L.act16() and not DBG.roi_fp32_bwd   # bf16 maps straight from fp32 registers
        keys = ctx.hub_keys
        if direct and HUB is not None and all(k is not None and k in HUB for k in keys):
            if ROI_BWD_FUSED:                         # wait for the other extractors' lists; the last one launches
                ret = []
                for i, k in enumerate(keys):
                    if HUB[k] is None:
                        HUB[k] = K.empty_nhwc(*shapes[i], K.L.act16(), g.device)
                        HUB['_fresh'].add(k)
                        ret.append(HUB[k])
                    else:
                        ret.append(None)
                HUB['_pending'].append((tuple(keys), tuple(shapes), strides, fs, g, rois, P, n_rot))
                HUB['_seen'] += 1
                if HUB['_seen'] >= HUB['_expected']:
                    _hub_flush()
                return (None, None, None, None, None) + tuple(ret)
            have = [HUB[k] for k in keys]
            if all(h is None for h in have):          # first consumer of these maps: the kernel writes every pixel
                grads = K.roi_align_bwd(g, rois, shapes, P, strides, fs, n_rot, rois_sorted=True, out_dtype=K.L.act16())
                for k, x in zip(keys, grads):
                    HUB[k] = x
                return (None, None, None, None, None) + tuple(grads)
            ret = []
            for i, k in enumerate(keys):               # levels nobody has touched yet start from zeros
                if have[i] is None:
                    have[i] = HUB[k] = K.zeros_nhwc(*shapes[i], K.L.act16(), g.device)
                    ret.append(have[i])
                else:
                    ret.append(None)
            K.roi_align_bwd(g, rois, shapes, P, strides, fs, n_rot, grad_feats=have, rois_sorted=True)
            return (None, None, None, None, None) + tuple(ret)
        grads = K.roi_align_bwd(g, rois, shapes, P, strides, fs, n_rot, rois_sorted=True,       # (rois in bbox2roi order)
                                out_dtype=K.L.act16() if direct else torch.float32)
        return (None, None, None, None, None) + tuple(x if x.dtype == dt else K.cast_bf16(x) for x in grads)   # (fp32 maps: as is)


def roi_align(feats, rois, P, strides, finest_scale=56, n_rot=1):
    before = HUB['_expected'] if HUB is not None else None
    y = _RoIAlignFn.apply(rois, P, tuple(strides), finest_scale, n_rot, *feats)
    if before is not None and HUB['_expected'] != before:
        y._loft_hub_counted = True            # (see roi_align_discard)
    return y


def roi_align_discard(y):
    """The caller drops the RoIAlign result ``y`` without ever differentiating through it (the RoI head's speculative bbox
    features when the sampler under-fills): the hub must not wait for that node's backward list, or the fused RoIAlign backward
    would only launch from the fallback flushes (roi.py)."""
    if HUB is not None and getattr(y, '_loft_hub_counted', False):
        HUB['_expected'] -= 1
        y._loft_hub_counted = False


class _FpnTopDownFn(torch.autograd.Function):
    """laterals[i-1] += nearest_x2(laterals[i]), coarsest to finest, in place (necks/fpn.py:176-181)."""

    @staticmethod
    def forward(ctx, *lats):
        _note_use(*lats)
        lats = list(lats)
        for i in range(len(lats) - 1, 0, -1):
            K.upsample2x_add_(lats[i - 1], lats[i])
        ctx.mark_dirty(*lats[:-1])
        return tuple(lats)

    @staticmethod
    def backward(ctx, *gs):
        if all(g.dtype == K.L.act16() and g.is_cuda for g in gs):
            # out[i] = gs[i] + blocksum(out[i-1]) into fresh maps: the incoming gradients are only read (no private clones)
            out = [to_nhwc(gs[0])]
            for i in range(1, len(gs)):
                out.append(K.downsum2x_sum(to_nhwc(gs[i]), out[i - 1]))
            return tuple(out)
        # gs[i] (i >= 1) are accumulated into in place -> private copies; the finest map is only read (no 268 MB clone)
        gs = [to_nhwc(g) if i == 0 else to_nhwc(g).clone() for i, g in enumerate(gs)]
        for i in range(1, len(gs)):
            K.downsum2x_add_(gs[i], gs[i - 1])
        return tuple(gs)


def fpn_top_down(lats):
    return _FpnTopDownFn.apply(*lats)


class _Subsample2Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _note_use(x)
        ctx.shape = tuple(x.shape)
        return K.subsample2(x)

    @staticmethod
    def backward(ctx, g):
        n, c, h, w = ctx.shape
        big = K.zeros_nhwc(n, c, h, w, g.dtype, g.device)
        return K.subsample2_adjoint_add_(big, to_nhwc(g))


def subsample2(x):
    return _Subsample2Fn.apply(x)


# ------------------------------------------------------------------ HRNet / HRFPN pieces (BASELINE config 5)

class _FuseSumReluFn(torch.autograd.Function):
    """y = relu(sum_j nearest_up(term_j, 2^shift_j)) -- HRModule fuse (mmdet/models/backbones/hrnet.py:177-195)."""

    @staticmethod
    def forward(ctx, shifts, *terms):
        _note_use(*terms)
        y = K.fuse_sum_relu(list(terms), shifts, relu=True)
        ctx.shifts = shifts
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        g = to_nhwc(g)
        if g.dtype != y.dtype:
            g = g.to(y.dtype)
        cache = {}
        grads = []
        for j, s in enumerate(ctx.shifts):
            if not ctx.needs_input_grad[1 + j]:
                grads.append(None)
                continue
            if s not in cache:
                cache[s] = K.blocksum_masked(g, y, s)
            grads.append(cache[s])
        return (None,) + tuple(grads)


def fuse_sum_relu(terms, shifts):
    return _FuseSumReluFn.apply(tuple(shifts), *terms)


class _HRFPNConcatFn(torch.autograd.Function):
    """torch.cat([x0] + [F.interpolate(x_i, scale_factor=2^i, mode='bilinear')], 1) (mmdet/models/necks/hrfpn.py:79-85),
    each term written straight into its channel slot."""

    @staticmethod
    def forward(ctx, *xs):
        _note_use(*xs)
        B, _, H, W = xs[0].shape
        ctot = sum(x.shape[1] for x in xs)
        out = K.empty_nhwc(B, ctot, H, W, xs[0].dtype, xs[0].device)
        off = 0
        for i, x in enumerate(xs):
            K.bilinear_up_slot_(x, out, i, off)
            off += x.shape[1]
        ctx.chans = [x.shape[1] for x in xs]
        return out

    @staticmethod
    def backward(ctx, g):
        g = to_nhwc(g)
        grads, off = [], 0
        for i, c in enumerate(ctx.chans):
            grads.append(K.bilinear_up_slot_bwd(g, c, i, off) if ctx.needs_input_grad[i] else None)
            off += c
        return tuple(grads)


def hrfpn_concat(xs):
    return _HRFPNConcatFn.apply(*xs)


class _AvgPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, shift):
        _note_use(x)
        ctx.shift = shift
        return K.avgpool(x, shift)

    @staticmethod
    def backward(ctx, g):
        return K.avgpool_bwd(to_nhwc(g), ctx.shift), None


def avgpool(x, shift):
    return _AvgPoolFn.apply(x, shift)


class _Stem3x3Fn(torch.autograd.Function):
    """relu(bn(conv3x3/2(img))) for the 3-channel image (HRNet conv1/norm1, hrnet.py:273-281,481-483)."""

    @staticmethod
    def forward(ctx, img, w, gamma, beta, mean, var, eps, out_dtype):
        scale = gamma * torch.rsqrt(var + eps)
        y = K.stem3x3s2_bn_relu(img, w, scale, beta - mean * scale, out_dtype)
        ctx.save_for_backward(img, w, gamma, beta, mean, var, y)
        ctx.eps = eps
        return y

    @staticmethod
    def backward(ctx, g):
        img, w, gamma, beta, mean, var, y = ctx.saved_tensors
        g = to_nhwc(g)
        if g.dtype != y.dtype:
            g = g.to(y.dtype)
        dwp, db = K.stem3x3s2_wgrad(img, g, y)
        dw, dg, dbeta = K.fold_unpack_bwd(dwp, db, w, (gamma, beta, mean, var), ctx.eps)
        return None, dw, dg, dbeta, None, None, None, None


def stem3x3s2(img, w, bn, out_dtype=None):
    return _Stem3x3Fn.apply(img, w, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, out_dtype)


# ------------------------------------------------------------------ sparse backward of the RPN head

def _as_img(t2d):
    """[n, C] row-major -> a channels_last [1, C, n, 1] view (what the tap / wgrad kernels index as n pixels of C channels)."""
    return t2d.view(1, t2d.shape[0], 1, t2d.shape[1]).permute(0, 3, 1, 2)


class _SparseRPNFn(torch.autograd.Function):
    """Differentiable handle on the RPN head outputs AT THE SAMPLED ANCHORS only.

    forward : returns ``vals`` [nsel, 5] (objectness logit + 4 deltas of each sampled anchor), gathered by the caller from the
              dense head outputs it computed without autograd (those also feed proposal generation).
    backward: the gradient of the dense outputs is zero except at <= 256 anchors per image, so instead of dense dgrad / wgrad
              launches over five pyramid levels (3.5 ms per step at 8 x 1024^2) it runs four small dense GEMMs over the nsel
              selected pixels on the same MFMA kernels, with loft_rpn_gather_rows / loft_rpn_scatter_add_rows moving the rows:
                gh   = relu'(h_sel) * (g_rows x W_head)                    [nsel,256]
                dW_head, db_head = g_rows^T x h_sel                         (cls + reg 1x1 convs)
                dW_conv, db_conv = gh^T x gather3x3(x)                      [256, 9*256]
                dx[level][pixel + tap] += gh x W_conv[tap]                  (scatter, bf16 packed atomics)
    Reference semantics: autograd of rpn_head.py:38-54 under the losses of anchor_head.py:429-497."""

    @staticmethod
    def forward(ctx, vals, rows, slot, A, nlev, w_conv, b_conv, w_cls, b_cls, w_reg, b_reg, *maps):
        _note_use(*maps)
        xs, hs = maps[:nlev], maps[nlev:]
        h_sel = K.rpn_gather_rows(list(hs), rows, 1)
        ctx.save_for_backward(rows, slot, w_conv, w_cls, w_reg, h_sel, *xs)
        ctx.A = A
        ctx.params = (w_conv, b_conv, w_cls, b_cls, w_reg, b_reg)        # the Parameter objects (their .grad may be an arena slot)
        _count_uses(*[p for p, need in zip(ctx.params, ctx.needs_input_grad[5:11]) if need])
        return vals.view_as(vals)     # (a fresh tensor object for autograd to hang the node on; same storage, no copy launch)

    @staticmethod
    @K.planes_scoped
    def backward(ctx, g):
        rows, slot, w_conv, w_cls, w_reg, h_sel = ctx.saved_tensors[:6]
        xs = ctx.saved_tensors[6:]
        A, nsel, dev = ctx.A, rows.shape[0], rows.device
        P = _NarrowHeadFn.PADW
        C = w_conv.shape[0]
        # output-gradient rows in the fused head's channel order (cls 0..A-1, reg A..5A-1), zero padded to 128 channels, and the two
        # dgrad operands [C, P] / [(tap, cin), cout]: one launch
        g_rows, w_headT, wd = K.rpn_sparse_prep(g, slot, A, P, w_cls, w_reg, w_conv)
        gimg = _as_img(g_rows)
        # gh = relu'(h) * (g_rows x W_head): data gradient of the 1x1 heads with the ReLU mask in the epilogue
        gh = K.conv2d_dgrad(gimg, w_headT[None, None], (nsel, 1), 1, 1, mask=_as_img(h_sel))
        dwp, db = K.conv2d_wgrad(gimg, _as_img(h_sel), 1, 1, with_bias=True)
        g_wcls, g_wreg = dwp[0, 0, :A].reshape(w_cls.shape), dwp[0, 0, A:5 * A].reshape(w_reg.shape)
        g_bcls, g_breg = db[0, :A], db[0, A:5 * A]
        gh2d = gh.permute(0, 2, 3, 1).reshape(nsel, C)
        xg = K.rpn_gather_rows(list(xs), rows, 3)
        dwc, dbc = K.conv2d_wgrad(gh, _as_img(xg), 1, 1, with_bias=True)
        g_wconv = dwc[0, 0].view(C, 9, C).permute(0, 2, 1).reshape(w_conv.shape)
        dxs = K.conv2d_fwd(_as_img(gh2d), wd[None, None], None, 1, 1)
        dxs = dxs.permute(0, 2, 3, 1).reshape(nsel, 9 * C)
        # Everything above ran on THIS node's stream -- the RPN losses' own stream when rpn.loss_fused forked one: ~25 small launches
        # that then execute beside the RoI heads' backward instead of behind the RoIAlign backward (they were 0.23 ms of thin
        # launches on the critical path).  What follows touches the shared per-level gradient maps, which live on the hub's
        # stream: join it for the flush + scatter, so that they stay ordered with the RoI extractors' pass before and the hub
        # node's backward after.
        cur = torch.cuda.current_stream() if g.is_cuda else None
        hub_stream = HUB.get('_stream') if HUB is not None else None
        switch = cur is not None and hub_stream is not None and hub_stream != cur
        if switch:
            hub_stream.wait_stream(cur)
        with (torch.cuda.stream(hub_stream) if switch else contextlib.nullcontext()):
            _hub_flush()                               # the RoI extractors' maps are complete before rows are added
            dxl, ret = [], []
            for x in xs:                               # hub-managed maps: scatter into the shared gradient map of the level
                k = x.data_ptr() if (HUB is not None and x.data_ptr() in HUB and x.dtype == K.L.act16()) else None
                if k is not None and HUB[k] is not None:
                    dxl.append(HUB[k])
                    ret.append(None)
                else:
                    z = torch.zeros_like(x)
                    if switch:
                        z.record_stream(cur)           # (allocated from the hub stream's pool, consumed by autograd on `cur`: ADVICE r4)
                    if k is not None:
                        HUB[k] = z
                    dxl.append(z)
                    ret.append(z)
            K.rpn_scatter_add_rows_(dxl, rows, dxs, 3)
            if switch:
                dxs.record_stream(hub_stream)
        if switch:
            cur.wait_stream(hub_stream)                # (autograd orders this node's consumers behind `cur`)
        pw_conv, pb_conv, pw_cls, pb_cls, pw_reg, pb_reg = ctx.params
        if all(ctx.needs_input_grad[5:11]) and _queue_param_grads([
                (pw_conv, dwc[0, 0], pb_conv, dbc[0], (C, 3, 3)),                 # dwc is [cout, (tap, cin)]: the flattened-map form
                (pw_cls, dwp[0, 0, :A], pb_cls, db[0, :A], None),
                (pw_reg, dwp[0, 0, A:5 * A], pb_reg, db[0, A:5 * A], None)]):
            pgrads = (None,) * 6
        else:
            # (no queue / no slots: autograd accumulates; the uses counted in forward are not sunk -- as _NarrowHeadFn does)
            for p_, need in zip(ctx.params, ctx.needs_input_grad[5:11]):
                if need and p_ is not None and getattr(p_, '_loft_pending', 0) > 0:
                    p_._loft_pending -= 1
            pgrads = (g_wconv, dbc[0], g_wcls, g_bcls, g_wreg, g_breg)
        return (None, None, None, None, None) + pgrads + tuple(ret) + (None,) * len(xs)


def rpn_sparse_outputs(vals, rows, slot, A, xs, hs, w_conv, b_conv, w_cls, b_cls, w_reg, b_reg):
    return _SparseRPNFn.apply(vals, rows, slot, A, len(xs), w_conv, b_conv, w_cls, b_cls, w_reg, b_reg, *xs, *hs)


# ------------------------------------------------------------------ residual block as ONE autograd node

def _rb_pack(w, bn, cin_p, cout_p, need_dgrad, pdt):
    """BN-folded operand packings of one conv+bn of a residual block (same cache policy as _ConvFn)."""
    Cout, Cin, R, S = w.shape
    stats = (bn.running_mean, bn.running_var)
    tensors = (w, bn.weight, bn.bias)
    key = None
    if all(_cacheable(t) for t in tensors):
        key = ('rb', pdt, cin_p, cout_p, bool(need_dgrad))
        cached = _pack_cache_get(key, tensors + stats)
        if cached is not None:
            return cached
    if PREPACK is not None and key is None and pdt == K.L.act16() and cout_p % 2 == 0 and cin_p % 2 == 0 and \
            isinstance(w, torch.nn.Parameter):
        return PREPACK.request((w,), (None,), (bn.weight, bn.bias, stats[0], stats[1]), bn.eps, cout_p, cin_p, need_dgrad)
    if PREPACK is not None and key is None and pdt == torch.float32 and isinstance(w, torch.nn.Parameter) and w.dim() == 4:
        # fp32 parity mode: fold + fp32 packings + their operand planes for all convs in two launches per step
        return PREPACK.request_f32((w,), (None,), (bn.weight, bn.bias, stats[0], stats[1]), bn.eps, cout_p, cin_p, need_dgrad)
    out = K.fold_pack(w, None, (bn.weight, bn.bias, stats[0], stats[1]), bn.eps, want_fwd=True, want_dgrad=need_dgrad, dtype=pdt,
                      cout_pad=cout_p, cin_pad=cin_p)
    out = (out[0][None], None if out[1] is None else out[1][None], out[2][None])
    if key is not None:
        _pack_cache_put(key, tensors + stats, out)
    return out


def _nsplit(dwp):
    """Split-K slots of a weight gradient that came back as [G, S, T, Cout, Cin] (kernels.conv_wgrad(slots_ok=True)), else 1."""
    return dwp.shape[1] if dwp.dim() == 5 else 1


def _rb_param_grads(g, x, w, bn, k, stride, pad, needs):
    """Weight / gamma / beta gradients of one conv+bn from the (already ReLU-masked) output gradient g and the conv input x.
    Returns (dw, dgamma, dbeta) for autograd, or Nones when the kernels accumulated straight into the trainer's arena."""
    need_w, need_g, need_b = needs
    if not (need_w or need_g or need_b):
        return None, None, None
    bnt = (bn.weight, bn.bias, bn.running_mean, bn.running_var)
    slots = (_direct_slot(w) if need_w else None, _direct_slot(bn.weight) if need_g else None, _direct_slot(bn.bias) if need_b else None)
    sunk = need_w and need_g and need_b and all(s is not None for s in slots)
    if sunk and UNPACK_Q is not None and WGRAD_STREAM is not None and g.is_cuda and K.PROFILE is None:
        # Nothing on the data-gradient chain reads a weight gradient, so the launch goes to a second stream and runs beside
        # the next blocks' dgrad kernels (its split-K atomics tail and its < 256-workgroup grids leave CUs idle otherwise);
        # the unpack queue waits for it with an event (UnpackQueue.add / flush).
        cur = torch.cuda.current_stream()
        WGRAD_STREAM.wait_stream(cur)
        g.record_stream(WGRAD_STREAM)
        x.record_stream(WGRAD_STREAM)
        _mark_sunk(w, bn.weight, bn.bias)
        with torch.cuda.stream(WGRAD_STREAM):
            dwp, db = K.conv2d_wgrad(g, x, k, k, stride, pad, with_bias=True, slots_ok=True)
            UNPACK_Q.add(dwp[0], db[0], w, bnt, bn.eps, slots, [(lambda q=q: _sink_done(q)) for q in (w, bn.weight, bn.bias)],
                         nsplit=_nsplit(dwp), params=(w, bn.weight, bn.bias))
        return None, None, None
    dwp, db = K.conv2d_wgrad(g, x, k, k, stride, pad, with_bias=True, slots_ok=sunk and UNPACK_Q is not None)
    if sunk:
        _mark_sunk(w, bn.weight, bn.bias)
        if UNPACK_Q is not None:
            UNPACK_Q.add(dwp[0], db[0], w, bnt, bn.eps, slots, [(lambda q=q: _sink_done(q)) for q in (w, bn.weight, bn.bias)],
                         nsplit=_nsplit(dwp), params=(w, bn.weight, bn.bias))
            return None, None, None
        K.fold_unpack_bwd(dwp[0], db[0], w, bnt, bn.eps, out=slots)
        _sink_done(w), _sink_done(bn.weight), _sink_done(bn.bias)
        return None, None, None
    dw, dg, dbeta = K.fold_unpack_bwd(dwp[0], db[0], w, bnt, bn.eps, need_dw=need_w)
    return (dw if need_w else None), (dg if need_g else None), (dbeta if need_b else None)


class _ResBlockFn(torch.autograd.Function):
    """out = relu(main(x) + shortcut(x)) with main = conv-bn(-relu) x n and shortcut = identity | conv-bn, as ONE autograd
    node (ResNet Bottleneck, mmdet/models/backbones/resnet.py:266-298; BasicBlock, resnet.py:65-92 -- also HRNet's branches).

    Why one node: with per-conv nodes the block input feeds two consumers, so autograd adds their gradients with an aten
    kernel and the block's ReLU backward is a further pass over the same tensor.  Here the shortcut gradient enters the first
    conv's data-gradient launch as its ``residual`` and the ReLU mask of the block INPUT (the previous block's output) is
    applied in that same epilogue: no add, no separate ReLU-backward launch per block (13 + 13 launches per step on R50,
    ~190 + ~245 on HRNet-W32).  The returned gradient is marked pre-masked for the producer of x.

    specs: tuple of (k, stride, pad, cout_pad) per main conv, then the same for the shortcut conv or None.
    tensors: (w, gamma, beta) per main conv, then for the shortcut conv; bns: the FrozenStatBN modules (running stats, eps)."""

    @staticmethod
    def forward(ctx, x, specs, bns, x_is_relu_out, pair, *tensors):
        _note_use(x)
        main_specs, sc_spec = specs
        n = len(main_specs)
        ctx.pair_prev = None
        pdt = torch.float32 if x.dtype == torch.float32 else K.L.act16()
        need_dx = ctx.needs_input_grad[0]
        # A 64-plane bottleneck nobody differentiates (the frozen layer1 of the LOFT backbone; any such block at inference):
        # conv1, then 3x3 + 1x1 expansion + shortcut + ReLU as ONE launch -- the 64-channel intermediate never leaves the CU and
        # a conv shortcut is folded into the expansion's accumulation (loft_bneck_tail_bf16, conv_mfma.hip).
        if pair is not None and not (pdt == K.L.act16() and x.dtype == pdt and n == 3 and not DBG.no_pair_fusion):
            pair = None
        if (not DBG.no_bneck_fusion and pdt == K.L.act16() and x.dtype == pdt and not any(ctx.needs_input_grad) and n == 3
                and main_specs == ((1, 1, 0, None), (3, 1, 1, None), (1, 1, 0, None))
                and tuple(tensors[3].shape) == (64, 64, 3, 3) and tuple(tensors[6].shape) == (256, 64, 1, 1)
                and (sc_spec is None and x.shape[1] == 256 or sc_spec == (1, 1, 0, None) and x.shape[1] == 64
                     and tuple(tensors[9].shape) == (256, 64, 1, 1))):
            wp1, _, b1 = _rb_pack(tensors[0], bns[0], x.shape[1], 64, False, pdt)
            wp2, _, b2 = _rb_pack(tensors[3], bns[1], 64, 64, False, pdt)
            wp3, _, b3 = _rb_pack(tensors[6], bns[2], 64, 256, False, pdt)
            t1 = K.conv2d_fwd(x, wp1, b1, 1, 1, 1, 0, relu=True, out_dtype=pdt)
            if sc_spec is None:
                return _begin_uses(K.bottleneck_tail(t1, wp2, b2, wp3, b3, x))
            wpd, _, bd = _rb_pack(tensors[9], bns[3], 64, 256, False, pdt)
            return _begin_uses(K.bottleneck_tail(t1, wp2, b2, wp3, b3, x, wpd, bd))
        if JOIN is not None and need_dx and sc_spec is not None and x.dtype == K.L.act16() and x.is_cuda and not DBG.no_grad_join:
            JOIN[x.data_ptr()] = None           # (open: another consumer of x may deposit its data gradient for this block)
        acts, packs = [x], []
        h = x
        sc = x
        if sc_spec is not None:
            k, s, p, cp = sc_spec
            wp, wpt, bias = _rb_pack(tensors[3 * n], bns[n], x.shape[1], cp or tensors[3 * n].shape[0], need_dx, pdt)
            sc = K.conv2d_fwd(x, wp, bias, k, k, s, p, out_dtype=pdt, planes_cache=True)
            packs.append(wpt)
        pre = pair.pop('pre', None) if pair is not None else None       # (t1 of THIS block from the previous block's pair launch, its backward hand-over)
        for i, (k, s, p, cp) in enumerate(main_specs):
            last = i == n - 1
            w = tensors[3 * i]
            wp, wpt, bias = _rb_pack(w, bns[i], h.shape[1], cp or w.shape[0], need_dx or i > 0, pdt)
            packs.insert(i, wpt)
            if i == 0 and pre is not None and pre['x'] == x.data_ptr() and tuple(pre['t1'].shape) == (x.shape[0], w.shape[0]) + tuple(x.shape[2:]):
                h = pre['t1']                    # conv1 + bn1 + relu of this block ran in the previous block's last launch
                if need_dx and sc_spec is None and x_is_relu_out and pre['bwd'] is not None and wpt is not None:
                    # (K8 data-gradient packing of the previous block's conv3, its t2; this conv1's own K8 data-gradient packing --
                    #  taken NOW: the trainer's registry serves it only during the forward pass)
                    ctx.pair_prev = pre['bwd'] + (_k8(wpt[0, 0]),)
            elif last and pair is not None and pair.get('next') is not None and _pair_fwd_ok(h, w, (k, s, p, cp), sc, pair['next']):
                # the END of this block and the BEGINNING of the next one as one launch (loft_bneck_pair_bf16): the block output
                # goes to HBM once and is not read back as the next conv1's operand
                wn, bnn = pair['next']
                need_next = any(ctx.needs_input_grad)
                wpn, _, biasn = _rb_pack(wn, bnn, w.shape[0], wn.shape[0], need_next, pdt)
                k3, k1n = _k8(wp[0, 0]), _k8(wpn[0, 0])
                t2 = h
                h, t1n = K.bneck_pair(t2, k3, bias[0], sc, k1n, biasn[0])
                pair['pre'] = dict(x=h.data_ptr(), t1=t1n, bwd=(_k8(wpt[0, 0]), t2) if (need_next and wpt is not None) else None)
            else:
                h = K.conv2d_fwd(h, wp, bias, k, k, s, p, relu=True, residual=sc if last else None, out_dtype=pdt, planes_cache=True)
            if not last:
                acts.append(h)
        ctx.specs, ctx.bns, ctx.x_is_relu_out = specs, bns, x_is_relu_out
        ctx.params = tensors
        for k_, t in enumerate(tensors):
            if ctx.needs_input_grad[5 + k_] and isinstance(t, torch.nn.Parameter):
                t._loft_pending = getattr(t, '_loft_pending', 0) + 1
        ctx.n_saved = (len(acts), len(packs))
        ctx.save_for_backward(h, *acts, *[p for p in packs if p is not None])
        ctx.pack_none = [p is None for p in packs]
        return _begin_uses(h)

    @staticmethod
    @K.planes_scoped
    def backward(ctx, g):
        main_specs, sc_spec = ctx.specs
        n = len(main_specs)
        saved = ctx.saved_tensors
        out = saved[0]
        na, npk = ctx.n_saved
        acts = list(saved[1:1 + na])
        it = iter(saved[1 + na:])
        packs = [None if isnone else next(it) for isnone in ctx.pack_none]
        x = acts[0]
        P, bns = ctx.params, ctx.bns
        needs = ctx.needs_input_grad
        g = to_nhwc(g)
        if g.dtype != x.dtype:          # bf16 training path; the fp32 parity mode stays fp32 throughout
            g = g.to(x.dtype)
        if not _premasked(g, out):
            g = K.relu_bwd(g, out)
        grads = [None] * len(P)
        need_dx = needs[0]
        # main path, last conv to first; gk = gradient w.r.t. the output of conv k (masked by that output's ReLU)
        gk = g
        pg = PAIR_G.pop(g.data_ptr(), None) if PAIR_G else None
        for i in range(n - 1, -1, -1):
            k, s, p, cp = main_specs[i]
            xin = acts[i]
            grads[3 * i:3 * i + 3] = _rb_param_grads(gk, xin, P[3 * i], bns[i], k, s, p, needs[5 + 3 * i:8 + 3 * i])
            if i > 0:       # input of conv i is the ReLU output of conv i-1: its mask rides in this dgrad's epilogue
                if i == n - 1 and pg is not None and pg[0].data_ptr() == g.data_ptr() and pg[1].shape == xin.shape:
                    gk = pg[1]                   # the next block's backward launch already produced it (pair fusion)
                else:
                    gk = K.conv2d_dgrad(gk, packs[i], tuple(xin.shape[2:]), k, k, s, p, mask=xin, out_dtype=x.dtype)
        gx = None
        dep = JOIN.pop(x.data_ptr(), None) if JOIN is not None else None
        if need_dx:
            k, s, p, cp = main_specs[0]
            mask = x if ctx.x_is_relu_out else None
            pp = ctx.pair_prev
            if (sc_spec is None and pp is not None and mask is not None and dep is None and not DBG.no_pair_fusion and packs[0] is not None
                    and g.dtype == K.L.act16() and gk.dtype == K.L.act16() and K.bneck_pair_ok(gk, x.shape[1])):
                # this block's first data gradient and the PREVIOUS block's last one in one launch (the mirror of the forward pair):
                # gx = relu'(x) (W1^T gk + g);  previous block: g_t2 = relu'(t2) (W3^T gx) -- handed over through PAIR_G
                gx, gt2p = K.bneck_pair(gk, pp[2], None, g, pp[0], None, mask1=x, mask2=pp[1])
                PAIR_G.clear()
                PAIR_G[gx.data_ptr()] = (gx, gt2p)
            elif sc_spec is None:
                # shortcut = identity: its gradient (g) is the residual of the first conv's data gradient
                gx = K.conv2d_dgrad(gk, packs[0], tuple(x.shape[2:]), k, k, s, p, residual=g, mask=mask, out_dtype=x.dtype)
            else:
                # (mask on both launches: a strided shortcut conv only covers one output parity class, the other positions
                #  are copies of the residual and must already be masked; the mask is idempotent)
                # (dep: the data gradient another consumer of x deposited -- gradient join above -- rides in as the residual)
                gx = K.conv2d_dgrad(gk, packs[0], tuple(x.shape[2:]), k, k, s, p, residual=dep, mask=mask, out_dtype=x.dtype)
                dep = None
                ks, ss, ps, cps = sc_spec
                # (in place into the main path's gradient: a strided 1x1 shortcut writes one position in stride^2)
                gx = K.conv2d_dgrad(g, packs[n], tuple(x.shape[2:]), ks, ks, ss, ps, residual=gx, mask=mask, out_dtype=x.dtype,
                                    out=gx if (ss > 1 and ks == 1) else None)
            if dep is not None:                  # (a deposit this block's launches could not take: plain sum)
                gx = gx + dep.to(gx.dtype)
                dep = None
                if mask is not None:             # (the sum is not masked yet: the tag below would claim it is -- ADVICE r5)
                    gx = K.relu_bwd(gx, x)
            if mask is not None:
                gx._loft_premasked = x.data_ptr()
        if sc_spec is not None:
            ks, ss, ps, cps = sc_spec
            grads[3 * n:3 * n + 3] = _rb_param_grads(g, x, P[3 * n], bns[n], ks, ss, ps, needs[5 + 3 * n:8 + 3 * n])
        return (gx, None, None, None, None) + tuple(grads)


def _k8(m):
    """K8 layout of a 16-bit [rows, K] packing (loft_bneck_pair_bf16's weight operands): from the trainer's registry (one batched
    launch per step) when the packing lives there, else converted now."""
    if PREPACK is not None:
        o = PREPACK.k8(m)
        if o is not None:
            return o
    return K.pack_k8([m])[0][0]


def _pair_fwd_ok(t2, w3, spec3, sc, nxt):
    """The last conv of a bottleneck (t2 -> out, + shortcut sc) and the next block's conv1 as one loft_bneck_pair_bf16 launch?"""
    wn, bnn = nxt
    P, C = t2.shape[1], w3.shape[0]
    return (spec3 == (1, 1, 0, None) and sc is not None and sc.dtype == K.L.act16() and t2.dtype == K.L.act16() and tuple(w3.shape) == (C, P, 1, 1)
            and tuple(wn.shape) == (P, C, 1, 1) and tuple(sc.shape) == (t2.shape[0], C) + tuple(t2.shape[2:]) and K.bneck_pair_ok(t2, C))


def res_block(x, main, shortcut=None, x_is_relu_out=True, pair=None):
    """main: list of (w, bn, k, stride, pad, cout_pad); shortcut: None (identity) or one such tuple (conv + bn, no ReLU).
    pair: a dict shared by the consecutive blocks of one stage (pair fusion, loft_bneck_pair_bf16): the caller sets pair['next'] =
    (conv1 weight, bn1) of the FOLLOWING block (stride-1 1x1 on this block's output) or None; this block leaves pair['pre'] for it."""
    specs = (tuple((k, s, p, cp) for (_, _, k, s, p, cp) in main),
             None if shortcut is None else tuple(shortcut[2:6]))
    mods = list(main) + ([shortcut] if shortcut is not None else [])
    tensors = []
    for (w, bn, *_r) in mods:
        tensors += [w, bn.weight, bn.bias]
    return _ResBlockFn.apply(x, specs, tuple(m[1] for m in mods), x_is_relu_out, pair, *tensors)
