"""RPNHead on the HIP kernels, batched over the images of the step.

Mirrors mmdet/models/dense_heads/rpn_head.py:12-168 (+ AnchorHead anchor_head.py:35-652,
BaseDenseHead.forward_train base_dense_head.py:22-59, RPNTestMixin rpn_test_mixin.py:25-38):
same constructor arguments, parameter names (rpn_conv / rpn_cls / rpn_reg), loss keys and
proposal semantics.  What changed underneath:

  * rpn_cls and rpn_reg run as ONE 1x1 contraction (15 outputs, fp32) per level;
  * target assignment is one IoU-assign launch for all images (no KxN matrix, no per-gt loop);
  * the per image x per level sort / top-k / decode / NMS python loops (rpn_head.py:116-168) are
    segment-batched: one sort, five decode launches, one NMS mask launch and one NMS scan launch for
    the whole batch, with no host synchronisation;
  * the losses are evaluated on the <=512 sampled anchors per image only (identical value: all other
    anchors carry weight 0 in anchor_head.py:180-245).
"""
import contextlib
import os

import torch
from torch import nn

from .. import kernels as K
from ..debug import DBG
from .. import nn as F2
from .backbone import ConvW
from .builder import HEADS, build_anchor_generator, build_assigner, build_bbox_coder, build_loss, build_sampler
from .core import pad_gts


TENSOR_GATHER = False   # tests: the tensor formulation of the sampled-anchor gather instead of loft_rpn_sample_gather


def _static_ready(device):
    """Called once after a cached device table has been queued for construction: wait until it IS constructed.  Cached tables
    outlive the stream that built them and are read from whichever stream the next caller runs on (the proposal chain has its own
    stream), so 'stream-ordered' is not enough for them.  One host synchronisation per table and shape, never per step."""
    if torch.device(device).type == 'cuda':
        torch.cuda.current_stream().synchronize()


@HEADS.register_module()
class RPNHead(nn.Module):
    def __init__(self, in_channels, feat_channels=256, anchor_generator=None, bbox_coder=None, reg_decoded_bbox=False,
                 background_label=0, loss_cls=None, loss_bbox=None, train_cfg=None, test_cfg=None, num_classes=1):
        super().__init__()
        self.in_channels, self.feat_channels = in_channels, feat_channels
        self.anchor_generator = build_anchor_generator(anchor_generator)
        self.num_anchors = self.anchor_generator.num_base_anchors[0]
        if any(isinstance(s, tuple) for s in self.anchor_generator.strides):
            raise NotImplementedError('RPNHead decodes with square strides (loft_rpn_decode); got ' + str(self.anchor_generator.strides))
        self.bbox_coder = build_bbox_coder(bbox_coder or dict(type='DeltaXYWHBBoxCoder'))
        self.loss_cls = build_loss(loss_cls or dict(type='CrossEntropyLoss', use_sigmoid=True, loss_weight=1.0))
        self.loss_bbox = build_loss(loss_bbox or dict(type='L1Loss', loss_weight=1.0))
        self.use_sigmoid_cls = getattr(self.loss_cls, 'use_sigmoid', True)
        if not self.use_sigmoid_cls or reg_decoded_bbox:
            raise NotImplementedError('RPN with softmax cls / decoded-box regression')
        self.cls_out_channels = 1
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        self.sampling = True
        if train_cfg is not None:
            self.assigner = build_assigner(train_cfg.assigner)
            self.sampler = build_sampler(train_cfg.get('sampler', dict(type='RandomSampler', num=256, pos_fraction=.5)))
        self.rpn_conv = ConvW(in_channels, feat_channels, 3, bias=True)
        self.rpn_cls = ConvW(feat_channels, self.num_anchors * self.cls_out_channels, 1, bias=True)
        self.rpn_reg = ConvW(feat_channels, self.num_anchors * 4, 1, bias=True)
        self._static = {}

    def init_weights(self):
        for m in (self.rpn_conv, self.rpn_cls, self.rpn_reg):
            nn.init.normal_(m.weight, 0, 0.01)
            nn.init.constant_(m.bias, 0)

    # ---------------------------------------------------------------- forward
    def forward_fused(self, feats, keep_hidden=False):
        """-> per level fp32 [B,16,H,W] NHWC: channels [0,A) objectness, [A,5A) deltas (anchor-major)."""
        A = self.num_anchors
        w = torch.cat([self.rpn_cls.weight.view(A, -1), self.rpn_reg.weight.view(4 * A, -1)], 0)
        b = torch.cat([self.rpn_cls.bias, self.rpn_reg.bias], 0)
        outs, hs = [], []
        pre = F2.narrow_head_prepack(w, b, feats[0].dtype)        # one packing for the five levels
        fuse = not torch.is_grad_enabled() and feats[0].is_cuda and feats[0].dtype == K.L.act16()
        for x in feats:
            o = None
            if fuse:    # objectness + deltas from the 3x3 conv's own epilogue where its tile holds all 256 channels (P2-P4 at 1024^2)
                h, o = F2.conv2d_with_head(x, self.rpn_conv.weight, self.rpn_conv.bias, pre, pad=1, relu=True)
            else:
                h = F2.conv2d(x, self.rpn_conv.weight, self.rpn_conv.bias, pad=1, relu=True)
            outs.append(o if o is not None else F2.narrow_head(h, w, b, prepacked=pre))
            hs.append(h)
        return (outs, hs) if keep_hidden else outs

    def forward(self, feats):
        """Reference signature: (cls_scores, bbox_preds), NCHW-shaped views of the fused outputs."""
        A = self.num_anchors
        fused = self.forward_fused(feats)
        return [f[:, :A] for f in fused], [f[:, A:5 * A] for f in fused]

    # ---------------------------------------------------------------- static geometry
    def _geometry(self, fused, device):
        return self._geometry_of(tuple((int(f.shape[2]), int(f.shape[3])) for f in fused), int(fused[0].shape[0]), device)

    def _geometry_of(self, sizes, B, device):
        key = (sizes, B, str(device))
        if key not in self._static:
            A = self.num_anchors
            n_l = [h * w * A for h, w in sizes]
            lvl_off = [0]
            for n in n_l:
                lvl_off.append(lvl_off[-1] + n)
            anchors = torch.cat(self.anchor_generator.grid_anchors(sizes, device), 0)
            seg = torch.tensor([b * lvl_off[-1] + o for b in range(B) for o in lvl_off[:-1]] + [B * lvl_off[-1]],
                               dtype=torch.int64, device=device)
            self._static[key] = dict(sizes=sizes, n_l=n_l, lvl_off=lvl_off, N=lvl_off[-1], anchors=anchors,
                                     anchors_b=anchors[None].expand(B, -1, -1).contiguous(), seg=seg,
                                     base=[b.to(device).contiguous() for b in self.anchor_generator.base_anchors])
            _static_ready(device)
        return self._static[key]

    def _flatten(self, fused):
        """[B, N] logits and [B, N, 4] deltas in the reference's (position, anchor) order."""
        A = self.num_anchors
        B = fused[0].shape[0]
        cls = torch.cat([f.permute(0, 2, 3, 1)[..., :A].reshape(B, -1) for f in fused], 1)
        reg = torch.cat([f.permute(0, 2, 3, 1)[..., A:5 * A].reshape(B, -1, 4) for f in fused], 1)
        return cls, reg

    # ---------------------------------------------------------------- targets
    def _assign_and_sample(self, geo, B, gts, ngt, dev):
        """MaxIoUAssigner over every anchor + RandomSampler (anchor_head.py:197-255 for the whole batch) -> sampled indices and the
        loss normaliser.  Depends on the static anchors and the gt boxes only -- not on anything the network computes."""
        with torch.no_grad():
            if 'nbox' not in geo:               # static: every anchor of every image is a candidate (allowed_border = -1)
                geo['nbox'] = torch.full((B,), geo['N'], dtype=torch.int32, device=dev)
                _static_ready(dev)
            gt_inds, _ = self.assigner.assign_batched(geo['anchors_b'], geo['nbox'], gts, ngt)
            smp = self.sampler.sample_batched(gt_inds)
            pidx, pval, nidx, nval = smp['pos_idx'], smp['pos_valid'], smp['neg_idx'], smp['neg_valid']
            if pval.is_cuda:
                avg = K.sampled_avg_factor(pval, nval).reshape(())
            else:
                num_pos = pval.sum(1).clamp(min=1).sum()
                num_neg = nval.sum(1).clamp(min=1).sum()
                avg = (num_pos + num_neg).float()
        return gt_inds, pidx, pval, nidx, nval, avg

    def prefetch_targets(self, img, gt_bboxes):
        """Called by the detector BEFORE the backbone runs: anchor assignment and sampling (two IoU passes over 8 x 262k anchors and the
        sampler's one-workgroup-per-image selection: ~0.35 ms of launches that fill a fraction of the chip) go to a side stream now and
        overlap the backbone, instead of sitting between the RPN convs and the RoI heads where nothing else can run
        (tools/probes/stage_events.py: that window was 0.80 ms).  loss_fused picks the result up if the geometry it sees is the
        one predicted here from the image size; otherwise it recomputes."""
        self._prefetched = None
        if not (img.is_cuda and self.sparse_backward and torch.is_grad_enabled() and K.PROFILE is None and not DBG.no_side_stream
                and not DBG.no_rpn_target_prefetch) or self.train_cfg.get('allowed_border', -1) >= 0:
            return
        dev = img.device
        sizes, (h, w) = [], (int(img.shape[2]), int(img.shape[3]))
        for st in self.anchor_generator.strides:
            st = st[0] if isinstance(st, (tuple, list)) else st
            if st & (st - 1):
                return                                  # (not a chain of stride-2 stages: no prediction, loss_fused computes)
            hh, ww = h, w
            for _ in range(int(st).bit_length() - 1):   # every stride-2 stage of the backbone / neck maps n -> ceil(n / 2)
                hh, ww = (hh + 1) // 2, (ww + 1) // 2
            sizes.append((hh, ww))
        B = int(img.shape[0])
        geo = self._geometry_of(tuple(sizes), B, dev)
        gts, ngt = pad_gts(gt_bboxes, dev)              # (on the calling stream: the RoI head reads the same cached pair)
        main = torch.cuda.current_stream()
        if getattr(self, '_tgt_stream', None) is None:
            self._tgt_stream = torch.cuda.Stream()
        side = self._tgt_stream
        side.wait_stream(main)
        gts.record_stream(side)                         # (allocated on `main`, read on `side`: the caching allocator must not hand
        ngt.record_stream(side)                         #  the blocks out again while the side stream still reads them -- ADVICE r4)
        with torch.cuda.stream(side):
            out = self._assign_and_sample(geo, B, gts, ngt, dev)
            done = torch.cuda.Event()
            done.record(side)
        self._prefetched = (geo, tuple((g.data_ptr(), g._version) for g in gt_bboxes), out, done)

    def _targets(self, geo, B, gt_bboxes, dev):
        gts, ngt = pad_gts(gt_bboxes, dev)
        pre, self._prefetched = getattr(self, '_prefetched', None), None
        if pre is not None and pre[0] is geo and pre[1] == tuple((g.data_ptr(), g._version) for g in gt_bboxes):
            main = torch.cuda.current_stream()
            main.wait_event(pre[3])
            for t in pre[2]:
                t.record_stream(main)
            return (gts, ngt) + pre[2]
        if pre is not None:
            # the prediction missed (another geometry or gt list than prefetch_targets saw): the side stream's work is dropped, but
            # not before it has finished with its inputs and outputs; say so once -- it is wasted work every step
            torch.cuda.current_stream().wait_event(pre[3])
            if not getattr(self, '_prefetch_miss_logged', False):
                import warnings
                warnings.warn('RPNHead.prefetch_targets: predicted map sizes / gt list did not match what loss_fused saw; targets are recomputed')
                self._prefetch_miss_logged = True
        return (gts, ngt) + self._assign_and_sample(geo, B, gts, ngt, dev)

    # ---------------------------------------------------------------- loss
    def loss_fused(self, fused, gt_bboxes, img_metas, gt_bboxes_ignore=None, sparse=None):
        dev = fused[0].device
        geo = self._geometry(fused, dev)
        if self.train_cfg.get('allowed_border', -1) >= 0:
            raise NotImplementedError('allowed_border >= 0 (configs/loft_foa use -1: every anchor is valid)')
        B, N = fused[0].shape[0], geo['N']
        gts, ngt, gt_inds, pidx, pval, nidx, nval, avg = self._targets(geo, B, gt_bboxes, dev)
        if sparse is not None and fused[0].is_cuda and not TENSOR_GATHER:
            # one launch: level / pixel / slot of every sampled anchor, its logit + deltas straight from the fused head outputs,
            # labels, weights and the positives' regression targets
            xs, hs = sparse
            A = self.num_anchors
            # The two RPN losses get a stream of their own.  Forward this changes nothing (a dozen small launches beside the proposal
            # chain either way); autograd replays a node on its forward stream, so BACKWARD the sparse RPN gradient -- ~25 small
            # launches -- runs beside the RoI heads' backward instead of after the RoIAlign backward on the main stream
            # (nn._SparseRPNFn.backward joins the hub's stream only for the scatter into the shared maps).
            main = torch.cuda.current_stream()
            side = None
            if torch.is_grad_enabled() and K.PROFILE is None and not DBG.no_side_stream and not DBG.no_rpn_loss_stream:
                if getattr(self, '_loss_stream', None) is None:
                    self._loss_stream = torch.cuda.Stream()
                side = self._loss_stream
                side.wait_stream(main)
            with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
                with torch.no_grad():
                    vals, rows, slot, tgt, label, w = K.rpn_sample_gather(list(fused), geo['lvl_off'], A, geo['anchors'], gts, gt_inds,
                                                                          pidx, pval, nidx, nval, self.bbox_coder.means,
                                                                          self.bbox_coder.stds)
                S = vals.shape[1]
                vals = F2.rpn_sparse_outputs(vals.reshape(-1, 5), rows, slot, A, list(xs), list(hs), self.rpn_conv.weight,
                                             self.rpn_conv.bias, self.rpn_cls.weight, self.rpn_cls.bias, self.rpn_reg.weight,
                                             self.rpn_reg.bias).view(B, S, 5)
                logit = vals[..., 0]
                pred = vals[:, :pidx.shape[1], 1:5]
                loss_cls = self.loss_cls(logit.reshape(-1, 1), label.reshape(-1), w.reshape(-1), avg_factor=avg)
                # (the positives' weights ARE pos_valid: read as one byte per row by the loss kernel, no expand / cast launches)
                loss_bbox = self.loss_bbox(pred, tgt, pval[..., None].expand_as(pred), avg_factor=avg)
            if side is not None:
                main.wait_stream(side)
                loss_cls.record_stream(main)
                loss_bbox.record_stream(main)
            return dict(loss_rpn_cls=loss_cls, loss_rpn_bbox=loss_bbox)
        with torch.no_grad():
            pos_anchor = geo['anchors'][pidx.reshape(-1)]
            pos_gt_i = (torch.gather(gt_inds, 1, pidx) - 1).clamp(min=0)
            pos_gt = torch.gather(gts, 1, pos_gt_i[..., None].expand(-1, -1, 4)).reshape(-1, 4)
            tgt = self.bbox_coder.encode(pos_anchor, pos_gt).view(B, -1, 4)
            tgt = torch.where(pval[..., None], tgt, torch.zeros_like(tgt))
        sel = torch.cat([pidx, nidx], 1)
        if sparse is None:
            cls, reg = self._flatten(fused)
            logit = torch.gather(cls, 1, sel)
            pred = torch.gather(reg, 1, pidx[..., None].expand(-1, -1, 4))
        else:       # autograd sees only the sampled anchors (bonai_amd.nn._SparseRPNFn); fused was computed without a graph
            xs, hs = sparse
            with torch.no_grad():
                cls, reg = self._flatten(fused)
                vals = torch.cat([torch.gather(cls, 1, sel)[..., None], torch.gather(reg, 1, sel[..., None].expand(-1, -1, 4))], 2)
                A = self.num_anchors
                if 'lvl_off_t' not in geo:      # static geometry, uploaded once
                    geo['lvl_off_t'] = torch.tensor(geo['lvl_off'], device=dev)
                    geo['lvl_w_t'] = torch.tensor([s[1] for s in geo['sizes']], device=dev)
                off = geo['lvl_off_t']
                lvl = torch.bucketize(sel, off[1:], right=True)
                local = sel - off[lvl]
                pix, slot = local // A, local % A
                wl = geo['lvl_w_t'][lvl]
                valid = torch.cat([pval, nval], 1)
                rows = torch.stack([torch.arange(B, device=dev)[:, None].expand_as(sel), torch.where(valid, lvl, torch.full_like(lvl, -1)),
                                    pix // wl, pix % wl], -1).reshape(-1, 4).int().contiguous()
            S = sel.shape[1]
            vals = F2.rpn_sparse_outputs(vals.reshape(-1, 5), rows, slot.reshape(-1), A, list(xs), list(hs), self.rpn_conv.weight,
                                         self.rpn_conv.bias, self.rpn_cls.weight, self.rpn_cls.bias, self.rpn_reg.weight,
                                         self.rpn_reg.bias).view(B, S, 5)
            logit = vals[..., 0]
            pred = vals[:, :pidx.shape[1], 1:5]
        label = torch.cat([pval.long(), torch.zeros_like(nval, dtype=torch.long)], 1)
        w = torch.cat([pval, nval], 1).float()
        loss_cls = self.loss_cls(logit.reshape(-1, 1), label.reshape(-1), w.reshape(-1), avg_factor=avg)
        loss_bbox = self.loss_bbox(pred, tgt, pval[..., None].float().expand_as(pred), avg_factor=avg)
        return dict(loss_rpn_cls=loss_cls, loss_rpn_bbox=loss_bbox)

    def loss(self, cls_scores, bbox_preds, gt_bboxes, img_metas, gt_bboxes_ignore=None):
        fused = [torch.cat([c, r, c.new_zeros(c.shape[0], 1, *c.shape[2:])], 1).contiguous(memory_format=torch.channels_last)
                 for c, r in zip(cls_scores, bbox_preds)]
        return self.loss_fused(fused, gt_bboxes, img_metas, gt_bboxes_ignore)

    # ---------------------------------------------------------------- proposals
    @torch.no_grad()
    def get_bboxes_fused(self, fused, img_metas, cfg=None):
        """-> (proposals fp32 [B, max_num, 5] sorted by score, padded with zeros; counts int64 [B])."""
        cfg = self.test_cfg if cfg is None else cfg
        if cfg.get('min_bbox_size', 0) > 0 or cfg.get('nms_across_levels', False):
            raise NotImplementedError('min_bbox_size > 0 / nms_across_levels (configs/loft_foa use 0 / False)')
        dev = fused[0].device
        geo = self._geometry(fused, dev)
        A, B, N = self.num_anchors, fused[0].shape[0], geo['N']
        img_shape = img_metas[0]['img_shape']
        for m in img_metas:
            if tuple(m['img_shape'][:2]) != tuple(img_shape[:2]):
                raise NotImplementedError('per-image img_shape inside one batch')
        # Seven launches, each for the whole batch and every level (tools/probes/stage_events.py: this chain is what the RoI heads
        # wait for; it was ~35 launches): scores -> two-stage top-k -> decode (+ candidate scores, + the per-image coordinate
        # maximum of batched_nms's level shift) -> NMS mask + scan (shift derived on the device) -> top-k of the survivors
        # (keep flags as a key mask) -> proposals + counts.
        nlv = len(fused)
        strides = []
        for st in self.anchor_generator.strides[:nlv]:
            if isinstance(st, (tuple, list)):
                if st[0] != st[1]:
                    raise NotImplementedError('anisotropic anchor strides in the device decode')
                st = st[0]
            strides.append(int(st))
        keys = torch.empty(B * N, dtype=torch.float32, device=dev)
        img_max = torch.empty(B, dtype=torch.float32, device=dev)
        K.rpn_scores_levels(fused, A, N, geo['lvl_off'], keys, img_max)
        topk = [min(cfg.nms_pre, n) if cfg.nms_pre > 0 else n for n in geo['n_l']]
        # only the first topk[l] entries of every (image, level) segment are read below: in-house select + sort, one workgroup per
        # segment up to 32 768 keys; the 3 x 256^2 anchors of a P2 level in two stages (top-k of ten sub-ranges, rank merge of the runs)
        skeys, sidx = K.segmented_topk_desc(keys, geo['seg'], max(topk), max_segment=max(geo['n_l']), seg_lengths=list(geo['n_l']) * B)
        coff = [0]
        for t in topk:
            coff.append(coff[-1] + t)
        C = coff[-1]
        cand = torch.empty(B, C, 4, dtype=torch.float32, device=dev)
        cscore = torch.empty(B, C, dtype=torch.float32, device=dev)
        K.rpn_decode_levels(fused, sidx, skeys, A, N, geo['lvl_off'], topk, geo['base'], strides, self.bbox_coder.means,
                            self.bbox_coder.stds, img_shape, C, coff, cand, cscore, img_max)
        # batched_nms: boxes + level * (max_coordinate + 1), one segment per (image, level)
        key = ('nms_seg', B, tuple(topk))
        if key not in self._static:
            self._static[key] = torch.tensor([b * C + o for b in range(B) for o in coff[:-1]] + [B * C], dtype=torch.int64,
                                             device=dev)
            self._static[('img_seg', B, C)] = torch.arange(B + 1, dtype=torch.int64, device=dev) * C
            _static_ready(dev)
        keep = K.nms_segmented(cand.view(-1, 4), self._static[key], cfg.nms_thr, max_segment=max(topk),
                               predicate=cfg.get('nms_predicate', 'device'), img_max=img_max, levels=nlv, covered=True)
        post = min(cfg.nms_post, cfg.max_num) if cfg.get('max_num', 0) > 0 else cfg.nms_post
        post = min(post, C)
        fs, fi = K.segmented_topk_desc(cscore.reshape(-1), self._static[('img_seg', B, C)], post, max_segment=C, key_mask=keep)
        return K.rpn_finalize(fs, fi, cand.view(-1, 4), B, C, post)

    def get_bboxes(self, cls_scores, bbox_preds, img_metas, cfg=None, rescale=False):
        """Reference return type: list of (n_i, 5) tensors."""
        fused = [torch.cat([c, r, c.new_zeros(c.shape[0], 1, *c.shape[2:])], 1).contiguous(memory_format=torch.channels_last)
                 for c, r in zip(cls_scores, bbox_preds)]
        props, counts = self.get_bboxes_fused(fused, img_metas, cfg)
        return [props[i, :int(n)] for i, n in enumerate(counts.tolist())]

    # ---------------------------------------------------------------- train / test entry points
    sparse_backward = True      # False (tests, A/B): the plain dense autograd path

    def forward_train(self, x, img_metas, gt_bboxes, gt_labels=None, gt_bboxes_ignore=None, proposal_cfg=None, **kwargs):
        if self.sparse_backward and torch.is_grad_enabled() and x[0].dtype == K.L.act16() and x[0].shape[1] % 128 == 0:
            if (proposal_cfg is not None and x[0].is_cuda and K.PROFILE is None and not DBG.no_side_stream
                    and not DBG.no_rpn_side_stream):
                return self._forward_train_two_streams(x, img_metas, gt_bboxes, gt_bboxes_ignore, proposal_cfg)
            with torch.no_grad():
                fused, hs = self.forward_fused(x, keep_hidden=True)
            losses = self.loss_fused(fused, gt_bboxes, img_metas, gt_bboxes_ignore, sparse=(x, hs))
        else:
            fused = self.forward_fused(x)
            losses = self.loss_fused(fused, gt_bboxes, img_metas, gt_bboxes_ignore)
        if proposal_cfg is None:
            return losses
        return losses, self.get_bboxes_fused([f.detach() for f in fused], img_metas, proposal_cfg)

    def _forward_train_two_streams(self, x, img_metas, gt_bboxes, gt_bboxes_ignore, proposal_cfg):
        """Target assignment + sampling + losses on the calling stream, proposal generation (score sort, decode, NMS, re-sort) on
        a second HIP stream: two chains of small launches that each keep only a few CUs busy (8 workgroups in the NMS scan and
        the sampler) and share nothing but the head's outputs.  Proposals carry no gradient, so autograd never sees the stream."""
        main = torch.cuda.current_stream()
        if getattr(self, '_prop_stream', None) is None:
            self._prop_stream = torch.cuda.Stream()
        side = self._prop_stream
        with torch.no_grad():
            fused, hs = self.forward_fused(x, keep_hidden=True)
        # The anchor tables BOTH chains read are built here, on the calling stream, before the fork.  Round 2 let the first
        # get_bboxes_fused build them lazily -- i.e. with kernels and copies queued on the SIDE stream -- while loss_fused on the
        # main stream read `anchors_b` in the IoU assignment with no ordering against them: on a model's first training step the
        # assigner could see a half-written anchor table (VERDICT round 2's nondeterministic gradient mismatch of
        # test_trainer_direct_grad_sink_matches_autograd_accumulation, rep 0).  _static_ready() additionally completes every
        # cached table before it can be handed to another stream.
        self._geometry(fused, fused[0].device)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            props, counts = self.get_bboxes_fused([f.detach() for f in fused], img_metas, proposal_cfg)
        losses = self.loss_fused(fused, gt_bboxes, img_metas, gt_bboxes_ignore, sparse=(x, hs))
        main.wait_stream(side)
        props.record_stream(main)
        counts.record_stream(main)
        return losses, (props, counts)

    def simple_test_rpn(self, x, img_metas):
        fused = self.forward_fused(x)
        return self.get_bboxes_fused(fused, img_metas)
