"""hipGraph capture of the fixed-shape section of the training step: backbone + neck, forward AND backward.

Why (north_star: "HIP streams and graphs instead of a tracing compiler"; VERDICT round 2, missing #4 / next #5b).  The step's
launches are issued from Python: ~27 us of host time per launch, 19.4 ms per step for the R50 model -- 5.8 ms short of the GPU's
25.2 ms at the random-init load -- and 65 of 71 ms for HRNet-W32 (config 5: ~2100 launches, host-bound outright).  More than half
of those launches belong to a section whose shapes never change from step to step: image in, five FPN maps out, and on the way back
five gradient maps in, weight gradients deposited into the trainer's flat arena.  That section is recorded ONCE into two hipGraphs
(``torch.cuda.CUDAGraph`` is the hipGraph object on ROCm) and replayed with one host call each.  Everything that depends on the
data -- RPN targets, proposals, the sampler's per-image counts, the RoI heads whose launch sizes follow the number of positives --
stays on the eager path between the two replays.

How it plugs in (no tracing, no second implementation): the SAME python code runs once under stream capture, so the graphs hold
exactly the kernels of bonai_amd/nn.py with their arguments baked in:

    forward graph   static_img -> extract_feat() -> static FPN maps (all activations live in the graphs' private pool)
    backward graph  static gradient maps -> autograd of the captured forward -> kernels deposit dW / dgamma / dbeta into the
                    arena (GRAD_SINK + UnpackQueue, flushed INSIDE the graph)

and a tiny autograd node (``_BridgeFn``) hands the static maps to the eager heads and, when their gradients arrive, copies them
into the static gradient maps and replays the backward graph.  What makes capture legal: every address the section touches is
persistent (arena slots, the PrepackRegistry's operand buffers, frozen-layer packings cached on their parameters, the private
pool), kernel argument structs are copied by value at launch, the descriptor tables of the batched unpack hold only such
addresses and are uploaded once, at capture, on a stream that is not capturing (kernels.h2d / H2D_KEEP), and nothing in the
section reads back to the host.

Not captured: the per-step BN fold + operand packing (one launch, runs before the forward replay), the optimizer, the collective.
Under data parallelism the backbone's gradient buckets are released right after the backward replay has been enqueued (the
reducer orders its collective behind the replay with an event, as for eager launches).
"""
import torch

from . import kernels as K
from . import nn as F2


class _BridgeFn(torch.autograd.Function):
    """Eager autograd <-> the two graphs.  Inputs: a dummy leaf (so the node exists in the graph) and the static maps."""

    @staticmethod
    def forward(ctx, owner, token, *maps):
        ctx.owner = owner
        return tuple(m.detach() for m in maps)

    @staticmethod
    def backward(ctx, *grads):
        ctx.owner._run_backward(grads)
        return (None, None) + (None,) * len(grads)


class FeatureGraphs:
    """Owns the two graphs of one model + trainer.  ``provider(img)`` replaces ``model.extract_feat`` inside train_step."""

    def __init__(self, trainer):
        self.tr = trainer
        self.model = trainer.model
        self.ready = False
        self.failed = None            # the exception text when capture was refused: the trainer then stays on the eager path
        self.keep = []                # device tables uploaded at capture (kernels.h2d): must outlive the graphs
        self.pack_deps = []           # (signature, source tensors, packing) of every frozen-layer packing the graphs read
        self.prepack_sig = None       # identity of the PrepackRegistry operand buffers the graphs read
        self.stale = None             # why the last set of graphs was dropped (recaptured at the next step)
        self.token = None

    # ------------------------------------------------------------------ capture
    def capture(self, img):
        """Record both graphs on the shapes of ``img``.  Called by the trainer in place of an eager forward once the prepack
        registry knows every conv of the section (step >= 2).  Nothing executes during capture; the caller replays afterwards."""
        tr, m = self.tr, self.model
        dev = img.device
        self.static_img = torch.empty_like(img)
        self.static_img.copy_(img)
        self.token = torch.zeros(1, device=dev, requires_grad=True)
        params = [p for p in list(m.backbone.parameters()) + (list(m.neck.parameters()) if m.with_neck else []) if p.requires_grad]
        self.params = params
        torch.cuda.synchronize()
        self.g_fwd, self.g_bwd = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        stream = torch.cuda.Stream()
        stream.wait_stream(torch.cuda.current_stream())
        from .loft import hrnet as H
        prev = (F2.GRAD_SINK, F2.UNPACK_Q, F2.WGRAD_STREAM, F2.PREPACK, F2.HUB_ENABLED, H.BRANCH_STREAMS)
        zp, sp = K._ZeroPool, K._ScratchPool
        zp_state, sp_state = (zp.buf, zp.off, zp.need, zp.active), (sp.buf, sp.off, sp.need, sp.active)
        zero_need = max(int(zp.need), int(zp.buf.numel()) if zp.buf is not None else 0)     # >= what the section's backward asks for
        sp.active = False
        zp.active = False
        K.H2D_KEEP = self.keep
        prev_trace, F2.PACK_TRACE = F2.PACK_TRACE, []
        # forks inside the graphs: HRNet's branches and the weight-gradient launches keep their own streams (parallel branches of
        # the captured graph; every fork is joined before the capture ends: HRModule joins its branches, the final UnpackQueue
        # flush waits for every weight-gradient launch)
        # (streams come from a pool of 32, round robin: the capture stream and its four forks must be five DIFFERENT queues, and none
        #  of them the queue of a stream the captured section uses for non-captured work -- kernels.h2d's upload stream re-draws itself)
        seen = {stream.cuda_stream}
        self.side = []
        for _ in range(64):
            st = torch.cuda.Stream()
            if st.cuda_stream not in seen:
                seen.add(st.cuda_stream)
                self.side.append(st)
            if len(self.side) == 4:
                break
        if len(self.side) < 4:
            raise RuntimeError('could not draw four distinct side streams for the capture')
        # No cyclic-garbage collection while a stream is capturing: a collected cycle may own device objects (an older trainer's
        # CUDAGraphs, events, streams) whose destructors call HIP APIs the capture forbids -- the runtime aborts the process
        # (seen as "Fatal Python error: Aborted ... Garbage-collecting" inside extract_feat).  Collect now, pause the collector
        # until both graphs are recorded.
        import gc
        gc.collect()
        gc_was_on = gc.isenabled()
        gc.disable()
        try:
            F2.PREPACK = tr.prepack
            F2.HUB_ENABLED = False
            H.BRANCH_STREAMS = self.side[:3]
            with torch.cuda.graph(self.g_fwd, stream=stream, capture_error_mode='relaxed'):
                feats = m.extract_feat(self.static_img)
            feats = tuple(feats)
            self.static_grads = tuple(torch.zeros_like(f) for f in feats)
            F2.GRAD_SINK = lambda p: None                        # (the reducer is told after each replay, not during capture)
            F2.WGRAD_STREAM = self.side[3]
            with torch.cuda.graph(self.g_bwd, pool=self.g_fwd.pool(), stream=stream, capture_error_mode='relaxed'):
                if zero_need > 0:
                    # the section's accumulation buffers: ONE slab of the graphs' private pool, zeroed by one node per replay
                    # (bump offsets are identical in every replay: the same launches in the same order)
                    zp.buf = torch.empty(int(zero_need * 1.05) + 1024, dtype=torch.float32, device=dev)
                    zp.buf.zero_()
                    zp.off, zp.need, zp.active = 0, 0, True
                    self.keep.append(zp.buf)
                F2.UNPACK_Q = K.UnpackQueue(limit=48)
                for p in params:
                    p._loft_sunk = False
                torch.autograd.backward(list(feats), grad_tensors=list(self.static_grads))
                F2.UNPACK_Q.flush()
            self.static_feats = tuple(f.detach() for f in feats)      # aliases of the static maps, cut loose from the captured graph
        finally:
            F2.GRAD_SINK, F2.UNPACK_Q, F2.WGRAD_STREAM, F2.PREPACK, F2.HUB_ENABLED, H.BRANCH_STREAMS = prev
            (zp.buf, zp.off, zp.need, zp.active), (sp.buf, sp.off, sp.need, sp.active) = zp_state, sp_state
            K.H2D_KEEP = None
            trace, F2.PACK_TRACE = F2.PACK_TRACE, prev_trace
            if gc_was_on:
                gc.enable()
        # Addresses the graphs do not own: the frozen layers' packings live in a cache on their Parameters and the
        # trainable layers' operands in the PrepackRegistry.  The graphs hold a reference to every such buffer (it cannot be freed
        # and handed to someone else under a live graph) and remember what it was computed from; provider() drops the graphs when
        # a source changed (load_state_dict / load_checkpoint mid-run, .data re-assignment) or the registry re-allocated.
        self.pack_deps = trace
        self.keep.extend(v for _, _, v in trace)
        self.prepack_sig = self._prepack_signature()
        self.keep.extend(t for grp in tr.prepack.jobs.values() for t in (grp['wp'], grp['wpt'], grp['bias']) if t is not None)
        torch.cuda.current_stream().wait_stream(stream)
        torch.cuda.synchronize()
        self.shape = tuple(img.shape)
        self.ready = True

    def _prepack_signature(self):
        reg = self.tr.prepack
        return tuple((k, grp['wp'].data_ptr(), 0 if grp['wpt'] is None else grp['wpt'].data_ptr(), grp['bias'].data_ptr())
                     for k, grp in ((k, reg.jobs[k]) for k in reg.order))

    def validate(self):
        """-> None when every address baked into the graphs still holds what the graphs expect, else the reason (the graphs are
        then dropped: ``ready`` False, recaptured by the trainer at the next step)."""
        why = None
        for sig, tensors, _ in self.pack_deps:
            if not F2._pack_sig_valid(sig, tensors):
                why = 'a frozen layer\'s weights changed after capture (load_state_dict / load_checkpoint / .data re-assignment)'
                break
        if why is None and self._prepack_signature()[:len(self.prepack_sig)] != self.prepack_sig:
            why = 'the PrepackRegistry re-allocated an operand buffer the graphs read'
        if why is not None:
            self.ready, self.stale = False, why
            self.g_fwd = self.g_bwd = None
            self.keep, self.pack_deps = [], []
        return why

    # ------------------------------------------------------------------ per step
    def provider(self, img):
        """extract_feat of the step: upload into the static image, replay the forward graph, hand the maps to autograd."""
        if tuple(img.shape) != self.shape or img.dtype != self.static_img.dtype:
            raise K.L.LoftHipError(f'FeatureGraphs were captured for images {self.shape}, got {tuple(img.shape)}: '
                                   'capture is per shape (Trainer(graph_features=True) expects fixed-size batches)')
        self.static_img.copy_(img)
        self.g_fwd.replay()
        return _BridgeFn.apply(self, self.token, *self.static_feats)

    def _run_backward(self, grads):
        for s, g in zip(self.static_grads, grads):
            if g is None:
                s.zero_()
            else:
                s.copy_(g)
        self.g_bwd.replay()
        sink = F2.GRAD_SINK
        if sink is not None:              # data parallelism: these parameters' gradients are final for the step (enqueued)
            for p in self.params:
                p._loft_sunk = True
                sink(p)
