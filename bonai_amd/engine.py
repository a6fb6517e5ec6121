"""Training engine: flat parameter arena, fused SGD, bucketed RCCL gradient all-reduce over xGMI.

Replaces, for the LOFT path, what the reference gets from mmcv/torch:
  * MMDistributedDataParallel + torch DDP reducer (mmdet/apis/train.py:71-79): one process per GPU,
    gradient buckets laid out CONTIGUOUSLY in one flat fp32 buffer in reverse execution order
    (FOA -> mask -> bbox -> RPN -> FPN -> layer4..layer2), each all-reduced (RCCL, backend 'nccl') on a
    side HIP stream as soon as autograd has produced its last gradient, overlapping the rest of backward;
  * OptimizerHook(grad_clip=dict(max_norm=35, norm_type=2)) + torch.optim.SGD(momentum, weight_decay)
    (configs/_base_/schedules/schedule_2x_bonai.py:2-3): two kernels over the flat arena
    (loft_sumsq_f32, loft_sgd_momentum_f32) instead of ~300 per-tensor launches;
  * the step/warm-up LR schedule (schedule_2x_bonai.py:5-10).
MI355X sizing: 81.2 M trainable parameters = 325 MB fp32 = 13 buckets of the default 25 MiB (torch DDP's
bucket_cap_mb, what the reference's wrapper runs with).  xGMI gives 7 links x ~153 GB/s per GPU: a 25 MiB ring
all-reduce over 8 GPUs moves 2 * 7/8 * 25 MiB per link, ~0.3 ms -- bandwidth- rather than latency-bound, and the
first bucket (FOA + part of the mask head) is on the wire while the mask / bbox branches are still in backward.
"""
import os

import torch
import torch.distributed as dist

from . import kernels as K
from .debug import DBG


class FlatArena:
    """All trainable parameters (and their gradients / momenta) as views into flat fp32 buffers."""

    def __init__(self, model):
        self.params = [p for p in model.parameters() if p.requires_grad]
        dev = self.params[0].device
        # reverse registration order ~ the order autograd finishes gradients in
        order = list(reversed(self.params))
        self.offsets, off = {}, 0
        for p in order:
            self.offsets[id(p)] = off
            off += (p.numel() + 7) // 8 * 8
        self.numel = off
        self.data = torch.zeros(off, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(off, dtype=torch.float32, device=dev)
        self.momentum = torch.zeros(off, dtype=torch.float32, device=dev)
        for p in order:
            o = self.offsets[id(p)]
            self.data[o:o + p.numel()].copy_(p.data.reshape(-1))
            p.data = self.data[o:o + p.numel()].view(p.shape)
            p.grad = self.grad[o:o + p.numel()].view(p.shape)
        self.order = order

    def rebind_grads(self):
        for p in self.order:
            o = self.offsets[id(p)]
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * o:
                p.grad = self.grad[o:o + p.numel()].view(p.shape)


class BucketedAllReduce:
    """Gradient averaging for data parallelism: contiguous buckets, side-stream RCCL all-reduce, event-ordered."""

    def __init__(self, arena, bucket_bytes=25 << 20):
        self.arena = arena
        # LOFT_FORCE_REDUCER=1 keeps the whole hook / side-stream / RCCL path active in a 1-rank group (GPU test)
        self.enabled = dist.is_available() and dist.is_initialized() and (
            dist.get_world_size() > 1 or os.environ.get('LOFT_FORCE_REDUCER') == '1')
        self.buckets, self.param_bucket = [], {}
        # The LAST buckets (the arena's tail: the backbone's first trainable layers, whose gradients finish last) are cut small,
        # from the end: what an N-rank step cannot hide is the collective of whatever is released at the very end of backward
        # (tools/probes/bucket_timeline.py: a 22 MiB last bucket became ready 0.1 ms before the optimizer).
        span = lambda p: (arena.offsets[id(p)], arena.offsets[id(p)] + (p.numel() + 7) // 8 * 8)
        tail, caps, n_tail = [], [bucket_bytes // 6, bucket_bytes // 3, bucket_bytes * 2 // 3], len(arena.order)
        pend = []
        for p in reversed(arena.order):
            if not caps:
                break
            if pend and (span(pend[0])[1] - span(p)[0]) * 4 > caps[0]:
                tail.append(dict(start=span(pend[-1])[0], end=span(pend[0])[1], params=pend[::-1]))
                n_tail -= len(pend)
                pend = []
                caps.pop(0)
                if not caps:
                    break
            pend.append(p)
        if len(tail) < 3 or n_tail < len(arena.order) // 2:      # tiny models (tests): no taper
            tail, n_tail = [], len(arena.order)
        head = arena.order[:n_tail]
        head_end = tail[-1]['start'] if tail else arena.numel
        start, pending = 0, []
        for p in head:
            o, end = span(p)
            if pending and (end - start) * 4 > bucket_bytes:     # a bucket exceeds the cap only for one oversized tensor (fc1)
                self.buckets.append(dict(start=start, end=o, params=pending))
                start, pending = o, []
            pending.append(p)
            if (end - start) * 4 >= bucket_bytes:
                self.buckets.append(dict(start=start, end=end, params=pending))
                start, pending = end, []
        if pending:
            self.buckets.append(dict(start=start, end=head_end, params=pending))
        self.buckets += tail[::-1]
        for bi, b in enumerate(self.buckets):
            for p in b['params']:
                self.param_bucket[id(p)] = bi
        self.on_gpu = arena.data.is_cuda
        self.stream = torch.cuda.Stream() if (self.enabled and self.on_gpu) else None
        self.works = []
        # Host cost of the reducer (+1.47 ms per step on the one-rank leg, none of it bytes).  autograd calls a
        # python post-accumulate hook for EVERY leaf -- ~300 per step -- although almost every gradient of this model is deposited
        # by a kernel and reported through the trainer's sink (the hook then only returns).  Hooks are therefore PRUNED: a
        # parameter that reported through the sink in PRUNE_AFTER consecutive steps loses its autograd hook; should it ever come
        # through autograd again (a debug switch, a fallback path, a rank without that branch) nothing is lost -- its bucket is
        # launched by finish(), in index order like every bucket, only later -- and finish() re-arms the hook of every parameter
        # that did not report, so the steady state is reached again.  Events of the hand-over to the side stream are created once.
        self._hooks, self._sunk_streak = {}, {}
        self.direct_issue = os.environ.get('LOFT_REDUCER_SIDE_STREAM_ONLY') != '1'
        self._issued_from = set()
        self._dryrun = os.environ.get('LOFT_REDUCER_DRYRUN') == '1' and self.enabled and dist.get_world_size() == 1
        self.PRUNE_AFTER = 2
        self.prune_hooks = os.environ.get('LOFT_REDUCER_KEEP_HOOKS') != '1'
        self._events = {}
        if self.enabled:
            for p in arena.order:
                self._arm(p)
        self._remaining = None
        self.measure, self.exposed = False, []      # bench.py: record (backward end, last collective end) event pairs per step
        self.capturing = False     # True while bonai_amd.graphs records a section: nothing may be released or launched from it

    def _arm(self, p):
        if id(p) not in self._hooks:
            self._hooks[id(p)] = p.register_post_accumulate_grad_hook(self._autograd_hook)
            self._sunk_streak[id(p)] = 0

    def hooks_armed(self):
        return len(self._hooks)

    def begin(self):
        self._remaining = [len(b['params']) for b in self.buckets]
        self._seen = set()
        self._via_sink = set()
        self._next = 0
        self.works = []
        self._streams = [dict() for _ in self.buckets]
        self._queued, self._nq = set(), [0] * len(self.buckets)
        self._issued_from = set()
        if self.on_gpu:
            self._dev = torch.cuda.current_device()

    def note_queued(self, params):
        """kernels.UnpackQueue.add: these parameters' gradient deposits now sit in the unpack queue (not launched yet).
        -> True when the next bucket in line has nothing else outstanding, i.e. the queue should flush NOW: buckets are
        released when their last gradient is computed, not when the queue happens to fill up."""
        for p in params:
            k = id(p)
            if getattr(p, '_loft_pending', 1) <= 1 and k not in self._seen and k not in self._queued and k in self.param_bucket:
                self._queued.add(k)
                self._nq[self.param_bucket[k]] += 1
        nb = self._next
        return nb < len(self.buckets) and self._remaining[nb] > 0 and self._remaining[nb] == self._nq[nb]

    def _autograd_hook(self, p):
        """autograd's post-accumulate callback.  It also runs for parameters whose Function returned None because a kernel
        deposits the gradient in the arena itself -- possibly later, at the next UnpackQueue flush: those report through the
        gradient sink (Trainer._sink -> _hook) once the deposit has been enqueued, never from here."""
        if getattr(p, '_loft_sunk', False) or self.capturing:
            return
        self._hook(p)

    def _sink(self, p):
        """The kernels' direct arena sink (bonai_amd.nn.GRAD_SINK): as _hook, and remembered as a sink report for hook pruning."""
        self._via_sink.add(id(p))
        self._hook(p)

    def _hook(self, p):
        """Gradient of ``p`` is final for this step (autograd post-accumulate hook, or the kernels' direct arena sink)."""
        if id(p) in self._seen or self.capturing:       # idempotent: a parameter must never release its bucket twice
            return
        self._seen.add(id(p))
        bi = self.param_bucket[id(p)]
        self._remaining[bi] -= 1
        if id(p) in self._queued:
            self._queued.discard(id(p))
            self._nq[bi] -= 1
        if self.on_gpu:
            # the RoI head's mask / bbox branches replay their backward on a second stream: a bucket can hold gradients
            # deposited on different streams, and its collective has to wait for the tail of every one of them
            raw = torch._C._cuda_getCurrentRawStream(self._dev)
            if raw not in self._streams[bi]:
                self._streams[bi][raw] = torch.cuda.current_stream()
        # collectives must be issued in the SAME order on every rank even when a rank's graph lacks some branch
        # (e.g. no positive RoI -> no mask/FOA gradients there): buckets are launched strictly in index order
        while self._next < len(self.buckets) and self._remaining[self._next] == 0:
            self._launch(self._next)
            self._next += 1

    def _launch(self, bi):
        b = self.buckets[bi]
        view = b.get('view')
        if view is None:
            view = b['view'] = self.arena.grad[b['start']:b['end']]
        if not self.on_gpu:   # gloo / CPU (tests): same bucket order, no stream juggling
            self.works.append(dist.all_reduce(view, async_op=True))
            return
        if self._dryrun:          # (LOFT_REDUCER_DRYRUN=1, one-rank experiments only: every host step of a release but the collective)
            return
        raw_cur = torch._C._cuda_getCurrentRawStream(self._dev)
        producers = self._streams[bi]
        if self.direct_issue and all(r == raw_cur for r in producers):
            # every gradient of the bucket was deposited on THIS stream (the backbone's buckets: the batched unpack runs on the
            # main stream): c10d's own stream takes an event of the calling stream -- which does not wait -- so the hop through
            # the side stream (two event operations and a stream switch per bucket on the host) buys nothing
            self.works.append(dist.all_reduce(view, async_op=True))
            self._issued_from.add(raw_cur)
            return
        cur = torch.cuda.current_stream()
        producers = dict(producers)
        producers[raw_cur] = cur
        for raw, st in producers.items():
            ev = self._events.get((bi, raw))
            if ev is None:
                ev = self._events[(bi, raw)] = torch.cuda.Event()
            ev.record(st)
            self.stream.wait_event(ev)
        with torch.cuda.stream(self.stream):
            self.works.append(dist.all_reduce(view, async_op=True))

    def finish(self):
        """Make the main stream wait for every outstanding collective (no host sync).  With ``measure`` set, two events bracket
        what the step could not hide: the end of backward on the main stream and the end of the last collective on the side
        stream (``exposed_ms()`` after a synchronisation)."""
        if not self.enabled:
            return
        while self._next < len(self.buckets):   # buckets holding parameters that got no gradient this step
            self._launch(self._next)
            self._next += 1
        if self.on_gpu and self.measure:
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record(torch.cuda.current_stream())
        if self.on_gpu:
            # c10d runs the collectives of one process group in issue order on ITS OWN stream; work.wait() makes the calling stream
            # wait for that work's end event.  One wait on the LAST work, taken by the side stream, covers all of them (the
            # per-work wait() loop did the same thing fifteen times, on the main stream); the main stream then waits for the side
            # stream's tail, and the measured pair brackets the real end of the last collective.
            # (whichever stream a collective was issued from, c10d ran it on its own stream: the side stream takes the wait)
            if self.works and dist.get_backend() == 'nccl':
                with torch.cuda.stream(self.stream):
                    self.works[-1].wait()
            else:                               # (gloo on device tensors -- the shared-GPU tests: every work is its own transfer)
                for w in self.works:
                    w.wait()
            if self.measure:
                e1 = torch.cuda.Event(enable_timing=True)
                e1.record(self.stream)
                self.exposed.append((e0, e1))
            torch.cuda.current_stream().wait_stream(self.stream)
        else:
            for w in self.works:
                w.wait()
        self._prune()

    def _prune(self):
        """Hook bookkeeping after a step (see __init__): drop the autograd hook of parameters the sink has served PRUNE_AFTER steps
        in a row; re-arm the hook of every parameter that did not report at all."""
        if not self.prune_hooks or self.capturing:
            return
        for p in self.arena.order:
            k = id(p)
            if k in self._via_sink:
                if k in self._hooks:
                    self._sunk_streak[k] += 1
                    if self._sunk_streak[k] >= self.PRUNE_AFTER:
                        self._hooks.pop(k).remove()
            elif k in self._hooks:
                self._sunk_streak[k] = 0
            elif k not in self._seen:
                self._arm(p)

    def exposed_ms(self):
        """Mean over the measured steps of max(0, end of the last gradient collective - end of backward): the all-reduce time the
        backward pass did not cover.  Call after torch.cuda.synchronize(); clears the record."""
        ev, self.exposed = self.exposed, []
        if not ev:
            return None
        return sum(max(0.0, a.elapsed_time(b)) for a, b in ev) / len(ev)


def step_lr(base_lr, it, epoch, warmup_iters=300, warmup_ratio=0.001, steps=(16, 22), gamma=0.1, step=None):
    """mmcv StepLrUpdaterHook + linear warm-up as configured in schedule_2x_bonai.py:5-10 (``step``: the config's key name)."""
    if step is not None:
        steps = step
    lr = base_lr * gamma ** sum(epoch >= s for s in steps)
    if it < warmup_iters:
        k = (1 - it / warmup_iters) * (1 - warmup_ratio)
        lr = lr * (1 - k)
    return lr


class Trainer:
    def __init__(self, model, lr=0.005, momentum=0.9, weight_decay=1e-4, max_norm=35.0, bucket_bytes=25 << 20,
                 loss_scale=1.0, graph_features=False):
        """loss_scale: the static scale of the reference's Fp16OptimizerHook (``fp16 = dict(loss_scale=512.)``,
        mmdet/core/fp16/hooks.py:64-96): the loss is multiplied before backward and the gradients are divided again inside
        the fused clip + SGD kernel (after the all-reduce, before the norm), exactly the hook's order.  bf16 activations have
        fp32's exponent range, so the scale is not needed for range here; it is honoured for runner parity."""
        self.model = model
        # graph_features: backbone + neck forward and backward as two hipGraphs, recorded at the third step and replayed from
        # then on (bonai_amd/graphs.py); fixed-size batches only.  Capture failures fall back to eager launches, loudly.
        self.graph_features = bool(graph_features)
        self._fgraphs = None
        self._unpack_stream = None
        self._steps_run = 0          # steps THIS trainer has run (a resumed trainer starts at iter > 0 with an empty prepack registry)
        self.lr, self.mu, self.wd, self.max_norm = lr, momentum, weight_decay, max_norm
        self.loss_scale = float(loss_scale)
        self.arena = FlatArena(model)
        self.reducer = BucketedAllReduce(self.arena, bucket_bytes)
        self.world = dist.get_world_size() if self.reducer.enabled else 1
        self.gnorm_sq = torch.zeros(1, dtype=torch.float32, device=self.arena.data.device)
        self.iter = 0
        self.prepack = K.PrepackRegistry()
        # kernels accumulate weight / BN gradients straight into the arena slots (bonai_amd.nn.GRAD_SINK); the callback
        # replaces the post-accumulate-grad hook for those parameters
        self._sink = self.reducer._sink if self.reducer.enabled else (lambda p: None)

    def _graph_step_setup(self, img):
        from .graphs import FeatureGraphs
        fg = self._fgraphs
        if fg is None:
            fg = self._fgraphs = FeatureGraphs(self)
        if fg.ready and fg.validate() is not None:
            import warnings
            warnings.warn(f'hipGraphs of backbone + neck dropped and recaptured: {fg.stale}', RuntimeWarning)
        if not fg.ready and fg.failed is None and self._steps_run >= 2 and self.prepack.order:
            self.reducer.capturing = True
            try:
                fg.capture(img)
            except Exception as e:      # noqa: BLE001 -- any capture refusal: stay eager, say so once
                import warnings
                fg.failed = f'{type(e).__name__}: {e}'
                warnings.warn(f'hipGraph capture of backbone + neck failed, the trainer stays on eager launches: {fg.failed}',
                              RuntimeWarning)
            finally:
                self.reducer.capturing = False
        if fg.ready:
            self.model.feat_provider = fg.provider

    def optimizer_state_dict(self):
        """The arena's SGD state in torch.optim.SGD.state_dict() layout (what mmcv's CheckpointHook stores under 'optimizer',
        apis/train.py:139-140 resumes from): parameters indexed in ``model.parameters()`` order (frozen ones included, as the
        reference builds its optimizer over all of them), one ``momentum_buffer`` per parameter that has been stepped."""
        a = self.arena
        state, idx = {}, {}
        for i, p in enumerate(self.model.parameters()):
            idx[id(p)] = i
        if self.iter > 0:
            for p in a.params:
                o = a.offsets[id(p)]
                state[idx[id(p)]] = dict(momentum_buffer=a.momentum[o:o + p.numel()].view(p.shape).detach().cpu().clone())
        group = dict(lr=self.lr, momentum=self.mu, dampening=0, weight_decay=self.wd, nesterov=False,
                     params=list(range(len(idx))))
        # sampler_calls: the RandomSampler kernel's draws are a function of (torch.initial_seed(), call count); a resumed run
        # continues the uninterrupted run's sequence only if the count travels with the optimizer state
        return dict(state=state, param_groups=[group], iter=self.iter, sampler_calls=int(K._SAMPLE_CALLS[0]))

    def load_optimizer_state(self, sd):
        """Inverse of optimizer_state_dict (also accepts a reference checkpoint's torch SGD state: state index i is the i-th
        entry of ``model.parameters()``, which tests/test_plugin_cpu.py pins to the reference model's registration order).
        Every buffer must have exactly its parameter's shape -- an index that points at another tensor of equal size would
        otherwise be written into the wrong momentum slot without a trace."""
        a = self.arena
        params = list(self.model.parameters())
        state = sd.get('state', {})
        groups = sd.get('param_groups')
        if groups and sum(len(g.get('params', ())) for g in groups) != len(params):
            raise RuntimeError(f"optimizer state covers {sum(len(g['params']) for g in groups)} parameters, the model has "
                               f'{len(params)}: it was written for a different model')
        todo = []
        for i, st in state.items():
            if not 0 <= int(i) < len(params):
                raise RuntimeError(f'optimizer state index {i} outside the model\'s {len(params)} parameters')
            p = params[int(i)]
            buf = st.get('momentum_buffer')
            if buf is None:
                continue
            if tuple(buf.shape) != tuple(p.shape):
                raise RuntimeError(f'optimizer state {i}: momentum_buffer {tuple(buf.shape)} does not fit parameter '
                                   f'{tuple(p.shape)} (parameter order differs from the checkpoint\'s)')
            if id(p) in a.offsets:
                todo.append((a.offsets[id(p)], p.numel(), buf))
        for o, n, buf in todo:                              # (nothing is written unless everything fits)
            a.momentum[o:o + n].copy_(buf.reshape(-1).to(a.momentum.device))
        self.iter = int(sd.get('iter', self.iter))
        if 'sampler_calls' in sd:
            K._SAMPLE_CALLS[0] = int(sd['sampler_calls'])

    def train_step(self, data, lr=None):
        """One full optimisation step: forward, losses, backward, gradient all-reduce, clip, SGD."""
        self.arena.grad.zero_()
        self.arena.rebind_grads()
        self.reducer.begin()
        for p in self.arena.params:
            p._loft_pending = 0
            p._loft_sunk = False
        from . import nn as F2
        prev_pp, prev_wpl = F2.PREPACK, K.WEIGHT_PLANES
        if self.arena.data.is_cuda and not DBG.no_prepack:
            F2.PREPACK = self.prepack
            K.WEIGHT_PLANES = self.prepack.wplanes         # (fp32 parity mode: the weight operands' planes, made by run() below)
            self.prepack.run(self.iter)                   # every trainable conv's BN fold + operand packing: one launch
        prev_hub = F2.HUB_ENABLED
        F2.HUB_ENABLED = self.arena.data.is_cuda and not DBG.no_feat_hub
        F2.JOIN = {} if (self.arena.data.is_cuda and not DBG.no_grad_join) else None      # lives through forward AND backward
        F2.PAIR_G.clear()                                 # (pair fusion's backward hand-over: nothing survives a step)
        if self.graph_features and self.arena.data.is_cuda and K.PROFILE is None and F2.PREPACK is not None \
                and hasattr(self.model, 'extract_feat'):
            self._graph_step_setup(data['img'])
        try:
            out = self.model.train_step(data)
        except BaseException:
            F2.JOIN = None
            K.WEIGHT_PLANES = prev_wpl
            raise
        finally:
            F2.PREPACK = prev_pp
            F2.HUB_ENABLED = prev_hub
            if self.graph_features:
                self.model.feat_provider = None
        prev, F2.GRAD_SINK = F2.GRAD_SINK, (self._sink if self.arena.data.is_cuda and not DBG.no_grad_sink else None)
        if self.arena.data.is_cuda and not DBG.no_zero_pool:
            K.zero_pool_begin(self.arena.data.device)     # one memset for all the backward's accumulation buffers
        if F2.GRAD_SINK is not None and not DBG.no_unpack_queue:
            # multi-GPU: smaller bursts, so the gradient buckets become ready (and their all-reduce starts) earlier in backward
            # (round 6, measured and NOT the default: the batched unpack launches on a stream of their own -- kernels.UnpackQueue,
            #  DBG.unpack_stream -- joined below: neutral on the plain step, +1.0 .. 1.3 ms on the forced-reducer leg)
            ustream = None
            if DBG.unpack_stream and not DBG.no_side_stream:
                if getattr(self, '_unpack_stream', None) is None:
                    self._unpack_stream = torch.cuda.Stream()
                ustream = self._unpack_stream
            lim = int(os.environ.get('LOFT_UNPACK_LIMIT', '0')) or (24 if self.reducer.enabled else 48)
            note = self.reducer.note_queued if (self.reducer.enabled and os.environ.get('LOFT_NO_NOTE_FLUSH') != '1') else None
            F2.UNPACK_Q = K.UnpackQueue(limit=lim, note=note, stream=ustream)
            if not DBG.no_side_stream and not DBG.no_wgrad_stream:
                if getattr(self, '_wgrad_stream', None) is None:
                    self._wgrad_stream = torch.cuda.Stream()
                F2.WGRAD_STREAM = self._wgrad_stream
        try:
            (out['loss'] if self.loss_scale == 1.0 else out['loss'] * self.loss_scale).backward()
            if F2.UNPACK_Q is not None:
                F2.UNPACK_Q.flush()
        finally:
            if F2.UNPACK_Q is not None:
                F2.UNPACK_Q.join()              # the main stream waits for the unpack stream before the norm / optimizer kernels
            F2.UNPACK_Q = None
            F2.WGRAD_STREAM = None
            F2.GRAD_SINK = prev
            F2.HUB = None
            F2.JOIN = None
            K.WEIGHT_PLANES = prev_wpl          # (the backward's data-gradient launches read the weight planes too)
            K.zero_pool_end()
        self.reducer.finish()
        self.gnorm_sq.zero_()
        K.sumsq_(self.arena.grad, self.gnorm_sq)
        K.sgd_momentum_(self.arena.data, self.arena.grad, self.arena.momentum, self.gnorm_sq, self.max_norm,
                        self.lr if lr is None else lr, self.mu, self.wd, grad_scale=1.0 / (self.world * self.loss_scale))
        self.iter += 1
        self._steps_run += 1
        return out
