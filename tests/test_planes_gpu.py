"""The fp32 parity mode on OPERAND PLANES (round 5; include/loft_hip.h loft_conv_tap_planes / loft_conv_wgrad_planes /
loft_split_planes_f32): every fp32 operand becomes two binary16 planes under a power-of-two scale (or three bfloat16 planes) and the
contraction runs on the software-pipelined 16-bit stream kernels with fp32 accumulation.  Checked here against fp64 convolutions
at fp32-rounding tolerances, on shapes that reach every tile shape of the stream kernel (256 / 128 / 64 pixels x 256 / 128 couts,
pixel-major RoI maps, 1x1 layers, strided data gradients, grouped launches) and both weight-gradient kernels.
Reference lines: the reference evaluates these layers in fp32 on the CPU (detectors/base.py:159-173; resnet.py:266-298, fpn.py:170-199,
offset_head_expand_feature.py:134-161)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

MODES = ['planes_f16', 'planes_bf16']


@pytest.fixture(params=MODES)
def planes_mode(request):
    from bonai_amd import kernels as K
    prev = K.F32_CONTRACT
    K.F32_CONTRACT = {'planes_f16': K.F32_PLANES_F16, 'planes_bf16': K.F32_PLANES_BF16}[request.param]
    yield request.param
    K.F32_CONTRACT = prev


def _cl(t):
    return t.cuda().contiguous(memory_format=torch.channels_last)


def _conv64(x, w, b, s, p):
    """fp64 convolution of fp32-valued operands (on the device when the backend has an fp64 convolution, else on the host)."""
    try:
        return F.conv2d(x.cuda().double(), w.cuda().double(), None if b is None else b.cuda().double(), s, p).cpu()
    except RuntimeError:
        return F.conv2d(x.double(), w.double(), None if b is None else b.double(), s, p)


def test_split_planes_reconstructs_the_tensor(planes_mode):
    """plane_0 + plane_1 (+ plane_2) = x * scale to 2^-22 (binary16, two planes) / 2^-24 (bfloat16, three), the scale a power of two
    that keeps the high plane finite; tensors far outside binary16's range, zeros, and a tensor with one huge outlier."""
    from bonai_amd import kernels as K
    dt = torch.float16 if planes_mode == 'planes_f16' else torch.bfloat16
    torch.manual_seed(0)
    for scale in (1.0, 1e-7, 3e5):
        x = (torch.randn(4096 * 8) * scale).cuda()
        x[5] = 0.0
        x[17] = 900.0 * scale                       # an outlier 2^10 above the bulk
        planes, amax = K.split_planes(x, dt)
        assert planes.dtype == dt and planes.shape == (2 if dt == torch.float16 else 3, x.numel())
        rec = planes.double().sum(0)
        assert torch.isfinite(rec).all()
        if amax is not None:
            assert amax[0].item() == x.abs().max().item()
            e = torch.floor(torch.log2(amax[0])).item()
            rec = rec * 2.0 ** (e - 14)
        tol = (2.0 ** -21 if dt == torch.float16 else 2.0 ** -23)
        err = (rec - x.double()).abs()
        # elements within 2^-13 of the absmax keep full relative precision; smaller ones are exact to 2^-36 of the absmax
        assert (err <= tol * x.double().abs() + 2.0 ** -36 * x.abs().max().item()).all(), (scale, err.max().item())
    z = torch.zeros(64).cuda()
    planes, amax = K.split_planes(z, dt)
    assert (planes == 0).all()


# (B, cin, cout, k, s, p, hw, groups): the tile shape loft_conv_tap_planes picks is noted per case
CASES = [
    (2, 64, 256, 3, 1, 1, 40, 1),      # 64 px x 128 cout (few tiles)
    (3, 64, 256, 3, 1, 1, 128, 1),     # 256 px x 256 cout
    (3, 64, 512, 1, 1, 0, 64, 1),      # 128 px x 256 cout, pointwise
    (6, 128, 512, 1, 1, 0, 32, 1),     # 64 px x 256 cout, pointwise
    (3, 64, 128, 3, 1, 1, 128, 1),     # 256 px x 128 cout
    (256, 64, 256, 3, 1, 1, 7, 1),     # pixel-major RoI map (7 x 7)
    (256, 64, 256, 3, 1, 1, 7, 2),     # ... grouped (the FOA branches' form)
    (2, 256, 128, 1, 2, 0, 17, 1),     # strided 1x1, odd map
    (2, 64, 256, 3, 2, 1, 33, 1),      # strided 3x3
]


@pytest.mark.parametrize('B,cin,cout,k,s,p,hw,groups', CASES)
def test_conv_tap_planes_forward_vs_fp64(B, cin, cout, k, s, p, hw, groups, planes_mode):
    from bonai_amd import kernels as K
    torch.manual_seed(B + cin + cout + k + hw)
    x = torch.randn(groups * B, cin, hw, hw) * 3.0
    w = torch.randn(groups, cout, cin, k, k) * 0.05
    b = torch.randn(groups, cout)
    oh = (hw + 2 * p - k) // s + 1
    res = torch.randn(groups * B, cout, oh, oh)
    ref = torch.cat([F.relu(_conv64(x[i * B:(i + 1) * B], w[i], b[i], s, p) + res[i * B:(i + 1) * B].double()) for i in range(groups)])
    wp = torch.stack([K.pack_w_fwd(w[i].cuda(), torch.float32) for i in range(groups)])
    n0 = dict(K.PLANES_STATS)
    y = K.conv2d_fwd(_cl(x), wp, b.cuda(), k, k, s, p, relu=True, residual=_cl(res), out_dtype=torch.float32, groups=groups)
    assert K.PLANES_STATS['planes'] == n0['planes'] + 1 and K.PLANES_STATS['fallback'] == n0['fallback'], 'took the fallback kernel'
    err = (y.cpu().double() - ref).abs().max().item() / max(1.0, ref.abs().max().item())
    print(f'planes conv ({planes_mode}) {B}x{cin}->{cout} k{k}s{s} {hw}^2 g{groups}: max err / scale = {err:.2e}')
    assert err < 1e-5


@pytest.mark.parametrize('B,cin,cout,k,s,p,hw', [(2, 256, 64, 3, 1, 1, 40), (3, 256, 64, 3, 2, 1, 64), (2, 512, 64, 1, 1, 0, 48)])
def test_conv_tap_planes_data_gradient_vs_fp64(B, cin, cout, k, s, p, hw, planes_mode):
    """The data gradient of a cin -> cout convolution is a planes launch over Cout = cin output channels (stride 2: four parity
    classes with offset outputs and the ReLU-backward mask of the input in the epilogue)."""
    from bonai_amd import kernels as K
    torch.manual_seed(B + cin + cout + k)
    x = torch.randn(B, cin, hw, hw, dtype=torch.float64, requires_grad=True)
    w = torch.randn(cout, cin, k, k, dtype=torch.float64) * 0.05
    oh = (hw + 2 * p - k) // s + 1
    g = torch.randn(B, cout, oh, oh, dtype=torch.float64) * 1e-4          # (gradient-sized values: the scale has to follow them)
    F.conv2d(F.relu(x), w, None, s, p).backward(g)
    wpt = w.float().cuda().permute(2, 3, 1, 0).reshape(k * k, cin, cout).contiguous()[None]
    n0 = dict(K.PLANES_STATS)
    gx = K.conv2d_dgrad(_cl(g.float()), wpt, (hw, hw), k, k, s, p, mask=_cl(x.detach().float()), out_dtype=torch.float32)
    assert K.PLANES_STATS['fallback'] == n0['fallback'] and K.PLANES_STATS['planes'] > n0['planes']
    err = (gx.cpu().double() - x.grad).abs().max().item() / x.grad.abs().max().item()
    print(f'planes dgrad ({planes_mode}) {cin}<-{cout} k{k}s{s}: max err / scale = {err:.2e}')
    assert err < 2e-5


@pytest.mark.parametrize('B,cin,cout,k,s,p,hw,groups', [
    (2, 128, 128, 3, 1, 1, 40, 1),      # 128 x 128 ring kernel, same-size taps
    (2, 128, 384, 1, 1, 0, 33, 1),      # ring kernel, dense
    (4, 256, 256, 3, 1, 1, 96, 1),      # 256 x 256 stream kernel
    (2, 256, 128, 3, 2, 1, 33, 1),      # strided: generic row decode
    (256, 256, 256, 3, 1, 1, 7, 1),     # RoI maps: position-major reduction
    (256, 256, 256, 3, 1, 1, 7, 2),     # ... grouped
])
def test_conv_wgrad_planes_vs_fp64(B, cin, cout, k, s, p, hw, groups, planes_mode):
    from bonai_amd import kernels as K
    torch.manual_seed(B + cin + cout + k + groups)
    x = torch.randn(groups * B, cin, hw, hw) * 2.0
    oh = (hw + 2 * p - k) // s + 1
    g = torch.randn(groups * B, cout, oh, oh) * 1e-3
    want = []
    for i in range(groups):
        for dev in ('cuda', 'cpu'):
            try:
                w = torch.zeros(cout, cin, k, k, dtype=torch.float64, device=dev, requires_grad=True)
                F.conv2d(x[i * B:(i + 1) * B].to(dev).double(), w, None, s, p).backward(g[i * B:(i + 1) * B].to(dev).double())
                want.append(w.grad.cpu())
                break
            except RuntimeError:
                if dev == 'cpu':
                    raise
    n0 = dict(K.PLANES_STATS)
    dwp, db = K.conv2d_wgrad(_cl(g), _cl(x), k, k, s, p, groups=groups, with_bias=True)
    assert K.PLANES_STATS['planes'] == n0['planes'] + 1 and K.PLANES_STATS['fallback'] == n0['fallback'], 'took the fallback kernel'
    assert K.PLANES_STATS.get('db_fused', 0) == n0.get('db_fused', 0) + 1, 'the bias gradient did not ride in the plane launch'
    assert K.PLANES_STATS.get('db_fused', 0) == n0.get('db_fused', 0) + 1, 'the bias gradient did not ride in the plane launch'
    for i in range(groups):
        dw = K.unpack_dw(dwp[i], (cout, cin, k, k)).cpu().double()
        err = (dw - want[i]).abs().max().item() / want[i].abs().max().item()
        print(f'planes wgrad ({planes_mode}) {cin}->{cout} k{k}s{s} {hw}^2 g{groups}[{i}]: max err / scale = {err:.2e}')
        assert err < 2e-5, i
    want_b = g.double().view(groups, B, cout, -1).sum(dim=(1, 3))
    assert (db[:, :cout].cpu().double() - want_b).abs().max().item() <= 2e-5 * want_b.abs().max().item()
    # the separate column-sum pass (round 5's form) agrees with the fused one
    K.PLANES_DB_FUSED = False
    try:
        _, db2 = K.conv2d_wgrad(_cl(g), _cl(x), k, k, s, p, groups=groups, with_bias=True)
    finally:
        K.PLANES_DB_FUSED = True
    assert (db2[:, :cout].cpu().double() - want_b).abs().max().item() <= 2e-5 * want_b.abs().max().item()
    assert (db2 - db).abs().max().item() <= 2e-5 * want_b.abs().max().item()


@pytest.mark.parametrize('B,cin,cout,k,s,p,hw,groups', [
    (3, 64, 256, 3, 1, 1, 128, 1),      # dense map, 256-pixel tiles
    (2, 256, 512, 1, 1, 0, 100, 1),     # two channel tiles, a row count that leaves a partial tile; data gradient over 256 channels
    (300, 64, 256, 3, 1, 1, 7, 2),      # pixel-major RoI maps, grouped, padding rows
])
def test_staged_fp32_epilogue_is_bit_identical_to_the_direct_one(B, cin, cout, k, s, p, hw, groups, planes_mode):
    """Round 6: the 256 x 256 plane launches collect their fp32 tile in LDS and store whole rows (pipe_epilogue_f32_staged);
    loft_conv_stream_form(form | 4) keeps the direct epilogue.  Same operations in the same order: forward with bias + residual +
    ReLU and the data gradient with the ReLU-backward mask are torch.equal, and so is the absmax either launch leaves."""
    from bonai_amd import kernels as K
    from bonai_amd import lib as L
    torch.manual_seed(B + cin + cout)
    x = _cl(torch.randn(groups * B, cin, hw, hw) * 3.0)
    w = torch.randn(groups, cout, cin, k, k) * 0.05
    b = torch.randn(groups, cout).cuda()
    oh = (hw + 2 * p - k) // s + 1
    res = _cl(torch.randn(groups * B, cout, oh, oh))
    wp = torch.stack([K.pack_w_fwd(w[i].cuda(), torch.float32) for i in range(groups)])
    wpt = torch.stack([w[i].cuda().permute(2, 3, 1, 0).reshape(k * k, cin, cout).contiguous() for i in range(groups)])
    gfix = _cl(torch.randn(groups * B, cout, oh, oh) * 1e-3)
    libs = [L.load_for(torch.bfloat16), L.load_for(torch.float16)]
    outs = []
    prev = [lib.loft_conv_stream_form(-1) for lib in libs]
    try:
        for form in (2, 6):
            for lib in libs:
                lib.loft_conv_stream_form(form)
            y = K.conv2d_fwd(x, wp, b, k, k, s, p, relu=True, residual=res, out_dtype=torch.float32, groups=groups)
            o = [y, None if K._known_amax(y) is None else K._known_amax(y).clone()]
            if cin % 128 == 0:
                o.append(K.conv2d_dgrad(gfix, wpt, (hw, hw), k, k, s, p, mask=x, out_dtype=torch.float32, groups=groups))
            outs.append(o)
    finally:
        for lib, f in zip(libs, prev):
            lib.loft_conv_stream_form(f)
    torch.cuda.synchronize()
    assert len(outs[0]) == len(outs[1])
    for a_, b_ in zip(outs[0], outs[1]):
        if a_ is None or b_ is None:
            assert a_ is None and b_ is None
        else:
            assert torch.equal(a_, b_)


def test_planes_modes_fall_back_on_unsupported_shapes(planes_mode):
    """Channel counts the stream kernels do not serve take the SPLIT6 kernels, with the same answer."""
    from bonai_amd import kernels as K
    torch.manual_seed(3)
    x = torch.randn(2, 96, 9, 9)
    w = torch.randn(36, 96, 3, 3) * 0.05
    ref = _conv64(x, w, None, 1, 1)
    n0 = dict(K.PLANES_STATS)
    y = K.conv2d_fwd(_cl(x), K.pack_w_fwd(w.cuda(), torch.float32)[None], None, 3, 3, 1, 1, out_dtype=torch.float32)
    assert K.PLANES_STATS['fallback'] == n0['fallback'] + 1
    assert (y.cpu().double() - ref).abs().max().item() < 1e-5 * ref.abs().max().item()


def test_producer_absmax_note_is_exact_and_dropped_by_in_place_kernels():
    """A plane launch's epilogue leaves max |out| with its output (kernels._known_amax) so that the next layer's split skips its
    absmax pass: the note equals the tensor's absmax, a chain of two layers gives the same bits with and without it, and both a
    raw in-place kernel (FPN's upsample-add) and an aten in-place op invalidate it."""
    from bonai_amd import kernels as K
    prev = K.F32_CONTRACT
    K.F32_CONTRACT = K.F32_PLANES_F16
    try:
        torch.manual_seed(11)
        x = _cl(torch.randn(2, 64, 32, 32) * 2.0)
        w1 = K.pack_w_fwd((torch.randn(256, 64, 3, 3) * 0.05).cuda(), torch.float32)[None]
        w2 = K.pack_w_fwd((torch.randn(256, 256, 3, 3) * 0.05).cuda(), torch.float32)[None]
        outs = []
        for flag in (True, False):
            K.AMAX_FROM_PRODUCER = flag
            y = K.conv2d_fwd(x, w1, None, 3, 3, 1, 1, relu=True, out_dtype=torch.float32)
            note = K._known_amax(y)
            assert (note is not None) == flag
            if flag:
                assert note[0].item() == y.abs().max().item()
            outs.append(K.conv2d_fwd(y, w2, None, 3, 3, 1, 1, out_dtype=torch.float32))
        assert torch.equal(outs[0], outs[1])
        K.AMAX_FROM_PRODUCER = True
        y = K.conv2d_fwd(x, w1, None, 3, 3, 1, 1, relu=True, out_dtype=torch.float32)
        coarse = _cl(torch.randn(2, 256, 16, 16) * 50.0)
        K.upsample2x_add_(y, coarse)
        assert K._known_amax(y) is None
        y2 = K.conv2d_fwd(x, w1, None, 3, 3, 1, 1, relu=True, out_dtype=torch.float32)
        y2.mul_(100.0)
        assert K._known_amax(y2) is None
        z = K.conv2d_fwd(y2, w2, None, 3, 3, 1, 1, out_dtype=torch.float32)
        assert torch.isfinite(z).all()
    finally:
        K.F32_CONTRACT = prev
        K.AMAX_FROM_PRODUCER = True


def test_batched_f32_prepack_equals_the_per_conv_launches():
    """Round 6: under the trainer the fp32 parity mode folds BN, packs and splits the weights of EVERY conv in two launches per step
    (PrepackRegistry.request_f32: loft_fold_f32_multi + loft_split_planes_f32_multi).  Packings, biases, planes and the absmax slot
    must be bit-identical to loft_fold_pack + loft_absmax_split_planes_f32 per conv -- first step (registered on the fly), second
    step (the batched launches over all records, after the weights changed), a grouped record (one scale for the G members) and a
    record the plane kernels do not serve (no planes, packings only)."""
    from bonai_amd import kernels as K
    torch.manual_seed(11)
    dev = 'cuda'

    def conv(co, ci, k, bn=True):
        w = torch.nn.Parameter(torch.randn(co, ci, k, k, device=dev) * 0.05)
        stats = (torch.rand(co, device=dev) + 0.5, torch.randn(co, device=dev) * 0.1, torch.randn(co, device=dev) * 0.1,
                 torch.rand(co, device=dev) + 0.5) if bn else None            # gamma, beta, mean, var
        return w, stats

    recs = [((conv(256, 128, 3),), 256, 128, True), ((conv(128, 256, 1),), 128, 256, True),
            (tuple(conv(256, 256, 3, bn=False) for _ in range(4)), 256, 256, True),      # FOA-style grouped launch
            ((conv(64, 64, 3),), 64, 64, False)]                                          # not served by the plane kernels
    reg = K.PrepackRegistry()
    prev = K.WEIGHT_PLANES
    K.WEIGHT_PLANES = reg.wplanes
    try:
        for step in range(2):
            reg.run(step)
            for members, cop, cip, dgrad in recs:
                ws = tuple(m[0] for m in members)
                bn = members[0][1]
                wp, wpt, bias = reg.request_f32(ws, (None,) * len(ws), bn, 1e-5, cop, cip, dgrad)
                for g, (w, _) in enumerate(members):
                    rwp, rwpt, rb = K.fold_pack(w.detach(), None, bn, 1e-5, want_dgrad=dgrad, dtype=torch.float32, cout_pad=cop, cin_pad=cip)
                    assert torch.equal(wp[g], rwp) and torch.equal(bias[g], rb), (step, g)
                    assert (wpt is None) == (rwpt is None) and (wpt is None or torch.equal(wpt[g], rwpt))
                for t in (wp, wpt):
                    if t is None:
                        continue
                    prev_reg, K.WEIGHT_PLANES = K.WEIGHT_PLANES, None
                    want_p, want_a = K.split_planes(t, torch.float16)                 # the per-tensor launches
                    K.WEIGHT_PLANES = prev_reg
                    got = reg.wplanes.get((t.data_ptr(), t.numel()))
                    served = cip % 64 == 0 and cop % 128 == 0 if t is wp else cop % 64 == 0 and cip % 128 == 0
                    assert (got is not None) == served, (step, tuple(t.shape))
                    if got is not None:
                        gp, ga = K.split_planes(t, torch.float16)                      # (what a conv launch gets: the registry's)
                        assert gp.data_ptr() == got[0].data_ptr()
                        assert torch.equal(gp, want_p) and float(ga[0]) == float(want_a[0]), (step, tuple(t.shape))
            with torch.no_grad():                                                      # "SGD": the next step's batched launches see new weights
                for members, *_ in recs:
                    for w, _ in members:
                        w.mul_(1.5).add_(0.01)
    finally:
        K.WEIGHT_PLANES = prev
