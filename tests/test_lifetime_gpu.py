"""Object lifetime on the device: nothing the frozen layers cache may outlive the model it was computed from.

Round-2 regression: the packed operands of frozen convs were cached in a process-global dict
keyed by (data_ptr, _version, shape).  A model built after another one was freed gets the freed model's addresses from the
caching allocator (same shapes, same version counts) and silently ran with the PREVIOUS model's stem / layer1 weights."""
import gc
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, 'tests', 'golden')


def _cfg():
    from bonai_amd.config import Config
    return Config.fromfile(os.path.join(ROOT, 'configs', 'loft_foa', 'loft_foa_r50_fpn_2x_bonai.py'))


def _random_model(seed):
    from bonai_amd.loft import build_detector
    cfg = _cfg()
    torch.manual_seed(seed)
    return build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg).cuda().train()


def _synth_model():
    from bonai_amd.loft import build_detector
    from oracle.synth_weights import synth_tensor
    cfg = _cfg()
    m = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg)
    m.load_state_dict({k: synth_tensor(k, v.shape) for k, v in m.state_dict().items()})
    return m.cuda().train()


def _check_features(m, gd, data):
    with torch.no_grad():
        feats = m.extract_feat(data['img'])
    for i, f in enumerate(feats):
        want = torch.from_numpy(gd[f'feat_{i}_crop'])
        got = f[:, :8, :6, :6].float().cpu()
        scale = float(gd[f'feat_{i}_absmean'])
        assert (got - want).abs().max().item() < 0.03 * scale * 4, (i, (got - want).abs().max().item(), scale)


@pytest.mark.parametrize('release', ['keep_blocks', 'empty_cache'])
def test_frozen_packs_do_not_outlive_their_model(release):
    """random-init model A runs and dies; synth-weight model B takes its place in memory and must reproduce the
    reference-made fixture (which a B running on A's packed stem / layer1 weights cannot)."""
    from bonai_amd.synth import make_batch
    gd = np.load(os.path.join(GOLD, 'e2e_256.npz'))
    size, batch, num_gt = [int(v) for v in gd['meta']]
    data = make_batch(batch, size, num_gt, device='cuda')
    for seed in (0, 1):
        a = _random_model(seed)
        with torch.no_grad():
            a.extract_feat(data['img'])
        ptrs = {n: p.data_ptr() for n, p in a.named_parameters()}
        del a
        gc.collect()
        if release == 'empty_cache':
            torch.cuda.empty_cache()
        b = _synth_model()
        same = sum(ptrs[n] == p.data_ptr() for n, p in b.named_parameters())
        print(f'[{release} seed {seed}] parameters of B at the addresses of A: {same} of {len(ptrs)}')
        _check_features(b, gd, data)
        del b
        gc.collect()


def test_frozen_packs_follow_in_place_updates():
    """load_state_dict / copy_ into the frozen parameters of a LIVE model must be seen by the next forward."""
    from bonai_amd.synth import make_batch
    from oracle.synth_weights import synth_tensor
    gd = np.load(os.path.join(GOLD, 'e2e_256.npz'))
    size, batch, num_gt = [int(v) for v in gd['meta']]
    data = make_batch(batch, size, num_gt, device='cuda')
    m = _random_model(3)
    with torch.no_grad():
        m.extract_feat(data['img'])                       # packs of the random weights are cached now
    m.load_state_dict({k: synth_tensor(k, v.shape) for k, v in m.state_dict().items()})
    _check_features(m, gd, data)
    with torch.no_grad():                                  # .data re-assignment (what FlatArena / .cuda() do)
        for p in m.backbone.parameters():
            p.data = p.data.clone()
    _check_features(m, gd, data)
