"""CPU: config loader, registry, model construction / parameter naming (no kernels launched)."""
import os

import pytest
import torch

from bonai_amd.config import Config
from bonai_amd.registry import Registry, build_from_cfg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = os.path.join(ROOT, 'configs', 'loft_foa', 'loft_foa_r50_fpn_2x_bonai.py')


def test_registry_both_decorator_forms():
    R = Registry('r')

    @R.register_module()
    class A:
        def __init__(self, x=1):
            self.x = x

    @R.register_module
    class Bb:
        pass
    assert R.get('A') is A and R.get('Bb') is Bb
    assert build_from_cfg(dict(type='A', x=3), R).x == 3
    assert build_from_cfg(dict(type='A'), R, dict(x=5)).x == 5


def test_config_bases_and_access():
    cfg = Config.fromfile(CFG)
    assert cfg.model.type == 'LOFT' and cfg.model.roi_head.offset_head.num_convs == 10
    assert cfg.train_cfg.rcnn.sampler.num == 1024 and cfg.train_cfg.rpn.get('allowed_border') == -1
    assert cfg.optimizer.lr == 0.005 and cfg.total_epochs == 24
    assert cfg.model.roi_head.offset_head.loss_offset.loss_weight == 16.0
    cfg.merge_from_dict({'optimizer.lr': 0.01})
    assert cfg.optimizer.lr == 0.01


def test_build_detector_and_state_dict_names():
    from bonai_amd.loft import build_detector
    cfg = Config.fromfile(CFG)
    m = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg)
    sd = m.state_dict()
    total = sum(p.numel() for p in m.parameters())
    trainable = sum(p.numel() for p in m.parameters() if p.requires_grad)
    assert abs(total / 1e6 - 81.472088) < 1e-6 and abs(trainable / 1e6 - 81.246744) < 1e-6   # reference counts
    for k in ['backbone.conv1.weight', 'backbone.bn1.running_var', 'backbone.layer1.0.downsample.0.weight',
              'backbone.layer4.2.bn3.weight', 'neck.lateral_convs.3.conv.weight', 'neck.fpn_convs.0.conv.bias',
              'rpn_head.rpn_conv.weight', 'rpn_head.rpn_cls.bias', 'rpn_head.rpn_reg.weight',
              'roi_head.bbox_head.shared_fcs.0.weight', 'roi_head.bbox_head.fc_cls.weight', 'roi_head.bbox_head.fc_reg.bias',
              'roi_head.mask_head.convs.3.conv.weight', 'roi_head.mask_head.upsample.weight',
              'roi_head.mask_head.conv_logits.weight', 'roi_head.offset_head.expand_convs.3.9.weight',
              'roi_head.offset_head.fcs.0.weight', 'roi_head.offset_head.fcs.1.bias', 'roi_head.offset_head.fc_offset.weight']:
        assert k in sd, k
    assert sd['roi_head.offset_head.fcs.0.weight'].shape == (1024, 12544)
    assert sd['roi_head.mask_head.upsample.weight'].shape == (256, 256, 2, 2)
    assert not m.backbone.layer1[0].conv1.weight.requires_grad and m.backbone.layer2[0].conv1.weight.requires_grad


def test_checkpoint_formats(tmp_path):
    """Reference checkpoint layouts load by key: {'meta','state_dict','optimizer'} with 'module.' prefixes (MMDDP), a bare
    state_dict, and a torchvision-keyed ResNet-50 (conv1.weight, layer1.0..., fc.*) into model.backbone."""
    import torch
    from bonai_amd.checkpoint import load_checkpoint, save_checkpoint
    from bonai_amd.config import Config
    from bonai_amd.loft import build_detector
    cfg = Config.fromfile(os.path.join(ROOT, 'configs', 'loft_foa', 'loft_foa_r50_fpn_2x_bonai.py'))
    a = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg)
    b = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg)
    with torch.no_grad():
        for p in a.parameters():
            p.add_(0.5)
    ck = dict(meta=dict(epoch=3), state_dict={'module.' + k: v for k, v in a.state_dict().items()}, optimizer=dict(state={}))
    path = str(tmp_path / 'epoch_3.pth')
    torch.save(ck, path)
    out = load_checkpoint(b, path, strict=True)
    assert out['meta']['epoch'] == 3
    for (k, v), (_, w) in zip(a.state_dict().items(), b.state_dict().items()):
        assert torch.equal(v, w), k
    # torchvision-style backbone checkpoint (plus the classifier the detector does not have)
    tv = {k[len('backbone.'):]: v + 1 for k, v in a.state_dict().items() if k.startswith('backbone.') and v.is_floating_point()}
    tv['fc.weight'] = torch.zeros(1000, 2048)
    tv['fc.bias'] = torch.zeros(1000)
    load_checkpoint(b, tv)
    assert torch.equal(b.state_dict()['backbone.layer3.2.conv2.weight'], a.state_dict()['backbone.layer3.2.conv2.weight'] + 1)
    assert torch.equal(b.state_dict()['rpn_head.rpn_conv.weight'], a.state_dict()['rpn_head.rpn_conv.weight'])
    # backbone.init_weights(pretrained=path) path (resnet.py:591-600)
    p2 = str(tmp_path / 'r50.pth')
    torch.save(tv, p2)
    b.backbone.init_weights(pretrained=p2)
    # round trip through save_checkpoint
    p3 = save_checkpoint(a, str(tmp_path / 'latest.pth'), meta=dict(iter=7))
    c = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg)
    assert load_checkpoint(c, p3, strict=True)['meta']['iter'] == 7
    with pytest.raises(RuntimeError):
        load_checkpoint(c, {'state_dict': {'backbone.conv1.weight': torch.zeros(1, 1)}})
