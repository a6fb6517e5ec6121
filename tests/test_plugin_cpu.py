"""CPU, build container only: the reference-side plugin (bonai_amd/mmdet_plugin.py) EXECUTED against the
reference tree -- the reference's own ``mmdet.models.build_detector`` (mmdet/models/builder.py:65-67) on the reference's own
config file must come back built from bonai_amd classes, with the state_dict surface of the reference's model.

Needs /root/reference (absent on the GPU box: skipped there) and the stand-in mmcv of oracle/ref_harness/mmcv_stub.py."""
import os
import sys
import warnings

import pytest

REF = '/root/reference'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, 'mmdet')), reason='reference tree not present on this box')


def test_reference_build_detector_resolves_to_native_classes():
    sys.path.insert(0, os.path.join(ROOT, 'oracle', 'ref_harness'))
    import mmcv_stub
    mmcv_stub.install()
    from mmdet.models import build_detector          # the REFERENCE's builder
    from bonai_amd.config import Config
    import bonai_amd.loft as L
    cfg = Config.fromfile(os.path.join(REF, 'configs', 'loft_foa', 'loft_foa_r50_fpn_2x_bonai.py'))   # the REFERENCE's config
    cfg.model['pretrained'] = None
    ref_model = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg)
    assert type(ref_model).__module__.startswith('mmdet.')
    ref_keys = {k: tuple(v.shape) for k, v in ref_model.state_dict().items()}

    import bonai_amd.mmdet_plugin as plugin
    done = plugin.register()
    for name in ('LOFT', 'ResNet', 'FPN', 'RPNHead', 'LoftRoIHead', 'SingleRoIExtractor', 'Shared2FCBBoxHead', 'FCNMaskHead',
                 'OffsetHeadExpandFeature', 'OffsetHead', 'CrossEntropyLoss', 'L1Loss', 'SmoothL1Loss', 'MaxIoUAssigner',
                 'RandomSampler', 'DeltaXYWHBBoxCoder', 'DeltaXYOffsetCoder', 'AnchorGenerator', 'BboxOverlaps2D', 'HRNet', 'HRFPN'):
        assert any(name in v for v in done.values()), name
    model = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg)      # same call, same config
    assert type(model) is L.DETECTORS.get('LOFT')
    native = [m for m in model.modules() if type(m).__module__.startswith('bonai_amd.')]
    foreign = [type(m).__name__ for m in model.modules()
               if not type(m).__module__.startswith(('bonai_amd.', 'torch.'))]
    assert not foreign and len(native) > 50, foreign[:5]
    keys = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    assert keys == ref_keys                           # reference checkpoints load by key, shapes included
    # ... and in the reference's REGISTRATION ORDER: torch.optim.SGD's state_dict indexes parameters by their position in
    # model.parameters(), so Trainer.load_optimizer_state can take a reference checkpoint's 'optimizer' entry only if the
    # i-th parameter here is the i-th parameter there (engine.py)
    ref_order = [(n, tuple(p.shape), p.requires_grad) for n, p in ref_model.named_parameters()]
    own_order = [(n, tuple(p.shape), p.requires_grad) for n, p in model.named_parameters()]
    assert own_order == ref_order
    # sub-builders of the reference resolve too
    from mmdet.models.builder import build_backbone, build_head
    from mmdet.core.bbox.builder import build_bbox_coder
    assert type(build_backbone(dict(cfg.model.backbone))).__module__.startswith('bonai_amd.')
    assert type(build_bbox_coder(dict(type='DeltaXYOffsetCoder'))).__module__.startswith('bonai_amd.')
    assert type(build_head(dict(type='OffsetHead', num_convs=1))).__module__.startswith('bonai_amd.')


def test_pretrained_uri_is_resolved_or_reported(tmp_path, monkeypatch):
    """a model-zoo URI must never silently become random init."""
    import torch
    from bonai_amd.loft.detector import resolve_pretrained
    monkeypatch.delenv('LOFT_PRETRAINED_DIR', raising=False)
    monkeypatch.delenv('LOFT_PRETRAINED_STRICT', raising=False)
    with pytest.warns(RuntimeWarning, match='RANDOM initialisation'):
        assert resolve_pretrained('torchvision://resnet50') is None
    monkeypatch.setenv('LOFT_PRETRAINED_STRICT', '1')
    with pytest.raises(FileNotFoundError):
        resolve_pretrained('torchvision://resnet50')
    monkeypatch.delenv('LOFT_PRETRAINED_STRICT')
    f = tmp_path / 'resnet50-19c8e357.pth'
    torch.save({}, f)
    monkeypatch.setenv('LOFT_PRETRAINED_DIR', str(tmp_path))
    with warnings.catch_warnings():
        warnings.simplefilter('error')
        assert resolve_pretrained('torchvision://resnet50') == str(f)
    assert resolve_pretrained('/some/local.pth') == '/some/local.pth' and resolve_pretrained(None) is None
