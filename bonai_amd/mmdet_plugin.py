"""Reference-side plugin: make the reference's own entry points build the MI355X-native classes.

    import bonai_amd.mmdet_plugin            # from tools/train.py / tools/test.py of the reference, or a config's custom import
    bonai_amd.mmdet_plugin.register()

After ``register()`` the reference's ``mmdet.models.build_detector(cfg.model, train_cfg, test_cfg)``
(mmdet/models/builder.py:65-67) resolves ``type='LOFT'`` -- and every other type string of configs/loft_foa/* -- to the
classes of ``bonai_amd.loft`` (``force=True`` replaces the stock entries of the registries declared at
mmdet/models/builder.py:4-10, mmdet/core/bbox/builder.py:3-5, mmdet/core/anchor/builder.py:3,
mmdet/core/bbox/iou_calculators/builder.py:3).  The classes keep the reference's constructor arguments and state_dict keys, so
reference configs and checkpoints are consumed unchanged.  Executed by tests/test_plugin_cpu.py against the reference tree.
"""
import importlib

_PAIRS = (
    ('mmdet.models.builder', 'DETECTORS'), ('mmdet.models.builder', 'BACKBONES'), ('mmdet.models.builder', 'NECKS'),
    ('mmdet.models.builder', 'HEADS'), ('mmdet.models.builder', 'ROI_EXTRACTORS'), ('mmdet.models.builder', 'SHARED_HEADS'),
    ('mmdet.models.builder', 'LOSSES'), ('mmdet.core.bbox.builder', 'BBOX_ASSIGNERS'), ('mmdet.core.bbox.builder', 'BBOX_SAMPLERS'),
    ('mmdet.core.bbox.builder', 'BBOX_CODERS'), ('mmdet.core.anchor.builder', 'ANCHOR_GENERATORS'),
    ('mmdet.core.bbox.iou_calculators.builder', 'IOU_CALCULATORS'),
)


def register(force=True):
    """Register every bonai_amd.loft class under its reference name in the reference's registries.
    -> {registry name: [type strings registered]}."""
    import bonai_amd.loft as L
    done = {}
    for modname, regname in _PAIRS:
        ref_reg = getattr(importlib.import_module(modname), regname)
        ours = getattr(L, regname)
        for name, cls in ours.module_dict.items():
            ref_reg.register_module(name=name, force=force, module=cls)
        done[regname] = sorted(ours.module_dict)
    return done
