"""Stand-in `mmcv` (and cv2 / pycocotools / torchvision / terminaltables) modules that let the
reference's pure-torch python import and run on CPU **in the survey/build container only**.

TEST INFRASTRUCTURE.  Nothing under /root/reference is copied; the reference's own modules are
imported from where they lie.  mmcv==1.0.5 itself is absent from this image, so the handful of
mmcv helpers the LOFT path touches are restated here from their published behaviour
(Registry/build_from_cfg, ConvModule = conv->norm->act, the weight-init helpers); the compiled
mmcv ops (RoIAlign, nms, soft_nms) are routed to the plain-C oracle (oracle/loft_oracle.c) --
which is why goldens that cross those ops are "parity unpinned" (SURVEY.md section 8c).

Usage (make_goldens.py):   import mmcv_stub; mmcv_stub.install(); import mmdet...
"""
import inspect
import math
import os
import sys
import types

import torch
import torch.nn as nn

REFERENCE_ROOT = os.environ.get('BONAI_REFERENCE', '/root/reference')


class _Auto(types.ModuleType):
    """Module whose unknown attributes resolve to an inert placeholder class."""

    def __getattr__(self, name):
        if name.startswith('__'):
            raise AttributeError(name)
        ph = type(name, (), {'__init__': lambda self, *a, **k: None})
        setattr(self, name, ph)
        return ph


def _mod(name, auto=True):
    m = (_Auto if auto else types.ModuleType)(name)
    m.__path__ = []
    sys.modules[name] = m
    parent, _, child = name.rpartition('.')
    if parent and parent in sys.modules:
        setattr(sys.modules[parent], child, m)
    return m


# ---------------------------------------------------------------- mmcv.utils
class Registry:
    def __init__(self, name):
        self._name = name
        self._module_dict = {}

    @property
    def name(self):
        return self._name

    @property
    def module_dict(self):
        return self._module_dict

    def get(self, key):
        return self._module_dict.get(key)

    def _register(self, cls, name=None, force=False):
        key = name or cls.__name__
        if not force and key in self._module_dict:
            raise KeyError(f'{key} already in {self._name}')
        self._module_dict[key] = cls

    def register_module(self, name=None, force=False, module=None):
        if inspect.isclass(name):  # bare-decorator form
            self._register(name, force=force)
            return name
        if module is not None:
            self._register(module, name, force)
            return module

        def deco(cls):
            self._register(cls, name, force)
            return cls
        return deco


def build_from_cfg(cfg, registry, default_args=None):
    args = dict(cfg)
    if default_args:
        for k, v in default_args.items():
            args.setdefault(k, v)
    typ = args.pop('type')
    cls = registry.get(typ) if isinstance(typ, str) else typ
    if cls is None:
        raise KeyError(f'{typ} not in {registry.name}')
    return cls(**args)


class ConfigDict(dict):
    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError:
            raise AttributeError(k)
        return v

    def __setattr__(self, k, v):
        self[k] = v


def to_cfg(obj):
    if isinstance(obj, dict):
        return ConfigDict({k: to_cfg(v) for k, v in obj.items()})
    if isinstance(obj, (list, tuple)):
        return type(obj)(to_cfg(v) for v in obj)
    return obj


# ---------------------------------------------------------------- mmcv.cnn
def constant_init(module, val, bias=0):
    if getattr(module, 'weight', None) is not None:
        nn.init.constant_(module.weight, val)
    if getattr(module, 'bias', None) is not None:
        nn.init.constant_(module.bias, bias)


def xavier_init(module, gain=1, bias=0, distribution='normal'):
    if distribution == 'uniform':
        nn.init.xavier_uniform_(module.weight, gain=gain)
    else:
        nn.init.xavier_normal_(module.weight, gain=gain)
    if getattr(module, 'bias', None) is not None:
        nn.init.constant_(module.bias, bias)


def normal_init(module, mean=0, std=1, bias=0):
    nn.init.normal_(module.weight, mean, std)
    if getattr(module, 'bias', None) is not None:
        nn.init.constant_(module.bias, bias)


def kaiming_init(module, a=0, mode='fan_out', nonlinearity='relu', bias=0, distribution='normal'):
    if distribution == 'uniform':
        nn.init.kaiming_uniform_(module.weight, a=a, mode=mode, nonlinearity=nonlinearity)
    else:
        nn.init.kaiming_normal_(module.weight, a=a, mode=mode, nonlinearity=nonlinearity)
    if getattr(module, 'bias', None) is not None:
        nn.init.constant_(module.bias, bias)


def build_conv_layer(cfg, *args, **kwargs):
    assert cfg is None or cfg.get('type', 'Conv2d') in ('Conv', 'Conv2d'), cfg
    return nn.Conv2d(*args, **kwargs)


def build_norm_layer(cfg, num_features, postfix=''):
    cfg = dict(cfg)
    typ = cfg.pop('type')
    requires_grad = cfg.pop('requires_grad', True)
    cfg.setdefault('eps', 1e-5)
    assert typ == 'BN', typ
    layer = nn.BatchNorm2d(num_features, **cfg)
    for p in layer.parameters():
        p.requires_grad = requires_grad
    return 'bn' + str(postfix), layer


def build_upsample_layer(cfg, *args, **kwargs):
    cfg = dict(cfg)
    typ = cfg.pop('type')
    assert typ == 'deconv', typ
    return nn.ConvTranspose2d(*args, **kwargs, **cfg)


class ConvModule(nn.Module):
    """conv -> norm -> act, bias='auto' (= no bias when a norm follows)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 bias='auto', conv_cfg=None, norm_cfg=None, act_cfg=dict(type='ReLU'), inplace=True,
                 with_spectral_norm=False, padding_mode='zeros', order=('conv', 'norm', 'act')):
        super().__init__()
        assert order == ('conv', 'norm', 'act') and padding_mode == 'zeros'
        self.with_norm = norm_cfg is not None
        self.with_activation = act_cfg is not None
        if bias == 'auto':
            bias = not self.with_norm
        self.conv = build_conv_layer(conv_cfg, in_channels, out_channels, kernel_size, stride=stride,
                                     padding=padding, dilation=dilation, groups=groups, bias=bias)
        if self.with_norm:
            self.norm_name, norm = build_norm_layer(norm_cfg, out_channels)
            self.add_module(self.norm_name, norm)
        if self.with_activation:
            assert act_cfg['type'] == 'ReLU'
            self.activate = nn.ReLU(inplace=inplace)
        kaiming_init(self.conv, a=0, nonlinearity='relu')
        if self.with_norm:
            constant_init(getattr(self, self.norm_name), 1, bias=0)

    def forward(self, x, activate=True, norm=True):
        x = self.conv(x)
        if norm and self.with_norm:
            x = getattr(self, self.norm_name)(x)
        if activate and self.with_activation:
            x = self.activate(x)
        return x


# ---------------------------------------------------------------- mmcv.ops (-> plain-C oracle)
class RoIAlign(nn.Module):
    def __init__(self, output_size, spatial_scale=1.0, sampling_ratio=0, pool_mode='avg', aligned=True,
                 use_torchvision=False):
        super().__init__()
        self.output_size = torch.nn.modules.utils._pair(output_size)
        self.spatial_scale = float(spatial_scale)
        self.sampling_ratio = int(sampling_ratio)
        self.pool_mode = pool_mode
        self.aligned = aligned

    def forward(self, x, rois):
        from oracle import cops
        return cops.roi_align(x, rois, self.output_size, self.spatial_scale, self.sampling_ratio,
                              self.pool_mode, self.aligned)


def install():
    """Populate sys.modules with the stand-ins and put the reference on sys.path."""
    if 'mmcv' in sys.modules and getattr(sys.modules['mmcv'], '_bonai_stub', False):
        return
    repo = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    if repo not in sys.path:
        sys.path.insert(0, repo)
    from oracle import cops

    mmcv = _mod('mmcv')
    mmcv.__version__ = '1.0.5'
    mmcv._bonai_stub = True
    mmcv.is_tuple_of = lambda seq, t: isinstance(seq, tuple) and all(isinstance(s, t) for s in seq)
    mmcv.is_list_of = lambda seq, t: isinstance(seq, list) and all(isinstance(s, t) for s in seq)
    mmcv.is_str = lambda x: isinstance(x, str)
    mmcv.ConfigDict = ConfigDict
    mmcv.Config = ConfigDict

    u = _mod('mmcv.utils')
    u.Registry, u.build_from_cfg = Registry, build_from_cfg
    u.print_log = lambda msg, logger=None, level=None: None
    u.get_logger = lambda *a, **k: __import__('logging').getLogger('mmdet')
    u.ConfigDict = ConfigDict
    u.Config = ConfigDict

    c = _mod('mmcv.cnn')
    for f in (constant_init, xavier_init, normal_init, kaiming_init, build_conv_layer, build_norm_layer,
              build_upsample_layer, ConvModule):
        setattr(c, f.__name__, f)
    c.bias_init_with_prob = lambda p: float(-math.log((1 - p) / p))
    c.build_plugin_layer = lambda *a, **k: (_ for _ in ()).throw(NotImplementedError())
    c.Scale = type('Scale', (nn.Module,), {})
    _mod('mmcv.cnn.bricks')
    cw = _mod('mmcv.cnn.bricks.wrappers')
    _mod('mmcv.cnn.utils')

    o = _mod('mmcv.ops')
    o.Conv2d = nn.Conv2d
    o.ConvTranspose2d = nn.ConvTranspose2d
    o.Linear = nn.Linear
    o.MaxPool2d = nn.MaxPool2d
    cw.Conv2d = nn.Conv2d
    o.RoIAlign = RoIAlign
    o.roi_align = cops.roi_align
    o.nms, o.soft_nms, o.batched_nms = cops.nms, cops.soft_nms, cops.batched_nms
    on = _mod('mmcv.ops.nms')
    on.nms, on.soft_nms, on.batched_nms = cops.nms, cops.soft_nms, cops.batched_nms
    orr = _mod('mmcv.ops.roi_align')
    orr.roi_align, orr.RoIAlign = cops.roi_align, RoIAlign
    for sub in ('carafe', 'merge_cells', 'point_sample', 'deform_conv', 'modulated_deform_conv', 'corner_pool',
                'focal_loss', 'masked_conv', 'saconv', 'context_block', 'plugin'):
        _mod('mmcv.ops.' + sub)

    r = _mod('mmcv.runner')
    r.load_checkpoint = lambda *a, **k: None
    r.OptimizerHook = type('OptimizerHook', (), {})
    r.Hook = type('Hook', (), {})
    r.get_dist_info = lambda: (0, 1)
    _mod('mmcv.runner.hooks')
    p = _mod('mmcv.parallel')
    p.DataContainer = type('DataContainer', (), {})
    _mod('mmcv.image')
    _mod('mmcv.visualization')
    for name in ('cv2', 'pycocotools', 'pycocotools.mask', 'pycocotools.coco', 'pycocotools.cocoeval',
                 'terminaltables', 'torchvision', 'torchvision.ops', 'torchvision.models', 'six', 'six.moves',
                 'shapely', 'shapely.geometry', 'seaborn', 'matplotlib', 'matplotlib.pyplot', 'pandas_stub'):
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                _mod(name)
    six_moves = sys.modules.get('six.moves')
    if isinstance(six_moves, _Auto):
        six_moves.map, six_moves.zip = map, zip
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
