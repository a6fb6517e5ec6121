"""Which tensors still take a separate ReLU-backward pass (not folded into a producer's dgrad epilogue)."""
import os, sys, traceback
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bonai_amd import kernels as K
from bonai_amd.config import Config
from bonai_amd.engine import Trainer
from bonai_amd.loft import build_detector
from bonai_amd.synth import make_batch
cfg = Config.fromfile(os.path.join(ROOT, 'configs', 'loft_foa', 'loft_foa_r50_fpn_2x_bonai.py'))
torch.manual_seed(0)
model = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg).cuda().train()
tr = Trainer(model)
data = make_batch(8, 1024, 80, device='cuda')
tr.train_step(data)
orig = K.relu_bwd
def dbg(g, y):
    print('relu_bwd', tuple(g.shape), g.dtype)
    return orig(g, y)
K.relu_bwd = dbg
tr.train_step(data)
torch.cuda.synchronize()
