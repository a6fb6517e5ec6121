"""Three training steps in the fp32 parity mode (model.backbone.compute_dtype = torch.float32) at the bench workload -- run under
rocprofv3 --kernel-trace --stats to see where the 1e-3 mode's step goes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bonai_amd.config import Config
from bonai_amd.engine import Trainer
from bonai_amd.loft import build_detector
from bonai_amd.synth import make_batch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
cfg = Config.fromfile(os.path.join(ROOT, 'configs/loft_foa', 'loft_foa_r50_fpn_2x_bonai.py'))
torch.manual_seed(0)
m = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg).cuda().train()
tr = Trainer(m, lr=0.005)
data = make_batch(8, 1024, 80, device='cuda')
m.backbone.compute_dtype = torch.float32
from bonai_amd import kernels as K
K.F32_CONTRACT = {'planes_f16': K.F32_PLANES_F16, 'planes_bf16': K.F32_PLANES_BF16, 'split6': K.F32_SPLIT6, 'split3': K.F32_SPLIT3,
                  'exact': K.F32_EXACT}[os.environ.get('F32', 'planes_f16')]
for i in range(int(os.environ.get('STEPS', '3'))):
    torch.cuda.synchronize(); t0 = time.time()
    tr.train_step(data)
    torch.cuda.synchronize(); print(f'step {i}: {(time.time() - t0) * 1e3:.1f} ms', flush=True)
