"""cProfile of the host side of one training step (run on the GPU box): python tools/probes/host_profile.py [config]"""
import cProfile, os, pstats, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch
from bonai_amd.config import Config
from bonai_amd.engine import Trainer
from bonai_amd.loft import build_detector
from bonai_amd.synth import make_batch
ROOT = sys.path[0]
name = sys.argv[1] if len(sys.argv) > 1 else 'loft_foa_r50_fpn_2x_bonai.py'
cfg = Config.fromfile(os.path.join(ROOT, 'configs/loft_foa', name))
torch.manual_seed(0)
m = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg).cuda().train()
tr = Trainer(m, lr=1e-4)
data = make_batch(8, 1024, 80, device='cuda')
for _ in range(3):
    tr.train_step(data)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(4):
    tr.train_step(data)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(22); st.sort_stats('cumulative').print_stats('bonai_amd', 30)
