"""Exploratory: fp32 parity mode vs the reference fixture (prints discrepancy statistics)."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bonai_amd import kernels as K
from bonai_amd.config import Config
from bonai_amd.loft import build_detector
from bonai_amd.synth import make_batch
from oracle.synth_weights import synth_tensor
import torch.nn.functional as F

# unit: conv_tap_f32 vs CPU fp64
torch.manual_seed(0)
for (cin, cout, k, s, p, hw) in [(64, 64, 3, 1, 1, 20), (256, 128, 1, 2, 0, 17), (32, 36, 3, 2, 1, 9), (96, 256, 1, 1, 0, 7)]:
    x = torch.randn(2, cin, hw, hw)
    w = torch.randn(cout, cin, k, k) * 0.05
    b = torch.randn(cout)
    ref = F.relu(F.conv2d(x.double(), w.double(), b.double(), s, p)).float()
    xp = x.cuda().contiguous(memory_format=torch.channels_last)
    wp = K.pack_w_fwd(w.cuda(), torch.float32)[None]
    y = K.conv2d_fwd(xp, wp, b.cuda(), k, k, s, p, relu=True, out_dtype=torch.float32)
    print('conv', cin, cout, k, s, 'max err', (y.cpu() - ref).abs().max().item(), 'ref max', ref.abs().max().item())

gd = np.load(os.path.join(ROOT, 'tests', 'golden', 'e2e_test_256.npz'))
size = int(gd['meta'][0])
cfg = Config.fromfile(os.path.join(ROOT, 'configs', 'loft_foa', 'loft_foa_r50_fpn_2x_bonai.py'))
m = build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
m.load_state_dict({k: synth_tensor(k, v.shape) for k, v in m.state_dict().items()})
m = m.cuda().eval()
m.backbone.compute_dtype = torch.float32
data = make_batch(1, size, 4, device='cuda')
with torch.no_grad():
    bbox_results, segm_results, offset_results = m(img=[data['img']], img_metas=[data['img_metas']], return_loss=False, rescale=True)
det = bbox_results[0]
want = gd['det']
print('n det', det.shape, 'want', want.shape)
n = min(len(det), len(want))
d = np.abs(det[:n] - want[:n])
print('rowwise max abs diff box', d[:, :4].max(), 'score', d[:, 4].max())
print('rows with box diff > 1e-3 px:', (d[:, :4].max(1) > 1e-3).sum(), ' >1e-2:', (d[:, :4].max(1) > 1e-2).sum())
print('rows with score diff > 1e-5:', (d[:, 4] > 1e-5).sum())
od = np.abs(offset_results[:n] - gd['offsets'][:n])
print('offset max diff', od.max(), 'rows>1e-3', (od.max(1) > 1e-3).sum())
areas = np.array([s.sum() for s in segm_results[0]])
print('mask area diffs: n!=', (areas[:n] != gd['mask_area'][:n]).sum(), 'max', np.abs(areas[:n] - gd['mask_area'][:n]).max())
rs = np.stack([s.sum(1) for s in segm_results[0][:64]]).astype(np.int32)
print('rowsum mismatch', (rs != gd['mask_rowsum']).sum(), 'of', rs.size)
dt, wt = torch.from_numpy(det), torch.from_numpy(want)
cd = (wt[:, None, :] - dt[None, :, :]).abs().amax(-1)     # [2000,2000]
best, arg = cd.min(1)
print('order-insensitive: ref rows with a match within 1e-3:', (best < 1e-3).sum().item(), ' within 1e-2:', (best < 1e-2).sum().item(),
      'max', best.max().item())
print('first mismatching row index (rowwise):', int(np.argmax(d[:, :4].max(1) > 1e-3)))
bad = np.where(best.numpy() >= 1e-3)[0]
print('unmatched ref rows', bad[:20], want[bad[:5]], det[arg[bad[:5]].numpy()])
offm = torch.from_numpy(offset_results)[arg]
ok = best < 1e-3
print('offset diff on matched', (offm[ok] - torch.from_numpy(gd['offsets'])[ok]).abs().max().item())
am = torch.from_numpy(areas)[arg]
print('mask area diff on matched: n!=', (am[ok] != torch.from_numpy(gd['mask_area'])[ok]).sum().item(), 'max', (am[ok] - torch.from_numpy(gd['mask_area'])[ok]).abs().max().item())
