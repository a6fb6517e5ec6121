import sys; sys.path.insert(0,'/root/repo')
import numpy as np, torch
from oracle import ops_ref
from bonai_amd import kernels as K
sys.path.insert(0,'/root/repo/tests')
from test_roi_nms_gpu import _rand_rois
for P in (7,14):
    rng = np.random.RandomState(0)
    B, C, size = 2, 64, 256
    strides = [4, 8, 16, 32]
    torch.manual_seed(0)
    feats = [torch.randn(B, C, size // s, size // s).bfloat16().float() for s in strides]
    rois = _rand_rois(rng, 300, B, size)
    rois[0, 1:] = torch.tensor([10., 10., 10., 10.]); rois[1, 1:] = torch.tensor([-500., -500., -400., -400.]); rois[2, 1:] = torch.tensor([0., 0., float(size), float(size)])
    ref = ops_ref.roi_extract(feats, rois, P, strides)
    dfeats = [f.to('cuda', torch.bfloat16).contiguous(memory_format=torch.channels_last) for f in feats]
    out = K.roi_align_fwd(dfeats, rois.cuda(), P, strides).float().cpu()
    err = (out - ref).abs().amax(dim=(1,))   # [N,P,P]
    bad = (err.amax(dim=(1,2)) > 0.05).nonzero().flatten()
    lv = ops_ref.map_roi_levels(rois)
    print('P', P, 'bad rois', len(bad))
    for k in bad[:6].tolist():
        r = rois[k]; s = strides[lv[k]]
        print(k, 'lvl', int(lv[k]), 'roi/stride', [round(float(v)/s,2) for v in r[1:]], 'W', size//s)
        print((err[k] > 0.05).int())
