"""Which stage owns the bf16 path's offset error against the reference?
Offsets of the FOA head on the REFERENCE's 2000 boxes of tests/golden/e2e_test_256.npz, with the backbone + neck ("features") and
the RoI extractor + FOA head ("head") each run in fp32 parity mode or in bf16: 2 x 2 combinations (run on the GPU box)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from bonai_amd.config import Config
from bonai_amd.loft import build_detector
from bonai_amd.synth import make_batch
from oracle.synth_weights import synth_tensor


def main():
    gd = np.load(os.path.join(ROOT, 'tests', 'golden', 'e2e_test_256.npz'))
    size = int(gd['meta'][0])
    cfg = Config.fromfile(os.path.join(ROOT, 'configs', 'loft_foa', 'loft_foa_r50_fpn_2x_bonai.py'))
    m = build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
    m.load_state_dict({k: synth_tensor(k, v.shape) for k, v in m.state_dict().items()})
    m = m.cuda().eval()
    data = make_batch(1, size, 4, device='cuda')
    want, off_ref = torch.from_numpy(gd['det']), gd['offsets']
    rb = want[:, :4].cuda().contiguous()
    rois = torch.cat([rb.new_zeros(rb.shape[0], 1), rb], 1).contiguous()
    mag = np.sqrt((off_ref ** 2).sum(1))
    feats = {}
    with torch.no_grad():
        for name, dt in (('fp32', torch.float32), ('bf16', torch.bfloat16)):
            m.backbone.compute_dtype = dt
            feats[name] = [f.clone() for f in m.extract_feat(data['img'])]
        print(f'{"features":>9s} {"head":>6s} {"aEPE px":>9s} {"max EPE":>9s} {"p99":>8s} {"rel aEPE":>9s}   (mean |offset| {mag.mean():.2f} px, {len(mag)} boxes)')
        res = {}
        for fname in ('fp32', 'bf16'):
            for hname, hdt in (('fp32', torch.float32), ('bf16', torch.bfloat16)):
                fs = [f.to(hdt).contiguous(memory_format=torch.channels_last) for f in feats[fname]]
                op = m.roi_head._offset_forward(fs, rois)
                o = np.asarray(m.roi_head.offset_head.get_offsets(op, rb, None, False))
                epe = np.sqrt(((o - off_ref) ** 2).sum(1))
                res[(fname, hname)] = epe
                print(f'{fname:>9s} {hname:>6s} {epe.mean():9.5f} {epe.max():9.4f} {np.percentile(epe, 99):8.4f} {(epe / np.maximum(mag, 1e-6)).mean():9.6f}')
        worst = int(np.argmax(res[('bf16', 'bf16')]))
        print(f'worst box of the all-bf16 run: #{worst}, box {want[worst, :4].tolist()}, |offset| {mag[worst]:.2f} px; its EPE per combination:',
              {k: round(float(v[worst]), 4) for k, v in res.items()})
        # the FOA head by sub-stage, on fp32 features: RoIAlign output + ten grouped 3x3 convs ("convs") and the two FCs + fc_offset
        # ("fcs") each in fp32 or bf16 (the dtype of the activation a layer receives selects its kernels)
        head = m.roi_head.offset_head
        ext = m.roi_head.offset_roi_extractor
        from bonai_amd import nn as F2
        print(f'{"convs":>9s} {"fcs":>6s} {"aEPE px":>9s} {"max EPE":>9s} {"p99":>8s}   (fp32 features)')
        for cdt in (torch.float32, torch.bfloat16):
            for fdt in (torch.float32, torch.bfloat16):
                fs = [f.to(cdt).contiguous(memory_format=torch.channels_last) for f in feats['fp32']]
                x4 = ext(fs[:ext.num_inputs], rois, n_rot=4)
                for i in range(head.num_convs):
                    x4 = F2.conv2d(x4, [head.expand_convs[k][i].weight for k in range(4)], [head.expand_convs[k][i].bias for k in range(4)],
                                   pad=1, relu=True, groups=4, input_relu=i > 0)
                op = head._fc_tail(x4.to(fdt).contiguous(memory_format=torch.channels_last))
                o = np.asarray(head.get_offsets(op, rb, None, False))
                epe = np.sqrt(((o - off_ref) ** 2).sum(1))
                n = {torch.float32: 'fp32', torch.bfloat16: 'bf16'}
                print(f'{n[cdt]:>9s} {n[fdt]:>6s} {epe.mean():9.5f} {epe.max():9.4f} {np.percentile(epe, 99):8.4f}')
        m.backbone.compute_dtype = None


if __name__ == '__main__':
    main()
