"""Where the step's weight packing launch (loft_fold_pack_multi) spends its time: all records, and by record kind."""
import os, struct, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bonai_amd import kernels as K
from bonai_amd.config import Config
from bonai_amd.engine import Trainer
from bonai_amd.loft import build_detector
from bonai_amd.synth import make_batch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
cfg = Config.fromfile(os.path.join(ROOT, 'configs', 'loft_foa', 'loft_foa_r50_fpn_2x_bonai.py'))
torch.manual_seed(0)
model = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg).cuda().train()
tr = Trainer(model)
data = make_batch(2, 512, 20, device='cuda')
for _ in range(3):
    tr.train_step(data)
torch.cuda.synchronize()
reg = tr.prepack
lib = K.L.load()


def rows_of(pred):
    rows, chunk, elems = [], 0, 0
    for j in (m for key in reg.order for m in reg.jobs[key]['members']):
        Cout, Cin, RS, CoutP, CinP = j['dims']
        if not pred(RS):
            continue
        bn = j['bn']
        p = lambda t: 0 if t is None else t.data_ptr()
        eps_bits = struct.unpack('<i', struct.pack('<f', j['eps']))[0]
        rows.append([p(j['w']), p(j['cb']), p(bn[0]) if bn else 0, p(bn[1]) if bn else 0, p(bn[2]) if bn else 0, p(bn[3]) if bn else 0,
                     p(j['wp']), p(j['wpt']), p(j['bias']), eps_bits, Cout, Cin, RS, CoutP, CinP, chunk])
        elems += Cout * Cin * abs(RS)
        if RS < 0:
            chunk += Cout
        elif RS > K.FOLD_TILE_MAX_RS:
            chunk += (CoutP * CinP * RS + K.FOLD_CHUNK - 1) // K.FOLD_CHUNK
        else:
            nt = 64 if RS == 1 else K.FOLD_NT_TAPS
            chunk += ((CoutP + nt - 1) // nt) * ((CinP + 63) // 64)
    return rows, chunk, elems


def timeit(fn, iters=20):
    fn(); fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


for name, pred in (('all', lambda r: True), ('1x1 (RS = 1)', lambda r: r == 1), ('3x3 (RS = 9)', lambda r: r == 9),
                   ('other tiled (1 < RS < 9)', lambda r: 1 < r < 9), ('n-major (Linear on CHW)', lambda r: r < 0),
                   ('per-element (RS > 9)', lambda r: r > 9)):
    rows, chunk, elems = rows_of(pred)
    if not rows:
        continue
    desc = K.h2d(rows, torch.int64, 'cuda')
    torch.cuda.synchronize()
    t = timeit(lambda: K.L.check(lib.loft_fold_pack_multi(K.L.ptr(desc), len(rows), K.c_int64(chunk), K.L.stream()), 'pack'))
    print(f'{name:28s} {len(rows):4d} records {chunk:6d} chunks {elems / 1e6:7.2f} M weights  {t:7.1f} us   '
          f'{elems * 8 / t / 1e6:5.2f} TB/s (4 B read + 2 x 2 B written per weight)')
t = timeit(lambda: reg.run(reg.step))
print(f'PrepackRegistry.run (pack + the n-major transposes): {t:.1f} us')
