"""Where the host's time per step goes (cProfile over N steps on 128^2 tiles, where the GPU never holds the host back).
    python tools/probes/host_cprofile.py [config] [steps]
Note: the backward runs on autograd's own thread; cProfile sees the calling thread only, so the backward appears as the time
spent inside `run_backward` -- the per-function rows cover forward, loss glue, optimizer and the launch helpers they call.
A second pass profiles the autograd thread through threading.setprofile."""
import cProfile
import os
import pstats
import sys
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from bonai_amd.config import Config  # noqa: E402
from bonai_amd.engine import Trainer  # noqa: E402
from bonai_amd.loft import build_detector  # noqa: E402
from bonai_amd.synth import make_batch  # noqa: E402

cfgname = sys.argv[1] if len(sys.argv) > 1 else 'loft_foa_r50_fpn_2x_bonai.py'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
cfg = Config.fromfile(os.path.join(ROOT, 'configs/loft_foa', cfgname))
if cfg.get('fp16'):
    from bonai_amd import lib as L
    L.set_act16(torch.float16)
torch.manual_seed(0)
import warnings
warnings.simplefilter('ignore')
m = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg).cuda().train()
tr = Trainer(m, lr=0.005, loss_scale=(cfg.get('fp16') or {}).get('loss_scale', 1.0))
data = make_batch(8, 128, 8, device='cuda')
for _ in range(5):
    tr.train_step(data)
torch.cuda.synchronize()
profs = {}


def hook(frame, event, arg):            # threading.setprofile target: one cProfile per thread
    p = profs.get(threading.get_ident())
    if p is None:
        p = profs[threading.get_ident()] = cProfile.Profile()
        p.enable()
    return None


threading.setprofile(hook)
main = cProfile.Profile()
main.enable()
for _ in range(steps):
    tr.train_step(data)
torch.cuda.synchronize()
main.disable()
threading.setprofile(None)
for p in profs.values():
    p.disable()
for tag, p in [('calling thread', main)] + [(f'other thread {i}', p) for i, p in enumerate(profs.values())]:
    st = pstats.Stats(p)
    st.strip_dirs()
    tot = sum(v[2] for v in st.stats.values())
    print(f'==== {tag}: {1e3 * tot / steps:.2f} ms of own time per step, top functions by own time (ms per step)')
    rows = sorted(st.stats.items(), key=lambda kv: -kv[1][2])[:45]
    for (fn, line, name), (cc, nc, tt, ct, callers) in rows:
        print(f'{1e3 * tt / steps:8.3f} ms  {nc / steps:8.1f} calls  cum {1e3 * ct / steps:8.3f}  {fn}:{line}({name})')
