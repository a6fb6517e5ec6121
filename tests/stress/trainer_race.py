"""TEST INFRASTRUCTURE (a stress form of a GPU test; imports the oracle's name-seeded weights like the test it repeats).
Stress form of tests/test_edge_gpu.py::test_trainer_direct_grad_sink_matches_autograd_accumulation:
ONE fresh process = the reference gradients (plain autograd, one stream) and `reps` Trainer steps (gradient sink + unpack queue +
weight-gradient side stream + prepack + zero / scratch pools -- what bench.py times) on the same batch; prints one line
    RESULT <ok|BAD> switches=<LOFT_NO_* set> worst=<param> rel=<|got-want|/|want|> nbad=<#params over 1e-2> loss_ref=.. loss_tr=..
The driver script tests/stress/trainer_race.sh runs it N times per switch, several processes at once (their kernels interleave on
the one GPU, which moves every launch's timing), and tabulates."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # tests/stress/ -> repo root
sys.path.insert(0, ROOT)


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    size = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    prior = int(sys.argv[3]) if len(sys.argv) > 3 else 1     # random-init models built, run and freed first (allocator history)
    if os.environ.get('RACE_OLD_NN'):      # needs:  git show c4f7e1c:bonai_amd/nn.py > tools/probes/_abl/nn_r2.py  (not tracked)      # the round-2 bonai_amd/nn.py (process-global _PACK_CACHE) under the current tree
        import importlib.util
        import bonai_amd
        spec = importlib.util.spec_from_file_location('bonai_amd.nn', os.path.join(ROOT, 'tools', 'probes', '_abl', 'nn_r2.py'))
        mod = importlib.util.module_from_spec(spec)
        sys.modules['bonai_amd.nn'] = mod
        spec.loader.exec_module(mod)
        bonai_amd.nn = mod
    from bonai_amd.config import Config
    from bonai_amd.engine import Trainer
    from bonai_amd.loft import build_detector
    from bonai_amd.loft.core import RandomSampler
    from bonai_amd.synth import make_batch
    from oracle.synth_weights import synth_tensor
    RandomSampler.choice_mode = 'first'
    cfg = Config.fromfile(os.path.join(ROOT, 'configs', 'loft_foa', 'loft_foa_r50_fpn_2x_bonai.py'))
    data = make_batch(2, size, 8, device='cuda')
    switches = sorted(k for k in os.environ if k.startswith('LOFT_NO_') and os.environ[k])
    tag = ('oldnn,' if os.environ.get('RACE_OLD_NN') else '')
    import warnings
    warnings.simplefilter('ignore')
    for s in range(prior):
        torch.manual_seed(s)
        a = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg).cuda().train()
        Trainer(a, lr=1e-3).train_step(data)
        del a

    def build():
        m = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg)
        m.load_state_dict({k: synth_tensor(k, v.shape) for k, v in m.state_dict().items()})
        return m.cuda().train()
    from bonai_amd.debug import DBG
    ref = build()
    with DBG.override(**dict({k: False for k in DBG.active()}, no_side_stream=True)):
        out = ref.train_step(data)
        out['loss'].backward()
    loss_ref = float(out['loss'])
    want = {n: p.grad.clone() for n, p in ref.named_parameters() if p.grad is not None}
    m = build()
    tr = Trainer(m, lr=0.0, momentum=0.0, weight_decay=0.0)
    worst, nbad, losses = ('', 0.0), 0, []
    for rep in range(reps):
        o = tr.train_step(data, lr=0.0)
        torch.cuda.synchronize()
        losses.append(float(o['loss']))
        got = {n: p.grad for n, p in m.named_parameters() if p.requires_grad}
        for n, w in want.items():
            rel = (got[n] - w).norm().item() / (w.norm().item() + 1e-12)
            if rel > 1e-2:
                nbad += 1
            if rel > worst[1]:
                worst = (f'{n}@rep{rep}', rel)
    ok = nbad == 0
    print(f"RESULT {'ok' if ok else 'BAD'} switches={tag}{','.join(switches) or '-'} worst={worst[0]} rel={worst[1]:.3e} nbad={nbad} "
          f"loss_ref={loss_ref:.5f} loss_tr={','.join(f'{v:.5f}' for v in losses)}", flush=True)


if __name__ == '__main__':
    main()
