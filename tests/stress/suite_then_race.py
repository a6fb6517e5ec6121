"""TEST INFRASTRUCTURE (a stress form of a GPU test; imports the oracle's name-seeded weights like the test it repeats).
The driver's failing situation, reproduced in ONE process: the GPU test files that precede tests/test_edge_gpu.py in
alphabetical order run first (pytest.main, in-process: allocator history, stream pool, library state as on the driver's box),
then the body of test_trainer_direct_grad_sink_matches_autograd_accumulation is repeated N times with diagnostics.
    python tests/stress/suite_then_race.py [N] [--no-history]        (RACE_OLD_NN=1: round-2 bonai_amd/nn.py)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # tests/stress/ -> repo root
sys.path.insert(0, ROOT)
os.chdir(ROOT)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 20
    if os.environ.get('RACE_OLD_NN'):      # needs:  git show c4f7e1c:bonai_amd/nn.py > tools/probes/_abl/nn_r2.py  (not tracked)
        import importlib.util
        import bonai_amd
        spec = importlib.util.spec_from_file_location('bonai_amd.nn', os.path.join(ROOT, 'tools', 'probes', '_abl', 'nn_r2.py'))
        mod = importlib.util.module_from_spec(spec)
        sys.modules['bonai_amd.nn'] = mod
        spec.loader.exec_module(mod)
        mod.roi_align_discard = lambda y: None
        bonai_amd.nn = mod
    tag = 'oldnn' if os.environ.get('RACE_OLD_NN') else 'new'
    if '--no-history' not in sys.argv:
        import pytest
        files = ['tests/test_conv_gpu.py', 'tests/test_conv_variants_gpu.py', 'tests/test_data_gpu.py', 'tests/test_ddp_gpu.py',
                 'tests/test_deform_gpu.py', 'tests/test_e2e_gpu.py',
                 'tests/test_edge_gpu.py::test_no_gt_and_ragged_gt', 'tests/test_edge_gpu.py::test_rccl_reducer_path_single_rank',
                 'tests/test_edge_gpu.py::test_forward_dummy_shapes']
        rc = pytest.main(files + ['-q', '-m', 'gpu', '-x', '-p', 'no:cacheprovider', '-o', 'addopts='])
        print(f'HISTORY[{tag}] pytest rc={rc}  allocated={torch.cuda.memory_allocated() >> 20} MiB reserved={torch.cuda.memory_reserved() >> 20} MiB',
              flush=True)
    import warnings
    warnings.simplefilter('ignore')
    from bonai_amd.config import Config
    from bonai_amd.debug import DBG
    from bonai_amd.engine import Trainer
    from bonai_amd.loft import build_detector
    from bonai_amd.loft.core import RandomSampler
    from bonai_amd.synth import make_batch
    from oracle.synth_weights import synth_tensor
    RandomSampler.choice_mode = 'first'
    cfg = Config.fromfile(os.path.join(ROOT, 'configs', 'loft_foa', 'loft_foa_r50_fpn_2x_bonai.py'))
    data = make_batch(2, 256, 8, device='cuda')

    def build():
        m = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg)
        m.load_state_dict({k: synth_tensor(k, v.shape) for k, v in m.state_dict().items()})
        return m.cuda().train()
    first_want = None
    nbad_runs = 0
    for it in range(n):
        ref = build()
        with DBG.override(no_side_stream=True):
            o = ref.train_step(data)
            o['loss'].backward()
        loss_ref = float(o['loss'])
        want = {k: p.grad.clone() for k, p in ref.named_parameters() if p.grad is not None}
        if first_want is None:
            first_want = want
        drift = max((want[k] - first_want[k]).norm().item() / (first_want[k].norm().item() + 1e-12) for k in want)
        m = build()
        tr = Trainer(m, lr=0.0, momentum=0.0, weight_decay=0.0)
        worst, bad, losses = ('', 0.0), [], []
        for rep in range(3):
            o = tr.train_step(data, lr=0.0)
            torch.cuda.synchronize()
            losses.append(float(o['loss']))
            got = {k: p.grad for k, p in m.named_parameters() if p.requires_grad}
            for k, w in want.items():
                rel = (got[k] - w).norm().item() / (w.norm().item() + 1e-12)
                if rel > 1e-2:
                    bad.append((rep, k, round(rel, 3)))
                if rel > worst[1]:
                    worst = (f'{k}@rep{rep}', rel)
        nbad_runs += bool(bad) or drift > 1e-2
        print(f"ITER[{tag}] {it:3d} {'BAD' if bad or drift > 1e-2 else 'ok '} ref_drift_vs_iter0={drift:.2e} worst={worst[0]} rel={worst[1]:.2e} "
              f"nbad={len(bad)} loss_ref={loss_ref:.5f} loss_tr={losses} first_bad={bad[:4]}", flush=True)
        del ref, m, tr, want, got
    print(f'SUMMARY[{tag}] {nbad_runs} bad of {n}', flush=True)


if __name__ == '__main__':
    main()
