"""Where a prefetched batch spends its host time (run on the GPU box): decode fan-out, annotation side, device batch."""
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from PIL import Image
from bonai_amd.data import to_device_batch
from bonai_amd.dataset import BonaiDataset, decode_tile_into
from bonai_amd.synth import synth_bonai_anns


def main():
    d = tempfile.mkdtemp()
    size, n_tiles, bs = 1024, 32, 8
    rng = np.random.RandomState(0)
    base = (rng.randint(0, 255, (size // 8, size // 8, 3)).astype(np.uint8)).repeat(8, 0).repeat(8, 1)
    images, annotations, aid = [], [], 0
    for i in range(n_tiles):
        name = f'tile_{i}.png'
        Image.fromarray(np.roll(base, 17 * i, axis=1)).save(os.path.join(d, name), compress_level=3)
        images.append(dict(id=10 + i, file_name=name, width=size, height=size))
        for a in synth_bonai_anns(seed=i, n=80, size=size):
            aid += 1
            annotations.append(dict(a, id=aid, image_id=10 + i))
    f = os.path.join(d, 'ann.json')
    json.dump(dict(images=images, annotations=annotations, categories=[dict(id=1, name='building')]), open(f, 'w'))
    ds = BonaiDataset(f, d, flip_ratio=0.5, flip_direction='vertical', seed=2)
    torch.zeros(1, device='cuda')
    nw = min(16, os.cpu_count() or 8)
    t = time.time()
    pool = ds._decoder_pool(nw)
    print(f'pool of {nw} forked in {time.time() - t:.2f} s')
    st = ds._staging_block(5, bs, size, size, True)
    print('pinned:', st['registered'] is not None)
    buf = st['whole'][0].numpy()
    side = torch.cuda.Stream()
    for rep in range(4):
        g = list(range(rep * bs, rep * bs + bs))
        t0 = time.time()
        futs = [pool.submit(decode_tile_into, os.path.join(d, ds.data_infos[j]['filename']), st['shm'].name, i * size * size * 3, size, size)
                for i, j in enumerate(g)]
        t1 = time.time()
        samples = [ds.prepare_train_img(j, 0.3 if i % 2 else 0.9, buf[i], decode=False) for i, j in enumerate(g)]
        t2 = time.time()
        for fu in futs:
            assert fu.result() is None
        t3 = time.time()
        with torch.cuda.stream(side):
            batch = to_device_batch(samples, device='cuda', staged=st['whole'][0][:bs])
        t4 = time.time()
        side.synchronize()
        t5 = time.time()
        print(f'batch {rep}: submit {1e3 * (t1 - t0):.1f} ms  annotations {1e3 * (t2 - t1):.1f}  wait decode {1e3 * (t3 - t2):.1f}  '
              f'to_device_batch (host) {1e3 * (t4 - t3):.1f}  device drain {1e3 * (t5 - t4):.1f}')
    import cProfile
    import pstats
    pr = cProfile.Profile()
    pr.enable()
    with torch.cuda.stream(side):
        for _ in range(5):
            to_device_batch(samples, device='cuda', staged=st['whole'][0][:bs])
    pr.disable()
    pstats.Stats(pr).sort_stats('cumulative').print_stats(14)
    ds.close()


if __name__ == '__main__':
    main()
