"""Build libloft_hip.so and libloft_hip_f16.so (gfx950) in-tree with hipcc.  `python -m bonai_amd.build [--force]`.

Each .hip translation unit is compiled to an object (so per-file flags are possible: the
bit-exact integer/box kernels are built with -ffp-contract=off) and linked into one shared
library, bonai_amd/csrc/libloft_hip.so, which travels to the GPU box with the repo snapshot.
The same sources are compiled a second time with -DLOFT_ACT_F16 (16-bit type = IEEE binary16 instead of
bfloat16; loft_common.h) into libloft_hip_f16.so: same C-ABI, used by the fp16 configs.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'csrc')
LIB = os.path.join(CSRC, 'libloft_hip.so')
LIB_F16 = os.path.join(CSRC, 'libloft_hip_f16.so')
ARCH = 'gfx950'
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')

COMMON = ['--offload-arch=' + ARCH, '-O3', '-std=c++17', '-fPIC', '-fvisibility=hidden', '-Wno-unused-result']
# per-file extra flags
FLAGS = {
    'roi_align.hip': ['-ffp-contract=off'],
    'nms.hip': ['-ffp-contract=off'],
    'boxes.hip': ['-ffp-contract=off'],
    'deform.hip': ['-ffp-contract=off'],
}


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith('.hip'))


def _newest_header():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    hs.append(os.path.join(os.path.dirname(CSRC), '..', 'include', 'loft_hip.h'))
    return max(os.path.getmtime(h) for h in hs if os.path.exists(h))


def _compile(src, force, f16=False):
    obj = os.path.join(CSRC, src[:-4] + ('.f16.o' if f16 else '.o'))
    spath = os.path.join(CSRC, src)
    if (not force and os.path.exists(obj) and os.path.getmtime(obj) >= os.path.getmtime(spath)
            and os.path.getmtime(obj) >= _newest_header()):
        return obj, False
    cmd = [HIPCC] + COMMON + FLAGS.get(src, []) + (['-DLOFT_ACT_F16=1'] if f16 else []) + ['-c', spath, '-o', obj]
    subprocess.check_call(cmd)
    return obj, True


def build(force=False, verbose=False):
    srcs = sources()
    jobs = [(s, f16) for f16 in (False, True) for s in srcs]
    with ThreadPoolExecutor(max_workers=min(16, len(jobs), os.cpu_count() or 8)) as ex:
        res = list(ex.map(lambda j: _compile(j[0], force, j[1]), jobs))
    for lib, part in ((LIB, res[:len(srcs)]), (LIB_F16, res[len(srcs):])):
        if force or any(r[1] for r in part) or not os.path.exists(lib):
            cmd = [HIPCC, '--offload-arch=' + ARCH, '-shared', '-fPIC', '-o', lib] + [r[0] for r in part]
            subprocess.check_call(cmd)
            if verbose:
                print('linked', lib)
    return LIB


def source_hash():
    """sha256 (first 16 hex digits) over the kernel sources and headers -- what a committed profile under profiles/ was measured
    on.  bench.py quotes a profile file only when the hash recorded in it equals the running tree's (the GPU box has no .git)."""
    import hashlib
    h = hashlib.sha256()
    files = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(('.hip', '.h'))]
    files.append(os.path.join(os.path.dirname(CSRC), '..', 'include', 'loft_hip.h'))
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, 'rb').read())
    return h.hexdigest()[:16]


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
