"""Compile the plain-C oracle (oracle/loft_oracle.c) into oracle/_build/liboracle.so.

TEST INFRASTRUCTURE.  Called by __graft_entry__.build() and lazily by oracle.cops.
"""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, 'loft_oracle.c')
OUT_DIR = os.path.join(HERE, '_build')
OUT = os.path.join(OUT_DIR, 'liboracle.so')


def build(force=False):
    os.makedirs(OUT_DIR, exist_ok=True)
    if (not force and os.path.exists(OUT)
            and os.path.getmtime(OUT) >= os.path.getmtime(SRC)):
        return OUT
    cmd = ['gcc', '-O2', '-ffp-contract=off', '-fno-fast-math', '-std=c99', '-shared', '-fPIC',
           SRC, '-o', OUT, '-lm']
    subprocess.check_call(cmd)
    return OUT


if __name__ == '__main__':
    print(build(force=True))
