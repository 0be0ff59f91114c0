"""Builds libpworld.so (HIP, gfx950) in-tree.  Run: python -m multiagent_rl_amd.build_native"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, 'csrc', 'pworld.hip')
OUT = os.path.join(HERE, 'libpworld.so')
DEPS = [os.path.join(HERE, 'csrc', f) for f in sorted(os.listdir(os.path.join(HERE, 'csrc')))] + \
       [os.path.join(ROOT, 'include', 'pworld.h'), os.path.join(ROOT, 'include', 'pworld_math.h')]

# -ffp-contract=off + correctly rounded div/sqrt: the kernels must reproduce the float32
# oracle bit for bit (HIP's device default is fp-contract=fast).
FLAGS = ['-O3', '--offload-arch=gfx950', '-std=c++17', '-fPIC', '-shared',
         '-ffp-contract=off', '-fno-fast-math', '-fhip-fp32-correctly-rounded-divide-sqrt',
         '-Wall', '-Wno-unused-function', '-Wno-cuda-compat', '-Wno-pass-failed', '-I', os.path.join(ROOT, 'include')]


def build(force=False, verbose=False):
    if not force and os.path.exists(OUT) and all(os.path.getmtime(d) <= os.path.getmtime(OUT) for d in DEPS):
        return OUT
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    cmd = [hipcc] + [f for f in FLAGS if f] + ['-o', OUT, SRC]
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd)
    return OUT


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
