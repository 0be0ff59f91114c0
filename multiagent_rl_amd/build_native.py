"""Builds libpworld.so (HIP, gfx950) in-tree.  Run: python -m multiagent_rl_amd.build_native"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRCS = [os.path.join(HERE, 'csrc', 'pworld.hip'), os.path.join(HERE, 'csrc', 'pworld_policy.hip')]
OUT = os.path.join(HERE, 'libpworld.so')
OBJ_DIR = os.path.join(HERE, 'csrc', '_obj')  # git-ignored (*.o); objects are kept so that one unit rebuilds alone
DEPS = [os.path.join(HERE, 'csrc', f) for f in sorted(os.listdir(os.path.join(HERE, 'csrc'))) if f.endswith(('.hip', '.hpp'))] + \
       [os.path.join(ROOT, 'include', 'pworld.h'), os.path.join(ROOT, 'include', 'pworld_math.h')]

# -ffp-contract=off + correctly rounded div/sqrt: the kernels must reproduce the float32
# oracle bit for bit (HIP's device default is fp-contract=fast).
FLAGS = ['-O3', '--offload-arch=gfx950', '-std=c++17', '-fPIC',
         '-ffp-contract=off', '-fno-fast-math', '-fhip-fp32-correctly-rounded-divide-sqrt',
         '-Wall', '-Wno-unused-function', '-Wno-cuda-compat', '-Wno-pass-failed', '-I', os.path.join(ROOT, 'include')]


def _unit_deps(obj):
    """The files one object was compiled from, as hipcc itself recorded them (-MD -MF <obj>.d): the unit's source, every
    header it includes under csrc/ and include/.  None if there is no record yet (-> compile)."""
    dep = obj + '.d'
    if not os.path.exists(dep):
        return None
    words = open(dep).read().replace('\\\n', ' ').split()
    files = [w for w in words[1:] if not w.endswith(':')]
    return [f for f in files if f.startswith(ROOT)] or None


def _stale(obj):
    if not os.path.exists(obj):
        return True
    deps = _unit_deps(obj)
    if deps is None:
        return True
    t = os.path.getmtime(obj)
    return any(not os.path.exists(d) or os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and os.path.exists(OUT) and all(os.path.getmtime(d) <= os.path.getmtime(OUT) for d in DEPS):
        return OUT
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    os.makedirs(OBJ_DIR, exist_ok=True)
    procs = []
    objs = []
    for src in SRCS:  # the two units compile in parallel
        obj = os.path.join(OBJ_DIR, os.path.basename(src)[:-4] + '.o')
        objs.append(obj)
        if not force and not _stale(obj):   # an actor experiment leaves the env unit alone and vice versa
            continue
        cmd = [hipcc] + FLAGS + ['-MD', '-MF', obj + '.d', '-c', '-o', obj, src]
        if verbose:
            print(' '.join(cmd))
        procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    link = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', OUT] + objs
    if verbose:
        print(' '.join(link))
    subprocess.check_call(link)
    return OUT


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
