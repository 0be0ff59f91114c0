"""Builds libpworld.so (HIP, gfx950) in-tree.  Run: python -m multiagent_rl_amd.build_native"""
import hashlib
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


# Objects are named after the flags they were compiled with: an object left behind by a build with other flags (e.g. a tools/
# timing experiment that passed -D switches through HIPCC_EXTRA_FLAGS) is never picked up by the next product build.
def _flags():
    extra = os.environ.get('HIPCC_EXTRA_FLAGS', '').split()
    if any(f.startswith('-DPW_EXP') for f in extra):
        raise SystemExit('build_native: PW_EXPERIMENTS / PW_EXP_* are timing-only overlays (wrong results by design); they are built '
                         'from tools/ (tools/experiments/pw_experiments.hpp), never into libpworld.so')
    return FLAGS + extra


def _flag_tag(flags):
    return hashlib.sha256(' '.join(flags).encode()).hexdigest()[:8]


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


def unit_sources(unit):
    """Every file translation unit `unit` ('pworld' = environment / replay / wire, 'pworld_policy' = actor and policy rollouts) is
    compiled from: the .hip file, the quoted includes it reaches under csrc/, and the two public headers.  Found by reading the
    sources (no compiler, no recorded paths), so it gives the same answer in any copy of the tree."""
    import re
    csrc = os.path.join(HERE, 'csrc')
    seen, todo = [], [os.path.join(csrc, unit + '.hip')]
    while todo:
        f = todo.pop()
        if f in seen or not os.path.exists(f):
            continue
        seen.append(f)
        for inc in re.findall(r'^\s*#\s*include\s+"([^"]+)"', open(f).read(), flags=re.M):
            for d in (os.path.dirname(f), csrc, os.path.join(ROOT, 'include')):
                if os.path.exists(os.path.join(d, inc)):
                    todo.append(os.path.normpath(os.path.join(d, inc)))
                    break
    return sorted(seen)


KERNEL_FAMILIES = {
    # the device code of the environment kernels (pw_spread_* / pw_tag_* / pw_rollout_kernel / pw_reference_*): headers only -- the host
    # side (pworld.hip: dispatch, entry points) and the replay / wire kernels do not change what these kernels execute, and a
    # different dispatch choice shows up as a different kernel NAME, which bench.py compares as well
    'env': ['csrc/pw_common.hpp', 'csrc/pw_kernels_spread.hpp', 'csrc/pw_kernels_spread_quad.hpp', 'csrc/pw_kernels_tag.hpp',
            'csrc/pw_kernels_generic.hpp', 'csrc/pw_kernels_reference.hpp', '../include/pworld_math.h'],
    # the one-launch policy rollouts and the fused actor: the environment arithmetic above + the actor headers
    'policy': ['csrc/pw_common.hpp', 'csrc/pw_kernels_spread.hpp', 'csrc/pw_kernels_tag.hpp', 'csrc/pw_kernels_reference.hpp',
               '../include/pworld_math.h', 'csrc/pw_kernels_actor16.hpp', 'csrc/pw_kernels_policy.hpp', 'csrc/pw_kernels_policy2.hpp',
               'csrc/pw_kernels_policy3.hpp', 'csrc/pw_kernels_policy3j.hpp', 'csrc/pw_kernels_policy_ref.hpp',
               'csrc/pw_kernels_policy_tag.hpp'],
}


def kernel_source_hash(family):
    """sha256 (first 16 hex digits) over the device sources of a kernel family and the compile flags: what a rocprofv3 summary under
    profiles/ records about the kernels it measured (tools/summarize_prof.py), and what bench.py compares before it quotes that
    summary's counters -- a summary collected from other kernel code is refused, not replayed."""
    h = hashlib.sha256(' '.join(FLAGS[:-1]).encode())   # without the -I path (it differs between copies of the tree)
    for rel in KERNEL_FAMILIES[family]:
        h.update(rel.encode())
        h.update(open(os.path.normpath(os.path.join(HERE, rel)), 'rb').read())
    return h.hexdigest()[:16]


def kernel_family(kernel_name):
    return 'policy' if ('pw_policy_rollout' in kernel_name or 'pw_actor_' in kernel_name) else 'env'


def objects():
    """The per-unit objects of the current flags (tools/code_object.py reads the compiler's verdict on every kernel from them)."""
    tag = _flag_tag(_flags())
    return [os.path.join(OBJ_DIR, '%s.%s.o' % (os.path.basename(src)[:-4], tag)) for src in SRCS]


def build(force=False, verbose=False):
    if not force and os.path.exists(OUT) and all(os.path.getmtime(d) <= os.path.getmtime(OUT) for d in DEPS):
        return OUT
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    os.makedirs(OBJ_DIR, exist_ok=True)
    flags = _flags()
    tag = _flag_tag(flags)
    procs = []
    objs = []
    for src in SRCS:  # the two units compile in parallel
        obj = os.path.join(OBJ_DIR, '%s.%s.o' % (os.path.basename(src)[:-4], tag))
        objs.append(obj)
        if not force and not _stale(obj):   # an actor experiment leaves the env unit alone and vice versa
            continue
        cmd = [hipcc] + flags + ['-MD', '-MF', obj + '.d', '-c', '-o', obj, src]
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
