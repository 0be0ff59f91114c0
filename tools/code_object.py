#!/usr/bin/env python3
"""What the compiler made of the kernels in libpworld.so, read from the built objects (multiagent_rl_amd/csrc/_obj/*.o): the gfx950
code object of each translation unit is unbundled from the object's .hip_fatbin section, its AMDGPU metadata notes give
registers / spills / private segment per kernel, its disassembly the number of scratch-memory instructions per kernel.

    python3 tools/code_object.py            # table of every kernel that spills, has a private segment or touches scratch
    python3 tools/code_object.py --all

tests/test_code_object.py holds the rule: no kernel spills a VGPR and no kernel executes a scratch instruction."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = '/opt/rocm/lib/llvm/bin'
FIELDS = ('.vgpr_count', '.agpr_count', '.vgpr_spill_count', '.sgpr_count', '.sgpr_spill_count', '.private_segment_fixed_size',
          '.group_segment_fixed_size')


def tools_present():
    return all(os.path.exists(os.path.join(LLVM, t)) for t in ('llvm-objcopy', 'clang-offload-bundler', 'llvm-readelf', 'llvm-objdump'))


def _demangle(names):
    out = subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True, text=True).stdout.splitlines()
    return dict(zip(names, out)) if len(out) == len(names) else {n: n for n in names}


def code_object(obj, workdir):
    """-> path of the gfx950 code object embedded in a host object compiled by hipcc."""
    base = os.path.join(workdir, os.path.basename(obj))
    subprocess.check_call([os.path.join(LLVM, 'llvm-objcopy'), '--dump-section', '.hip_fatbin=%s.fatbin' % base, obj, os.devnull])
    subprocess.check_call([os.path.join(LLVM, 'clang-offload-bundler'), '--unbundle', '--type=o',
                           '--targets=hipv4-amdgcn-amd-amdhsa--gfx950', '--input=%s.fatbin' % base, '--output=%s.co' % base])
    return base + '.co'


def kernels(co):
    """-> {mangled kernel name: {field: int, 'scratch_insts': int}}"""
    notes = subprocess.run([os.path.join(LLVM, 'llvm-readelf'), '--notes', co], capture_output=True, text=True, check=True).stdout
    out = {}
    # one block per kernel: from its '.agpr_count' (first key of the sorted map after .args) to the next
    for blk in re.split(r'\n\s+- \.agpr_count:', notes)[1:]:
        blk = '\n    .agpr_count:' + blk
        name = re.search(r'\n\s+\.name:\s+(\S+)', blk).group(1)
        out[name] = {f: int(re.search(r'\n\s+%s:\s+(\d+)' % re.escape(f), blk).group(1)) for f in FIELDS}
        out[name]['scratch_insts'] = 0
    dis = subprocess.run([os.path.join(LLVM, 'llvm-objdump'), '-d', '--no-show-raw-insn', co], capture_output=True, text=True,
                         check=True).stdout
    cur = None
    for line in dis.splitlines():
        m = re.match(r'^[0-9a-f]+ <(\S+)>:', line)
        if m:
            cur = m.group(1) if m.group(1) in out else None
        elif cur and ('scratch_' in line):
            out[cur]['scratch_insts'] += 1
    return out


def all_kernels():
    from multiagent_rl_amd import build_native
    objs = build_native.objects()
    res = {}
    with tempfile.TemporaryDirectory() as wd:
        for o in objs:
            res.update(kernels(code_object(o, wd)))
    names = _demangle(list(res))
    return {names[k].replace('(anonymous namespace)::', ''): v for k, v in res.items()}


if __name__ == '__main__':
    sys.path.insert(0, ROOT)
    ks = all_kernels()
    print('%-86s %5s %5s %6s %6s %5s %7s' % ('kernel', 'vgpr', 'vspill', 'sspill', 'priv B', 'LDS', 'scratch'))
    for n, d in sorted(ks.items()):
        if '--all' in sys.argv or d['.vgpr_spill_count'] or d['.private_segment_fixed_size'] or d['scratch_insts']:
            print('%-86s %5d %5d %6d %6d %5d %7d' % (n.split('(')[0][:86], d['.vgpr_count'], d['.vgpr_spill_count'], d['.sgpr_spill_count'],
                                                     d['.private_segment_fixed_size'], d['.group_segment_fixed_size'], d['scratch_insts']))
    print('%d kernels' % len(ks))
