"""CPU: what the compiler made of every kernel in libpworld.so ("no scratch in a rollout kernel's step loop").

Read from the built objects with tools/code_object.py: the AMDGPU metadata notes of the gfx950 code objects (registers, spills,
private segment) and their disassembly (scratch-memory instructions).  The rule, for EVERY kernel of both translation units:
  * no VGPR is spilled (.vgpr_spill_count == 0) -- SGPR spills are fine, they go to lanes of a VGPR, not to memory;
  * no scratch-memory instruction exists anywhere in the kernel (so none can sit in a step loop).
A kernel may still carry a small private segment that nothing accesses (the compiler reserves 36 bytes for some instantiations of the
one-launch policy rollouts and then never touches them): allowed, and bounded here so that a real stack object does not hide there.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import code_object  # noqa: E402

pytestmark = pytest.mark.skipif(not code_object.tools_present(), reason='ROCm LLVM binary tools not installed')


@pytest.fixture(scope='module')
def kernels():
    from multiagent_rl_amd import build_native
    if not all(os.path.exists(o) for o in build_native.objects()):
        build_native.build(force=True)          # a tree that carries only the .so: the objects are rebuilt (2 minutes)
    return code_object.all_kernels()


def test_both_units_are_there(kernels):
    names = ' '.join(kernels)
    for family in ('pw_spread_quad_kernel', 'pw_spread_duo_kernel', 'pw_tag_duo_kernel', 'pw_rollout_kernel', 'pw_reference_rollout_kernel',
                   'pw_policy_rollout3_kernel', 'pw_policy_rollout3j_kernel', 'pw_policy_rollout_tag_kernel',
                   'pw_policy_rollout_ref_kernel', 'pw_actor_fused16_kernel', 'pw_replay_gather_kernel'):
        assert family in names, family
    assert len(kernels) > 150


def test_no_kernel_spills_vector_registers(kernels):
    bad = {n: d['.vgpr_spill_count'] for n, d in kernels.items() if d['.vgpr_spill_count']}
    assert not bad, 'VGPR spills (scratch memory traffic): %r' % bad


def test_no_kernel_executes_a_scratch_instruction(kernels):
    bad = {n: d['scratch_insts'] for n, d in kernels.items() if d['scratch_insts']}
    assert not bad, 'scratch_* instructions in: %r' % bad


def test_private_segments_are_dead_and_tiny(kernels):
    """A private segment without a single scratch instruction is a reservation nothing uses; anything larger than a few words
    would be a real stack object (a run-time indexed local array) on its way back."""
    for n, d in kernels.items():
        assert d['.private_segment_fixed_size'] <= 64, (n, d)
        if d['.private_segment_fixed_size']:
            assert d['scratch_insts'] == 0 and d['.vgpr_spill_count'] == 0, (n, d)


def test_rollout_kernels_fit_two_waves_per_simd(kernels):
    """The 512-thread rollout workgroups need two waves per SIMD: at most 256 registers per lane (arch + accumulation)."""
    for n, d in kernels.items():
        if 'pw_policy_rollout' in n or 'pw_actor_fused16' in n:
            assert d['.vgpr_count'] <= 256, (n, d['.vgpr_count'])
