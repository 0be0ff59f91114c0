"""CPU: libpworld.so loads and exports every symbol include/pworld.h declares; host-side argument
checking of the C ABI (no compute launches without a GPU)."""
import ctypes as C
import os
import re

import pytest

from multiagent_rl_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, 'include', 'pworld.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(pw_[a-z_0-9]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), 'libpworld.so lacks %s declared in include/pworld.h' % n
    assert set(names) == set(_lib.SIGNATURES), 'ctypes table and header disagree: %s' % (
        set(names) ^ set(_lib.SIGNATURES))
    import re
    want = int(re.search(r'#define\s+PW_VERSION\s+(\d+)', open(os.path.join(ROOT, 'include', 'pworld.h')).read()).group(1))
    assert lib.pw_version() == want      # the library and the header it was built against agree


def test_config_default_matches_canonical_constants():
    lib = _lib.load()
    cfg = _lib.PwConfig()
    assert lib.pw_config_default(C.byref(cfg), _lib.PW_SIMPLE_SPREAD, 4096, 6, -1, 0) == 0
    assert cfg.struct_size == C.sizeof(_lib.PwConfig)
    assert (cfg.num_agents, cfg.num_landmarks, cfg.max_episode_len, cfg.seed) == (6, 6, 25, 12345678)
    assert abs(cfg.dt - 0.1) < 1e-7 and cfg.damping == 0.25 and cfg.contact_force == 100.0
    assert abs(cfg.contact_margin - 1e-3) < 1e-9 and cfg.default_sensitivity == 5.0
    assert abs(cfg.agent_size[0] - 0.15) < 1e-7 and cfg.agent_accel[0] < 0 and cfg.landmark_collide == 0
    assert lib.pw_config_default(C.byref(cfg), _lib.PW_SIMPLE_TAG, 8192, 6, -1, 4) == 0
    assert (cfg.num_landmarks, cfg.num_adversaries, cfg.landmark_collide) == (2, 4, 1)
    assert abs(cfg.agent_size[0] - 0.075) < 1e-7 and abs(cfg.agent_size[5] - 0.05) < 1e-7
    assert (cfg.agent_accel[0], cfg.agent_accel[5]) == (3.0, 4.0)
    assert abs(cfg.agent_max_speed[5] - 1.3) < 1e-6 and abs(cfg.landmark_size - 0.2) < 1e-7


def test_handle_geometry_and_errors():
    lib = _lib.load()
    cfg = _lib.PwConfig()
    lib.pw_config_default(C.byref(cfg), _lib.PW_SIMPLE_SPREAD, 4096, 6, -1, 0)
    h = C.c_void_p()
    assert lib.pw_create(C.byref(cfg), C.byref(h)) == 0
    assert lib.pw_obs_dim(h) == 16
    assert lib.pw_algorithmic_bytes_per_env_step(h) == 678          # 57N + 8L + 8NL
    lay = _lib.PwStateLayout()
    assert lib.pw_get_state_layout(h, C.byref(lay)) == 0
    assert lay.total_bytes == lib.pw_state_bytes(h) and lay.pos_y - lay.pos_x >= 4096 * 6 * 4
    assert all(getattr(lay, f) % 256 == 0 for f, _ in lay._fields_)
    # stepping before a state block is bound is an error, not a crash
    io = _lib.PwStepIO()
    assert lib.pw_step(h, C.byref(io), None) == -2 and b'not bound' in lib.pw_last_error()
    assert lib.pw_bind_state(h, C.c_void_p(0x1001)) == -1 and b'aligned' in lib.pw_last_error()
    lib.pw_destroy(h)
    for n, want in [(3, 267), (12, 1932), (24, 6168), (48, 21552)]:
        lib.pw_config_default(C.byref(cfg), _lib.PW_SIMPLE_SPREAD, 4096, n, -1, 0)
        assert lib.pw_create(C.byref(cfg), C.byref(h)) == 0
        assert lib.pw_algorithmic_bytes_per_env_step(h) == want
        lib.pw_destroy(h)
    # bad configurations
    lib.pw_config_default(C.byref(cfg), _lib.PW_SIMPLE_SPREAD, 16, 6, -1, 0)
    cfg.num_agents = 65
    assert lib.pw_create(C.byref(cfg), C.byref(h)) == -1 and b'num_agents' in lib.pw_last_error()
    cfg.num_agents, cfg.struct_size = 6, 12
    assert lib.pw_create(C.byref(cfg), C.byref(h)) == -1 and b'struct_size' in lib.pw_last_error()
    assert lib.pw_config_default(C.byref(cfg), 7, 16, 6, -1, 0) == -1 and b'scenario' in lib.pw_last_error()


def test_actor_precision_is_a_handle_property_outside_the_dispatch(monkeypatch):
    """pw_set_actor_precision / pw_get_actor_precision: exact float32 by default, the opt-in bf16x3 mode by call ONLY (no
    environment variable may change results: PW_ACTOR_BF16X3=1 in the creating process is ignored), other values refused -- and
    it is NOT a pw_dispatch field (the dispatch never changes results; this switch does).  pw_actor_set_bf16x3 returns the
    previous process-wide value."""
    lib = _lib.load()
    monkeypatch.delenv('PW_ACTOR_BF16X3', raising=False)
    cfg = _lib.PwConfig()
    lib.pw_config_default(C.byref(cfg), _lib.PW_SIMPLE_SPREAD, 64, 6, -1, 0)
    h, h2 = C.c_void_p(), C.c_void_p()
    assert lib.pw_create(C.byref(cfg), C.byref(h)) == 0
    assert lib.pw_get_actor_precision(h) == 0
    assert lib.pw_set_actor_precision(h, 1) == 0 and lib.pw_get_actor_precision(h) == 1
    assert lib.pw_set_actor_precision(h, 2) == -1 and b'PW_ACTOR' in lib.pw_last_error()
    assert lib.pw_get_actor_precision(h) == 1
    assert lib.pw_set_actor_precision(h, 0) == 0 and lib.pw_get_actor_precision(h) == 0
    monkeypatch.setenv('PW_ACTOR_BF16X3', '1')
    assert lib.pw_create(C.byref(cfg), C.byref(h2)) == 0
    assert lib.pw_get_actor_precision(h2) == 0 and lib.pw_get_actor_precision(h) == 0   # the environment does not reach it
    assert 'actor' not in ' '.join(n for n, _ in _lib.PwDispatch._fields_)
    assert lib.pw_actor_set_bf16x3(0) == 0 and lib.pw_actor_set_bf16x3(1) == 0 and lib.pw_actor_set_bf16x3(0) == 1
    lib.pw_destroy(h)
    lib.pw_destroy(h2)


def test_dispatch_lives_in_the_handle(monkeypatch):
    """pw_dispatch (kernel selection): defaults, the one-time overlay of PWORLD_* at pw_create, set / get, range checks --
    and the launch path of csrc/pworld.hip contains no getenv at all (the only reads are in dispatch_from_environment)."""
    lib = _lib.load()
    names = [n for n, _ in _lib.PwDispatch._fields_ if n != 'struct_size']
    auto = dict(force_generic=0, no_stream=0, duo=-1, quad=-1, obs_block=-1, trio=-1, p_prio=-1, envs_per_wave=0, policy_form=0)
    d = _lib.PwDispatch()
    assert lib.pw_dispatch_default(C.byref(d)) == 0 and d.struct_size == C.sizeof(_lib.PwDispatch)
    assert {n: getattr(d, n) for n in names} == auto
    for k in [k for k in os.environ if k.startswith('PWORLD_')]:
        monkeypatch.delenv(k)
    cfg = _lib.PwConfig()
    lib.pw_config_default(C.byref(cfg), _lib.PW_SIMPLE_SPREAD, 4096, 6, -1, 0)
    h, h2 = C.c_void_p(), C.c_void_p()
    assert lib.pw_create(C.byref(cfg), C.byref(h)) == 0
    got = _lib.PwDispatch()
    assert lib.pw_get_dispatch(h, C.byref(got)) == 0 and {n: getattr(got, n) for n in names} == auto
    monkeypatch.setenv('PWORLD_NO_QUAD', '1')
    monkeypatch.setenv('PWORLD_OBS_BLOCK', '1')
    monkeypatch.setenv('PWORLD_POLICY_V3J', '1')
    monkeypatch.setenv('PWORLD_POLICY_V2', '1')                    # retired with forms 1 / 2: no longer read
    assert lib.pw_create(C.byref(cfg), C.byref(h2)) == 0          # read once, here
    monkeypatch.delenv('PWORLD_NO_QUAD')
    lib.pw_get_dispatch(h2, C.byref(got))
    assert {n: getattr(got, n) for n in names} == dict(auto, quad=0, obs_block=1, policy_form=4)
    lib.pw_get_dispatch(h, C.byref(got))
    assert {n: getattr(got, n) for n in names} == auto          # the older handle is untouched
    got.duo, got.envs_per_wave = 0, 5
    assert lib.pw_set_dispatch(h, C.byref(got)) == 0
    back = _lib.PwDispatch()
    lib.pw_get_dispatch(h, C.byref(back))
    assert (back.duo, back.envs_per_wave) == (0, 5)
    got.trio = 2
    assert lib.pw_set_dispatch(h, C.byref(got)) == -1 and b'out of range' in lib.pw_last_error()
    got.trio, got.struct_size = -1, 8
    assert lib.pw_set_dispatch(h, C.byref(got)) == -1 and b'struct_size' in lib.pw_last_error()
    lib.pw_destroy(h)
    lib.pw_destroy(h2)
    src = open(os.path.join(ROOT, 'multiagent_rl_amd', 'csrc', 'pworld.hip')).read()
    body = src[src.index('void dispatch_from_environment'):]
    body = body[:body.index('\n}\n')]
    assert src.count('getenv') == body.count('getenv') > 0, 'getenv outside dispatch_from_environment'
    for f in sorted(os.listdir(os.path.join(ROOT, 'multiagent_rl_amd', 'csrc'))):
        if f != 'pworld.hip' and os.path.isfile(os.path.join(ROOT, 'multiagent_rl_amd', 'csrc', f)):
            assert 'getenv' not in open(os.path.join(ROOT, 'multiagent_rl_amd', 'csrc', f)).read(), f


def test_product_path_fails_loudly_without_gpu_or_library(monkeypatch, tmp_path):
    import torch
    from multiagent_rl_amd.env import BatchedParticleEnv
    if not torch.cuda.is_available():
        with pytest.raises(_lib.PworldError, match='no CPU fallback'):
            BatchedParticleEnv('simple_spread', 4, num_agents=3)
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', str(tmp_path / 'libpworld.so'))
    with pytest.raises(_lib.PworldError, match='HIP extension is required'):
        _lib.load()


def test_product_never_imports_the_oracle():
    """The product path may not import, link or execute anything under oracle/."""
    pat = re.compile(r'(^|\n)\s*(from|import)\s+oracle\b|oracle[/\\.]c_oracle|libpworld_oracle|particle_oracle|'
                     r'pworld_oracle|["\']oracle["\']')
    pkg = os.path.join(ROOT, 'multiagent_rl_amd')
    seen = 0
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h')):
                seen += 1
                m = pat.search(open(os.path.join(d, f)).read())
                assert m is None, '%s references the oracle (%r): the product must not depend on it' % (f, m.group(0))
    assert seen >= 8


def test_header_is_plain_c_and_the_c_host_example_links(tmp_path):
    """include/pworld.h must be consumable by a C compiler (it is the FFI surface), and examples/c_host.c -- a
    caller with no Python and no PyTorch -- must compile warning-free as C11 and link against libpworld.so.
    (Run on a GPU box: ./examples/c_host.bin > out.txt && python tools/check_c_host.py out.txt -- the checker
    compares every printed value with the CPU oracle, bit for bit.)"""
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if shutil.which('gcc') is None or not os.path.isdir('/opt/rocm/include'):
        pytest.skip('needs gcc and the ROCm headers')
    from multiagent_rl_amd import _lib
    assert os.path.exists(_lib.LIB_PATH)
    out = str(tmp_path / 'c_host')
    cmd = ['gcc', '-std=c11', '-Wall', '-Wextra', '-Werror', '-D__HIP_PLATFORM_AMD__', '-I', os.path.join(root, 'include'),
           '-I', '/opt/rocm/include', os.path.join(root, 'examples', 'c_host.c'), '-L', os.path.dirname(_lib.LIB_PATH),
           '-lpworld', '-L', '/opt/rocm/lib', '-lamdhip64', '-o', out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    hdr = str(tmp_path / 'only_header.c')
    open(hdr, 'w').write('#include "pworld.h"\n#include "pworld_math.h"\nint main(void) { return pw_exp(0.0f) == 1.0f ? 0 : 1; }\n')
    r = subprocess.run(['gcc', '-std=c99', '-Wall', '-Wextra', '-Werror', '-pedantic', '-I', os.path.join(root, 'include'),
                        '-c', hdr, '-o', str(tmp_path / 'h.o')], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_integration_doc_stub_matches_the_abi_structs():
    """INTEGRATION.md shows a hand-written ctypes binding: its struct field lists must stay in step with
    include/pworld.h (as mirrored, and size-checked against the library, by multiagent_rl_amd/_lib.py)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    doc = open(os.path.join(root, 'INTEGRATION.md')).read()
    cfg_block = doc[doc.index('class pw_config(C.Structure)'):doc.index('class pw_step_io(C.Structure)')]
    doc_cfg = re.findall(r"\('(\w+)', C\.", cfg_block)
    assert doc_cfg == [f[0] for f in _lib.PwConfig._fields_]
    io_block = doc[doc.index('class pw_step_io(C.Structure)'):doc.index('cfg, h = pw_config()')]
    doc_io = re.findall(r"'(\w+)'", io_block[io_block.index('for n in'):])
    assert doc_io == [f[0] for f in _lib.PwStepIO._fields_]
    # every entry point the doc's table names exists in the header
    hdr = open(os.path.join(root, 'include', 'pworld.h')).read()
    for name in set(re.findall(r'`(pw_[a-z_0-9]+)`', doc)):
        assert re.search(r'\b%s\b' % name, hdr), name


def test_wire_layouts_of_every_scenario_at_baseline_sizes():
    """Host arithmetic of the wire blocks (no GPU): bytes per env-step at the BASELINE shapes -- simple_spread C2 114 B (row block 414),
    simple_tag C3 (4 + 2, L = 2, B = 8192) <= 150 B (row block 539), simple_reference <= 80 B (row block 181) -- and what the layouts refuse."""
    lib = _lib.load()
    T = 100
    sw = _lib.PwStateWire()
    assert lib.pw_state_wire_layout_scn(_lib.PW_SIMPLE_SPREAD, T, 4096, 6, 6, 0, 25, C.byref(sw)) == 0
    assert (sw.D, sw.F, sw.scenario, sw.num_adversaries) == (16, 4, _lib.PW_SIMPLE_SPREAD, 0) and 113.0 < sw.total_bytes / (T * 4096.0) < 116.0
    old = _lib.PwStateWire()
    assert lib.pw_state_wire_layout(T, 4096, 6, 6, 25, C.byref(old)) == 0 and bytes(old) == bytes(sw)       # the 0.1.5 entry point = the spread case
    assert lib.pw_state_wire_layout_scn(_lib.PW_SIMPLE_TAG, T, 8192, 6, 2, 4, 25, C.byref(sw)) == 0
    assert (sw.D, sw.F, sw.scenario, sw.num_adversaries) == (22, 4, _lib.PW_SIMPLE_TAG, 4)
    tag_bytes = sw.total_bytes / (T * 8192.0)
    cw = _lib.PwChunkWire()
    assert lib.pw_chunk_wire_layout(T, 8192, 6, 22, 25, C.byref(cw)) == 0
    assert tag_bytes <= 150.0 and tag_bytes < 0.25 * cw.total_bytes / (T * 8192.0)
    assert lib.pw_state_wire_layout_scn(_lib.PW_SIMPLE_TAG, T, 8192, 6, 2, 7, 25, C.byref(sw)) < 0            # more adversaries than agents
    assert lib.pw_state_wire_layout_scn(_lib.PW_SIMPLE_REFERENCE, T, 4096, 2, 3, 0, 25, C.byref(sw)) < 0     # its own compact-row layout
    rw = _lib.PwRefWire()
    assert lib.pw_ref_wire_layout(T, 4096, 25, C.byref(rw)) == 0 and rw.F == 4 and rw.total_bytes % 256 == 0
    ref_bytes = rw.total_bytes / (T * 4096.0)
    assert lib.pw_chunk_wire_layout(T, 4096, 2, 21, 25, C.byref(cw)) == 0
    assert ref_bytes <= 80.0 and ref_bytes < 0.45 * cw.total_bytes / (T * 4096.0)
    assert lib.pw_ref_wire_layout(1000, 8, 2, C.byref(rw)) < 0 and b'126' in lib.pw_last_error()


def test_state_ring_arguments_are_checked_on_the_host():
    """A STATE ring (pw_replay_store.state_rows) is filled by pw_replay_add_state_wire and read by pw_replay_gather only; every other writer and
    every inconsistent shape is refused before anything is launched (no GPU needed)."""
    lib = _lib.load()
    st = _lib.PwReplayStore()
    st.obs, st.next_obs, st.rew, st.done, st.act, st.lm = 4096, 8192, 12288, 16384, 20480, 24576      # fake, aligned "device pointers"
    st.capacity, st.num_agents, st.obs_dim = 1000, 6, 16
    st.state_rows, st.num_landmarks, st.scenario, st.num_adversaries = 1, 6, _lib.PW_SIMPLE_SPREAD, 0
    p = C.c_void_p(4096)
    assert lib.pw_replay_add(C.byref(st), 0, None, 4, p, p, p, p, None, None, None, None) < 0 and b'STATE ring' in lib.pw_last_error()
    io = _lib.PwStepIO()
    io.obs = io.rew_shared = io.terminal = 4096
    assert lib.pw_replay_add_rollout(C.byref(st), 0, 4, 2, p, C.byref(io), p, None, None, None, None, None) < 0
    assert b'STATE ring' in lib.pw_last_error()
    assert lib.pw_replay_add_packed(C.byref(st), 0, 4, p, None) < 0 and b'STATE ring' in lib.pw_last_error()
    idx = C.c_void_p(4096)
    st.obs_dim = 18                                                        # not 4 + 2L
    assert lib.pw_replay_gather(C.byref(st), idx, 4, p, p, p, p, p, None) < 0 and b'STATE ring' in lib.pw_last_error()
    st.obs_dim, st.act_heads = 16, 2
    assert lib.pw_replay_gather(C.byref(st), idx, 4, p, p, p, p, p, None) < 0
    st.act_heads, st.scenario, st.num_adversaries, st.num_landmarks, st.obs_dim = 0, _lib.PW_SIMPLE_TAG, 4, 2, 22
    sw = _lib.PwStateWire()
    assert lib.pw_state_wire_layout_scn(_lib.PW_SIMPLE_SPREAD, 10, 8, 6, 6, 0, 25, C.byref(sw)) == 0
    assert lib.pw_replay_add_state_wire(C.byref(st), 0, C.byref(sw), C.c_void_p(1 << 20), None) < 0          # a spread block into a tag ring
