"""Test infrastructure (never imported by the product): torch restatements of the HIP launches of
multiagent_rl_amd.dist, so that the exchange choreography can run with the gloo backend on CPU tensors
(tests/test_dist_gloo.py, tests/test_bench_launcher.py via ``PW_BENCH_STUB=1``).  The GPU parity of the real
launches against these same functions is in tests/test_gpu_engine.py."""
import torch

from multiagent_rl_amd.dist import FullTransitionGather, SampledTransitionGather


class HostStateRing(object):
    """The learner rank's STATE ring on the host: per transition {vel, pos} before / after + the episode's landmarks; ``sample_index``
    rebuilds the rows (``pw_replay_gather`` on a state ring)."""

    def __init__(self, scenario='simple_spread', num_adversaries=0):
        self.scenario, self.A, self.transitions = scenario, num_adversaries, []

    def __len__(self):
        return sum(t['rew'].shape[0] for t in self.transitions)

    def clear(self):
        self.transitions = []

    def append(self, tr):
        self.transitions.append({k: tr[k] for k in ('state', 'next_state', 'lm', 'act', 'rew', 'done')})

    def sample_index(self, idx):
        cat = {k: torch.cat([t[k] for t in self.transitions]) for k in ('state', 'next_state', 'lm', 'act', 'rew', 'done')}
        i = torch.as_tensor(idx, dtype=torch.long)
        return (rows_from_state(cat['state'][i], cat['lm'][i], self.scenario, self.A),
                torch.nn.functional.one_hot(cat['act'][i].long(), 5).float(), cat['rew'][i],
                rows_from_state(cat['next_state'][i], cat['lm'][i], self.scenario, self.A), cat['done'][i])


class HostRing(object):
    """Tuple ring in the reference's order (rls/replay_buffer.py:30-37), unbounded: what the root ingested."""

    def __init__(self):
        self.rows = []          # SampledTransitionGather: packed row blocks
        self.transitions = []   # FullTransitionGather: dicts of [T*B, ...] tensors, one per ingested block

    def __len__(self):
        return sum(t['rew'].shape[0] for t in self.transitions) + sum(r.shape[0] for r in self.rows)

    def clear(self):
        self.rows, self.transitions = [], []


def pack_reference(out, actions, sel_t, sel_e):
    """Row layout of include/pworld.h: [obs ND | next_obs ND | act N | rew | done]."""
    t, e = sel_t.long(), sel_e.long()
    obs = out['obs'][t - 1, e].reshape(len(t), -1)
    nxt = torch.where(out['terminal'][t, e].bool()[:, None, None], out['final_obs'][t, e], out['obs'][t, e])
    return torch.cat([obs, nxt.reshape(len(t), -1), actions[t, e].float(), out['rew_shared'][t, e][:, None],
                      torch.zeros(len(t), 1)], dim=1)


class CpuSampledGather(SampledTransitionGather):
    def _make_memory(self):
        return HostRing()

    def _pack(self, out, actions, sel_t, sel_e, rows):
        rows.copy_(pack_reference(out, actions, sel_t, sel_e))

    def _ingest(self, rows):
        self.memory.rows.append(rows.clone())


def wire_finalize_reference(g, block, obs0):
    """pw_chunk_wire_finalize: obs0, the k-th episode end's pre-reset row per env, byte actions, episode-end map."""
    v = g.views(block)
    T, B, F = g.T, g.B, g.lay.F
    v['obs0'].copy_(obs0)
    term, fin = g.side['terminal'].bool(), g.side['final_obs']
    k = torch.zeros(B, dtype=torch.long)
    for t in range(T):
        m = term[t] & (k < F) if fin is not None else torch.zeros(B, dtype=torch.bool)
        v['fin_slot'][t] = torch.where(m, k, torch.full_like(k, 255)).to(torch.uint8)
        e = torch.nonzero(m).flatten()
        if e.numel():
            v['final_rows'][k[e], e] = fin[t, e]
        k = k + m.long()
    v['act'].copy_(g.side['act'].to(torch.uint8))


def wire_transitions_reference(g, block):
    """pw_replay_add_wire: the block's T*B transitions in (t, e) order as the reference's tuple fields."""
    v = g.views(block)
    T, B, N, D = g.T, g.B, g.N, g.D
    obs = torch.cat([v['obs0'][None], v['obs'][:-1]], 0)
    fs = v['fin_slot'].long()
    nxt = v['obs'].clone()
    t, e = torch.nonzero(fs != 255, as_tuple=True)
    if t.numel():
        nxt[t, e] = v['final_rows'][fs[t, e], e]
    return dict(obs=obs.reshape(T * B, N, D).clone(), next_obs=nxt.reshape(T * B, N, D),
                act=v['act'].reshape(T * B, N).clone(), rew=v['rew_shared'].reshape(T * B).clone(),
                done=torch.zeros(T * B))


# ---- state-only wire blocks (include/pworld.h pw_state_wire): simple_spread, local observation --------------------------
def rows_from_state(state, lm, scenario='simple_spread', num_adversaries=0):
    """The observation rows of agents with ``state`` [..., N, 4] = {vx, vy, px, py} among landmarks ``lm`` [..., L, 2], float32, one
    subtraction per entry.  simple_spread, local observation (experiments/scenarios.py:6-20): [vel, pos, lm_0 - pos, lm_1 - pos, ...].
    simple_tag (upstream simple_tag.observation): the same, then [pos_j - pos for j != a], then [vel_j for the GOOD agents j != a],
    zero-padded to the adversaries' width 4 + 2L + 2(N - 1) + 2(N - A)."""
    pos = state[..., 2:4]
    rel = lm[..., None, :, :] - pos[..., :, None, :]                      # [..., N, L, 2]
    rows = torch.cat([state, rel.reshape(*rel.shape[:-2], -1)], dim=-1)
    if scenario == 'simple_spread':
        return rows
    N, A = state.shape[-2], num_adversaries
    D = rows.shape[-1] + 2 * (N - 1) + 2 * (N - A)
    out = torch.zeros(*state.shape[:-1], D, dtype=state.dtype)
    out[..., :rows.shape[-1]] = rows
    for a in range(N):
        k = rows.shape[-1]
        for j in range(N):
            if j != a:
                out[..., a, k:k + 2] = pos[..., j, :] - pos[..., a, :]
                k += 2
        for j in range(A, N):
            if j != a:
                out[..., a, k:k + 2] = state[..., j, 0:2]
                k += 2
    return out


def state_wire_begin_reference(g, block, state0, lm0, ep0):
    """pw_state_wire_begin: the chunk's start, from the env's state planes."""
    v = g.views(block)
    v['state0'].copy_(state0)
    v['lm'][0].copy_(lm0)
    v['ep0'].copy_(ep0.to(torch.int32))


def state_wire_finalize_reference(g, block, landmarks_of_episode):
    """pw_state_wire_finalize: columns 0..3 of every row, the k-th episode end's pre-reset state, the landmarks the k-th reset
    drew (``landmarks_of_episode(env_indices, episode_numbers) -> [n, L, 2]``: the reset's own generator), byte actions, and per
    step: episode ends before it (bits 0..6) | this step ended one (bit 7)."""
    v = g.views(block)
    T, B, F = g.T, g.B, g.lay.F
    v['state'].copy_(g.side['obs'][..., :4])
    term, fin = g.side['terminal'].bool(), g.side['final_obs']
    ep0 = v['ep0'].long()
    k = torch.zeros(B, dtype=torch.long)
    for t in range(T):
        m = term[t] & (k < F) if fin is not None else torch.zeros(B, dtype=torch.bool)
        v['epi'][t] = (k + 128 * m.long()).to(torch.uint8)
        e = torch.nonzero(m).flatten()
        if e.numel():
            v['final_state'][k[e], e] = fin[t, e][..., :4]
            v['lm'][k[e] + 1, e] = landmarks_of_episode(e, ep0[e] + k[e] + 1)
        k = k + m.long()
    v['act'].copy_(g.side['act'].to(torch.uint8))


def state_wire_transitions_reference(g, block):
    """pw_replay_add_state_wire: the block's T*B transitions in (t, e) order with the rows REBUILT from the states."""
    v = g.views(block)
    T, B, N, D = g.T, g.B, g.N, g.D
    epi = v['epi'].long()
    k, ended = epi & 127, (epi & 128) != 0
    e_idx = torch.arange(B)[None, :].expand(T, B)
    lm = v['lm'][k, e_idx]                                                # [T, B, L, 2]: the episode in progress at step t
    s_obs = torch.cat([v['state0'][None], v['state'][:-1]], 0)
    s_next = v['state'].clone()
    t, e = torch.nonzero(ended, as_tuple=True)
    if t.numel():
        s_next[t, e] = v['final_state'][k[t, e], e]
    scen, A = getattr(g, 'scenario', 'simple_spread'), getattr(g, 'A', 0)
    return dict(obs=rows_from_state(s_obs, lm, scen, A).reshape(T * B, N, D), next_obs=rows_from_state(s_next, lm, scen, A).reshape(T * B, N, D),
                act=v['act'].reshape(T * B, N).clone(), rew=v['rew_shared'].reshape(T * B).clone(), done=torch.zeros(T * B),
                # what a STATE ring keeps of the same transitions (pw_replay_add_state_wire into pw_replay_store.state_rows)
                state=s_obs.reshape(T * B, N, 4).clone(), next_state=s_next.reshape(T * B, N, 4).clone(),
                lm=lm.reshape(T * B, lm.shape[-2], 2).clone())


# ---- compact-row wire blocks of simple_reference (include/pworld.h pw_ref_wire) ---------------------------------------------
REF_COLOURS = torch.tensor([[0.75, 0.25, 0.25], [0.25, 0.75, 0.25], [0.25, 0.25, 0.75]])


def ref_rows(head, goal, symbol):
    """The 21-number rows of simple_reference (experiments/scenarios.py:23-42) from their three parts: ``head`` [..., 2, 8] =
    [p_vel, landmark - p_pos x 3] as the env wrote it, ``goal`` [..., 2] the agent's goal landmark (0.75 on its colour channel),
    ``symbol`` [..., 2] the symbol the OTHER agent emitted (255: none -> zeros)."""
    sym = symbol.long()
    onehot = torch.nn.functional.one_hot(sym.clamp(max=9), 10).float() * (sym != 255)[..., None].float()
    return torch.cat([head, REF_COLOURS[goal.long()], onehot], dim=-1)


def ref_wire_finalize_reference(g, block, obs0):
    """pw_ref_wire_finalize: heads of every row, the chunk's start (head, goal, visible symbol), per episode end the pre-reset head and
    the goal the reset drew (read off the post-reset row), both action heads as bytes, the episode map."""
    v = g.views(block)
    T, B, F = g.T, g.B, g.lay.F
    obs, fin, term = g.side['obs'], g.side['final_obs'], g.side['terminal'].bool()
    v['head0'].copy_(obs0[..., :8])
    v['goal'][0].copy_(obs0[..., 8:11].argmax(-1).to(torch.uint8))
    c0 = obs0[..., 11:]
    v['comm0'].copy_(torch.where(c0.any(-1), c0.argmax(-1), torch.full_like(c0.argmax(-1), 255)).to(torch.uint8))
    v['head'].copy_(obs[..., :8])
    k = torch.zeros(B, dtype=torch.long)
    for t in range(T):
        m = term[t] & (k < F) if fin is not None else torch.zeros(B, dtype=torch.bool)
        v['epi'][t] = (k + 128 * m.long()).to(torch.uint8)
        e = torch.nonzero(m).flatten()
        if e.numel():
            v['final_head'][k[e], e] = fin[t, e][..., :8]
            v['goal'][k[e] + 1, e] = obs[t, e][..., 8:11].argmax(-1).to(torch.uint8)
        k = k + m.long()
    v['act'].copy_(g.side['act'].to(torch.uint8))


def ref_wire_transitions_reference(g, block):
    """pw_replay_add_ref_wire: the block's T*B transitions in (t, e) order, rows rebuilt; act [T*B, 2, 2] = (movement, symbol)."""
    v = g.views(block)
    T, B = g.T, g.B
    epi = v['epi'].long()
    k, ended = epi & 127, (epi & 128) != 0
    e_idx = torch.arange(B)[None, :].expand(T, B)
    goal = v['goal'][k, e_idx]                                            # [T, B, 2]: the episode in progress at step t
    head_obs = torch.cat([v['head0'][None], v['head'][:-1]], 0)
    head_next = v['head'].clone()
    t, e = torch.nonzero(ended, as_tuple=True)
    if t.numel():
        head_next[t, e] = v['final_head'][k[t, e], e]
    sym_now = v['act'][..., 1].flip(-1)                                   # [T, B, 2]: what agent a sees = the OTHER agent's symbol
    prev_ended = torch.cat([torch.zeros(1, B, dtype=torch.bool), ended[:-1]], 0)
    sym_prev = torch.cat([v['comm0'][None], sym_now[:-1]], 0)
    sym_prev = torch.where(prev_ended[:, :, None], torch.full_like(sym_prev, 255), sym_prev)
    return dict(obs=ref_rows(head_obs, goal, sym_prev).reshape(T * B, 2, 21), next_obs=ref_rows(head_next, goal, sym_now).reshape(T * B, 2, 21),
                act=v['act'].reshape(T * B, 2, 2).clone(), rew=v['rew_shared'].reshape(T * B).clone(), done=torch.zeros(T * B))


class CpuFullGather(FullTransitionGather):
    def _layout(self, PwChunkWire):
        # the layout arithmetic is host code of libpworld (no GPU needed)
        return super()._layout(PwChunkWire)

    def _make_memory(self):
        if self.ref_wire:
            return HostRing()
        if getattr(self, 'ring_kind', 'rows') == 'state':
            return HostStateRing(self.scenario, self.A)
        return HostRing()

    def _begin(self, block):
        state_wire_begin_reference(self, block, *self.env.wire_start())     # the stub env's (state0, landmarks, episode numbers)

    def _finalize(self, block, obs0):
        if self.ref_wire:
            ref_wire_finalize_reference(self, block, obs0)
        elif self.state_wire:
            state_wire_finalize_reference(self, block, self.env.landmarks_of_episode)
        else:
            wire_finalize_reference(self, block, obs0)

    def _ingest(self, block):
        tr = ref_wire_transitions_reference(self, block) if self.ref_wire else \
            state_wire_transitions_reference(self, block) if self.state_wire else wire_transitions_reference(self, block)
        if isinstance(self.memory, HostStateRing):
            self.memory.append(tr)
        else:
            self.memory.transitions.append(tr)
