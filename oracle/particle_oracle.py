"""CPU ORACLE (test infrastructure, NOT product code) -- scalar NumPy float64
restatement of the particle world the reference drives.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  The product path (``multiagent_rl_amd``) never
does; it fails loudly when the HIP library is missing.

PARITY UNPINNED.  The arithmetic below is not in ``/root/reference``: the
reference only *imports* it from the third-party ``multiagent`` package
(OpenAI multiagent-particle-envs lineage; no version pin, not installed, not
fetchable -- SURVEY.md section 0 / 8(c)).  This file restates the canonical,
published algorithm (``multiagent/core.py``, ``environment.py``,
``scenarios/simple_spread.py``, ``scenarios/simple_tag.py``) from its
documented semantics and anchors on the reference's own
call sites:

* ``experiments/scenarios.py:6-20``    local observation layout (R3)
* ``experiments/scenarios.py:124-192`` make_env: collaborative=False,
  force_discrete_action=True, ``make_world(num_agents=n)`` (R2)
* ``experiments/run.py:28,44,60``      reset/step call sites (R1)
* ``main.py:41-49``                    seed protocol: global NumPy stream (R7)

It is pinned by hand-derivable known-answer tests (tests/test_oracle_kat.py)
and by the version-stable legacy NumPy MT19937 stream, nothing else.

Loop structure deliberately mirrors upstream (E^2 pair loop calling a per-pair
function, reward recomputed per agent) so that (a) the summation order of
forces is the reference's and (b) timing it is a fair "reference-style
Python/NumPy step" CPU baseline.
"""
import numpy as np

# ----------------------------------------------------------------------------
# Philox4x32-10 counter RNG (public algorithm: Salmon et al., SC'11).  Used
# only by the *batched* device reset; restated here so tests can reproduce the
# kernel's initial states bit for bit.  Layout of counter/key is the build's
# own convention (include/pworld.h, "Reset RNG").
# ----------------------------------------------------------------------------
_PHILOX_M0 = 0xD2511F53
_PHILOX_M1 = 0xCD9E8D57
_PHILOX_W0 = 0x9E3779B9
_PHILOX_W1 = 0xBB67AE85
_MASK32 = 0xFFFFFFFF


def philox4x32_10(counter, key):
    """counter: 4 ints, key: 2 ints -> 4 uint32 as python ints."""
    c0, c1, c2, c3 = [int(c) & _MASK32 for c in counter]
    k0, k1 = [int(k) & _MASK32 for k in key]
    for _ in range(10):
        p0 = _PHILOX_M0 * c0
        p1 = _PHILOX_M1 * c2
        hi0, lo0 = p0 >> 32, p0 & _MASK32
        hi1, lo1 = p1 >> 32, p1 & _MASK32
        c0, c1, c2, c3 = (hi1 ^ c1 ^ k0) & _MASK32, lo1, (hi0 ^ c3 ^ k1) & _MASK32, lo0
        k0 = (k0 + _PHILOX_W0) & _MASK32
        k1 = (k1 + _PHILOX_W1) & _MASK32
    return c0, c1, c2, c3


def philox_entity_xy(seed, env_id, episode, entity, lo, hi):
    """Initial (x, y) of one entity, float32, exactly as the kernel draws it.

    counter = (entity, episode, env_id lo32, env_id hi32), key = (seed lo32,
    seed hi32); u = (r >> 8) * 2^-24; value = lo + (hi - lo) * u, every
    operation rounded to float32, no fused multiply-add.
    """
    r = philox4x32_10((entity, episode, env_id & _MASK32, (env_id >> 32) & _MASK32),
                      (seed & _MASK32, (seed >> 32) & _MASK32))
    f32 = np.float32
    span = f32(f32(hi) - f32(lo))
    out = []
    for w in r[:2]:
        u = f32(f32(w >> 8) * f32(2.0 ** -24))
        out.append(f32(f32(span * u) + f32(lo)))
    return out[0], out[1]


# ----------------------------------------------------------------------------
# core (U3-U7)
# ----------------------------------------------------------------------------
class EntityState(object):
    def __init__(self):
        self.p_pos = None
        self.p_vel = None


class AgentState(EntityState):
    def __init__(self):
        super(AgentState, self).__init__()
        self.c = None


class Action(object):
    def __init__(self):
        self.u = None
        self.c = None


class Entity(object):
    def __init__(self):
        self.name = ''
        self.size = 0.050
        self.movable = False
        self.collide = True
        self.density = 25.0
        self.color = None
        self.max_speed = None
        self.accel = None
        self.state = EntityState()
        self.initial_mass = 1.0

    @property
    def mass(self):
        return self.initial_mass


class Landmark(Entity):
    def __init__(self):
        super(Landmark, self).__init__()
        self.boundary = False


class Agent(Entity):
    def __init__(self):
        super(Agent, self).__init__()
        self.movable = True
        self.silent = False
        self.blind = False
        self.u_noise = None
        self.c_noise = None
        self.u_range = 1.0
        self.state = AgentState()
        self.action = Action()
        self.action_callback = None
        self.adversary = False


class World(object):
    """U3: one particle world.  Constants are the canonical upstream ones."""

    def __init__(self):
        self.agents = []
        self.landmarks = []
        self.dim_c = 0
        self.dim_p = 2
        self.dim_color = 3
        self.dt = 0.1
        self.damping = 0.25
        self.contact_force = 1e+2
        self.contact_margin = 1e-3
        # fork-dependent knob: canonical OpenAI applies action.u
        # as the force directly (scale 1).  The MAAC fork multiplies by
        # mass*accel; kept explicit so either can be restated.
        self.action_force_uses_accel = False

    @property
    def entities(self):
        return self.agents + self.landmarks

    @property
    def policy_agents(self):
        return [a for a in self.agents if a.action_callback is None]

    def step(self):
        p_force = [None] * len(self.entities)
        p_force = self.apply_action_force(p_force)
        p_force = self.apply_environment_force(p_force)
        self.integrate_state(p_force)
        for agent in self.agents:
            self.update_agent_state(agent)

    def apply_action_force(self, p_force):
        # U4
        for i, agent in enumerate(self.agents):
            if agent.movable:
                noise = np.random.randn(*agent.action.u.shape) * agent.u_noise if agent.u_noise else 0.0
                if self.action_force_uses_accel:
                    scale = agent.mass * agent.accel if agent.accel is not None else agent.mass
                    p_force[i] = scale * agent.action.u + noise
                else:
                    p_force[i] = agent.action.u + noise
        return p_force

    def apply_environment_force(self, p_force):
        # U5: every unordered pair a<b in lexicographic order
        ents = self.entities
        for a, entity_a in enumerate(ents):
            for b, entity_b in enumerate(ents):
                if b <= a:
                    continue
                f_a, f_b = self.get_collision_force(entity_a, entity_b)
                if f_a is not None:
                    if p_force[a] is None:
                        p_force[a] = 0.0
                    p_force[a] = f_a + p_force[a]
                if f_b is not None:
                    if p_force[b] is None:
                        p_force[b] = 0.0
                    p_force[b] = f_b + p_force[b]
        return p_force

    def get_collision_force(self, entity_a, entity_b):
        if (not entity_a.collide) or (not entity_b.collide):
            return [None, None]
        if entity_a is entity_b:
            return [None, None]
        delta_pos = entity_a.state.p_pos - entity_b.state.p_pos
        dist = np.sqrt(np.sum(np.square(delta_pos)))
        dist_min = entity_a.size + entity_b.size
        k = self.contact_margin
        penetration = np.logaddexp(0, -(dist - dist_min) / k) * k
        with np.errstate(invalid='ignore', divide='ignore'):
            force = self.contact_force * delta_pos / dist * penetration
        force_a = +force if entity_a.movable else None
        force_b = -force if entity_b.movable else None
        return [force_a, force_b]

    def integrate_state(self, p_force):
        # U6: damped semi-implicit Euler
        for i, entity in enumerate(self.entities):
            if not entity.movable:
                continue
            entity.state.p_vel = entity.state.p_vel * (1 - self.damping)
            if p_force[i] is not None:
                entity.state.p_vel = entity.state.p_vel + (p_force[i] / entity.mass) * self.dt
            if entity.max_speed is not None:
                speed = np.sqrt(np.square(entity.state.p_vel[0]) + np.square(entity.state.p_vel[1]))
                if speed > entity.max_speed:
                    entity.state.p_vel = entity.state.p_vel / np.sqrt(
                        np.square(entity.state.p_vel[0]) + np.square(entity.state.p_vel[1])) * entity.max_speed
            entity.state.p_pos = entity.state.p_pos + entity.state.p_vel * self.dt

    def update_agent_state(self, agent):
        # U7
        if agent.silent:
            agent.state.c = np.zeros(self.dim_c)
        else:
            noise = np.random.randn(*agent.action.c.shape) * agent.c_noise if agent.c_noise else 0.0
            agent.state.c = agent.action.c + noise


# ----------------------------------------------------------------------------
# scenarios (U8, U9)
# ----------------------------------------------------------------------------
class SimpleSpread(object):
    """U8.  ``make_world(num_agents=n)`` is the signature the reference calls
    (experiments/scenarios.py:170); L defaults to N (SURVEY section 8 assumption)."""

    name = 'simple_spread'

    def make_world(self, num_agents=3, num_landmarks=None):
        world = World()
        world.dim_c = 2
        if num_landmarks is None:
            num_landmarks = num_agents
        world.collaborative = True
        world.agents = [Agent() for _ in range(num_agents)]
        for i, agent in enumerate(world.agents):
            agent.name = 'agent %d' % i
            agent.collide = True
            agent.silent = True
            agent.size = 0.15
        world.landmarks = [Landmark() for _ in range(num_landmarks)]
        for i, landmark in enumerate(world.landmarks):
            landmark.name = 'landmark %d' % i
            landmark.collide = False
            landmark.movable = False
        self.reset_world(world)
        return world

    def reset_world(self, world):
        # draw order: all agents, then all landmarks; 2 doubles each, from the
        # GLOBAL legacy NumPy stream (main.py:47 seeds it)
        for agent in world.agents:
            agent.state.p_pos = np.random.uniform(-1, +1, world.dim_p)
            agent.state.p_vel = np.zeros(world.dim_p)
            agent.state.c = np.zeros(world.dim_c)
        for landmark in world.landmarks:
            landmark.state.p_pos = np.random.uniform(-1, +1, world.dim_p)
            landmark.state.p_vel = np.zeros(world.dim_p)

    def is_collision(self, agent1, agent2):
        delta_pos = agent1.state.p_pos - agent2.state.p_pos
        dist = np.sqrt(np.sum(np.square(delta_pos)))
        dist_min = agent1.size + agent2.size
        return True if dist < dist_min else False

    def reward(self, agent, world):
        rew = 0
        for l in world.landmarks:
            dists = [np.sqrt(np.sum(np.square(a.state.p_pos - l.state.p_pos))) for a in world.agents]
            rew -= min(dists)
        if agent.collide:
            for a in world.agents:
                if self.is_collision(a, agent):  # includes a is agent: always -1
                    rew -= 1
        return rew

    def benchmark_data(self, agent, world):
        rew = 0
        collisions = 0
        occupied_landmarks = 0
        min_dists = 0
        for l in world.landmarks:
            dists = [np.sqrt(np.sum(np.square(a.state.p_pos - l.state.p_pos))) for a in world.agents]
            min_dists += min(dists)
            rew -= min(dists)
            if min(dists) < 0.1:
                occupied_landmarks += 1
        if agent.collide:
            for a in world.agents:
                if self.is_collision(a, agent):
                    rew -= 1
                    collisions += 1
        return (rew, collisions, min_dists, occupied_landmarks)

    def observation_full(self, agent, world):
        # layout recorded at experiments/scenarios.py:10
        entity_pos = [e.state.p_pos - agent.state.p_pos for e in world.landmarks]
        comm, other_pos = [], []
        for other in world.agents:
            if other is agent:
                continue
            comm.append(other.state.c)
            other_pos.append(other.state.p_pos - agent.state.p_pos)
        return np.concatenate([agent.state.p_vel] + [agent.state.p_pos] + entity_pos + other_pos + comm)

    def observation_local(self, agent, world):
        # experiments/scenarios.py:6-20
        entity_pos = [e.state.p_pos - agent.state.p_pos for e in world.landmarks]
        return np.concatenate([agent.state.p_vel] + [agent.state.p_pos] + entity_pos)


class SimpleTag(object):
    """U9 (predator-prey).  Agents i < num_adversaries are adversaries."""

    name = 'simple_tag'

    def make_world(self, num_good=1, num_adversaries=3, num_landmarks=2):
        world = World()
        world.dim_c = 2
        world.agents = [Agent() for _ in range(num_adversaries + num_good)]
        for i, agent in enumerate(world.agents):
            agent.name = 'agent %d' % i
            agent.collide = True
            agent.silent = True
            agent.adversary = True if i < num_adversaries else False
            agent.size = 0.075 if agent.adversary else 0.05
            agent.accel = 3.0 if agent.adversary else 4.0
            agent.max_speed = 1.0 if agent.adversary else 1.3
        world.landmarks = [Landmark() for _ in range(num_landmarks)]
        for i, landmark in enumerate(world.landmarks):
            landmark.name = 'landmark %d' % i
            landmark.collide = True
            landmark.movable = False
            landmark.size = 0.2
            landmark.boundary = False
        self.reset_world(world)
        return world

    def reset_world(self, world):
        for agent in world.agents:
            agent.state.p_pos = np.random.uniform(-1, +1, world.dim_p)
            agent.state.p_vel = np.zeros(world.dim_p)
            agent.state.c = np.zeros(world.dim_c)
        for landmark in world.landmarks:
            if not landmark.boundary:
                landmark.state.p_pos = np.random.uniform(-0.9, +0.9, world.dim_p)
                landmark.state.p_vel = np.zeros(world.dim_p)

    def is_collision(self, agent1, agent2):
        delta_pos = agent1.state.p_pos - agent2.state.p_pos
        dist = np.sqrt(np.sum(np.square(delta_pos)))
        dist_min = agent1.size + agent2.size
        return True if dist < dist_min else False

    def good_agents(self, world):
        return [a for a in world.agents if not a.adversary]

    def adversaries(self, world):
        return [a for a in world.agents if a.adversary]

    def reward(self, agent, world):
        return self.adversary_reward(agent, world) if agent.adversary else self.agent_reward(agent, world)

    def agent_reward(self, agent, world):
        rew = 0
        if agent.collide:
            for a in self.adversaries(world):
                if self.is_collision(a, agent):
                    rew -= 10

        def bound(x):
            if x < 0.9:
                return 0
            if x < 1.0:
                return (x - 0.9) * 10
            return min(np.exp(2 * x - 2), 10)

        for p in range(world.dim_p):
            x = abs(agent.state.p_pos[p])
            rew -= bound(x)
        return rew

    def adversary_reward(self, agent, world):
        rew = 0
        if agent.collide:
            for ag in self.good_agents(world):
                for adv in self.adversaries(world):
                    if self.is_collision(ag, adv):
                        rew += 10
        return rew

    def benchmark_data(self, agent, world):
        if agent.adversary:
            collisions = 0
            for a in self.good_agents(world):
                if self.is_collision(a, agent):
                    collisions += 1
            return collisions
        return 0

    def observation_full(self, agent, world):
        entity_pos = [e.state.p_pos - agent.state.p_pos for e in world.landmarks if not e.boundary]
        other_pos, other_vel = [], []
        for other in world.agents:
            if other is agent:
                continue
            other_pos.append(other.state.p_pos - agent.state.p_pos)
            if not other.adversary:
                other_vel.append(other.state.p_vel)
        return np.concatenate([agent.state.p_vel] + [agent.state.p_pos] + entity_pos + other_pos + other_vel)

    observation_local = observation_full  # the reference patches no local obs for simple_tag


class SimpleReference(object):
    """Canonical simple_reference (main.py:24 lists it): two agents that move AND
    speak; each must get the OTHER agent to the landmark only it knows (goal_b).  dim_c = 10, nobody
    collides.  The observation is the one the reference patches in (experiments/scenarios.py:23-42), which is
    also upstream's: [p_vel] + landmark_rel + [goal_b.color] + other agents' comm."""

    name = 'simple_reference'
    LANDMARK_COLORS = [(0.75, 0.25, 0.25), (0.25, 0.75, 0.25), (0.25, 0.25, 0.75)]

    def make_world(self):
        world = World()
        world.dim_c = 10
        world.collaborative = True
        world.agents = [Agent() for _ in range(2)]
        for i, agent in enumerate(world.agents):
            agent.name = 'agent %d' % i
            agent.collide = False
        world.landmarks = [Landmark() for _ in range(3)]
        for i, landmark in enumerate(world.landmarks):
            landmark.name = 'landmark %d' % i
            landmark.collide = False
            landmark.movable = False
        self.reset_world(world)
        return world

    def reset_world(self, world):
        for agent in world.agents:
            agent.goal_a = None
            agent.goal_b = None
        # draw order on the global NumPy stream: two choices, then agents, then landmarks
        world.agents[0].goal_a = world.agents[1]
        world.agents[0].goal_b = np.random.choice(world.landmarks)
        world.agents[1].goal_a = world.agents[0]
        world.agents[1].goal_b = np.random.choice(world.landmarks)
        for agent in world.agents:
            agent.color = np.array([0.25, 0.25, 0.25])
        for landmark, col in zip(world.landmarks, self.LANDMARK_COLORS):
            landmark.color = np.array(col)
        world.agents[0].goal_a.color = world.agents[0].goal_b.color
        world.agents[1].goal_a.color = world.agents[1].goal_b.color
        for agent in world.agents:
            agent.state.p_pos = np.random.uniform(-1, +1, world.dim_p)
            agent.state.p_vel = np.zeros(world.dim_p)
            agent.state.c = np.zeros(world.dim_c)
        for landmark in world.landmarks:
            landmark.state.p_pos = np.random.uniform(-1, +1, world.dim_p)
            landmark.state.p_vel = np.zeros(world.dim_p)

    def reward(self, agent, world):
        if agent.goal_a is None or agent.goal_b is None:
            return 0.0
        dist2 = np.sum(np.square(agent.goal_a.state.p_pos - agent.goal_b.state.p_pos))
        return -dist2

    def benchmark_data(self, agent, world):
        return self.reward(agent, world)

    def observation_full(self, agent, world):
        goal_color = [np.zeros(world.dim_color), np.zeros(world.dim_color)]
        if agent.goal_b is not None:
            goal_color[1] = agent.goal_b.color
        entity_pos = [e.state.p_pos - agent.state.p_pos for e in world.landmarks]
        comm = [other.state.c for other in world.agents if other is not agent]
        return np.concatenate([agent.state.p_vel] + entity_pos + [goal_color[1]] + comm)

    observation_local = observation_full  # experiments/scenarios.py:23-42 is the same expression


class SimpleSpeakerListener(object):
    """Canonical simple_speaker_listener (main.py:24 lists it): agent 0 is a fixed
    speaker that knows the goal landmark and emits one of dim_c = 3 symbols, agent 1 a silent listener that
    moves; both are rewarded with -|p_listener - p_goal|^2.  Action spaces differ per agent (Discrete(3) for
    the speaker, Discrete(5) for the listener), as upstream's environment.py builds them.
    observation_local is the one the reference patches in (experiments/scenarios.py:45-64): [p_vel] +
    landmark_rel + [goal_b.color, or zeros for the listener] -- 11 numbers for BOTH agents, and the spoken symbol
    is collected but not included.  observation_full is upstream's (speaker: 3 numbers, listener: 11)."""

    name = 'simple_speaker_listener'
    LANDMARK_COLORS = [(0.65, 0.15, 0.15), (0.15, 0.65, 0.15), (0.15, 0.15, 0.65)]

    def make_world(self):
        world = World()
        world.dim_c = 3
        world.collaborative = True
        world.agents = [Agent() for _ in range(2)]
        for i, agent in enumerate(world.agents):
            agent.name = 'agent %d' % i
            agent.collide = False
            agent.size = 0.075
        world.agents[0].movable = False   # speaker
        world.agents[1].silent = True     # listener
        world.landmarks = [Landmark() for _ in range(3)]
        for i, landmark in enumerate(world.landmarks):
            landmark.name = 'landmark %d' % i
            landmark.collide = False
            landmark.movable = False
            landmark.size = 0.04
        self.reset_world(world)
        return world

    def reset_world(self, world):
        for agent in world.agents:
            agent.goal_a = None
            agent.goal_b = None
        # draw order on the global NumPy stream: one choice, then agents, then landmarks
        world.agents[0].goal_a = world.agents[1]
        world.agents[0].goal_b = np.random.choice(world.landmarks)
        for agent in world.agents:
            agent.color = np.array([0.25, 0.25, 0.25])
        for landmark, col in zip(world.landmarks, self.LANDMARK_COLORS):
            landmark.color = np.array(col)
        world.agents[0].goal_a.color = world.agents[0].goal_b.color + np.array([0.45, 0.45, 0.45])
        for agent in world.agents:
            agent.state.p_pos = np.random.uniform(-1, +1, world.dim_p)
            agent.state.p_vel = np.zeros(world.dim_p)
            agent.state.c = np.zeros(world.dim_c)
        for landmark in world.landmarks:
            landmark.state.p_pos = np.random.uniform(-1, +1, world.dim_p)
            landmark.state.p_vel = np.zeros(world.dim_p)

    def reward(self, agent, world):
        a = world.agents[0]
        dist2 = np.sum(np.square(a.goal_a.state.p_pos - a.goal_b.state.p_pos))
        return -dist2

    def benchmark_data(self, agent, world):
        return self.reward(agent, world)

    def observation_full(self, agent, world):
        goal_color = np.zeros(world.dim_color)
        if agent.goal_b is not None:
            goal_color = agent.goal_b.color
        entity_pos = [e.state.p_pos - agent.state.p_pos for e in world.landmarks]
        comm = [other.state.c for other in world.agents if other is not agent and other.state.c is not None]
        if not agent.movable:   # speaker
            return np.concatenate([goal_color])
        return np.concatenate([agent.state.p_vel] + entity_pos + comm)   # listener

    def observation_local(self, agent, world):
        goal_color = np.zeros(world.dim_color)
        if agent.goal_b is not None:
            goal_color = agent.goal_b.color
        entity_pos = [e.state.p_pos - agent.state.p_pos for e in world.landmarks]
        return np.concatenate([agent.state.p_vel] + entity_pos + [goal_color])


# ----------------------------------------------------------------------------
# environment (U1, U2)
# ----------------------------------------------------------------------------
class _Discrete(object):
    """Stand-in for gym.spaces.Discrete: has ``.n`` and NO ``.high``
    (main.py:51-58 tells Discrete from MultiDiscrete by hasattr(.., 'high'))."""

    def __init__(self, n):
        self.n = n

    def __repr__(self):
        return 'Discrete(%d)' % self.n


class _MultiDiscrete(object):
    """Stand-in for multiagent.multi_discrete.MultiDiscrete: ``.low`` / ``.high`` arrays (main.py:52-54 reads
    ``.high + 1`` as the per-head action sizes)."""

    def __init__(self, array_of_param_array):
        self.low = np.array([x[0] for x in array_of_param_array])
        self.high = np.array([x[1] for x in array_of_param_array])
        self.num_discrete_space = self.low.shape[0]
        self.shape = (self.num_discrete_space,)

    def __repr__(self):
        return 'MultiDiscrete' + str(self.num_discrete_space)


class _Box(object):
    def __init__(self, shape):
        self.shape = shape

    def __repr__(self):
        return 'Box%s' % (self.shape,)


class OracleMultiAgentEnv(object):
    """U1/U2: the MultiAgentEnv surface run.py consumes."""

    def __init__(self, world, reset_callback=None, reward_callback=None,
                 observation_callback=None, info_callback=None, done_callback=None,
                 post_step_callback=None, shared_viewer=True, discrete_action=True):
        self.world = world
        self.agents = self.world.policy_agents
        self.n = len(world.policy_agents)
        self.reset_callback = reset_callback
        self.reward_callback = reward_callback
        self.observation_callback = observation_callback
        self.info_callback = info_callback
        self.done_callback = done_callback
        self.post_step_callback = post_step_callback
        self.discrete_action_space = discrete_action
        self.discrete_action_input = False
        self.force_discrete_action = world.discrete_action if hasattr(world, 'discrete_action') else False
        self.shared_reward = world.collaborative if hasattr(world, 'collaborative') else False
        self.time = 0
        self.action_space = []
        self.observation_space = []
        for agent in self.agents:
            total = []
            if agent.movable:
                total.append(_Discrete(world.dim_p * 2 + 1))
            if not agent.silent:
                total.append(_Discrete(world.dim_c))
            if len(total) > 1:
                self.action_space.append(_MultiDiscrete([[0, sp.n - 1] for sp in total]))
            else:
                self.action_space.append(total[0])
            obs_dim = len(observation_callback(agent, self.world))
            self.observation_space.append(_Box((obs_dim,)))
            agent.action.c = np.zeros(self.world.dim_c)

    def seed(self, seed=None):
        # gym.Env.seed default: nothing is seeded here; reset draws from the
        # global NumPy stream (main.py:45 vs :47)
        return []

    def step(self, action_n):
        obs_n, reward_n, done_n, info_n = [], [], [], {'n': []}
        self.agents = self.world.policy_agents
        for i, agent in enumerate(self.agents):
            self._set_action(action_n[i], agent, self.action_space[i])
        self.world.step()
        if self.post_step_callback is not None:
            self.post_step_callback(self.world)
        for agent in self.agents:
            obs_n.append(self._get_obs(agent))
            reward_n.append(self._get_reward(agent))
            done_n.append(self._get_done(agent))
            info_n['n'].append(self._get_info(agent))
        reward = np.sum(reward_n)
        if self.shared_reward:
            reward_n = [reward] * self.n
        return obs_n, reward_n, done_n, info_n

    def reset(self):
        self.reset_callback(self.world)
        self.agents = self.world.policy_agents
        return [self._get_obs(agent) for agent in self.agents]

    def render(self, mode='human'):
        return []

    def _get_info(self, agent):
        return {} if self.info_callback is None else self.info_callback(agent, self.world)

    def _get_obs(self, agent):
        if self.observation_callback is None:
            return np.zeros(0)
        return self.observation_callback(agent, self.world)

    def _get_done(self, agent):
        if self.done_callback is None:
            return False
        return self.done_callback(agent, self.world)

    def _get_reward(self, agent):
        if self.reward_callback is None:
            return 0.0
        return self.reward_callback(agent, self.world)

    def _set_action(self, action, agent, action_space, time=None):
        agent.action.u = np.zeros(self.world.dim_p)
        agent.action.c = np.zeros(self.world.dim_c)
        if isinstance(action_space, _MultiDiscrete):
            act, index = [], 0
            for size in action_space.high - action_space.low + 1:
                act.append(action[index:(index + size)])
                index += size
            action = act
        else:
            action = [action]
        if agent.movable:
            if self.discrete_action_input:
                agent.action.u = np.zeros(self.world.dim_p)
                if action[0] == 1:
                    agent.action.u[0] = +1.0
                if action[0] == 2:
                    agent.action.u[0] = -1.0
                if action[0] == 3:
                    agent.action.u[1] = +1.0
                if action[0] == 4:
                    agent.action.u[1] = -1.0
            else:
                if self.force_discrete_action:
                    d = np.argmax(action[0])
                    action[0][:] = 0.0
                    action[0][d] = 1.0
                if self.discrete_action_space:
                    agent.action.u[0] += action[0][1] - action[0][2]
                    agent.action.u[1] += action[0][3] - action[0][4]
                else:
                    agent.action.u = action[0]
            sensitivity = 5.0
            if agent.accel is not None:
                sensitivity = agent.accel
            agent.action.u *= sensitivity
            action = action[1:]
        if not agent.silent:
            if self.discrete_action_input:
                agent.action.c = np.zeros(self.world.dim_c)
                agent.action.c[action[0]] = 1.0
            else:
                agent.action.c = action[0]
            action = action[1:]
        assert len(action) == 0


def make_oracle_env(scenario_name, n=None, local_observation=True, benchmark=False,
                    discrete_action=True, **world_kwargs):
    """Restates experiments/scenarios.py:124-192 (make_env) on the oracle."""
    if scenario_name == 'simple_spread':
        scenario = SimpleSpread()
        world = scenario.make_world(**world_kwargs) if n is None else scenario.make_world(num_agents=n, **world_kwargs)
    elif scenario_name == 'simple_tag':
        scenario = SimpleTag()
        world = scenario.make_world(**world_kwargs)
    elif scenario_name == 'simple_reference':
        scenario = SimpleReference()
        world = scenario.make_world(**world_kwargs)
    elif scenario_name == 'simple_speaker_listener':
        scenario = SimpleSpeakerListener()
        world = scenario.make_world(**world_kwargs)
    else:
        raise ValueError('unsupported scenario: %r' % (scenario_name,))
    observation = scenario.observation_local if local_observation else scenario.observation_full
    world.collaborative = False  # experiments/scenarios.py:171
    env = OracleMultiAgentEnv(world, reset_callback=scenario.reset_world,
                              reward_callback=scenario.reward,
                              observation_callback=observation,
                              info_callback=scenario.benchmark_data if benchmark else None,
                              discrete_action=discrete_action)
    env.force_discrete_action = True  # experiments/scenarios.py:191
    env.scenario = scenario
    return env


# ----------------------------------------------------------------------------
# batched helpers used by parity tests: drive B independent oracle worlds
# ----------------------------------------------------------------------------
def set_world_state(world, pos, vel, lm_pos):
    """pos/vel: [N,2], lm_pos: [L,2] (any float dtype; stored as float64)."""
    for i, a in enumerate(world.agents):
        a.state.p_pos = np.array(pos[i], dtype=np.float64)
        a.state.p_vel = np.array(vel[i], dtype=np.float64)
        a.state.c = np.zeros(world.dim_c)
    for i, l in enumerate(world.landmarks):
        l.state.p_pos = np.array(lm_pos[i], dtype=np.float64)
        l.state.p_vel = np.zeros(world.dim_p)


def get_world_state(world):
    pos = np.stack([a.state.p_pos for a in world.agents])
    vel = np.stack([a.state.p_vel for a in world.agents])
    lm = np.stack([l.state.p_pos for l in world.landmarks]) if world.landmarks else np.zeros((0, 2))
    return pos, vel, lm


def onehot(idx, n=5):
    a = np.zeros(n)
    a[int(idx)] = 1.0
    return a
