/* CPU ORACLE body, included twice by pworld_oracle.c with
 *   REAL = float  / SUFFIX(x) = x##_f32   (deterministic float32 arithmetic)
 *   REAL = double / SUFFIX(x) = x##_f64   (reference precision: NumPy float64)
 * Test infrastructure only -- see the header comment of pworld_oracle.c.
 *
 * Loop order follows the canonical upstream code:
 *   _set_action -> apply_action_force -> apply_environment_force (pairs a<b in
 *   lexicographic order over entities = agents + landmarks) -> integrate_state
 *   -> per agent: observation, reward, done.
 */

static REAL SUFFIX(po_softplus)(REAL x)
{
#if PO_IS_F32
    return po_softplus_det_f32(x);
#else
    /* np.logaddexp(0, x): stable form used by npy_logaddexp */
    if (x > 0) return x + log1p(exp(-x));
    return log1p(exp(x));
#endif
}

static REAL SUFFIX(po_exp)(REAL x)
{
#if PO_IS_F32
    return po_exp_det_f32(x);
#else
    return exp(x);
#endif
}

static REAL SUFFIX(po_sqrt)(REAL x)
{
#if PO_IS_F32
    return sqrtf(x);
#else
    return sqrt(x);
#endif
}

/* dist between two points, np.sqrt(np.sum(np.square(delta))) */
static REAL SUFFIX(po_dist)(REAL ax, REAL ay, REAL bx, REAL by)
{
    REAL dx = ax - bx, dy = ay - by;
    return SUFFIX(po_sqrt)(dx * dx + dy * dy);
}

static void SUFFIX(po_observe_env)(const po_config *c, const REAL *pos, const REAL *vel,
                                   const REAL *lm, REAL *obs)
{
    const int N = c->num_agents, L = c->num_landmarks, D = po_obs_dim(c);
    for (int i = 0; i < N; ++i) {
        REAL *o = obs + (size_t)i * D;
        int k = 0;
        o[k++] = vel[2 * i];
        o[k++] = vel[2 * i + 1];
        o[k++] = pos[2 * i];
        o[k++] = pos[2 * i + 1];
        for (int l = 0; l < L; ++l) {
            o[k++] = lm[2 * l] - pos[2 * i];
            o[k++] = lm[2 * l + 1] - pos[2 * i + 1];
        }
        if (c->scenario == PO_SIMPLE_SPREAD) {
            if (c->obs_mode == PO_OBS_FULL) {
                for (int j = 0; j < N; ++j) {
                    if (j == i) continue;
                    o[k++] = pos[2 * j] - pos[2 * i];
                    o[k++] = pos[2 * j + 1] - pos[2 * i + 1];
                }
                for (int j = 0; j < N; ++j) {
                    if (j == i) continue;
                    o[k++] = 0; /* comm: silent agents, dim_c = 2 */
                    o[k++] = 0;
                }
            }
        } else { /* simple_tag */
            for (int j = 0; j < N; ++j) {
                if (j == i) continue;
                o[k++] = pos[2 * j] - pos[2 * i];
                o[k++] = pos[2 * j + 1] - pos[2 * i + 1];
            }
            for (int j = 0; j < N; ++j) {
                if (j == i) continue;
                if (j >= c->num_adversaries) { /* other is a good agent */
                    o[k++] = vel[2 * j];
                    o[k++] = vel[2 * j + 1];
                }
            }
        }
        while (k < D) o[k++] = 0; /* ragged rows padded (good agents in simple_tag) */
    }
}

/* rewards + collision masks from the CURRENT (post-integration) state */
static void SUFFIX(po_reward_env)(const po_config *c, const REAL *pos, const REAL *lm,
                                  REAL *rew, uint64_t *coll)
{
    const int N = c->num_agents, L = c->num_landmarks, A = c->num_adversaries;
    for (int i = 0; i < N; ++i) {
        uint64_t m = 0;
        for (int j = 0; j < N; ++j) {
            REAL d = SUFFIX(po_dist)(pos[2 * j], pos[2 * j + 1], pos[2 * i], pos[2 * i + 1]);
            REAL dmin = (REAL)c->agent_size[j] + (REAL)c->agent_size[i];
            if (d < dmin) m |= (uint64_t)1 << j;
        }
        if (coll) coll[i] = m;
        REAL r = 0;
        if (c->scenario == PO_SIMPLE_SPREAD) {
            for (int l = 0; l < L; ++l) {
                REAL best = 0;
                for (int a = 0; a < N; ++a) {
                    REAL d = SUFFIX(po_dist)(pos[2 * a], pos[2 * a + 1], lm[2 * l], lm[2 * l + 1]);
                    if (a == 0 || d < best) best = d; /* python min(): first minimum */
                }
                r -= best;
            }
            for (int a = 0; a < N; ++a)
                if ((m >> a) & 1) r -= 1; /* includes a == i */
        } else {
            if (i >= A) { /* good agent */
                for (int a = 0; a < A; ++a)
                    if ((m >> a) & 1) r -= 10;
                for (int p = 0; p < 2; ++p) {
                    REAL x = pos[2 * i + p];
                    if (x < 0) x = -x;
                    REAL b;
                    if (x < (REAL)0.9) b = 0;
                    else if (x < (REAL)1.0) b = (x - (REAL)0.9) * 10;
                    else {
                        b = SUFFIX(po_exp)(2 * x - 2);
                        if (!(b < 10)) b = 10; /* min(exp, 10) */
                    }
                    r -= b;
                }
            } else { /* adversary: +10 per (good, adversary) colliding pair */
                for (int g = A; g < N; ++g)
                    for (int a = 0; a < A; ++a) {
                        REAL d = SUFFIX(po_dist)(pos[2 * g], pos[2 * g + 1], pos[2 * a], pos[2 * a + 1]);
                        REAL dmin = (REAL)c->agent_size[g] + (REAL)c->agent_size[a];
                        if (d < dmin) r += 10;
                    }
            }
        }
        rew[i] = r;
    }
}

static void SUFFIX(po_reset_env)(const po_config *c, uint64_t env_id, uint32_t episode,
                                 REAL *pos, REAL *vel, REAL *lm)
{
    const int N = c->num_agents, L = c->num_landmarks;
    for (int i = 0; i < N; ++i) {
        float x, y;
        po_philox_xy(c->seed, env_id, episode, (uint32_t)i, -1.0f, 1.0f, &x, &y);
        pos[2 * i] = x; pos[2 * i + 1] = y;
        vel[2 * i] = 0; vel[2 * i + 1] = 0;
    }
    const float lo = c->scenario == PO_SIMPLE_TAG ? -0.9f : -1.0f;
    for (int l = 0; l < L; ++l) {
        float x, y;
        po_philox_xy(c->seed, env_id, episode, (uint32_t)(N + l), lo, -lo, &x, &y);
        lm[2 * l] = x; lm[2 * l + 1] = y;
    }
}

/* U3-U6 for one env.  act: N action indices (act_idx) or N x 5 floats (act_vec). */
static void SUFFIX(po_world_step_env)(const po_config *c, REAL *pos, REAL *vel, const REAL *lm,
                                      const int32_t *act_idx, const REAL *act_vec)
{
    const int N = c->num_agents, L = c->num_landmarks, E = N + L;
    REAL fx[PO_MAX_ENTITIES], fy[PO_MAX_ENTITIES];
    /* U2 _set_action + U4 apply_action_force */
    for (int i = 0; i < N; ++i) {
        REAL a[5] = {0, 0, 0, 0, 0};
        if (act_idx) {
            a[act_idx[i]] = 1;
        } else {
            for (int k = 0; k < 5; ++k) a[k] = act_vec[5 * i + k];
            if (c->force_discrete_action) {
                int d = 0;
                for (int k = 1; k < 5; ++k) if (a[k] > a[d]) d = k; /* np.argmax: first max */
                for (int k = 0; k < 5; ++k) a[k] = 0;
                a[d] = 1;
            }
        }
        REAL ux = (REAL)0 + (a[1] - a[2]);
        REAL uy = (REAL)0 + (a[3] - a[4]);
        REAL sens = c->agent_accel[i] >= 0 ? (REAL)c->agent_accel[i] : (REAL)c->default_sensitivity;
        ux *= sens; uy *= sens;
        if (c->action_force_uses_accel) {
            REAL sc = c->agent_accel[i] >= 0 ? (REAL)c->mass * (REAL)c->agent_accel[i] : (REAL)c->mass;
            ux = sc * ux; uy = sc * uy;
        }
        fx[i] = ux + (REAL)0; fy[i] = uy + (REAL)0; /* + noise (0.0) */
    }
    /* U5 apply_environment_force */
    const REAL k = (REAL)c->contact_margin;
    for (int a = 0; a < E; ++a) {
        for (int b = a + 1; b < E; ++b) {
            const int a_agent = a < N, b_agent = b < N;
            const int a_coll = a_agent ? 1 : c->landmark_collide;
            const int b_coll = b_agent ? 1 : c->landmark_collide;
            if (!a_coll || !b_coll) continue;
            if (!a_agent && !b_agent) continue; /* neither movable: no force recorded */
            REAL ax = a_agent ? pos[2 * a] : lm[2 * (a - N)], ay = a_agent ? pos[2 * a + 1] : lm[2 * (a - N) + 1];
            REAL bx = b_agent ? pos[2 * b] : lm[2 * (b - N)], by = b_agent ? pos[2 * b + 1] : lm[2 * (b - N) + 1];
            REAL sa = a_agent ? (REAL)c->agent_size[a] : (REAL)c->landmark_size;
            REAL sb = b_agent ? (REAL)c->agent_size[b] : (REAL)c->landmark_size;
            REAL dx = ax - bx, dy = ay - by;
            REAL dist = SUFFIX(po_sqrt)(dx * dx + dy * dy);
            REAL dist_min = sa + sb;
            REAL pen = SUFFIX(po_softplus)(-(dist - dist_min) / k) * k;
            REAL Fx = (REAL)c->contact_force * dx / dist * pen;
            REAL Fy = (REAL)c->contact_force * dy / dist * pen;
            if (a_agent) { fx[a] = Fx + fx[a]; fy[a] = Fy + fy[a]; }
            if (b_agent) { fx[b] = -Fx + fx[b]; fy[b] = -Fy + fy[b]; }
        }
    }
    /* U6 integrate_state */
    const REAL damp = (REAL)1 - (REAL)c->damping, dt = (REAL)c->dt, mass = (REAL)c->mass;
    for (int i = 0; i < N; ++i) {
        REAL vx = vel[2 * i] * damp, vy = vel[2 * i + 1] * damp;
        vx = vx + (fx[i] / mass) * dt;
        vy = vy + (fy[i] / mass) * dt;
        if (c->agent_max_speed[i] >= 0) {
            REAL ms = (REAL)c->agent_max_speed[i];
            REAL speed = SUFFIX(po_sqrt)(vx * vx + vy * vy);
            if (speed > ms) {
                vx = vx / speed * ms;
                vy = vy / speed * ms;
            }
        }
        vel[2 * i] = vx; vel[2 * i + 1] = vy;
        pos[2 * i] = pos[2 * i] + vx * dt;
        pos[2 * i + 1] = pos[2 * i + 1] + vy * dt;
    }
}

/* One batched step, B envs.  Arrays are AoS: pos/vel [B,N,2], lm [B,L,2],
 * ep_step/ep_count [B].  Outputs (any may be NULL): obs [B,N,D] = what the
 * policy sees next (post-reset where an env auto-reset), final_obs [B,N,D] =
 * pre-reset observation (written only for envs that reset this step),
 * rew [B,N], done [B,N] (always 0: done_callback is None), terminal [B],
 * coll [B,N] bit j = is_collision(agent j, agent i). */
PO_EXPORT int SUFFIX(po_step)(const po_config *c, int B, REAL *pos, REAL *vel, REAL *lm,
                              int32_t *ep_step, uint32_t *ep_count,
                              const int32_t *act_idx, const REAL *act_vec,
                              REAL *obs, REAL *final_obs, REAL *rew, uint8_t *done,
                              uint8_t *terminal, uint64_t *coll)
{
    const int N = c->num_agents, L = c->num_landmarks, D = po_obs_dim(c);
    if (N + L > PO_MAX_ENTITIES || N > 64) return -1;
    REAL rtmp[64];
    for (int e = 0; e < B; ++e) {
        REAL *p = pos + (size_t)e * N * 2, *v = vel + (size_t)e * N * 2, *l = lm + (size_t)e * L * 2;
        SUFFIX(po_world_step_env)(c, p, v, l, act_idx ? act_idx + (size_t)e * N : NULL,
                                  act_vec ? act_vec + (size_t)e * N * 5 : NULL);
        SUFFIX(po_reward_env)(c, p, l, rew ? rew + (size_t)e * N : rtmp, coll ? coll + (size_t)e * N : NULL);
        if (done) for (int i = 0; i < N; ++i) done[(size_t)e * N + i] = 0;
        int term = 0;
        if (ep_step) {
            ep_step[e] += 1;
            term = c->max_episode_len > 0 && ep_step[e] >= c->max_episode_len;
        }
        if (terminal) terminal[e] = (uint8_t)term;
        if (term && c->auto_reset) {
            if (final_obs) SUFFIX(po_observe_env)(c, p, v, l, final_obs + (size_t)e * N * D);
            ep_count[e] += 1;
            ep_step[e] = 0;
            SUFFIX(po_reset_env)(c, c->env_id_base + (uint64_t)e, ep_count[e], p, v, l);
        }
        if (obs) SUFFIX(po_observe_env)(c, p, v, l, obs + (size_t)e * N * D);
    }
    return 0;
}

PO_EXPORT int SUFFIX(po_reset)(const po_config *c, int B, REAL *pos, REAL *vel, REAL *lm,
                               int32_t *ep_step, uint32_t *ep_count, const uint8_t *mask, REAL *obs)
{
    const int N = c->num_agents, L = c->num_landmarks, D = po_obs_dim(c);
    for (int e = 0; e < B; ++e) {
        REAL *p = pos + (size_t)e * N * 2, *v = vel + (size_t)e * N * 2, *l = lm + (size_t)e * L * 2;
        if (!mask || mask[e]) {
            ep_count[e] += 1;
            ep_step[e] = 0;
            SUFFIX(po_reset_env)(c, c->env_id_base + (uint64_t)e, ep_count[e], p, v, l);
        }
        if (obs) SUFFIX(po_observe_env)(c, p, v, l, obs + (size_t)e * N * D);
    }
    return 0;
}

PO_EXPORT int SUFFIX(po_observe)(const po_config *c, int B, const REAL *pos, const REAL *vel,
                                 const REAL *lm, REAL *obs)
{
    const int N = c->num_agents, L = c->num_landmarks, D = po_obs_dim(c);
    for (int e = 0; e < B; ++e)
        SUFFIX(po_observe_env)(c, pos + (size_t)e * N * 2, vel + (size_t)e * N * 2,
                               lm + (size_t)e * L * 2, obs + (size_t)e * N * D);
    return 0;
}

PO_EXPORT int SUFFIX(po_reward)(const po_config *c, int B, const REAL *pos, const REAL *lm,
                                REAL *rew, uint64_t *coll)
{
    const int N = c->num_agents, L = c->num_landmarks;
    for (int e = 0; e < B; ++e)
        SUFFIX(po_reward_env)(c, pos + (size_t)e * N * 2, lm + (size_t)e * L * 2,
                              rew + (size_t)e * N, coll ? coll + (size_t)e * N : NULL);
    return 0;
}


/* ------------------------------------------------------------------------------------------------
 * The two communication scenarios: N = 2 agents, L <= 3 landmarks, nobody collides.
 * Extra state: comm [B,N,dim_c] (state.c after the last step) and goal [B,N] (index of goal_b).
 *
 * simple_reference: both agents move and speak (dim_c = 10).  Actions: act_idx [B,N] (move) + act_comm [B,N]
 *   (symbol), or act_vec [B,N,15].  obs = [p_vel] + landmark_rel + goal_b.color + other agent's c
 *   (experiments/scenarios.py:23-42).  reward_i = -|p_other - p_goal_b(i)|^2.
 * simple_speaker_listener: agent 0 speaks and never moves (dim_c = 3), agent 1 moves and is silent.  Actions:
 *   act_idx [B,N] = (symbol of the speaker, movement of the listener), or act_vec [B,N,5] (the speaker's
 *   Discrete(3) vector in the first three entries).  obs = [p_vel] + landmark_rel + (goal_b.color for the
 *   speaker, zeros for the listener) (experiments/scenarios.py:45-64); reward of both = -|p_listener - p_goal|^2;
 *   goal[.,1] is unused (0).
 * ------------------------------------------------------------------------------------------------ */
static void SUFFIX(po_ref_observe_env)(const po_config *c, const REAL *pos, const REAL *vel, const REAL *lm,
                                       const REAL *comm, const int32_t *goal, REAL *obs)
{
    const int N = c->num_agents, L = c->num_landmarks, D = po_obs_dim(c);
    const int sl = c->scenario == PO_SIMPLE_SPEAKER_LISTENER, DC = po_dim_c(c);
    for (int i = 0; i < N; ++i) {
        REAL *o = obs + (size_t)i * D;
        int k = 0;
        o[k++] = vel[2 * i];
        o[k++] = vel[2 * i + 1];
        for (int l = 0; l < L; ++l) {
            o[k++] = lm[2 * l] - pos[2 * i];
            o[k++] = lm[2 * l + 1] - pos[2 * i + 1];
        }
        if (sl) {
            for (int q = 0; q < 3; ++q) o[k++] = i == 0 ? (REAL)PO_SL_COLOR[goal[0]][q] : (REAL)0;
            continue;
        }
        for (int q = 0; q < 3; ++q) o[k++] = (REAL)PO_LM_COLOR[goal[i]][q];
        for (int j = 0; j < N; ++j) {
            if (j == i) continue;
            for (int q = 0; q < DC; ++q) o[k++] = comm[(size_t)j * DC + q];
        }
    }
}

static void SUFFIX(po_ref_reset_env)(const po_config *c, uint64_t env_id, uint32_t episode, REAL *pos, REAL *vel,
                                     REAL *lm, REAL *comm, int32_t *goal)
{
    const int N = c->num_agents, L = c->num_landmarks;
    const int sl = c->scenario == PO_SIMPLE_SPEAKER_LISTENER, DC = po_dim_c(c);
    for (int i = 0; i < N; ++i) {
        float x, y;
        po_philox_xy(c->seed, env_id, episode, (uint32_t)i, -1.0f, 1.0f, &x, &y);
        pos[2 * i] = x; pos[2 * i + 1] = y;
        vel[2 * i] = 0; vel[2 * i + 1] = 0;
        goal[i] = (sl && i == 1) ? 0 : po_philox_goal(c->seed, env_id, episode, (uint32_t)i, L);
        for (int q = 0; q < DC; ++q) comm[(size_t)i * DC + q] = 0;
    }
    for (int l = 0; l < L; ++l) {
        float x, y;
        po_philox_xy(c->seed, env_id, episode, (uint32_t)(N + l), -1.0f, 1.0f, &x, &y);
        lm[2 * l] = x; lm[2 * l + 1] = y;
    }
}

PO_EXPORT int SUFFIX(po_ref_step)(const po_config *c, int B, REAL *pos, REAL *vel, REAL *lm, REAL *comm,
                                  int32_t *goal, int32_t *ep_step, uint32_t *ep_count, const int32_t *act_idx,
                                  const int32_t *act_comm, const REAL *act_vec, REAL *obs, REAL *final_obs,
                                  REAL *rew, uint8_t *done, uint8_t *terminal)
{
    const int N = c->num_agents, L = c->num_landmarks, D = po_obs_dim(c);
    const int sl = c->scenario == PO_SIMPLE_SPEAKER_LISTENER, DC = po_dim_c(c);
    if (N != 2 || L < 1 || L > 3 || DC == 0) return -1;
    const REAL damp = (REAL)1 - (REAL)c->damping, dt = (REAL)c->dt, mass = (REAL)c->mass;
    for (int e = 0; e < B; ++e) {
        REAL *p = pos + (size_t)e * N * 2, *v = vel + (size_t)e * N * 2, *l = lm + (size_t)e * L * 2;
        REAL *cm = comm + (size_t)e * N * DC;
        int32_t *g = goal + (size_t)e * N;
        for (int i = 0; i < N; ++i) { /* _set_action (per-agent action space) + World.step, no collisions */
            REAL a[5] = {0, 0, 0, 0, 0}, cv[PO_DIM_C];
            const int moves = !(sl && i == 0), speaks = !(sl && i == 1);
            for (int q = 0; q < DC; ++q) cv[q] = 0;
            if (act_idx) {
                const int ai = act_idx[(size_t)e * N + i];
                if (moves) a[ai] = 1;
                if (speaks) cv[sl ? ai : act_comm[(size_t)e * N + i]] = 1;
            } else {
                const REAL *av = act_vec + ((size_t)e * N + i) * (sl ? 5 : 5 + PO_DIM_C);
                if (moves) {
                    for (int q = 0; q < 5; ++q) a[q] = av[q];
                    if (c->force_discrete_action) {
                        int d = 0;
                        for (int q = 1; q < 5; ++q) if (a[q] > a[d]) d = q;
                        for (int q = 0; q < 5; ++q) a[q] = 0;
                        a[d] = 1;
                    }
                }
                if (speaks) for (int q = 0; q < DC; ++q) cv[q] = av[(sl ? 0 : 5) + q];
            }
            if (moves) { /* apply_action_force + integrate_state skip entities that are not movable */
                REAL ux = (REAL)0 + (a[1] - a[2]), uy = (REAL)0 + (a[3] - a[4]);
                ux *= (REAL)c->default_sensitivity; uy *= (REAL)c->default_sensitivity;
                const REAL fx = ux + (REAL)0, fy = uy + (REAL)0;
                REAL vx = v[2 * i] * damp, vy = v[2 * i + 1] * damp;
                vx = vx + (fx / mass) * dt;
                vy = vy + (fy / mass) * dt;
                v[2 * i] = vx; v[2 * i + 1] = vy;
                p[2 * i] = p[2 * i] + vx * dt;
                p[2 * i + 1] = p[2 * i + 1] + vy * dt;
            }
            /* update_agent_state: silent -> zeros, else action.c (+ 0 noise) */
            for (int q = 0; q < DC; ++q) cm[(size_t)i * DC + q] = speaks ? cv[q] + (REAL)0 : (REAL)0;
        }
        for (int i = 0; i < N; ++i) {
            const int o = sl ? 1 : 1 - i, gl = sl ? g[0] : g[i];
            const REAL dx = p[2 * o] - l[2 * gl], dy = p[2 * o + 1] - l[2 * gl + 1];
            if (rew) rew[(size_t)e * N + i] = -(dx * dx + dy * dy);
            if (done) done[(size_t)e * N + i] = 0;
        }
        int term = 0;
        if (ep_step) {
            ep_step[e] += 1;
            term = c->max_episode_len > 0 && ep_step[e] >= c->max_episode_len;
        }
        if (terminal) terminal[e] = (uint8_t)term;
        if (term && c->auto_reset) {
            if (final_obs) SUFFIX(po_ref_observe_env)(c, p, v, l, cm, g, final_obs + (size_t)e * N * D);
            ep_count[e] += 1;
            ep_step[e] = 0;
            SUFFIX(po_ref_reset_env)(c, c->env_id_base + (uint64_t)e, ep_count[e], p, v, l, cm, g);
        }
        if (obs) SUFFIX(po_ref_observe_env)(c, p, v, l, cm, g, obs + (size_t)e * N * D);
    }
    return 0;
}

PO_EXPORT int SUFFIX(po_ref_reset)(const po_config *c, int B, REAL *pos, REAL *vel, REAL *lm, REAL *comm,
                                   int32_t *goal, int32_t *ep_step, uint32_t *ep_count, REAL *obs)
{
    const int N = c->num_agents, L = c->num_landmarks, D = po_obs_dim(c);
    for (int e = 0; e < B; ++e) {
        REAL *p = pos + (size_t)e * N * 2, *v = vel + (size_t)e * N * 2, *l = lm + (size_t)e * L * 2;
        REAL *cm = comm + (size_t)e * N * po_dim_c(c);
        int32_t *g = goal + (size_t)e * N;
        ep_count[e] += 1;
        ep_step[e] = 0;
        SUFFIX(po_ref_reset_env)(c, c->env_id_base + (uint64_t)e, ep_count[e], p, v, l, cm, g);
        if (obs) SUFFIX(po_ref_observe_env)(c, p, v, l, cm, g, obs + (size_t)e * N * D);
    }
    return 0;
}

PO_EXPORT int SUFFIX(po_ref_observe)(const po_config *c, int B, const REAL *pos, const REAL *vel, const REAL *lm,
                                     const REAL *comm, const int32_t *goal, REAL *obs)
{
    const int N = c->num_agents, L = c->num_landmarks, D = po_obs_dim(c);
    for (int e = 0; e < B; ++e)
        SUFFIX(po_ref_observe_env)(c, pos + (size_t)e * N * 2, vel + (size_t)e * N * 2, lm + (size_t)e * L * 2,
                                   comm + (size_t)e * N * po_dim_c(c), goal + (size_t)e * N, obs + (size_t)e * N * D);
    return 0;
}
