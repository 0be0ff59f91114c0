// pw_kernels_replay.hpp -- part of libpworld.so (one translation unit: csrc/pworld.hip includes it).
// Device replay ring, transition packing and the fused multi-GPU exchange launch.
#pragma once

namespace {

// ------------------------------------------------------------------------------------------
// device replay ring (rls/replay_buffer.py ReplayBuffer.add / _encode_sample)
// ------------------------------------------------------------------------------------------
// hipGraph support: a captured launch freezes by-value arguments, so the two values that change from
// step to step (ring position, Philox step) can also be read from device memory and advanced by a
// one-thread kernel that is part of the same graph.
__global__ void pw_counter_add_kernel(int64_t *counter, const int64_t delta, const int64_t modulo)
{
    int64_t v = *counter + delta;
    if (modulo > 0) v %= modulo;
    *counter = v;
}

// Ring variants (pw_replay_store.act_heads / per_agent): H action indices per agent (MultiDiscrete: movement and
// communication symbol, experiments/run.py:39-41) and per-agent reward / done planes (the BiCNet tuple,
// experiments/run_BIC.py:46,50).  H = 1, per_agent = 0 is the plain ring.
__device__ __forceinline__ int store_heads(const pw_replay_store &st) { return st.act_heads > 1 ? 2 : 1; }

__global__ void pw_replay_add_kernel(const pw_replay_store st, int64_t start, const int64_t *start_dev, const int B,
                                     const float *obs, const int32_t *act_idx, const float *rew_shared,
                                     const float *next_obs, const float *final_obs, const uint8_t *terminal,
                                     const float *done)
{
    const int ND = st.num_agents * st.obs_dim, N = st.num_agents, NH = N * store_heads(st);
    const size_t total = (size_t)B * ND;
    if (start_dev) start = *start_dev;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t e = i / ND, c = i - e * ND;
        const size_t slot = (size_t)((start + (int64_t)e) % st.capacity);
        st.obs[slot * ND + c] = obs[i];
        const bool fin = final_obs && terminal && terminal[e];
        st.next_obs[slot * ND + c] = fin ? final_obs[i] : next_obs[i];
        if (c < (size_t)NH) st.act[slot * NH + c] = (uint8_t)act_idx[e * NH + c];  // ND >= 2N always (obs_dim >= 2)
        if (st.per_agent) {
            if (c < (size_t)N) {
                st.rew[slot * N + c] = rew_shared[e * N + c];
                st.done[slot * N + c] = done ? done[e * N + c] : 0.0f;
            }
        } else if (c == 0) {
            st.rew[slot] = rew_shared[e];
            st.done[slot] = done ? done[e] : 0.0f;
        }
    }
}

// pw_replay_add_kernel + the step's bookkeeping in the same launch (a captured rollout step ends with this
// kernel): one extra workgroup accumulates the episode returns (pw_episode_stats_kernel's arithmetic in the same
// order: 1024 strided partial sums, then a binary tree), publishes the NEXT ring position into a second
// cursor cell -- every workgroup reads `start_dev`, so it must not be overwritten here -- and advances the
// policy's Philox step.
struct ReplayTail {
    float *episode_return;
    double *finished_sum;
    int64_t *finished_count, *next_start_dev, *step_counter;
};
__global__ void __launch_bounds__(256) pw_replay_add_tail_kernel(const pw_replay_store st, int64_t start,
                                                                 const int64_t *start_dev, const int B, const float *obs,
                                                                 const int32_t *act_idx, const float *rew_shared,
                                                                 const float *next_obs, const float *final_obs,
                                                                 const uint8_t *terminal, const float *done,
                                                                 const ReplayTail tl)
{
    const int ND = st.num_agents * st.obs_dim, N = st.num_agents;
    const size_t total = (size_t)B * ND;
    if (start_dev) start = *start_dev;
    const unsigned copy_blocks = gridDim.x - 1;  // the last workgroup does the bookkeeping, concurrently
    if (blockIdx.x < copy_blocks) {
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)copy_blocks * blockDim.x) {
            const size_t e = i / ND, c = i - e * ND;
            const size_t slot = (size_t)((start + (int64_t)e) % st.capacity);
            st.obs[slot * ND + c] = obs[i];
            const bool fin = final_obs && terminal && terminal[e];
            st.next_obs[slot * ND + c] = fin ? final_obs[i] : next_obs[i];
            if (c < (size_t)N) st.act[slot * N + c] = (uint8_t)act_idx[e * N + c];
            if (c == 0) {
                st.rew[slot] = rew_shared[e];
                st.done[slot] = done ? done[e] : 0.0f;
            }
        }
        return;
    }
    __shared__ double s_sum[128];
    __shared__ int s_cnt[128];
    const int tid = threadIdx.x;
    if (tid == 0) {
        if (tl.next_start_dev) *tl.next_start_dev = (start + B) % st.capacity;
        if (tl.step_counter) *tl.step_counter += 1;
    }
    double acc[4] = {0.0, 0.0, 0.0, 0.0};  // the partial sums of "threads" tid, tid + 256, tid + 512, tid + 768
    int cnt[4] = {0, 0, 0, 0};
    for (int e0 = tid; e0 < B; e0 += 1024) {  // the four partial sums advance together: 4 independent load chains
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int e = e0 + 256 * v;
            if (e < B) {
                const float r = tl.episode_return[e] + rew_shared[e];
                if (terminal[e]) { acc[v] += (double)r; cnt[v] += 1; tl.episode_return[e] = 0.0f; }
                else tl.episode_return[e] = r;
            }
        }
    }
    // tree levels 512 and 256 are thread-local: s[i] += s[i + 512] (i < 512), then s[i] += s[i + 256] (i < 256)
    double sum = (acc[0] + acc[2]) + (acc[1] + acc[3]);
    int c = (cnt[0] + cnt[2]) + (cnt[1] + cnt[3]);
    for (int w = 128; w > 0; w >>= 1) {
        if (tid >= w && tid < 2 * w) { s_sum[tid - w] = sum; s_cnt[tid - w] = c; }
        __syncthreads();
        if (tid < w) { sum += s_sum[tid]; c += s_cnt[tid]; }
        __syncthreads();
    }
    if (tid == 0) {
        *tl.finished_sum += sum;
        *tl.finished_count += c;
    }
}

// A whole rollout chunk into the ring in ONE launch (the sink of pw_policy_rollout / pw_rollout): transition
// (t, e) -> slot (start + t * B + e) % capacity, i.e. the order in which T calls of pw_replay_add would have stored
// them; obs of step t is obs0 (t = 0) or the chunk's obs[t - 1], next_obs the PRE-reset observation where the env
// terminated (run.py:52 vs :60).  The last `stat_blocks` workgroups do the episode-return bookkeeping of the
// chunk: thread = env, walking its T steps in order (loads batched 8 deep), a fixed-order tree per workgroup, the
// partial (sum, count) into `scratch`, and the workgroup that arrives last adds the partials in index order --
// bit-reproducible without float atomics.  scratch: [2 * stat_blocks + 1] 8-byte words, zero before first use.
__global__ void __launch_bounds__(256) pw_replay_add_rollout_kernel(const pw_replay_store st, const int64_t start,
                                                                    const int B, const int T, const float *obs0,
                                                                    const pw_step_io io, const int32_t *act,
                                                                    const ReplayTail tl, const unsigned stat_blocks,
                                                                    unsigned long long *scratch)
{
    const int ND = st.num_agents * st.obs_dim, N = st.num_agents;
    const size_t per_step = (size_t)B * ND, total = (size_t)T * per_step;
    const unsigned copy_blocks = gridDim.x - stat_blocks;
    if (blockIdx.x < copy_blocks) {
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)copy_blocks * blockDim.x) {
            const size_t t = i / per_step, rem = i - t * per_step, e = rem / ND, c = rem - e * ND;
            const size_t te = t * B + e;
            const size_t slot = (size_t)((start + (int64_t)te) % st.capacity);
            st.obs[slot * ND + c] = t == 0 ? obs0[rem] : io.obs[i - per_step];
            const bool fin = io.final_obs && io.terminal[te];
            st.next_obs[slot * ND + c] = fin ? io.final_obs[i] : io.obs[i];
            const size_t NA = (size_t)N * (st.act_heads == 2 ? 2 : 1);  // two-head ring: act [cap,N,2], chunk act [T,B,N,2]
            if (c < NA) st.act[slot * NA + c] = (uint8_t)act[te * NA + c];
            if (c == 0) {
                st.rew[slot] = io.rew_shared[te];
                st.done[slot] = 0.0f;
            }
        }
        return;
    }
    const unsigned sb = blockIdx.x - copy_blocks;
    __shared__ double s_sum[256];
    __shared__ int s_cnt[256];
    __shared__ bool s_last;
    double acc = 0.0;
    int cnt = 0;
    const int e = (int)(sb * 256 + threadIdx.x);
    if (e < B) {
        float ret = tl.episode_return[e];
        for (int t0 = 0; t0 < T; t0 += 8) {
            float rw[8];
            uint8_t tm[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int t = t0 + q < T ? t0 + q : T - 1;
                rw[q] = io.rew_shared[(size_t)t * B + e];
                tm[q] = io.terminal[(size_t)t * B + e];
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if (t0 + q < T) {
                    const float r = ret + rw[q];
                    if (tm[q]) { acc += (double)r; cnt += 1; ret = 0.0f; }
                    else ret = r;
                }
            }
        }
        tl.episode_return[e] = ret;
    }
    s_sum[threadIdx.x] = acc;
    s_cnt[threadIdx.x] = cnt;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) {
            s_sum[threadIdx.x] += s_sum[threadIdx.x + w];
            s_cnt[threadIdx.x] += s_cnt[threadIdx.x + w];
        }
        __syncthreads();
    }
    double *part_sum = reinterpret_cast<double *>(scratch);
    long long *part_cnt = reinterpret_cast<long long *>(scratch + stat_blocks);
    unsigned long long *ticket = scratch + 2 * stat_blocks;
    if (threadIdx.x == 0) {
        part_sum[sb] = s_sum[0];
        part_cnt[sb] = s_cnt[0];
        __threadfence();
        s_last = atomicAdd(ticket, 1ull) == (unsigned long long)stat_blocks - 1;
    }
    __syncthreads();
    if (s_last && threadIdx.x == 0) {
        __threadfence();
        double ssum = 0.0;
        long long scnt = 0;
        for (unsigned i = 0; i < stat_blocks; ++i) {
            ssum += __builtin_nontemporal_load(part_sum + i);
            scnt += __builtin_nontemporal_load(part_cnt + i);
        }
        *tl.finished_sum += ssum;
        *tl.finished_count += scnt;
        *ticket = 0;
    }
}

__global__ void pw_replay_gather_kernel(const pw_replay_store st, const int64_t *idx, const int b,
                                        float *out_obs, float *out_act, float *out_rew, float *out_next_obs,
                                        float *out_done)
{
    const int ND = st.num_agents * st.obs_dim, N = st.num_agents, H = store_heads(st);
    const int W0 = st.head_width[0] > 0 ? st.head_width[0] : 5, W1 = H == 2 ? st.head_width[1] : 0, W = W0 + W1;
    const size_t total = (size_t)b * ND;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t e = i / ND, c = i - e * ND;
        const size_t slot = (size_t)idx[e];
        if (out_obs) out_obs[i] = st.obs[slot * ND + c];
        if (out_next_obs) out_next_obs[i] = st.next_obs[slot * ND + c];
        if (c < (size_t)N) {
            if (out_act) {  // the one-hot rows the reference stored: [head 0 | head 1] per agent (run.py:38-41)
                const int a0 = st.act[(slot * N + c) * H], a1 = H == 2 ? st.act[(slot * N + c) * H + 1] : -1;
                float *row = out_act + (e * N + c) * W;
                for (int k = 0; k < W0; ++k) row[k] = a0 == k ? 1.0f : 0.0f;
                for (int k = 0; k < W1; ++k) row[W0 + k] = a1 == k ? 1.0f : 0.0f;
            }
            if (st.per_agent) {
                if (out_rew) out_rew[e * N + c] = st.rew[slot * N + c];
                if (out_done) out_done[e * N + c] = st.done[slot * N + c];
            }
        }
        if (c == 0 && !st.per_agent) {
            if (out_rew) out_rew[e] = st.rew[slot];
            if (out_done) out_done[e] = st.done[slot];
        }
    }
}

// Transition rows for the multi-GPU exchange: [obs ND | next_obs ND | act N | rew | done], f32.
// Row r is transition (t, e) = (sel_t[r], sel_e[r]) of a rollout chunk, t >= 1: the observation the
// policy acted on is obs[t-1], the stored next observation is the PRE-reset one (run.py:52 vs :60).
__global__ void pw_pack_transitions_kernel(const pw_step_io io, const int B, const int N, const int D,
                                           const int32_t *sel_t, const int32_t *sel_e, const int R, float *rows)
{
    const int ND = N * D, W = 2 * ND + N + 2;
    const size_t total = (size_t)R * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / W), c = (int)(i - (size_t)r * W);
        const int t = sel_t[r], e = sel_e[r];
        const size_t te = (size_t)t * B + e;
        float v;
        if (c < ND) {
            v = io.obs[((size_t)(t - 1) * B + e) * ND + c];
        } else if (c < 2 * ND) {
            const bool fin = io.final_obs && io.terminal && io.terminal[te];
            v = (fin ? io.final_obs : io.obs)[te * ND + (c - ND)];
        } else if (c < 2 * ND + N) {
            v = (float)io.act_idx[te * N + (c - 2 * ND)];
        } else if (c == 2 * ND + N) {
            v = io.rew_shared[te];
        } else {
            v = 0.0f;  // done: upstream done_callback is None
        }
        rows[i] = v;
    }
}

__global__ void pw_replay_add_packed_kernel(const pw_replay_store st, const int64_t start, const int R,
                                            const float *rows)
{
    const int N = st.num_agents, ND = N * st.obs_dim, W = 2 * ND + N + 2;
    const size_t total = (size_t)R * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / W), c = (int)(i - (size_t)r * W);
        const size_t slot = (size_t)((start + r) % st.capacity);
        const float v = rows[i];
        if (c < ND) st.obs[slot * ND + c] = v;
        else if (c < 2 * ND) st.next_obs[slot * ND + (c - ND)] = v;
        else if (c < 2 * ND + N) st.act[slot * N + (c - 2 * ND)] = (uint8_t)v;
        else if (c == 2 * ND + N) st.rew[slot] = v;
        else st.done[slot] = v;
    }
}

// One launch per exchange: blocks [0, nb_in) append the rows received by the PREVIOUS collective to the
// ring, blocks [nb_in, ...) pack this chunk's sampled transitions for the NEXT one (two tiny dependent
// launches would cost more in launch gaps than in work).
__global__ void pw_exchange_kernel(const pw_replay_store st, const int64_t start, const int R_in, const float *rows_in,
                                   const int nb_in, const pw_step_io io, const int B, const int N, const int D,
                                   const int32_t *sel_t, const int32_t *sel_e, const int R_out, float *rows_out)
{
    const int ND = N * D, W = 2 * ND + N + 2;
    if ((int)blockIdx.x < nb_in) {
        const size_t total = (size_t)R_in * W;
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)nb_in * blockDim.x) {
            const int r = (int)(i / W), c = (int)(i - (size_t)r * W);
            const size_t slot = (size_t)((start + r) % st.capacity);
            const float v = rows_in[i];
            if (c < ND) st.obs[slot * ND + c] = v;
            else if (c < 2 * ND) st.next_obs[slot * ND + (c - ND)] = v;
            else if (c < 2 * ND + N) st.act[slot * N + (c - 2 * ND)] = (uint8_t)v;
            else if (c == 2 * ND + N) st.rew[slot] = v;
            else st.done[slot] = v;
        }
        return;
    }
    const int nb_out = gridDim.x - nb_in;
    const size_t total = (size_t)R_out * W;
    for (size_t i = (size_t)(blockIdx.x - nb_in) * blockDim.x + threadIdx.x; i < total; i += (size_t)nb_out * blockDim.x) {
        const int r = (int)(i / W), c = (int)(i - (size_t)r * W);
        const int t = sel_t[r], e = sel_e[r];
        const size_t te = (size_t)t * B + e;
        float v;
        if (c < ND) {
            v = io.obs[((size_t)(t - 1) * B + e) * ND + c];
        } else if (c < 2 * ND) {
            const bool fin = io.final_obs && io.terminal && io.terminal[te];
            v = (fin ? io.final_obs : io.obs)[te * ND + (c - ND)];
        } else if (c < 2 * ND + N) {
            v = (float)io.act_idx[te * N + (c - 2 * ND)];
        } else if (c == 2 * ND + N) {
            v = io.rew_shared[te];
        } else {
            v = 0.0f;
        }
        rows_out[i] = v;
    }
}

}  // namespace

namespace {

// ------------------------------------------------------------------------------------------
// Full gather of transitions: the chunk wire block (include/pworld.h, pw_chunk_wire).
// ------------------------------------------------------------------------------------------
struct WirePtrs {
    float *obs0, *obs, *final_rows, *rew_shared;
    uint8_t *act, *fin_slot;
};
__host__ __device__ inline WirePtrs wire_ptrs(const pw_chunk_wire &w, void *wire)
{
    unsigned char *b = static_cast<unsigned char *>(wire);
    WirePtrs p;
    p.obs0 = reinterpret_cast<float *>(b + w.obs0);
    p.obs = reinterpret_cast<float *>(b + w.obs);
    p.final_rows = reinterpret_cast<float *>(b + w.final_rows);
    p.rew_shared = reinterpret_cast<float *>(b + w.rew_shared);
    p.act = b + w.act;
    p.fin_slot = b + w.fin_slot;
    return p;
}

// Sender side.  Workgroups [0, row_blocks): thread = (env, observation column), walking the env's T steps in
// order: the k-th episode end's pre-reset row goes to final_rows[k][env] and column 0 records k in fin_slot.
// The remaining workgroups narrow the int32 action indices to bytes (grid-stride).
__global__ void __launch_bounds__(256) pw_chunk_wire_finalize_kernel(const pw_chunk_wire w, void *wire, const float *obs0,
                                                                     const float *final_obs, const uint8_t *terminal,
                                                                     const int32_t *act, const unsigned row_blocks)
{
    const WirePtrs p = wire_ptrs(w, wire);
    const int ND = w.N * w.D;
    const size_t per_step = (size_t)w.B * ND;
    if (blockIdx.x < row_blocks) {
        const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
        if (i >= per_step) return;
        const size_t e = i / ND, c = i - e * ND;
        p.obs0[i] = obs0[i];
        int k = 0;
        for (int t = 0; t < w.T; ++t) {
            const size_t te = (size_t)t * w.B + e;
            const bool term = terminal[te] != 0 && final_obs != nullptr && k < w.F;
            if (term) p.final_rows[(size_t)k * per_step + i] = final_obs[(size_t)t * per_step + i];
            if (c == 0) p.fin_slot[te] = term ? (uint8_t)k : (uint8_t)0xFF;
            k += term ? 1 : 0;
        }
        return;
    }
    const size_t total = (size_t)w.T * w.B * w.N;
    const unsigned nb = gridDim.x - row_blocks;
    for (size_t i = (size_t)(blockIdx.x - row_blocks) * blockDim.x + threadIdx.x; i < total; i += (size_t)nb * blockDim.x)
        p.act[i] = (uint8_t)act[i];
}

// Root side: the copy half of pw_replay_add_rollout_kernel reading a wire block.  V = floats per thread
// (4 when N*D is a multiple of 4: 16-byte loads and stores; the planes are 256-B aligned, ring rows 16-B).
template <int V>
__global__ void __launch_bounds__(256) pw_replay_add_wire_kernel(const pw_replay_store st, const int64_t start,
                                                                 const pw_chunk_wire w, const void *wire)
{
    typedef float vec_t __attribute__((ext_vector_type(V)));
    const WirePtrs p = wire_ptrs(w, const_cast<void *>(wire));
    const int ND = w.N * w.D, N = w.N, NDV = ND / V;
    const size_t per_step = (size_t)w.B * NDV, total = (size_t)w.T * per_step;
    const vec_t *obs0 = reinterpret_cast<const vec_t *>(p.obs0), *obs = reinterpret_cast<const vec_t *>(p.obs);
    const vec_t *fin = reinterpret_cast<const vec_t *>(p.final_rows);
    vec_t *r_obs = reinterpret_cast<vec_t *>(st.obs), *r_next = reinterpret_cast<vec_t *>(st.next_obs);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t t = i / per_step, rem = i - t * per_step, e = rem / NDV, c = rem - e * NDV;
        const size_t te = t * w.B + e;
        const size_t slot = (size_t)((start + (int64_t)te) % st.capacity);
        r_obs[slot * NDV + c] = t == 0 ? obs0[rem] : obs[i - per_step];
        const unsigned fs = p.fin_slot[te];
        r_next[slot * NDV + c] = fs != 0xFFu ? fin[(size_t)fs * per_step + rem] : obs[i];
        for (int q = 0; q < V; ++q) {
            const size_t cc = c * V + q;
            if (cc < (size_t)N) st.act[slot * N + cc] = p.act[te * N + cc];
        }
        if (c == 0) {
            st.rew[slot] = p.rew_shared[te];
            st.done[slot] = 0.0f;
        }
    }
}

// ------------------------------------------------------------------------------------------
// State-only wire block (include/pworld.h, pw_state_wire): simple_spread with the local observation.  The row of agent
// a is {vx, vy, px, py, lm_0 - p, lm_1 - p, ...} (stream_write_obs in pw_kernels_spread.hpp; experiments/scenarios.py:6-20):
// columns 0..3 ARE the state, every other column is one float32 subtraction of two numbers the block carries.
// ------------------------------------------------------------------------------------------
struct StateWirePtrs {
    float4 *state0, *state, *final_state;
    float2 *lm;
    uint32_t *ep0;
    float *rew_shared;
    uint8_t *act, *epi;
};
__host__ __device__ inline StateWirePtrs state_wire_ptrs(const pw_state_wire &w, void *wire)
{
    unsigned char *b = static_cast<unsigned char *>(wire);
    StateWirePtrs p;
    p.state0 = reinterpret_cast<float4 *>(b + w.state0);
    p.state = reinterpret_cast<float4 *>(b + w.state);
    p.final_state = reinterpret_cast<float4 *>(b + w.final_state);
    p.lm = reinterpret_cast<float2 *>(b + w.lm);
    p.ep0 = reinterpret_cast<uint32_t *>(b + w.ep0);
    p.rew_shared = reinterpret_cast<float *>(b + w.rew_shared);
    p.act = b + w.act;
    p.epi = b + w.epi;
    return p;
}

// Before the chunk's rollout: what the chunk starts from, read from the bound state planes (SoA over g = env * N + agent).
__global__ void __launch_bounds__(256) pw_state_wire_begin_kernel(const pw_state_wire w, void *wire, const float *pos_x,
                                                                  const float *pos_y, const float *vel_x, const float *vel_y,
                                                                  const float *lm_x, const float *lm_y, const uint32_t *ep_count)
{
    const StateWirePtrs p = state_wire_ptrs(w, wire);
    const size_t BN = (size_t)w.B * w.N, BL = (size_t)w.B * w.L;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < BN) p.state0[i] = make_float4(vel_x[i], vel_y[i], pos_x[i], pos_y[i]);
    if (i < BL) p.lm[i] = make_float2(lm_x[i], lm_y[i]);
    if (i < (size_t)w.B) p.ep0[i] = ep_count[i];
}

// After the chunk's rollout.  Workgroups [0, copy_blocks): columns 0..3 of every observation row -> state (grid-stride).
// Workgroups [copy_blocks, copy_blocks + env_blocks): thread = (env, agent) walks the env's T terminal flags: the k-th
// episode end's pre-reset state goes to final_state[k], the landmarks the reset drew (Philox key: seed, global env id,
// episode number ep0 + k + 1, entity N + l: reset_lane in pw_common.hpp) to lm[k + 1], and agent 0 writes the step's
// epi byte.  The remaining workgroups narrow the int32 action indices to bytes.
__global__ void __launch_bounds__(256) pw_state_wire_finalize_kernel(const pw_state_wire w, void *wire, const float *obs,
                                                                     const float *final_obs, const uint8_t *terminal,
                                                                     const int32_t *act, const uint64_t seed,
                                                                     const uint64_t env_id_base, const float lm_lo, const float lm_hi,
                                                                     const unsigned copy_blocks, const unsigned env_blocks)
{
    const StateWirePtrs p = state_wire_ptrs(w, wire);
    const size_t BN = (size_t)w.B * w.N;
    const int D = w.D;
    auto row_state = [&](const float *row) {  // rows are 8-byte aligned for every D = 4 + 2L
        const float2 v = *reinterpret_cast<const float2 *>(row), q = *reinterpret_cast<const float2 *>(row + 2);
        return make_float4(v.x, v.y, q.x, q.y);
    };
    if (blockIdx.x < copy_blocks) {
        const size_t total = (size_t)w.T * BN;
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)copy_blocks * blockDim.x)
            p.state[i] = row_state(obs + i * D);
        return;
    }
    if (blockIdx.x < copy_blocks + env_blocks) {
        const size_t i = (size_t)(blockIdx.x - copy_blocks) * blockDim.x + threadIdx.x;
        if (i >= BN) return;
        const size_t e = i / w.N;
        const int a = (int)(i - e * w.N);
        const uint32_t ep0 = p.ep0[e];
        int k = 0;
        for (int t = 0; t < w.T; ++t) {
            const size_t te = (size_t)t * w.B + e;
            const bool term = terminal[te] != 0 && final_obs != nullptr && k < w.F;
            if (a == 0) p.epi[te] = (uint8_t)(k | (term ? 0x80 : 0));
            if (term) {
                p.final_state[(size_t)k * BN + i] = row_state(final_obs + ((size_t)t * BN + i) * D);
                for (int l = a; l < w.L; l += w.N) {
                    float x, y;
                    pw_reset_xy(seed, env_id_base + e, ep0 + (uint32_t)k + 1u, (uint32_t)(w.N + l), lm_lo, lm_hi, &x, &y);
                    p.lm[((size_t)(k + 1) * w.B + e) * w.L + l] = make_float2(x, y);
                }
                ++k;
            }
        }
        return;
    }
    const size_t total = (size_t)w.T * BN;
    const unsigned first = copy_blocks + env_blocks, nb = gridDim.x - first;
    for (size_t i = (size_t)(blockIdx.x - first) * blockDim.x + threadIdx.x; i < total; i += (size_t)nb * blockDim.x)
        p.act[i] = (uint8_t)act[i];
}

// Root side: ReplayBuffer.add() of the block's T*B transitions with the observation rows REBUILT from the states.
// Thread = one V-float chunk of one agent's row pair (obs_t, next_obs_t).  V = 4 (even L): chunk 0 = the state itself, chunk
// c >= 1 = landmarks 2c-2, 2c-1 relative to the agent; V = 2 (odd L): {vel}, {pos}, then one landmark per chunk.
template <int V>
__global__ void __launch_bounds__(256) pw_replay_add_state_wire_kernel(const pw_replay_store st, const int64_t start,
                                                                       const pw_state_wire w, const void *wire)
{
    typedef float vec_t __attribute__((ext_vector_type(V)));
    const StateWirePtrs p = state_wire_ptrs(w, const_cast<void *>(wire));
    const int N = w.N, L = w.L, CH = w.D / V;
    const size_t BN = (size_t)w.B * N, per_step = BN * CH, total = (size_t)w.T * per_step;
    vec_t *r_obs = reinterpret_cast<vec_t *>(st.obs), *r_next = reinterpret_cast<vec_t *>(st.next_obs);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t t = i / per_step, rem = i - t * per_step, ea = rem / CH, e = ea / N;
        const int c = (int)(rem - ea * CH), a = (int)(ea - e * N);
        const size_t te = t * w.B + e;
        const unsigned epi = p.epi[te], k = epi & 0x7fu;
        const float4 s_obs = t == 0 ? p.state0[ea] : p.state[(t - 1) * BN + ea];
        const float4 s_next = (epi & 0x80u) ? p.final_state[(size_t)k * BN + ea] : p.state[t * BN + ea];
        const float2 *lm = p.lm + ((size_t)k * w.B + e) * L;
        vec_t o, n;
        if (V == 4) {
            if (c == 0) {
                o[0] = s_obs.x; o[1] = s_obs.y; o[2] = s_obs.z; o[3] = s_obs.w;
                n[0] = s_next.x; n[1] = s_next.y; n[2] = s_next.z; n[3] = s_next.w;
            } else {
                const float2 l0 = lm[2 * c - 2], l1 = lm[2 * c - 1];
                o[0] = l0.x - s_obs.z; o[1] = l0.y - s_obs.w; o[2] = l1.x - s_obs.z; o[3] = l1.y - s_obs.w;
                n[0] = l0.x - s_next.z; n[1] = l0.y - s_next.w; n[2] = l1.x - s_next.z; n[3] = l1.y - s_next.w;
            }
        } else {
            if (c == 0) { o[0] = s_obs.x; o[1] = s_obs.y; n[0] = s_next.x; n[1] = s_next.y; }
            else if (c == 1) { o[0] = s_obs.z; o[1] = s_obs.w; n[0] = s_next.z; n[1] = s_next.w; }
            else {
                const float2 l0 = lm[c - 2];
                o[0] = l0.x - s_obs.z; o[1] = l0.y - s_obs.w;
                n[0] = l0.x - s_next.z; n[1] = l0.y - s_next.w;
            }
        }
        const size_t slot = (size_t)((start + (int64_t)te) % st.capacity);
        const size_t at = (slot * N + a) * CH + c;
        r_obs[at] = o;
        r_next[at] = n;
        if (c == 0) {
            st.act[slot * N + a] = p.act[te * N + a];
            if (a == 0) {
                st.rew[slot] = p.rew_shared[te];
                st.done[slot] = 0.0f;
            }
        }
    }
}


// ------------------------------------------------------------------------------------------
// Rows as a function of the state, in 8-byte units (every row of these scenarios is a sequence of float2 units; D is even):
//   simple_spread, local observation (experiments/scenarios.py:6-20):  [vel] [pos] [lm_l - pos] x L
//   simple_tag (tag_write_obs in pw_kernels_tag.hpp; upstream simple_tag.observation):
//                                      [vel] [pos] [lm_l - pos] x L  [pos_j - pos] for j != a ascending  [vel_j] for the GOOD agents j != a
//                                      ascending, zero-padded to D (the adversaries' width)
// s = {vx, vy, px, py} of the env's N agents, lm = its L landmarks.  Each unit is the state itself or one float32 subtraction of two
// numbers the caller holds -- the same operands in the same operation as the kernels that wrote the row, hence the same bits.
// ------------------------------------------------------------------------------------------
struct RowGeom {
    int scenario, N, L, A;
};
__device__ __forceinline__ float2 state_row_unit(const RowGeom &G, const int a, const int k, const float4 *__restrict__ s,
                                                 const float2 *__restrict__ lm)
{
    const float4 me = s[a];
    if (k == 0) return make_float2(me.x, me.y);
    if (k == 1) return make_float2(me.z, me.w);
    int q = k - 2;
    if (q < G.L) {
        const float2 l = lm[q];
        return make_float2(l.x - me.z, l.y - me.w);
    }
    if (G.scenario == PW_SIMPLE_TAG) {
        q -= G.L;
        if (q < G.N - 1) {
            const float4 o = s[q < a ? q : q + 1];
            return make_float2(o.z - me.z, o.w - me.w);
        }
        q -= G.N - 1;
        int j = G.A + q;
        if (a >= G.A && j >= a) ++j;
        if (j < G.N) {
            const float4 o = s[j];
            return make_float2(o.x, o.y);
        }
    }
    return make_float2(0.0f, 0.0f);
}

// Root side, any state-wire scenario into a ROW ring: thread = one unit of one agent's row pair (obs_t, next_obs_t).  (simple_spread
// keeps its 16-byte form above where the alignment allows; this one serves simple_tag and is the reference for both.)
__global__ void __launch_bounds__(256) pw_replay_add_state_wire_units_kernel(const pw_replay_store st, const int64_t start,
                                                                             const pw_state_wire w, const void *wire)
{
    const StateWirePtrs p = state_wire_ptrs(w, const_cast<void *>(wire));
    const RowGeom G = {w.scenario, w.N, w.L, w.num_adversaries};
    const int N = w.N, U = w.D / 2;
    const size_t BN = (size_t)w.B * N, per_step = BN * U, total = (size_t)w.T * per_step;
    float2 *r_obs = reinterpret_cast<float2 *>(st.obs), *r_next = reinterpret_cast<float2 *>(st.next_obs);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t t = i / per_step, rem = i - t * per_step, ea = rem / U, e = ea / N;
        const int k = (int)(rem - ea * U), a = (int)(ea - e * N);
        const size_t te = t * w.B + e;
        const unsigned epi = p.epi[te], ke = epi & 0x7fu;
        const float4 *s_obs = (t == 0 ? p.state0 : p.state + (t - 1) * BN) + e * N;
        const float4 *s_next = ((epi & 0x80u) ? p.final_state + (size_t)ke * BN : p.state + t * BN) + e * N;
        const float2 *lm = p.lm + ((size_t)ke * w.B + e) * w.L;
        const size_t slot = (size_t)((start + (int64_t)te) % st.capacity);
        const size_t at = (slot * N + a) * U + k;
        r_obs[at] = state_row_unit(G, a, k, s_obs, lm);
        r_next[at] = state_row_unit(G, a, k, s_next, lm);
        if (k == 0) {
            st.act[slot * N + a] = p.act[te * N + a];
            if (a == 0) {
                st.rew[slot] = p.rew_shared[te];
                st.done[slot] = 0.0f;
            }
        }
    }
}

// Root side, into a STATE ring (pw_replay_store.state_rows): the transition keeps what the block carries -- {vel, pos} of every agent
// before and after the step (pre-reset where the step ended an episode) and the episode's landmarks; nothing is rebuilt here.
// Thread = (step, env, agent): 2 x 16 B of state, the byte action, its share of the L landmarks; agent 0 the reward / done words.
__global__ void __launch_bounds__(256) pw_replay_add_state_wire_to_state_ring_kernel(const pw_replay_store st, const int64_t start,
                                                                                     const pw_state_wire w, const void *wire)
{
    const StateWirePtrs p = state_wire_ptrs(w, const_cast<void *>(wire));
    const int N = w.N, L = w.L;
    const size_t BN = (size_t)w.B * N, total = (size_t)w.T * BN;
    float4 *r_s = reinterpret_cast<float4 *>(st.obs), *r_n = reinterpret_cast<float4 *>(st.next_obs);
    float2 *r_lm = reinterpret_cast<float2 *>(st.lm);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t t = i / BN, ea = i - t * BN, e = ea / N;
        const int a = (int)(ea - e * N);
        const size_t te = t * w.B + e;
        const unsigned epi = p.epi[te], ke = epi & 0x7fu;
        const size_t slot = (size_t)((start + (int64_t)te) % st.capacity);
        r_s[slot * N + a] = t == 0 ? p.state0[ea] : p.state[(t - 1) * BN + ea];
        r_n[slot * N + a] = (epi & 0x80u) ? p.final_state[(size_t)ke * BN + ea] : p.state[t * BN + ea];
        const float2 *lm = p.lm + ((size_t)ke * w.B + e) * L;
        for (int l = a; l < L; l += N) r_lm[slot * L + l] = lm[l];
        st.act[slot * N + a] = p.act[te * N + a];
        if (a == 0) {
            st.rew[slot] = p.rew_shared[te];
            st.done[slot] = 0.0f;
        }
    }
}

// sample_index() on a STATE ring: thread = one unit of one agent's row pair of one sampled transition, REBUILT from the slot's states
// and landmarks (state_row_unit); unit 0 also writes the agent's one-hot action row, agent 0's unit 0 the reward / done words.
__global__ void __launch_bounds__(256) pw_replay_gather_state_kernel(const pw_replay_store st, const int64_t *idx, const int b,
                                                                     float *out_obs, float *out_act, float *out_rew,
                                                                     float *out_next_obs, float *out_done)
{
    const RowGeom G = {st.scenario, st.num_agents, st.num_landmarks, st.num_adversaries};
    const int N = st.num_agents, U = st.obs_dim / 2, W0 = st.head_width[0] > 0 ? st.head_width[0] : 5;
    const size_t total = (size_t)b * N * U;
    const float4 *r_s = reinterpret_cast<const float4 *>(st.obs), *r_n = reinterpret_cast<const float4 *>(st.next_obs);
    const float2 *r_lm = reinterpret_cast<const float2 *>(st.lm);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t ea = i / U, e = ea / N;
        const int k = (int)(i - ea * U), a = (int)(ea - e * N);
        const size_t slot = (size_t)idx[e];
        const float2 *lm = r_lm + slot * G.L;
        if (out_obs) reinterpret_cast<float2 *>(out_obs)[i] = state_row_unit(G, a, k, r_s + slot * N, lm);
        if (out_next_obs) reinterpret_cast<float2 *>(out_next_obs)[i] = state_row_unit(G, a, k, r_n + slot * N, lm);
        if (k == 0) {
            if (out_act) {
                const int a0 = st.act[slot * N + a];
                float *row = out_act + ea * W0;
                for (int q = 0; q < W0; ++q) row[q] = a0 == q ? 1.0f : 0.0f;
            }
            if (a == 0) {
                if (out_rew) out_rew[e] = st.rew[slot];
                if (out_done) out_done[e] = st.done[slot];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Compact-row wire block of simple_reference (include/pworld.h, pw_ref_wire).  Row of agent a (ref_write_obs in pw_kernels_reference.hpp;
// experiments/scenarios.py:23-42): [vel (2), landmark - pos (6)] = the "head", carried as it is; [goal colour (3)]: 0.75 on the goal
// landmark's channel, 0.25 elsewhere; [the other agent's communication state (10)]: one-hot of its symbol, zeros after a reset.
// ------------------------------------------------------------------------------------------
constexpr int kRefN = 2, kRefD = 21, kRefHead = 8, kRefDimC = PW_DIM_C;
struct RefWirePtrs {
    float *head0, *head, *final_head, *rew_shared;
    uint8_t *goal, *comm0, *act, *epi;
};
__host__ __device__ inline RefWirePtrs ref_wire_ptrs(const pw_ref_wire &w, void *wire)
{
    unsigned char *b = static_cast<unsigned char *>(wire);
    RefWirePtrs p;
    p.head0 = reinterpret_cast<float *>(b + w.head0);
    p.head = reinterpret_cast<float *>(b + w.head);
    p.final_head = reinterpret_cast<float *>(b + w.final_head);
    p.rew_shared = reinterpret_cast<float *>(b + w.rew_shared);
    p.goal = b + w.goal;
    p.comm0 = b + w.comm0;
    p.act = b + w.act;
    p.epi = b + w.epi;
    return p;
}
__device__ __forceinline__ uint8_t ref_row_goal(const float *row)   // the channel that carries 0.75 (first one if the row is foreign)
{
    return (uint8_t)(row[9] > row[8] ? (row[10] > row[9] ? 2 : 1) : (row[10] > row[8] ? 2 : 0));
}
__device__ __forceinline__ uint8_t ref_row_symbol(const float *row)  // the other agent's symbol in columns 11..20, 0xFF for zeros
{
    for (int q = 0; q < kRefDimC; ++q)
        if (row[11 + q] != 0.0f) return (uint8_t)q;
    return 0xFF;
}

// Sender.  Workgroups [0, copy_blocks): the heads of every post-step row (grid-stride, one float per thread).  Workgroups
// [copy_blocks, copy_blocks + env_blocks): thread = (env, agent): head0 / goal[0] / comm0 from obs0, then the env's T terminal flags in
// order -- the k-th episode end's pre-reset head, the goal the reset drew (read off the post-reset row), agent 0 the epi byte.
// The rest narrow the two-head int32 actions to bytes.
__global__ void __launch_bounds__(256) pw_ref_wire_finalize_kernel(const pw_ref_wire w, void *wire, const float *obs0, const float *obs,
                                                                   const float *final_obs, const uint8_t *terminal, const int32_t *act,
                                                                   const unsigned copy_blocks, const unsigned env_blocks)
{
    const RefWirePtrs p = ref_wire_ptrs(w, wire);
    const size_t BN = (size_t)w.B * kRefN;
    if (blockIdx.x < copy_blocks) {
        const size_t total = (size_t)w.T * BN * kRefHead;
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)copy_blocks * blockDim.x) {
            const size_t row = i / kRefHead;
            p.head[i] = obs[row * kRefD + (i - row * kRefHead)];
        }
        return;
    }
    if (blockIdx.x < copy_blocks + env_blocks) {
        const size_t i = (size_t)(blockIdx.x - copy_blocks) * blockDim.x + threadIdx.x;
        if (i >= BN) return;
        const size_t e = i / kRefN;
        const int a = (int)(i - e * kRefN);
        const float *r0 = obs0 + i * kRefD;
        for (int c = 0; c < kRefHead; ++c) p.head0[i * kRefHead + c] = r0[c];
        p.goal[i] = ref_row_goal(r0);
        p.comm0[i] = ref_row_symbol(r0);
        int k = 0;
        for (int t = 0; t < w.T; ++t) {
            const size_t te = (size_t)t * w.B + e;
            const bool term = terminal[te] != 0 && final_obs != nullptr && k < w.F;
            if (a == 0) p.epi[te] = (uint8_t)(k | (term ? 0x80 : 0));
            if (term) {
                const float *fr = final_obs + ((size_t)t * BN + i) * kRefD;
                for (int c = 0; c < kRefHead; ++c) p.final_head[((size_t)k * BN + i) * kRefHead + c] = fr[c];
                p.goal[(size_t)(k + 1) * BN + i] = ref_row_goal(obs + ((size_t)t * BN + i) * kRefD);   // the post-reset row
                ++k;
            }
        }
        return;
    }
    const size_t total = (size_t)w.T * BN * 2;
    const unsigned first = copy_blocks + env_blocks, nb = gridDim.x - first;
    for (size_t i = (size_t)(blockIdx.x - first) * blockDim.x + threadIdx.x; i < total; i += (size_t)nb * blockDim.x)
        p.act[i] = (uint8_t)act[i];
}

// Root: ReplayBuffer.add() of the block's T*B transitions into the two-head ring, rows rebuilt.  Thread = one column of one agent's
// row pair (obs_t, next_obs_t).
__global__ void __launch_bounds__(256) pw_replay_add_ref_wire_kernel(const pw_replay_store st, const int64_t start, const pw_ref_wire w,
                                                                     const void *wire)
{
    const RefWirePtrs p = ref_wire_ptrs(w, const_cast<void *>(wire));
    const size_t BN = (size_t)w.B * kRefN, per_step = BN * kRefD, total = (size_t)w.T * per_step;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t t = i / per_step, rem = i - t * per_step, ea = rem / kRefD, e = ea / kRefN;
        const int c = (int)(rem - ea * kRefD), a = (int)(ea - e * kRefN);
        const size_t te = t * w.B + e, other = e * kRefN + (1 - a);
        const unsigned epi = p.epi[te], k = epi & 0x7fu;
        const bool ended = (epi & 0x80u) != 0;
        float o, n;
        if (c < kRefHead) {
            o = (t == 0 ? p.head0 : p.head + (t - 1) * BN * kRefHead)[ea * kRefHead + c];
            n = (ended ? p.final_head + (size_t)k * BN * kRefHead : p.head + t * BN * kRefHead)[ea * kRefHead + c];
        } else if (c < kRefHead + 3) {
            o = n = (c - kRefHead) == (int)p.goal[(size_t)k * BN + ea] ? 0.75f : 0.25f;   // one episode: obs_t and next_obs_t share the goal
        } else {
            const int q = c - kRefHead - 3;
            // what agent a sees of the other agent at step t: the symbol it sampled at step t - 1 (none right after a reset)
            const unsigned prev = t == 0 ? p.comm0[ea] : ((p.epi[te - w.B] & 0x80u) ? 0xFFu : p.act[((t - 1) * BN + other) * 2 + 1]);
            o = (unsigned)q == prev ? 1.0f : 0.0f;
            n = q == (int)p.act[(t * BN + other) * 2 + 1] ? 1.0f : 0.0f;   // pre-reset or not: this step's symbol
        }
        const size_t slot = (size_t)((start + (int64_t)te) % st.capacity);
        const size_t at = (slot * kRefN + a) * kRefD + c;
        st.obs[at] = o;
        st.next_obs[at] = n;
        if (c < 2) st.act[(slot * kRefN + a) * 2 + c] = p.act[(t * BN + ea) * 2 + c];
        if (c == 0 && a == 0) {
            st.rew[slot] = p.rew_shared[te];
            st.done[slot] = 0.0f;
        }
    }
}

}  // namespace
