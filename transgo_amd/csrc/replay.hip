// replay.hip -- device-resident replay store + batch sampler (replaces the host path of
// replay_buffer.py:30-47 + trainer.py:46-54).
//
// The reference appends 8 augmented tuples per position (self_play.py:943-965) and the trainer draws uniform indices over
// them, stacks object tuples with np.stack and converts to float32 tensors.  Here a position is stored ONCE, un-augmented and
// compact (bit-packed planes, raw visit counts, z, signed territory); entry e of the reference buffer is (position e/8,
// symmetry e%8) in the reference's append order, and the sampler materialises the requested entries directly as the four
// float32 arrays the trainer feeds the network: state [B][C][S][S], pi [B][A], z [B], own [B][S*S].
#include <vector>

#include "ctx.h"

namespace {

struct Replay {
    int S, C, P, A, cap, obs_words;
    long long appended = 0;          // positions ever appended (ring index = appended % cap)
    uint32_t* obs = nullptr;         // [cap][obs_words] bit-packed planes (bit i of plane-major flat index)
    int32_t* counts = nullptr;       // [cap][A] raw visit counts
    float* z = nullptr;              // [cap]
    int8_t* own = nullptr;           // [cap][P]
    tg::DevBuf idx, o_state, o_pi, o_z, o_own, stage;
};

// symmetry s = 2*(i-1) + f for the reference's loop "for i in 1..4: rot90(x, i); then fliplr of that" (self_play.py:944-965).
// Returns the SOURCE point of destination (r, c).  np.rot90(m, k)[r][c]: k=1: m[c][S-1-r]; k=2: m[S-1-r][S-1-c];
// k=3: m[S-1-c][r]; k=4: m[r][c].  fliplr(y)[r][c] = y[r][S-1-c].
__device__ __forceinline__ int sym_src(int s, int r, int c, int S) {
    const int k = (s >> 1) + 1, f = s & 1;
    if (f) c = S - 1 - c;
    int rr, cc;
    switch (k & 3) {
        case 1: rr = c; cc = S - 1 - r; break;
        case 2: rr = S - 1 - r; cc = S - 1 - c; break;
        case 3: rr = S - 1 - c; cc = r; break;
        default: rr = r; cc = c; break;
    }
    return rr * S + cc;
}

__global__ __launch_bounds__(256) void k_sample(const uint32_t* __restrict__ obs, const int32_t* __restrict__ counts,
                                                const float* __restrict__ z, const int8_t* __restrict__ own,
                                                const long long* __restrict__ entry, int B, int S, int C, int obs_words,
                                                float* __restrict__ o_state, float* __restrict__ o_pi,
                                                float* __restrict__ o_z, float* __restrict__ o_own) {
    const int b = blockIdx.x, tid = threadIdx.x;
    if (b >= B) return;
    const int P = S * S, A = P + 1;
    const long long e = entry[b];
    const long long pos = e >> 3;
    const int s = (int)(e & 7);
    const uint32_t* ob = obs + pos * obs_words;
    for (int i = tid; i < C * P; i += 256) {
        const int ch = i / P, d = i % P;
        const int src = ch * P + sym_src(s, d / S, d % S, S);
        o_state[(size_t)b * C * P + i] = (float)((ob[src >> 5] >> (src & 31)) & 1u);
    }
    // pi = counts / sum with counts == 1 zeroed (self_play.py:666-671), float64 division then float32 (trainer.py:52)
    __shared__ long long ssum;
    if (tid == 0) {
        long long t = 0;
        for (int a = 0; a < A; ++a) { int c = counts[pos * A + a]; t += (c == 1) ? 0 : c; }
        ssum = t;
    }
    __syncthreads();
    const double denom = (double)ssum;
    for (int a = tid; a < A; a += 256) {
        const int src = a < P ? sym_src(s, a / S, a % S, S) : P;
        int c = counts[pos * A + src];
        if (c == 1) c = 0;
        o_pi[(size_t)b * A + a] = (float)((double)c / denom);
    }
    for (int d = tid; d < P; d += 256) o_own[(size_t)b * P + d] = (float)own[pos * P + sym_src(s, d / S, d % S, S)];
    if (tid == 0) o_z[b] = z[pos];
}

}  // namespace

struct tg_replay { Replay r; tg_ctx* ctx; };

extern "C" {

int tg_replay_create(tg_ctx* ctx, int capacity_positions, tg_replay** out) {
    if (!ctx || !out || capacity_positions <= 0) return TG_ERR_ARG;
    TG_HIP(ctx, hipSetDevice(ctx->cfg.device));
    tg_replay* h = new tg_replay();
    h->ctx = ctx;
    Replay& r = h->r;
    r.S = ctx->S; r.C = ctx->cfg.encode_dim; r.P = ctx->P; r.A = ctx->A; r.cap = capacity_positions;
    r.obs_words = (r.C * r.P + 31) / 32;
    TG_HIP(ctx, hipMalloc((void**)&r.obs, sizeof(uint32_t) * (size_t)r.cap * r.obs_words));
    TG_HIP(ctx, hipMalloc((void**)&r.counts, sizeof(int32_t) * (size_t)r.cap * r.A));
    TG_HIP(ctx, hipMalloc((void**)&r.z, sizeof(float) * (size_t)r.cap));
    TG_HIP(ctx, hipMalloc((void**)&r.own, (size_t)r.cap * r.P));
    *out = h;
    return TG_OK;
}

void tg_replay_destroy(tg_replay* h) {
    if (!h) return;
    Replay& r = h->r;
    (void)hipSetDevice(h->ctx->cfg.device);
    void* ptrs[] = {r.obs, r.counts, r.z, r.own};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    r.idx.release(); r.o_state.release(); r.o_pi.release(); r.o_z.release(); r.o_own.release(); r.stage.release();
    delete h;
}

// Append n un-augmented positions (ring buffer over positions = the reference's ring over 8-tuples, replay_buffer.py:30-34):
// obs_bits u32[n][obs_words] (bit i = plane-major flat index i of env.encode), counts i32[n][A] raw root visit counts,
// z f32[n] (+1/-1, self_play.py:931-934), own i8[n][P] (territory from the mover's side, self_play.py:938-940).
static int replay_append(tg_replay* h, const uint32_t* obs_bits, const int32_t* counts, const float* z, const int8_t* own, int n,
                         hipMemcpyKind kind) {
    if (!h || !obs_bits || !counts || !z || !own || n < 0) return TG_ERR_ARG;
    tg_ctx* ctx = h->ctx; Replay& r = h->r;
    TG_HIP(ctx, hipSetDevice(ctx->cfg.device));
    int done = 0;
    while (done < n) {
        const int at = (int)(r.appended % r.cap);
        const int k = (n - done) < (r.cap - at) ? (n - done) : (r.cap - at);
        TG_HIP(ctx, hipMemcpyAsync(r.obs + (size_t)at * r.obs_words, obs_bits + (size_t)done * r.obs_words,
                                   sizeof(uint32_t) * (size_t)k * r.obs_words, kind, ctx->stream));
        TG_HIP(ctx, hipMemcpyAsync(r.counts + (size_t)at * r.A, counts + (size_t)done * r.A, sizeof(int32_t) * (size_t)k * r.A,
                                   kind, ctx->stream));
        TG_HIP(ctx, hipMemcpyAsync(r.z + at, z + done, sizeof(float) * k, kind, ctx->stream));
        TG_HIP(ctx, hipMemcpyAsync(r.own + (size_t)at * r.P, own + (size_t)done * r.P, (size_t)k * r.P, kind, ctx->stream));
        r.appended += k; done += k;
    }
    TG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return TG_OK;
}

int tg_replay_append(tg_replay* h, const uint32_t* obs_bits, const int32_t* counts, const float* z, const int8_t* own, int n) {
    return replay_append(h, obs_bits, counts, z, own, n, hipMemcpyHostToDevice);
}
// The same with the four arrays already in this GPU's memory (what tg_sp_harvest(device_out = 1) or an RCCL gather delivers):
// finished games go from the search engine into the store without touching the host.  The arrays must be complete (their
// producer synchronised) when this is called.
int tg_replay_append_dev(tg_replay* h, const uint32_t* obs_bits, const int32_t* counts, const float* z, const int8_t* own, int n) {
    return replay_append(h, obs_bits, counts, z, own, n, hipMemcpyDeviceToDevice);
}

// Number of reference-buffer entries currently addressable (8 per stored position) and the ring position, as info() does
// (replay_buffer.py:89-94).
int tg_replay_info(const tg_replay* h, long long* entries, long long* index, int* full) {
    if (!h) return TG_ERR_ARG;
    const Replay& r = h->r;
    const long long held = r.appended < r.cap ? r.appended : r.cap;
    if (entries) *entries = held * 8;
    if (index) *index = (r.appended % r.cap) * 8;
    if (full) *full = r.appended >= r.cap;
    return TG_OK;
}

// Materialise entries entry[0..B) (index into the reference's ring of augmented tuples: position = e/8, symmetry = e%8).
// Outputs are float32; `device_out` != 0: the four pointers are device memory (e.g. torch tensors on this GPU), else host.
int tg_replay_sample(tg_replay* h, const long long* entry, int B, float* state, float* pi, float* z, float* own, int device_out) {
    if (!h || !entry || B <= 0 || !state || !pi || !z || !own) return TG_ERR_ARG;
    tg_ctx* ctx = h->ctx; Replay& r = h->r;
    TG_HIP(ctx, hipSetDevice(ctx->cfg.device));
    const long long held = (r.appended < r.cap ? r.appended : r.cap) * 8;
    for (int i = 0; i < B; ++i) if (entry[i] < 0 || entry[i] >= held) TG_FAIL(ctx, TG_ERR_ARG, "replay entry index out of range");
    const size_t n_state = (size_t)B * r.C * r.P, n_pi = (size_t)B * r.A, n_own = (size_t)B * r.P;
    if (r.idx.reserve(sizeof(long long) * B)) TG_FAIL(ctx, TG_ERR_HIP, "hipMalloc failed");
    TG_HIP(ctx, hipMemcpyAsync(r.idx.p, entry, sizeof(long long) * B, hipMemcpyHostToDevice, ctx->stream));
    float *d_state = state, *d_pi = pi, *d_z = z, *d_own = own;
    if (!device_out) {
        if (r.o_state.reserve(sizeof(float) * n_state) || r.o_pi.reserve(sizeof(float) * n_pi) || r.o_z.reserve(sizeof(float) * B) ||
            r.o_own.reserve(sizeof(float) * n_own))
            TG_FAIL(ctx, TG_ERR_HIP, "hipMalloc failed");
        d_state = (float*)r.o_state.p; d_pi = (float*)r.o_pi.p; d_z = (float*)r.o_z.p; d_own = (float*)r.o_own.p;
    }
    hipLaunchKernelGGL(k_sample, dim3(B), dim3(256), 0, ctx->stream, (const uint32_t*)r.obs, (const int32_t*)r.counts,
                       (const float*)r.z, (const int8_t*)r.own, (const long long*)r.idx.p, B, r.S, r.C, r.obs_words, d_state,
                       d_pi, d_z, d_own);
    TG_HIP(ctx, hipGetLastError());
    if (!device_out) {
        TG_HIP(ctx, hipMemcpyAsync(state, d_state, sizeof(float) * n_state, hipMemcpyDeviceToHost, ctx->stream));
        TG_HIP(ctx, hipMemcpyAsync(pi, d_pi, sizeof(float) * n_pi, hipMemcpyDeviceToHost, ctx->stream));
        TG_HIP(ctx, hipMemcpyAsync(z, d_z, sizeof(float) * B, hipMemcpyDeviceToHost, ctx->stream));
        TG_HIP(ctx, hipMemcpyAsync(own, d_own, sizeof(float) * n_own, hipMemcpyDeviceToHost, ctx->stream));
    }
    TG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return TG_OK;
}

}  // extern "C"
