// env.hip -- context lifetime + the batched rules-engine entry points (tg_env_*) of include/transgo_hip.h.
// One 64-lane workgroup per state; see board_dev.h for the device algorithms.
#include <cstdio>
#include <cstring>
#include <new>

#include "ctx.h"

using namespace tg;

namespace {

std::string g_create_err;

template <int S>
__global__ __launch_bounds__(64) void k_env_reset(BoardState<S>* st, int n) {
    int g = blockIdx.x;
    if (g >= n) return;
    BoardState<S> s;
    state_reset(s);
    if (lane_id() == 0) st[g] = s;
}

template <int S>
__global__ __launch_bounds__(64) void k_env_step(const BoardState<S>* in, BoardState<S>* out, const int32_t* act, int n,
                                                 RulesCfg cfg, uint8_t* done, uint8_t* ok) {
    __shared__ WaveLds<S> lds;
    int g = blockIdx.x;
    if (g >= n) return;
    BoardWave<S> bw;
    bw.init(&lds);
    BoardState<S> st = in[g];
    bool good;
    bool d = state_step(bw, st, act[g], cfg, /*check=*/true, &good);
    if (lane_id() == 0) {
        out[g] = st;
        done[g] = d ? 1 : 0;
        ok[g] = good ? 1 : 0;
    }
}

template <int S>
__global__ __launch_bounds__(64) void k_env_query(const BoardState<S>* sts, int n, RulesCfg cfg, uint8_t* legal,
                                                  uint8_t* noeye, float* obs, float* score, float* terr, int32_t* player,
                                                  int32_t* step, uint8_t* term) {
    using G = Geo<S>;
    __shared__ WaveLds<S> lds;
    int g = blockIdx.x;
    if (g >= n) return;
    BoardWave<S> bw;
    bw.init(&lds);
    BoardState<S> st = sts[g];
    const int lane = lane_id();
    if (lane == 0) {
        if (player) player[g] = st.next_player;
        if (step) step[g] = st.step_count;
        if (term) term[g] = st.terminated;
    }
    if (legal || noeye || obs) {
        bw.load_colors(st.bb[0], st.bb[1]);
        bw.analyze();
#pragma unroll
        for (int k = 0; k < G::NW; ++k) {
            if (bw.pt[k] >= G::P) continue;
            bool lg = bw.legal(k, st, st.next_player);
            if (legal) legal[(size_t)g * G::A + bw.pt[k]] = lg ? 1 : 0;
            if (noeye) noeye[(size_t)g * G::A + bw.pt[k]] = (lg && !bw.true_eye(k, st.next_player)) ? 1 : 0;
        }
        if (lane == 0) {
            if (legal) legal[(size_t)g * G::A + G::P] = 1;      // go_env.cc:161-163: pass always appended
            if (noeye) noeye[(size_t)g * G::A + G::P] = 1;
        }
        if (obs) encode_planes(bw, st, cfg, obs + (size_t)g * cfg.encode_dim * G::P);
    }
    if (score || terr) {
        uint8_t owner[G::NW];
        float raw = tromp_taylor(bw, st, owner);
        if (score && lane == 0) score[g] = raw - cfg.komi;       // go_env.cc:129
        if (terr) {
#pragma unroll
            for (int k = 0; k < G::NW; ++k)
                if (bw.pt[k] < G::P) terr[(size_t)g * G::P + bw.pt[k]] = owner[k] == 1 ? 1.f : owner[k] == 2 ? -1.f : 0.f;
        }
    }
}

}  // namespace

extern "C" {

int tg_version(void) { return 1; }

void tg_config_default(tg_config* c) {
    memset(c, 0, sizeof(*c));
    c->board_size = 9; c->encode_dim = 10; c->max_step = 120; c->komi = 7.5f;      // environment.py:35-39
    c->n_games = 0; c->num_simulation = 210; c->parallel_readouts = 4; c->wu_loss = 2;  // configure.py:29-33
    c->c_puct1 = 3; c->c_puct2 = 0.05;                                                 // configure.py:26-27
    c->net_blocks = 6; c->net_filters = 128; c->device = 0;
    c->record_games = 1;
}

const char* tg_last_error(const tg_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

int tg_engine_create(tg_ctx* ctx);      // engine.hip
void tg_engine_destroy(tg_ctx* ctx);

int tg_create(const tg_config* cfg, tg_ctx** out) {
    if (!cfg || !out) { g_create_err = "null argument"; return TG_ERR_ARG; }
    *out = nullptr;
    if (cfg->board_size != 9 && cfg->board_size != 19) { g_create_err = "board_size must be 9 or 19"; return TG_ERR_ARG; }
    if (cfg->encode_dim != 9 && cfg->encode_dim != 10 && cfg->encode_dim != 13) {
        g_create_err = "encode_dim must be 9, 10 or 13"; return TG_ERR_ARG;
    }
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        g_create_err = "no HIP device available: libtransgo_hip has no CPU path";
        return TG_ERR_NO_DEVICE;
    }
    if (cfg->device < 0 || cfg->device >= ndev) { g_create_err = "device ordinal out of range"; return TG_ERR_ARG; }
    e = hipSetDevice(cfg->device);
    if (e != hipSuccess) { g_create_err = std::string("hipSetDevice: ") + hipGetErrorString(e); return TG_ERR_HIP; }
    tg_ctx* ctx = new (std::nothrow) tg_ctx();
    if (!ctx) { g_create_err = "out of host memory"; return TG_ERR_ARG; }
    ctx->cfg = *cfg;
    ctx->S = cfg->board_size; ctx->P = ctx->S * ctx->S; ctx->A = ctx->P + 1;
    ctx->state_bytes = ctx->S == 9 ? (int)sizeof(BoardState<9>) : (int)sizeof(BoardState<19>);
    ctx->rules.max_step = cfg->max_step; ctx->rules.komi = cfg->komi; ctx->rules.encode_dim = cfg->encode_dim;
    e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { g_create_err = std::string("hipStreamCreate: ") + hipGetErrorString(e); delete ctx; return TG_ERR_HIP; }
    if (cfg->n_games > 0) {
        int rc = tg_engine_create(ctx);
        if (rc != TG_OK) { g_create_err = ctx->err; tg_destroy(ctx); return rc; }
    }
    *out = ctx;
    return TG_OK;
}

void tg_destroy(tg_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->cfg.device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    tg_engine_destroy(ctx);
    ctx->env_in.release(); ctx->env_out.release(); ctx->env_act.release(); ctx->env_flags.release();
    ctx->env_u8.release(); ctx->env_f32.release(); ctx->env_i32.release();
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int tg_sync(tg_ctx* ctx) {
    if (!ctx) return TG_ERR_ARG;
    TG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return TG_OK;
}

int tg_state_size(const tg_ctx* ctx) { return ctx ? ctx->state_bytes : 0; }

int tg_env_reset(tg_ctx* ctx, void* states, int n) {
    if (!ctx || !states || n < 0) return TG_ERR_ARG;
    if (n == 0) return TG_OK;
    TG_HIP(ctx, hipSetDevice(ctx->cfg.device));
    size_t bytes = (size_t)n * ctx->state_bytes;
    if (ctx->env_out.reserve(bytes)) TG_FAIL(ctx, TG_ERR_HIP, "hipMalloc failed (env scratch)");
    if (ctx->S == 9) hipLaunchKernelGGL(k_env_reset<9>, dim3(n), dim3(64), 0, ctx->stream, (BoardState<9>*)ctx->env_out.p, n);
    else hipLaunchKernelGGL(k_env_reset<19>, dim3(n), dim3(64), 0, ctx->stream, (BoardState<19>*)ctx->env_out.p, n);
    TG_HIP(ctx, hipGetLastError());
    TG_HIP(ctx, hipMemcpyAsync(states, ctx->env_out.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
    TG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return TG_OK;
}

int tg_env_step(tg_ctx* ctx, const void* in, void* out, const int32_t* actions, int n, uint8_t* done, uint8_t* ok) {
    if (!ctx || !in || !out || !actions || !done || n < 0) return TG_ERR_ARG;
    if (n == 0) return TG_OK;
    TG_HIP(ctx, hipSetDevice(ctx->cfg.device));
    size_t bytes = (size_t)n * ctx->state_bytes;
    if (ctx->env_in.reserve(bytes) || ctx->env_out.reserve(bytes) || ctx->env_act.reserve((size_t)n * 4) ||
        ctx->env_flags.reserve((size_t)n * 2))
        TG_FAIL(ctx, TG_ERR_HIP, "hipMalloc failed (env scratch)");
    TG_HIP(ctx, hipMemcpyAsync(ctx->env_in.p, in, bytes, hipMemcpyHostToDevice, ctx->stream));
    TG_HIP(ctx, hipMemcpyAsync(ctx->env_act.p, actions, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    uint8_t* d_done = (uint8_t*)ctx->env_flags.p;
    uint8_t* d_ok = d_done + n;
    if (ctx->S == 9)
        hipLaunchKernelGGL(k_env_step<9>, dim3(n), dim3(64), 0, ctx->stream, (const BoardState<9>*)ctx->env_in.p,
                           (BoardState<9>*)ctx->env_out.p, (const int32_t*)ctx->env_act.p, n, ctx->rules, d_done, d_ok);
    else
        hipLaunchKernelGGL(k_env_step<19>, dim3(n), dim3(64), 0, ctx->stream, (const BoardState<19>*)ctx->env_in.p,
                           (BoardState<19>*)ctx->env_out.p, (const int32_t*)ctx->env_act.p, n, ctx->rules, d_done, d_ok);
    TG_HIP(ctx, hipGetLastError());
    TG_HIP(ctx, hipMemcpyAsync(out, ctx->env_out.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
    TG_HIP(ctx, hipMemcpyAsync(done, d_done, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    if (ok) TG_HIP(ctx, hipMemcpyAsync(ok, d_ok, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    TG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return TG_OK;
}

int tg_env_query(tg_ctx* ctx, const void* states, int n, uint8_t* legal, uint8_t* noeye, float* obs, float* score,
                 float* terr, int32_t* player, int32_t* step, uint8_t* terminated) {
    if (!ctx || !states || n < 0) return TG_ERR_ARG;
    if (n == 0) return TG_OK;
    TG_HIP(ctx, hipSetDevice(ctx->cfg.device));
    const size_t A = ctx->A, P = ctx->P, C = ctx->cfg.encode_dim;
    size_t bytes = (size_t)n * ctx->state_bytes;
    // device layout: u8 = [legal n*A][noeye n*A][term n]; f32 = [obs n*C*P][score n][terr n*P]; i32 = [player n][step n]
    if (ctx->env_in.reserve(bytes) || ctx->env_u8.reserve(n * (2 * A + 1)) ||
        ctx->env_f32.reserve(sizeof(float) * n * (C * P + 1 + P)) || ctx->env_i32.reserve(sizeof(int32_t) * n * 2))
        TG_FAIL(ctx, TG_ERR_HIP, "hipMalloc failed (env scratch)");
    uint8_t* d_legal = (uint8_t*)ctx->env_u8.p; uint8_t* d_noeye = d_legal + n * A; uint8_t* d_term = d_noeye + n * A;
    float* d_obs = (float*)ctx->env_f32.p; float* d_score = d_obs + n * C * P; float* d_terr = d_score + n;
    int32_t* d_player = (int32_t*)ctx->env_i32.p; int32_t* d_step = d_player + n;
    TG_HIP(ctx, hipMemcpyAsync(ctx->env_in.p, states, bytes, hipMemcpyHostToDevice, ctx->stream));
#define TG_Q(SZ)                                                                                                     \
    hipLaunchKernelGGL(k_env_query<SZ>, dim3(n), dim3(64), 0, ctx->stream, (const BoardState<SZ>*)ctx->env_in.p, n,    \
                       ctx->rules, legal ? d_legal : nullptr, noeye ? d_noeye : nullptr, obs ? d_obs : nullptr,        \
                       score ? d_score : nullptr, terr ? d_terr : nullptr, player ? d_player : nullptr,                \
                       step ? d_step : nullptr, terminated ? d_term : nullptr)
    if (ctx->S == 9) TG_Q(9); else TG_Q(19);
#undef TG_Q
    TG_HIP(ctx, hipGetLastError());
    if (legal) TG_HIP(ctx, hipMemcpyAsync(legal, d_legal, n * A, hipMemcpyDeviceToHost, ctx->stream));
    if (noeye) TG_HIP(ctx, hipMemcpyAsync(noeye, d_noeye, n * A, hipMemcpyDeviceToHost, ctx->stream));
    if (terminated) TG_HIP(ctx, hipMemcpyAsync(terminated, d_term, n, hipMemcpyDeviceToHost, ctx->stream));
    if (obs) TG_HIP(ctx, hipMemcpyAsync(obs, d_obs, sizeof(float) * n * C * P, hipMemcpyDeviceToHost, ctx->stream));
    if (score) TG_HIP(ctx, hipMemcpyAsync(score, d_score, sizeof(float) * n, hipMemcpyDeviceToHost, ctx->stream));
    if (terr) TG_HIP(ctx, hipMemcpyAsync(terr, d_terr, sizeof(float) * n * P, hipMemcpyDeviceToHost, ctx->stream));
    if (player) TG_HIP(ctx, hipMemcpyAsync(player, d_player, sizeof(int32_t) * n, hipMemcpyDeviceToHost, ctx->stream));
    if (step) TG_HIP(ctx, hipMemcpyAsync(step, d_step, sizeof(int32_t) * n, hipMemcpyDeviceToHost, ctx->stream));
    TG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return TG_OK;
}

int tg_env_show(tg_ctx* ctx, const void* state) {      // go_env.cc:184-200, board.cc:32-44
    if (!ctx || !state) return TG_ERR_ARG;
    const int S = ctx->S, NW = (ctx->P + 63) / 64;
    const uint64_t* bb = (const uint64_t*)state;
    printf("    ");
    for (int x = 0; x < S; ++x) printf("%c ", "ABCDEFGHJKLMNOPQRST"[x]);
    printf("\n");
    for (int y = 0; y < S; ++y) {
        printf("%2d | ", y + 1);
        for (int x = 0; x < S; ++x) {
            int p = y * S + x;
            bool b = (bb[p >> 6] >> (p & 63)) & 1, w = (bb[NW + (p >> 6)] >> (p & 63)) & 1;
            printf("%s ", b ? "X" : w ? "O" : ".");
        }
        printf("\n");
    }
    fflush(stdout);
    return TG_OK;
}

}  // extern "C"
