// engine.h -- host-side engine state shared by engine.hip (tree) and net.hip (network)
#pragma once
#include <vector>

#include "ctx.h"
#include "tree_dev.h"

namespace tg {

enum { CNT_ROWS = 0, CNT_ACTIVE = 1, CNT_ERRORS = 2, CNT_N = 8 };
enum BatchKind { BATCH_NONE = 0, BATCH_ROOTS = 1, BATCH_LEAVES = 2 };

// Everything the tree kernels need, passed by value.
struct EngineDev {
    NodeRec* arena = nullptr;       // [G][2][arena_slots]
    GameCtl* ctl = nullptr;         // [G]
    tg_mt19937* rng = nullptr;      // [G]  NumPy-legacy MT19937 stream per game
    int32_t* path_nodes = nullptr;  // [G][R][maxd]
    int32_t* path_len = nullptr;    // [G][R]
    int32_t* path_row = nullptr;    // [G][R]  row of the evaluation batch that serves the path
    int32_t* row_game = nullptr;    // [rows_cap]
    float* obs = nullptr;           // [rows_cap][C][P]   evaluation batch, reference plane layout
    float* policy = nullptr;        // [rows_cap][A]
    float* value = nullptr;         // [rows_cap]
    int32_t* counters = nullptr;    // [CNT_N]
    SearchCfg sc;
    RulesCfg rules;
};

struct Net;                          // net.hip

struct Engine {
    int G = 0, R = 0, rows_cap = 0;
    EngineDev dev;
    BatchKind batch_kind = BATCH_NONE;
    bool batch_ready = false;
    bool all_reset = false;
    int last_rows = 0;
    size_t arena_bytes = 0;
    double* d_noise = nullptr;
    int32_t* d_i32 = nullptr;
    float* d_f32 = nullptr;
    uint8_t* d_u8 = nullptr;
    std::vector<tg_mt19937> h_rng;
    std::vector<double> h_noise;
    std::vector<int32_t> h_nchild;
    Net* net = nullptr;
    // HIP-event timing of the tree stage (k_collect, k_absorb) on the engine's stream, switched by tg_prof_enable
    bool tprof = false; std::vector<hipEvent_t> tev; size_t tev_used = 0; double collect_ms = 0, absorb_ms = 0; long tree_waves = 0;
    std::vector<char> tev_kind;
};

}  // namespace tg

extern "C" int tg_net_forward(tg_ctx* ctx, int rows);   // obs[rows] -> policy[rows], value[rows] on ctx->stream
extern "C" void tg_net_destroy(tg_ctx* ctx);
extern "C" int tg_net_load_arch(tg_ctx* ctx, const char* arch, const float* blob, size_t n_floats, int rows_cap);
