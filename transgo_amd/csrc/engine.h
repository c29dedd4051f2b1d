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
    NodeRec* arena = nullptr;       // the chunk pool: [pool_chunks][chunk_slots] (tree_dev.h)
    int32_t* ring = nullptr;        // [pool_chunks] free ring of chunk ids
    PoolCtl* pool = nullptr;        // its counters
    int32_t* chunk_ids = nullptr;   // [G][2][max_chunks] chunk-id lists of the games' trees
    GameCtl* ctl = nullptr;         // [G]
    tg_mt19937* rng = nullptr;      // [G]  NumPy-legacy MT19937 stream per game
    int32_t* path_nodes = nullptr;  // [G][R][maxd]
    int32_t* path_len = nullptr;    // [G][R]
    int32_t* path_row = nullptr;    // [G][R]  slot (0..R-1) of the game's evaluation-batch entry that serves the path
    // Evaluation batch.  Game g writes the positions it wants evaluated into ITS OWN slots g*R + 0.. (env.encode bit-packed,
    // obs_words u32 each) and their number into game_nslot[g]; k_compact then turns the counts into dense rows: game_off[g] =
    // first row of game g, row_slot[row] = slot.  Rows are therefore ordered by game (deterministic), nothing is allocated with
    // atomics (round 1: ~16 k increments of one counter per wave cost a fifth of the tree stage), and the batch costs 104 B per
    // position instead of 3240 B of float planes; the network's input kernel gathers through row_slot.
    uint32_t* obs_bits = nullptr;   // [rows_cap = G*R][obs_words]
    int32_t* game_nslot = nullptr;  // [G]
    int32_t* game_off = nullptr;    // [G]
    int32_t* row_slot = nullptr;    // [rows_cap]
    uint8_t* game_act = nullptr;    // [G]  game was below its visit target when the wave started
    float* policy = nullptr;        // [rows_cap][A]
    float* value = nullptr;         // [rows_cap]
    int32_t* counters = nullptr;    // [CNT_N]
    // per-game record of the moves played so far (the observation/pi/player lists of self_play.py:917-926), ply-major per game
    uint32_t* hist_obs = nullptr;   // [G][hist_T][obs_words]  env.encode(root) bit-packed (bit i = plane-major flat index i)
    int32_t* hist_cnt = nullptr;    // [G][hist_T][A]          raw root visit counts
    uint8_t* hist_pl = nullptr;     // [G][hist_T]             side to move
    int hist_T = 0, obs_words = 0, G = 0;
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
    int n_errors = 0;                // games parked in error as of the last counter read
    size_t arena_bytes = 0;
    double* d_noise = nullptr;
    int32_t* d_i32 = nullptr;
    float* d_f32 = nullptr;
    uint8_t* d_u8 = nullptr;
    tg_mt19937* h_rng = nullptr;     // pinned host mirror of dev.rng; authoritative while rng_on_host (the host-side draws of a
    bool rng_on_host = false;        // move -- choice(A, p) then the next Dirichlet -- share one round trip)
    std::vector<int32_t> h_moves;    // moves played per game, as of the last tg_sp_play
    std::vector<int32_t> fin_slot, fin_off;   // games finished by the last tg_sp_play (ascending slot) and their position offsets
    int fin_positions = 0;
    int32_t* d_fin = nullptr;        // [2][G] device copy of fin_slot / fin_off
    DevBuf hv_obs, hv_cnt, hv_z, hv_own, hv_pl, hv_game;   // harvest staging when the caller wants host arrays
    DevBuf obs_f32;                  // float planes of the pending batch, only materialised for tg_sp_batch_obs (tests, host evaluators)
    std::vector<double> h_noise;
    std::vector<int32_t> h_nchild;
    Net* net = nullptr;
    // HIP-event timing of the tree stage (k_collect, k_absorb) on the engine's stream, switched by tg_prof_enable
    bool tprof = false; std::vector<hipEvent_t> tev; size_t tev_used = 0; double collect_ms = 0, absorb_ms = 0; long tree_waves = 0;
    std::vector<char> tev_kind;
};

}  // namespace tg

extern "C" int tg_net_forward(tg_ctx* ctx, int rows);   // pending batch (obs_bits through row_slot) -> policy[rows], value[rows] on ctx->stream
extern "C" void tg_net_destroy(tg_ctx* ctx);
extern "C" int tg_net_adopt_ready(tg_ctx* ctx);         // tg_sp_begin_move: a completed background weight refresh becomes the live set
extern "C" int tg_net_load_arch(tg_ctx* ctx, const char* arch, const float* blob, size_t n_floats, int rows_cap);
