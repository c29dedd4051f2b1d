/* A host in plain C on top of include/transgo_hip.h: what a non-Python integration of the self-play path looks like.
 * 8 concurrent 9x9 games with a 3-ply limit, a 2-block x 32-filter tower with synthetic weights, three moves of 32 simulations each
 * with the most visited move played (no sampling, to keep the example free of NumPy semantics); the third move ends every game,
 * and the finished games go record -> tg_sp_harvest -> device replay store -> one sampled training batch without ever becoming
 * host objects.  Build (see tests/test_gpu_selfplay.py):
 *     gcc -O2 -Iinclude examples/c_host_min.c -Ltransgo_amd -ltransgo_hip -Wl,-rpath,$PWD/transgo_amd -lm -o c_host_min */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "transgo_hip.h"

#define CHECK(call) do { int rc_ = (call); if (rc_ != 0) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, tg_last_error(ctx)); return 1; } } while (0)

int main(void) {
    enum { G = 8, S = 9, P = S * S, A = P + 1, SIMS = 32, F = 32, NB = 2 };
    tg_ctx* ctx = NULL;
    tg_config cfg;
    tg_config_default(&cfg);
    cfg.board_size = S; cfg.n_games = G; cfg.num_simulation = SIMS; cfg.net_filters = F; cfg.net_blocks = NB; cfg.max_step = 3;
    if (tg_create(&cfg, &ctx) != 0) { fprintf(stderr, "tg_create: %s\n", tg_last_error(NULL)); return 1; }

    /* synthetic weights in the packed layout (transgo_amd/model.py documents it): small values, BN scales near 1 would be nicer
     * but any finite blob evaluates */
    const size_t nf = tg_net_blob_floats(S, cfg.encode_dim, F, NB);
    float* blob = (float*)malloc(nf * sizeof(float));
    uint32_t lcg = 12345u;
    for (size_t i = 0; i < nf; ++i) { lcg = lcg * 1664525u + 1013904223u; blob[i] = ((int)(lcg >> 9) % 2001 - 1000) * 2e-5f; }
    CHECK(tg_net_load(ctx, blob, nf, 0));
    free(blob);

    uint32_t seeds[G];
    for (int g = 0; g < G; ++g) seeds[g] = 1000u + (uint32_t)g;
    CHECK(tg_sp_reset(ctx, seeds, NULL));                 /* empty boards; a batch of root positions is pending */
    CHECK(tg_sp_eval(ctx));                               /* network on the pending batch */
    CHECK(tg_sp_expand_roots(ctx));

    static int32_t visits[G * A], root_n[G], player[G], step[G], actions[G];
    uint8_t done[G];
    for (int move = 0; move < 3; ++move) {
        int32_t waves = 0;
        CHECK(tg_sp_begin_move(ctx, /*selfplay=*/1, 0));  /* Dirichlet root noise + visit target */
        CHECK(tg_sp_search(ctx, &waves));                 /* collect -> network -> absorb until every game reached its target */
        CHECK(tg_sp_root_info(ctx, visits, root_n, player, step, NULL));
        for (int g = 0; g < G; ++g) {
            int best = 0;
            for (int a = 1; a < A; ++a) if (visits[g * A + a] > visits[g * A + best]) best = a;
            actions[g] = best;
        }
        CHECK(tg_sp_play(ctx, actions, done));            /* re-root (tree reuse); new roots without children await evaluation */
        CHECK(tg_sp_eval(ctx));
        CHECK(tg_sp_expand_roots(ctx));
        printf("move %d: %d waves; game 0: player %d ply %d root visits %d -> action %d\n", move, waves, player[0], step[0], root_n[0], actions[0]);
    }
    /* ---- game end (self_play.py:929-967): every game hit the ply limit on the third move ---- */
    int32_t n_games = 0, n_pos = 0;
    CHECK(tg_sp_finished(ctx, &n_games, &n_pos));
    if (n_games != G || n_pos != 3 * G) { fprintf(stderr, "expected %d finished games / %d positions, got %d / %d\n", G, 3 * G, n_games, n_pos); return 1; }
    enum { W = (10 * P + 31) / 32 };
    static uint32_t obs_bits[3 * G * W]; static int32_t counts[3 * G * A]; static float z[3 * G]; static int8_t own[3 * G * P];
    static uint8_t mover[3 * G]; static int32_t slot[G], n_moves[G], winner[G]; static float score[G]; static int8_t terr[G * P];
    CHECK(tg_sp_harvest(ctx, obs_bits, counts, z, own, mover, /*device_out=*/0, slot, n_moves, winner, score, terr));
    printf("game in slot %d: %d moves, winner %d, score %.1f; first position: mover %d, z %+.0f\n", slot[0], n_moves[0], winner[0],
           score[0], mover[0], z[0]);
    tg_replay* rp = NULL;
    CHECK(tg_replay_create(ctx, 1024, &rp));
    CHECK(tg_replay_append(rp, obs_bits, counts, z, own, n_pos));
    long long entries = 0, index = 0; int full = 0;
    CHECK(tg_replay_info(rp, &entries, &index, &full));
    enum { B = 4 };
    const long long want[B] = {0, 7, 8 * 5 + 3, entries - 1};                /* (position, symmetry) = (e / 8, e % 8) */
    static float st[B * 10 * P], pi[B * A], zz[B], ow[B * P];
    CHECK(tg_replay_sample(rp, want, B, st, pi, zz, ow, /*device_out=*/0));
    float pisum = 0.f; for (int a = 0; a < A; ++a) pisum += pi[a];
    printf("replay: %lld entries; sampled entry 0: z %+.0f, sum(pi) %.4f\n", entries, zz[0], pisum);
    if (entries != 8LL * n_pos || pisum < 0.999f || pisum > 1.001f) return 1;
    tg_replay_destroy(rp);
    for (int g = 0; g < G; ++g) { seeds[g] += 100u; done[g] = 1; }
    CHECK(tg_sp_reset(ctx, seeds, done));                 /* restart the finished slots with new seeds */
    CHECK(tg_sp_eval(ctx));
    CHECK(tg_sp_expand_roots(ctx));
    uint64_t sims = 0, evals = 0, depth = 0, ties = 0; int32_t errors = 0, slots = 0;
    CHECK(tg_sp_stats(ctx, &sims, &evals, &depth, &ties, &errors, &slots));
    printf("c_host_min ok: %llu simulations, %llu evaluated leaves, %d errors\n", (unsigned long long)sims, (unsigned long long)evals, errors);
    tg_destroy(ctx);
    return errors != 0;
}
