/* A host in plain C on top of include/transgo_hip.h: what a non-Python integration of the self-play path looks like.
 * 8 concurrent 9x9 games, a 2-block x 32-filter tower with synthetic weights, three moves of 32 simulations each with the most
 * visited move played (no sampling, to keep the example free of NumPy semantics).  Build (see tests/test_gpu_selfplay.py):
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
    cfg.board_size = S; cfg.n_games = G; cfg.num_simulation = SIMS; cfg.net_filters = F; cfg.net_blocks = NB; cfg.max_step = 120;
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
    uint64_t sims = 0, evals = 0, depth = 0, ties = 0; int32_t errors = 0, slots = 0;
    CHECK(tg_sp_stats(ctx, &sims, &evals, &depth, &ties, &errors, &slots));
    printf("c_host_min ok: %llu simulations, %llu evaluated leaves, %d errors\n", (unsigned long long)sims, (unsigned long long)evals, errors);
    tg_destroy(ctx);
    return errors != 0;
}
