/* Sanitizer driver for the oracle's C restatement (TEST INFRASTRUCTURE ONLY): plays seeded random games through every og_*
 * entry point at 9x9 and 19x19, compiled together with go_oracle.c under -fsanitize=address,undefined (make -C oracle san).
 * Exit code 0 = no report.  GPU AddressSanitizer is not available on the pool, so this covers the CPU side only. */
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "go_oracle.c"

static uint32_t lcg(uint32_t *s) { *s = *s * 1664525u + 1013904223u; return *s >> 8; }

int main(void) {
    static int32_t legal[OG_MAXP + 1], noeye[OG_MAXP + 1];
    static float planes[13 * OG_MAXP], terr[OG_MAXP];
    static uint8_t chk[OG_MAXP + 1];
    long plies = 0;
    const int sizes[2] = {9, 19}, dims[3] = {9, 10, 13};
    for (int si = 0; si < 2; ++si)
        for (int g = 0; g < (si ? 6 : 60); ++g) {
            og_cfg cfg = {sizes[si], si ? 300 : 120, 7.5f, dims[g % 3]};
            const int P = cfg.size * cfg.size;
            uint32_t seed = 12345u + 77u * g + si;
            og_state st, nx;
            og_reset(&cfg, &st);
            int done = 0;
            while (!done) {
                const int n = og_legal_actions(&cfg, &st, legal);
                og_legal_no_eye(&cfg, &st, noeye);
                og_encode(&cfg, &st, planes);
                og_check_all(&cfg, &st, chk);
                (void)og_score(&cfg, &st); (void)og_territory(&cfg, &st, terr);
                (void)og_player(&st); (void)og_step_count(&st); (void)og_terminated(&st);
                const uint32_t r = lcg(&seed) % 100;
                int a = r < 2 ? P : r < 5 ? (int)(lcg(&seed) % (P + 3)) - 1 : legal[lcg(&seed) % (n > 1 ? n - 1 : 1)];   /* incl. illegal / out-of-range */
                int ok = 0;
                done = og_step(&cfg, &st, &nx, a, &ok);
                (void)og_check_action(&cfg, &st, a);
                st = nx;
                if (lcg(&seed) % 50 == 0) { int ok2; (void)og_step_inplace(&cfg, &st, legal[0], &ok2); done = og_terminated(&st); }
                ++plies;
            }
        }
    printf("san_driver: %ld plies, no sanitizer report\n", plies);
    return 0;
}
