// Model of the opt-in AGC (rh_agc_fast): how often is "the quietest desired gain of the tile in front is a reset point" wrong, and how long are the
// replays?  Reference chain in f32 (agc.rs:397-504, defaults) -> d[n], g[n]; then per tile k of T samples: c = argmin d over tile k-1,
// assumption d[c] <= g[c-1]; on failure the replay runs from the tile start to the first reset.   gcc -O2 -o agc_segments agc_segments.c -lm
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#define W 8192
static uint64_t rng = 88172645463325252ull;
static double urand(void) { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return (double)(rng >> 11) / 9007199254740992.0; }
int main(int argc, char **argv) {
    const size_t n = (size_t)2 << 20;
    float *x = malloc(n * 4), *d = malloc(n * 4), *g = malloc(n * 4);
    for (int input = 0; input < 5; ++input) {
        const char *nm;
        if (input == 0) { nm = "U(-0.9,0.9)"; for (size_t i = 0; i < n; ++i) x[i] = (float)((urand() * 2 - 1) * 0.9); }
        else if (input == 1) { nm = "quiet swell 0.05"; for (size_t i = 0; i < n; ++i) x[i] = (float)((urand() * 2 - 1) * 0.05 * (1.0 + 0.8 * sin(i * 6.5e-5))); }
        else if (input == 2) { nm = "silence/burst/silence"; for (size_t i = 0; i < n; ++i) x[i] = (i > n / 3 && i < 2 * n / 3) ? (float)((urand() * 2 - 1) * 0.8) : 0.0f; }
        else if (input == 3) { nm = "sine 440 Hz 0.5 + noise 0.01"; for (size_t i = 0; i < n; ++i) x[i] = (float)(0.5 * sin(i * 2 * M_PI * 440 / 96000.0) + (urand() * 2 - 1) * 0.01); }
        else { nm = "music-like: AM noise, loud"; for (size_t i = 0; i < n; ++i) x[i] = (float)((urand() * 2 - 1) * 0.7 * (0.55 + 0.45 * sin(i * 3.1e-4)) * (0.6 + 0.4 * sin(i * 7.7e-6))); }
        const float target = 1.0f, maxg = 7.0f, attack = expf(-1.0f / (4.0f * 48000.0f));
        static float buf[W];
        memset(buf, 0, sizeof buf);
        float sum = 0, gain = 1;
        size_t idx = 0, resets = 0;
        for (size_t i = 0; i < n; ++i) {
            const float a = fabsf(x[i]), sq = a * a;
            sum = sum - buf[idx] + sq; buf[idx] = sq; idx = (idx + 1) & (W - 1);
            const float rms = sqrtf(sum / (float)W), rg = rms > 0 ? target / rms : maxg, pg = a > 0 ? fminf(target / a, maxg) : maxg;
            const float de = fmaxf(fminf(rg, pg), 0.0f);
            d[i] = de;
            if (!(de > gain)) resets++;
            const float sp = de > gain ? attack : 0.0f;
            gain = gain * sp + de * (1.0f - sp);
            gain = gain < 0.1f ? 0.1f : (gain > maxg ? maxg : gain);
            g[i] = gain;
        }
        printf("%s: resets %.3f %% of samples\n", nm, 100.0 * resets / n);
        for (size_t T = 512; T <= 8192; T *= 2) {
            size_t fails = 0, replay = 0, longest = 0;
            for (size_t k = 1; k < n / T; ++k) {
                size_t c = (k - 1) * T;
                for (size_t i = (k - 1) * T; i < k * T; ++i) if (d[i] < d[c]) c = i;
                const float gprev = c ? g[c - 1] : 1.0f;
                if (d[c] > gprev) {  // not a reset: replay from the tile start to the first reset
                    fails++;
                    size_t j = k * T;
                    while (j < n && d[j] > g[j - 1]) ++j;
                    replay += j - k * T;
                    if (j - k * T > longest) longest = j - k * T;
                }
            }
            printf("    T = %5zu: %6zu tiles, %5zu wrong (%.2f %%), replayed samples %zu (%.2f %% of the stream), longest replay %zu\n", T, n / T, fails, 100.0 * fails / (n / T), replay, 100.0 * replay / n, longest);
        }
    }
    return 0;
}
