// How far from rodio's AGC (agc.rs:397-504, default settings: target 1, attack 4 s, release 0, max gain 7, floor 0) is each way of breaking
// its two sequential chains?  The reference chain in f32, sample by sample, against:
//   (a) the window sum WITHOUT the reference's drift: exact block sums (f64) -- what any two-level / prefix form computes;
//   (b) the gain recurrence in f64 (what a scan in higher precision computes), window sum as the reference's;
//   (c) both;
//   (d) the window sum re-summed in f32 in a different ORDER (blocks of 64 then the blocks): the cheapest parallel form.
// Inputs: the bench's (U(-0.9, 0.9), 2 Mi samples a stream) and silence -> burst -> silence.   gcc -O2 -o agc_distance agc_distance.c -lm
// (test infrastructure: an experiment whose numbers DESIGN.md quotes; not part of the product)
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#define W 8192
static uint64_t rng = 88172645463325252ull;
static double urand(void) { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return (double)(rng >> 11) / 9007199254740992.0; }
typedef struct { int sum_mode, gain_f64; } Mode;  // sum_mode 0: reference; 1: exact (f64 window); 2: f32, blocks of 64
static void run(const float *x, float *y, size_t n, Mode m, unsigned rate) {
    const float target = getenv("AGC_TARGET") ? (float)atof(getenv("AGC_TARGET")) : 1.0f, maxg = 7.0f, floor_ = 0.0f;
    const float attack = expf(-1.0f / (4.0f * (float)rate)), release = 0.0f;  // duration_to_coefficient(0) = exp(-inf) = 0
    static float buf[W];
    static double bufd[W];
    memset(buf, 0, sizeof buf);
    memset(bufd, 0, sizeof bufd);
    float sum = 0.f, peak = 0.f, gain = 1.f;
    double sumd = 0.0, gaind = 1.0;
    size_t idx = 0;
    for (size_t i = 0; i < n; ++i) {
        const float s = x[i], a = fabsf(s);
        const float coeff = a > peak ? 0.0f : release;
        peak = peak * coeff + a * (1.0f - coeff);
        const float sq = a * a;
        float wsum;
        if (m.sum_mode == 0) {
            sum = sum - buf[idx] + sq;
            buf[idx] = sq;
            wsum = sum;
        } else if (m.sum_mode == 1) {
            sumd = sumd - bufd[idx] + (double)sq;  // exact to ~1e-16: f32 squares have 48 significant bits at most, 8192 of them fit a double
            bufd[idx] = (double)sq;
            if ((i & 0xffff) == 0xffff) { sumd = 0; for (int k = 0; k < W; ++k) sumd += bufd[k]; }
            wsum = (float)sumd;
        } else {
            buf[idx] = sq;
            float tot = 0.f;  // blocks of 64 in ring order from the oldest, then the 128 blocks (f32 throughout)
            if ((i & 63) == 63 || 1) {
                for (int b = 0; b < W / 64; ++b) {
                    float bs = 0.f;
                    for (int k = 0; k < 64; ++k) bs += buf[(idx + 1 + b * 64 + k) & (W - 1)];
                    tot += bs;
                }
            }
            wsum = tot;
        }
        idx = (idx + 1) & (W - 1);
        const float rms = sqrtf(wsum / (float)W);
        const float rms_gain = rms > 0.0f ? target / rms : maxg;
        const float peak_gain = peak > 0.0f ? fminf(target / peak, maxg) : maxg;
        const float desired = fmaxf(fminf(rms_gain, peak_gain), floor_);
        if (!m.gain_f64) {
            const float sp = desired > gain ? attack : release;
            gain = gain * sp + desired * (1.0f - sp);
            gain = gain < 0.1f ? 0.1f : (gain > maxg ? maxg : gain);
            y[i] = s * gain;
        } else {
            const double sp = (double)desired > gaind ? (double)attack : 0.0;
            gaind = gaind * sp + (double)desired * (1.0 - sp);
            gaind = gaind < (double)0.1f ? (double)0.1f : (gaind > 7.0 ? 7.0 : gaind);
            y[i] = s * (float)gaind;
        }
    }
}
static double dist(const float *a, const float *b, size_t n, double *rel) {
    double d = 0, pk = 0;
    for (size_t i = 0; i < n; ++i) {
        const double e = fabs((double)a[i] - (double)b[i]);
        if (e > d) d = e;
        if (fabs((double)a[i]) > pk) pk = fabs((double)a[i]);
    }
    *rel = pk > 0 ? d / pk : 0;
    return d;
}
int main(int argc, char **argv) {
    const size_t n = argc > 1 ? (size_t)atol(argv[1]) : (size_t)2 << 20;
    float *x = malloc(n * 4), *ref = malloc(n * 4), *y = malloc(n * 4);
    const char *names[] = {"(a) window sum exact (no drift)", "(b) gain recurrence in f64", "(c) both", "(d) window re-summed in f32, blocks of 64"};
    const Mode modes[] = {{1, 0}, {0, 1}, {1, 1}, {2, 0}};
    for (int input = 0; input < 4; ++input) {
        const char *in_name;
        if (input == 0) { in_name = "bench input: U(-0.9, 0.9)"; for (size_t i = 0; i < n; ++i) x[i] = (float)((urand() * 2 - 1) * 0.9); }
        else if (input == 1) { in_name = "quiet music-like: 0.05 * U(-1,1) with a slow 0.5 Hz swell"; for (size_t i = 0; i < n; ++i) x[i] = (float)((urand() * 2 - 1) * 0.05 * (1.0 + 0.8 * sin(i * 6.5e-5))); }
        else if (input == 2) { in_name = "silence -> burst (0.8) -> silence, thirds"; for (size_t i = 0; i < n; ++i) x[i] = (i > n / 3 && i < 2 * n / 3) ? (float)((urand() * 2 - 1) * 0.8) : 0.0f; }
        else { in_name = "near-silence 1e-4 -> burst 0.8 -> near-silence, fifths"; for (size_t i = 0; i < n; ++i) x[i] = (float)((urand() * 2 - 1) * (((i / (n / 5)) & 1) ? 0.8 : 1e-4)); }
        Mode r = {0, 0};
        run(x, ref, n, r, 48000);
        printf("%s, %zu samples @ 48 kHz\n", in_name, n);
        const int nm = n > ((size_t)1 << 19) ? 3 : 4;  // (d) is O(n * 8192): short runs only
        for (int k = 0; k < nm; ++k) {
            run(x, y, n, modes[k], 48000);
            double rel;
            const double d = dist(ref, y, n, &rel);
            printf("    %-44s max |y - ref| = %.3e   (%.3e of the output's peak)   %s\n", names[k], d, rel, d <= 1e-5 ? "within 1e-5" : "OUTSIDE 1e-5");
        }
    }
    return 0;
}
