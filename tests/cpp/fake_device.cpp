// fake_device.cpp -- TEST INFRASTRUCTURE ONLY.  A CPU stand-in for librodio_hip.so that lets the HOST LOGIC of
// include/rodio_hip.hpp (span readers, planners, block pumps, generations, late joins, format changes) run in the
// `-m "not gpu"` suite, where there is no device.  It is linked into tests/cpp/host_mirror_test_fake and into nothing else:
// the product (rodio_amd/, librodio_hip.so, the host mirror itself) never sees it, and no GPU test, smoke() or bench.py leg
// runs through it.  It is NOT a CPU fallback of the library: it exists so that a planner bug shows up here in seconds instead
// of on the GPU box in minutes.
//
// What it is: the entry points of include/rodio_hip.h that the host mirror calls, executed synchronously on the calling thread
// ("device memory" is host memory, streams and events are no-ops), with the arithmetic of the reference restated in the
// reference's order (like oracle/rodio_oracle.cpp, whose iterator classes it does not use: the C ABI works on blocks with
// carried state, so the loops are written block-wise here).  Entry points the host mirror does not call are absent, and the
// every adapter of the mirror has its entry point here (rh_dither with the counter-based noise the header states).  The fused stream (rh_rlm_stream_block_v) is emulated by its CONTRACT (whole
// tiles while sources are live, everything once all have ended, one common *consumed_frames), not by its kernels.
//
// Build: g++ -std=c++17 -O2 -ffp-contract=off -I include tests/cpp/host_mirror_test.cpp tests/cpp/fake_device.cpp -o tests/cpp/host_mirror_test_fake
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <numeric>
#include <string>
#include <vector>

#include "rodio_hip.h"

namespace {
constexpr float LOG2_10 = 3.32192809488736234787f;
constexpr float LOG10_2 = 0.301029995663981195214f;
constexpr float PI_F = 3.14159265358979323846264338327950288f;
inline float lerp(float a, float b, uint32_t num, uint32_t den) { return a + (b - a) * (float)num / (float)den; }  // math.rs:23-26
inline float duration_to_float(uint64_t ns) { return (float)(ns / 1000000000ull) + (float)(uint32_t)(ns % 1000000000ull) / 1000000000.0f; }
inline uint64_t lerp_ready(uint64_t n, uint64_t F, uint64_t T) { return n == 0 ? 0 : (uint64_t)((((unsigned __int128)(n - 1) * T) + F - 1) / F); }
struct Ratio {
    uint64_t F, T;
};
inline Ratio reduce(uint32_t from, uint32_t to) {
    const uint32_t g = std::gcd(from, to);
    return Ratio{from / g, to / g};
}
bool g_init = false;
}  // namespace

extern "C" {

// ---------------------------------------------------------------- runtime ----
int32_t rh_version(void) { return 100; }
const char *rh_status_string(rh_status s) {
    switch (s) {
        case RH_OK: return "ok";
        case RH_ERR_INVALID: return "invalid argument";
        case RH_ERR_HIP: return "HIP runtime error";
        case RH_ERR_UNSUPPORTED: return "unsupported configuration";
        case RH_ERR_NOMEM: return "out of memory";
        case RH_ERR_TIMEOUT: return "in-kernel wait timed out";
        case RH_ERR_NOT_INITIALIZED: return "rh_init() has not succeeded";
        case RH_ERR_CAPACITY: return "output buffer too small";
        default: return "unknown status";
    }
}
const char *rh_last_hip_error(void) { return "(fake device)"; }
rh_status rh_init(int32_t) {
    g_init = true;
    return RH_OK;
}
rh_status rh_bind_thread(void) { return RH_OK; }
rh_status rh_async_status(void) { return RH_OK; }
rh_status rh_malloc(void **out, size_t bytes) {
    *out = std::malloc(bytes ? bytes : 1);
    if (*out) std::memset(*out, 0xff, bytes ? bytes : 1);  // fresh device memory reads as NaN: nothing may rely on zeros
    return *out ? RH_OK : RH_ERR_NOMEM;
}
rh_status rh_free(void *p) {
    std::free(p);
    return RH_OK;
}
rh_status rh_host_alloc(void **out, size_t bytes) { return rh_malloc(out, bytes); }
rh_status rh_host_free(void *p) { return rh_free(p); }
rh_status rh_memset(void *p, int32_t value, size_t bytes, rh_stream) {
    if (bytes) std::memset(p, value, bytes);
    return RH_OK;
}
rh_status rh_memcpy_h2d(void *dst, const void *src, size_t bytes, rh_stream) {
    if (bytes) std::memmove(dst, src, bytes);
    return RH_OK;
}
rh_status rh_memcpy_h2d_rows(void *dst, const void *src, size_t pitch, size_t width, size_t rows, rh_stream) {
    for (size_t r = 0; r < rows; ++r) std::memmove((char *)dst + r * pitch, (const char *)src + r * pitch, width);
    return RH_OK;
}
rh_status rh_memcpy_d2h(void *dst, const void *src, size_t bytes, rh_stream s) { return rh_memcpy_h2d(dst, src, bytes, s); }
rh_status rh_memcpy_d2h_async(void *dst, const void *src, size_t bytes, rh_stream s) { return rh_memcpy_h2d(dst, src, bytes, s); }
rh_status rh_memcpy_d2d(void *dst, const void *src, size_t bytes, rh_stream s) { return rh_memcpy_h2d(dst, src, bytes, s); }
rh_status rh_stream_create(rh_stream *out) {
    *out = std::malloc(8);
    return RH_OK;
}
rh_status rh_stream_destroy(rh_stream s) {
    std::free(s);
    return RH_OK;
}
rh_status rh_stream_synchronize(rh_stream) { return RH_OK; }
rh_status rh_event_create(void **out) {
    *out = std::malloc(8);
    return RH_OK;
}
rh_status rh_event_destroy(void *ev) {
    std::free(ev);
    return RH_OK;
}
rh_status rh_event_record(void *, rh_stream) { return RH_OK; }
rh_status rh_event_synchronize(void *) { return RH_OK; }
rh_status rh_stream_wait_event(rh_stream, void *) { return RH_OK; }

// ---------------------------------------------------------------- math.rs helpers ----
float rh_db_to_linear(float db) { return powf(2.0f, db * 0.05f * LOG2_10); }
float rh_linear_to_db(float lin) { return log2f(lin) * LOG10_2 * 20.0f; }
float rh_duration_to_coefficient(uint64_t ns, uint32_t rate) { return expf(-1.0f / (duration_to_float(ns) * (float)rate)); }
uint64_t rh_delay_samples(uint64_t ns, uint32_t rate, uint32_t ch) { return (uint64_t)((unsigned __int128)ns * ch * rate / 1000000000ull); }
rh_status rh_spatial_gains(const float emitter[3], const float left[3], const float right[3], float out[2]) {  // spatial.rs:19-24, :48-69
    auto dist_sq = [](const float *a, const float *b) {
        float s = 0.0f;
        for (int k = 0; k < 3; ++k) s += (a[k] - b[k]) * (a[k] - b[k]);
        return s;
    };
    const float lsq = dist_sq(left, emitter), rsq = dist_sq(right, emitter), max_diff = std::sqrt(dist_sq(left, right)), ld = std::sqrt(lsq), rd = std::sqrt(rsq);
    out[0] = std::fmin(((ld - rd) / max_diff + 1.0f) / 4.0f + 0.5f, 1.0f) * std::fmin(1.0f / lsq, 1.0f);
    out[1] = std::fmin(((rd - ld) / max_diff + 1.0f) / 4.0f + 0.5f, 1.0f) * std::fmin(1.0f / rsq, 1.0f);
    return RH_OK;
}

// ---------------------------------------------------------------- elementwise ----
rh_status rh_amplify(float *dst, const float *src, size_t n, float f, rh_stream) {
    for (size_t i = 0; i < n; ++i) dst[i] = src[i] * f;
    return RH_OK;
}
rh_status rh_channels_convert(float *dst, const float *src, size_t frames, uint32_t from, uint32_t to, rh_stream) {  // channels.rs:57-85
    if (!from || !to) return RH_ERR_INVALID;
    for (size_t f = 0; f < frames; ++f)
        for (uint32_t k = 0; k < to; ++k) dst[f * to + k] = k < from ? src[f * from + k] : (k == 1 && from == 1 ? src[f * from] : 0.0f);
    return RH_OK;
}
rh_status rh_channel_volume(float *dst, const float *src, size_t frames, uint32_t in_ch, const float *gains, uint32_t out_ch, rh_stream) {  // channel_volume.rs:71-88
    for (size_t f = 0; f < frames; ++f) {
        float sum = 0.0f;
        for (uint32_t c = 0; c < in_ch; ++c) sum += src[f * in_ch + c];
        const float m = sum / (float)in_ch;
        for (uint32_t k = 0; k < out_ch; ++k) dst[f * out_ch + k] = m * gains[k];
    }
    return RH_OK;
}
rh_status rh_mix_sum(float *dst, size_t out_len, const float *const *srcs, const uint64_t *start, const uint64_t *len, uint32_t n, rh_stream) {  // mixer.rs:185-198
    for (size_t t = 0; t < out_len; ++t) {
        float sum = 0.0f;
        for (uint32_t s = 0; s < n; ++s)
            if (t >= start[s] && t < start[s] + len[s]) sum += srcs[s][t - start[s]];
        dst[t] = sum;
    }
    return RH_OK;
}
rh_status rh_wide_mix_block(float *dst, uint32_t channels, uint32_t to_rate, uint64_t out_frames, const rh_wide_src *srcs, uint32_t n, rh_stream) {  // amplify.rs:64, sample_rate.rs:131-201, channels.rs:57-85, mixer.rs:185-198
    if (out_frames == 0) return RH_OK;
    if (!dst || !channels || !to_rate || (n && !srcs)) return RH_ERR_INVALID;
    for (uint32_t s = 0; s < n; ++s) {
        if (!srcs[s].frames) continue;
        if (!srcs[s].data || !srcs[s].channels || !srcs[s].from_rate || srcs[s].frames > out_frames) return RH_ERR_INVALID;
        const uint32_t g = std::gcd(srcs[s].from_rate, to_rate);
        if ((uint64_t)(srcs[s].from_rate / g) * (to_rate / g) > 0xffffffffull) return RH_ERR_UNSUPPORTED;
        if (srcs[s].phase >= to_rate / g) return RH_ERR_INVALID;
    }
    for (uint64_t j = 0; j < out_frames; ++j)
        for (uint32_t c = 0; c < channels; ++c) {
            float sum = 0.0f;
            for (uint32_t s = 0; s < n; ++s) {
                const rh_wide_src &x = srcs[s];
                if (j >= x.frames) continue;
                const uint32_t g = std::gcd(x.from_rate, to_rate), F = x.from_rate / g, T = to_rate / g;
                const uint64_t p = (uint64_t)x.phase + j * F, i = p / T;
                const uint32_t num = (uint32_t)(p - i * T);
                float v = 0.0f;
                if (c < x.channels || (c == 1 && x.channels == 1)) {
                    const uint32_t k = c < x.channels ? c : 0;
                    const float a = x.data[i * x.channels + k] * x.gain;
                    v = a;
                    if (F != T && i < x.last) {
                        const float b = x.data[(i + 1) * x.channels + k] * x.gain;
                        v = a + (b - a) * (float)num / (float)T;
                    }
                }
                sum += v;
            }
            dst[j * channels + c] = sum;
        }
    return RH_OK;
}
rh_status rh_distortion(float *dst, const float *src, size_t n, float gain, float threshold, rh_stream) {  // distortion.rs:66-72
    if (!(threshold >= 0.0f)) return RH_ERR_INVALID;  // (f32::clamp panics on min > max and on NaN)
    for (size_t i = 0; i < n; ++i) {
        float v = src[i] * gain;
        if (v < -threshold) v = -threshold;
        if (v > threshold) v = threshold;
        dst[i] = v;
    }
    return RH_OK;
}
// dither.rs:217-242 with the counter-based noise rodio_hip.h states for rh_dither (the noise of sample k = sample_offset + i is a function of (seed, k))
static uint64_t dither_mix(uint64_t z) {
    z ^= z >> 30;
    z *= 0xbf58476d1ce4e5b9ull;
    z ^= z >> 27;
    z *= 0x94d049bb133111ebull;
    z ^= z >> 31;
    return z;
}
rh_status rh_dither(float *dst, const float *src, size_t n, uint64_t sample_offset, uint32_t channels, uint32_t target_bits, int32_t algorithm, uint64_t seed, rh_stream) {
    if (!channels || target_bits < 1 || target_bits > 32 || algorithm < 0 || algorithm > 3) return RH_ERR_INVALID;
    const float lsb = (float)(1.0 / (double)(1ull << (target_bits - 1)));
    auto bits_at = [seed](uint64_t k) { return dither_mix(seed ^ dither_mix(k + 1)); };
    auto u1 = [](uint64_t h) { return (float)((int32_t)(h >> 40) - 8388608) * 1.1920928955078125e-07f; };
    auto u2 = [](uint64_t h) { return (float)((int32_t)((h >> 16) & 0xffffffu) - 8388608) * 1.1920928955078125e-07f; };
    for (size_t i = 0; i < n; ++i) {
        const uint64_t k = sample_offset + i, h = bits_at(k);
        float noise;
        if (algorithm == 3) {
            noise = (u1(h) + u2(h)) * 0.5f;
        } else if (algorithm == 2) {
            noise = u1(h);
        } else if (algorithm == 1) {
            noise = u1(h) - (k >= channels ? u1(bits_at(k - channels)) : 0.0f);
        } else {
            const float a = (float)((uint32_t)(h >> 40) + 1u) * 5.9604644775390625e-08f, b = (float)((uint32_t)(h >> 16) & 0xffffffu) * 5.9604644775390625e-08f;
            noise = std::sqrt(-2.0f * std::log(a)) * std::cos(6.2831853071795864769f * b) * 0.6f;
        }
        dst[i] = src[i] - noise * lsb;
    }
    return RH_OK;
}
// linear_ramp.rs:79-110, sample by sample from the start of the stream (the entry point is stateless: sample_offset says where the block lies;
// a test's streams are short enough to walk)
rh_status rh_linear_gain_ramp(float *dst, const float *src, size_t n, uint64_t sample_offset, uint32_t channels, uint32_t sample_rate, uint64_t duration_ns, float start_gain, float end_gain,
                              int32_t clamp_end, rh_stream) {
    if (!channels || !sample_rate) return RH_ERR_INVALID;
    uint64_t elapsed = 0, sample_idx = 0;
    const uint64_t dt = 1000000000ull / sample_rate;
    for (uint64_t k = 0; k < sample_offset + n; ++k) {
        float factor;
        if (elapsed >= duration_ns) {
            factor = clamp_end ? end_gain : 1.0f;
        } else {
            sample_idx += 1;
            const float p = duration_to_float(elapsed) / duration_to_float(duration_ns);
            factor = start_gain * (1.0f - p) + end_gain * p;
        }
        if (sample_idx % channels == 0) elapsed += dt;
        if (k >= sample_offset) dst[k - sample_offset] = src[k - sample_offset] * factor;
    }
    return RH_OK;
}
rh_status rh_delay(float *dst, const float *src, uint64_t n, uint64_t delay_samples, rh_stream) {  // delay.rs:68-75: the silence, then the input
    for (uint64_t i = 0; i < delay_samples; ++i) dst[i] = 0.0f;
    for (uint64_t i = 0; i < n; ++i) dst[delay_samples + i] = src[i];
    return RH_OK;
}
// take.rs:96-148: samples while remaining >= duration_per_sample, the fade-out filter before the decrement, the cut frame completed with silence
rh_status rh_take_duration_from(float *dst, const float *src, uint64_t n, uint64_t remaining_ns, uint64_t requested_ns, uint32_t frame_phase, uint32_t channels, uint32_t sample_rate, int32_t fade_out,
                                uint64_t *out_samples, int32_t *ended, uint64_t *remaining_after_ns, rh_stream) {
    if (!channels || !sample_rate || !out_samples || frame_phase >= channels) return RH_ERR_INVALID;
    const uint64_t dps = 1000000000ull / ((uint64_t)sample_rate * channels);
    if (!dps) return RH_ERR_UNSUPPORTED;
    uint64_t remaining = remaining_ns, m = 0;
    bool expired = false;
    for (uint64_t i = 0;; ++i) {
        if (remaining < dps) {  // (also right behind the block's last sample: rodio's next call would answer None without pulling)
            expired = true;
            const uint64_t in_frame = (frame_phase + i) % channels;
            for (uint64_t z = in_frame ? channels - in_frame : 0; z > 0; --z) dst[m++] = 0.0f;
            break;
        }
        if (i == n) break;
        float v = src[i];
        if (fade_out) v = v * (float)(remaining / 1000000ull) / (float)(requested_ns / 1000000ull);
        remaining -= dps;
        dst[m++] = v;
    }
    *out_samples = m;
    if (ended) *ended = expired ? 1 : 0;
    if (remaining_after_ns) *remaining_after_ns = remaining;
    return RH_OK;
}
rh_status rh_take_duration(float *dst, const float *src, uint64_t n, uint64_t sample_offset, uint32_t channels, uint32_t sample_rate, uint64_t duration_ns, int32_t fade_out, uint64_t *out_samples,
                           int32_t *ended, rh_stream s) {
    if (!channels || !sample_rate || !out_samples) return RH_ERR_INVALID;
    const uint64_t dps = 1000000000ull / ((uint64_t)sample_rate * channels);
    if (!dps) return RH_ERR_UNSUPPORTED;
    const uint64_t done = std::min(sample_offset, duration_ns / dps);
    return rh_take_duration_from(dst, src, n, duration_ns - done * dps, duration_ns, (uint32_t)(done % channels), channels, sample_rate, fade_out, out_samples, ended, nullptr, s);
}

// ---------------------------------------------------------------- BltFilter ----
rh_status rh_biquad_coeffs(int32_t kind, uint32_t freq, float q, uint32_t fs, float out[5]) {  // blt.rs:502-544
    if (!fs || !out || kind < 0 || kind > 1) return RH_ERR_INVALID;
    const float w0 = 2.0f * PI_F * (float)freq / (float)fs;
    float rb0, rb1, rb2, ra0, ra1, ra2;
    if (kind == 0) {
        const float alpha = sinf(w0) / (2.0f * q);
        rb1 = 1.0f - cosf(w0);
        rb0 = rb1 / 2.0f;
        rb2 = rb0;
        ra0 = 1.0f + alpha;
        ra1 = -2.0f * cosf(w0);
        ra2 = 1.0f - alpha;
    } else {
        const float cw = cosf(w0), alpha = sinf(w0) / (2.0f * q);
        rb0 = (1.0f + cw) / 2.0f;
        rb1 = -1.0f - cw;
        rb2 = rb0;
        ra0 = 1.0f + alpha;
        ra1 = -2.0f * cw;
        ra2 = 1.0f - alpha;
    }
    out[0] = rb0 / ra0, out[1] = rb1 / ra0, out[2] = rb2 / ra0, out[3] = ra1 / ra0, out[4] = ra2 / ra0;
    return RH_OK;
}
int32_t rh_filter_scan_ok(int32_t kind, uint32_t freq, float q, uint32_t rate) {  // the library's rule, restated (rh_recurrence.hip)
    float c[5];
    if (rh_biquad_coeffs(kind, freq, q, rate, c) != RH_OK) return 0;
    const double a1 = c[3], a2 = c[4], disc = a1 * a1 - 4.0 * a2;
    const double r = disc >= 0.0 ? fmax(fabs((-a1 + sqrt(disc)) / 2.0), fabs((-a1 - sqrt(disc)) / 2.0)) : sqrt(a2 > 0.0 ? a2 : 0.0);
    return (1.0 - r) >= (kind == 0 ? 0.0125 : 0.075) ? 1 : 0;
}
// state: 4 floats per channel {x1, x2, y1, y2}; both modes run the reference's order here
rh_status rh_biquad(float *dst, const float *src, uint64_t frames, uint32_t ch, uint32_t n_streams, const float co[5], float *state, int32_t, rh_stream) {
    if (!ch || !co) return RH_ERR_INVALID;
    for (uint32_t s = 0; s < n_streams; ++s) {
        std::vector<float> st(4 * ch, 0.0f);
        if (state) std::memcpy(st.data(), state + (size_t)s * 4 * ch, 4 * ch * sizeof(float));
        const float *x = src + (size_t)s * frames * ch;
        float *y = dst + (size_t)s * frames * ch;
        for (uint64_t f = 0; f < frames; ++f)
            for (uint32_t c = 0; c < ch; ++c) {
                float *q = st.data() + 4 * c;
                const float in = x[f * ch + c];
                const float r = co[0] * in + co[1] * q[0] + co[2] * q[1] - co[3] * q[2] - co[4] * q[3];  // blt.rs:559
                q[3] = q[2], q[1] = q[0], q[2] = r, q[0] = in;
                y[f * ch + c] = r;
            }
        if (state) std::memcpy(state + (size_t)s * 4 * ch, st.data(), 4 * ch * sizeof(float));
    }
    return RH_OK;
}

// ---------------------------------------------------------------- Limit ----
rh_status rh_limit(float *dst, const float *src, uint64_t frames, uint32_t ch, uint32_t rate, uint32_t n_streams, const rh_limit_params *p, float *state, rh_stream) {
    if (!p || !ch) return RH_ERR_INVALID;
    const float attack = rh_duration_to_coefficient(p->attack_ns, rate), release = rh_duration_to_coefficient(p->release_ns, rate);
    const float thr = p->threshold_db, knee = p->knee_width_db, inv8 = 1.0f / (8.0f * knee);
    for (uint32_t s = 0; s < n_streams; ++s) {
        std::vector<float> st(2 * ch, 0.0f);  // {integrator, peak} per channel
        if (state) std::memcpy(st.data(), state + (size_t)s * 2 * ch, 2 * ch * sizeof(float));
        for (uint64_t f = 0; f < frames; ++f)
            for (uint32_t c = 0; c < ch; ++c) {
                const float x = src[((size_t)s * frames + f) * ch + c];
                const float bias = rh_linear_to_db(fabsf(x) + std::numeric_limits<float>::min()) - thr, kb = bias * 2.0f;
                const float g = kb < -knee ? 0.0f : (fabsf(kb) <= knee ? (kb + knee) * (kb + knee) * inv8 : bias);
                float &I = st[2 * c], &P = st[2 * c + 1];
                I = fmaxf(g, release * I + (1.0f - release) * g);
                P = attack * P + (1.0f - attack) * I;
                float mx;
                if (ch == 1) mx = st[1];
                else if (ch == 2) mx = fmaxf(st[1], st[3]);
                else {
                    mx = 0.0f;
                    for (uint32_t k = 0; k < ch; ++k) mx = fmaxf(mx, st[2 * k + 1]);
                }
                dst[((size_t)s * frames + f) * ch + c] = x * rh_db_to_linear(-mx);
            }
        if (state) std::memcpy(state + (size_t)s * 2 * ch, st.data(), 2 * ch * sizeof(float));
    }
    return RH_OK;
}

// ---------------------------------------------------------------- AGC ----
namespace {
constexpr size_t kWin = 8192;
struct AgcState {  // laid out in the caller's floats: gain, peak, sum, index, window[8192]
    float gain, peak, sum, index;
    float win[kWin];
};
}  // namespace
size_t rh_agc_state_floats(void) { return sizeof(AgcState) / sizeof(float); }
rh_status rh_agc_state_init(float *state, uint32_t n_streams, rh_stream) {
    for (uint32_t s = 0; s < n_streams; ++s) {
        AgcState *a = reinterpret_cast<AgcState *>(state) + s;
        std::memset(a, 0, sizeof *a);
        a->gain = 1.0f;
    }
    return RH_OK;
}
rh_status rh_agc(float *dst, const float *src, uint64_t n, uint32_t rate, uint32_t n_streams, const rh_agc_params *p, float *state, rh_stream) {
    if (!p) return RH_ERR_INVALID;
    const uint64_t ten = 10000000000ull;
    const float attack = rh_duration_to_coefficient(std::min(p->attack_ns, ten), rate), release = rh_duration_to_coefficient(std::min(p->release_ns, ten), rate);
    for (uint32_t s = 0; s < n_streams; ++s) {
        AgcState local;
        AgcState *a = state ? reinterpret_cast<AgcState *>(state) + s : &local;
        if (!state) {
            std::memset(&local, 0, sizeof local);
            local.gain = 1.0f;
        }
        for (uint64_t i = 0; i < n; ++i) {  // agc.rs:433-504
            const float x = src[(size_t)s * n + i], v = fabsf(x);
            const float coeff = v > a->peak ? 0.0f : release;
            a->peak = a->peak * coeff + v * (1.0f - coeff);
            const float sq = v * v;
            size_t idx = (size_t)a->index;
            a->sum = a->sum - a->win[idx] + sq;
            a->win[idx] = sq;
            a->index = (float)((idx + 1) & (kWin - 1));
            const float rms = sqrtf(a->sum / (float)kWin);
            const float rms_gain = rms > 0.0f ? p->target_level / rms : p->absolute_max_gain;
            const float peak_gain = a->peak > 0.0f ? fminf(p->target_level / a->peak, p->absolute_max_gain) : p->absolute_max_gain;
            const float desired = fmaxf(fminf(rms_gain, peak_gain), p->floor);
            const float speed = desired > a->gain ? attack : release;
            a->gain = a->gain * speed + desired * (1.0f - speed);
            if (a->gain < 0.1f) a->gain = 0.1f;
            else if (a->gain > p->absolute_max_gain) a->gain = p->absolute_max_gain;
            dst[(size_t)s * n + i] = x * a->gain;
        }
    }
    return RH_OK;
}

// ---------------------------------------------------------------- SampleRateConverter / reverb, block streaming ----
struct rh_resampler {
    uint32_t from, to, ch;
    uint64_t fed = 0, m = 0;      // input frames received / output frames emitted
    std::vector<float> hist;      // frames [base, fed) still needed
    uint64_t base = 0;
};
rh_status rh_resampler_create(rh_resampler **out, uint32_t from, uint32_t to, uint32_t ch) {
    if (!out || !from || !to || !ch) return RH_ERR_INVALID;
    *out = new rh_resampler{from, to, ch, 0, 0, {}, 0};
    return RH_OK;
}
rh_status rh_resampler_destroy(rh_resampler *p) {
    delete p;
    return RH_OK;
}
rh_status rh_resampler_process(rh_resampler *p, float *dst, uint64_t cap, const float *src, uint64_t in_frames, int32_t flush, uint64_t *out_frames, rh_stream) {
    const Ratio r = reduce(p->from, p->to);
    p->hist.insert(p->hist.end(), src, src + in_frames * p->ch);
    p->fed += in_frames;
    uint64_t n = 0;
    for (;; ++p->m, ++n) {
        const uint64_t i = (uint64_t)(((unsigned __int128)p->m * r.F) / r.T);
        const uint32_t num = (uint32_t)(((unsigned __int128)p->m * r.F) % r.T);
        const float *a = p->hist.data() + (i - p->base) * p->ch;
        if (r.F == r.T ? i < p->fed : i + 1 < p->fed) {
            if (n >= cap) return RH_ERR_CAPACITY;
            for (uint32_t c = 0; c < p->ch; ++c) dst[n * p->ch + c] = r.F == r.T ? a[c] : lerp(a[c], a[p->ch + c], num, (uint32_t)r.T);
        } else if (flush && p->fed && i == p->fed - 1) {  // the last frame verbatim, then the end (sample_rate.rs:193-200)
            if (n >= cap) return RH_ERR_CAPACITY;
            for (uint32_t c = 0; c < p->ch; ++c) dst[n * p->ch + c] = a[c];
            ++n, ++p->m;
            break;
        } else break;
    }
    const uint64_t keep = std::min<uint64_t>((uint64_t)(((unsigned __int128)p->m * r.F) / r.T), p->fed);
    p->hist.erase(p->hist.begin(), p->hist.begin() + (keep - p->base) * p->ch);
    p->base = keep;
    *out_frames = n;
    return RH_OK;
}
struct rh_echo {
    uint64_t d;
    float gain;
    std::vector<float> tail;  // the last d input samples (what the delayed clone still has to play)
    uint64_t seen = 0;
};
rh_status rh_echo_create(rh_echo **out, uint64_t d, float gain) {
    *out = new rh_echo{d, gain, {}, 0};
    return RH_OK;
}
rh_status rh_echo_destroy(rh_echo *p) {
    delete p;
    return RH_OK;
}
rh_status rh_echo_process(rh_echo *p, float *dst, const float *src, uint64_t n, rh_stream) {  // y[k] = x[k] + (k < d ? 0.0 : gain * x[k - d])   (mix.rs:43-53)
    std::vector<float> all(p->tail);
    all.insert(all.end(), src, src + n);
    const uint64_t off = p->tail.size();  // all[off + i] = x[seen + i]
    for (uint64_t i = 0; i < n; ++i) {
        const uint64_t k = p->seen + i;
        const float echo = k < p->d ? 0.0f : all[off + i - p->d] * p->gain;
        dst[i] = src[i] + echo;
    }
    p->seen += n;
    const uint64_t keep = std::min<uint64_t>(p->d, all.size());
    p->tail.assign(all.end() - keep, all.end());
    return RH_OK;
}
rh_status rh_echo_flush(rh_echo *p, float *dst, rh_stream) {  // the d samples of the delayed clone that outlive the source
    for (uint64_t j = 0; j < p->d; ++j) {
        const uint64_t k = p->seen + j;  // output position; the clone plays x[k - d]
        const int64_t src_idx = (int64_t)k - (int64_t)p->d;  // < seen
        float v = 0.0f;  // (positions before the clone starts: Delay's zeros, amplified -> 0.0)
        if (src_idx >= 0) v = p->tail[p->tail.size() - (p->seen - (uint64_t)src_idx)] * p->gain;
        dst[j] = v;
    }
    return RH_OK;
}

// ---------------------------------------------------------------- UniformSourceIterator, span by span ----
rh_status rh_resample_out_frames(uint64_t in_frames, uint32_t from, uint32_t to, uint32_t ch, uint64_t span_len, uint64_t *out) {
    if (!out || !from || !to || !ch) return RH_ERR_INVALID;
    const Ratio r = reduce(from, to);
    auto one = [&](uint64_t n) -> uint64_t { return r.F == r.T ? n : (n == 0 ? 0 : lerp_ready(n, r.F, r.T) + ((unsigned __int128)lerp_ready(n, r.F, r.T) * r.F < (unsigned __int128)n * r.T ? 1 : 0)); };
    if (!span_len) {
        *out = one(in_frames);
        return RH_OK;
    }
    const uint64_t sf = std::min<uint64_t>(span_len, 32768) / ch;
    uint64_t m = 0;
    for (uint64_t f = 0; f < in_frames; f += sf) m += one(std::min(sf, in_frames - f));
    *out = m;
    return RH_OK;
}
rh_status rh_uniform_span_frames(uint64_t n, uint32_t from, uint32_t to, int32_t complete, uint64_t *out) {
    if (!out || !from || !to) return RH_ERR_INVALID;
    const Ratio r = reduce(from, to);
    if (r.F * r.T > 0xffffffffull) return RH_ERR_UNSUPPORTED;
    if (r.F == r.T) {
        *out = n;
        return RH_OK;
    }
    uint64_t c = lerp_ready(n, r.F, r.T);
    if (complete && n && (unsigned __int128)c * r.F < (unsigned __int128)n * r.T) c += 1;
    *out = c;
    return RH_OK;
}
rh_status rh_uniform_first_tap(uint64_t m, uint32_t from, uint32_t to, uint64_t *in_frame) {
    if (!in_frame || !from || !to) return RH_ERR_INVALID;
    const Ratio r = reduce(from, to);
    *in_frame = (uint64_t)(((unsigned __int128)m * r.F) / r.T);
    return RH_OK;
}
// The tail of a span that ends inside a frame (see include/rodio_hip.h, rh_uniform_cut_tail_samples).
static void cut_counts(uint64_t q, uint64_t F, uint64_t T, uint64_t &m_first, uint64_t &k, uint64_t &e) {
    if (F == T) {
        m_first = q, k = 0, e = 1;
        return;
    }
    m_first = lerp_ready(q, F, T);
    k = q >= 1 ? lerp_ready(q + 1, F, T) - lerp_ready(q, F, T) : 0;
    e = lerp_ready(q + 2, F, T) - lerp_ready(q + 1, F, T) >= 1 ? 1 : 0;
}
rh_status rh_uniform_cut_tail_samples(uint64_t q, uint32_t t, uint32_t from_rate, uint32_t to_rate, uint32_t from_ch, uint32_t to_ch, uint64_t *out) {
    if (!out || !from_rate || !to_rate || !from_ch || !to_ch || t == 0 || t >= from_ch) return RH_ERR_INVALID;
    const Ratio r = reduce(from_rate, to_rate);
    if (r.F * r.T > 0xffffffffull) return RH_ERR_UNSUPPORTED;
    uint64_t mf, k, e;
    cut_counts(q, r.F, r.T, mf, k, e);
    const uint64_t u = (k + e) * t, G = u / from_ch, g = u % from_ch;
    *out = G * to_ch + std::min<uint64_t>(g, to_ch);
    return RH_OK;
}
static void convert_segment(const rh_uniform_seg &g) {
    const Ratio r = reduce(g.from_rate, g.to_rate);
    const uint32_t fc = g.from_ch, tc = g.to_ch, nc = std::min(fc, tc);
    if (g.reserved) {  // cut tail: output SAMPLES [m0, m1)
        const uint32_t t = g.reserved;
        const uint64_t q = g.span_frames;
        uint64_t mf, k, e;
        cut_counts(q, r.F, r.T, mf, k, e);
        const float *last = g.src;                          // the frame in front of the cut (read only when k > 0: then the host put it there)
        const float *p = g.src + g.src_frames * fc;         // the cut frame's t samples (src_frames: 0 or 1, as rh_uniform.hip reads it)
        for (uint64_t j = g.m0; j < g.m1; ++j) {
            const uint64_t grp = j / tc, pos = j % tc;
            float v = 0.0f;
            if (pos < fc) {
                const uint64_t idx = grp * fc + pos, rr = idx / t, c = idx % t;
                if (rr < k) {
                    const uint64_t m = mf + rr;
                    const uint32_t num = (uint32_t)(((unsigned __int128)m * r.F) % r.T);
                    v = lerp(last[c] * g.gain, p[c] * g.gain, num, (uint32_t)r.T);
                } else v = p[c] * g.gain;
            }
            g.dst[j - g.m0] = v;
        }
        return;
    }
    for (uint64_t m = g.m0; m < g.m1; ++m) {
        float *o = g.dst + (m - g.m0) * tc;
        const uint64_t i = r.F == r.T ? m : (uint64_t)(((unsigned __int128)m * r.F) / r.T);
        const uint32_t num = r.F == r.T ? 0 : (uint32_t)(((unsigned __int128)m * r.F) % r.T);
        const bool verbatim = r.F == r.T || i + 1 >= g.span_frames;
        const float *a = g.src + (i - g.src_frame0) * fc;
        for (uint32_t k = 0; k < nc; ++k) {
            const float av = a[k] * g.gain;
            o[k] = verbatim ? av : lerp(av, a[fc + k] * g.gain, num, (uint32_t)r.T);
        }
        if (tc > fc) {
            if (fc == 1) o[1] = o[0];
            for (uint32_t k = (fc == 1 ? 2u : fc); k < tc; ++k) o[k] = 0.0f;
        }
    }
}
rh_status rh_uniform_segments(const rh_uniform_seg *segs, uint32_t n, rh_stream) {
    // On the device the segments of a launch run side by side: a table whose result depends on their order is a bug of the host.  Here they run
    // LAST TO FIRST, so that a segment that overwrites what an earlier one wrote shows (round 5: the copy of a row's left-over frames took one
    // sample too many, which belonged to the next segment -- invisible first to last).
    for (uint32_t k = n; k-- > 0;) {
        const rh_uniform_seg &g = segs[k];
        if (!g.from_rate || !g.to_rate || !g.from_ch || !g.to_ch || g.m1 < g.m0) return RH_ERR_INVALID;
        if (g.m1 > g.m0 && (!g.src || !g.dst)) return RH_ERR_INVALID;
        convert_segment(g);
    }
    return RH_OK;
}
rh_status rh_uniform_segments_dev(const rh_uniform_seg *segs, uint32_t n, uint64_t, rh_stream s) { return rh_uniform_segments(segs, n, s); }

// ---------------------------------------------------------------- the fused stream, by its contract ----
struct rh_rlm {
    rh_rlm_config cfg;
    std::vector<float> gains;
    bool on = false, done = false;
    uint64_t g0 = 0, m = 0;  // global input frame of the rows' first frame; output frames emitted
    std::vector<uint64_t> total;           // input frames of a source once it has ended (~0: live)
    std::vector<std::vector<float>> state;  // biquad state per source
    float co[5];
};
rh_status rh_rlm_create(rh_rlm **out, const rh_rlm_config *cfg) {
    if (!out || !cfg || !cfg->from_rate || !cfg->to_rate || (cfg->channels != 1 && cfg->channels != 2)) return RH_ERR_INVALID;
    const Ratio r = reduce(cfg->from_rate, cfg->to_rate);
    if (2 * r.F > 9 * r.T || r.F * r.T > 0xffffffffull) return RH_ERR_UNSUPPORTED;
    rh_rlm *p = new rh_rlm();
    p->cfg = *cfg;
    if (cfg->filter_kind == 0 || cfg->filter_kind == 1) (void)rh_biquad_coeffs(cfg->filter_kind, cfg->filter_freq, cfg->filter_q, cfg->to_rate, p->co);
    *out = p;
    return RH_OK;
}
rh_status rh_rlm_destroy(rh_rlm *p) {
    delete p;
    return RH_OK;
}
rh_status rh_rlm_set_exclusive(rh_rlm *, int32_t) { return RH_OK; }
rh_status rh_rlm_set_gains(rh_rlm *p, const float *g, uint32_t n) {
    p->gains.assign(g, g + n);
    return RH_OK;
}
rh_status rh_rlm_stream_begin(rh_rlm *p) {
    p->on = true, p->done = false, p->g0 = p->m = 0;
    p->total.clear();
    p->state.clear();
    return RH_OK;
}
rh_status rh_rlm_stream_keep_history(rh_rlm *, int32_t) { return RH_OK; }
rh_status rh_rlm_stream_overlap(rh_rlm *, int32_t) { return RH_OK; }
rh_status rh_rlm_stream_one_launch_blocks(rh_rlm *, uint32_t *blocks) {
    if (blocks) *blocks = 0;
    return RH_OK;
}
rh_status rh_rlm_stream_overlapped_blocks(rh_rlm *, uint32_t *blocks) {
    if (blocks) *blocks = 0;
    return RH_OK;
}
rh_status rh_rlm_last_status(rh_rlm *) { return RH_OK; }
rh_status rh_rlm_stream_block_v(rh_rlm *p, const float *const *srcs, const uint64_t *avail, const uint8_t *ended, uint32_t S, float *dst, uint64_t cap, uint64_t *out_frames, uint64_t *consumed, rh_stream) {
    if (!p || !p->on || p->done || !out_frames || !consumed) return RH_ERR_INVALID;
    if (S == 0 || S > p->cfg.max_sources) return RH_ERR_CAPACITY;
    const Ratio r = reduce(p->cfg.from_rate, p->cfg.to_rate);
    const uint32_t C = p->cfg.channels;
    const uint64_t L = 64ull * (p->cfg.frames_per_lane ? p->cfg.frames_per_lane : 4);
    if (p->total.empty()) {
        p->total.assign(S, ~0ull);
        p->state.assign(S, std::vector<float>(4 * C, 0.0f));
    }
    auto ready = [&](uint64_t N) { return r.F == r.T ? N : lerp_ready(N, r.F, r.T); };
    auto total_out = [&](uint64_t N) {
        if (r.F == r.T) return N;
        const uint64_t c = lerp_ready(N, r.F, r.T);
        return c + (N && (unsigned __int128)c * r.F < (unsigned __int128)N * r.T ? 1 : 0);
    };
    uint64_t live_min = ~0ull, ended_max = 0;
    bool any_live = false;
    for (uint32_t s = 0; s < S; ++s) {
        if (avail[s] > p->cfg.max_in_frames) return RH_ERR_CAPACITY;
        if (p->total[s] == ~0ull && ended[s]) p->total[s] = p->g0 + avail[s];
        if (p->total[s] == ~0ull) {
            const uint64_t e = ready(p->g0 + avail[s]);
            live_min = std::min(live_min, e > p->m ? e - p->m : 0);
            any_live = true;
        } else {
            const uint64_t M = total_out(p->total[s]);
            ended_max = std::max(ended_max, M > p->m ? M - p->m : 0);
        }
    }
    const bool final_block = !any_live;
    const uint64_t out = final_block ? ended_max : live_min / L * L;
    if (out > cap) return RH_ERR_CAPACITY;
    for (uint64_t k = 0; k < out; ++k) {
        const uint64_t m = p->m + k;
        float acc[2] = {0.0f, 0.0f};
        for (uint32_t s = 0; s < S; ++s) {
            if (p->total[s] != ~0ull && m >= total_out(p->total[s])) continue;
            const uint64_t i = r.F == r.T ? m : (uint64_t)(((unsigned __int128)m * r.F) / r.T);
            const uint32_t num = r.F == r.T ? 0 : (uint32_t)(((unsigned __int128)m * r.F) % r.T);
            const float *a = srcs[s] + (i - p->g0) * C;
            const bool verbatim = r.F == r.T || (p->total[s] != ~0ull && i + 1 >= p->total[s]);
            const float g = s < p->gains.size() ? p->gains[s] : 1.0f;
            for (uint32_t c = 0; c < C; ++c) {
                const float av = a[c] * g;
                float v = verbatim ? av : lerp(av, a[C + c] * g, num, (uint32_t)r.T);
                if (p->cfg.filter_kind == 0 || p->cfg.filter_kind == 1) {
                    float *q = p->state[s].data() + 4 * c;
                    const float y = p->co[0] * v + p->co[1] * q[0] + p->co[2] * q[1] - p->co[3] * q[2] - p->co[4] * q[3];
                    q[3] = q[2], q[1] = q[0], q[2] = y, q[0] = v;
                    v = y;
                }
                acc[c] += v;
            }
        }
        for (uint32_t c = 0; c < C; ++c) dst[k * C + c] = acc[c];
    }
    p->m += out;
    if (final_block) {
        p->done = true;
        uint64_t mx = 0;
        for (uint32_t s = 0; s < S; ++s) mx = std::max(mx, avail[s]);
        *consumed = mx;
    } else {
        const uint64_t keep_from = p->m >= 2 ? (r.F == r.T ? p->m - 2 : (uint64_t)(((unsigned __int128)(p->m - 2) * r.F) / r.T)) : 0;
        *consumed = keep_from > p->g0 ? keep_from - p->g0 : 0;
        p->g0 += *consumed;
    }
    *out_frames = out;
    return RH_OK;
}

}  // extern "C"
