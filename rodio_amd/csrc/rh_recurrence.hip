// rh_recurrence.hip -- the per-stream recurrences, executed in the reference's own order:
//   BltFilter   src/source/blt.rs:502-544 (coefficients), :558-560 (apply), :397-492 (state)
//   Limit       src/source/limit.rs:94-130, :853-873, :903-916, :927-988
//   AGC         src/source/agc.rs:133-171, :397-504
//
// "mode 0" kernels: one lane per independent stream, time-sequential, identical operation
// order to the Rust iterators (this TU is built with -ffp-contract=off), so the biquad is
// bit-exact and limiter/AGC differ only through log2f/exp2f/sqrtf library rounding.  Streams
// are the only parallel axis here (S sources x C channels for the biquad, S for limiter/AGC),
// so these kernels are latency-bound by construction; the time-parallel biquad used by the
// headline pipeline lives in rh_pipeline.hip.
#include <cmath>
#include <cstdlib>

#include "rh_common.h"

namespace {

constexpr int kBlock = 64;  // one wave per workgroup: spread few streams over many CUs

struct Biquad5 {
    float b0, b1, b2, a1, a2;
};

__global__ __launch_bounds__(kBlock) void k_biquad_seq(float *__restrict__ dst, const float *__restrict__ src, uint64_t frames, uint32_t channels, uint32_t n_streams, Biquad5 k, float *__restrict__ state) {
    const uint32_t id = blockIdx.x * kBlock + threadIdx.x;
    if (id >= n_streams * channels) return;
    const uint32_t stream = id / channels, c = id - stream * channels;
    const float *x = src + (uint64_t)stream * frames * channels + c;
    float *y = dst + (uint64_t)stream * frames * channels + c;
    float x1 = 0.f, x2 = 0.f, y1 = 0.f, y2 = 0.f;
    float *st = state ? state + (uint64_t)id * 4 : nullptr;
    if (st) {
        x1 = st[0];
        x2 = st[1];
        y1 = st[2];
        y2 = st[3];
    }
    for (uint64_t n = 0; n < frames; ++n) {
        const float xn = x[n * channels];
        // blt.rs:559, left to right
        const float r = k.b0 * xn + k.b1 * x1 + k.b2 * x2 - k.a1 * y1 - k.a2 * y2;
        y2 = y1;
        x2 = x1;
        y1 = r;
        x1 = xn;
        y[n * channels] = r;
    }
    if (st) {
        st[0] = x1;
        st[1] = x2;
        st[2] = y1;
        st[3] = y2;
    }
}

// The same recurrence with the memory side done properly: one lane per STREAM (its C channels are C independent chains,
// which fills the pipeline between the dependent steps), 16-byte loads and stores of 16 samples at a time, the next 16
// samples requested before the current ones are worked on.  k_biquad_seq above fetched 4 bytes per lane and used them at
// once -- a full memory round trip per sample: 1 236 ms for 64 x 1 Mi stereo frames.  The arithmetic (operation order of
// blt.rs:559, no contraction) and hence every output bit is unchanged; what bounds it now is the dependent chain itself
// (3 operations deep per sample), which only more streams can fill.
typedef float v4f_t __attribute__((ext_vector_type(4)));
template <int C>
__global__ __launch_bounds__(kBlock) void k_biquad_vec(float *__restrict__ dst, const float *__restrict__ src, uint64_t frames, uint32_t n_streams, Biquad5 k, float *__restrict__ state) {
    static_assert(16 % C == 0, "a block of 16 samples is whole frames");
    const uint32_t stream = blockIdx.x * kBlock + threadIdx.x;
    if (stream >= n_streams) return;
    const uint64_t total = frames * C;
    const float *x = src + (uint64_t)stream * total;
    float *y = dst + (uint64_t)stream * total;
    float x1[C], x2[C], y1[C], y2[C];
    float *st = state ? state + (uint64_t)stream * C * 4 : nullptr;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        x1[c] = st ? st[4 * c] : 0.f;
        x2[c] = st ? st[4 * c + 1] : 0.f;
        y1[c] = st ? st[4 * c + 2] : 0.f;
        y2[c] = st ? st[4 * c + 3] : 0.f;
    }
    auto step = [&](float xn, int c) {
        const float r = k.b0 * xn + k.b1 * x1[c] + k.b2 * x2[c] - k.a1 * y1[c] - k.a2 * y2[c];  // blt.rs:559, left to right
        y2[c] = y1[c];
        x2[c] = x1[c];
        y1[c] = r;
        x1[c] = xn;
        return r;
    };
    const uint64_t blocks = total / 16;
    const v4f_t *xv = reinterpret_cast<const v4f_t *>(x);
    v4f_t *yv = reinterpret_cast<v4f_t *>(y);
    v4f_t cur[4], nxt[4];
    if (blocks) {
#pragma unroll
        for (int j = 0; j < 4; ++j) cur[j] = __builtin_nontemporal_load(xv + j);
    }
    for (uint64_t b = 0; b < blocks; ++b) {
        if (b + 1 < blocks) {
#pragma unroll
            for (int j = 0; j < 4; ++j) nxt[j] = __builtin_nontemporal_load(xv + (b + 1) * 4 + j);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v4f_t o;
            o.x = step(cur[j].x, (4 * j + 0) % C);
            o.y = step(cur[j].y, (4 * j + 1) % C);
            o.z = step(cur[j].z, (4 * j + 2) % C);
            o.w = step(cur[j].w, (4 * j + 3) % C);
            yv[b * 4 + j] = o;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) cur[j] = nxt[j];
    }
    for (uint64_t i = blocks * 16; i < total; ++i) y[i] = step(x[i], (int)(i % C));
    if (st) {
#pragma unroll
        for (int c = 0; c < C; ++c) {
            st[4 * c] = x1[c];
            st[4 * c + 1] = x2[c];
            st[4 * c + 2] = y1[c];
            st[4 * c + 3] = y2[c];
        }
    }
}

constexpr float LOG2_10 = 3.32192809488736234787f;
constexpr float LOG10_2 = 0.301029995663981195214f;
constexpr int kMaxCh = 8;

struct LimitK {
    float threshold, knee_width, inv_knee_8, attack, release;
};

__device__ __forceinline__ float limit_process_sample(float sample, const LimitK &k) {
    // limit.rs:853-873; f32::MIN_POSITIVE = 2^-126
    const float bias_db = log2f(fabsf(sample) + 1.17549435e-38f) * LOG10_2 * 20.0f - k.threshold;
    const float knee_boundary_db = bias_db * 2.0f;
    if (knee_boundary_db < -k.knee_width) return 0.0f;
    if (fabsf(knee_boundary_db) <= k.knee_width) {
        const float x = knee_boundary_db + k.knee_width;
        return x * x * k.inv_knee_8;
    }
    return bias_db;
}

__global__ __launch_bounds__(kBlock) void k_limit_seq(float *__restrict__ dst, const float *__restrict__ src, uint64_t frames, uint32_t channels, uint32_t n_streams, LimitK k, float *__restrict__ state) {
    const uint32_t stream = blockIdx.x * kBlock + threadIdx.x;
    if (stream >= n_streams) return;
    const uint64_t total = frames * channels;
    const float *x = src + (uint64_t)stream * total;
    float *y = dst + (uint64_t)stream * total;
    float integ[kMaxCh], peak[kMaxCh];
    float *st = state ? state + (uint64_t)stream * channels * 2 : nullptr;
    for (uint32_t c = 0; c < channels; ++c) {
        integ[c] = st ? st[2 * c] : 0.0f;
        peak[c] = st ? st[2 * c + 1] : 0.0f;
    }
    uint32_t c = 0;
    for (uint64_t n = 0; n < total; ++n) {
        const float sample = x[n];
        const float limiter_db = limit_process_sample(sample, k);
        // limit.rs:909-913
        integ[c] = fmaxf(limiter_db, k.release * integ[c] + (1.0f - k.release) * limiter_db);
        peak[c] = k.attack * peak[c] + (1.0f - k.attack) * integ[c];
        float max_peak;
        if (channels == 1) max_peak = peak[0];
        else if (channels == 2) max_peak = fmaxf(peak[0], peak[1]);
        else {
            max_peak = 0.0f;
            for (uint32_t j = 0; j < channels; ++j) max_peak = fmaxf(max_peak, peak[j]);
        }
        // math.rs:51-56: 2^(dB * 0.05 * log2(10))
        y[n] = sample * exp2f(-max_peak * 0.05f * LOG2_10);
        c = (c + 1 == channels) ? 0 : c + 1;
    }
    if (st) {
        for (uint32_t j = 0; j < channels; ++j) {
            st[2 * j] = integ[j];
            st[2 * j + 1] = peak[j];
        }
    }
}

constexpr uint32_t kRmsWindow = 8192;              // agc.rs:51
constexpr size_t kAgcStateFloats = 4 + kRmsWindow;  // {sum, index(bits), peak_level, current_gain, ring[8192]}

struct AgcK {
    float target_level, attack_coeff, release_coeff, absolute_max_gain, floor;
};

__global__ __launch_bounds__(kBlock) void k_agc_seq(float *__restrict__ dst, const float *__restrict__ src, uint64_t n_samples, uint32_t n_streams, AgcK k, float *__restrict__ state, int fresh) {
    const uint32_t stream = blockIdx.x * kBlock + threadIdx.x;
    if (stream >= n_streams) return;
    const float *x = src + (uint64_t)stream * n_samples;
    float *y = dst + (uint64_t)stream * n_samples;
    float *st = state + (uint64_t)stream * kAgcStateFloats;
    float *ring = st + 4;
    float sum, peak_level, current_gain;
    uint32_t index;
    if (fresh) {  // agc.rs:209-236: gain 1.0, peak 0.0, zeroed window
        sum = 0.0f;
        index = 0;
        peak_level = 0.0f;
        current_gain = 1.0f;
        for (uint32_t i = 0; i < kRmsWindow; ++i) ring[i] = 0.0f;
    } else {
        sum = st[0];
        index = __float_as_uint(st[1]) & (kRmsWindow - 1);
        peak_level = st[2];
        current_gain = st[3];
    }
    for (uint64_t n = 0; n < n_samples; ++n) {
        const float sample = x[n];
        const float sample_value = fabsf(sample);
        // update_peak_level, agc.rs:397-407
        const float coeff = sample_value > peak_level ? 0.0f : k.release_coeff;
        peak_level = peak_level * coeff + sample_value * (1.0f - coeff);
        // update_rms, agc.rs:413-417 + CircularBuffer::push :152-163
        const float squared = sample_value * sample_value;
        const float old_value = ring[index];
        sum = sum - old_value + squared;
        ring[index] = squared;
        index = (index + 1) & (kRmsWindow - 1);
        const float rms = sqrtf(sum / (float)kRmsWindow);
        const float rms_gain = rms > 0.0f ? k.target_level / rms : k.absolute_max_gain;
        const float peak_gain = peak_level > 0.0f ? fminf(k.target_level / peak_level, k.absolute_max_gain) : k.absolute_max_gain;
        const float desired_gain = fmaxf(fminf(rms_gain, peak_gain), k.floor);
        const float attack_speed = desired_gain > current_gain ? k.attack_coeff : k.release_coeff;
        current_gain = current_gain * attack_speed + desired_gain * (1.0f - attack_speed);
        current_gain = current_gain < 0.1f ? 0.1f : (current_gain > k.absolute_max_gain ? k.absolute_max_gain : current_gain);
        y[n] = sample * current_gain;
    }
    st[0] = sum;
    st[1] = __uint_as_float(index);
    st[2] = peak_level;
    st[3] = current_gain;
}

// AGC with the memory side done properly: one lane per stream (the recurrences of agc.rs:397-504 branch on their own state
// and the running sum `sum - old + new` must round in the reference's order, so time stays sequential), but 16 samples per
// lane arrive with 16-byte loads one block ahead of their use, the squares that leave the RMS window are recomputed from the
// input 8192 samples back (the same f32 product, bit for bit) instead of a read-modify-write ring, and results leave with
// 16-byte stores.  The ring of the carried state is only read for the first 8192 samples of a block and rewritten once at
// its end.  Same arithmetic as k_agc_seq, same bits.
__global__ __launch_bounds__(kBlock) void k_agc_vec(float *__restrict__ dst, const float *__restrict__ src, uint64_t n_samples, uint32_t n_streams, AgcK k, float *__restrict__ state) {
    const uint32_t stream = blockIdx.x * kBlock + threadIdx.x;
    if (stream >= n_streams) return;
    const float *x = src + (uint64_t)stream * n_samples;
    float *y = dst + (uint64_t)stream * n_samples;
    float *st = state ? state + (uint64_t)stream * kAgcStateFloats : nullptr;
    float *ring = st ? st + 4 : nullptr;
    float sum = 0.0f, peak_level = 0.0f, current_gain = 1.0f;
    uint32_t index = 0;
    if (st) {
        sum = st[0];
        index = __float_as_uint(st[1]) & (kRmsWindow - 1);
        peak_level = st[2];
        current_gain = st[3];
    }
    auto step = [&](float sample, float old_value) {
        const float sample_value = fabsf(sample);
        const float coeff = sample_value > peak_level ? 0.0f : k.release_coeff;  // agc.rs:397-407
        peak_level = peak_level * coeff + sample_value * (1.0f - coeff);
        const float squared = sample_value * sample_value;  // agc.rs:413-417, :152-163
        sum = sum - old_value + squared;
        const float rms = sqrtf(sum / (float)kRmsWindow);
        const float rms_gain = rms > 0.0f ? k.target_level / rms : k.absolute_max_gain;
        const float peak_gain = peak_level > 0.0f ? fminf(k.target_level / peak_level, k.absolute_max_gain) : k.absolute_max_gain;
        const float desired_gain = fmaxf(fminf(rms_gain, peak_gain), k.floor);
        const float attack_speed = desired_gain > current_gain ? k.attack_coeff : k.release_coeff;
        current_gain = current_gain * attack_speed + desired_gain * (1.0f - attack_speed);
        current_gain = current_gain < 0.1f ? 0.1f : (current_gain > k.absolute_max_gain ? k.absolute_max_gain : current_gain);
        return sample * current_gain;
    };
    const uint64_t blocks = n_samples / 16;
    const v4f_t *xv = reinterpret_cast<const v4f_t *>(x);
    v4f_t *yv = reinterpret_cast<v4f_t *>(y);
    // what leaves the window while sample n enters: the square of sample n - 8192 of this block, or, for the block's first
    // 8192 samples, what the carried ring holds (zeros for a fresh AGC)
    auto old_block = [&](uint64_t b, v4f_t (&o)[4]) {
        if (b * 16 >= kRmsWindow) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const v4f_t v = xv[(b * 16 - kRmsWindow) / 4 + j];  // re-read 32 KiB behind: L2
                o[j].x = fabsf(v.x) * fabsf(v.x);
                o[j].y = fabsf(v.y) * fabsf(v.y);
                o[j].z = fabsf(v.z) * fabsf(v.z);
                o[j].w = fabsf(v.w) * fabsf(v.w);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t i0 = index + (uint32_t)(b * 16) + 4 * j;
                o[j].x = ring ? ring[(i0 + 0) & (kRmsWindow - 1)] : 0.0f;
                o[j].y = ring ? ring[(i0 + 1) & (kRmsWindow - 1)] : 0.0f;
                o[j].z = ring ? ring[(i0 + 2) & (kRmsWindow - 1)] : 0.0f;
                o[j].w = ring ? ring[(i0 + 3) & (kRmsWindow - 1)] : 0.0f;
            }
        }
    };
    v4f_t cur[4], nxt[4], ocur[4], onxt[4];
    if (blocks) {
#pragma unroll
        for (int j = 0; j < 4; ++j) cur[j] = xv[j];
        old_block(0, ocur);
    }
    for (uint64_t b = 0; b < blocks; ++b) {
        if (b + 1 < blocks) {
#pragma unroll
            for (int j = 0; j < 4; ++j) nxt[j] = xv[(b + 1) * 4 + j];  // (read again 8192 samples later: no streaming hint)
            old_block(b + 1, onxt);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v4f_t o;
            o.x = step(cur[j].x, ocur[j].x);
            o.y = step(cur[j].y, ocur[j].y);
            o.z = step(cur[j].z, ocur[j].z);
            o.w = step(cur[j].w, ocur[j].w);
            yv[b * 4 + j] = o;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) cur[j] = nxt[j], ocur[j] = onxt[j];
    }
    for (uint64_t i = blocks * 16; i < n_samples; ++i) {
        float old_value;
        if (i >= kRmsWindow) old_value = fabsf(x[i - kRmsWindow]) * fabsf(x[i - kRmsWindow]);
        else old_value = ring ? ring[(index + (uint32_t)i) & (kRmsWindow - 1)] : 0.0f;
        y[i] = step(x[i], old_value);
    }
    if (st) {  // the window the next block starts from: the squares of this block's last 8192 samples, at their ring positions
        const uint64_t from = n_samples > kRmsWindow ? n_samples - kRmsWindow : 0;
        for (uint64_t i = from; i < n_samples; ++i) ring[(index + (uint32_t)i) & (kRmsWindow - 1)] = fabsf(x[i]) * fabsf(x[i]);
        st[0] = sum;
        st[1] = __uint_as_float((index + (uint32_t)n_samples) & (kRmsWindow - 1));
        st[2] = peak_level;
        st[3] = current_gain;
    }
}

// Duration::as_secs_f32 then exp(-1/(t*sr)): math.rs:110-122.  Host side, f32.
float duration_to_coefficient(uint64_t ns, uint32_t sample_rate) {
    const uint64_t secs = ns / 1000000000ull;
    const uint32_t nanos = (uint32_t)(ns % 1000000000ull);
    const float t = (float)secs + (float)nanos / 1000000000.0f;
    return expf(-1.0f / (t * (float)sample_rate));
}

}  // namespace

namespace rh {
// The reference-order limiter (one lane per stream): what rh_limit (rh_limit.hip) takes for rows that are not 16-byte
// aligned or for coefficients outside the scan's premises.  k5 = {threshold, knee_width, inv_knee_8, attack, release}.
rh_status limit_seq_launch(float *dst, const float *src, uint64_t frames, uint32_t channels, uint32_t n_streams, const float k5[5], float *state, hipStream_t s) {
    if (channels > (uint32_t)kMaxCh) return RH_ERR_UNSUPPORTED;
    LimitK k;
    k.threshold = k5[0];
    k.knee_width = k5[1];
    k.inv_knee_8 = k5[2];
    k.attack = k5[3];
    k.release = k5[4];
    hipLaunchKernelGGL(k_limit_seq, dim3((n_streams + kBlock - 1) / kBlock), dim3(kBlock), 0, s, dst, src, frames, channels, n_streams, k, state);
    RH_CHECK_LAUNCH();
    return RH_OK;
}
float duration_to_coefficient_f32(uint64_t ns, uint32_t sample_rate) { return duration_to_coefficient(ns, sample_rate); }
}  // namespace rh

extern "C" {

rh_status rh_biquad_coeffs(int32_t kind, uint32_t freq, float q, uint32_t sample_rate, float out[5]) {
    if (!out || sample_rate == 0 || (kind != 0 && kind != 1)) return RH_ERR_INVALID;
    const float PI_F = 3.14159265358979323846264338327950288f;
    const float w0 = 2.0f * PI_F * (float)freq / (float)sample_rate;
    float b0, b1, b2, a0, a1, a2;
    if (kind == 0) {  // blt.rs:504-521
        const float alpha = sinf(w0) / (2.0f * q);
        b1 = 1.0f - cosf(w0);
        b0 = b1 / 2.0f;
        b2 = b0;
        a0 = 1.0f + alpha;
        a1 = -2.0f * cosf(w0);
        a2 = 1.0f - alpha;
    } else {  // blt.rs:523-542
        const float cos_w0 = cosf(w0);
        const float alpha = sinf(w0) / (2.0f * q);
        b0 = (1.0f + cos_w0) / 2.0f;
        b1 = -1.0f - cos_w0;
        b2 = b0;
        a0 = 1.0f + alpha;
        a1 = -2.0f * cos_w0;
        a2 = 1.0f - alpha;
    }
    out[0] = b0 / a0;
    out[1] = b1 / a0;
    out[2] = b2 / a0;
    out[3] = a1 / a0;
    out[4] = a2 / a0;
    return RH_OK;
}

// Where the time-parallel evaluation of a low_pass / high_pass stays within 1e-5 of rodio's OWN f32 recurrence for a full-scale source
// (|x| <= 1).  The scan is the more accurate of the two (profiles/r04_filter_contract.txt: 400 000 frames of full-scale noise; its
// distance from an f64 evaluation is 3e-8 .. 5e-6 where the reference's is 1e-7 .. 6e-3), so |scan - reference| IS the reference's
// rounding noise -- the recurrence y = b0 x + b1 x1 + b2 x2 - a1 y1 - a2 y2 amplifies every rounding by 1 / A(z), i.e. by
// ~1 / (1 - r)^2 for poles of radius r, and a high-pass (b = {1, -2, 1} / a0: full-size terms that cancel) feeds it 30 times more of
// it than a low-pass (tiny b).  Measured, 44.1 / 48 / 96 kHz, q = 0.5:
//     low_pass : 1 - r >= 0.0125 (100 Hz at 48 kHz, 200 Hz at 96 kHz): <= 7.4e-6;   below: 1.2e-5 (50 Hz) .. 4e-5 (10 Hz)
//     high_pass: 1 - r >= 0.075  (600 Hz at 48 kHz)                   : <= 5e-6;     below: 8.6e-6 (500 Hz), 1.9e-5 (300), 3.3e-5 (200) .. 3e-3 (10)
// The error scales with the signal's peak.  Outside the region a drop-in takes the reference-order kernel (rh_biquad mode 0, bit
// for bit); include/rodio_hip.hpp does that on its own.
int32_t rh_filter_scan_ok(int32_t kind, uint32_t freq, float q, uint32_t sample_rate) {
    float c[5];
    if (rh_biquad_coeffs(kind, freq, q, sample_rate, c) != RH_OK) return 0;
    const double a1 = c[3], a2 = c[4], disc = a1 * a1 - 4.0 * a2;
    double r;
    if (disc >= 0.0) {
        const double sq = sqrt(disc);
        r = fmax(fabs((-a1 + sq) / 2.0), fabs((-a1 - sq) / 2.0));
    } else {
        r = sqrt(a2 > 0.0 ? a2 : 0.0);
    }
    return (1.0 - r) >= (kind == 0 ? 0.0125 : 0.075) ? 1 : 0;
}

}  // extern "C" (reopened below)
namespace rh {
rh_status agc_chain_launch(float *dst, const float *src, uint64_t n_samples, uint32_t n_streams, const float k5[5], float *state, hipStream_t s);
rh_status biquad_scan_launch(float *dst, const float *src, uint64_t frames, uint32_t channels, uint32_t n_streams, const float co[5], float *state, hipStream_t s);
}
extern "C" {

rh_status rh_biquad(float *dst, const float *src, uint64_t frames, uint32_t channels, uint32_t n_streams, const float coeffs5_host[5], float *state, int32_t mode, rh_stream stream) {
    RH_REQUIRE_INIT();
    if (channels == 0 || !coeffs5_host) return RH_ERR_INVALID;
    if (frames == 0 || n_streams == 0) return RH_OK;
    if (!dst || !src) return RH_ERR_INVALID;
    if (mode == 1) {  // time-parallel: the dedicated scan kernel (rh_biquad_scan.hip).  What it does not take -- rows that are not
                      // 16-byte aligned, a filter that does not forget within 64 tiles (a very low cutoff on short tiles), more than
                      // 8 channels -- continues in the reference-order kernel below: the same state layout, the exact bits, slower.
        const uint64_t total = frames * channels * (uint64_t)n_streams;
        const bool overlap = dst < src + total && src < dst + total;  // in place: the scan would read halo frames a neighbour has overwritten
        if (!overlap) {
            const rh_status st = rh::biquad_scan_launch(dst, src, frames, channels, n_streams, coeffs5_host, state, rh::as_stream(stream));
            if (st != RH_ERR_UNSUPPORTED || rh::knob(rh::K_BIQUAD_NO_FALLBACK)) return st;
        }
        mode = 0;
    }
    if (mode != 0) return RH_ERR_INVALID;
    const Biquad5 k{coeffs5_host[0], coeffs5_host[1], coeffs5_host[2], coeffs5_host[3], coeffs5_host[4]};
    // rows that start on 16-byte boundaries take the vector kernel (same arithmetic, same bits); anything else the 4-byte one
    const bool aligned = ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15u) == 0 && (n_streams == 1 || (frames * channels) % 4 == 0) && !rh::knob(rh::K_BIQUAD_SEQ);
    const dim3 grid_v((n_streams + kBlock - 1) / kBlock);
    if (aligned && channels == 1) hipLaunchKernelGGL(k_biquad_vec<1>, grid_v, dim3(kBlock), 0, rh::as_stream(stream), dst, src, frames, n_streams, k, state);
    else if (aligned && channels == 2) hipLaunchKernelGGL(k_biquad_vec<2>, grid_v, dim3(kBlock), 0, rh::as_stream(stream), dst, src, frames, n_streams, k, state);
    else if (aligned && channels == 4) hipLaunchKernelGGL(k_biquad_vec<4>, grid_v, dim3(kBlock), 0, rh::as_stream(stream), dst, src, frames, n_streams, k, state);
    else if (aligned && channels == 8) hipLaunchKernelGGL(k_biquad_vec<8>, grid_v, dim3(kBlock), 0, rh::as_stream(stream), dst, src, frames, n_streams, k, state);
    else {
        const uint32_t lanes = n_streams * channels;
        hipLaunchKernelGGL(k_biquad_seq, dim3((lanes + kBlock - 1) / kBlock), dim3(kBlock), 0, rh::as_stream(stream), dst, src, frames, channels, n_streams, k, state);
    }
    RH_CHECK_LAUNCH();
    return RH_OK;
}

size_t rh_agc_state_floats(void) { return kAgcStateFloats; }

// agc.rs:397-421: everything zero but current_gain = 1.0 (word 3 of a stream's state); one kernel on the caller's stream
__global__ void k_agc_state_init(float *state, size_t n) {
    const size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x, step = (size_t)gridDim.x * blockDim.x;
    for (size_t i = i0; i < n; i += step) state[i] = (i % kAgcStateFloats == 3) ? 1.0f : 0.0f;
}

rh_status rh_agc_state_init(float *state, uint32_t n_streams, rh_stream stream) {
    RH_REQUIRE_INIT();
    if (!state) return RH_ERR_INVALID;
    hipStream_t s = rh::as_stream(stream);
    hipLaunchKernelGGL(k_agc_state_init, dim3(rh::grid_for((size_t)kAgcStateFloats * n_streams)), dim3(256), 0, s, state, (size_t)kAgcStateFloats * n_streams);
    RH_CHECK_LAUNCH();
    return RH_OK;
}

rh_status rh_agc(float *dst, const float *src, uint64_t n_samples, uint32_t sample_rate, uint32_t n_streams, const rh_agc_params *p, float *state, rh_stream stream) {
    RH_REQUIRE_INIT();
    if (sample_rate == 0 || !p) return RH_ERR_INVALID;
    if (n_samples == 0 || n_streams == 0) return RH_OK;
    if (!dst || !src) return RH_ERR_INVALID;
    const uint64_t ten_s = 10000000000ull;  // source/mod.rs:432-433
    AgcK k;
    k.target_level = p->target_level;
    k.attack_coeff = duration_to_coefficient(p->attack_ns < ten_s ? p->attack_ns : ten_s, sample_rate);
    k.release_coeff = duration_to_coefficient(p->release_ns < ten_s ? p->release_ns : ten_s, sample_rate);
    k.absolute_max_gain = p->absolute_max_gain;
    k.floor = p->floor;
    hipStream_t s = rh::as_stream(stream);
    const bool aligned = ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15u) == 0 && (n_streams == 1 || n_samples % 4 == 0);
    // in place (dst == src) the window's tail could not be re-read from the input: the ring kernel then
    if (aligned && dst != src && !rh::knob(rh::K_AGC_SEQ) && !rh::knob(rh::K_AGC_VEC)) {  // the chains taken apart (rh_agc.hip): same operations, same bits
        const bool overlap = dst < src + (uint64_t)n_streams * n_samples && src < dst + (uint64_t)n_streams * n_samples;
        if (!overlap) {
            const float k5[5] = {k.target_level, k.attack_coeff, k.release_coeff, k.absolute_max_gain, k.floor};
            const rh_status cs = rh::agc_chain_launch(dst, src, n_samples, n_streams, k5, state, s);
            if (cs != RH_ERR_UNSUPPORTED) return cs;
        }
    }
    if (aligned && dst != src && !rh::knob(rh::K_AGC_SEQ)) {
        hipLaunchKernelGGL(k_agc_vec, dim3((n_streams + kBlock - 1) / kBlock), dim3(kBlock), 0, s, dst, src, n_samples, n_streams, k, state);
        RH_CHECK_LAUNCH();
        return RH_OK;
    }
    float *st = state;
    std::unique_lock<std::mutex> scratch_hold;  // a state-less call borrows the stream's scratch for the kernel's own state
    if (!st) RH_HIP_TRY(rh::stream_scratch(s, sizeof(float) * kAgcStateFloats * n_streams, reinterpret_cast<void **>(&st), scratch_hold));
    hipLaunchKernelGGL(k_agc_seq, dim3((n_streams + kBlock - 1) / kBlock), dim3(kBlock), 0, s, dst, src, n_samples, n_streams, k, st, state ? 0 : 1);
    hipError_t le = hipGetLastError();
    if (le != hipSuccess) {
        rh::set_hip_error(le, "k_agc_seq launch");
        return RH_ERR_HIP;
    }
    return RH_OK;
}

// ---- src/math.rs:51-56,86-90,110-113: the scalar helpers behind Amplify::set_log_factor / amplify_decibel
// (amplify.rs:33-35), the limiter and the AGC.  Host arithmetic, the reference's expressions.
float rh_db_to_linear(float decibels) { return powf(2.0f, decibels * 0.05f * 3.32192809488736234787f); }
float rh_linear_to_db(float linear) { return log2f(linear) * 0.30102999566398119521f * 20.0f; }
float rh_duration_to_coefficient(uint64_t duration_ns, uint32_t sample_rate) { return duration_to_coefficient(duration_ns, sample_rate); }
}  // extern "C"
