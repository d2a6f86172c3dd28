// rh_stream.hip -- the block-streaming forms of the two stateful adapters that are not recurrences:
//   SampleRateConverter  src/conversions/sample_rate.rs:52-90,110-122,131-201, src/math.rs:23-26
//   reverb = Mix(x, Delay(Amplify(x)))   src/source/mod.rs:628-634, delay.rs:8-16,68-75, mix.rs:43-53
// A Rust `Source` shim pulls its upstream in blocks; these handles carry what the reference's
// iterators keep between samples (the resampler's position and current frame; the delayed clone's
// D samples of history), so that ANY split of a stream into blocks gives the bits of one pass.
// (rh_biquad / rh_limit / rh_agc carry their state through their `state` argument already.)
// This TU is compiled with -ffp-contract=off: the lerp must not become an FMA.
#include <numeric>

#include "rh_common.h"

namespace {

constexpr int kBlock = 256;

// Output frames m in [m0, m1) of the continuous stream; input frame i(m) = floor(m*F/T) lives at
// src[i - n0] for i >= n0 and in `carry` for i == n0-1 (the last frame of the previous block).
// `verbatim_from`: frames with i >= it are emitted without interpolation (only the stream's last
// frame at flush, sample_rate.rs:193-200).
template <int C>
__global__ __launch_bounds__(kBlock) void k_resample_stream(float *__restrict__ dst, const float *__restrict__ src, const float *__restrict__ carry,
                                                            uint64_t m0, uint64_t m1, uint64_t n0, uint32_t F, uint32_t T, uint64_t verbatim_from, uint32_t channels) {
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    const float Tf = (float)T;
    const uint32_t ch = (C == 0) ? channels : (uint32_t)C;
    for (uint64_t m = m0 + (uint64_t)blockIdx.x * kBlock + threadIdx.x; m < m1; m += stride) {
        const unsigned __int128 pp = (unsigned __int128)m * F;  // the reference's u32 product wraps after ~2^32/F frames; a stream must not
        const uint64_t i = (uint64_t)(pp / T);
        const float numf = (float)(uint32_t)(pp - (unsigned __int128)i * T);
        const bool verbatim = i >= verbatim_from;
        const float *a = (i >= n0) ? src + (i - n0) * ch : carry;
        const float *b = src + (i + 1 - n0) * ch;  // i+1 >= n0 always (i >= n0-1)
        float *o = dst + (m - m0) * ch;
        for (uint32_t c = 0; c < ch; ++c) {
            const float av = a[c];
            const float bv = verbatim ? av : b[c];  // the frame after the stream's last one does not exist
            o[c] = verbatim ? av : av + (bv - av) * numf / Tf;
        }
    }
}

// n new samples at global sample index g0: y[i] = x[i] + 0 (g < D) | x[i] + a*x[g-D]; the delayed tap
// is in this block (i >= D) or in hist[D + i - D] = hist[i] (hist = the D samples before this block).
__global__ __launch_bounds__(kBlock) void k_echo_stream(float *__restrict__ dst, const float *__restrict__ src, const float *__restrict__ hist, uint64_t n, uint64_t g0,
                                                        uint64_t delay, float gain) {
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        float s2 = 0.0f;
        if (g0 + i >= delay) s2 = (i >= delay ? src[i - delay] : hist[i]) * gain;
        dst[i] = src[i] + s2;
    }
}
// the tail after the end of the stream: a*x[N-D+i] for i in [0, D); zeros where the stream was shorter than D
__global__ __launch_bounds__(kBlock) void k_echo_tail(float *__restrict__ dst, const float *__restrict__ hist, uint64_t delay, uint64_t total, float gain) {
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < delay; i += stride) {
        // global index of the delayed source sample: total - delay + i (negative: still inside Delay's zeros)
        dst[i] = (total + i >= delay) ? hist[i] * gain : 0.0f;
    }
}
// hist <- the last D samples of (hist ++ block)
__global__ __launch_bounds__(kBlock) void k_hist_update(float *__restrict__ hist_new, const float *__restrict__ hist_old, const float *__restrict__ src, uint64_t n, uint64_t delay) {
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < delay; i += stride) {
        // element i of the new history is element i + n of (old ++ block)
        const uint64_t j = i + n;
        hist_new[i] = j < delay ? hist_old[j] : src[j - delay];
    }
}

}  // namespace

struct rh_resampler {
    uint32_t F, T, channels;
    uint64_t total_in = 0, total_out = 0;
    bool finished = false;
    float *d_carry = nullptr;  // the last input frame seen
};
struct rh_echo {
    uint64_t delay;
    float gain;
    uint64_t total = 0;
    bool finished = false;
    float *d_hist[2] = {nullptr, nullptr};  // ping-pong: the last `delay` input samples
    int cur = 0;
};

extern "C" {

rh_status rh_resampler_create(rh_resampler **out, uint32_t from_rate, uint32_t to_rate, uint32_t channels) {
    RH_REQUIRE_INIT();
    if (!out || from_rate == 0 || to_rate == 0 || channels == 0) return RH_ERR_INVALID;
    const uint32_t g = std::gcd(from_rate, to_rate);
    rh_resampler *p = new rh_resampler();
    p->F = from_rate / g;
    p->T = to_rate / g;
    p->channels = channels;
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&p->d_carry), sizeof(float) * channels);
    if (e != hipSuccess) {
        rh::set_hip_error(e, "rh_resampler_create");
        delete p;
        return e == hipErrorOutOfMemory ? RH_ERR_NOMEM : RH_ERR_HIP;
    }
    *out = p;
    return RH_OK;
}
rh_status rh_resampler_reset(rh_resampler *p) {
    if (!p) return RH_ERR_INVALID;
    p->total_in = p->total_out = 0;
    p->finished = false;
    return RH_OK;
}
rh_status rh_resampler_destroy(rh_resampler *p) {
    if (!p) return RH_OK;
    if (p->d_carry) (void)hipFree(p->d_carry);
    delete p;
    return RH_OK;
}
// frames the next process() call would emit for in_frames more input frames
static uint64_t resampler_emit_upto(const rh_resampler *p, uint64_t total_in, bool flush) {
    if (total_in == 0) return 0;
    if (p->F == p->T) return total_in;
    // every m with floor(m*F/T) <= N-2, i.e. m < ceil((N-1)*T/F); at flush also the m that lands on N-1
    const unsigned __int128 num = (unsigned __int128)(total_in - 1) * p->T;
    const uint64_t c1 = (uint64_t)((num + p->F - 1) / p->F);
    if (!flush) return c1;
    const bool lands = (unsigned __int128)c1 * p->F < (unsigned __int128)total_in * p->T;
    return c1 + (lands ? 1 : 0);
}
rh_status rh_resampler_pending_frames(rh_resampler *p, uint64_t in_frames, int32_t flush, uint64_t *out_frames) {
    if (!p || !out_frames) return RH_ERR_INVALID;
    *out_frames = resampler_emit_upto(p, p->total_in + in_frames, flush != 0) - p->total_out;
    return RH_OK;
}
rh_status rh_resampler_process(rh_resampler *p, float *dst, uint64_t dst_capacity_frames, const float *src, uint64_t in_frames, int32_t flush, uint64_t *out_frames,
                               rh_stream stream) {
    RH_REQUIRE_INIT();
    if (!p || (in_frames && !src)) return RH_ERR_INVALID;
    if (p->finished) return RH_ERR_INVALID;  // a flushed stream is over (None): reset first
    hipStream_t s = rh::as_stream(stream);
    const uint64_t n0 = p->total_in, total = n0 + in_frames;
    const uint64_t m0 = p->total_out, m1 = resampler_emit_upto(p, total, flush != 0);
    if (out_frames) *out_frames = m1 - m0;
    if (m1 - m0 > dst_capacity_frames) return RH_ERR_CAPACITY;
    if (m1 > m0) {
        if (!dst) return RH_ERR_INVALID;
        if (p->F == p->T) {  // sample_rate.rs:133-136 passthrough
            RH_HIP_TRY(rh::copy_d2d(dst, src, in_frames * p->channels * sizeof(float), s));
        } else {
            const uint64_t verbatim_from = flush ? total - 1 : ~0ull;
            const unsigned grid = rh::grid_for(m1 - m0);
            if (p->channels == 2) hipLaunchKernelGGL((k_resample_stream<2>), dim3(grid), dim3(kBlock), 0, s, dst, src, p->d_carry, m0, m1, n0, p->F, p->T, verbatim_from, p->channels);
            else if (p->channels == 1) hipLaunchKernelGGL((k_resample_stream<1>), dim3(grid), dim3(kBlock), 0, s, dst, src, p->d_carry, m0, m1, n0, p->F, p->T, verbatim_from, p->channels);
            else hipLaunchKernelGGL((k_resample_stream<0>), dim3(grid), dim3(kBlock), 0, s, dst, src, p->d_carry, m0, m1, n0, p->F, p->T, verbatim_from, p->channels);
            RH_CHECK_LAUNCH();
        }
    }
    if (in_frames)  // the next block's frame n0-1 (enqueued behind the kernel that still reads the old one)
        RH_HIP_TRY(rh::copy_d2d(p->d_carry, src + (in_frames - 1) * p->channels, sizeof(float) * p->channels, s));
    p->total_in = total;
    p->total_out = m1;
    p->finished = flush != 0;
    return RH_OK;
}

rh_status rh_echo_create(rh_echo **out, uint64_t delay_samples, float gain) {
    RH_REQUIRE_INIT();
    if (!out) return RH_ERR_INVALID;
    rh_echo *p = new rh_echo();
    p->delay = delay_samples;
    p->gain = gain;
    for (int k = 0; k < 2 && delay_samples; ++k) {
        hipError_t e = hipMalloc(reinterpret_cast<void **>(&p->d_hist[k]), sizeof(float) * delay_samples);
        if (e == hipSuccess) e = rh::fill_now(p->d_hist[k], 0, sizeof(float) * delay_samples);  // before the first block's kernel writes it
        if (e != hipSuccess) {
            rh::set_hip_error(e, "rh_echo_create");
            rh_echo_destroy(p);
            return e == hipErrorOutOfMemory ? RH_ERR_NOMEM : RH_ERR_HIP;
        }
    }
    *out = p;
    return RH_OK;
}
rh_status rh_echo_destroy(rh_echo *p) {
    if (!p) return RH_OK;
    for (int k = 0; k < 2; ++k)
        if (p->d_hist[k]) (void)hipFree(p->d_hist[k]);
    delete p;
    return RH_OK;
}
rh_status rh_echo_reset(rh_echo *p) {
    if (!p) return RH_ERR_INVALID;
    p->total = 0;
    p->finished = false;
    return RH_OK;
}
rh_status rh_echo_process(rh_echo *p, float *dst, const float *src, uint64_t n, rh_stream stream) {
    RH_REQUIRE_INIT();
    if (!p || p->finished) return RH_ERR_INVALID;
    if (n == 0) return RH_OK;
    if (!dst || !src) return RH_ERR_INVALID;
    hipStream_t s = rh::as_stream(stream);
    hipLaunchKernelGGL(k_echo_stream, dim3(rh::grid_for(n)), dim3(kBlock), 0, s, dst, src, p->d_hist[p->cur], n, p->total, p->delay, p->gain);
    RH_CHECK_LAUNCH();
    if (p->delay) {
        hipLaunchKernelGGL(k_hist_update, dim3(rh::grid_for(p->delay)), dim3(kBlock), 0, s, p->d_hist[p->cur ^ 1], p->d_hist[p->cur], src, n, p->delay);
        RH_CHECK_LAUNCH();
        p->cur ^= 1;
    }
    p->total += n;
    return RH_OK;
}
rh_status rh_echo_flush(rh_echo *p, float *dst, rh_stream stream) {
    RH_REQUIRE_INIT();
    if (!p || p->finished) return RH_ERR_INVALID;
    p->finished = true;
    if (p->delay == 0) return RH_OK;
    if (!dst) return RH_ERR_INVALID;
    hipLaunchKernelGGL(k_echo_tail, dim3(rh::grid_for(p->delay)), dim3(kBlock), 0, rh::as_stream(stream), dst, p->d_hist[p->cur], p->delay, p->total, p->gain);
    RH_CHECK_LAUNCH();
    return RH_OK;
}

}  // extern "C"
