// rh_resample.hip -- SampleRateConverter + UniformSourceIterator span chunking, and the Mixer sum.
//
//   src/conversions/sample_rate.rs:52-90,110-122,131-201   (the streaming lerp state machine)
//   src/math.rs:23-26                                        (lerp: mul, divide, add -- in that order)
//   src/source/uniform.rs:50-97                              (restart every min(span,32768) samples)
//   src/mixer.rs:185-198                                     (ordered f32 sum)
//
// The reference's state machine is a pure function of the output frame index m:
//     i = floor(m*F/T), num = (m*F) mod T        (F/T = from/to reduced by gcd, sample_rate.rs:74)
//     i <= N-2 : x[i] + (x[i+1]-x[i]) * num / T  (lerp)
//     i == N-1 : x[N-1] verbatim, then the stream ends (the drain at :193-200)
// applied independently per chunk of min(span_len,32768) samples.  That closed form is what
// the kernel evaluates -- one lane per output frame, both taps read straight from HBM/L2
// (adjacent lanes read adjacent frames, so every input line is fetched once).
// This TU is compiled with -ffp-contract=off: the lerp must not become an FMA.
#include <cstdlib>
#include <numeric>

#include <cstring>

#include "rh_common.h"

namespace rh {

struct ResampleGeom {
    uint32_t F, T;
    uint64_t in_frames;
    uint64_t chunk_in;   // input frames per full chunk (== in_frames when unchunked)
    uint64_t chunk_out;  // output frames per full chunk
    uint64_t n_chunks;
    uint64_t last_in;    // input frames of the last chunk
    uint64_t out_frames;
    int fits32;          // (chunk_out * F) < 2^32: index math in u32 like the reference
};

// Output frames of one independently converted run of n input frames:
// every m with floor(mF/T) <= n-2, plus one verbatim frame if some m lands on i == n-1.
static uint64_t run_out_frames(uint64_t n, uint64_t F, uint64_t T) {
    if (n == 0) return 0;
    if (F == T) return n;
    const unsigned __int128 num = (unsigned __int128)(n - 1) * T;
    const uint64_t c1 = (uint64_t)((num + F - 1) / F);  // #m with i(m) <= n-2
    const bool lands = (unsigned __int128)c1 * F < (unsigned __int128)n * T;
    return c1 + (lands ? 1 : 0);
}

rh_status make_resample_geom(uint64_t in_frames, uint32_t from_rate, uint32_t to_rate, uint32_t channels,
                             uint64_t span_len, ResampleGeom *g) {
    if (from_rate == 0 || to_rate == 0 || channels == 0) return RH_ERR_INVALID;
    const uint32_t gc = std::gcd(from_rate, to_rate);
    g->F = from_rate / gc;
    g->T = to_rate / gc;
    // The reference multiplies from*pos in u32 (sample_rate.rs:157,173; doc :45-47).
    if ((uint64_t)g->F * g->T > 0xffffffffull) return RH_ERR_UNSUPPORTED;
    g->in_frames = in_frames;
    uint64_t chunk_in = in_frames;
    if (span_len != 0) {
        const uint64_t span = span_len < 32768 ? span_len : 32768;  // uniform.rs:56
        if (span % channels != 0) return RH_ERR_UNSUPPORTED;        // a span that splits a frame
        chunk_in = span / channels;
        if (chunk_in == 0) return RH_ERR_INVALID;
    }
    if (chunk_in >= in_frames || in_frames == 0) {
        g->chunk_in = in_frames;
        g->n_chunks = in_frames ? 1 : 0;
        g->last_in = in_frames;
        g->chunk_out = run_out_frames(in_frames, g->F, g->T);
        g->out_frames = g->chunk_out;
    } else {
        g->chunk_in = chunk_in;
        g->n_chunks = (in_frames + chunk_in - 1) / chunk_in;
        g->last_in = in_frames - (g->n_chunks - 1) * chunk_in;
        g->chunk_out = run_out_frames(chunk_in, g->F, g->T);
        g->out_frames = (g->n_chunks - 1) * g->chunk_out + run_out_frames(g->last_in, g->F, g->T);
    }
    const unsigned __int128 pmax = (unsigned __int128)(g->chunk_out + 1) * g->F;
    g->fits32 = pmax < ((unsigned __int128)1 << 32);
    return RH_OK;
}

}  // namespace rh

namespace {

constexpr int kBlock = 256;

template <int C, bool P32>
__global__ __launch_bounds__(kBlock) void k_resample_linear(float *__restrict__ dst, const float *__restrict__ src, rh::ResampleGeom g, uint32_t channels) {
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    const float Tf = (float)g.T;
    for (uint64_t m = (uint64_t)blockIdx.x * kBlock + threadIdx.x; m < g.out_frames; m += stride) {
        uint64_t k = 0, ml = m;
        if (g.n_chunks > 1) {
            k = m / g.chunk_out;
            if (k > g.n_chunks - 1) k = g.n_chunks - 1;
            ml = m - k * g.chunk_out;
        }
        uint64_t il;
        uint32_t num;
        if (P32) {
            const uint32_t p = (uint32_t)ml * g.F;
            il = p / g.T;
            num = p - (uint32_t)il * g.T;
        } else {
            const uint64_t p = ml * g.F;
            il = p / g.T;
            num = (uint32_t)(p - il * g.T);
        }
        const uint64_t nc = g.chunk_in + (k == g.n_chunks - 1 ? g.last_in - g.chunk_in : 0);  // (as a sum: the select between two kernel arguments became an indexed read of a scratch copy)
        const bool verbatim = il + 1 >= nc;  // i == nc-1: the drained last frame
        const uint64_t i = k * g.chunk_in + (verbatim ? nc - 1 : il);
        const float numf = (float)num;
        if (C == 2) {
            const float2 a = reinterpret_cast<const float2 *>(src)[i];
            float2 o = a;
            if (!verbatim) {
                const float2 b = reinterpret_cast<const float2 *>(src)[i + 1];
                o.x = a.x + (b.x - a.x) * numf / Tf;
                o.y = a.y + (b.y - a.y) * numf / Tf;
            }
            reinterpret_cast<float2 *>(dst)[m] = o;
        } else {
            const uint32_t ch = (C == 0) ? channels : (uint32_t)C;
            for (uint32_t c = 0; c < ch; ++c) {
                const float a = src[i * ch + c];
                float o = a;
                if (!verbatim) {
                    const float b = src[(i + 1) * ch + c];
                    o = a + (b - a) * numf / Tf;
                }
                dst[m * ch + c] = o;
            }
        }
    }
}

// The same closed form, a TILE of output frames per workgroup (round 6, last session).  The lane-per-frame kernel above asks for both taps
// of every frame from memory (8 bytes a lane for stereo, 4 for mono, a loop of 4-byte loads for 5.1) and stores a frame a lane:
// tools/bench_rows.py measured 0.51 / 0.31 / 0.23 of 8 TB/s for stereo / mono / 5.1 at 44.1 -> 48 kHz.  Here the input frames a tile needs --
// one contiguous run, chunk boundaries included: a chunk's verbatim last frame is followed by the next chunk's first -- come in once as
// aligned 16-byte vectors (the first starts up to 12 bytes in front of the run, inside the same 16 bytes), are parked in LDS, and a lane
// produces four consecutive output SAMPLES (one 16-byte store) from taps it reads there.  Index arithmetic, operations and their order are
// the kernel's above; a tile whose run does not fit the LDS it was given (a burst of tiny chunks) reads its taps from memory instead.
template <bool P32>
__device__ __forceinline__ void resample_pos(const rh::ResampleGeom &g, uint64_t m, uint64_t &i, uint32_t &num, bool &verbatim) {
    uint64_t k = 0, ml = m;
    if (g.n_chunks > 1) {
        k = m / g.chunk_out;
        if (k > g.n_chunks - 1) k = g.n_chunks - 1;
        ml = m - k * g.chunk_out;
    }
    uint64_t il;
    if (P32) {
        const uint32_t p = (uint32_t)ml * g.F;
        il = p / g.T;
        num = p - (uint32_t)il * g.T;
    } else {
        const uint64_t p = ml * g.F;
        il = p / g.T;
        num = (uint32_t)(p - il * g.T);
    }
    const uint64_t nc = g.chunk_in + (k == g.n_chunks - 1 ? g.last_in - g.chunk_in : 0);  // (as a sum: the select between two kernel arguments became an indexed read of a scratch copy)
    verbatim = il + 1 >= nc;  // i == nc-1: the drained last frame
    i = k * g.chunk_in + (verbatim ? nc - 1 : il);
}
// Where a tile sits: chunk k, the position (il0, num0) of its first output frame inside the chunk, and whether all its frames belong to
// that chunk -- then frame f of the tile is at il0 + (num0 + f * F) / T, a 32-bit division (the launcher checked T + tile_frames * F < 2^32).
// A lane-per-frame kernel pays a 64-bit multiply and division per FRAME once the row is longer than 2^32 / F frames (29 M at 44.1 -> 48 kHz):
// that, not memory, is what held the kernel above at 0.51 of the roofline on the 64 Mi-frame row of tools/bench_rows.py.
struct TilePos {
    uint64_t k, il0, nc;  // chunk, first frame's position in it, the chunk's input frames
    uint32_t num0;
    bool one_chunk;
};
template <bool P32>
__device__ __forceinline__ TilePos resample_tile_pos(const rh::ResampleGeom &g, uint32_t tile, uint32_t tile_frames, uint32_t nf, uint32_t qA, uint32_t rA, int small_out) {
    TilePos t;
    if (g.n_chunks <= 1) {  // m0 * F = tile * (tile_frames * F) = tile * (qA * T + rA)
        t.k = 0;
        const uint64_t br = (uint64_t)tile * rA;
        uint64_t d;
        if (rA == 0) d = 0, t.num0 = 0;  // tiles of whole periods (the launcher's choice wherever a period fits a tile)
        else if ((br >> 32) == 0) d = (uint32_t)br / g.T, t.num0 = (uint32_t)br - (uint32_t)d * g.T;
        else d = br / g.T, t.num0 = (uint32_t)(br - d * g.T);
        t.il0 = (uint64_t)tile * qA + d;
        t.nc = g.last_in;
        t.one_chunk = true;
        return t;
    }
    const uint64_t m0 = (uint64_t)tile * tile_frames;
    uint64_t k = small_out ? (uint64_t)((uint32_t)m0 / (uint32_t)g.chunk_out) : m0 / g.chunk_out;
    if (k > g.n_chunks - 1) k = g.n_chunks - 1;
    const uint64_t ml = m0 - k * g.chunk_out;
    if (P32) {
        const uint32_t p = (uint32_t)ml * g.F;
        t.il0 = p / g.T, t.num0 = p - (uint32_t)t.il0 * g.T;
    } else {
        const uint64_t p = ml * g.F;
        t.il0 = p / g.T, t.num0 = (uint32_t)(p - t.il0 * g.T);
    }
    t.k = k;
    t.nc = g.chunk_in + (k == g.n_chunks - 1 ? g.last_in - g.chunk_in : 0);
    t.one_chunk = k == g.n_chunks - 1 || ml + nf <= g.chunk_out;
    return t;
}
template <bool P32, typename TapPtr>
__device__ __forceinline__ void resample_tile_out(const rh::ResampleGeom &g, TapPtr taps, float *__restrict__ out, uint64_t m0, uint64_t i_lo, uint32_t nf, uint32_t ch, int vec_ok) {
    const float Tf = (float)g.T;
    const uint32_t total = nf * ch, nv = (total + 3u) / 4u;
    for (uint32_t v = threadIdx.x; v < nv; v += kBlock) {
        const uint32_t o0 = 4u * v;
        uint32_t f = o0 / ch, c = o0 - f * ch;
        float e[4];
        uint64_t i = 0;
        uint32_t num = 0;
        bool vb = true, have = false;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (o0 + j < total) {
                if (!have) resample_pos<P32>(g, m0 + f, i, num, vb), have = true;
                const uint32_t t = (uint32_t)(i - i_lo) * ch + c;
                const float a = taps[t];
                float o = a;
                if (!vb) {
                    const float b = taps[t + ch];
                    o = a + (b - a) * (float)num / Tf;
                }
                e[j] = o;
            } else {
                e[j] = 0.0f;
            }
            if (++c == ch) c = 0, ++f, have = false;
        }
        if (vec_ok && o0 + 4u <= total) {
            rh::st_nt(reinterpret_cast<float4 *>(out + o0), make_float4(e[0], e[1], e[2], e[3]));
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (o0 + j < total) out[o0 + j] = e[j];
        }
    }
}
// The tile inside ONE chunk (every tile of an unchunked row, all but a few of a spanned one): everything per frame in 32 bits, relative to the
// tile.  Frame f sits d = (num0 + f * F) / T input frames behind the tile's first tap; it is the chunk's verbatim last frame from d = dv on.
// A lane finds its first frame's d by a float estimate corrected by one (d is a few thousand at most, the estimate is off by less than 0.01)
// and walks to its next frames by adding F mod T / F div T -- the per-frame 64-bit division of the kernel above was what bound it, and a
// 32-bit division a frame still left this one at 0.40 of the roofline.  CH: 1 / 2 known at compile time, 0 = any.
template <int CH, typename TapPtr>
__device__ __forceinline__ void resample_tile_out_fast(TapPtr taps, float *__restrict__ out, uint32_t num0, uint32_t dv, uint32_t nf, uint32_t chr, uint32_t F, uint32_t T, float invT,
                                                       uint32_t qF, uint32_t rF, int vec_ok) {
    const uint32_t ch = CH ? (uint32_t)CH : chr;
    const float Tf = (float)T, inv_ch = 1.0f / (float)ch;
    const uint32_t total = nf * ch, nv = (total + 3u) / 4u;
    for (uint32_t v = threadIdx.x; v < nv; v += kBlock) {
        const uint32_t o0 = 4u * v;
        uint32_t f, c;
        if (CH == 1) {
            f = o0, c = 0;
        } else if (CH == 2) {
            f = o0 >> 1, c = 0;
        } else {
            f = (uint32_t)((float)o0 * inv_ch);  // o0 < 2^24: exact as a float; the estimate is one off at most
            int32_t cc = (int32_t)(o0 - f * ch);
            if (cc < 0) --f, cc += (int32_t)ch;
            else if ((uint32_t)cc >= ch) ++f, cc -= (int32_t)ch;
            c = (uint32_t)cc;
        }
        const uint32_t p = num0 + f * F;  // < T + tile_frames * F < 2^32 (the launcher's condition)
        uint32_t d = (uint32_t)((float)p * invT);
        int32_t nn = (int32_t)(p - d * T);  // in (-T, 2T), T < 2^30
        if (nn < 0) --d, nn += (int32_t)T;
        else if ((uint32_t)nn >= T) ++d, nn -= (int32_t)T;
        uint32_t num = (uint32_t)nn;
        float e[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float o = 0.0f;
            if (o0 + j < total) {
                const bool vb = d >= dv;
                const uint32_t t = (vb ? dv : d) * ch + c;
                const float a = taps[t];
                o = a;
                if (!vb) {
                    const float b = taps[t + ch];
                    o = a + (b - a) * (float)num / Tf;  // math.rs:25 (the IEEE division: a shortened one measured no faster here, and is not exact everywhere -- rh_common.h)
                }
            }
            e[j] = o;
            if (CH == 1 || ++c == ch) {
                c = 0, num += rF, d += qF;
                if (num >= T) num -= T, ++d;
            }
        }
        if (vec_ok && o0 + 4u <= total) {
            rh::st_nt(reinterpret_cast<float4 *>(out + o0), make_float4(e[0], e[1], e[2], e[3]));
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (o0 + j < total) out[o0 + j] = e[j];
        }
    }
}
template <bool P32, int CH>
__global__ __launch_bounds__(kBlock) void k_resample_tile(float *__restrict__ dst, const float *__restrict__ src, rh::ResampleGeom g, uint32_t ch, uint32_t tile_frames, uint32_t lds_floats, uint32_t qA, uint32_t rA,
                                                          float invT, uint32_t qF, uint32_t rF, int small_out, int vec_ok) {
    extern __shared__ uint4 rs_tile[];
    const uint64_t m0 = (uint64_t)blockIdx.x * tile_frames;  // (tile_frames is a multiple of 4: the tile's first output sample starts a 16-byte vector)
    const uint32_t nf = (uint32_t)(g.out_frames - m0 < tile_frames ? g.out_frames - m0 : tile_frames);
    const TilePos tp = resample_tile_pos<P32>(g, blockIdx.x, tile_frames, nf, qA, rA, small_out);
    float *out = dst + m0 * ch;
    if (tp.one_chunk) {
        const uint64_t dv64 = tp.nc >= 1 + tp.il0 ? tp.nc - 1 - tp.il0 : 0;  // the first d at which the frame is the chunk's last (verbatim)
        const uint32_t dv = dv64 < 0xffffffffull ? (uint32_t)dv64 : 0xffffffffu;
        const uint64_t i_lo = tp.k * g.chunk_in + tp.il0 + (dv64 ? 0 : tp.nc - 1 - tp.il0);  // == k * chunk_in + min(il0, nc - 1)
        const uint32_t d_last = (tp.num0 + (nf - 1) * g.F) / g.T;
        const uint32_t rel_hi = d_last < dv ? d_last + 1 : dv;  // the last tap the tile reads, in frames behind i_lo
        const uintptr_t p0 = reinterpret_cast<uintptr_t>(src + i_lo * ch), a0 = p0 & ~(uintptr_t)15;
        const uint32_t shift = (uint32_t)(p0 - a0) / 4u;
        const uint64_t run = ((uint64_t)rel_hi + 1) * ch + shift;  // floats from a0
        const bool in_lds = run <= lds_floats;                      // (uniform over the workgroup)
        if (in_lds) {
            const uint32_t nvec = (uint32_t)((run + 3) / 4);
            for (uint32_t v = threadIdx.x; v < nvec; v += kBlock) rs_tile[v] = rh::ld_nt(reinterpret_cast<const uint4 *>(a0) + v);
        }
        __syncthreads();
        // (two calls each, so that the taps' address space is known where they are read: ds_read for the LDS image, global loads otherwise)
        if (in_lds) resample_tile_out_fast<CH>(reinterpret_cast<const float *>(rs_tile) + shift, out, tp.num0, dv, nf, ch, g.F, g.T, invT, qF, rF, vec_ok);
        else resample_tile_out_fast<CH>(src + i_lo * ch, out, tp.num0, dv, nf, ch, g.F, g.T, invT, qF, rF, vec_ok);
        return;
    }
    uint64_t i_lo, i_hi;  // a tile across chunk boundaries: frame by frame
    {
        uint32_t num;
        bool vb;
        resample_pos<P32>(g, m0, i_lo, num, vb);
        resample_pos<P32>(g, m0 + nf - 1, i_hi, num, vb);
        i_hi += vb ? 0 : 1;  // the second tap
    }
    const uintptr_t p0 = reinterpret_cast<uintptr_t>(src + i_lo * ch), a0 = p0 & ~(uintptr_t)15;
    const uint32_t shift = (uint32_t)(p0 - a0) / 4u;
    const uint64_t run = (i_hi - i_lo + 1) * ch + shift;
    const bool in_lds = run <= lds_floats;
    if (in_lds) {
        const uint32_t nvec = (uint32_t)((run + 3) / 4);
        for (uint32_t v = threadIdx.x; v < nvec; v += kBlock) rs_tile[v] = rh::ld_nt(reinterpret_cast<const uint4 *>(a0) + v);
    }
    __syncthreads();
    if (in_lds) resample_tile_out<P32>(g, reinterpret_cast<const float *>(rs_tile) + shift, out, m0, i_lo, nf, ch, vec_ok);
    else resample_tile_out<P32>(g, src + i_lo * ch, out, m0, i_lo, nf, ch, vec_ok);
}

struct MixDesc {
    const float *data;
    uint64_t start;
    uint64_t len;
};

// One lane per output sample, sources visited in insertion order: the rounding sequence of
// mixer.rs:185-198 (`sum = 0.0; sum += v_s`).  No atomics, no tree: bit-identical to the CPU.
// The source table travels BY VALUE as a kernel argument, kMixChunk sources per launch: nothing to upload, nothing
// to synchronise, nothing that a later call could overwrite while this launch is still queued.  More sources take
// further launches that continue from the stored partial sum -- the same left-to-right sequence of f32 additions.
constexpr uint32_t kMixChunk = 128;  // 24 bytes a source: 3 KiB of the 4 KiB a kernel's arguments may take (32 until round 6: 256 short rows were 8 launches, 0.39 of 8 TB/s)
struct MixTable {
    MixDesc d[kMixChunk];
};
template <bool CONT>
__global__ __launch_bounds__(kBlock) void k_mix_sum(float *__restrict__ dst, uint64_t out_len, const MixTable tbl, uint32_t n_sources) {
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t o = (uint64_t)blockIdx.x * kBlock + threadIdx.x; o < out_len; o += stride) {
        float acc = CONT ? dst[o] : 0.0f;
        for (uint32_t s = 0; s < n_sources; ++s) {
            const MixDesc d = tbl.d[s];
            const uint64_t rel = o - d.start;  // wraps to huge when o < start
            if (rel < d.len) acc += d.data[rel];
        }
        dst[o] = acc;
    }
}

// Same, four consecutive samples per lane (float4 loads) when every start is a multiple of 4
// samples and every pointer is 16-byte aligned -- the common "all sources start together" case.
template <bool CONT>
__global__ __launch_bounds__(kBlock) void k_mix_sum_v4(float *__restrict__ dst, uint64_t out_len, const MixTable tbl, uint32_t n_sources) {
    const uint64_t nvec = (out_len + 3) / 4;
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t v = (uint64_t)blockIdx.x * kBlock + threadIdx.x; v < nvec; v += stride) {
        const uint64_t o = v * 4;
        float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (CONT) {
            for (int k = 0; k < 4; ++k)
                if (o + k < out_len) acc[k] = dst[o + k];
        }
        for (uint32_t s = 0; s < n_sources; ++s) {
            const MixDesc d = tbl.d[s];
            const uint64_t rel = o - d.start;
            if (rel < d.len) {
                if (rel + 4 <= d.len) {
                    const float4 x = *reinterpret_cast<const float4 *>(d.data + rel);
                    acc[0] += x.x;
                    acc[1] += x.y;
                    acc[2] += x.z;
                    acc[3] += x.w;
                } else {
                    for (int k = 0; k < 4; ++k)
                        if (rel + k < d.len) acc[k] += d.data[rel + k];
                }
            }
        }
        if (o + 4 <= out_len) {
            *reinterpret_cast<float4 *>(dst + o) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        } else {
            for (int k = 0; k < 4; ++k)
                if (o + k < out_len) dst[o + k] = acc[k];
        }
    }
}
// The same for rows that start ANYWHERE (a late join two samples into a vector, a row inside a larger buffer): every source through
// rh::ld4_at -- aligned vectors around the samples, 0.0 where the source does not reach (an addition of +0.0 leaves a sum that started at +0.0
// as it is: the sum is never -0.0).  The sample-a-lane kernel it replaces there ran at a fraction of this one's rate.
template <bool CONT>
__global__ __launch_bounds__(kBlock) void k_mix_sum_any(float *__restrict__ dst, uint64_t out_len, const MixTable tbl, uint32_t n_sources, int dst_vec) {
    const uint64_t nvec = (out_len + 3) / 4;
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t v = (uint64_t)blockIdx.x * kBlock + threadIdx.x; v < nvec; v += stride) {
        const uint64_t o = v * 4;
        float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (CONT) {
            for (int k = 0; k < 4; ++k)
                if (o + k < out_len) acc[k] = dst[o + k];
        }
        for (uint32_t s = 0; s < n_sources; ++s) {
            const MixDesc d = tbl.d[s];
            const int64_t rel = (int64_t)o - (int64_t)d.start;
            if (rel > -4 && rel < (int64_t)d.len) {  // (uniform nowhere, cheap everywhere: the sources a vector does not touch are skipped)
                const float4 x = rh::ld4_at(d.data, rel, d.len);
                acc[0] += x.x;
                acc[1] += x.y;
                acc[2] += x.z;
                acc[3] += x.w;
            }
        }
        if (dst_vec && o + 4 <= out_len) {
            *reinterpret_cast<float4 *>(dst + o) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        } else {
            for (int k = 0; k < 4; ++k)
                if (o + k < out_len) dst[o + k] = acc[k];
        }
    }
}
// The same for SHORT rows of many sources (a block of a wide mixer: 256 rows of 2 MiB are 512 workgroups): the loop above keeps a load or two a
// lane in flight, which is enough -- and kinder to DRAM pages -- when the launch has thousands of workgroups (32 rows of 16 MiB: 0.78 of 8 TB/s,
// against 0.51-0.66 for any grouped form), and not when it has two a CU (0.39-0.44): there a lane asks for eight sources at once, the next
// eight before it adds these (0.63).  The additions stay in insertion order; a group any lane of the wave cannot take whole goes source by source.
template <bool CONT, uint32_t kGroup>
__global__ __launch_bounds__(kBlock) void k_mix_sum_v4_grp(float *__restrict__ dst, uint64_t out_len, const MixTable tbl, uint32_t n_sources) {
    const uint64_t nvec = (out_len + 3) / 4;
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    //                                 sources whose loads leave together  // sources whose loads leave together (round 6, last session: one source after the other, each behind its own
                                    // bounds branch, kept ONE load a lane in flight -- 256 rows of 2 MiB: 0.39 of 8 TB/s)
    for (uint64_t v = (uint64_t)blockIdx.x * kBlock + threadIdx.x; v < nvec; v += stride) {
        const uint64_t o = v * 4;
        float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (CONT) {
            for (int k = 0; k < 4; ++k)
                if (o + k < out_len) acc[k] = dst[o + k];
        }
        auto one = [&](const MixDesc d) {  // a source that may start late, end inside the vector or not reach it
            const uint64_t rel = o - d.start;  // wraps to huge when o < start
            if (rel < d.len) {
                if (rel + 4 <= d.len) {
                    const float4 x = *reinterpret_cast<const float4 *>(d.data + rel);
                    acc[0] += x.x;
                    acc[1] += x.y;
                    acc[2] += x.z;
                    acc[3] += x.w;
                } else {
                    for (int k = 0; k < 4; ++k)
                        if (rel + k < d.len) acc[k] += d.data[rel + k];
                }
            }
        };
        // group g + 1's loads leave BEFORE group g's are added (two groups a lane in flight; waiting for a group with nothing behind it measured
        // 0.53 of 8 TB/s at 32 rows where the old loop had 0.78)
        auto is_full = [&](uint32_t s0) {  // every source of the group holds the whole vector -- in every lane of the wave, or the group goes source by source
            bool full = true;
#pragma unroll
            for (uint32_t u = 0; u < kGroup; ++u) {
                const uint64_t rel = o - tbl.d[s0 + u].start;
                full = full && rel < tbl.d[s0 + u].len && rel + 4 <= tbl.d[s0 + u].len;
            }
            return __builtin_amdgcn_ballot_w64(!full) == 0;
        };
        uint32_t s = 0;
        float4 cur[kGroup], nxt[kGroup];
        bool cur_ok = false, nxt_ok = false;
        if (kGroup <= n_sources) {
            cur_ok = is_full(0);
            if (cur_ok) {
#pragma unroll
                for (uint32_t u = 0; u < kGroup; ++u) cur[u] = *reinterpret_cast<const float4 *>(tbl.d[u].data + (o - tbl.d[u].start));
            }
        }
        for (; s + kGroup <= n_sources; s += kGroup) {
            nxt_ok = s + 2 * kGroup <= n_sources && is_full(s + kGroup);
            if (nxt_ok) {
#pragma unroll
                for (uint32_t u = 0; u < kGroup; ++u) nxt[u] = *reinterpret_cast<const float4 *>(tbl.d[s + kGroup + u].data + (o - tbl.d[s + kGroup + u].start));
            }
            if (cur_ok) {
#pragma unroll
                for (uint32_t u = 0; u < kGroup; ++u) {  // the additions in insertion order (mixer.rs:185-198)
                    acc[0] += cur[u].x;
                    acc[1] += cur[u].y;
                    acc[2] += cur[u].z;
                    acc[3] += cur[u].w;
                }
            } else {
                for (uint32_t u = 0; u < kGroup; ++u) one(tbl.d[s + u]);
            }
#pragma unroll
            for (uint32_t u = 0; u < kGroup; ++u) cur[u] = nxt[u];
            cur_ok = nxt_ok;
        }
        for (; s < n_sources; ++s) one(tbl.d[s]);
        if (o + 4 <= out_len) {
            *reinterpret_cast<float4 *>(dst + o) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        } else {
            for (int k = 0; k < 4; ++k)
                if (o + k < out_len) dst[o + k] = acc[k];
        }
    }
}

}  // namespace

extern "C" {

rh_status rh_resample_out_frames(uint64_t in_frames, uint32_t from_rate, uint32_t to_rate, uint32_t channels, uint64_t span_len, uint64_t *out_frames) {
    if (!out_frames) return RH_ERR_INVALID;
    rh::ResampleGeom g;
    rh_status st = rh::make_resample_geom(in_frames, from_rate, to_rate, channels, span_len, &g);
    if (st != RH_OK) return st;
    *out_frames = g.out_frames;
    return RH_OK;
}

rh_status rh_resample_linear(float *dst, const float *src, uint64_t in_frames, uint32_t from_rate, uint32_t to_rate, uint32_t channels, uint64_t span_len, rh_stream stream) {
    RH_REQUIRE_INIT();
    rh::ResampleGeom g;
    rh_status st = rh::make_resample_geom(in_frames, from_rate, to_rate, channels, span_len, &g);
    if (st != RH_OK) return st;
    if (g.out_frames == 0) return RH_OK;
    if (!dst || !src) return RH_ERR_INVALID;
    hipStream_t s = rh::as_stream(stream);
    if (g.F == g.T) {  // sample_rate.rs:133-136 passthrough
        RH_HIP_TRY(rh::copy_d2d(dst, src, in_frames * channels * sizeof(float), s));
        return RH_OK;
    }
    if (!rh::knob(rh::K_PCM_NO_TILE)) {
        // what a tile moves, in + out: TF output frames need ~TF * F / T input frames.  32 KiB here (rh_wav.hip's tiles are best at 10): a tile
        // starts with a few divisions every lane does alike (where the tile sits in the row), and they want amortising -- 44.1 -> 48 kHz on
        // 512 MiB, stereo / mono / 5.1: 10 KiB 0.49 / 0.51 / 0.48 of 8 TB/s, 24 KiB 0.70 / 0.67 / 0.58, 32-48 KiB 0.72 / 0.64 / 0.59
        // (profiles/r06_rows.txt; the lane-per-frame kernel: 0.51 / 0.31 / 0.23)
        uint32_t kb = 32;
        if (const char *k = rh::knob(rh::K_PCM_TILE_KB)) kb = (uint32_t)std::atoi(k);
        if (kb < 1 || kb > 48) kb = 32;
        const double per_frame0 = 4.0 * channels * (1.0 + (double)g.F / (double)g.T);
        if (!rh::knob(rh::K_PCM_TILE_KB))  // a short row (a chain's block): smaller tiles, so that the launch is more than a handful of workgroups
            while (kb > 8 && (double)g.out_frames * per_frame0 < 128.0 * kb * 1024.0) kb /= 2;
        const double per_frame = 4.0 * channels * (1.0 + (double)g.F / (double)g.T);
        uint64_t tf = (uint64_t)(kb * 1024.0 / per_frame) & ~3ull;
        {   // whole periods of the converter where they fit: a tile that starts on a tap (tile_frames a multiple of T) finds its place in the row
            // by one multiplication (rA == 0 below) instead of a division every lane repeats
            const uint64_t period = std::lcm<uint64_t>(g.T, 4);
            if (g.n_chunks <= 1 && tf >= period) tf -= tf % period;
        }
        const uint64_t in_floats = ((uint64_t)((double)tf * g.F / g.T) + 8) * channels + 8;  // the run of a tile inside one chunk, with room for a few chunk boundaries
        const uint64_t A = tf * g.F;  // input positions a tile advances, in units of 1 / T of an input frame
        if (tf >= 16 && in_floats * 4 <= 56 * 1024 && (g.out_frames + tf - 1) / tf <= 0x7fffffffull && (uint64_t)tf * channels < (1ull << 30) && A + g.T < (1ull << 32) && g.T < (1u << 30)) {
            const uint32_t qA = (uint32_t)(A / g.T), rA = (uint32_t)(A % g.T);
            const int small_out = g.out_frames < (1ull << 32);
            const dim3 tgrid((unsigned)((g.out_frames + tf - 1) / tf));
            const uint32_t lds_floats = (uint32_t)((in_floats + 3) & ~3ull);
            const int vec_ok = reinterpret_cast<uintptr_t>(dst) % 16 == 0;
            const float invT = 1.0f / (float)g.T;       // for the position estimate (corrected by one)
            const uint32_t qF = g.F / g.T, rF = g.F % g.T;
#define RH_RST(P, CH) hipLaunchKernelGGL((k_resample_tile<P, CH>), tgrid, dim3(kBlock), lds_floats * 4, s, dst, src, g, channels, (uint32_t)tf, lds_floats, qA, rA, invT, qF, rF, small_out, vec_ok)
            if (g.fits32) {
                if (channels == 1) RH_RST(true, 1);
                else if (channels == 2) RH_RST(true, 2);
                else RH_RST(true, 0);
            } else {
                if (channels == 1) RH_RST(false, 1);
                else if (channels == 2) RH_RST(false, 2);
                else RH_RST(false, 0);
            }
#undef RH_RST
            RH_CHECK_LAUNCH();
            return RH_OK;
        }
    }
    const unsigned grid = rh::grid_for(g.out_frames);
    const bool f2 = channels == 2 && (reinterpret_cast<uintptr_t>(dst) % 8 == 0) && (reinterpret_cast<uintptr_t>(src) % 8 == 0);
    if (f2) {
        if (g.fits32) hipLaunchKernelGGL((k_resample_linear<2, true>), dim3(grid), dim3(kBlock), 0, s, dst, src, g, channels);
        else hipLaunchKernelGGL((k_resample_linear<2, false>), dim3(grid), dim3(kBlock), 0, s, dst, src, g, channels);
    } else if (channels == 1) {
        if (g.fits32) hipLaunchKernelGGL((k_resample_linear<1, true>), dim3(grid), dim3(kBlock), 0, s, dst, src, g, channels);
        else hipLaunchKernelGGL((k_resample_linear<1, false>), dim3(grid), dim3(kBlock), 0, s, dst, src, g, channels);
    } else {
        if (g.fits32) hipLaunchKernelGGL((k_resample_linear<0, true>), dim3(grid), dim3(kBlock), 0, s, dst, src, g, channels);
        else hipLaunchKernelGGL((k_resample_linear<0, false>), dim3(grid), dim3(kBlock), 0, s, dst, src, g, channels);
    }
    RH_CHECK_LAUNCH();
    return RH_OK;
}

rh_status rh_mix_sum(float *dst, size_t out_len, const float *const *srcs_host, const uint64_t *start_host, const uint64_t *len_host, uint32_t n_sources, rh_stream stream) {
    RH_REQUIRE_INIT();
    if (out_len == 0) return RH_OK;
    if (!dst || (n_sources && (!srcs_host || !start_host || !len_host))) return RH_ERR_INVALID;
    hipStream_t s = rh::as_stream(stream);
    if (n_sources == 0) {  // an empty mixer yields nothing (mixer.rs:131-135); caller asked for zeros
        RH_HIP_TRY(rh::fill_async(dst, 0, out_len * sizeof(float), s));
        return RH_OK;
    }
    bool vec_ok = (reinterpret_cast<uintptr_t>(dst) % 16 == 0);
    for (uint32_t i = 0; i < n_sources; ++i) {
        if (len_host[i] && !srcs_host[i]) return RH_ERR_INVALID;
        vec_ok = vec_ok && (start_host[i] % 4 == 0) && (reinterpret_cast<uintptr_t>(srcs_host[i]) % 16 == 0);
    }
    for (uint32_t first = 0; first < n_sources; first += kMixChunk) {
        const uint32_t n = n_sources - first < kMixChunk ? n_sources - first : kMixChunk;
        MixTable t;
        std::memset(&t, 0, sizeof(t));
        for (uint32_t i = 0; i < n; ++i) t.d[i] = MixDesc{srcs_host[first + i], start_host[first + i], len_host[first + i]};
        if (vec_ok) {
            const dim3 grid(rh::grid_tiles((out_len + 3) / 4));
            const bool grouped = rh::knob(rh::K_MIX_GROUPS) ? std::atoi(rh::knob(rh::K_MIX_GROUPS)) > 1 : (grid.x < 8u * (unsigned)rh::g_num_cus && n >= 16);
            if (grouped) {
                if (first) hipLaunchKernelGGL((k_mix_sum_v4_grp<true, 8>), grid, dim3(kBlock), 0, s, dst, (uint64_t)out_len, t, n);
                else hipLaunchKernelGGL((k_mix_sum_v4_grp<false, 8>), grid, dim3(kBlock), 0, s, dst, (uint64_t)out_len, t, n);
            } else {
                if (first) hipLaunchKernelGGL(k_mix_sum_v4<true>, grid, dim3(kBlock), 0, s, dst, (uint64_t)out_len, t, n);
                else hipLaunchKernelGGL(k_mix_sum_v4<false>, grid, dim3(kBlock), 0, s, dst, (uint64_t)out_len, t, n);
            }
        } else if (!rh::knob(rh::K_PCM_NO_TILE)) {
            const dim3 grid(rh::grid_tiles((out_len + 3) / 4));
            const int dst_vec = reinterpret_cast<uintptr_t>(dst) % 16 == 0;
            if (first) hipLaunchKernelGGL(k_mix_sum_any<true>, grid, dim3(kBlock), 0, s, dst, (uint64_t)out_len, t, n, dst_vec);
            else hipLaunchKernelGGL(k_mix_sum_any<false>, grid, dim3(kBlock), 0, s, dst, (uint64_t)out_len, t, n, dst_vec);
        } else {
            if (first) hipLaunchKernelGGL(k_mix_sum<true>, dim3(rh::grid_for(out_len)), dim3(kBlock), 0, s, dst, (uint64_t)out_len, t, n);
            else hipLaunchKernelGGL(k_mix_sum<false>, dim3(rh::grid_for(out_len)), dim3(kBlock), 0, s, dst, (uint64_t)out_len, t, n);
        }
        RH_CHECK_LAUNCH();
    }
    return RH_OK;
}

}  // extern "C"
