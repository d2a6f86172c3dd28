// rh_wav.hip -- the on-disk PCM format either side of the path:
//   ingest  src/decoder/wav.rs:94-172 (hound: 8-bit unsigned, 16/24/32-bit signed LE, 32-bit float ->
//           `to_sample::<f32>()`); a trailing partial frame is completed with silence (:161-169)
//   egress  src/wav_output.rs:62-96 (`wav_to_writer`: 32-bit float WAVE, whole frames only, :98-140)
// The RIFF chunk walk and the header are host work (rh_wav_probe_host / rh_wav_header_f32_host: plain C,
// no GPU); the sample conversion runs on the device straight from the file bytes, so a file -> file job
// never does the int -> float pass on the CPU.  hound is an un-vendored dependency (Cargo.lock): its
// byte-level behaviour is restated from the WAVE specification and pinned by the six files the reference's own test decodes
// (tests/wav_test.rs:1-33; tests/golden/wav/) against scipy.io.wavfile, bit for bit (tests/test_wav_assets.py).
#include <cstdlib>
#include <cstring>

#include "rh_common.h"

namespace {

constexpr int kBlock = 256;

// 24-bit packed little-endian samples: byte address 3*i is unaligned, so every lane reads 3 bytes.
__global__ __launch_bounds__(kBlock) void k_pcm24_to_f32(float *__restrict__ dst, const uint8_t *__restrict__ src, size_t n) {
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const uint8_t *b = src + 3 * i;
        int32_t v = (int32_t)b[0] | ((int32_t)b[1] << 8) | ((int32_t)(int8_t)b[2] << 16);  // sign-extended I24 (wav.rs:126-131)
        dst[i] = (float)v / 8388608.0f;
    }
}
// zero-fill [from, to): the silence that completes a cut frame
__global__ __launch_bounds__(kBlock) void k_fill_zero(float *__restrict__ dst, size_t from, size_t to) {
    const size_t i = from + (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < to) dst[i] = 0.0f;
}

// ---- decode AND ChannelCountConverter in one pass (BASELINE config 5: the decoded block goes straight into the converter) -------------
// out[f][k] = k < from ? sample(f, k) : (k == 1 && from == 1 ? sample(f, 0) : 0.0)   (channels.rs:59-70), sample = hound's to_sample::<f32>()
// of the file's bytes (wav.rs:107-151), 0.0 where the data chunk ends inside the last frame (wav.rs:161-169).  The two-launch form writes
// the decoded f32 block (4 B a sample) and reads it again; here a 6-channel PCM16 frame costs 12 B in + 4 * to B out instead of 36 + 24 + 4 * to.
// A lane produces four consecutive output samples (one 16-byte store); its few loads hit lines its neighbours touch too.
// BYTES: the data chunk sits where the chunks in front of it left it -- the reference's own assets/audacity32bit_int.wav has its 32-bit
// samples at file offset 102, lmms16bit.wav at 94 -- so a caller that uploads the file and passes `file + data_offset` hands over samples
// at ANY byte address; there every sample is put together from its bytes (hound reads a byte stream and never sees an address).
template <int FMT, bool BYTES>  // 0: u8, 1: i16, 2: packed i24, 3: i32, 4: f32 (all little-endian, as the file holds them)
__device__ __forceinline__ float pcm_sample(const uint8_t *__restrict__ src, uint64_t i) {
    if (FMT == 0) return (float)((int)src[i] - 128) / 128.0f;
    if (FMT == 1) {
        if (!BYTES) return (float)reinterpret_cast<const int16_t *>(src)[i] / 32768.0f;
        const uint8_t *b = src + 2 * i;
        return (float)((int32_t)b[0] | ((int32_t)(int8_t)b[1] << 8)) / 32768.0f;
    }
    if (FMT == 2) {
        const uint8_t *b = src + 3 * i;
        const int32_t v = (int32_t)b[0] | ((int32_t)b[1] << 8) | ((int32_t)(int8_t)b[2] << 16);
        return (float)v / 8388608.0f;
    }
    uint32_t w;
    if (!BYTES) {
        w = reinterpret_cast<const uint32_t *>(src)[i];
    } else {
        const uint8_t *b = src + 4 * i;
        w = (uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16) | ((uint32_t)b[3] << 24);
    }
    if (FMT == 3) return (float)(int32_t)w / 2147483648.0f;
    return __uint_as_float(w);
}
template <int FMT, bool BYTES>
__global__ __launch_bounds__(kBlock) void k_pcm_to_channels(float *__restrict__ dst, const uint8_t *__restrict__ src, uint64_t n_samples, uint64_t frames, uint32_t from, uint32_t to, int vec_ok) {
    const uint64_t total = frames * to, nvec = (total + 3) / 4;
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t v = (uint64_t)blockIdx.x * kBlock + threadIdx.x; v < nvec; v += stride) {
        const uint64_t o0 = 4 * v;
        uint64_t f = o0 / to;
        uint32_t k = (uint32_t)(o0 - f * to);
        float e[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool has = k < from || (k == 1u && from == 1u);
            const uint64_t idx = f * from + (k < from ? k : 0u);
            e[i] = (has && o0 + i < total && idx < n_samples) ? pcm_sample<FMT, BYTES>(src, idx) : 0.0f;
            if (++k == to) k = 0, ++f;
        }
        if (vec_ok && o0 + 4 <= total) {
            rh::st_nt(reinterpret_cast<float4 *>(dst + o0), make_float4(e[0], e[1], e[2], e[3]));
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (o0 + i < total) dst[o0 + i] = e[i];
        }
    }
}

// The same, a TILE of frames per workgroup (round 6): the tile's bytes come in as aligned 16-byte vectors, every byte of the file once and
// in address order whatever the frame size (a 6-channel PCM16 frame is 12 bytes of which 6 -> 2 keeps 4: the lane-per-output form above asks
// for them as 2-byte loads 12 bytes apart, four instructions over the same lines for one store), are parked in LDS, and the lanes pick their
// samples there.  The vectors are aligned in MEMORY, not in the file: the first one starts up to 15 bytes in front of the tile (inside the same
// 16 bytes as its first byte, so inside the caller's buffer's page), `shift` says where the tile begins.
template <int FMT, bool BYTES>
__device__ __forceinline__ float pcm_sample_lds(const uint8_t *lds, uint32_t byte) {
    if (FMT == 0) return (float)((int)lds[byte] - 128) / 128.0f;
    if (FMT == 1) {
        if (!BYTES) return (float)*reinterpret_cast<const int16_t *>(lds + byte) / 32768.0f;
        return (float)((int32_t)lds[byte] | ((int32_t)(int8_t)lds[byte + 1] << 8)) / 32768.0f;
    }
    if (FMT == 2) {
        const int32_t v = (int32_t)lds[byte] | ((int32_t)lds[byte + 1] << 8) | ((int32_t)(int8_t)lds[byte + 2] << 16);
        return (float)v / 8388608.0f;
    }
    uint32_t w;
    if (!BYTES) w = *reinterpret_cast<const uint32_t *>(lds + byte);
    else w = (uint32_t)lds[byte] | ((uint32_t)lds[byte + 1] << 8) | ((uint32_t)lds[byte + 2] << 16) | ((uint32_t)lds[byte + 3] << 24);
    if (FMT == 3) return (float)(int32_t)w / 2147483648.0f;
    return __uint_as_float(w);
}
template <int FMT, bool BYTES>
__global__ __launch_bounds__(kBlock) void k_pcm_to_channels_tile(float *__restrict__ dst, const uint8_t *__restrict__ src, uint64_t n_samples, uint64_t frames, uint32_t from, uint32_t to,
                                                                  uint32_t tile_frames, int vec_ok) {
    extern __shared__ uint4 pcm_tile[];
    constexpr uint32_t bps = FMT == 0 ? 1u : FMT == 1 ? 2u : FMT == 2 ? 3u : 4u;
    const uint64_t f0 = (uint64_t)blockIdx.x * tile_frames;  // (tile_frames is a multiple of 4: a tile's first output sample starts a 16-byte vector)
    const uint32_t nf = (uint32_t)(frames - f0 < tile_frames ? frames - f0 : tile_frames);
    const uint64_t s0 = f0 * from;
    const uint32_t n_in = (uint32_t)(n_samples - s0 < (uint64_t)nf * from ? n_samples - s0 : (uint64_t)nf * from);  // the data chunk may end inside the last frame
    const uintptr_t p0 = reinterpret_cast<uintptr_t>(src) + s0 * bps, a0 = p0 & ~(uintptr_t)15;
    const uint32_t shift = (uint32_t)(p0 - a0), nvec = (shift + n_in * bps + 15u) / 16u;
    for (uint32_t v = threadIdx.x; v < nvec; v += kBlock) pcm_tile[v] = rh::ld_nt(reinterpret_cast<const uint4 *>(a0) + v);
    __syncthreads();
    const uint8_t *lds = reinterpret_cast<const uint8_t *>(pcm_tile) + shift;
    const uint32_t total = nf * to, nv = (total + 3u) / 4u;
    float *out = dst + f0 * to;
    for (uint32_t v = threadIdx.x; v < nv; v += kBlock) {
        const uint32_t o0 = 4u * v;
        uint32_t f = o0 / to, k = o0 - f * to;
        float e[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool has = k < from || (k == 1u && from == 1u);
            const uint32_t idx = f * from + (k < from ? k : 0u);
            e[i] = (has && o0 + i < total && idx < n_in) ? pcm_sample_lds<FMT, BYTES>(lds, idx * bps) : 0.0f;
            if (++k == to) k = 0, ++f;
        }
        if (vec_ok && o0 + 4u <= total) {
            rh::st_nt(reinterpret_cast<float4 *>(out + o0), make_float4(e[0], e[1], e[2], e[3]));
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (o0 + i < total) out[o0 + i] = e[i];
        }
    }
}

uint32_t rd32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
uint16_t rd16(const uint8_t *p) { return (uint16_t)(p[0] | (p[1] << 8)); }
void wr32(uint8_t *p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); }
void wr16(uint8_t *p, uint16_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); }

}  // namespace

namespace rh {
// A tile of frames per workgroup (k_pcm_to_channels_tile) wherever it pays: launches and returns true, or returns false and the caller
// runs its lane-per-output kernel.  Sized by what a tile MOVES, in + out: tools/bench_channels.py on 768 MiB (profiles/r06_channels_tile.txt)
// -- 6 -> 2 from PCM16 is best with ~10 KiB a tile, 2 -> 6 with less input still (its output is three times its input); against the
// lane-per-output kernels 0.43-0.68 -> 0.72-0.82 of 8 TB/s.  fmt: 0 u8, 1 i16, 2 packed i24, 3 i32, 4 f32.
bool pcm_tile_try(float *dst, const uint8_t *data, uint64_t n_samples, uint64_t frames, uint32_t channels, uint32_t to_channels, int fmt, hipStream_t s) {
    static const uint32_t bps_of[5] = {1, 2, 3, 4, 4};
    if (knob(K_PCM_NO_TILE) || frames == 0) return false;
    const uint64_t frame_in = (uint64_t)channels * bps_of[fmt], frame_out = 4ull * to_channels;
    uint32_t kb = 10;
    if (const char *k = knob(K_PCM_TILE_KB)) kb = (uint32_t)std::atoi(k);
    if (kb < 1 || kb > 48) kb = 10;
    uint64_t tf = ((uint64_t)kb * 1024 / (frame_in + frame_out)) & ~3ull;  // frames a tile: a multiple of 4 (a tile's output starts a 16-byte vector)
    if (tf < 8 || tf * frame_in > 60 * 1024 || (frames + tf - 1) / tf > 0x7fffffffull) return false;  // frames of hundreds of channels: a lane per output sample
    const uintptr_t align = fmt == 1 ? 1u : (fmt >= 3 ? 3u : 0u);
    const bool bytes = (reinterpret_cast<uintptr_t>(data) & align) != 0;  // file + data_offset: any byte address (see pcm_sample)
    const int vec_ok = (reinterpret_cast<uintptr_t>(dst) & 15u) == 0;
    const dim3 tgrid((unsigned)((frames + tf - 1) / tf));
    const size_t lds = (size_t)(tf * frame_in) + 32;
#define RH_PCMT(F, B) hipLaunchKernelGGL((k_pcm_to_channels_tile<F, B>), tgrid, dim3(kBlock), lds, s, dst, data, n_samples, frames, channels, to_channels, (uint32_t)tf, vec_ok)
    if (fmt == 0) RH_PCMT(0, false);
    else if (fmt == 2) RH_PCMT(2, false);
    else if (fmt == 1) { if (bytes) RH_PCMT(1, true); else RH_PCMT(1, false); }
    else if (fmt == 3) { if (bytes) RH_PCMT(3, true); else RH_PCMT(3, false); }
    else { if (bytes) RH_PCMT(4, true); else RH_PCMT(4, false); }
#undef RH_PCMT
    return true;
}
}  // namespace rh

extern "C" {

rh_status rh_wav_probe_host(const uint8_t *bytes, size_t size, rh_wav_info *info) {
    if (!bytes || !info || size < 12) return RH_ERR_INVALID;
    if (std::memcmp(bytes, "RIFF", 4) != 0 || std::memcmp(bytes + 8, "WAVE", 4) != 0) return RH_ERR_INVALID;
    std::memset(info, 0, sizeof(*info));
    bool have_fmt = false;
    size_t pos = 12;
    while (pos + 8 <= size) {
        const uint32_t len = rd32(bytes + pos + 4);
        const uint8_t *body = bytes + pos + 8;
        if (std::memcmp(bytes + pos, "fmt ", 4) == 0) {
            if (len < 16 || pos + 8 + 16 > size) return RH_ERR_INVALID;
            uint16_t tag = rd16(body);
            info->channels = rd16(body + 2);
            info->sample_rate = rd32(body + 4);
            info->bits_per_sample = rd16(body + 14);
            if (tag == 0xFFFE && len >= 40 && pos + 8 + 40 <= size) tag = rd16(body + 24);  // WAVE_FORMAT_EXTENSIBLE: sub-format GUID
            info->is_float = tag == 3;
            if (tag != 1 && tag != 3) return RH_ERR_UNSUPPORTED;
            have_fmt = true;
        } else if (std::memcmp(bytes + pos, "data", 4) == 0) {
            if (!have_fmt) return RH_ERR_INVALID;
            info->data_offset = pos + 8;
            uint64_t n = len;
            if (pos + 8 + n > size) n = size - (pos + 8);  // truncated file: what is there
            info->data_bytes = n;
            const uint32_t bps = (info->bits_per_sample + 7u) / 8u;
            if (bps == 0 || info->channels == 0 || info->sample_rate == 0) return RH_ERR_INVALID;
            info->samples = n / bps;
            return RH_OK;
        }
        pos += 8 + (size_t)len + (len & 1u);
    }
    return RH_ERR_INVALID;
}

rh_status rh_wav_decode(float *dst, const uint8_t *data, uint64_t n_samples, uint32_t channels, uint32_t bits_per_sample, int32_t is_float, uint64_t *out_samples, rh_stream stream) {
    RH_REQUIRE_INIT();
    if (channels == 0 || !out_samples) return RH_ERR_INVALID;
    const uint64_t rem = n_samples % channels;
    const uint64_t total = n_samples + (rem ? channels - rem : 0);  // wav.rs:161-169
    *out_samples = total;
    if (total == 0) return RH_OK;
    if (!dst || (n_samples && !data)) return RH_ERR_INVALID;
    hipStream_t s = rh::as_stream(stream);
    rh_status st = RH_OK;
    {   // `file + data_offset`: the canonical header is 44 bytes, so the samples of an uploaded file image start 4 bytes off a 16-byte boundary (and
        // anywhere behind odd-sized chunks).  The converters below want rows on vector boundaries and fall back to a sample a lane otherwise; the
        // tile kernel reads aligned vectors AROUND the row and does not care, at the same rate (from == to: every frame verbatim, the cut one
        // completed with silence).  Samples off their OWN size's boundary always go there (put together from their bytes).
        const uintptr_t a = reinterpret_cast<uintptr_t>(data);
        const bool off_size = (bits_per_sample == 16 && (a & 1u)) || (bits_per_sample == 32 && (a & 3u));
        const bool off_vector = (bits_per_sample == 16 && (a & 7u)) || (bits_per_sample == 8 && (a & 3u)) || (bits_per_sample == 32 && !is_float && (a & 15u));
        if (off_size) return rh_wav_decode_channels(dst, data, n_samples, channels, bits_per_sample, is_float, channels, out_samples, stream);
        if (off_vector && !is_float) {
            const int fmt = bits_per_sample == 8 ? 0 : bits_per_sample == 16 ? 1 : 3;
            if (rh::pcm_tile_try(dst, data, n_samples, total / channels, channels, channels, fmt, s)) {
                RH_CHECK_LAUNCH();
                return RH_OK;
            }
        }
    }
    if (is_float) {
        if (bits_per_sample != 32) return RH_ERR_UNSUPPORTED;  // wav.rs:107-117
        RH_HIP_TRY(rh::copy_d2d(dst, data, n_samples * 4, s));
    } else if (bits_per_sample == 8) {
        st = rh_convert_u8_to_f32(dst, data, n_samples, stream);  // the file holds unsigned bytes; hound hands rodio i8 = b - 128
    } else if (bits_per_sample == 16) {
        st = rh_convert_i16_to_f32(dst, reinterpret_cast<const int16_t *>(data), n_samples, stream);
    } else if (bits_per_sample == 24) {
        // packed 3-byte samples: the tile kernel with the layout kept (its 16-byte loads carry 5.3 samples each; three byte loads a lane measured
        // 0.52 of 8 TB/s against 0.74-0.80, profiles/r06_channels_tile.txt); it completes the cut frame itself
        if (rh::pcm_tile_try(dst, data, n_samples, total / channels, channels, channels, 2, s)) {
            RH_CHECK_LAUNCH();
            return RH_OK;
        }
        if (n_samples) {
            hipLaunchKernelGGL(k_pcm24_to_f32, dim3(rh::grid_for(n_samples)), dim3(kBlock), 0, s, dst, data, (size_t)n_samples);
            RH_CHECK_LAUNCH();
        }
    } else if (bits_per_sample == 32) {
        st = rh_convert_i32_to_f32(dst, reinterpret_cast<const int32_t *>(data), n_samples, stream);
    } else {
        return RH_ERR_UNSUPPORTED;  // "unofficial" depths (wav.rs:137-151)
    }
    if (st != RH_OK) return st;
    if (total > n_samples) {
        hipLaunchKernelGGL(k_fill_zero, dim3((unsigned)((total - n_samples + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, dst, (size_t)n_samples, (size_t)total);  // up to channels - 1 samples: a header may say 65 535 channels
        RH_CHECK_LAUNCH();
    }
    return RH_OK;
}

rh_status rh_wav_decode_channels(float *dst, const uint8_t *data, uint64_t n_samples, uint32_t channels, uint32_t bits_per_sample, int32_t is_float, uint32_t to_channels,
                                 uint64_t *out_samples, rh_stream stream) {
    RH_REQUIRE_INIT();
    if (channels == 0 || to_channels == 0 || !out_samples) return RH_ERR_INVALID;
    const uint64_t frames = (n_samples + channels - 1) / channels;  // a cut last frame is completed with silence (wav.rs:161-169) and converted like the others
    *out_samples = frames * to_channels;
    if (frames == 0) return RH_OK;
    if (!dst || !data) return RH_ERR_INVALID;
    int fmt;
    if (is_float) {
        if (bits_per_sample != 32) return RH_ERR_UNSUPPORTED;  // wav.rs:107-117
        fmt = 4;
    } else if (bits_per_sample == 8) fmt = 0;
    else if (bits_per_sample == 16) fmt = 1;
    else if (bits_per_sample == 24) fmt = 2;
    else if (bits_per_sample == 32) fmt = 3;
    else return RH_ERR_UNSUPPORTED;  // "unofficial" depths (wav.rs:137-151)
    const uintptr_t align = fmt == 1 ? 1u : (fmt >= 3 ? 3u : 0u);
    const bool bytes = (reinterpret_cast<uintptr_t>(data) & align) != 0;  // file + data_offset: any byte address (see pcm_sample)
    const int vec_ok = (reinterpret_cast<uintptr_t>(dst) & 15u) == 0;
    hipStream_t s = rh::as_stream(stream);
    if (rh::pcm_tile_try(dst, data, n_samples, frames, channels, to_channels, fmt, s)) {
        RH_CHECK_LAUNCH();
        return RH_OK;
    }
    const dim3 grid(rh::grid_tiles((size_t)((frames * to_channels + 3) / 4)));
#define RH_PCM(F, B) hipLaunchKernelGGL((k_pcm_to_channels<F, B>), grid, dim3(kBlock), 0, s, dst, data, n_samples, frames, channels, to_channels, vec_ok)
    if (fmt == 0) RH_PCM(0, false);
    else if (fmt == 2) RH_PCM(2, false);
    else if (fmt == 1) { if (bytes) RH_PCM(1, true); else RH_PCM(1, false); }
    else if (fmt == 3) { if (bytes) RH_PCM(3, true); else RH_PCM(3, false); }
    else { if (bytes) RH_PCM(4, true); else RH_PCM(4, false); }
#undef RH_PCM
    RH_CHECK_LAUNCH();
    return RH_OK;
}

size_t rh_wav_header_f32_host(uint8_t *out, size_t cap, uint32_t channels, uint32_t sample_rate, uint64_t n_samples) {
    // canonical 44-byte header, format tag 3 (IEEE float), 32 bits -- what wav_output.rs:66-71 asks hound for
    if (!out || cap < 44 || channels == 0 || channels > 65535) return 0;
    const uint64_t whole = n_samples - n_samples % channels;  // WholeFrames, wav_output.rs:98-140
    const uint64_t bytes = whole * 4;
    if (bytes > 0xffffffffull - 36) return 0;
    std::memcpy(out, "RIFF", 4);
    wr32(out + 4, (uint32_t)(36 + bytes));
    std::memcpy(out + 8, "WAVEfmt ", 8);
    wr32(out + 16, 16);
    wr16(out + 20, 3);
    wr16(out + 22, (uint16_t)channels);
    wr32(out + 24, sample_rate);
    wr32(out + 28, sample_rate * channels * 4);
    wr16(out + 32, (uint16_t)(channels * 4));
    wr16(out + 34, 32);
    std::memcpy(out + 36, "data", 4);
    wr32(out + 40, (uint32_t)bytes);
    return 44;
}

}  // extern "C"
