// Driver for tests/test_host_mirror.py: runs rodio-style pull chains through include/rodio_hip.hpp (the C++
// host mirror over the C ABI) on raw f32 files and writes what `next()` returned.  Test infrastructure:
// the expected values come from the oracle on the Python side.
//
//   host_mirror_test mixer <dir> <S> <from_rate> <to_rate> <filter_kind> <freq> <block_frames> <frames_per_lane>
//       <dir>/src_<i>.f32 (stereo, interleaved), <dir>/gains.f32  ->  <dir>/out.f32
//   host_mirror_test selftest
//       host-only checks of the Source mirror (no GPU): the reference's SamplesBuffer tests, buffer.rs:148-207
//   host_mirror_test mixany <dir> <S> <to_rate> <filter_kind> <freq> <block_frames> <frames_per_lane>
//       like mixer, but every source has its own layout: <dir>/spec.txt holds "channels rate gain" per source
//   host_mirror_test late <dir> <S0> <S1> <from_rate> <to_rate> <filter_kind> <freq> <block_frames> <frames_per_lane> <pull_first>
//       sources 0..S0-1 are added before the first next(); after <pull_first> samples were served, S1 more are added
//       (Mixer::add on a running mixer).  Writes out.f32 and join.txt (the output frame at which they joined).
//   host_mirror_test bench <S> <frames> <block_frames> [host_threads]
//       times the pull path end to end (host samples in, mixed host samples out: PCIe inclusive) on S synthetic sources
//   host_mirror_test chain <dir> <channels> <rate> <block_frames> <op> [<op> ...]
//       <dir>/src_0.f32  ->  <dir>/out.f32 ; ops: amplify:F low_pass:HZ high_pass:HZ reverb:NS:AMP uniform:CH:RATE take:NS:FADE delay:NS
//       channels:N limit agc fade_in:NS fade_out:NS distortion:G:T dither:BITS:ALG:SEED channel_volume:G0,G1,.. spatial exact (filters in reference order)
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <sstream>
#include <string>
#include <vector>

#include "rodio_hip.hpp"

namespace rh = rodio_hip;

static std::vector<float> read_f32(const std::string &path) {
    std::vector<float> v;
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("cannot open " + path);
    std::fseek(f, 0, SEEK_END);
    const long n = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    v.resize((size_t)n / 4);
    if (n && std::fread(v.data(), 4, v.size(), f) != v.size()) throw std::runtime_error("short read " + path);
    std::fclose(f);
    return v;
}
static void write_f32(const std::string &path, const std::vector<float> &v) {
    FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) throw std::runtime_error("cannot write " + path);
    if (!v.empty()) std::fwrite(v.data(), 4, v.size(), f);
    std::fclose(f);
}
// the block size of a run; RH_TEST_BLOCK overrides what the command line says (tools/fuzz_mixer_chains.py: the same cases in blocks of 64 or 100 000 frames)
static size_t block_arg(const char *a) {
    const char *e = std::getenv("RH_TEST_BLOCK");
    return (size_t)std::atoll(e && *e ? e : a);
}
static std::vector<std::string> split(const std::string &s, char sep) {
    std::vector<std::string> out;
    std::stringstream ss(s);
    std::string item;
    while (std::getline(ss, item, sep)) out.push_back(item);
    return out;
}
// The sources of the chains.  RH_TEST_SOURCE picks what `current_span_len()` says (the samples are the same):
//   test (default)  None -- the TestSource of rodio's benches (benches/shared.rs:32-34): one continuous stream
//   buffer          rodio's SamplesBuffer: Some(len) until exhausted, then Some(0)  (buffer.rs:76-82)
//   spans:K         Some(K) throughout: spans of K samples, like a decoder's packets or Buffered's spans (buffered.rs:109)
//   mixed           source i: test / buffer / spans:4096 by i % 3
class TestSource : public rh::SamplesBuffer {
public:
    using rh::SamplesBuffer::SamplesBuffer;
    std::optional<std::size_t> current_span_len() const override { return std::nullopt; }
    rh::SizeHint size_hint() const override { return rh::SizeHint{}; }  // benches/shared.rs:14-21 implements only next(): the trait's default (0, None); its total_duration is GIVEN (:47-49; here: the buffer's)
};
class SpanSource : public rh::SamplesBuffer {
public:
    SpanSource(std::uint16_t ch, std::uint32_t rate, std::vector<float> d, std::size_t span) : rh::SamplesBuffer(ch, rate, std::move(d)), span_(span) {}
    std::optional<std::size_t> current_span_len() const override { return span_; }

private:
    std::size_t span_;
};
// A source of several spans with formats of their own (the oracle's SeqSource): current_span_len() = Some(len of the current part); the moment
// a part's last sample has been taken the next part's parameters are the ones reported; Some(0) when everything has been taken.
class SeqSource : public rh::Source {
public:
    struct Part {
        std::vector<float> data;
        std::uint16_t ch;
        std::uint32_t rate;
    };
    explicit SeqSource(std::vector<Part> parts) : parts_(std::move(parts)) { settle(); }
    std::optional<float> next() override {
        settle();
        if (cur_ >= parts_.size() || pos_ >= parts_[cur_].data.size()) return std::nullopt;
        const float v = parts_[cur_].data[pos_++];
        settle();
        return v;
    }
    std::optional<std::size_t> current_span_len() const override {
        if (cur_ >= parts_.size() || pos_ >= parts_[cur_].data.size()) return 0;
        return parts_[cur_].data.size();
    }
    std::uint16_t channels() const override { return parts_[std::min(cur_, parts_.size() - 1)].ch; }
    std::uint32_t sample_rate() const override { return parts_[std::min(cur_, parts_.size() - 1)].rate; }

private:
    void settle() {
        while (cur_ < parts_.size() && pos_ >= parts_[cur_].data.size()) {
            if (cur_ + 1 == parts_.size()) break;
            ++cur_;
            pos_ = 0;
        }
    }
    std::vector<Part> parts_;
    std::size_t cur_ = 0, pos_ = 0;
};
// <dir>/seq_<index>.txt ("channels rate" per part, samples in <dir>/seq_<index>_<k>.f32) makes source <index> a SeqSource
static rh::BoxSource make_seq_source(const std::string &dir, int index) {
    std::FILE *f = std::fopen((dir + "/seq_" + std::to_string(index) + ".txt").c_str(), "r");
    if (!f) return nullptr;
    std::vector<SeqSource::Part> parts;
    unsigned ch = 0, rate = 0;
    for (int k = 0; std::fscanf(f, "%u %u", &ch, &rate) == 2; ++k)
        parts.push_back(SeqSource::Part{read_f32(dir + "/seq_" + std::to_string(index) + "_" + std::to_string(k) + ".f32"), (std::uint16_t)ch, rate});
    std::fclose(f);
    if (parts.empty()) throw std::runtime_error("seq source without parts");
    return std::make_unique<SeqSource>(std::move(parts));
}
static rh::BoxSource make_source(std::uint16_t ch, std::uint32_t rate, std::vector<float> data, int index = 0) {
    const char *e = std::getenv("RH_TEST_SOURCE");
    std::string kind = e ? e : "test";
    if (kind == "mixed") kind = index % 3 == 0 ? "test" : index % 3 == 1 ? "buffer" : "spans:4096";  // one mixer, all three
    if (kind == "buffer") return std::make_unique<rh::SamplesBuffer>(ch, rate, std::move(data));
    if (kind.rfind("spans:", 0) == 0) return std::make_unique<SpanSource>(ch, rate, std::move(data), (std::size_t)std::atoll(kind.c_str() + 6));
    if (kind != "test") throw std::runtime_error("RH_TEST_SOURCE=" + kind);
    return std::make_unique<TestSource>(ch, rate, std::move(data));
}
// the consumer: like the cpal callback, it takes samples one at a time; every so often in bulk, as wav_to_writer would
// RH_TEST_TRACK_HINTS (the mixer modes): size_hint() in front of every sample drain() takes and behind the last (written to hints.i64 as
// lower, upper or -1 at exit; `late` / `latewide`: the samples pulled before the late add are not tracked, the file starts at the join)
static std::vector<long long> g_hints;
static void note_hint(const rh::Source &src) {
    const rh::SizeHint h = src.size_hint();
    g_hints.push_back((long long)h.lower);
    g_hints.push_back(h.upper ? (long long)*h.upper : -1);
}
static std::vector<float> drain(rh::Source &src) {
    std::vector<float> out;
    if (std::getenv("RH_TEST_TRACK_HINTS")) {
        for (;;) {
            note_hint(src);
            const std::optional<float> v = src.next();
            if (!v) break;
            out.push_back(*v);
        }
        return out;
    }
    float chunk[777];
    for (int round = 0;; ++round) {
        if (round % 3 == 2) {
            const size_t k = src.read(chunk, 777);
            out.insert(out.end(), chunk, chunk + k);
            if (k < 777) break;
        } else {
            bool ended = false;
            for (int i = 0; i < 1000; ++i) {
                const std::optional<float> v = src.next();
                if (!v) {
                    ended = true;
                    break;
                }
                out.push_back(*v);
            }
            if (ended) break;
        }
    }
    if (src.next()) throw std::runtime_error("a source that ended produced another sample");
    return out;
}

// One adapter of a chain, in the driver's spelling (see the header of this file).
static void apply_op(rh::GpuSource &g, const std::string &arg) {
    const std::vector<std::string> t = split(arg, ':');
    const std::string &op = t[0];
    if (op == "exact") g.exact_filters(true);
    else if (op == "amplify") g.amplify(std::stof(t.at(1)));
    else if (op == "low_pass") g.low_pass((uint32_t)std::stoul(t.at(1)));
    else if (op == "high_pass") g.high_pass((uint32_t)std::stoul(t.at(1)));
    else if (op == "reverb") g.reverb(rh::Nanos(std::stoll(t.at(1))), std::stof(t.at(2)));
    else if (op == "uniform") g.uniform((uint16_t)std::stoul(t.at(1)), (uint32_t)std::stoul(t.at(2)));
    else if (op == "channels") g.convert_channels((uint16_t)std::stoul(t.at(1)));
    else if (op == "limit") g.limit(rh_limit_params{-1.0f, 4.0f, 5000000ull, 100000000ull});
    else if (op == "agc") g.automatic_gain_control(rh_agc_params{1.0f, 4000000000ull, 0ull, 7.0f, 0.0f});
    else if (op == "take") g.take_duration(rh::Nanos(std::stoll(t.at(1))), std::stoi(t.at(2)) != 0);
    else if (op == "delay") g.delay(rh::Nanos(std::stoll(t.at(1))));
    else if (op == "fade_in") g.fade_in(rh::Nanos(std::stoll(t.at(1))));
    else if (op == "fade_out") g.fade_out(rh::Nanos(std::stoll(t.at(1))));
    else if (op == "dither") g.dither((uint32_t)std::stoul(t.at(1)), (rh::GpuSource::DitherAlgorithm)std::stoi(t.at(2)), std::stoull(t.at(3)));
    else if (op == "distortion") g.distortion(std::stof(t.at(1)), std::stof(t.at(2)));
    else if (op == "channel_volume") {
        std::vector<float> gains;
        for (const std::string &x : split(t.at(1), ',')) gains.push_back(std::stof(x));
        g.channel_volume(gains);
    } else if (op == "spatial") {
        const float e[3] = {0.5f, 0.0f, 1.0f}, l[3] = {-1.0f, 0.0f, 0.0f}, r[3] = {1.0f, 0.0f, 0.0f};
        g.spatial(e, l, r);
    } else throw std::runtime_error("unknown op " + op);
}

int main(int argc, char **argv) {
    if (argc == 2 && std::string(argv[1]) == "selftest") {
        auto expect = [](bool ok, const char *what) {
            if (!ok) {
                std::fprintf(stderr, "selftest failed: %s\n", what);
                std::exit(1);
            }
        };
        {  // buffer.rs:153-159 duration_basic
            rh::SamplesBuffer buf(2, 2, std::vector<float>(6, 0.0f));
            expect(buf.total_duration() && buf.total_duration()->count() == 1500000000ll, "duration_basic");
        }
        {  // buffer.rs:161-171 iteration
            rh::SamplesBuffer buf(1, 44100, {1, 2, 3, 4, 5, 6});
            for (float v = 1; v <= 6; v += 1) expect(buf.next() == std::optional<float>(v), "iteration");
            expect(!buf.next() && !buf.next(), "iteration end");
        }
        {  // buffer.rs:180-205 try_seek::channel_order_stays_correct
            std::vector<float> d(2000);
            for (int i = 0; i < 2000; ++i) d[(size_t)i] = (float)i;
            rh::SamplesBuffer buf(2, 100, d);
            expect(buf.try_seek(std::chrono::seconds(5)), "seek supported");
            expect(buf.next() == std::optional<float>(5.0f * 100 * 2), "seek lands on the frame");
            auto odd = [&]() { const auto s = buf.next(); return s && ((int)*s % 2 == 1); };
            auto even = [&]() { const auto s = buf.next(); return s && ((int)*s % 2 == 0); };
            expect(odd(), "channel 1 after the seek");
            expect(even(), "channel 0 next");
            expect(buf.try_seek(std::chrono::seconds(6)) && odd(), "a seek from channel 1 continues with channel 1");
            expect(buf.try_seek(std::chrono::seconds(1000)) && !buf.next(), "seek saturates at the end");
        }
        {  // buffer.rs:76-82: the whole buffer is one span; Some(0) once exhausted (is_exhausted, source/mod.rs:203-207)
            rh::SamplesBuffer buf(2, 48000, {1, 2, 3, 4});
            expect(buf.current_span_len() == std::optional<std::size_t>(4), "span = len");
            (void)buf.next();
            expect(buf.current_span_len() == std::optional<std::size_t>(4), "span stays the total length");
            for (int i = 0; i < 3; ++i) (void)buf.next();
            expect(buf.current_span_len() == std::optional<std::size_t>(0), "Some(0) when exhausted");
        }
        {  // SpanReader pulls like UniformSourceIterator (uniform.rs:50-68): min(span, 32768) samples per chain, then it asks again
            std::vector<float> d(70000);
            for (size_t i = 0; i < d.size(); ++i) d[i] = (float)i;
            rh::SamplesBuffer buf(2, 44100, d);
            rh::detail::SpanReader rd(&buf);
            std::vector<float> tmp(70000);
            rh::detail::Piece pc;
            expect(rd.read_piece(tmp.data(), 20000, pc) && pc.n == 32768 && pc.opens && pc.closes && pc.ch == 2 && pc.rate == 44100, "first span: 32768 samples");
            expect(rd.read_piece(tmp.data(), 10000, pc) && pc.n == 20000 && pc.opens && !pc.closes, "second span opens");
            expect(rd.read_piece(tmp.data(), 100000, pc) && pc.n == 12768 && !pc.opens && pc.closes && tmp[0] == 52768.0f, "second span closes at 32768");
            expect(rd.read_piece(tmp.data(), 100000, pc) && pc.n == 70000 - 65536 && pc.opens && pc.closes && rd.ended(), "the last span is what is left: the source returns None inside it");
            expect(!rd.read_piece(tmp.data(), 100000, pc), "the stream is over");
            {  // min(span, 32768) cuts a frame of 6 channels (32768 = 5461 * 6 + 2): the cut frame's samples come with the span, the next span
               // starts behind them -- rotated, as in rodio; the source ends 4 samples into a frame
                std::vector<float> d6(6 * 6000);
                for (size_t i = 0; i < d6.size(); ++i) d6[i] = (float)i;
                rh::SamplesBuffer six(6, 48000, d6);
                rh::detail::SpanReader r6(&six);
                std::vector<float> t6(40000);
                expect(r6.read_piece(t6.data(), 100000, pc) && pc.n == 32768 && pc.tail == 2 && pc.closes && !pc.by_none && t6[32767] == 32767.0f, "the span's cut frame comes with it");
                expect(r6.read_piece(t6.data(), 100000, pc) && pc.n == 36000 - 32768 && pc.tail == (36000 - 32768) % 6 && pc.closes && pc.by_none && t6[0] == 32768.0f, "the next span starts behind the cut");
                // what the converters make of it (sample_rate.rs:174-200, channels.rs:57-85): at the mixer's rate the whole frames pass through and the cut
                // frame gives the output channels its samples cover -- 2 samples of a stereo frame here ...
                rh::detail::UniformPlanner same(2, 48000);
                std::vector<rh::detail::UniformPlanner::Seg> segs;
                same.begin_block();
                same.add(rh::detail::Piece{32768, true, true, 6, 48000, 2, false}, segs);
                expect(segs.size() == 2 && segs[0].g.m1 == 5461 && segs[0].g.reserved == 0 && segs[1].g.reserved == 2 && segs[1].g.m1 == 2 && segs[1].g.span_frames == 5461 && segs[1].dst_off == 5461 * 2 &&
                           same.out_samples() == 5461 * 2 + 2,
                       "a cut frame at the mixer's rate: the samples it covers");
                // ... and in front of a rate conversion the output frames that lerp towards the cut frame are cut to its length, the cut frame follows verbatim,
                // and the channel converter regroups: 48 -> 44.1 kHz, 5461 whole frames + 2 samples: one output has its first tap on frame 5460 (2 samples),
                // one lands on the cut frame (2 samples): 4 converter samples = an incomplete group of 6 -> the 2 output channels it covers
                segs.clear();
                rh::detail::UniformPlanner other(2, 44100);
                other.begin_block();
                other.add(rh::detail::Piece{32768, true, true, 6, 48000, 2, false}, segs);
                uint64_t whole = 0, tail = 0;
                expect(rh_uniform_span_frames(5461, 48000, 44100, 0, &whole) == RH_OK && rh_uniform_cut_tail_samples(5461, 2, 48000, 44100, 6, 2, &tail) == RH_OK, "span arithmetic");
                expect(segs.size() == 2 && segs[0].g.m1 == whole && segs[0].g.span_frames == UINT64_MAX && segs[1].g.reserved == 2 && segs[1].g.m1 == tail && segs[1].g.src_frames == 1 && tail == 2 &&
                           other.out_samples() == whole * 2 + tail,
                       "a cut frame in front of a rate conversion");
            }
        }
        {  // NonZero channels / rate (buffer.rs:40: the types cannot hold 0)
            bool threw = false;
            try {
                rh::SamplesBuffer bad(0, 44100, {});
            } catch (const std::invalid_argument &) {
                threw = true;
            }
            expect(threw, "zero channels rejected");
        }
        {  // the threads that pull a block's sources: every item once, jobs back to back, the first exception reaches the caller
            rh::detail::Workers pool(4);
            expect(pool.threads() == 4, "workers: thread count");
            for (int round = 0; round < 200; ++round) {
                const size_t n = (size_t)(round % 37) + (round % 5 == 0 ? 0 : 1);
                std::vector<int> hit(n + 3, 0);
                pool.run(3, n + 3, [&](size_t i) { hit[i] += 1; });
                bool ok = hit[0] == 0 && hit[1] == 0 && hit[2] == 0;
                for (size_t i = 3; i < n + 3; ++i) ok = ok && hit[i] == 1;
                expect(ok, "workers: every item once");
            }
            bool threw = false;
            try {
                pool.run(0, 64, [&](size_t i) {
                    if (i == 17) throw std::runtime_error("item 17");
                });
            } catch (const std::runtime_error &e) {
                threw = std::string(e.what()) == "item 17";
            }
            expect(threw, "workers: the exception of an item reaches the caller");
            std::atomic<int> count{0};
            pool.run(0, 1000, [&](size_t) { count.fetch_add(1); });
            expect(count.load() == 1000, "workers: usable after an exception");
        }
        std::printf("selftest ok\n");
        return 0;
    }
    if (argc < 3) {
        std::fprintf(stderr, "usage: host_mirror_test mixer|chain <dir> ... (see the source)\n");
        return 2;
    }
    try {
        const std::string mode = argv[1], dir = argv[2];
        rh::init(0);
        std::vector<float> out;
        if (mode == "bench" && (argc == 5 || argc == 6)) {
            const int S = std::atoi(argv[2]);
            const size_t frames = (size_t)std::atoll(argv[3]);
            rh::GpuMixer::Options opt;
            opt.filter_kind = 0;
            opt.filter_freq = 200;
            opt.block_frames = block_arg(argv[4]);
            opt.host_threads = argc == 6 ? (unsigned)std::atoi(argv[5]) : 0;
            rh::GpuMixer mixer(48000, opt);
            uint32_t lcg = 12345u;
            for (int i = 0; i < S; ++i) {
                std::vector<float> x(frames * 2);
                for (float &v : x) {
                    lcg = lcg * 1664525u + 1013904223u;
                    v = ((float)(lcg >> 8) / 8388608.0f - 1.0f) / (float)S;
                }
                mixer.add(make_source(2, 44100, std::move(x)));
            }
            std::vector<float> chunk(1u << 16);
            // the control thread starts the stream (plan, page-locked blocks, device rows, the first block) BEFORE the consumer's first
            // read, as a host does before it hands the mixer to the audio callback (RH_BENCH_NO_PREPARE=1: the first read does it)
            const auto tp = std::chrono::steady_clock::now();
            if (!std::getenv("RH_BENCH_NO_PREPARE")) mixer.prepare();
            const auto t0 = std::chrono::steady_clock::now();
            size_t total = mixer.read(chunk.data(), chunk.size());
            const size_t first = total;
            const auto t1 = std::chrono::steady_clock::now();
            // ... and the end is timed apart too: the reads that see the stream finish (the last block; the teardown is a thread's)
            size_t steady = 0;
            auto t2 = t1;
            double slowest_read = 0, last_read = 0;
            for (;;) {
                const auto r0 = std::chrono::steady_clock::now();
                const size_t k = mixer.read(chunk.data(), chunk.size());
                last_read = std::chrono::duration<double>(std::chrono::steady_clock::now() - r0).count();
                slowest_read = std::max(slowest_read, last_read);
                total += k;
                if (k < chunk.size()) break;
                if (mixer.timing().blocks * opt.block_frames < frames) {  // the upstreams still have frames to give: steady state
                    steady = total - first;
                    t2 = std::chrono::steady_clock::now();
                }
            }
            const auto t3 = std::chrono::steady_clock::now();
            const double prep_s = std::chrono::duration<double>(t0 - tp).count();
            const double start_s = std::chrono::duration<double>(t1 - t0).count();
            const double sec = std::chrono::duration<double>(t2 - t1).count();
            const double end_s = std::chrono::duration<double>(t3 - t2).count();
            // end to end: EVERY input byte over the time from the consumer's first read to its last (and with the control thread's
            // prepare() in front of it)
            const double all_bytes = (double)S * (double)frames * 2.0 * sizeof(float);
            const double e2e = all_bytes / std::chrono::duration<double>(t3 - t0).count() / 1e9, e2e_prep = all_bytes / std::chrono::duration<double>(t3 - tp).count() / 1e9;
            const double in_samples = (double)S * (double)frames * 2.0 * (double)steady / (double)total;  // the share of the input behind the steady part
            // the bound of this path is the host link: every input sample crosses it once (PCIe 5.0 x16: 63 GB/s one way)
            const double gbps = in_samples * sizeof(float) / sec / 1e9;
            std::printf("{\"pull_path\": \"GpuMixer\", \"sources\": %d, \"in_frames\": %zu, \"block_frames\": %zu, \"host_threads\": %u, \"out_samples\": %zu, \"start_seconds\": %.4f, \"seconds\": %.4f, \"end_seconds\": %.4f, "
                        "\"Msamples_per_s_in\": %.1f, \"host_link\": {\"achieved\": %.2f, \"peak\": 63.0, \"unit\": \"GB/s\", \"frac\": %.3f}, \"host_seconds\": {\"pull\": %.4f, \"prefetch\": %.4f, \"submit\": %.4f, \"wait\": %.4f, \"blocks\": %llu}, "
                        "\"prepare_seconds\": %.4f, \"first_read_seconds\": %.5f, \"last_read_seconds\": %.5f, \"slowest_read_seconds\": %.5f, "
                        "\"end_to_end\": {\"achieved\": %.2f, \"frac\": %.3f, \"with_prepare\": %.2f, \"frac_with_prepare\": %.3f, \"unit\": \"GB/s\", \"peak\": 63.0}}\n",
                        S, frames, opt.block_frames, opt.host_threads, total, start_s, sec, end_s, in_samples / sec / 1e6, gbps, gbps / 63.0, mixer.timing().pull_s, mixer.timing().prefetch_s, mixer.timing().submit_s,
                        mixer.timing().wait_s, (unsigned long long)mixer.timing().blocks, prep_s, start_s, last_read, slowest_read, e2e, e2e / 63.0, e2e_prep, e2e_prep / 63.0);
            return 0;
        }
        if (mode == "benchwide" && argc == 6) {
            // host_mirror_test benchwide <S> <frames> <block_frames> <mixer_channels>: S continuous 5.1-style sources at 44.1 kHz (host memory) into
            // mixer::mixer(channels, 48 kHz), pulled to the end; RH_TEST_WIDE_CHAINS=1: every source a chain of its own (the form of round 5)
            const int S = std::atoi(argv[2]);
            const size_t frames = (size_t)std::atoll(argv[3]);
            const uint16_t mch = (uint16_t)std::atoi(argv[5]);
            rh::GpuMixer::Options opt;
            opt.block_frames = block_arg(argv[4]);
            opt.wide_chains = std::getenv("RH_TEST_WIDE_CHAINS") != nullptr;
            rh::GpuMixer mixer(mch, 48000, opt);
            uint32_t lcg = 12345u;
            for (int i = 0; i < S; ++i) {
                std::vector<float> x(frames * mch);
                for (float &v : x) {
                    lcg = lcg * 1664525u + 1013904223u;
                    v = ((float)(lcg >> 8) / 8388608.0f - 1.0f) / (float)S;
                }
                mixer.add(make_source(mch, 44100, std::move(x)), 0.5f + 0.01f * (float)i);
            }
            std::vector<float> chunk(1u << 16);
            mixer.prepare();
            const auto t0 = std::chrono::steady_clock::now();
            size_t total = 0;
            double sum = 0;
            for (;;) {
                const size_t k = mixer.read(chunk.data(), chunk.size());
                total += k;
                for (size_t q = 0; q < k; q += 4097) sum += chunk[q];
                if (k < chunk.size()) break;
            }
            const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            const double in_bytes = (double)S * (double)frames * mch * 4.0;
            std::printf("{\"mode\": \"%s\", \"sources\": %d, \"channels\": %u, \"frames\": %zu, \"block_frames\": %zu, \"out_samples\": %zu, \"seconds\": %.4f, \"Msamples_per_s_in\": %.1f, "
                        "\"host_link_GBps\": %.2f, \"host_link_frac_of_63\": %.3f, \"wide_fused_blocks\": %llu, \"blocks\": %llu, \"checksum\": %.9g, "
                        "\"host_seconds\": {\"pull\": %.4f, \"prefetch\": %.4f, \"submit\": %.4f, \"wait\": %.4f}}\n",
                        opt.wide_chains ? "a chain per source" : "one launch a block", S, (unsigned)mch, frames, opt.block_frames, total, sec, in_bytes / 4.0 / sec / 1e6, in_bytes / sec / 1e9,
                        in_bytes / sec / 1e9 / 63.0, (unsigned long long)mixer.wide_fused_blocks(), (unsigned long long)mixer.timing().blocks, sum, mixer.timing().pull_s, mixer.timing().prefetch_s,
                        mixer.timing().submit_s, mixer.timing().wait_s);
            return 0;
        }
        if (mode == "mixany" && argc == 9) {
            const int S = std::atoi(argv[3]);
            const uint32_t to = (uint32_t)std::atoll(argv[4]);
            rh::GpuMixer::Options opt;
            opt.filter_kind = std::atoi(argv[5]);
            opt.filter_freq = (uint32_t)std::atoll(argv[6]);
            opt.block_frames = block_arg(argv[7]);
            opt.frames_per_lane = (uint32_t)std::atoll(argv[8]);
            std::FILE *sf = std::fopen((dir + "/spec.txt").c_str(), "r");
            if (!sf) throw std::runtime_error("spec.txt");
            rh::GpuMixer mixer(to, opt);
            for (int i = 0; i < S; ++i) {
                unsigned ch = 0, rate = 0;
                float gain = 1.0f;
                if (std::fscanf(sf, "%u %u %f", &ch, &rate, &gain) != 3) throw std::runtime_error("spec.txt: short");
                mixer.add(make_source((uint16_t)ch, rate, read_f32(dir + "/src_" + std::to_string(i) + ".f32"), i), gain);
            }
            std::fclose(sf);
            out = drain(mixer);
            std::fprintf(stderr, "pull_threads=%u\n", mixer.pull_threads());
        } else if (mode == "late" && argc == 12) {
            const int S0 = std::atoi(argv[3]), S1 = std::atoi(argv[4]);
            const uint32_t from = (uint32_t)std::atoll(argv[5]), to = (uint32_t)std::atoll(argv[6]);
            rh::GpuMixer::Options opt;
            opt.filter_kind = std::atoi(argv[7]);
            opt.filter_freq = (uint32_t)std::atoll(argv[8]);
            opt.block_frames = block_arg(argv[9]);
            opt.frames_per_lane = (uint32_t)std::atoll(argv[10]);
            const size_t pull_first = (size_t)std::atoll(argv[11]);
            const std::vector<float> gains = read_f32(dir + "/gains.f32");
            rh::GpuMixer mixer(to, opt);
            auto add = [&](int i) { mixer.add(make_source(2, from, read_f32(dir + "/src_" + std::to_string(i) + ".f32")), gains.at((size_t)i)); };
            for (int i = 0; i < S0; ++i) add(i);
            for (size_t k = 0; k < pull_first; ++k) {
                const std::optional<float> v = mixer.next();
                if (!v) break;
                out.push_back(*v);
            }
            for (int i = S0; i < S0 + S1; ++i) add(i);
            // a consumer keeps calling next() after a None (the cpal callback plays silence): an ended mixer that was given a
            // new source answers None until MixerSource's channel position is back at 0 (mixer.rs:120-136)
            int nones = 0;
            std::optional<float> v;
            while (nones < 4 && !(v = mixer.next())) ++nones;
            if (v) out.push_back(*v);
            const std::vector<float> rest = drain(mixer);
            out.insert(out.end(), rest.begin(), rest.end());
            std::FILE *nf = std::fopen((dir + "/nones.txt").c_str(), "w");
            if (nf) {
                std::fprintf(nf, "%d\n", nones);
                std::fclose(nf);
            }
            std::FILE *jf = std::fopen((dir + "/join.txt").c_str(), "w");
            if (jf) {
                std::fprintf(jf, "%llu\n", (unsigned long long)mixer.last_join_frame());
                std::fclose(jf);
            }
        } else if (mode == "latewide" && argc == 9) {
            // host_mirror_test latewide <dir> <S0> <S1> <mixer_channels> <to_rate> <block_frames> <pull_first>: like `late`, any layouts (spec.txt: "channels rate gain"), any mixer
            const int S0 = std::atoi(argv[3]), S1 = std::atoi(argv[4]);
            const uint16_t mch = (uint16_t)std::atoi(argv[5]);
            const uint32_t to = (uint32_t)std::atoll(argv[6]);
            rh::GpuMixer::Options opt;
            opt.block_frames = block_arg(argv[7]);
            const size_t pull_first = (size_t)std::atoll(argv[8]);
            std::FILE *sf = std::fopen((dir + "/spec.txt").c_str(), "r");
            if (!sf) throw std::runtime_error("spec.txt");
            rh::GpuMixer mixer(mch, to, opt);
            auto add = [&](int i) {
                unsigned ch = 0, rate = 0;
                float gain = 1.0f;
                if (std::fscanf(sf, "%u %u %f", &ch, &rate, &gain) != 3) throw std::runtime_error("spec.txt: short");
                mixer.add(make_source((uint16_t)ch, rate, read_f32(dir + "/src_" + std::to_string(i) + ".f32"), i), gain);
            };
            for (int i = 0; i < S0; ++i) add(i);
            for (size_t k = 0; k < pull_first; ++k) {
                const std::optional<float> v = mixer.next();
                if (!v) break;
                out.push_back(*v);
            }
            for (int i = S0; i < S0 + S1; ++i) add(i);
            std::fclose(sf);
            // (as in `late`: an ended mixer that was given a new source answers None until its channel position is back at 0, mixer.rs:120-136)
            int nones = 0;
            std::optional<float> v;
            while (nones < 16 && !(v = mixer.next())) ++nones;
            if (v) out.push_back(*v);
            if (v) {
                const std::vector<float> rest = drain(mixer);
                out.insert(out.end(), rest.begin(), rest.end());
            }
            std::FILE *nf = std::fopen((dir + "/nones.txt").c_str(), "w");
            if (nf) {
                std::fprintf(nf, "%d\n", nones);
                std::fclose(nf);
            }
        } else if (mode == "mixer" && argc == 10) {
            const int S = std::atoi(argv[3]);
            const uint32_t from = (uint32_t)std::atoll(argv[4]), to = (uint32_t)std::atoll(argv[5]);
            rh::GpuMixer::Options opt;
            opt.filter_kind = std::atoi(argv[6]);
            opt.filter_freq = (uint32_t)std::atoll(argv[7]);
            opt.block_frames = block_arg(argv[8]);
            opt.frames_per_lane = (uint32_t)std::atoll(argv[9]);
            const std::vector<float> gains = read_f32(dir + "/gains.f32");
            rh::GpuMixer mixer(to, opt);
            for (int i = 0; i < S; ++i)
                mixer.add(make_source(2, from, read_f32(dir + "/src_" + std::to_string(i) + ".f32")), gains.at((size_t)i));
            if (mixer.channels() != 2 || mixer.sample_rate() != to) throw std::runtime_error("format");
            out = drain(mixer);
        } else if (mode == "latechain" && argc == 10) {
            // host_mirror_test latechain <dir> <S0> <S1> <mixer_channels> <to_rate> <block_frames> <on_device 0|1> <pull_first>: `chainmix`'s sources
            // (spec.txt: "channels rate gain filter_kind filter_freq op,op,..."), the last S1 of them added to the RUNNING mixer as `latewide` does
            const int S0 = std::atoi(argv[3]), S1 = std::atoi(argv[4]);
            const uint16_t mch = (uint16_t)std::atoi(argv[5]);
            const uint32_t to = (uint32_t)std::atoll(argv[6]);
            rh::GpuMixer::Options opt;
            opt.block_frames = block_arg(argv[7]);
            const bool on_device = std::atoi(argv[8]) != 0;
            const size_t pull_first = (size_t)std::atoll(argv[9]);
            std::FILE *sf = std::fopen((dir + "/spec.txt").c_str(), "r");
            if (!sf) throw std::runtime_error("spec.txt");
            rh::GpuMixer mixer(mch, to, opt);
            auto add = [&](int i) {
                unsigned ch = 0, rate = 0, ffreq = 0;
                int fkind = -1;
                float gain = 1.0f;
                char ops[1024];
                if (std::fscanf(sf, "%u %u %f %d %u %1023s", &ch, &rate, &gain, &fkind, &ffreq, ops) != 6) throw std::runtime_error("spec.txt: short");
                rh::BoxSource src = make_source((uint16_t)ch, rate, read_f32(dir + "/src_" + std::to_string(i) + ".f32"), i);
                const rh::GpuMixer::Filter filt{fkind, ffreq, 0.5f};
                if (std::string(ops) == "-") {
                    mixer.add(std::move(src), gain, filt);
                } else {
                    auto g = std::make_unique<rh::GpuSource>(std::move(src), opt.block_frames);
                    for (const std::string &o : split(ops, ',')) apply_op(*g, o);
                    if (on_device) mixer.add(std::move(g), gain, filt);
                    else mixer.add(rh::BoxSource(std::move(g)), gain, filt);
                }
            };
            for (int i = 0; i < S0; ++i) add(i);
            for (size_t k = 0; k < pull_first; ++k) {
                const std::optional<float> v = mixer.next();
                if (!v) break;
                out.push_back(*v);
            }
            for (int i = S0; i < S0 + S1; ++i) add(i);
            std::fclose(sf);
            int nones = 0;
            std::optional<float> v;
            while (nones < 16 && !(v = mixer.next())) ++nones;
            if (v) {
                out.push_back(*v);
                const std::vector<float> rest = drain(mixer);
                out.insert(out.end(), rest.begin(), rest.end());
            }
            std::FILE *nf = std::fopen((dir + "/nones.txt").c_str(), "w");
            if (nf) {
                std::fprintf(nf, "%d\n", nones);
                std::fclose(nf);
            }
        } else if (mode == "chainmix" && argc == 8) {
            // host_mirror_test chainmix <dir> <S> <mixer_channels> <to_rate> <block_frames> <on_device 0|1>
            //   <dir>/spec.txt, one line per source: "channels rate gain filter_kind filter_freq op,op,..." ("-" = no adapters: the source
            //   goes into the mixer as it is).  A source with adapters is a GpuSource chain handed to GpuMixer::add by value; with
            //   on_device = 1 its blocks reach the mixer device-to-device.  Writes out.f32 and stats.txt.
            const int S = std::atoi(argv[3]);
            const uint16_t mch = (uint16_t)std::atoi(argv[4]);
            const uint32_t to = (uint32_t)std::atoll(argv[5]);
            rh::GpuMixer::Options opt;
            opt.block_frames = block_arg(argv[6]);
            opt.wide_chains = std::getenv("RH_TEST_WIDE_CHAINS") != nullptr;  // (a comparison aid: every source of a wide mixer a chain of its own)
            const bool on_device = std::atoi(argv[7]) != 0;
            std::FILE *sf = std::fopen((dir + "/spec.txt").c_str(), "r");
            if (!sf) throw std::runtime_error("spec.txt");
            rh::GpuMixer mixer(mch, to, opt);
            for (int i = 0; i < S; ++i) {
                unsigned ch = 0, rate = 0, ffreq = 0;
                int fkind = -1;
                float gain = 1.0f;
                char ops[1024];
                if (std::fscanf(sf, "%u %u %f %d %u %1023s", &ch, &rate, &gain, &fkind, &ffreq, ops) != 6) throw std::runtime_error("spec.txt: short");
                rh::BoxSource src = make_seq_source(dir, i);
                if (!src) src = make_source((uint16_t)ch, rate, read_f32(dir + "/src_" + std::to_string(i) + ".f32"), i);
                const rh::GpuMixer::Filter filt{fkind, ffreq, 0.5f};
                if (std::string(ops) == "-") {
                    mixer.add(std::move(src), gain, filt);
                } else {
                    auto g = std::make_unique<rh::GpuSource>(std::move(src), opt.block_frames);
                    for (const std::string &o : split(ops, ',')) apply_op(*g, o);
                    if (on_device) mixer.add(std::move(g), gain, filt);
                    else mixer.add(rh::BoxSource(std::move(g)), gain, filt);
                }
            }
            std::fclose(sf);
            if (mixer.channels() != mch || mixer.sample_rate() != to) throw std::runtime_error("format");
            mixer.prepare();  // what a control thread does before it hands the mixer to the audio callback
            const auto t0 = std::chrono::steady_clock::now();
            float one[8];
            const size_t k0 = mixer.read(one, mch);  // the callback's first read: must not start anything
            const double first_read = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            out.assign(one, one + k0);
            const std::vector<float> rest = drain(mixer);
            out.insert(out.end(), rest.begin(), rest.end());
            const rh::GpuMixer::ChainStats cs = mixer.chain_stats();  // (the chains themselves are gone with their generation)
            std::FILE *st = std::fopen((dir + "/stats.txt").c_str(), "w");
            if (st) {
                std::fprintf(st, "{\"chains\": %llu, \"chains_on_device\": %llu, \"chain_d2h_samples\": %llu, \"chain_device_samples\": %llu, \"first_read_seconds\": %.6f, \"prepare_seconds\": %.6f, "
                                 "\"wide_fused_blocks\": %llu}\n",
                             (unsigned long long)cs.chains, (unsigned long long)cs.on_device, (unsigned long long)cs.d2h_samples, (unsigned long long)cs.device_samples, first_read,
                             mixer.timing().first_advance_s, (unsigned long long)mixer.wide_fused_blocks());
                std::fclose(st);
            }
        } else if (mode == "chain" && argc >= 6) {
            const uint16_t ch = (uint16_t)std::atoi(argv[3]);
            const uint32_t rate = (uint32_t)std::atoll(argv[4]);
            rh::BoxSource seq = make_seq_source(dir, 0);
            rh::GpuSource g(seq ? std::move(seq) : make_source(ch, rate, read_f32(dir + "/src_0.f32")), block_arg(argv[5]));
            for (int a = 6; a < argc; ++a) apply_op(g, argv[a]);
            std::FILE *meta = std::fopen((dir + "/format.txt").c_str(), "w");
            if (meta) {
                std::fprintf(meta, "%u %u\n", (unsigned)g.channels(), (unsigned)g.sample_rate());
                std::fclose(meta);
            }
            if (const char *after = std::getenv("RH_TEST_SEEK_AFTER")) {  // pull, try_seek, pull on (test_gpu_source_try_seek)
                const size_t k = (size_t)std::atoll(after);
                for (size_t i = 0; i < k; ++i) {
                    const std::optional<float> v = g.next();
                    if (!v) break;
                    out.push_back(*v);
                }
                const char *ns = std::getenv("RH_TEST_SEEK_NS");
                const bool ok = g.try_seek(rh::Nanos(ns ? std::atoll(ns) : 0));
                std::FILE *sk = std::fopen((dir + "/seek.txt").c_str(), "w");
                if (sk) {
                    std::fprintf(sk, "%d %zu\n", ok ? 1 : 0, out.size());
                    std::fclose(sk);
                }
            }
            if (std::getenv("RH_TEST_TRACK_HINTS")) {  // sample by sample: size_hint() in front of every sample and behind the last (hints.i64: lower, upper or -1), total_duration() (duration.txt: ns or -1)
                std::vector<long long> hints;
                auto note = [&]() {
                    const rh::SizeHint h = g.size_hint();
                    hints.push_back((long long)h.lower);
                    hints.push_back(h.upper ? (long long)*h.upper : -1);
                };
                const std::optional<rh::Nanos> d0 = g.total_duration();
                for (;;) {
                    note();
                    const std::optional<float> v = g.next();
                    if (!v) break;
                    out.push_back(*v);
                }
                const std::optional<rh::Nanos> d1 = g.total_duration();
                std::FILE *hf = std::fopen((dir + "/hints.i64").c_str(), "wb");
                if (hf) {
                    if (!hints.empty()) std::fwrite(hints.data(), sizeof(long long), hints.size(), hf);
                    std::fclose(hf);
                }
                std::FILE *df = std::fopen((dir + "/duration.txt").c_str(), "w");
                if (df) {
                    std::fprintf(df, "%lld %lld\n", d0 ? (long long)d0->count() : -1, d1 ? (long long)d1->count() : -1);
                    std::fclose(df);
                }
            } else if (std::getenv("RH_TEST_TRACK_FORMAT")) {  // sample by sample, noting what the chain reports in front of every sample (test_gpu_source_follows_a_format_change)
                std::FILE *ff = std::fopen((dir + "/formats.txt").c_str(), "w");
                unsigned lc = 0, lr = 0;
                long long ls = -2;
                for (;;) {
                    const unsigned c = g.channels(), r = g.sample_rate();
                    const std::optional<std::size_t> sp = g.current_span_len();
                    const long long spv = sp ? (long long)*sp : -1;
                    if (c != lc || r != lr || spv != ls) {
                        if (ff) std::fprintf(ff, "%zu %u %u %lld\n", out.size(), c, r, spv);
                        lc = c, lr = r, ls = spv;
                    }
                    const std::optional<float> v = g.next();
                    if (!v) break;
                    out.push_back(*v);
                }
                if (ff) std::fclose(ff);
            } else {
                const std::vector<float> rest = drain(g);
                out.insert(out.end(), rest.begin(), rest.end());
            }
        } else {
            std::fprintf(stderr, "bad arguments\n");
            return 2;
        }
        write_f32(dir + "/out.f32", out);
        if (!g_hints.empty()) {
            std::FILE *hf = std::fopen((dir + "/hints.i64").c_str(), "wb");
            if (hf) {
                std::fwrite(g_hints.data(), sizeof(long long), g_hints.size(), hf);
                std::fclose(hf);
            }
        }
        std::printf("%zu samples\n", out.size());
        return 0;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "host_mirror_test: %s\n", e.what());
        return 1;
    }
}
