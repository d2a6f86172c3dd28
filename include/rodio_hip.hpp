// rodio_hip.hpp -- C++17 host-side mirror of rodio's `Source` interface over the C ABI of rodio_hip.h.
//
// rodio is compiled code whose extension point is the trait `Source: Iterator<Item = f32>`
// (/root/reference/src/source/mod.rs:179-218): anything that implements it can be handed to
// `Mixer::add` (src/mixer.rs:58-66), `Player::append` (src/player.rs:104-108) or `queue.append`
// (src/queue.rs:62-70).  This header is that boundary for a C++ host (and the model for the Rust shim of
// INTEGRATION.md): pull-model adapters that own their upstream by value, pre-pull a block, run it
// through librodio_hip.so on the GPU and serve `next()` from page-locked memory, one block ahead.
//
//   GpuSource  one upstream + a chain of adapters built with rodio's method names
//              (amplify, low_pass, high_pass, reverb, channel_volume, limit, automatic_gain_control, ...)
//   GpuMixer   mixer::mixer(2, rate) where every added source goes through
//              UniformSourceIterator(.., 2, rate) [.low_pass(f) / .high_pass(f)] [.amplify(g)] and the ordered sum:
//              the fused kernel, block by block, one filter state per source (rh_rlm_stream_block_v)
//
// Header-only; needs nothing but rodio_hip.h and librodio_hip.so.  There is no CPU compute path: every
// arithmetic operation on samples happens in the library; a missing GPU surfaces as rodio_hip::Error.
// One object is used from one thread at a time (rodio: `Send`, not `Sync`).
#ifndef RODIO_HIP_HPP
#define RODIO_HIP_HPP

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <exception>
#include <memory>
#include <mutex>
#include <numeric>
#include <optional>
#include <stdexcept>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "rodio_hip.h"

namespace rodio_hip {

class Error : public std::runtime_error {
public:
    Error(rh_status st, const std::string &what) : std::runtime_error(what + ": " + rh_status_string(st) + hip_detail(st)), status(st) {}
    rh_status status;

private:
    static std::string hip_detail(rh_status st) { return st == RH_ERR_HIP ? std::string(" (") + rh_last_hip_error() + ")" : std::string(); }
};
inline void check(rh_status st, const char *what) {
    if (st != RH_OK) throw Error(st, what);
}
/// Binds the library to a gfx950 device (once per process).
inline void init(int device = 0) { check(rh_init(device), "rh_init"); }

using Nanos = std::chrono::nanoseconds;
/// `Iterator::size_hint()`: (lower, upper), upper == nullopt for "no upper bound".
struct SizeHint {
    std::size_t lower = 0;
    std::optional<std::size_t> upper = std::nullopt;
    bool operator==(const SizeHint &o) const { return lower == o.lower && upper == o.upper; }
};

// ---------------------------------------------------------------- trait Source (source/mod.rs:179-218) ----
class Source {
public:
    virtual ~Source() = default;
    /// `Iterator::next`: the next interleaved sample, or nullopt at the end of the stream (never an error).
    virtual std::optional<float> next() = 0;
    virtual std::optional<std::size_t> current_span_len() const { return std::nullopt; }
    virtual std::uint16_t channels() const = 0;
    virtual std::uint32_t sample_rate() const = 0;
    virtual std::optional<Nanos> total_duration() const { return std::nullopt; }
    /// `Iterator::size_hint`: the trait's default is (0, None) -- what a source that only implements next() answers (benches/shared.rs:14-21).
    virtual SizeHint size_hint() const { return SizeHint{}; }
    /// `try_seek`: false == SeekError::NotSupported (source/mod.rs:766-787).
    virtual bool try_seek(Nanos) { return false; }
    /// Bulk form of next() that block adapters pull with; the default is the per-sample loop.
    virtual std::size_t read(float *dst, std::size_t n) {
        std::size_t k = 0;
        for (; k < n; ++k) {
            const std::optional<float> v = next();
            if (!v) break;
            dst[k] = *v;
        }
        return k;
    }
};
using BoxSource = std::unique_ptr<Source>;

/// buffer.rs:20-71: a source over samples held in memory.
class SamplesBuffer : public Source {
public:
    SamplesBuffer(std::uint16_t channels, std::uint32_t sample_rate, std::vector<float> data) : ch_(channels), rate_(sample_rate), data_(std::move(data)) {
        if (!channels || !sample_rate) throw std::invalid_argument("channels and sample_rate are NonZero in rodio");
    }
    std::optional<float> next() override { return pos_ < data_.size() ? std::optional<float>(data_[pos_++]) : std::nullopt; }
    std::size_t read(float *dst, std::size_t n) override {
        const std::size_t k = std::min(n, data_.size() - pos_);
        std::memcpy(dst, data_.data() + pos_, k * sizeof(float));
        pos_ += k;
        return k;
    }
    /// buffer.rs:76-82: the whole buffer is one span -- Some(len) until it is exhausted, then Some(0).
    std::optional<std::size_t> current_span_len() const override { return pos_ >= data_.size() ? 0 : data_.size(); }
    std::uint16_t channels() const override { return ch_; }
    std::uint32_t sample_rate() const override { return rate_; }
    std::optional<Nanos> total_duration() const override {  // buffer.rs:45-51
        return Nanos((std::int64_t)(1000000000ull * (std::uint64_t)data_.size() / rate_ / ch_));
    }
    SizeHint size_hint() const override {  // buffer.rs:134-137
        const std::size_t remaining = data_.size() - pos_;
        return SizeHint{remaining, remaining};
    }
    /// buffer.rs:99-121: jump to the sample for `pos`, saturating at the end, keeping the channel the consumer is at.
    bool try_seek(Nanos pos) override {
        const std::size_t curr_channel = pos_ % ch_;
        const std::uint64_t ns = (std::uint64_t)pos.count();
        const float secs = (float)(ns / 1000000000ull) + (float)(ns % 1000000000ull) / 1000000000.0f;  // math.rs:118-122 duration_to_float
        const float fpos = secs * (float)rate_ * (float)ch_;
        std::size_t np = fpos >= 1.8446744e19f ? data_.size() : (std::size_t)fpos;
        np = std::min(np, data_.size());
        np = (np + ch_ - 1) / ch_ * ch_;  // next_multiple_of(channels)
        pos_ = np - curr_channel;
        return true;
    }

private:
    std::uint16_t ch_;
    std::uint32_t rate_;
    std::vector<float> data_;
    std::size_t pos_ = 0;
};

// ---------------------------------------------------------------- device plumbing ----
namespace detail {
class DeviceBuf {
public:
    DeviceBuf() = default;
    explicit DeviceBuf(std::size_t floats) { reset(floats); }
    DeviceBuf(const DeviceBuf &) = delete;
    DeviceBuf &operator=(const DeviceBuf &) = delete;
    ~DeviceBuf() {
        if (p_) (void)rh_free(p_);
    }
    void reset(std::size_t floats) {
        if (floats <= n_) return;
        if (p_) check(rh_free(p_), "rh_free");
        p_ = nullptr;
        check(rh_malloc(&p_, floats * sizeof(float)), "rh_malloc");
        n_ = floats;
    }
    float *get() const { return static_cast<float *>(p_); }
    std::size_t size() const { return n_; }
    void swap(DeviceBuf &o) {
        std::swap(p_, o.p_);
        std::swap(n_, o.n_);
    }

private:
    void *p_ = nullptr;
    std::size_t n_ = 0;
};
/// A few host threads for the per-source pulls of one block.  Sources are independent objects and one thread drives one source at
/// a time (rodio: `Source: Send`); the caller takes items too, and the first exception of an item reaches the caller.
class Workers {
public:
    explicit Workers(unsigned threads) {
        for (unsigned i = 1; i < threads; ++i) th_.emplace_back([this] { loop(); });
    }
    Workers(const Workers &) = delete;
    Workers &operator=(const Workers &) = delete;
    ~Workers() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            quit_ = true;
        }
        wake_.notify_all();
        for (std::thread &t : th_) t.join();
    }
    unsigned threads() const { return (unsigned)th_.size() + 1; }
    /// fn(i) for i in [first, last), in any order, on any of the threads; returns when all of them have returned.
    template <class F>
    void run(std::size_t first, std::size_t last, F &&fn) {
        if (th_.empty() || last - first < 2) {
            for (std::size_t i = first; i < last; ++i) fn(i);
            return;
        }
        std::function<void(std::size_t)> f = std::ref(fn);
        {
            std::lock_guard<std::mutex> lk(mu_);
            job_ = &f;
            next_.store(first, std::memory_order_relaxed);
            last_ = last;
            err_ = nullptr;
            ++gen_;
        }
        wake_.notify_all();
        work(f, last);
        std::unique_lock<std::mutex> lk(mu_);
        job_ = nullptr;  // a thread that wakes up late finds nothing to do
        idle_.wait(lk, [this] { return active_ == 0; });
        if (err_) std::rethrow_exception(err_);
    }

private:
    void work(const std::function<void(std::size_t)> &f, std::size_t last) {
        for (;;) {
            const std::size_t i = next_.fetch_add(1, std::memory_order_relaxed);
            if (i >= last) return;
            try {
                f(i);
            } catch (...) {
                std::lock_guard<std::mutex> lk(mu_);
                if (!err_) err_ = std::current_exception();
                next_.store(last, std::memory_order_relaxed);  // the other items are not started
            }
        }
    }
    void loop() {
        (void)rh_bind_thread();  // (the pulls may run GPU-backed upstreams: the thread works on the device rh_init() bound)
        std::uint64_t seen = 0;
        std::unique_lock<std::mutex> lk(mu_);
        for (;;) {
            wake_.wait(lk, [&] { return quit_ || gen_ != seen; });
            if (quit_) return;
            seen = gen_;
            if (!job_) continue;
            const std::function<void(std::size_t)> *f = job_;
            const std::size_t last = last_;
            ++active_;
            lk.unlock();
            work(*f, last);
            lk.lock();
            if (--active_ == 0) idle_.notify_all();
        }
    }
    std::vector<std::thread> th_;
    std::mutex mu_;
    std::condition_variable wake_, idle_;
    const std::function<void(std::size_t)> *job_ = nullptr;
    std::atomic<std::size_t> next_{0};
    std::size_t last_ = 0;
    std::uint64_t gen_ = 0;
    unsigned active_ = 0;
    bool quit_ = false;
    std::exception_ptr err_;
};
class PinnedBuf {
public:
    PinnedBuf() = default;
    PinnedBuf(const PinnedBuf &) = delete;
    PinnedBuf &operator=(const PinnedBuf &) = delete;
    ~PinnedBuf() {
        if (p_) (void)rh_host_free(p_);
    }
    void reset(std::size_t floats) {
        if (floats <= n_) return;
        if (p_) check(rh_host_free(p_), "rh_host_free");
        p_ = nullptr;
        check(rh_host_alloc(&p_, floats * sizeof(float)), "rh_host_alloc");
        n_ = floats;
    }
    float *get() const { return static_cast<float *>(p_); }
    std::size_t size() const { return n_; }

private:
    void *p_ = nullptr;
    std::size_t n_ = 0;
};
class Event {
public:
    Event() { check(rh_event_create(&e_), "rh_event_create"); }
    Event(const Event &) = delete;
    Event &operator=(const Event &) = delete;
    ~Event() {
        if (e_) (void)rh_event_destroy(e_);
    }
    void *get() const { return e_; }

private:
    void *e_ = nullptr;
};

/// One run of samples pulled from a source inside ONE span of it.
struct Piece {
    std::size_t n;       // samples: whole frames of `ch` channels, and behind them `tail` samples of a frame the span's end cuts
    bool opens, closes;  // the first / the last samples of their span
    std::uint16_t ch;
    std::uint32_t rate;
    std::size_t tail = 0;   // closes only: samples of a CUT last frame (uniform.rs:56: `.min(32768)` cuts frames of 3, 5, 6, 7 channels; so does a source that ends inside a frame)
    bool by_none = false;   // closes only: the span ended because the source returned None, not because its samples were counted out
    std::optional<std::size_t> span_len = std::nullopt;  // opens only: what current_span_len() answered when the span's chain was built
    // The SOURCE's own spans, as an adapter that merely follows them counts them (SpanTracker, span.rs:66-101: current_span_len() samples, then it
    // looks again) -- UniformSourceIterator's chains of min(span, 32768) samples need not end where they do (a SamplesBuffer of 40 000 samples in a
    // queue: the second chain runs on into the next sound, converting it with the first one's parameters; an adapter in FRONT of the
    // iterator sees the new parameters at sample 40 000).  A piece never crosses either boundary.
    std::uint16_t src_ch = 0;     // the format the source reported for these samples
    std::uint32_t src_rate = 0;
    bool src_opens = false;       // the first samples of a span of the source
    std::optional<std::size_t> src_span = std::nullopt;  // src_opens only: current_span_len() at that moment
};

/// Pulls a source the way UniformSourceIterator does (uniform.rs:50-97): whenever its converter chain has run dry it asks
/// `current_span_len()`, `channels()` and `sample_rate()` -- in that order, at exactly that position of the stream -- and
/// admits min(span, 32768) samples (`Take`, uniform.rs:56,148-178) to the chain it builds for them.  A span ends when that
/// many samples were taken or when the source returns None; the stream ends when a fresh chain yields nothing
/// (Some(0), or None at once).  read_piece() never crosses a span boundary.
class SpanReader {
public:
    static constexpr std::size_t kOpenEnded = ~std::size_t(0);
    /// take_32768 = false: the spans as the source reports them, whole -- how an adapter that only FOLLOWS its input's spans meets them
    /// (SpanTracker, span.rs:66-101: counts current_span_len() samples, then looks at the parameters again).
    explicit SpanReader(Source *up = nullptr, bool take_32768 = true) : up_(up), clamp_(take_32768) {}
    bool ended() const { return ended_; }
    /// Format of the span the next read_piece() continues or opens (builds the next chain if none is open); false at the
    /// end of the stream.
    bool peek(std::uint16_t &ch, std::uint32_t &rate) {
        if (!open_ && !bootstrap()) return false;
        ch = ch_;
        rate = rate_;
        return true;
    }
    /// After peek(): the next piece is the first of its span.
    bool opens_next() const { return fresh_; }
    /// What current_span_len() answered when the open span's chain was built.
    std::optional<std::size_t> span_answer() const { return span_; }
    /// Frames the open span still admits (kOpenEnded: current_span_len() was None).
    std::size_t left_frames() const { return left_ == kOpenEnded ? kOpenEnded : left_ / ch_; }
    /// Up to max_frames frames of the current span into dst (dst has room for one more frame: the samples of a frame the span's end
    /// cuts come along with the last whole frames, Piece::tail); false: the stream is over and nothing was produced.
    bool read_piece(float *dst, std::size_t max_frames, Piece &out) {
        if (ended_ || (!open_ && !bootstrap())) return false;
        if (!t_open_) {  // the source's own span opens here
            t_span_ = up_->current_span_len();
            t_ch_ = up_->channels();
            t_rate_ = up_->sample_rate();
            t_left_ = t_span_ && *t_span_ ? *t_span_ : kOpenEnded;  // (Some(0): the source is exhausted -- the read below finds that out)
            t_open_ = t_fresh_ = true;
        }
        std::size_t want = max_frames > kOpenEnded / ch_ ? kOpenEnded : max_frames * ch_;
        want = std::min(want, left_);
        want -= want % ch_;
        if (left_ != kOpenEnded && left_ - want < ch_) want = left_;  // the rest of the span is a cut frame: its samples belong to this span's chain
        // the source's span ends inside the piece: the piece ends with it -- also inside a frame of the format the iterator converts in (the
        // adapters in front of the iterator re-make their state at that very sample: GpuSource splits its runs there; the planner counts the
        // open span in samples, so a piece may stop anywhere).  A source that ENDS there is read on: the None shows below.
        bool by_span = false;
        if (t_left_ != kOpenEnded && t_left_ < want && t_left_ != 0) {
            want = t_left_;
            by_span = true;
        }
        std::size_t got = want ? up_->read(dst, want) : 0;
        bool none = got < want;  // the source returned None inside the span
        if (by_span && !none) {  // the source's span is through: a source that says it is exhausted (Some(0): buffer.rs:76-82) would return None now
            const std::optional<std::size_t> after = up_->current_span_len();
            none = after && *after == 0;
        }
        if (left_ != kOpenEnded) left_ -= got;
        span_got_ += got;
        const bool t_was_fresh = t_fresh_;
        if (got) t_fresh_ = false;
        if (t_left_ != kOpenEnded) {
            t_left_ -= std::min(got, t_left_);
            if (t_left_ == 0) t_open_ = false;
        }
        const bool closes = none || left_ == 0;
        const std::size_t tail = closes ? span_got_ % ch_ : 0;  // the samples of the frame the span's end cuts, over all its pieces (what becomes of them is the planner's business: it knows the target format)
        out = Piece{got, fresh_, closes, ch_, rate_, tail, none, fresh_ ? span_ : std::nullopt, t_ch_, t_rate_, t_was_fresh, t_was_fresh ? t_span_ : std::nullopt};
        const bool produced = got != 0 || (closes && !fresh_);  // a span that had samples before ends here: its last frame is due
        if (got) fresh_ = false;
        if (closes) open_ = false;
        if (none) ended_ = true;  // the chain rodio builds next yields nothing: None
        return produced;
    }
    /// After a seek of the source: what was pulled ahead is gone, the next read builds a fresh chain.
    void restart() {
        open_ = false;
        ended_ = false;
        t_open_ = false;
    }
    /// The format the SOURCE reports for the sample a read would take next, and its answer to current_span_len() there (peek() first).
    void source_format(std::uint16_t &ch, std::uint32_t &rate, std::optional<std::size_t> &span) const {
        if (t_open_) ch = t_ch_, rate = t_rate_, span = t_span_;
        else ch = up_->channels(), rate = up_->sample_rate(), span = up_->current_span_len();
    }

private:
    bool bootstrap() {  // uniform.rs:50-68
        const std::optional<std::size_t> span = up_->current_span_len();
        span_ = span;
        ch_ = up_->channels();
        rate_ = up_->sample_rate();
        if (!ch_ || !rate_) throw std::invalid_argument("channels and sample_rate are NonZero in rodio");
        if (span && *span == 0) {  // Take{n: 0}: the chain is empty, next() is None
            ended_ = true;
            return false;
        }
        left_ = span ? (clamp_ ? std::min<std::size_t>(*span, 32768) : *span) : kOpenEnded;
        // source/mod.rs:196-200 asks for spans of whole frames; `.min(32768)` breaks that for 3, 5, 6, 7 ... channels: the chain rodio
        // builds for such a span ends inside a frame, and the next chain starts there -- every later span has its channels rotated.
        // The reader hands the cut frame's samples over with the span (Piece::tail) and goes on at the sample behind them, as rodio's
        // source does; what the converters make of a cut frame is UniformPlanner::add's business.
        open_ = true;
        fresh_ = true;
        span_got_ = 0;
        return true;
    }
    Source *up_;
    bool clamp_ = true;
    bool open_ = false, fresh_ = true, ended_ = false;
    std::size_t left_ = 0, span_got_ = 0;  // samples the open span still admits / has delivered
    std::uint16_t ch_ = 0;
    std::uint32_t rate_ = 0;
    std::optional<std::size_t> span_;
    bool t_open_ = false, t_fresh_ = true;  // the source's own span (see Piece)
    std::size_t t_left_ = 0;
    std::uint16_t t_ch_ = 0;
    std::uint32_t t_rate_ = 0;
    std::optional<std::size_t> t_span_;
};

/// Turns the pieces of one source into the segments rh_uniform_segments converts (UniformSourceIterator::new(src, to_ch,
/// to_rate), span by span).  The planner only counts: the owner lays the samples out as one row
///     [ the frames held() says the previous block left | the samples of this block's pieces, back to back ]
/// (on the host or on the device), hands every piece to add() in order, and after the last one keeps the frames
/// [keep_offset(), keep_offset() + keep_samples()) of the row for the next block.
/// What comes out is a stream of SAMPLES, not of frames: a span that ends inside a frame (Piece::tail) makes rodio's converters
/// emit a run of samples that need not fill an output frame (rh_uniform_cut_tail_samples), and the next span's output follows
/// directly behind it -- out_samples() counts samples, Seg::dst_off is a sample offset.
class UniformPlanner {
public:
    struct Seg {
        std::size_t src_off;  // samples from the start of the row
        std::size_t dst_off;  // output SAMPLES from the first sample this block produces
        rh_uniform_seg g;     // everything but the two pointers
    };
    UniformPlanner(std::uint16_t to_ch = 2, std::uint32_t to_rate = 48000) : to_ch_(to_ch), to_rate_(to_rate) {}
    /// Start of a block: the row begins with held_samples() samples of the open span (none if no span is open).
    void begin_block() {
        pos_ = held_;
        row_off_ = 0;
        out_ = 0;
        keep_off_ = 0;  // a block that brings nothing for the open span keeps what it held
        keep_n_ = held_;
    }
    std::size_t held_samples() const { return held_; }
    /// The largest number of further input frames of the open (or a fresh) span that cannot produce more than `room`
    /// output frames, and the smallest that produces at least `want` (both before the span's verbatim last frame).
    void budget(std::uint32_t rate, bool fresh, std::uint64_t want, std::uint64_t room, std::uint64_t &need, std::uint64_t &most) const {
        const std::uint64_t in = fresh ? 0 : span_in_, m = fresh ? 0 : span_m_;
        // ready(N) = ceil((N-1)*T/F) for N >= 1 (lerp taps), +1 when the span closes.  first_tap(k) = floor(k*F/T).
        std::uint64_t a = 0, b = 0;
        check(rh_uniform_first_tap(m + want, rate, to_rate_, &a), "rh_uniform_first_tap");              // ready(N) >= m + want  <=  N-1 >= ceil((m+want)*F/T)
        check(rh_uniform_first_tap(room ? m + room - 1 : m, rate, to_rate_, &b), "rh_uniform_first_tap");  // ready(N) + 1 <= m + room  <=  N-1 <= floor((m+room-1)*F/T)
        const std::uint64_t n_need = a + 2, n_most = room ? b + 1 : in;
        need = n_need > in ? n_need - in : 0;
        most = n_most > in ? n_most - in : 0;
    }
    /// Output frames a span's end can add beyond what budget() counts: its verbatim last frame -- or, when the span ends inside a
    /// frame, the samples rodio's converters make of the cut (at most ceil(T/F) + 1 short frames regrouped: sample_rate.rs:174-200).
    static std::uint64_t close_slack_frames(std::uint32_t rate, std::uint32_t to_rate) { return (std::uint64_t)to_rate / rate + 3; }
    void add(const Piece &p, std::vector<Seg> &segs) {
        if (p.opens) {
            span_in_ = span_m_ = span_samples_ = 0;
            row_off_ = pos_;  // the span's frame 0 sits here
            row_frame0_ = 0;
        }
        // (a piece may stop inside a frame where the source's own span does: the open span is counted in samples, the samples of a frame
        // that is not whole yet lie behind the whole frames in the row and are kept with them)
        const std::uint64_t span_total = span_samples_ + p.n;
        const std::size_t tail = p.closes ? (std::size_t)(span_total % p.ch) : 0;  // the samples of the frame the span's end cuts (they may have come with earlier pieces)
        span_samples_ = span_total - tail;
        span_in_ = span_samples_ / p.ch;
        const std::size_t partial = (std::size_t)(span_samples_ % p.ch);
        pos_ += p.n;
        const bool cut = tail != 0;
        // A span that ends inside a frame (uniform.rs:56: `.min(32768)` on 3, 5, 6, 7 channels; a source that returns None inside a frame).
        // rodio's SampleRateConverter meets a SHORT frame: every output frame that lerps towards it is cut to its length (zip,
        // sample_rate.rs:174-179), the short frame itself comes out verbatim when an output lands on it (:193-200), and the
        // ChannelCountConverter behind regroups those runs into frames of `from` samples (channels.rs:57-85) -- reproduced sample for
        // sample by a segment of its own (rh_uniform_seg::reserved = the cut frame's samples); the whole frames in front of it convert as
        // the frames of a span that is still open (no verbatim last frame: that role went to the cut frame).  The next span starts at the
        // sample behind the cut, its channels rotated, as in rodio.
        std::uint64_t ready = 0;
        check(rh_uniform_span_frames(span_in_, p.rate, to_rate_, (p.closes && !cut) ? 1 : 0, &ready), "rh_uniform_span_frames");
        if (ready > span_m_) {
            Seg sg;
            sg.src_off = row_off_;
            sg.dst_off = out_;
            std::memset(&sg.g, 0, sizeof sg.g);
            sg.g.src_frame0 = row_frame0_;
            sg.g.src_frames = span_in_ - row_frame0_;
            sg.g.m0 = span_m_;
            sg.g.m1 = ready;
            sg.g.span_frames = (p.closes && !cut) ? span_in_ : UINT64_MAX;
            sg.g.from_rate = p.rate;
            sg.g.to_rate = to_rate_;
            sg.g.from_ch = p.ch;
            sg.g.to_ch = to_ch_;
            sg.g.gain = 1.0f;
            segs.push_back(sg);
            out_ += (std::size_t)(ready - span_m_) * to_ch_;
            span_m_ = ready;
        }
        if (cut) {
            std::uint64_t tail_out = 0;
            check(rh_uniform_cut_tail_samples(span_in_, (std::uint32_t)tail, p.rate, to_rate_, p.ch, to_ch_, &tail_out), "rh_uniform_cut_tail_samples");
            if (tail_out) {
                // the frame in front of the cut is in the row whenever an output lerps towards the cut frame: that output's first tap is
                // this very frame, and the frames from the next output's first tap on are what a block keeps
                const bool have_last = span_in_ >= 1 && span_in_ - 1 >= row_frame0_;
                const std::uint64_t f0 = have_last ? span_in_ - 1 : span_in_;
                Seg sg;
                sg.src_off = row_off_ + (std::size_t)(f0 - row_frame0_) * p.ch;
                sg.dst_off = out_;
                std::memset(&sg.g, 0, sizeof sg.g);
                sg.g.src_frame0 = f0;
                sg.g.src_frames = span_in_ - f0;
                sg.g.m0 = 0;
                sg.g.m1 = tail_out;  // output SAMPLES of the tail
                sg.g.span_frames = span_in_;
                sg.g.from_rate = p.rate;
                sg.g.to_rate = to_rate_;
                sg.g.from_ch = p.ch;
                sg.g.to_ch = to_ch_;
                sg.g.gain = 1.0f;
                sg.g.reserved = (std::uint32_t)tail;
                segs.push_back(sg);
                out_ += (std::size_t)tail_out;
            }
        }
        if (p.closes) {
            held_ = 0;
            keep_off_ = keep_n_ = 0;
        } else {  // the frames the span's next output frame reads first stay
            std::uint64_t first = 0;
            check(rh_uniform_first_tap(span_m_, p.rate, to_rate_, &first), "rh_uniform_first_tap");
            first = std::max(first, row_frame0_);
            first = std::min(first, span_in_);
            keep_off_ = row_off_ + (std::size_t)(first - row_frame0_) * p.ch;
            keep_n_ = (std::size_t)(span_in_ - first) * p.ch + partial;
            held_ = keep_n_;
            next_frame0_ = first;
        }
    }
    /// End of a block: the next row starts with the kept frames.
    void end_block() {
        row_frame0_ = next_frame0_;
        if (!held_) row_frame0_ = span_in_;
    }
    std::size_t keep_offset() const { return keep_off_; }
    std::size_t keep_samples() const { return keep_n_; }
    std::size_t out_samples() const { return out_; }
    std::uint16_t to_channels() const { return to_ch_; }

private:
    std::uint16_t to_ch_;
    std::uint32_t to_rate_;
    std::uint64_t span_in_ = 0, span_m_ = 0;        // input frames received / output frames planned of the open span
    std::uint64_t span_samples_ = 0;                // ... and its samples (whole frames + the frame that is not whole yet)
    std::uint64_t row_frame0_ = 0, next_frame0_ = 0;  // span frame index of the first frame the row holds of the open span
    std::size_t row_off_ = 0, pos_ = 0, held_ = 0, keep_off_ = 0, keep_n_ = 0;
    std::size_t out_ = 0;                            // output samples planned in this block
};

/// What a source answered to size_hint() at the positions it was asked (the adapters read a block ahead of the consumer; rodio's adapters ask
/// their input where THEY stand).  at(q): the answer at sample q of the source -- the last answer recorded at or before q, less the samples taken
/// since (exact for a source that counts its samples down, buffer.rs:134-137, and for one that answers the trait's default (0, None); a valid
/// bound for any other).
class HintLog {
public:
    void note(std::uint64_t pos, const SizeHint &h) {
        if (!log_.empty() && log_.back().first == pos) log_.back().second = h;
        else log_.emplace_back(pos, h);
        if (log_.size() > 256) log_.pop_front();  // (nobody has asked for a long while: a question that old is answered from the oldest answer kept)
    }
    bool empty() const { return log_.empty(); }
    void clear() { log_.clear(); }
    SizeHint at(std::uint64_t q) const {
        std::size_t i = 0;
        while (i + 1 < log_.size() && log_[i + 1].first <= q) ++i;
        if (i) log_.erase(log_.begin(), log_.begin() + (std::ptrdiff_t)i);  // (the questions only move forward)
        const std::uint64_t since = q > log_.front().first ? q - log_.front().first : 0;
        SizeHint h = log_.front().second;
        h.lower = h.lower > since ? h.lower - (std::size_t)since : 0;
        if (h.upper) h.upper = *h.upper > since ? *h.upper - (std::size_t)since : 0;
        return h;
    }

private:
    mutable std::deque<std::pair<std::uint64_t, SizeHint>> log_;
};

/// UniformSourceIterator::size_hint() (uniform.rs:100-108): the lower bound of the converter chain that is open -- ChannelCountConverter
/// (channels.rs:88-102) over SampleRateConverter (sample_rate.rs:204-238) over Take (uniform.rs:181-196) -- and no upper bound.  Those bounds are
/// functions of the converters' COUNTERS (position in the chunk, samples waiting in the output buffer, the length of the frame read ahead,
/// what Take still admits); the samples themselves are converted on the device, so this class runs the three iterators' state machines over
/// counts only, fed with the spans the samples were pulled in (Piece), lazily: nothing happens until somebody asks.
class UniformCounter {
public:
    /// bare_converter: a SampleRateConverter on its own (no Take in front, no ChannelCountConverter behind): both of ITS bounds.
    UniformCounter(std::uint16_t to_ch = 2, std::uint32_t to_rate = 48000, bool bare_converter = false) : to_ch_(to_ch), to_rate_(to_rate), bare_(bare_converter) {}
    /// A piece of the input, in pull order; `up_pos`: samples taken from the iterator's input in front of it.
    void feed(const Piece &p, std::uint64_t up_pos) {
        if (p.opens || spans_.empty() || spans_.back().closed) {
            compact();
            Span sp;
            sp.ch = p.ch, sp.rate = p.rate, sp.up_pos = up_pos;
            sp.limited = p.span_len.has_value();
            sp.take_n = sp.limited ? std::min<std::size_t>(*p.span_len, 32768) : 0;  // uniform.rs:56
            spans_.push_back(sp);
        }
        spans_.back().got += p.n;
        if (p.closes) spans_.back().closed = true;
    }
    void input_ended() {
        ended_ = true;
        if (!spans_.empty()) spans_.back().closed = true;
    }
    /// size_hint() once the iterator has returned `e` samples; nullopt: it returned None before that (its stream is shorter than e).
    /// `in_hint(q)`: the input's size_hint() at sample q of the input.
    std::optional<SizeHint> hint_at(std::uint64_t e, const std::function<SizeHint(std::uint64_t)> &in_hint) {
        while (done_ < e)
            if (!step()) return std::nullopt;
        if (bare_ && !started_ && !spans_.empty()) (void)bootstrap();  // SampleRateConverter::new reads its first two frames (sample_rate.rs:58-71)
        if (!started_) return SizeHint{in_hint(taken_total()).lower, std::nullopt};  // uniform.rs:105: no chain has been built yet -- the pending input itself
        if (!open_) return SizeHint{0, std::nullopt};  // (every chain has run dry and no span was pulled for another one)
        // Take (uniform.rs:181-196)
        SizeHint h = in_hint(taken_total());
        if (cur().limited) {
            h.lower = std::min(h.lower, left_);
            h.upper = h.upper && *h.upper < left_ ? *h.upper : left_;
        }
        // SampleRateConverter (sample_rate.rs:204-238; usize / u32 arithmetic as written there)
        auto apply = [&](std::size_t samples) {
            std::size_t after = samples;
            if (pos_in_chunk_ == from_ - 1) after += next_len_;
            const std::uint32_t a = pos_in_chunk_ + 2;
            const std::size_t unread = (std::size_t)(from_ > a ? from_ - a : 0) * cur().ch;
            after = after > unread ? after - unread : 0;
            after = after * (std::size_t)to_ / (std::size_t)from_;
            return (std::size_t)(to_ - out_pos_) * cur().ch + after + buf_len_;
        };
        if (from_ != to_) {
            h.lower = apply(h.lower);
            if (h.upper) h.upper = apply(*h.upper);
        }
        if (bare_) return h;
        // ChannelCountConverter (channels.rs:88-102); UniformSourceIterator keeps the lower bound (uniform.rs:100-108)
        const std::size_t consumed = std::min<std::size_t>(cur().ch, ccc_pos_);
        const std::size_t x = (h.lower + consumed) / cur().ch * to_ch_;
        return SizeHint{x > ccc_pos_ ? x - ccc_pos_ : 0, std::nullopt};
    }

private:
    struct Span {
        std::uint16_t ch = 0;
        std::uint32_t rate = 0;
        std::uint64_t up_pos = 0;
        bool limited = false, closed = false;
        std::size_t take_n = 0;
        std::uint64_t got = 0;   // samples the span's chain took from the input
        std::uint64_t reps = 1;  // ... that many spans like this one, back to back (a SamplesBuffer comes in spans of 32768: uniform.rs:56)
    };
    void compact() {  // the spans nobody has asked about yet, run-length encoded
        if (spans_.size() < 2) return;
        Span &a = spans_[spans_.size() - 2];
        const Span &b = spans_.back();
        if (a.closed && b.closed && a.ch == b.ch && a.rate == b.rate && a.limited == b.limited && a.take_n == b.take_n && a.got == b.got && b.reps == 1) {
            a.reps += 1;
            spans_.pop_back();
        }
    }
    const Span &cur() const { return spans_.front(); }
    std::uint64_t taken_total() const { return open_ ? cur().up_pos + used_ : (spans_.empty() ? 0 : spans_.front().up_pos); }
    // Take::next over the counts (uniform.rs:160-177)
    bool take_next() {
        if (cur().limited) {
            if (left_ == 0) return false;
            left_ -= 1;
        }
        if (used_ < cur().got) {
            used_ += 1;
            return true;
        }
        if (cur().closed) return false;  // the input returned None here
        used_ += 1;  // (asked beyond what has been pulled: the source has not said None, the sample is taken to exist)
        return true;
    }
    std::size_t take_frame() {  // sample_rate.rs:58-71,113-121: up to `channels` samples
        std::size_t k = 0;
        while (k < cur().ch && take_next()) ++k;
        return k;
    }
    void next_input_span() {  // sample_rate.rs:110-122
        pos_in_chunk_ += 1;
        cur_len_ = next_len_;
        next_len_ = take_frame();
    }
    bool src_next() {  // sample_rate.rs:131-201
        if (from_ == to_) return take_next();
        if (buf_len_) {
            buf_len_ -= 1;
            return true;
        }
        if (out_pos_ == to_) {
            out_pos_ = 0;
            next_input_span();
            while (pos_in_chunk_ != from_) next_input_span();
            pos_in_chunk_ = 0;
        } else {
            const std::uint32_t req = (from_ * out_pos_ / to_) % from_;
            while (pos_in_chunk_ != req) next_input_span();
        }
        const std::size_t n = std::min(cur_len_, next_len_);  // zip
        out_pos_ += 1;
        if (n) {
            buf_len_ = n - 1;
            return true;
        }
        if (!cur_len_) return false;  // :193-200 draining `current_span`
        buf_len_ = cur_len_ - 1;
        cur_len_ = 0;
        return true;
    }
    bool ccc_next() {  // channels.rs:57-85
        const std::uint16_t from = cur().ch, to = to_ch_;
        bool some;
        if (ccc_pos_ == 0) {
            some = src_next();
            have_repeat_ = some;
        } else if (ccc_pos_ < from) {
            some = src_next();
        } else if (ccc_pos_ == 1) {
            some = have_repeat_;
        } else {
            some = true;
        }
        if (some) ccc_pos_ += 1;
        if (ccc_pos_ == to) {
            ccc_pos_ = 0;
            for (std::uint16_t k = to; k < from; ++k) (void)src_next();
        }
        return some;
    }
    bool bootstrap() {  // uniform.rs:50-68,82-92: the next span's chain
        if (open_) {
            if (spans_.front().reps > 1) {
                spans_.front().reps -= 1;
                spans_.front().up_pos += spans_.front().got;
            } else {
                spans_.pop_front();
            }
        }
        started_ = true;
        if (spans_.empty()) {
            open_ = false;
            return false;  // nothing was pulled for another chain: the one rodio builds here is empty
        }
        open_ = true;
        used_ = 0;
        left_ = cur().take_n;
        std::uint64_t a = cur().rate, b = to_rate_;
        while (b) {
            const std::uint64_t t = a % b;
            a = b;
            b = t;
        }
        from_ = (std::uint32_t)(cur().rate / a);
        to_ = (std::uint32_t)(to_rate_ / a);
        pos_in_chunk_ = out_pos_ = 0;
        buf_len_ = 0;
        ccc_pos_ = 0;
        have_repeat_ = false;
        cur_len_ = next_len_ = 0;
        if (from_ != to_) {
            cur_len_ = take_frame();
            next_len_ = take_frame();
        }
        return true;
    }
    bool step() {  // UniformSourceIterator::next (uniform.rs:76-97)
        if (bare_) {  // the converter alone: one chain, built when the adapter is
            if (!started_ && !bootstrap()) return false;
            if (!open_ || !src_next()) return false;
            done_ += 1;
            return true;
        }
        if (open_ && ccc_next()) {
            done_ += 1;
            return true;
        }
        if (!bootstrap()) return false;
        if (!ccc_next()) return false;
        done_ += 1;
        return true;
    }
    std::uint16_t to_ch_;
    std::uint32_t to_rate_;
    bool bare_ = false;
    std::deque<Span> spans_;  // front(): the span of the open chain
    bool open_ = false, started_ = false, ended_ = false;
    std::uint64_t done_ = 0;   // samples the iterator has returned in the simulation
    std::uint64_t used_ = 0;   // samples the open chain has taken from the input
    std::size_t left_ = 0;     // Take::n
    std::uint32_t from_ = 1, to_ = 1, pos_in_chunk_ = 0, out_pos_ = 0;
    std::size_t cur_len_ = 0, next_len_ = 0, buf_len_ = 0;
    std::uint16_t ccc_pos_ = 0;
    bool have_repeat_ = false;
};

/// What every GPU-backed source shares: two page-locked result blocks, one served while the other is in
/// flight.  A subclass implements enqueue(): pull upstream, copy in, launch, copy out -- all asynchronous on
/// `stream_`; the sizes are known on the host when it returns.
class BlockPump : public Source {
public:
    BlockPump() {
        check(rh_stream_create(&stream_), "rh_stream_create");
        for (Slot &s : slot_) {
            check(rh_event_create(&s.done), "rh_event_create");
            check(rh_event_create(&s.taken), "rh_event_create");
        }
    }
    ~BlockPump() override {
        (void)rh_stream_synchronize(stream_);
        for (Slot &s : slot_) {
            if (s.taken_pending) (void)rh_event_synchronize(s.taken);  // a consumer's copy may still read the slot's device block
            if (s.done) (void)rh_event_destroy(s.done);
            if (s.taken) (void)rh_event_destroy(s.taken);
        }
        (void)rh_stream_destroy(stream_);
    }
    std::optional<float> next() override {
        if (device_out_) throw std::logic_error("this source keeps its blocks on the device: read_device()");
        while (pos_ == cur().n) {
            if (!advance()) return std::nullopt;
        }
        ++handed_out_;
        return cur().out.get()[pos_++];
    }
    /// Everything a first next() would do before it can serve a sample -- start the stream (plans, page-locked blocks, device rows),
    /// pull and process the first block, put the second one in flight -- done NOW, on the calling thread.  rodio builds its sources
    /// on a control thread and hands them to the audio callback (src/stream.rs:538-545), where next() is expected not to block: call
    /// this before the hand-over and the callback's first next() finds its block waiting.
    void prepare() {
        if (!primed_ && !ended_) (void)advance();
    }
    /// Device-resident hand-off (GpuMixer::add(std::unique_ptr<GpuSource>, ..)): the blocks of this source stay in device memory,
    /// and a consumer takes them with read_device() -- device-to-device copies ordered by events: no host copy of the samples, no
    /// host wait.  Switched on before the first block; next() / read() are then not available.
    void keep_blocks_on_device(bool on = true) {
        if (primed_) throw std::logic_error("keep_blocks_on_device() after the stream has started");
        device_out_ = on;
    }
    bool blocks_on_device() const { return device_out_; }
    bool started() const { return primed_; }
    /// Up to n samples of the stream into DEVICE memory `ddst`, enqueued on `consumer` (a stream of the caller's).  Returns the
    /// count: less than n at the end of the stream.
    std::size_t read_device(float *ddst, std::size_t n, rh_stream consumer) {
        if (!device_out_) throw std::logic_error("read_device() needs keep_blocks_on_device()");
        std::size_t k = 0;
        while (k < n) {
            if (pos_ == cur().n && !advance()) break;
            Slot &s = cur();
            const std::size_t take = std::min(n - k, s.n - pos_);
            check(rh_stream_wait_event(consumer, s.done), "rh_stream_wait_event");  // the block is complete when its event fires: the consumer's STREAM waits for it
            check(rh_memcpy_d2d(ddst + k, s.dev.get() + pos_, take * sizeof(float), consumer), "rh_memcpy_d2d");
            check(rh_event_record(s.taken, consumer), "rh_event_record");            // ... and the producer rewrites the slot's block only behind this copy
            s.taken_pending = true;
            pos_ += take;
            k += take;
        }
        handed_out_ += k;
        timing_.device_samples += k;
        return k;
    }
    std::size_t read(float *dst, std::size_t n) override {
        if (device_out_) throw std::logic_error("this source keeps its blocks on the device: read_device()");
        std::size_t k = 0;
        while (k < n) {
            if (pos_ == cur().n && !advance()) break;
            const std::size_t take = std::min(n - k, cur().n - pos_);
            std::memcpy(dst + k, cur().out.get() + pos_, take * sizeof(float));
            pos_ += take;
            k += take;
        }
        handed_out_ += k;
        return k;
    }

    /// Where the host's time went: submitting blocks, preparing the next one ahead of the wait, pulling the upstreams (`pull_s`: a
    /// part of the other two), waiting for the device.
    struct Timing {
        double submit_s = 0, prefetch_s = 0, pull_s = 0, wait_s = 0;
        std::uint64_t blocks = 0;
        std::uint64_t d2h_samples = 0;     // samples of processed blocks copied to the host (0 for a source that keeps its blocks on the device)
        std::uint64_t device_samples = 0;  // samples handed to a consumer device-to-device (read_device)
        double first_advance_s = 0;        // the advance that started the stream (prepare(), or the first next())
    };
    const Timing &timing() const { return timing_; }

protected:
    Timing timing_;
    /// The format of a block's samples from offset `off` on (a source's channels(), sample_rate() and current_span_len() hold for the sample
    /// next() returns next: source/mod.rs:196-207).
    struct FormatMark {
        std::size_t off = 0;
        std::uint16_t ch = 0;
        std::uint32_t rate = 0;
        std::optional<std::size_t> span = std::nullopt;
    };
    struct Slot {
        std::vector<FormatMark> marks;  // format changes inside the block (empty: the source's format is constant)
        FormatMark next;                // ... and the format of the sample behind the block's last one
        PinnedBuf in, out;  // staging of the pulled samples / the processed block
        std::size_t n = 0;  // samples in `out`
        bool last = false;  // upstream ended with this block
        void *done = nullptr;
        DeviceBuf dev;               // keep_blocks_on_device(): the processed block, on the device
        void *taken = nullptr;       // ... recorded on the consumer's stream behind its copies out of `dev`
        bool taken_pending = false;
    };
    /// Fills `s` (n, last) and enqueues everything that produces s.out on stream_.
    virtual void enqueue(Slot &s) = 0;
    /// The host-only part of the next enqueue() (pulling the upstreams into a staging block), called while the device still works
    /// on the block about to be served: what it prepares, the next enqueue() finds done.  Optional.
    virtual void prefetch() {}
    /// A source that returned None may yield samples again (rodio's MixerSource after a later Mixer::add).
    virtual bool can_resume() const { return false; }
    /// Called when a block's work has completed, before it is served: a place to surface device-side failures.
    virtual void block_done() {}
    rh_stream stream_ = nullptr;
    bool device_out_ = false;  // keep_blocks_on_device()

    // What a subclass needs to patch blocks that are already scheduled (GpuMixer: a source that joins a running mixer at
    // the next frame, mixer.rs:175-183): the block being served, the one in flight behind it, the read position.
    Slot &cur() { return slot_[cur_]; }
    const Slot &cur_slot() const { return slot_[cur_]; }
    Slot &other() { return slot_[cur_ ^ 1]; }
    int slot_index(const Slot &s) const { return &s == &slot_[0] ? 0 : 1; }
    int cur_index() const { return cur_; }
    bool running() const { return primed_ && !ended_; }
    bool other_in_flight() const { return primed_ && !ended_ && !slot_[cur_].last; }
    std::size_t position() const { return pos_; }
    std::uint64_t handed_out() const { return handed_out_; }  // samples served so far
    void submit(Slot &s) {
        const auto t0 = std::chrono::steady_clock::now();
        enqueue(s);
        check(rh_event_record(s.done, stream_), "rh_event_record");
        timing_.submit_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        ++timing_.blocks;
    }
    /// Forget everything that was pulled and processed ahead (after a seek of the upstream): the next next() starts over.
    /// `keep_phase` = the source's channel count: the stream that follows resumes at the channel the consumer is at (the first
    /// handed_out % channels samples of the new position are skipped), as rodio's seekable sources do (buffer.rs:110-120) --
    /// a consumer in the middle of a frame must not see left and right swap.
    void restart(std::size_t keep_phase = 0) {
        skip_ = keep_phase ? (std::size_t)(handed_out_ % keep_phase) : 0;
        check(rh_stream_synchronize(stream_), "rh_stream_synchronize");
        for (Slot &s : slot_) {
            s.n = 0;
            s.last = false;
        }
        cur_ = 0;
        pos_ = 0;
        primed_ = ended_ = false;
    }

private:
    bool advance() {
        if (ended_) {
            if (!can_resume()) return false;
            ended_ = primed_ = false;
        }
        const bool starting = !primed_;
        const auto ta = std::chrono::steady_clock::now();
        if (!primed_) {
            submit(slot_[0]);
            primed_ = true;
            cur_ = 0;
        } else {
            if (cur().last) {
                ended_ = true;
                return false;
            }
            cur_ ^= 1;  // the block that was enqueued while the previous one was being served
            if (!cur().last) {
                const auto t0 = std::chrono::steady_clock::now();
                prefetch();
                timing_.prefetch_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            }
        }
        if (!device_out_) {  // (a device-resident consumer orders its copies behind the block's event on ITS stream: the host does not wait)
            const auto t0 = std::chrono::steady_clock::now();
            check(rh_event_synchronize(cur().done), "rh_event_synchronize");
            timing_.wait_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            block_done();
        }
        pos_ = std::min(skip_, cur().n);
        skip_ = 0;
        if (!cur().last) submit(slot_[cur_ ^ 1]);  // prefetch: pull and process one block ahead
        if (starting) timing_.first_advance_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - ta).count();
        return true;
    }
    Slot slot_[2];
    int cur_ = 0;
    std::size_t pos_ = 0, skip_ = 0;
    std::uint64_t handed_out_ = 0;
    bool primed_ = false, ended_ = false;
};
}  // namespace detail

// ---------------------------------------------------------------- GpuSource: adapter chain on one upstream ----
/// `upstream.amplify(..).low_pass(..)...` with the chain executed block-wise on the GPU.  Adapters with
/// memory (filters, limiter, AGC, reverb, converters) carry it across blocks: any block size gives the bits
/// of one pass.  An upstream that changes its format between spans (a queue of sounds) is followed span by span by the adapters
/// that rodio's own SpanTracker serves -- filters, limiter, AGC, `uniform` -- and by everything that does not look at the format; the
/// others (reverb, delay, take_duration, channel_volume, ramps, dither) refuse such a change loudly: `.uniform()` in front of them
/// makes the format one.
class GpuSource : public detail::BlockPump {
public:
    explicit GpuSource(BoxSource upstream, std::size_t block_frames = 1u << 15) : up_(std::move(upstream)), block_frames_(block_frames ? block_frames : 1) {
        if (!up_) throw std::invalid_argument("upstream");
        ch_ = up_->channels();
        rate_ = up_->sample_rate();
        cur_in_ch_ = in_ch0_ = ch_;
        cur_in_rate_ = in_rate0_ = rate_;
        reader_ = detail::SpanReader(up_.get());
    }
    ~GpuSource() override { (void)rh_stream_synchronize(stream_); }  // nothing of the chain may still run when its buffers go
    // -- Source.  The format is the one of the sample next() returns next: an upstream that changes its format between spans (a queue of
    // sounds of different formats) is followed span by span -- every adapter does at the boundary what rodio's does (BltFilter: new
    // coefficients, blt.rs:119-141; Limit: a new channel count resets its state, limit.rs:652-695; AutomaticGainControl: new coefficients
    // and a fresh window, agc.rs:524-548) -- and the chain reports the new format from that sample on.
    std::uint16_t channels() const override { return format_at_cursor().ch; }
    std::uint32_t sample_rate() const override { return format_at_cursor().rate; }
    /// What rodio's adapters answer: the input's span length behind adapters that hand on one sample per sample (amplify.rs:78-80,
    /// blt.rs:153-155, ...), None behind Mix and the converters (mix.rs:92-94, uniform.rs:104-106), and the adapters' own arithmetic
    /// behind take_duration / delay / channel_volume (span_behind()).
    std::optional<std::size_t> current_span_len() const override {
        int rule = 0;  // of the last adapter that does anything to the spans
        for (const Stage &st : stages_)
            if (st.span_rule) rule = st.span_rule;
        if (rule == 0) return format_at_cursor().span;
        // Rule 1: None whatever comes in (Mix, the converters).  Rule 2: rodio's adapter answers with span arithmetic of its own (span_behind()).
        if (rule == 1) return std::nullopt;
        return span_behind(stages_.size(), handed_out());
    }
    /// What rodio's adapters answer, adapter by adapter: the input's duration behind the ones that keep it (amplify.rs:95-97, blt.rs:171-173,
    /// limit.rs:592-594, agc.rs:588-590, channel_volume.rs:119-121, ...), plus the delay behind `delay` (delay.rs:111-115), the shorter of the
    /// two behind `take_duration` (take.rs:209-219), the longer behind `reverb` (mix.rs:104-112 over delay.rs:111-115), and -- behind `reverb`
    /// and `uniform` -- the value the input gave when the adapter was BUILT (buffered.rs:16, uniform.rs:37).
    std::optional<Nanos> total_duration() const override {
        std::optional<Nanos> d = up_->total_duration();
        for (const Stage &st : stages_)
            if (st.dur_fn) d = st.dur_fn(d);
        return d;
    }
    /// `Iterator::size_hint()` as rodio's adapter chain would answer it where the CONSUMER stands (the chain itself reads a block ahead):
    /// every adapter's own arithmetic (delay.rs:78-84, take.rs:151-171, mix.rs:56-67, channels.rs:88-102, sample_rate.rs:204-238,
    /// uniform.rs:100-108; the input's answer behind the adapters that hand it on: amplify.rs:68-70, blt.rs:144-146, ...) over the upstream's
    /// answer at the sample the question reaches (detail::HintLog).
    SizeHint size_hint() const override { return size_hint_at(handed_out()); }
    /// ... once `emitted` samples of the chain's stream have been served (a consumer that takes the blocks ahead of ITS consumer: GpuMixer).
    SizeHint size_hint_at(std::uint64_t emitted) const {
        if (hint_->log.empty()) hint_->before = up_->size_hint();  // nothing pulled (since the last seek): the upstream stands where the question reaches it
        return hint_->at(emitted);
    }
    /// The last adapter of the chain is a `uniform` (what Mixer::add would wrap the chain in is already there).
    bool ends_with_uniform() const { return last_kind_ == 1; }
    Source &inner() { return *up_; }
    BoxSource into_inner() { return std::move(up_); }
    /// The chain's stream can end inside a frame although its upstream keeps `Source`'s contract: reverb and delay count their
    /// silence in SAMPLES (delay.rs:14: every second stereo reverb ends inside a frame), and `uniform` over spans that cut frames hands
    /// on what rodio's converters make of the cut.  A consumer that works on whole frames (GpuMixer's fused streams) asks.
    bool may_end_inside_a_frame() const { return may_cut_; }
    /// The stream has ended where rodio's ChannelVolume would return one more frame -- made of a stale sum -- to a consumer that asks again after
    /// its None (channel_volume.rs:71-88: its input ended inside a frame).  A consumer that stops at None has everything; one that would ask
    /// again (UniformSourceIterator does, once: a mixer) refuses the chain.
    bool ended_with_a_stale_frame() const { return stale_frame_; }
    /// The chain's current_span_len() comes from an adapter's own arithmetic (take_duration, delay, channel_volume: span_behind()): a consumer
    /// that asks -- a UniformSourceIterator, a mixer -- converts it in the chains those answers make, and stops where they say Some(0).
    bool answers_with_adapter_spans() const {
        int rule = 0;
        for (const Stage &st : stages_)
            if (st.span_rule) rule = st.span_rule;
        return rule == 2;
    }
    /// `try_seek` through the chain, adapter by adapter as rodio does it: an adapter that cannot seek (reverb = Mix,
    /// mix.rs:116-120) fails the call before anything moved; otherwise the upstream seeks, what was pulled and processed
    /// ahead is dropped, and every adapter does to its state what its `try_seek` does -- filters and the limiter start
    /// from zero (blt.rs:350-377, limit.rs:1139-1158), a gain ramp continues from `pos` (linear_ramp.rs:141-146), the AGC
    /// and the sample-rate converter keep theirs (agc.rs:593-597, uniform.rs:136-144).
    bool try_seek(Nanos pos) override {
        for (const Stage &st : stages_)
            if (!st.seekable) return false;
        if (!up_->try_seek(pos)) return false;
        reader_.restart();
        restart(ch_);
        for (Stage &st : stages_)
            if (st.on_seek) st.on_seek(pos);
        // (the adapters' counts start over: what the chain emits from here on is the stream behind the new position, less the samples that keep
        // the consumer's channel -- restart())
        hint_->log.clear();
        hint_->out_base = handed_out();
        hint_->skip = handed_out() % ch_;
        hint_->in_base = pulled_total_;
        if (span_log_on_) {  // what was pulled ahead is gone: the next sample pulled is the one the consumer's cursor reaches
            std::uint64_t q = handed_out();
            for (std::size_t k = stages_.size(); k-- > 0;)
                if (stages_[k].span_in_pos) q = stages_[k].span_in_pos(q);
            in_total_ = q;
            span_log_.clear();
            up_ended_ = false;
        }
        return true;
    }

    // -- builder methods (source/mod.rs:255-731); call before the first next()
    GpuSource &amplify(float factor) {  // amplify.rs:64
        return push([factor](Ctx &c) { check(rh_amplify(c.out, c.in, c.n, factor, c.stream), "rh_amplify"); return c.n; }).any_format();
    }
    GpuSource &amplify_decibel(float db) { return amplify(rh_db_to_linear(db)); }  // amplify.rs:33-35, math.rs:51-56
    GpuSource &distortion(float gain, float threshold) {  // distortion.rs:66-72
        return push([=](Ctx &c) { check(rh_distortion(c.out, c.in, c.n, gain, threshold, c.stream), "rh_distortion"); return c.n; }).any_format();
    }
    enum class DitherAlgorithm { GPDF = 0, HighPass = 1, RPDF = 2, TPDF = 3 };  // dither.rs:40-69
    /// dither.rs:217-242; the noise of sample k is a function of (seed, k) -- see rh_dither.
    GpuSource &dither(std::uint32_t target_bits, DitherAlgorithm algorithm = DitherAlgorithm::TPDF, std::uint64_t seed = 0) {
        const std::uint16_t ch = ch_;
        auto pos = std::make_shared<std::uint64_t>(0);
        return push([=](Ctx &c) {
            check(rh_dither(c.out, c.in, c.n, *pos, ch, target_bits, (std::int32_t)algorithm, seed, c.stream), "rh_dither");
            *pos += c.n;
            return c.n;
        });
    }
    /// How the filters of this chain run.  By default (neither call) every filter decides for itself by THE FILTER CONTRACT
    /// (rodio_hip.h, rh_filter_scan_ok): time-parallel (rh_biquad mode 1) where that stays within 1e-5 of rodio's own f32
    /// recurrence for a full-scale source, the reference's operation order (mode 0, bit for bit, one lane per channel) where it
    /// would not -- low cutoffs, where rodio's recurrence amplifies its own rounding noise past 1e-5.  exact_filters(true): the
    /// reference's order throughout.  exact_filters(false): time-parallel throughout -- closer to the exact response than rodio
    /// is, further than 1e-5 from rodio at low cutoffs.  (Blocks the scan kernel does not take run in reference order on the
    /// same state either way.)
    GpuSource &exact_filters(bool on = true) {
        filter_mode_ = on ? 1 : 2;
        return *this;
    }
    GpuSource &low_pass(std::uint32_t freq) { return blt(0, freq, 0.5f); }   // blt.rs:11-16
    GpuSource &high_pass(std::uint32_t freq) { return blt(1, freq, 0.5f); }  // blt.rs:18-24
    GpuSource &low_pass_with_q(std::uint32_t freq, float q) { return blt(0, freq, q); }
    GpuSource &high_pass_with_q(std::uint32_t freq, float q) { return blt(1, freq, q); }
    GpuSource &reverb(Nanos duration, float amplitude) {  // source/mod.rs:628-634
        const std::uint64_t d = rh_delay_samples((std::uint64_t)duration.count(), rate_, ch_);
        may_cut_ = may_cut_ || d % ch_ != 0;
        auto h = std::make_shared<Handle<rh_echo>>();
        check(rh_echo_create(&h->p, d, amplitude), "rh_echo_create");
        h->destroy = [](rh_echo *e) { (void)rh_echo_destroy(e); };
        struct Seen {
            std::uint64_t n = 0;  // samples of the input so far
            bool all = false;     // ... and that is all of it
        };
        auto seen = std::make_shared<Seen>();
        // (size_hint) Once the plain branch has found its Buffered source at its end, its iterator builds a chain on `Span::End` at every call --
        // ONE channel at 44100 Hz (buffered.rs:218-233), whatever the source was -- and an empty SampleRateConverter 44100 -> rate answers a whole
        // chunk of `to` samples less the one its failed next() counted (sample_rate.rs:190,226-229: the chunk it believes it is in), which the
        // ChannelCountConverter 1 -> channels behind it multiplies (channels.rs:93): rodio's lower bound while only the echo plays.  Restated, not judged.
        std::uint64_t g44 = 44100, gb = rate_;
        while (gb) {
            const std::uint64_t t = g44 % gb;
            g44 = gb;
            gb = t;
        }
        const std::size_t end_chunk = rate_ == 44100 ? 0 : (std::size_t)(rate_ / g44 - 1) * ch_;
        return push(
            [h, d, seen](Ctx &c) {
                seen->n += c.n;
                seen->all = seen->all || c.flush;
                if (c.n) check(rh_echo_process(h->p, c.out, c.in, c.n, c.stream), "rh_echo_process");
                if (!c.flush) return c.n;
                if (d) check(rh_echo_flush(h->p, c.out + c.n, c.stream), "rh_echo_flush");  // the delayed clone outlives the source
                return c.n + (std::size_t)d;
            },
            [d](std::size_t n) { return n + (std::size_t)d; })
            .not_seekable()
            .spans(1)  // Mix::current_span_len() is None (mix.rs:92-94)
            .duration([captured = total_duration(), duration](std::optional<Nanos>) -> std::optional<Nanos> {  // mix.rs:104-112: max(f1, f1 + d), f1 asked when the source was buffered (buffered.rs:16)
                return captured ? std::optional<Nanos>(*captured + duration) : std::nullopt;
            })
            // mix.rs:56-67 over two UniformSourceIterators (mix.rs:10-22) at the source's own format: (the larger lower bound, None).  The plain
            // branch reads a Buffered source, whose bounds are (0, None) (buffered.rs:192-195): 0.  The echo -- Delay over Amplify over the clone --
            // owes d - e samples of silence (delay.rs:78-84), of which its iterator's open chain admits what Take has left (uniform.rs:56,181-196:
            // chains of min(the buffered span + the silence owed, 32768) samples), counted in whole frames from where the chain stands
            // (channels.rs:88-102); before the first sample no chain exists and the iterator answers with Delay's own bound (uniform.rs:105).
            .hints([d, ch = ch_, s0 = current_span_len().value_or(32768), seen, end_chunk](const HintAt &, std::uint64_t e) {
                if (e == 0) return SizeHint{(std::size_t)d, std::nullopt};
                const std::size_t plain = seen->all && e > seen->n ? end_chunk : 0;  // (the plain branch has returned None: see above)
                if (e >= d) return SizeHint{plain, std::nullopt};
                std::uint64_t start = 0, n = std::min<std::uint64_t>(s0 + d, 32768);
                while (n && start + n <= e) {  // the chains that have run dry (the silence comes first: the buffered span stays where it is)
                    start += n;
                    n = std::min<std::uint64_t>(s0 + (d - start), 32768);
                }
                const std::uint64_t in_chain = e - start, pos = in_chain % ch;
                const std::uint64_t lo = std::min<std::uint64_t>(d - e, n > in_chain ? n - in_chain : 0);
                const std::uint64_t x = (lo + pos) / ch * ch;
                return SizeHint{std::max(plain, (std::size_t)(x > pos ? x - pos : 0)), std::nullopt};
            });
    }
    GpuSource &channel_volume(std::vector<float> gains) {  // channel_volume.rs:71-88
        const std::uint16_t in_ch = ch_;
        const std::uint16_t out_ch = (std::uint16_t)gains.size();
        if (!out_ch) throw std::invalid_argument("channel_volume: no output channels");
        auto rg = std::make_shared<Regroup>();
        auto emitted = std::make_shared<std::uint64_t>(0);
        const std::size_t self = stages_.size();
        push([this, gains, in_ch, out_ch, rg, emitted, self](Ctx &c) {
            return run_grouped(c, in_ch, *rg, [&](const float *in, std::size_t n) {
                const std::size_t frames = n / in_ch;  // (a frame the stream ends in is dropped: `input.next()?`, channel_volume.rs:71-79)
                check(rh_channel_volume(c.out, in, frames, in_ch, gains.data(), out_ch, c.stream), "rh_channel_volume");
                *emitted += frames * out_ch;
                if (n % in_ch) {
                    // ... and rodio's ChannelVolume is left with its channel position at 0 and the sum of the samples it did get: asked AGAIN
                    // after that None it returns one more frame made of that stale sum (:71-88).  Adapters that pass samples on one by one end
                    // with the None; a converter behind it polls its input again while it drains (sample_rate.rs:110-122), a
                    // UniformSourceIterator once more when its chain has ended (uniform.rs:76-97), a Mix through both (refused); and so
                    // does a mixer the chain was handed to (which looks at ended_with_a_stale_frame()).
                    for (std::size_t k = self + 1; k < stages_.size(); ++k)
                        if (stages_[k].span_rule == 1)
                            throw Error(RH_ERR_UNSUPPORTED, "GpuSource::channel_volume: its input ends inside a frame: what rodio's ChannelVolume returns to the converter behind it, "
                                                            "which polls it again, is a frame of its stale sum (channel_volume.rs:71-88)");
                    stale_frame_ = true;
                }
                return frames * out_ch;
            });
        }, [in_ch, out_ch](std::size_t n) { return (n / in_ch + 1) * out_ch; }).on_seek([rg](Nanos) { rg->n = 0; })
            .span_arithmetic(  // channel_volume.rs:103-105: the input's answer as it is -- counted in ITS samples, of which a frame has been taken whenever an output frame begins (:71-79)
                [](std::optional<std::size_t> in, std::uint64_t) { return in; },
                [in_ch, out_ch](std::uint64_t emitted) { return (emitted + out_ch - 1) / out_ch * in_ch; });
        ch_ = out_ch;
        may_cut_ = false;  // (an open last frame is dropped: `input.next()?`)
        return *this;
    }
    GpuSource &spatial(const float emitter[3], const float left_ear[3], const float right_ear[3]) {  // spatial.rs:19-24,48-69
        float g[2];
        check(rh_spatial_gains(emitter, left_ear, right_ear, g), "rh_spatial_gains");
        return channel_volume({g[0], g[1]});
    }
    GpuSource &convert_channels(std::uint16_t to) {  // ChannelCountConverter, channels.rs:57-85
        const std::uint16_t from = ch_;
        if (!to) throw std::invalid_argument("channels are NonZero in rodio");
        auto rg = std::make_shared<Regroup>();
        push([from, to, rg](Ctx &c) {
            return run_grouped(c, from, *rg, [&](const float *in, std::size_t n) {
                const std::size_t frames = n / from;
                check(rh_channels_convert(c.out, in, frames, from, to, c.stream), "rh_channels_convert");
                // A stream that ends inside a frame (reverb with a delay that is no whole number of frames: delay.rs:14 counts samples): the
                // converter hands on the samples of the open frame as far as both layouts go -- positions below `from` are plain input.next()
                // (channels.rs:57-67), a None ends the stream -- so min(rest, to) samples follow the whole frames.
                const std::size_t rest = std::min<std::size_t>(n % from, to);
                if (rest) check(rh_memcpy_d2d(c.out + frames * to, in + frames * from, rest * sizeof(float), c.stream), "rh_memcpy_d2d");
                return frames * to + rest;
            });
        }, [from, to](std::size_t n) { return (n / from + 1) * to + to; }).spans(1).on_seek([rg](Nanos) { rg->n = 0; })  // a bare converter is an iterator, not a Source: what wraps it sees no spans
            // channels.rs:88-102: whole frames of what the input still holds, from where the converter stands -- it has taken min(pos, from) samples of
            // the frame it is in (the surplus channels go when the frame is complete, :77-81)
            .hints([from, to](const HintAt &in, std::uint64_t e) {
                const std::uint64_t pos = e % to, consumed = std::min<std::uint64_t>(from, pos);
                const SizeHint h = in(e / to * from + consumed);
                auto f = [&](std::size_t v) {
                    const std::uint64_t x = (v + consumed) / from * to;
                    return (std::size_t)(x > pos ? x - pos : 0);
                };
                return SizeHint{f(h.lower), h.upper ? std::optional<std::size_t>(f(*h.upper)) : std::nullopt};
            });
        ch_ = to;
        return *this;
    }
    GpuSource &convert_sample_rate(std::uint32_t to) {  // SampleRateConverter, sample_rate.rs:52-201
        const std::uint32_t from = rate_;
        const std::uint16_t ch = ch_;
        if (!to) throw std::invalid_argument("sample_rate is NonZero in rodio");
        if (from == to) return *this;  // sample_rate.rs:133-136
        auto h = std::make_shared<Handle<rh_resampler>>();
        check(rh_resampler_create(&h->p, from, to, ch), "rh_resampler_create");
        h->destroy = [](rh_resampler *r) { (void)rh_resampler_destroy(r); };
        auto rg = std::make_shared<Regroup>();
        // sample_rate.rs:204-238: the converter's bounds are functions of its counters; they run here, over counts (detail::UniformCounter)
        auto cnt = std::make_shared<detail::UniformCounter>(ch, to, true);
        auto fed = std::make_shared<std::uint64_t>(0);
        cnt->feed(detail::Piece{0, true, false, ch, from, 0, false, std::nullopt, ch, from, true, std::nullopt}, 0);
        push([h, ch, rg, cnt, fed, from](Ctx &c) {
            cnt->feed(detail::Piece{c.n, false, c.flush, ch, from, 0, c.flush, std::nullopt, ch, from, false, std::nullopt}, *fed);
            *fed += c.n;
            return run_grouped(c, ch, *rg, [&](const float *in, std::size_t n) {
                std::uint64_t m = 0;  // (a frame the stream ends in is not converted here: `uniform` is the adapter that knows what rodio's converter makes of it)
                check(rh_resampler_process(h->p, c.out, c.out_cap / ch, in, n / ch, c.flush ? 1 : 0, &m, c.stream), "rh_resampler_process");
                return (std::size_t)m * ch;
            });
        }, [from, to, ch](std::size_t n) { return (std::size_t)((std::uint64_t)(n / ch + 3) * to / from + 2) * ch; }).spans(1)
            .hints([cnt](const HintAt &in, std::uint64_t e) { return cnt->hint_at(e, in).value_or(SizeHint{0, std::size_t(0)}); })
            .on_seek([cnt, fed, ch, from, to](Nanos) {  // (the counts start over with the stream behind the new position)
                *cnt = detail::UniformCounter(ch, to, true);
                cnt->feed(detail::Piece{0, true, false, ch, from, 0, false, std::nullopt, ch, from, true, std::nullopt}, 0);
                *fed = 0;
            });
        rate_ = to;
        return *this;
    }
    /// UniformSourceIterator::new(src, channels, rate) (uniform.rs:50-97).  The upstream is pulled the way rodio pulls it:
    /// `current_span_len()` is asked whenever the converter chain has run dry, min(span, 32768) samples go to a FRESH
    /// SampleRateConverter -> ChannelCountConverter pair, and every span ends with its last frame verbatim -- so a
    /// SamplesBuffer or a decoder (which report spans) and a generator (None: one continuous conversion) each come out as
    /// they do in rodio.  Spans travel through the adapters in front that keep the sample count (amplify, filters, limiter,
    /// ...: they forward current_span_len()); behind reverb (Mix: None, mix.rs:92-94) or a bare converter the stream is
    /// continuous.  Behind take_duration / delay the spans are the ones rodio's adapters report there (TakeDuration:
    /// Some(what it still admits) unless the input's span is shorter -- so chains of 32768 samples over a generator -- and
    /// Some(0) in front of the silence that completes a cut frame; Delay: the input's answer plus the silence it still owes:
    /// span_behind(); ChannelVolume: its input's answer, counted in the input's samples).
    GpuSource &uniform(std::uint16_t channels, std::uint32_t sample_rate) {
        if (!channels || !sample_rate) throw std::invalid_argument("channels and sample_rate are NonZero in rodio");
        int rule = 0;
        for (const Stage &st : stages_)
            if (st.span_rule) rule = st.span_rule;
        // (rule 2 over a continuous upstream: the spans are the ones rodio's adapters report there -- TakeDuration's Some(what it still admits),
        // in chains of 32768 samples, ending with Some(0) in front of the silence that completes a cut frame -- computed at every bootstrap from
        // the adapters' sample counts: span_behind())
        const bool computed = rule == 2;
        const std::size_t upto = stages_.size();
        if (computed) (void)span_behind(upto, 0);  // (combinations the counts do not cover are refused here, not at the first block)
        // (rule != 0: the input is continuous from here on -- current_span_len() is None behind a converter or a Mix -- so the iterator builds
        // ChannelCountConverter(SampleRateConverter(..)) once, uniform.rs:62-67: ONE span that opens with the first sample and closes with the
        // stream, through the same planner, which also knows what rodio's converters make of a stream that ends inside a frame)
        const bool one_span = rule != 0;
        if (!one_span) span_aware_ = true;
        may_cut_ = may_cut_ || (!one_span && ch_ > 1 && up_->current_span_len().has_value());  // (a span that cuts a frame leaves a run of samples that need not fill an output frame)
        may_cut_ = may_cut_ || (computed && ch_ > 1);
        auto started = std::make_shared<bool>(false);
        struct Computed {  // the open span of the computed mode
            std::uint64_t taken = 0;  // samples of the input consumed so far
            std::size_t left = 0;     // samples the open span still admits (SpanReader::kOpenEnded: the answer was None)
            bool open = false, fresh = true, ended = false;
        };
        auto cs = std::make_shared<Computed>();
        auto plan = std::make_shared<detail::UniformPlanner>(channels, sample_rate);
        auto win = std::make_shared<detail::DeviceBuf>();
        auto keep = std::make_shared<detail::DeviceBuf>();
        // What UniformSourceIterator emits is a stream of SAMPLES: a span that ends inside a frame leaves a run that need not fill a frame of
        // `channels`, and the next span's output follows directly behind it.  The adapters behind work on frames, so a block hands on
        // whole frames and the samples of a frame that is not complete yet wait here for the next block (at the end of the stream they
        // are handed on as they are: rodio's adapters take them too).
        auto part = std::make_shared<detail::DeviceBuf>(std::max<std::size_t>(64, channels));
        auto part_n = std::make_shared<std::size_t>(0);
        const std::uint16_t in_ch = ch_;
        const std::uint32_t from = rate_, to = sample_rate;
        // uniform.rs:100-108: the bounds of the converter chain that is open are functions of the converters' counters: they run here, over
        // counts, fed with the same pieces as the planner (detail::UniformCounter)
        auto cnt = std::make_shared<detail::UniformCounter>(channels, sample_rate);
        auto fed = std::make_shared<std::uint64_t>(0);
        const std::optional<Nanos> captured = total_duration();  // uniform.rs:37: asked once, when the iterator is built
        auto feed = [cnt, fed](const detail::Piece &p) {
            cnt->feed(p, *fed);
            *fed += p.n;
        };
        push(
            [=](Ctx &c) {
                plan->begin_block();
                const std::size_t hs = plan->held_samples();
                win->reset(hs + c.n + 4);
                if (hs) check(rh_memcpy_d2d(win->get(), keep->get(), hs * sizeof(float), c.stream), "rh_memcpy_d2d");
                if (c.n) check(rh_memcpy_d2d(win->get() + hs, c.in, c.n * sizeof(float), c.stream), "rh_memcpy_d2d");
                std::vector<detail::UniformPlanner::Seg> segs;
                if (computed) {
                    std::size_t off = 0;
                    while (!cs->ended && (off < c.n || (c.flush && cs->open))) {
                        std::optional<std::size_t> answer;
                        if (!cs->open) {  // uniform.rs:50-68: bootstrap
                            answer = span_behind(upto, cs->taken);
                            if (answer && *answer == 0) {  // Take{n: 0}: the chain is empty, next() is None -- whatever the input still holds
                                cs->ended = true;
                                break;
                            }
                            cs->left = answer ? std::min<std::size_t>(*answer, 32768) : detail::SpanReader::kOpenEnded;
                            cs->open = cs->fresh = true;
                        }
                        const std::size_t take = std::min(cs->left, c.n - off);
                        if (cs->left != detail::SpanReader::kOpenEnded) cs->left -= take;
                        const bool last = c.flush && off + take == c.n;  // the input returned None inside the span
                        const bool closes = cs->left == 0 || last;
                        if (take || (closes && !cs->fresh)) {
                            detail::Piece p{take, cs->fresh, closes, in_ch, from, 0, last && cs->left != 0, cs->fresh ? answer : std::nullopt, in_ch, from, cs->fresh, std::nullopt};
                            plan->add(p, segs);
                            feed(p);
                        }
                        if (take) cs->fresh = false;
                        off += take;
                        cs->taken += take;
                        if (closes) cs->open = false;
                        if (last) cs->ended = true;
                    }
                    if (cs->ended) c.end = true;  // the iterator has returned None: the stream ends here although the input may have more
                } else if (one_span) {
                    if (c.n || (c.flush && *started)) {
                        detail::Piece p{c.n, !*started, c.flush, in_ch, from, 0, c.flush, std::nullopt, in_ch, from, !*started, std::nullopt};
                        *started = true;
                        plan->add(p, segs);
                        feed(p);
                    }
                } else {
                    for (const detail::Piece &p : pieces_) {
                        plan->add(p, segs);
                        feed(p);
                    }
                }
                if (c.flush || c.end) cnt->input_ended();
                plan->end_block();
                const std::size_t carried = *part_n;
                if (carried) check(rh_memcpy_d2d(c.out, part->get(), carried * sizeof(float), c.stream), "rh_memcpy_d2d");
                std::vector<rh_uniform_seg> table;
                for (const detail::UniformPlanner::Seg &sg : segs) {
                    rh_uniform_seg g = sg.g;
                    g.src = win->get() + sg.src_off;
                    g.dst = c.out + carried + sg.dst_off;
                    table.push_back(g);
                }
                const std::size_t total = carried + plan->out_samples();
                if (total + channels > c.out_cap) throw Error(RH_ERR_CAPACITY, "GpuSource::uniform: block capacity");
                check(rh_uniform_segments(table.data(), (std::uint32_t)table.size(), c.stream), "rh_uniform_segments");
                if (const std::size_t kn = plan->keep_samples()) {
                    keep->reset(kn);
                    check(rh_memcpy_d2d(keep->get(), win->get() + plan->keep_offset(), kn * sizeof(float), c.stream), "rh_memcpy_d2d");
                }
                const std::size_t rest = c.flush ? 0 : total % channels;
                if (rest) check(rh_memcpy_d2d(part->get(), c.out + total - rest, rest * sizeof(float), c.stream), "rh_memcpy_d2d");
                *part_n = rest;
                return total - rest;
            },
            [this, in_ch, channels, from, to, one_span, computed](std::size_t n) {  // every span may add its verbatim last frame, or what the converters make of a cut frame
                // (the block's pieces may come in other formats than the chain was built for: the fewest channels and the lowest rate among them bound it)
                const std::uint64_t ich = one_span ? in_ch : std::min<std::uint64_t>(in_ch, block_min_ch_ ? block_min_ch_ : in_ch), ifrom = one_span ? from : std::min<std::uint64_t>(from, block_min_rate_ ? block_min_rate_ : from);
                const std::uint64_t f = n / ich + 1;
                return (std::size_t)(std::max<std::uint64_t>(f, f * to / ifrom + 2) + (detail::UniformPlanner::close_slack_frames((std::uint32_t)ifrom, to) + 1) * ((computed ? pieces_.size() * ((in_ch + in_ch0() - 1) / in_ch0()) + n / 32768 + 4 : one_span ? 1 : pieces_.size()) + 2) + 1) * channels;  // (computed: a span of the upstream per piece -- more where a channel_volume in between made more samples of them --, the cuts at 32768, a take's end)
            })
            .on_seek([plan, part_n, started, cs, channels, sample_rate, cnt, fed](Nanos) {  // what was pulled ahead is gone: the next span starts a fresh chain
                *plan = detail::UniformPlanner(channels, sample_rate);
                *part_n = 0;
                *started = false;
                cs->open = false;
                *cnt = detail::UniformCounter(channels, sample_rate);
                *fed = 0;
            })
            .duration([captured](std::optional<Nanos>) { return captured; })  // uniform.rs:131-133
            .hints([cnt](const HintAt &in, std::uint64_t e) { return cnt->hint_at(e, in).value_or(SizeHint{0, std::nullopt}); });
        stages_.back().span_rule = 1;
        last_kind_ = 1;
        stages_.back().fmt = one_span ? 0 : 3;  // every piece comes with its own format (uniform.rs:58-59: read at every bootstrap); behind another converter nothing changes any more
        ch_ = channels;
        rate_ = sample_rate;
        return *this;
    }
    GpuSource &limit(const rh_limit_params &settings) {  // limit.rs:94-130,853-988
        auto chp = std::make_shared<std::uint16_t>(ch_);
        const std::uint32_t rate = rate_;  // (the coefficients belong to the rate the limiter was built for: LimitBase outlives a change of format, limit.rs:669-693)
        auto st = state(2u * ch_);
        const rh_stream sm = stream_;
        scan_kernels_ = true;
        auto fc = std::make_shared<FrameCarry>();
        return push([=](Ctx &c) {
            const std::uint16_t ch = *chp;
            return run_framewise(c, ch, *fc, st->get(), 2u * ch, [&](float *out, const float *in, std::size_t frames, float *state) {
                check(rh_limit(out, in, frames, ch, rate, 1, &settings, state, c.stream), "rh_limit");
            });
        }).on_seek([st, chp, fc, sm](Nanos) {  // limit.rs:1139-1158
            check(rh_memset(st->get(), 0, 2u * *chp * sizeof(float), sm), "rh_memset");
            fc->n = fc->lead = 0;
        }).on_format([st, chp, fc, sm](std::uint16_t ch, std::uint32_t) {  // limit.rs:652-695: another channel count rebuilds the state (and the channel position); another rate changes nothing
                if (ch == *chp) return;
                st->reset(2u * ch);
                check(rh_memset(st->get(), 0, 2u * ch * sizeof(float), sm), "rh_memset");
                *chp = ch;
                fc->n = fc->lead = 0;
            });
    }
    GpuSource &automatic_gain_control(const rh_agc_params &settings) {  // agc.rs:133-171,397-504
        auto ratep = std::make_shared<std::uint32_t>(rate_);
        auto st = std::make_shared<detail::DeviceBuf>(rh_agc_state_floats());
        check(rh_agc_state_init(st->get(), 1, stream_), "rh_agc_state_init");
        const rh_stream sm = stream_;
        return push([=](Ctx &c) {
            check(rh_agc(c.out, c.in, c.n, *ratep, 1, &settings, st->get(), c.stream), "rh_agc");
            return c.n;
        }).on_format([st, ratep, sm](std::uint16_t, std::uint32_t rate) {  // agc.rs:524-548: coefficients for the new rate, a fresh window, peak 0, gain 1
            *ratep = rate;
            check(rh_agc_state_init(st->get(), 1, sm), "rh_agc_state_init");
        });
    }
    GpuSource &linear_gain_ramp(Nanos duration, float start_gain, float end_gain, bool clamp_end) {  // linear_ramp.rs:79-110
        const std::uint16_t ch = ch_;
        const std::uint32_t rate = rate_;
        auto pos = std::make_shared<std::uint64_t>(0);
        return push([=](Ctx &c) {
            check(rh_linear_gain_ramp(c.out, c.in, c.n, *pos, ch, rate, (std::uint64_t)duration.count(), start_gain, end_gain, clamp_end ? 1 : 0, c.stream), "rh_linear_gain_ramp");
            *pos += c.n;
            return c.n;
        }).on_seek([pos, ch, rate](Nanos p) {  // linear_ramp.rs:141-146: elapsed = pos
            const std::uint64_t ns = (std::uint64_t)p.count();  // frames = floor(ns * rate / 1e9) without leaving 64 bits
            *pos = (ns / 1000000000ull * rate + ns % 1000000000ull * rate / 1000000000ull) * ch;
        });
    }
    /// `take_duration(d)`, with `set_filter_fadeout()` when `fade_out` (take.rs:96-148): the samples the duration admits, a cut
    /// frame completed with zeros, then the end of the stream.  `try_seek` moves the upstream and starts the duration over from
    /// the new position: what is left is the requested duration less `pos` (take.rs:222-231).
    GpuSource &take_duration(Nanos duration, bool fade_out = false) {
        const std::uint16_t ch = ch_;
        const std::uint32_t rate = rate_;
        const std::uint64_t requested = (std::uint64_t)duration.count();
        struct Took {
            std::uint64_t remaining;     // of the duration, at the next sample the adapter takes
            std::uint32_t phase = 0;     // samples of the current frame already emitted (take.rs:124-131)
            bool done = false;
            std::uint64_t since = 0;     // what was left of the duration at the start of the stream / behind the last seek (size_hint counts from there)
        };
        auto tk = std::make_shared<Took>();
        tk->remaining = tk->since = requested;
        // the samples the duration admits: one per duration_per_sample = 1e9 / (rate * channels) ns, integer (take.rs:21-24,124-131)
        const std::uint64_t per_sample = 1000000000ull / ((std::uint64_t)rate * ch);
        const std::uint64_t admits = per_sample ? requested / per_sample : 0;
        return push(
            [=](Ctx &c) {
                c.end = true;
                if (tk->done) return std::size_t(0);
                std::uint64_t m = 0, after = 0;
                std::int32_t ended = 0;
                check(rh_take_duration_from(c.out, c.in, c.n, tk->remaining, requested, tk->phase, ch, rate, fade_out ? 1 : 0, &m, &ended, &after, c.stream), "rh_take_duration_from");
                if (per_sample) tk->phase = (std::uint32_t)((tk->phase + (tk->remaining - after) / per_sample) % ch);
                tk->remaining = after;
                tk->done = ended != 0;
                c.end = tk->done;
                return (std::size_t)m;
            },
            [ch](std::size_t n) { return n + ch; })
            .span_arithmetic(
                [admits](std::optional<std::size_t> in, std::uint64_t emitted) -> std::optional<std::size_t> {  // take.rs:176-195
                    const std::uint64_t rem = admits > emitted ? admits - emitted : 0;
                    if (rem == 0) return std::size_t(0);
                    return in && *in < rem ? in : std::optional<std::size_t>((std::size_t)rem);
                },
                [admits](std::uint64_t emitted) { return std::min(emitted, admits); })  // (behind them: the silence that completes a cut frame)
            .keeps_the_sample_count()
            .on_seek([tk, requested](Nanos pos) {  // take.rs:222-231
                const std::uint64_t p = (std::uint64_t)pos.count();
                tk->remaining = tk->since = requested > p ? requested - p : 0;
                tk->phase = 0;
                tk->done = false;
            })
            .duration([duration](std::optional<Nanos> in) -> std::optional<Nanos> {  // take.rs:209-219
                if (!in) return std::nullopt;
                return *in < duration ? *in : duration;
            })
            // take.rs:151-171: what the remaining duration admits, cut to the input's bounds -- an upper bound even over an input that has none
            .hints([tk, per_sample](const HintAt &in, std::uint64_t e) {
                const std::uint64_t can = per_sample ? tk->since / per_sample : 0, took = std::min(e, can);
                const std::uint64_t remaining = tk->since - took * per_sample;
                if (!per_sample || remaining == 0) return SizeHint{0, std::size_t(0)};
                const std::size_t rs = (std::size_t)(remaining / per_sample);
                const SizeHint h = in(took);
                return SizeHint{std::min(h.lower, rs), h.upper ? std::min(*h.upper, rs) : rs};
            });
    }
    /// `delay(d)` (delay.rs:8-16,68-75): rh_delay_samples() zeros in front of the stream.  Not seekable here (rodio's Delay
    /// splits the position between the silence and the input; the shim's seek hands every adapter the same position).
    GpuSource &delay(Nanos duration) {
        const std::uint64_t d = rh_delay_samples((std::uint64_t)duration.count(), rate_, ch_);
        may_cut_ = may_cut_ || d % ch_ != 0;
        const std::uint16_t ch = ch_;
        auto first = std::make_shared<bool>(true);
        // The silence counts SAMPLES (delay.rs:14): one that is no whole number of frames shifts the stream inside its frames.  The adapters
        // behind work on frames, so a block hands on whole frames and the samples of the frame that is not complete yet wait here for the next
        // block (as behind `uniform`); the end of the stream hands them on as they are.
        auto part = std::make_shared<detail::DeviceBuf>(64);
        auto part_n = std::make_shared<std::size_t>(0);
        return push(
                   [=](Ctx &c) {
                       std::size_t k;
                       if (*first) {
                           *first = false;
                           check(rh_delay(c.out, c.in, c.n, d, c.stream), "rh_delay");
                           k = c.n + (std::size_t)d;
                       } else {
                           const std::size_t carried = *part_n;
                           if (carried) check(rh_memcpy_d2d(c.out, part->get(), carried * sizeof(float), c.stream), "rh_memcpy_d2d");
                           check(rh_amplify(c.out + carried, c.in, c.n, 1.0f, c.stream), "rh_amplify");  // x * 1.0 == x: a copy into the other buffer
                           k = carried + c.n;
                       }
                       const std::size_t rest = c.flush ? 0 : k % ch;
                       if (rest) {
                           part->reset(ch);
                           check(rh_memcpy_d2d(part->get(), c.out + k - rest, rest * sizeof(float), c.stream), "rh_memcpy_d2d");
                       }
                       *part_n = rest;
                       return k - rest;
                   },
                   [d, ch](std::size_t n) { return n + (std::size_t)d + ch; })
            .not_seekable()
            .span_arithmetic(
                [d](std::optional<std::size_t> in, std::uint64_t emitted) -> std::optional<std::size_t> {  // delay.rs:94-98: the input's answer + the silence still owed
                    if (!in) return std::nullopt;
                    return *in + (std::size_t)(d > emitted ? d - emitted : 0);
                },
                [d](std::uint64_t emitted) { return emitted > d ? emitted - d : 0; })
            .duration([duration](std::optional<Nanos> in) { return in ? std::optional<Nanos>(*in + duration) : std::nullopt; })  // delay.rs:111-115: + the REQUESTED delay
            .hints([d](const HintAt &in, std::uint64_t e) {  // delay.rs:78-84: the input's bounds plus the silence still owed
                const std::size_t owed = (std::size_t)(d > e ? d - e : 0);
                SizeHint h = in(e > d ? e - d : 0);
                h.lower += owed;
                if (h.upper) h.upper = *h.upper + owed;
                return h;
            });
    }
    GpuSource &fade_in(Nanos duration) { return linear_gain_ramp(duration, 0.0f, 1.0f, false); }  // fadein.rs:11-13
    GpuSource &fade_out(Nanos duration) { return linear_gain_ramp(duration, 1.0f, 0.0f, true); }  // fadeout.rs:13

protected:
    void block_done() override {  // a bounded wait inside the limiter's scan expired (never seen on a healthy device): fail loudly
        if (scan_kernels_) check(rh_async_status(), "rh_async_status");
    }
    void enqueue(Slot &s) override {
        if (!follow_known_) {  // an upstream that reports spans may change its format between them: it is pulled span by span
            follow_known_ = true;
            follow_spans_ = !span_aware_ && up_->current_span_len().has_value();
            if (follow_spans_) reader_ = detail::SpanReader(up_.get(), false);
            for (const Stage &st : stages_) span_log_on_ = span_log_on_ || st.span_fn;
        }
        const std::size_t want = block_frames_ * cur_in_ch_;
        hint_->log.note(pulled_total_, up_->size_hint());  // (size_hint(): what the upstream answers where this block's first sample is pulled)
        // The slot's page-locked staging block is about to be rewritten: the copy that read it two blocks ago must have run.  A host
        // consumer has waited for that block's event already (advance()); one that takes the blocks on the device never waits on the
        // host, so the wait is here (otherwise the refill races the asynchronous host-to-device copy of the block before).
        // Almost always satisfied by the time the slot comes round again.
        if (device_out_) check(rh_event_synchronize(s.done), "rh_event_synchronize");
        s.in.reset(want + std::max<std::size_t>(64, cur_in_ch_));  // (+ room for a cut frame of any layout)
        // 1. pull: runs of samples of one format (a format changes only between two spans)
        struct Run {
            std::size_t off, n;
            std::uint16_t ch;
            std::uint32_t rate;
            std::size_t p0, p1;  // its pieces
        };
        std::vector<Run> runs;
        std::vector<detail::Piece> all;
        std::vector<std::size_t> piece_off;
        std::size_t n = 0;
        bool flush = false;
        if (span_aware_ || follow_spans_) {  // span by span, asking for the span where rodio asks (UniformSourceIterator: uniform.rs:50-68; the adapters' SpanTracker: span.rs:66-101)
            while (n < want) {
                std::uint16_t pch = 0;
                std::uint32_t prate = 0;
                if (!reader_.peek(pch, prate)) break;
                const std::size_t room = (want - n) / pch;
                if (!room) break;
                detail::Piece pc;
                const bool produced = reader_.read_piece(s.in.get() + n, room, pc);
                if (produced) {
                    if (runs.empty() || pc.src_ch != runs.back().ch || pc.src_rate != runs.back().rate) runs.push_back(Run{n, 0, pc.src_ch, pc.src_rate, all.size(), all.size()});
                    runs.back().n += pc.n;
                    runs.back().p1 = all.size() + 1;
                    piece_off.push_back(n);
                    all.push_back(pc);
                    if (span_log_on_ && pc.src_opens) span_log_.emplace_back(in_total_ + n, pc.src_span);
                    n += pc.n;
                }
                if (reader_.ended() || !produced) break;
            }
            flush = reader_.ended();
        } else {
            n = up_->read(s.in.get(), want);
            // (sources end on frame boundaries, source/mod.rs:169-178; one that does not hands its cut frame to the adapters as a spanned one
            // does: they carry open frames anyway)
            flush = n < want;
            runs.push_back(Run{0, n, cur_in_ch_, cur_in_rate_, 0, 0});
        }
        in_total_ += n;
        pulled_total_ += n;
        up_ended_ = up_ended_ || flush;
        if (span_log_.size() > 4096) {  // nobody has asked for a while: what lies in front of the sample the consumer's cursor reaches is not asked for any more
            std::uint64_t q = handed_out();
            for (std::size_t k = stages_.size(); k-- > 0;)
                if (stages_[k].span_in_pos) q = stages_[k].span_in_pos(q);
            std::size_t i = 0;
            while (i + 1 < span_log_.size() && span_log_[i + 1].first <= q) ++i;
            span_log_.erase(span_log_.begin(), span_log_.begin() + (std::ptrdiff_t)i);
        }
        if (runs.empty()) runs.push_back(Run{0, 0, cur_in_ch_, cur_in_rate_, 0, 0});
        // capacity of the ping-pong buffers: the largest block any stage can emit
        pieces_.assign(all.begin(), all.end());
        block_min_ch_ = 0, block_min_rate_ = 0;
        for (const detail::Piece &pc : all) {
            block_min_ch_ = block_min_ch_ ? std::min(block_min_ch_, pc.ch) : pc.ch;
            block_min_rate_ = block_min_rate_ ? std::min(block_min_rate_, pc.rate) : pc.rate;
        }
        std::size_t cap = want + 64, m = want + 64;
        for (const Stage &st : stages_) cap = std::max(cap, m = st.bound(m));
        cap = ((cap + 3) & ~std::size_t(3)) + 64;  // + room for one padding frame
        a_.reset(cap);
        b_.reset(cap);
        bool counted = spans_stay_in_place(), fixed = false;  // every adapter hands on one sample per sample / an adapter gives one format out whatever comes in
        for (const Stage &st : stages_) fixed = fixed || st.fmt == 3;
        if (runs.size() > 1) acc_.reset(cap * runs.size());
        // 2. run by run through the adapters
        s.marks.clear();
        std::size_t total = 0;
        float *result = a_.get();
        bool ends = false;
        for (std::size_t r = 0; r < runs.size(); ++r) {
            const Run &run = runs[r];
            if (run.ch != cur_in_ch_ || run.rate != cur_in_rate_) {  // the boundary: every adapter does what rodio's does there
                for (Stage &st : stages_) {
                    if (st.fmt == 3) break;  // (it meets the spans itself; what lies behind it sees one format)
                    if (st.fmt == 2) throw Error(RH_ERR_UNSUPPORTED, "GpuSource: the upstream changed its format mid-stream in front of an adapter that is not mirrored across such a change (convert with .uniform() first)");
                    if (st.fmt == 1) st.on_format(run.ch, run.rate);
                }
                cur_in_ch_ = run.ch;
                cur_in_rate_ = run.rate;
            }
            pieces_.assign(all.begin() + (std::ptrdiff_t)run.p0, all.begin() + (std::ptrdiff_t)run.p1);
            float *cur = a_.get(), *oth = b_.get();
            std::size_t k = run.n;
            if (k) check(rh_memcpy_h2d(cur, s.in.get() + run.off, k * sizeof(float), stream_), "rh_memcpy_h2d");
            bool run_ends = flush && r + 1 == runs.size();  // the upstream ended, or a stage says so: the stages behind it see the end of their input
            for (Stage &st : stages_) {
                Ctx c{oth, cur, k, cap, run_ends, stream_};
                k = st.run(c);
                run_ends = run_ends || c.end;
                std::swap(cur, oth);
            }
            ends = ends || run_ends;
            // the format of what came out: the adapters' own where one of them fixes it (or nothing changed), the run's otherwise
            const bool as_built = fixed || (run.ch == in_ch0() && run.rate == in_rate0());
            FormatMark mk{total, as_built ? ch_ : run.ch, as_built ? rate_ : run.rate, std::nullopt};
            if (counted && (span_aware_ || follow_spans_)) {  // current_span_len() is the input's: a mark per span that opens in the run
                bool first = true;
                for (std::size_t pi = run.p0; pi < run.p1; ++pi) {
                    if (all[pi].src_opens) open_span_ = all[pi].src_span;
                    if (first || all[pi].src_opens) {
                        mk.off = total + (piece_off[pi] - run.off);
                        mk.span = open_span_;
                        s.marks.push_back(mk);
                    }
                    first = false;
                }
                if (run.p0 == run.p1) s.marks.push_back(mk);
            } else {
                s.marks.push_back(mk);
            }
            if (runs.size() > 1) {
                if (k) check(rh_memcpy_d2d(acc_.get() + total, cur, k * sizeof(float), stream_), "rh_memcpy_d2d");
                result = acc_.get();
            } else {
                result = cur;
            }
            total += k;
            if (ends) break;
        }
        n = total;
        // ... and the format of the sample behind the block (the next span's, if the block ended on a boundary)
        {
            std::uint16_t nch = cur_in_ch_;
            std::uint32_t nrate = cur_in_rate_;
            std::optional<std::size_t> nspan = std::nullopt;
            if ((span_aware_ || follow_spans_) && !ends) {
                if (reader_.peek(nch, nrate)) reader_.source_format(nch, nrate, nspan);
            }
            if (ends && follow_spans_) nspan = reader_.ended() ? up_->current_span_len() : std::optional<std::size_t>(0);  // a source that has given everything stands where the question reaches it: Some(0) from a SamplesBuffer (buffer.rs:76-82)
            const bool as_built = fixed || (nch == in_ch0() && nrate == in_rate0());
            s.next = FormatMark{n, as_built ? ch_ : nch, as_built ? rate_ : nrate, counted ? nspan : std::nullopt};
        }
        if (device_out_) {  // the block stays on the device, in the slot's own buffer (a_ / b_ belong to the next block's stages)
            s.dev.reset(std::max(cap, n + 64));
            if (s.taken_pending) {  // the consumer's copies out of this slot's previous block
                check(rh_stream_wait_event(stream_, s.taken), "rh_stream_wait_event");
                s.taken_pending = false;
            }
            if (n) check(rh_memcpy_d2d(s.dev.get(), result, n * sizeof(float), stream_), "rh_memcpy_d2d");
        } else if (n) {
            s.out.reset(std::max(cap, n + 64));
            check(rh_memcpy_d2h_async(s.out.get(), result, n * sizeof(float), stream_), "rh_memcpy_d2h_async");
            timing_.d2h_samples += n;
        }
        s.n = n;
        s.last = ends;
    }
    /// The format of the sample next() returns next.
    FormatMark format_at_cursor() const {
        if (!started()) {  // nothing pulled yet: the upstream's own answers
            return FormatMark{0, ch_, rate_, spans_stay_in_place() ? up_->current_span_len() : std::nullopt};
        }
        const Slot &sl = cur_slot();
        const std::size_t pos = position();
        if (pos >= sl.n || sl.marks.empty()) return sl.next;
        const FormatMark *m = &sl.marks.front();
        for (const FormatMark &k : sl.marks)
            if (k.off <= pos) m = &k;
        return *m;
    }

private:
    struct Ctx {
        float *out;
        const float *in;
        std::size_t n, out_cap;
        bool flush;
        rh_stream stream;
        bool end = false;  // set by a stage: the stream ends with this block although the upstream has more (take_duration)
    };
    struct Stage {
        std::function<std::size_t(Ctx &)> run;
        std::function<std::size_t(std::size_t)> bound;
        bool seekable = true;                        // false: the adapter answers SeekError::NotSupported (mix.rs:116-120)
        std::function<void(Nanos)> on_seek = nullptr;  // what the adapter does to its own state after its input was sought
        int span_rule = 0;  // what current_span_len() is behind the adapter: 0 the input's (one sample out per sample in), 1 None (Mix, the converters), 2 the input's with another sample count
        // The upstream's format changes between two spans.  0: the adapter does not care (amplify).  1: on_format does what rodio's adapter does at
        // the boundary.  2: not mirrored (loud error).  3: the adapter meets the spans itself and gives one format out (UniformSourceIterator): the
        // adapters behind it never see a change.
        int fmt = 2;
        std::function<void(std::uint16_t, std::uint32_t)> on_format = nullptr;
        // span_rule 2, where the arithmetic is mirrored (take_duration, delay): rodio's answer behind the adapter from its input's answer and
        // the number of samples the adapter has emitted; and how many of its input's samples it has taken by then (span_behind())
        std::function<std::optional<std::size_t>(std::optional<std::size_t>, std::uint64_t)> span_fn = nullptr;
        std::function<std::uint64_t(std::uint64_t)> span_in_pos = nullptr;
        bool span_keeps_count = false;  // ... and it hands on one sample per sample until it ends the stream (take_duration): the input's spans lie where they lay
        // total_duration() behind the adapter from its input's (null: the input's, amplify.rs:95-97 and the like)
        std::function<std::optional<Nanos>(std::optional<Nanos>)> dur_fn = nullptr;
    };
    // size_hint(): what every adapter makes of its input's answer.  `in(q)`: the input's size_hint() once q of ITS samples have been taken;
    // `emitted`: samples the adapter has emitted (both counted from the start of the stream, or from the last seek).  The state lives apart from
    // the chain (shared): a consumer that keeps asking after the chain has been retired (GpuMixer) holds on to it.
    using HintAt = std::function<SizeHint(std::uint64_t)>;
    struct HintStage {
        std::function<SizeHint(const HintAt &in, std::uint64_t emitted)> fn = nullptr;  // null: in(in_pos(emitted))
        std::function<std::uint64_t(std::uint64_t)> in_pos = nullptr;                   // null: one sample in per sample out
    };
    struct HintState {
        std::vector<HintStage> stages;
        detail::HintLog log;         // the upstream's answers, by the sample at which it was asked
        SizeHint before;             // ... and its answer while nothing has been pulled
        std::uint64_t out_base = 0, in_base = 0, skip = 0;  // (after a seek) samples the chain had served / the upstream had given by then; samples dropped to keep the consumer's channel
        SizeHint at(std::uint64_t emitted) const {
            const std::uint64_t rel = (emitted > out_base ? emitted - out_base : 0) + skip;
            HintAt h = [this](std::uint64_t q) { return log.empty() ? before : log.at(in_base + q); };
            for (const HintStage &st : stages) {
                if (st.fn) h = [prev = std::move(h), &st](std::uint64_t x) { return st.fn(prev, x); };
                else if (st.in_pos) h = [prev = std::move(h), &st](std::uint64_t x) { return prev(st.in_pos(x)); };
            }
            return h(rel);
        }
    };
    template <class T>
    struct Handle {
        T *p = nullptr;
        void (*destroy)(T *) = nullptr;
        ~Handle() {
            if (p && destroy) destroy(p);
        }
    };
    template <class F>
    GpuSource &push(F run) {
        stages_.push_back(Stage{std::function<std::size_t(Ctx &)>(run), [](std::size_t n) { return n; }, true, nullptr, 0, 2, nullptr});
        hint_->stages.emplace_back();
        last_kind_ = 0;
        return *this;
    }
    template <class F, class B>
    GpuSource &push(F run, B bound) {
        stages_.push_back(Stage{std::function<std::size_t(Ctx &)>(run), std::function<std::size_t(std::size_t)>(bound), true, nullptr, 2, 2, nullptr});
        hint_->stages.emplace_back();
        last_kind_ = 0;
        return *this;
    }
    GpuSource &duration(std::function<std::optional<Nanos>(std::optional<Nanos>)> f) {  // for the stage pushed last
        stages_.back().dur_fn = std::move(f);
        return *this;
    }
    GpuSource &hints(std::function<SizeHint(const HintAt &, std::uint64_t)> f) {  // for the stage pushed last
        hint_->stages.back().fn = std::move(f);
        return *this;
    }
    GpuSource &any_format() {  // for the stage pushed last
        stages_.back().fmt = 0;
        return *this;
    }
    GpuSource &on_format(std::function<void(std::uint16_t, std::uint32_t)> f) {
        stages_.back().fmt = 1;
        stages_.back().on_format = std::move(f);
        return *this;
    }
    GpuSource &on_seek(std::function<void(Nanos)> f) {  // for the stage pushed last (a second call adds to the first)
        if (stages_.back().on_seek) {
            f = [a = std::move(stages_.back().on_seek), b = std::move(f)](Nanos p) {
                a(p);
                b(p);
            };
        }
        stages_.back().on_seek = std::move(f);
        return *this;
    }
    GpuSource &not_seekable() {
        stages_.back().seekable = false;
        return *this;
    }
    GpuSource &spans(int rule) {
        stages_.back().span_rule = rule;
        return *this;
    }
    // every adapter hands on one sample per sample (until one of them ends the stream): the spans of the input lie at the same samples of the output
    bool spans_stay_in_place() const {
        for (const Stage &st : stages_)
            if (st.span_rule != 0 && !st.span_keeps_count) return false;
        return true;
    }
    GpuSource &keeps_the_sample_count() {
        stages_.back().span_keeps_count = true;
        return *this;
    }
    GpuSource &span_arithmetic(std::function<std::optional<std::size_t>(std::optional<std::size_t>, std::uint64_t)> fn, std::function<std::uint64_t(std::uint64_t)> in_pos) {
        stages_.back().span_fn = std::move(fn);
        hint_->stages.back().in_pos = in_pos;
        stages_.back().span_in_pos = std::move(in_pos);
        return *this;
    }
    /// rodio's answer to current_span_len() behind the first `upto` adapters at sample `pos` of their output, where the last of them that
    /// does anything to the spans is a take_duration / delay / channel_volume.  TakeDuration answers with what its duration still admits
    /// unless its input's span is shorter -- Some(..) over an input that says None too, and Some(0) once it is spent (take.rs:176-195);
    /// Delay adds the silence it still owes to its input's answer (delay.rs:94-98); ChannelVolume hands its input's answer on
    /// (channel_volume.rs:103-105).  Computed from the adapters' sample counts: every adapter maps the samples it has emitted to the
    /// samples it has taken, down to the sample of the upstream the question reaches (whose answers the chain keeps by sample
    /// position: span_log_).  An adapter with rule 2 and no such mapping would be refused, loudly (there is none at present).
    std::optional<std::size_t> span_behind(std::size_t upto, std::uint64_t pos) const {
        std::size_t first = 0;
        for (std::size_t k = 0; k < upto; ++k)
            if (stages_[k].span_rule == 1) first = k + 1;
        const bool upstream_asked = first == 0 && up_->current_span_len().has_value();  // the adapters' questions reach an upstream that reports spans
        std::vector<std::uint64_t> at(upto + 1, 0);  // at[k + 1]: samples adapter k has emitted when the chain has emitted `pos`; at[first]: samples taken from what lies in front
        at[upto] = pos;
        bool counts = false;
        for (std::size_t k = first; k < upto; ++k) counts = counts || stages_[k].span_fn;
        for (std::size_t k = upto; k-- > first;) {
            const Stage &st = stages_[k];
            if (st.span_rule == 2 && !st.span_fn)  // (an adapter that changes the sample count and says nothing about how: none at present)
                throw Error(RH_ERR_UNSUPPORTED, "GpuSource::current_span_len behind an adapter whose span arithmetic is not mirrored");
            at[k] = st.span_in_pos ? st.span_in_pos(at[k + 1]) : at[k + 1];
        }
        // Over an upstream that reports spans the adapters ask IT, wherever the consumer asks them: the chain keeps the upstream's answers
        // by sample position (span_log_: what it answered when each of its spans opened -- a SamplesBuffer, a decoder's packets: sources
        // whose answer holds for the whole span), and at[0] is the upstream sample the question reaches.
        std::optional<std::size_t> ans = upstream_asked ? upstream_answer_at(at[0]) : std::nullopt;
        if (!counts) return ans;
        for (std::size_t k = first; k < upto; ++k)
            if (stages_[k].span_fn) ans = stages_[k].span_fn(ans, at[k + 1]);
        return ans;
    }
    std::optional<std::size_t> upstream_answer_at(std::uint64_t q) const {
        if (span_log_.empty()) return up_->current_span_len();  // nothing pulled yet: the upstream itself
        if (up_ended_ && q >= in_total_) return up_->current_span_len();  // it has given everything, and stands where the question reaches it: Some(0) from a SamplesBuffer (buffer.rs:76-82)
        std::size_t i = 0;
        while (i + 1 < span_log_.size() && span_log_[i + 1].first <= q) ++i;
        span_log_.erase(span_log_.begin(), span_log_.begin() + (std::ptrdiff_t)i);  // (the questions only move forward)
        return span_log_.front().second;
    }
    // Adapters that work on whole frames and carry state (BltFilter, Limit) over an input that may break off inside a frame: at the end of a
    // span (rodio's adapters run sample by sample -- blt.rs:431-451, limit.rs:927-988 -- and their channel position simply goes on into the
    // next span) or of the stream.  `kernel(out, in, frames, state)` runs whole frames from `state`.  The samples of a broken frame are
    // EMITTED AT ONCE, from a copy of the state with the frame completed by zeros (what follows a sample in time does not reach back to
    // it) -- one sample out per sample in, so the format marks of the block stay where rodio reports them -- and KEPT, so that the real
    // state meets them again in front of the samples that complete the frame; on_format / on_seek drop them where rodio starts afresh.
    // Adapters that regroup the flat stream into frames of `from` samples and keep no state between them (ChannelCountConverter,
    // ChannelVolume, the bare SampleRateConverter): a block may stop inside a frame -- a spanned upstream whose spans do, adapters in
    // front that emit an open frame's samples at once -- and rodio's iterators simply go on with the next sample.  Here the samples of the
    // open frame wait for the next block; the end of the stream hands them to `fn` as they are.  fn(in, n) -> samples written to c.out.
    struct Regroup {
        detail::DeviceBuf part, tin;
        std::size_t n = 0;
    };
    template <class F>
    static std::size_t run_grouped(Ctx &c, std::uint16_t from, Regroup &rg, F fn) {
        const std::size_t carried = rg.n, total = carried + c.n;
        const float *in = c.in;
        if (carried) {
            rg.tin.reset(total + from);
            check(rh_memcpy_d2d(rg.tin.get(), rg.part.get(), carried * sizeof(float), c.stream), "rh_memcpy_d2d");
            if (c.n) check(rh_memcpy_d2d(rg.tin.get() + carried, c.in, c.n * sizeof(float), c.stream), "rh_memcpy_d2d");
            in = rg.tin.get();
        }
        const std::size_t rest = c.flush ? 0 : total % from;
        const std::size_t k = fn(in, total - rest);
        if (rest) {
            rg.part.reset(from);
            check(rh_memcpy_d2d(rg.part.get(), in + total - rest, rest * sizeof(float), c.stream), "rh_memcpy_d2d");
        }
        rg.n = rest;
        return k;
    }
    struct FrameCarry {
        detail::DeviceBuf part, pad, st2, tin, tout;
        std::size_t n = 0;     // samples of an open frame, already emitted from a copy of the state and kept for the real one
        std::size_t lead = 0;  // channels of the open frame that are done for good (commit_open_frame): the next samples go on at this channel
    };
    // One frame of which only the channels [c0, c0 + k) are real (the others zeros), through a copy of the state; the outputs of those channels
    // go to `out`, and -- `keep` -- their state entries (per_ch floats a channel, one channel after the other: rh_biquad's layout) back into the
    // real state.  The channels are independent in the filters; the limiter never needs `keep` (its coefficients do not change with the rate).
    template <class K>
    static void run_part_of_a_frame(FrameCarry &fc, std::uint16_t ch, std::size_t c0, std::size_t k, const float *in, float *out, float *state, std::size_t state_floats, bool keep, rh_stream s,
                                    K kernel) {
        fc.pad.reset(2u * ch);
        fc.st2.reset(state_floats);
        check(rh_memset(fc.pad.get(), 0, ch * sizeof(float), s), "rh_memset");
        check(rh_memcpy_d2d(fc.pad.get() + c0, in, k * sizeof(float), s), "rh_memcpy_d2d");
        check(rh_memcpy_d2d(fc.st2.get(), state, state_floats * sizeof(float), s), "rh_memcpy_d2d");
        kernel(fc.pad.get() + ch, fc.pad.get(), 1, fc.st2.get());
        if (out) check(rh_memcpy_d2d(out, fc.pad.get() + ch + c0, k * sizeof(float), s), "rh_memcpy_d2d");
        if (keep) {
            const std::size_t per_ch = state_floats / ch;
            check(rh_memcpy_d2d(state + per_ch * c0, fc.st2.get() + per_ch * c0, per_ch * k * sizeof(float), s), "rh_memcpy_d2d");
        }
    }
    // The adapter's parameters are about to change (a filter's coefficients at a new sample rate) while a frame is open: the samples that were
    // emitted from a copy of the state meet the OLD parameters for good -- their channels' state entries go into the real state -- and the
    // frame goes on at the next channel with the new ones (rodio's filter runs sample by sample: blt.rs:431-451; the recreated applier,
    // blt.rs:119-141, simply meets the next sample).
    template <class K>
    static void commit_open_frame(FrameCarry &fc, std::uint16_t ch, float *state, std::size_t state_floats, rh_stream s, K kernel) {
        if (!fc.n) return;
        run_part_of_a_frame(fc, ch, fc.lead, fc.n, fc.part.get(), nullptr, state, state_floats, true, s, kernel);
        fc.lead += fc.n;  // (< ch: an open frame is never whole)
        fc.n = 0;
    }
    template <class K>
    static std::size_t run_framewise(Ctx &c, std::uint16_t ch, FrameCarry &fc, float *state, std::size_t state_floats, K kernel) {
        const float *cin = c.in;
        float *cout = c.out;
        std::size_t cn = c.n;
        if (fc.lead) {  // a frame whose first channels are done (see commit_open_frame): its next samples one part at a time, straight into the state
            const std::size_t k = std::min<std::size_t>(ch - fc.lead, cn);
            if (k) run_part_of_a_frame(fc, ch, fc.lead, k, cin, cout, state, state_floats, true, c.stream, kernel);
            fc.lead = fc.lead + k == ch ? 0 : fc.lead + k;
            cin += k, cout += k, cn -= k;
            if (fc.lead) return c.n;  // (the block ended inside that frame)
        }
        const std::size_t carried = fc.n, total = carried + cn, frames = total / ch, rem = total % ch;
        const float *in = cin;
        float *out = cout;
        if (carried) {  // (rare: once behind every span that ends inside a frame)
            fc.tin.reset(total + ch);
            fc.tout.reset(total + ch);
            check(rh_memcpy_d2d(fc.tin.get(), fc.part.get(), carried * sizeof(float), c.stream), "rh_memcpy_d2d");
            if (cn) check(rh_memcpy_d2d(fc.tin.get() + carried, cin, cn * sizeof(float), c.stream), "rh_memcpy_d2d");
            in = fc.tin.get();
            out = fc.tout.get();
        }
        if (frames) kernel(out, in, frames, state);
        if (carried && frames * ch > carried) check(rh_memcpy_d2d(cout, out + carried, (frames * ch - carried) * sizeof(float), c.stream), "rh_memcpy_d2d");
        if (rem) {
            const std::size_t before = frames ? 0 : carried;  // of the open frame's samples: emitted by an earlier call
            if (rem > before) {
                // (all of the open frame's samples through the copy again: the ones emitted before give the same outputs, only the new ones are taken)
                fc.pad.reset(2u * ch);
                run_part_of_a_frame(fc, ch, 0, rem, in + frames * ch, nullptr, state, state_floats, false, c.stream, kernel);
                check(rh_memcpy_d2d(cout + (frames * ch + before - carried), fc.pad.get() + ch + before, (rem - before) * sizeof(float), c.stream), "rh_memcpy_d2d");
                fc.part.reset(ch);
                check(rh_memcpy_d2d(fc.part.get(), in + frames * ch, rem * sizeof(float), c.stream), "rh_memcpy_d2d");
            }
        }
        fc.n = rem;
        return c.n;
    }
    std::shared_ptr<detail::DeviceBuf> state(std::size_t floats) {
        auto st = std::make_shared<detail::DeviceBuf>(floats);
        check(rh_memset(st->get(), 0, floats * sizeof(float), stream_), "rh_memset");
        return st;
    }
    GpuSource &blt(int kind, std::uint32_t freq, float q) {  // blt.rs:502-544,558-560
        const std::uint16_t ch = ch_;
        struct Applier {
            float co[5];
            bool exact;
        };
        const int mode = filter_mode_;
        auto make = [kind, freq, q, mode](std::uint32_t rate) {
            Applier a;
            check(rh_biquad_coeffs(kind, freq, q, rate, a.co), "rh_biquad_coeffs");
            a.exact = mode == 1 || (mode == 0 && !rh_filter_scan_ok(kind, freq, q, rate));  // the filter contract (rodio_hip.h)
            return a;
        };
        auto ap = std::make_shared<Applier>(make(rate_));
        auto st = state(4u * ch);
        auto fc = std::make_shared<FrameCarry>();
        const bool behind_uniform = last_kind_ == 1;
        struct Mark {  // (runs when the builder expression is complete: the adapter pushed last is this filter)
            GpuSource *g;
            bool on;
            ~Mark() { g->last_kind_ = on ? 2 : 0; }
        } mark{this, behind_uniform};
        return push([=](Ctx &c) {
            return run_framewise(c, ch, *fc, st->get(), 4u * ch, [&](float *out, const float *in, std::size_t frames, float *state) {
                check(rh_biquad(out, in, frames, ch, 1, ap->co, state, ap->exact ? 0 : 1, c.stream), "rh_biquad");
            });
        }).on_seek([st, ch, fc, sm = stream_](Nanos) {  // blt.rs:350-377
            check(rh_memset(st->get(), 0, 4u * ch * sizeof(float), sm), "rh_memset");
            fc->n = fc->lead = 0;
        })
            .on_format([ap, make, ch, fc, st, sm = stream_](std::uint16_t new_ch, std::uint32_t rate) {
                // blt.rs:119-141: `recreate_applier` for the new rate; the state stays.  (A new channel COUNT: the branch that would rebuild the
                // filter compares the count with itself, blt.rs:128, so rodio goes on filtering frames of the new layout with the state of
                // the old one -- channels meet each other's history.  Not mirrored.)
                if (new_ch != ch) throw Error(RH_ERR_UNSUPPORTED, "GpuSource: a filter across a change of the channel count (" + std::to_string(ch) + " -> " + std::to_string(new_ch) + ")");
                // The span in front of the change may have ended inside a frame (source/mod.rs:169-178 asks sources for whole frames; queues of
                // odd sounds exist): rodio's filter goes on at the next channel with the new coefficients, the channels in front of it have met
                // the old ones -- commit_open_frame.
                commit_open_frame(*fc, ch, st->get(), 4u * ch, sm, [&](float *out, const float *in, std::size_t frames, float *state) {
                    check(rh_biquad(out, in, frames, ch, 1, ap->co, state, ap->exact ? 0 : 1, sm), "rh_biquad");
                });
                *ap = make(rate);
            });
    }

    BoxSource up_;
    std::size_t block_frames_;
    std::uint16_t ch_ = 0;
    std::uint32_t rate_ = 0;
    std::uint16_t cur_in_ch_ = 0;    // the upstream's format as of the last sample pulled
    std::uint32_t cur_in_rate_ = 0;
    std::uint16_t in_ch0_ = 0;       // ... and when the chain was built (ch_ / rate_ are the chain's output format for THAT input)
    std::uint32_t in_rate0_ = 0;
    std::uint16_t in_ch0() const { return in_ch0_; }
    std::uint32_t in_rate0() const { return in_rate0_; }
    std::optional<std::size_t> open_span_;  // what current_span_len() answered for the span that is open
    // (chains with take_duration / delay over an upstream that reports spans: span_behind()) the upstream's answers by the sample at which each
    // of its spans opened, the samples pulled from it so far, and whether it has ended
    mutable std::deque<std::pair<std::uint64_t, std::optional<std::size_t>>> span_log_;
    std::uint64_t in_total_ = 0;
    bool up_ended_ = false, span_log_on_ = false;
    std::shared_ptr<HintState> hint_ = std::make_shared<HintState>();  // size_hint(): see HintState
    int last_kind_ = 0;  // the adapter pushed last: 1 a `uniform`, 2 a filter right behind a `uniform`, 0 anything else
    friend class GpuMixer;
    std::uint64_t pulled_total_ = 0;                                   // samples pulled from the upstream, whatever was sought in between
    std::uint16_t block_min_ch_ = 0;        // the fewest channels / the lowest rate among the pieces of the block being enqueued (0: none)
    std::uint32_t block_min_rate_ = 0;
    bool follow_spans_ = false, follow_known_ = false;  // the upstream reports spans: it is pulled span by span (asked once, at the first block)
    detail::DeviceBuf acc_;          // a block whose samples have more than one format: the runs' outputs, back to back
    std::vector<Stage> stages_;
    detail::SpanReader reader_{nullptr};
    std::vector<detail::Piece> pieces_;  // the spans of the block being enqueued
    bool span_aware_ = false;
    int filter_mode_ = 0;  // 0: by the filter contract, per filter; 1: reference order throughout; 2: time-parallel throughout
    detail::DeviceBuf a_, b_;
    bool scan_kernels_ = false;  // the chain launches handle-less scan kernels: their failure word is read per block
    bool stale_frame_ = false;   // ended_with_a_stale_frame()
    bool may_cut_ = false;       // an adapter of the chain can make the stream end inside a frame (may_end_inside_a_frame())
};

// ---------------------------------------------------------------- GpuMixer: the fused mixer path ----
/// What rodio spells
///     let (mixer, mixed) = mixer::mixer(nz!(2), rate);
///     mixer.add(UniformSourceIterator::new(src.amplify(g), nz!(2), rate).low_pass(f));   // per source
/// as ONE source: every block is one launch of the fused kernel (resample + filter + ordered sum), each source
/// keeping its own converter position and filter state across blocks; sources end when they end.
/// Every source is pulled the way Mixer::add's UniformSourceIterator pulls it (uniform.rs:50-97).  Sources whose
/// current_span_len() is None (generators, anything behind a UniformSourceIterator) are continuous streams: they go
/// straight into the fused kernel (resample + filter + sum in one pass, one fused stream per input rate).  If a source
/// reports SPANS (SamplesBuffer, Buffered, the decoders), rodio converts it span by span -- a fresh converter every
/// min(span, 32768) samples, each span's last frame verbatim, rate and layout free to change between spans -- and so
/// does the generation it joins: its sources are converted segment by segment on the device (rh_uniform_segments, one
/// launch per block for all of them, any rates and layouts) and the fused kernel filters and mixes the converted rows.
///
/// add() may be called at any time (Mixer::add, mixer.rs:58-66).  Sources added before the first next() start
/// with the stream.  Sources added later start at the next FRAME of the output, as mixer.rs:175-183 admits them
/// (late_join: the blocks already mixed beyond that frame -- at most two -- are patched on the device): they form
/// a new *generation* with its own clock and its own fused stream, and the generations' mixes are summed in
/// insertion order by rh_mix_sum.  An empty mixer is an ended stream (there is nothing to pull); it resumes after
/// a later add() at the channel position MixerSource would be at.
class GpuMixer : public detail::BlockPump {
public:
    /// The filter a source carries into the mixer: `mixer.add(UniformSourceIterator::new(src, ch, rate).low_pass(200))` for one,
    /// `.high_pass(300)` for the next, none for a third (source/mod.rs:686-721: every source has its own adapters).
    struct Filter {
        int kind = -1;           // -1 none, 0 low_pass, 1 high_pass
        std::uint32_t freq = 0;
        float q = 0.5f;          // rodio's low_pass() / high_pass() (blt.rs:11-24)
        static Filter none() { return Filter{}; }
        static Filter low_pass(std::uint32_t f, float q = 0.5f) { return Filter{0, f, q}; }
        static Filter high_pass(std::uint32_t f, float q = 0.5f) { return Filter{1, f, q}; }
        bool operator==(const Filter &o) const { return kind == o.kind && (kind < 0 || (freq == o.freq && q == o.q)); }
    };
    struct Options {
        std::size_t block_frames = 1u << 15;  // input frames pulled per source and block
        int filter_kind = -1;                 // the filter of sources added WITHOUT one of their own: -1 none, 0 low_pass, 1 high_pass (q = 0.5, blt.rs:11-24)
        std::uint32_t filter_freq = 0;
        float filter_q = 0.5f;
        std::uint32_t frames_per_lane = 0;    // 0 = the library's choice
        unsigned host_threads = 0;            // threads that pull the sources of a block (0 = min(cores, 16); 1 = the caller alone)
        // THE FILTER CONTRACT (rodio_hip.h, rh_filter_scan_ok).  The fused kernel evaluates the filters time-parallel, which stays within
        // 1e-5 of rodio's own f32 recurrence only where that recurrence does not amplify its rounding noise past it (low_pass >= 100 Hz,
        // high_pass >= 600 Hz at 48 kHz, full-scale sources).  true (default): a source whose filter lies outside gets a chain of its own --
        // amplify -> UniformSourceIterator -> the filter in rodio's operation order, bit for bit -- and enters the mix unfiltered, on the
        // device: rodio's samples, at the price of per-source launches.  false: every filter stays in the fused kernel (closer to the
        // exact response than rodio is; at low cutoffs further than 1e-5 from rodio).
        bool reference_exact_filters = true;
        // Mixers of more than two channels: true = every source a chain of its own, whatever it is (the form of rounds 5; a comparison aid).
        // false (default): a generation of plain continuous sources without filters converts and sums a block in one launch (rh_wide_mix_block)
        bool wide_chains = false;
    };
    /// mixer::mixer(channels, sample_rate) (mixer.rs:25).  One and two channels: the sources are mixed as stereo frames by the fused
    /// kernel, and a mono mixer keeps channel 0 of the mix (ChannelCountConverter(2 -> 1) commutes with the sum: channels.rs:57-85).
    /// MORE than two channels (a 5.1 mix): sources that join together and are all plain continuous ones without a filter (decoded assets,
    /// generators: current_span_len() == None) are converted and summed a block at a time in ONE launch (rh_wide_mix_block: Amplify ->
    /// SampleRateConverter -> ChannelCountConverter per source and the ordered sum, any rates and layouts side by side).  Anything else --
    /// a filter, spans, a GpuSource chain -- becomes a chain of its own on the device, [amplify] -> UniformSourceIterator(channels, rate)
    /// -> [its filter] (GpuSource, the blocks staying in device memory), and the mixer adds the chains' blocks in insertion order
    /// (rh_mix_sum): rodio's Mixer::add + MixerSource::next for any layout, bit for bit either way.
    GpuMixer(std::uint16_t channels, std::uint32_t sample_rate, Options opt) : rate_(sample_rate), opt_(opt), out_ch_(channels), qch_(channels > 2 ? channels : 2) {
        if (!sample_rate || !channels) throw std::invalid_argument("channels and sample_rate are NonZero in rodio");
        if (!opt_.block_frames) opt_.block_frames = 1;
        check(rh_stream_create(&copy_stream_), "rh_stream_create");
    }
    GpuMixer(std::uint32_t sample_rate, Options opt) : GpuMixer(2, sample_rate, opt) {}
    explicit GpuMixer(std::uint32_t sample_rate) : GpuMixer(2, sample_rate, Options()) {}
    ~GpuMixer() override {
        (void)rh_stream_synchronize(copy_stream_);
        (void)rh_stream_synchronize(stream_);
        gens_.clear();
        reaper_.reset();  // joins: generations that were retired are gone before the streams are
        (void)rh_stream_destroy(copy_stream_);
    }
    /// Mixer::add (mixer.rs:58-66), with the source's volume.  Any source: mono sources form fused streams of their own (the kernel
    /// reads mono frames, the mono mix becomes stereo once per block); a channel count above 2 is staged in the source's own
    /// layout and converted on the device in front of the fused launch (ChannelCountConverter, rh_channels_convert); a rate more
    /// than 4.5 times the mixer's first runs through the GPU SampleRateConverter adapter, still one pull chain.
    void add(BoxSource src, float gain = 1.0f) { add(std::move(src), gain, Filter{opt_.filter_kind, opt_.filter_freq, opt_.filter_q}); }
    /// ... with the source's own filter (behind its UniformSourceIterator, at the mixer's rate).  Sources of one filter share a fused
    /// stream -- one launch per block for all of them, summed first where they run together -- and the streams' mixes are added.
    void add(BoxSource src, float gain, Filter filter) {
        if (!src) throw std::invalid_argument("source");
        std::uint16_t ch = src->channels();
        const std::uint32_t from = src->sample_rate();
        if (!ch || !from) throw std::invalid_argument("channels and sample_rate are NonZero in rodio");
        if (filter.kind > 1) throw std::invalid_argument("filter kind");
        if (wide()) {  // a mixer of more than two channels: the source's own chain, the mixer only sums --
            Src item;
            item.up = std::move(src);
            item.gain = gain;
            item.ch = ch;
            item.filt = filter;
            // ... unless the source is a plain continuous one without a filter: those stay as they are, and a generation of such sources
            // converts and sums a block in ONE launch (rh_wide_mix_block; start_stream_wide decides)
            item.fusew = !opt_.wide_chains && filter.kind < 0 && !dynamic_cast<GpuSource *>(item.up.get()) && !item.up->current_span_len().has_value() && !ratio_beyond_u32(from, rate_);
            if (!item.fusew) make_wide(item);
            if (joins_a_running_mix()) late_join(std::move(item));
            else pending_.push_back(std::move(item));
            return;
        }
        // A source whose spans can end inside a frame (uniform.rs:56: 32768 is no multiple of 3, 5, 6, 7 channels) may end its converted
        // stream inside an output frame; the fused kernel filters whole frames, so such a source takes its filter along in a chain of its
        // own, which filters exactly the samples rodio's BltFilter sees (blt.rs:431-451), and enters the mix unfiltered.
        if (GpuSource *gs = dynamic_cast<GpuSource *>(src.get()); gs && completes_with_uniform(*gs, gain, filter)) {  // (see add(chain): the same, whoever owns the chain)
            src.release();
            add(std::unique_ptr<GpuSource>(gs), gain, filter);
            return;
        }
        // (... and one whose span at hand does not hold whole frames -- a SamplesBuffer of an odd number of stereo samples, packets of 37.)
        const std::optional<std::size_t> span_now = src->current_span_len();
        const bool cuts = span_now.has_value() && ((32768u % ch) != 0 || (*span_now % ch) != 0);
        const bool may_cut = cuts && filter.kind >= 0;
        if ((filter.kind >= 0 && opt_.reference_exact_filters && !rh_filter_scan_ok(filter.kind, filter.freq, filter.q, rate_)) || may_cut) {
            // outside the filter contract: the source's own chain, the filter in the reference's order, the mixer only sums
            auto chain = std::make_unique<GpuSource>(std::move(src), opt_.block_frames);
            if (filter.kind >= 0 && opt_.reference_exact_filters && !rh_filter_scan_ok(filter.kind, filter.freq, filter.q, rate_)) chain->exact_filters(true);
            if (gain != 1.0f) chain->amplify(gain);
            chain->uniform(out_ch_, rate_);
            if (filter.kind == 0) chain->low_pass_with_q(filter.freq, filter.q);
            else if (filter.kind == 1) chain->high_pass_with_q(filter.freq, filter.q);
            add(std::move(chain), 1.0f, Filter::none());
            return;
        }
        Src item;
        item.up = std::move(src);
        item.gain = gain;
        item.ch = ch;
        item.filt = filter;
        if (joins_a_running_mix()) late_join(std::move(item));  // mixer.rs:175-183: admitted at the next frame
        else pending_.push_back(std::move(item));    // starts with the stream (or resumes an ended one)
    }
private:
    // A chain whose stream can end inside a frame and is not yet what Mixer::add would make of it (see add(chain))
    bool completes_with_uniform(const GpuSource &gs, float gain, const Filter &filter) const {
        // (... or whose spans are an adapter's own -- take_duration's Some(what it still admits), Some(0) in front of the silence that completes a cut
        // frame --: its own `uniform` computes them where rodio's iterator asks, and the result stays on the device; pulled through the host the
        // mixer's reader would ask the same questions)
        if (wide() || gs.started()) return false;
        if (gs.answers_with_adapter_spans()) return true;
        return gs.may_end_inside_a_frame() && !(gs.channels() == out_ch_ && gs.sample_rate() == rate_ && gain == 1.0f && filter.kind < 0);
    }

public:
    /// A GpuSource chain handed to the mixer by value, as rodio's adapters are (`mixer.add(src.reverb(..).limit(..))`, amplify.rs:19-22,
    /// mixer.rs:58-72): its blocks stay in device memory and the mixer takes them device-to-device -- the chain's output never
    /// crosses to the host and back.  (A chain that is not stereo at a rate the fused converter takes, or one that has already
    /// started, is pulled like any other source: the same samples, through the host.)
    void add(std::unique_ptr<GpuSource> chain, float gain = 1.0f) { add(std::move(chain), gain, Filter{opt_.filter_kind, opt_.filter_freq, opt_.filter_q}); }
    void add(std::unique_ptr<GpuSource> chain, float gain, Filter filter) {
        if (!chain) throw std::invalid_argument("source");
        GpuSource *const gs = chain.get();
        if (completes_with_uniform(*gs, gain, filter)) {
            // The fused streams convert whole frames; what rodio's UniformSourceIterator makes of a stream that ends inside one (the cut
            // tail of sample_rate.rs:174-200 / channels.rs:57-85) is what `uniform` makes of it: the chain is completed to
            // `chain.amplify(gain) -> UniformSourceIterator(channels, rate) [-> filter]` -- Mixer::add's own wrapping, mixer.rs:58-66 -- and
            // enters the mix as it is, sample for sample (a stream at the mixer's rate is tracked to its last sample: Gen::track).
            if (gain != 1.0f) chain->amplify(gain);
            chain->uniform(out_ch_, rate_);
            if (filter.kind == 0) chain->low_pass_with_q(filter.freq, filter.q);
            else if (filter.kind == 1) chain->high_pass_with_q(filter.freq, filter.q);
            add(std::move(chain), 1.0f, Filter::none());
            return;
        }
        if (wide() && !gs->started()) {  // the chain goes on as the source's chain of a wide mixer: amplify -> uniform(channels, rate) -> filter behind what it has
            Src item;
            item.ch = gs->channels();
            item.up = std::move(chain);
            item.dev = gs;
            item.gain = gain;
            item.filt = filter;
            make_wide(item);
            if (joins_a_running_mix()) late_join(std::move(item));
            else pending_.push_back(std::move(item));
            return;
        }
        // (a chain that forwards its input's spans -- a SamplesBuffer behind amplify / a filter -- is converted span by span like any spanned
        // source, rodio's UniformSourceIterator restarting at every span: that path pulls through the host)
        const bool on_device = !wide() && gs->channels() == 2 && !gs->started() && !fused_ratio_unsupported(gs->sample_rate(), rate_) && gs->sample_rate() != 0 && !gs->current_span_len().has_value();
        if (!on_device) {
            add(BoxSource(std::move(chain)), gain, filter);
            return;
        }
        gs->keep_blocks_on_device();
        Src item;
        item.up = std::move(chain);
        item.dev = gs;
        item.gain = gain;
        item.ch = 2;
        item.filt = filter;
        device_chains_ = true;
        if (joins_a_running_mix()) late_join(std::move(item));
        else pending_.push_back(std::move(item));
    }
    // MixerSource::next advances its channel position on every call, also on the ones that return None (mixer.rs:120-136),
    // and admits pending sources only at channel 0: an ended mixer that gets a new source after an odd number of calls
    // returns one more None before the source's first sample.  `calls_` is that position.
    std::optional<float> next() override {
        resume_ok_ = (calls_ % out_ch_) == 0;
        ++calls_;
        return detail::BlockPump::next();
    }
    std::size_t read(float *dst, std::size_t n) override {
        resume_ok_ = (calls_ % out_ch_) == 0;  // blocks hold whole frames: the position only matters where the stream had ended
        const std::size_t k = detail::BlockPump::read(dst, n);
        calls_ += k + (k < n ? 1 : 0);
        return k;
    }
    std::uint16_t channels() const override { return out_ch_; }
    std::uint32_t sample_rate() const override { return rate_; }
    /// mixer.rs:104-106: a mixer has no duration.
    std::optional<Nanos> total_duration() const override { return std::nullopt; }
    /// mixer.rs:139-166: (0, Some(0)) while no source plays -- sources that wait for the next frame do not count --; otherwise the largest lower
    /// bound among the sources that play where the CONSUMER stands, and no upper bound (every source sits in a UniformSourceIterator,
    /// mixer.rs:58-66, whose upper bound is None: uniform.rs:100-108).  A source plays from the call that admitted it to the call in which it
    /// returns None (mixer.rs:185-198).  Which adapters rodio's spelling of add(src, gain, filter) has: see HintTrack.
    SizeHint size_hint() const override {
        const std::uint64_t consumed = started() ? slot_base_[cur_index()] * out_ch_ + position() : 0;
        bool any = false;
        std::size_t lower = 0;
        for (std::size_t i = 0; i < hints_.size();) {
            HintTrack &t = *hints_[i];
            const std::uint64_t first = t.join_frame * out_ch_;
            if (consumed <= first) {
                ++i;
                continue;  // admitted by a call still to come
            }
            const std::optional<std::size_t> lo = t.lower_at(consumed - first, out_ch_);
            if (!lo) {  // it has returned None in front of the consumer: gone for good
                hints_.erase(hints_.begin() + (std::ptrdiff_t)i);
                continue;
            }
            any = true;
            lower = std::max(lower, *lo);
            ++i;
        }
        return any ? SizeHint{lower, std::nullopt} : SizeHint{0, std::size_t(0)};
    }
    /// What became of the GpuSource chains handed to add(): how many there were, how many delivered their blocks on the device, and
    /// the samples of chain output that crossed to the host (0 when every chain stayed on the device) / went device-to-device.
    struct ChainStats {
        std::uint64_t chains = 0, on_device = 0, d2h_samples = 0, device_samples = 0;
    };
    ChainStats chain_stats() const {
        ChainStats st = retired_chains_;
        for (const auto &gp : gens_) count_chains(*gp, st);
        for (const Src &x : pending_) count_chain(x, st);
        return st;
    }
    /// Blocks of wide generations that ran as one launch (rh_wide_mix_block).
    std::uint64_t wide_fused_blocks() const { return wide_fused_blocks_; }
    /// Output frame (of this mixer) at which the most recently started generation joined.
    std::uint64_t last_join_frame() const { return last_join_; }
    /// Threads that have pulled sources so far (1 until a block was large enough for the pool).
    unsigned pull_threads() const { return pool_ ? pool_->threads() : 1; }

protected:
    bool can_resume() const override { return !pending_.empty() && resume_ok_; }  // mixer.rs:117-136: None while empty, samples again after add()
    void block_done() override {  // a bounded wait inside the fused kernel expired (never seen on a healthy device): fail loudly
        for (auto &g : gens_)
            if (g->plan) check(rh_rlm_last_status(g->plan), "rh_rlm_last_status");
        if (device_chains_) check(rh_async_status(), "rh_async_status");  // ... or inside a scan kernel of a chain that hands its blocks over on the device
    }
    void enqueue(Slot &s) override {
        if (!pending_.empty()) start_generation();
        if (gens_.empty()) {  // nothing to pull
            s.n = 0;
            s.last = true;
            slot_base_[slot_index(s)] = scheduled_;
            slot_frames_[slot_index(s)] = 0;
            return;
        }
        s.out.reset(out_cap_frames_ * 2 * qch_);
        if (debug_poison()) std::memset(s.out.get(), 0xff, out_cap_frames_ * 2 * qch_ * sizeof(float));  // diagnostics: a block served before it arrived reads NaN
        // 1. every live generation converts, filters and mixes one block of its sources behind what its queue holds
        // (one that runs ahead of the slowest -- another rate, other tile boundaries -- waits with a full queue)
        for (auto &gp : gens_)
            if (!gp->done && gp->fill < out_cap_frames_) run_block(*gp, s);
        // 2. the frames every unfinished generation has reached; finished ones give what they have left
        std::uint64_t n = ~0ull, most = 0;
        bool any_live = false;
        for (auto &gp : gens_) {
            most = std::max(most, gp->fill);
            if (!gp->done) {
                any_live = true;
                n = std::min(n, gp->fill);
            }
        }
        if (!any_live) n = most;
        // 3. sum the generations in insertion order (a single one is already the mix) and send the block to the host
        if (n) {
            const float *mixed = gens_.front()->queue();
            if (gens_.size() > 1) {
                std::vector<const float *> ptrs;
                std::vector<std::uint64_t> start, len;
                for (auto &gp : gens_) {
                    ptrs.push_back(gp->queue());
                    start.push_back(0);
                    len.push_back(std::min(gp->fill, n) * qch_);
                }
                dmix_.reset(out_cap_frames_ * 2 * qch_);
                if (debug_poison()) check(rh_memset(dmix_.get(), 0xff, out_cap_frames_ * 2 * qch_ * sizeof(float), stream_), "rh_memset");
                check(rh_mix_sum(dmix_.get(), n * qch_, ptrs.data(), start.data(), len.data(), (std::uint32_t)ptrs.size(), stream_), "rh_mix_sum");
                mixed = dmix_.get();
            }
            send_block(s, mixed, n);
            // the block also stays on the device until it has been served: a source that joins in the middle of it is added there
            dkeep_[slot_index(s)].reset(out_cap_frames_ * 2 * qch_);
            check(rh_memcpy_d2d(dkeep_[slot_index(s)].get(), mixed, n * qch_ * sizeof(float), stream_), "rh_memcpy_d2d");
        }
        slot_base_[slot_index(s)] = scheduled_;
        slot_frames_[slot_index(s)] = n;
        // 4. what a generation produced beyond n waits at the front of its (other) queue buffer
        for (auto &gp : gens_) {
            Gen &g = *gp;
            const std::uint64_t used = std::min(g.fill, n), rem = g.fill - used, pad = rem & 1;  // the fused kernel writes 16-byte aligned blocks behind it
            if (rem) check(rh_memcpy_d2d(g.q[g.cur ^ 1].get() + pad * qch_, g.queue() + used * qch_, rem * qch_ * sizeof(float), stream_), "rh_memcpy_d2d");
            g.cur ^= 1;
            g.head = pad;
            g.fill = rem;
        }
        // the slot's staging block stays in use until this block's copies have run: generations are only retired here,
        // after their last frames were scheduled; their plans are destroyed once the stream has drained them
        scheduled_ += n;
        bool all_done = true;
        for (auto &gp : gens_) all_done = all_done && gp->done && gp->fill == 0;
        std::size_t trim = 0;
        if (all_done && out_ch_ > 1) {
            // MixerSource::next returns None the moment no source is left (mixer.rs:120-136): when the sources that last longest end inside a
            // frame -- the converters' output of a span that ends inside a frame need not fill one -- so does the mix.  The blocks hold whole
            // frames (the missing samples were added as +0.0, which leaves the sums as they are); the last one is cut to what rodio returns.
            std::uint64_t last = 0;
            std::uint32_t valid = 0;
            for (auto &gp : gens_) last = std::max(last, gp->join + gp->emitted);
            for (auto &gp : gens_)
                if (gp->join + gp->emitted == last) valid = std::max<std::uint32_t>(valid, gp->last_valid ? gp->last_valid : qch_);
            if (n && valid && valid < qch_) trim = qch_ - valid;
        }
        if (all_done) {
            check(rh_stream_synchronize(stream_), "rh_stream_synchronize");
            check(rh_stream_synchronize(copy_stream_), "rh_stream_synchronize");
            for (auto &gp : gens_)
                if (gp->plan) check(rh_rlm_last_status(gp->plan), "rh_rlm_last_status");  // the last blocks too: nothing is served unchecked
            // The generations' plans, page-locked blocks and device rows are NOT freed here: this runs inside the consumer's next()
            // (the audio callback), and freeing a few hundred MB of page-locked memory takes hundreds of milliseconds.  A thread of
            // the mixer's does it (nothing on the device refers to them any more: both streams were waited for above).
            for (auto &gp : gens_) count_chains(*gp, retired_chains_);
            if (!reaper_) reaper_.reset(new Reaper());
            reaper_->retire(std::move(gens_));
            gens_.clear();
        }
        s.n = (std::size_t)n * out_ch_ - trim;
        s.last = gens_.empty() && pending_.empty();
    }

private:
    /// The mixed stereo block `mixed` (n frames, on the device) on its way to the host block of slot `s`, in the mixer's layout.
    void send_block(Slot &s, const float *mixed, std::uint64_t n) {
        if (out_ch_ == qch_) {  // stereo, or a wide mix formed in the mixer's own layout
            check(rh_memcpy_d2h_async(s.out.get(), mixed, n * qch_ * sizeof(float), stream_), "rh_memcpy_d2h_async");
            return;
        }
        dout_.reset(out_cap_frames_ * 2 * out_ch_);  // a mono mixer: ChannelCountConverter(2 -> 1) on the mix (channels.rs:57-85), once per block
        check(rh_channels_convert(dout_.get(), mixed, (std::size_t)n, 2, out_ch_, stream_), "rh_channels_convert");
        check(rh_memcpy_d2h_async(s.out.get(), dout_.get(), n * out_ch_ * sizeof(float), stream_), "rh_memcpy_d2h_async");
    }
    struct Src;
    struct Gen;
    /// size_hint() of ONE source of the mix, where the mixer's consumer stands.  rodio's spelling of add(src, gain, filter):
    ///     mixer.add(src)                                                          gain 1, no filter
    ///     mixer.add(src.amplify(g))                                               a gain (amplify.rs:68-70 hands the bounds on)
    ///     mixer.add(UniformSourceIterator::new(src.amplify(g), ch, rate).low_pass(f))   a filter: it runs at the mixer's rate, behind a converter of its own
    /// and Mixer::add wraps what it gets in a UniformSourceIterator (mixer.rs:58-66).  So the bounds are those of ONE iterator over the source
    /// (`counter`: the mixer itself converts; or the chain's own last `uniform`), and behind a filter of one more -- a pass-through at the
    /// mixer's own format, whose ChannelCountConverter still counts in whole frames from where it stands (channels.rs:88-102).
    struct HintTrack {
        std::uint64_t join_frame = 0;                     // the mixer frame of the source's first frame
        std::shared_ptr<detail::UniformCounter> counter;  // the iterator the mixer stands for (null: the chain ends with its own)
        detail::HintLog log;                              // a plain source's answers by the sample at which it was asked
        std::shared_ptr<GpuSource::HintState> chain;      // a GpuSource chain: its own arithmetic (kept alive beyond the chain)
        std::uint64_t pulled = 0;                         // samples taken from the source
        bool wrapped_again = false;                       // a filter sits behind the iterator: Mixer::add's own iterator comes on top
        bool total_known = false;                         // a chain that ends with its own iterator: the length of its stream, once it has ended
        std::uint64_t total = 0;
        SizeHint source_at(std::uint64_t q) const { return chain ? chain->at(q) : log.empty() ? SizeHint{} : log.at(q); }
        std::optional<std::size_t> lower_at(std::uint64_t e, std::uint16_t ch) {
            std::size_t lo;
            if (counter) {
                const std::optional<SizeHint> h = counter->hint_at(e, [this](std::uint64_t q) { return source_at(q); });
                if (!h) return std::nullopt;
                lo = h->lower;
            } else {
                if (total_known && e > total) return std::nullopt;
                lo = source_at(e).lower;
            }
            if (!wrapped_again) return lo;
            const std::uint64_t pos = e % ch;  // channels.rs:88-102 with from == to: ((lower + pos) / ch * ch) - pos
            const std::uint64_t x = (lo + pos) / ch * ch;
            return (std::size_t)(x > pos ? x - pos : 0);
        }
    };
    mutable std::vector<std::shared_ptr<HintTrack>> hints_;
    void track(Gen &g) {  // (start_stream / start_stream_wide: the generation's sources start playing)
        for (Src &x : g.srcs) {
            auto t = std::make_shared<HintTrack>();
            t->join_frame = g.join;
            const GpuSource *gs = x.up ? dynamic_cast<const GpuSource *>(x.up.get()) : nullptr;
            if (gs) t->chain = gs->hint_;
            if (gs && gs->hint_->log.empty()) gs->hint_->before = gs->up_->size_hint();
            const bool own = gs && gs->ends_with_uniform() && gs->channels() == out_ch_ && gs->sample_rate() == rate_ && x.gain == 1.0f && x.filt.kind < 0;
            const bool own_filtered = gs && !own && chain_ends_with_uniform_and_filter(*gs) && x.gain == 1.0f && x.filt.kind < 0;
            if (own || own_filtered) {
                t->wrapped_again = own_filtered;
            } else {
                t->counter = std::make_shared<detail::UniformCounter>(out_ch_, rate_);
                t->wrapped_again = x.filt.kind >= 0;
            }
            x.hint = t;
            hints_.push_back(std::move(t));
        }
    }
    // (make_wide / add(chain) complete a chain to `.. -> uniform(channels, rate) -> filter`: the iterator is the last adapter but one)
    static bool chain_ends_with_uniform_and_filter(const GpuSource &gs) {
        return gs.last_kind_ == 2;
    }
    static void count_chain(const Src &x, ChainStats &st) {
        const GpuSource *g = x.up ? dynamic_cast<const GpuSource *>(x.up.get()) : nullptr;
        if (!g) return;
        st.chains += 1;
        st.on_device += g->blocks_on_device() ? 1 : 0;
        st.d2h_samples += g->timing().d2h_samples;
        st.device_samples += g->timing().device_samples;
    }
    static void count_chains(const Gen &g, ChainStats &st) {
        for (const Src &x : g.srcs) count_chain(x, st);
    }
    struct Src {
        BoxSource up;
        GpuSource *dev = nullptr;  // == up.get() when the source is a chain that hands its blocks over on the device
        std::uint64_t dheld = 0, dheld_off = 0;  // ... frames the converter has not consumed: `dheld` of them from frame `dheld_off` of the row of the block before
        Filter filt;
        float gain = 1.0f;
        std::vector<float> held;  // pulled, not yet consumed by the converter (interleaved, in the source's own channel layout)
        bool ended = false;
        std::uint16_t ch = 2;     // the source's channel count (what is not stereo is converted on the device, block by block)
        // span-by-span generations: how rodio's UniformSourceIterator would pull this source, and where its converted frames wait
        detail::SpanReader reader{nullptr};
        detail::UniformPlanner plan;
        std::uint64_t have_s = 0, off_s = 0;  // converted SAMPLES not yet mixed: `have_s` of them from sample `off_s` of the source's device row
        std::uint64_t total_s = 0;            // samples of the source's stream in the mixer's layout so far (where the generation tracks them: Gen::track)
        // a wide mixer's plain sources (Gen::widefused): `held` starts at frame `wpos` of the source's stream; its reduced rates
        bool fusew = false;
        std::uint64_t wpos = 0;
        std::uint32_t wF = 1, wT = 1;
        std::shared_ptr<HintTrack> hint;      // size_hint(): see HintTrack
        // a pull of `got` samples (ONE continuous span: the source reports none); `ended`: it returned None behind them
        void note_pull(std::size_t got, bool ended_now) {
            if (!hint) return;
            if (hint->counter) {
                detail::Piece p{got, hint->pulled == 0, ended_now, ch, 0, 0, ended_now, std::nullopt, ch, 0, hint->pulled == 0, std::nullopt};
                p.rate = p.src_rate = up->sample_rate();
                if (got || ended_now) hint->counter->feed(p, hint->pulled);
                if (ended_now) hint->counter->input_ended();
            }
            hint->pulled += got;
        }
    };
    struct Gen {  // sources that joined together: one clock, one fused stream
        std::vector<Src> srcs;
        rh_rlm *plan = nullptr;
        detail::DeviceBuf din[3], q[2];  // staged input rows, THREE sets in rotation: the rows of a block stay untouched until the block after it has run
                                         // (rh_rlm_stream_keep_history: sources that run together are summed first); mixed output not yet served (ping-pong)
        int dnext = 0, pd = 0, pd_prev = -1;  // the row set the next pull takes / the one of the block that was pulled last / the one before it
        detail::Event copied[2];         // ... recorded on the copy stream behind the copies that fill din[i]
        detail::PinnedBuf stage[2];   // one staging block per slot in flight
        detail::PinnedBuf side[2];    // ... and one for the sources that are not stereo, in their own layout
        detail::DeviceBuf dside[2];
        int cur = 0, slot = 0;
        std::uint64_t head = 0, fill = 0;  // q[cur] holds `fill` frames from frame `head` on (head in {0,1}: the END stays 16-byte aligned)
        bool done = false;                 // the stream emitted its last frame
        // the host half of a block that has been pulled and not yet issued (pull_block / issue_block)
        bool pulled = false;
        int pslot = 0;
        std::vector<const float *> pptrs;
        std::vector<std::uint64_t> pavail;
        std::vector<std::uint8_t> pended;
        std::vector<std::size_t> pside_off;
        std::size_t pside = 0, ptotal = 0;
        std::vector<rh_uniform_seg> ptable;
        std::uint64_t pmax_out = 0;
        bool mono = false;                 // a fused stream of mono sources (rh_rlm_config.channels = 1): mono rows in, a mono mix out ...
        detail::DeviceBuf qm;              // ... which ChannelCountConverter(1 -> 2) (channels.rs:64-73) turns into the stereo queue, once per block
        // span-by-span generations (`staged`): the sources are converted to the mixer's format first, row by row
        bool staged = false;
        std::size_t sch = 2;                 // ... in frames of this many channels: 2, or 1 for a mono mixer (UniformSourceIterator(src, 1, rate) per source: a span that ends
                                             // inside a frame gives one sample either way, channels.rs:57-67, so the mono stream cannot be had from a stereo one)
        std::uint64_t target = 0, crow = 0;  // converted frames a block tops every row up to; capacity of a row (frames)
        detail::DeviceBuf conv[2], dtab;     // converted rows (ping-pong: what a block leaves moves to the front of the other set); segment table
        detail::PinnedBuf tab[2];
        int ccur = 0;
        std::size_t qch = 2;                 // channels of the queue's frames (the mixer's, when it has more than two)
        const float *queue() const { return q[cur].get() + head * qch; }
        float *queue_end() { return q[cur].get() + (head + fill) * qch; }
        Filter filt;
        // a generation of a wide mixer: every source a chain in the mixer's layout, summed by rh_mix_sum
        bool wide = false;
        detail::DeviceBuf drow;
        // ... or, where every source is a plain continuous one (Src::fusew), ONE launch a block that converts and sums them (rh_wide_mix_block)
        bool widefused = false;
        std::uint64_t wm = 0, wend = 0;  // output frames planned so far; (once every source has ended) where the longest stream ends
        std::size_t wrow = 0;            // floats per staged row
        std::vector<rh_wide_src> wtab;   // the block that has been pulled: its sources as the launch sees them,
        std::uint64_t wout = 0;          // its output frames,
        bool wlast = false;              // and whether it is the generation's last
        // where the generation's stream lies in the mixer's and how it ended (enqueue(): the last block of a mix that ends inside a frame)
        std::uint64_t join = 0, emitted = 0;  // mixer frame of its first frame; frames it has produced
        bool track = false;                   // its sources' streams are counted in samples (Src::total_s)
        std::uint32_t last_valid = 0;         // samples of its LAST frame that rodio's sources cover (0: the whole frame)
        void finish() {  // (when `done` is set) the frame the longest sources end in
            if (!track) return;
            std::uint64_t most = 0;
            for (const Src &x : srcs) most = std::max(most, (x.total_s + qch - 1) / qch);
            std::uint32_t v = 0;
            for (const Src &x : srcs)
                if ((x.total_s + qch - 1) / qch == most) v = std::max<std::uint32_t>(v, x.total_s % qch ? (std::uint32_t)(x.total_s % qch) : (std::uint32_t)qch);
            last_valid = v == qch ? 0 : v;
        }
        Gen() = default;
        Gen(const Gen &) = delete;
        Gen &operator=(const Gen &) = delete;
        ~Gen() {
            if (plan) (void)rh_rlm_destroy(plan);
        }
    };
    /// Frees retired generations on a thread of its own (see enqueue()).  The sources a generation owned are DESTROYED ON THAT THREAD
    /// (rodio: `Source: Send` -- a source handed to Mixer::add may be dropped by another thread than the one that made it).
    class Reaper {
    public:
        Reaper() : th_([this] { loop(); }) {}
        ~Reaper() {
            {
                std::lock_guard<std::mutex> lk(mu_);
                quit_ = true;
            }
            cv_.notify_all();
            th_.join();
        }
        void retire(std::vector<std::unique_ptr<Gen>> g) {
            {
                std::lock_guard<std::mutex> lk(mu_);
                for (auto &p : g) dead_.push_back(std::move(p));
            }
            cv_.notify_all();
        }

    private:
        void loop() {
            (void)rh_bind_thread();  // HIP's current device is per thread: what this thread frees belongs to the mixer's device
            std::unique_lock<std::mutex> lk(mu_);
            for (;;) {
                cv_.wait(lk, [this] { return quit_ || !dead_.empty(); });
                std::vector<std::unique_ptr<Gen>> take;
                take.swap(dead_);
                lk.unlock();
                take.clear();  // the destructors: rh_rlm_destroy, rh_host_free, rh_free
                lk.lock();
                if (quit_ && dead_.empty()) return;
            }
        }
        std::mutex mu_;
        std::condition_variable cv_;
        std::vector<std::unique_ptr<Gen>> dead_;
        bool quit_ = false;
        std::thread th_;
    };
    static bool fused_ratio_unsupported(std::uint32_t from, std::uint32_t to) {  // rh_rlm_create: reduced from/to <= 4.5, from*to within u32
        std::uint64_t a = from, b = to;
        while (b) {
            const std::uint64_t t = a % b;
            a = b;
            b = t;
        }
        const std::uint64_t F = from / a, T = to / a;
        return 2 * F > 9 * T || F * T > 0xffffffffull;
    }
    static bool spanned(const Src &x) { return x.up->current_span_len().has_value(); }
    /// A continuous source the fused kernel cannot take as it is (rate ratio above 4.5) gets the GPU converter adapter in front.
    void make_direct(Src &x) {
        if (x.dev || !fused_ratio_unsupported(x.up->sample_rate(), rate_)) return;
        auto conv = std::make_unique<GpuSource>(std::move(x.up), opt_.block_frames);
        if (x.ch != 2) conv->convert_channels(2);
        conv->convert_sample_rate(rate_);
        x.up = std::move(conv);
        x.ch = 2;
    }
    bool wide() const { return out_ch_ > 2; }
    /// A source of a wide mixer: its own chain on the device, [amplify] -> UniformSourceIterator(channels, rate) -> [filter] (mixer.rs:58-66 with
    /// rodio's adapters in rodio's order), the blocks handed over in device memory.
    void make_wide(Src &x) {
        GpuSource *gs = x.dev;
        if (!gs) {
            auto chain = std::make_unique<GpuSource>(std::move(x.up), opt_.block_frames);
            gs = chain.get();
            x.up = std::move(chain);
            x.dev = gs;
        }
        if (x.gain != 1.0f) gs->amplify(x.gain);
        gs->uniform(out_ch_, rate_);
        if (x.filt.kind >= 0) {
            if (opt_.reference_exact_filters && !rh_filter_scan_ok(x.filt.kind, x.filt.freq, x.filt.q, rate_)) gs->exact_filters(true);
            if (x.filt.kind == 0) gs->low_pass_with_q(x.filt.freq, x.filt.q);
            else gs->high_pass_with_q(x.filt.freq, x.filt.q);
        }
        gs->keep_blocks_on_device();
        x.gain = 1.0f;
        x.filt = Filter::none();
        x.ch = out_ch_;
        device_chains_ = true;
    }
    void start_generation() {  // the sources that joined together
        std::vector<Src> all = std::move(pending_);
        pending_.clear();
        if (wide()) {
            start_stream_wide(std::move(all), scheduled_);
            return;
        }
        bool any_spans = false;
        for (const Src &x : all) any_spans = any_spans || (!x.dev && spanned(x));
        if (any_spans) {  // span by span, as rodio converts them: one stream per filter for all of them, in insertion order
            std::vector<Filter> fk;
            for (const Src &x : all)
                if (!x.dev && std::find(fk.begin(), fk.end(), x.filt) == fk.end()) fk.push_back(x.filt);
            for (const Filter &f : fk) {
                std::vector<Src> group;
                for (Src &x : all)
                    if (x.up && !x.dev && x.filt == f) group.push_back(std::move(x));
                start_stream(std::move(group), true);
            }
            // (chains that hand their blocks over on the device are continuous streams in the fused kernel's layout: streams of their own, below)
            std::vector<Src> rest;
            for (Src &x : all)
                if (x.up) rest.push_back(std::move(x));
            all = std::move(rest);
            if (all.empty()) return;
        }
        // continuous sources: one fused stream per (input rate, filter) -- and one for its mono sources, which the kernel reads as
        // they are (4 bytes per frame) -- in order of first appearance
        for (Src &x : all) make_direct(x);
        struct Key {
            std::uint32_t rate;
            bool mono;
            Filter filt;
            bool operator==(const Key &o) const { return rate == o.rate && mono == o.mono && filt == o.filt; }
        };
        std::vector<Key> keys;
        for (const Src &x : all) {
            const Key k{x.up->sample_rate(), x.ch == 1, x.filt};
            if (std::find(keys.begin(), keys.end(), k) == keys.end()) keys.push_back(k);
        }
        for (const Key &k : keys) {
            std::vector<Src> group;
            for (Src &x : all)
                if (x.up && Key{x.up->sample_rate(), x.ch == 1, x.filt} == k) group.push_back(std::move(x));
            start_stream(std::move(group), false, k.mono);
        }
    }
    void start_stream_wide(std::vector<Src> srcs, std::uint64_t join) {
        auto gp = std::make_unique<Gen>();
        Gen &g = *gp;
        bool fused = !srcs.empty();
        for (const Src &x : srcs) fused = fused && x.fusew;
        for (Src &x : srcs) {
            if (!fused && x.fusew) make_wide(x);  // (one chain among them: the rows of all are summed in insertion order by rh_mix_sum)
            x.fusew = fused;
            if (fused) {
                const std::uint32_t from = x.up->sample_rate(), gc = std::gcd(from, rate_);
                x.wF = from / gc, x.wT = rate_ / gc;
            }
        }
        g.srcs = std::move(srcs);
        g.wide = true;
        g.widefused = fused;
        g.track = true;
        g.qch = qch_;
        g.join = join;
        // (a block takes at most block_frames frames of every chain, whatever the chains' own blocks are)
        out_cap_frames_ = std::max<std::uint64_t>(out_cap_frames_, opt_.block_frames + 64);
        for (auto &other : gens_)
            for (auto &b : other->q) grow_keep(b, out_cap_frames_ * 2 * qch_, (other->head + other->fill) * qch_);
        for (auto &b : g.q) b.reset(out_cap_frames_ * 2 * qch_);
        last_join_ = join;
        track(g);
        gens_.push_back(std::move(gp));
    }
    void start_stream(std::vector<Src> srcs, bool staged, bool mono = false) {
        auto gp = std::make_unique<Gen>();
        Gen &g = *gp;
        g.join = scheduled_;
        g.srcs = std::move(srcs);
        g.filt = g.srcs.front().filt;  // (one filter per stream: start_generation / late_join group by it)
        g.staged = staged;
        g.sch = staged && out_ch_ == 1 ? 1 : 2;
        g.mono = staged ? g.sch == 1 : mono;
        const std::uint32_t from = staged ? rate_ : g.srcs.front().up->sample_rate();  // staged: the fused kernel sees converted rows
        rh_rlm_config cfg;
        std::memset(&cfg, 0, sizeof cfg);
        cfg.from_rate = from;
        cfg.to_rate = rate_;
        cfg.channels = g.mono ? 1 : 2;
        cfg.span_len = 0;
        cfg.filter_kind = g.filt.kind;
        cfg.filter_freq = g.filt.freq;
        cfg.filter_q = g.filt.q;
        cfg.max_sources = (std::uint32_t)g.srcs.size();
        cap_frames_ = opt_.block_frames + 4096;  // a block can hold what the previous one left over: less than two tiles' worth of input
        if (staged) {
            g.target = opt_.block_frames + 64 * 20 + 8;  // a block emits whole tiles (at most 64 * 20 frames each) and keeps two frames of history
            g.crow = g.target + 64;
            for (Src &x : g.srcs) {
                x.reader = detail::SpanReader(x.up.get());
                x.plan = detail::UniformPlanner((std::uint16_t)g.sch, rate_);
            }
            const std::size_t crowf = ((std::size_t)g.crow * 2 + 3) & ~std::size_t(3);
            for (auto &b : g.conv) b.reset(g.srcs.size() * crowf);
        }
        cfg.max_in_frames = staged ? g.crow : cap_frames_;
        cfg.frames_per_lane = opt_.frames_per_lane;
        check(rh_rlm_create(&g.plan, &cfg), "rh_rlm_create");
        check(rh_rlm_set_exclusive(g.plan, 0), "rh_rlm_set_exclusive");  // the copy stream's launches (and other mixers) share the CUs: tiles by ticket
        std::vector<float> gains;
        for (const Src &x : g.srcs) gains.push_back(staged ? 1.0f : x.gain);  // staged: the factor sits in front of the converter, where Mixer::add(src.amplify(g)) has it
        check(rh_rlm_set_gains(g.plan, gains.data(), (std::uint32_t)gains.size()), "rh_rlm_set_gains");
        check(rh_rlm_stream_begin(g.plan), "rh_rlm_stream_begin");
        // sources that start together run together until the first of them ends: their blocks are summed first (rodio_hip.h).  The rows of
        // a direct generation rotate through three sets, which is what the recovery at that moment needs; converted rows (staged) do not.
        if (!staged) check(rh_rlm_stream_keep_history(g.plan, 1), "rh_rlm_stream_keep_history");
        row_ = (cap_frames_ * 2 + 3) & ~std::size_t(3);  // 16-byte aligned rows
        std::uint64_t m = 0;
        check(rh_resample_out_frames(staged ? g.crow : cap_frames_, from, rate_, cfg.channels, 0, &m), "rh_resample_out_frames");
        out_cap_frames_ = std::max<std::uint64_t>(out_cap_frames_, m + 64);
        for (auto &other : gens_)  // rates differ between generations: every queue holds two of the largest blocks
            for (auto &b : other->q) grow_keep(b, out_cap_frames_ * 2 * qch_, (other->head + other->fill) * qch_);
        for (auto &b : g.q) b.reset(out_cap_frames_ * 2 * qch_);
        g.track = staged || from == rate_;  // the sources' streams reach the mix sample for sample: a stream that ends inside a frame is seen
        last_join_ = scheduled_;
        track(g);
        gens_.push_back(std::move(gp));
    }
    void grow_keep(detail::DeviceBuf &b, std::size_t floats, std::size_t keep) {
        if (b.size() >= floats) return;
        detail::DeviceBuf n(floats);
        if (keep) check(rh_memcpy_d2d(n.get(), b.get(), keep * sizeof(float), stream_), "rh_memcpy_d2d");
        check(rh_stream_synchronize(stream_), "rh_stream_synchronize");
        b.swap(n);
    }
    /// One block of a generation = its host half (pull_block: the upstreams are pulled into a page-locked staging block, by a few
    /// threads) and its device half (issue_block: copies and launches on stream_).  BlockPump runs the host half of the next
    /// block while the device still works on the block about to be served (prefetch()).
    void run_block(Gen &g, Slot &) {
        if (!g.pulled) pull_block(g);
        issue_block(g);
    }
    void prefetch() override {
        if (!pending_.empty()) return;  // a new generation starts with the next block: enqueue() does it all
        for (auto &gp : gens_)
            if (!gp->done && gp->fill < out_cap_frames_ && !gp->pulled) pull_block(*gp);
    }
    void pull_block(Gen &g) {
        g.pslot = g.slot;
        g.slot ^= 1;
        g.pd_prev = g.pd_prev < 0 && g.dnext == 0 ? -1 : g.pd;
        g.pd = g.dnext;
        g.dnext = (g.dnext + 1) % 3;
        if (g.wide) {  // (chains pull their own upstreams, a block ahead)
            if (g.widefused) pull_block_widefused(g);
        }
        else if (g.staged) pull_block_staged(g);
        else pull_block_direct(g);
        g.pulled = true;
    }
    void issue_block(Gen &g) {
        g.pulled = false;
        const std::uint64_t before = g.fill;
        if (g.widefused) issue_block_widefused(g);
        else if (g.wide) issue_block_wide(g);
        else if (g.staged) issue_block_staged(g);
        else issue_block_direct(g);
        g.emitted += g.fill - before;
        for (const Src &x : g.srcs)  // (rodio's UniformSourceIterator asks a source again, once, after its None)
            if (x.ended)
                if (const GpuSource *gs = x.up ? dynamic_cast<const GpuSource *>(x.up.get()) : nullptr; gs && gs->ended_with_a_stale_frame())
                    throw Error(RH_ERR_UNSUPPORTED, "GpuMixer: a chain ended inside a frame of its channel_volume's input: asked again, as rodio's mixer asks, ChannelVolume returns a frame of its "
                                                    "stale sum (channel_volume.rs:71-88)");
        if (g.done) g.finish();
    }
    /// A block of a wide generation of PLAIN sources (continuous, no filter: Src::fusew), host half.  Output frames g.wm .. of the mix are
    /// planned here: every live source is pulled until both taps of frame g.wm + block_frames - 1 are there (sources of different rates
    /// give different numbers of frames), the block emits what every live source can give -- everything once all have ended -- and one
    /// pitched copy brings the rows `[what the block before left | the new frames]`, each in its source's own layout, to the device.
    // (64-bit arithmetic throughout: m F is split at T, so nothing overflows however long the mixer has played)
    static std::uint64_t wide_first_tap(std::uint64_t m, std::uint32_t F, std::uint32_t T) { return m / T * F + m % T * F / T; }  // floor(m F / T)
    static std::uint64_t wide_ready(std::uint64_t n, std::uint32_t F, std::uint32_t T) {  // output frames a stream of n frames (so far) can give: both taps there
        if (F == T || n == 0) return n;
        const std::uint64_t a = n - 1;  // ceil(a T / F): the frames m with floor(m F / T) <= n - 2
        return a / F * T + (a % F * T + F - 1) / F;
    }
    static std::uint64_t wide_total(std::uint64_t n, std::uint32_t F, std::uint32_t T) {  // ... of an ended one: + the verbatim last frame (sample_rate.rs:193-200)
        if (F == T || n == 0) return n;
        const std::uint64_t c1 = wide_ready(n, F, T);
        return c1 + (wide_first_tap(c1, F, T) < n ? 1 : 0);  // (the next frame's first tap is the last frame: it lands on it)
    }
    static bool ratio_beyond_u32(std::uint32_t from, std::uint32_t to) {  // the reference multiplies in u32 (sample_rate.rs:157,173)
        const std::uint32_t gc = std::gcd(from, to);
        return (std::uint64_t)(from / gc) * (to / gc) > 0xffffffffull;
    }
    void pull_block_widefused(Gen &g) {
        const std::size_t S = g.srcs.size();
        const std::uint64_t m0 = g.wm, target = m0 + opt_.block_frames;
        // frames every live source must hold for the block, and the row that fits them all
        std::vector<std::uint64_t> need(S, 0);
        std::size_t roww = 4;
        for (std::size_t i = 0; i < S; ++i) {
            Src &x = g.srcs[i];
            const std::uint64_t have = x.held.size() / x.ch;
            need[i] = x.ended ? x.wpos + have : (x.wF == x.wT ? target : wide_first_tap(target - 1, x.wF, x.wT) + 2);
            const std::uint64_t fr = std::max(need[i], x.wpos + have) - x.wpos;
            roww = std::max<std::size_t>(roww, ((std::size_t)fr * x.ch + 3) & ~std::size_t(3));
        }
        // (rows a little longer than this block needs, and never shorter than the block before: the page-locked block and the device rows are
        //  allocated once -- a block whose rows come out a frame longer must not allocate 50 MB of page-locked memory again)
        roww = std::max(g.wrow, (roww + 64 * qch_ + 1023) & ~std::size_t(1023));
        g.wrow = roww;
        detail::PinnedBuf &stage = g.stage[g.pslot];
        stage.reset(S * roww);
        detail::DeviceBuf &din = g.din[g.pd];
        din.reset(S * roww);
        std::vector<std::size_t> haves(S, 0);  // samples in every row: [what the block before left | the new ones]
        pull_sources(S * opt_.block_frames, S, [&](std::size_t i) {
            Src &x = g.srcs[i];
            const std::size_t ch = x.ch;
            float *row = stage.get() + i * roww;
            std::size_t have = x.held.size();
            if (have) std::memcpy(row, x.held.data(), have * sizeof(float));
            if (!x.ended && need[i] > x.wpos + have / ch) {
                const std::size_t want = (std::size_t)(need[i] - x.wpos - have / ch) * ch;
                if (x.hint && !x.hint->chain) x.hint->log.note(x.hint->pulled, x.up->size_hint());
                const std::size_t got = x.up->read(row + have, want);
                x.ended = got < want;
                x.note_pull(got, x.ended);
                if (got % ch) throw Error(RH_ERR_UNSUPPORTED, "GpuMixer: a source whose current_span_len() is None ended inside a frame (source/mod.rs:169-178 asks for whole frames)");
                have += got;
            }
            haves[i] = have;
        });
        // what the block emits
        bool any_live = false;
        std::uint64_t m_end = target, longest = 0;
        for (std::size_t i = 0; i < S; ++i) {
            const Src &x = g.srcs[i];
            const std::uint64_t n = x.wpos + haves[i] / x.ch;
            if (x.ended) longest = std::max(longest, wide_total(n, x.wF, x.wT));
            else any_live = true, m_end = std::min(m_end, wide_ready(n, x.wF, x.wT));
        }
        if (!any_live) m_end = std::min(target, longest);
        if (m_end < m0) m_end = m0;
        const std::uint64_t out = m_end - m0;
        g.wout = out;
        g.wlast = !any_live && m_end >= longest;
        g.wend = longest;
        g.wtab.assign(S, rh_wide_src{});
        std::size_t width = 0;
        for (std::size_t i = 0; i < S; ++i) {
            Src &x = g.srcs[i];
            const std::uint64_t have = haves[i] / x.ch, n = x.wpos + have;
            const std::uint64_t i0 = wide_first_tap(m0, x.wF, x.wT);
            rh_wide_src &d = g.wtab[i];
            const std::uint64_t end = x.ended ? wide_total(n, x.wF, x.wT) : m_end;
            d.frames = std::min(end, m_end) > m0 ? std::min(end, m_end) - m0 : 0;
            d.data = d.frames ? din.get() + i * roww + (std::size_t)(i0 - x.wpos) * x.ch : nullptr;  // (i0 >= wpos: the block before dropped up to its own i(m_end) only)
            d.channels = x.ch;
            d.from_rate = x.up->sample_rate();
            d.phase = (std::uint32_t)(m0 % x.wT * x.wF % x.wT);
            d.last = x.ended ? (std::uint32_t)(n ? n - 1 - std::min(i0, n - 1) : 0) : 0xffffffffu;
            d.gain = x.gain;
            if (d.frames) width = std::max<std::size_t>(width, haves[i]);
            if (x.ended && g.wlast) x.total_s = wide_total(n, x.wF, x.wT) * qch_;
            // the frames from the next block's first tap on are kept for it (a few: the block consumes what it was pulled for)
            const std::uint64_t keep_from = std::min(wide_first_tap(m_end, x.wF, x.wT), n);
            const float *row = stage.get() + i * roww;
            x.held.assign(row + (std::size_t)(keep_from - x.wpos) * x.ch, row + haves[i]);
            x.wpos = keep_from;
            if (x.ended && x.hint && (!d.frames || std::min(end, m_end) == end)) x.hint->total_known = true, x.hint->total = wide_total(n, x.wF, x.wT) * qch_;
        }
        g.wm = m_end;
        if (width) check(rh_memcpy_h2d_rows(din.get(), stage.get(), roww * sizeof(float), width * sizeof(float), S, copy_stream_), "rh_memcpy_h2d_rows");
        check(rh_event_record(g.copied[g.pslot].get(), copy_stream_), "rh_event_record");
    }
    /// ... and its device half: the one launch (per source Amplify -> SampleRateConverter -> ChannelCountConverter, and the ordered sum)
    /// behind what the queue holds.
    void issue_block_widefused(Gen &g) {
        check(rh_stream_wait_event(stream_, g.copied[g.pslot].get()), "rh_stream_wait_event");
        const std::uint64_t out = g.wout;
        if (out > out_cap_frames_ * 2 - g.fill - g.head) throw Error(RH_ERR_CAPACITY, "GpuMixer: the queue of a wide generation");
        if (out) {
            check(rh_wide_mix_block(g.queue_end(), (std::uint32_t)qch_, rate_, out, g.wtab.data(), (std::uint32_t)g.wtab.size(), stream_), "rh_wide_mix_block");
            ++wide_fused_blocks_;
        }
        g.fill += out;
        g.done = g.wlast;
    }
    /// A block of a wide generation: up to block_frames frames of every chain, device to device into a row each, and the ordered sum
    /// of the rows (mixer.rs:185-198) behind what the queue holds.  A chain that ends inside a frame has its last frame completed with
    /// +0.0 (which leaves the sums as they are); Gen::finish remembers how far rodio's samples reach.
    void issue_block_wide(Gen &g) {
        const std::size_t S = g.srcs.size();
        const std::size_t want = opt_.block_frames * qch_, roww = ((want + 3) & ~std::size_t(3)) + 4 * ((qch_ + 3) / 4);
        g.drow.reset(S * roww);
        std::vector<const float *> ptrs(S);
        std::vector<std::uint64_t> start(S, 0), len(S, 0);
        std::uint64_t most = 0;
        bool all_ended = true;
        for (std::size_t i = 0; i < S; ++i) {
            Src &x = g.srcs[i];
            float *row = g.drow.get() + i * roww;
            ptrs[i] = row;
            if (!x.ended) {
                std::size_t got = x.dev->read_device(row, want, stream_);
                x.total_s += got;
                x.note_pull(got, got < want);
                if (got < want) {
                    x.ended = true;
                    if (x.hint) x.hint->total_known = true, x.hint->total = x.total_s;
                    if (const std::size_t cut = got % qch_) {
                        check(rh_memset(row + got, 0, (qch_ - cut) * sizeof(float), stream_), "rh_memset");
                        got += qch_ - cut;
                    }
                }
                len[i] = got;
                most = std::max<std::uint64_t>(most, got);
            }
            all_ended = all_ended && x.ended;
        }
        const std::uint64_t out = most / qch_;
        if (out > out_cap_frames_ * 2 - g.fill - g.head) throw Error(RH_ERR_CAPACITY, "GpuMixer: the queue of a wide generation");
        if (out) check(rh_mix_sum(g.queue_end(), (std::size_t)out * qch_, ptrs.data(), start.data(), len.data(), (std::uint32_t)S, stream_), "rh_mix_sum");
        g.fill += out;
        g.done = all_ended;
    }
    template <class F>
    void pull_sources(std::size_t frames, std::size_t n, F &&fn) {
        const auto t0 = std::chrono::steady_clock::now();
        pullers(frames).run(0, n, fn);
        timing_.pull_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    /// A span-by-span generation.  Every source is topped up to `target` converted frames: it is pulled piece by piece (a piece
    /// never crosses a span; its length is budgeted so that its output fits the row whatever the span does), the pieces are
    /// planned into segments; then ONE copy brings all rows to the device, ONE launch converts all segments of all sources (plus
    /// the frames the last block left over, moved to the front of the other row set), and the fused kernel -- its converter
    /// passing through -- filters and mixes the rows.
    void pull_block_staged(Gen &g) {
        const std::size_t S = g.srcs.size();
        detail::PinnedBuf &stage = g.stage[g.pslot];
        const std::size_t crowf = ((std::size_t)g.crow * 2 + 3) & ~std::size_t(3);
        // 1. layout of the staging block: a row per live source, sized for what it is about to pull in its current format
        std::vector<std::size_t> row_off(S, 0), row_cap(S, 0);
        std::size_t total = 0;
        for (std::size_t i = 0; i < S; ++i) {
            Src &x = g.srcs[i];
            row_off[i] = total;
            std::uint16_t ch = 0;
            std::uint32_t rate = 0;
            if (!x.ended && !x.reader.peek(ch, rate)) x.ended = true;  // the chain rodio would build now is empty
            if (x.ended) continue;
            const std::uint64_t have = x.have_s / 2, want = g.target > have ? g.target - have : 0;
            const std::uint64_t in_frames = want * rate / rate_ + 8;
            row_cap[i] = x.plan.held_samples() + (std::size_t)(in_frames + 1) * ch;  // (+ a frame: read_piece brings a cut frame's samples along with the last whole ones)
            total += (row_cap[i] + 3) & ~std::size_t(3);
        }
        stage.reset(total ? total : 4);
        detail::DeviceBuf &din = g.din[g.pslot];
        din.reset(total ? total : 4);
        g.ptotal = total;
        std::vector<rh_uniform_seg> &table = g.ptable;
        table.clear();
        g.pmax_out = 0;
        const int oc = g.ccur, nc = g.ccur ^ 1;
        for (std::size_t i = 0; i < S; ++i) {  // what the last block left over: to the front of the other row set
            Src &x = g.srcs[i];
            if (!x.have_s) continue;
            rh_uniform_seg sg;
            std::memset(&sg, 0, sizeof sg);
            sg.src = g.conv[oc].get() + i * crowf + x.off_s;
            sg.dst = g.conv[nc].get() + i * crowf;
            sg.src_frames = sg.m1 = x.have_s;  // copied as a MONO stream, sample for sample: the count may be odd, and the sample behind the last one belongs
                                                // to a segment of this very launch
            sg.span_frames = UINT64_MAX;
            sg.from_rate = sg.to_rate = rate_;
            sg.from_ch = sg.to_ch = 1;
            sg.gain = 1.0f;
            table.push_back(sg);
            g.pmax_out = std::max(g.pmax_out, x.have_s);
        }
        // 2. pull and plan (one source per thread at a time; the segments join the table in source order)
        std::vector<std::vector<rh_uniform_seg>> planned(S);
        pull_sources(total / 2, S, [&](std::size_t i) {
            Src &x = g.srcs[i];
            if (x.ended) return;
            float *row = stage.get() + row_off[i];
            x.plan.begin_block();
            std::size_t fill = x.plan.held_samples();
            if (fill) std::memcpy(row, x.held.data(), fill * sizeof(float));
            std::vector<detail::UniformPlanner::Seg> segs;
            for (;;) {
                std::uint16_t ch = 0;
                std::uint32_t rate = 0;
                if (!x.reader.peek(ch, rate)) {
                    x.ended = true;
                    break;
                }
                const std::uint64_t now = (x.have_s + x.plan.out_samples() + g.sch - 1) / g.sch;
                if (now >= g.target) break;
                std::uint64_t need = 0, most = 0;
                const std::uint64_t slack = detail::UniformPlanner::close_slack_frames(rate, rate_);  // what the span's end may add
                x.plan.budget(rate, x.reader.opens_next(), g.target - now, g.crow > now + slack ? g.crow - now - slack : 0, need, most);
                // (whole frames that fit the row AND leave room for the up to ch - 1 samples of a frame the span's end cuts: spans of 37 samples
                // end inside a frame every time, and a row that was filled to its last frame took the cut samples into the next row)
                const std::uint64_t room = row_cap[i] - fill >= ch ? (row_cap[i] - fill - (ch - 1)) / ch : 0;
                const std::uint64_t n = std::min<std::uint64_t>(std::min(need, most), room);
                if (!n) break;
                detail::Piece pc;
                if (x.hint && !x.hint->chain) x.hint->log.note(x.hint->pulled, x.up->size_hint());
                const bool produced = x.reader.read_piece(row + fill, (std::size_t)n, pc);  // straight into the staging block
                if (produced) {
                    fill += pc.n;
                    x.plan.add(pc, segs);
                    if (x.hint && x.hint->counter) x.hint->counter->feed(pc, x.hint->pulled);
                    if (x.hint) x.hint->pulled += pc.n;
                }
                if (x.reader.ended()) {
                    x.ended = true;
                    if (x.hint && x.hint->counter) x.hint->counter->input_ended();
                    break;
                }
                if (!produced) break;
            }
            x.plan.end_block();
            x.held.assign(row + x.plan.keep_offset(), row + x.plan.keep_offset() + x.plan.keep_samples());
            for (const detail::UniformPlanner::Seg &sg : segs) {
                rh_uniform_seg t = sg.g;
                t.src = din.get() + row_off[i] + sg.src_off;
                t.dst = g.conv[nc].get() + i * crowf + x.have_s + sg.dst_off;
                t.gain = x.gain;
                planned[i].push_back(t);
            }
            x.have_s += x.plan.out_samples();
            x.total_s += x.plan.out_samples();
            if (x.have_s + g.sch - 1 > g.crow * g.sch) throw Error(RH_ERR_CAPACITY, "GpuMixer: converted frames exceed the row");
        });
        for (std::size_t i = 0; i < S; ++i)
            for (const rh_uniform_seg &t : planned[i]) {
                table.push_back(t);
                g.pmax_out = std::max<std::uint64_t>(g.pmax_out, t.m1 - t.m0);
            }
        // 3. one copy for all rows, on the copy stream: it runs beside the launches of the block before
        if (total) check(rh_memcpy_h2d(din.get(), stage.get(), total * sizeof(float), copy_stream_), "rh_memcpy_h2d");
        check(rh_event_record(g.copied[g.pslot].get(), copy_stream_), "rh_event_record");
    }
    void issue_block_staged(Gen &g) {
        const std::size_t S = g.srcs.size();
        detail::PinnedBuf &tabh = g.tab[g.pslot];
        const std::size_t crowf = ((std::size_t)g.crow * 2 + 3) & ~std::size_t(3);
        const std::vector<rh_uniform_seg> &table = g.ptable;
        const int nc = g.ccur ^ 1;
        // ... one conversion launch behind it
        check(rh_stream_wait_event(stream_, g.copied[g.pslot].get()), "rh_stream_wait_event");
        if (!table.empty()) {
            const std::size_t tf = table.size() * sizeof(rh_uniform_seg) / sizeof(float);
            tabh.reset(tf);
            g.dtab.reset(tf);
            std::memcpy(tabh.get(), table.data(), tf * sizeof(float));
            check(rh_memcpy_h2d(g.dtab.get(), tabh.get(), tf * sizeof(float), stream_), "rh_memcpy_h2d");
            check(rh_uniform_segments_dev(reinterpret_cast<const rh_uniform_seg *>(g.dtab.get()), (std::uint32_t)table.size(), g.pmax_out, stream_), "rh_uniform_segments_dev");
        }
        g.ccur = nc;
        // 4. filter + ordered sum of the converted rows
        std::vector<const float *> ptrs(S);
        std::vector<std::uint64_t> avail(S);
        std::vector<std::uint8_t> ended(S);
        bool all_ended = true;
        for (std::size_t i = 0; i < S; ++i) {
            Src &x = g.srcs[i];
            x.off_s = 0;
            ptrs[i] = g.conv[nc].get() + i * crowf;
            if (g.sch == 2 && x.ended && (x.have_s & 1)) {
                // the source's stream ends inside a frame: rodio's mixer adds its last sample and, at the next one, finds the source gone
                // (mixer.rs:185-198).  Adding +0.0 for the missing sample leaves the sum as it is -- through a filter it would not.
                if (g.filt.kind >= 0) throw Error(RH_ERR_UNSUPPORTED, "GpuMixer: a filtered source of 1, 2, 4 or 8 channels whose stream ends inside a frame (its spans do not hold whole frames: source/mod.rs:196-200)");
                check(rh_memset(g.conv[nc].get() + i * crowf + x.have_s, 0, sizeof(float), stream_), "rh_memset");
                x.have_s += 1;
            }
            avail[i] = x.have_s / g.sch;
            ended[i] = x.ended ? 1 : 0;
            all_ended = all_ended && x.ended;
        }
        std::uint64_t out = 0, consumed = 0;
        if (g.mono) {  // the mono mix of the block, then ChannelCountConverter(1 -> 2) behind the stereo queue (send_block takes channel 0 of it again)
            g.qm.reset(out_cap_frames_ * 2);
            check(rh_rlm_stream_block_v(g.plan, ptrs.data(), avail.data(), ended.data(), (std::uint32_t)S, g.qm.get(), out_cap_frames_ * 2 - g.fill - g.head, &out, &consumed, stream_),
                  "rh_rlm_stream_block_v");
            if (out) check(rh_channels_convert(g.queue_end(), g.qm.get(), (std::size_t)out, 1, 2, stream_), "rh_channels_convert");
        } else {
            check(rh_rlm_stream_block_v(g.plan, ptrs.data(), avail.data(), ended.data(), (std::uint32_t)S, g.queue_end(), out_cap_frames_ * 2 - g.fill - g.head, &out, &consumed, stream_),
                  "rh_rlm_stream_block_v");
        }
        g.fill += out;
        for (Src &x : g.srcs) {
            const std::uint64_t d = std::min(consumed, x.have_s / g.sch);
            x.off_s = d * g.sch;
            x.have_s -= d * g.sch;
        }
        g.done = all_ended;  // the call that saw every source ended emitted everything that was left
    }
    /// A generation of one format.  Row i of the page-locked block = [frames the previous block left unconsumed | one freshly
    /// pulled block]; a source that is not in the layout the fused launch reads has its row in the side block, in its own layout.
    std::size_t row_floats(const Gen &g) const { return g.mono ? ((cap_frames_ + 3) & ~std::size_t(3)) : row_; }
    void pull_block_direct(Gen &g) {
        const std::size_t S = g.srcs.size();
        detail::PinnedBuf &stage = g.stage[g.pslot], &side = g.side[g.pslot];
        const std::size_t native = g.mono ? 1 : 2;  // channels of the rows the fused launch reads
        const std::size_t row_ = row_floats(g);
        stage.reset(S * row_);
        detail::DeviceBuf &din = g.din[g.pd], &dside = g.dside[g.pslot];
        din.reset(S * row_);
        g.pptrs.assign(S, nullptr);
        g.pavail.assign(S, 0);
        g.pended.assign(S, 0);
        g.pside_off.assign(S, 0);
        g.pside = 0;
        for (std::size_t i = 0; i < S; ++i)
            if (g.srcs[i].ch != native) {
                g.pside_off[i] = g.pside;
                g.pside += (cap_frames_ * g.srcs[i].ch + 3) & ~std::size_t(3);
            }
        if (g.pside) {
            side.reset(g.pside);
            dside.reset(g.pside);
        }
        pull_sources(S * opt_.block_frames, S, [&](std::size_t i) {
            Src &x = g.srcs[i];
            if (x.dev) return;  // its block arrives device-to-device, below
            const std::size_t ch = x.ch;
            float *row = ch == native ? stage.get() + i * row_ : side.get() + g.pside_off[i];
            std::size_t have = x.held.size();
            if (have / ch + (x.ended ? 0 : opt_.block_frames) > cap_frames_) throw Error(RH_ERR_CAPACITY, "GpuMixer: held frames exceed the plan");
            if (have) std::memcpy(row, x.held.data(), have * sizeof(float));
            if (!x.ended) {
                const std::size_t want = opt_.block_frames * ch;
                if (x.hint && !x.hint->chain) x.hint->log.note(x.hint->pulled, x.up->size_hint());
                std::size_t got = x.up->read(row + have, want);  // straight into the staging block
                x.ended = got < want;
                x.note_pull(got, x.ended);
                // sources end on frame boundaries (source/mod.rs:169-178); one that reports no spans and ends inside a frame all the same is
                // refused rather than shortened (rodio would convert the cut frame's samples; span-reporting sources take the staged path, which does)
                if (got % ch) throw Error(RH_ERR_UNSUPPORTED, "GpuMixer: a source whose current_span_len() is None ended inside a frame (source/mod.rs:169-178 asks for whole frames)");
                have += got;
                x.total_s += got / ch * 2;
            }
            g.pptrs[i] = din.get() + i * row_;
            g.pavail[i] = have / ch;
            g.pended[i] = x.ended ? 1 : 0;
        });
        // one copy for all rows -- what the longest row holds of every row: the rows differ by a few frames, the unused ends stay --
        // on the copy stream: it runs beside the launches of the block before
        std::size_t width = 0;
        for (std::size_t i = 0; i < S; ++i)
            if (g.srcs[i].ch == native) width = std::max<std::size_t>(width, (std::size_t)g.pavail[i] * native);
        if (width) check(rh_memcpy_h2d_rows(din.get(), stage.get(), row_ * sizeof(float), width * sizeof(float), S, copy_stream_), "rh_memcpy_h2d_rows");
        if (g.pside) check(rh_memcpy_h2d(dside.get(), side.get(), g.pside * sizeof(float), copy_stream_), "rh_memcpy_h2d");
        // chains that hand their blocks over on the device: [what the converter left of the block before | the chain's next samples],
        // device-to-device on the copy stream, behind the pitched copy (which also moved the rows' unused staging bytes)
        for (std::size_t i = 0; i < S; ++i) {
            Src &x = g.srcs[i];
            if (!x.dev) continue;
            float *row = din.get() + i * row_;
            std::size_t have = (std::size_t)x.dheld * 2;
            if (have / 2 + (x.ended ? 0 : opt_.block_frames) > cap_frames_) throw Error(RH_ERR_CAPACITY, "GpuMixer: held frames exceed the plan");
            if (have) check(rh_memcpy_d2d(row, g.din[g.pd_prev].get() + i * row_ + x.dheld_off * 2, have * sizeof(float), copy_stream_), "rh_memcpy_d2d");
            if (!x.ended) {
                const std::size_t want = opt_.block_frames * 2;
                std::size_t got = x.dev->read_device(row + have, want, copy_stream_);
                x.ended = got < want;
                x.total_s += got;
                x.note_pull(got, x.ended);
                if (x.ended && x.hint) x.hint->total_known = true, x.hint->total = x.total_s;
                if (got % 2) {  // the chain's stream ends inside a frame (want is whole frames): its last sample is mixed, the missing one is +0.0
                    check(rh_memset(row + have + got, 0, sizeof(float), copy_stream_), "rh_memset");
                    got += 1;
                }
                have += got;
            }
            g.pptrs[i] = row;
            g.pavail[i] = have / 2;
            g.pended[i] = x.ended ? 1 : 0;
        }
        check(rh_event_record(g.copied[g.pslot].get(), copy_stream_), "rh_event_record");
    }
    void issue_block_direct(Gen &g) {
        const std::size_t S = g.srcs.size();
        detail::PinnedBuf &stage = g.stage[g.pslot], &side = g.side[g.pslot];
        const std::size_t native = g.mono ? 1 : 2;
        const std::size_t row_ = row_floats(g);
        check(rh_stream_wait_event(stream_, g.copied[g.pslot].get()), "rh_stream_wait_event");
        if (g.pside)  // ChannelCountConverter on the device (channels.rs:57-85), into the rows the fused launch reads
            for (std::size_t i = 0; i < S; ++i)
                if (g.srcs[i].ch != native && g.pavail[i])
                    check(rh_channels_convert(g.din[g.pd].get() + i * row_, g.dside[g.pslot].get() + g.pside_off[i], (std::size_t)g.pavail[i], g.srcs[i].ch, 2, stream_), "rh_channels_convert");
        std::uint64_t out = 0, consumed = 0;
        if (debug_poison()) check(rh_memset(g.queue_end(), 0xff, (out_cap_frames_ * 2 - g.fill - g.head) * 2 * sizeof(float), stream_), "rh_memset");
        if (g.mono) {  // the mono mix of the block, then ChannelCountConverter(1 -> 2) behind the stereo queue (one pass over the MIX, not per source)
            g.qm.reset(out_cap_frames_ * 2);
            check(rh_rlm_stream_block_v(g.plan, g.pptrs.data(), g.pavail.data(), g.pended.data(), (std::uint32_t)S, g.qm.get(), out_cap_frames_ * 2 - g.fill - g.head, &out, &consumed, stream_),
                  "rh_rlm_stream_block_v");
            if (out) check(rh_channels_convert(g.queue_end(), g.qm.get(), (std::size_t)out, 1, 2, stream_), "rh_channels_convert");
        } else {
            check(rh_rlm_stream_block_v(g.plan, g.pptrs.data(), g.pavail.data(), g.pended.data(), (std::uint32_t)S, g.queue_end(), out_cap_frames_ * 2 - g.fill - g.head, &out, &consumed, stream_),
                  "rh_rlm_stream_block_v");
        }
        g.fill += out;
        bool all_ended = true;
        for (std::size_t i = 0; i < S; ++i) {  // keep what the converter has not consumed (a few hundred frames)
            Src &x = g.srcs[i];
            if (x.dev) {  // ... which stays where it is: the next block copies it from this block's row
                x.dheld_off = std::min<std::uint64_t>(consumed, g.pavail[i]);
                x.dheld = g.pavail[i] - x.dheld_off;
                all_ended = all_ended && x.ended;
                continue;
            }
            const std::size_t ch = x.ch;
            const float *row = ch == native ? stage.get() + i * row_ : side.get() + g.pside_off[i];
            const std::size_t have = (std::size_t)g.pavail[i] * ch, drop = std::min<std::size_t>((std::size_t)consumed * ch, have);
            x.held.assign(row + drop, row + have);
            all_ended = all_ended && x.ended;
        }
        g.done = all_ended;  // the call that saw every source ended emitted everything that was left
    }

    /// Mixer::add on a running mixer.  rodio admits the source at the next frame boundary of the output (mixer.rs:175-183).
    /// Here up to two blocks are already mixed beyond that frame (the one being served, the one in flight), so the new
    /// source -- its own generation, its own fused stream and clock from frame J on -- is run ahead synchronously until it
    /// covers them, added onto their device copies at its offset (rh_mix_sum: old mix first, the newcomer last = insertion
    /// order, bit for bit without a filter) and the blocks travel to the host again.  From the next block on it is one more
    /// generation.
    // Mixer::add on a running mixer admits the source at the next frame of the mix (late_join) -- unless the mix is about to END inside a
    // frame (its last block is cut to what rodio's sources cover) and the consumer is already in that frame: rodio's mixer then answers None
    // for the rest of it (no source is left, and a pending one only starts on a frame boundary: mixer.rs:120-136,175-183) and goes on
    // with the new source afterwards.  That is what an ENDED mixer does with a pending source (can_resume()), so the source waits for that.
    bool joins_a_running_mix() {
        if (!running()) return false;
        const int ci = cur_index();
        const bool flight = other_in_flight();
        const int li = flight ? ci ^ 1 : ci;
        const Slot &ls = flight ? other() : cur();
        if (!ls.last || ls.n == (std::size_t)slot_frames_[li] * out_ch_) return true;
        const std::uint64_t consumed = slot_base_[ci] * out_ch_ + position();
        return (consumed + out_ch_ - 1) / out_ch_ < slot_base_[li] + slot_frames_[li];
    }
    void late_join(Src item) {
        const int ci = cur_index();
        const std::uint64_t consumed = slot_base_[ci] * out_ch_ + position();  // samples already handed out
        const std::uint64_t J = (consumed + out_ch_ - 1) / out_ch_;             // the next frame boundary
        const bool flight = other_in_flight();
        const int li = flight ? ci ^ 1 : ci;                               // the last block that is scheduled
        const std::uint64_t sched_end = slot_base_[li] + slot_frames_[li];
        check(rh_stream_synchronize(stream_), "rh_stream_synchronize");   // the blocks about to be patched have been produced
        const bool staged = !wide() && !item.dev && spanned(item);
        if (!staged && !wide()) make_direct(item);
        std::vector<Src> one;
        one.push_back(std::move(item));
        if (wide()) start_stream_wide(std::move(one), J);
        else start_stream(std::move(one), staged);
        last_join_ = J;
        Gen &g = *gens_.back();
        g.join = J;
        for (Src &x : g.srcs)
            if (x.hint) x.hint->join_frame = J;
        const std::uint64_t need = sched_end > J ? sched_end - J : 0;
        if (need + 64 > out_cap_frames_) {
            // the newcomer catches up with everything that is scheduled -- up to two blocks, and a block of a generation that ran behind a
            // full queue can hold nearly two of the newcomer's own -- and its last block of that may reach a block beyond: the queues grow
            out_cap_frames_ = need + 64;
            for (auto &other : gens_)
                for (auto &b : other->q) grow_keep(b, out_cap_frames_ * 2 * qch_, (other->head + other->fill) * qch_);
        }
        while (g.fill < need && !g.done) {
            run_block(g, cur());
            check(rh_stream_synchronize(stream_), "rh_stream_synchronize");  // the staging blocks alternate: never more than one copy in flight here
        }
        for (int k = 0; k < (flight ? 2 : 1); ++k) {
            const int si = k == 0 ? ci : ci ^ 1;
            Slot &sl = k == 0 ? cur() : other();
            const std::uint64_t b0 = slot_base_[si], b1 = b0 + slot_frames_[si];
            const std::uint64_t lo = std::max(J, b0), hi = std::min(b1, J + g.fill);
            if (hi <= lo) continue;
            const float *ptrs[2] = {dkeep_[si].get(), g.queue() + (lo - J) * qch_};
            const std::uint64_t start[2] = {0, (lo - b0) * qch_}, len[2] = {slot_frames_[si] * qch_, (hi - lo) * qch_};
            dmix_.reset(out_cap_frames_ * 2 * qch_);
            check(rh_mix_sum(dmix_.get(), slot_frames_[si] * qch_, ptrs, start, len, 2, stream_), "rh_mix_sum");
            check(rh_memcpy_d2d(dkeep_[si].get(), dmix_.get(), slot_frames_[si] * qch_ * sizeof(float), stream_), "rh_memcpy_d2d");
            send_block(sl, dkeep_[si].get(), slot_frames_[si]);
        }
        {  // the newcomer's queue moves on to the frame the next block starts at
            const std::uint64_t used = std::min(g.fill, need), rem = g.fill - used, pad = rem & 1;
            if (rem) check(rh_memcpy_d2d(g.q[g.cur ^ 1].get() + pad * qch_, g.queue() + used * qch_, rem * qch_ * sizeof(float), stream_), "rh_memcpy_d2d");
            g.cur ^= 1;
            g.head = pad;
            g.fill = rem;
        }
        check(rh_stream_synchronize(stream_), "rh_stream_synchronize");
        block_done();
        // a mixer that was about to end goes on: the block that carried the end mark loses it, and if nothing was in flight the
        // next block is requested now
        Slot &last_slot = flight ? other() : cur();
        const bool plays_on = !(g.done && g.fill == 0);
        if (last_slot.last && last_slot.n < (std::size_t)slot_frames_[li] * out_ch_) {
            // ... and a mix that was about to end INSIDE its last frame (the block was cut to the samples rodio's sources cover; the device copy
            // holds the whole frame, +0.0 where nothing played) has that frame completed by a newcomer that plays beyond it, or that ends in it
            const std::size_t whole = (std::size_t)slot_frames_[li] * out_ch_;
            if (plays_on) last_slot.n = whole;
            else if (J + g.emitted == sched_end) last_slot.n = std::max(last_slot.n, whole - (g.last_valid ? qch_ - g.last_valid : 0));
        }
        if (last_slot.last && plays_on) {
            last_slot.last = false;
            if (!flight) submit(other());
        }
    }
    static bool debug_poison() {  // RODIO_HIP_DEBUG_POISON=1: every buffer a block passes through is filled with NaN patterns first
        static const bool on = std::getenv("RODIO_HIP_DEBUG_POISON") != nullptr;
        return on;
    }
    std::uint32_t rate_;
    Options opt_;
    std::uint16_t out_ch_ = 2;       // mixer::mixer(channels, ..)
    std::size_t qch_ = 2;            // channels of the frames the mix is formed in: 2 (the fused kernel's), or the mixer's own when it has more
    bool device_chains_ = false;     // a chain hands its blocks over on the device: its scan kernels' failure word is read per block
    std::unique_ptr<Reaper> reaper_;
    ChainStats retired_chains_;
    std::uint64_t wide_fused_blocks_ = 0;
    detail::DeviceBuf dout_;         // the mixed block in the mixer's channel layout (channels != 2)
    std::vector<Src> pending_;
    std::vector<std::unique_ptr<Gen>> gens_;
    // the threads that pull a block's sources: made at the first block that is worth them
    detail::Workers &pullers(std::size_t frames) {
        if (opt_.host_threads == 1 || frames < (std::size_t(1) << 18)) return solo_;
        if (!pool_) pool_.reset(new detail::Workers(opt_.host_threads ? opt_.host_threads : std::min(16u, std::max(1u, std::thread::hardware_concurrency()))));
        return *pool_;
    }
    detail::Workers solo_{1};
    rh_stream copy_stream_ = nullptr;  // host-to-device copies of the staged rows: the link stays busy while stream_ runs the block before
    std::unique_ptr<detail::Workers> pool_;
    std::size_t cap_frames_ = 0, row_ = 0;
    std::uint64_t out_cap_frames_ = 0, scheduled_ = 0, last_join_ = 0;
    detail::DeviceBuf dmix_, dkeep_[2];                       // scratch of rh_mix_sum; device copies of the two scheduled blocks
    std::uint64_t slot_base_[2] = {0, 0}, slot_frames_[2] = {0, 0};  // mixer frame of a scheduled block's first frame; its length
    std::uint64_t calls_ = 0;                                 // next() calls so far (MixerSource's channel position, mod 2)
    bool resume_ok_ = true;
};

}  // namespace rodio_hip
#endif  // RODIO_HIP_HPP
