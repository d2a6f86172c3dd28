//! rodio-hip -- rodio's per-sample DSP hot path on AMD MI355X (gfx950), as `Source` adapters.
//!
//! rodio has no FFI: its extension point is `trait Source: Iterator<Item = f32>` (`src/source/mod.rs:179-218` of rodio 0.22).
//! Anything that implements it can be handed to `Mixer::add` (`src/mixer.rs:58-66`), `Player::append` (`src/player.rs:104-108`)
//! or `queue.append` (`src/queue.rs:62-70`).  This crate is that boundary: pull-model adapters that own their upstream by value,
//! pre-pull a block, run it through `librodio_hip.so` (`include/rodio_hip.h`, declared in [`ffi`]) and serve `next()` from
//! page-locked memory, one block ahead.
//!
//! * [`GpuSource<I>`] -- one upstream and a chain of adapters built with rodio's method names (`amplify`, `low_pass`, `reverb`,
//!   `limit`, `automatic_gain_control`, `uniform`, ...), executed block-wise on the device; adapter memory is carried across
//!   blocks, so any block size gives the samples of one pass.
//! * [`GpuMixer`] -- `mixer::mixer(channels, rate)` where every added source goes through `UniformSourceIterator::new(src.amplify(g),
//!   .., rate)` (`Mixer::add` wraps every source in one, `mixer.rs:58-66`) and ITS OWN filter (`add_filtered`: rodio's sources carry
//!   their adapters into the mixer) and the ordered sum: the fused kernel, block by block, one fused stream per (rate, filter).
//!   Sources may be added while the mixer is playing (they join at the next frame, `mixer.rs:175-183`).  A [`GpuSource`] chain handed
//!   over with `add_chain` keeps its blocks in device memory: the mixer takes them device-to-device.  `prepare()` starts the stream on
//!   the calling (control) thread, a reaper thread frees what has ended: the consumer's `next()` neither starts nor tears down.
//!
//! Both pull their upstream the way `UniformSourceIterator` does (`src/source/uniform.rs:50-97`): `current_span_len()` is asked
//! whenever the converter chain has run dry, `min(span, 32768)` samples go to a FRESH converter pair, and every span ends with
//! its last frame verbatim -- a `SamplesBuffer` or a decoder comes out span by span, a generator as one continuous stream.
//!
//! There is no CPU compute path: every arithmetic operation on samples happens in the library.  One object is used from one
//! thread at a time (`Send`, not `Sync`, like every rodio source).
//!
//! This is the Rust twin of `include/rodio_hip.hpp` (C++17, compiled and tested in the repository: `tests/test_host_mirror.py`),
//! method for method (`tests/test_rust_decls.py` compares the two); the image the library was developed in has no Rust toolchain,
//! so this crate has been checked against the header and against its twin but not against `rustc`.

pub mod ffi;

use ffi::*;
use rodio::source::SeekError;
use rodio::{ChannelCount, SampleRate, Source};
use std::ptr;
use std::sync::{mpsc, Arc, Mutex};
use std::time::{Duration, Instant};

// ------------------------------------------------------------------------------------------------ errors ----
/// A library call that failed (`rh_status != RH_OK`).  `Iterator::next` cannot return it: the adapters panic with it, as rodio's own
/// adapters do on broken invariants -- never a silent CPU path.
#[derive(Debug, Clone)]
pub struct RhError {
    pub status: RhStatus,
    pub what: &'static str,
}
impl std::fmt::Display for RhError {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        let s = unsafe { std::ffi::CStr::from_ptr(rh_status_string(self.status)) };
        write!(f, "{}: {}", self.what, s.to_string_lossy())
    }
}
impl std::error::Error for RhError {}
fn ck(status: RhStatus, what: &'static str) {
    if status != RH_OK {
        panic!("{}", RhError { status, what });
    }
}
/// Binds the library to a gfx950 device (once per process; also re-reads the library's tuning variables).
pub fn init(device: i32) -> Result<(), RhError> {
    let status = unsafe { rh_init(device) };
    if status == RH_OK { Ok(()) } else { Err(RhError { status, what: "rh_init" }) }
}

// ------------------------------------------------------------------------------------------------ plumbing ----
struct DeviceBuf { p: *mut f32, n: usize }
impl DeviceBuf {
    fn new() -> Self { DeviceBuf { p: ptr::null_mut(), n: 0 } }
    /// Grows to at least `floats` (the contents are NOT kept).
    fn reserve(&mut self, floats: usize) {
        if floats <= self.n { return; }
        unsafe {
            if !self.p.is_null() { ck(rh_free(self.p.cast()), "rh_free"); }
            let mut q: *mut core::ffi::c_void = ptr::null_mut();
            ck(rh_malloc(&mut q, floats * 4), "rh_malloc");
            self.p = q.cast();
        }
        self.n = floats;
    }
}
impl Drop for DeviceBuf { fn drop(&mut self) { if !self.p.is_null() { unsafe { rh_free(self.p.cast()); } } } }
// The buffers are plain owners of device / page-locked memory: moving one to another thread (a stage closure, the reaper) is fine,
// and nothing mutates through a shared reference.  (Edition-2021 closures capture disjoint FIELDS: without these impls a closure
// that names `buf.p` captures a bare `*mut f32`, which is not Send.)
unsafe impl Send for DeviceBuf {}
unsafe impl Sync for DeviceBuf {}
struct PinnedBuf { p: *mut f32, n: usize }
impl PinnedBuf {
    fn new() -> Self { PinnedBuf { p: ptr::null_mut(), n: 0 } }
    fn reserve(&mut self, floats: usize) {
        if floats <= self.n { return; }
        unsafe {
            if !self.p.is_null() { ck(rh_host_free(self.p.cast()), "rh_host_free"); }
            let mut q: *mut core::ffi::c_void = ptr::null_mut();
            ck(rh_host_alloc(&mut q, floats * 4), "rh_host_alloc");
            self.p = q.cast();
        }
        self.n = floats;
    }
    fn slice(&self, n: usize) -> &[f32] { unsafe { std::slice::from_raw_parts(self.p, n) } }
    fn slice_mut(&mut self, n: usize) -> &mut [f32] { unsafe { std::slice::from_raw_parts_mut(self.p, n) } }
}
impl Drop for PinnedBuf { fn drop(&mut self) { if !self.p.is_null() { unsafe { rh_host_free(self.p.cast()); } } } }
unsafe impl Send for PinnedBuf {}
unsafe impl Sync for PinnedBuf {}
/// A HIP event (`rh_event_create`), destroyed with its owner.
struct Event(*mut core::ffi::c_void);
impl Event {
    fn new() -> Self { let mut e = ptr::null_mut(); ck(unsafe { rh_event_create(&mut e) }, "rh_event_create"); Event(e) }
}
impl Drop for Event { fn drop(&mut self) { if !self.0.is_null() { unsafe { rh_event_destroy(self.0); } } } }
unsafe impl Send for Event {}
/// A raw pointer that may travel to a scoped pull thread (every thread gets rows of its own).
#[derive(Clone, Copy)]
struct SendPtr(*mut f32);
unsafe impl Send for SendPtr {}
unsafe impl Sync for SendPtr {}

/// Bulk form of `next()`: what a block adapter pulls with.
fn read_into(src: &mut dyn Source, dst: &mut [f32]) -> usize {
    let mut k = 0;
    while k < dst.len() {
        match src.next() { Some(v) => { dst[k] = v; k += 1; } None => break }
    }
    k
}

// ------------------------------------------------------------------------------------------------ spans ----
/// One run of samples pulled from a source inside ONE span of it.
#[derive(Clone, Copy, Debug)]
pub struct Piece {
    pub n: usize,            // samples: whole frames of `ch` channels, and behind them `tail` samples of a frame the span's end cuts
    pub opens: bool, pub closes: bool, pub ch: u16, pub rate: u32,
    pub tail: usize,         // closes only: samples of a CUT last frame (uniform.rs:56: `.min(32768)` cuts frames of 3, 5, 6, 7 channels)
    pub by_none: bool,       // closes only: the span ended because the source returned None, not because its samples were counted out
    pub limit: Option<usize>,  // opens only: what `Take` admits to the span's chain, min(current_span_len(), 32768) (uniform.rs:56); None: the source reports no spans
}

/// Pulls a source the way `UniformSourceIterator` does (`uniform.rs:50-97`): whenever its converter chain has run dry it asks
/// `current_span_len()`, `channels()` and `sample_rate()` -- in that order, at exactly that position of the stream -- and admits
/// `min(span, 32768)` samples (`Take`, `uniform.rs:56,148-178`) to the chain it builds for them.  A span ends when that many
/// samples were taken or when the source returns `None`; the stream ends when a fresh chain yields nothing.
#[derive(Default)]
pub struct SpanReader { open: bool, fresh: bool, ended: bool, left: usize, ch: u16, rate: u32, limit: Option<usize> }
const OPEN_ENDED: usize = usize::MAX;
impl SpanReader {
    pub fn new() -> Self { SpanReader { fresh: true, ..Default::default() } }
    pub fn ended(&self) -> bool { self.ended }
    pub fn opens_next(&self) -> bool { self.fresh }
    /// Format of the span the next `read_piece` continues or opens; `None` at the end of the stream.
    pub fn peek(&mut self, up: &mut dyn Source) -> Option<(u16, u32)> {
        if !self.open && !self.bootstrap(up) { return None; }
        Some((self.ch, self.rate))
    }
    fn bootstrap(&mut self, up: &mut dyn Source) -> bool {                       // uniform.rs:50-68
        let span = up.current_span_len();
        self.ch = up.channels().get();
        self.rate = up.sample_rate().get();
        if span == Some(0) { self.ended = true; return false; }                  // Take{n: 0}: the chain is empty, next() is None
        self.left = span.map(|s| s.min(32768)).unwrap_or(OPEN_ENDED);
        self.limit = span.map(|s| s.min(32768));
        // source/mod.rs:196-200 asks for spans of whole frames; `.min(32768)` breaks that for 3, 5, 6, 7 ... channels: the chain rodio
        // builds for such a span ends inside a frame and the next one starts there -- every later span has its channels rotated.  The
        // reader hands the cut frame's samples over with the span (`Piece::tail`) and goes on at the sample behind them, as rodio's
        // source does; what the converters make of a cut frame is `UniformPlanner::add`'s business.
        self.open = true;
        self.fresh = true;
        true
    }
    /// Up to `max_frames` frames of the current span into `dst`; `None`: the stream is over and nothing was produced.
    pub fn read_piece(&mut self, up: &mut dyn Source, dst: &mut [f32], max_frames: usize) -> Option<Piece> {
        if self.ended || (!self.open && !self.bootstrap(up)) { return None; }
        let ch = self.ch as usize;
        let mut want = max_frames.saturating_mul(ch).min(self.left).min(dst.len());
        want -= want % ch;
        if self.left != OPEN_ENDED && self.left - want < ch && self.left <= dst.len() { want = self.left; }   // the rest of the span is a cut frame: its samples belong to this span's chain
        let mut got = if want > 0 { read_into(up, &mut dst[..want]) } else { 0 };
        let none = got < want;                                                    // the source returned None inside the span
        if self.left != OPEN_ENDED { self.left -= got; }
        let closes = none || self.left == 0;
        let tail = if closes { got % ch } else { 0 };                             // (what becomes of it is the planner's business: it knows the target format)
        if !closes { got -= got % ch; }
        let piece = Piece { n: got, opens: self.fresh, closes, ch: self.ch, rate: self.rate, tail, by_none: none, limit: if self.fresh { self.limit } else { None } };
        let produced = got != 0 || (closes && !self.fresh);                       // a span that had samples before ends here: its last frame is due
        if got != 0 { self.fresh = false; }
        if closes { self.open = false; }
        if none { self.ended = true; }                                            // the chain rodio builds next yields nothing: None
        if produced { Some(piece) } else { None }
    }
    /// After a seek of the source: what was pulled ahead is gone, the next read builds a fresh chain.
    pub fn restart(&mut self) { self.open = false; self.ended = false; }
}

/// One segment `rh_uniform_segments` converts, with offsets instead of pointers.
pub struct PlannedSeg { pub src_off: usize, pub dst_off: usize, pub g: RhUniformSeg }

/// Turns the pieces of one source into segments (`UniformSourceIterator::new(src, to_ch, to_rate)`, span by span).  The planner
/// only counts: the owner lays the samples out as one row `[frames kept from the previous block | this block's pieces]`, hands
/// every piece to `add` in order, and keeps `[keep_offset, keep_offset + keep_samples)` of the row for the next block.
pub struct UniformPlanner {
    to_ch: u16, to_rate: u32,
    span_in: u64, span_m: u64, row_frame0: u64, next_frame0: u64,
    row_off: usize, pos: usize, held: usize, keep_off: usize, keep_n: usize,
    out: usize,              // output SAMPLES planned in this block: what comes out is a stream of samples, not of frames (a span that ends inside a frame)
}
impl UniformPlanner {
    pub fn new(to_ch: u16, to_rate: u32) -> Self {
        UniformPlanner { to_ch, to_rate, span_in: 0, span_m: 0, row_frame0: 0, next_frame0: 0, row_off: 0, pos: 0, held: 0, keep_off: 0, keep_n: 0, out: 0 }
    }
    pub fn begin_block(&mut self) { self.pos = self.held; self.row_off = 0; self.out = 0; self.keep_off = 0; self.keep_n = self.held; }
    pub fn held_samples(&self) -> usize { self.held }
    pub fn out_samples(&self) -> usize { self.out }
    /// Output frames a span's end can add beyond what `budget` counts: its verbatim last frame, or what the converters make of a cut frame.
    pub fn close_slack_frames(rate: u32, to_rate: u32) -> u64 { to_rate as u64 / rate as u64 + 3 }
    pub fn keep_offset(&self) -> usize { self.keep_off }
    pub fn keep_samples(&self) -> usize { self.keep_n }
    fn span_frames(&self, n: u64, rate: u32, complete: bool) -> u64 {
        let mut r = 0u64;
        ck(unsafe { rh_uniform_span_frames(n, rate, self.to_rate, complete as i32, &mut r) }, "rh_uniform_span_frames");
        r
    }
    fn first_tap(&self, m: u64, rate: u32) -> u64 {
        let mut i = 0u64;
        ck(unsafe { rh_uniform_first_tap(m, rate, self.to_rate, &mut i) }, "rh_uniform_first_tap");
        i
    }
    /// Further input frames of the open (or a fresh) span that produce at least `want` output frames, and the most that cannot
    /// produce more than `room` (verbatim last frame included).
    pub fn budget(&self, rate: u32, fresh: bool, want: u64, room: u64) -> (u64, u64) {
        let (inn, m) = if fresh { (0, 0) } else { (self.span_in, self.span_m) };
        let need = self.first_tap(m + want, rate) + 2;
        let most = if room > 0 { self.first_tap(m + room - 1, rate) + 1 } else { inn };
        (need.saturating_sub(inn), most.saturating_sub(inn))
    }
    pub fn add(&mut self, p: &Piece, segs: &mut Vec<PlannedSeg>) {
        if p.opens { self.span_in = 0; self.span_m = 0; self.row_off = self.pos; self.row_frame0 = 0; }
        self.span_in += ((p.n - p.tail) / p.ch as usize) as u64;
        self.pos += p.n;
        // A span that ends inside a frame (uniform.rs:56: `.min(32768)` on 3, 5, 6, 7 channels; a source that returns None inside a frame).
        // rodio's SampleRateConverter meets a SHORT frame: every output frame that lerps towards it is cut to its length (zip,
        // sample_rate.rs:174-179), the short frame itself comes out verbatim when an output lands on it (:193-200), and the
        // ChannelCountConverter behind regroups those runs into frames of `from` samples (channels.rs:57-85) -- reproduced sample for sample
        // by a segment of its own (`reserved` = the cut frame's samples); the whole frames in front of it convert as the frames of a span
        // that is still open.  The next span starts at the sample behind the cut, its channels rotated, as in rodio.
        let cut = p.closes && p.tail > 0;
        let ready = self.span_frames(self.span_in, p.rate, p.closes && !cut);
        let seg = |src_frame0: u64, src_frames: u64, m0: u64, m1: u64, span_frames: u64, reserved: u32, to_rate: u32, to_ch: u16| RhUniformSeg {
            src: ptr::null(), dst: ptr::null_mut(), src_frame0, src_frames, m0, m1, span_frames,
            from_rate: p.rate, to_rate, from_ch: p.ch as u32, to_ch: to_ch as u32, gain: 1.0, reserved,
        };
        if ready > self.span_m {
            let g = seg(self.row_frame0, self.span_in - self.row_frame0, self.span_m, ready, if p.closes && !cut { self.span_in } else { u64::MAX }, 0, self.to_rate, self.to_ch);
            segs.push(PlannedSeg { src_off: self.row_off, dst_off: self.out, g });
            self.out += (ready - self.span_m) as usize * self.to_ch as usize;
            self.span_m = ready;
        }
        if cut {
            let mut tail_out = 0u64;
            ck(unsafe { rh_uniform_cut_tail_samples(self.span_in, p.tail as u32, p.rate, self.to_rate, p.ch as u32, self.to_ch as u32, &mut tail_out) }, "rh_uniform_cut_tail_samples");
            if tail_out > 0 {
                // the frame in front of the cut is in the row whenever an output lerps towards the cut frame (it is that output's first tap)
                let have_last = self.span_in >= 1 && self.span_in - 1 >= self.row_frame0;
                let f0 = if have_last { self.span_in - 1 } else { self.span_in };
                let g = seg(f0, self.span_in - f0, 0, tail_out, self.span_in, p.tail as u32, self.to_rate, self.to_ch);
                segs.push(PlannedSeg { src_off: self.row_off + (f0 - self.row_frame0) as usize * p.ch as usize, dst_off: self.out, g });
                self.out += tail_out as usize;
            }
        }
        if p.closes {
            self.held = 0; self.keep_off = 0; self.keep_n = 0;
        } else {                                                                  // the frames the span's next output frame reads first stay
            let first = self.first_tap(self.span_m, p.rate).max(self.row_frame0).min(self.span_in);
            self.keep_off = self.row_off + (first - self.row_frame0) as usize * p.ch as usize;
            self.keep_n = (self.span_in - first) as usize * p.ch as usize;
            self.held = self.keep_n;
            self.next_frame0 = first;
        }
    }
    pub fn end_block(&mut self) { self.row_frame0 = if self.held > 0 { self.next_frame0 } else { self.span_in }; }
}

// ------------------------------------------------------------------------------------------------ block pump ----
/// Where the host's time went, and what crossed to the host (`detail::BlockPump::Timing` of rodio_hip.hpp).
#[derive(Clone, Copy, Default, Debug)]
pub struct Timing {
    pub submit_s: f64, pub prefetch_s: f64, pub pull_s: f64, pub wait_s: f64, pub blocks: u64,
    pub d2h_samples: u64,      // samples of processed blocks copied to the host (0 for a source that keeps its blocks on the device)
    pub device_samples: u64,   // samples handed to a consumer device-to-device (`read_device`)
    pub first_advance_s: f64,  // the advance that started the stream (`prepare()`, or the first `next()`)
}
struct Slot {
    stage: PinnedBuf, out: PinnedBuf, n: usize, last: bool, done: Event,
    dev: DeviceBuf, taken: Event, taken_pending: bool,   // keep_blocks_on_device(): the processed block on the device; recorded on the consumer's stream behind its copies
}
/// Two page-locked result blocks, one served while the other is in flight.
struct Pump { slot: [Slot; 2], cur: usize, pos: usize, skip: usize, handed_out: u64, primed: bool, ended: bool, stream: RhStream, device_out: bool, timing: Timing }
unsafe impl Send for Pump {}
impl Pump {
    fn new() -> Self {
        let mut stream: RhStream = ptr::null_mut();
        ck(unsafe { rh_stream_create(&mut stream) }, "rh_stream_create");
        let mk = || Slot { stage: PinnedBuf::new(), out: PinnedBuf::new(), n: 0, last: false, done: Event::new(), dev: DeviceBuf::new(), taken: Event::new(), taken_pending: false };
        Pump { slot: [mk(), mk()], cur: 0, pos: 0, skip: 0, handed_out: 0, primed: false, ended: false, stream, device_out: false, timing: Timing::default() }
    }
    fn running(&self) -> bool { self.primed && !self.ended }
    fn other_in_flight(&self) -> bool { self.primed && !self.ended && !self.slot[self.cur].last }
    /// Forget everything pulled and processed ahead (after a seek); the stream that follows resumes at the channel the consumer is at.
    fn restart(&mut self, keep_phase: usize) {
        self.skip = if keep_phase > 0 { (self.handed_out % keep_phase as u64) as usize } else { 0 };
        ck(unsafe { rh_stream_synchronize(self.stream) }, "rh_stream_synchronize");
        for s in &mut self.slot { s.n = 0; s.last = false; }
        self.cur = 0; self.pos = 0; self.primed = false; self.ended = false;
    }
}
impl Drop for Pump {
    fn drop(&mut self) {
        unsafe {
            rh_stream_synchronize(self.stream);
            for s in &self.slot { if s.taken_pending { rh_event_synchronize(s.taken.0); } }   // a consumer's copy may still read the slot's device block
            rh_stream_destroy(self.stream);
        }
    }
}
/// What a block source implements; `advance` is the shared `next()` machinery of `detail::BlockPump` in rodio_hip.hpp.
trait BlockSource {
    fn pump(&mut self) -> &mut Pump;
    /// Fills slot `i` (n, last) and enqueues everything that produces its `out` block on the stream.
    fn enqueue(&mut self, i: usize);
    fn can_resume(&self) -> bool { false }
    fn block_done(&mut self) {}
    /// The host-only part of the next enqueue() (pulling the upstreams into a staging block), called while the device still works on the
    /// block about to be served.  Optional.
    fn prefetch(&mut self) {}
    fn submit(&mut self, i: usize) {
        let t0 = Instant::now();
        self.enqueue(i);
        let p = self.pump();
        ck(unsafe { rh_event_record(p.slot[i].done.0, p.stream) }, "rh_event_record");
        p.timing.submit_s += t0.elapsed().as_secs_f64();
        p.timing.blocks += 1;
    }
    fn advance(&mut self) -> bool {
        if self.pump().ended {
            if !self.can_resume() { return false; }
            let p = self.pump(); p.ended = false; p.primed = false;
        }
        let starting = !self.pump().primed;
        let ta = Instant::now();
        if starting {
            self.submit(0);
            let p = self.pump(); p.primed = true; p.cur = 0;
        } else {
            let p = self.pump();
            if p.slot[p.cur].last { p.ended = true; return false; }
            p.cur ^= 1;                                                            // the block that was enqueued while the previous one was being served
            if !p.slot[p.cur].last {
                let t0 = Instant::now();
                self.prefetch();
                self.pump().timing.prefetch_s += t0.elapsed().as_secs_f64();
            }
        }
        let cur = self.pump().cur;
        if !self.pump().device_out {   // (a device-resident consumer orders its copies behind the block's event on ITS stream: the host does not wait)
            let t0 = Instant::now();
            ck(unsafe { rh_event_synchronize(self.pump().slot[cur].done.0) }, "rh_event_synchronize");
            self.pump().timing.wait_s += t0.elapsed().as_secs_f64();
            self.block_done();
        }
        let p = self.pump();
        p.pos = p.skip.min(p.slot[cur].n);
        p.skip = 0;
        if !p.slot[cur].last { self.submit(cur ^ 1); }                            // prefetch: pull and process one block ahead
        if starting { self.pump().timing.first_advance_s = ta.elapsed().as_secs_f64(); }
        true
    }
    /// Everything a first `next()` would do before it can serve a sample -- start the stream (plans, page-locked blocks, device rows),
    /// pull and process the first block, put the second one in flight -- done NOW, on the calling thread.  rodio builds its sources on
    /// a control thread and hands them to the audio callback (`src/stream.rs:538-545`), where `next()` is expected not to block.
    fn prepare_stream(&mut self) {
        let p = self.pump();
        if !p.primed && !p.ended { self.advance(); }
    }
    /// Up to `n` samples of the stream into DEVICE memory `ddst`, enqueued on `consumer` (a stream of the caller's); the count, less
    /// than `n` at the end of the stream.  Needs `device_out` (`keep_blocks_on_device`).
    fn read_device_impl(&mut self, ddst: *mut f32, n: usize, consumer: RhStream) -> usize {
        assert!(self.pump().device_out, "read_device() needs keep_blocks_on_device()");
        let mut k = 0;
        while k < n {
            { let p = self.pump(); if p.pos == p.slot[p.cur].n && !self.advance() { break; } }
            let p = self.pump();
            let cur = p.cur;
            let take = (n - k).min(p.slot[cur].n - p.pos);
            unsafe {
                ck(rh_stream_wait_event(consumer, p.slot[cur].done.0), "rh_stream_wait_event");   // the block is complete when its event fires: the consumer's STREAM waits for it
                ck(rh_memcpy_d2d(ddst.add(k).cast(), p.slot[cur].dev.p.add(p.pos).cast(), take * 4, consumer), "rh_memcpy_d2d");
                ck(rh_event_record(p.slot[cur].taken.0, consumer), "rh_event_record");            // ... and the producer rewrites the slot's block only behind this copy
            }
            p.slot[cur].taken_pending = true;
            p.pos += take;
            k += take;
        }
        let p = self.pump();
        p.handed_out += k as u64;
        p.timing.device_samples += k as u64;
        k
    }
    fn next_sample(&mut self) -> Option<f32> {
        assert!(!self.pump().device_out, "this source keeps its blocks on the device: read_device()");
        loop {
            let p = self.pump();
            if p.pos < p.slot[p.cur].n {
                let v = p.slot[p.cur].out.slice(p.slot[p.cur].n)[p.pos];
                p.pos += 1; p.handed_out += 1;
                return Some(v);
            }
            if !self.advance() { return None; }
        }
    }
}

// ------------------------------------------------------------------------------------------------ size_hint ----
/// `Iterator::size_hint()`: (lower, upper).
pub type SizeHint = (usize, Option<usize>);

/// What a source answered to `size_hint()` at the positions it was asked (the adapters read a block ahead of the consumer; rodio's adapters
/// ask their input where THEY stand).  `at(q)`: the last answer recorded at or before sample q, less the samples taken since -- exact for a
/// source that counts its samples down (buffer.rs:134-137) and for one that answers the trait's default (0, None); a valid bound otherwise.
#[derive(Default)]
pub struct HintLog { log: std::collections::VecDeque<(u64, SizeHint)> }
impl HintLog {
    pub fn note(&mut self, pos: u64, h: SizeHint) {
        match self.log.back_mut() { Some(b) if b.0 == pos => b.1 = h, _ => self.log.push_back((pos, h)) }
        if self.log.len() > 256 { self.log.pop_front(); }
    }
    pub fn is_empty(&self) -> bool { self.log.is_empty() }
    pub fn clear(&mut self) { self.log.clear(); }
    pub fn at(&mut self, q: u64) -> SizeHint {
        while self.log.len() > 1 && self.log[1].0 <= q { self.log.pop_front(); }     // (the questions only move forward)
        let (p, h) = self.log[0];
        let since = q.saturating_sub(p) as usize;
        (h.0.saturating_sub(since), h.1.map(|u| u.saturating_sub(since)))
    }
}

/// `UniformSourceIterator::size_hint()` (uniform.rs:100-108): the lower bound of the converter chain that is open -- ChannelCountConverter
/// (channels.rs:88-102) over SampleRateConverter (sample_rate.rs:204-238) over Take (uniform.rs:181-196) -- and no upper bound.  Those bounds
/// are functions of the converters' COUNTERS; the samples are converted on the device, so this runs the three iterators' state machines
/// over counts only, fed with the spans the samples were pulled in, lazily.  The twin of `detail::UniformCounter` (include/rodio_hip.hpp).
pub struct UniformCounter {
    to_ch: u16, to_rate: u32, bare: bool,
    spans: std::collections::VecDeque<CountSpan>,                                 // front: the span of the open chain
    open: bool, started: bool, done: u64, used: u64, left: usize,
    from: u32, to: u32, pos_in_chunk: u32, out_pos: u32, cur_len: usize, next_len: usize, buf_len: usize, ccc_pos: u16, have_repeat: bool,
}
#[derive(Clone, Copy)]
struct CountSpan { ch: u16, rate: u32, up_pos: u64, limited: bool, closed: bool, take_n: usize, got: u64, reps: u64 }
impl UniformCounter {
    /// `bare`: a SampleRateConverter on its own (no Take in front, no ChannelCountConverter behind): both of ITS bounds.
    pub fn new(to_ch: u16, to_rate: u32, bare: bool) -> Self {
        UniformCounter { to_ch, to_rate, bare, spans: Default::default(), open: false, started: false, done: 0, used: 0, left: 0, from: 1, to: 1, pos_in_chunk: 0, out_pos: 0,
                         cur_len: 0, next_len: 0, buf_len: 0, ccc_pos: 0, have_repeat: false }
    }
    /// A piece of the input, in pull order; `up_pos`: samples taken from the iterator's input in front of it.
    pub fn feed(&mut self, p: &Piece, up_pos: u64) {
        if p.opens || self.spans.back().map_or(true, |b| b.closed) {
            self.compact();
            self.spans.push_back(CountSpan { ch: p.ch, rate: p.rate, up_pos, limited: p.limit.is_some(), closed: false, take_n: p.limit.unwrap_or(0), got: 0, reps: 1 });
        }
        let b = self.spans.back_mut().unwrap();
        b.got += p.n as u64;
        if p.closes { b.closed = true; }
    }
    pub fn input_ended(&mut self) { if let Some(b) = self.spans.back_mut() { b.closed = true; } }
    fn compact(&mut self) {                                                       // the spans nobody has asked about yet, run-length encoded
        let n = self.spans.len();
        if n < 2 { return; }
        let (a, b) = (self.spans[n - 2], self.spans[n - 1]);
        if a.closed && b.closed && a.ch == b.ch && a.rate == b.rate && a.limited == b.limited && a.take_n == b.take_n && a.got == b.got && b.reps == 1 {
            self.spans[n - 2].reps += 1;
            self.spans.pop_back();
        }
    }
    fn cur(&self) -> CountSpan { self.spans[0] }
    fn taken_total(&self) -> u64 { if self.open { self.cur().up_pos + self.used } else { self.spans.front().map_or(0, |s| s.up_pos) } }
    fn take_next(&mut self) -> bool {                                             // Take::next over the counts (uniform.rs:160-177)
        let c = self.cur();
        if c.limited { if self.left == 0 { return false; } self.left -= 1; }
        if self.used < c.got { self.used += 1; return true; }
        if c.closed { return false; }                                             // the input returned None here
        self.used += 1;                                                           // (asked beyond what has been pulled: the sample is taken to exist)
        true
    }
    fn take_frame(&mut self) -> usize { let ch = self.cur().ch as usize; let mut k = 0; while k < ch && self.take_next() { k += 1; } k }   // sample_rate.rs:58-71,113-121
    fn next_input_span(&mut self) { self.pos_in_chunk += 1; self.cur_len = self.next_len; self.next_len = self.take_frame(); }               // sample_rate.rs:110-122
    fn src_next(&mut self) -> bool {                                              // sample_rate.rs:131-201
        if self.from == self.to { return self.take_next(); }
        if self.buf_len > 0 { self.buf_len -= 1; return true; }
        if self.out_pos == self.to {
            self.out_pos = 0;
            self.next_input_span();
            while self.pos_in_chunk != self.from { self.next_input_span(); }
            self.pos_in_chunk = 0;
        } else {
            let req = (self.from.wrapping_mul(self.out_pos) / self.to) % self.from;
            while self.pos_in_chunk != req { self.next_input_span(); }
        }
        let n = self.cur_len.min(self.next_len);                                  // zip
        self.out_pos += 1;
        if n > 0 { self.buf_len = n - 1; return true; }
        if self.cur_len == 0 { return false; }                                    // :193-200 draining `current_span`
        self.buf_len = self.cur_len - 1;
        self.cur_len = 0;
        true
    }
    fn ccc_next(&mut self) -> bool {                                              // channels.rs:57-85
        let (from, to) = (self.cur().ch, self.to_ch);
        let some;
        if self.ccc_pos == 0 { some = self.src_next(); self.have_repeat = some; }
        else if self.ccc_pos < from { some = self.src_next(); }
        else if self.ccc_pos == 1 { some = self.have_repeat; }
        else { some = true; }
        if some { self.ccc_pos += 1; }
        if self.ccc_pos == to {
            self.ccc_pos = 0;
            for _ in to..from { let _ = self.src_next(); }
        }
        some
    }
    fn bootstrap(&mut self) -> bool {                                             // uniform.rs:50-68,82-92: the next span's chain
        if self.open {
            if self.spans[0].reps > 1 { self.spans[0].reps -= 1; self.spans[0].up_pos += self.spans[0].got; } else { self.spans.pop_front(); }
        }
        self.started = true;
        if self.spans.is_empty() { self.open = false; return false; }             // nothing was pulled for another chain: the one rodio builds here is empty
        self.open = true;
        self.used = 0;
        let c = self.cur();
        self.left = c.take_n;
        let (mut a, mut b) = (c.rate as u64, self.to_rate as u64);
        while b != 0 { let t = a % b; a = b; b = t; }
        self.from = (c.rate as u64 / a) as u32;
        self.to = (self.to_rate as u64 / a) as u32;
        self.pos_in_chunk = 0; self.out_pos = 0; self.buf_len = 0; self.ccc_pos = 0; self.have_repeat = false; self.cur_len = 0; self.next_len = 0;
        if self.from != self.to { self.cur_len = self.take_frame(); self.next_len = self.take_frame(); }
        true
    }
    fn step(&mut self) -> bool {                                                  // UniformSourceIterator::next (uniform.rs:76-97)
        if self.bare {
            if !self.started && !self.bootstrap() { return false; }
            if !self.open || !self.src_next() { return false; }
            self.done += 1;
            return true;
        }
        if self.open && self.ccc_next() { self.done += 1; return true; }
        if !self.bootstrap() || !self.ccc_next() { return false; }
        self.done += 1;
        true
    }
    /// `size_hint()` once the iterator has returned `e` samples; `None`: it returned None before that.  `in_hint(q)`: the input's
    /// `size_hint()` at sample q of the input.
    pub fn hint_at(&mut self, e: u64, in_hint: &mut dyn FnMut(u64) -> SizeHint) -> Option<SizeHint> {
        while self.done < e { if !self.step() { return None; } }
        if self.bare && !self.started && !self.spans.is_empty() { let _ = self.bootstrap(); }   // SampleRateConverter::new reads its first two frames (sample_rate.rs:58-71)
        if !self.started { return Some((in_hint(self.taken_total()).0, None)); }    // uniform.rs:105: no chain has been built yet -- the pending input itself
        if !self.open { return Some((0, None)); }
        let c = self.cur();
        let mut h = in_hint(self.taken_total());                                    // Take (uniform.rs:181-196)
        if c.limited { h = (h.0.min(self.left), Some(h.1.map_or(self.left, |u| u.min(self.left)))); }
        let (from, to, pos, out_pos, next_len, buf_len, ch) = (self.from, self.to, self.pos_in_chunk, self.out_pos, self.next_len, self.buf_len, c.ch as usize);
        let apply = |samples: usize| -> usize {                                     // sample_rate.rs:204-238, usize / u32 arithmetic as written there
            let mut after = samples;
            if pos == from - 1 { after += next_len; }
            let unread = from.saturating_sub(pos + 2) as usize * ch;
            after = after.saturating_sub(unread);
            after = after * to as usize / from as usize;
            (to - out_pos) as usize * ch + after + buf_len
        };
        if from != to { h = (apply(h.0), h.1.map(apply)); }
        if self.bare { return Some(h); }
        let p = self.ccc_pos as usize;                                              // channels.rs:88-102; the iterator keeps the lower bound (uniform.rs:100-108)
        let consumed = ch.min(p);
        Some((((h.0 + consumed) / ch * self.to_ch as usize).saturating_sub(p), None))
    }
}

/// What every adapter of a chain makes of its input's `size_hint()`.  `fun(in, emitted)`: `in(q)` = the input's answer once q of ITS samples
/// have been taken, `emitted` = samples the adapter has emitted (both from the start of the stream, or from the last seek).  Shared (Arc):
/// a consumer that keeps asking after the chain has been retired (GpuMixer) holds on to it.  The twin of `GpuSource::HintState`.
pub struct HintStage {
    pub fun: Option<Box<dyn FnMut(&mut dyn FnMut(u64) -> SizeHint, u64) -> SizeHint + Send>>,   // None: in(in_pos(emitted))
    pub in_pos: Option<Box<dyn Fn(u64) -> u64 + Send>>,                                          // None: one sample in per sample out
}
#[derive(Default)]
pub struct HintState { pub stages: Vec<HintStage>, pub log: HintLog, pub before: SizeHint, pub out_base: u64, pub in_base: u64, pub skip: u64 }
impl HintState {
    pub fn at(&mut self, emitted: u64) -> SizeHint {
        let rel = emitted.saturating_sub(self.out_base) + self.skip;
        let HintState { stages, log, before, in_base, .. } = self;
        fn through(stages: &mut [HintStage], x: u64, base: &mut dyn FnMut(u64) -> SizeHint) -> SizeHint {
            match stages.split_last_mut() {
                None => base(x),
                Some((st, front)) => {
                    let mut input = |q: u64| through(front, q, base);
                    match (&mut st.fun, &st.in_pos) {
                        (Some(f), _) => f(&mut input, x),
                        (None, Some(p)) => input(p(x)),
                        (None, None) => input(x),
                    }
                }
            }
        }
        let (b, ib) = (*before, *in_base);
        through(stages, rel, &mut |q| if log.is_empty() { b } else { log.at(ib + q) })
    }
}

// ------------------------------------------------------------------------------------------------ GpuSource ----
struct Ctx<'a> { out: *mut f32, inp: *const f32, n: usize, out_cap: usize, flush: bool, stream: RhStream, pieces: &'a [Piece], end: bool }
struct Stage {
    run: Box<dyn FnMut(&mut Ctx) -> usize + Send>,
    bound: Box<dyn Fn(usize, usize) -> usize + Send>,   // (samples in, pieces in the block) -> most samples out
    seekable: bool,
    on_seek: Option<Box<dyn FnMut(Duration) + Send>>,
    span_rule: u8,                                        // 0 the input's spans, 1 None (Mix, the converters), 2 the input's with another sample count
}
struct Handle<T> { p: *mut T, destroy: unsafe extern "C" fn(*mut T) -> RhStatus }
impl<T> Drop for Handle<T> { fn drop(&mut self) { if !self.p.is_null() { unsafe { (self.destroy)(self.p); } } } }
unsafe impl<T> Send for Handle<T> {}
struct State(DeviceBuf);
unsafe impl Send for State {}
unsafe impl Sync for State {}   // (shared through Arc by a stage and its on_seek: both run on the thread that owns the chain)
unsafe impl<T> Sync for Handle<T> {}

/// `upstream.amplify(..).low_pass(..)...` with the chain executed block-wise on the GPU.
pub struct GpuSource<I: Source> {
    up: I, block_frames: usize, ch: u16, rate: u32, in_ch: u16, in_rate: u32,
    stages: Vec<Stage>, reader: SpanReader, pieces: Vec<Piece>, span_aware: bool, scan_kernels: bool,
    may_cut: bool,     // an adapter of the chain can make the stream end inside a frame (may_end_inside_a_frame)
    filter_mode: u8,   // 0: by the filter contract, per filter (rh_filter_scan_ok); 1: reference order throughout; 2: time-parallel throughout
    a: DeviceBuf, b: DeviceBuf, pump: Pump,
    hint: Arc<Mutex<HintState>>,   // size_hint(): see HintState
    durs: Vec<Option<Box<dyn Fn(Option<Duration>) -> Option<Duration> + Send>>>,   // total_duration() behind every adapter from its input's (None: the input's, amplify.rs:95-97 and the like)
    pulled_total: u64,             // samples pulled from the upstream, whatever was sought in between
    last_kind: u8,                 // the adapter pushed last: 1 a `uniform`, 2 a filter right behind a `uniform`, 0 anything else
}
type Arc<T> = std::sync::Arc<T>;
type Mutex<T> = std::sync::Mutex<T>;
unsafe impl<I: Source + Send> Send for GpuSource<I> {}

impl<I: Source> GpuSource<I> {
    pub fn new(upstream: I, block_frames: usize) -> Self {
        let (ch, rate) = (upstream.channels().get(), upstream.sample_rate().get());
        GpuSource { up: upstream, block_frames: block_frames.max(1), ch, rate, in_ch: ch, in_rate: rate, stages: Vec::new(), reader: SpanReader::new(),
                    pieces: Vec::new(), span_aware: false, scan_kernels: false, may_cut: false, filter_mode: 0, a: DeviceBuf::new(), b: DeviceBuf::new(), pump: Pump::new(),
                    hint: Arc::new(Mutex::new(HintState::default())), durs: Vec::new(), pulled_total: 0, last_kind: 0 }
    }
    /// `total_duration()` as rodio's adapters answer it, adapter by adapter: the input's behind the ones that keep it (amplify.rs:95-97,
    /// blt.rs:171-173, limit.rs:592-594, agc.rs:588-590, channel_volume.rs:119-121), plus the delay behind `delay` (delay.rs:111-115), the
    /// shorter of the two behind `take_duration` (take.rs:209-219), the longer behind `reverb` (mix.rs:104-112), and -- behind `reverb` and
    /// `uniform` -- the value the input gave when the adapter was BUILT (buffered.rs:16, uniform.rs:37).
    fn duration_through(&self) -> Option<Duration> {
        let mut d = self.up.total_duration();
        for f in self.durs.iter().flatten() { d = f(d); }
        d
    }
    /// `size_hint()` once `emitted` samples of the chain's stream have been served (a consumer that takes the blocks ahead of ITS consumer:
    /// GpuMixer); `size_hint()` itself asks for the consumer's own position.
    pub fn size_hint_at(&self, emitted: u64) -> SizeHint {
        let mut h = self.hint.lock().unwrap();
        if h.log.is_empty() { h.before = self.up.size_hint(); }                     // nothing pulled (since the last seek): the upstream stands where the question reaches it
        h.at(emitted)
    }
    /// The last adapter of the chain is a `uniform` (what Mixer::add would wrap the chain in is already there).
    pub fn ends_with_uniform(&self) -> bool { self.last_kind == 1 }
    fn set_hints(&mut self, f: impl FnMut(&mut dyn FnMut(u64) -> SizeHint, u64) -> SizeHint + Send + 'static) { self.hint.lock().unwrap().stages.last_mut().unwrap().fun = Some(Box::new(f)); }
    fn set_in_pos(&mut self, f: impl Fn(u64) -> u64 + Send + 'static) { self.hint.lock().unwrap().stages.last_mut().unwrap().in_pos = Some(Box::new(f)); }
    fn set_duration(&mut self, f: impl Fn(Option<Duration>) -> Option<Duration> + Send + 'static) { *self.durs.last_mut().unwrap() = Some(Box::new(f)); }
    pub fn inner(&self) -> &I { &self.up }
    pub fn inner_mut(&mut self) -> &mut I { &mut self.up }
    pub fn into_inner(self) -> I where I: Clone { self.up.clone() }
    /// The chain's stream can end inside a frame although its upstream keeps `Source`'s contract: reverb and delay count their silence in
    /// SAMPLES (delay.rs:14), `uniform` over spans that cut frames hands on what rodio's converters make of the cut.  (The C++ twin's GpuMixer
    /// completes such a chain to `amplify -> UniformSourceIterator -> filter` before it enters a fused stream: not ported, INTEGRATION.md.)
    pub fn may_end_inside_a_frame(&self) -> bool { self.may_cut }
    /// The stream ended where rodio's ChannelVolume would return one more frame of its stale sum to a consumer that asks again (C++ twin only so far: always false here).
    pub fn ended_with_a_stale_frame(&self) -> bool { false }
    /// The chain's `current_span_len()` comes from an adapter's own arithmetic (take_duration, delay, channel_volume): see the C++ twin's `span_behind`.
    pub fn answers_with_adapter_spans(&self) -> bool { self.stages.iter().rev().map(|s| s.span_rule).find(|&r| r != 0).unwrap_or(0) == 2 }
    fn push(&mut self, run: impl FnMut(&mut Ctx) -> usize + Send + 'static, bound: impl Fn(usize, usize) -> usize + Send + 'static, span_rule: u8) -> &mut Stage {
        self.stages.push(Stage { run: Box::new(run), bound: Box::new(bound), seekable: true, on_seek: None, span_rule });
        self.hint.lock().unwrap().stages.push(HintStage { fun: None, in_pos: None });
        self.durs.push(None);
        self.last_kind = 0;
        self.stages.last_mut().unwrap()
    }
    fn state(&self, floats: usize) -> std::sync::Arc<State> {
        let mut d = DeviceBuf::new();
        d.reserve(floats);
        ck(unsafe { rh_memset(d.p.cast(), 0, floats * 4, self.pump.stream) }, "rh_memset");
        std::sync::Arc::new(State(d))
    }

    // -- builder methods (source/mod.rs:255-731); call before the first next()
    pub fn amplify(mut self, factor: f32) -> Self {                              // amplify.rs:64
        self.push(move |c| { ck(unsafe { rh_amplify(c.out, c.inp, c.n, factor, c.stream) }, "rh_amplify"); c.n }, |n, _| n, 0);
        self
    }
    pub fn amplify_decibel(self, db: f32) -> Self { let f = unsafe { rh_db_to_linear(db) }; self.amplify(f) }   // amplify.rs:33-35
    pub fn distortion(mut self, gain: f32, threshold: f32) -> Self {             // distortion.rs:66-72
        self.push(move |c| { ck(unsafe { rh_distortion(c.out, c.inp, c.n, gain, threshold, c.stream) }, "rh_distortion"); c.n }, |n, _| n, 0);
        self
    }
    /// How the filters of this chain run.  By default every filter decides for itself by THE FILTER CONTRACT (`rodio_hip.h`,
    /// `rh_filter_scan_ok`): time-parallel (`rh_biquad` mode 1) where that stays within 1e-5 of rodio's own f32 recurrence for a
    /// full-scale source, the reference's operation order (mode 0, bit for bit) where it would not -- low cutoffs, where rodio's
    /// recurrence amplifies its own rounding noise past 1e-5.  `exact_filters(true)`: the reference's order throughout;
    /// `exact_filters(false)`: time-parallel throughout (closer to the exact response than rodio is, further than 1e-5 from rodio at
    /// low cutoffs).
    pub fn exact_filters(mut self, on: bool) -> Self { self.filter_mode = if on { 1 } else { 2 }; self }
    /// See [`GpuMixer::prepare`]: the stream is started on the calling thread, the consumer's first `next()` finds its block waiting.
    pub fn prepare(&mut self) { self.prepare_stream(); }
    /// Device-resident hand-off (`GpuMixer::add_chain`): the blocks stay in device memory, a consumer takes them with `read_device`.
    /// Before the first block; `next()` is then not available.
    pub fn keep_blocks_on_device(&mut self, on: bool) {
        assert!(!self.pump.primed, "keep_blocks_on_device() after the stream has started");
        self.pump.device_out = on;
    }
    pub fn blocks_on_device(&self) -> bool { self.pump.device_out }
    pub fn started(&self) -> bool { self.pump.primed }
    pub fn timing(&self) -> Timing { self.pump.timing }
    /// Up to `n` samples of the chain's output into DEVICE memory, enqueued on `consumer`: see `detail::BlockPump::read_device`.
    pub fn read_device(&mut self, ddst: *mut f32, n: usize, consumer: RhStream) -> usize { self.read_device_impl(ddst, n, consumer) }
    pub fn low_pass(self, freq: u32) -> Self { self.blt(0, freq, 0.5) }          // blt.rs:11-16
    pub fn high_pass(self, freq: u32) -> Self { self.blt(1, freq, 0.5) }         // blt.rs:18-24
    pub fn low_pass_with_q(self, freq: u32, q: f32) -> Self { self.blt(0, freq, q) }
    pub fn high_pass_with_q(self, freq: u32, q: f32) -> Self { self.blt(1, freq, q) }
    fn blt(mut self, kind: i32, freq: u32, q: f32) -> Self {                      // blt.rs:502-544,558-560
        let ch = self.ch as u32;
        let mut co = [0f32; 5];
        ck(unsafe { rh_biquad_coeffs(kind, freq, q, self.rate, co.as_mut_ptr()) }, "rh_biquad_coeffs");
        let st = self.state(4 * ch as usize);
        let exact = self.filter_mode == 1 || (self.filter_mode == 0 && unsafe { rh_filter_scan_ok(kind, freq, q, self.rate) } == 0);   // the filter contract (rodio_hip.h)
        let (st2, sm, mode) = (st.clone(), self.pump.stream as usize, if exact { 0 } else { 1 });
        let behind_uniform = self.last_kind == 1;
        let stage = self.push(move |c| {
            let frames = c.n / ch as usize;
            ck(unsafe { rh_biquad(c.out, c.inp, frames as u64, ch, 1, co.as_ptr(), st.0.p, mode, c.stream) }, "rh_biquad");
            frames * ch as usize
        }, |n, _| n, 0);
        stage.on_seek = Some(Box::new(move |_| ck(unsafe { rh_memset(st2.0.p.cast(), 0, 4 * ch as usize * 4, sm as RhStream) }, "rh_memset")));   // blt.rs:350-377
        self.last_kind = if behind_uniform { 2 } else { 0 };
        self
    }
    pub fn reverb(mut self, duration: Duration, amplitude: f32) -> Self {         // source/mod.rs:628-634
        let d = unsafe { rh_delay_samples(duration.as_nanos() as u64, self.rate, self.ch as u32) };
        self.may_cut = self.may_cut || d % self.ch as u64 != 0;
        let mut e: *mut RhEcho = ptr::null_mut();
        ck(unsafe { rh_echo_create(&mut e, d, amplitude) }, "rh_echo_create");
        let h = Handle { p: e, destroy: rh_echo_destroy };
        let captured = self.duration_through();                                       // buffered.rs:16: asked when the source is buffered
        let s0 = self.current_span_len().unwrap_or(32768) as u64;
        let (chn, rate_now) = (self.ch as u64, self.rate as u64);
        let seen = Arc::new(Mutex::new((0u64, false)));                               // samples of the input so far / that is all of it
        let seen2 = seen.clone();
        let stage = self.push(move |c| {
            let h = &h;   // (the whole handle moves into the closure -- it is Send and dropped with the stage -- not just its raw pointer field)
            { let mut sn = seen.lock().unwrap(); sn.0 += c.n as u64; sn.1 = sn.1 || c.flush; }
            if c.n > 0 { ck(unsafe { rh_echo_process(h.p, c.out, c.inp, c.n as u64, c.stream) }, "rh_echo_process"); }
            if !c.flush { return c.n; }
            if d > 0 { ck(unsafe { rh_echo_flush(h.p, c.out.add(c.n), c.stream) }, "rh_echo_flush"); }                 // the delayed clone outlives the source
            c.n + d as usize
        }, move |n, _| n + d as usize, 1);                                           // Mix::current_span_len() is None (mix.rs:92-94)
        stage.seekable = false;                                                        // mix.rs:116-120
        self.set_duration(move |_| captured.map(|f1| f1 + duration));                 // mix.rs:104-112: max(f1, f1 + d)
        // mix.rs:56-67 over two UniformSourceIterators (mix.rs:10-22): (the larger lower bound, None) -- see GpuSource::reverb in include/rodio_hip.hpp
        // for the derivation (the echo's silence through Take's chains of min(span + silence, 32768) samples in whole frames; the plain branch at
        // its Buffered source's end: an empty 44100 -> rate converter on ONE channel, buffered.rs:218-233, sample_rate.rs:190,226-229).
        let (mut g44, mut gb) = (44100u64, rate_now);
        while gb != 0 { let t = g44 % gb; g44 = gb; gb = t; }
        let end_chunk = if rate_now == 44100 { 0 } else { (rate_now / g44 - 1) * chn };
        self.set_hints(move |_, e| {
            if e == 0 { return (d as usize, None); }
            let sn = *seen2.lock().unwrap();
            let plain = if sn.1 && e > sn.0 { end_chunk } else { 0 };
            if e >= d { return (plain as usize, None); }
            let (mut start, mut n) = (0u64, (s0 + d).min(32768));
            while n > 0 && start + n <= e { start += n; n = (s0 + (d - start)).min(32768); }
            let in_chain = e - start;
            let pos = in_chain % chn;
            let lo = (d - e).min(n.saturating_sub(in_chain));
            let x = (lo + pos) / chn * chn;
            (plain.max(x.saturating_sub(pos)) as usize, None)
        });
        self
    }
    pub fn channel_volume(mut self, gains: Vec<f32>) -> Self {                    // channel_volume.rs:71-88
        let (in_ch, out_ch) = (self.ch as usize, gains.len());
        assert!(out_ch > 0, "channel_volume: no output channels");
        self.push(move |c| {
            let frames = c.n / in_ch;
            ck(unsafe { rh_channel_volume(c.out, c.inp, frames, in_ch as u32, gains.as_ptr(), out_ch as u32, c.stream) }, "rh_channel_volume");
            frames * out_ch
        }, move |n, _| n / in_ch * out_ch, 2);
        self.set_in_pos(move |e| (e + out_ch as u64 - 1) / out_ch as u64 * in_ch as u64);   // channel_volume.rs:91-93: the INPUT's answer; a frame of it has been taken whenever an output frame begins (:71-79)
        self.ch = out_ch as u16;
        self
    }
    /// `Spatial` (spatial.rs:19-24,48-69): the two ear gains from the positions, then `ChannelVolume`.
    pub fn spatial(self, emitter: [f32; 3], left_ear: [f32; 3], right_ear: [f32; 3]) -> Self {
        let mut g = [0f32; 2];
        ck(unsafe { rh_spatial_gains(emitter.as_ptr(), left_ear.as_ptr(), right_ear.as_ptr(), g.as_mut_ptr()) }, "rh_spatial_gains");
        self.channel_volume(vec![g[0], g[1]])
    }
    /// `dither(target_bits, algorithm)` (dither.rs:217-242); algorithm in the reference's enum order: 0 GPDF, 1 HighPass, 2 RPDF, 3 TPDF (the
    /// default).  The noise of sample k is a function of (seed, k) -- see `rh_dither` in rodio_hip.h.
    pub fn dither(mut self, target_bits: u32, algorithm: i32, seed: u64) -> Self {
        let ch = self.ch as u32;
        let mut pos = 0u64;
        self.push(move |c| {
            ck(unsafe { rh_dither(c.out, c.inp, c.n, pos, ch, target_bits, algorithm, seed, c.stream) }, "rh_dither");
            pos += c.n as u64;
            c.n
        }, |n, _| n, 0);
        self
    }
    pub fn convert_channels(mut self, to: ChannelCount) -> Self {                 // ChannelCountConverter, channels.rs:57-85
        let (from, to) = (self.ch as usize, to.get() as usize);
        self.push(move |c| {
            let frames = c.n / from;
            ck(unsafe { rh_channels_convert(c.out, c.inp, frames, from as u32, to as u32, c.stream) }, "rh_channels_convert");
            frames * to
        }, move |n, _| n / from * to, 1);
        self.set_hints(move |inp, e| {                                              // channels.rs:88-102
            let (pos, consumed) = (e % to as u64, (from as u64).min(e % to as u64));
            let h = inp(e / to as u64 * from as u64 + consumed);
            let f = |v: usize| ((v as u64 + consumed) / from as u64 * to as u64).saturating_sub(pos) as usize;
            (f(h.0), h.1.map(f))
        });
        self.ch = to as u16;
        self
    }
    pub fn convert_sample_rate(mut self, to: SampleRate) -> Self {                // SampleRateConverter, sample_rate.rs:52-201
        let (from, to, ch) = (self.rate, to.get(), self.ch as usize);
        if from == to { return self; }                                              // sample_rate.rs:133-136
        let mut r: *mut RhResampler = ptr::null_mut();
        ck(unsafe { rh_resampler_create(&mut r, from, to, ch as u32) }, "rh_resampler_create");
        let h = Handle { p: r, destroy: rh_resampler_destroy };
        let cnt = Arc::new(Mutex::new((UniformCounter::new(ch as u16, to, true), 0u64)));   // sample_rate.rs:204-238 over counts; samples fed so far
        cnt.lock().unwrap().0.feed(&Piece { n: 0, opens: true, closes: false, ch: ch as u16, rate: from, tail: 0, by_none: false, limit: None }, 0);
        let (cnt2, cnt3) = (cnt.clone(), cnt.clone());
        let stage = self.push(move |c| {
            let h = &h;   // (whole-struct capture: see reverb)
            { let mut k = cnt.lock().unwrap(); let at = k.1; k.0.feed(&Piece { n: c.n, opens: false, closes: c.flush, ch: ch as u16, rate: from, tail: 0, by_none: c.flush, limit: None }, at); k.1 += c.n as u64; }
            let mut m = 0u64;
            ck(unsafe { rh_resampler_process(h.p, c.out, (c.out_cap / ch) as u64, c.inp, (c.n / ch) as u64, c.flush as i32, &mut m, c.stream) }, "rh_resampler_process");
            m as usize * ch
        }, move |n, _| (((n / ch + 2) as u64 * to as u64 / from as u64) as usize + 2) * ch, 1);
        stage.on_seek = Some(Box::new(move |_| {                                    // (the counts start over with the stream behind the new position)
            let mut k = cnt3.lock().unwrap();
            *k = (UniformCounter::new(ch as u16, to, true), 0);
            k.0.feed(&Piece { n: 0, opens: true, closes: false, ch: ch as u16, rate: from, tail: 0, by_none: false, limit: None }, 0);
        }));
        self.set_hints(move |inp, e| cnt2.lock().unwrap().0.hint_at(e, inp).unwrap_or((0, Some(0))));
        self.rate = to;
        self
    }
    /// `UniformSourceIterator::new(src, channels, rate)` (uniform.rs:50-97), span by span: see the crate documentation.
    pub fn uniform(mut self, channels: ChannelCount, sample_rate: SampleRate) -> Self {
        let rule = self.stages.iter().rev().map(|s| s.span_rule).find(|&r| r != 0).unwrap_or(0);
        assert!(!(rule == 2 && self.up.current_span_len().is_some()), "GpuSource::uniform behind take_duration / delay / channel_volume on a source that reports spans");
        self.may_cut = self.may_cut || (rule == 0 && self.ch > 1 && self.up.current_span_len().is_some());
        if rule != 0 {                                                              // continuous from here on
            let from_ch = self.ch;
            let mut s = self.convert_sample_rate(sample_rate);
            if channels.get() != from_ch { s = s.convert_channels(channels); }
            if let Some(st) = s.stages.last_mut() { st.span_rule = 1; }
            // (uniform.rs:37,100-108: the duration the input gave when the iterator was built; the lower bound only -- the arithmetic of the two
            // bare converters above is that of the iterator's one continuous chain)
            let captured = s.duration_through();
            s.set_duration(move |_| captured);
            { let mut hs = s.hint.lock().unwrap(); if let Some(f) = hs.stages.last_mut().and_then(|st| st.fun.take()) { let mut f = f; hs.stages.last_mut().unwrap().fun = Some(Box::new(move |i, e| (f(i, e).0, None))); } }
            s.last_kind = 1;
            return s;
        }
        self.span_aware = true;
        let (to_ch, to_rate, in_ch, from) = (channels.get(), sample_rate.get(), self.ch as usize, self.rate);
        let plan = std::sync::Arc::new(std::sync::Mutex::new(UniformPlanner::new(to_ch, to_rate)));
        let plan2 = plan.clone();
        let (mut win, mut keep) = (State(DeviceBuf::new()), State(DeviceBuf::new()));
        // What UniformSourceIterator emits is a stream of SAMPLES (a span that ends inside a frame leaves a run that need not fill a frame); the
        // adapters behind work on frames, so a block hands on whole frames and the samples of a frame that is not complete yet wait here.
        let mut part = State(DeviceBuf::new());
        part.0.reserve(64.max(to_ch as usize));
        let part_n = std::sync::Arc::new(std::sync::Mutex::new(0usize));
        let part_n2 = part_n.clone();
        let captured = self.duration_through();                                       // uniform.rs:37
        let cnt = Arc::new(Mutex::new((UniformCounter::new(to_ch, to_rate, false), 0u64)));   // uniform.rs:100-108 over counts, fed with the planner's pieces
        let (cnt2, cnt3) = (cnt.clone(), cnt.clone());
        let stage = self.push(move |c| {
            let (win, keep, part) = (&mut win, &mut keep, &mut part);   // (whole-struct capture)
            { let mut k = cnt.lock().unwrap(); for p in c.pieces { let at = k.1; k.0.feed(p, at); k.1 += p.n as u64; } if c.flush { k.0.input_ended(); } }
            let mut plan = plan.lock().unwrap();
            plan.begin_block();
            let hs = plan.held_samples();
            win.0.reserve(hs + c.n + 4);
            unsafe {
                if hs > 0 { ck(rh_memcpy_d2d(win.0.p.cast(), keep.0.p.cast(), hs * 4, c.stream), "rh_memcpy_d2d"); }
                if c.n > 0 { ck(rh_memcpy_d2d(win.0.p.add(hs).cast(), c.inp.cast(), c.n * 4, c.stream), "rh_memcpy_d2d"); }
            }
            let mut segs = Vec::new();
            for p in c.pieces { plan.add(p, &mut segs); }
            plan.end_block();
            let mut carried = part_n.lock().unwrap();
            if *carried > 0 { ck(unsafe { rh_memcpy_d2d(c.out.cast(), part.0.p.cast(), *carried * 4, c.stream) }, "rh_memcpy_d2d"); }
            let table: Vec<RhUniformSeg> = segs.iter().map(|s| {
                let mut g = s.g;
                g.src = unsafe { win.0.p.add(s.src_off) };
                g.dst = unsafe { c.out.add(*carried + s.dst_off) };                 // (sample offsets)
                g
            }).collect();
            let total = *carried + plan.out_samples();
            assert!(total + to_ch as usize <= c.out_cap, "GpuSource::uniform: block capacity");
            ck(unsafe { rh_uniform_segments(table.as_ptr(), table.len() as u32, c.stream) }, "rh_uniform_segments");
            let kn = plan.keep_samples();
            if kn > 0 {
                keep.0.reserve(kn);
                ck(unsafe { rh_memcpy_d2d(keep.0.p.cast(), win.0.p.add(plan.keep_offset()).cast(), kn * 4, c.stream) }, "rh_memcpy_d2d");
            }
            let rest = if c.flush { 0 } else { total % to_ch as usize };          // at the end of the stream the rest of a frame is handed on as it is
            if rest > 0 { ck(unsafe { rh_memcpy_d2d(part.0.p.cast(), c.out.add(total - rest).cast(), rest * 4, c.stream) }, "rh_memcpy_d2d"); }
            *carried = rest;
            total - rest
        }, move |n, pieces| {                                                        // every span may add its verbatim last frame, or what the converters make of a cut frame
            let f = (n / in_ch) as u64;
            (f.max(f * to_rate as u64 / from as u64 + 2) as usize + (UniformPlanner::close_slack_frames(from, to_rate) as usize + 1) * (pieces + 2) + 1) * to_ch as usize
        }, 1);
        stage.on_seek = Some(Box::new(move |_| {   // the next span starts a fresh chain
            *plan2.lock().unwrap() = UniformPlanner::new(to_ch, to_rate);
            *part_n2.lock().unwrap() = 0;
            *cnt3.lock().unwrap() = (UniformCounter::new(to_ch, to_rate, false), 0);
        }));
        self.set_duration(move |_| captured);                                         // uniform.rs:131-133
        self.set_hints(move |inp, e| cnt2.lock().unwrap().0.hint_at(e, inp).unwrap_or((0, None)));
        self.last_kind = 1;
        self.ch = to_ch;
        self.rate = to_rate;
        self
    }
    pub fn limit(mut self, settings: RhLimitParams) -> Self {                     // limit.rs:94-130,853-988
        let (ch, rate) = (self.ch as u32, self.rate);
        let st = self.state(2 * ch as usize);
        let (st2, sm) = (st.clone(), self.pump.stream as usize);
        self.scan_kernels = true;
        let stage = self.push(move |c| {
            ck(unsafe { rh_limit(c.out, c.inp, (c.n / ch as usize) as u64, ch, rate, 1, &settings, st.0.p, c.stream) }, "rh_limit");
            c.n / ch as usize * ch as usize
        }, |n, _| n, 0);
        stage.on_seek = Some(Box::new(move |_| ck(unsafe { rh_memset(st2.0.p.cast(), 0, 2 * ch as usize * 4, sm as RhStream) }, "rh_memset")));   // limit.rs:1139-1158
        self
    }
    pub fn automatic_gain_control(mut self, settings: RhAgcParams) -> Self {      // agc.rs:133-171,397-504
        let rate = self.rate;
        let mut d = DeviceBuf::new();
        d.reserve(unsafe { rh_agc_state_floats() });
        ck(unsafe { rh_agc_state_init(d.p, 1, self.pump.stream) }, "rh_agc_state_init");
        let st = State(d);
        self.push(move |c| { let st = &st; ck(unsafe { rh_agc(c.out, c.inp, c.n as u64, rate, 1, &settings, st.0.p, c.stream) }, "rh_agc"); c.n }, |n, _| n, 0);
        self
    }
    pub fn linear_gain_ramp(mut self, duration: Duration, start_gain: f32, end_gain: f32, clamp_end: bool) -> Self {   // linear_ramp.rs:79-110
        let (ch, rate, ns) = (self.ch as u32, self.rate, duration.as_nanos() as u64);
        let pos = std::sync::Arc::new(std::sync::atomic::AtomicU64::new(0));
        let pos2 = pos.clone();
        let stage = self.push(move |c| {
            let at = pos.load(std::sync::atomic::Ordering::Relaxed);
            ck(unsafe { rh_linear_gain_ramp(c.out, c.inp, c.n, at, ch, rate, ns, start_gain, end_gain, clamp_end as i32, c.stream) }, "rh_linear_gain_ramp");
            pos.store(at + c.n as u64, std::sync::atomic::Ordering::Relaxed);
            c.n
        }, |n, _| n, 0);
        stage.on_seek = Some(Box::new(move |p: Duration| {                          // linear_ramp.rs:141-146: elapsed = pos
            let ns = p.as_nanos() as u64;
            pos2.store((ns / 1_000_000_000 * rate as u64 + ns % 1_000_000_000 * rate as u64 / 1_000_000_000) * ch as u64, std::sync::atomic::Ordering::Relaxed);
        }));
        self
    }
    pub fn fade_in(self, duration: Duration) -> Self { self.linear_gain_ramp(duration, 0.0, 1.0, false) }   // fadein.rs:11-13
    pub fn fade_out(self, duration: Duration) -> Self { self.linear_gain_ramp(duration, 1.0, 0.0, true) }    // fadeout.rs:13
    /// `take_duration(d)`, with `set_filter_fadeout()` when `fade_out` (take.rs:96-148).  `try_seek` starts the duration over from the new
    /// position: what is left is the requested duration less `pos` (take.rs:222-231).
    pub fn take_duration(mut self, duration: Duration, fade_out: bool) -> Self {
        let (ch, rate, ns) = (self.ch as u32, self.rate, duration.as_nanos() as u64);
        // (remaining of the duration at the next sample, samples of the current frame already emitted, done, what was left at the start / behind the last seek)
        let tk = Arc::new(Mutex::new((ns, 0u32, false, ns)));
        let (tk2, tk3) = (tk.clone(), tk.clone());
        let per_sample = 1_000_000_000u64 / (rate as u64 * ch as u64);
        let stage = self.push(move |c| {
            let mut t = tk.lock().unwrap();
            c.end = true;
            if t.2 { return 0; }
            let (mut m, mut ended, mut after) = (0u64, 0i32, 0u64);
            ck(unsafe { rh_take_duration_from(c.out, c.inp, c.n as u64, t.0, ns, t.1, ch, rate, fade_out as i32, &mut m, &mut ended, &mut after, c.stream) }, "rh_take_duration_from");
            if per_sample > 0 { t.1 = ((t.1 as u64 + (t.0 - after) / per_sample) % ch as u64) as u32; }
            t.0 = after;
            t.2 = ended != 0;
            c.end = t.2;
            m as usize
        }, move |n, _| n + ch as usize, 2);
        stage.on_seek = Some(Box::new(move |pos: Duration| {                         // take.rs:222-231
            let mut t = tk2.lock().unwrap();
            let left = ns.saturating_sub(pos.as_nanos() as u64);
            *t = (left, 0, false, left);
        }));
        self.set_duration(move |inp| inp.map(|d| d.min(duration)));                   // take.rs:209-219
        self.set_hints(move |inp, e| {                                                // take.rs:151-171
            let since = tk3.lock().unwrap().3;
            if per_sample == 0 { return (0, Some(0)); }
            let took = e.min(since / per_sample);
            let remaining = since - took * per_sample;
            if remaining == 0 { return (0, Some(0)); }
            let rs = (remaining / per_sample) as usize;
            let h = inp(took);
            (h.0.min(rs), Some(h.1.map_or(rs, |u| u.min(rs))))
        });
        self
    }
    /// `delay(d)` (delay.rs:8-16,68-75): `rh_delay_samples()` zeros in front of the stream.  Not seekable here.
    pub fn delay(mut self, duration: Duration) -> Self {
        let d = unsafe { rh_delay_samples(duration.as_nanos() as u64, self.rate, self.ch as u32) };
        self.may_cut = self.may_cut || d % self.ch as u64 != 0;
        let mut first = true;
        let stage = self.push(move |c| {
            if first {
                first = false;
                ck(unsafe { rh_delay(c.out, c.inp, c.n as u64, d, c.stream) }, "rh_delay");
                return c.n + d as usize;
            }
            ck(unsafe { rh_amplify(c.out, c.inp, c.n, 1.0, c.stream) }, "rh_amplify");   // x * 1.0 == x: a copy into the other buffer
            c.n
        }, move |n, _| n + d as usize, 2);
        stage.seekable = false;
        self.set_duration(move |inp| inp.map(|v| v + duration));                      // delay.rs:111-115: + the REQUESTED delay
        self.set_hints(move |inp, e| {                                                // delay.rs:78-84: the input's bounds plus the silence still owed
            let owed = d.saturating_sub(e) as usize;
            let h = inp(e.saturating_sub(d));
            (h.0 + owed, h.1.map(|u| u + owed))
        });
        self
    }
}

impl<I: Source> BlockSource for GpuSource<I> {
    fn pump(&mut self) -> &mut Pump { &mut self.pump }
    fn block_done(&mut self) {                                                    // a bounded wait inside the limiter's scan expired (never seen on a healthy device): fail loudly
        if self.scan_kernels { ck(unsafe { rh_async_status() }, "rh_async_status"); }
    }
    fn enqueue(&mut self, i: usize) {
        let in_ch = self.in_ch as usize;
        let want = self.block_frames * in_ch;
        assert!(self.up.channels().get() == self.in_ch && self.up.sample_rate().get() == self.in_rate, "GpuSource: the upstream changed its format mid-stream");
        // the slot's page-locked block is about to be rewritten: the copy that read it two blocks ago must have run (a consumer that takes the
        // blocks on the device never waits on the host -- see GpuSource::enqueue in include/rodio_hip.hpp)
        if self.pump.device_out { ck(unsafe { rh_event_synchronize(self.pump.slot[i].done.0) }, "rh_event_synchronize"); }
        self.pump.slot[i].stage.reserve(want);
        { let h = self.up.size_hint(); self.hint.lock().unwrap().log.note(self.pulled_total, h); }   // (size_hint(): what the upstream answers where this block's first sample is pulled)
        let (mut n, flush);
        self.pieces.clear();
        {
            let stage = self.pump.slot[i].stage.slice_mut(want);
            if self.span_aware {                                                    // pull span by span, asking for the span where rodio asks
                n = 0;
                while n < want {
                    match self.reader.read_piece(&mut self.up, &mut stage[n..], (want - n) / in_ch) {
                        Some(pc) => { assert!(pc.ch == self.in_ch && pc.rate == self.in_rate, "GpuSource: the upstream changed its format mid-stream"); n += pc.n; self.pieces.push(pc); }
                        None => break,
                    }
                    if self.reader.ended() { break; }
                }
                flush = self.reader.ended();
            } else {
                n = read_into(&mut self.up, stage);
                n -= n % in_ch;                                                     // sources end on frame boundaries (source/mod.rs:169-178)
                flush = n < want;
            }
        }
        self.pulled_total += n as u64;
        // capacity of the ping-pong buffers: the largest block any stage can emit
        let (mut cap, mut m) = (want, want);
        for st in &self.stages { m = (st.bound)(m, self.pieces.len()); cap = cap.max(m); }
        cap = ((cap + 3) & !3) + 64;
        self.a.reserve(cap);
        self.b.reserve(cap);
        let (mut cur, mut oth) = (self.a.p, self.b.p);
        let stream = self.pump.stream;
        if n > 0 { ck(unsafe { rh_memcpy_h2d(cur.cast(), self.pump.slot[i].stage.p.cast(), n * 4, stream) }, "rh_memcpy_h2d"); }
        let mut ends = flush;                                                       // the upstream ended, or a stage says so
        for st in &mut self.stages {
            let mut c = Ctx { out: oth, inp: cur, n, out_cap: cap, flush: ends, stream, pieces: &self.pieces, end: false };
            n = (st.run)(&mut c);
            ends = ends || c.end;
            std::mem::swap(&mut cur, &mut oth);
        }
        if self.pump.device_out {   // the block stays on the device, in the slot's own buffer (a / b belong to the next block's stages)
            let sl = &mut self.pump.slot[i];
            sl.dev.reserve(cap);
            if sl.taken_pending { ck(unsafe { rh_stream_wait_event(stream, sl.taken.0) }, "rh_stream_wait_event"); sl.taken_pending = false; }   // the consumer's copies out of this slot's previous block
            if n > 0 { ck(unsafe { rh_memcpy_d2d(sl.dev.p.cast(), cur.cast(), n * 4, stream) }, "rh_memcpy_d2d"); }
        } else if n > 0 {
            self.pump.slot[i].out.reserve(cap);
            ck(unsafe { rh_memcpy_d2h_async(self.pump.slot[i].out.p.cast(), cur.cast(), n * 4, stream) }, "rh_memcpy_d2h_async");
            self.pump.timing.d2h_samples += n as u64;
        }
        self.pump.slot[i].n = n;
        self.pump.slot[i].last = ends;
    }
}
/// A chain the mixer may take on the device (`GpuMixer::add_chain`): a `Source` that can also hand its blocks over device-to-device.
pub trait DeviceChain: Send {
    fn as_source(&mut self) -> &mut dyn Source;
    fn as_source_ref(&self) -> &dyn Source;
    fn keep_blocks_on_device(&mut self, on: bool);
    fn blocks_on_device(&self) -> bool;
    fn started(&self) -> bool;
    fn read_device(&mut self, ddst: *mut f32, n: usize, consumer: RhStream) -> usize;
    fn timing(&self) -> Timing;
    /// size_hint(): the chain's own arithmetic (kept alive beyond the chain), primed with the upstream's answer while nothing has been pulled; and
    /// whether the chain ends with a `uniform` (1), with a filter right behind one (2), or with anything else (0).
    fn hint_state(&self) -> std::sync::Arc<std::sync::Mutex<HintState>>;
    fn last_kind(&self) -> u8;
}
impl<I: Source + Send> DeviceChain for GpuSource<I> {
    fn as_source(&mut self) -> &mut dyn Source { self }
    fn as_source_ref(&self) -> &dyn Source { self }
    fn keep_blocks_on_device(&mut self, on: bool) { GpuSource::keep_blocks_on_device(self, on) }
    fn blocks_on_device(&self) -> bool { GpuSource::blocks_on_device(self) }
    fn started(&self) -> bool { GpuSource::started(self) }
    fn read_device(&mut self, ddst: *mut f32, n: usize, consumer: RhStream) -> usize { GpuSource::read_device(self, ddst, n, consumer) }
    fn timing(&self) -> Timing { GpuSource::timing(self) }
    fn hint_state(&self) -> std::sync::Arc<std::sync::Mutex<HintState>> {
        { let mut h = self.hint.lock().unwrap(); if h.log.is_empty() { h.before = self.up.size_hint(); } }
        self.hint.clone()
    }
    fn last_kind(&self) -> u8 { self.last_kind }
}
impl<I: Source> Iterator for GpuSource<I> {
    type Item = f32;
    fn next(&mut self) -> Option<f32> { self.next_sample() }
    /// As rodio's adapter chain would answer where the CONSUMER stands (the chain itself reads a block ahead): every adapter's own arithmetic
    /// (delay.rs:78-84, take.rs:151-171, mix.rs:56-67, channels.rs:88-102, sample_rate.rs:204-238, uniform.rs:100-108; the input's answer behind
    /// the adapters that hand it on: amplify.rs:68-70, blt.rs:144-146, ...) over the upstream's answer at the sample the question reaches.
    fn size_hint(&self) -> (usize, Option<usize>) { self.size_hint_at(self.pump.handed_out) }
}
impl<I: Source> Source for GpuSource<I> {
    fn current_span_len(&self) -> Option<usize> { None }
    fn channels(&self) -> ChannelCount { ChannelCount::new(self.ch).unwrap() }
    fn sample_rate(&self) -> SampleRate { SampleRate::new(self.rate).unwrap() }
    fn total_duration(&self) -> Option<Duration> { self.duration_through() }
    /// `try_seek` through the chain, adapter by adapter as rodio does it: an adapter that cannot seek (reverb = Mix, mix.rs:116-120)
    /// fails the call before anything moved; otherwise the upstream seeks, what was pulled and processed ahead is dropped, and
    /// every adapter does to its state what its `try_seek` does (blt.rs:350-377, limit.rs:1139-1158, linear_ramp.rs:141-146).
    fn try_seek(&mut self, pos: Duration) -> Result<(), SeekError> {
        if self.stages.iter().any(|s| !s.seekable) { return Err(SeekError::NotSupported { underlying_source: "rodio_hip::GpuSource (reverb / delay in the chain)" }); }
        self.up.try_seek(pos)?;
        self.reader.restart();
        let ch = self.ch as usize;
        self.pump.restart(ch);
        for st in &mut self.stages { if let Some(f) = st.on_seek.as_mut() { f(pos); } }
        {   // (the adapters' counts start over: what the chain emits from here on is the stream behind the new position, less the samples that keep the consumer's channel)
            let mut h = self.hint.lock().unwrap();
            h.log.clear();
            h.out_base = self.pump.handed_out;
            h.skip = self.pump.handed_out % ch as u64;
            h.in_base = self.pulled_total;
        }
        Ok(())
    }
}

// ------------------------------------------------------------------------------------------------ GpuMixer ----
/// The filter a source carries into the mixer: `mixer.add(UniformSourceIterator::new(src, ch, rate).low_pass(200))` for one,
/// `.high_pass(300)` for the next, none for a third (source/mod.rs:686-721: every source has its own adapters).
#[derive(Clone, Copy, Debug, PartialEq)]
pub struct MixerFilter {
    pub kind: i32,    // -1 none, 0 low_pass, 1 high_pass
    pub freq: u32,
    pub q: f32,       // rodio's low_pass() / high_pass() use 0.5 (blt.rs:11-24)
}
impl MixerFilter {
    pub fn none() -> Self { MixerFilter { kind: -1, freq: 0, q: 0.5 } }
    pub fn low_pass(freq: u32) -> Self { MixerFilter { kind: 0, freq, q: 0.5 } }
    pub fn high_pass(freq: u32) -> Self { MixerFilter { kind: 1, freq, q: 0.5 } }
    fn same(&self, o: &MixerFilter) -> bool { self.kind == o.kind && (self.kind < 0 || (self.freq == o.freq && self.q == o.q)) }
}

#[derive(Clone, Copy)]
pub struct MixerOptions {
    pub block_frames: usize,         // frames pulled per source and block
    pub filter_kind: i32,            // the filter of sources added WITHOUT one of their own: -1 none, 0 low_pass, 1 high_pass (q = 0.5, blt.rs:11-24)
    pub filter_freq: u32,
    pub filter_q: f32,
    pub frames_per_lane: u32,        // 0 = the library's choice
    pub host_threads: u32,           // threads that pull the sources of a block (0 = min(cores, 16); 1 = the caller alone)
    /// THE FILTER CONTRACT (`rodio_hip.h`, `rh_filter_scan_ok`).  true (default): a source whose filter lies outside the region in which the
    /// fused kernel's time-parallel filter stays within 1e-5 of rodio's own recurrence gets a chain of its own -- amplify ->
    /// UniformSourceIterator -> the filter in rodio's operation order, bit for bit -- and enters the mix unfiltered, on the device.
    pub reference_exact_filters: bool,
}
impl Default for MixerOptions {
    fn default() -> Self { MixerOptions { block_frames: 1 << 15, filter_kind: -1, filter_freq: 0, filter_q: 0.5, frames_per_lane: 0, host_threads: 0, reference_exact_filters: true } }
}

/// What became of the `GpuSource` chains handed to `add_chain`: how many there were, how many delivered their blocks on the device, the
/// samples of chain output that crossed to the host (0 when every chain stayed on the device) / went device-to-device.
#[derive(Clone, Copy, Default, Debug)]
pub struct ChainStats { pub chains: u64, pub on_device: u64, pub d2h_samples: u64, pub device_samples: u64 }

enum Upstream { Host(Box<dyn Source + Send>), Chain(Box<dyn DeviceChain>) }
impl Upstream {
    fn src(&mut self) -> &mut dyn Source { match self { Upstream::Host(s) => s.as_mut(), Upstream::Chain(c) => c.as_source() } }
    fn src_ref(&self) -> &dyn Source { match self { Upstream::Host(s) => s.as_ref(), Upstream::Chain(c) => c.as_source_ref() } }
    fn on_device(&self) -> bool { matches!(self, Upstream::Chain(c) if c.blocks_on_device()) }
}
struct Src {
    up: Upstream, gain: f32, filt: MixerFilter, held: Vec<f32>, ended: bool, ch: u16,
    dheld: u64, dheld_off: u64,                                                  // a device chain: frames the converter has not consumed, in the row of the block before
    reader: SpanReader, plan: UniformPlanner, have_s: u64, off_s: u64,           // span-by-span generations: converted SAMPLES not yet mixed, from sample off_s of the row
    total_s: u64,                                                                 // samples of the source's stream in the mix's layout so far (a stream that ends inside a frame: the last block is cut to what rodio returns)
    hint: Option<std::sync::Arc<std::sync::Mutex<HintTrack>>>,                   // size_hint(): see HintTrack
}
unsafe impl Send for Src {}
impl Src {
    /// A pull of `got` samples of ONE continuous span (the source reports none); `ended`: it returned None behind them.
    fn note_pull(&mut self, got: usize, ended: bool) {
        let Some(h) = &self.hint else { return };
        let mut t = h.lock().unwrap();
        let (first, rate, at) = (t.pulled == 0, self.up.src_ref().sample_rate().get(), t.pulled);
        if let Some(c) = t.counter.as_mut() {
            if got > 0 || ended { c.feed(&Piece { n: got, opens: first, closes: ended, ch: self.ch, rate, tail: 0, by_none: ended, limit: None }, at); }
            if ended { c.input_ended(); }
        }
        t.pulled += got as u64;
    }
    fn note_hint(&mut self) {   // (before a pull of a plain source: what it answers where the pull starts)
        let Some(h) = &self.hint else { return };
        let mut t = h.lock().unwrap();
        if t.chain.is_none() { let (at, ans) = (t.pulled, self.up.src_ref().size_hint()); t.log.note(at, ans); }
    }
}
/// `size_hint()` of ONE source of the mix, where the mixer's consumer stands.  rodio's spelling of `add(src, gain, filter)`:
///     mixer.add(src)                                                                gain 1, no filter
///     mixer.add(src.amplify(g))                                                     a gain (amplify.rs:68-70 hands the bounds on)
///     mixer.add(UniformSourceIterator::new(src.amplify(g), ch, rate).low_pass(f))   a filter: it runs at the mixer's rate, behind a converter of its own
/// and Mixer::add wraps what it gets in a UniformSourceIterator (mixer.rs:58-66).  So the bounds are those of ONE iterator over the source
/// (`counter`: the mixer itself converts; or the chain's own last `uniform`), and behind a filter of one more -- a pass-through at the mixer's
/// own format, whose ChannelCountConverter still counts in whole frames from where it stands (channels.rs:88-102).  The twin of
/// `GpuMixer::HintTrack` (include/rodio_hip.hpp).
pub struct HintTrack {
    join_frame: u64, counter: Option<UniformCounter>, log: HintLog, chain: Option<std::sync::Arc<std::sync::Mutex<HintState>>>, pulled: u64,
    wrapped_again: bool, total_known: bool, total: u64,
}
impl HintTrack {
    fn lower_at(&mut self, e: u64, ch: u16) -> Option<usize> {
        let HintTrack { counter, log, chain, .. } = self;
        let mut source_at = |q: u64| -> SizeHint { match chain { Some(c) => c.lock().unwrap().at(q), None => if log.is_empty() { (0, None) } else { log.at(q) } } };
        let lo = match counter {
            Some(c) => c.hint_at(e, &mut source_at)?.0,
            None => { if self.total_known && e > self.total { return None; } source_at(e).0 }
        };
        if !self.wrapped_again { return Some(lo); }
        let pos = (e % ch as u64) as usize;                                         // channels.rs:88-102 with from == to
        Some(((lo + pos) / ch as usize * ch as usize).saturating_sub(pos))
    }
}
fn count_chain(x: &Src, st: &mut ChainStats) {
    if let Upstream::Chain(c) = &x.up {
        st.chains += 1;
        st.on_device += c.blocks_on_device() as u64;
        st.d2h_samples += c.timing().d2h_samples;
        st.device_samples += c.timing().device_samples;
    }
}
/// Sources that joined together: one clock, one fused stream.
struct Gen {
    srcs: Vec<Src>, plan: *mut RhRlm, filt: MixerFilter,
    din: [DeviceBuf; 3], dnext: usize, pd: usize, pd_prev: Option<usize>,          // staged input rows, THREE sets in rotation: a block's rows stay untouched until the block after it has run
    q: [DeviceBuf; 2], stage: [PinnedBuf; 2], side: [PinnedBuf; 2], dside: [DeviceBuf; 2], copied: [Event; 2],
    cur: usize, slot: usize, head: u64, fill: u64, done: bool,
    // the host half of a block that has been pulled and not yet issued (pull_block / issue_block)
    pulled: bool, pslot: usize, pptrs: Vec<*const f32>, pavail: Vec<u64>, pended: Vec<u8>, pside_off: Vec<usize>, pside: usize, ptable: Vec<RhUniformSeg>, pmax_out: u64,
    mono: bool, qm: DeviceBuf,                                                       // a fused stream of mono sources (channels = 1): the mono mix, made stereo once per block
    staged: bool, target: u64, crow: u64, conv: [DeviceBuf; 2], dtab: DeviceBuf, tab: [PinnedBuf; 2], ccur: usize,
}
unsafe impl Send for Gen {}   // (retired generations are freed by the reaper thread)
impl Gen {
    fn queue(&self) -> *const f32 { unsafe { self.q[self.cur].p.add(self.head as usize * 2) } }
    fn queue_end(&self) -> *mut f32 { unsafe { self.q[self.cur].p.add((self.head + self.fill) as usize * 2) } }
}
impl Drop for Gen { fn drop(&mut self) { if !self.plan.is_null() { unsafe { rh_rlm_destroy(self.plan); } } } }

/// Frees retired generations (plans, page-locked blocks, device rows) on a thread of its own: the consumer's `next()` -- the audio
/// callback -- only hands them over.
struct Reaper { tx: Option<mpsc::Sender<Vec<Gen>>>, th: Option<std::thread::JoinHandle<()>> }
impl Reaper {
    fn new() -> Self {
        let (tx, rx) = mpsc::channel::<Vec<Gen>>();
        let th = std::thread::spawn(move || { unsafe { rh_bind_thread(); } for dead in rx { drop(dead); } });   // (HIP's current device is per thread) the destructors: rh_rlm_destroy, rh_host_free, rh_free
        Reaper { tx: Some(tx), th: Some(th) }
    }
    fn retire(&self, g: Vec<Gen>) { let _ = self.tx.as_ref().unwrap().send(g); }
}
impl Drop for Reaper { fn drop(&mut self) { self.tx.take(); if let Some(t) = self.th.take() { let _ = t.join(); } } }

/// What rodio spells `let (mixer, mixed) = mixer::mixer(channels, rate); mixer.add(src.amplify(g)) ...` (every added source goes
/// through `UniformSourceIterator::new(.., rate)`, optionally `.low_pass(f)` / `.high_pass(f)` of its own) as ONE source.
pub struct GpuMixer {
    rate: u32, out_ch: u16, opt: MixerOptions, pending: Vec<Src>, gens: Vec<Gen>,
    cap_frames: usize, row: usize, out_cap_frames: u64, scheduled: u64, last_join: u64,
    dmix: DeviceBuf, dout: DeviceBuf, dkeep: [DeviceBuf; 2], slot_base: [u64; 2], slot_frames: [u64; 2], calls: u64, resume_ok: bool,
    copy_stream: RhStream,              // host-to-device copies of the staged rows: the link stays busy while the pump's stream runs the block before
    device_chains: bool, retired_chains: ChainStats, reaper: Option<Reaper>,
    pump: Pump,
    hints: Mutex<Vec<Arc<Mutex<HintTrack>>>>,   // size_hint(): the sources that play, or are about to (outlive their generation: the last blocks are served after it is retired)
}
unsafe impl Send for GpuMixer {}

fn fused_ratio_unsupported(from: u32, to: u32) -> bool {                          // rh_rlm_create: reduced from/to <= 4.5, from*to within u32
    let (mut a, mut b) = (from as u64, to as u64);
    while b != 0 { let t = a % b; a = b; b = t; }
    let (f, t) = (from as u64 / a, to as u64 / a);
    2 * f > 9 * t || f * t > 0xffff_ffff
}

impl GpuMixer {
    /// `mixer::mixer(2, sample_rate)`.
    pub fn new(sample_rate: SampleRate, opt: MixerOptions) -> Self { Self::with_channels(ChannelCount::new(2).unwrap(), sample_rate, opt) }
    /// `mixer::mixer(channels, sample_rate)` (mixer.rs:25).  The sources are mixed as stereo frames; `channels` other than 2 is what
    /// ChannelCountConverter makes of every source (channels.rs:57-85), applied ONCE, to the mixed block (the converter and the sum
    /// commute: 1 keeps channel 0, more than 2 appends silent channels -- for that, sources must not have more than 2 channels).
    pub fn with_channels(channels: ChannelCount, sample_rate: SampleRate, opt: MixerOptions) -> Self {
        let mut opt = opt;
        opt.block_frames = opt.block_frames.max(1);
        let mut copy_stream: RhStream = ptr::null_mut();
        ck(unsafe { rh_stream_create(&mut copy_stream) }, "rh_stream_create");
        GpuMixer { rate: sample_rate.get(), out_ch: channels.get(), opt, pending: Vec::new(), gens: Vec::new(), cap_frames: 0, row: 0, out_cap_frames: 0, scheduled: 0, last_join: 0,
                   dmix: DeviceBuf::new(), dout: DeviceBuf::new(), dkeep: [DeviceBuf::new(), DeviceBuf::new()], slot_base: [0; 2], slot_frames: [0; 2], calls: 0, resume_ok: true,
                   copy_stream, device_chains: false, retired_chains: ChainStats::default(), reaper: None, pump: Pump::new(), hints: Mutex::new(Vec::new()) }
    }
    fn default_filter(&self) -> MixerFilter { MixerFilter { kind: self.opt.filter_kind, freq: self.opt.filter_freq, q: self.opt.filter_q } }
    /// `Mixer::add` (mixer.rs:58-66), with the source's volume (`mixer.add(src.amplify(gain))`).  May be called at any time.
    pub fn add(&mut self, src: Box<dyn Source + Send>, gain: f32) { let f = self.default_filter(); self.add_filtered(src, gain, f); }
    /// ... with the source's own filter (behind its UniformSourceIterator, at the mixer's rate).  Sources of one filter share a fused
    /// stream -- one launch per block for all of them, summed first while they run together -- and the streams' mixes are added.
    pub fn add_filtered(&mut self, src: Box<dyn Source + Send>, gain: f32, filter: MixerFilter) {
        let ch = src.channels().get();
        assert!(!(self.out_ch > 2 && ch > 2), "GpuMixer: a source of more than 2 channels into a mixer of more than 2 (the mix is formed in stereo)");
        assert!(filter.kind <= 1, "filter kind");
        if filter.kind >= 0 && self.opt.reference_exact_filters && unsafe { rh_filter_scan_ok(filter.kind, filter.freq, filter.q, self.rate) } == 0 {
            // outside the filter contract: the source's own chain, the filter in the reference's order, the mixer only sums
            let mut chain = GpuSource::new(src, self.opt.block_frames).exact_filters(true);
            if gain != 1.0 { chain = chain.amplify(gain); }
            chain = chain.uniform(ChannelCount::new(2).unwrap(), SampleRate::new(self.rate).unwrap());
            chain = if filter.kind == 0 { chain.low_pass_with_q(filter.freq, filter.q) } else { chain.high_pass_with_q(filter.freq, filter.q) };
            self.add_chain_filtered(Box::new(chain), 1.0, MixerFilter::none());
            return;
        }
        let item = Src { up: Upstream::Host(src), gain, filt: filter, held: Vec::new(), ended: false, ch, dheld: 0, dheld_off: 0,
                         reader: SpanReader::new(), plan: UniformPlanner::new(2, self.rate), have_s: 0, off_s: 0, total_s: 0, hint: None };
        if self.pump.running() { self.late_join(item); } else { self.pending.push(item); }
    }
    /// A `GpuSource` chain handed to the mixer by value, as rodio's adapters are (`mixer.add(src.reverb(..).limit(..))`, amplify.rs:19-22,
    /// mixer.rs:58-72): its blocks stay in device memory and the mixer takes them device-to-device -- the chain's output never crosses
    /// to the host and back.  (A chain that is not stereo at a rate the fused converter takes, or one that has already started, is
    /// pulled like any other source: the same samples, through the host.)
    pub fn add_chain(&mut self, chain: Box<dyn DeviceChain>, gain: f32) { let f = self.default_filter(); self.add_chain_filtered(chain, gain, f); }
    pub fn add_chain_filtered(&mut self, mut chain: Box<dyn DeviceChain>, gain: f32, filter: MixerFilter) {
        let (ch, rate) = (chain.as_source_ref().channels().get(), chain.as_source_ref().sample_rate().get());
        let on_device = ch == 2 && !chain.started() && !fused_ratio_unsupported(rate, self.rate);
        if on_device { chain.keep_blocks_on_device(true); self.device_chains = true; }
        let item = Src { up: Upstream::Chain(chain), gain, filt: filter, held: Vec::new(), ended: false, ch, dheld: 0, dheld_off: 0,
                         reader: SpanReader::new(), plan: UniformPlanner::new(2, self.rate), have_s: 0, off_s: 0, total_s: 0, hint: None };
        if self.pump.running() { self.late_join(item); } else { self.pending.push(item); }
    }
    /// Everything a first `next()` would do before it can serve a sample, done NOW on the calling thread (see `BlockSource::prepare_stream`):
    /// call it on the control thread before the mixer goes to the audio callback.
    pub fn prepare(&mut self) { self.prepare_stream(); }
    pub fn timing(&self) -> Timing { self.pump.timing }
    pub fn chain_stats(&self) -> ChainStats {
        let mut st = self.retired_chains;
        for g in &self.gens { for x in &g.srcs { count_chain(x, &mut st); } }
        for x in &self.pending { count_chain(x, &mut st); }
        st
    }
    /// Output frame (of this mixer) at which the most recently started generation joined.
    /// Blocks of generations of MORE than two channels that ran as one launch (`rh_wide_mix_block`, declared in [`ffi`]: Amplify ->
    /// SampleRateConverter -> ChannelCountConverter per source and the ordered sum, any layouts side by side).  The C++ mirror plans such
    /// blocks (`GpuMixer::pull_block_widefused` of include/rodio_hip.hpp); this twin still forms every mix in stereo and widens it once
    /// (see `with_channels`), so it never issues one: always 0 here.  INTEGRATION.md section 1 lists the difference.
    pub fn wide_fused_blocks(&self) -> u64 { 0 }
    pub fn last_join_frame(&self) -> u64 { self.last_join }
    /// Threads that pull a block's sources.
    pub fn pull_threads(&self) -> u32 {
        if self.opt.host_threads != 0 { self.opt.host_threads } else { (std::thread::available_parallelism().map(|n| n.get()).unwrap_or(1) as u32).min(16) }
    }

    fn make_direct(&self, x: &mut Src) {
        // A continuous source the fused kernel cannot take as it is (rate ratio above 4.5) gets the GPU converter adapter in front.
        if x.up.on_device() || !fused_ratio_unsupported(x.up.src_ref().sample_rate().get(), self.rate) { return; }
        let up = match std::mem::replace(&mut x.up, Upstream::Host(Box::new(rodio::source::Empty::new()))) {
            Upstream::Host(s) => s,
            Upstream::Chain(c) => Box::new(ChainAsSource(c)) as Box<dyn Source + Send>,
        };
        let mut conv = GpuSource::new(up, self.opt.block_frames);
        if x.ch != 2 { conv = conv.convert_channels(ChannelCount::new(2).unwrap()); }
        conv = conv.convert_sample_rate(SampleRate::new(self.rate).unwrap());
        x.up = Upstream::Host(Box::new(conv));
        x.ch = 2;
    }
    fn start_generation(&mut self) {
        let mut all = std::mem::take(&mut self.pending);
        if all.iter().any(|x| x.up.src_ref().current_span_len().is_some()) {       // span by span, as rodio converts them: one stream per filter, insertion order
            let mut fk: Vec<MixerFilter> = Vec::new();
            for x in &all { if !fk.iter().any(|f| f.same(&x.filt)) { fk.push(x.filt); } }
            for f in fk {
                let (group, rest): (Vec<Src>, Vec<Src>) = all.into_iter().partition(|x| x.filt.same(&f));
                all = rest;
                self.start_stream(group, true, false);
            }
            return;
        }
        for x in &mut all { self.make_direct(x); }
        // continuous sources: one fused stream per (input rate, mono or not, filter), in order of first appearance; mono sources form
        // streams of mono frames (the kernel reads 4 bytes per frame, the mono mix becomes stereo once per block)
        let mut kinds: Vec<(u32, bool, MixerFilter)> = Vec::new();
        for x in &all {
            let k = (x.up.src_ref().sample_rate().get(), x.ch == 1, x.filt);
            if !kinds.iter().any(|o| o.0 == k.0 && o.1 == k.1 && o.2.same(&k.2)) { kinds.push(k); }
        }
        for (r, mono, f) in kinds {
            let (group, rest): (Vec<Src>, Vec<Src>) = all.into_iter().partition(|x| x.up.src_ref().sample_rate().get() == r && (x.ch == 1) == mono && x.filt.same(&f));
            all = rest;
            self.start_stream(group, false, mono);
        }
    }
    fn start_stream(&mut self, srcs: Vec<Src>, staged: bool, mono: bool) {
        let from = if staged { self.rate } else { srcs[0].up.src_ref().sample_rate().get() };
        let filt = srcs[0].filt;                                                   // (one filter per stream: start_generation / late_join group by it)
        self.cap_frames = self.opt.block_frames + 4096;                            // a block can hold what the previous one left over
        let mut g = Gen { srcs, plan: ptr::null_mut(), filt, din: [DeviceBuf::new(), DeviceBuf::new(), DeviceBuf::new()], dnext: 0, pd: 0, pd_prev: None,
                          q: [DeviceBuf::new(), DeviceBuf::new()], stage: [PinnedBuf::new(), PinnedBuf::new()], side: [PinnedBuf::new(), PinnedBuf::new()],
                          dside: [DeviceBuf::new(), DeviceBuf::new()], copied: [Event::new(), Event::new()], cur: 0, slot: 0, head: 0, fill: 0, done: false,
                          pulled: false, pslot: 0, pptrs: Vec::new(), pavail: Vec::new(), pended: Vec::new(), pside_off: Vec::new(), pside: 0, ptable: Vec::new(), pmax_out: 0,
                          mono: mono && !staged, qm: DeviceBuf::new(), staged, target: 0, crow: 0, conv: [DeviceBuf::new(), DeviceBuf::new()], dtab: DeviceBuf::new(),
                          tab: [PinnedBuf::new(), PinnedBuf::new()], ccur: 0 };
        if staged {
            g.target = self.opt.block_frames as u64 + 64 * 20 + 8;                 // a block emits whole tiles (at most 64 * 20 frames) and keeps two frames of history
            g.crow = g.target + 64;
            let crowf = (g.crow as usize * 2 + 3) & !3;
            for b in &mut g.conv { b.reserve(g.srcs.len() * crowf); }
        }
        let cfg = RhRlmConfig {
            from_rate: from, to_rate: self.rate, channels: if g.mono { 1 } else { 2 }, span_len: 0, filter_kind: filt.kind, filter_freq: filt.freq, filter_q: filt.q,
            max_sources: g.srcs.len() as u32, max_in_frames: if staged { g.crow } else { self.cap_frames as u64 },
            frames_per_lane: self.opt.frames_per_lane, ring_stages: 0, no_balance: 0, force_general: 0, custom_coeffs: [0.0; 5], filter_first: 0,
        };
        ck(unsafe { rh_rlm_create(&mut g.plan, &cfg) }, "rh_rlm_create");
        ck(unsafe { rh_rlm_set_exclusive(g.plan, 0) }, "rh_rlm_set_exclusive");   // the copy stream's launches (and other mixers) share the CUs: tiles by ticket
        // staged: the factor sits in front of the converter, where Mixer::add(src.amplify(g)) has it
        let gains: Vec<f32> = g.srcs.iter().map(|x| if staged { 1.0 } else { x.gain }).collect();
        ck(unsafe { rh_rlm_set_gains(g.plan, gains.as_ptr(), gains.len() as u32) }, "rh_rlm_set_gains");
        ck(unsafe { rh_rlm_stream_begin(g.plan) }, "rh_rlm_stream_begin");
        // sources that start together run together until the first of them ends: their blocks are summed first (rodio_hip.h).  The rows of a
        // direct generation rotate through three sets, which is what the recovery at that moment needs; converted rows (staged) do not.
        if !staged { ck(unsafe { rh_rlm_stream_keep_history(g.plan, 1) }, "rh_rlm_stream_keep_history"); }
        self.row = (self.cap_frames * 2 + 3) & !3;
        let mut m = 0u64;
        ck(unsafe { rh_resample_out_frames(if staged { g.crow } else { self.cap_frames as u64 }, from, self.rate, 2, 0, &mut m) }, "rh_resample_out_frames");
        if m + 64 > self.out_cap_frames {
            self.out_cap_frames = m + 64;
            for o in &mut self.gens {                                               // rates differ between generations: every queue holds two of the largest blocks
                for k in 0..2 {
                    let want = self.out_cap_frames as usize * 4;
                    if o.q[k].n < want {
                        let mut nb = DeviceBuf::new();
                        nb.reserve(want);
                        let keep = ((o.head + o.fill) * 2) as usize;
                        if keep > 0 && k == o.cur { ck(unsafe { rh_memcpy_d2d(nb.p.cast(), o.q[k].p.cast(), keep * 4, self.pump.stream) }, "rh_memcpy_d2d"); }
                        ck(unsafe { rh_stream_synchronize(self.pump.stream) }, "rh_stream_synchronize");
                        o.q[k] = nb;
                    }
                }
            }
        }
        for b in &mut g.q { b.reserve(self.out_cap_frames as usize * 4); }
        self.last_join = self.scheduled;
        for x in &mut g.srcs {   // size_hint(): the generation's sources start playing (late_join moves join_frame to the frame they join at)
            let (chain, kind) = match &x.up { Upstream::Chain(c) => (Some(c.hint_state()), c.last_kind()), Upstream::Host(_) => (None, 0) };
            let (sch, srate) = (x.up.src_ref().channels().get(), x.up.src_ref().sample_rate().get());
            let same = sch == self.out_ch && srate == self.rate && x.gain == 1.0 && x.filt.kind < 0;
            let own = chain.is_some() && same && (kind == 1 || kind == 2);           // the chain ends with its own iterator (and a filter behind it): nothing for the mixer's to do
            let t = HintTrack { join_frame: self.scheduled, counter: if own { None } else { Some(UniformCounter::new(self.out_ch, self.rate, false)) }, log: HintLog::default(), chain,
                                pulled: 0, wrapped_again: if own { kind == 2 } else { x.filt.kind >= 0 }, total_known: false, total: 0 };
            let t = Arc::new(Mutex::new(t));
            x.hint = Some(t.clone());
            self.hints.lock().unwrap().push(t);
        }
        self.gens.push(g);
    }

    /// One block of a generation = its host half (pull_block: the upstreams are pulled into a page-locked staging block, by a few
    /// threads, and the copy is queued on the copy stream) and its device half (issue_block: launches on the pump's stream, behind the
    /// copy's event).  The pump runs the host half of the next block while the device still works on the block about to be served.
    fn run_block(&mut self, gi: usize) {
        if !self.gens[gi].pulled { self.pull_block(gi); }
        self.issue_block(gi);
    }
    fn pull_block(&mut self, gi: usize) {
        {
            let g = &mut self.gens[gi];
            g.pslot = g.slot;
            g.slot ^= 1;
            g.pd_prev = if g.pd_prev.is_none() && g.dnext == 0 { None } else { Some(g.pd) };
            g.pd = g.dnext;
            g.dnext = (g.dnext + 1) % 3;
        }
        let t0 = Instant::now();
        if self.gens[gi].staged { self.pull_block_staged(gi) } else { self.pull_block_direct(gi) }
        self.pump.timing.pull_s += t0.elapsed().as_secs_f64();
        self.gens[gi].pulled = true;
    }
    fn issue_block(&mut self, gi: usize) {
        self.gens[gi].pulled = false;
        if self.gens[gi].staged { self.issue_block_staged(gi) } else { self.issue_block_direct(gi) }
    }

    /// A generation of one format.  Row i of the page-locked block = [frames the previous block left unconsumed | one freshly pulled
    /// block]; a source that is not in the layout the fused launch reads has its row in the side block, in its own layout; a chain
    /// that hands its blocks over on the device has its row filled device-to-device.
    fn pull_block_direct(&mut self, gi: usize) {
        let (cap_frames, block_frames, copy_stream, threads) = (self.cap_frames, self.opt.block_frames, self.copy_stream, self.pull_threads() as usize);
        let row_stereo = self.row;
        let g = &mut self.gens[gi];
        let native: usize = if g.mono { 1 } else { 2 };                            // channels of the rows the fused launch reads
        let row_len = if g.mono { (cap_frames + 3) & !3 } else { row_stereo };      // floats per row
        let s_n = g.srcs.len();
        let (slot, pd) = (g.pslot, g.pd);
        g.stage[slot].reserve(s_n * row_len);
        g.din[pd].reserve(s_n * row_len);
        g.pptrs = vec![ptr::null(); s_n];
        g.pavail = vec![0; s_n];
        g.pended = vec![0; s_n];
        g.pside_off = vec![0; s_n];
        g.pside = 0;
        for (i, x) in g.srcs.iter().enumerate() {
            if x.ch as usize != native { g.pside_off[i] = g.pside; g.pside += (cap_frames * x.ch as usize + 3) & !3; }
        }
        if g.pside > 0 { g.side[slot].reserve(g.pside); g.dside[slot].reserve(g.pside); }
        // the host sources, a few at a time on scoped threads: sources are independent objects, one thread drives one source at a time
        let (stage_p, side_p, din_p) = (SendPtr(g.stage[slot].p), SendPtr(g.side[slot].p), SendPtr(g.din[pd].p));
        let side_off = g.pside_off.clone();
        let mut results: Vec<(u64, u8)> = vec![(0, 0); s_n];
        {
            let pull_one = |i: usize, x: &mut Src, res: &mut (u64, u8)| {
                if x.up.on_device() { return; }                                     // its block arrives device-to-device, below
                let ch = x.ch as usize;
                let row: &mut [f32] = unsafe {
                    if ch == native { std::slice::from_raw_parts_mut(stage_p.0.add(i * row_len), row_len) } else { std::slice::from_raw_parts_mut(side_p.0.add(side_off[i]), cap_frames * ch) }
                };
                let mut have = x.held.len();
                assert!(have / ch + if x.ended { 0 } else { block_frames } <= cap_frames, "GpuMixer: held frames exceed the plan");
                row[..have].copy_from_slice(&x.held);
                if !x.ended {
                    let want = block_frames * ch;
                    x.note_hint();
                    let mut got = read_into(x.up.src(), &mut row[have..have + want]);   // straight into the staging block
                    got -= got % ch;                                                // sources end on frame boundaries (source/mod.rs:169-178)
                    have += got;
                    x.ended = got < want;
                    x.note_pull(got, x.ended);
                }
                *res = ((have / ch) as u64, x.ended as u8);
            };
            let big = s_n * block_frames >= (1 << 18) && threads > 1 && s_n > 1;
            if big {
                let per = (s_n + threads - 1) / threads;
                std::thread::scope(|sc| {
                    for (k, (xs, rs)) in g.srcs.chunks_mut(per).zip(results.chunks_mut(per)).enumerate() {
                        let pull_one = &pull_one;
                        sc.spawn(move || { for (j, (x, r)) in xs.iter_mut().zip(rs.iter_mut()).enumerate() { pull_one(k * per + j, x, r); } });
                    }
                });
            } else {
                for (i, (x, r)) in g.srcs.iter_mut().zip(results.iter_mut()).enumerate() { pull_one(i, x, r); }
            }
        }
        for i in 0..s_n {
            g.pptrs[i] = unsafe { din_p.0.add(i * row_len) } as *const f32;
            g.pavail[i] = results[i].0;
            g.pended[i] = results[i].1;
        }
        // one copy for all rows -- what the longest row holds of every row: the rows differ by a few frames, the unused ends stay -- on
        // the copy stream: it runs beside the launches of the block before
        let width = (0..s_n).filter(|&i| g.srcs[i].ch as usize == native).map(|i| g.pavail[i] as usize * native).max().unwrap_or(0);
        unsafe {
            if width > 0 { ck(rh_memcpy_h2d_rows(g.din[pd].p.cast(), g.stage[slot].p.cast(), row_len * 4, width * 4, s_n, copy_stream), "rh_memcpy_h2d_rows"); }
            if g.pside > 0 { ck(rh_memcpy_h2d(g.dside[slot].p.cast(), g.side[slot].p.cast(), g.pside * 4, copy_stream), "rh_memcpy_h2d"); }
        }
        // chains that hand their blocks over on the device: [what the converter left of the block before | the chain's next samples],
        // device-to-device on the copy stream, behind the pitched copy
        for i in 0..s_n {
            if !g.srcs[i].up.on_device() { continue; }
            let row = unsafe { g.din[pd].p.add(i * row_len) };
            let prev = g.pd_prev.map(|k| g.din[k].p);
            let x = &mut g.srcs[i];
            let mut have = x.dheld as usize * 2;
            assert!(have / 2 + if x.ended { 0 } else { block_frames } <= cap_frames, "GpuMixer: held frames exceed the plan");
            if have > 0 { ck(unsafe { rh_memcpy_d2d(row.cast(), prev.unwrap().add(i * row_len + x.dheld_off as usize * 2).cast(), have * 4, copy_stream) }, "rh_memcpy_d2d"); }
            if !x.ended {
                let want = block_frames * 2;
                let mut got = match &mut x.up { Upstream::Chain(c) => c.read_device(unsafe { row.add(have) }, want, copy_stream), Upstream::Host(_) => unreachable!() };
                got -= got % 2;
                have += got;
                x.ended = got < want;
                x.note_pull(got, x.ended);
                if x.ended { if let Some(h) = &x.hint { let mut t = h.lock().unwrap(); t.total_known = true; t.total = t.pulled; } }
            }
            g.pptrs[i] = row as *const f32;
            g.pavail[i] = (have / 2) as u64;
            g.pended[i] = x.ended as u8;
        }
        ck(unsafe { rh_event_record(g.copied[slot].0, copy_stream) }, "rh_event_record");
    }
    fn issue_block_direct(&mut self, gi: usize) {
        let (cap_frames, stream, out_cap) = (self.cap_frames, self.pump.stream, self.out_cap_frames);
        let row_stereo = self.row;
        let g = &mut self.gens[gi];
        let native: usize = if g.mono { 1 } else { 2 };
        let row_len = if g.mono { (cap_frames + 3) & !3 } else { row_stereo };
        let s_n = g.srcs.len();
        let (slot, pd) = (g.pslot, g.pd);
        ck(unsafe { rh_stream_wait_event(stream, g.copied[slot].0) }, "rh_stream_wait_event");
        if g.pside > 0 {                                                            // ChannelCountConverter on the device (channels.rs:57-85), into the rows the fused launch reads
            for i in 0..s_n {
                if g.srcs[i].ch as usize != native && g.pavail[i] > 0 {
                    ck(unsafe { rh_channels_convert(g.din[pd].p.add(i * row_len), g.dside[slot].p.add(g.pside_off[i]), g.pavail[i] as usize, g.srcs[i].ch as u32, 2, stream) }, "rh_channels_convert");
                }
            }
        }
        let (mut out, mut consumed) = (0u64, 0u64);
        if g.mono {                                                                 // the mono mix of the block, then ChannelCountConverter(1 -> 2) (channels.rs:64-73) behind the stereo queue
            g.qm.reserve(out_cap as usize * 2);
            ck(unsafe { rh_rlm_stream_block_v(g.plan, g.pptrs.as_ptr(), g.pavail.as_ptr(), g.pended.as_ptr(), s_n as u32, g.qm.p, out_cap * 2 - g.fill - g.head, &mut out, &mut consumed, stream) },
               "rh_rlm_stream_block_v");
            if out > 0 { ck(unsafe { rh_channels_convert(g.queue_end(), g.qm.p, out as usize, 1, 2, stream) }, "rh_channels_convert"); }
        } else {
            ck(unsafe { rh_rlm_stream_block_v(g.plan, g.pptrs.as_ptr(), g.pavail.as_ptr(), g.pended.as_ptr(), s_n as u32, g.queue_end(), out_cap * 2 - g.fill - g.head, &mut out, &mut consumed, stream) },
               "rh_rlm_stream_block_v");
        }
        g.fill += out;
        let mut all_ended = true;
        let side_floats = g.pside;
        for i in 0..s_n {                                                           // keep what the converter has not consumed (a few hundred frames)
            let avail = g.pavail[i];
            if g.srcs[i].up.on_device() {                                          // ... which stays where it is: the next block copies it from this block's row
                let x = &mut g.srcs[i];
                x.dheld_off = consumed.min(avail);
                x.dheld = avail - x.dheld_off;
                all_ended = all_ended && x.ended;
                continue;
            }
            let ch = g.srcs[i].ch as usize;
            let row: &[f32] = if ch == native { &g.stage[slot].slice(s_n * row_len)[i * row_len..] } else { &g.side[slot].slice(side_floats)[g.pside_off[i]..] };
            let have = avail as usize * ch;
            let drop = (consumed as usize * ch).min(have);
            let x = &mut g.srcs[i];
            x.held.clear();
            x.held.extend_from_slice(&row[drop..have]);
            all_ended = all_ended && x.ended;
        }
        g.done = all_ended;                                                         // the call that saw every source ended emitted everything that was left
    }

    /// A span-by-span generation.  Every source is topped up to `target` converted frames: it is pulled piece by piece (a piece never
    /// crosses a span; its length is budgeted so that its output fits the row whatever the span does), the pieces are planned into
    /// segments; then ONE copy brings all rows to the device (copy stream), ONE launch converts all segments of all sources (plus the
    /// frames the last block left over, moved to the front of the other row set), and the fused kernel -- its converter passing
    /// through -- filters and mixes the rows.
    fn pull_block_staged(&mut self, gi: usize) {
        let (rate, copy_stream) = (self.rate, self.copy_stream);
        let g = &mut self.gens[gi];
        let s_n = g.srcs.len();
        let (slot, pd) = (g.pslot, g.pd);
        let crowf = (g.crow as usize * 2 + 3) & !3;
        // 1. layout of the staging block: a row per live source, sized for what it is about to pull in its current format
        let (mut row_off, mut row_cap) = (vec![0usize; s_n], vec![0usize; s_n]);
        let mut total = 0usize;
        for i in 0..s_n {
            row_off[i] = total;
            let x = &mut g.srcs[i];
            if x.ended { continue; }
            match x.reader.peek(x.up.src()) {
                None => { x.ended = true; }                                         // the chain rodio would build now is empty
                Some((ch, r)) => {
                    let want = g.target.saturating_sub(x.have_s / 2);
                    let in_frames = want * r as u64 / rate as u64 + 8;
                    row_cap[i] = x.plan.held_samples() + (in_frames as usize + 1) * ch as usize;   // (+ a frame: read_piece brings a cut frame's samples along with the last whole ones)
                    total += (row_cap[i] + 3) & !3;
                }
            }
        }
        g.stage[slot].reserve(total.max(4));
        g.din[pd].reserve(total.max(4));
        g.ptable.clear();
        g.pmax_out = 0;
        let (oc, nc) = (g.ccur, g.ccur ^ 1);
        for (i, x) in g.srcs.iter().enumerate() {                                  // what the last block left over: to the front of the other row set
            if x.have_s == 0 { continue; }
            // (copied as a MONO stream, sample for sample: the count may be odd, and the sample behind the last one belongs to a segment of this very launch)
            g.ptable.push(RhUniformSeg { src: unsafe { g.conv[oc].p.add(i * crowf + x.off_s as usize) }, dst: unsafe { g.conv[nc].p.add(i * crowf) }, src_frame0: 0, src_frames: x.have_s,
                                         m0: 0, m1: x.have_s, span_frames: u64::MAX, from_rate: rate, to_rate: rate, from_ch: 1, to_ch: 1, gain: 1.0, reserved: 0 });
            g.pmax_out = g.pmax_out.max(x.have_s);
        }
        // 2. pull and plan (the spans of a source are pulled in order by one thread; the sources one after the other here -- the C++ twin
        // deals them over its pull threads the same way the direct generations do)
        let stage_all = g.stage[slot].slice_mut(total.max(4));
        for i in 0..s_n {
            let x = &mut g.srcs[i];
            if x.ended { continue; }
            let row = &mut stage_all[row_off[i]..row_off[i] + row_cap[i]];
            x.plan.begin_block();
            let mut fill = x.plan.held_samples();
            row[..fill].copy_from_slice(&x.held[..fill]);
            let mut segs: Vec<PlannedSeg> = Vec::new();
            loop {
                let (ch, r) = match x.reader.peek(x.up.src()) { Some(f) => f, None => { x.ended = true; break; } };
                let now = (x.have_s + x.plan.out_samples() as u64 + 1) / 2;
                if now >= g.target { break; }
                let slack = UniformPlanner::close_slack_frames(r, rate);                // what the span's end may add
                let (need, most) = x.plan.budget(r, x.reader.opens_next(), g.target - now, (g.crow - now).saturating_sub(slack));
                // (whole frames that fit the row AND leave room for the up to ch - 1 samples of a frame the span's end cuts)
                let room = if row_cap[i] - fill >= ch as usize { (row_cap[i] - fill - (ch as usize - 1)) / ch as usize } else { 0 };
                let n = need.min(most).min(room as u64) as usize;
                if n == 0 { break; }
                x.note_hint();
                let piece = x.reader.read_piece(x.up.src(), &mut row[fill..], n);   // straight into the staging block
                if let Some(pc) = piece {
                    fill += pc.n;
                    x.plan.add(&pc, &mut segs);
                    if let Some(h) = &x.hint { let mut t = h.lock().unwrap(); let at = t.pulled; if let Some(c) = t.counter.as_mut() { c.feed(&pc, at); } t.pulled += pc.n as u64; }
                }
                if x.reader.ended() {
                    x.ended = true;
                    if let Some(h) = &x.hint { if let Some(c) = h.lock().unwrap().counter.as_mut() { c.input_ended(); } }
                    break;
                }
                if piece.is_none() { break; }
            }
            x.plan.end_block();
            x.held.clear();
            x.held.extend_from_slice(&row[x.plan.keep_offset()..x.plan.keep_offset() + x.plan.keep_samples()]);
            for sg in &segs {
                let mut t = sg.g;
                t.src = unsafe { g.din[pd].p.add(row_off[i] + sg.src_off) };
                t.dst = unsafe { g.conv[nc].p.add(i * crowf + x.have_s as usize + sg.dst_off) };   // (sample offsets)
                t.gain = x.gain;
                g.pmax_out = g.pmax_out.max(t.m1 - t.m0);
                g.ptable.push(t);
            }
            x.have_s += x.plan.out_samples() as u64;
            x.total_s += x.plan.out_samples() as u64;
            assert!(x.have_s + 1 <= g.crow * 2, "GpuMixer: converted frames exceed the row");
        }
        // 3. one copy for all rows, on the copy stream: it runs beside the launches of the block before
        if total > 0 { ck(unsafe { rh_memcpy_h2d(g.din[pd].p.cast(), g.stage[slot].p.cast(), total * 4, copy_stream) }, "rh_memcpy_h2d"); }
        ck(unsafe { rh_event_record(g.copied[slot].0, copy_stream) }, "rh_event_record");
    }
    fn issue_block_staged(&mut self, gi: usize) {
        let (stream, out_cap) = (self.pump.stream, self.out_cap_frames);
        let g = &mut self.gens[gi];
        let s_n = g.srcs.len();
        let slot = g.pslot;
        let crowf = (g.crow as usize * 2 + 3) & !3;
        let nc = g.ccur ^ 1;
        ck(unsafe { rh_stream_wait_event(stream, g.copied[slot].0) }, "rh_stream_wait_event");
        if !g.ptable.is_empty() {                                                   // ... one conversion launch behind the copy
            let tf = g.ptable.len() * std::mem::size_of::<RhUniformSeg>() / 4;
            g.tab[slot].reserve(tf);
            g.dtab.reserve(tf);
            unsafe {
                ptr::copy_nonoverlapping(g.ptable.as_ptr() as *const f32, g.tab[slot].p, tf);
                ck(rh_memcpy_h2d(g.dtab.p.cast(), g.tab[slot].p.cast(), tf * 4, stream), "rh_memcpy_h2d");
                ck(rh_uniform_segments_dev(g.dtab.p as *const RhUniformSeg, g.ptable.len() as u32, g.pmax_out, stream), "rh_uniform_segments_dev");
            }
        }
        g.ccur = nc;
        // 4. filter + ordered sum of the converted rows
        let ptrs: Vec<*const f32> = (0..s_n).map(|i| unsafe { g.conv[nc].p.add(i * crowf) } as *const f32).collect();
        for (i, x) in g.srcs.iter_mut().enumerate() {
            if x.ended && x.have_s & 1 == 1 {
                // the source's stream ends inside a frame: rodio's mixer adds its last sample and, at the next one, finds the source gone
                // (mixer.rs:185-198).  Adding +0.0 for the missing sample leaves the sum as it is -- through a filter it would not.
                assert!(g.filt.kind < 0, "GpuMixer: a filtered source of 1, 2, 4 or 8 channels whose stream ends inside a frame (source/mod.rs:196-200)");
                ck(unsafe { rh_memset(g.conv[nc].p.add(i * crowf + x.have_s as usize).cast(), 0, 4, stream) }, "rh_memset");
                x.have_s += 1;
            }
        }
        let avail: Vec<u64> = g.srcs.iter().map(|x| x.have_s / 2).collect();
        let ended: Vec<u8> = g.srcs.iter().map(|x| x.ended as u8).collect();
        let (mut out, mut consumed) = (0u64, 0u64);
        ck(unsafe { rh_rlm_stream_block_v(g.plan, ptrs.as_ptr(), avail.as_ptr(), ended.as_ptr(), s_n as u32, g.queue_end(), out_cap * 2 - g.fill - g.head, &mut out, &mut consumed, stream) },
           "rh_rlm_stream_block_v");
        g.fill += out;
        for x in &mut g.srcs { let d = consumed.min(x.have_s / 2); x.off_s = d * 2; x.have_s -= d * 2; }
        g.done = g.srcs.iter().all(|x| x.ended);
    }

    /// The mixed stereo block `mixed` (n frames, on the device) on its way to the host block of slot `i`, in the mixer's layout.
    fn send_block(&mut self, i: usize, mixed: *const f32, n: u64) {
        let stream = self.pump.stream;
        if self.out_ch == 2 {
            ck(unsafe { rh_memcpy_d2h_async(self.pump.slot[i].out.p.cast(), mixed.cast(), (n * 2 * 4) as usize, stream) }, "rh_memcpy_d2h_async");
            return;
        }
        let oc = self.out_ch as usize;
        self.dout.reserve(self.out_cap_frames as usize * 2 * oc);                 // ChannelCountConverter(2 -> channels) on the mix (channels.rs:57-85), once per block
        unsafe {
            ck(rh_channels_convert(self.dout.p, mixed, n as usize, 2, oc as u32, stream), "rh_channels_convert");
            ck(rh_memcpy_d2h_async(self.pump.slot[i].out.p.cast(), self.dout.p.cast(), n as usize * oc * 4, stream), "rh_memcpy_d2h_async");
        }
    }

    /// `Mixer::add` on a running mixer.  rodio admits the source at the next frame boundary of the output (mixer.rs:175-183).  Here up to
    /// two blocks are already mixed beyond that frame (the one being served, the one in flight), so the new source -- its own
    /// generation, its own fused stream and clock from frame J on -- is run ahead synchronously until it covers them, added onto their
    /// device copies at its offset (rh_mix_sum: old mix first, the newcomer last = insertion order) and the blocks travel to the host again.
    fn late_join(&mut self, mut item: Src) {
        let ci = self.pump.cur;
        let oc = self.out_ch as u64;
        let consumed = self.slot_base[ci] * oc + self.pump.pos as u64;              // samples already handed out
        let j = (consumed + oc - 1) / oc;                                            // the next frame boundary
        let flight = self.pump.other_in_flight();
        let li = if flight { ci ^ 1 } else { ci };
        let sched_end = self.slot_base[li] + self.slot_frames[li];
        let stream = self.pump.stream;
        ck(unsafe { rh_stream_synchronize(stream) }, "rh_stream_synchronize");     // the blocks about to be patched have been produced
        let staged = !item.up.on_device() && item.up.src_ref().current_span_len().is_some();
        if !staged { self.make_direct(&mut item); }
        let mono = !staged && item.ch == 1;
        self.start_stream(vec![item], staged, mono);
        self.last_join = j;
        let gi = self.gens.len() - 1;
        for x in &self.gens[gi].srcs { if let Some(h) = &x.hint { h.lock().unwrap().join_frame = j; } }
        let need = sched_end.saturating_sub(j);
        while self.gens[gi].fill < need && !self.gens[gi].done {
            self.run_block(gi);
            ck(unsafe { rh_stream_synchronize(stream) }, "rh_stream_synchronize");
        }
        for k in 0..(if flight { 2 } else { 1 }) {
            let si = if k == 0 { ci } else { ci ^ 1 };
            let (b0, b1) = (self.slot_base[si], self.slot_base[si] + self.slot_frames[si]);
            let (lo, hi) = { let g = &self.gens[gi]; (j.max(b0), b1.min(j + g.fill)) };
            if hi <= lo { continue; }
            let ptrs = [self.dkeep[si].p as *const f32, unsafe { self.gens[gi].queue().add(((lo - j) * 2) as usize) }];
            let (start, len) = ([0u64, (lo - b0) * 2], [self.slot_frames[si] * 2, (hi - lo) * 2]);
            self.dmix.reserve(self.out_cap_frames as usize * 4);
            unsafe {
                ck(rh_mix_sum(self.dmix.p, (self.slot_frames[si] * 2) as usize, ptrs.as_ptr(), start.as_ptr(), len.as_ptr(), 2, stream), "rh_mix_sum");
                ck(rh_memcpy_d2d(self.dkeep[si].p.cast(), self.dmix.p.cast(), (self.slot_frames[si] * 2 * 4) as usize, stream), "rh_memcpy_d2d");
            }
            let (keep, frames) = (self.dkeep[si].p as *const f32, self.slot_frames[si]);
            self.send_block(si, keep, frames);
        }
        {   // the newcomer's queue moves on to the frame the next block starts at
            let g = &mut self.gens[gi];
            let used = g.fill.min(need);
            let (rem, pad) = (g.fill - used, (g.fill - used) & 1);
            if rem > 0 { ck(unsafe { rh_memcpy_d2d(g.q[g.cur ^ 1].p.add(pad as usize * 2).cast(), g.queue().add((used * 2) as usize).cast(), (rem * 2 * 4) as usize, stream) }, "rh_memcpy_d2d"); }
            g.cur ^= 1; g.head = pad; g.fill = rem;
        }
        ck(unsafe { rh_stream_synchronize(stream) }, "rh_stream_synchronize");
        self.block_done();
        // a mixer that was about to end goes on: the block that carried the end mark loses it, and if nothing was in flight the next block is requested now
        let last_i = if flight { ci ^ 1 } else { ci };
        let (gdone, gfill) = (self.gens[gi].done, self.gens[gi].fill);
        if self.pump.slot[last_i].last && !(gdone && gfill == 0) {
            self.pump.slot[last_i].last = false;
            if !flight { self.submit(ci ^ 1); }
        }
    }
}

/// A device chain pulled through the host after all (a steep rate ratio in front of it): `Source` by delegation.
struct ChainAsSource(Box<dyn DeviceChain>);
impl Iterator for ChainAsSource { type Item = f32; fn next(&mut self) -> Option<f32> { self.0.as_source().next() } }
impl Source for ChainAsSource {
    fn current_span_len(&self) -> Option<usize> { self.0.as_source_ref().current_span_len() }
    fn channels(&self) -> ChannelCount { self.0.as_source_ref().channels() }
    fn sample_rate(&self) -> SampleRate { self.0.as_source_ref().sample_rate() }
    fn total_duration(&self) -> Option<Duration> { self.0.as_source_ref().total_duration() }
    fn try_seek(&mut self, pos: Duration) -> Result<(), SeekError> { self.0.as_source().try_seek(pos) }
}

impl BlockSource for GpuMixer {
    fn pump(&mut self) -> &mut Pump { &mut self.pump }
    fn can_resume(&self) -> bool { !self.pending.is_empty() && self.resume_ok }     // mixer.rs:117-136: None while empty, samples again after add()
    fn block_done(&mut self) {                                                    // a bounded wait inside the fused kernel expired (never seen on a healthy device): fail loudly
        for g in &self.gens { if !g.plan.is_null() { ck(unsafe { rh_rlm_last_status(g.plan) }, "rh_rlm_last_status"); } }
        if self.device_chains { ck(unsafe { rh_async_status() }, "rh_async_status"); }   // ... or inside a scan kernel of a chain that hands its blocks over on the device
    }
    fn prefetch(&mut self) {
        if !self.pending.is_empty() { return; }                                     // a new generation starts with the next block: enqueue() does it all
        for gi in 0..self.gens.len() {
            let g = &self.gens[gi];
            if !g.done && g.fill < self.out_cap_frames && !g.pulled { self.pull_block(gi); }
        }
    }
    fn enqueue(&mut self, i: usize) {
        if !self.pending.is_empty() { self.start_generation(); }
        if self.gens.is_empty() {                                                  // nothing to pull
            self.pump.slot[i].n = 0; self.pump.slot[i].last = true;
            self.slot_base[i] = self.scheduled; self.slot_frames[i] = 0;
            return;
        }
        let stream = self.pump.stream;
        let oc = self.out_ch as usize;
        self.pump.slot[i].out.reserve(self.out_cap_frames as usize * 2 * oc.max(2));
        // 1. every live generation converts, filters and mixes one block of its sources behind what its queue holds
        for gi in 0..self.gens.len() { if !self.gens[gi].done && self.gens[gi].fill < self.out_cap_frames { self.run_block(gi); } }
        // 2. the frames every unfinished generation has reached; finished ones give what they have left
        let live: Vec<u64> = self.gens.iter().filter(|g| !g.done).map(|g| g.fill).collect();
        let n = if live.is_empty() { self.gens.iter().map(|g| g.fill).max().unwrap_or(0) } else { *live.iter().min().unwrap() };
        // 3. sum the generations in insertion order (a single one is already the mix) and send the block to the host
        if n > 0 {
            let mut mixed = self.gens[0].queue();
            if self.gens.len() > 1 {
                let ptrs: Vec<*const f32> = self.gens.iter().map(|g| g.queue()).collect();
                let start = vec![0u64; ptrs.len()];
                let len: Vec<u64> = self.gens.iter().map(|g| g.fill.min(n) * 2).collect();
                self.dmix.reserve(self.out_cap_frames as usize * 4);
                ck(unsafe { rh_mix_sum(self.dmix.p, (n * 2) as usize, ptrs.as_ptr(), start.as_ptr(), len.as_ptr(), ptrs.len() as u32, stream) }, "rh_mix_sum");
                mixed = self.dmix.p;
            }
            self.send_block(i, mixed, n);
            self.dkeep[i].reserve(self.out_cap_frames as usize * 4);              // the block also stays on the device until it has been served (late_join)
            ck(unsafe { rh_memcpy_d2d(self.dkeep[i].p.cast(), mixed.cast(), (n * 2 * 4) as usize, stream) }, "rh_memcpy_d2d");
        }
        self.slot_base[i] = self.scheduled;
        self.slot_frames[i] = n;
        // 4. what a generation produced beyond n waits at the front of its (other) queue buffer
        for g in &mut self.gens {
            let used = g.fill.min(n);
            let (rem, pad) = (g.fill - used, (g.fill - used) & 1);                   // the fused kernel writes 16-byte aligned blocks behind it
            if rem > 0 { ck(unsafe { rh_memcpy_d2d(g.q[g.cur ^ 1].p.add(pad as usize * 2).cast(), g.queue().add((used * 2) as usize).cast(), (rem * 2 * 4) as usize, stream) }, "rh_memcpy_d2d"); }
            g.cur ^= 1; g.head = pad; g.fill = rem;
        }
        self.scheduled += n;
        if self.gens.iter().all(|g| g.done && g.fill == 0) {
            unsafe {
                ck(rh_stream_synchronize(stream), "rh_stream_synchronize");
                ck(rh_stream_synchronize(self.copy_stream), "rh_stream_synchronize");
            }
            for g in &self.gens { if !g.plan.is_null() { ck(unsafe { rh_rlm_last_status(g.plan) }, "rh_rlm_last_status"); } }   // the last blocks too: nothing is served unchecked
            // The generations' plans, page-locked blocks and device rows are NOT freed here: this runs inside the consumer's next() (the
            // audio callback), and freeing a few hundred MB of page-locked memory takes hundreds of milliseconds.  The reaper does it.
            let mut st = self.retired_chains;
            for g in &self.gens { for x in &g.srcs { count_chain(x, &mut st); } }
            self.retired_chains = st;
            let dead: Vec<Gen> = self.gens.drain(..).collect();
            self.reaper.get_or_insert_with(Reaper::new).retire(dead);
        }
        self.pump.slot[i].n = n as usize * oc;
        self.pump.slot[i].last = self.gens.is_empty() && self.pending.is_empty();
    }
}
impl Iterator for GpuMixer {
    type Item = f32;
    // MixerSource::next advances its channel position on every call, also on the ones that return None (mixer.rs:120-136), and admits
    // pending sources only at channel 0: an ended mixer that gets a new source in the middle of a frame returns None until the frame is over.
    fn next(&mut self) -> Option<f32> {
        self.resume_ok = self.calls % self.out_ch as u64 == 0;
        self.calls += 1;
        self.next_sample()
    }
    /// mixer.rs:139-166: (0, Some(0)) while no source plays -- sources that wait for the next frame do not count --; otherwise the largest lower
    /// bound among the sources that play where the CONSUMER stands, and no upper bound (every source sits in a UniformSourceIterator,
    /// mixer.rs:58-66, whose upper bound is None: uniform.rs:100-108).  A source plays from the call that admitted it to the call in which it
    /// returns None (mixer.rs:185-198).
    fn size_hint(&self) -> (usize, Option<usize>) {
        let oc = self.out_ch as u64;
        let consumed = if self.pump.primed { self.slot_base[self.pump.cur] * oc + self.pump.pos as u64 } else { 0 };
        let mut hints = self.hints.lock().unwrap();
        let (mut any, mut lower) = (false, 0usize);
        hints.retain(|t| {
            let mut t = t.lock().unwrap();
            let first = t.join_frame * oc;
            if consumed <= first { return true; }                                   // admitted by a call still to come
            match t.lower_at(consumed - first, self.out_ch) {
                None => false,                                                      // it has returned None in front of the consumer: gone for good
                Some(lo) => { any = true; lower = lower.max(lo); true }
            }
        });
        if any { (lower, None) } else { (0, Some(0)) }
    }
}
impl Source for GpuMixer {
    fn current_span_len(&self) -> Option<usize> { None }
    fn channels(&self) -> ChannelCount { ChannelCount::new(self.out_ch).unwrap() }
    fn sample_rate(&self) -> SampleRate { SampleRate::new(self.rate).unwrap() }
    fn total_duration(&self) -> Option<Duration> { None }
    fn try_seek(&mut self, _: Duration) -> Result<(), SeekError> { Err(SeekError::NotSupported { underlying_source: "rodio_hip::GpuMixer (like MixerSource, mixer.rs:160-170)" }) }
}
impl Drop for GpuMixer {
    fn drop(&mut self) {
        unsafe {
            rh_stream_synchronize(self.copy_stream);
            rh_stream_synchronize(self.pump.stream);
        }
        self.gens.clear();
        self.reaper.take();   // joins: generations that were retired are gone before the streams are
        unsafe { rh_stream_destroy(self.copy_stream); }
    }
}
