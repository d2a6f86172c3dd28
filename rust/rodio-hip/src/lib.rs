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
//! * [`GpuMixer`] -- `mixer::mixer(2, rate)` where every added source goes through `UniformSourceIterator::new(src.amplify(g), 2,
//!   rate)` (`Mixer::add` wraps every source in one, `mixer.rs:58-66`) `[.low_pass(f)]` and the ordered sum: the fused kernel,
//!   block by block.  Sources may be added while the mixer is playing (they join at the next frame, `mixer.rs:175-183`).
//!
//! Both pull their upstream the way `UniformSourceIterator` does (`src/source/uniform.rs:50-97`): `current_span_len()` is asked
//! whenever the converter chain has run dry, `min(span, 32768)` samples go to a FRESH converter pair, and every span ends with
//! its last frame verbatim -- a `SamplesBuffer` or a decoder comes out span by span, a generator as one continuous stream.
//!
//! There is no CPU compute path: every arithmetic operation on samples happens in the library.  One object is used from one
//! thread at a time (`Send`, not `Sync`, like every rodio source).
//!
//! This is the Rust twin of `include/rodio_hip.hpp` (C++17, compiled and tested in the repository: `tests/test_host_mirror.py`);
//! the image the library was developed in has no Rust toolchain, so this crate has been checked against the header
//! (`tests/test_rust_decls.py`) but not against `rustc`.

pub mod ffi;

use ffi::*;
use rodio::source::SeekError;
use rodio::{ChannelCount, SampleRate, Source};
use std::ptr;
use std::time::Duration;

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
pub struct Piece { pub n: usize, pub opens: bool, pub closes: bool, pub ch: u16, pub rate: u32 }

/// Pulls a source the way `UniformSourceIterator` does (`uniform.rs:50-97`): whenever its converter chain has run dry it asks
/// `current_span_len()`, `channels()` and `sample_rate()` -- in that order, at exactly that position of the stream -- and admits
/// `min(span, 32768)` samples (`Take`, `uniform.rs:56,148-178`) to the chain it builds for them.  A span ends when that many
/// samples were taken or when the source returns `None`; the stream ends when a fresh chain yields nothing.
#[derive(Default)]
pub struct SpanReader { open: bool, fresh: bool, ended: bool, left: usize, ch: u16, rate: u32 }
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
        // source/mod.rs:196-200 asks for spans of whole frames; `.min(32768)` breaks that for 3, 5, 6, 7 ... channels, and rodio
        // then rotates the channels of every later span.  That is not reproduced: it is refused.
        assert!(self.left == OPEN_ENDED || self.left % self.ch as usize == 0, "a span of {} samples cuts a frame of {} channels", self.left, self.ch);
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
        let mut got = if want > 0 { read_into(up, &mut dst[..want]) } else { 0 };
        let none = got < want;                                                    // the source returned None inside the span
        got -= got % ch;                                                          // sources end on frame boundaries (source/mod.rs:169-178)
        if self.left != OPEN_ENDED { self.left -= got; }
        let closes = none || self.left == 0;
        let piece = Piece { n: got, opens: self.fresh, closes, ch: self.ch, rate: self.rate };
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
    row_off: usize, pos: usize, held: usize, keep_off: usize, keep_n: usize, out: u64,
}
impl UniformPlanner {
    pub fn new(to_ch: u16, to_rate: u32) -> Self {
        UniformPlanner { to_ch, to_rate, span_in: 0, span_m: 0, row_frame0: 0, next_frame0: 0, row_off: 0, pos: 0, held: 0, keep_off: 0, keep_n: 0, out: 0 }
    }
    pub fn begin_block(&mut self) { self.pos = self.held; self.row_off = 0; self.out = 0; self.keep_off = 0; self.keep_n = self.held; }
    pub fn held_samples(&self) -> usize { self.held }
    pub fn out_frames(&self) -> u64 { self.out }
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
        self.span_in += (p.n / p.ch as usize) as u64;
        self.pos += p.n;
        let ready = self.span_frames(self.span_in, p.rate, p.closes);
        if ready > self.span_m {
            let g = RhUniformSeg {
                src: ptr::null(), dst: ptr::null_mut(), src_frame0: self.row_frame0, src_frames: self.span_in - self.row_frame0,
                m0: self.span_m, m1: ready, span_frames: if p.closes { self.span_in } else { u64::MAX },
                from_rate: p.rate, to_rate: self.to_rate, from_ch: p.ch as u32, to_ch: self.to_ch as u32, gain: 1.0, reserved: 0,
            };
            segs.push(PlannedSeg { src_off: self.row_off, dst_off: self.out as usize, g });
            self.out += ready - self.span_m;
            self.span_m = ready;
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
struct Slot { stage: PinnedBuf, out: PinnedBuf, n: usize, last: bool, done: *mut core::ffi::c_void }
/// Two page-locked result blocks, one served while the other is in flight.
struct Pump { slot: [Slot; 2], cur: usize, pos: usize, skip: usize, handed_out: u64, primed: bool, ended: bool, stream: RhStream }
impl Pump {
    fn new() -> Self {
        let mut stream: RhStream = ptr::null_mut();
        ck(unsafe { rh_stream_create(&mut stream) }, "rh_stream_create");
        let mk = || {
            let mut ev = ptr::null_mut();
            ck(unsafe { rh_event_create(&mut ev) }, "rh_event_create");
            Slot { stage: PinnedBuf::new(), out: PinnedBuf::new(), n: 0, last: false, done: ev }
        };
        Pump { slot: [mk(), mk()], cur: 0, pos: 0, skip: 0, handed_out: 0, primed: false, ended: false, stream }
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
            for s in &self.slot { rh_event_destroy(s.done); }
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
    fn submit(&mut self, i: usize) {
        self.enqueue(i);
        let p = self.pump();
        ck(unsafe { rh_event_record(p.slot[i].done, p.stream) }, "rh_event_record");
    }
    fn advance(&mut self) -> bool {
        if self.pump().ended {
            if !self.can_resume() { return false; }
            let p = self.pump(); p.ended = false; p.primed = false;
        }
        if !self.pump().primed {
            self.submit(0);
            let p = self.pump(); p.primed = true; p.cur = 0;
        } else {
            let p = self.pump();
            if p.slot[p.cur].last { p.ended = true; return false; }
            p.cur ^= 1;                                                            // the block that was enqueued while the previous one was being served
        }
        let cur = self.pump().cur;
        ck(unsafe { rh_event_synchronize(self.pump().slot[cur].done) }, "rh_event_synchronize");
        self.block_done();
        let p = self.pump();
        p.pos = p.skip.min(p.slot[cur].n);
        p.skip = 0;
        if !p.slot[cur].last { self.submit(cur ^ 1); }                            // prefetch: pull and process one block ahead
        true
    }
    fn next_sample(&mut self) -> Option<f32> {
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

/// `upstream.amplify(..).low_pass(..)...` with the chain executed block-wise on the GPU.
pub struct GpuSource<I: Source> {
    up: I, block_frames: usize, ch: u16, rate: u32, in_ch: u16, in_rate: u32,
    stages: Vec<Stage>, reader: SpanReader, pieces: Vec<Piece>, span_aware: bool, exact_filters: bool, scan_kernels: bool,
    a: DeviceBuf, b: DeviceBuf, pump: Pump,
}
unsafe impl<I: Source + Send> Send for GpuSource<I> {}

impl<I: Source> GpuSource<I> {
    pub fn new(upstream: I, block_frames: usize) -> Self {
        let (ch, rate) = (upstream.channels().get(), upstream.sample_rate().get());
        GpuSource { up: upstream, block_frames: block_frames.max(1), ch, rate, in_ch: ch, in_rate: rate, stages: Vec::new(), reader: SpanReader::new(),
                    pieces: Vec::new(), span_aware: false, exact_filters: false, scan_kernels: false, a: DeviceBuf::new(), b: DeviceBuf::new(), pump: Pump::new() }
    }
    pub fn inner(&self) -> &I { &self.up }
    pub fn inner_mut(&mut self) -> &mut I { &mut self.up }
    pub fn into_inner(self) -> I where I: Clone { self.up.clone() }
    fn push(&mut self, run: impl FnMut(&mut Ctx) -> usize + Send + 'static, bound: impl Fn(usize, usize) -> usize + Send + 'static, span_rule: u8) -> &mut Stage {
        self.stages.push(Stage { run: Box::new(run), bound: Box::new(bound), seekable: true, on_seek: None, span_rule });
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
    /// The filters run time-parallel (rh_biquad mode 1: <= 1e-5 from rodio's f32 recurrence); `exact_filters(true)`: the reference's
    /// operation order throughout, bit for bit.
    pub fn exact_filters(mut self, on: bool) -> Self { self.exact_filters = on; self }
    pub fn low_pass(self, freq: u32) -> Self { self.blt(0, freq, 0.5) }          // blt.rs:11-16
    pub fn high_pass(self, freq: u32) -> Self { self.blt(1, freq, 0.5) }         // blt.rs:18-24
    pub fn low_pass_with_q(self, freq: u32, q: f32) -> Self { self.blt(0, freq, q) }
    pub fn high_pass_with_q(self, freq: u32, q: f32) -> Self { self.blt(1, freq, q) }
    fn blt(mut self, kind: i32, freq: u32, q: f32) -> Self {                      // blt.rs:502-544,558-560
        let ch = self.ch as u32;
        let mut co = [0f32; 5];
        ck(unsafe { rh_biquad_coeffs(kind, freq, q, self.rate, co.as_mut_ptr()) }, "rh_biquad_coeffs");
        let st = self.state(4 * ch as usize);
        let (st2, sm, mode) = (st.clone(), self.pump.stream as usize, if self.exact_filters { 0 } else { 1 });
        let stage = self.push(move |c| {
            let frames = c.n / ch as usize;
            ck(unsafe { rh_biquad(c.out, c.inp, frames as u64, ch, 1, co.as_ptr(), st.0.p, mode, c.stream) }, "rh_biquad");
            frames * ch as usize
        }, |n, _| n, 0);
        stage.on_seek = Some(Box::new(move |_| ck(unsafe { rh_memset(st2.0.p.cast(), 0, 4 * ch as usize * 4, sm as RhStream) }, "rh_memset")));   // blt.rs:350-377
        self
    }
    pub fn reverb(mut self, duration: Duration, amplitude: f32) -> Self {         // source/mod.rs:628-634
        let d = unsafe { rh_delay_samples(duration.as_nanos() as u64, self.rate, self.ch as u32) };
        let mut e: *mut RhEcho = ptr::null_mut();
        ck(unsafe { rh_echo_create(&mut e, d, amplitude) }, "rh_echo_create");
        let h = Handle { p: e, destroy: rh_echo_destroy };
        let stage = self.push(move |c| {
            if c.n > 0 { ck(unsafe { rh_echo_process(h.p, c.out, c.inp, c.n as u64, c.stream) }, "rh_echo_process"); }
            if !c.flush { return c.n; }
            if d > 0 { ck(unsafe { rh_echo_flush(h.p, c.out.add(c.n), c.stream) }, "rh_echo_flush"); }                 // the delayed clone outlives the source
            c.n + d as usize
        }, move |n, _| n + d as usize, 1);                                           // Mix::current_span_len() is None (mix.rs:92-94)
        stage.seekable = false;                                                        // mix.rs:116-120
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
        self.ch = out_ch as u16;
        self
    }
    pub fn convert_channels(mut self, to: ChannelCount) -> Self {                 // ChannelCountConverter, channels.rs:57-85
        let (from, to) = (self.ch as usize, to.get() as usize);
        self.push(move |c| {
            let frames = c.n / from;
            ck(unsafe { rh_channels_convert(c.out, c.inp, frames, from as u32, to as u32, c.stream) }, "rh_channels_convert");
            frames * to
        }, move |n, _| n / from * to, 1);
        self.ch = to as u16;
        self
    }
    pub fn convert_sample_rate(mut self, to: SampleRate) -> Self {                // SampleRateConverter, sample_rate.rs:52-201
        let (from, to, ch) = (self.rate, to.get(), self.ch as usize);
        if from == to { return self; }                                              // sample_rate.rs:133-136
        let mut r: *mut RhResampler = ptr::null_mut();
        ck(unsafe { rh_resampler_create(&mut r, from, to, ch as u32) }, "rh_resampler_create");
        let h = Handle { p: r, destroy: rh_resampler_destroy };
        self.push(move |c| {
            let mut m = 0u64;
            ck(unsafe { rh_resampler_process(h.p, c.out, (c.out_cap / ch) as u64, c.inp, (c.n / ch) as u64, c.flush as i32, &mut m, c.stream) }, "rh_resampler_process");
            m as usize * ch
        }, move |n, _| (((n / ch + 2) as u64 * to as u64 / from as u64) as usize + 2) * ch, 1);
        self.rate = to;
        self
    }
    /// `UniformSourceIterator::new(src, channels, rate)` (uniform.rs:50-97), span by span: see the crate documentation.
    pub fn uniform(mut self, channels: ChannelCount, sample_rate: SampleRate) -> Self {
        let rule = self.stages.iter().rev().map(|s| s.span_rule).find(|&r| r != 0).unwrap_or(0);
        assert!(!(rule == 2 && self.up.current_span_len().is_some()), "GpuSource::uniform behind take_duration / delay / channel_volume on a source that reports spans");
        if rule != 0 {                                                              // continuous from here on
            let from_ch = self.ch;
            let mut s = self.convert_sample_rate(sample_rate);
            if channels.get() != from_ch { s = s.convert_channels(channels); }
            if let Some(st) = s.stages.last_mut() { st.span_rule = 1; }
            return s;
        }
        self.span_aware = true;
        let (to_ch, to_rate, in_ch, from) = (channels.get(), sample_rate.get(), self.ch as usize, self.rate);
        let plan = std::sync::Arc::new(std::sync::Mutex::new(UniformPlanner::new(to_ch, to_rate)));
        let plan2 = plan.clone();
        let (mut win, mut keep) = (State(DeviceBuf::new()), State(DeviceBuf::new()));
        let stage = self.push(move |c| {
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
            let table: Vec<RhUniformSeg> = segs.iter().map(|s| {
                let mut g = s.g;
                g.src = unsafe { win.0.p.add(s.src_off) };
                g.dst = unsafe { c.out.add(s.dst_off * to_ch as usize) };
                g
            }).collect();
            assert!((plan.out_frames() as usize + 1) * to_ch as usize <= c.out_cap, "GpuSource::uniform: block capacity");
            ck(unsafe { rh_uniform_segments(table.as_ptr(), table.len() as u32, c.stream) }, "rh_uniform_segments");
            let kn = plan.keep_samples();
            if kn > 0 {
                keep.0.reserve(kn);
                ck(unsafe { rh_memcpy_d2d(keep.0.p.cast(), win.0.p.add(plan.keep_offset()).cast(), kn * 4, c.stream) }, "rh_memcpy_d2d");
            }
            plan.out_frames() as usize * to_ch as usize
        }, move |n, pieces| {                                                        // every span may add its verbatim last frame
            let f = (n / in_ch) as u64;
            (f.max(f * to_rate as u64 / from as u64 + 2) as usize + 2 * (pieces + 2)) * to_ch as usize
        }, 1);
        stage.on_seek = Some(Box::new(move |_| *plan2.lock().unwrap() = UniformPlanner::new(to_ch, to_rate)));   // the next span starts a fresh chain
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
        self.push(move |c| { ck(unsafe { rh_agc(c.out, c.inp, c.n as u64, rate, 1, &settings, st.0.p, c.stream) }, "rh_agc"); c.n }, |n, _| n, 0);
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
    /// `take_duration(d)`, with `set_filter_fadeout()` when `fade_out` (take.rs:96-148).
    pub fn take_duration(mut self, duration: Duration, fade_out: bool) -> Self {
        let (ch, rate, ns) = (self.ch as u32, self.rate, duration.as_nanos() as u64);
        let (mut pos, mut done) = (0u64, false);
        self.push(move |c| {
            c.end = true;
            if done { return 0; }
            let (mut m, mut ended) = (0u64, 0i32);
            ck(unsafe { rh_take_duration(c.out, c.inp, c.n as u64, pos, ch, rate, ns, fade_out as i32, &mut m, &mut ended, c.stream) }, "rh_take_duration");
            pos += c.n as u64;
            done = ended != 0;
            c.end = done;
            m as usize
        }, move |n, _| n + ch as usize, 2);
        self
    }
    /// `delay(d)` (delay.rs:8-16,68-75): `rh_delay_samples()` zeros in front of the stream.  Not seekable here.
    pub fn delay(mut self, duration: Duration) -> Self {
        let d = unsafe { rh_delay_samples(duration.as_nanos() as u64, self.rate, self.ch as u32) };
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
        self.pump.slot[i].stage.reserve(want);
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
        // capacity of the ping-pong buffers: the largest block any stage can emit
        let (mut cap, mut m) = (want, want);
        for st in &self.stages { m = (st.bound)(m, self.pieces.len()); cap = cap.max(m); }
        cap = ((cap + 3) & !3) + 64;
        self.a.reserve(cap);
        self.b.reserve(cap);
        self.pump.slot[i].out.reserve(cap);
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
        if n > 0 { ck(unsafe { rh_memcpy_d2h_async(self.pump.slot[i].out.p.cast(), cur.cast(), n * 4, stream) }, "rh_memcpy_d2h_async"); }
        self.pump.slot[i].n = n;
        self.pump.slot[i].last = ends;
    }
}
impl<I: Source> Iterator for GpuSource<I> {
    type Item = f32;
    fn next(&mut self) -> Option<f32> { self.next_sample() }
}
impl<I: Source> Source for GpuSource<I> {
    fn current_span_len(&self) -> Option<usize> { None }
    fn channels(&self) -> ChannelCount { ChannelCount::new(self.ch).unwrap() }
    fn sample_rate(&self) -> SampleRate { SampleRate::new(self.rate).unwrap() }
    fn total_duration(&self) -> Option<Duration> { None }
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
        Ok(())
    }
}

// ------------------------------------------------------------------------------------------------ GpuMixer ----
#[derive(Clone, Copy)]
pub struct MixerOptions {
    pub block_frames: usize,         // frames pulled per source and block
    pub filter_kind: i32,            // -1 none, 0 low_pass, 1 high_pass (q = 0.5, blt.rs:11-24)
    pub filter_freq: u32,
    pub filter_q: f32,
    pub frames_per_lane: u32,        // 0 = the library's choice
}
impl Default for MixerOptions { fn default() -> Self { MixerOptions { block_frames: 1 << 15, filter_kind: -1, filter_freq: 0, filter_q: 0.5, frames_per_lane: 0 } } }

struct Src {
    up: Box<dyn Source + Send>, gain: f32, held: Vec<f32>, ended: bool, ch: u16,
    reader: SpanReader, plan: UniformPlanner, have: u64, off: u64,               // span-by-span generations
}
/// Sources that joined together: one clock, one fused stream.
struct Gen {
    srcs: Vec<Src>, plan: *mut RhRlm, din: DeviceBuf, q: [DeviceBuf; 2], stage: [PinnedBuf; 2], side: [PinnedBuf; 2], dside: DeviceBuf,
    cur: usize, slot: usize, head: u64, fill: u64, done: bool,
    mono: bool, qm: DeviceBuf,                                                       // a fused stream of mono sources (channels = 1): the mono mix, made stereo once per block
    staged: bool, target: u64, crow: u64, conv: [DeviceBuf; 2], dtab: DeviceBuf, tab: [PinnedBuf; 2], ccur: usize,
}
impl Gen {
    fn queue(&self) -> *const f32 { unsafe { self.q[self.cur].p.add(self.head as usize * 2) } }
    fn queue_end(&self) -> *mut f32 { unsafe { self.q[self.cur].p.add((self.head + self.fill) as usize * 2) } }
}

/// What rodio spells `let (mixer, mixed) = mixer::mixer(nz!(2), rate); mixer.add(src.amplify(g)) ...` (every added source goes
/// through `UniformSourceIterator::new(.., 2, rate)`, optionally `.low_pass(f)` / `.high_pass(f)`) as ONE source.
pub struct GpuMixer {
    rate: u32, opt: MixerOptions, pending: Vec<Src>, gens: Vec<Gen>,
    cap_frames: usize, row: usize, out_cap_frames: u64, scheduled: u64, last_join: u64,
    dmix: DeviceBuf, dkeep: [DeviceBuf; 2], slot_base: [u64; 2], slot_frames: [u64; 2], calls: u64, resume_ok: bool,
    pump: Pump,
}
unsafe impl Send for GpuMixer {}

fn fused_ratio_unsupported(from: u32, to: u32) -> bool {                          // rh_rlm_create: reduced from/to <= 4.5, from*to within u32
    let (mut a, mut b) = (from as u64, to as u64);
    while b != 0 { let t = a % b; a = b; b = t; }
    let (f, t) = (from as u64 / a, to as u64 / a);
    2 * f > 9 * t || f * t > 0xffff_ffff
}

impl GpuMixer {
    pub fn new(sample_rate: SampleRate, opt: MixerOptions) -> Self {
        let mut opt = opt;
        opt.block_frames = opt.block_frames.max(1);
        GpuMixer { rate: sample_rate.get(), opt, pending: Vec::new(), gens: Vec::new(), cap_frames: 0, row: 0, out_cap_frames: 0, scheduled: 0, last_join: 0,
                   dmix: DeviceBuf::new(), dkeep: [DeviceBuf::new(), DeviceBuf::new()], slot_base: [0; 2], slot_frames: [0; 2], calls: 0, resume_ok: true, pump: Pump::new() }
    }
    /// `Mixer::add` (mixer.rs:58-66), with the source's volume (`mixer.add(src.amplify(gain))`).  May be called at any time.
    pub fn add(&mut self, src: Box<dyn Source + Send>, gain: f32) {
        let ch = src.channels().get();
        let item = Src { up: src, gain, held: Vec::new(), ended: false, ch, reader: SpanReader::new(), plan: UniformPlanner::new(2, self.rate), have: 0, off: 0 };
        if self.pump.running() { self.late_join(item); } else { self.pending.push(item); }
    }
    /// Output frame (of this mixer) at which the most recently started generation joined.
    pub fn last_join_frame(&self) -> u64 { self.last_join }

    fn make_direct(&self, x: &mut Src) {
        // A continuous source the fused kernel cannot take as it is (rate ratio above 4.5) gets the GPU converter adapter in front.
        if !fused_ratio_unsupported(x.up.sample_rate().get(), self.rate) { return; }
        let up = std::mem::replace(&mut x.up, Box::new(rodio::source::Empty::new()));
        let mut conv = GpuSource::new(up, self.opt.block_frames);
        if x.ch != 2 { conv = conv.convert_channels(ChannelCount::new(2).unwrap()); }
        conv = conv.convert_sample_rate(SampleRate::new(self.rate).unwrap());
        x.up = Box::new(conv);
        x.ch = 2;
    }
    fn start_generation(&mut self) {
        let mut all = std::mem::take(&mut self.pending);
        if all.iter().any(|x| x.up.current_span_len().is_some()) {               // span by span, as rodio converts them: one stream, insertion order
            self.start_stream(all, true, false);
            return;
        }
        for x in &mut all { self.make_direct(x); }
        // continuous sources: one fused stream per (input rate, mono or not), in order of first appearance; mono sources form
        // streams of mono frames (the kernel reads 4 bytes per frame, the mono mix becomes stereo once per block)
        let mut kinds: Vec<(u32, bool)> = Vec::new();
        for x in &all { let k = (x.up.sample_rate().get(), x.ch == 1); if !kinds.contains(&k) { kinds.push(k); } }
        for (r, mono) in kinds {
            let (group, rest): (Vec<Src>, Vec<Src>) = all.into_iter().partition(|x| x.up.sample_rate().get() == r && (x.ch == 1) == mono);
            all = rest;
            self.start_stream(group, false, mono);
        }
    }
    fn start_stream(&mut self, srcs: Vec<Src>, staged: bool, mono: bool) {
        let from = if staged { self.rate } else { srcs[0].up.sample_rate().get() };
        self.cap_frames = self.opt.block_frames + 4096;                            // a block can hold what the previous one left over
        let mut g = Gen { srcs, plan: ptr::null_mut(), din: DeviceBuf::new(), q: [DeviceBuf::new(), DeviceBuf::new()], stage: [PinnedBuf::new(), PinnedBuf::new()],
                          side: [PinnedBuf::new(), PinnedBuf::new()], dside: DeviceBuf::new(), cur: 0, slot: 0, head: 0, fill: 0, done: false,
                          mono: mono && !staged, qm: DeviceBuf::new(), staged, target: 0, crow: 0, conv: [DeviceBuf::new(), DeviceBuf::new()], dtab: DeviceBuf::new(), tab: [PinnedBuf::new(), PinnedBuf::new()], ccur: 0 };
        if staged {
            g.target = self.opt.block_frames as u64 + 64 * 20 + 8;                 // a block emits whole tiles (at most 64 * 20 frames) and keeps two frames of history
            g.crow = g.target + 64;
            let crowf = (g.crow as usize * 2 + 3) & !3;
            for b in &mut g.conv { b.reserve(g.srcs.len() * crowf); }
        }
        let cfg = RhRlmConfig {
            from_rate: from, to_rate: self.rate, channels: if g.mono { 1 } else { 2 }, span_len: 0, filter_kind: self.opt.filter_kind, filter_freq: self.opt.filter_freq, filter_q: self.opt.filter_q,
            max_sources: g.srcs.len() as u32, max_in_frames: if staged { g.crow } else { self.cap_frames as u64 },
            frames_per_lane: self.opt.frames_per_lane, ring_stages: 0, no_balance: 0, force_general: 0, custom_coeffs: [0.0; 5], filter_first: 0,
        };
        ck(unsafe { rh_rlm_create(&mut g.plan, &cfg) }, "rh_rlm_create");
        // staged: the factor sits in front of the converter, where Mixer::add(src.amplify(g)) has it
        let gains: Vec<f32> = g.srcs.iter().map(|x| if staged { 1.0 } else { x.gain }).collect();
        ck(unsafe { rh_rlm_set_gains(g.plan, gains.as_ptr(), gains.len() as u32) }, "rh_rlm_set_gains");
        ck(unsafe { rh_rlm_stream_begin(g.plan) }, "rh_rlm_stream_begin");
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
        self.gens.push(g);
    }

    /// One block of a continuous generation: every source's row = [frames the previous block left unconsumed | a freshly pulled
    /// block]; one copy, the fused launch (resample + filter + ordered sum), what the converter has not consumed is kept.
    fn run_block_direct(&mut self, gi: usize) {
        let (cap_frames, block_frames, stream, out_cap) = (self.cap_frames, self.opt.block_frames, self.pump.stream, self.out_cap_frames);
        let g = &mut self.gens[gi];
        let native: u16 = if g.mono { 1 } else { 2 };                              // channels of the rows the fused launch reads
        let row_len = if g.mono { (cap_frames + 3) & !3 } else { self.row };        // floats per row
        let s_n = g.srcs.len();
        let slot = g.slot;
        g.slot ^= 1;
        g.stage[slot].reserve(s_n * row_len);
        g.din.reserve(s_n * row_len);
        let mut side_off = vec![0usize; s_n];
        let mut side_floats = 0usize;
        for (i, x) in g.srcs.iter().enumerate() {
            if x.ch != native { side_off[i] = side_floats; side_floats += (cap_frames * x.ch as usize + 3) & !3; }
        }
        if side_floats > 0 { g.side[slot].reserve(side_floats); g.dside.reserve(side_floats); }
        let (mut ptrs, mut avail, mut ended) = (Vec::with_capacity(s_n), Vec::with_capacity(s_n), Vec::with_capacity(s_n));
        for i in 0..s_n {
            let x = &mut g.srcs[i];
            let ch = x.ch as usize;
            let row: &mut [f32] = if ch == native as usize { &mut g.stage[slot].slice_mut(s_n * row_len)[i * row_len..(i + 1) * row_len] }
                                  else { &mut g.side[slot].slice_mut(side_floats)[side_off[i]..] };
            let mut have = x.held.len();
            assert!(have / ch + if x.ended { 0 } else { block_frames } <= cap_frames, "GpuMixer: held frames exceed the plan");
            row[..have].copy_from_slice(&x.held);
            if !x.ended {
                let want = block_frames * ch;
                let mut got = read_into(x.up.as_mut(), &mut row[have..have + want]);   // straight into the staging block
                got -= got % ch;                                                    // sources end on frame boundaries (source/mod.rs:169-178)
                have += got;
                x.ended = got < want;
            }
            ptrs.push(unsafe { g.din.p.add(i * row_len) } as *const f32);
            avail.push((have / ch) as u64);
            ended.push(x.ended as u8);
        }
        ck(unsafe { rh_memcpy_h2d(g.din.p.cast(), g.stage[slot].p.cast(), s_n * row_len * 4, stream) }, "rh_memcpy_h2d");
        if side_floats > 0 {                                                        // ChannelCountConverter on the device (channels.rs:57-85), into the stereo rows
            ck(unsafe { rh_memcpy_h2d(g.dside.p.cast(), g.side[slot].p.cast(), side_floats * 4, stream) }, "rh_memcpy_h2d");
            for i in 0..s_n {
                if g.srcs[i].ch != native && avail[i] > 0 {
                    ck(unsafe { rh_channels_convert(g.din.p.add(i * row_len), g.dside.p.add(side_off[i]), avail[i] as usize, g.srcs[i].ch as u32, 2, stream) }, "rh_channels_convert");
                }
            }
        }
        let (mut out, mut consumed) = (0u64, 0u64);
        if g.mono {                                                                 // the mono mix of the block, then ChannelCountConverter(1 -> 2) (channels.rs:64-73) behind the stereo queue
            g.qm.reserve(out_cap as usize * 2);
            ck(unsafe { rh_rlm_stream_block_v(g.plan, ptrs.as_ptr(), avail.as_ptr(), ended.as_ptr(), s_n as u32, g.qm.p, out_cap * 2 - g.fill - g.head, &mut out, &mut consumed, stream) },
               "rh_rlm_stream_block_v");
            if out > 0 { ck(unsafe { rh_channels_convert(g.queue_end(), g.qm.p, out as usize, 1, 2, stream) }, "rh_channels_convert"); }
        } else {
            ck(unsafe { rh_rlm_stream_block_v(g.plan, ptrs.as_ptr(), avail.as_ptr(), ended.as_ptr(), s_n as u32, g.queue_end(), out_cap * 2 - g.fill - g.head, &mut out, &mut consumed, stream) },
               "rh_rlm_stream_block_v");
        }
        g.fill += out;
        let mut all_ended = true;
        for i in 0..s_n {                                                           // keep what the converter has not consumed (a few hundred frames)
            let ch = g.srcs[i].ch as usize;
            let row: &[f32] = if ch == native as usize { &g.stage[slot].slice(s_n * row_len)[i * row_len..] } else { &g.side[slot].slice(side_floats)[side_off[i]..] };
            let have = avail[i] as usize * ch;
            let drop = (consumed as usize * ch).min(have);
            let x = &mut g.srcs[i];
            x.held.clear();
            x.held.extend_from_slice(&row[drop..have]);
            all_ended = all_ended && x.ended;
        }
        g.done = all_ended;                                                         // the call that saw every source ended emitted everything that was left
    }

    /// One block of a span-by-span generation.  Every source is topped up to `target` converted frames: it is pulled piece by piece
    /// (a piece never crosses a span; its length is budgeted so that its output fits the row whatever the span does), the pieces are
    /// planned into segments, ONE copy brings all rows to the device, ONE launch converts all segments of all sources (plus the frames
    /// the last block left over, moved to the front of the other row set), and the fused kernel -- its converter passing through --
    /// filters and mixes the rows.
    fn run_block_staged(&mut self, gi: usize) {
        let (rate, stream, out_cap) = (self.rate, self.pump.stream, self.out_cap_frames);
        let g = &mut self.gens[gi];
        let s_n = g.srcs.len();
        let slot = g.slot;
        g.slot ^= 1;
        let crowf = (g.crow as usize * 2 + 3) & !3;
        // 1. layout of the staging block: a row per live source, sized for what it is about to pull in its current format
        let (mut row_off, mut row_cap) = (vec![0usize; s_n], vec![0usize; s_n]);
        let mut total = 0usize;
        for i in 0..s_n {
            row_off[i] = total;
            let x = &mut g.srcs[i];
            if x.ended { continue; }
            match x.reader.peek(x.up.as_mut()) {
                None => { x.ended = true; }                                         // the chain rodio would build now is empty
                Some((ch, r)) => {
                    let want = g.target.saturating_sub(x.have);
                    let in_frames = want * r as u64 / rate as u64 + 8;
                    row_cap[i] = x.plan.held_samples() + in_frames as usize * ch as usize;
                    total += (row_cap[i] + 3) & !3;
                }
            }
        }
        g.stage[slot].reserve(total.max(4));
        g.din.reserve(total.max(4));
        let mut table: Vec<RhUniformSeg> = Vec::new();
        let mut max_out = 0u64;
        let (oc, nc) = (g.ccur, g.ccur ^ 1);
        for (i, x) in g.srcs.iter().enumerate() {                                  // what the last block left over: to the front of the other row set
            if x.have == 0 { continue; }
            table.push(RhUniformSeg { src: unsafe { g.conv[oc].p.add(i * crowf + x.off as usize * 2) }, dst: unsafe { g.conv[nc].p.add(i * crowf) }, src_frame0: 0, src_frames: x.have,
                                      m0: 0, m1: x.have, span_frames: u64::MAX, from_rate: rate, to_rate: rate, from_ch: 2, to_ch: 2, gain: 1.0, reserved: 0 });
            max_out = max_out.max(x.have);
        }
        // 2. pull and plan
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
                let (ch, r) = match x.reader.peek(x.up.as_mut()) { Some(f) => f, None => { x.ended = true; break; } };
                let now = x.have + x.plan.out_frames();
                if now >= g.target { break; }
                let (need, most) = x.plan.budget(r, x.reader.opens_next(), g.target - now, g.crow - now);
                let n = need.min(most).min(((row_cap[i] - fill) / ch as usize) as u64) as usize;
                if n == 0 { break; }
                let piece = x.reader.read_piece(x.up.as_mut(), &mut row[fill..], n);   // straight into the staging block
                if let Some(pc) = piece { fill += pc.n; x.plan.add(&pc, &mut segs); }
                if x.reader.ended() { x.ended = true; break; }
                if piece.is_none() { break; }
            }
            x.plan.end_block();
            x.held.clear();
            x.held.extend_from_slice(&row[x.plan.keep_offset()..x.plan.keep_offset() + x.plan.keep_samples()]);
            for sg in &segs {
                let mut t = sg.g;
                t.src = unsafe { g.din.p.add(row_off[i] + sg.src_off) };
                t.dst = unsafe { g.conv[nc].p.add(i * crowf + (x.have as usize + sg.dst_off) * 2) };
                t.gain = x.gain;
                max_out = max_out.max(t.m1 - t.m0);
                table.push(t);
            }
            x.have += x.plan.out_frames();
            assert!(x.have <= g.crow, "GpuMixer: converted frames exceed the row");
        }
        // 3. one copy, one conversion launch
        if total > 0 { ck(unsafe { rh_memcpy_h2d(g.din.p.cast(), g.stage[slot].p.cast(), total * 4, stream) }, "rh_memcpy_h2d"); }
        if !table.is_empty() {
            let tf = table.len() * std::mem::size_of::<RhUniformSeg>() / 4;
            g.tab[slot].reserve(tf);
            g.dtab.reserve(tf);
            unsafe {
                ptr::copy_nonoverlapping(table.as_ptr() as *const f32, g.tab[slot].p, tf);
                ck(rh_memcpy_h2d(g.dtab.p.cast(), g.tab[slot].p.cast(), tf * 4, stream), "rh_memcpy_h2d");
                ck(rh_uniform_segments_dev(g.dtab.p as *const RhUniformSeg, table.len() as u32, max_out, stream), "rh_uniform_segments_dev");
            }
        }
        g.ccur = nc;
        // 4. filter + ordered sum of the converted rows
        let ptrs: Vec<*const f32> = (0..s_n).map(|i| unsafe { g.conv[nc].p.add(i * crowf) } as *const f32).collect();
        let avail: Vec<u64> = g.srcs.iter().map(|x| x.have).collect();
        let ended: Vec<u8> = g.srcs.iter().map(|x| x.ended as u8).collect();
        let (mut out, mut consumed) = (0u64, 0u64);
        ck(unsafe { rh_rlm_stream_block_v(g.plan, ptrs.as_ptr(), avail.as_ptr(), ended.as_ptr(), s_n as u32, g.queue_end(), out_cap * 2 - g.fill - g.head, &mut out, &mut consumed, stream) },
           "rh_rlm_stream_block_v");
        g.fill += out;
        for x in &mut g.srcs { let d = consumed.min(x.have); x.off = d; x.have -= d; }
        g.done = g.srcs.iter().all(|x| x.ended);
    }
    fn run_block(&mut self, gi: usize) { if self.gens[gi].staged { self.run_block_staged(gi) } else { self.run_block_direct(gi) } }

    /// `Mixer::add` on a running mixer.  rodio admits the source at the next frame boundary of the output (mixer.rs:175-183).  Here up to
    /// two blocks are already mixed beyond that frame (the one being served, the one in flight), so the new source -- its own
    /// generation, its own fused stream and clock from frame J on -- is run ahead synchronously until it covers them, added onto their
    /// device copies at its offset (rh_mix_sum: old mix first, the newcomer last = insertion order) and the blocks travel to the host again.
    fn late_join(&mut self, mut item: Src) {
        let ci = self.pump.cur;
        let consumed = self.slot_base[ci] * 2 + self.pump.pos as u64;               // samples already handed out
        let j = (consumed + 1) / 2;                                                  // the next frame boundary
        let flight = self.pump.other_in_flight();
        let li = if flight { ci ^ 1 } else { ci };
        let sched_end = self.slot_base[li] + self.slot_frames[li];
        let stream = self.pump.stream;
        ck(unsafe { rh_stream_synchronize(stream) }, "rh_stream_synchronize");     // the blocks about to be patched have been produced
        let staged = item.up.current_span_len().is_some();
        if !staged { self.make_direct(&mut item); }
        let mono = !staged && item.ch == 1;
        self.start_stream(vec![item], staged, mono);
        self.last_join = j;
        let gi = self.gens.len() - 1;
        let need = sched_end.saturating_sub(j);
        while self.gens[gi].fill < need && !self.gens[gi].done {
            self.run_block(gi);
            ck(unsafe { rh_stream_synchronize(stream) }, "rh_stream_synchronize");
        }
        for k in 0..(if flight { 2 } else { 1 }) {
            let si = if k == 0 { ci } else { ci ^ 1 };
            let (b0, b1) = (self.slot_base[si], self.slot_base[si] + self.slot_frames[si]);
            let g = &self.gens[gi];
            let (lo, hi) = (j.max(b0), b1.min(j + g.fill));
            if hi <= lo { continue; }
            let ptrs = [self.dkeep[si].p as *const f32, unsafe { g.queue().add(((lo - j) * 2) as usize) }];
            let (start, len) = ([0u64, (lo - b0) * 2], [self.slot_frames[si] * 2, (hi - lo) * 2]);
            self.dmix.reserve(self.out_cap_frames as usize * 4);
            unsafe {
                ck(rh_mix_sum(self.dmix.p, (self.slot_frames[si] * 2) as usize, ptrs.as_ptr(), start.as_ptr(), len.as_ptr(), 2, stream), "rh_mix_sum");
                ck(rh_memcpy_d2d(self.dkeep[si].p.cast(), self.dmix.p.cast(), (self.slot_frames[si] * 2 * 4) as usize, stream), "rh_memcpy_d2d");
                ck(rh_memcpy_d2h_async(self.pump.slot[si].out.p.cast(), self.dkeep[si].p.cast(), (self.slot_frames[si] * 2 * 4) as usize, stream), "rh_memcpy_d2h_async");
            }
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
        let g = &self.gens[gi];
        if self.pump.slot[last_i].last && !(g.done && g.fill == 0) {
            self.pump.slot[last_i].last = false;
            if !flight { self.submit(ci ^ 1); }
        }
    }
}

impl BlockSource for GpuMixer {
    fn pump(&mut self) -> &mut Pump { &mut self.pump }
    fn can_resume(&self) -> bool { !self.pending.is_empty() && self.resume_ok }     // mixer.rs:117-136: None while empty, samples again after add()
    fn block_done(&mut self) {                                                    // a bounded wait inside the fused kernel expired (never seen on a healthy device): fail loudly
        for g in &self.gens { if !g.plan.is_null() { ck(unsafe { rh_rlm_last_status(g.plan) }, "rh_rlm_last_status"); } }
    }
    fn enqueue(&mut self, i: usize) {
        if !self.pending.is_empty() { self.start_generation(); }
        if self.gens.is_empty() {                                                  // nothing to pull
            self.pump.slot[i].n = 0; self.pump.slot[i].last = true;
            self.slot_base[i] = self.scheduled; self.slot_frames[i] = 0;
            return;
        }
        let stream = self.pump.stream;
        self.pump.slot[i].out.reserve(self.out_cap_frames as usize * 4);
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
            self.dkeep[i].reserve(self.out_cap_frames as usize * 4);              // the block also stays on the device until it has been served (late_join)
            unsafe {
                ck(rh_memcpy_d2h_async(self.pump.slot[i].out.p.cast(), mixed.cast(), (n * 2 * 4) as usize, stream), "rh_memcpy_d2h_async");
                ck(rh_memcpy_d2d(self.dkeep[i].p.cast(), mixed.cast(), (n * 2 * 4) as usize, stream), "rh_memcpy_d2d");
            }
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
            ck(unsafe { rh_stream_synchronize(stream) }, "rh_stream_synchronize");
            for g in self.gens.drain(..) {
                if !g.plan.is_null() {
                    ck(unsafe { rh_rlm_last_status(g.plan) }, "rh_rlm_last_status"); // the last blocks too: nothing is served unchecked
                    ck(unsafe { rh_rlm_destroy(g.plan) }, "rh_rlm_destroy");
                }
            }
        }
        self.pump.slot[i].n = n as usize * 2;
        self.pump.slot[i].last = self.gens.is_empty() && self.pending.is_empty();
    }
}
impl Iterator for GpuMixer {
    type Item = f32;
    // MixerSource::next advances its channel position on every call, also on the ones that return None (mixer.rs:120-136), and admits
    // pending sources only at channel 0: an ended mixer that gets a new source after an odd number of calls returns one more None.
    fn next(&mut self) -> Option<f32> {
        self.resume_ok = self.calls % 2 == 0;
        self.calls += 1;
        self.next_sample()
    }
}
impl Source for GpuMixer {
    fn current_span_len(&self) -> Option<usize> { None }
    fn channels(&self) -> ChannelCount { ChannelCount::new(2).unwrap() }
    fn sample_rate(&self) -> SampleRate { SampleRate::new(self.rate).unwrap() }
    fn total_duration(&self) -> Option<Duration> { None }
    fn try_seek(&mut self, _: Duration) -> Result<(), SeekError> { Err(SeekError::NotSupported { underlying_source: "rodio_hip::GpuMixer (like MixerSource, mixer.rs:160-170)" }) }
}
impl Drop for GpuMixer {
    fn drop(&mut self) {
        unsafe {
            rh_stream_synchronize(self.pump.stream);
            for g in &self.gens { if !g.plan.is_null() { rh_rlm_destroy(g.plan); } }
        }
    }
}
