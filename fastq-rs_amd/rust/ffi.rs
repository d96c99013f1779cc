//! ffi.rs — the binding a maintainer of the `fastq` crate would add to call libfastq_hip.so.
//!
//! NOT COMPILED IN THIS REPOSITORY: the build image has no rustc/cargo.  It is the reference-side
//! half of the drop-in boundary; the tested half is the C ABI itself
//! (include/fastq_hip.h, exercised through ctypes and through the C++ mirror host/fastq.hpp, which
//! has exactly the shape this file gives the Rust side).
//!
//! Where it slots into the crate (fastq 0.6.0):
//!   * `RecordSetIter::next` (src/lib.rs:364-425) and `RecordRefIter::advance` (src/lib.rs:255-303)
//!     stop calling `IdxRecord::from_buffer` (src/records.rs:201-247) once per record; they pull a
//!     whole chunk's `Vec<IdxRecord>` from `GpuScanner::next_chunk` instead.
//!   * `Buffer` (src/buffer.rs) is replaced by the pinned ring behind `fqh_stream_*`.
#![allow(non_camel_case_types, dead_code)]
use std::io::{Error, ErrorKind, Read, Result};
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)]
#[derive(Clone, Copy)]
pub struct fqh_stream_times { pub wall_ms: f64, pub copy_busy_ms: f64, pub scan_busy_ms: f64, pub both_busy_ms: f64, pub n_slots: u64 }
#[repr(C)]
#[derive(Clone, Copy)]
pub struct fqh_shard_result { pub status: i32, pub phase: u32, pub n_records: u64, pub n_newlines: u64, pub err_offset: u64,
                              pub head_len: u64, pub tail_len: u64, pub flags: u64 }
// flags: bit 0 = n_newlines stops where the stream stopped; bit 1 = the read callback failed for bytes of this range (not a failure
// of the run: the range defers, the rank that parses the gap it lies in reads its bytes again in file order — include/fastq_hip.h)
#[repr(C)] pub struct fqh_ctx { _p: [u8; 0] }
#[repr(C)] pub struct fqh_stream { _p: [u8; 0] }

#[repr(C)] #[derive(Clone, Copy)]
pub struct fqh_idx_record { pub start: u64, pub head: u32, pub seq: u32, pub sep: u32, pub qual: u32 }

#[repr(C)]
pub struct fqh_chunk {
    pub parse_status: i32, pub is_final: i32, pub n_records: u64, pub base_offset: u64,
    pub data_len: u64, pub lead_len: u64, pub h_data: *const u8, pub h_index: *const fqh_idx_record,
    pub h_rec_start: *const u64, pub d_data: *const u8, pub d_rec_start: *const u64,
    pub err_record: u64, pub err_offset: u64, pub err_need: u64,
}

#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct fqh_carry { pub base_offset: u64, pub nl_count: u64, pub back: [u64; 4] }

#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct fqh_summary {
    pub n_records: u64, pub bytes_consumed: u64, pub parse_status: i32, pub reserved: i32,
    pub err_record: u64, pub err_offset: u64, pub n_newlines: u64, pub tail_len: u64,
    pub max_record_len: u64, pub n_line_starts: u64,
}

#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct fqh_timing { pub total_ms: f32, pub index_ms: f32, pub prefix_ms: f32, pub emit_ms: f32, pub stats_ms: f32 }

#[repr(C)] pub struct fqh_comm { _p: [u8; 0] }
pub const FQH_COMM_ID_BYTES: usize = 128;
pub const FQH_OPT_FAST_PATH: c_int = 1;
pub const FQH_OPT_SINGLE_PASS: c_int = 2;
pub const FQH_OPT_PLACE_TRIES: c_int = 3;
pub const FQH_OPT_SPIN_WAIT: c_int = 4;
pub const FQH_OPT_REUSE_INDEX: c_int = 5;      // default 0: fqh_stats after fqh_scan on the same buffer reads the input again
pub const FQH_OPT_OWN_STREAM_NONBLOCKING: c_int = 7;   // default 0: the context's own stream is ordered against the legacy null stream
pub const FQH_OPT_ADAPT_LINES: c_int = 6;       // default 3: a second line buffer is tried for big inputs that are scanned again
pub const FQH_SHARD_WORDS: usize = 8;
pub const FQH_SHARD_STREAM_WORDS: usize = 10;
pub const FQH_SHARD_MAX_RANKS: c_int = 256;
pub const FQH_NO_ERROR_KEY: u64 = u64::MAX;
pub const FQH_SHARD_EMPTY: u32 = 0xFFFF_FFFF;  // fqh_shard_result.phase of an empty byte range
pub const FQH_SHARD_PASS: u32 = 0xFFFF_FFFE;   // ... of a byte range without a record start
pub const FQH_SHARD_DEFER: u32 = 0xFFFF_FFFD;  // ... of a byte range whose window does not single out a line phase
pub const FQH_OPT_KEEP_RING: c_int = 8;         // default 0: a destroyed ring's pinned slots are freed, not kept for the next ring
pub const FQH_ABI_VERSION: c_int = 1;
pub const FQH_NSCALARS: usize = 8;
// fqh_status (the five parse errors carry the crate's own messages: fqh_strerror)
pub const FQH_E_HEADER: c_int = 1;        // src/records.rs:143-146
pub const FQH_E_SEP: c_int = 2;           // src/records.rs:157-160
pub const FQH_E_LEN_MISMATCH: c_int = 3;  // src/records.rs:234-237
pub const FQH_E_TRUNCATED: c_int = 4;     // src/lib.rs:287-290
pub const FQH_E_TOO_LONG: c_int = 5;      // src/lib.rs:279-282
pub const FQH_E_IO: c_int = 6;
pub const FQH_E_DEVICE: c_int = 7;
pub const FQH_E_ARG: c_int = 8;
pub const FQH_E_AGAIN: c_int = 10;

#[link(name = "fastq_hip")]
extern "C" {
    pub fn fqh_create(device: c_int, out: *mut *mut fqh_ctx) -> c_int;
    pub fn fqh_destroy(ctx: *mut fqh_ctx);
    pub fn fqh_strerror(status: c_int) -> *const c_char;
    pub fn fqh_last_error(ctx: *mut fqh_ctx) -> *const c_char;
    pub fn fqh_abi_version() -> c_int;
    pub fn fqh_set_stream(ctx: *mut fqh_ctx, hip_stream: *mut c_void) -> c_int;
    pub fn fqh_set_bufsize(ctx: *mut fqh_ctx, bufsize: u64) -> c_int;
    pub fn fqh_set_option(ctx: *mut fqh_ctx, option: c_int, value: c_int) -> c_int;
    pub fn fqh_last_scan_fast(ctx: *mut fqh_ctx) -> c_int;
    pub fn fqh_last_stats_route(ctx: *mut fqh_ctx) -> c_int;
    pub fn fqh_placement(ctx: *mut fqh_ctx, n_candidates: *mut c_int, ms: *mut f32) -> c_int;   // ms: [f32; 10]
    pub fn fqh_line_buffers(ctx: *mut fqh_ctx, n_alive: *mut c_int, n_unsettled: *mut c_int, bytes: *mut u64) -> c_int;

    // ---- whole buffers in HBM: IdxRecord::from_buffer over every record (src/records.rs:201-247)
    pub fn fqh_scan(ctx: *mut fqh_ctx, d_buf: *const u8, len: u64, is_final: c_int, carry_in: *const fqh_carry,
                    d_rec_start: *mut u64, cap: u64, out: *mut fqh_summary, carry_out: *mut fqh_carry) -> c_int;
    pub fn fqh_scan_launch(ctx: *mut fqh_ctx, d_buf: *const u8, len: u64, is_final: c_int, carry_in: *const fqh_carry,
                           d_rec_start: *mut u64, cap: u64) -> c_int;
    pub fn fqh_scan_finish(ctx: *mut fqh_ctx, out: *mut fqh_summary, carry_out: *mut fqh_carry) -> c_int;
    pub fn fqh_index_records(ctx: *mut fqh_ctx, d_index: *mut fqh_idx_record, cap: u64) -> c_int;
    pub fn fqh_invalidate(ctx: *mut fqh_ctx) -> c_int;

    // ---- the closure of Parser::each that reads seq() / qual() (src/lib.rs:226-237, src/records.rs:75-90)
    pub fn fqh_stats(ctx: *mut fqh_ctx, d_buf: *const u8, len: u64, is_final: c_int, carry_in: *const fqh_carry,
                     lmax: u32, d_qual_hist: *mut u64, d_base_hist: *mut u64, d_scalars: *mut u64,
                     out: *mut fqh_summary, carry_out: *mut fqh_carry) -> c_int;
    pub fn fqh_stats_launch(ctx: *mut fqh_ctx, d_buf: *const u8, len: u64, is_final: c_int, carry_in: *const fqh_carry,
                            lmax: u32, d_qual_hist: *mut u64, d_base_hist: *mut u64, d_scalars: *mut u64) -> c_int;
    pub fn fqh_stats_launch_lead(ctx: *mut fqh_ctx, d_buf: *const u8, len: u64, lead_len: u64, is_final: c_int,
                                 carry_in: *const fqh_carry, lmax: u32, d_qual_hist: *mut u64, d_base_hist: *mut u64,
                                 d_scalars: *mut u64) -> c_int;
    pub fn fqh_stats_finish(ctx: *mut fqh_ctx, out: *mut fqh_summary, carry_out: *mut fqh_carry) -> c_int;
    pub fn fqh_scan_stats(ctx: *mut fqh_ctx, d_buf: *const u8, len: u64, is_final: c_int, carry_in: *const fqh_carry,
                          d_rec_start: *mut u64, cap: u64, lmax: u32, d_qual_hist: *mut u64, d_base_hist: *mut u64,
                          d_scalars: *mut u64, out: *mut fqh_summary, carry_out: *mut fqh_carry) -> c_int;
    pub fn fqh_scan_stats_launch(ctx: *mut fqh_ctx, d_buf: *const u8, len: u64, is_final: c_int,
                                 carry_in: *const fqh_carry, d_rec_start: *mut u64, cap: u64, lmax: u32,
                                 d_qual_hist: *mut u64, d_base_hist: *mut u64, d_scalars: *mut u64) -> c_int;
    pub fn fqh_scan_stats_finish(ctx: *mut fqh_ctx, out: *mut fqh_summary, carry_out: *mut fqh_carry) -> c_int;

    // ---- byte-range shards across GPUs (the crate is single-process; nearest: src/lib.rs:553-559)
    pub fn fqh_shard_prescan(ctx: *mut fqh_ctx, d_buf: *const u8, len: u64, n_newlines: *mut u64,
                             n_line_starts: *mut u64, back_zero_carry: *mut u64) -> c_int;
    pub fn fqh_shard_prescan_launch(ctx: *mut fqh_ctx, d_buf: *const u8, len: u64, d_words: *mut u64) -> c_int;
    pub fn fqh_shard_rescan_launch(ctx: *mut fqh_ctx, is_final: c_int, d_all_words: *const u64, n_ranks: c_int, rank: c_int,
                                   d_rec_start: *mut u64, cap: u64, d_counts: *mut u64) -> c_int;
    pub fn fqh_carry_combine(prev: *const fqh_carry, len: u64, n_newlines: u64, n_line_starts: u64,
                             back_zero_carry: *const u64, next: *mut fqh_carry) -> c_int;
    pub fn fqh_rescan_launch(ctx: *mut fqh_ctx, is_final: c_int, carry_in: *const fqh_carry, d_rec_start: *mut u64,
                             cap: u64) -> c_int;
    pub fn fqh_shard_align(ctx: *mut fqh_ctx, d_buf: *const u8, len: u64, prev_is_newline: c_int, phase: *mut u32,
                           first_record_offset: *mut u64) -> c_int;
    pub fn fqh_comm_unique_id(id: *mut u8) -> c_int;
    pub fn fqh_comm_create(ctx: *mut fqh_ctx, n_ranks: c_int, rank: c_int, id: *const u8, out: *mut *mut fqh_comm) -> c_int;
    pub fn fqh_comm_destroy(comm: *mut fqh_comm);
    pub fn fqh_allgather(ctx: *mut fqh_ctx, comm: *mut fqh_comm, d_send: *const c_void, d_recv: *mut c_void,
                         bytes_per_rank: u64) -> c_int;
    pub fn fqh_allreduce_u64(ctx: *mut fqh_ctx, comm: *mut fqh_comm, d_buf: *mut u64, n: u64) -> c_int;
    pub fn fqh_allreduce_min_u64(ctx: *mut fqh_ctx, comm: *mut fqh_comm, d_buf: *mut u64, n: u64) -> c_int;
    pub fn fqh_sync(ctx: *mut fqh_ctx) -> c_int;
    // ---- the sharded, host-streamed mode (one process per GPU): the gather + error return of Parser::parallel_each,
    // src/lib.rs:544-564, over byte-range shards
    pub fn fqh_shard_stream_run(ctx: *mut fqh_ctx, read: extern "C" fn(*mut c_void, *mut u8, u64, u64) -> c_int, user: *mut c_void,
                                lo: u64, hi: u64, file_len: u64, slot_bytes: u64, n_slots: u32, lmax: u32,
                                d_qual_hist: *mut u64, d_base_hist: *mut u64, d_scalars: *mut u64, res: *mut fqh_shard_result) -> c_int;
    pub fn fqh_shard_stream_run_mapped(ctx: *mut fqh_ctx, read: extern "C" fn(*mut c_void, *mut u8, u64, u64) -> c_int,
                                       map: extern "C" fn(*mut c_void, u64, u64, *mut u64) -> *const u8, user: *mut c_void,
                                       lo: u64, hi: u64, file_len: u64, slot_bytes: u64, n_slots: u32, lmax: u32,
                                       d_qual_hist: *mut u64, d_base_hist: *mut u64, d_scalars: *mut u64, res: *mut fqh_shard_result) -> c_int;
    pub fn fqh_shard_result_words(res: *const fqh_shard_result, lo: u64, hi: u64, words: *mut u64);
    pub fn fqh_shard_failed_words(why: c_int, lo: u64, hi: u64, words: *mut u64);
    pub fn fqh_shard_stream_finish(ctx: *mut fqh_ctx, read: extern "C" fn(*mut c_void, *mut u8, u64, u64) -> c_int, user: *mut c_void,
                                   file_len: u64, h_all_words: *const u64, n_ranks: c_int, rank: c_int, slot_bytes: u64, n_slots: u32,
                                   lmax: u32, d_qual_hist: *mut u64, d_base_hist: *mut u64, d_scalars: *mut u64, out: *mut u64) -> c_int;
    pub fn fqh_shard_failure_key(rank: c_int, offset: u64, why: c_int) -> u64;
    pub fn fqh_shard_stream_outcome(min_key: u64, records_per_rank: *const u64, n_ranks: c_int, status: *mut i32,
                                    n_records: *mut u64, err_offset: *mut u64) -> c_int;

    // ---- Buffer + thread_reader (src/buffer.rs, src/thread_reader.rs:182-200): the pinned ring
    pub fn fqh_stream_create(ctx: *mut fqh_ctx, slot_bytes: u64, n_slots: u32, flags: u32,
                             out: *mut *mut fqh_stream) -> c_int;
    pub fn fqh_stream_destroy(st: *mut fqh_stream);
    pub fn fqh_stream_acquire(st: *mut fqh_stream, h_dst: *mut *mut u8, cap: *mut u64) -> c_int;
    pub fn fqh_stream_submit(st: *mut fqh_stream, nbytes: u64, is_final: c_int) -> c_int;
    // the slot's bytes straight from the host's own page-locked memory (an mmap'ed file registered once): no staging copy — the
    // copy src/thread_reader.rs:90-97 makes
    pub fn fqh_stream_submit_external(st: *mut fqh_stream, h_src: *const u8, nbytes: u64, is_final: c_int) -> c_int;
    pub fn fqh_host_register(ctx: *mut fqh_ctx, h_ptr: *mut c_void, bytes: u64) -> c_int;
    pub fn fqh_host_unregister(ctx: *mut fqh_ctx, h_ptr: *mut c_void) -> c_int;
    pub fn fqh_stream_collect(st: *mut fqh_stream, out: *mut fqh_chunk) -> c_int;
    pub fn fqh_stream_release(st: *mut fqh_stream) -> c_int;
    pub fn fqh_stream_release_chunk(st: *mut fqh_stream, c: *const fqh_chunk) -> c_int;   // RecordSets that borrow a slot (src/lib.rs:306-318)
    pub fn fqh_stream_carry(st: *mut fqh_stream, out: *mut fqh_carry) -> c_int;
    pub fn fqh_stream_set_origin(st: *mut fqh_stream, file_offset: u64) -> c_int;
    pub fn fqh_stream_note_read(st: *mut fqh_stream, got: u64, asked: u64) -> c_int;   // a reader that comes back short (src/buffer.rs:74-100)
    pub fn fqh_stream_timing(st: *mut fqh_stream, out: *mut fqh_stream_times) -> c_int;
    // histograms per delivered record (FQH_STREAM_STATS) and the device-side filter (flags + gather);
    // not needed by Parser itself, bound for consumers that want them
    pub fn fqh_stream_set_stats(st: *mut fqh_stream, lmax: u32, d_qual_hist: *mut u64, d_base_hist: *mut u64,
                                d_scalars: *mut u64) -> c_int;
    pub fn fqh_len_hist(ctx: *mut fqh_ctx, d_base_hist: *const u64, d_scalars: *const u64, lmax: u32, d_len_hist: *mut u64) -> c_int;
    pub fn fqh_record_flags(ctx: *mut fqh_ctx, d_buf: *const u8, len: u64, base_offset: u64,
                            d_index: *const fqh_idx_record, n: u64, d_flags: *mut u8) -> c_int;
    pub fn fqh_gather_records(ctx: *mut fqh_ctx, d_buf: *const u8, len: u64, base_offset: u64,
                              d_index: *const fqh_idx_record, n: u64, d_flags: *const u8, mask: u8, want: u8,
                              d_out: *mut u8, out_cap: u64, n_selected: *mut u64, out_bytes: *mut u64) -> c_int;

    // ---- benchmarks: kernel times of the last launch, the synthetic file of SURVEY 8(d), the bare streaming read
    pub fn fqh_last_timing(ctx: *mut fqh_ctx, out: *mut fqh_timing) -> c_int;
    pub fn fqh_synth_fill(ctx: *mut fqh_ctx, d_out: *mut u8, byte_off: u64, len: u64, seed: u64) -> c_int;
    pub fn fqh_read_ceiling(ctx: *mut fqh_ctx, d_buf: *const u8, len: u64, checksum: *mut u64, ms: *mut f32) -> c_int;

    // ---- device memory for hosts without a HIP binding of their own
    pub fn fqh_dev_alloc(ctx: *mut fqh_ctx, bytes: u64, d_ptr: *mut *mut c_void) -> c_int;
    pub fn fqh_dev_free(ctx: *mut fqh_ctx, d_ptr: *mut c_void) -> c_int;
    pub fn fqh_memcpy_h2d(ctx: *mut fqh_ctx, d_dst: *mut c_void, h_src: *const c_void, bytes: u64) -> c_int;
    pub fn fqh_memcpy_d2h(ctx: *mut fqh_ctx, h_dst: *mut c_void, d_src: *const c_void, bytes: u64) -> c_int;
    pub fn fqh_memset(ctx: *mut fqh_ctx, d_dst: *mut c_void, value: c_int, bytes: u64) -> c_int;
}

const FQH_OK: c_int = 0;
const FQH_BUFSIZE: usize = 68 * 1024;   // src/lib.rs:128-129
const FQH_E_CAPACITY: c_int = 9;
const FQH_STREAM_INDEX: u32 = 1;
#[allow(dead_code)]
const FQH_STREAM_STATS: u32 = 2;
#[allow(dead_code)]
const FQH_STREAM_TIMING: u32 = 4;
#[allow(dead_code)]
const FQH_STREAM_EXTERNAL: u32 = 8;

/// What `Parser` holds instead of `buffer::Buffer`.
pub struct GpuScanner<R: Read> {
    reader: R, ctx: *mut fqh_ctx, st: *mut fqh_stream, eof: bool,
    /// a chunk's parse error waits here while the records in front of it are handed out
    pending: Option<Error>,
    /// no record delivered yet: chunks end as soon as four newlines have come in (the first record early, src/lib.rs:264-275)
    startup: bool, startup_newlines: usize, startup_target: usize,
}

impl<R: Read> GpuScanner<R> {
    pub fn new(reader: R) -> Result<Self> {
        unsafe {
            let mut ctx = std::ptr::null_mut();
            if fqh_create(0, &mut ctx) != FQH_OK { return Err(Error::new(ErrorKind::Other, "no MI355X")); }
            // BUFSIZE emulation ("Fastq record is too long", src/lib.rs:278-283) stays on: 68 KiB.
            let mut st = std::ptr::null_mut();
            if fqh_stream_create(ctx, 32 << 20, 3, FQH_STREAM_INDEX, &mut st) != FQH_OK {
                fqh_destroy(ctx);
                return Err(Error::new(ErrorKind::Other, "fqh_stream_create"));
            }
            Ok(GpuScanner { reader, ctx, st, eof: false, pending: None, startup: true, startup_newlines: 0, startup_target: 0 })
        }
    }

    /// read() -> pinned slot -> async H2D, as long as the ring has a free slot
    fn fill(&mut self) -> Result<()> {
        unsafe {
            while !self.eof {
                let (mut dst, mut cap) = (std::ptr::null_mut(), 0u64);
                match fqh_stream_acquire(self.st, &mut dst, &mut cap) {
                    FQH_OK => {}
                    FQH_E_CAPACITY => return Ok(()),
                    _ => return Err(Error::new(ErrorKind::Other, "fqh_stream_acquire")),
                }
                let first = self.startup;
                if first {   // per chunk: the newline count starts over, the chunks double until a record is out
                    self.startup_newlines = 0;
                    self.startup_target = if self.startup_target == 0 { (cap as usize).min(FQH_BUFSIZE) } else { (cap as usize).min(2 * self.startup_target) };
                }
                let target = if first { self.startup_target } else { cap as usize };
                let slot = std::slice::from_raw_parts_mut(dst, target);
                let mut n = 0usize;
                while n < slot.len() {
                    let asked = slot.len() - n;
                    match self.reader.read(&mut slot[n..]) {
                        Ok(0) => { self.eof = true; break; }
                        Ok(k) => {
                            // a reader that comes back short decides the 69 618 .. 69 632-byte band (src/buffer.rs:74-100)
                            if fqh_stream_note_read(self.st, k as u64, asked as u64) != FQH_OK {
                                return Err(Error::new(ErrorKind::Other, "fqh_stream_note_read"));
                            }
                            if first { self.startup_newlines += slot[n..n + k].iter().filter(|&&b| b == b'\n').count(); }
                            n += k;
                            if first && self.startup_newlines >= 4 { break; }
                        }
                        Err(ref e) if e.kind() == ErrorKind::Interrupted => {} // src/buffer.rs:85-97
                        Err(e) => return Err(e),
                    }
                }
                if fqh_stream_submit(self.st, n as u64, self.eof as c_int) != FQH_OK {
                    return Err(Error::new(ErrorKind::Other, "fqh_stream_submit"));
                }
                if first { return Ok(()); }   // collect it before anything else is read
            }
            Ok(())
        }
    }

    /// One chunk of records: the GPU's replacement for ~hundreds of thousands of
    /// `IdxRecord::from_buffer` calls.  The slices stay valid until the next call.
    ///
    /// A chunk that ends in a parse error still carries the `n_records` valid records in front of
    /// the bad one; `Parser::each` (src/lib.rs:226-237) hands those to the closure before it returns
    /// the error, so they are returned first and the error by the call after.
    pub fn next_chunk(&mut self) -> Result<Option<(&[u8], &[fqh_idx_record], u64, bool)>> {
        if let Some(e) = self.pending.take() { return Err(e); }
        self.fill()?;
        unsafe {
            let mut c: fqh_chunk = std::mem::zeroed();
            fqh_stream_release(self.st); // releases the previous chunk, if any
            if fqh_stream_collect(self.st, &mut c) != FQH_OK { return Ok(None); }
            if c.n_records != 0 { self.startup = false; }
            if c.parse_status != FQH_OK {
                // the error text is the crate's own (fqh_strerror returns the exact strings)
                let msg = std::ffi::CStr::from_ptr(fqh_strerror(c.parse_status)).to_string_lossy().into_owned();
                let e = Error::new(ErrorKind::InvalidData, msg);
                if c.n_records == 0 { return Err(e); }
                self.pending = Some(e);
            }
            let lead = c.lead_len as usize;
            let bytes = std::slice::from_raw_parts(c.h_data.sub(lead), lead + c.data_len as usize);
            let idx = std::slice::from_raw_parts(c.h_index, c.n_records as usize);
            Ok(Some((bytes, idx, c.base_offset - c.lead_len, c.is_final != 0)))
        }
    }
}

impl<R: Read> Drop for GpuScanner<R> {
    fn drop(&mut self) { unsafe { fqh_stream_destroy(self.st); fqh_destroy(self.ctx); } }
}

// In src/lib.rs the iterator then becomes (sketch):
//
//   impl<R: Read> RecordRefIter<R> {
//       pub fn advance(&mut self) -> Result<()> {
//           self.i += 1;
//           while self.i >= self.idx.len() {
//               match self.scanner.next_chunk()? { None => { self.current = None; return Ok(()) }
//                   Some((bytes, idx, origin, _)) => { self.bytes = bytes; self.idx = idx; self.origin = origin; self.i = 0; } }
//           }
//           let r = self.idx[self.i];
//           let at = (r.start - self.origin) as usize;
//           self.current = Some(IdxRecord { head: r.head as usize, seq: r.seq as usize, sep: r.sep as usize,
//                                           qual: r.qual as usize, data: (at, at + r.qual as usize + 1) });
//           Ok(())
//       }
//   }

// The sharded mode in the crate's terms (sketch; the C++ mirror fastq::each_sharded in host/fastq.hpp is the tested form):
//
//   pub fn each_sharded(ctx, comm, n_ranks, rank, file: &File, lmax, d_hist) -> io::Result<u64> {
//       extern "C" fn read_at(user: *mut c_void, dst: *mut u8, off: u64, n: u64) -> c_int {
//           let f = unsafe { &*(user as *const File) };
//           let buf = unsafe { std::slice::from_raw_parts_mut(dst, n as usize) };
//           f.read_exact_at(buf, off).map(|_| 0).unwrap_or(1)                 // std::os::unix::fs::FileExt
//       }
//       let (lo, hi) = (len / n * rank, if rank + 1 == n { len } else { len / n * (rank + 1) });
//       let st = fqh_shard_stream_run(ctx, read_at, file as *const _ as *mut c_void, lo, hi, len, 32 << 20, 3, lmax, d_q, d_b, d_sc, &mut res);
//       // a rank that failed still takes part in the exchange (the others would wait for it forever):
//       if st == FQH_OK { fqh_shard_result_words(&res, lo, hi, words.as_mut_ptr()) } else { fqh_shard_failed_words(st, lo, hi, words.as_mut_ptr()) }
//       // one fqh_allgather of the FQH_SHARD_STREAM_WORDS words, then
//       let st = fqh_shard_stream_finish(ctx, read_at, user, len, all_words.as_ptr(), n, rank, 32 << 20, 3, lmax, d_q, d_b, d_sc, out.as_mut_ptr());
//       if st != FQH_OK { out = [0, fqh_shard_failure_key(rank, lo, st)] }
//       // fqh_allreduce_u64 over [records_per_rank (out[0] in slot `rank`) | scalars | histograms], fqh_allreduce_min_u64 over out[1],
//       // fqh_shard_stream_outcome(min_key, records_per_rank, n, &mut status, &mut n_records, &mut err_offset):
//       // Err(InvalidData, fqh_strerror(status)) — what parallel_each returns when the parse fails (src/lib.rs:561-564) — or Ok(n_records)
//   }
