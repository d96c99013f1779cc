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

#[link(name = "fastq_hip")]
extern "C" {
    pub fn fqh_create(device: c_int, out: *mut *mut fqh_ctx) -> c_int;
    pub fn fqh_destroy(ctx: *mut fqh_ctx);
    pub fn fqh_strerror(status: c_int) -> *const c_char;
    pub fn fqh_set_bufsize(ctx: *mut fqh_ctx, bufsize: u64) -> c_int;
    pub fn fqh_stream_create(ctx: *mut fqh_ctx, slot_bytes: u64, n_slots: u32, flags: u32,
                             out: *mut *mut fqh_stream) -> c_int;
    pub fn fqh_stream_destroy(st: *mut fqh_stream);
    pub fn fqh_stream_acquire(st: *mut fqh_stream, h_dst: *mut *mut u8, cap: *mut u64) -> c_int;
    pub fn fqh_stream_submit(st: *mut fqh_stream, nbytes: u64, is_final: c_int) -> c_int;
    pub fn fqh_stream_collect(st: *mut fqh_stream, out: *mut fqh_chunk) -> c_int;
    pub fn fqh_stream_release(st: *mut fqh_stream) -> c_int;
    // histograms per delivered record (FQH_STREAM_STATS) and the device-side filter (flags + gather);
    // not needed by Parser itself, bound for consumers that want them
    pub fn fqh_stream_set_stats(st: *mut fqh_stream, lmax: u32, d_qual_hist: *mut u64, d_base_hist: *mut u64,
                                d_scalars: *mut u64) -> c_int;
    pub fn fqh_record_flags(ctx: *mut fqh_ctx, d_buf: *const u8, len: u64, base_offset: u64,
                            d_index: *const fqh_idx_record, n: u64, d_flags: *mut u8) -> c_int;
    pub fn fqh_gather_records(ctx: *mut fqh_ctx, d_buf: *const u8, len: u64, base_offset: u64,
                              d_index: *const fqh_idx_record, n: u64, d_flags: *const u8, mask: u8, want: u8,
                              d_out: *mut u8, out_cap: u64, n_selected: *mut u64, out_bytes: *mut u64) -> c_int;
}

const FQH_OK: c_int = 0;
const FQH_E_CAPACITY: c_int = 9;
const FQH_STREAM_INDEX: u32 = 1;
#[allow(dead_code)]
const FQH_STREAM_STATS: u32 = 2;

/// What `Parser` holds instead of `buffer::Buffer`.
pub struct GpuScanner<R: Read> {
    reader: R, ctx: *mut fqh_ctx, st: *mut fqh_stream, eof: bool,
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
            Ok(GpuScanner { reader, ctx, st, eof: false })
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
                let slot = std::slice::from_raw_parts_mut(dst, cap as usize);
                let mut n = 0usize;
                while n < slot.len() {
                    match self.reader.read(&mut slot[n..]) {
                        Ok(0) => { self.eof = true; break; }
                        Ok(k) => n += k,
                        Err(ref e) if e.kind() == ErrorKind::Interrupted => {} // src/buffer.rs:85-97
                        Err(e) => return Err(e),
                    }
                }
                fqh_stream_submit(self.st, n as u64, self.eof as c_int);
            }
            Ok(())
        }
    }

    /// One chunk of records: the GPU's replacement for ~hundreds of thousands of
    /// `IdxRecord::from_buffer` calls.  The slices stay valid until the next call.
    pub fn next_chunk(&mut self) -> Result<Option<(&[u8], &[fqh_idx_record], u64, bool)>> {
        self.fill()?;
        unsafe {
            let mut c: fqh_chunk = std::mem::zeroed();
            fqh_stream_release(self.st); // releases the previous chunk, if any
            if fqh_stream_collect(self.st, &mut c) != FQH_OK { return Ok(None); }
            if c.parse_status != FQH_OK {
                // records before the error were delivered with the previous chunks / this index;
                // the error text is the crate's own (fqh_strerror returns the exact strings)
                let msg = std::ffi::CStr::from_ptr(fqh_strerror(c.parse_status)).to_string_lossy().into_owned();
                return Err(Error::new(ErrorKind::InvalidData, msg));
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
