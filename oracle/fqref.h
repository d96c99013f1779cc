/* fqref.h — CPU oracle for the FASTQ record-scan / per-read-statistics path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only as the
 * checker (or as the timed CPU baseline).  The product path (fastq-rs_amd/csrc) never links it.
 *
 * It is a plain-C restatement of the reference crate `fastq` 0.6.0 (aseyboldt/fastq-rs):
 *   src/records.rs:137-163   read_header / read_sep
 *   src/records.rs:201-247   IdxRecord::from_buffer
 *   src/records.rs:65-97     trim_winline + RefRecord accessors
 *   src/records.rs:19-33     validate_dna / validate_dnan
 *   src/buffer.rs:1-112      Buffer (clean / replace_buffer / read_into / consume)
 *   src/lib.rs:221-304       Parser::each / RecordRefIter::advance (streaming driver)
 *   src/lib.rs:364-425       RecordSetIter::next (batch driver)
 *   src/lib.rs:509-565       Parser::parallel_each (round-robin deal of record sets)
 * The reference is Rust; no rustc/cargo exists in the build image, so the reference itself cannot
 * be compiled here.  Parity is pinned by the reference's own 12 unit tests + doc-test
 * (src/lib.rs:611-811, :474-508), committed as tests/golden/reference_unit_tests.json and checked
 * by tests/test_oracle_golden.py.  `memchr` (crates.io, ">=0.1", unpinned) is restated with libc
 * memchr: "index of the first occurrence of a byte".
 */
#ifndef FQREF_H
#define FQREF_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FQREF_BUFSIZE (68u * 1024u) /* src/lib.rs:128-129 */

enum {
    FQREF_OK = 0,
    FQREF_E_HEADER = 1,   /* "Fastq headers must start with '@'"        records.rs:143-146 */
    FQREF_E_SEP = 2,      /* "Sequence and quality not separated by +"  records.rs:157-160 */
    FQREF_E_LEN = 3,      /* "Sequence and quality length mismatch"     records.rs:234-237 */
    FQREF_E_TRUNCATED = 4,/* "Possibly truncated input file"            lib.rs:287-290 (407-410) */
    FQREF_E_TOO_LONG = 5  /* "Fastq record is too long"                 lib.rs:279-282 (399-402) */
};

/* One parsed record, offsets exactly as IdxRecord (records.rs:56-63) but with `start` rebased to
 * the byte offset in the whole input (the reference only knows buffer-relative offsets). */
typedef struct {
    uint64_t start;              /* global byte offset of '@'                          */
    uint64_t head, seq, sep, qual; /* offsets of the 4 '\n' relative to `start`        */
} fqref_idx;

typedef struct {
    int32_t status;       /* FQREF_OK or FQREF_E_*                                              */
    int32_t stopped;      /* 1 if the callback returned 0 (each() -> Ok(false))                 */
    uint64_t n_records;   /* records delivered to the callback before EOF / error / stop        */
    uint64_t bytes_consumed; /* global offset just past the last delivered record               */
} fqref_result;

/* Callback: `rec` points at the record's '@'; offsets as in fqref_idx.  Return 0 to stop. */
typedef int (*fqref_cb)(void *user, const uint8_t *rec, const fqref_idx *idx);

/* Parser::each over an in-memory reader (std::io::Cursor semantics).  bufsize = BUFSIZE
 * (FQREF_BUFSIZE for the real crate, 64 under cfg(fuzzing)); max_read > 0 caps every read() call
 * to emulate short reads (0 = unlimited, i.e. Cursor). */
void fqref_each(const uint8_t *data, uint64_t len, uint64_t bufsize, uint64_t max_read,
                fqref_cb cb, void *user, fqref_result *res);

/* Convenience loops on top of fqref_each. */
/* each_zipped (src/lib.rs:577-609) with a scripted callback: see fqref.c */
void fqref_each_zipped(const uint8_t *d1, uint64_t l1, const uint8_t *d2, uint64_t l2, uint64_t bufsize,
                       const uint8_t *flags, uint64_t nflags, uint64_t *trace, uint64_t cap, uint64_t *ncalls,
                       int32_t fin[2], int32_t *status);
/* the same over a file on disk (one read(2) per Buffer refill): examples/fastq-count.rs; -1 if it cannot be opened */
int fqref_count_file(const char *path, uint64_t bufsize, fqref_result *res);
void fqref_count(const uint8_t *data, uint64_t len, uint64_t bufsize, uint64_t max_read,
                 fqref_result *res);
/* Writes up to cap index entries; res->n_records is the true count. */
void fqref_index(const uint8_t *data, uint64_t len, uint64_t bufsize, uint64_t max_read,
                 fqref_idx *out, uint64_t cap, fqref_result *res);
/* Writes up to cap record-start offsets. */
void fqref_offsets(const uint8_t *data, uint64_t len, uint64_t bufsize, uint64_t max_read,
                   uint64_t *rec_start, uint64_t cap, fqref_result *res);

/* Per-read statistics ("CPU loop over Record::qual/seq"), over delivered records:
 *   qual_hist[p*256 + qual()[p]] += 1      for p < lmax
 *   base_hist[p*8 + cls(seq()[p])] += 1    for p < lmax; cls: A0 C1 G2 T3 N4 other5 (uppercase only)
 * scalars[0]=n_records [1]=n_bases(sum len seq()) [2]=n_qual(sum len qual())
 *        [3]=n_valid_dna [4]=n_valid_dnan [5]=seq bytes at p>=lmax [6]=qual bytes at p>=lmax [7]=0
 * Arrays are ACCUMULATED into (caller zeroes them). */
#define FQREF_NSCALARS 8
void fqref_stats(const uint8_t *data, uint64_t len, uint64_t bufsize, uint64_t max_read,
                 uint32_t lmax, uint64_t *qual_hist, uint64_t *base_hist, uint64_t *scalars,
                 fqref_result *res);

/* Parser::record_sets / parallel_each (lib.rs:364-425, 509-565).  set_sizes receives the number of
 * records of every yielded Ok(RecordSet) in order (first one is always 0), up to cap_sets;
 * *n_sets is the true number.  worker_counts[n_threads] receives the records each worker would see
 * with the round-robin deal of lib.rs:535 (NULL to skip).  res->n_records counts the records in
 * yielded sets only: a set under construction when an error hits is dropped (lib.rs:399-410). */
void fqref_record_sets(const uint8_t *data, uint64_t len, uint64_t bufsize, uint64_t max_read,
                       uint32_t n_threads, uint64_t *set_sizes, uint64_t cap_sets, uint64_t *n_sets,
                       uint64_t *worker_counts, fqref_result *res);

/* Accessors (records.rs:65-97): pointer+length of head()/seq()/qual() for one record. */
void fqref_accessors(const uint8_t *rec, const fqref_idx *idx, uint64_t *head_off,
                     uint64_t *head_len, uint64_t *seq_off, uint64_t *seq_len, uint64_t *qual_off,
                     uint64_t *qual_len);
int fqref_validate_dna(const uint8_t *seq, uint64_t n);  /* records.rs:19-23 */
int fqref_validate_dnan(const uint8_t *seq, uint64_t n); /* records.rs:29-33 */

const char *fqref_strerror(int status);

/* Synthetic 150 bp FASTQ: 330 bytes per record, record i depends only on (seed, i).
 * Fills out[0..len) with bytes [byte_off, byte_off+len) of the infinite synthetic file. */
#define FQREF_SYNTH_RECLEN 330u
#define FQREF_SYNTH_SEED 0x5EEDF00D2026ull
void fqref_synth_range(uint8_t *out, uint64_t byte_off, uint64_t len, uint64_t seed);

#ifdef __cplusplus
}
#endif
#endif
