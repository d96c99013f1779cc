/* fastq_hip.h — C ABI of libfastq_hip.so: the MI355X (gfx950) FASTQ record-scan and per-read
 * statistics path that sits behind the `fastq` crate's Parser / Record / parallel_each surface.
 *
 * The reference crate (aseyboldt/fastq-rs, fastq 0.6.0) has no FFI of its own; this header is the
 * seam a Rust (or C/C++/Python) host binds.  Every entry point names the reference interface it
 * replaces.  Plain C: opaque handle, plain pointers and sizes, status codes, no exceptions, no
 * torch types.  Device pointers are raw HIP device addresses (hipMalloc / torch `data_ptr()`).
 *
 *   reference (CPU, one record at a time)                        this ABI (GPU, whole buffers)
 *   ------------------------------------------------------------ ------------------------------
 *   IdxRecord::from_buffer          src/records.rs:201-247        fqh_scan
 *   read_header / read_sep          src/records.rs:137-163        fqh_scan (validation keys)
 *   memchr('\n')                    src/records.rs:141,155,214,228 fqh_scan (byte-scan kernel)
 *   RecordRefIter::advance / each   src/lib.rs:221-304            fqh_scan + fqh_summary
 *   RecordSetIter::next             src/lib.rs:364-425            fqh_scan (record offsets) +
 *                                                                 fqh_index_records
 *   loop over Record::seq()/qual()  src/records.rs:75-90 (a8)     fqh_stats, fqh_scan_stats (one read)
 *   validate_dna / validate_dnan    src/records.rs:19-33          fqh_stats (scalars 3,4), fqh_record_flags
 *   read lengths (sum of seq().len()) fuzz/fuzz_targets/fuzz_target_1.rs:16  fqh_len_hist
 *   Record::write (filter loops)    src/records.rs:93-96          fqh_gather_records
 *   Buffer                          src/buffer.rs:1-112           fqh_stream_* (pinned ring)
 *   thread_reader                   src/thread_reader.rs:182-200  fqh_stream_* (copy stream)
 *   ... its copy box -> buffer      src/thread_reader.rs:90-97    removed: fqh_stream_submit_external (DMA from the host's memory)
 *   ... its recycled boxes          src/thread_reader.rs:60-75    FQH_OPT_KEEP_RING
 *   RecordSet owning its buffer     src/lib.rs:306-318, 384-385   fqh_stream_release_chunk (sets borrow ring slots)
 *   parallel_each, one worker / GPU src/lib.rs:509-565            fqh_shard_stream_run[_mapped] / _finish / _outcome
 *   parallel_each gather            src/lib.rs:553-559            fqh_allgather / fqh_allreduce_u64 (RCCL)
 */
#ifndef FASTQ_HIP_H
#define FASTQ_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* libfastq_hip.so is built with -fvisibility=hidden; the functions declared between here and the matching pop are its whole
 * dynamic symbol table (tests/test_abi.py holds `nm -D` against this header).  A host includes the header without the macro. */
#ifdef FQH_BUILDING_LIBRARY
#pragma GCC visibility push(default)
#endif

#define FQH_ABI_VERSION 1
#define FQH_BUFSIZE (68u * 1024u) /* BUFSIZE, src/lib.rs:128-129 */

typedef struct fqh_ctx fqh_ctx; /* one per device and host thread; not thread-safe */

typedef enum {
    FQH_OK = 0,
    FQH_E_HEADER = 1,       /* "Fastq headers must start with '@'"        src/records.rs:143-146 */
    FQH_E_SEP = 2,          /* "Sequence and quality not separated by +"  src/records.rs:157-160 */
    FQH_E_LEN_MISMATCH = 3, /* "Sequence and quality length mismatch"     src/records.rs:234-237 */
    FQH_E_TRUNCATED = 4,    /* "Possibly truncated input file"            src/lib.rs:287-290     */
    FQH_E_TOO_LONG = 5,     /* "Fastq record is too long"                 src/lib.rs:279-282     */
    FQH_E_IO = 6,
    FQH_E_DEVICE = 7,       /* HIP runtime error; fqh_last_error() has the text */
    FQH_E_ARG = 8,
    FQH_E_CAPACITY = 9,     /* rec_start / index capacity too small; summary.n_records is exact */
    FQH_E_AGAIN = 10        /* fqh_scan_finish after fqh_shard_rescan_launch: some shard left the fast path; take the host recipe.
                               fqh_stream_collect: the next slot of the ring is still held (fqh_stream_release_chunk), call again */
} fqh_status;

/* Parser state at a byte boundary of the input: everything a scan of the NEXT chunk needs to be
 * bit-exact without seeing earlier bytes.  All-zero = start of file.  (The reference keeps the same
 * information implicitly in its 68 KiB window: src/lib.rs:255-303.) */
typedef struct {
    uint64_t base_offset; /* file offset of the chunk's first byte — the TRUE one whenever bufsize != 0: "Fastq record is
                             too long" is judged on the record's file offset mod 16 (fqh_set_bufsize), for every chunk of a
                             chain, not only the last; a driver that numbers its chunks from 0 sets bufsize 0 and applies
                             the rule itself                                                     */
    uint64_t nl_count;    /* '\n' seen before the chunk (nl_count % 4 = line phase, / 4 = record) */
    uint64_t back[4];     /* back[i] = chunk_start - (i-th most recent line start <= chunk_start);
                             back[0] is the column of the chunk's first byte; starts older than
                             the file start are clamped to the file start                        */
} fqh_carry;

typedef struct {
    uint64_t n_records;      /* records that END in this chunk, before the first error            */
    uint64_t bytes_consumed; /* chunk-relative offset just past the last of them (0 if none)      */
    int32_t parse_status;    /* FQH_OK or FQH_E_HEADER..FQH_E_TOO_LONG (is_final decides EOF rule)*/
    int32_t reserved;
    uint64_t err_record;     /* global index of the failing record (nl_count/4-based)             */
    uint64_t err_offset;     /* file offset of the failing record's first byte                    */
    uint64_t n_newlines;     /* '\n' in this chunk                                                */
    uint64_t tail_len;       /* bytes after the last complete record (carried when !is_final)     */
    uint64_t max_record_len; /* longest record ending in this chunk (bytes, incl. last '\n')      */
    uint64_t n_line_starts;  /* line starts at chunk offsets 1..len (len counts if the last byte is
                                '\n'): how many entries of carry_out.back[] lie inside this chunk  */
} fqh_summary;

/* One record of the index, the GPU-side counterpart of IdxRecord (src/records.rs:56-63):
 * start = file offset of '@'; head/seq/sep/qual = offsets of the four '\n' relative to start. */
typedef struct {
    uint64_t start;
    uint32_t head, seq, sep, qual;
} fqh_idx_record;

#define FQH_NSCALARS 8
/* d_scalars layout of fqh_stats: [0] n_records [1] n_bases = sum len(seq()) [2] sum len(qual())
 * [3] n_valid_dna [4] n_valid_dnan [5] seq bytes at position >= lmax [6] qual bytes >= lmax [7] 0 */

fqh_status fqh_create(int device, fqh_ctx **out);
void fqh_destroy(fqh_ctx *ctx);
const char *fqh_strerror(fqh_status s); /* the reference's exact message strings for 1..5 */
const char *fqh_last_error(fqh_ctx *ctx);
int fqh_abi_version(void);

/* Launch on a caller-owned hipStream_t (e.g. torch's current stream); NULL = the context's own.  The context's OWN stream is a
 * blocking stream: every call on it is ordered, in both directions, against whatever the process does on the legacy null
 * stream — safe for a host that never thinks about streams, and a hidden coupling for one that overlaps other null-stream work
 * with its scans: such a host passes a stream of its own.  Device memory handed to a
 * call must be ready ON THAT STREAM: the context's own stream is a blocking one, i.e. ordered against work on the legacy null
 * stream (hipMemsetAsync(.., 0), torch's default stream) and against nothing else; work on any other stream needs an event or
 * a synchronize before the call, as for any kernel launch. */
fqh_status fqh_set_stream(fqh_ctx *ctx, void *hip_stream);
/* BUFSIZE used for the "record is too long" rule (default FQH_BUFSIZE; 64 = cfg(fuzzing),
 * src/lib.rs:126-127; 0 = no limit).  The reference's verdict on a record depends on the record's FILE OFFSET mod 16 and on
 * nothing else (the closed form of src/buffer.rs's arithmetic for a reader that fills every read, csrc/replay.h): whole files,
 * chunks chained with fqh_carry (base_offset is the chunk's file offset), ring slots and byte-range shards are all judged on
 * file offsets and give Parser::each's answer. */
fqh_status fqh_set_bufsize(fqh_ctx *ctx, uint64_t bufsize);

/* Knobs (defaults in brackets; the environment variables FQH_SPEC / FQH_FUSED set the same at fqh_create):
 *   FQH_OPT_FAST_PATH [1]   fqh_scan first tries to PROVE the input valid with a quarter of the index traffic
 *                            and reruns the exact path on any doubt; 0 = exact path only.  Setting it
 *                           also clears the back-off the context keeps after failed attempts.
 *   FQH_OPT_SINGLE_PASS [1] whole-file fqh_stats / fqh_scan_stats count in the scan's own pass over the input
 *                           (k_scan_stats); 0 = always the exact scan followed by the histogram kernel.  A pass that has to
 *                           be given up (lines of 512 bytes and more among shorter ones, more lines with bytes outside the
 *                           alphabets than its dump area holds) is counted over the exact index instead, bit-exact, and the context's
 *                           next 1, 2, 4 .. 64 statistics calls go there directly; setting the option forgets that back-off.
 *                           lmax is the caller's choice and may be far below the reads' length (the first 150 cycles of
 *                           kilobase reads): the columns beyond it are looked at (n_valid_dna / n_valid_dnan cover every
 *                           base), not counted.  It may as well be far above it (1000 rows, whatever comes): the pass keeps
 *                           the rows the READS need — a look at four 64 KiB windows of every NEW input (its first bytes and three
 *                           more, a quarter of it apart; the same buffer again and the next chunk of the same file are not
 *                           new), then what the calls on that input find; setting the option forgets that as well — and any
 *                           lmax is counted exactly.  The back-off after a pass that was given up belongs to that input too.
 *   FQH_OPT_PLACE_TRIES [0] where the fast path's per-tile lines (1.6 % of the input size) land in device memory can decide
 *                           whether the byte scan runs at 2.65-2.70 or at 2.85-2.95 ms per 16 GiB: the same allocation call
 *                           gives either kind, and the kind stays with the allocation.  With a value of 2..8
 *                           the first scan of 1 GiB or more on a context's fast path times the index kernel on the first
 *                           GiBs of the caller's input, once without its line stores and then with candidate line buffers
 *                           (at most this many, about 1 ms each, once per context; that first *_launch call blocks while it
 *                           does) until one costs no more than 3.5 % on top of the store-less run; the fastest is kept, each
 *                           loser is freed as soon as it has lost.  Off by default: on some boxes every allocation is of one
 *                           kind and the search buys nothing.  fqh_placement reports what a search found.
 *   FQH_OPT_REUSE_INDEX [0] 1: a fqh_stats* call on the buffer, length and carry of the last finished fqh_scan counts over that scan's
 *                           tile index instead of scanning again (two-pass route only).  The caller thereby VOUCHES that the bytes
 *                           have not changed since the scan: pointer identity proves nothing about a buffer the caller's own
 *                           kernels may have rewritten.  Off: every statistics call reads its input itself.
 *   FQH_OPT_ADAPT_LINES [3] whether the byte scan of 16 GiB takes 2.65 or 2.83 ms is a property of the PAIR (the caller's input
 *                           allocation, the context's allocation of the fast path's per-tile lines): the same line buffer is of the
 *                           fast kind for one input and of the slow kind for another.  A context therefore keeps up to
 *                           TWO line buffers (1.6 % of the input size each) and learns, for each input of 2 GiB or more that it is
 *                           given AGAIN (same address, length and kind of call; four inputs are remembered), which one that input
 *                           runs faster with — from the HIP-event time of the real calls, one buffer per call, no extra launches:
 *                           calls 1 and 2 store to the first, calls 3 and 4 to a second one (allocated then), later calls to the faster (of each pair of
 *                           measurements the faster one counts); an
 *                           alternate that measures like the first (within 2.5 %) is given back and another is tried, at most this
 *                           many times per input.  0 = off.  Results never depend on the choice.
 *   FQH_OPT_OWN_STREAM_NONBLOCKING [0]  1: the context's own stream (used while fqh_set_stream has not given it another) is
 *                           created again as a NON-blocking stream: no implicit ordering against the process's legacy
 *                           null-stream work in either direction — for hosts that overlap other null-stream work with their
 *                           scans and order what they hand in with events.  0: the blocking stream of fqh_create.  Not while a
 *                           launch is pending.
 *   FQH_OPT_KEEP_RING [0]   1: fqh_stream_destroy leaves the ring's pinned slots and their device twins with the context, and the
 *                           next fqh_stream_create of the same geometry (slot_bytes, flags' slot layout) takes them instead of
 *                           pinning memory again (3 x 255 MiB: 70 ms) — for hosts that open one ring per file, and for
 *                           fqh_shard_stream_run, which opens one per call (the reference recycles its two 4 MiB boxes the same
 *                           way, src/thread_reader.rs:60-75).  One geometry is kept, the biggest; 0 (or fqh_destroy) frees it.
 *   FQH_OPT_SPIN_WAIT [0]   microseconds fqh_*_finish polls the stream before it sleeps on it (hipStreamSynchronize wakes up
 *                           ~15 us after the last kernel); a host core spinning inside a library call is the caller's choice.
 * fqh_last_scan_fast: did the last finished scan (or single-pass statistics call) keep the fast path's result (1), or
 * was it rerun on the exact path (0)?  Results are identical either way; this is for benchmarks and tests. */
#define FQH_OPT_FAST_PATH 1
#define FQH_OPT_SINGLE_PASS 2
#define FQH_OPT_PLACE_TRIES 3
#define FQH_OPT_SPIN_WAIT 4
#define FQH_OPT_REUSE_INDEX 5
#define FQH_OPT_ADAPT_LINES 6
#define FQH_OPT_OWN_STREAM_NONBLOCKING 7
#define FQH_OPT_KEEP_RING 8
fqh_status fqh_set_option(fqh_ctx *ctx, int option, int value);
int fqh_last_scan_fast(fqh_ctx *ctx);
/* How the last finished statistics call (fqh_stats*, fqh_scan_stats*) counted: 1 = in the scan's own pass over the input
 * (k_scan_stats); 2 = the same, and the lines that pass does not count itself — batches of eight with a byte outside ACGTN
 * or '!'..'`', lines longer than the pass's rows — were counted one by one behind it (a few KiB re-read); 0 = in a second pass over the
 * input (reads of 512 bases and more — whatever lmax is —, a parse error, or more such lines than one per 512 KiB).  Results are
 * identical; for benchmarks and tests.  A count the single pass declines is no doubt about the parse: it neither reruns the
 * scan nor touches the fast path's back-off. */
int fqh_last_stats_route(fqh_ctx *ctx);
/* What the placement search of this context (FQH_OPT_PLACE_TRIES) measured: *n_candidates line buffers tried (0: no search
 * ran), ms[0 .. n-1] the index kernel's time on the sample with each, ms[8] the kept one's, ms[9] the same kernel without
 * its line stores (the yardstick).  For benchmarks: says whether the search engaged and what it bought. */
fqh_status fqh_placement(fqh_ctx *ctx, int *n_candidates, float ms[10]);
/* What FQH_OPT_ADAPT_LINES holds right now: *n_alive line buffers (the one in use, the alternate, alternates held back until the
 * input they were tried for is settled — never more than 2 + the option's value), *bytes of device memory in all of them, and
 * *n_unsettled remembered inputs that have not been through their measurements yet (0: every later call on those inputs stores
 * to its chosen buffer and allocates nothing).  A benchmark warms up until n_unsettled is 0; a test holds n_alive and the
 * device's free memory against repeated calls. */
fqh_status fqh_line_buffers(fqh_ctx *ctx, int *n_alive, int *n_unsettled, uint64_t *bytes);

/* Record scan.  d_buf[0..len) are device-resident bytes; `in` (NULL = start of file) describes
 * where in the file they sit.  Writes d_rec_start[0..n_records]: [0] = file offset of the record in
 * progress at the chunk start, [i] = file offset just past the i-th record that ends in this chunk
 * (so record i of the chunk is [d_rec_start[i], d_rec_start[i+1])).  d_rec_start may be NULL
 * (count + validate only); cap = its capacity in elements.  Blocking: returns with `out` filled. */
fqh_status fqh_scan(fqh_ctx *ctx, const uint8_t *d_buf, uint64_t len, int is_final,
                    const fqh_carry *in, uint64_t *d_rec_start, uint64_t cap, fqh_summary *out,
                    fqh_carry *carry_out);

/* Same scan split in two so a host can overlap the launch with other work (and so a benchmark can
 * time the kernels alone): fqh_scan_launch enqueues every kernel and returns; fqh_scan_finish
 * waits, resolves the summary.  Exactly one finish per launch. */
fqh_status fqh_scan_launch(fqh_ctx *ctx, const uint8_t *d_buf, uint64_t len, int is_final,
                           const fqh_carry *in, uint64_t *d_rec_start, uint64_t cap);
fqh_status fqh_scan_finish(fqh_ctx *ctx, fqh_summary *out, fqh_carry *carry_out);

/* Byte-range sharding: the byte scan of a shard needs nothing from other shards, only
 * the emit step needs the carry.  So every rank first scans its shard as if it began the file
 * (fqh_scan with in = NULL, is_final = 0, d_rec_start = NULL), the ranks exchange
 * (len, summary.n_newlines, summary.n_line_starts, carry_out.back[4]) — 7 words — and fold them in
 * rank order with fqh_carry_combine to get each shard's true carry-in; fqh_rescan_launch then redoes
 * only the cheap emit/validate step on the tile index the first call left in the context (the
 * buffer must be unchanged).  Finish with fqh_scan_finish. */
/* Step 1 of that recipe without the (wasted) emit pass: byte scan + tile prefix only.  Leaves the
 * tile index in the context for fqh_rescan_launch.  back_zero_carry[i] = len - (i-th most recent line
 * start inside the shard), meaningful for i < n_line_starts. */
fqh_status fqh_shard_prescan(fqh_ctx *ctx, const uint8_t *d_buf, uint64_t len, uint64_t *n_newlines,
                             uint64_t *n_line_starts, uint64_t back_zero_carry[4]);
fqh_status fqh_carry_combine(const fqh_carry *prev, uint64_t len, uint64_t n_newlines,
                             uint64_t n_line_starts, const uint64_t back_zero_carry[4],
                             fqh_carry *next);
fqh_status fqh_rescan_launch(fqh_ctx *ctx, int is_final, const fqh_carry *in, uint64_t *d_rec_start,
                             uint64_t cap);
/* The same recipe with the exchange ON THE DEVICE — no host hop between the byte scan and the emit step (the host
 * recipe costs two: 0.14 ms of a 3 ms step).  fqh_shard_prescan_launch enqueues step 1 and writes this rank's
 * FQH_SHARD_WORDS words (len, newlines, line starts, back_zero_carry[4], "left the fast path") to d_words; the host
 * enqueues an all-gather of them on the same stream (fqh_allgather, or any collective library on the context's
 * stream); fqh_shard_rescan_launch enqueues the fold of the rows in front of `rank` (fqh_carry_combine, on the device),
 * the emit / validate step under the carry it gives, and — if d_counts is not NULL — writes (records, 1 if this
 * shard has a parse error or the recipe cannot be used) there for the sum over the ranks; fqh_scan_finish ends the
 * launch as usual (summary, carry-out).  fqh_scan_finish — not the launch calls, which only enqueue — returns FQH_E_AGAIN
 * on EVERY rank when some rank's byte scan could not keep the fast path (its words are not to be used): the ranks then run
 * the host recipe above.  fqh_shard_rescan_launch only continues a fqh_shard_prescan_launch (FQH_E_ARG otherwise).  A parse error inside
 * a shard is reported by that shard's fqh_scan_finish as usual; the word d_counts[1] tells the other ranks. */
#define FQH_SHARD_WORDS 8
fqh_status fqh_shard_prescan_launch(fqh_ctx *ctx, const uint8_t *d_buf, uint64_t len, uint64_t *d_words);
fqh_status fqh_shard_rescan_launch(fqh_ctx *ctx, int is_final, const uint64_t *d_all_words, int n_ranks, int rank,
                                   uint64_t *d_rec_start, uint64_t cap, uint64_t *d_counts);
/* Where does the first record of a byte-range shard begin, and at which line phase does the shard start?  For hosts
 * that cannot wait for the shards in front of theirs (the host-streamed sharded mode, BASELINE configs[4]: each rank
 * streams its own range and the ranks talk once, at the end).  d_buf[0..len) is a window at the shard's start (a few
 * MiB; the byte before it is a newline or not: prev_is_newline).  The four possible phases (newlines in front of
 * the shard, mod 4) are tried on the window's tile index; the record in progress at the window's start is left
 * unvalidated (it belongs to the stitch with the previous shard), every other record of the window is validated in
 * the reference's order (src/records.rs:201-247).  *phase = the one phase under which the window parses,
 * *first_record_offset = where the record in progress ends (0 if the shard begins with a record).  FQH_E_HEADER: no
 * phase validates; FQH_E_ARG: several do (window too small to tell).  The ranks confirm the phase at the end against
 * the true newline count of the shards in front of them (fqh_carry.nl_count of their streams, fqh_stream_carry). */
fqh_status fqh_shard_align(fqh_ctx *ctx, const uint8_t *d_buf, uint64_t len, int prev_is_newline, uint32_t *phase,
                           uint64_t *first_record_offset);
/* The exchange steps of the sharded modes, over RCCL (xGMI) on the context's stream — for hosts without a collective
 * library of their own (a Rust or C++ driver; bench.py uses torch.distributed, which is the same RCCL).  librccl.so is
 * bound at run time by the first of these calls.  One rank makes an id (fqh_comm_unique_id) and hands it to the
 * others by any means; every rank then calls fqh_comm_create on its own context (= its GPU).  fqh_allgather: every
 * rank contributes bytes_per_rank bytes (the 7 carry words of fqh_shard_prescan, the 8 words + tail of a streamed
 * shard), d_recv gets all of them in rank order.  fqh_allreduce_u64: element-wise sum in place (counts, scalars,
 * histograms) — the gather at the end of Parser::parallel_each, src/lib.rs:553-559.  Both are enqueued; fqh_sync waits
 * for the context's stream. */
typedef struct fqh_comm fqh_comm;
#define FQH_COMM_ID_BYTES 128
fqh_status fqh_comm_unique_id(uint8_t id[FQH_COMM_ID_BYTES]);
fqh_status fqh_comm_create(fqh_ctx *ctx, int n_ranks, int rank, const uint8_t id[FQH_COMM_ID_BYTES], fqh_comm **out);
void fqh_comm_destroy(fqh_comm *comm);
fqh_status fqh_allgather(fqh_ctx *ctx, fqh_comm *comm, const void *d_send, void *d_recv, uint64_t bytes_per_rank);
fqh_status fqh_allreduce_u64(fqh_ctx *ctx, fqh_comm *comm, uint64_t *d_buf, uint64_t n);
/* Element-wise MINIMUM in place (ncclMin): the global first error of a sharded parse is the minimum over the ranks' packed
 * (record, kind) keys — what Parser::parallel_each returns when the parse fails, src/lib.rs:544-547, 561-564. */
fqh_status fqh_allreduce_min_u64(fqh_ctx *ctx, fqh_comm *comm, uint64_t *d_buf, uint64_t n);
fqh_status fqh_sync(fqh_ctx *ctx);

/* ---- The sharded, host-streamed mode (BASELINE configs[4]) -----------------------------------------------------------
 * One rank = one GPU = one pinned ring; the file is cut at arbitrary byte offsets and every rank streams its own range
 * [lo, hi) PHASE-FREE (it cannot wait for the ranks in front of it); the ranks talk once, at the end.  The reference's
 * analogue is Parser::parallel_each over a thread_reader pipeline: per-worker results gathered at the end
 * (src/lib.rs:553-559), a parse error returned for the whole call (src/lib.rs:544-547, 561-564).  The outcome — status and
 * number of records delivered before the first error — is EXACTLY Parser::each's over the same bytes, whatever the cuts
 * (inside lines, sixteen bytes apart, empty ranges, a file of three lines on eight ranks), and so are the histograms of a file
 * that parses.  (Of a file that does NOT parse the summed histograms also hold records behind the error, counted by the ranks
 * behind it: discard them, as parallel_each discards its workers' results when the parse fails, src/lib.rs:561-564.)
 *
 * fqh_shard_stream_run: rank r > 0 first settles its line phase and the offset R of its first record on the 4 MiB of the file
 * behind lo (fqh_shard_align; it may look past hi: a range of a few lines settles nothing by itself), then streams
 * [lo + R, hi) through a ring of n_slots x slot_bytes like a file of its own that begins at file offset lo + R, calling
 * `read(user, h_dst, file_offset, nbytes)` (0 = ok) to fill pinned memory — a pread, a memcpy, a decompressor with an index.
 * lmax != 0: every record the rank delivers is added to the histograms (as fqh_stats); the three arrays must be this rank's
 * own for this call (zeroed by the caller, summed over the ranks afterwards).  *res: the rank's summary.  A parse error inside
 * the rank's records is res->status, not the return value.  A range that holds no record start (it lies inside one record:
 * FQH_SHARD_PASS), or whose window does not single out one line phase (several fit a few lines at the end of the file; none
 * fits a window that begins with a parse error: FQH_SHARD_DEFER), streams nothing, counts its newlines and leaves its bytes to
 * the rank that parses the gap it lies in (below); an EMPTY range (lo == hi) is fine.  No byte range is refused.
 *
 * Then ONE exchange: fqh_shard_result_words(res, lo, hi) — FQH_SHARD_STREAM_WORDS words — of every rank, all-gathered in rank
 * order (fqh_allgather or the host's own collective).  A rank whose run FAILED (a device error, an exception of the host
 * around the call) must still take part — the others would wait for it forever: it sends fqh_shard_failed_words(status, lo,
 * hi) and goes on.  Bytes the `read` callback cannot deliver do NOT fail the run (res->flags bit 1, below): the error the
 * whole call ends in is the one the sequential reader would meet first, a parse error in front of the unreadable bytes or
 * FQH_E_IO at them.
 *
 * fqh_shard_stream_finish: every rank derives the same picture from the words — the TRUE newline count in front of every
 * range, hence which ranks parsed under the true line phase.  Between the last complete record of one such rank and the first
 * record of the next lies a GAP: the record that straddles the cut, and every PASS / DEFER / wrongly-phased range in between.
 * The rank a gap ends at parses it through its own `read` callback — a sequential parse from a true record start over the
 * file's bytes, on the GPU, which must land on its first record; what lies behind the last such rank is parsed to the end of
 * the file by the last rank that holds bytes.  (A rank that parsed under a wrong phase contributes nothing; finish zeroes its
 * histogram arrays.)  out[0] = records this rank contributes (gap + streamed), out[1] = its first error as a packed key
 * (file offset of the failing record << 11 | rank << 3 | kind), or FQH_NO_ERROR_KEY.  ctx and read may be NULL for a rank
 * without a gap to parse.  If finish itself fails on a rank, that rank goes on with out[1] = fqh_shard_failure_key(rank, lo, status).
 *
 * One SUM over [records_per_rank[n_ranks] (rank r puts out[0] into slot r), scalars, histograms] (fqh_allreduce_u64) and one
 * MIN over out[1] (fqh_allreduce_min_u64) give every rank the totals, or the first error in FILE order;
 * fqh_shard_stream_outcome turns the two into Parser::each's result: *status, *n_records = records delivered before the error
 * (the slots up to the failing rank's; all of them when the parse succeeded), *err_offset = where the failing record starts. */
typedef int (*fqh_read_fn)(void *user, uint8_t *h_dst, uint64_t file_offset, uint64_t nbytes);
typedef struct {
    int32_t status;      /* first parse error among the rank's own streamed records (FQH_OK: none)            */
    uint32_t phase;      /* newlines in front of the range, mod 4, as settled on its window; or FQH_SHARD_*    */
    uint64_t n_records;  /* records delivered (before the first error)                                        */
    uint64_t n_newlines; /* '\n' in [lo, hi)                                                                  */
    uint64_t err_offset; /* file offset of the failing record                                                 */
    uint64_t head_len;   /* R: [lo, lo + R) ends the record the ranks in front began                           */
    uint64_t tail_len;   /* bytes behind the rank's last complete record                                      */
    uint64_t flags;      /* bit 0: n_newlines stops where the stream stopped (an error in a range of many MiB, unreadable bytes);
                            bit 1: the read callback failed for bytes of this range.  Not a failure of the run: the range
                            contributes nothing by itself (phase FQH_SHARD_DEFER), and the rank that parses the gap it lies in
                            reads its bytes again in file order — it reports a parse error in front of the unreadable bytes,
                            as the sequential reader would, FQH_E_IO when it meets them, and nothing if ITS callback can read
                            them */
} fqh_shard_result;
#define FQH_SHARD_STREAM_WORDS 10
#define FQH_SHARD_MAX_RANKS 256
#define FQH_NO_ERROR_KEY UINT64_MAX
#define FQH_SHARD_EMPTY 0xFFFFFFFFu /* fqh_shard_result.phase of an empty byte range (lo == hi) */
#define FQH_SHARD_PASS 0xFFFFFFFEu  /* ... of a byte range that holds NO record start (it lies inside one record) */
#define FQH_SHARD_DEFER 0xFFFFFFFDu /* ... of a byte range whose window does not single out a line phase: parsed after the exchange */
fqh_status fqh_shard_stream_run(fqh_ctx *ctx, fqh_read_fn read, void *user, uint64_t lo, uint64_t hi, uint64_t file_len,
                                uint64_t slot_bytes, uint32_t n_slots, uint32_t lmax, uint64_t *d_qual_hist,
                                uint64_t *d_base_hist, uint64_t *d_scalars, fqh_shard_result *res);
/* The same with the range's bytes taken IN PLACE from page-locked host memory (fqh_stream_submit_external): for the bytes it
 * streams, the run asks `map(user, file_offset, want, &avail)` for a pointer to the file's bytes [file_offset, file_offset +
 * avail), 1 <= avail (more or less than `want` — a slot takes min(slot_bytes, avail)); NULL = these bytes cannot be had (as a
 * failing read).  The ring then has no pinned data slots at all (FQH_STREAM_EXTERNAL).  `read` still serves the 4 MiB alignment
 * window, newline counts of ranges that stream nothing, and the gap parse of fqh_shard_stream_finish.  map == NULL: exactly
 * fqh_shard_stream_run. */
typedef const uint8_t *(*fqh_map_fn)(void *user, uint64_t file_offset, uint64_t want, uint64_t *avail);
fqh_status fqh_shard_stream_run_mapped(fqh_ctx *ctx, fqh_read_fn read, fqh_map_fn map, void *user, uint64_t lo, uint64_t hi,
                                       uint64_t file_len, uint64_t slot_bytes, uint32_t n_slots, uint32_t lmax,
                                       uint64_t *d_qual_hist, uint64_t *d_base_hist, uint64_t *d_scalars, fqh_shard_result *res);
void fqh_shard_result_words(const fqh_shard_result *res, uint64_t lo, uint64_t hi, uint64_t words[FQH_SHARD_STREAM_WORDS]);
void fqh_shard_failed_words(fqh_status why, uint64_t lo, uint64_t hi, uint64_t words[FQH_SHARD_STREAM_WORDS]);
fqh_status fqh_shard_stream_finish(fqh_ctx *ctx, fqh_read_fn read, void *user, uint64_t file_len, const uint64_t *h_all_words,
                                   int n_ranks, int rank, uint64_t slot_bytes, uint32_t n_slots, uint32_t lmax,
                                   uint64_t *d_qual_hist, uint64_t *d_base_hist, uint64_t *d_scalars, uint64_t out[2]);
uint64_t fqh_shard_failure_key(int rank, uint64_t offset, fqh_status why);
fqh_status fqh_shard_stream_outcome(uint64_t min_key, const uint64_t *records_per_rank, int n_ranks, int32_t *status,
                                    uint64_t *n_records, uint64_t *err_offset);

/* Forget the cached tile index.  The index describes the BYTES of the last scanned buffer; fqh_memcpy_h2d, fqh_memset and
 * fqh_synth_fill drop it themselves when they write into that buffer, writes the library cannot see (the caller's
 * own kernels, an allocator that hands the same address out again) need this call before fqh_stats /
 * fqh_index_records / fqh_rescan_launch on the same (pointer, length, carry). */
fqh_status fqh_invalidate(fqh_ctx *ctx);

/* Full IdxRecord-style index of the records found by the LAST fqh_scan on this context (same
 * d_buf): d_index[0..n) with n = min(n_records, cap). */
fqh_status fqh_index_records(fqh_ctx *ctx, fqh_idx_record *d_index, uint64_t cap);

/* Per-position statistics over the records the scan delivers (everything before the first
 * error): d_qual_hist[p*256 + qual()[p]] and d_base_hist[p*8 + class(seq()[p])] for p < lmax
 * (classes A0 C1 G2 T3 N4 other5), d_scalars as above.  All are u64 device arrays that are ADDED
 * to (zero them first).  Reads its input itself; only with FQH_OPT_REUSE_INDEX set (the caller vouches
 * that the bytes are unchanged) does a call on the (d_buf, len, carry) of the last finished fqh_scan count over that
 * scan's tile index.  The *_launch forms return without waiting for the device — with one exception: the first statistics
 * call on a NEW input (another buffer, length or file offset than the context's last statistics call; and a later one over a GiB
 * or more whose rows might be too few) looks at four 64 KiB windows of the input to size the pass by the reads
 * (FQH_OPT_SINGLE_PASS above) and waits for that look, i.e. for what the context's stream holds in front of it, once. */
fqh_status fqh_stats(fqh_ctx *ctx, const uint8_t *d_buf, uint64_t len, int is_final,
                     const fqh_carry *in, uint32_t lmax, uint64_t *d_qual_hist,
                     uint64_t *d_base_hist, uint64_t *d_scalars, fqh_summary *out,
                     fqh_carry *carry_out);
fqh_status fqh_stats_launch(fqh_ctx *ctx, const uint8_t *d_buf, uint64_t len, int is_final,
                            const fqh_carry *in, uint32_t lmax, uint64_t *d_qual_hist,
                            uint64_t *d_base_hist, uint64_t *d_scalars);
fqh_status fqh_stats_finish(fqh_ctx *ctx, fqh_summary *out, fqh_carry *carry_out);
/* Record scan AND statistics of one buffer in a single call — for reads of up to 511 bases, whatever lmax is, in a single READ of the input: one kernel scans,
 * validates and counts (k_scan_stats), the way the reference's Parser::each hands each record to the closure that reads
 * seq()/qual() (src/lib.rs:226-237); whole files, chunks with a carry and chunks that are not the file's last alike.  Outputs as
 * fqh_scan (d_rec_start may be NULL) plus fqh_stats.  fqh_stats on its own takes the same single-pass route.  Lines with bytes
 * outside ACGTN / '!'..'`' and lines longer than the pass's rows are counted one by one behind that pass (fqh_last_stats_route() == 2).
 * Reads of 512 bases and more, or more such lines than one per 512 KiB, send the HISTOGRAMS to a second pass over
 * the input (the scan's result stands); an input the fast path cannot prove valid (any parse error) runs the exact scan
 * followed by the histogram kernel; results are identical either way.  FQH_E_CAPACITY (d_rec_start shorter than
 * n_records + 1) is reported by the blocking call / the finish on either route, with the summary, the carry-out and the
 * histograms complete, exactly as fqh_scan reports it. */
fqh_status fqh_scan_stats(fqh_ctx *ctx, const uint8_t *d_buf, uint64_t len, int is_final, const fqh_carry *in,
                          uint64_t *d_rec_start, uint64_t cap, uint32_t lmax, uint64_t *d_qual_hist,
                          uint64_t *d_base_hist, uint64_t *d_scalars, fqh_summary *out,
                          fqh_carry *carry_out);
fqh_status fqh_scan_stats_launch(fqh_ctx *ctx, const uint8_t *d_buf, uint64_t len, int is_final,
                                 const fqh_carry *in, uint64_t *d_rec_start, uint64_t cap, uint32_t lmax,
                                 uint64_t *d_qual_hist, uint64_t *d_base_hist, uint64_t *d_scalars);
fqh_status fqh_scan_stats_finish(fqh_ctx *ctx, fqh_summary *out, fqh_carry *carry_out);
/* As fqh_stats_launch, for a chunk whose buffer also holds the beginning of the record in progress
 * at the chunk start: d_buf[-lead_len .. -1] is valid device memory and ends with the bytes of that
 * record that precede the chunk (in->back[in->nl_count & 3] of them; the streaming ring and a
 * Buffer-style compaction, src/buffer.rs:51-72, both keep them there).  The record is then counted
 * with the chunk it ends in, so the calls over consecutive chunks of a file add up to exactly the
 * whole-file histograms.  Without enough lead the record is left out, as in fqh_stats_launch. */
fqh_status fqh_stats_launch_lead(fqh_ctx *ctx, const uint8_t *d_buf, uint64_t len, uint64_t lead_len,
                                 int is_final, const fqh_carry *in, uint32_t lmax,
                                 uint64_t *d_qual_hist, uint64_t *d_base_hist, uint64_t *d_scalars);

/* Read-length histogram (the optional len_hist[len(seq())] of the statistics contract; the consumer it
 * stands in for sums rec.seq().len(), fuzz/fuzz_targets/fuzz_target_1.rs:16).  Derived on the device from what fqh_stats /
 * fqh_scan_stats left in d_base_hist and d_scalars over the SAME records: column p of the base histogram holds one count
 * per read longer than p, so d_len_hist[L] = reads with len(seq()) == L for L < lmax, and d_len_hist[lmax] = reads of
 * lmax bases or more.  d_len_hist (lmax + 1 u64) is ADDED to, like the other arrays. */
fqh_status fqh_len_hist(fqh_ctx *ctx, const uint64_t *d_base_hist, const uint64_t *d_scalars, uint32_t lmax,
                        uint64_t *d_len_hist);

/* ---- Filter and rewrite: scan -> select -> gather -> write ----------------------
 * Per-record alphabet flags of the sequence lines: bit 0 = Record::validate_dna (all of ACGT), bit 1 =
 * Record::validate_dnan (all of ACGTN), src/records.rs:19-33, over seq() (one trailing '\r' trimmed).
 * d_index[0..n): the IdxRecord-style index of fqh_index_records for this buffer; base_offset: file
 * offset of d_buf[0] (the scan's carry-in base_offset; 0 for a whole file). */
fqh_status fqh_record_flags(fqh_ctx *ctx, const uint8_t *d_buf, uint64_t len, uint64_t base_offset,
                            const fqh_idx_record *d_index, uint64_t n, uint8_t *d_flags);
/* Copies the raw bytes of every record with (d_flags[i] & mask) == want, in order and back to back,
 * to d_out: what a filter loop over Record::write (RefRecord::write copies the record's bytes,
 * src/records.rs:93-96) writes on the CPU.  *n_selected and *out_bytes always receive the totals;
 * d_out may be NULL to size the output; FQH_E_CAPACITY if out_bytes > out_cap (records that do not
 * fit completely are not written). */
fqh_status fqh_gather_records(fqh_ctx *ctx, const uint8_t *d_buf, uint64_t len, uint64_t base_offset,
                              const fqh_idx_record *d_index, uint64_t n, const uint8_t *d_flags,
                              uint8_t mask, uint8_t want, uint8_t *d_out, uint64_t out_cap,
                              uint64_t *n_selected, uint64_t *out_bytes);

/* ---- Streaming ingest: the GPU counterpart of Buffer + thread_reader --------------------------
 * src/buffer.rs keeps one 68 KiB window and memmoves the partial trailing record to its front;
 * src/thread_reader.rs fills two 4 MiB boxes in a background thread.  Here the window is a ring of
 * pinned host slots (MiBs each) with a device twin per slot: fqh_stream_submit starts the
 * host-to-device copy on a side stream and returns; fqh_stream_collect scans the oldest submitted
 * slot (carry chained from the previous one, so slots may end anywhere), brings the record index
 * back and copies the partial trailing record in front of the next slot, so every record is
 * contiguous in host memory when the caller walks it.  Copies of later slots overlap the scan and
 * the caller's work on earlier ones.  Single producer/consumer, one thread.
 * Between two fqh_stream_collect calls the CONTEXT belongs to the stream: a collect may already have enqueued the next slot's
 * scan on it (the scan then runs while the caller walks this chunk), so other calls on the same context — fqh_scan, fqh_stats,
 * fqh_record_flags, fqh_gather_records — return FQH_E_ARG ("a launch is pending") until the stream is drained or destroyed.  A
 * consumer that filters a collected chunk on the device (d_data / d_rec_start) does so on a second context of the same device. */
typedef struct fqh_stream fqh_stream;
typedef struct {
    int32_t parse_status;  /* FQH_OK or FQH_E_HEADER..FQH_E_TOO_LONG; an error ends the stream         */
    int32_t is_final;
    uint64_t n_records;    /* records that END in this chunk (before the first error)                  */
    uint64_t base_offset;  /* file offset of the chunk's first new byte                                */
    uint64_t data_len;     /* new bytes in the chunk                                                   */
    uint64_t lead_len;     /* bytes of the record in progress available in front of h_data             */
    const uint8_t *h_data; /* pinned host memory: byte at base_offset; h_data[-lead_len..] is valid    */
    const fqh_idx_record *h_index; /* n_records entries, `start` = file offset; NULL without FQH_STREAM_INDEX */
    const uint64_t *h_rec_start;   /* n_records + 1 boundaries (file offsets)                          */
    const uint8_t *d_data;         /* device twin of the new bytes                                     */
    const uint64_t *d_rec_start;   /* device copy of the boundaries                                    */
    uint64_t err_record, err_offset;
    uint64_t err_need;     /* bytes of the failing record that must be visible to report its error
                              (0: truncated tail, reported at EOF only; UINT64_MAX: no failing record):
                              input of a host-side replay of the reference's Buffer                   */
} fqh_chunk;
#define FQH_STREAM_INDEX 1u /* also build + download the IdxRecord-style index per chunk */
#define FQH_STREAM_STATS 2u /* also add every delivered record to the histograms of fqh_stream_set_stats */
#define FQH_STREAM_TIMING 4u /* HIP events around every slot's copy and scan: fqh_stream_timing */
#define FQH_STREAM_EXTERNAL 8u /* every slot is fed by fqh_stream_submit_external: no pinned data area per slot (only the lead
                                  area for the record in progress), fqh_stream_acquire is FQH_E_ARG */
/* (slot_bytes: a slot costs about 0.2 ms of device-side launches whatever it holds, and a host-to-device copy of a few MiB does
 * not fill the link — measured with the slots filled once and submitted again and again: 255 MiB slots 52 GB/s, 32 MiB 39 GB/s,
 * 4 MiB 11 GB/s; slots of 64 MiB and more keep the copy engine the bound) */
fqh_status fqh_stream_create(fqh_ctx *ctx, uint64_t slot_bytes, uint32_t n_slots, uint32_t flags,
                             fqh_stream **out);
void fqh_stream_destroy(fqh_stream *st);
/* FQH_STREAM_STATS: device arrays as for fqh_stats (ADDED to).  Every record the stream delivers is
 * counted exactly once, with the chunk it ends in; the device slots keep the partial trailing record in
 * front of the next chunk for that. */
fqh_status fqh_stream_set_stats(fqh_stream *st, uint32_t lmax, uint64_t *d_qual_hist, uint64_t *d_base_hist,
                                uint64_t *d_scalars);
/* Where to put the next input bytes (pinned host memory, *cap bytes).  FQH_E_CAPACITY if every slot
 * is submitted or held by the caller. */
fqh_status fqh_stream_acquire(fqh_stream *st, uint8_t **h_dst, uint64_t *cap);
fqh_status fqh_stream_submit(fqh_stream *st, uint64_t nbytes, int is_final);
/* The slot's bytes straight from the CALLER's memory, no staging copy: h_src[0..nbytes) is page-locked host memory
 * (fqh_host_register, hipHostRegister / hipHostMalloc of the host's own, an mmap'ed file registered once) and the DMA engine
 * reads it directly — the host moves 1 byte of DRAM traffic per input byte instead of 3 (read source + write pinned slot + DMA
 * read), which is what bounds eight rings on one host.  The reference's thread_reader makes the same copy
 * this removes (src/thread_reader.rs:90-97).  Takes the next free slot by itself (no fqh_stream_acquire; FQH_E_CAPACITY when the
 * ring is full); nbytes <= slot_bytes; h_src must stay unchanged until the chunk's fqh_stream_release.  The chunk's h_data is
 * h_src itself and its lead_len is 0: the beginning of the record in progress lies at the end of the previous chunk's memory
 * (d_data keeps it in front, as always).  Pageable h_src works too, at the speed of the driver's own staging. */
fqh_status fqh_stream_submit_external(fqh_stream *st, const uint8_t *h_src, uint64_t nbytes, int is_final);
/* Page-lock / release a range of the host's own memory for fqh_stream_submit_external (hipHostRegister / hipHostUnregister). */
fqh_status fqh_host_register(fqh_ctx *ctx, void *h_ptr, uint64_t bytes);
fqh_status fqh_host_unregister(fqh_ctx *ctx, void *h_ptr);
fqh_status fqh_stream_collect(fqh_stream *st, fqh_chunk *out);
fqh_status fqh_stream_release(fqh_stream *st); /* done with the chunk of the last collect */
/* Done with THIS chunk, whichever collect handed it out: a host may hold several chunks at once (RecordSets that borrow a
 * slot's pinned memory while worker threads walk them, the way a RecordSet of the reference owns its buffer,
 * src/lib.rs:306-318, 384-385) and give them back in any order.  The ring still fills its slots in order: a slot that is held
 * stops the producer when its turn comes (fqh_stream_acquire: FQH_E_CAPACITY), and fqh_stream_collect returns FQH_E_AGAIN —
 * nothing done — while the slot BEHIND the one it would collect is held (the partial trailing record goes in front of that
 * slot's data).  Same thread as every other call on the stream. */
fqh_status fqh_stream_release_chunk(fqh_stream *st, const fqh_chunk *c);
/* FQH_STREAM_TIMING: how the ingest overlapped the scan, measured with HIP events on the two streams over the slots collected
 * so far — the point of src/thread_reader.rs:131-139 (the producer's read() runs while the consumer parses).  copy_busy_ms /
 * scan_busy_ms: time the side stream spent in host-to-device copies / the context's stream in the slots' kernels;
 * both_busy_ms: time both were busy at once; wall_ms: first begin to last end. */
typedef struct {
    double wall_ms, copy_busy_ms, scan_busy_ms, both_busy_ms;
    uint64_t n_slots;
} fqh_stream_times;
fqh_status fqh_stream_timing(fqh_stream *st, fqh_stream_times *out);
/* For a stream that does not begin the file (a byte range of it, from a record start): the file offset of its first byte.
 * Boundaries and error offsets are then file offsets, and "Fastq record is too long" — which depends on a record's file
 * offset mod 16 and on nothing else (csrc/replay.h) — is judged as the reference judges it in the whole file.  Before the
 * first fqh_stream_acquire. */
fqh_status fqh_stream_set_origin(fqh_stream *st, uint64_t file_offset);
/* For hosts whose reader may come back SHORT (a pipe, a socket, a decompressor): Buffer::read_into makes one reader.read() per
 * refill and takes what it gets (src/buffer.rs:74-100), so with such a reader the reference's verdict on a record of
 * BUFSIZE - 15 .. BUFSIZE bytes depends on the sizes of the reads, not only on the record's file offset.  Call this once per
 * read() the host makes into an acquired slot — `got` bytes of the `asked` — from the stream's first slot on (FQH_E_ARG later,
 * or for a stream with an origin); slots of at least BUFSIZE bytes.  "Fastq record is too long" is then decided by the replay of
 * the reference's Buffer under a reader that hands out what the notes say (csrc/replay.h: exact for a reader with one cap per
 * call, tests/replay_fuzz.cpp; for a pipe whose reads depend on timing, the closest statement there is).  Without any note the
 * reader is taken to fill every read — a file — and the rule's closed form is used.  FQH_E_ARG: slots smaller than BUFSIZE, or a
 * BUFSIZE (fqh_set_bufsize) that is no longer the one the first note was made under.  (The replay stops where the reference's
 * reader would block, up to BUFSIZE bytes behind the chunk it was given; the verdict on a record never needs those bytes — it
 * falls when the reference's buffer is full and the record still open, before the record's last byte is read — so the chunk a
 * record ends in is also the chunk that judges it: tests/test_gpu_stream.py.) */
fqh_status fqh_stream_note_read(fqh_stream *st, uint64_t got, uint64_t asked);
/* Parser state behind the last collected chunk (nl_count = newlines the stream has seen: what the next shard's phase is
 * checked against in the sharded mode). */
fqh_status fqh_stream_carry(fqh_stream *st, fqh_carry *out);

/* Timing of the kernels of the last launch/finish pair, measured with HIP events on the
 * launch stream: total and per-kernel milliseconds (index, prefix, emit, stats). */
typedef struct {
    float total_ms, index_ms, prefix_ms, emit_ms, stats_ms;
} fqh_timing;
fqh_status fqh_last_timing(fqh_ctx *ctx, fqh_timing *out);

/* Synthetic 150 bp FASTQ of SURVEY §8d, generated in HBM: bytes [byte_off, byte_off+len) of the
 * infinite synthetic file (330 bytes per record, record i a pure function of (seed, i)). */
fqh_status fqh_synth_fill(fqh_ctx *ctx, uint8_t *d_out, uint64_t byte_off, uint64_t len,
                          uint64_t seed);

/* Plain 16-byte-load read-reduction over d_buf: the measured streaming-read ceiling the roofline
 * fraction is compared with.  *checksum receives the 64-bit sum of all dwords. */
fqh_status fqh_read_ceiling(fqh_ctx *ctx, const uint8_t *d_buf, uint64_t len, uint64_t *checksum,
                            float *ms);

/* Device memory helpers for hosts without their own allocator (the C++ mirror, the CLI). */
fqh_status fqh_dev_alloc(fqh_ctx *ctx, uint64_t bytes, void **d_ptr);
fqh_status fqh_dev_free(fqh_ctx *ctx, void *d_ptr);
fqh_status fqh_memcpy_h2d(fqh_ctx *ctx, void *d_dst, const void *h_src, uint64_t bytes);
fqh_status fqh_memcpy_d2h(fqh_ctx *ctx, void *h_dst, const void *d_src, uint64_t bytes);
fqh_status fqh_memset(fqh_ctx *ctx, void *d_dst, int value, uint64_t bytes);

#ifdef FQH_BUILDING_LIBRARY
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif
