// fqh_internal.h — layouts shared by the HIP kernels and the host side of libfastq_hip.so.
//
// Data layout in HBM:
//   input        d_buf[len]                       caller-owned, 16-byte aligned
//   wave tile    16 KiB of input = 16 pieces of 1 KiB; one wavefront owns one tile
//   line list    u16 list[n_tiles][list_cap]      one entry per LINE START inside the tile
//                bits 0..13 tile-relative offset, bit 14 byte=='@', bit 15 byte=='+'
//   tile_count   u32 [n_tiles]                    entries of each tile (true count, may exceed cap)
//   tile_prefix  u32 [n_tiles] + block_prefix u64 [n_tiles/2048 + 1]   exclusive scan of tile_count
//   rec_start    u64 [n_records + 1]              caller-owned output
//   index        fqh_idx_record [n_records]       optional (stats / RecordSet hand-back)
#pragma once
#include <hip/hip_runtime.h>

#include <mutex>
#include <stdint.h>

#include "../../include/fastq_hip.h"

namespace fqh {

constexpr uint32_t WT_BYTES = 16384;    // bytes per wave tile
constexpr uint32_t WT_SHIFT = 14;
constexpr uint32_t PIECE_BYTES = 1024;  // one wave-wide 16-byte load
constexpr uint32_t WT_PIECES = WT_BYTES / PIECE_BYTES;
constexpr uint32_t LIST_CAP_DEFAULT = 512;   // entries per tile list (average line >= 32 bytes); a scan that
                                             // overflows reruns with 2048, then 16384, and the context keeps it
constexpr uint32_t SCAN_CHUNK = 2048;        // tiles per block of the tile-count scan
constexpr uint32_t SCAN_SHIFT = 11;
constexpr unsigned long long NOKEY = ~0ull;

// Everything the emit / finalize kernels need to know about one scan call.
struct ScanArgs {
    const uint8_t *buf;
    uint64_t len;
    // carry-in (fqh_carry)
    uint64_t base_offset, nl_count;
    uint64_t back[4];
    int32_t is_final;
    uint32_t v_start;   // 1 if a line starts at chunk offset 0 (back[0]==0 && len>0)
    uint64_t bufsize;   // 0 = unlimited
    uint32_t max_walk;  // tiles a predecessor search may cross before the record is "too long"
    uint32_t head_unchecked;  // the record in progress at the chunk start began at unknown distances (a shard whose line
                              // phase is being probed, fqh_shard_align): its length rule and length are not checked
    // tile index
    const uint16_t *list;
    uint32_t list_cap;
    const uint32_t *tile_count;
    const uint16_t *fast_rs;       // fast path only: [tile][128]: 64 record-start offsets, then 8 edge entries, count, alignment
    const uint32_t *tile_prefix;   // exclusive prefix inside its SCAN_CHUNK block
    const uint64_t *block_prefix;  // exclusive prefix of block sums; [n_blocks] = total
    uint64_t n_tiles;
    uint64_t n_blocks;
    // outputs
    uint64_t *rec_start;
    uint64_t cap;
    fqh_idx_record *idx;
    uint64_t idx_cap;
    struct DevOut *mirror;         // pinned host copy of the results, written by the finalize kernel (nullable)
    uint32_t prescan;              // fqh_shard_prescan*: no emit, nothing that depends on the line phase is validated
    const struct DevCarry *dcarry; // the carry-in lives in device memory (the device-side shard protocol: k_carry_fold wrote it
                                   // on this stream); the emit / finalize kernels then take it from there
};

// A carry in device (or pinned host) memory, laid out as fqh_carry, and what the shard exchange found out.
struct DevCarry {
    uint64_t base_offset, nl_count;
    uint64_t back[4];
    uint64_t any_fail;             // some rank's prescan left the fast path: the exchanged words are not to be used
};
constexpr uint32_t SHARD_WORDS = 8;  // FQH_SHARD_WORDS: len, newlines, line starts, back_zero_carry[4], "left the fast path"

// Arguments of the line-parallel histogram kernels (stats_kernels.hip: k_stats_oct, k_stats_lines).
struct StatsArgs {
    const uint8_t *buf;
    uint64_t len;
    uint64_t valid_end;          // chunk-relative end of the last delivered record
    uint64_t nl_count;           // carry-in: newlines before the chunk
    uint64_t line_lo, line_hi;   // global line indices [lo, hi): lines of the records that count
    const uint16_t *list;
    uint32_t list_cap;
    const uint32_t *tile_count;
    const uint32_t *tile_prefix;
    const uint64_t *block_prefix;
    uint64_t n_tiles;
    uint32_t lmax;               // rows of the caller's histograms
    uint32_t col0, lc;           // this pass of k_stats_oct counts columns col0 .. col0 + lc - 1 (lc <= 256 rows in LDS)
    uint32_t max_line;           // no line that counts is longer than this (0: unknown); bounds the number of passes
    uint32_t last;               // this is the pass that holds the last of the caller's rows: it sees sequence lines to their end
    uint32_t *flagmap;           // several passes: one bit per record that counts, "has an N or worse" in words [0, flag_words),
    uint64_t flag_words;         //   "has a byte outside ACGTN" behind them (zeroed by the caller); NULL with one pass
    uint32_t *cr_flag;           // several passes: pass 0 sets it if any line that counts ends in "\r\n"; the others look for '\r' only then
    uint32_t *scratch;           // [gridDim.x][SO_WORDS] per-block partial histograms
    unsigned long long *qual_hist, *base_hist, *scalars;
    uint32_t dbg;                // timing experiments only (FQH_STATS_DBG, k_stats_oct<5, true>): 1 no LDS atomics, 4 no counting,
                                 // 8 generic tile path, 16 no '\\r' probes, 64 sequence lines only, 128 / 256 one / no load per batch,
                                 // 512 no batches, 1024 only walk the tiles, 8192 report section cycles, 16384 every load twice,
                                 // 32768 all groups read one line, 65536 half the waves idle
};

// Device-resident accumulators and the finalize kernel's results (one D2H copy per scan).
struct DevOut {
    // accumulated by k_index / k_emit with atomics; reset before every scan
    unsigned long long min_key;     // min(record * 4 + stage) over violated predicates
    unsigned long long first_long;  // min global record index with length >= bufsize - 15
    unsigned long long max_len;     // longest complete record ending in the chunk
    unsigned long long overflow;    // tiles whose list overflowed list_cap (=> rerun)
    unsigned long long spec_fail;   // fast path: some tile could not be proven valid (=> exact rerun)
    // the single pass (k_scan_stats): what it could not COUNT — no doubt about the parse.  Batches of eight lines with a byte
    // outside ACGTN / '!'..'`' are dumped (registers and all) to FusedArgs::decl_b, lines longer than the histogram's rows are
    // listed in FusedArgs::decl_l; k_stats_declined counts both exactly behind k_stats_commit.  stats_declined: something could
    // be neither dumped nor listed (a line of more than ~500 bytes, a full dump area): nothing of the pass is committed and
    // the host counts in a second pass — the scan's result stands, the fast path's back-off is not touched
    unsigned long long decl_batches, decl_lines, stats_declined;
    // written by k_finalize
    unsigned long long total_entries;
    unsigned long long lastnl;
    long long recent[4];            // most recent line starts <= len, chunk-relative (may be < 0)
    unsigned long long n_newlines;
    unsigned long long final_key;   // min_key incl. the truncation rule; NOKEY if none
    unsigned long long n_records;   // records ending in the chunk before the first error
    long long end_off;              // chunk-relative end of the last good record (may be < 0)
    long long err_start;            // chunk-relative start of the failing record
    unsigned long long err_need;    // bytes of the failing record needed to see its error (0: n/a)
    unsigned long long tail_len;
    unsigned long long need_list;    // fast path without a line-list workspace: a tile has more record starts than its two lines hold
                                     // (reads shorter than ~50 bp): rerun the fast path with the workspace (reset by finalize)
    unsigned long long stats_commit; // written by k_finalize_fast: 1 = the fast path's result stands AND the single pass counted (or
                                     // dumped / listed) every line (k_stats_commit may add what k_scan_stats counted to the caller's
                                     // histograms), 0 = it is discarded
    unsigned long long decl_b, decl_l;  // ... and how many dumped batches / listed lines k_stats_declined has to count
};

// Arguments of k_scan_stats (fused_kernels.hip): the input, the fast path's outputs of the byte scan, and where the
// counts go.
struct FusedArgs {
    const uint8_t *buf;
    uint64_t len;
    uint64_t n_tiles;
    uint16_t *list;       // list area: record starts beyond a tile's two lines (reads shorter than ~25 bp)
    uint32_t list_cap;
    uint16_t *fast_rs;    // per tile one 128-byte line (+ a second one), as k_index_fast writes them
    DevOut *out;          // spec_fail
    uint32_t lmax, lc;    // the caller's lmax; lc = rows of the LDS histogram in use (set by the launcher)
    uint32_t rows;        // rows the single pass is to keep (scan_stats_rows: <= lmax, by what is known about the reads' length)
    uint32_t *scratch;    // [gridDim.x][SO_WORDS] per-block partial histograms
    unsigned long long *scalars;  // FQH_NSCALARS totals (a zeroed side array: k_stats_commit adds them to the caller's)
    uint32_t skip_head;   // the chunk begins inside a record (carry-in): the lines of that record are k_stats_edge's, not this kernel's
    uint32_t *decl_b;     // dump area of declined batches: [decl_cap][(1 + NSL) * 64] words: the lanes' P (bit 31: quality lines), then their raw dwords per step
    uint64_t *decl_l;     // declined lines: [decl_cap] x 4 u32: tile, place of the line's first byte in the wave's data area | group << 16, length, kind (1: quality)
    uint32_t decl_cap;
    uint32_t wave_base;   // set by the launcher: bytes of histogram in front of the wavefronts' LDS areas
    uint32_t dbg;         // knock-out flags for timing experiments (FQH_FZ_DBG; results are wrong by design)
};

// hipFuncAttributeMaxDynamicSharedMemorySize of one kernel: set per DEVICE (a process may hold contexts on several), raised
// when a launch asks for more than the device's function was last given, and an error is the caller's to report.
// (Contexts are single-threaded, a PROCESS is not: two host threads with a context each may launch the same kernel at once —
// the function-local statics that hold these are shared, so the bookkeeping is under a lock.)
struct LdsAttr {
    size_t set[64] = {};
    std::mutex mu;
    hipError_t ensure(const void *fn, size_t lds) {
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
        std::lock_guard<std::mutex> lock(mu);
        if (lds <= set[dev]) return hipSuccess;
        e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess) set[dev] = lds;
        return e;
    }
};

void launch_len_hist(hipStream_t s, const unsigned long long *base_hist, const unsigned long long *scalars, uint32_t lmax,
                     unsigned long long *len_hist);
}  // namespace fqh
