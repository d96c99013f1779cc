// scan_kernels.hip — gfx950 kernels of the record scan.
//
//   k_index_t   streaming byte-scan: a wavefront owns a 16 KiB tile, 16-byte coalesced loads,
//               LDS transposition, SWAR newline detection, ballot/mbcnt in-wave prefix, emits the
//               tile's line-start list (replaces the memchr loop of src/records.rs:141,155,214,228).
//               Phase-free: needs no information from any other tile, shard or GPU.
//   k_prefix_*  exclusive scan of the per-tile counts.
//   k_emit      walks the line-start lists with the global line index known: record offsets,
//               '@' / '+' / length checks in the order of src/records.rs:201-247.
//   k_finalize  EOF rule (src/lib.rs:264-294), carry-out, summary.
#include <hip/hip_runtime.h>

#include <atomic>

#include <cstdlib>
#include <type_traits>

#include "scan_dev.h"

namespace fqh {


// ---------------------------------------------------------------------------------------------
// exclusive scan of tile_count: per-block local prefix + block sums, then the block sums.
// The counts are read at cnt[t * stride]: stride 1 = the dense tile_count array; stride 32 = the count
// slot of the fast path's per-tile line (k_index_fast then issues one whole-line store per tile and
// nothing else), copied to dense_out for later readers.
__global__ __launch_bounds__(256) void k_prefix_local(const uint32_t *__restrict__ cnt, uint32_t stride,
                                                      uint32_t *__restrict__ dense_out,
                                                      uint32_t *__restrict__ tile_prefix,
                                                      uint64_t *__restrict__ block_sum,
                                                      uint64_t n_tiles) {
    __shared__ uint32_t wsum[4];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint64_t t0 = (uint64_t)blockIdx.x * SCAN_CHUNK + (uint64_t)tid * 8;
    uint32_t c[8];
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        c[i] = (t0 + i < n_tiles) ? cnt[(t0 + i) * stride] : 0u;
        s += c[i];
    }
    if (dense_out) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (t0 + i < n_tiles) dense_out[t0 + i] = c[i];
    }
    // inclusive wave scan of s
    uint32_t inc = s;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t o = __shfl_up(inc, d);
        if (lane >= (uint32_t)d) inc += o;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    uint32_t wbase = 0;
    for (uint32_t w = 0; w < wave; ++w) wbase += wsum[w];
    uint32_t ex = wbase + inc - s;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (t0 + i < n_tiles) tile_prefix[t0 + i] = ex;
        ex += c[i];
    }
    if (tid == 255) block_sum[blockIdx.x] = (uint64_t)wbase + inc;
}

// single block: block_prefix[b] = exclusive prefix of block_sum, block_prefix[n_blocks] = total.
// In place (block_sum == block_prefix is allowed: array has n_blocks + 1 slots).
__global__ __launch_bounds__(1024) void k_prefix_top(uint64_t *__restrict__ block_prefix, uint64_t n_blocks) {
    __shared__ unsigned long long wsum[16];
    __shared__ unsigned long long carry;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (uint64_t base = 0; base < n_blocks; base += 1024) {
        const uint64_t i = base + tid;
        unsigned long long s = (i < n_blocks) ? block_prefix[i] : 0ull;
        unsigned long long inc = s;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            unsigned long long o = __shfl_up(inc, d);
            if (lane >= (uint32_t)d) inc += o;
        }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        unsigned long long wbase = carry;
        for (uint32_t w = 0; w < wave; ++w) wbase += wsum[w];
        if (i < n_blocks) block_prefix[i] = wbase + inc - s;
        __syncthreads();
        if (tid == 1023) carry = wbase + inc;
        __syncthreads();
    }
    if (tid == 0) block_prefix[n_blocks] = carry;
}

// ---------------------------------------------------------------------------------------------
// helpers on the tile index
// The carry-in of a launch whose carry was worked out on the device (fqh_shard_rescan_launch): the DEVC variants of the
// emit / finalize kernels work on a copy of their arguments in LDS with the carry filled in.  (Not on the by-value
// argument itself: a kernel that writes to its argument struct and indexes it gets it in scratch memory — 200 to 400
// bytes per lane, which the runtime multiplies by a full device of lanes — and the plain variants would pay for it too.)
// (COPY: the single-thread finalize kernels index their arguments with run-time subscripts; from the LDS copy that costs
// an LDS read, from the by-value argument the whole struct went to scratch.)
#define FQH_ARGS_WITH_CARRY(DEVC, COPY, a_in)                                                   \
    __shared__ ScanArgs a_sh_;                                                                   \
    if (DEVC || COPY) {                                                                          \
        if (threadIdx.x == 0) {                                                                  \
            a_sh_ = a_in;                                                                        \
            if (DEVC) {                                                                          \
                a_sh_.base_offset = a_in.dcarry->base_offset;                                    \
                a_sh_.nl_count = a_in.dcarry->nl_count;                                          \
                for (int i_ = 0; i_ < 4; ++i_) a_sh_.back[i_] = a_in.dcarry->back[i_];           \
                a_sh_.v_start = (a_sh_.back[0] == 0 && a_in.len > 0) ? 1u : 0u;                  \
            }                                                                                    \
        }                                                                                        \
        __syncthreads();                                                                         \
    }                                                                                            \
    const ScanArgs &a = (DEVC || COPY) ? a_sh_ : a_in;
__device__ __forceinline__ uint64_t tile_pref(const ScanArgs &a, uint64_t t) {
    return a.block_prefix[t >> SCAN_SHIFT] + a.tile_prefix[t];
}
__device__ __forceinline__ uint32_t tile_cnt(const ScanArgs &a, uint64_t t) {
    uint32_t c = a.tile_count[t];
    return c < a.list_cap ? c : a.list_cap;
}
__device__ __forceinline__ long long entry_start(const ScanArgs &a, uint64_t t, uint32_t i) {
    return (long long)((t << WT_SHIFT) + (a.list[t * a.list_cap + i] & 0x3FFFu));
}

// The (up to) four line starts that precede entry i of tile t, most recent first.  t == n_tiles,
// i == 0 asks for the starts preceding the end of the chunk.  Starts before the chunk come from the
// virtual start entry (offset 0) and the carry.  Returns false if the search crossed more than
// max_walk tiles: then the enclosing record is longer than BUFSIZE and the rest is filled with a
// far-away sentinel.
__device__ bool collect_prev(const ScanArgs &a, uint64_t t, uint32_t i, long long out[4]) {
    int n = 0;
    long long ti = (long long)t;
    long long ii = (long long)i - 1;
    uint32_t walked = 0;
    bool ok = true;
    while (n < 4) {
        if (ii >= 0) {
            out[n++] = entry_start(a, (uint64_t)ti, (uint32_t)ii);
            --ii;
            continue;
        }
        --ti;
        if (ti < 0) break;
        if (++walked > a.max_walk) {
            ok = false;
            break;
        }
        ii = (long long)tile_cnt(a, (uint64_t)ti) - 1;
    }
    if (!ok) {
        while (n < 4) out[n++] = -(1ll << 60);
        return false;
    }
    if (n < 4) {
        int j = 0;
        if (a.v_start) {
            out[n++] = 0;
            j = 1;
        }
        while (n < 4) {
            out[n++] = -(long long)a.back[j < 4 ? j : 3];
            ++j;
        }
    }
    return true;
}

struct Acc {
    unsigned long long key, first_long, max_len;
};

// Checks of the record that ENDS right before the line start S (global line index l, l % 4 == 0):
// length rule src/records.rs:233-238, record length for the too-long rule, index entry.
__device__ __forceinline__ void close_record(const ScanArgs &a, uint64_t t, uint32_t i, long long S,
                                             unsigned long long l, Acc &acc) {
    long long p[4];
    const bool ok = collect_prev(a, t, i, p);
    const unsigned long long rec = (l >> 2) - 1;  // the record that just ended
    unsigned long long reclen;
    if (a.head_unchecked && rec == (a.nl_count >> 2)) return;  // began before the chunk, at distances nobody knows yet
    if (ok) {
        // newlines: nl0=p[2]-1 nl1=p[1]-1 nl2=p[0]-1 nl3=S-1; raw line lengths nl3-nl2 vs nl1-nl0
        if ((S - p[0]) != (p[1] - p[2])) {
            unsigned long long k = rec * 4 + 2;
            if (k < acc.key) acc.key = k;
        }
        reclen = (unsigned long long)(S - p[3]);
    } else {
        reclen = 1ull << 60;
    }
    if (reclen > acc.max_len) acc.max_len = reclen;
    if (a.bufsize && reclen + 15 >= a.bufsize && rec < acc.first_long) acc.first_long = rec;
    const unsigned long long r = (l >> 2) - (a.nl_count >> 2);  // local index of the NEXT record
    if (a.idx && ok && r - 1 < a.idx_cap) {
        fqh_idx_record ir;
        ir.start = a.base_offset + (unsigned long long)p[3];
        ir.head = (uint32_t)(p[2] - 1 - p[3]);
        ir.seq = (uint32_t)(p[1] - 1 - p[3]);
        ir.sep = (uint32_t)(p[0] - 1 - p[3]);
        ir.qual = (uint32_t)(S - 1 - p[3]);
        a.idx[r - 1] = ir;
    }
}

__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        unsigned long long o = __shfl_xor(v, d);
        v = o < v ? o : v;
    }
    return v;
}
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        unsigned long long o = __shfl_xor(v, d);
        v = o > v ? o : v;
    }
    return v;
}

// ---------------------------------------------------------------------------------------------
// k_emit: persistent grid, a wavefront takes tiles round-robin.
//
// Per tile: the line-start list (u16 entries) is staged in LDS behind a 4-entry header holding the
// last four entries of the PREVIOUS tile, so a record-start entry finds its four predecessor line
// starts with LDS reads even when the record began in the previous tile.  '@'/'+' bits are checked
// while staging (one lane per entry); the per-record work (offset store, length rule of
// src/records.rs:233-238, record length) runs one lane per RECORD in 32-bit tile-relative
// arithmetic.  Everything 64-bit (error keys, generic predecessor search for tiles whose
// predecessor has fewer than four entries) sits behind wave-uniform rare branches.
// A three-stage software pipeline (counts -> list entries -> process) keeps two tiles of loads in
// flight per wavefront, so no memory latency is exposed per tile.

__device__ __forceinline__ uint64_t uniform64(uint64_t v) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}

// generic (slow) per-entry processing of one tile: any list length, any predecessor situation
__device__ void emit_tile_generic(const ScanArgs &a, uint64_t t, uint32_t cnt, unsigned long long lbase,
                                  uint32_t lane, Acc &acc) {
    const unsigned long long r0 = a.nl_count >> 2;
    const uint16_t *__restrict__ tl = a.list + t * a.list_cap;
    for (uint32_t i = lane; i < cnt; i += 64) {
        const uint32_t e = tl[i];
        const long long S = (long long)((t << WT_SHIFT) + (e & 0x3FFFu));
        const unsigned long long l = lbase + i;
        const uint32_t ph = (uint32_t)l & 3u;
        if (ph == 0) {
            if (!(e & 0x4000u)) {  // read_header: src/records.rs:138-147
                unsigned long long k = (l >> 2) * 4 + 0;
                if (k < acc.key) acc.key = k;
            }
            const unsigned long long r = (l >> 2) - r0;
            if (a.rec_start && r < a.cap) a.rec_start[r] = a.base_offset + (unsigned long long)S;
            close_record(a, t, i, S, l, acc);
        } else if (ph == 2) {
            if (!(e & 0x8000u)) {  // read_sep: src/records.rs:152-161
                unsigned long long k = (l >> 2) * 4 + 1;
                if (k < acc.key) acc.key = k;
            }
        }
    }
}

// A wavefront handles a SUPER-TILE of four consecutive tiles (64 KiB of input, ~800 line starts,
// ~200 records) per iteration: the four lists are compacted into one LDS array (their exclusive
// prefixes give the slots), which amortises the per-tile scalar overhead and fills the lanes of the
// record pass.  LDS word: bits 0..16 = offset relative to the super-tile + 16384 (so the tail of the
// previous tile, at negative offsets, needs no special case), bit 17 = '@', bit 18 = '+'.
constexpr uint32_t EMIT_G = 4;
constexpr uint32_t EMIT_WORDS = 1280;  // staged entries per wave; more -> generic path

template <bool DEVC>
__global__ __launch_bounds__(256) void k_emit(ScanArgs a_in, DevOut *__restrict__ out) {
    FQH_ARGS_WITH_CARRY(DEVC, false, a_in)
    __shared__ uint32_t stage_all[4][EMIT_WORDS + 4];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wv = threadIdx.x >> 6;
    uint32_t *const st = stage_all[wv] + 4;  // st[-4..-1]: tail of the previous tile
    const uint64_t nwaves = (uint64_t)gridDim.x * 4;
    const uint64_t n_super = (a.n_tiles + EMIT_G - 1) / EMIT_G;
    const unsigned long long r0 = a.nl_count >> 2;
    const uint32_t bufsize32 = a.bufsize > 0x7FFFFFFFull ? 0x7FFFFFFFu : (uint32_t)a.bufsize;
    Acc acc = {NOKEY, NOKEY, 0};
    uint32_t maxlen32 = 0;

    struct StageA { uint32_t cnt[4], tp[4], cprev; unsigned long long bp; };
    struct StageB { StageA h; uint32_t e[4][4]; uint32_t tail; };
    auto fetchA = [&](uint64_t sidx, StageA &p) {
        if (sidx < n_super) {
            const uint64_t t0 = sidx * EMIT_G;
#pragma unroll
            for (uint32_t k = 0; k < EMIT_G; ++k) {
                const bool in = t0 + k < a.n_tiles;
                p.cnt[k] = in ? a.tile_count[t0 + k] : 0u;
                p.tp[k] = in ? a.tile_prefix[t0 + k] : 0u;
            }
            p.bp = a.block_prefix[t0 >> SCAN_SHIFT];
            p.cprev = t0 ? a.tile_count[t0 - 1] : 0u;
        }
    };
    auto fetchB = [&](uint64_t sidx, const StageA &q, StageB &p) {
        if (sidx < n_super) {
            const uint64_t t0 = sidx * EMIT_G;
            p.h = q;
#pragma unroll
            for (uint32_t k = 0; k < EMIT_G; ++k) {
                // list_cap >= 256: always inside the workspace (a partial last super-tile re-reads tile n-1)
                const uint64_t tk = t0 + k < a.n_tiles ? t0 + k : a.n_tiles - 1;
                const uint16_t *tl = a.list + tk * a.list_cap + lane;
                p.e[k][0] = tl[0]; p.e[k][1] = tl[64]; p.e[k][2] = tl[128]; p.e[k][3] = tl[192];
            }
            const uint32_t cp = q.cprev < a.list_cap ? q.cprev : a.list_cap;
            p.tail = (lane < 4 && cp >= 4) ? (uint32_t)a.list[(t0 - 1) * a.list_cap + cp - 4 + lane] : 0u;
        }
    };
    uint64_t sidx = (uint64_t)blockIdx.x * 4 + wv;
    StageA sa;
    StageB sb;
    fetchA(sidx, sa);
    fetchB(sidx, sa, sb);
    fetchA(sidx + nwaves, sa);
    for (; sidx < n_super; sidx += nwaves) {
        const StageB cur = sb;
        fetchB(sidx + nwaves, sa, sb);
        fetchA(sidx + 2 * nwaves, sa);
        const uint64_t t0 = sidx * EMIT_G;
        uint32_t cnt[EMIT_G], slot[EMIT_G];
        uint32_t total = 0;
        bool capped = false;
#pragma unroll
        for (uint32_t k = 0; k < EMIT_G; ++k) {
            const uint32_t c = (uint32_t)__builtin_amdgcn_readfirstlane((int)cur.h.cnt[k]);
            capped |= c > a.list_cap;
            cnt[k] = c < a.list_cap ? c : a.list_cap;
            slot[k] = total;
            total += cnt[k];
        }
        if (total == 0) continue;
        const unsigned long long lbase = a.nl_count + 1 + uniform64(cur.h.bp) +
                                         (uint32_t)__builtin_amdgcn_readfirstlane((int)cur.h.tp[0]);
        const uint32_t cprev = (uint32_t)__builtin_amdgcn_readfirstlane((int)cur.h.cprev);
        if (total > EMIT_WORDS || capped || t0 == 0 || cprev < 4) {
            // dense tiles, first tiles of the chunk, or a predecessor tile with < 4 line starts
            unsigned long long lb = lbase;
            for (uint32_t k = 0; k < EMIT_G; ++k) {
                if (cnt[k]) emit_tile_generic(a, t0 + k, cnt[k], lb, lane, acc);
                lb += cnt[k];
            }
            continue;
        }
        const uint32_t lb3 = (uint32_t)lbase & 3u;
        // ---- compact the four lists into LDS; check '@' (phase 0) and '+' (phase 2) on the way
        __builtin_amdgcn_wave_barrier();
        uint32_t bad = 0;
        auto put = [&](uint32_t k, uint32_t i, uint32_t e) {
            if (i < cnt[k]) {
                const uint32_t j = slot[k] + i;
                st[j] = ((e & 0x3FFFu) + (k << WT_SHIFT) + WT_BYTES) | ((e & 0xC000u) << 3);
                const uint32_t ph = (lb3 + j) & 3u;
                bad |= ((ph == 0 && !(e & 0x4000u)) || (ph == 2 && !(e & 0x8000u))) ? 1u : 0u;
            }
        };
#pragma unroll
        for (uint32_t k = 0; k < EMIT_G; ++k) {
            put(k, lane, cur.e[k][0]); put(k, lane + 64, cur.e[k][1]);
            put(k, lane + 128, cur.e[k][2]); put(k, lane + 192, cur.e[k][3]);
            if (cnt[k] > 256) {
                const uint16_t *__restrict__ tl = a.list + (t0 + k) * a.list_cap;
                for (uint32_t i = lane + 256; i < cnt[k]; i += 64) put(k, i, tl[i]);
            }
        }
        if (lane < 4) st[(int)lane - 4] = (cur.tail & 0x3FFFu);  // previous tile: offset - 16384, biased + 16384
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (__ballot(bad)) {  // rare: some header / separator byte is wrong -> error keys
            for (uint32_t j = lane; j < total; j += 64) {
                const uint32_t v = st[j];
                const unsigned long long l = lbase + j;
                const uint32_t ph = (uint32_t)l & 3u;
                if (ph == 0 && !(v & 0x20000u)) { unsigned long long k = (l >> 2) * 4; if (k < acc.key) acc.key = k; }
                if (ph == 2 && !(v & 0x40000u)) { unsigned long long k = (l >> 2) * 4 + 1; if (k < acc.key) acc.key = k; }
            }
        }
        // ---- one lane per record: entry i0 + 4 m starts record m of the super-tile, closes the one before
        const uint32_t i0 = (4u - lb3) & 3u;
        if (i0 >= total) continue;
        const uint32_t nrec = (total - i0 + 3) >> 2;
        const unsigned long long rbase = ((lbase + i0) >> 2) - r0;  // local index of the first record start
        const unsigned long long vbase = a.base_offset + (t0 << WT_SHIFT) - WT_BYTES;  // minus the bias
        const bool cap_ok = rbase + nrec <= a.cap;
        uint64_t *__restrict__ rs = a.rec_start ? a.rec_start + rbase : nullptr;
        fqh_idx_record *__restrict__ ix = a.idx ? a.idx + (rbase - 1) : nullptr;
        uint32_t mism = 0;
        for (uint32_t m = lane; m < nrec; m += 64) {
            const int j = (int)(i0 + 4 * m);
            const uint32_t o = st[j] & 0x1FFFFu, o1 = st[j - 1] & 0x1FFFFu, o2 = st[j - 2] & 0x1FFFFu;
            const uint32_t o3 = st[j - 3] & 0x1FFFFu, o4 = st[j - 4] & 0x1FFFFu;
            if (rs && (cap_ok || rbase + m < a.cap)) __builtin_nontemporal_store((uint64_t)(vbase + o), rs + m);  // written once, not read here
            mism |= ((o - o1) != (o2 - o3)) ? 1u : 0u;  // src/records.rs:233-238
            const uint32_t reclen = o - o4;
            maxlen32 = reclen > maxlen32 ? reclen : maxlen32;
            if (bufsize32 && reclen + 15 >= bufsize32) {
                const unsigned long long rec = r0 + rbase + m - 1;
                if (rec < acc.first_long) acc.first_long = rec;
            }
            if (ix && rbase + m - 1 < a.idx_cap) {
                fqh_idx_record ir;
                ir.start = vbase + o4;
                ir.head = o3 - 1 - o4;
                ir.seq = o2 - 1 - o4;
                ir.sep = o1 - 1 - o4;
                ir.qual = o - 1 - o4;
                ix[m] = ir;
            }
        }
        if (__ballot(mism)) {  // rare: a length mismatch -> error keys
            for (uint32_t m = lane; m < nrec; m += 64) {
                const int j = (int)(i0 + 4 * m);
                const uint32_t o = st[j] & 0x1FFFFu, o1 = st[j - 1] & 0x1FFFFu, o2 = st[j - 2] & 0x1FFFFu;
                const uint32_t o3 = st[j - 3] & 0x1FFFFu;
                if ((o - o1) != (o2 - o3)) {
                    const unsigned long long k = (r0 + rbase + m - 1) * 4 + 2;
                    if (k < acc.key) acc.key = k;
                }
            }
        }
    }
    if (maxlen32 > acc.max_len) acc.max_len = maxlen32;
    // one set of atomics per wavefront of the persistent grid
    const unsigned long long k = wave_min_u64(acc.key);
    const unsigned long long fl = wave_min_u64(acc.first_long);
    const unsigned long long ml = wave_max_u64(acc.max_len);
    if (lane == 0) {
        if (k != NOKEY) atomicMin(&out->min_key, k);
        if (fl != NOKEY) atomicMin(&out->first_long, fl);
        if (ml) atomicMax(&out->max_len, ml);
    }
}

// ---------------------------------------------------------------------------------------------
// chunk-relative start of global line L (must be one of: a line start before the chunk that the
// carry still knows, the virtual start, a list entry, or the virtual end).
__device__ long long start_of_line(const ScanArgs &a, unsigned long long L, unsigned long long E,
                                   bool lastnl) {
    if (L <= a.nl_count) {
        unsigned long long j = a.nl_count - L;
        if (j == 0 && a.v_start) return 0;
        return -(long long)a.back[j < 4 ? j : 3];
    }
    unsigned long long e = L - (a.nl_count + 1);
    if (e >= E) return (long long)a.len;  // virtual end (only asked for when lastnl)
    // largest tile whose exclusive prefix is <= e
    uint64_t lo = 0, hi = a.n_tiles;  // invariant: pref(lo) <= e
    while (hi - lo > 1) {
        uint64_t mid = lo + (hi - lo) / 2;
        if (tile_pref(a, mid) <= e) lo = mid; else hi = mid;
    }
    return entry_start(a, lo, (uint32_t)(e - tile_pref(a, lo)));
}

// k_finalize: one thread.  EOF rule of src/lib.rs:264-294, virtual entries at both chunk ends,
// carry-out, summary.
// Last act of a finalize kernel: hand the results to the host (pinned memory, visible when the kernel has
// completed: no separate device-to-host copy) and leave the accumulators clean for the next scan (no
// host-to-device copy in front of it).
__device__ __forceinline__ void publish_and_reset(const ScanArgs &a, DevOut *out) {
    if (!a.mirror) return;
    *a.mirror = *out;
    out->min_key = NOKEY;
    out->first_long = NOKEY;
    out->max_len = 0;
    out->overflow = 0;
    out->spec_fail = 0;
    out->need_list = 0;
    out->decl_batches = 0;
    out->decl_lines = 0;
    out->stats_declined = 0;
}

template <bool DEVC>
__global__ void k_finalize(ScanArgs a_in, DevOut *__restrict__ out) {
    FQH_ARGS_WITH_CARRY(DEVC, true, a_in)
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const unsigned long long E = a.n_tiles ? a.block_prefix[a.n_blocks] : 0ull;
    const bool lastnl = a.len > 0 && a.buf[a.len - 1] == '\n';
    Acc acc = {out->min_key, out->first_long, out->max_len};
    const unsigned long long r0 = a.nl_count >> 2;

    // virtual start entry: a line starts at chunk offset 0
    if (a.v_start) {
        const unsigned long long l = a.nl_count;
        const uint8_t b = a.buf[0];
        if ((l & 3) == 0 && b != '@') { unsigned long long k = (l >> 2) * 4; if (k < acc.key) acc.key = k; }
        if ((l & 3) == 2 && b != '+') { unsigned long long k = (l >> 2) * 4 + 1; if (k < acc.key) acc.key = k; }
    }
    // start of the record in progress at the chunk start
    if (a.rec_start && a.cap > 0)
        a.rec_start[0] = a.base_offset - a.back[a.nl_count & 3];

    // line starts preceding the end of the chunk
    long long p[4];
    collect_prev(a, a.n_tiles, 0, p);
    long long recent[4];
    if (lastnl) {
        // virtual end entry: a line would start at offset len
        const unsigned long long l = a.nl_count + 1 + E;
        if ((l & 3) == 0) {
            close_record(a, a.n_tiles, 0, (long long)a.len, l, acc);
            const unsigned long long r = (l >> 2) - r0;
            if (a.rec_start && r < a.cap) a.rec_start[r] = a.base_offset + a.len;
        }
        recent[0] = (long long)a.len; recent[1] = p[0]; recent[2] = p[1]; recent[3] = p[2];
    } else {
        recent[0] = p[0]; recent[1] = p[1]; recent[2] = p[2]; recent[3] = p[3];
    }
    const unsigned long long n_newlines = a.len ? E + (lastnl ? 1 : 0) : 0;
    const unsigned long long T = a.nl_count + n_newlines;
    const unsigned long long col = (unsigned long long)((long long)a.len - recent[0]);
    const bool tail = (T & 3) != 0 || col > 0;
    if (a.is_final && tail) {  // "Possibly truncated input file", src/lib.rs:286-291
        unsigned long long k = (T >> 2) * 4 + 3;
        if (k < acc.key) acc.key = k;
    }
    const unsigned long long k_end = T >> 2;
    unsigned long long n_good = k_end;
    if (acc.key != NOKEY && (acc.key >> 2) < n_good) n_good = acc.key >> 2;
    // end of the last good record = start of line 4 * n_good
    long long end_off;
    if (n_good == k_end) end_off = recent[T & 3];
    else end_off = start_of_line(a, n_good * 4, E, lastnl);
    long long err_start = end_off;
    unsigned long long need = 0;
    if (acc.key != NOKEY) {
        const unsigned long long e = acc.key >> 2;
        const uint32_t stage = (uint32_t)acc.key & 3u;
        err_start = (e == k_end) ? recent[T & 3] : start_of_line(a, e * 4, E, lastnl);
        if (stage == 0) need = 1;
        else if (stage == 1) need = (unsigned long long)(start_of_line(a, e * 4 + 2, E, lastnl) + 1 - err_start);
        else if (stage == 2) need = (unsigned long long)(start_of_line(a, e * 4 + 4, E, lastnl) - err_start);
    }
    out->min_key = acc.key;
    out->first_long = acc.first_long;
    out->max_len = acc.max_len;
    out->total_entries = E;
    out->lastnl = lastnl;
    for (int i = 0; i < 4; ++i) out->recent[i] = recent[i];
    out->n_newlines = n_newlines;
    out->final_key = acc.key;
    out->n_records = n_good - r0;
    out->end_off = end_off;
    out->err_start = err_start;
    out->err_need = need;
    out->tail_len = (unsigned long long)((long long)a.len - recent[T & 3]);
    publish_and_reset(a, out);
}

// ---------------------------------------------------------------------------------------------
// launchers (host)
// ---------------------------------------------------------------------------------------------
// k_index_t: persistent grid, a wavefront takes 16 KiB tiles round-robin.  It fetches its tile with
// coalesced 16-byte loads (lane-strided, 1 KiB per instruction), but stages each 4 KiB group in
// LDS and reads it back so that lane l owns the 64 CONTIGUOUS bytes [64 l, 64 l + 64): one in-wave
// prefix, one boundary shuffle and ~3 emit-loop iterations per 4 KiB instead of per 1 KiB.
// LDS image: logical 16-byte chunk c of the group lives in slot c ^ ((c >> 4) & 3) — conflict-free
// for the lane-strided ds_write_b128 and for the lane-contiguous ds_read_b128 (MI355X LDS services
// b128 reads in 16-lane groups over a 16-slot bank row).
__global__ __launch_bounds__(256) void k_index_t(const uint8_t *__restrict__ buf, uint64_t len,
                                                 uint16_t *__restrict__ list, uint32_t list_cap,
                                                 uint32_t *__restrict__ tile_count,
                                                 uint64_t n_tiles, DevOut *__restrict__ out) {
    __shared__ __attribute__((aligned(16))) uint8_t lds_all[4][4096];
    const uint32_t lane = threadIdx.x & 63u;
    uint8_t *const lds = lds_all[threadIdx.x >> 6];
    const uint64_t nwaves = (uint64_t)gridDim.x * 4;
    uint32_t n_over = 0;
    // per-lane constants of the LDS image
    const uint32_t wslot = lane ^ ((lane >> 4) & 3u);        // write: chunk 64 j + lane -> slot 64 j + wslot
    const uint32_t s4 = ((lane >> 2) & 3u) << 4;              // read:  byte Q of the lane at 64 lane + (Q ^ s4)
    uint8_t *const wptr = lds + wslot * 16;
    const uint8_t *const rptr = lds + lane * 64;
    for (uint64_t tile = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6); tile < n_tiles; tile += nwaves) {
        const uint64_t tbase = tile << WT_SHIFT;
        uint16_t *__restrict__ tl = list + tile * list_cap;
        uint32_t run = 0;
        uint32_t prev = 0;
        if (tile > 0) prev = (buf[tbase - 1] == '\n') ? 1u : 0u;
        const uint32_t lo = lane * 16;
        if (tbase + WT_BYTES <= len) {
            const uint8_t *p = buf + tbase + lo;
            uint4 n0 = load16_nt(p), n1 = load16_nt(p + PIECE_BYTES);
            uint4 n2 = load16_nt(p + 2 * PIECE_BYTES), n3 = load16_nt(p + 3 * PIECE_BYTES);
#pragma unroll 1
            for (uint32_t g = 0; g < WT_PIECES / 4; ++g) {
                __builtin_amdgcn_wave_barrier();
                *reinterpret_cast<uint4 *>(wptr) = n0;
                *reinterpret_cast<uint4 *>(wptr + 1024) = n1;
                *reinterpret_cast<uint4 *>(wptr + 2048) = n2;
                *reinterpret_cast<uint4 *>(wptr + 3072) = n3;
                if (g + 1 < WT_PIECES / 4) {  // the next group's loads are in flight while this one is processed
                    p += 4 * PIECE_BYTES;
                    n0 = load16_nt(p); n1 = load16_nt(p + PIECE_BYTES);
                    n2 = load16_nt(p + 2 * PIECE_BYTES); n3 = load16_nt(p + 3 * PIECE_BYTES);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const uint4 d0 = *reinterpret_cast<const uint4 *>(rptr + (0u ^ s4));
                const uint4 d1 = *reinterpret_cast<const uint4 *>(rptr + (16u ^ s4));
                const uint4 d2 = *reinterpret_cast<const uint4 *>(rptr + (32u ^ s4));
                const uint4 d3 = *reinterpret_cast<const uint4 *>(rptr + (48u ^ s4));
                const uint32_t m_lo = eqmask16<1>(d0, 0x0A0A0A0Au) | (eqmask16<1>(d1, 0x0A0A0A0Au) << 16);
                const uint32_t m_hi = eqmask16<1>(d2, 0x0A0A0A0Au) | (eqmask16<1>(d3, 0x0A0A0A0Au) << 16);
                // line starts: the byte after a newline
                uint32_t ls_lo = (m_lo << 1) | wave_shr1(m_hi >> 31, prev);
                uint32_t ls_hi = __builtin_amdgcn_alignbit(m_hi, m_lo, 31);
                prev = ((uint32_t)__builtin_amdgcn_readlane((int)m_hi, 63)) >> 31;
                const uint32_t c = __popc(ls_lo) + __popc(ls_hi);
                const unsigned long long b1 = __ballot(c >= 1), b2 = __ballot(c >= 2), b3 = __ballot(c >= 3);
                uint32_t pre = __builtin_amdgcn_mbcnt_hi((uint32_t)(b1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b1, 0));
                pre = __builtin_amdgcn_mbcnt_hi((uint32_t)(b2 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b2, pre));
                pre = __builtin_amdgcn_mbcnt_hi((uint32_t)(b3 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b3, pre));
                uint32_t tot = (uint32_t)__popcll(b1) + (uint32_t)__popcll(b2) + (uint32_t)__popcll(b3);
                if (__ballot(c >= 4)) {
                    for (uint32_t k = 4;; ++k) {
                        const unsigned long long b = __ballot(c >= k);
                        if (!b) break;
                        pre = __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, pre));
                        tot += (uint32_t)__popcll(b);
                    }
                }
                const uint32_t ebase = g * 4 * PIECE_BYTES + lane * 64;
                if (run + tot <= list_cap) {  // uniform
                    uint16_t *__restrict__ dst = tl + run + pre;
                    while (ls_lo) {
                        const uint32_t q = __ffs(ls_lo) - 1;
                        ls_lo &= ls_lo - 1;
                        const uint32_t b = rptr[q ^ s4];
                        *dst++ = (uint16_t)((ebase + q) | ((b == '@') ? 0x4000u : 0u) | ((b == '+') ? 0x8000u : 0u));
                    }
                    while (ls_hi) {
                        const uint32_t q = __ffs(ls_hi) + 31;
                        ls_hi &= ls_hi - 1;
                        const uint32_t b = rptr[q ^ s4];
                        *dst++ = (uint16_t)((ebase + q) | ((b == '@') ? 0x4000u : 0u) | ((b == '+') ? 0x8000u : 0u));
                    }
                }
                run += tot;
            }
        } else {
#pragma unroll 1
            for (uint32_t j = 0; j < WT_PIECES; ++j) {
                const uint64_t off = tbase + (uint64_t)j * PIECE_BYTES + lo;
                if (tbase + (uint64_t)j * PIECE_BYTES >= len) break;  // uniform
                const uint4 v = load16(buf, off, len);
                index_piece<false, 1>(v, off, len, j * PIECE_BYTES + lo, lane, prev, run, tl, list_cap);
            }
        }
        if (lane == 0) tile_count[tile] = run;
        if (run > list_cap) ++n_over;
    }
    if (lane == 0 && n_over) atomicAdd(&out->overflow, (unsigned long long)n_over);
}

// ---------------------------------------------------------------------------------------------
// k_index_fast: the fast path's index kernel — the byte scan of k_index_t, restructured around
// the one thing that keeps it off the read ceiling: on gfx950 stores share vmcnt with loads, and the
// compiler's wait in front of a group's LDS write is vmcnt(0).  A wavefront that stores its tile's
// results and then loads the next tile sits out the store acknowledgement and the load latency once
// per tile.  Here
//   * the loads of the NEXT tile's first 4 KiB group (and the byte before that tile) are issued
//     during the last group of the current tile,
//   * the tile's results (two stores, each a whole 128-byte line, plus the dense entry count for
//     the prefix scan) are kept in registers and issued
//     in the NEXT tile's first group, right after that group's prefetch: the next vmcnt(0) is a
//     whole group (~5 us of work per wave) away, by which time loads and stores have both landed.
// Whole tiles only; the (at most one) partial tile at the end of the buffer is taken by wave 0 of
// block 0 through the generic piece loop afterwards.
__global__ __launch_bounds__(256) void k_index_fast(const uint8_t *__restrict__ buf, uint64_t len,
                                                    uint16_t *__restrict__ list, uint32_t list_cap,
                                                    uint16_t *__restrict__ fast_rs,
                                                    uint64_t n_tiles, DevOut *__restrict__ out) {
    __shared__ __attribute__((aligned(16))) uint8_t lds_all[4][4096 + FAST_ENTRIES * 2 + 16];
    const uint32_t lane = threadIdx.x & 63u;
    uint8_t *const lds = lds_all[threadIdx.x >> 6];
    uint16_t *const lst = reinterpret_cast<uint16_t *>(lds + 4096);
    const uint64_t nwaves = (uint64_t)gridDim.x * 4;
    const uint64_t wave0 = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint64_t n_full = len >> WT_SHIFT;  // whole tiles
    uint32_t n_over = 0;
    const uint32_t wslot = lane ^ ((lane >> 4) & 3u);        // write: chunk 64 j + lane -> slot 64 j + wslot
    const uint32_t s4 = ((lane >> 2) & 3u) << 4;              // read:  byte Q of the lane at 64 lane + (Q ^ s4)
    uint8_t *const wptr = lds + wslot * 16;
    const uint8_t *const rptr = lds + lane * 64;
    const uint32_t lo = lane * 16;

    // evaluate the four alignments over the staged list; rv = this lane's slot of the tile's line
    auto finish_tile = [&](uint64_t tile, uint32_t run, uint32_t nstaged, uint32_t &rv, bool tail) {
        uint32_t hyp = 7;
        if (tail && nstaged == run && run < 8) {  // the short tile at the end of the buffer: left to k_finalize_fast
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            rv = (lane >= FR_EDGE && lane < FR_EDGE + run) ? (uint32_t)lst[lane - FR_EDGE] : 0u;
            rv = lane == FR_CNT ? run : lane == FR_HYP ? (FR_SMALL | 4u) : rv;
            return;
        }
        if (nstaged == run && run >= 8) {  // uniform
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            auto window_ok = [&](uint32_t i) -> bool {  // entries i .. i + 4 as header, sequence, separator, quality, next header
                const uint32_t e0 = lst[i], e1 = lst[i + 1], e2 = lst[i + 2], e3 = lst[i + 3], e4 = lst[i + 4];
                return (e0 & 0x4000u) && (e2 & 0x8000u) &&
                       ((e2 & 0x3FFFu) - (e1 & 0x3FFFu)) == ((e4 & 0x3FFFu) - (e3 & 0x3FFFu));
            };
            // the alignment: among the tile's first four entries exactly one starts with '@' and has a '+' two entries on (a
            // header and its separator: a sequence line starts with neither, and a quality line that starts with '@' is followed
            // two lines on by a sequence line) ...
            const bool cand = lane < 4 && lane + 4 < run && (lst[lane] & 0x4000u) && (lst[lane + 2] & 0x8000u);
            uint32_t cons = (uint32_t)__ballot(cand);
            if (!cons || (cons & (cons - 1))) {
                // (rare: none or several — a quality line that starts with '@' in front of a sequence line that starts with
                // '+', which the parser accepts: the tile's first 64 windows under all four alignments must single out one)
                const bool in0 = lane + 4 < run;
                const bool ok0 = in0 && window_ok(lane);
                cons = 0;
#pragma unroll
                for (uint32_t r = 0; r < 4; ++r)
                    if (__ballot(in0 && (lane & 3u) == r) && !__ballot(in0 && !ok0 && (lane & 3u) == r)) cons |= 1u << r;
            }
            if (cons && !(cons & (cons - 1))) hyp = (uint32_t)__ffs(cons) - 1;
            // ... under which the whole tile is checked, one lane per RECORD (a 16 KiB tile of 150 bp reads: 49 lanes, one trip;
            // the windows of the three other alignments — three quarters of the work until round 3 — decide nothing: the tile's
            // records are valid if they are valid under the TRUE alignment, and k_emit_fast holds the tile's against the true
            // line index; a tile that guesses wrong fails there and the scan reruns on the exact path)
            if (hyp < 4) {
                bool bad = false;
                for (uint32_t i = hyp + 4 * lane; i + 4 < run; i += 256) bad = bad || !window_ok(i);
                if (__ballot(bad) != 0) hyp = 7;
            }
        }
        rv = 0;
        if (hyp < 4) {
            if (lane < FR_N) rv = hyp + 4 * lane < run ? (uint32_t)(lst[hyp + 4 * lane] & 0x3FFFu) : 0u;
            else if (lane < FR_EDGE + 4) rv = lst[lane - FR_EDGE];
            else if (lane < FR_EDGE + 8) rv = lst[run - 8 + (lane - FR_EDGE)];
            if (hyp + 4 * FR_N < run && fast_rs) {  // more record starts than the line holds: a second whole line ...
                const uint32_t j2 = FR_N + lane;
                __builtin_nontemporal_store(hyp + 4 * j2 < run ? (uint16_t)(lst[hyp + 4 * j2] & 0x3FFFu) : (uint16_t)0,
                                            fast_rs + fr2_off(n_tiles) + tile * FR2_N + lane);
                if (hyp + 4 * (FR_N + FR2_N) < run) {  // ... and the list area for the rest (reads shorter than ~25 bp)
                    uint16_t *__restrict__ tl = list + tile * list_cap;
                    for (uint32_t j = FR_N + FR2_N + lane; hyp + 4 * j < run; j += 64) tl[8 + j] = lst[hyp + 4 * j] & 0x3FFFu;
                }
            }
        } else {
            ++n_over;  // counted into spec_fail below
        }
        rv = lane == FR_CNT ? (run & 0xFFFFu) : lane == FR_CNT + 1 ? (run >> 16) : lane == FR_HYP ? hyp : rv;
    };
    auto store_tile = [&](uint64_t tile, uint32_t run, uint32_t rv) {
        // one whole 128-byte line, non-temporal: written once, read once by k_emit_fast.  On the boxes where
        // a plain store costs the kernel 0.45 ms, this one costs 0.2.  (No plain/nt switch here: the
        // optimizer merges two stores to one address and drops the hint.)
        // (fast_rs == nullptr: the yardstick launch of place_fast_rs, scan_dispatch.hip — the same scan without its stores)
        if (fast_rs) __builtin_nontemporal_store((uint16_t)rv, fast_rs + tile * FR_STRIDE + lane);
        (void)run;
    };

    // A wavefront takes a contiguous RUN of tiles (round 4; it took every nwaves-th tile before): 4 096 streams of 4 MiB each
    // instead of one 64 MiB window that 4 096 wavefronts sweep together — 2.84 against 2.87–2.90 ms per 16 GiB step on inputs of
    // the fast kind, 3.00 against 2.96 on the slow kind (tools/exp_buf_ab.py; DESIGN.md 4b: the kind is the input allocation's)
    const uint64_t per_w = (n_full + nwaves - 1) / nwaves;
    uint64_t tile = wave0 * per_w;
    const uint64_t tend = tile + per_w < n_full ? tile + per_w : n_full;
    if (tile < tend) {
        const uint8_t *p = buf + (tile << WT_SHIFT) + lo;
        uint32_t pb = tile ? buf[(tile << WT_SHIFT) - 1] : 0u;  // the byte before the tile
        uint4 n0 = load16_nt(p), n1 = load16_nt(p + PIECE_BYTES);
        uint4 n2 = load16_nt(p + 2 * PIECE_BYTES), n3 = load16_nt(p + 3 * PIECE_BYTES);
        bool pending = false;          // the previous tile's two lines are still in registers
        uint64_t ptile = 0;
        uint32_t prv = 0, prun = 0;
        for (; tile < tend; tile += 1) {
            const uint64_t nxt = tile + 1 < tend ? tile + 1 : tile;  // clamped: the prefetch is unconditional
            uint32_t run = 0, nstaged = 0;
            uint32_t prev = (tile && pb == '\n') ? 1u : 0u;
#pragma unroll 1
            for (uint32_t g = 0; g < WT_PIECES / 4; ++g) {
                __builtin_amdgcn_wave_barrier();
                *reinterpret_cast<uint4 *>(wptr) = n0;
                *reinterpret_cast<uint4 *>(wptr + 1024) = n1;
                *reinterpret_cast<uint4 *>(wptr + 2048) = n2;
                *reinterpret_cast<uint4 *>(wptr + 3072) = n3;
                {  // next group of this tile, or the first group of the wave's next tile
                    const bool last = g + 1 == WT_PIECES / 4;
                    p = last ? buf + (nxt << WT_SHIFT) + lo : p + 4 * PIECE_BYTES;
                    if (last) pb = buf[(nxt << WT_SHIFT) - (nxt ? 1 : 0)];
                    n0 = load16_nt(p); n1 = load16_nt(p + PIECE_BYTES);
                    n2 = load16_nt(p + 2 * PIECE_BYTES); n3 = load16_nt(p + 3 * PIECE_BYTES);
                }
                if (g == 0 && pending) store_tile(ptile, prun, prv);  // a whole group before the next wait
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const uint4 d0 = *reinterpret_cast<const uint4 *>(rptr + (0u ^ s4));
                const uint4 d1 = *reinterpret_cast<const uint4 *>(rptr + (16u ^ s4));
                const uint4 d2 = *reinterpret_cast<const uint4 *>(rptr + (32u ^ s4));
                const uint4 d3 = *reinterpret_cast<const uint4 *>(rptr + (48u ^ s4));
#ifdef FQH_IDX_NOMASK   // tuning knock-out: no newline search (one cheap use of every loaded dword keeps the LDS reads alive)
                const uint32_t m_lo = (d0.x ^ d1.y ^ d2.z ^ d3.w) == 0x12345678u ? 1u : 0u, m_hi = (d0.y ^ d1.z ^ d2.w ^ d3.x ^ d0.z ^ d0.w ^ d1.x ^ d1.w ^ d2.x ^ d2.y ^ d3.y ^ d3.z) == 0x12345678u ? 1u : 0u;
#else
                const uint32_t m_lo = eqmask16<1>(d0, 0x0A0A0A0Au) | (eqmask16<1>(d1, 0x0A0A0A0Au) << 16);
                const uint32_t m_hi = eqmask16<1>(d2, 0x0A0A0A0Au) | (eqmask16<1>(d3, 0x0A0A0A0Au) << 16);
#endif
                // line starts: the byte after a newline
                uint32_t ls_lo = (m_lo << 1) | wave_shr1(m_hi >> 31, prev);
                uint32_t ls_hi = __builtin_amdgcn_alignbit(m_hi, m_lo, 31);
                prev = ((uint32_t)__builtin_amdgcn_readlane((int)m_hi, 63)) >> 31;
                const uint32_t c = __popc(ls_lo) + __popc(ls_hi);
                const unsigned long long b1 = __ballot(c >= 1), b2 = __ballot(c >= 2), b3 = __ballot(c >= 3);
                uint32_t pre = __builtin_amdgcn_mbcnt_hi((uint32_t)(b1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b1, 0));
                pre = __builtin_amdgcn_mbcnt_hi((uint32_t)(b2 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b2, pre));
                pre = __builtin_amdgcn_mbcnt_hi((uint32_t)(b3 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b3, pre));
                uint32_t tot = (uint32_t)__popcll(b1) + (uint32_t)__popcll(b2) + (uint32_t)__popcll(b3);
                if (__ballot(c >= 4)) {
                    for (uint32_t k = 4;; ++k) {
                        const unsigned long long b = __ballot(c >= k);
                        if (!b) break;
                        pre = __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, pre));
                        tot += (uint32_t)__popcll(b);
                    }
                }
                const uint32_t ebase = g * 4 * PIECE_BYTES + lane * 64;
#ifdef FQH_IDX_NOSTAGE
                if (false) {
#else
                if (run == nstaged && run + tot <= FAST_ENTRIES) {  // uniform: stage in LDS
#endif
                    // the lane's first three line starts without a loop: positions by find-first-set, the three class bytes read in
                    // ONE LDS round trip, then the entries (ordinary reads have two to three line starts in a lane's 64 bytes;
                    // a fourth and later ones take the loop)
                    uint16_t *dst = lst + run + pre;
                    unsigned long long lsm = ((unsigned long long)ls_hi << 32) | ls_lo;
                    const uint32_t q0 = lsm ? (uint32_t)__ffsll((long long)lsm) - 1u : 0u;
                    lsm &= lsm - 1ull;
                    const uint32_t q1 = lsm ? (uint32_t)__ffsll((long long)lsm) - 1u : 0u;
                    lsm &= lsm - 1ull;
                    const uint32_t q2 = lsm ? (uint32_t)__ffsll((long long)lsm) - 1u : 0u;
                    lsm &= lsm - 1ull;
                    const uint32_t b0 = rptr[q0 ^ s4], b1 = rptr[q1 ^ s4], b2 = rptr[q2 ^ s4];
                    if (c > 0) dst[0] = (uint16_t)((ebase + q0) | ((b0 == '@') ? 0x4000u : 0u) | ((b0 == '+') ? 0x8000u : 0u));
                    if (c > 1) dst[1] = (uint16_t)((ebase + q1) | ((b1 == '@') ? 0x4000u : 0u) | ((b1 == '+') ? 0x8000u : 0u));
                    if (c > 2) dst[2] = (uint16_t)((ebase + q2) | ((b2 == '@') ? 0x4000u : 0u) | ((b2 == '+') ? 0x8000u : 0u));
                    if (__ballot(c > 3) != 0) {
                        dst += 3;
                        while (lsm) {
                            const uint32_t q = (uint32_t)__ffsll((long long)lsm) - 1u;
                            lsm &= lsm - 1ull;
                            const uint32_t b = rptr[q ^ s4];
                            *dst++ = (uint16_t)((ebase + q) | ((b == '@') ? 0x4000u : 0u) | ((b == '+') ? 0x8000u : 0u));
                        }
                    }
                    nstaged = run + tot;
                }  // else: more than FAST_ENTRIES line starts in a tile: left to the exact path
                run += tot;
            }
#ifdef FQH_IDX_NOFINISH
            prv = run;
#else
            finish_tile(tile, run, nstaged, prv, false);
#endif
            prun = run;
            ptile = tile;
            pending = true;
            __builtin_amdgcn_wave_barrier();
        }
        if (pending) store_tile(ptile, prun, prv);
    }
    // the partial tile at the end of the buffer
    if (wave0 == 0 && n_full < n_tiles) {
        const uint64_t t = n_full, tbase = t << WT_SHIFT;
        uint32_t run = 0;
        uint32_t prev = (t && buf[tbase - 1] == '\n') ? 1u : 0u;
#pragma unroll 1
        for (uint32_t j = 0; j < WT_PIECES; ++j) {
            const uint64_t off = tbase + (uint64_t)j * PIECE_BYTES + lo;
            if (tbase + (uint64_t)j * PIECE_BYTES >= len) break;  // uniform
            const uint4 v = load16(buf, off, len);
            index_piece<false, 1>(v, off, len, j * PIECE_BYTES + lo, lane, prev, run, lst, FAST_ENTRIES);
        }
        uint32_t rv;
        finish_tile(t, run, run <= FAST_ENTRIES ? run : 0u, rv, true);
        store_tile(t, run, rv);
    }
    if (lane == 0 && n_over) atomicAdd(&out->spec_fail, (unsigned long long)n_over);
}

// k_emit_fast: the fast path's emit.  Per tile: verify the tile's alignment against the true line
// index, store the record starts, and validate the one record that straddles into the tile from
// the previous one (its five line starts are in the two tiles' edge entries).  Any doubt sets
// spec_fail; error keys are never produced here.  The per-tile scalar work is done once per LANE, not
// once per wave: a wavefront takes 64 consecutive tiles.  Phase A, lane = tile: line index of the tile's
// first entry, alignment check, position of its record starts in the output, and the validation of
// the record that straddles into it (five line starts out of the two tiles' 16-byte edge blocks).
// Phase B, one iteration per tile: readlane the three scalars, one 2-byte load and one 8-byte
// store per lane, in rounds of 16 loads then 16 stores.  ~25 instructions per tile instead of ~110.
template <uint32_t EMIT_ROUND, bool DEVC>
__global__ __launch_bounds__(256) void k_emit_fast(ScanArgs a_in, DevOut *__restrict__ out) {
    // (the arguments stay where they are — pointers read back from an LDS copy are flat pointers, and a flat access makes
    // every later s_waitcnt a vmcnt(0) lgkmcnt(0); of the carry this kernel needs two words, read from device memory)
    const ScanArgs &a = a_in;
    const unsigned long long carry_nl = DEVC ? a_in.dcarry->nl_count : a_in.nl_count;
    const unsigned long long carry_base = DEVC ? a_in.dcarry->base_offset : a_in.base_offset;
    __shared__ uint16_t stage_all[4][EMIT_ROUND * 64];
    uint16_t *const stage = stage_all[threadIdx.x >> 6];
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t nwaves = (uint64_t)gridDim.x * 4;
    const uint64_t ngroups = (a.n_tiles + 63) >> 6;
    const unsigned long long r0 = carry_nl >> 2;
    const uint32_t bufsize32 = a.bufsize > 0x7FFFFFFFull ? 0x7FFFFFFFu : (uint32_t)a.bufsize;
    unsigned long long first_long = NOKEY;
    uint32_t maxlen32 = 0, fail = 0;
    for (uint64_t g = (uint64_t)blockIdx.x * 4 + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); g < ngroups; g += nwaves) {
        const uint64_t t0 = g << 6;
        // ---- phase A: lane = tile t0 + lane
        const uint64_t T = t0 + lane;
        const bool live = T < a.n_tiles;
        uint32_t cnt = 0, tp = 0, hyp = 7, cprev = 0;
        uint4 e = make_uint4(0, 0, 0, 0);
        if (live) {
            cnt = a.tile_count[T];
            tp = a.tile_prefix[T];
            const uint16_t *meta = a.fast_rs + T * FR_STRIDE + FR_EDGE;        // 8-byte aligned
            const uint2 ea = *reinterpret_cast<const uint2 *>(meta), eb = *reinterpret_cast<const uint2 *>(meta + 4);
            e = make_uint4(ea.x, ea.y, eb.x, eb.y);                           // entries 0..3 first, 4..7 last four
            hyp = meta[FR_HYP - FR_EDGE];
        }
        const unsigned long long bp = a.block_prefix[t0 >> SCAN_SHIFT];    // 64 | 2^SCAN_SHIFT: one block
        uint32_t c0 = 0, pz0 = 0, pw0 = 0;                                // tile t0 - 1, for lane 0
        if (t0) {
            c0 = a.tile_count[t0 - 1];
            const uint2 pe0 = *reinterpret_cast<const uint2 *>(a.fast_rs + (t0 - 1) * FR_STRIDE + FR_EDGE + 4);
            pz0 = pe0.x;
            pw0 = pe0.y;
        }
        cprev = wave_shr1(cnt, c0);
        const uint32_t pz = wave_shr1(e.z, pz0), pw = wave_shr1(e.w, pw0);  // previous tile's last four entries
        const unsigned long long lbase = carry_nl + 1 + bp + tp;
        const uint32_t r = (4u - ((uint32_t)lbase & 3u)) & 3u;             // entry index of the first record start
        const bool small = hyp >= FR_SMALL;   // the short tile at the end of the buffer: k_finalize_fast's
        const bool has = cnt != 0 && !small;
        bool bad = (has && hyp != r) || (small && T + 1 != a.n_tiles);
        const uint32_t nrs = has && cnt > r ? (cnt - r + 3) >> 2 : 0u;
        const unsigned long long rbase = ((lbase + r) >> 2) - r0;
        if (has && T) {  // the record that ends at entry r started in the previous tile (tile 0: k_finalize_fast)
            bad = bad || cprev < 4 || cnt < 4;
            // halfwords r .. r+4 of [previous tile's last four | this tile's first four]
            const unsigned long long A = ((unsigned long long)pw << 32) | pz, B = ((unsigned long long)e.y << 32) | e.x;
            const uint32_t sh = 16u * r;
            const unsigned long long lo = r ? (A >> sh) | (B << (64u - sh)) : A;
            const uint32_t ev0 = (uint32_t)lo & 0xFFFFu, ev1 = (uint32_t)(lo >> 16) & 0xFFFFu;
            const uint32_t ev2 = (uint32_t)(lo >> 32) & 0xFFFFu, ev3 = (uint32_t)(lo >> 48);
            const uint32_t ev4 = (uint32_t)(B >> sh) & 0xFFFFu;
            // entries with index r + k < 4 belong to the previous tile: offset - 16 KiB
            const int o0 = (int)(ev0 & 0x3FFFu) - (int)WT_BYTES;  // k = 0: always the previous tile
            const int o1 = (int)(ev1 & 0x3FFFu) - (r + 1 < 4 ? (int)WT_BYTES : 0);
            const int o2 = (int)(ev2 & 0x3FFFu) - (r + 2 < 4 ? (int)WT_BYTES : 0);
            const int o3 = (int)(ev3 & 0x3FFFu) - (r + 3 < 4 ? (int)WT_BYTES : 0);
            const int o4 = (int)(ev4 & 0x3FFFu);                  // k = 4: always this tile
            const bool ok = (ev0 & 0x4000u) && (ev2 & 0x8000u) && (o2 - o1) == (o4 - o3);
            bad = bad || !ok;
            if (!bad) {
                const uint32_t reclen = (uint32_t)(o4 - o0);
                maxlen32 = reclen > maxlen32 ? reclen : maxlen32;
                if (bufsize32 && reclen + 15 >= bufsize32) {
                    const unsigned long long rec = r0 + rbase - 1;
                    if (rec < first_long) first_long = rec;
                }
            }
        }
        if (!a.list && nrs > FR_N + FR2_N) {  // the tile's later record starts went to a list area this launch does not have
            bad = true;                        // (k_index_fast wrote them to a dummy): the host reruns with the workspace
            out->need_list = 1;
        }
        fail |= bad ? 1u : 0u;
        const uint32_t n_emit = bad ? 0u : nrs;   // a failed tile stores nothing (the result is discarded anyway)
        // ---- phase B: rounds of 16 tiles: 16 loads, then 16 stores.  Stores share vmcnt with loads on
        // gfx950, so a load that is consumed a few stores after it was issued waits for those stores'
        // acknowledgements; in rounds, the only wait is the one in front of a round's first store and
        // the previous round's stores have had a whole load latency to land.
        const uint32_t ntl = (uint32_t)(a.n_tiles - t0 < 64 ? a.n_tiles - t0 : 64);
        const uint16_t *const rs0 = a.fast_rs + t0 * FR_STRIDE + lane;
        // ---- the streamlined phase B: every tile of the group keeps its record starts in its one line, all of them fit
        // the caller's array, and no record the fast path can validate reaches the Buffer's limit (such a record would
        // hold a whole tile with at most four line starts, which fails the fast path in k_index_fast already: the
        // per-record "too long" test only matters for limits below two tiles).  Then a tile costs three v_readlane, one
        // 8-byte store with a scalar base and the record-length maximum: ~16 instructions, no branch, no LDS.  The
        // phase is bound by instruction issue (vector and scalar unit each ~55 % busy with the generic loop below), not
        // by its 550 MB of traffic.
        const bool no_long = bufsize32 == 0 || bufsize32 > 2 * WT_BYTES + 15;
        if (no_long && __ballot(n_emit > FR_N + FR2_N || (n_emit != 0 && a.rec_start != nullptr && rbase + n_emit > a.cap)) == 0) {
            const uint64_t p0 = (uint64_t)(uintptr_t)(a.rec_start + rbase);   // lane = tile here: where its records go
            const uint32_t plo = (uint32_t)p0, phi = (uint32_t)(p0 >> 32);
            const unsigned long long vbase0 = carry_base + (t0 << WT_SHIFT);
            // (an explicit global pointer: rebuilt from integers it would be a flat one, and flat stores count on lgkmcnt too)
            typedef __attribute__((address_space(1))) uint64_t g_u64;
            // TWO: some tile of the group has record starts in its second line (reads shorter than ~140 bp) — that line is
            // loaded and stored for every tile of the group then, with 8 tiles per round for the registers' sake
            auto phase_b = [&](auto store_tag, auto two_tag) {
                constexpr bool STORE = decltype(store_tag)::value, TWO = decltype(two_tag)::value;
                constexpr uint32_t R = TWO ? EMIT_ROUND / 2 : EMIT_ROUND;
                uint32_t qa[R], qb[R], ra[TWO ? R : 1], rb[TWO ? R : 1];
                const uint16_t *const rs2 = a.fast_rs + fr2_off(a.n_tiles) + t0 * FR2_N + lane;
                auto body = [&](uint32_t i, uint32_t o, uint32_t o2) {
                    const uint32_t n = (uint32_t)__builtin_amdgcn_readlane((int)n_emit, (int)i);
                    if (n == 0) return;  // (wave-uniform)
                    const uint32_t pl = (uint32_t)__builtin_amdgcn_readlane((int)plo, (int)i);
                    const uint32_t ph = (uint32_t)__builtin_amdgcn_readlane((int)phi, (int)i);
                    const unsigned long long vb = vbase0 + ((unsigned long long)i << WT_SHIFT);
                    g_u64 *const dst = reinterpret_cast<g_u64 *>(((uint64_t)ph << 32) | pl);
                    if (lane < n && lane < FR_N) {
                        if (STORE) __builtin_nontemporal_store((uint64_t)(vb + o), dst + lane);
                        const uint32_t reclen = o - wave_shr1(o, o);  // lane 0: 0 (the record that ends there was measured in phase A)
                        maxlen32 = reclen > maxlen32 ? reclen : maxlen32;
                    }
                    if (TWO && n > FR_N) {  // (wave-uniform) record starts FR_N .. n - 1 out of the second line
                        const uint32_t last1 = (uint32_t)__builtin_amdgcn_readlane((int)o, (int)(FR_N - 1));
                        if (lane < n - FR_N) {
                            if (STORE) __builtin_nontemporal_store((uint64_t)(vb + o2), dst + FR_N + lane);
                            const uint32_t reclen = o2 - wave_shr1(o2, last1);
                            maxlen32 = reclen > maxlen32 ? reclen : maxlen32;
                        }
                    }
                };
#pragma unroll
                for (uint32_t j = 0; j < R; ++j) {
                    qa[j] = rs0[j * FR_STRIDE];  // (the arrays have 64 tiles of slack)
                    if (TWO) ra[j] = rs2[j * FR2_N];
                }
#pragma unroll 1
                for (uint32_t ib = 0; ib < 64; ib += 2 * R) {
#pragma unroll
                    for (uint32_t j = 0; j < R; ++j) {
                        qb[j] = rs0[(ib + R + j) * FR_STRIDE];
                        if (TWO) rb[j] = rs2[(ib + R + j) * FR2_N];
                    }
#pragma unroll
                    for (uint32_t j = 0; j < R; ++j) body(ib + j, qa[j], TWO ? ra[j] : 0u);
                    if (ib + 2 * R < 64) {
#pragma unroll
                        for (uint32_t j = 0; j < R; ++j) {
                            qa[j] = rs0[(ib + 2 * R + j) * FR_STRIDE];
                            if (TWO) ra[j] = rs2[(ib + 2 * R + j) * FR2_N];
                        }
                    }
#pragma unroll
                    for (uint32_t j = 0; j < R; ++j) body(ib + R + j, qb[j], TWO ? rb[j] : 0u);
                }
            };
            const bool two = __ballot(n_emit > FR_N) != 0;
            if (a.rec_start) {
                if (two) phase_b(std::true_type{}, std::true_type{});
                else phase_b(std::true_type{}, std::false_type{});
            } else {
                if (two) phase_b(std::false_type{}, std::true_type{});
                else phase_b(std::false_type{}, std::false_type{});
            }
            continue;
        }
        for (uint32_t ib = 0; ib < ntl; ib += EMIT_ROUND) {
            uint32_t q[EMIT_ROUND];
#pragma unroll
            for (uint32_t j = 0; j < EMIT_ROUND; ++j) q[j] = rs0[(ib + j) * FR_STRIDE];  // the array has 64 tiles of slack
            // through LDS, so that the store loop below is ONE body indexed by j (unrolled, its two
            // paths times 16 need 180+ VGPRs and halve the occupancy)
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (uint32_t j = 0; j < EMIT_ROUND; ++j) stage[j * 64 + lane] = (uint16_t)q[j];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll 1
            for (uint32_t j = 0; j < EMIT_ROUND; ++j) {
                const uint32_t i = ib + j;
                if (i >= ntl) break;
                const uint32_t o = stage[j * 64 + lane];
                const uint32_t n = (uint32_t)__builtin_amdgcn_readlane((int)n_emit, (int)i);
                if (n == 0) continue;
                const unsigned long long rb = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(rbase >> 32), (int)i) << 32) |
                                              (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)rbase, (int)i);
                const unsigned long long vbase = carry_base + ((t0 + i) << WT_SHIFT);
                const bool cap_ok = rb + n <= a.cap;
                uint64_t *__restrict__ rs = a.rec_start ? a.rec_start + rb : nullptr;
                const uint16_t *__restrict__ tl = a.list ? a.list + (t0 + i) * a.list_cap + 8 : nullptr;
                if (n <= FR_N) {  // the common case: everything is in the tile's line
                    const uint32_t oprev = wave_shr1(o, 0u);
                    if (lane < n) {
                        if (rs && (cap_ok || rb + lane < a.cap)) {
                            __builtin_nontemporal_store((uint64_t)(vbase + o), rs + lane);  // written once, not read here
                        }
                        if (lane) {  // record lane-1 of the tile lies inside it: its length
                            const uint32_t reclen = o - oprev;
                            maxlen32 = reclen > maxlen32 ? reclen : maxlen32;
                            if (bufsize32 && reclen + 15 >= bufsize32) {
                                const unsigned long long rec = r0 + rb + lane - 1;
                                if (rec < first_long) first_long = rec;
                            }
                        }
                    }
                    continue;
                }
                uint32_t carry = 0;  // offset of record start mb - 1 (last lane of the previous group of 64)
                for (uint32_t mb = 0; mb < n; mb += 64) {
                    const uint32_t m = mb + lane;
                    const uint32_t oo = m < FR_N ? o
                                        : m >= n ? 0u
                                        : m < FR_N + FR2_N ? (uint32_t)a.fast_rs[fr2_off(a.n_tiles) + (t0 + i) * FR2_N + (m - FR_N)]
                                                           : (tl ? (uint32_t)tl[m] : 0u);
                    const uint32_t oprev = wave_shr1(oo, carry);
                    carry = (uint32_t)__builtin_amdgcn_readlane((int)oo, 63);
                    if (m < n) {
                        if (rs && (cap_ok || rb + m < a.cap)) __builtin_nontemporal_store((uint64_t)(vbase + oo), rs + m);
                        if (m) {  // record m-1 of the tile lies inside it: its length
                            const uint32_t reclen = oo - oprev;
                            maxlen32 = reclen > maxlen32 ? reclen : maxlen32;
                            if (bufsize32 && reclen + 15 >= bufsize32) {
                                const unsigned long long rec = r0 + rb + m - 1;
                                if (rec < first_long) first_long = rec;
                            }
                        }
                    }
                }
            }
        }
    }
    const unsigned long long fl = wave_min_u64(first_long);
    const unsigned long long ml = wave_max_u64(maxlen32);
    if (bufsize32 > 2 * WT_BYTES + 15 && ml + 15 >= bufsize32) fail = 1;  // (cannot happen, see the streamlined phase B: left to the exact path if it does)
    if (lane == 0) {
        if (fl != NOKEY) atomicMin(&out->first_long, fl);
        if (ml) atomicMax(&out->max_len, ml);
    }
    if (__ballot(fail) && lane == 0) atomicAdd(&out->spec_fail, 1ull);
}

// k_finalize_fast: one thread.  Everything k_emit_fast leaves out: the record in progress at the
// chunk start (validated with the carry), the lines after the last complete in-tile group, the EOF
// rule, carry-out and summary.  Needs at least four line starts in the first and in the last tile;
// otherwise, or on any violation, it sets spec_fail.
template <bool DEVC>
__global__ void k_finalize_fast(ScanArgs a_in, DevOut *__restrict__ out) {
    // (one thread works; its small arrays live in LDS: as private arrays they were 400 bytes of scratch per lane, and the
    // runtime sizes a queue's scratch for a full device of such lanes: 200 MB for the sake of this kernel)
    __shared__ long long sh_ll[4 + 5 + 4 + 12];
    __shared__ uint32_t sh_u32[5 + 4 + 12];
    FQH_ARGS_WITH_CARRY(DEVC, true, a_in)
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    bool fail = out->spec_fail != 0 || a.n_tiles == 0;
    const unsigned long long E = a.n_tiles ? a.block_prefix[a.n_blocks] : 0ull;
    const bool lastnl = a.len > 0 && a.buf[a.len - 1] == '\n';
    const uint64_t tl_last = a.n_tiles ? a.n_tiles - 1 : 0;
    const uint32_t hyp_last = a.n_tiles ? a.fast_rs[tl_last * FR_STRIDE + FR_HYP] : 7u;
    const bool small = hyp_last >= FR_SMALL;  // a short tile at the end: all of its (< 8) entries are in its edge slots
    // a shard's prescan runs under a made-up carry (the shard begins the file): what it is asked for are the newline
    // count and the last line starts; what depends on the line phase is validated by the rescan under the true carry
    const bool chk = !a.prescan;
    if (!fail && (a.tile_count[0] < 4 || E < 8)) fail = true;
    if (!fail && (small ? (a.n_tiles < 2 || a.tile_count[tl_last] >= 8 || a.tile_count[tl_last - 1] < 4) : a.tile_count[tl_last] < 4))
        fail = true;
    unsigned long long max_len = out->max_len, first_long = out->first_long;
    const unsigned long long r0 = a.nl_count >> 2;
    long long *const recent = sh_ll;
    for (int i = 0; i < 4; ++i) recent[i] = 0;
    unsigned long long n_newlines = 0, T = a.nl_count;
    if (!fail) {
        // ---- chunk start: the virtual line start at offset 0 and the record in progress
        const unsigned long long lbase0 = a.nl_count + 1;
        if (a.v_start) {
            const unsigned long long l = a.nl_count;
            const uint8_t b = a.buf[0];
            if (((l & 3) == 0 && b != '@') || ((l & 3) == 2 && b != '+')) fail = chk;
        }
        const uint32_t rr = (4u - ((uint32_t)lbase0 & 3u)) & 3u;  // first entry of tile 0 that starts a record
        long long *const S = sh_ll + 4;   // [5] line starts rr-4 .. rr of tile 0, chunk-relative
        uint32_t *const cls = sh_u32;     // [5] bit 0 '@', bit 1 '+' where known from this chunk (entries); 3 = not checkable here
        for (int k = 0; k < 5; ++k) {
            const int j = (int)rr - 4 + k;
            if (j >= 0) {
                const uint32_t e = a.fast_rs[FR_EDGE + j];
                S[k] = (long long)(e & 0x3FFFu);
                cls[k] = ((e & 0x4000u) ? 1u : 0u) | ((e & 0x8000u) ? 2u : 0u);
            } else {
                // line starts before entry 0, most recent first: the virtual start at offset 0 (if a
                // line starts there), then the carry's
                const int idx = -j - 1;
                S[k] = (a.v_start && idx == 0) ? 0 : -(long long)a.back[idx < 4 ? idx : 3];
                cls[k] = 3;
            }
        }
        // class checks of the entries 0..rr (lines of the record in progress that start in this chunk)
        for (int k = 0; k < 5; ++k) {
            const int j = (int)rr - 4 + k;
            if (j < 0) continue;
            const unsigned long long l = lbase0 + j;
            if ((l & 3) == 0 && !(cls[k] & 1u)) fail = chk;
            if ((l & 3) == 2 && !(cls[k] & 2u)) fail = chk;
        }
        // length rule and length of the record that ends at entry rr (needs >= 1 earlier record line)
        if (lbase0 + rr >= 4 && !a.head_unchecked) {
            if ((S[4] - S[3]) != (S[2] - S[1])) fail = chk;
            const unsigned long long reclen = (unsigned long long)(S[4] - S[0]);
            if (reclen > max_len) max_len = reclen;
            const unsigned long long rec = ((lbase0 + rr) >> 2) - 1;
            if (a.bufsize && reclen + 15 >= a.bufsize && rec < first_long) first_long = rec;
        }
        if (a.rec_start && a.cap > 0) a.rec_start[0] = a.base_offset - a.back[a.nl_count & 3];

        // ---- chunk end: the last four entries (with their class bits)
        uint32_t *const le = sh_u32 + 5;   // [4]
        long long *const ls = sh_ll + 9;   // [4]
        if (!small) {
            const uint16_t *el = a.fast_rs + tl_last * FR_STRIDE + FR_EDGE + 4;
            for (int k = 0; k < 4; ++k) {  // k = 0: most recent
                le[k] = el[3 - k];
                ls[k] = (long long)((tl_last << WT_SHIFT) + (le[k] & 0x3FFFu));
            }
        } else {
            // the short tail tile: k_emit_fast skipped it.  Its entries behind the previous tile's last four: every
            // record that ends at one of them has its five line starts in this list (src/records.rs:201-247)
            const uint32_t cnt = a.tile_count[tl_last];
            uint32_t *const ce = sh_u32 + 9;   // [12]
            long long *const cs = sh_ll + 13;  // [12]
            const uint16_t *ep = a.fast_rs + (tl_last - 1) * FR_STRIDE + FR_EDGE + 4;
            const uint16_t *eq = a.fast_rs + tl_last * FR_STRIDE + FR_EDGE;
            for (uint32_t k = 0; k < 4; ++k) {
                ce[k] = ep[k];
                cs[k] = (long long)(((tl_last - 1) << WT_SHIFT) + (ce[k] & 0x3FFFu));
            }
            for (uint32_t k = 0; k < cnt; ++k) {
                ce[4 + k] = eq[k];
                cs[4 + k] = (long long)((tl_last << WT_SHIFT) + (ce[4 + k] & 0x3FFFu));
            }
            const unsigned long long lq0 = a.nl_count + 1 + tile_pref(a, tl_last);  // line index of the tile's entry 0
            if (!(hyp_last & 4u) && (hyp_last & 3u) != ((4u - ((uint32_t)lq0 & 3u)) & 3u)) fail = chk;  // counted under another alignment
            for (uint32_t j = 0; j < 4 + cnt; ++j) {
                const unsigned long long l = lq0 + j - 4;
                if ((l & 3) == 0 && !(ce[j] & 0x4000u)) fail = chk;
                if ((l & 3) == 2 && !(ce[j] & 0x8000u)) fail = chk;
                if (j >= 4 && (l & 3) == 0) {  // a record ends in front of this entry
                    if ((cs[j] - cs[j - 1]) != (cs[j - 2] - cs[j - 3])) fail = chk;
                    const unsigned long long reclen = (unsigned long long)(cs[j] - cs[j - 4]);
                    if (reclen > max_len) max_len = reclen;
                    const unsigned long long rec = (l >> 2) - 1;
                    if (a.bufsize && reclen + 15 >= a.bufsize && rec < first_long) first_long = rec;
                    const unsigned long long r = (l >> 2) - r0;
                    if (a.rec_start && r < a.cap) a.rec_start[r] = a.base_offset + (unsigned long long)cs[j];
                }
            }
            for (uint32_t k = 0; k < 4; ++k) {
                le[k] = ce[4 + cnt - 1 - k];
                ls[k] = cs[4 + cnt - 1 - k];
            }
        }
        n_newlines = E + (lastnl ? 1 : 0);
        T = a.nl_count + n_newlines;
        // global line index of the most recent entry
        const unsigned long long l_last = a.nl_count + E;  // entries are lines nl_count+1 .. nl_count+E
        // entries after the last complete in-tile group were not class-checked: the last group start g
        // (largest line index == 0 mod 4 among the last four entries) and what follows it
        for (int k = 0; k < 4; ++k) {
            const unsigned long long l = l_last - k;
            // a group starting at l was validated in its tile only if its fifth line start exists as an entry
            const bool group_done = (l & 3) == 0 ? (l + 4 <= l_last) : ((l & ~3ull) + 4 <= l_last);
            if (group_done) continue;
            if ((l & 3) == 0 && !(le[k] & 0x4000u)) fail = chk;
            if ((l & 3) == 2 && !(le[k] & 0x8000u)) fail = chk;
        }
        if (lastnl) {
            const unsigned long long lv = a.nl_count + 1 + E;  // a line would start at offset len
            if ((lv & 3) == 0) {  // a record ends exactly at the chunk end: lines lv-4 .. lv-1 = the last four entries
                if ((((long long)a.len - ls[0]) != (ls[1] - ls[2]))) fail = chk;
                const unsigned long long reclen = (unsigned long long)((long long)a.len - ls[3]);
                if (reclen > max_len) max_len = reclen;
                const unsigned long long rec = (lv >> 2) - 1;
                if (a.bufsize && reclen + 15 >= a.bufsize && rec < first_long) first_long = rec;
                const unsigned long long r = (lv >> 2) - r0;
                if (a.rec_start && r < a.cap) a.rec_start[r] = a.base_offset + a.len;
            }
            recent[0] = (long long)a.len; recent[1] = ls[0]; recent[2] = ls[1]; recent[3] = ls[2];
        } else {
            recent[0] = ls[0]; recent[1] = ls[1]; recent[2] = ls[2]; recent[3] = ls[3];
        }
        const unsigned long long col = (unsigned long long)((long long)a.len - recent[0]);
        if (a.is_final && ((T & 3) != 0 || col > 0)) fail = chk;  // truncated: the exact path reports it
    }
    out->spec_fail = fail ? 1 : 0;
    out->stats_commit = (fail || out->stats_declined) ? 0 : 1;
    out->decl_b = out->decl_batches;
    out->decl_l = out->decl_lines;
    out->min_key = NOKEY;
    out->first_long = first_long;
    out->max_len = max_len;
    out->total_entries = E;
    out->lastnl = lastnl;
    for (int i = 0; i < 4; ++i) out->recent[i] = recent[i];
    out->n_newlines = n_newlines;
    out->final_key = NOKEY;
    out->n_records = (T >> 2) - r0;
    out->end_off = recent[T & 3];
    out->err_start = recent[T & 3];
    out->err_need = 0;
    out->tail_len = (unsigned long long)((long long)a.len - recent[T & 3]);
    publish_and_reset(a, out);
}

// ---------------------------------------------------------------------------------------------
// The shard exchange without host hops: what a rank contributes, the fold over the ranks in front
// of it, and what it contributes to the sum at the end.  One thread each.
__global__ void k_shard_words(const DevOut *__restrict__ out, const DevOut *__restrict__ mirror, uint64_t len,
                              unsigned long long *__restrict__ w) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    w[0] = len;
    w[1] = out->n_newlines;
    w[2] = out->total_entries + out->lastnl;
    for (int i = 0; i < 4; ++i) {  // (as resolve() builds the carry-out of a chunk that began the file)
        const long long v = (long long)len - out->recent[i];
        w[3 + i] = v < 0 ? 0ull : ((unsigned long long)v > len ? len : (unsigned long long)v);
    }
    // (the finalize kernel has reset the accumulators behind the copy it published)
    w[7] = (mirror->spec_fail || mirror->overflow) ? 1ull : 0ull;
}
// fqh_carry_combine over the rows of the ranks in front of `rank` (src of the host version: scan_dispatch.hip)
__global__ void k_carry_fold(const unsigned long long *__restrict__ all, int n_ranks, int rank, DevCarry *__restrict__ dc,
                             DevCarry *__restrict__ hc, DevOut *__restrict__ out) {
    __shared__ DevCarry cs[2];  // (in LDS, not in scratch: see FQH_ARGS_WITH_CARRY)
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    DevCarry *c = &cs[0], *n = &cs[1];
    *c = DevCarry{};
    unsigned long long any_fail = 0;
    for (int r = 0; r < n_ranks; ++r) any_fail |= all[r * SHARD_WORDS + 7];
    for (int r = 0; r < rank; ++r) {
        const unsigned long long *w = all + r * SHARD_WORDS;
        const unsigned long long len = w[0], nls = w[2];
        *n = DevCarry{};
        n->base_offset = c->base_offset + len;
        n->nl_count = c->nl_count + w[1];
        int k = 0;
        for (; k < 4 && (unsigned long long)k < nls; ++k) n->back[k] = w[3 + k];
        if (len == 0) {
            for (int i = 0; i < 4; ++i) n->back[i] = c->back[i];
        } else {
            for (int j = 0; k < 4; ++k, ++j) {
                const unsigned long long v = c->back[j < 4 ? j : 3] + len;
                n->back[k] = v > n->base_offset ? n->base_offset : v;
            }
        }
        DevCarry *t = c;
        c = n;
        n = t;
    }
    c->any_fail = any_fail;
    *dc = *c;
    *hc = *c;  // pinned: the host reads it after the stream has drained (fqh_scan_finish)
    if (any_fail) out->spec_fail = 1;  // nothing of this launch is used
}
__global__ void k_shard_counts(const DevOut *__restrict__ out, const DevOut *__restrict__ mirror, const DevCarry *__restrict__ dc,
                               unsigned long long *__restrict__ counts) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    counts[0] = out->n_records;
    // (a fast path that failed under the true carry: the host reruns the exact path in fqh_scan_finish, after the sum)
    counts[1] = (out->final_key != NOKEY || mirror->spec_fail || mirror->overflow || dc->any_fail) ? 1ull : 0ull;
}
void launch_shard_words(hipStream_t s, const DevOut *out, const DevOut *mirror, uint64_t len, uint64_t *d_words) {
    hipLaunchKernelGGL(k_shard_words, dim3(1), dim3(64), 0, s, out, mirror, len, (unsigned long long *)d_words);
}
void launch_carry_fold(hipStream_t s, const uint64_t *d_all, int n_ranks, int rank, DevCarry *dc, DevCarry *hc, DevOut *out) {
    hipLaunchKernelGGL(k_carry_fold, dim3(1), dim3(64), 0, s, (const unsigned long long *)d_all, n_ranks, rank, dc, hc, out);
}
void launch_shard_counts(hipStream_t s, const DevOut *out, const DevOut *mirror, const DevCarry *dc, uint64_t *d_counts) {
    hipLaunchKernelGGL(k_shard_counts, dim3(1), dim3(64), 0, s, out, mirror, dc, (unsigned long long *)d_counts);
}

void launch_index(hipStream_t s, const uint8_t *buf, uint64_t len, uint16_t *list, uint32_t list_cap,
                  uint32_t *tile_count, uint16_t *fast_rs, uint64_t n_tiles, DevOut *out, int n_cu, bool fast) {
    if (!n_tiles) return;
    // persistent grid = exactly the blocks that are resident at once: a static round-robin of tiles over
    // a grid with one non-resident block per CU would run that block as a tail
    static std::atomic<int> occ[2];   // (zero-initialised; two host threads with a context each may be here at once: both find the same answer)
    int oc = occ[fast].load(std::memory_order_relaxed);
    if (!oc) {
        int o = 0;
        const hipError_t e = fast ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&o, k_index_fast, 256, 0)
                                  : hipOccupancyMaxActiveBlocksPerMultiprocessor(&o, k_index_t, 256, 0);
        if (e != hipSuccess || o < 1) o = 4;
        oc = o > 8 ? 8 : o;
        occ[fast].store(oc, std::memory_order_relaxed);
    }
    uint64_t blocks = (n_tiles + 3) / 4;
    const uint64_t maxb = (uint64_t)(n_cu > 0 ? n_cu : 256) * oc;
    if (blocks > maxb) blocks = maxb;
    if (fast)  // (the entry count travels in the tile's line: the prefix scan leaves the dense copy)
        hipLaunchKernelGGL(k_index_fast, dim3((uint32_t)blocks), dim3(256), 0, s, buf, len, list, list_cap,
                           fast_rs, n_tiles, out);
    else
        hipLaunchKernelGGL(k_index_t, dim3((uint32_t)blocks), dim3(256), 0, s, buf, len, list, list_cap, tile_count,
                           n_tiles, out);
}
void launch_emit_fast(hipStream_t s, const ScanArgs &a, DevOut *out, int n_cu) {
    if (a.n_tiles) {
        const uint64_t ngroups = (a.n_tiles + 63) >> 6;  // 64 tiles per wavefront and round
        uint64_t blocks = (ngroups + 3) / 4;
#ifndef FQH_EMIT_BPC
#define FQH_EMIT_BPC 4
#endif
#ifndef FQH_EMIT_ROUND
#define FQH_EMIT_ROUND 16
#endif
        const uint64_t maxb = (uint64_t)(n_cu > 0 ? n_cu : 256) * FQH_EMIT_BPC;
        if (blocks > maxb) blocks = maxb;
        if (a.dcarry) hipLaunchKernelGGL((k_emit_fast<FQH_EMIT_ROUND, true>), dim3((uint32_t)blocks), dim3(256), 0, s, a, out);
        else hipLaunchKernelGGL((k_emit_fast<FQH_EMIT_ROUND, false>), dim3((uint32_t)blocks), dim3(256), 0, s, a, out);
    }
}
void launch_finalize_fast(hipStream_t s, const ScanArgs &a, DevOut *out) {
    if (a.dcarry) hipLaunchKernelGGL(k_finalize_fast<true>, dim3(1), dim3(64), 0, s, a, out);
    else hipLaunchKernelGGL(k_finalize_fast<false>, dim3(1), dim3(64), 0, s, a, out);
}
void launch_prefix(hipStream_t s, uint32_t *tile_count, const uint16_t *fast_rs, uint32_t *tile_prefix,
                   uint64_t *block_prefix, uint64_t n_tiles, uint64_t n_blocks) {
    if (!n_tiles) return;
    if (fast_rs)
        hipLaunchKernelGGL(k_prefix_local, dim3((uint32_t)n_blocks), dim3(256), 0, s,
                           reinterpret_cast<const uint32_t *>(fast_rs + FR_CNT), FR_STRIDE / 2, tile_count, tile_prefix,
                           block_prefix, n_tiles);
    else
        hipLaunchKernelGGL(k_prefix_local, dim3((uint32_t)n_blocks), dim3(256), 0, s, (const uint32_t *)tile_count, 1u,
                           (uint32_t *)nullptr, tile_prefix, block_prefix, n_tiles);
    hipLaunchKernelGGL(k_prefix_top, dim3(1), dim3(1024), 0, s, block_prefix, n_blocks);
}
void launch_emit(hipStream_t s, const ScanArgs &a, DevOut *out, int n_cu) {
    if (!a.n_tiles) return;
    static std::atomic<int> occ_a;
    int occ = occ_a.load(std::memory_order_relaxed);
    if (!occ) {
        int o = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&o, k_emit<false>, 256, 0) != hipSuccess || o < 1) o = 4;
        occ = o > 8 ? 8 : o;
        occ_a.store(occ, std::memory_order_relaxed);
    }
    uint64_t blocks = ((a.n_tiles + EMIT_G - 1) / EMIT_G + 3) / 4;
    const uint64_t maxb = (uint64_t)(n_cu > 0 ? n_cu : 256) * occ;
    if (blocks > maxb) blocks = maxb;
    if (a.dcarry) hipLaunchKernelGGL(k_emit<true>, dim3((uint32_t)blocks), dim3(256), 0, s, a, out);
    else hipLaunchKernelGGL(k_emit<false>, dim3((uint32_t)blocks), dim3(256), 0, s, a, out);
}
void launch_finalize(hipStream_t s, const ScanArgs &a, DevOut *out) {
    if (a.dcarry) hipLaunchKernelGGL(k_finalize<true>, dim3(1), dim3(64), 0, s, a, out);
    else hipLaunchKernelGGL(k_finalize<false>, dim3(1), dim3(64), 0, s, a, out);
}

}  // namespace fqh
