// stats_kernels.hip — per-position Phred-quality and base-composition histograms over a full tile index
// (k_stats_oct, the second-pass kernel behind chunked / sharded / error-limited fqh_stats calls;
// whole-file calls take the single-pass k_scan_stats of fused_kernels.hip), the synthetic-FASTQ generator and
// the streaming-read ceiling probe.  Accessors as src/records.rs:75-90 (one trailing '\r' trimmed), alphabets as
// src/records.rs:19-33.  Counters are integers: addition commutes, so the result is bit-exact whatever the order.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "stats_dev.h"

namespace fqh {

uint32_t stats_blocks(int n_cu) { return (uint32_t)(n_cu > 0 ? n_cu : 256); }

template <uint32_t NSL, bool DBG>
__global__ __launch_bounds__(SO_THREADS) void k_stats_oct(StatsArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t hist[];  // sequence + quality regions, the staged lists
    const uint32_t lc = a.lc;
    const uint32_t listw = NSL >= 8 ? SO_LISTW : SO_LISTW_REG;  // list entries of a tile kept in LDS; the rest is read from memory
    // Reads longer than the 256 rows the LDS holds are counted in PASSES of 256 columns (launch_stats_oct): this launch
    // counts columns col0 .. col0 + lc - 1 of every line and nothing else.  Totals and the columns beyond the caller's
    // lmax (plain arithmetic on the line lengths) belong to pass 0; every pass flags the sequence lines in which it met
    // an 'N' or worse (a record's bit in a.flagmap says whether an earlier pass has counted it), and the last pass
    // looks at what sequence lines hold beyond the caller's rows for that purpose only.
    const uint32_t col0 = a.col0;
    const bool pass0 = col0 == 0;
    auto seg = [&](uint32_t l) -> uint32_t { return l > col0 ? (l - col0 < lc ? l - col0 : lc) : 0u; };
    for (uint32_t i = threadIdx.x; i < SO_WORDS; i += SO_THREADS) hist[i] = 0;
    __syncthreads();

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // a scalar: so is the tile
    const uint32_t m4 = (lane & 7u) * 4u, g8 = lane >> 3;
    // Two ways of having a tile's words at hand a tile ahead (DMA below).  Without: the wave's staged list sits right
    // behind the histogram.  With: two tile slots per wave, [listw list entries (u16) | 64 words, six of them the
    // tile's], BEHIND every address a lane that counts nothing can form (such a lane subtracts 0 wherever its bytes
    // point, SO_ADDR_SPAN): the slots are filled by LDS-DMA, and a read-modify-write of the LDS, even of +0, is not
    // atomic against that (found the hard way: lists that lost entries whenever a batch had idle lanes).
    constexpr bool DMA = NSL >= 8;
    uint8_t *const wslot = DMA ? reinterpret_cast<uint8_t *>(hist) + SO_ADDR_SPAN + wv * (2u * SO_SLOT_BYTES)
                               : reinterpret_cast<uint8_t *>(hist + SO_WORDS) + wv * (2u * SO_LISTW_REG);
    // The address registers assume the histogram starts at LDS address 0 (it is the kernel's only
    // LDS object); a shared-memory pointer is its LDS address in the low 32 bits.
    if ((uint32_t)(uintptr_t)hist != 0) __builtin_trap();
    SoLane c;
    c.slots = 0;
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) {
        const uint32_t j = k ^ (g8 & 3u);
        c.sel[k] = 0x0C0C0004u + k + (j << 8);
        c.slots |= (((lane & 7u) + 8u * j) * 4u) << (8u * k);
    }
    SoAcc acc = {0, 0, 0};
    SoTotals T = {0, 0, 0, 0};
    bool cr_seen = false;  // wave-uniform: a line of this wave's tiles ended in "\r\n" -- look for it from the next tile on
    bool cr_any = false;   // wave-uniform: this wave trimmed a '\r' (pass 0 tells the later passes through a.cr_flag)
    const bool cr_file = (pass0 || !a.cr_flag) ? true : __builtin_nontemporal_load(a.cr_flag) != 0;
    // DBG (FQH_STATS_DBG & 8192): cycles this wave spent waiting for a tile's words, staging its list, working out
    // its lines, and in its batches (added to qual_hist[0..3] at the end; tools/exp_statsdbg.py prints them)
    unsigned long long dbgt[4] = {0, 0, 0, 0};
    SoShape<NSL> S = {};
    S.key = 0xFFFFFFFFu;  // no P has this key: the first batch works the shape out

    // What a wave needs to know about a tile before it can start on it is loaded one tile ahead (the per-tile chain
    // count -> prefix -> list -> '\r' bytes -> first dwords is otherwise paid in full, 256 times per wave: 2.5 of the
    // kernel's 6 ms): six words — 0 the tile's count, 1 its prefix, 2 the next tile's count, 3 that tile's first entry,
    // 4-5 the block prefix — and the first listw list entries.
    //   * Reads of up to 160 bp (NSL <= 5): into registers (one for the six words, four for the list), staged with two
    //     ds_write when the tile's turn comes.
    //   * The 256-column variant has no registers to spare: held in registers, these words were the first thing the
    //     allocator spilled, and a spill waits for the very load it was meant to hide (3 000 cycles per tile).  There
    //     they go STRAIGHT INTO LDS (global_load_lds), into the slot the wave is not working from.  (Not for the other
    //     variants: next to LDS-DMA the compiler waits vmcnt(0) for every ordinary load, which costs the 150 bp shape
    //     the overlap of a batch's loads with the batch before it: 3.7 instead of 3.2 ms.)
    struct TilePre {
        uint32_t meta;
        uint2 l0, l1;   // list entries 4 lane .. 4 lane + 3 and 256 + 4 lane .. + 3
    };
    TilePre nextP;
    typedef const __attribute__((address_space(1))) void *gptr_t;
    typedef __attribute__((address_space(3))) void *lptr_t;
    auto prefetch = [&](uint64_t t, uint32_t slot) {
        const uint64_t tc = t < a.n_tiles ? t : a.n_tiles - 1;  // clamped: the loads are unconditional
        uint8_t *const dst = wslot + slot * SO_SLOT_BYTES;
        if constexpr (DMA) {
            const uint32_t *__restrict__ tl = reinterpret_cast<const uint32_t *>(a.list + tc * a.list_cap) + lane;  // list_cap >= 512
            __builtin_amdgcn_global_load_lds((gptr_t)tl, (lptr_t)dst, 4, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)(tl + 64), (lptr_t)(dst + 256), 4, 0, 0);
        } else {
            const uint16_t *__restrict__ tl = a.list + tc * a.list_cap;
            nextP.l0 = *reinterpret_cast<const uint2 *>(tl + lane * 4);
            nextP.l1 = *reinterpret_cast<const uint2 *>(tl + 256 + lane * 4);  // list_cap >= 512
        }
        const uint64_t t1 = tc + 1 < a.n_tiles ? tc + 1 : tc;
        const uint32_t *mp = a.tile_count + tc;
        if (lane == 1) mp = a.tile_prefix + tc;
        if (lane == 2) mp = a.tile_count + t1;
        if (lane == 3) mp = reinterpret_cast<const uint32_t *>(a.list + t1 * a.list_cap);  // (list_cap is even)
        if (lane == 4 || lane == 5)
            mp = reinterpret_cast<const uint32_t *>(a.block_prefix + (tc >> SCAN_SHIFT)) + (lane - 4);
        if constexpr (DMA) __builtin_amdgcn_global_load_lds((gptr_t)mp, (lptr_t)(dst + 2u * SO_LISTW), 4, 0, 0);
        else nextP.meta = *mp;
    };
    const uint64_t tstride = (uint64_t)gridDim.x * SO_WAVES;
    uint64_t tile = (uint64_t)blockIdx.x * SO_WAVES + wv;
    uint32_t slot = 0;
    if (tile < a.n_tiles && a.len >= 4) prefetch(tile, 0);
    if (DBG && (a.dbg & 65536u) && wv >= 8) tile = a.n_tiles;  // half the waves idle: does the other half get faster?
    for (; tile < a.n_tiles && a.len >= 4; tile += tstride) {
        unsigned long long tk0 = 0, tk1 = 0, tk2 = 0, tk3 = 0;
        if (DBG) tk0 = __builtin_readcyclecounter();
        uint16_t *const wl = reinterpret_cast<uint16_t *>(wslot + (DMA ? slot * SO_SLOT_BYTES : 0u));
        uint32_t cnt, cur_tp, cur_cnt1, cur_first1;
        unsigned long long cur_bp;
        TilePre cur;
        if constexpr (DMA) {
            // the slot's loads were issued a whole tile ago (nothing orders an LDS read behind them but this wait)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            const uint32_t *const mw = reinterpret_cast<const uint32_t *>(wslot + slot * SO_SLOT_BYTES + 2u * SO_LISTW);
            const uint4 m03 = *reinterpret_cast<const uint4 *>(mw);
            const uint2 m45 = *reinterpret_cast<const uint2 *>(mw + 4);
            // (as scalars, the tile's bookkeeping runs on the scalar unit)
            cnt = (uint32_t)__builtin_amdgcn_readfirstlane((int)m03.x);
            cur_tp = (uint32_t)__builtin_amdgcn_readfirstlane((int)m03.y);
            cur_cnt1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)m03.z);
            cur_first1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)m03.w);
            cur_bp = (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)m45.x) |
                     ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)m45.y) << 32);
            slot ^= 1u;
            prefetch(tile + tstride, slot);
        } else {
            cur = nextP;
            prefetch(tile + tstride, 0);
            // (six lanes hold the tile's six words: as scalars, the tile's bookkeeping runs on the scalar unit)
            cnt = (uint32_t)__builtin_amdgcn_readlane((int)cur.meta, 0);
            cur_tp = (uint32_t)__builtin_amdgcn_readlane((int)cur.meta, 1);
            cur_cnt1 = (uint32_t)__builtin_amdgcn_readlane((int)cur.meta, 2);
            cur_first1 = (uint32_t)__builtin_amdgcn_readlane((int)cur.meta, 3);
            cur_bp = (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)cur.meta, 4) |
                     ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)cur.meta, 5) << 32);
        }
        cnt = cnt < a.list_cap ? cnt : a.list_cap;
        if (cnt == 0) continue;
        const unsigned long long lbase = a.nl_count + 1 + cur_bp + cur_tp;
        if (lbase >= a.line_hi || lbase + cnt <= a.line_lo) continue;
        if (DBG && (a.dbg & 1024u)) { acc.rec += cnt; continue; }  // only the walk over the tiles' words
        if (DBG) { tk1 = __builtin_readcyclecounter(); dbgt[0] += tk1 - tk0; }
        const uint16_t *__restrict__ tl = a.list + tile * a.list_cap;
        const uint64_t tb = tile << WT_SHIFT;
        if constexpr (!DMA) {  // stage the list
            __builtin_amdgcn_wave_barrier();
            *reinterpret_cast<uint2 *>(wl + lane * 4) = cur.l0;
            *reinterpret_cast<uint2 *>(wl + 256 + lane * 4) = cur.l1;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        if (DBG) { tk2 = __builtin_readcyclecounter(); dbgt[1] += tk2 - tk1; }
        // start of the first line after this tile (ends the tile's last line), tile-relative
        uint64_t next_first = a.valid_end;
        if (tile + 1 < a.n_tiles) {
            if (cur_cnt1) {
                next_first = ((tile + 1) << WT_SHIFT) + (cur_first1 & 0x3FFFu);
            } else {
                for (uint64_t t2 = tile + 2; t2 < a.n_tiles; ++t2) {
                    if (a.tile_count[t2]) { next_first = (t2 << WT_SHIFT) + (a.list[t2 * a.list_cap] & 0x3FFFu); break; }
                }
            }
        }
        const uint64_t vend = a.valid_end > tb ? a.valid_end - tb : 0;
        const uint32_t vend_rel = vend < 0x7FFFFFFFull ? (uint32_t)vend : 0x7FFFFFFFu;
        const uint64_t nf = next_first > tb ? next_first - tb : 0;
        const uint32_t nf_rel = nf < 0x7FFFFFFFull ? (uint32_t)nf : 0x7FFFFFFFu;
        // entries of this tile whose lines count: global line index in [line_lo, line_hi)
        const uint32_t e_lo = a.line_lo > lbase ? (uint32_t)(a.line_lo - lbase) : 0u;
        const uint32_t e_hi = a.line_hi - lbase < cnt ? (uint32_t)(a.line_hi - lbase) : cnt;
        const uint32_t lb3 = (uint32_t)lbase & 3u;
        const uint8_t *const tbase = a.buf + tb;
        const uint8_t *const wbase = tbase + col0;  // column col0 of a line that starts at tile offset s: wbase[s]
        // every unconditional load of a line that starts in this tile stays inside the buffer; the
        // (few) tiles at the end of the buffer for which that does not hold take the exact
        // path for everything
        const bool safe = tb + WT_BYTES + col0 + 32u * NSL + 8u <= a.len;
        const uint32_t lce = safe ? lc : 0u;  // LDS rows in use for this tile: none => every column is exact

        // one lane per line: start and raw length of line `sbl` of `kind`; false: no line that counts
        auto line_of = [&](uint32_t kind, uint32_t sbl, uint32_t &s_rel, uint32_t &len) -> bool {
            const uint32_t i0 = ((kind ? 3u : 1u) - lb3) & 3u;
            const uint32_t i = i0 + 4u * sbl;
            if (i >= cnt || i < e_lo || i >= e_hi) return false;
            s_rel = (i < listw ? wl[i] : tl[i]) & 0x3FFFu;
            uint32_t n_rel = i + 1 < cnt ? ((i + 1 < listw ? wl[i + 1] : tl[i + 1]) & 0x3FFFu) : nf_rel;
            n_rel = n_rel < vend_rel ? n_rel : vend_rel;
            len = n_rel - 1 - s_rel;  // raw line, without its '\n'
            return true;
        };
        const uint32_t i0s = (1u - lb3) & 3u, i0q = (3u - lb3) & 3u;
        const uint32_t nls = i0s < cnt ? (cnt - i0s + 3) >> 2 : 0u, nlq = i0q < cnt ? (cnt - i0q + 3) >> 2 : 0u;
        // index, among the records that count, of the record whose sequence line is the tile's first (may be negative)
        const long long lrec0 = ((long long)(lbase + i0s) - (long long)a.line_lo) >> 2;

        if (safe && nls <= 64 && nlq <= 64 && !(DBG && (a.dbg & 8u))) {
            // ---- the usual tile: at most 64 lines of each kind.  Both kinds' lines are worked out at
            // once (their '\r' bytes are in flight together), then all batches run as one sequence so that
            // the first quality batch is fetched while the last sequence batch is counted.
            uint32_t s_s = 0, l_s = 0, s_q = 0, l_q = 0;
            const bool has_s = line_of(0, lane, s_s, l_s), has_q = line_of(1, lane, s_q, l_q);
            // The byte before each line's '\n' is only looked at (two scattered loads per tile, and their latency before the
            // first batch can be packed) once the wave has met a "\r\n"; until then so_count's exact path does the trimming.
            // (that only works where the line's last byte is one of the bytes the pass counts: a tile with a line longer
            // than the rows, and every pass but the first, look right away)
            const bool probe = (pass0 ? (cr_seen || __ballot((has_s && l_s > lce) || (has_q && l_q > lce)) != 0) : cr_file) &&
                               !(DBG && (a.dbg & 16u));
            const uint32_t cr_s = (probe && has_s && l_s) ? tbase[s_s + l_s - 1] : 0u;
            const uint32_t cr_q = (probe && has_q && l_q) ? tbase[s_q + l_q - 1] : 0u;
            if (cr_s == '\r') --l_s;                                                   // trim_winline, src/records.rs:66-73
            if (cr_q == '\r') --l_q;
            if (probe && __ballot(cr_s == '\r' || cr_q == '\r') != 0) cr_any = true;
            const bool trimmed = probe || !pass0;  // (a later pass without probes: pass 0 found no "\r\n" in the file)
            if (pass0) {
                if (has_s) { ++acc.rec; acc.bases += l_s; }
                if (has_q) acc.qual += l_q;
                T.over_s += so_over(l_s, has_s, a.lmax);
                T.over_q += so_over(l_q, has_q, a.lmax);
            }
            // the columns of this pass; the last pass sees a sequence line to its end (the alphabet flags)
            l_s = (a.last && l_s > col0) ? l_s - col0 : seg(l_s);
            l_q = seg(l_q);
            const uint32_t P_s = (has_s && (pass0 || l_s)) ? so_pack(s_s, l_s, lce) : 0u;
            const uint32_t P_q = (has_q && (pass0 || l_q)) ? so_pack(s_q, l_q, lce) : 0u;
            if (DBG) { tk3 = __builtin_readcyclecounter(); dbgt[2] += tk3 - tk2; }
            const uint32_t nbs = (nls + 7) >> 3, nbq = (DBG && (a.dbg & 64u)) ? 0u : (nlq + 7) >> 3;  // 64: sequence lines only
            // Batch f = 2 k + kind: the k-th eight sequence lines, then the k-th eight quality lines -- the lines of (nearly) the
            // same records, a few hundred bytes apart, so the second batch's loads find the first one's cache lines.  A kind
            // that has run out of lines gets an empty batch (P = 0 counts nothing).
            const uint32_t nbm = nbs > nbq ? nbs : nbq;
            const uint32_t nbt = (DBG && (a.dbg & 512u)) ? 0u : 2u * nbm;                               // 512: no batches
            // The line words of a batch are looked up a batch early, before the atomics of the batch being counted:
            // the ds_bpermute then does not queue behind them, and the loads go out as soon as their turn comes.
            auto lookup = [&](uint32_t f) -> uint32_t {
                const bool isq = (f & 1u) != 0;
                const uint32_t b = f >> 1;
                uint32_t P = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(32u * b + 4u * g8), (int)(isq ? P_q : P_s));
                if (DBG && (a.dbg & 32768u))  // every group reads the first group's line: an eighth of the cache lines
                    P = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(32u * b), (int)(isq ? P_q : P_s));
                if (b >= (isq ? nbq : nbs)) P = 0;
                __builtin_amdgcn_sched_barrier(0);  // (keep it where it is: ahead of the count's atomics)
                return P;
            };
            auto fetch = [&](uint32_t P, SoBatch<NSL> &B) {
                B.P = P;
                const uint32_t o = (B.P >> SO_P_SREL) + m4;
                if (DBG && (a.dbg & 384u)) {  // 128: only the first step's load, 256: none
#pragma unroll
                    for (uint32_t u = 0; u < NSL; ++u) B.w[u] = o;
                    if (a.dbg & 128u) B.w[0] = load4_fast(wbase + o);
                    return;
                }
#pragma unroll
                for (uint32_t u = 0; u < NSL; ++u) B.w[u] = load4_fast(wbase + (o + 32 * u));
                if (DBG && (a.dbg & 16384u)) {  // the same loads once more, one byte on: what does a load that hits cost?
#pragma unroll
                    for (uint32_t u = 0; u < NSL; ++u) B.w[u] ^= load4_fast(wbase + (o + 32 * u + 1));
                }
            };
            auto count_s = [&](uint32_t f, SoBatch<NSL> &B) {  // f even
                if (DBG && (a.dbg & 4u)) { acc.rec += B.w[0] == 0x12345u; return; }
                so_count<true, NSL, DBG>(a, wbase, B, S, lane, lce, hist, c, l_s, 32u * (f >> 1) + 4u * g8, T, acc, trimmed, cr_seen, lrec0);
            };
            auto count_q = [&](uint32_t f, SoBatch<NSL> &B) {  // f odd
                if (DBG && (a.dbg & 4u)) { acc.rec += B.w[0] == 0x12345u; return; }
                so_count<false, NSL, DBG>(a, wbase, B, S, lane, lce, hist, c, l_q, 32u * (f >> 1) + 4u * g8, T, acc, trimmed, cr_seen, 0);
            };
            // The fetches are unconditional inside the loops (the index is clamped instead) so that the
            // compiler's s_waitcnt for the batch it needs leaves the next one's loads in flight.
            if (nbt) {
                SoBatch<NSL> B0, B1;  // ping-pong: the loads of one are in flight while the other is counted
                const uint32_t fl = nbt - 1;  // (nbt is even: B0 holds the sequence batches, B1 the quality batches)
                uint32_t pa = lookup(0), pb = lookup(1);  // the words of the next even / odd batch
                fetch(pa, B0);
                for (uint32_t f = 0; f < nbt; f += 2) {
                    fetch(pb, B1);
                    pa = lookup(f + 2 < fl ? f + 2 : fl);  // (past the end: the last batch once more, not counted)
                    count_s(f, B0);
                    fetch(pa, B0);
                    pb = lookup(f + 3 < fl ? f + 3 : fl);
                    count_q(f + 1, B1);
                }
            }
            if (DBG) dbgt[3] += __builtin_readcyclecounter() - tk3;
            continue;
        }

        // ---- any other tile (more than 64 lines of a kind, or too close to the end of the buffer for
        // unconditional loads): one kind after the other, 64 lines at a time
        for (uint32_t kind = 0; kind < 2; ++kind) {        // 0: sequence lines, 1: quality lines
            const uint32_t nlines = kind ? nlq : nls;
            if (!nlines) continue;
            for (uint32_t sb = 0; sb < nlines; sb += 64) {
                uint32_t my_P = 0, my_len = 0, full_len = 0;
                bool mine = false, crl = false;
                {
                    uint32_t s_rel = 0, len = 0;
                    if (sb + lane < nlines && line_of(kind, sb + lane, s_rel, len)) {
                        if (len && tbase[s_rel + len - 1] == '\r') {                   // trim_winline, src/records.rs:66-73
                            --len;
                            crl = true;
                        }
                        mine = true;
                        full_len = len;
                        if (pass0) {
                            if (kind == 0) {
                                ++acc.rec;
                                acc.bases += len;
                            } else {
                                acc.qual += len;
                            }
                        }
                        len = (kind == 0 && a.last && len > col0) ? len - col0 : seg(len);
                        my_len = len;
                        my_P = (pass0 || len) ? so_pack(s_rel, len, lce) : 0u;
                    }
                }
                if (__ballot(crl) != 0) cr_any = true;
                if (pass0) {
                    if (kind == 0) T.over_s += so_over(full_len, mine, a.lmax);
                    else T.over_q += so_over(full_len, mine, a.lmax);
                }
                const uint32_t nbat = ((nlines - sb < 64 ? nlines - sb : 64u) + 7) >> 3;
                SoBatch<NSL> B0;
                for (uint32_t b = 0; b < nbat; ++b) {
                    B0.P = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(32u * b + 4u * g8), (int)my_P);
                    const uint32_t o = (B0.P >> SO_P_SREL) + m4;
                    if (safe) {
#pragma unroll
                        for (uint32_t u = 0; u < NSL; ++u) B0.w[u] = load4_fast(wbase + (o + 32 * u));
                    } else {  // lce == 0: every column goes through the exact path, which loads for itself
#pragma unroll
                        for (uint32_t u = 0; u < NSL; ++u) B0.w[u] = 0;
                    }
                    if (kind == 0) so_count<true, NSL, DBG>(a, wbase, B0, S, lane, lce, hist, c, my_len, 32u * b + 4u * g8, T, acc, true, cr_seen, lrec0 + sb);
                    else so_count<false, NSL, DBG>(a, wbase, B0, S, lane, lce, hist, c, my_len, 32u * b + 4u * g8, T, acc, true, cr_seen, 0);
                }
            }
        }
    }
    if (pass0 && a.cr_flag && (cr_any || cr_seen) && lane == 0) atomicOr(a.cr_flag, 1u);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the inline ds_add of lds_add
    __syncthreads();
    uint32_t *__restrict__ dst = a.scratch + (uint64_t)blockIdx.x * SO_WORDS;
    for (uint32_t i = threadIdx.x; i < SO_WORDS; i += SO_THREADS) dst[i] = hist[i];
    if (DBG && (a.dbg & 8192u) && lane == 0)
        for (int j = 0; j < 4; ++j) atomicAdd(&a.qual_hist[j], dbgt[j]);
    // per-line totals: rec / bases / qual were summed by the lane that owned the line; the two
    // "not DNA" counts are wave-uniform
    unsigned long long sc[7] = {acc.rec, acc.bases, acc.qual, 0, 0, T.over_s, T.over_q};  // [5], [6]: bytes at positions >= lmax
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        unsigned long long v = sc[j];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
        sc[j] = v;
    }
    if (lane == 0) {
        // [3], [4]: records whose sequence is pure ACGT / ACGTN = all of them (pass 0) less the flagged ones (any pass)
        sc[3] = sc[0] - T.not_dna;
        sc[4] = sc[0] - T.not_dnan;
#pragma unroll
        for (int j = 0; j < 7; ++j)
            if (sc[j]) atomicAdd(&a.scalars[j], sc[j]);
    }
}

// Sum the per-block partial histograms of one pass into the caller's u64 arrays (coalesced reads): LDS row r is
// column col0 + r.
__global__ __launch_bounds__(256) void k_stats_reduce_oct(const uint32_t *__restrict__ scratch, uint32_t n_blocks,
                                                          uint32_t lc, uint32_t col0,
                                                          unsigned long long *__restrict__ qual_hist,
                                                          unsigned long long *__restrict__ base_hist) {
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= SO_WORDS) return;
    const bool isq = id >= SO_SBYTES / 4;
    const uint32_t r = isq ? id - SO_SBYTES / 4 : id;
    const uint32_t rb = isq ? r >> 12 : r >> 9;
    const uint32_t bin = isq ? (r >> 6) & 63u : (r >> 6) & 7u;
    const uint32_t row = rb * 64 + so_row6(r & 63u);
    if (row >= lc) return;
    const uint32_t b0 = blockIdx.y * RED_GROUP;
    const uint32_t b1 = b0 + RED_GROUP < n_blocks ? b0 + RED_GROUP : n_blocks;
    unsigned long long s = 0;
    for (uint32_t b = b0; b < b1; ++b) s += scratch[(uint64_t)b * SO_WORDS + id];
    if (!s) return;
    if (isq) atomicAdd(&qual_hist[(uint64_t)(col0 + row) * 256 + 33 + bin], s);
    else atomicAdd(&base_hist[(uint64_t)(col0 + row) * 8 + bin_to_class(bin)], s);  // bins 0,2,5 share class 5
}

size_t stats_oct_scratch_bytes(uint32_t, int n_cu) { return (size_t)stats_blocks(n_cu) * SO_WORDS * sizeof(uint32_t); }
template <uint32_t NSL, bool DBG>
static hipError_t launch_stats_oct_n(hipStream_t s, const StatsArgs &a, uint32_t blocks, size_t lds) {
    // Lanes without a whole dword subtract 0 at the address their bytes happen to form: any bin byte
    // (< 64 KiB) plus the largest row-block offset.  The allocation covers all of them.
    if (lds < SO_ADDR_SPAN) lds = SO_ADDR_SPAN;
    if (lds > SO_LDS_MAX) return hipErrorInvalidValue;
    static LdsAttr attr;
    if (hipError_t e = attr.ensure(reinterpret_cast<const void *>(k_stats_oct<NSL, DBG>), lds); e != hipSuccess) return e;
    hipLaunchKernelGGL((k_stats_oct<NSL, DBG>), dim3(blocks), dim3(SO_THREADS), lds, s, a);
    return hipSuccess;
}
hipError_t launch_stats_oct(hipStream_t s, StatsArgs a, int n_cu) {
#ifdef FQH_TUNING  // timing variant of the kernel (tools/exp_statsdbg.py); not part of the product library
    static const uint32_t dbg = getenv("FQH_STATS_DBG") ? (uint32_t)atoi(getenv("FQH_STATS_DBG")) : 0u;
#else
    const uint32_t dbg = 0;
#endif
    a.dbg = dbg;
    const size_t lds = (size_t)SO_ADDR_SPAN + (size_t)SO_WAVES * 2 * SO_SLOT_BYTES;
    const uint32_t blocks = stats_blocks(n_cu);
    // one pass per 256 columns, as far as the caller's rows and the longest line go (reads of up to 256 bp: one)
    const uint32_t span = (a.max_line && a.max_line < a.lmax) ? a.max_line : a.lmax;
    if (span <= SO_LC_MAX) {  // one pass
        a.flagmap = nullptr;
        a.cr_flag = nullptr;
    }
    for (uint32_t col0 = 0; col0 == 0 || col0 < span; col0 += SO_LC_MAX) {
        a.col0 = col0;
        a.lc = a.lmax - col0 < SO_LC_MAX ? a.lmax - col0 : SO_LC_MAX;
        a.last = col0 + SO_LC_MAX >= span ? 1u : 0u;
        const uint32_t nsl = (a.lc + 31) / 32;  // steps that hold LDS rows
        hipError_t e =
#ifdef FQH_TUNING
                       dbg        ? (nsl <= 5 ? launch_stats_oct_n<5, true>(s, a, blocks, lds) : launch_stats_oct_n<8, true>(s, a, blocks, lds)) :  // timing experiments
#endif
                       nsl <= 2   ? launch_stats_oct_n<2, false>(s, a, blocks, lds)
                       : nsl <= 4 ? launch_stats_oct_n<4, false>(s, a, blocks, lds)
                       : nsl <= 5 ? launch_stats_oct_n<5, false>(s, a, blocks, lds)
                                  : launch_stats_oct_n<8, false>(s, a, blocks, lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_stats_reduce_oct, dim3((SO_WORDS + 255) / 256, (blocks + RED_GROUP - 1) / RED_GROUP), dim3(256), 0, s,
                           a.scratch, blocks, a.lc, a.col0, a.qual_hist, a.base_hist);
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// k_stats_long — reads longer than the 256 bank-scheduled LDS rows (kilobase reads), counted in ONE walk over the record
// index instead of one pass of k_stats_oct per 256 columns (each of which walks every tile for the few lines it holds: 5 kbp
// reads took 20 passes, 557 GB/s).  The reference treats every record up to BUFSIZE alike (src/lib.rs:276-283,
// src/records.rs:75-90); so does this: the work item is (record, block of 256 columns), found by plain arithmetic on the
// IdxRecord-style index (fqh_idx_record: start + the four newline offsets) — no tile lists, no line-start search.
//   * a block of 1024 threads owns one work item: a column block cb and a range of the records that reach it (the plan is
//     made on the device from the reads' lengths: k_long_plan below); its LDS holds the
//     bank-scheduled histogram of those 256 columns (stats_dev.h: 8 KiB sequence + 128 KiB quality: 128 bins per column,
//     '!' .. 0xA0 — HiFi reads are mostly '~' (Q93), and a 64-bin window sent every such byte to the caller's arrays);
//   * eight lanes walk a line's 256 columns, eight records per wavefront and round.  A lane loads EIGHT contiguous bytes four
//     times (columns 64 v + 8 m .. + 7: half the load instructions of one dword per lane and step, 2.21 -> 2.13 ms per 4 GiB
//     of 5 kbp reads) and counts them under a bank-scheduled layout of its own (so_slot_long): byte j of register
//     2 v + h is row 64 v + 8 m + 4 h + j, slot m + 8 j + 32 h — the half h (128 bytes) goes into the instruction's
//     immediate, m + 8 j is the lane's slot byte: the 32 lanes of four line slots are on 32 distinct banks.
//     Whole dwords of bytes inside the alphabet / window cost one v_perm_b32 and one ds_sub_u32 per byte; a dword with a byte outside, or with fewer than four bytes of the line, takes the per-byte
//     statement (so_exact_step: window bytes to LDS, the rest to the caller's arrays);
//   * per-record facts: lengths and the columns beyond lmax are arithmetic, done by column block 0; "has an N / a byte
//     outside ACGTN" is ORed over a line's column blocks through one bit per record and flag in scratch (atomicOr: the
//     block that sets a bit first counts the record), sequence columns beyond lmax included;
//   * every input byte of a sequence / quality line is read once; algorithmic bytes: the buffer's length.
// The plan of one launch, made ON THE DEVICE from the lengths of the records at hand (k_long_census / k_long_plan below): the
// work items (column block, range of records), slice-major.
struct LongItem {
    uint32_t cb, pad;
    uint64_t r_lo, r_hi;           // positions in the column block's LIST of records (LongPlan::listed), or records
};
struct LongPlan {
    uint32_t n_items;              // work items in use (<= the launch's blocks)
    uint32_t listed;               // 1: records of several length classes: every column block walks the list of the records that reach it
    uint32_t q_max;                // slices of column block 0 (the most any column block has)
    uint32_t w;                    // records per item
};
struct LongArgs {
    const uint8_t *buf;            // chunk-relative: the byte at file offset o is buf[o - base_offset]
    uint64_t len, base_offset;
    const fqh_idx_record *idx;     // the records that count: idx[0 .. n)
    const uint32_t *lists;         // per column block, the records that reach it (LongPlan::listed): list cb starts at lists[list_off[cb]]
    const uint64_t *list_off;
    uint64_t n;
    uint32_t lmax;
    const LongPlan *plan;
    const LongItem *items;
    uint32_t *part;                // [items][SO_LWORDS]: every block's rows as it leaves them (k_stats_long_reduce adds them up)
    uint32_t *flagmap;             // 2 x flag_words words, zeroed: [has N or worse | has a byte outside ACGTN]
    uint64_t flag_words;
    unsigned long long *qual_hist, *base_hist, *scalars;
};
__global__ __launch_bounds__(SO_THREADS) void k_stats_long(LongArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t hist[];
    // Workgroups go to the eight XCDs round-robin (blockIdx % 8), and every XCD has an L2 of its own.  The column blocks of ONE
    // slice of the records read neighbouring 256-byte pieces of the same lines — pieces that are not aligned to the 128-byte
    // cache lines, so that neighbours share the line at either edge — at about the same time: the work items (slice-major:
    // slice s of every column block that has one, then slice s + 1; every item holds LongPlan::w records, so slice s of two
    // column blocks is the same records) are dealt out in eight CONTIGUOUS runs, one per XCD, so that a slice's column blocks
    // sit on one XCD (two at a run's edge) and the shared lines are fetched once.  (Numbered across the XCDs, every column
    // block of a slice pulled its edge lines through another L2: 1.7 x the input's bytes from memory, PMC FETCH_SIZE.)
    const uint32_t n_items = a.plan->n_items, per_xcd = (n_items + 7u) / 8u;
    const uint32_t xcd = blockIdx.x & 7u, item = xcd * per_xcd + (blockIdx.x >> 3);
    if ((blockIdx.x >> 3) >= per_xcd || item >= n_items) return;   // (the grid is the most items a plan may use)
    const LongItem it = a.items[item];
    const uint32_t *const lst = a.plan->listed ? a.lists + a.list_off[it.cb] : nullptr;   // (block-uniform)
    for (uint32_t i = threadIdx.x; i < SO_LWORDS; i += SO_THREADS) hist[i] = 0;
    __syncthreads();
    if ((uint32_t)(uintptr_t)hist != 0) __builtin_trap();  // the address registers assume the histogram starts at LDS address 0
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint32_t m = lane & 7u, m8 = m * 8u, g8 = lane >> 3;
    SoLane c;
    c.slots = 0;
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) {
        const uint32_t j = k ^ (g8 & 3u);
        c.sel[k] = 0x0C0C0004u + k + (j << 8);
        c.slots |= ((m + 8u * j) * 4u) << (8u * k);
    }
    const uint32_t cb = it.cb;
    const uint32_t col0 = cb * SO_LC_MAX;
    const uint32_t lc = a.lmax > col0 ? (a.lmax - col0 < SO_LC_MAX ? a.lmax - col0 : SO_LC_MAX) : 0u;  // rows of this block that the caller has
    StatsArgs sa = {};           // what so_exact_step wants to know
    sa.lc = lc;
    sa.col0 = col0;
    sa.qual_hist = a.qual_hist;
    sa.base_hist = a.base_hist;
    const uint64_t r_lo = it.r_lo, r_hi = it.r_hi;
    unsigned long long recs = 0, bases = 0, quals = 0, over_s = 0, over_q = 0;   // per lane (lane m == 0 of a record's group adds)
    uint32_t newn = 0, newi = 0;
    const uint8_t *const bend = a.buf + a.len;
    // A wavefront's round is eight records' pieces of this column block: sixteen loads per lane, then 128 LDS atomics per lane.
    // Counted one round at a time, a wavefront alternates between waiting for its loads and counting (the loads alone take
    // 1.29 ms per 4 GiB of 5 kbp reads, loads + count 2.35: the sum, not the maximum).  So the rounds are software-pipelined:
    //   * the words of round k + 1 are loaded (buffer B) before round k (buffer A) is counted, and the other way round;
    //   * the three dependent loads of a round — the record's index entry, the byte in front of each line's '\n'
    //     (trim_winline), the lines' words — are spread over three rounds: the entry of round k + 2 is fetched at the top of
    //     round k, its two probe bytes between the counts of round k's two lines (the entry has arrived by then), its words at
    //     the top of round k + 1.  Behind a plan with lists (LongPlan::listed) there is a fourth in front of them: the record's
    //     number, read from the column block's list three rounds ahead.
    constexpr uint64_t STEP = (uint64_t)SO_WAVES * 8;
    const uint64_t p_first = r_lo + (uint64_t)wv * 8 + g8;   // this lane group's position (in the list, or in the index) in the first round
    struct Round {               // a round whose words are in flight / loaded
        uint32_t id;             // the record's number if the plan has lists (else it is the position)
        bool has;
        uint32_t segs[2], ws[2][8];
    };
    auto id_of = [&](uint64_t p) { return (lst && p < r_hi) ? lst[p] : 0u; };
    auto entry_of = [&](uint64_t p, uint32_t id) {
        fqh_idx_record ir = {};
        if (p < r_hi) ir = a.idx[lst ? (uint64_t)id : p];
        return ir;
    };
    auto probe = [&](const fqh_idx_record &ir, uint64_t r, uint32_t cr[2]) {   // the byte in front of each line's '\n'
#pragma unroll
        for (int kind = 0; kind < 2; ++kind) {
            cr[kind] = 0;
            if (r < r_hi) {
                const uint32_t len = (kind ? ir.qual - ir.sep : ir.seq - ir.head) - 1u;
                if (len) cr[kind] = (a.buf + (ir.start - a.base_offset) + (kind ? ir.sep : ir.head) + 1)[len - 1];
            }
        }
    };
    // geometry of record r's two lines in this column block, the record's totals (column block 0), and the loads of its words
    auto issue = [&](Round &R, uint64_t r, uint32_t id, const fqh_idx_record &ir, const uint32_t cr[2]) {
        R.id = id;
        R.has = r < r_hi;
#pragma unroll
        for (int kind = 0; kind < 2; ++kind) {
            const uint8_t *line = a.buf + (ir.start - a.base_offset) + (kind ? ir.sep : ir.head) + 1;
            uint32_t len = R.has ? (kind ? ir.qual - ir.sep : ir.seq - ir.head) - 1u : 0u;   // raw line, without its '\n'
            if (len && cr[kind] == '\r') --len;                                               // trim_winline, src/records.rs:66-73
            if (cb == 0 && m == 0 && R.has) {
                if (kind) { quals += len; over_q += len > a.lmax ? len - a.lmax : 0u; }
                else { ++recs; bases += len; over_s += len > a.lmax ? len - a.lmax : 0u; }
            }
            // this block's columns of the line: [col0, col0 + 256); sequence lines are LOOKED AT to their end (the alphabet
            // flags cover every base), counted up to the caller's rows
            uint32_t seg = len > col0 ? (len - col0 < SO_LC_MAX ? len - col0 : SO_LC_MAX) : 0u;  // columns of the line in this block
            if (kind && seg > lc) seg = lc;   // (quality bytes beyond the caller's rows are nobody's business: not loaded, not looked at)
            R.segs[kind] = seg;
            // eight contiguous bytes per lane and load (registers 2 v, 2 v + 1: columns 64 v + 8 m .. + 7)
            const uint8_t *p = line + col0 + m8;
            if (__ballot(seg != 0 && p + 256 > bend) == 0) {   // (wave-uniform) every load of the round lies inside the buffer
#pragma unroll
                for (uint32_t v = 0; v < 4; ++v) {
                    uint2 w2 = make_uint2(0u, 0u);
                    if (64u * v + m8 < seg) __builtin_memcpy(&w2, p + 64u * v, 8);
                    R.ws[kind][2 * v] = w2.x;
                    R.ws[kind][2 * v + 1] = w2.y;
                }
            } else {
#pragma unroll
                for (uint32_t u = 0; u < 8; ++u) R.ws[kind][u] = 64u * (u >> 1) + 4u * (u & 1u) + m8 < seg ? load4_any(p + 64u * (u >> 1) + 4u * (u & 1u), bend) : 0u;
            }
        }
    };
    const int lim_rows = (int)lc - (int)m8;   // (the caller's rows, seen from this lane's first column)
    auto count_line = [&](const Round &R, const int kind, uint32_t &any_n, uint32_t &any_inv) {
        const uint32_t seg = R.segs[kind];
        if (__ballot(seg != 0) == 0) return;
        constexpr uint32_t RB = SO_LRB;
        if (lc == 0) {
            // (block-uniform) a column block beyond the caller's rows — lmax is the caller's choice, the reads' length is not: the
            // sequence bytes are looked at (the alphabet flags cover every base of a record), nothing is counted.  Until round 5
            // every such dword took the exact statement below: 5 kbp reads with lmax = 150 ran at a NINTH of the speed of lmax = 5000.
            const int lim = (int)seg - (int)m8;
#pragma unroll
            for (uint32_t u = 0; u < 8; ++u) {
                const uint32_t wu = R.ws[kind][u];
                const int t = lim - (int)(64u * (u >> 1) + 4u * (u & 1u));
                if (t > 0) {
                    const uint32_t sh = t < 4 ? 32u - 8u * (uint32_t)t : 0u;
                    any_n |= wu << sh;
                    any_inv |= ((wu ^ __builtin_amdgcn_perm(0x474EFF54u, 0x43FF41FFu, wu & 0x07070707u)) << sh) ? 1u : 0u;
                }
            }
            return;
        }
        // a dword at column C + 8 m (C a constant of the step) is whole iff C + 4 <= min(seg, lc) - 8 m, and holds bytes of the
        // line at all iff C < seg - 8 m: one compare with a constant each (the limits are per line, not per dword)
        const int lim_any = (int)seg - (int)m8, lim_whole = lim_any < lim_rows ? lim_any : lim_rows;
#pragma unroll
        for (uint32_t u = 0; u < 8; ++u) {
            const uint32_t wu = R.ws[kind][u];
            const int C = (int)(64u * (u >> 1) + 4u * (u & 1u));   // column of the dword's first byte for lane m == 0, relative to col0
            const bool whole = lim_whole >= C + 4;                  // four bytes of the line, all inside the caller's rows
            uint32_t pb, chk;
            if (kind == 0) {
                pb = wu & 0x07070707u;
                chk = wu ^ __builtin_amdgcn_perm(0x474EFF54u, 0x43FF41FFu, pb);   // != 0: a byte outside ACGTN
            } else {
                pb = wu - 0x21212121u;
                chk = pb & 0x80808080u;                                             // != 0: a byte outside '!' .. 0xA0 (128 bins)
                pb &= 0x7F7F7F7Fu;                                                  // (whatever the bytes are, the address stays inside the rows)
            }
            const uint32_t f = (whole && !chk) ? 0xFFFFFFFFu : 0u;
            if (kind == 0) any_n |= wu & f;   // (bit 3 of a byte is set in 'N' only: masked once, in flags())
            const uint32_t off = (kind ? SO_SBYTES : 0u) + 128u * (u & 1u) + (kind ? RB : 2048u) * (u >> 1);
#pragma unroll
            for (int k = 0; k < 4; ++k)
                (void)__hip_atomic_fetch_sub((so_lds_u32 *)(uintptr_t)(__builtin_amdgcn_perm(c.slots, pb, c.sel[k]) + off), f,
                                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            bool exact = lim_any > C && !f;           // bytes of the line that the four atomics above did not count
            if (__ballot(exact) != 0) {
                // The line's last, partial dword (a line has one unless its length is a multiple of four — and in reads of many
                // lengths every record's sits in another step): its one to three bytes, inside the caller's rows and inside the
                // alphabet / window, are counted as the whole dwords are, under a mask per byte.  (Until round 5 they took the
                // exact statement below, a loop per byte: unseen in the benchmarks, whose reads were 600 .. 20 000 bases long,
                // all multiples of four — reads of 250 .. 1000 bases ran at HALF the speed per piece.)
                const int t = lim_any - C;            // bytes of the line in this dword (if in 1 .. 3)
                if (kind == 0 && exact && lim_rows <= C) {   // a dword of the sequence line beyond the caller's rows: looked at, not counted
                    const uint32_t sh = t < 4 ? 32u - 8u * (uint32_t)t : 0u;
                    any_n |= wu << sh;
                    any_inv |= (chk << sh) ? 1u : 0u;
                    exact = false;
                }
                if (exact && t < 4 && lim_whole >= lim_any) {
                    const uint32_t sh = 32u - 8u * (uint32_t)t;
                    if ((chk << sh) == 0) {
                        if (kind == 0) any_n |= wu << sh;
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            (void)__hip_atomic_fetch_sub((so_lds_u32 *)(uintptr_t)(__builtin_amdgcn_perm(c.slots, pb, c.sel[k]) + off),
                                                         (int)((uint32_t)k ^ (g8 & 3u)) < t ? 0xFFFFFFFFu : 0u, __ATOMIC_RELAXED,
                                                         __HIP_MEMORY_SCOPE_WORKGROUP);
                        exact = false;
                    }
                }
                if (__ballot(exact) != 0) {           // (rare) a byte outside, or columns beyond the caller's rows
                    if (exact) {
                        const uint32_t pos = (uint32_t)C + m8;
                        uint32_t an = 0, ai = 0;
                        if (kind == 0) so_exact_step<true, SO_LQBITS, true>(sa, wu, pos, seg, lc, hist, an, ai);
                        else so_exact_step<false, SO_LQBITS, true>(sa, wu, pos, seg, lc, hist, an, ai);
                        any_n |= an ? 0x08u : 0u;   // (flags() looks at bit 3 of each byte)
                        any_inv |= ai;
                    }
                }
            }
        }
    };
    // a record's alphabet flags: ORed over the 8 lanes of its group here, over its column blocks through the flag maps
    auto flags = [&](const Round &R, uint64_t p, uint32_t any_n, uint32_t any_inv) {
        const unsigned long long bi = __ballot(any_inv != 0), bn = __ballot((any_n & 0x08080808u) != 0) | bi;
        if (bn) {
            const uint32_t sh = lane & 56u;
            const bool gn = ((bn >> sh) & 0xFFull) != 0, gi = ((bi >> sh) & 0xFFull) != 0;
            if (m == 0 && gn && R.has) {
                const uint64_t rid = lst ? (uint64_t)R.id : p;   // the record's number among those that count
                const uint32_t bit = 1u << (rid & 31u);
                if (!(atomicOr(&a.flagmap[rid >> 5], bit) & bit)) ++newn;
                if (gi && !(atomicOr(&a.flagmap[a.flag_words + (rid >> 5)], bit) & bit)) ++newi;
            }
        }
    };
    // one round: the NEXT round's words into `N`, then the words of `R` (loaded a round ago) are counted; q: the entry two rounds
    // ahead, fetched here; e1 / cr1: the next round's entry and probe bytes (ready), replaced by the ones after them
    Round A, B;
    uint32_t id1 = id_of(p_first + STEP), id2 = id_of(p_first + 2 * STEP), id3;
    fqh_idx_record e1 = entry_of(p_first + STEP, id1), e2;
    uint32_t cr1[2], cr2[2];
    {
        const uint32_t id0 = id_of(p_first);
        const fqh_idx_record e0 = entry_of(p_first, id0);
        uint32_t cr0[2];
        probe(e0, p_first, cr0);
        probe(e1, p_first + STEP, cr1);
        issue(A, p_first, id0, e0, cr0);
    }
    auto round = [&](Round &R, Round &N, uint64_t p) {   // p: the position R's record was taken from
        issue(N, p + STEP, id1, e1, cr1);
        e2 = entry_of(p + 2 * STEP, id2);
        id3 = id_of(p + 3 * STEP);
        uint32_t any_n = 0, any_inv = 0;
        count_line(R, 0, any_n, any_inv);
        probe(e2, p + 2 * STEP, cr2);
        count_line(R, 1, any_n, any_inv);
        flags(R, p, any_n, any_inv);
        e1 = e2;
        cr1[0] = cr2[0];
        cr1[1] = cr2[1];
        id1 = id2;
        id2 = id3;
    };
    for (uint64_t r0 = r_lo + (uint64_t)wv * 8; r0 < r_hi; r0 += 2 * STEP) {   // (wave-uniform bounds; two rounds per trip: the buffers swap roles)
        round(A, B, r0 + g8);
        if (r0 + STEP < r_hi) round(B, A, r0 + STEP + g8);
        else break;
    }
    // ---- the block's rows -> scratch, as they are (the S slices of a column block adding ~26 000 counters each to the caller's
    // arrays with 64-bit atomics took 0.15 of the kernel's 1.37 ms per 4 GiB; plain stores + k_stats_long_reduce: 0.02), totals
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    {
        uint4 *dst = reinterpret_cast<uint4 *>(a.part + (uint64_t)item * SO_LWORDS);
        const uint4 *src = reinterpret_cast<const uint4 *>(hist);
        for (uint32_t i = threadIdx.x; i < SO_LWORDS / 4; i += SO_THREADS) dst[i] = src[i];
    }
    unsigned long long t[7] = {recs, bases, quals, over_s, over_q, newn, newi};
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        unsigned long long v = t[j];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
        t[j] = v;
    }
    if (lane == 0) {
        if (t[0]) atomicAdd(&a.scalars[0], t[0]);
        if (t[1]) atomicAdd(&a.scalars[1], t[1]);
        if (t[2]) atomicAdd(&a.scalars[2], t[2]);
        if (t[0] || t[5]) atomicAdd(&a.scalars[3], t[0] - t[5]);   // valid_dna = records - (records with an N or worse), summed over the blocks
        if (t[0] || t[6]) atomicAdd(&a.scalars[4], t[0] - t[6]);
        if (t[3]) atomicAdd(&a.scalars[5], t[3]);
        if (t[4]) atomicAdd(&a.scalars[6], t[4]);
    }
}
// the rows of every slice of a column block, summed and added to the caller's arrays (one thread per counter: no atomics).
// Slice s of column block cb is work item slice_start[s] + cb.
__global__ __launch_bounds__(256) void k_stats_long_reduce(const uint32_t *__restrict__ part, const uint32_t *__restrict__ qs,
                                                           const uint32_t *__restrict__ slice_start, uint32_t lmax,
                                                           unsigned long long *__restrict__ qual_hist, unsigned long long *__restrict__ base_hist) {
    const uint32_t id = blockIdx.x * 256 + threadIdx.x, cb = blockIdx.y;
    if (id >= SO_LWORDS) return;
    unsigned long long v = 0;
    const uint32_t n_slices = qs[cb];
    for (uint32_t sl = 0; sl < n_slices; ++sl) v += part[(uint64_t)(slice_start[sl] + cb) * SO_LWORDS + id];
    if (!v) return;
    const uint32_t col0 = cb * SO_LC_MAX;
    const uint32_t lc = lmax > col0 ? (lmax - col0 < SO_LC_MAX ? lmax - col0 : SO_LC_MAX) : 0u;
    const bool isq = id >= SO_SBYTES / 4;
    const uint32_t q = isq ? id - SO_SBYTES / 4 : id;
    const uint32_t rb = isq ? q >> (6 + SO_LQBITS) : q >> 9;
    const uint32_t bin = isq ? (q >> 6) & ((1u << SO_LQBITS) - 1u) : (q >> 6) & 7u;
    const uint32_t row = rb * 64 + so_row6_long(q & 63u);
    if (row >= lc) return;
    // (sequence bins 0 and 5 both mean "other": two counters of this launch may belong to one of the caller's)
    if (isq) qual_hist[(uint64_t)(col0 + row) * 256 + 33 + bin] += v;
    else atomicAdd(&base_hist[(uint64_t)(col0 + row) * 8 + bin_to_class(bin)], v);
}

// ---- the plan of a launch, made on the device ---------------------------------------------------------------------
// Reads of ONE length fill every column block alike, and "the records in equal slices per column block" is a balanced plan.
// Real long reads are not of one length (a nanopore or HiFi run spreads over a decade), and then the column blocks towards
// the ends of the longest reads hold few bytes: with equal slices their CUs idle for most of the one round the launch runs in
// (round 4: 1.89 against 1.22 ms per 4 GiB for lengths uniform in 4 .. 16 kbp; a log-normal spread 3.6 ms), and slices in
// proportion to the bytes do not help as long as a block WALKS every record of its slice to find the few that reach its
// columns (round 4: a block's time follows the records it walks, 14 ms).  So, before the walk, on the stream and without the
// host:
//   * k_long_census counts the records by CLASS = the number of column blocks they reach (ceil(raw line length / 256), at
//     least 1: column block 0 sees every record — it counts them);
//   * k_long_plan turns that into cnt[cb] = records that reach column block cb (a suffix sum) and cuts every column block's
//     records into items of w, w the smallest for which the items fit the launch's blocks (a binary search over
//     sum ceil(cnt[cb] / w)); the items are numbered slice-major: slice s = the s-th item of every column block that has one;
//   * k_long_lists writes, per column block, the LIST of the records that reach it, in the file's order (up to the order in
//     which blocks of 4096 records reserve their share) — what an item is a range of.  Skipped when every record is of one
//     class: every list would be 0, 1, 2, ...
// Every item then holds the same number of 256-byte pieces, whatever the lengths, and reads them in the file's order.  (First
// built as ONE order for all column blocks — the index sorted by class, longest first, column block cb's records a prefix
// of it: balanced as well, and 2.2 x faster than equal slices on a log-normal spread around 5 kbp, but SLOWER than equal
// slices on reads of 250 .. 1000 bp, 2.59 against 2.51 ms per 4 GiB: in class-major order neighbouring pieces are far apart
// in the file; the same data with the FILE sorted by length ran in 1.58 ms, which is what told the two effects apart.)
constexpr uint32_t LONG_PLAN_MAX = 4096;       // column blocks and items the planner's LDS arrays hold; beyond: equal slices, no census
constexpr uint32_t LONG_W_MIN = 8 * SO_WAVES;  // records a block's wavefronts take per round: no item is cut smaller
constexpr uint32_t LONG_SORT_PER_BLOCK = 4096; // records per block of the census (four per thread)
__device__ __forceinline__ uint32_t long_class(const fqh_idx_record *idx, uint64_t r, uint32_t n_cb) {
    uint2 hs;                                  // head, seq: the line's raw length is the distance of the two newlines
    __builtin_memcpy(&hs, reinterpret_cast<const uint8_t *>(idx + r) + 8, 8);
    const uint32_t raw = hs.y - hs.x - 1u, k = raw ? (raw - 1u) / SO_LC_MAX + 1u : 1u;
    return k < n_cb ? k : n_cb;
}
// the block's records per class -> lh[1 .. n_cb] (zeroed by the caller); a wavefront whose records are of one class (reads of
// one length) adds once
__device__ __forceinline__ void long_count(uint32_t *lh, uint32_t k, bool has) {
    const unsigned long long hm = __ballot(has);
    if (!hm) return;
    const uint32_t k0 = (uint32_t)__builtin_amdgcn_readlane((int)k, (int)(__ffsll((long long)hm) - 1));
    if (__ballot(has && k != k0) == 0) {
        if (__lane_id() == (uint32_t)(__ffsll((long long)hm) - 1)) atomicAdd(&lh[k0], (uint32_t)__popcll(hm));
    } else if (has) {
        atomicAdd(&lh[k], 1u);
    }
}
__global__ __launch_bounds__(1024) void k_long_census(const fqh_idx_record *__restrict__ idx, uint64_t n, uint32_t n_cb, uint32_t *__restrict__ hist) {
    extern __shared__ uint32_t lh[];           // [n_cb + 1]
    for (uint32_t i = threadIdx.x; i <= n_cb; i += 1024) lh[i] = 0;
    __syncthreads();
    const uint64_t r0 = (uint64_t)blockIdx.x * LONG_SORT_PER_BLOCK + threadIdx.x;
#pragma unroll
    for (uint32_t t = 0; t < LONG_SORT_PER_BLOCK / 1024; ++t) {
        const uint64_t r = r0 + t * 1024u;
        const bool has = r < n;
        long_count(lh, has ? long_class(idx, r, n_cb) : 0u, has);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i <= n_cb; i += 1024)
        if (lh[i]) atomicAdd(&hist[i], lh[i]);
}
struct LongPlanArgs {
    uint64_t n;
    uint32_t n_cb, nb;             // column blocks; items the launch has blocks (and rows in `part`) for, >= n_cb
    uint32_t nb_one_length;        // ... of which reads of ONE class use this many (the host's plan for them: one round of blocks)
    uint32_t census;               // 0: no census was taken (more column blocks / items than the planner holds): every record reaches every column block
    const uint32_t *hist;          // [n_cb + 1] records per class (census)
    uint64_t *list_off;            // [n_cb + 1] -> where column block cb's list starts
    uint32_t *q;                   // [n_cb] -> slices per column block
    uint32_t *slice_start;         // [nb + 1] -> first item of slice s
    LongItem *items;               // [nb]
    LongPlan *plan;
};
__global__ __launch_bounds__(1024) void k_long_plan(LongPlanArgs a) {
    extern __shared__ uint32_t lp[];           // cnt[n_cb + 1], then start[nb + 1] (census); nothing otherwise
    __shared__ uint32_t sh_w, sh_qmax, sh_listed;
    const uint32_t tid = threadIdx.x, n_cb = a.n_cb;
    uint32_t *const cnt = lp, *const start = lp + n_cb + 1;
    if (!a.census) {
        // equal slices, closed form (what the host's plan assumed): nb / n_cb slices of every column block
        const uint32_t S = a.nb_one_length / n_cb;
        const uint64_t w = (a.n + S - 1) / S;
        for (uint32_t i = tid; i < n_cb; i += 1024) a.q[i] = S;
        for (uint32_t i = tid; i <= S; i += 1024) a.slice_start[i] = i * n_cb;
        for (uint32_t i = tid; i < S * n_cb; i += 1024) {
            const uint32_t sl = i / n_cb;
            const uint64_t lo = (uint64_t)sl * w, hi = lo + w < a.n ? lo + w : a.n;
            a.items[i] = LongItem{i % n_cb, 0u, lo < a.n ? lo : a.n, hi};
        }
        if (tid == 0) *a.plan = LongPlan{S * n_cb, 0u, S, (uint32_t)(w < 0xFFFFFFFFu ? w : 0xFFFFFFFFu)};
        return;
    }
    for (uint32_t i = tid; i <= n_cb; i += 1024) cnt[i] = a.hist[i];
    __syncthreads();
    if (tid == 0) {   // cnt[cb] = records of a class above cb (classes 1 .. n_cb)
        uint32_t run = 0, classes = 0;
        for (uint32_t k = n_cb; k >= 1; --k) {
            const uint32_t h = cnt[k];
            cnt[k] = run;
            run += h;
            classes += h ? 1u : 0u;
        }
        cnt[0] = run;                          // (== n)
        sh_listed = classes > 1 ? 1u : 0u;
        unsigned long long off = 0;            // the lists, one behind the other
        for (uint32_t cb = 0; cb <= n_cb; ++cb) {
            a.list_off[cb] = off;
            off += cnt[cb];
        }
    }
    __syncthreads();
    // Reads of one class fill one round of blocks evenly, and a second round costs them 3 % (20 kbp: 1.142 / 1.177 / 1.221 ms per
    // 4 GiB in one / two / three rounds).  Reads of many lengths leave column blocks with a few records each, which take an
    // item (a block, a CU for that round) all the same: the launch has blocks for a finer cut, and the blocks that finish early
    // make room for the rest (log-normal around 5 kbp, 118 column blocks: 1.95 / 1.71 / 1.69 ms).
    const uint32_t nb = sh_listed ? a.nb : a.nb_one_length;
    // the smallest w with sum over cb of ceil(cnt[cb] / w) <= nb (one wavefront; the sum does not grow with w)
    if (tid < 64) {
        const uint32_t n32 = cnt[0];
        auto items_at = [&](uint32_t w) {
            unsigned long long t = 0;
            for (uint32_t cb = tid; cb < n_cb; cb += 64) t += ((unsigned long long)cnt[cb] + w - 1) / w;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) t += __shfl_xor(t, d);
            return t;
        };
        uint32_t lo = n32 < LONG_W_MIN ? (n32 ? n32 : 1u) : LONG_W_MIN, hi = n32 ? n32 : 1u;
        if (items_at(lo) <= nb) hi = lo;
        while (lo < hi) {                      // invariant: items_at(hi) <= nb (at w = n every column block in use has one item, and nb >= n_cb)
            const uint32_t mid = lo + (hi - lo) / 2;
            if (items_at(mid) <= nb) hi = mid;
            else lo = mid + 1;
        }
        if (tid == 0) {
            sh_w = hi;
            sh_qmax = (uint32_t)(((unsigned long long)n32 + hi - 1) / hi);
        }
    }
    __syncthreads();
    const uint32_t w = sh_w, q_max = sh_qmax;
    auto q_of = [&](uint32_t cb) { return cb < n_cb ? (uint32_t)(((unsigned long long)cnt[cb] + w - 1) / w) : 0u; };
    // column blocks per slice: q_of does not grow with cb, so the column blocks that have a slice s are a prefix — slice s
    // has cb + 1 of them for s in [q_of(cb + 1), q_of(cb))
    for (uint32_t cb = tid; cb < n_cb; cb += 1024) {
        const uint32_t qc = q_of(cb);
        a.q[cb] = qc;
        for (uint32_t sl = q_of(cb + 1); sl < qc; ++sl) start[sl + 1] = cb + 1;
    }
    __syncthreads();
    if (tid == 0) {
        start[0] = 0;
        for (uint32_t sl = 0; sl < q_max; ++sl) start[sl + 1] += start[sl];
    }
    __syncthreads();
    const uint32_t n_items = start[q_max];
    for (uint32_t i = tid; i <= q_max; i += 1024) a.slice_start[i] = start[i];
    for (uint32_t i = tid; i < n_items; i += 1024) {
        uint32_t lo = 0, hi = q_max;           // the slice of item i: start[sl] <= i < start[sl + 1]
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) / 2;
            if (start[mid] <= i) lo = mid;
            else hi = mid;
        }
        const uint32_t cb = i - start[lo];
        const uint64_t r_lo = (uint64_t)lo * w, r_hi = r_lo + w < cnt[cb] ? r_lo + w : cnt[cb];
        a.items[i] = LongItem{cb, 0u, r_lo, r_hi};
    }
    if (tid == 0) *a.plan = LongPlan{n_items, sh_listed, q_max, w};
}
// The lists: a block takes 1024 consecutive records (one per thread), counts them by class (as the census did), turns that
// into its records per column block, reserves that many places at the end of every list it adds to (one atomic per list and
// block), and its wavefronts write their records' numbers: for column block cb the lanes whose record reaches it, in lane
// order, behind what the block's wavefronts have written there so far (one LDS atomic per wavefront and column block, four
// column blocks in flight).  (4096 records per block, one column block at a time: 125 us for 300 000 reads of 370 .. 30 000
// bases — 74 blocks, each wavefront 4 x 118 dependent LDS round trips — of the 1.5 ms the walk behind it takes.)
constexpr uint32_t LONG_LIST_PER_BLOCK = 1024;
__global__ __launch_bounds__(1024) void k_long_lists(const fqh_idx_record *__restrict__ idx, uint64_t n, uint32_t n_cb, const LongPlan *__restrict__ plan,
                                                     const uint64_t *__restrict__ list_off, uint32_t *__restrict__ cursor, uint32_t *__restrict__ lists) {
    extern __shared__ uint32_t lh[];           // [n_cb + 1] the block's records per class -> written so far per column block; [n_cb] -> its place in each list
    if (!plan->listed) return;
    uint32_t *const lbase = lh + n_cb + 1;
    for (uint32_t i = threadIdx.x; i <= n_cb; i += 1024) lh[i] = 0;
    __syncthreads();
    const uint64_t r = (uint64_t)blockIdx.x * LONG_LIST_PER_BLOCK + threadIdx.x;
    const uint32_t kt = r < n ? long_class(idx, r, n_cb) : 0u;
    long_count(lh, kt, r < n);
    __syncthreads();
    if (threadIdx.x < 64) {                    // lbase[cb] = the block's records of a class above cb: a suffix sum, 64 entries at a time from the top
        uint32_t carry = 0;
        for (int hi = (int)n_cb; hi >= 1; hi -= 64) {
            const int kk = hi - (int)threadIdx.x;   // class of this lane (descending)
            uint32_t v = kk >= 1 ? lh[kk] : 0u;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t u = __shfl_up(v, d);
                if ((int)threadIdx.x >= d) v += u;
            }
            const uint32_t total = __shfl(v, 63);
            if (kk >= 1) lbase[kk - 1] = carry + v; // classes kk .. n_cb reach column block kk - 1 (not written over lh: the next 64 are still to be read)
            carry += total;
        }
    }
    __syncthreads();
    for (uint32_t cb = threadIdx.x; cb < n_cb; cb += 1024) {
        const uint32_t c = lbase[cb];
        lbase[cb] = c ? atomicAdd(&cursor[cb], c) : 0u;
        lh[cb] = 0;                            // -> written so far
    }
    __syncthreads();
    uint32_t kmax = kt;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) kmax = max(kmax, (uint32_t)__shfl_xor((int)kmax, d));
    kmax = (uint32_t)__builtin_amdgcn_readfirstlane((int)kmax);
    const uint32_t rid = (uint32_t)r;
    const unsigned long long below = (1ull << __lane_id()) - 1ull;
    for (uint32_t cb0 = 0; cb0 < kmax; cb0 += 4) {
        unsigned long long reach[4];
        uint32_t base[4];
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) {
            reach[j] = __ballot(kt > cb0 + j);     // (0 beyond kmax)
            base[j] = 0;
            if (__lane_id() == 0 && reach[j]) base[j] = atomicAdd(&lh[cb0 + j], (uint32_t)__popcll(reach[j]));
        }
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) {
            const uint32_t b = (uint32_t)__builtin_amdgcn_readfirstlane((int)base[j]);
            if (kt > cb0 + j) lists[list_off[cb0 + j] + lbase[cb0 + j] + b + (uint32_t)__popcll(reach[j] & below)] = rid;
        }
    }
}

// One block is resident per CU (its histogram fills the LDS), so the launch runs in ROUNDS of `cus` blocks, and a round
// costs ~3.5 % beyond its share of the records (zeroing and storing 136 KiB of counters, the start and the drain of its
// wavefronts' pipelines: 20 kbp reads, per 4 GiB, in one / two / three rounds 1.142 / 1.177 / 1.221 ms; before the rounds
// were software-pipelined the same cost 0.2 ms a round); a round that is not full wastes CUs for a whole block time.  So: R
// rounds of as many slices as fill them, R = the cheapest of 1 .. 4 under time ~ R * (cus / blocks + 0.05)  (units: the
// records' work spread evenly over the CUs).  This is the host's half of the
// plan — how many items the launch has blocks for — made as if every read reached every column block; the device deals them
// out by what the reads' lengths are (above).
static void stats_long_plan(uint64_t n, uint32_t max_line, int n_cu, uint32_t *n_cb, uint32_t *n_slices, uint32_t *n_slices_varied) {
    *n_cb = (max_line + SO_LC_MAX - 1) / SO_LC_MAX;
    if (*n_cb == 0) *n_cb = 1;
    const uint32_t cus = stats_blocks(n_cu);
    const uint64_t max_slices = std::max<uint64_t>(1, (n + LONG_W_MIN - 1) / LONG_W_MIN);
    uint32_t rounds_lo = 1, rounds_hi = 4;
#ifdef FQH_TUNING
    if (getenv("FQH_LONG_ROUNDS") && atoi(getenv("FQH_LONG_ROUNDS")) > 0) rounds_lo = rounds_hi = (uint32_t)atoi(getenv("FQH_LONG_ROUNDS"));
#endif
    double best = 1e30;
    *n_slices = 1;
    for (uint32_t R = rounds_lo; R <= rounds_hi; ++R) {
        uint64_t S = std::max<uint64_t>(1, (uint64_t)R * cus / *n_cb);
        if (S > max_slices) S = max_slices;
        const uint64_t blocks = S * *n_cb, r = (blocks + cus - 1) / cus;
        const double cost = (double)r * ((double)cus / (double)blocks + 0.05);
        if (cost < best) {
            best = cost;
            *n_slices = (uint32_t)S;
        }
    }
    // reads of many lengths (the device finds out): items for at least two rounds once the column blocks are many — most of them
    // then hold a few records each
    *n_slices_varied = *n_slices;
    if (*n_cb > cus / 8) {
        const uint64_t S2 = std::min<uint64_t>(std::max<uint64_t>(1, (uint64_t)2 * cus / *n_cb), max_slices);
        if (S2 > *n_slices_varied) *n_slices_varied = (uint32_t)S2;
    }
#ifdef FQH_TUNING
    if (getenv("FQH_LONG_ROUNDS")) *n_slices_varied = *n_slices;
#endif
}
// the scratch of one launch: every item's rows, the plan's arrays, the column blocks' lists
struct LongScratch {
    uint32_t n_cb, nb, nb_one_length;
    bool census;
    size_t part, hist, cursor, list_off, q, slice_start, items, plan, lists, bytes;   // byte offsets
};
static LongScratch stats_long_scratch(uint64_t n, uint64_t len, uint32_t max_line, int n_cu) {
    LongScratch L = {};
    uint32_t n_slices, n_slices_varied;
    stats_long_plan(n, max_line, n_cu, &L.n_cb, &n_slices, &n_slices_varied);
    L.nb_one_length = L.n_cb * n_slices;
    L.nb = L.n_cb * n_slices_varied;
    L.census = L.n_cb <= LONG_PLAN_MAX && L.nb <= LONG_PLAN_MAX && n < 0xFFFF0000ull;
#ifdef FQH_TUNING
    if (getenv("FQH_LONG_CENSUS") && atoi(getenv("FQH_LONG_CENSUS")) == 0) L.census = false;
#endif
    size_t o = 0;
    auto take = [&](size_t bytes) {
        const size_t at = o;
        o += (bytes + 15) & ~(size_t)15;
        return at;
    };
    L.part = take((size_t)L.nb * SO_LWORDS * sizeof(uint32_t));
    L.hist = take(((size_t)L.n_cb + 1) * sizeof(uint32_t));      // (hist and cursor: zeroed by one memset)
    L.cursor = take(((size_t)L.n_cb + 1) * sizeof(uint32_t));
    L.list_off = take(((size_t)L.n_cb + 1) * sizeof(uint64_t));
    L.q = take((size_t)L.n_cb * sizeof(uint32_t));
    L.slice_start = take(((size_t)L.nb + 1) * sizeof(uint32_t));
    L.items = take((size_t)L.nb * sizeof(LongItem));
    L.plan = take(sizeof(LongPlan));
    // a record is on max(1, ceil(line / 256)) lists, and the sequence lines of all records are at most half of the input
    L.lists = take(L.census ? (size_t)(len / (2 * SO_LC_MAX) + n + 1) * sizeof(uint32_t) : 0);
    L.bytes = o;
    return L;
}
// bytes of scratch launch_stats_long wants (a multiple of 16)
size_t stats_long_scratch_bytes(uint64_t n, uint64_t len, uint32_t max_line, int n_cu) { return stats_long_scratch(n, len, max_line, n_cu).bytes; }
// records [0, n) of idx; max_line: no sequence / quality line is longer (bounds the column blocks); flagmap: 2 * flag_words zeroed words;
// scratch: stats_long_scratch_bytes()
hipError_t launch_stats_long(hipStream_t s, const uint8_t *buf, uint64_t len, uint64_t base_offset, const fqh_idx_record *idx, uint64_t n,
                             uint32_t lmax, uint32_t max_line, uint32_t *flagmap, uint64_t flag_words, unsigned long long *qual_hist,
                             unsigned long long *base_hist, unsigned long long *scalars, int n_cu, uint32_t *scratch) {
    if (!n) return hipSuccess;
    const LongScratch L = stats_long_scratch(n, len, max_line, n_cu);
    uint8_t *const sc = reinterpret_cast<uint8_t *>(scratch);
    LongPlanArgs pa = {};
    pa.n = n;
    pa.n_cb = L.n_cb;
    pa.nb = L.nb;
    pa.nb_one_length = L.nb_one_length;
    pa.census = L.census ? 1u : 0u;
    pa.hist = reinterpret_cast<uint32_t *>(sc + L.hist);
    pa.list_off = reinterpret_cast<uint64_t *>(sc + L.list_off);
    pa.q = reinterpret_cast<uint32_t *>(sc + L.q);
    pa.slice_start = reinterpret_cast<uint32_t *>(sc + L.slice_start);
    pa.items = reinterpret_cast<LongItem *>(sc + L.items);
    pa.plan = reinterpret_cast<LongPlan *>(sc + L.plan);
    uint32_t *const lists = reinterpret_cast<uint32_t *>(sc + L.lists);
    const uint32_t sort_blocks = (uint32_t)((n + LONG_SORT_PER_BLOCK - 1) / LONG_SORT_PER_BLOCK);
    const size_t class_lds = ((size_t)L.n_cb + 1) * sizeof(uint32_t);
    if (L.census) {
        if (hipError_t e = hipMemsetAsync(sc + L.hist, 0, L.list_off - L.hist, s); e != hipSuccess) return e;
        hipLaunchKernelGGL(k_long_census, dim3(sort_blocks), dim3(1024), class_lds, s, idx, n, L.n_cb, reinterpret_cast<uint32_t *>(sc + L.hist));
    }
    hipLaunchKernelGGL(k_long_plan, dim3(1), dim3(1024), L.census ? ((size_t)L.n_cb + L.nb + 2) * sizeof(uint32_t) : 0, s, pa);
    if (L.census)
        hipLaunchKernelGGL(k_long_lists, dim3((uint32_t)((n + LONG_LIST_PER_BLOCK - 1) / LONG_LIST_PER_BLOCK)), dim3(1024), 2 * class_lds, s, idx, n, L.n_cb, pa.plan, pa.list_off,
                           reinterpret_cast<uint32_t *>(sc + L.cursor), lists);
    LongArgs a = {};
    a.buf = buf;
    a.len = len;
    a.base_offset = base_offset;
    a.idx = idx;
    a.lists = lists;
    a.list_off = pa.list_off;
    a.n = n;
    a.lmax = lmax;
    a.plan = pa.plan;
    a.items = pa.items;
    a.part = reinterpret_cast<uint32_t *>(sc + L.part);
    a.flagmap = flagmap;
    a.flag_words = flag_words;
    a.qual_hist = qual_hist;
    a.base_hist = base_hist;
    a.scalars = scalars;
    const size_t lds = SO_LADDR_SPAN;  // (a lane without a whole dword subtracts 0 wherever its bytes point: all of that is allocated)
    static LdsAttr attr;
    if (hipError_t e = attr.ensure(reinterpret_cast<const void *>(k_stats_long), lds); e != hipSuccess) return e;
    hipLaunchKernelGGL(k_stats_long, dim3(8u * ((L.nb + 7u) / 8u)), dim3(SO_THREADS), lds, s, a);
    hipLaunchKernelGGL(k_stats_long_reduce, dim3((SO_LWORDS + 255) / 256, L.n_cb), dim3(256), 0, s, a.part, pa.q, pa.slice_start, lmax, qual_hist,
                       base_hist);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// k_stats_head — the record in progress at the chunk start, for callers whose buffer also holds its
// beginning in front of the chunk (fqh_stats_launch_lead; the streaming ring).  One wavefront.  The
// record's line starts before the chunk come from the carry (distances back[]), those inside from
// the full line lists; it ends at the line start with global index 4 (r0 + 1), or at the end of the
// chunk.  Counting is the plain per-byte statement on the caller's u64 arrays.
__global__ __launch_bounds__(64) void k_stats_head(StatsArgs a, unsigned long long b0, unsigned long long b1,
                                                   unsigned long long b2, unsigned long long b3) {
    const uint32_t lane = threadIdx.x;
    const unsigned long long back[4] = {b0, b1, b2, b3};
    const unsigned long long r0 = a.nl_count >> 2;
    auto entry = [&](unsigned long long j, long long &off) -> bool {  // j-th line start of the chunk
        unsigned long long cum = 0;
        for (uint64_t t = 0; t < a.n_tiles; ++t) {
            uint32_t c = a.tile_count[t];
            c = c < a.list_cap ? c : a.list_cap;
            if (j < cum + c) {
                off = (long long)((t << WT_SHIFT) + (a.list[t * a.list_cap + (uint32_t)(j - cum)] & 0x3FFFu));
                return true;
            }
            cum += c;
        }
        return false;
    };
    long long p[5];
    bool ok = true;
    for (int i = 0; i < 5; ++i) {
        const unsigned long long g = 4 * r0 + i;  // global line index
        if (g <= a.nl_count) {
            p[i] = -(long long)back[a.nl_count - g];
        } else if (!entry(g - a.nl_count - 1, p[i])) {
            if (i == 4) p[i] = (long long)a.len;  // the record's last '\n' is the last byte of the chunk
            else ok = false;
        }
    }
    if (!ok) return;
    const uint8_t *const base = a.buf;
    unsigned long long n_bases = 0, n_qual = 0, oseq = 0, oqual = 0;
    uint32_t any_n = 0, any_inv = 0;
    for (int kind = 0; kind < 2; ++kind) {
        const long long s = kind ? p[3] : p[1];
        long long len = (kind ? p[4] : p[2]) - 1 - s;            // raw line, without its '\n'
        if (len > 0 && base[s + len - 1] == '\r') --len;         // trim_winline, src/records.rs:66-73
        if (len < 0) len = 0;
        if (kind) n_qual = (unsigned long long)len; else n_bases = (unsigned long long)len;
        for (long long col = lane; col < len; col += 64) {
            const uint32_t b = base[s + col];
            if (kind == 0) {
                const uint32_t c = base_class(b);
                any_inv |= c == 5 ? 1u : 0u;
                any_n |= c == 4 ? 1u : 0u;
                if (col < (long long)a.lmax) atomicAdd(&a.base_hist[(uint64_t)col * 8 + c], 1ull);
                else ++oseq;
            } else {
                if (col < (long long)a.lmax) atomicAdd(&a.qual_hist[(uint64_t)col * 256 + b], 1ull);
                else ++oqual;
            }
        }
    }
    const bool gi = __ballot(any_inv != 0) != 0, gn = __ballot(any_n != 0) != 0;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        oseq += __shfl_xor(oseq, d);
        oqual += __shfl_xor(oqual, d);
    }
    if (lane == 0) {
        atomicAdd(&a.scalars[0], 1ull);
        if (n_bases) atomicAdd(&a.scalars[1], n_bases);
        if (n_qual) atomicAdd(&a.scalars[2], n_qual);
        if (!gi && !gn) atomicAdd(&a.scalars[3], 1ull);
        if (!gi) atomicAdd(&a.scalars[4], 1ull);
        if (oseq) atomicAdd(&a.scalars[5], oseq);
        if (oqual) atomicAdd(&a.scalars[6], oqual);
    }
}
void launch_stats_head(hipStream_t s, const StatsArgs &a, const uint64_t back[4]) {
    hipLaunchKernelGGL(k_stats_head, dim3(1), dim3(64), 0, s, a, (unsigned long long)back[0],
                       (unsigned long long)back[1], (unsigned long long)back[2], (unsigned long long)back[3]);
}

// ---------------------------------------------------------------------------------------------
// Synthetic 150 bp FASTQ: byte b of record i is a pure function of (seed, i, b); the
// tests regenerate any sub-range on the CPU from the same map.
__device__ __forceinline__ unsigned long long mix64(unsigned long long x) {
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27; x *= 0x94D049BB133111EBull;
    x ^= x >> 31;
    return x;
}
__device__ __forceinline__ unsigned long long synth_hash(unsigned long long seed, unsigned long long rec,
                                                         uint32_t stream, uint32_t p) {
    return mix64(seed + rec * 0x9E3779B97F4A7C15ull +
                 ((((unsigned long long)stream) << 32 | p) + 1) * 0xD6E8FEB86659FD93ull);
}
__device__ uint32_t synth_byte(unsigned long long seed, unsigned long long rec, uint32_t b) {
    if (b < 26) {
        if (b < 5) return (uint32_t)("@SYN."[b]);
        if (b < 17) {
            unsigned long long v = rec % 1000000000000ull;
            for (uint32_t k = 16; k > b; --k) v /= 10;
            return '0' + (uint32_t)(v % 10);
        }
        return (uint32_t)(" 1:N:0:1\n"[b - 17]);
    }
    if (b < 176) {
        unsigned long long h = synth_hash(seed, rec, 1, b - 26);
        if ((uint32_t)(h >> 32) % 100u == 0) return 'N';
        return (uint32_t)("ACGT"[h & 3]);
    }
    if (b == 176) return '\n';
    if (b == 177) return '+';
    if (b == 178) return '\n';
    if (b < 329) {
        unsigned long long h = synth_hash(seed, rec, 2, b - 179);
        return '#' + (uint32_t)(h >> 32) % 39u;
    }
    return '\n';
}
__global__ __launch_bounds__(256) void k_synth(uint8_t *__restrict__ out, uint64_t byte_off, uint64_t len,
                                               unsigned long long seed) {
    const uint64_t nchunks = (len + 15) / 16;
    for (uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; c < nchunks;
         c += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t o = c * 16;
        unsigned long long pos = byte_off + o;
        unsigned long long rec = pos / 330;
        uint32_t b = (uint32_t)(pos % 330);
        uint32_t w0 = 0, w1 = 0, w2 = 0, w3 = 0;
        const uint32_t n = (len - o) < 16 ? (uint32_t)(len - o) : 16u;
        for (uint32_t i = 0; i < n; ++i) {
            const uint32_t v = synth_byte(seed, rec, b) << ((i & 3) * 8);
            if (i < 4) w0 |= v; else if (i < 8) w1 |= v; else if (i < 12) w2 |= v; else w3 |= v;
            if (++b == 330) { b = 0; ++rec; }
        }
        if (n == 16) {
            *reinterpret_cast<uint4 *>(out + o) = make_uint4(w0, w1, w2, w3);
        } else {
            for (uint32_t i = 0; i < n; ++i) {
                const uint32_t w = i < 4 ? w0 : i < 8 ? w1 : i < 12 ? w2 : w3;
                out[o + i] = (uint8_t)(w >> ((i & 3) * 8));
            }
        }
    }
}
void launch_synth(hipStream_t s, uint8_t *out, uint64_t byte_off, uint64_t len, uint64_t seed) {
    if (!len) return;
    uint64_t blocks = ((len + 15) / 16 + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(k_synth, dim3((uint32_t)blocks), dim3(256), 0, s, out, byte_off, len,
                       (unsigned long long)seed);
}

// ---------------------------------------------------------------------------------------------
// Streaming-read ceiling: the same access pattern as k_index (a wavefront reads a 16 KiB tile as
// 1 KiB pieces, four 16-byte loads in flight per lane) with only an integer sum as work.  Persistent
// grid so that the final atomics do not serialise.
__global__ __launch_bounds__(256) void k_read_ceiling(const uint8_t *__restrict__ buf, uint64_t len,
                                                      unsigned long long *__restrict__ sum) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t n_tiles = (len + WT_BYTES - 1) >> WT_SHIFT;
    const uint64_t wave0 = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint64_t nwaves = (uint64_t)gridDim.x * 4;
    unsigned long long acc = 0;
    for (uint64_t tile = wave0; tile < n_tiles; tile += nwaves) {
        const uint64_t tbase = tile << WT_SHIFT;
        if (tbase + WT_BYTES <= len) {
#pragma unroll 1
            for (uint32_t g = 0; g < WT_PIECES / 4; ++g) {
                const uint8_t *p = buf + tbase + g * 4 * PIECE_BYTES + lane * 16;
                typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
                const u32x4 v0 = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p));
                const u32x4 v1 = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p + PIECE_BYTES));
                const u32x4 v2 = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p + 2 * PIECE_BYTES));
                const u32x4 v3 = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p + 3 * PIECE_BYTES));
                acc += (unsigned long long)v0.x + v0.y + v0.z + v0.w;
                acc += (unsigned long long)v1.x + v1.y + v1.z + v1.w;
                acc += (unsigned long long)v2.x + v2.y + v2.z + v2.w;
                acc += (unsigned long long)v3.x + v3.y + v3.z + v3.w;
            }
        } else {
            for (uint64_t o = tbase + lane * 4; o + 4 <= len && o < tbase + WT_BYTES; o += 256)
                acc += *reinterpret_cast<const uint32_t *>(buf + o);
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d);
    if (lane == 0 && acc) atomicAdd(sum, acc);
}
void launch_read_ceiling(hipStream_t s, const uint8_t *buf, uint64_t len, uint64_t *sum, int n_cu) {
    const uint64_t n_tiles = (len + WT_BYTES - 1) / WT_BYTES;
    if (!n_tiles) return;
    uint64_t blocks = (n_tiles + 3) / 4;
    const uint64_t maxb = (uint64_t)(n_cu > 0 ? n_cu : 256) * 8;
    if (blocks > maxb) blocks = maxb;
    hipLaunchKernelGGL(k_read_ceiling, dim3((uint32_t)blocks), dim3(256), 0, s, buf, len,
                       (unsigned long long *)sum);
}

// k_len_hist: the read-length histogram out of the base histogram (fqh_len_hist): reads longer than p = the counts of column p.
__global__ __launch_bounds__(256) void k_len_hist(const unsigned long long *__restrict__ base_hist,
                                                  const unsigned long long *__restrict__ scalars, uint32_t lmax,
                                                  unsigned long long *__restrict__ len_hist) {
    const uint32_t L = blockIdx.x * blockDim.x + threadIdx.x;
    if (L > lmax) return;
    auto longer_than = [&](uint32_t p) {  // reads with more than p bases
        unsigned long long c = 0;
        for (uint32_t k = 0; k < 8; ++k) c += base_hist[(uint64_t)p * 8 + k];
        return c;
    };
    const unsigned long long at_least = L == 0 ? scalars[0] : longer_than(L - 1);   // reads with at least L bases
    const unsigned long long v = L == lmax ? at_least : at_least - longer_than(L);
    if (v) atomicAdd(&len_hist[L], v);
}
void launch_len_hist(hipStream_t s, const unsigned long long *base_hist, const unsigned long long *scalars, uint32_t lmax,
                     unsigned long long *len_hist) {
    hipLaunchKernelGGL(k_len_hist, dim3((lmax + 1 + 255) / 256), dim3(256), 0, s, base_hist, scalars, lmax, len_hist);
}

}  // namespace fqh