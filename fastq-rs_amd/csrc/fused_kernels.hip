// fused_kernels.hip — k_scan_stats: record scan AND per-position histograms in ONE read of the input
// (DESIGN.md §5b).  The reference touches every record once: Parser::each hands it to the closure that
// reads seq()/qual() (src/lib.rs:226-237, src/records.rs:83-90).  Here the byte scan of the fast path
// (k_index_fast, scan_kernels.hip) and the eight-lanes-per-line count of k_stats_oct (stats_dev.h) run in
// the same wavefront on the same LDS image of the data:
//
//   * one 1024-thread block per CU owns the CU's LDS: the bank-scheduled histogram (stats_dev.h) and,
//     behind it, 5 KiB per wavefront: [512 B tail of the previous group | 4 KiB group | 128 B tile line |
//     4 + 188 line-start entries];
//   * a wavefront takes 16 KiB tiles round-robin, 4 KiB groups at a time, exactly like k_index_fast:
//     16-byte non-temporal loads a group ahead, LDS transposition, SWAR newline masks, ballot prefix;
//   * the group's line starts are staged in LDS; 5-entry windows ('@' on line i, '+' on line i+2, equal
//     raw lengths of lines i+1 and i+3) are checked under the four possible alignments.  The tile's FIRST
//     group must single out one alignment: that is the phase the tile's lines are counted under, and at
//     the tile's end it must still be the only consistent one.  k_emit_fast later checks it against the
//     true global line index; any doubt sets spec_fail and nothing of this pass is used;
//   * every line that ENDS in the group (its successor's start is one of the group's entries) is counted
//     from LDS: one lane per line works out start / length, batches of eight lines are read back with
//     aligned ds_read_b32 (the tail keeps the 512 bytes in front of the group, so a line that straddles
//     two groups is contiguous) and shifted into place with v_alignbyte and two DPP moves; then the
//     straight-line pass 1 / pass 2 of so_count: one v_perm_b32 + one ds_sub_u32 per byte;
//   * the tile's last line ends in another wavefront's tile: it is counted byte-wise from global memory
//     (one line per tile), as is nothing else in well-formed input;
//   * per-block partial histograms, per-lane totals and the exact path's 64-bit counters go to scratch;
//     k_stats_commit adds them to the caller's arrays only if the scan's finalize kernel found no reason
//     to doubt the fast path.
//
// Algorithmic bytes: len per launch — the only read of the input for offsets, validation and histograms.
#include <hip/hip_runtime.h>

#include "scan_dev.h"
#include "stats_dev.h"

namespace fqh {

constexpr uint32_t FZ_THREADS = 1024;
constexpr uint32_t FZ_WAVES = FZ_THREADS / 64;
constexpr uint32_t FZ_GROUP = 4096;                 // bytes per group: 64 contiguous bytes per lane
constexpr uint32_t FZ_TAIL = 512;                   // bytes of the previous group kept in front of the group
constexpr uint32_t FZ_DATA = FZ_TAIL + FZ_GROUP;
constexpr uint32_t FZ_GLIST = 188;                  // line starts per group that can be staged
constexpr uint32_t FZ_WAVE_BYTES = FZ_DATA + 128 + (4 + FZ_GLIST) * 2;  // 5120
constexpr uint32_t FZ_SLACK = 512;                  // a batch reads up to 32 NSL + 32 bytes past a line's start
static_assert(FZ_WAVE_BYTES % 64 == 0, "wave areas must keep the 64-byte blocks of the swizzle");

// LDS image of a wave's data: logical address L -> physical L ^ (bits 8-9 of L moved to bits 4-5).  Permutes
// the four 16-byte chunks of every aligned 64-byte block: conflict-free for the lane-strided ds_write_b128
// of the loads and for the lane-contiguous ds_read_b128 of the scan, whatever the area's (64-byte aligned) base.
__device__ __forceinline__ uint32_t fz_swz(uint32_t L) { return L ^ ((L >> 4) & 0x30u); }

template <uint32_t NSL>
__global__ __launch_bounds__(FZ_THREADS) void k_scan_stats(StatsArgs a, FusedArgs z) {
    extern __shared__ __attribute__((aligned(16))) uint32_t hist[];
    const uint32_t lc = a.lc;
    const uint32_t wb0 = z.wave_base;  // bytes of histogram in front of the waves' areas
    for (uint32_t i = threadIdx.x; i < wb0 / 4; i += FZ_THREADS) hist[i] = 0;
    __syncthreads();
    // The address registers assume the histogram starts at LDS address 0 (the kernel's only LDS object).
    if ((uint32_t)(uintptr_t)hist != 0) __builtin_trap();
    uint8_t *const lds8 = reinterpret_cast<uint8_t *>(hist);

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t m4 = (lane & 7u) * 4u, g8 = lane >> 3;
    const bool is7 = (lane & 7u) == 7u;
    const uint32_t wbase = wb0 + wv * FZ_WAVE_BYTES;                 // logical address of y = 0
    uint8_t *const wptr = lds8 + fz_swz(wbase + FZ_TAIL + 16u * lane);  // chunk 64 j + lane of the group: + 1024 j
    const uint32_t L0 = wbase + FZ_TAIL + 64u * lane;                // this lane's 64 contiguous bytes
    const uint32_t s4 = ((L0 >> 8) & 3u) << 4;                       // byte Q of the lane at L0 + (Q ^ s4)
    const uint8_t *const rptr = lds8 + L0;
    uint16_t *const tline = reinterpret_cast<uint16_t *>(lds8 + wbase + FZ_DATA);  // the tile's line, 64 u16
    uint16_t *const lst = tline + 64 + 4;  // the group's entries; lst[-4 .. -1]: the last four before the group

    SoLane c;
    c.slots = 0;
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) {
        const uint32_t j = k ^ (g8 & 3u);
        c.sel[k] = 0x0C0C0004u + k + (j << 8);
        c.slots |= (((lane & 7u) + 8u * j) * 4u) << (8u * k);
    }
    SoAcc acc = {0, 0, 0};
    SoTotals T = {0, 0};
    bool cr_seen = false;
    SoShape<NSL> S = {};
    S.key = 0xFFFFFFFFu;

    const uint8_t *__restrict__ const buf = a.buf;
    const uint64_t len = a.len;
    const uint64_t n_tiles = z.n_tiles;
    const uint64_t n_full = len >> WT_SHIFT;
    const uint64_t nwaves = (uint64_t)gridDim.x * FZ_WAVES;
    const uint32_t lo = lane * 16u;
    uint32_t n_over = 0;

    // the four 16-byte pieces of this lane for group g of tile t (whole tiles: unconditional loads)
    auto fetch_group = [&](uint64_t t, uint32_t g, uint4 &n0, uint4 &n1, uint4 &n2, uint4 &n3) {
        const uint64_t off = (t << WT_SHIFT) + g * FZ_GROUP + lo;
        if (t < n_full) {
            const uint8_t *p = buf + off;
            n0 = load16_nt(p); n1 = load16_nt(p + PIECE_BYTES);
            n2 = load16_nt(p + 2 * PIECE_BYTES); n3 = load16_nt(p + 3 * PIECE_BYTES);
        } else {  // the partial tile at the end of the buffer: bytes at or beyond len read as 0
            n0 = load16(buf, off, len); n1 = load16(buf, off + PIECE_BYTES, len);
            n2 = load16(buf, off + 2 * PIECE_BYTES, len); n3 = load16(buf, off + 3 * PIECE_BYTES, len);
        }
    };

    uint64_t tile = (uint64_t)blockIdx.x * FZ_WAVES + wv;
    if (tile < n_tiles) {
        uint4 n0, n1, n2, n3;
        fetch_group(tile, 0, n0, n1, n2, n3);
        uint32_t pb = tile ? buf[(tile << WT_SHIFT) - 1] : 0u;  // the byte before the tile
        bool pending = false;  // the previous tile's line is still in a register
        uint64_t ptile = 0;
        uint32_t prv = 0;
        for (; tile < n_tiles; tile += nwaves) {
            const uint64_t nxt = tile + nwaves < n_tiles ? tile + nwaves : tile;  // clamped: the prefetch is unconditional
            const uint64_t tb = tile << WT_SHIFT;
            const bool full = tile < n_full;
            const uint32_t tile_bytes = full ? WT_BYTES : (uint32_t)(len - tb);
            uint32_t run = 0;        // entries of the tile before the current group
            uint32_t tot = 0;        // entries of the current group
            uint32_t prev = (tile && pb == '\n') ? 1u : 0u;
            uint32_t hyp = 7;        // the tile's alignment: entries hyp, hyp + 4, .. start records
            bool tile_bad = false;
            uint32_t have = 0, bad = 0;  // bit r: some / some failing window of five entries starting at r (mod 4)
            __builtin_amdgcn_wave_barrier();
            tline[lane] = 0;
#pragma unroll 1
            for (uint32_t g = 0; g < WT_BYTES / FZ_GROUP; ++g) {
                __builtin_amdgcn_wave_barrier();
                *reinterpret_cast<uint4 *>(wptr) = n0;
                *reinterpret_cast<uint4 *>(wptr + 1024) = n1;
                *reinterpret_cast<uint4 *>(wptr + 2048) = n2;
                *reinterpret_cast<uint4 *>(wptr + 3072) = n3;
                {  // next group of this tile, or the first group of the wave's next tile
                    const bool last = g + 1 == WT_BYTES / FZ_GROUP;
                    if (last) pb = buf[(nxt << WT_SHIFT) - (nxt ? 1 : 0)];
                    fetch_group(last ? nxt : tile, last ? 0u : g + 1, n0, n1, n2, n3);
                }
                if (g == 0 && pending)  // a whole group before the next wait on vmcnt
                    __builtin_nontemporal_store((uint16_t)prv, z.fast_rs + ptile * FR_STRIDE + lane);
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
                if (!full) {  // a line start must be an existing byte
                    const int nv = (int)tile_bytes - (int)(g * FZ_GROUP + lane * 64u);
                    const uint32_t nvalid = nv < 0 ? 0u : nv > 64 ? 64u : (uint32_t)nv;
                    const unsigned long long keep = nvalid >= 64 ? ~0ull : ((1ull << nvalid) - 1ull);
                    ls_lo &= (uint32_t)keep;
                    ls_hi &= (uint32_t)(keep >> 32);
                }
                const uint32_t cl = __popc(ls_lo) + __popc(ls_hi);
                const unsigned long long b1 = __ballot(cl >= 1), b2 = __ballot(cl >= 2), b3 = __ballot(cl >= 3);
                uint32_t pre = __builtin_amdgcn_mbcnt_hi((uint32_t)(b1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b1, 0));
                pre = __builtin_amdgcn_mbcnt_hi((uint32_t)(b2 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b2, pre));
                pre = __builtin_amdgcn_mbcnt_hi((uint32_t)(b3 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b3, pre));
                uint32_t gtot = (uint32_t)__popcll(b1) + (uint32_t)__popcll(b2) + (uint32_t)__popcll(b3);
                if (__ballot(cl >= 4)) {
                    for (uint32_t k = 4;; ++k) {
                        const unsigned long long b = __ballot(cl >= k);
                        if (!b) break;
                        pre = __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, pre));
                        gtot += (uint32_t)__popcll(b);
                    }
                }
                run += tot;   // the previous group's entries are behind us now
                tot = gtot;
                if (tot > FZ_GLIST) {  // lines shorter than ~22 bytes on average: left to the exact path
                    tile_bad = true;
                    tot = 0;           // (the header below stays what it was: nothing more is counted in this tile)
                    run += gtot;
                    continue;
                }
                const uint32_t ebase = g * FZ_GROUP + lane * 64u;
                {
                    uint16_t *dst = lst + pre;
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
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                // ---- windows of five entries that end in this group (src/records.rs:141,155,233 under each alignment)
                for (uint32_t p = lane; p < tot; p += 64) {
                    const uint32_t ti = run + p;
                    if (ti < 4) {
                        tline[FR_EDGE + ti] = lst[p];  // the tile's first four entries
                        continue;
                    }
                    const uint32_t e0 = lst[(int)p - 4], e1 = lst[(int)p - 3], e2 = lst[(int)p - 2], e3 = lst[(int)p - 1], e4 = lst[p];
                    const bool ok = (e0 & 0x4000u) && (e2 & 0x8000u) &&
                                    ((e2 & 0x3FFFu) - (e1 & 0x3FFFu)) == ((e4 & 0x3FFFu) - (e3 & 0x3FFFu));
                    have |= 1u << (ti & 3u);
                    bad |= ok ? 0u : 1u << (ti & 3u);
                }
                if (g == 0) {  // the first group must single out the alignment the tile is counted under
                    uint32_t cons = 0;
#pragma unroll
                    for (uint32_t r = 0; r < 4; ++r)
                        if (__ballot((have >> r) & 1u) && !__ballot((bad >> r) & 1u)) cons |= 1u << r;
                    if (cons && !(cons & (cons - 1))) hyp = (uint32_t)__ffs(cons) - 1;
                    else tile_bad = true;
                }
                if (hyp < 4 && !tile_bad && tot) {
                    // ---- one lane per line that ends in this group.  Entry p closes the line that entry p - 1 starts.
                    // Lines of tile index i: i == hyp (mod 4) header, + 1 sequence, + 2 separator, + 3 quality.
                    const uint32_t pq = (hyp - run) & 3u;          // these entries start a record and close a quality line
                    const uint32_t ps = (pq + 2u) & 3u;            // these close a sequence line
                    const int gofs = (int)FZ_TAIL - (int)(g * FZ_GROUP);  // tile offset -> y (the wave's data area)
                    const uint8_t *const ybase = buf + tb - gofs;  // global address of y = 0
                    uint32_t P_s = 0, P_q = 0, l_s = 0, l_q = 0;
                    bool far = false;
                    {
                        const uint32_t p = ps + 4u * lane;
                        if (p < tot && (run | p) != 0) {
                            const int s = (int)(lst[(int)p - 1] & 0x3FFFu), n = (int)(lst[p] & 0x3FFFu);
                            const int ys = s + gofs;
                            uint32_t l = (uint32_t)(n - 1 - s);
                            if (ys < 0) far = true;
                            else {
                                if (cr_seen && l && lds8[fz_swz(wbase + (uint32_t)ys + l - 1)] == '\r') --l;  // trim_winline, src/records.rs:66-73
                                l_s = l;
                                P_s = so_pack((uint32_t)ys, l, lc);
                                ++acc.rec;
                                acc.bases += l;
                            }
                        }
                    }
                    {
                        const uint32_t p = pq + 4u * lane;
                        if (p < tot) {
                            const uint32_t e = lst[p] & 0x3FFFu;
                            // record start k of the tile: the first FR_N in the tile's line, then a second line, then the list area
                            const uint32_t k = (run + p - hyp) >> 2;
                            if (run + p >= hyp) {
                                if (k < FR_N) tline[k] = (uint16_t)e;
                                else if (k < FR_N + FR2_N) z.fast_rs[fr2_off(n_tiles) + tile * FR2_N + (k - FR_N)] = (uint16_t)e;
                                else z.list[tile * z.list_cap + 8 + k] = (uint16_t)e;
                            }
                            if ((run | p) != 0) {
                                const int s = (int)(lst[(int)p - 1] & 0x3FFFu);
                                const int ys = s + gofs;
                                uint32_t l = (uint32_t)((int)e - 1 - s);
                                if (ys < 0) far = true;
                                else {
                                    if (cr_seen && l && lds8[fz_swz(wbase + (uint32_t)ys + l - 1)] == '\r') --l;
                                    l_q = l;
                                    P_q = so_pack((uint32_t)ys, l, lc);
                                    acc.qual += l;
                                }
                            }
                        }
                    }
                    if (__ballot(far)) tile_bad = true;  // a line that began before the kept tail (longer than ~500 bytes)
                    const uint32_t nls = tot > ps ? (tot - ps + 3) >> 2 : 0u, nlq = tot > pq ? (tot - pq + 3) >> 2 : 0u;
                    const uint32_t nbs = (nls + 7) >> 3, nbq = (nlq + 7) >> 3;
                    const uint32_t nbm = nbs > nbq ? nbs : nbq;
                    const uint32_t nbt = 2u * nbm;
                    const bool probe = cr_seen;
                    auto lookup = [&](uint32_t f) -> uint32_t {
                        const bool isq = (f & 1u) != 0;
                        const uint32_t b = f >> 1;
                        uint32_t P = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(32u * b + 4u * g8), (int)(isq ? P_q : P_s));
                        if (b >= (isq ? nbq : nbs)) P = 0;
                        __builtin_amdgcn_sched_barrier(0);  // (keep it ahead of the count's atomics)
                        return P;
                    };
                    // this lane's dword of every step of its line: aligned reads, the lane above supplies the
                    // bytes that complete it (lane 7 of a line: lane 0's dword of the next step)
                    auto fetch = [&](uint32_t P, SoBatch<NSL> &B) {
                        B.P = P;
                        const uint32_t ys = P >> SO_P_SREL;
                        const uint32_t L = wbase + (ys & ~3u) + m4;
                        uint32_t W[NSL + 1];
#pragma unroll
                        for (uint32_t u = 0; u <= NSL; ++u)
                            W[u] = *reinterpret_cast<const uint32_t *>(lds8 + fz_swz(L + 32u * u));
#pragma unroll
                        for (uint32_t u = 0; u < NSL; ++u) {
                            const uint32_t up = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)W[u], 0x101, 0xF, 0xF, false);      // row_shl:1
                            const uint32_t nx = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)W[u + 1], 0x117, 0xF, 0xF, false);  // row_shr:7
                            B.w[u] = __builtin_amdgcn_alignbyte(is7 ? nx : up, W[u], ys & 3u);
                        }
                    };
                    if (nbt) {
                        SoBatch<NSL> B0, B1;  // ping-pong: the reads of one are in flight while the other is counted
                        const uint32_t fl = nbt - 1;
                        uint32_t pa = lookup(0), pbq = lookup(1);
                        fetch(pa, B0);
                        for (uint32_t f = 0; f < nbt; f += 2) {
                            fetch(pbq, B1);
                            pa = lookup(f + 2 < fl ? f + 2 : fl);
                            so_count<true, NSL, false>(a, ybase, B0, S, lane, lc, hist, c, l_s, 32u * (f >> 1) + 4u * g8, T, acc, probe, cr_seen);
                            fetch(pa, B0);
                            pbq = lookup(f + 3 < fl ? f + 3 : fl);
                            so_count<false, NSL, false>(a, ybase, B1, S, lane, lc, hist, c, l_q, 32u * (f >> 1) + 4u * g8, T, acc, probe, cr_seen);
                        }
                    }
                }
                // ---- the next group finds this one's last 512 bytes and last four entries in front of its own
                __builtin_amdgcn_wave_barrier();
                {
                    const uint2 tv = *reinterpret_cast<const uint2 *>(lds8 + wbase + FZ_GROUP + 8u * lane);
                    const uint32_t hv = lane < 4 ? (uint32_t)lst[(int)tot - 4 + (int)lane] : 0u;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    *reinterpret_cast<uint2 *>(lds8 + wbase + 8u * lane) = tv;
                    if (lane < 4) lst[(int)lane - 4] = (uint16_t)hv;
                }
            }
            run += tot;  // entries of the whole tile
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // ---- the tile's alignment must still be the only consistent one
            {
                uint32_t cons = 0;
#pragma unroll
                for (uint32_t r = 0; r < 4; ++r)
                    if (__ballot((have >> r) & 1u) && !__ballot((bad >> r) & 1u)) cons |= 1u << r;
                if (run < 8 || cons != (1u << hyp)) tile_bad = true;
            }
            // ---- the tile's last line ends in another wavefront's tile (or with the tile's, or the buffer's, last
            // byte): counted here, byte-wise from global memory
            if (!tile_bad) {
                const uint32_t kind = (run - 1u - hyp) & 3u;
                if (kind == 1u || kind == 3u) {
                    const uint32_t e_last = lst[-1] & 0x3FFFu;  // (the header holds the tile's last four entries now)
                    const uint64_t S0 = tb + e_last;
                    const uint8_t *const bend = buf + len;
                    const bool last_nl = full ? prev != 0 : buf[len - 1] == '\n';
                    uint64_t end = 0;
                    bool found = false;
                    if (last_nl) {
                        end = tb + tile_bytes - 1;
                        found = true;
                    } else {
                        uint64_t pos = tb + tile_bytes;
                        for (uint32_t it = 0; it < 64 && pos < len && !found; ++it, pos += 256) {
                            const uint32_t w = load4_any(buf + pos + 4u * lane, bend);
                            const uint32_t fl = eq_flags(w, 0x0A0A0A0Au);
                            const unsigned long long bm = __ballot(fl != 0);
                            if (bm) {
                                const uint32_t first = (uint32_t)__ffsll((long long)bm) - 1u;
                                const uint32_t ff = (uint32_t)__builtin_amdgcn_readlane((int)fl, (int)first);
                                end = pos + 4u * first + ((uint32_t)__ffs(ff) - 1u) / 8u;
                                found = true;
                            }
                        }
                        if (!found && pos < len) tile_bad = true;  // a line of more than 16 KiB
                    }
                    if (found) {
                        uint32_t l = (uint32_t)(end - S0);
                        if (l && buf[end - 1] == '\r') --l;  // trim_winline, src/records.rs:66-73
                        uint32_t any_n = 0, any_inv = 0;
                        for (uint32_t pos = 4u * lane; pos < l; pos += 256) {
                            const uint32_t w = load4_any(buf + S0 + pos, bend);
                            if (kind == 1u) so_exact_step<true>(a, w, pos, l, lc, hist, any_n, any_inv);
                            else so_exact_step<false>(a, w, pos, l, lc, hist, any_n, any_inv);
                        }
                        if (kind == 1u) {
                            const bool gi = __ballot(any_inv != 0) != 0, gn = __ballot(any_n != 0) != 0;
                            T.not_dna += (gi || gn) ? 1u : 0u;
                            T.not_dnan += gi ? 1u : 0u;
                            if (lane == 0) { ++acc.rec; acc.bases += l; }
                        } else if (lane == 0) {
                            acc.qual += l;
                        }
                    }
                }
            }
            // ---- the tile's line: record starts (the lanes wrote them), first and last four entries, count, alignment
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            {
                uint32_t rv = tline[lane];
                if (lane >= FR_EDGE + 4 && lane < FR_EDGE + 8) rv = lst[(int)lane - (int)(FR_EDGE + 8)];
                if (tile_bad) ++n_over;
                prv = lane == FR_CNT ? (run & 0xFFFFu) : lane == FR_CNT + 1 ? (run >> 16) : lane == FR_HYP ? (tile_bad ? 7u : hyp) : rv;
            }
            ptile = tile;
            pending = true;
            __builtin_amdgcn_wave_barrier();
        }
        if (pending) __builtin_nontemporal_store((uint16_t)prv, z.fast_rs + ptile * FR_STRIDE + lane);
    }
    if (lane == 0 && n_over) atomicAdd(&z.out->spec_fail, (unsigned long long)n_over);

    // ---- per-block partial histogram, per-wave totals
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    uint32_t *__restrict__ dst = a.scratch + (uint64_t)blockIdx.x * SO_WORDS;
    for (uint32_t i = threadIdx.x; i < wb0 / 4; i += FZ_THREADS) dst[i] = hist[i];
    unsigned long long sc[5] = {acc.rec, acc.bases, acc.qual, 0, 0};
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        unsigned long long v = sc[j];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
        sc[j] = v;
    }
    if (lane == 0) {
        sc[3] = sc[0] - T.not_dna;
        sc[4] = sc[0] - T.not_dnan;
#pragma unroll
        for (int j = 0; j < 5; ++j)
            if (sc[j]) atomicAdd(&a.scalars[j], sc[j]);
    }
}

// k_stats_commit: adds what k_scan_stats left in scratch to the caller's arrays — per-block partial histograms
// (bank-scheduled layout), the exact path's 64-bit counters and the scalars — if and only if the scan's
// finalize kernel kept the fast path's result (DevOut::stats_commit).
__global__ __launch_bounds__(256) void k_stats_commit(const DevOut *__restrict__ out, const uint32_t *__restrict__ scratch,
                                                      uint32_t n_blocks, uint32_t lc, uint32_t lmax, uint32_t words,
                                                      const unsigned long long *__restrict__ src_qual,
                                                      const unsigned long long *__restrict__ src_base,
                                                      const unsigned long long *__restrict__ src_scalars,
                                                      unsigned long long *__restrict__ qual_hist,
                                                      unsigned long long *__restrict__ base_hist,
                                                      unsigned long long *__restrict__ scalars) {
    if (!out->stats_commit) return;
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (blockIdx.y == 0) {  // the 64-bit side arrays (out-of-window bytes, columns beyond the LDS rows), once
        for (uint32_t i = id; i < lmax * 256u; i += gridDim.x * blockDim.x) {
            const unsigned long long v = src_qual[i];
            if (v) atomicAdd(&qual_hist[i], v);
        }
        for (uint32_t i = id; i < lmax * 8u; i += gridDim.x * blockDim.x) {
            const unsigned long long v = src_base[i];
            if (v) atomicAdd(&base_hist[i], v);
        }
        if (id < FQH_NSCALARS && src_scalars[id]) atomicAdd(&scalars[id], src_scalars[id]);
    }
    if (id >= words) return;
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
    if (isq) atomicAdd(&qual_hist[(uint64_t)row * 256 + 33 + bin], s);
    else atomicAdd(&base_hist[(uint64_t)row * 8 + bin_to_class(bin)], s);  // bins 0,2,5 share class 5
}

uint32_t stats_blocks(int n_cu);

// can the single-pass kernel take this call's lmax?  (columns beyond the 256 bank-scheduled rows have no LDS here)
bool scan_stats_supports(uint32_t lmax) { return lmax >= 1 && lmax <= SO_LC_MAX; }
uint32_t scan_stats_blocks(uint64_t n_tiles, int n_cu) {
    const uint64_t want = (n_tiles + FZ_WAVES - 1) / FZ_WAVES;
    const uint32_t cus = stats_blocks(n_cu);
    return (uint32_t)(want < cus ? (want ? want : 1) : cus);
}
size_t scan_stats_scratch_bytes(int n_cu) { return (size_t)stats_blocks(n_cu) * SO_WORDS * sizeof(uint32_t); }

template <uint32_t NSL>
static hipError_t launch_scan_stats_n(hipStream_t s, const StatsArgs &a, FusedArgs z, uint32_t blocks) {
    z.wave_base = SO_SBYTES + ((NSL + 1) / 2) * 16384u;
    const size_t lds = (size_t)z.wave_base + (size_t)FZ_WAVES * FZ_WAVE_BYTES + FZ_SLACK;
    static_assert(SO_SBYTES + ((NSL + 1) / 2) * 16384u + FZ_WAVES * FZ_WAVE_BYTES + FZ_SLACK <= SO_LDS_MAX, "LDS budget");
    // lanes without a whole dword subtract 0 at the address their bytes form (any bin byte plus the largest
    // row-block offset): inside the allocation, and harmless wherever it lands (stats_dev.h)
    static_assert(65536 + SO_SBYTES + 128 + ((NSL - 1) / 2) * 16384u <= SO_SBYTES + ((NSL + 1) / 2) * 16384u + FZ_WAVES * FZ_WAVE_BYTES,
                  "garbage addresses must stay inside the allocation");
    static bool set = false;
    if (!set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_scan_stats<NSL>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        set = true;
    }
    hipLaunchKernelGGL((k_scan_stats<NSL>), dim3(blocks), dim3(FZ_THREADS), lds, s, a, z);
    return hipSuccess;
}

// a: buf, len, lmax, scratch (scan_stats_scratch_bytes), qual_hist / base_hist / scalars = ZEROED side arrays of
// lmax * 256, lmax * 8 and FQH_NSCALARS u64 (not the caller's: see k_stats_commit)
hipError_t launch_scan_stats(hipStream_t s, StatsArgs a, FusedArgs z, int n_cu) {
    a.lc = a.lmax < SO_LC_MAX ? a.lmax : SO_LC_MAX;
    a.lx = 0;
    a.listw = 0;
    a.dbg = 0;
    const uint32_t blocks = scan_stats_blocks(z.n_tiles, n_cu);
    const uint32_t nsl = (a.lc + 31) / 32;
    hipError_t e = nsl <= 5 ? launch_scan_stats_n<5>(s, a, z, blocks) : launch_scan_stats_n<8>(s, a, z, blocks);
    if (e != hipSuccess) return e;
    return hipGetLastError();
}
void launch_stats_commit(hipStream_t s, const DevOut *out, const StatsArgs &a, uint32_t blocks,
                         unsigned long long *qual_hist, unsigned long long *base_hist, unsigned long long *scalars) {
    const uint32_t lc = a.lmax < SO_LC_MAX ? a.lmax : SO_LC_MAX;
    const uint32_t nsl = (lc + 31) / 32;
    const uint32_t words = (SO_SBYTES + ((nsl <= 5 ? 5u : 8u) + 1) / 2 * 16384u) / 4;
    hipLaunchKernelGGL(k_stats_commit, dim3((words + 255) / 256, (blocks + RED_GROUP - 1) / RED_GROUP), dim3(256), 0, s, out,
                       a.scratch, blocks, lc, a.lmax, words, a.qual_hist, a.base_hist, a.scalars, qual_hist, base_hist, scalars);
}

}  // namespace fqh
