// replay.h — host-side replay of the reference's 68 KiB Buffer, used to decide "Fastq record is
// too long" exactly as the crate does.
//
// The rule (src/lib.rs:276-283) fires when IdxRecord::from_buffer returns Incomplete and the buffer
// has no free space after Buffer::clean() (src/buffer.rs:51-72).  clean() re-aligns the unconsumed
// tail so that the NEXT read is 16-byte aligned, and reads are multiples of 4096 bytes
// (src/buffer.rs:74-100), so whether a record of 69 618..69 632 bytes fits depends on where the
// previous refills happened to end.  That is pure integer arithmetic over record boundaries, which
// the GPU scan provides; no input byte is touched here.  The replay is incremental so that a
// streamed input (fqh_stream_*) can feed it chunk by chunk.
#pragma once
#include <stdint.h>

#include <algorithm>
#include <vector>

namespace fqh {

struct BufferReplay {
    static constexpr uint64_t NO_BAD = UINT64_MAX;  // no non-valid record known (yet)
    uint64_t B = 0;
    uint64_t start = 0, end = 0;  // Buffer::start / Buffer::end
    uint64_t fpos = 0;            // file offset of buffer[start]
    uint64_t rd = 0;              // file offset of buffer[end] (bytes the reader has delivered)
    uint64_t k = 0;               // global index of the record boundary at fpos
    uint64_t kbase = 0;           // global index of pend[0]
    std::vector<uint64_t> pend;   // boundaries not yet consumed, pend[0] is boundary kbase
    // sets == true replays RecordSetIter::next (src/lib.rs:364-425) instead of RecordRefIter::advance:
    // every refill goes through Buffer::replace_buffer (src/buffer.rs:30-48), which ALWAYS re-aligns,
    // and ends one RecordSet.  set_sizes receives the number of records of every set that a step
    // completes (the first one is empty: lib.rs:381-391).
    bool sets = false;
    uint64_t set_first = 0;       // global index of the first record of the set under construction
    // Short reads (pipes, decompressors): Buffer::read_into makes ONE reader.read() per refill and takes what it gets
    // (src/buffer.rs:74-100), so with a reader that returns less than it is asked for the buffer's end is no multiple of 16 any
    // more, clean() leaves the record in progress somewhere else, and whether a record of BUFSIZE - 15 .. BUFSIZE bytes fits
    // depends on the sizes of the reads.  The host notes the reads IT made (note_read); the replay then gives the reference's
    // read at stream offset p what a reader that "hands out at most c bytes per call" would give it: min(asked, c).  c is what
    // the host's own reads say about the reader at p: a read that came back SHORT (got < asked) sets c = got; a read that came
    // back full says "c >= got" — it lifts the cap if it got more than c, and says nothing otherwise (the last read of a slot
    // asks for the few bytes the slot has left).  The host's slots must be at least BUFSIZE bytes, so that its first ask of
    // a slot is at least anything the reference ever asks for.  For a reader with one cap for every call — the oracle's
    // max_read, tests/replay_fuzz.cpp — that is exactly the reference's sequence of reads; for a pipe whose reads depend on
    // timing it is the closest statement there is (the reference's own outcome depends on that timing).  No note_read at all,
    // or full reads only: the case of a file.
    struct ReadCap {
        uint64_t upto, cap;       // bytes [previous upto, upto) of the stream came in reads of `cap` bytes (UINT64_MAX: full reads)
    };
    std::vector<ReadCap> caps;
    uint64_t noted = 0;           // stream offset up to which reads have been noted
    uint64_t cap_now = UINT64_MAX; // what the reads so far say the reader hands out per call

    void reset(uint64_t bufsize, bool record_sets = false) {
        B = bufsize;
        start = end = fpos = rd = k = kbase = 0;
        pend.assign(1, 0);
        sets = record_sets;
        set_first = 0;
        caps.clear();
        noted = 0;
        cap_now = UINT64_MAX;
    }
    // One reader.read() of the host: `got` bytes of `asked` (got < asked: the reader came back short; got == 0: nothing to note).
    void note_read(uint64_t got, uint64_t asked) {
        if (!got) return;
        if (got < asked) cap_now = got;
        else if (got > cap_now) cap_now = UINT64_MAX;
        const uint64_t cap = cap_now, before = noted;
        noted += got;
        if (!caps.empty() && caps.back().cap == cap) {
            caps.back().upto = noted;
        } else if (cap != UINT64_MAX || !caps.empty()) {
            // (full reads are not listed while nothing else is: the first short read must not claim the bytes in front of it —
            // a file's LAST read comes back short, and took every earlier refill of the replay for one of its size)
            if (caps.empty() && before) caps.push_back(ReadCap{before, UINT64_MAX});
            caps.push_back(ReadCap{noted, cap});
        }
    }
    uint64_t cap_at(uint64_t pos) {
        size_t drop = 0;
        while (drop < caps.size() && caps[drop].upto <= pos) ++drop;
        if (drop) caps.erase(caps.begin(), caps.begin() + (long)drop);
        return caps.empty() ? UINT64_MAX : caps.front().cap;
    }

    // New information: boundaries rs[0..n] of the records that end in the latest chunk (rs[0] is the
    // boundary the previous chunk ended with, global index k0); bytes [0, known_end) of the file exist;
    // eof: known_end is the end of the file.  `need`: if the record starting at rs[n] is not valid,
    // the number of its bytes that must be visible to report its own error (0: it is a truncated
    // tail, reported only at EOF); NO_BAD otherwise.  Returns true (and *which) as soon as the
    // reference would report "too long" for record *which.
    bool step(const uint64_t *rs, uint64_t k0, uint64_t n, uint64_t known_end, bool eof, uint64_t need,
              uint64_t *which, std::vector<uint64_t> *set_sizes = nullptr, bool *finished = nullptr) {
        if (finished) *finished = false;
        if (B == 0) return false;
        // append the new boundaries after the pending ones
        const uint64_t have_last = kbase + pend.size() - 1;  // global index of the last pending boundary
        for (uint64_t i = 0; i <= n; ++i)
            if (k0 + i > have_last) pend.push_back(rs[i]);
        const uint64_t klast = kbase + pend.size() - 1;  // boundary index where the non-valid record starts
        bool tripped = false;
        for (;;) {
            // consume every complete valid record inside the window [fpos, rd)
            const uint64_t *b = pend.data();
            const uint64_t *hi = std::upper_bound(b + (k - kbase), b + pend.size(), rd);
            const uint64_t j = kbase + (uint64_t)(hi - b) - 1;
            if (j > k) {
                start += b[j - kbase] - fpos;
                fpos = b[j - kbase];
                k = j;
            }
            if (k == klast) {
                if (!sets && eof && need == NO_BAD && fpos == known_end && start == end) {  // clean EOF
                    if (finished) *finished = true;
                    break;
                }
                if (need != NO_BAD && need != 0 && fpos + need <= rd) {  // its own error shows
                    if (finished) *finished = true;
                    break;
                }
            }
            uint64_t nstart = start, nend = end;
            if (start == end) {  // EmptyBuffer: clean() / replace_buffer() of nothing
                nstart = nend = 0;
            } else {             // Incomplete
                const uint64_t m = end - start;
                const uint64_t new_end = (m + 15) & ~(uint64_t)15;
                const uint64_t new_start = new_end - m;
                if (sets || (start && new_start < start)) { nstart = new_start; nend = new_end; }
                if (B - nend == 0) {  // n_free() == 0 => "Fastq record is too long"
                    start = nstart; end = nend;
                    *which = k;
                    tripped = true;
                    break;
                }
            }
            const uint64_t n_free = B - nend;
            uint64_t num = n_free < 4096 ? n_free : n_free - n_free % 4096;
            num = std::min(num, cap_at(rd));   // (a reader that comes back short: note_read)
            uint64_t got;
            if (rd + num <= known_end) got = num;
            else if (eof) got = known_end - rd;
            else break;  // the reader would block: wait for the next chunk; nothing was committed
            start = nstart; end = nend;
            const bool was_empty = start == end;
            if (got == 0 && !(sets && was_empty)) {  // EOF inside a record: "truncated", caller's status stands
                if (finished) *finished = true;
                break;
            }
            if (sets) {  // this refill ends a RecordSet (lib.rs:381-415)
                if (set_sizes) set_sizes->push_back(k - set_first);
                set_first = k;
            }
            if (got == 0) {  // EmptyBuffer at EOF: reader_at_end, the last set was just yielded
                if (finished) *finished = true;
                break;
            }
            end += got;
            rd += got;
        }
        // drop boundaries behind the window
        if (k > kbase) {
            pend.erase(pend.begin(), pend.begin() + (k - kbase));
            kbase = k;
        }
        return tripped;
    }
};

// ---- The same rule in closed form, for RecordRefIter::advance (Parser::each) over a reader that fills every read.
//
// Two invariants of src/buffer.rs make the replay above unnecessary for `each` (it stays the statement of record for
// RecordSets, whose CUTS do depend on where the refills end):
//   (1) every read lands at a buffer offset that is a multiple of 16 (clean() / replace_buffer() round the kept bytes up to
//       16, src/buffer.rs:33, 57) and asks for a multiple of 16 bytes (BUFSIZE - end with both multiples of 16, or a multiple
//       of 4096, src/buffer.rs:75-80), so the file offset rd of the buffer's end is a multiple of 16 until EOF; hence a byte
//       at file offset p sits at a buffer offset == p (mod 16);
//   (2) after clean() the record in progress therefore starts at buffer offset fpos & 15 — moved there, or there already
//       (`new_start >= self.start`, src/buffer.rs:60) — and the refills that follow fill the buffer to its last byte.
// So the parser sees the record that starts at file offset fpos through the WINDOW [fpos, fpos + BUFSIZE - (fpos & 15)) and
// reports "too long" iff from_buffer (src/records.rs:201-247) is still Incomplete on all of it: a record of L bytes iff
// L > BUFSIZE - (fpos & 15); a record whose own error needs `need` bytes to show iff need > BUFSIZE - (fpos & 15); the
// incomplete record at the end of the input iff at least a window of its bytes exists.  No state, hence no history: a byte
// range of a file can be judged by itself as long as its record boundaries are TRUE file offsets (the sharded mode), and
// nothing waits for a refill that a stream which ends at a cut would never see.  tests/replay_fuzz.cpp holds it against the
// oracle's streaming restatement (and against the replay) at BUFSIZE 64 and 69632.
struct TooLong {
    static constexpr uint64_t NO_BAD = UINT64_MAX;
    static uint64_t window(uint64_t B, uint64_t fpos) { return B - (fpos & 15); }
    // rs[0..n]: boundaries (true file offsets) of n complete valid records; what follows rs[n]: `avail` bytes that are not a
    // complete valid record — an invalid one whose error needs `need` bytes (0: it is only incomplete), or NO_BAD: the bytes
    // of a record still in progress (more may follow) or nothing.  -> true and *which = index (0..n) of the first record the
    // reference calls too long.
    static bool first(uint64_t B, const uint64_t *rs, uint64_t n, uint64_t avail, uint64_t need, uint64_t *which) {
        if (!B) return false;
        for (uint64_t i = 0; i < n; ++i)
            if (rs[i + 1] - rs[i] > window(B, rs[i])) { *which = i; return true; }
        const uint64_t w = window(B, rs[n]);
        const bool t = (need == NO_BAD || need == 0) ? avail >= w : need > w;
        if (t) *which = n;
        return t;
    }
};

}  // namespace fqh
