// fastq.hpp — C++17 mirror of the `fastq` crate's public surface (aseyboldt/fastq-rs 0.6.0) on top
// of the C ABI of libfastq_hip.so.  Same names, argument meaning and error behaviour as the crate,
// so the reference's own tests read the same here (tests/host_tests.cpp):
//
//   crate (src/lib.rs, src/records.rs, src/thread_reader.rs)        here
//   --------------------------------------------------------------- -------------------------------
//   trait Record { head seq qual write validate_dna validate_dnan } fastq::RefRecord / OwnedRecord
//   RefRecord::to_owned_record, OwnedRecord                         RefRecord::to_owned_record()
//   Parser::new(reader)                                             fastq::Parser<Reader>(reader)
//   Parser::each(FnMut(RefRecord)->bool) -> io::Result<bool>        Parser::each(f) -> bool, throws Error
//   Parser::ref_iter() / RecordRefIter::{advance,get}               Parser::ref_iter() -> RecordRefIter
//   Parser::record_sets() (crate-private) / RecordSet::{iter,len}   Parser::record_sets(f) / RecordSet
//   Parser::parallel_each(n, Fn(iterator<RecordSet>) -> O)          Parser::parallel_each<O>(n, f)
//   each_zipped(p1, p2, callback)                                   fastq::each_zipped
//   parse_path(Option<path>, FnOnce(Parser))                        fastq::parse_path (plain input only)
//   thread_reader(bufsize, queuelen, reader, f)                     fastq::thread_reader
//   parallel_each over byte-range shards, one process per GPU        fastq::each_sharded (no counterpart in the crate)
//   io::Error(InvalidData, msg)                                     fastq::Error{kind(), what()}
//
// What runs where: bytes go reader -> pinned ring slot -> (hipMemcpyAsync on a side stream) -> HBM;
// record boundaries, the four newline offsets of every record and all syntax checks come from the
// HIP kernels; this header only walks the index the GPU produced and replays the reference's
// 68 KiB Buffer arithmetic (csrc/replay.h) to reproduce "record too long" and the RecordSet
// boundaries exactly.  A Reader is anything with `size_t read(uint8_t *dst, size_t n)` returning 0
// at end of input (std::io::Read); it may throw.
#pragma once
#include <dlfcn.h>
#include <errno.h>
#include <stdint.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <optional>
#include <ostream>
#include <stdexcept>
#include <string>
#include <string_view>
#include <thread>
#include <vector>

#include "../../include/fastq_hip.h"
#ifndef FASTQ_NO_ZLIB
#include <zlib.h>
#endif
#include "../csrc/replay.h"

namespace fastq {

constexpr size_t BUFSIZE = FQH_BUFSIZE;  // src/lib.rs:128-129

enum class ErrorKind { InvalidData, BrokenPipe, Other };

class Error : public std::runtime_error {
  public:
    Error(ErrorKind k, const std::string &m) : std::runtime_error(m), kind_(k) {}
    ErrorKind kind() const { return kind_; }

  private:
    ErrorKind kind_;
};

using bytes_view = std::basic_string_view<uint8_t>;

inline bytes_view trim_winline(bytes_view line) {  // src/records.rs:66-73
    if (!line.empty() && line.back() == '\r') line.remove_suffix(1);
    return line;
}
inline bool all_of_alphabet(bytes_view s, bool allow_n) {  // src/records.rs:19-33
    for (uint8_t x : s)
        if (!(x == 'A' || x == 'C' || x == 'T' || x == 'G' || (allow_n && x == 'N'))) return false;
    return true;
}

struct OwnedRecord {  // src/records.rs:47-54, 99-129
    std::string head, seq, qual;
    std::optional<std::string> sep;
    size_t write(std::ostream &w) const {
        w.put('@'); w << head; w.put('\n'); w << seq; w.put('\n');
        if (sep) w << *sep; else w.put('+');
        w.put('\n'); w << qual; w.put('\n');
        return 1 + head.size() + 1 + seq.size() + 1 + (sep ? sep->size() : 1) + 1 + qual.size() + 1;
    }
    bool validate_dna() const { return all_of_alphabet({(const uint8_t *)seq.data(), seq.size()}, false); }
    bool validate_dnan() const { return all_of_alphabet({(const uint8_t *)seq.data(), seq.size()}, true); }
};

// A record that borrows its bytes (src/records.rs:36-45): data = the raw record including the final
// '\n'; head/seq/sep/qual = offsets of the four newlines, exactly IdxRecord's fields.
class RefRecord {
  public:
    RefRecord(const uint8_t *d, size_t n, uint32_t h, uint32_t s, uint32_t p, uint32_t q)
        : data_(d), len_(n), head_(h), seq_(s), sep_(p), qual_(q) {}
    bytes_view head() const { return trim_winline({data_ + 1, (size_t)head_ - 1}); }            // :77-80
    bytes_view seq() const { return trim_winline({data_ + head_ + 1, (size_t)(seq_ - head_ - 1)}); }   // :83-85
    bytes_view qual() const { return trim_winline({data_ + sep_ + 1, (size_t)(qual_ - sep_ - 1)}); }   // :88-90
    bytes_view data() const { return {data_, len_}; }
    size_t write(std::ostream &w) const {                                                        // :93-96
        w.write((const char *)data_, (std::streamsize)len_);
        return len_;
    }
    bool validate_dna() const { return all_of_alphabet(seq(), false); }
    bool validate_dnan() const { return all_of_alphabet(seq(), true); }
    OwnedRecord to_owned_record() const {                                                        // :167-174
        auto str = [](bytes_view v) { return std::string((const char *)v.data(), v.size()); };
        OwnedRecord o;
        o.head = str(head()); o.seq = str(seq()); o.qual = str(qual());
        o.sep = str(trim_winline({data_ + seq_ + 1, (size_t)(sep_ - seq_ - 1)}));
        return o;
    }

  private:
    const uint8_t *data_;
    size_t len_;
    uint32_t head_, seq_, sep_, qual_;
};

// A batch of records (src/lib.rs:306-353).  The reference's RecordSet OWNS its 68 KiB buffer: RecordSetIter::next swaps a fresh
// buffer into the parser and moves the full one into the set (src/lib.rs:384-385, src/buffer.rs:30-48) — one move, no copy
// per record.  Here a set CO-OWNS the ring slot its records lie in (pinned host memory, `keep_`): handing a set out costs a
// reference count, and the slot goes back to the ring when the last set that borrows it is dropped — on whatever thread that
// happens (Send, like the reference's).  Only a set whose records lie in two slots (at most one per slot) is a copy.
class RecordSet {
  public:
    RecordSet() = default;
    size_t len() const { return n_; }
    bool is_empty() const { return n_ == 0; }
    RefRecord at(size_t i) const {
        const fqh_idx_record &r = idx_[i];
        return RefRecord(base_ + (int64_t)(r.start - origin_), (size_t)r.qual + 1, r.head, r.seq, r.sep, r.qual);
    }
    class iterator {
      public:
        iterator(const RecordSet *s, size_t i) : s_(s), i_(i) {}
        RefRecord operator*() const { return s_->at(i_); }
        iterator &operator++() { ++i_; return *this; }
        bool operator!=(const iterator &o) const { return i_ != o.i_; }
      private:
        const RecordSet *s_;
        size_t i_;
    };
    iterator begin() const { return iterator(this, 0); }
    iterator end() const { return iterator(this, n_); }
    const RecordSet &iter() const { return *this; }

    // records idx[0 .. n) whose bytes lie at base[start - origin ..]; `keep` keeps both arrays alive
    static RecordSet view(std::shared_ptr<const void> keep, const uint8_t *base, uint64_t origin, const fqh_idx_record *idx, size_t n) {
        RecordSet s;
        s.keep_ = std::move(keep);
        s.base_ = base;
        s.origin_ = origin;
        s.idx_ = idx;
        s.n_ = n;
        return s;
    }
    // an owned copy, for sets whose records come from more than one place
    struct Block {
        std::vector<uint8_t> buf;
        std::vector<fqh_idx_record> idx;   // start = offset into buf
        void push(const uint8_t *rec, const fqh_idx_record &r) {
            fqh_idx_record x = r;
            x.start = buf.size();
            buf.insert(buf.end(), rec, rec + (size_t)r.qual + 1);
            idx.push_back(x);
        }
    };
    static RecordSet of_block(std::shared_ptr<const Block> b) {
        const Block *p = b.get();
        return view(std::move(b), p->buf.data(), 0, p->idx.data(), p->idx.size());
    }

  private:
    std::shared_ptr<const void> keep_;
    const uint8_t *base_ = nullptr;
    uint64_t origin_ = 0;
    const fqh_idx_record *idx_ = nullptr;
    size_t n_ = 0;
};

struct Options {
    int device = 0;
    uint64_t slot_bytes = 32ull << 20;  // pinned ring slot (the GPU-side "BUFSIZE")
    uint32_t n_slots = 3;               // (record_sets / parallel_each open at least 4: their sets hold slots while workers walk them)
    uint64_t bufsize = BUFSIZE;         // the reference's BUFSIZE, for its "too long" rule (64 = cfg(fuzzing))
    // Pipes and sockets: the reference delivers records as soon as a 68 KiB refill holds one (src/lib.rs:255-303); the
    // ring waits for a whole slot.  With low_latency a slot is submitted as it is when the reader comes back with less than
    // was asked for and nothing else is on its way to the GPU (the consumer would only wait).  Results are the same;
    // a chunk then costs a GPU round trip per read() of the pipe, so it is off for files.
    bool low_latency = false;
    // parse_path on a regular file: pread()s side by side per slot (FileReader).  0 = as many as an eighth of the host's hardware
    // threads, at most 8 (one thread's copy out of the page cache feeds the GPU at the oracle's own rate);
    // 1 = the reference's one reader (src/lib.rs:186-192)
    unsigned read_threads = 0;
    // A thread of the parser's own keeps the ring's slots filled while the calling thread collects, replays and hands out
    // (thread_reader's producer, src/thread_reader.rs:131-139, one level down: its destination IS the pinned slot, no copy
    // in between).  For readers that fill every read (files): the reads are not noted for the replay of the "too long" band
    // (csrc/replay.h), and record 0 is delivered after its slot, not after its first 68 KiB.  Off: one thread does it all.
    bool read_ahead = false;
};

namespace detail {
inline const char *message(int status, bool sets) {
    // the crate words two errors differently in RecordSetIter::next (lib.rs:399-402, 407-410)
    if (sets && status == FQH_E_TOO_LONG) return "Fastq record is too long.";
    if (sets && status == FQH_E_TRUNCATED) return "Truncated input file.";
    return fqh_strerror((fqh_status)status);
}
struct Handles {
    fqh_ctx *ctx = nullptr;
    fqh_stream *st = nullptr;
    ~Handles() {
        if (st) fqh_stream_destroy(st);
        if (ctx) fqh_destroy(ctx);
    }
};
// The chunks RecordSets have borrowed: a set co-owns a Lease; the last one to go puts the chunk on the list, and the parser's
// thread — the only one that may talk to the ring — gives the slot back (fqh_stream_release_chunk).
struct Returns {
    std::mutex m;
    std::condition_variable cv;
    std::vector<fqh_chunk> back;
    size_t out = 0;  // leases alive
};
struct Lease {
    std::shared_ptr<Handles> h;  // the ring (and its context) outlive every set, whatever happens to the parser
    std::shared_ptr<Returns> ret;
    fqh_chunk c;
    ~Lease() {
        {
            std::lock_guard<std::mutex> lk(ret->m);
            ret->back.push_back(c);
            --ret->out;
        }
        ret->cv.notify_all();
    }
};
}  // namespace detail

template <class Reader>
class RecordRefIter;

template <class Reader>
class Parser {
  public:
    explicit Parser(Reader reader, Options opt = Options()) : reader_(std::move(reader)), opt_(opt) {}
    Parser(const Parser &) = delete;
    Parser &operator=(const Parser &) = delete;
    ~Parser() { stop_filler(); }

    // Parser::each (src/lib.rs:221-239): true if the input was exhausted, false if f stopped it.
    template <class F>
    bool each(F f) {
        open(false);
        for (;;) {
            Chunk c = next_chunk();
            for (uint64_t i = 0; i < c.n; ++i)
                if (!f(c.record(i))) { release(); return false; }
            const int status = c.status;
            const bool fin = c.is_final;
            release();
            if (status != FQH_OK) throw Error(ErrorKind::InvalidData, detail::message(status, false));
            if (fin) return true;
        }
    }

    RecordRefIter<Reader> ref_iter() { return RecordRefIter<Reader>(this); }

    // Parser::record_sets (src/lib.rs:428-436): f(RecordSet&&) -> bool (false stops).  Sets have the
    // same boundaries as the crate's (one per 68 KiB refill, the first one empty).  Throws on a parse
    // error; like the crate, records of the set under construction are dropped with the error.
    template <class F>
    void record_sets(F f) {
        open(true);
        // Records scanned but not yet assigned to a set (the replay of the reference's refills lags the scan by < BUFSIZE): the
        // chunks they lie in, each under a lease that the sets cut from it share.
        struct Pending {
            std::shared_ptr<const void> lease;   // the slot's lease, or an owned copy of what is left of it (below)
            const uint8_t *base;
            uint64_t origin;
            const fqh_idx_record *idx;
            uint64_t n, done;
            bool copied;
        };
        std::deque<Pending> pend;
        std::vector<uint64_t> sizes;
        for (;;) {
            Chunk c = next_chunk(&sizes);
            const int status = c.status;
            const bool fin = c.is_final;
            {
                auto lease = std::make_shared<detail::Lease>();
                lease->h = h_;
                lease->ret = returns_;
                lease->c = c.raw;
                {
                    std::lock_guard<std::mutex> lk(returns_->m);
                    ++returns_->out;
                }
                held_ = false;  // (the lease gives the slot back, not release())
                if (c.n) pend.push_back(Pending{std::move(lease), c.h_data, c.base, c.idx, c.n, 0, false});
            }
            for (uint64_t want : sizes) {  // one RecordSet per refill of the reference's buffer
                RecordSet out;
                while (!pend.empty() && pend.front().done == pend.front().n) pend.pop_front();
                if (want && !pend.empty() && pend.front().n - pend.front().done >= want) {
                    Pending &p = pend.front();   // the common case: a view into one slot
                    out = RecordSet::view(p.lease, p.base, p.origin, p.idx + p.done, (size_t)want);
                    p.done += want;
                } else if (want) {               // the set straddles two chunks: copied
                    auto blk = std::make_shared<RecordSet::Block>();
                    for (uint64_t left = want; left;) {
                        while (!pend.empty() && pend.front().done == pend.front().n) pend.pop_front();
                        if (pend.empty()) throw Error(ErrorKind::Other, "record_sets: the replay asks for records that were not scanned");
                        Pending &p = pend.front();
                        const uint64_t k = std::min<uint64_t>(left, p.n - p.done);
                        for (uint64_t i = 0; i < k; ++i) {
                            const fqh_idx_record &r = p.idx[p.done + i];
                            blk->push(p.base + (int64_t)(r.start - p.origin), r);
                        }
                        p.done += k;
                        left -= k;
                    }
                    out = RecordSet::of_block(std::move(blk));
                }
                while (!pend.empty() && pend.front().done == pend.front().n) pend.pop_front();
                if (!f(std::move(out))) return;
            }
            sizes.clear();
            if (status != FQH_OK) throw Error(ErrorKind::InvalidData, detail::message(status, true));
            if (fin) return;
            // The parser itself keeps ONE slot across the next refill (the newest: its last records wait for the replay).  What is
            // left of older chunks — less than BUFSIZE bytes of records, and only with slots smaller than that — is copied out,
            // so that the ring never waits for a lease its own consumer holds.
            for (size_t j = 0; j + 1 < pend.size(); ++j) {
                Pending &p = pend[j];
                if (p.copied) continue;
                auto blk = std::make_shared<RecordSet::Block>();
                for (uint64_t i = p.done; i < p.n; ++i) blk->push(p.base + (int64_t)(p.idx[i].start - p.origin), p.idx[i]);
                const RecordSet::Block *q = blk.get();
                p = Pending{std::move(blk), q->buf.data(), 0, q->idx.data(), (uint64_t)q->idx.size(), 0, true};
            }
        }
    }

    // Parser::parallel_each (src/lib.rs:509-565): n_threads workers, each fed RecordSets round-robin
    // through a bounded queue of 10; the worker closure gets a pull function (the crate's iterator):
    // `std::optional<RecordSet> next()`.  Returns the workers' results in worker order; on a parse
    // error the results are discarded and the error is thrown after all workers have finished.
    template <class O, class F>
    std::vector<O> parallel_each(size_t n_threads, F func) {
        // sync_channel(10) per worker (lib.rs:521-532).  A set is 68 KiB of records — microseconds of work for a light closure — so
        // the hand-off itself is what a count-only run pays for: a consumer that finds its queue empty polls it for a few dozen
        // microseconds before it sleeps (the producer's notify is then no system call: nobody waits), and a notify goes out
        // only when somebody does wait.
        struct Queue {
            std::mutex m;
            std::condition_variable cv_put, cv_get;
            std::deque<RecordSet> q;
            std::atomic<size_t> avail{0};
            std::atomic<bool> done{false};
            bool closed = false, consumer_gone = false;
            int get_waiting = 0, put_waiting = 0;
        };
        std::vector<std::unique_ptr<Queue>> qs;
        for (size_t i = 0; i < n_threads; ++i) qs.emplace_back(new Queue());
        std::vector<O> results(n_threads);
        std::vector<std::exception_ptr> panics(n_threads);
        std::vector<std::thread> threads;
        for (size_t i = 0; i < n_threads; ++i) {
            threads.emplace_back([&, i] {
                Queue &q = *qs[i];
                auto next = [&q]() -> std::optional<RecordSet> {
                    if (!q.avail.load(std::memory_order_acquire) && !q.done.load(std::memory_order_acquire)) {
                        const auto until = std::chrono::steady_clock::now() + std::chrono::microseconds(50);
                        while (!q.avail.load(std::memory_order_acquire) && !q.done.load(std::memory_order_acquire) &&
                               std::chrono::steady_clock::now() < until) {
                        }
                    }
                    std::unique_lock<std::mutex> lk(q.m);
                    ++q.get_waiting;
                    q.cv_get.wait(lk, [&] { return !q.q.empty() || q.closed; });
                    --q.get_waiting;
                    if (q.q.empty()) return std::nullopt;
                    RecordSet s = std::move(q.q.front());
                    q.q.pop_front();
                    q.avail.fetch_sub(1, std::memory_order_release);
                    if (q.put_waiting) q.cv_put.notify_one();
                    return s;
                };
                try { results[i] = func(next); } catch (...) { panics[i] = std::current_exception(); }
                std::lock_guard<std::mutex> lk(q.m);
                q.consumer_gone = true;
                q.q.clear();
                q.avail.store(0, std::memory_order_release);
                q.cv_put.notify_all();
            });
        }
        std::exception_ptr io_error;
        size_t turn = 0;
        try {
            if (n_threads) {
                record_sets([&](RecordSet &&s) {
                    Queue &q = *qs[turn % n_threads];  // senders.iter().cycle(), lib.rs:535
                    ++turn;
                    std::unique_lock<std::mutex> lk(q.m);
                    ++q.put_waiting;
                    q.cv_put.wait(lk, [&] { return q.q.size() < 10 || q.consumer_gone; });  // sync_channel(10)
                    --q.put_waiting;
                    if (q.consumer_gone) return false;  // send error: stop parsing (lib.rs:538-543)
                    q.q.push_back(std::move(s));
                    q.avail.fetch_add(1, std::memory_order_release);
                    if (q.get_waiting) q.cv_get.notify_one();
                    return true;
                });
            }
        } catch (...) {
            io_error = std::current_exception();
        }
        for (auto &q : qs) {  // drop(senders)
            std::lock_guard<std::mutex> lk(q->m);
            q->closed = true;
            q->done.store(true, std::memory_order_release);
            q->cv_get.notify_all();
        }
        for (auto &t : threads) t.join();
        for (auto &p : panics)
            if (p) std::rethrow_exception(p);  // "Panic in worker thread"
        if (io_error) std::rethrow_exception(io_error);
        return results;
    }

  private:
    friend class RecordRefIter<Reader>;
    struct Chunk {
        uint64_t n = 0;
        int status = FQH_OK;
        bool is_final = false;
        const uint8_t *h_data = nullptr;
        uint64_t base = 0;
        const fqh_idx_record *idx = nullptr;
        fqh_chunk raw = {};
        const uint8_t *record_ptr(uint64_t i) const { return h_data + (int64_t)(idx[i].start - base); }
        RefRecord record(uint64_t i) const {
            const fqh_idx_record &r = idx[i];
            return RefRecord(record_ptr(i), (size_t)r.qual + 1, r.head, r.seq, r.sep, r.qual);
        }
    };

    void open(bool sets) {
        if (h_->ctx) throw Error(ErrorKind::Other, "parser already consumed");
        if (fqh_create(opt_.device, &h_->ctx) != FQH_OK)
            throw Error(ErrorKind::Other, std::string("fqh_create: ") + fqh_last_error(nullptr));
        fqh_set_bufsize(h_->ctx, 0);  // the "too long" rule is replayed here, mode-exact (each vs record sets)
        // (sets borrow slots: one being filled, one being scanned, one or two under the workers' hands)
        if (fqh_stream_create(h_->ctx, opt_.slot_bytes, sets ? std::max<uint32_t>(opt_.n_slots, 4) : opt_.n_slots, FQH_STREAM_INDEX, &h_->st) != FQH_OK)
            throw Error(ErrorKind::Other, std::string("fqh_stream_create: ") + fqh_last_error(h_->ctx));
        replay_.reset(opt_.bufsize, sets);
        sets_ = sets;
        if (opt_.read_ahead) {
            ahead_ = true;
            startup_ = false;
            filler_ = std::thread([this] { fill_ahead(); });
        }
    }

    // ---- Options::read_ahead: the filler thread and the calling thread share the ring under ring_mu_ (the ring itself is
    // single-threaded); the read() into an acquired slot runs outside the lock
    struct RingLock {
        std::unique_lock<std::mutex> lk;
        explicit RingLock(Parser *p) : lk(p->ring_mu_, std::defer_lock) { if (p->ahead_) lk.lock(); }
    };
    void fill_ahead() {
        try {
            bool eof = false;
            while (!eof) {
                uint8_t *dst = nullptr;
                uint64_t cap = 0;
                {
                    std::unique_lock<std::mutex> lk(ring_mu_);
                    for (;;) {
                        if (filler_stop_) return;
                        const fqh_status st = fqh_stream_acquire(h_->st, &dst, &cap);
                        if (st == FQH_OK) break;
                        if (st != FQH_E_CAPACITY) throw Error(ErrorKind::Other, fqh_last_error(h_->ctx));
                        ring_cv_.wait(lk);   // (a release, or a lease that came back, frees a slot)
                    }
                }
                uint64_t n = 0;
                while (n < cap) {
                    const size_t got = reader_.read(dst + n, (size_t)(cap - n));
                    if (got == 0) { eof = true; break; }
                    n += got;
                }
                {
                    std::lock_guard<std::mutex> lk(ring_mu_);
                    if (fqh_stream_submit(h_->st, n, eof ? 1 : 0) != FQH_OK) throw Error(ErrorKind::Other, fqh_last_error(h_->ctx));
                    ++in_flight_;
                }
                { std::lock_guard<std::mutex> lk(returns_->m); }
                returns_->cv.notify_all();   // (the calling thread waits there for chunks and for leases alike)
            }
        } catch (...) {
            std::lock_guard<std::mutex> lk(ring_mu_);
            filler_err_ = std::current_exception();
        }
        { std::lock_guard<std::mutex> lk(returns_->m); }
        returns_->cv.notify_all();
    }
    void stop_filler() {
        if (!filler_.joinable()) return;
        {
            std::lock_guard<std::mutex> lk(ring_mu_);
            filler_stop_ = true;
        }
        ring_cv_.notify_all();
        filler_.join();
    }
    // the calling thread, until the filler has submitted something to collect
    void wait_ahead() {
        for (;;) {
            {
                std::lock_guard<std::mutex> lk(ring_mu_);
                if (in_flight_ > 0) return;
                if (filler_err_) std::rethrow_exception(filler_err_);
            }
            take_back(false);   // (slots whose sets are gone: the filler may be waiting for one)
            std::unique_lock<std::mutex> lk(returns_->m);
            if (returns_->back.empty()) returns_->cv.wait_for(lk, std::chrono::microseconds(200));
        }
    }

    // Slots whose last RecordSet is gone go back to the ring (this thread is the only one that talks to it).  wait: block until
    // at least one comes back (false: none is out — waiting would be for ever).
    bool take_back(bool wait) {
        std::vector<fqh_chunk> back;
        {
            std::unique_lock<std::mutex> lk(returns_->m);
            if (wait) {
                if (returns_->back.empty() && returns_->out == 0) return false;
                returns_->cv.wait(lk, [&] { return !returns_->back.empty(); });
            }
            back.swap(returns_->back);
        }
        {
            RingLock rl(this);
            for (const fqh_chunk &c : back)
                if (fqh_stream_release_chunk(h_->st, &c) != FQH_OK) throw Error(ErrorKind::Other, "fqh_stream_release_chunk");
        }
        if (ahead_ && !back.empty()) ring_cv_.notify_all();
        return true;
    }

    void fill() {  // keep the ring busy: read -> pinned slot -> async H2D
        while (!eof_) {
            uint8_t *dst;
            uint64_t cap;
            take_back(false);
            fqh_status st = fqh_stream_acquire(h_->st, &dst, &cap);
            if (st == FQH_E_CAPACITY && in_flight_ == 0 && !held_) {
                // nothing to collect and no slot to fill: every slot is under some RecordSet's lease — wait for one
                if (!take_back(true)) throw Error(ErrorKind::Other, "fqh_stream_acquire: the ring is full and nothing is held");
                continue;
            }
            if (st == FQH_E_CAPACITY) return;
            if (st != FQH_OK) throw Error(ErrorKind::Other, fqh_last_error(h_->ctx));
            // Until the stream has delivered its FIRST record, a chunk ends as soon as the reads so far may hold it — four
            // newlines have come in — or after BUFSIZE bytes, is submitted at once and collected before anything else is read:
            // record 0 reaches the caller after no more input than the reference's refills take to hold it (src/lib.rs:264-275;
            // a producer on a pipe may be waiting for an answer to it).  Every later chunk fills its slot.
            // (a chunk without a record — a first record longer than BUFSIZE with bufsize 0, blank lines in front — must not
            // leave every later chunk of the start-up one read() long: the count starts over per chunk and the chunks double)
            const bool first = startup_;
            if (first) {
                startup_newlines_ = 0;
                startup_target_ = startup_target_ ? std::min<uint64_t>(cap, 2 * startup_target_) : std::min<uint64_t>(cap, opt_.bufsize ? opt_.bufsize : BUFSIZE);
            }
            const uint64_t target = first ? startup_target_ : cap;
            uint64_t n = 0;
            while (n < target) {
                const size_t want = (size_t)(target - n);
                size_t got = reader_.read(dst + n, want);
                if (got == 0) { eof_ = true; break; }
                replay_.note_read(got, want);   // (a reader that comes back short decides the "too long" band: csrc/replay.h)
                if (first) startup_newlines_ += (uint64_t)std::count(dst + n, dst + n + got, (uint8_t)'\n');
                n += got;
                if (first && startup_newlines_ >= 4) break;
                if (opt_.low_latency && got < want && in_flight_ == 0) break;  // what there is, now
            }
            if (fqh_stream_submit(h_->st, n, eof_ ? 1 : 0) != FQH_OK) throw Error(ErrorKind::Other, fqh_last_error(h_->ctx));
            ++in_flight_;
            if (first) return;
        }
    }

    Chunk next_chunk(std::vector<uint64_t> *set_sizes = nullptr) {
        if (ahead_) wait_ahead(); else fill();
        fqh_chunk c;
        for (;;) {
            fqh_status cs;
            {
                RingLock rl(this);
                cs = fqh_stream_collect(h_->st, &c);
                if (cs == FQH_OK) --in_flight_;
            }
            if (cs == FQH_OK) break;
            // (the slot behind the one to collect is still under a RecordSet's lease: the ring cannot put the record in progress in front of it)
            if (cs != FQH_E_AGAIN || !take_back(true)) throw Error(ErrorKind::Other, fqh_last_error(h_->ctx));
        }
        held_ = true;
        if (c.n_records) startup_ = false;
        Chunk out;
        out.n = c.n_records;
        out.status = c.parse_status;
        out.is_final = c.is_final != 0;
        out.h_data = c.h_data;
        out.base = c.base_offset;
        out.idx = c.h_index;
        out.raw = c;
        // the reference's Buffer, replayed over the boundaries: "too long" and (sets) the set cuts
        uint64_t which = 0;
        bool finished = false;
        const bool bad = c.parse_status != FQH_OK;
        const uint64_t known_end = c.base_offset + c.data_len;
        if (replay_.step(c.h_rec_start, records_done_, c.n_records, known_end, out.is_final || bad,
                         bad ? c.err_need : fqh::BufferReplay::NO_BAD, &which, set_sizes, &finished)) {
            out.status = FQH_E_TOO_LONG;
            out.n = which >= records_done_ ? which - records_done_ : 0;
            out.is_final = true;
        }
        records_done_ += c.n_records;
        return out;
    }

    void release() {
        if (held_) {
            {
                RingLock rl(this);
                fqh_stream_release(h_->st);
            }
            if (ahead_) ring_cv_.notify_all();
        }
        held_ = false;
    }

    Reader reader_;
    Options opt_;
    std::shared_ptr<detail::Handles> h_ = std::make_shared<detail::Handles>();
    std::shared_ptr<detail::Returns> returns_ = std::make_shared<detail::Returns>();
    fqh::BufferReplay replay_;
    bool sets_ = false, eof_ = false, held_ = false, startup_ = true;
    // Options::read_ahead
    bool ahead_ = false, filler_stop_ = false;
    std::thread filler_;
    std::mutex ring_mu_;
    std::condition_variable ring_cv_;
    std::exception_ptr filler_err_;
    uint64_t records_done_ = 0, startup_newlines_ = 0, startup_target_ = 0;
    int in_flight_ = 0;
};

// RecordRefIter (src/lib.rs:241-304): advance() then get(); get() is empty at end of input.
template <class Reader>
class RecordRefIter {
  public:
    explicit RecordRefIter(Parser<Reader> *p) : p_(p) {}
    void advance() {
        if (!opened_) { p_->open(false); opened_ = true; }
        ++i_;
        while (!have_ || i_ >= c_.n) {
            if (have_) {
                const int status = c_.status;
                const bool fin = c_.is_final;
                p_->release();
                have_ = false;
                if (status != FQH_OK) throw Error(ErrorKind::InvalidData, detail::message(status, false));
                if (fin) { done_ = true; return; }
            }
            if (done_) return;
            c_ = p_->next_chunk();
            have_ = true;
            i_ = 0;
        }
    }
    std::optional<RefRecord> get() const {
        if (done_ || !have_ || i_ >= c_.n) return std::nullopt;
        return c_.record(i_);
    }

  private:
    Parser<Reader> *p_;
    typename Parser<Reader>::Chunk c_;
    bool opened_ = false, have_ = false, done_ = false;
    uint64_t i_ = (uint64_t)-1;
};

// each_sharded — the byte-range sharded, host-streamed mode (BASELINE configs[4]): one process per GPU, every rank calls this
// with the same arguments but its own rank.  The analogue of Parser::parallel_each with a histogram closure
// (src/lib.rs:509-565): the ranks' results are gathered at the end (src/lib.rs:553-559) and a parse error — the FIRST one in
// file order, kind and record exactly as Parser::each meets it — is what every rank throws (src/lib.rs:544-547, 561-564).
// read_at(dst, file_offset, n) fills host memory (a pread, a memcpy).  d_hist: device array of 1 + 8 + lmax * 264 u64, this
// rank's own, zeroed: [records | the 8 scalars of fqh_stats | quality histogram lmax x 256 | base histogram lmax x 8]; on return
// it holds the sums over the ranks.  Returns the number of records of the whole file; *err (if given) receives status, records
// delivered before the error and the failing record's file offset instead of a throw.  comm may be NULL when n_ranks == 1.  The
// driver itself is the library's (fqh_shard_stream_run / fqh_shard_stream_finish / fqh_shard_stream_outcome); this is the exchange
// around it: one all-gather, the SUMs, one MIN.  A rank whose part fails (the callback throws, a device error) still takes part
// in every collective — the others would wait for it forever — and every rank learns of the failure from the MIN.
struct ShardedOutcome {
    int32_t status = FQH_OK;
    uint64_t n_records = 0, err_offset = 0;
};
template <class ReadAt>
uint64_t each_sharded(fqh_ctx *ctx, fqh_comm *comm, int n_ranks, int rank, ReadAt read_at, uint64_t file_len, uint32_t lmax,
                      uint64_t *d_hist, Options opt = Options(), ShardedOutcome *err = nullptr) {
    auto chk = [&](fqh_status st, const char *what) {
        if (st != FQH_OK) throw Error(ErrorKind::Other, std::string(what) + ": " + fqh_last_error(ctx));
    };
    if (n_ranks < 1 || n_ranks > FQH_SHARD_MAX_RANKS || rank < 0 || rank >= n_ranks || (n_ranks > 1 && !comm))
        throw Error(ErrorKind::Other, "each_sharded: bad rank / communicator");
    // byte ranges of equal size (any cut is fine: inside a line, a file of three lines on eight ranks)
    const uint64_t lo = file_len / (uint64_t)n_ranks * (uint64_t)rank;
    const uint64_t hi = rank + 1 == n_ranks ? file_len : file_len / (uint64_t)n_ranks * (uint64_t)(rank + 1);
    struct Cb {
        ReadAt *f;
        static int call(void *user, uint8_t *dst, uint64_t off, uint64_t n) {
            try {
                (*static_cast<Cb *>(user)->f)(dst, off, n);
                return 0;
            } catch (...) {
                return 1;  // (an exception must not cross the C frames: reported as FQH_E_IO)
            }
        }
    } cb{&read_at};
    constexpr uint64_t ROW = FQH_SHARD_STREAM_WORDS;
    std::vector<uint64_t> mine(ROW, 0), words((size_t)n_ranks * ROW, 0);
    uint64_t *d_sc = d_hist + 1, *d_q = d_hist + 9, *d_b = d_hist + 9 + (uint64_t)lmax * 256;
    fqh_shard_result res;
    std::string local_failure;
    const fqh_status run_st = fqh_shard_stream_run(ctx, &Cb::call, &cb, lo, hi, file_len, opt.slot_bytes, opt.n_slots, lmax, d_q, d_b, d_sc, &res);
    if (run_st == FQH_OK) {
        fqh_shard_result_words(&res, lo, hi, mine.data());
    } else {  // this rank failed: the others must not wait for it in the exchange
        local_failure = std::string("fqh_shard_stream_run: ") + fqh_last_error(ctx);
        fqh_shard_failed_words(run_st, lo, hi, mine.data());
    }
    // ---- the one exchange: FQH_SHARD_STREAM_WORDS words of every rank; device scratch: [mine | all | records per rank | key]
    void *d_x = nullptr;
    const uint64_t n_x = ROW + (uint64_t)n_ranks * ROW + (uint64_t)n_ranks + 1;
    chk(fqh_dev_alloc(ctx, n_x * 8, &d_x), "fqh_dev_alloc");
    struct Free {
        fqh_ctx *c;
        void *p;
        ~Free() { (void)fqh_dev_free(c, p); }
    } free_x{ctx, d_x};
    uint64_t *d_mine = static_cast<uint64_t *>(d_x), *d_all = d_mine + ROW, *d_slots = d_all + (uint64_t)n_ranks * ROW, *d_key = d_slots + n_ranks;
    if (n_ranks > 1) {
        fqh_status st = fqh_memcpy_h2d(ctx, d_mine, mine.data(), ROW * 8);
        if (st == FQH_OK) st = fqh_allgather(ctx, comm, d_mine, d_all, ROW * 8);
        if (st == FQH_OK) st = fqh_memcpy_d2h(ctx, words.data(), d_all, (uint64_t)n_ranks * ROW * 8);
        chk(st, "each_sharded: all-gather");
    } else {
        words = mine;
    }
    // ---- true-phase check, the gap in front of this rank, its first-error key; then the SUMs and the MIN over the ranks
    uint64_t out[2] = {0, FQH_NO_ERROR_KEY};
    const fqh_status fin_st = fqh_shard_stream_finish(ctx, &Cb::call, &cb, file_len, words.data(), n_ranks, rank, opt.slot_bytes, opt.n_slots,
                                                      lmax, d_q, d_b, d_sc, out);
    if (fin_st != FQH_OK) {
        local_failure = std::string("fqh_shard_stream_finish: ") + fqh_last_error(ctx);
        out[0] = 0;
        out[1] = fqh_shard_failure_key(rank, lo, fin_st);
    }
    std::vector<uint64_t> slots((size_t)n_ranks + 1, 0);
    slots[(size_t)rank] = out[0];
    slots[(size_t)n_ranks] = out[1];
    fqh_status st = fqh_memcpy_h2d(ctx, d_slots, slots.data(), ((uint64_t)n_ranks + 1) * 8);
    if (st == FQH_OK && n_ranks > 1) st = fqh_allreduce_u64(ctx, comm, d_slots, (uint64_t)n_ranks);
    if (st == FQH_OK && n_ranks > 1) st = fqh_allreduce_u64(ctx, comm, d_hist + 1, 8 + (uint64_t)lmax * 264);
    if (st == FQH_OK && n_ranks > 1) st = fqh_allreduce_min_u64(ctx, comm, d_key, 1);
    if (st == FQH_OK) st = fqh_memcpy_d2h(ctx, slots.data(), d_slots, ((uint64_t)n_ranks + 1) * 8);
    chk(st, "each_sharded: all-reduce");
    ShardedOutcome o;
    chk(fqh_shard_stream_outcome(slots[(size_t)n_ranks], slots.data(), n_ranks, &o.status, &o.n_records, &o.err_offset), "fqh_shard_stream_outcome");
    chk(fqh_memcpy_h2d(ctx, d_hist, &o.n_records, 8), "each_sharded: record count");
    if (err) {
        *err = o;
        return o.n_records;
    }
    if (o.status != FQH_OK && o.status <= FQH_E_TOO_LONG) throw Error(ErrorKind::InvalidData, detail::message(o.status, false));
    if (o.status != FQH_OK)
        throw Error(ErrorKind::Other, local_failure.empty() ? std::string("each_sharded: a rank failed: ") + fqh_strerror((fqh_status)o.status) : local_failure);
    return o.n_records;
}

// each_zipped (src/lib.rs:577-609)
template <class R1, class R2, class F>
std::pair<bool, bool> each_zipped(Parser<R1> &p1, Parser<R2> &p2, F callback) {
    auto it1 = p1.ref_iter();
    auto it2 = p2.ref_iter();
    std::pair<bool, bool> finished{false, false};
    it1.advance();
    it2.advance();
    for (;;) {
        auto v1 = finished.first ? std::nullopt : it1.get();
        auto v2 = finished.second ? std::nullopt : it2.get();
        finished = {!v1.has_value(), !v2.has_value()};
        std::pair<bool, bool> adv = callback(v1, v2);
        if ((!adv.first && !adv.second) || (finished.first && finished.second)) return finished;
        if (adv.first && !finished.first) it1.advance();
        if (adv.second && !finished.second) it2.advance();
    }
}

// Readers ---------------------------------------------------------------------------------------
class MemReader {  // std::io::Cursor<&[u8]>
  public:
    MemReader(const uint8_t *d, size_t n) : d_(d), n_(n) {}
    explicit MemReader(const std::string &s) : d_((const uint8_t *)s.data()), n_(s.size()) {}
    size_t read(uint8_t *dst, size_t n) {
        size_t k = n < n_ - pos_ ? n : n_ - pos_;
        if (k) memcpy(dst, d_ + pos_, k);
        pos_ += k;
        return k;
    }
  private:
    const uint8_t *d_;
    size_t n_, pos_ = 0;
};

class FileReader {  // std::fs::File / stdin
  public:
    // read_threads > 1: reads of 8 MiB and more from a REGULAR file are split into that many pread()s side by side (one thread's
    // copy out of the page cache runs at a fraction of what the link to the GPU takes; the reference reads on one thread
    // because its parser is the slower part, src/lib.rs:186-192).  Pipes, stdin and small reads: one read() per call.
    explicit FileReader(FILE *f, bool own, unsigned read_threads = 1) : f_(f), own_(own), par_(read_threads ? read_threads : 1) {
        struct stat sb;
        const off_t at = ftello(f);
        if (at >= 0 && fstat(fileno(f), &sb) == 0 && S_ISREG(sb.st_mode)) {
            regular_ = true;   // (from here on the FILE's own buffer is bypassed: pread at a position kept here)
            pos_ = (uint64_t)at;
        }
    }
    FileReader(FileReader &&o) noexcept : f_(o.f_), own_(o.own_), par_(o.par_), regular_(o.regular_), pos_(o.pos_) { o.f_ = nullptr; }
    FileReader(const FileReader &) = delete;
    ~FileReader() { if (f_ && own_) fclose(f_); }
    size_t read(uint8_t *dst, size_t n) {
        if (!regular_) {
            size_t k = fread(dst, 1, n, f_);
            if (k == 0 && ferror(f_)) throw Error(ErrorKind::Other, "read error");
            return k;
        }
        const int fd = fileno(f_);
        auto pread_all = [fd](uint8_t *d, size_t want, uint64_t at) -> ssize_t {  // up to `want` bytes, less only at the end of the file
            size_t got = 0;
            while (got < want) {
                const ssize_t k = ::pread(fd, d + got, want - got, (off_t)(at + got));
                if (k < 0) {
                    if (errno == EINTR) continue;  // src/buffer.rs:85-97
                    return -1;
                }
                if (k == 0) break;
                got += (size_t)k;
            }
            return (ssize_t)got;
        };
        size_t got = 0;
        if (par_ > 1 && n >= (8u << 20)) {
            const size_t piece = ((n + par_ - 1) / par_ + 4095) & ~(size_t)4095;
            std::vector<ssize_t> res((n + piece - 1) / piece, 0);
            std::vector<std::thread> th;
            for (size_t j = 1; j < res.size(); ++j)
                th.emplace_back([&, j] { res[j] = pread_all(dst + j * piece, std::min(piece, n - j * piece), pos_ + j * piece); });
            res[0] = pread_all(dst, std::min(piece, n), pos_);
            for (auto &t : th) t.join();
            for (size_t j = 0; j < res.size(); ++j) {
                if (res[j] < 0) throw Error(ErrorKind::Other, "read error");
                got += (size_t)res[j];
                if ((size_t)res[j] < std::min(piece, n - j * piece)) break;  // the end of the file lies in this piece
            }
        } else {
            const ssize_t k = pread_all(dst, n, pos_);
            if (k < 0) throw Error(ErrorKind::Other, "read error");
            got = (size_t)k;
        }
        pos_ += got;
        return got;
    }
  private:
    FILE *f_;
    bool own_;
    unsigned par_ = 1;
    bool regular_ = false;
    uint64_t pos_ = 0;
};

// thread_reader (src/thread_reader.rs:182-200): `queuelen` recycled buffers of `bufsize` bytes are
// filled by a background thread; the consumer sees a Reader.  A reader's error reaches the consumer's read() as it
// is (the reference forwards the io::Result of every read in its BufferMessage).
template <class Reader>
class ThreadReader {
  public:
    ThreadReader(Reader r, size_t bufsize, size_t queuelen) : reader_(std::move(r)), bufsize_(bufsize) {
        for (size_t i = 0; i < (queuelen ? queuelen : 1); ++i) empty_.push_back(std::vector<uint8_t>(bufsize));
        th_ = std::thread([this] { serve(); });
    }
    ~ThreadReader() {
        { std::lock_guard<std::mutex> lk(m_); stop_ = true; }
        cv_.notify_all();
        if (th_.joinable()) th_.join();
    }
    size_t read(uint8_t *dst, size_t n) {
        if (cur_pos_ == cur_len_) {
            std::unique_lock<std::mutex> lk(m_);
            if (have_cur_) { empty_.push_back(std::move(cur_)); have_cur_ = false; cv_.notify_all(); }
            cv_.wait(lk, [&] { return !full_.empty() || failed_; });
            if (full_.empty()) {
                // the reader's own error, as the BufferMessage carries the io::Result of its read()
                // (src/thread_reader.rs:134-137); BrokenPipe only if the thread went away without one
                if (err_) std::rethrow_exception(err_);
                throw Error(ErrorKind::BrokenPipe, "reader thread died");
            }
            auto m = std::move(full_.front());
            full_.pop_front();
            cur_ = std::move(m.first);
            cur_len_ = m.second;
            cur_pos_ = 0;
            have_cur_ = true;
            if (cur_len_ == 0) return 0;  // a read() of 0 bytes is forwarded (thread_reader.rs:40-50)
        }
        size_t k = n < cur_len_ - cur_pos_ ? n : cur_len_ - cur_pos_;
        memcpy(dst, cur_.data() + cur_pos_, k);
        cur_pos_ += k;
        return k;
    }
  private:
    void serve() {
        for (;;) {
            std::vector<uint8_t> b;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return !empty_.empty() || stop_; });
                if (stop_) return;
                b = std::move(empty_.front());
                empty_.pop_front();
            }
            size_t got = 0;
            try { got = reader_.read(b.data(), bufsize_); } catch (...) {
                std::lock_guard<std::mutex> lk(m_);
                err_ = std::current_exception();
                failed_ = true;
                cv_.notify_all();
                return;
            }
            std::lock_guard<std::mutex> lk(m_);
            full_.emplace_back(std::move(b), got);
            cv_.notify_all();
            if (got == 0) { /* keep serving: the consumer may poll EOF again */ }
        }
    }
    Reader reader_;
    size_t bufsize_;
    std::mutex m_;
    std::condition_variable cv_;
    std::deque<std::vector<uint8_t>> empty_;
    std::deque<std::pair<std::vector<uint8_t>, size_t>> full_;
    std::vector<uint8_t> cur_;
    size_t cur_len_ = 0, cur_pos_ = 0;
    bool have_cur_ = false, stop_ = false, failed_ = false;
    std::exception_ptr err_;
    std::thread th_;
};

template <class Reader, class F>
auto thread_reader(size_t bufsize, size_t queuelen, Reader reader, F func) {
    ThreadReader<Reader> tr(std::move(reader), bufsize, queuelen);
    struct Ref {  // the closure gets a `&mut ThreadReader`
        ThreadReader<Reader> *t;
        size_t read(uint8_t *d, size_t n) { return t->read(d, n); }
    };
    return func(Ref{&tr});
}

// `&mut dyn Read`: what parse_path hands to Parser::new (src/lib.rs:170, 186-192).
class DynReader {
  public:
    template <class R>
    explicit DynReader(R *r) : read_([r](uint8_t *d, size_t n) { return r->read(d, n); }) {}
    size_t read(uint8_t *dst, size_t n) { return read_(dst, n); }
  private:
    std::function<size_t(uint8_t *, size_t)> read_;
};

// Compression sniffing.  The reference delegates it to the niffler crate (src/lib.rs:173-185, dependency
// `niffler >= 2.4.0`, unpinned and not vendored => parity unpinned here): niffler reads the first five
// bytes, fails with FileTooShort when there are fewer, matches the magic numbers below and then replays
// those bytes in front of the stream.
enum class Compression { No, Gzip, Bzip2, Lzma, Zstd, Lz4 };
inline Compression sniff_compression(const uint8_t *b5) {
    if (b5[0] == 0x1f && b5[1] == 0x8b) return Compression::Gzip;
    if (b5[0] == 0x42 && b5[1] == 0x5a) return Compression::Bzip2;
    if (b5[0] == 0xfd && b5[1] == 0x37 && b5[2] == 0x7a && b5[3] == 0x58 && b5[4] == 0x5a) return Compression::Lzma;
    if (b5[0] == 0x28 && b5[1] == 0xb5 && b5[2] == 0x2f && b5[3] == 0xfd) return Compression::Zstd;
    // (the crate's documentation names lz4 next to gzip, src/lib.rs:137-141 and README.md:39-45: its numbers were taken
    // with an lz4 file; the LZ4 frame magic, little endian 0x184D2204)
    if (b5[0] == 0x04 && b5[1] == 0x22 && b5[2] == 0x4d && b5[3] == 0x18) return Compression::Lz4;
    return Compression::No;
}

template <class Reader>
class PrefixReader {  // io::Cursor(first_bytes).chain(stream)
  public:
    PrefixReader(Reader r, const uint8_t *pre, size_t n) : r_(std::move(r)), pre_(pre, pre + n) {}
    size_t read(uint8_t *dst, size_t n) {
        if (pos_ < pre_.size()) {
            size_t k = n < pre_.size() - pos_ ? n : pre_.size() - pos_;
            memcpy(dst, pre_.data() + pos_, k);
            pos_ += k;
            return k;
        }
        return r_.read(dst, n);
    }
  private:
    Reader r_;
    std::vector<uint8_t> pre_;
    size_t pos_ = 0;
};

#ifndef FASTQ_NO_ZLIB
// Multi-member gzip decoder on the host (flate2::read::MultiGzDecoder in niffler); it runs on the
// thread_reader thread and feeds the pinned ring of the scanner like any other Reader.
template <class Reader>
class GzReader {
  public:
    explicit GzReader(Reader r) : r_(std::move(r)), in_(1 << 16) {
        memset(&z_, 0, sizeof z_);
        if (inflateInit2(&z_, 15 + 16) != Z_OK) throw Error(ErrorKind::Other, "zlib init failed");
        init_ = true;
    }
    GzReader(GzReader &&o) noexcept : r_(std::move(o.r_)), in_(std::move(o.in_)) {
        // a z_stream holds a pointer to itself (state->strm): re-create instead of copying
        if (o.started_) std::terminate();
        memset(&z_, 0, sizeof z_);
        if (inflateInit2(&z_, 15 + 16) != Z_OK) std::terminate();
        init_ = true;
    }
    GzReader(const GzReader &) = delete;
    ~GzReader() { if (init_) inflateEnd(&z_); }
    size_t read(uint8_t *dst, size_t n) {
        started_ = true;
        if (n == 0) return 0;
        z_.next_out = dst;
        z_.avail_out = (uInt)(n < 0x40000000u ? n : 0x40000000u);
        while (!done_ && z_.next_out == dst) {  // until some output exists or the last member ended
            if (z_.avail_in == 0 && !in_eof_) {
                const size_t got = r_.read(in_.data(), in_.size());
                if (got == 0) in_eof_ = true;
                z_.next_in = in_.data();
                z_.avail_in = (uInt)got;
            }
            if (member_end_) {  // between members: more input => next member, none => EOF
                if (z_.avail_in == 0) { done_ = true; break; }
                if (inflateReset(&z_) != Z_OK) throw Error(ErrorKind::InvalidData, "corrupt gzip stream");
                member_end_ = false;
            }
            if (z_.avail_in == 0) throw Error(ErrorKind::InvalidData, "unexpected end of gzip stream");
            const int rc = inflate(&z_, Z_NO_FLUSH);
            if (rc == Z_STREAM_END) member_end_ = true;
            else if (rc != Z_OK && rc != Z_BUF_ERROR) throw Error(ErrorKind::InvalidData, "corrupt gzip stream");
        }
        return (size_t)(z_.next_out - dst);
    }
  private:
    Reader r_;
    std::vector<uint8_t> in_;
    z_stream z_;
    bool init_ = false, started_ = false, in_eof_ = false, member_end_ = false, done_ = false;
};
#endif


// bzip2, xz and zstd — the other formats niffler sniffs for parse_path (src/lib.rs:173-190).  The image ships the
// run-time libraries (libbz2.so.1, liblzma.so.5, libzstd.so.1) but no headers: the three streaming decoders are bound
// with dlopen, through the few stable declarations of their public C interfaces restated here.  A box without the
// library gets niffler's "no decoder" error for that format.  Concatenated streams / frames are decoded back to back,
// as bzip2 -d, xz -d and zstd -d do.  Runs on the thread_reader thread.
namespace codec {
inline void *open_lib(const char *a, const char *b) {
    void *h = dlopen(a, RTLD_NOW | RTLD_LOCAL);
    if (!h && b) h = dlopen(b, RTLD_NOW | RTLD_LOCAL);
    return h;
}
template <class T>
inline bool sym(void *h, const char *name, T &fn) {
    fn = reinterpret_cast<T>(dlsym(h, name));
    return fn != nullptr;
}

struct Bz2 {  // bzlib.h 1.0: bz_stream, BZ2_bzDecompress{Init,,End}; BZ_OK 0, BZ_STREAM_END 4
    struct Stream {
        char *next_in; unsigned avail_in, total_in_lo32, total_in_hi32;
        char *next_out; unsigned avail_out, total_out_lo32, total_out_hi32;
        void *state; void *(*bzalloc)(void *, int, int); void (*bzfree)(void *, void *); void *opaque;
    };
    int (*init)(Stream *, int, int) = nullptr;
    int (*run)(Stream *) = nullptr;
    int (*end)(Stream *) = nullptr;
    Stream z{};
    bool live = false;
    static const char *name() { return "bzip2"; }
    bool load() {
        void *h = open_lib("libbz2.so.1", "libbz2.so.1.0");
        // (the struct above is bzlib 1.0's: a library that does not say "1.0.x" is not used)
        const char *(*ver)() = nullptr;
        if (!h || !sym(h, "BZ2_bzlibVersion", ver) || strncmp(ver(), "1.0.", 4) != 0) return false;
        return sym(h, "BZ2_bzDecompressInit", init) && sym(h, "BZ2_bzDecompress", run) && sym(h, "BZ2_bzDecompressEnd", end);
    }
    bool start() { memset(&z, 0, sizeof z); live = init(&z, 0, 0) == 0; return live; }
    void stop() { if (live) end(&z); live = false; }
    // -> 0 progress, 1 stream ended, -1 corrupt
    int step(const uint8_t *&in, size_t &n_in, uint8_t *&out, size_t &n_out) {
        z.next_in = (char *)in; z.avail_in = (unsigned)(n_in < 0x40000000u ? n_in : 0x40000000u);
        z.next_out = (char *)out; z.avail_out = (unsigned)(n_out < 0x40000000u ? n_out : 0x40000000u);
        const unsigned ai = z.avail_in, ao = z.avail_out;
        const int rc = run(&z);
        in += ai - z.avail_in; n_in -= ai - z.avail_in;
        out += ao - z.avail_out; n_out -= ao - z.avail_out;
        return rc == 4 ? 1 : rc == 0 ? 0 : -1;
    }
};

struct Xz {  // lzma/base.h 5.2: lzma_stream (136 bytes on LP64), lzma_stream_decoder, lzma_code, lzma_end
    struct Stream {
        const uint8_t *next_in; size_t avail_in; uint64_t total_in;
        uint8_t *next_out; size_t avail_out; uint64_t total_out;
        const void *allocator; void *internal;
        void *reserved_ptr[4]; uint64_t reserved_int1, reserved_int2; size_t reserved_int3, reserved_int4;
        int reserved_enum1, reserved_enum2;
        uint64_t slack[8];  // (room, should a later 5.x grow the struct)
    };
    int (*init)(Stream *, uint64_t, uint32_t) = nullptr;
    int (*run)(Stream *, int) = nullptr;
    void (*end)(Stream *) = nullptr;
    Stream z{};
    bool live = false;
    static const char *name() { return "xz"; }
    bool load() {
        void *h = open_lib("liblzma.so.5", nullptr);
        // (lzma_stream as declared above: liblzma 5.x — 5.0 .. 5.8 kept the layout; anything else is not used)
        uint32_t (*ver)() = nullptr;
        if (!h || !sym(h, "lzma_version_number", ver) || ver() / 10000000u != 5u) return false;
        return sym(h, "lzma_stream_decoder", init) && sym(h, "lzma_code", run) && sym(h, "lzma_end", end);
    }
    bool start() { memset(&z, 0, sizeof z); live = init(&z, UINT64_MAX, 0) == 0; return live; }  // one .xz stream per start
    void stop() { if (live) end(&z); live = false; }
    int step(const uint8_t *&in, size_t &n_in, uint8_t *&out, size_t &n_out) {
        z.next_in = in; z.avail_in = n_in; z.next_out = out; z.avail_out = n_out;
        const int rc = run(&z, 0 /* LZMA_RUN */);
        in = z.next_in; n_in = z.avail_in; out = z.next_out; n_out = z.avail_out;
        return rc == 1 /* LZMA_STREAM_END */ ? 1 : (rc == 0 || rc == 10 /* LZMA_BUF_ERROR: no progress possible yet */) ? 0 : -1;
    }
};

struct Zstd {  // zstd.h 1.4: ZSTD_DStream, ZSTD_inBuffer / ZSTD_outBuffer, ZSTD_decompressStream (0: a frame is complete)
    struct Buf { void *p; size_t size, pos; };
    void *(*create)() = nullptr;
    size_t (*destroy)(void *) = nullptr;
    size_t (*init)(void *) = nullptr;
    size_t (*run)(void *, Buf *, Buf *) = nullptr;
    unsigned (*is_error)(size_t) = nullptr;
    void *d = nullptr;
    static const char *name() { return "zstd"; }
    bool load() {
        void *h = open_lib("libzstd.so.1", nullptr);
        // (the streaming API and ZSTD_inBuffer / ZSTD_outBuffer as above are stable since zstd 1.3; 1.x only)
        unsigned (*ver)() = nullptr;
        if (!h || !sym(h, "ZSTD_versionNumber", ver) || ver() < 10300u || ver() >= 20000u) return false;
        return sym(h, "ZSTD_createDStream", create) && sym(h, "ZSTD_freeDStream", destroy) && sym(h, "ZSTD_initDStream", init) &&
               sym(h, "ZSTD_decompressStream", run) && sym(h, "ZSTD_isError", is_error);
    }
    bool start() {
        if (!d) d = create();
        return d && !is_error(init(d));
    }
    void stop() { if (d) destroy(d); d = nullptr; }
    int step(const uint8_t *&in, size_t &n_in, uint8_t *&out, size_t &n_out) {
        Buf o = {out, n_out, 0}, i = {(void *)in, n_in, 0};
        const size_t rc = run(d, &o, &i);
        in += i.pos; n_in -= i.pos; out += o.pos; n_out -= o.pos;
        if (is_error(rc)) return -1;
        return rc == 0 ? 1 : 0;
    }
};
}  // namespace codec

template <class Codec, class Reader>
class CodecReader {
  public:
    explicit CodecReader(Reader r) : r_(std::move(r)), in_(1 << 16) {
        if (!c_.load()) throw Error(ErrorKind::InvalidData, std::string("Niffler failled in compression detection: no decoder library on this machine for ") + Codec::name());
    }
    CodecReader(CodecReader &&o) noexcept : r_(std::move(o.r_)), in_(std::move(o.in_)), c_(o.c_) {
        if (o.started_) std::terminate();  // (a decoder's state holds pointers into itself: only a fresh reader moves)
    }
    CodecReader(const CodecReader &) = delete;
    ~CodecReader() { c_.stop(); }
    size_t read(uint8_t *dst, size_t n) {
        started_ = true;
        uint8_t *out = dst;
        size_t left = n;
        while (left == n && !done_) {  // until some output exists or the input is exhausted at a stream boundary
            if (avail_ == 0 && !in_eof_) {
                avail_ = r_.read(in_.data(), in_.size());
                pos_ = in_.data();
                if (avail_ == 0) in_eof_ = true;
            }
            if (!in_stream_) {  // between streams: more input => the next stream, none => EOF
                if (avail_ == 0) { done_ = true; break; }
                if (!c_.start()) throw Error(ErrorKind::Other, std::string(Codec::name()) + " decoder init failed");
                in_stream_ = true;
            }
            // (input exhausted inside a stream: the decoder may still hold output and the stream's end — the step before may have
            // consumed the last input bytes with its output buffer full — so it is stepped once more with no input; only a step
            // that then yields nothing and no end is a truncated stream)
            const bool drained = avail_ == 0;
            const size_t left0 = left;
            const uint8_t *ip = pos_;
            const int rc = c_.step(ip, avail_, out, left);
            pos_ = ip;
            if (rc < 0) throw Error(ErrorKind::InvalidData, std::string("corrupt ") + Codec::name() + " stream");
            if (rc == 1) { c_.stop(); in_stream_ = false; }
            else if (drained && left == left0) throw Error(ErrorKind::InvalidData, std::string("unexpected end of ") + Codec::name() + " stream");
        }
        return n - left;
    }
  private:
    Reader r_;
    std::vector<uint8_t> in_;
    Codec c_;
    const uint8_t *pos_ = nullptr;
    size_t avail_ = 0;
    bool started_ = false, in_eof_ = false, in_stream_ = false, done_ = false;
};

// LZ4 frame decoder on the host (the published LZ4 Frame format 1.6.x and LZ4 block format: no dependency, the image has
// no lz4 headers).  Concatenated frames and skippable frames are accepted like `lz4 -d` does; blocks may depend on the
// previous ones (a 64 KiB window is kept); checksums are skipped, not verified.  Runs on the thread_reader thread.
template <class Reader>
class Lz4Reader {
  public:
    explicit Lz4Reader(Reader r) : r_(std::move(r)), in_(1 << 16) {}
    size_t read(uint8_t *dst, size_t n) {
        size_t got = 0;
        while (got < n) {
            if (out_pos_ == out_.size()) {
                if (done_ || !next_block()) break;
                continue;
            }
            const size_t k = std::min(n - got, out_.size() - out_pos_);
            memcpy(dst + got, out_.data() + out_pos_, k);
            out_pos_ += k;
            got += k;
            if (got) break;  // (a Read may return short: hand over what one block gave)
        }
        return got;
    }

  private:
    bool fill(uint8_t *p, size_t n, bool eof_ok = false) {  // exactly n bytes, or (eof_ok) nothing at all
        size_t have = 0;
        while (have < n) {
            if (in_pos_ == in_len_) {
                in_len_ = r_.read(in_.data(), in_.size());
                in_pos_ = 0;
                if (in_len_ == 0) {
                    if (have == 0 && eof_ok) return false;
                    throw Error(ErrorKind::InvalidData, "unexpected end of lz4 stream");
                }
            }
            const size_t k = std::min(n - have, in_len_ - in_pos_);
            memcpy(p + have, in_.data() + in_pos_, k);
            in_pos_ += k;
            have += k;
        }
        return true;
    }
    static uint32_t le32(const uint8_t *p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }
    bool next_block() {  // decodes the next data block into out_ (behind the kept window); false at the end of input
        for (;;) {
            if (!in_frame_) {
                uint8_t m[4];
                if (!fill(m, 4, true)) { done_ = true; return false; }
                const uint32_t magic = le32(m);
                if ((magic & 0xFFFFFFF0u) == 0x184D2A50u) {  // skippable frame
                    uint8_t sz[4];
                    fill(sz, 4);
                    uint8_t skip[4096];  // (the size is untrusted, up to 4 GiB: skipped in pieces, never allocated)
                    for (uint64_t left = le32(sz); left;) {
                        const size_t k = (size_t)std::min<uint64_t>(left, sizeof skip);
                        fill(skip, k);
                        left -= k;
                    }
                    continue;
                }
                if (magic != 0x184D2204u) throw Error(ErrorKind::InvalidData, "corrupt lz4 stream: bad frame magic");
                uint8_t fb[2];
                fill(fb, 2);
                if ((fb[0] >> 6) != 1) throw Error(ErrorKind::InvalidData, "corrupt lz4 stream: unsupported frame version");
                independent_ = (fb[0] >> 5) & 1;
                {   // block maximum size (BD byte, bits 4-6): 4 = 64 KiB .. 7 = 4 MiB; a block's OUTPUT may not exceed it
                    const uint32_t bd = (fb[1] >> 4) & 7u;
                    if (bd < 4) throw Error(ErrorKind::InvalidData, "corrupt lz4 stream: block maximum size");
                    block_max_ = (size_t)1 << (8 + 2 * bd);
                }
                block_checksum_ = (fb[0] >> 4) & 1;
                const bool content_size = (fb[0] >> 3) & 1;
                content_checksum_ = (fb[0] >> 2) & 1;
                const bool dict_id = fb[0] & 1;
                uint8_t rest[13];
                fill(rest, (content_size ? 8 : 0) + (dict_id ? 4 : 0) + 1);  // + header checksum
                in_frame_ = true;
                window_.clear();
            }
            uint8_t bs[4];
            fill(bs, 4);
            const uint32_t word = le32(bs);
            if (word == 0) {  // EndMark
                if (content_checksum_) { uint8_t c[4]; fill(c, 4); }
                in_frame_ = false;
                continue;
            }
            const bool stored = (word >> 31) != 0;
            const uint32_t size = word & 0x7FFFFFFFu;
            if (size > block_max_) throw Error(ErrorKind::InvalidData, "corrupt lz4 stream: block size");
            blk_.resize(size);
            if (size) fill(blk_.data(), size);
            if (block_checksum_) { uint8_t c[4]; fill(c, 4); }
            // out_ = [window (up to 64 KiB of earlier output) | this block's output]; out_pos_ starts behind the window
            out_.assign(window_.begin(), window_.end());
            const size_t base = out_.size();
            if (stored) out_.insert(out_.end(), blk_.begin(), blk_.end());
            else decode_block(base);
            out_pos_ = base;
            if (!independent_) {
                const size_t keep = std::min<size_t>(out_.size(), 65536);
                window_.assign(out_.end() - keep, out_.end());
            }
            if (out_.size() > base) return true;
        }
    }
    void decode_block(size_t base) {  // LZ4 block format: token, literals, 2-byte offset, match length
        const uint8_t *p = blk_.data(), *end = p + blk_.size();
        auto bad = [] { throw Error(ErrorKind::InvalidData, "corrupt lz4 stream: block data"); };
        while (p < end) {
            const uint32_t tok = *p++;
            size_t lit = tok >> 4;
            if (lit == 15) {
                uint8_t b;
                do { if (p >= end) bad(); b = *p++; lit += b; } while (b == 255);
            }
            if ((size_t)(end - p) < lit || out_.size() - base + lit > block_max_) bad();
            out_.insert(out_.end(), p, p + lit);
            p += lit;
            if (p >= end) break;  // the last sequence is literals only
            if (end - p < 2) bad();
            const size_t off = p[0] | (p[1] << 8);
            p += 2;
            size_t len = (tok & 15) + 4;
            if ((tok & 15) == 15) {
                uint8_t b;
                do { if (p >= end) bad(); b = *p++; len += b; } while (b == 255);
            }
            if (off == 0 || off > out_.size()) bad();
            if (out_.size() - base + len > block_max_) bad();   // (a match may not expand a block beyond the frame's block maximum)
            const size_t from = out_.size() - off;
            out_.reserve(out_.size() + len);
            for (size_t i = 0; i < len; ++i) out_.push_back(out_[from + i]);  // (may overlap its own output)
        }
    }
    Reader r_;
    std::vector<uint8_t> in_, blk_, out_, window_;
    size_t in_pos_ = 0, in_len_ = 0, out_pos_ = 0, block_max_ = 4u << 20;
    bool in_frame_ = false, done_ = false, independent_ = true, block_checksum_ = false, content_checksum_ = false;
};

// parse_path (src/lib.rs:167-196): open the file (nullopt / "-" = stdin), sniff the compression
// format, hand the closure a Parser over plain bytes.  Plain input goes straight to the parser;
// compressed input is decoded on a thread_reader thread with the reference's parameters (4 MiB
// buffers, queue of 2; src/lib.rs:191).  Gzip is decoded with zlib; bzip2 / xz / zstd with the system's run-time libraries
// (codec::*, bound with dlopen; an input in a format whose library is missing fails like niffler built without that feature).
// lz4 frames are an EXTENSION of the mirror: the crate's docs name lz4 (src/lib.rs:137-141), but niffler >= 2.4 has no lz4
// format, so the reference as built today would read an .lz4 file as plain bytes and fail on its header.
template <class F>
auto with_plain_reader(const std::optional<std::string> &path, F use, unsigned read_threads = 1) {  // use(DynReader &) sees plain bytes
    FILE *f = stdin;
    bool own = false;
    if (path && *path != "-") {
        f = fopen(path->c_str(), "rb");
        if (!f) throw Error(ErrorKind::Other, "cannot open " + *path);
        own = true;
    }
    FileReader file(f, own, read_threads);
    uint8_t first[5];
    size_t have = 0;
    while (have < 5) {
        size_t k = file.read(first + have, 5 - have);
        if (k == 0) break;
        have += k;
    }
    if (have < 5)
        throw Error(ErrorKind::InvalidData,
                    "Niffler failled in compression detection File is too short, less than five bytes");
    const Compression fmt = sniff_compression(first);
    PrefixReader<FileReader> chained(std::move(file), first, have);
    if (fmt == Compression::No) {
        DynReader dyn(&chained);
        return use(dyn);
    }
#ifndef FASTQ_NO_ZLIB
    if (fmt == Compression::Gzip) {
        return thread_reader(1 << 22, 2, GzReader<PrefixReader<FileReader>>(std::move(chained)), [&](auto reader) {
            DynReader dyn(&reader);
            return use(dyn);
        });
    }
#endif
    if (fmt == Compression::Lz4) {
        return thread_reader(1 << 22, 2, Lz4Reader<PrefixReader<FileReader>>(std::move(chained)), [&](auto reader) {
            DynReader dyn(&reader);
            return use(dyn);
        });
    }
    if (fmt == Compression::Bzip2) {
        return thread_reader(1 << 22, 2, CodecReader<codec::Bz2, PrefixReader<FileReader>>(std::move(chained)), [&](auto reader) {
            DynReader dyn(&reader);
            return use(dyn);
        });
    }
    if (fmt == Compression::Lzma) {
        return thread_reader(1 << 22, 2, CodecReader<codec::Xz, PrefixReader<FileReader>>(std::move(chained)), [&](auto reader) {
            DynReader dyn(&reader);
            return use(dyn);
        });
    }
    if (fmt == Compression::Zstd) {
        return thread_reader(1 << 22, 2, CodecReader<codec::Zstd, PrefixReader<FileReader>>(std::move(chained)), [&](auto reader) {
            DynReader dyn(&reader);
            return use(dyn);
        });
    }
    throw Error(ErrorKind::InvalidData,
                std::string("Niffler failled in compression detection: no decoder in this build for ") +
                    (fmt == Compression::Gzip ? "gzip" : fmt == Compression::Bzip2 ? "bzip2"
                     : fmt == Compression::Lzma ? "xz" : "zstd"));
}

template <class F>
auto parse_path(const std::optional<std::string> &path, F func, Options opt = Options()) {
    return with_plain_reader(path, [&](DynReader &dyn) {
        Parser<DynReader> p(std::move(dyn), opt);
        return func(p);
    }, opt.read_threads ? opt.read_threads : std::max(1u, std::min(8u, std::thread::hardware_concurrency() / 8u)));
}

}  // namespace fastq
