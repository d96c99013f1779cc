// host_tests — the reference's unit tests (src/lib.rs:611-811) and doc-test (src/lib.rs:474-508)
// written against the C++ mirror, same names and assertions, plus a differential mode used by
// tests/test_gpu_host_mirror.py:   host_tests --dump FILE THREADS BUFSIZE
// prints status, record count, RecordSet sizes and per-worker counts for comparison with the oracle.
#include <cassert>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <stdexcept>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "fastq.hpp"

using namespace fastq;

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #c); exit(1); } } while (0)

static std::string sv(bytes_view v) { return std::string((const char *)v.data(), v.size()); }
static Options small() { Options o; o.slot_bytes = 1 << 20; return o; }

static void correct() {  // lib.rs:616-652
    std::string d = "@hi\nNN\n+\n++\n@hallo\nTCC\n+\nabc\n";
    Parser<MemReader> p(MemReader(d), small());
    int i = 0;
    bool ok = p.each([&](const RefRecord &r) {
        if (i == 0) {
            CHECK(sv(r.head()) == "hi" && sv(r.seq()) == "NN" && sv(r.qual()) == "++");
            OwnedRecord o = r.to_owned_record();
            CHECK(o.head == "hi" && o.seq == "NN" && o.qual == "++");
            std::ostringstream out;
            CHECK(o.write(out) == 12 && out.str() == "@hi\nNN\n+\n++\n");
        } else {
            CHECK(sv(r.head()) == "hallo" && sv(r.seq()) == "TCC" && sv(r.qual()) == "abc");
            std::ostringstream out;
            CHECK(r.to_owned_record().write(out) == 17 && out.str() == "@hallo\nTCC\n+\nabc\n");
        }
        CHECK(i < 2);
        ++i;
        return true;
    });
    CHECK(ok && i == 2);
}
static void empty_id() {  // lib.rs:654-666
    std::string d = "@\nNN\n+\n++\n";
    Parser<MemReader> p(MemReader(d), small());
    p.each([&](const RefRecord &r) { CHECK(sv(r.head()) == "" && sv(r.seq()) == "NN" && sv(r.qual()) == "++"); return true; });
}
static void missing_lines() {  // lib.rs:668-686
    std::string d = "@hi\nNN\n+\n++\n@hi\nNN";
    Parser<MemReader> p(MemReader(d), small());
    int seen = 0;
    try {
        p.each([&](const RefRecord &r) { CHECK(sv(r.head()) == "hi"); ++seen; return true; });
        CHECK(!"should fail");
    } catch (const Error &e) { CHECK(e.kind() == ErrorKind::InvalidData && seen == 1); }
}
static void truncated() {  // lib.rs:688-697
    std::string d = "@hi\nNN\n+\n++";
    Parser<MemReader> p(MemReader(d), small());
    try { p.each([&](const RefRecord &) { CHECK(false); return true; }); CHECK(false); } catch (const Error &) {}
}
static void second_idline() {  // lib.rs:699-714
    std::string d = "@hi\nNN\n+hi\n++\n@hi\nNN\n+hi\n++\n";
    Parser<MemReader> p(MemReader(d), small());
    p.each([&](const RefRecord &r) {
        std::ostringstream out;
        CHECK(r.write(out) == 14 && out.str() == "@hi\nNN\n+hi\n++\n");
        return true;
    });
}
static void windows_lineend() {  // lib.rs:716-727
    std::string d = "@hi\r\nNN\r\n+\r\n++\r\n@hi\r\nNN\r\n+\r\n++\r\n";
    Parser<MemReader> p(MemReader(d), small());
    int n = 0;
    p.each([&](const RefRecord &r) { CHECK(sv(r.head()) == "hi" && sv(r.seq()) == "NN" && sv(r.qual()) == "++"); ++n; return true; });
    CHECK(n == 2);
}
static void length_mismatch() {  // lib.rs:729-738
    std::string d = "@hi\nNN\n+\n+\n";
    Parser<MemReader> p(MemReader(d), small());
    try { p.each([&](const RefRecord &) { CHECK(false); return true; }); CHECK(false); }
    catch (const Error &e) { CHECK(std::string(e.what()) == "Sequence and quality length mismatch"); }
}
static std::string huge() { std::string d = "@"; for (size_t i = 0; i < BUFSIZE; ++i) d += "longid"; return d; }
static void huge_incomplete() {  // lib.rs:740-750
    std::string d = huge();
    Parser<MemReader> p(MemReader(d), small());
    try { p.each([&](const RefRecord &) { return true; }); CHECK(false); }
    catch (const Error &e) { CHECK(std::string(e.what()) == "Fastq record is too long"); }
}
static void bufflen() {  // lib.rs:752-774
    std::string d = "@" + std::string(BUFSIZE - 8, 'a') + "\nA\n+\nB\n";
    Parser<MemReader> p(MemReader(d), small());
    auto vals = p.parallel_each<uint64_t>(2, [](auto next) { uint64_t c = 0; while (auto s = next()) for (auto r : s->iter()) { (void)r; ++c; } return c; });
    CHECK(vals.size() == 2 && vals[0] + vals[1] == 1);
}
static void refset() {  // lib.rs:776-791
    std::string d = "@hi\nNN\n+\n++\n@hi\nNN\n+\n++\n";
    Parser<MemReader> p(MemReader(d), small());
    size_t count = 0, sets = 0;
    bool first_empty = false;
    p.record_sets([&](RecordSet &&s) {
        if (sets++ == 0) first_empty = s.is_empty();
        for (auto r : s.iter()) { ++count; CHECK(sv(r.head()) == "hi" && sv(r.seq()) == "NN" && sv(r.qual()) == "++"); }
        return true;
    });
    CHECK(count == 2 && first_empty);
}
static void refset_incomplete() {  // lib.rs:793-798
    std::string d = "@hi\nNN\n+\n++\n@hi\nNN\n+\n++";
    Parser<MemReader> p(MemReader(d), small());
    try { p.record_sets([&](RecordSet &&) { return true; }); CHECK(false); }
    catch (const Error &e) { CHECK(std::string(e.what()) == "Truncated input file."); }
}
static void refset_huge_incomplete() {  // lib.rs:800-810
    std::string d = huge();
    Parser<MemReader> p(MemReader(d), small());
    try { p.record_sets([&](RecordSet &&) { return true; }); CHECK(false); }
    catch (const Error &e) { CHECK(std::string(e.what()) == "Fastq record is too long."); }
}
static void doctest_parallel_each() {  // lib.rs:474-508
    std::string d = "@hi\nATTAATTAATTA\n+\n++++++++++++\n";
    Parser<MemReader> p(MemReader(d), small());
    auto res = p.parallel_each<std::optional<std::string>>(4, [](auto next) -> std::optional<std::string> {
        while (auto set = next())
            for (auto r : set->iter())
                if (sv(r.seq()).rfind("ATTAATTA", 0) == 0) return sv(r.seq());
        return std::nullopt;
    });
    int found = 0;
    for (auto &r : res) if (r) { CHECK(*r == "ATTAATTAATTA"); ++found; }
    CHECK(found == 1);
}
static void zipped_and_thread_reader() {  // each_zipped lib.rs:577-609, thread_reader.rs:182-200
    std::string a = "@a1\nAC\n+\nII\n@a2\nGG\n+\nII\n", b = "@b1\nTT\n+\nII\n";
    Parser<MemReader> p1(MemReader(a), small()), p2(MemReader(b), small());
    std::vector<std::string> seen;
    auto fin = each_zipped(p1, p2, [&](std::optional<RefRecord> x, std::optional<RefRecord> y) {
        seen.push_back((x ? sv(x->head()) : "-") + "/" + (y ? sv(y->head()) : "-"));
        return std::make_pair(true, true);
    });
    CHECK(fin.first && fin.second);
    CHECK(seen.size() == 3 && seen[0] == "a1/b1" && seen[1] == "a2/-" && seen[2] == "-/-");
    std::string big;
    for (int i = 0; i < 20000; ++i) big += "@r" + std::to_string(i) + "\nACGTN\n+\nIIIII\n";
    size_t n = thread_reader(1 << 16, 2, MemReader(big), [&](auto reader) {
        Parser<decltype(reader)> p(reader, small());
        size_t c = 0;
        p.each([&](const RefRecord &r) { c += r.validate_dnan() && !r.validate_dna(); return true; });
        return c;
    });
    CHECK(n == 20000);
}

// a pipe: read() comes back with at most 3001 bytes at a time
struct DribbleReader {
    MemReader m;
    explicit DribbleReader(const std::string &s) : m(s) {}
    size_t read(uint8_t *dst, size_t n) { return m.read(dst, n < 3001 ? n : 3001); }
};

// a reader that hands out at most `cap` bytes per call (the oracle's max_read) and counts what it has handed out
struct CappedReader {
    MemReader m;
    size_t cap;
    size_t *handed;
    CappedReader(const std::string &s, size_t c, size_t *h) : m(s), cap(c), handed(h) {}
    size_t read(uint8_t *dst, size_t n) {
        const size_t k = m.read(dst, n < cap ? n : cap);
        *handed += k;
        return k;
    }
};

// --pipe FILE CAP BUFSIZE SLOT: Parser::each and record_sets over a reader that comes back with at most CAP bytes per read()
// (0: a reader that fills every read).  The reference's outcome for records of BUFSIZE - 15 .. BUFSIZE bytes depends on the
// reader's read sizes (src/buffer.rs:51-100): the mirror notes its own reads and replays the reference's (csrc/replay.h).
// Also prints how many bytes the reader had handed out when the closure saw record 0 (src/lib.rs:264-275: one refill).
static int pipe_mode(const char *file, size_t cap, uint64_t bufsize, uint64_t slot) {
    std::ifstream f(file, std::ios::binary);
    std::string d((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    Options o;
    o.bufsize = bufsize;
    o.slot_bytes = slot;
    if (!cap) cap = (size_t)-1;
    {
        size_t handed = 0, n = 0, bases = 0, first_after = 0;
        Parser<CappedReader> p(CappedReader(d, cap, &handed), o);
        std::string err = "ok";
        try { p.each([&](const RefRecord &r) { if (!n) first_after = handed; ++n; bases += r.seq().size(); return true; }); }
        catch (const Error &e) { err = e.what(); }
        printf("each %zu %zu %s\n", n, bases, err.c_str());
        printf("first %zu\n", first_after);
    }
    {
        size_t handed = 0;
        Parser<CappedReader> p(CappedReader(d, cap, &handed), o);
        std::string err = "ok", sizes;
        try { p.record_sets([&](RecordSet &&s) { sizes += std::to_string(s.len()) + ","; return true; }); }
        catch (const Error &e) { err = e.what(); }
        printf("sets %s %s\n", sizes.empty() ? "-" : sizes.c_str(), err.c_str());
    }
    return 0;
}

// an order-independent digest of the records a consumer saw: sum over the records of FNV-1a(head | 0 | seq | 0 | qual | raw bytes)
static uint64_t digest(const RefRecord &r) {
    uint64_t h = 1469598103934665603ull;
    auto eat = [&](fastq::bytes_view v) {
        for (uint8_t x : v) h = (h ^ x) * 1099511628211ull;
        h = (h ^ 0xFF) * 1099511628211ull;
    };
    eat(r.head()); eat(r.seq()); eat(r.qual()); eat(r.data());
    return h;
}

static int dump(const char *file, int threads, uint64_t bufsize, uint64_t slot) {
    std::ifstream f(file, std::ios::binary);
    std::string d((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    Options o;
    o.bufsize = bufsize;
    o.slot_bytes = slot;
    // each, record_sets and parallel_each; pre = "" for the parser that does everything on the calling thread, "ahead_" for the one
    // with a filler thread of its own (Options::read_ahead: a MemReader fills every read, as a file does)
    auto three = [&](Options oo, const char *pre) {
        {   // each
            Parser<MemReader> p(MemReader(d), oo);
            size_t n = 0, bases = 0;
            uint64_t sum = 0;
            std::string err = "ok";
            try { p.each([&](const RefRecord &r) { ++n; bases += r.seq().size(); sum += digest(r); return true; }); }
            catch (const Error &e) { err = e.what(); }
            printf("%seach %zu %zu %s\n", pre, n, bases, err.c_str());
            printf("%seachsum %llu\n", pre, (unsigned long long)sum);
        }
        {   // record_sets: the sets are kept until the parse is over (they co-own ring slots: the ring must cope), then walked
            Parser<MemReader> p(MemReader(d), oo);
            std::string err = "ok", sizes;
            std::vector<RecordSet> kept;
            uint64_t sum = 0, n = 0;
            try {
                p.record_sets([&](RecordSet &&s) {
                    sizes += std::to_string(s.len()) + ",";
                    if (kept.size() >= 40) {  // (at most a few slots' worth held at a time: the oldest are walked and dropped)
                        for (const RefRecord &r : kept.front()) { sum += digest(r); ++n; }
                        kept.erase(kept.begin());
                    }
                    kept.push_back(std::move(s));
                    return true;
                });
            } catch (const Error &e) { err = e.what(); }
            for (const RecordSet &s : kept)
                for (const RefRecord &r : s) { sum += digest(r); ++n; }
            printf("%ssets %s %s\n", pre, sizes.empty() ? "-" : sizes.c_str(), err.c_str());
            printf("%ssetsum %llu %llu\n", pre, (unsigned long long)n, (unsigned long long)sum);
        }
        {   // parallel_each
            Parser<MemReader> p(MemReader(d), oo);
            std::string err = "ok", counts;
            uint64_t sum = 0;
            try {
                auto res = p.parallel_each<std::pair<size_t, uint64_t>>((size_t)threads, [](auto next) {
                    size_t c = 0;
                    uint64_t h = 0;
                    while (auto s = next()) {
                        c += s->len();
                        for (const RefRecord &r : *s) h += digest(r);
                    }
                    return std::make_pair(c, h);
                });
                for (auto &c : res) { counts += std::to_string(c.first) + ","; sum += c.second; }
            } catch (const Error &e) { err = e.what(); }
            printf("%sworkers %s %s\n", pre, counts.empty() ? "-" : counts.c_str(), err.c_str());
            printf("%sworksum %llu\n", pre, (unsigned long long)sum);
        }
    };
    three(o, "");
    {   // each over a pipe, chunks submitted as the reads come in (Options::low_latency): the same records
        Options ol = o;
        ol.low_latency = true;
        Parser<DribbleReader> p(DribbleReader(d), ol);
        size_t n = 0, bases = 0;
        std::string err = "ok";
        try { p.each([&](const RefRecord &r) { ++n; bases += r.seq().size(); return true; }); }
        catch (const Error &e) { err = e.what(); }
        printf("pipe %zu %zu %s\n", n, bases, err.c_str());
    }
    Options oa = o;
    oa.read_ahead = true;
    three(oa, "ahead_");
    return 0;
}

// --zip FILE1 FILE2 FLAGS BUFSIZE SLOT: each_zipped (src/lib.rs:577-609) with a scripted callback: call i returns the
// advance flags FLAGS[i % len] ('0'..'3': bit 0 parser 1, bit 1 parser 2).  Prints "call <head1|-> <head2|->" per call,
// then "zipped <finished1> <finished2> <error|ok>".
static int zip(const char *file1, const char *file2, const char *flags, uint64_t bufsize, uint64_t slot) {
    auto slurp = [](const char *f) {
        std::ifstream in(f, std::ios::binary);
        return std::string((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    };
    const std::string d1 = slurp(file1), d2 = slurp(file2), fl = flags;
    Options o;
    o.bufsize = bufsize;
    o.slot_bytes = slot;
    Parser<MemReader> p1(MemReader(d1), o), p2(MemReader(d2), o);
    std::string err = "ok";
    std::pair<bool, bool> fin{false, false};
    size_t i = 0;
    try {
        fin = each_zipped(p1, p2, [&](const std::optional<RefRecord> &a, const std::optional<RefRecord> &b) {
            auto str = [](fastq::bytes_view v) { return std::string((const char *)v.data(), v.size()); };
            const std::string h1 = a ? str(a->head()) : "-", h2 = b ? str(b->head()) : "-";
            printf("call %s %s\n", h1.c_str(), h2.c_str());
            const int f = fl.empty() ? 3 : fl[i % fl.size()] - '0';
            ++i;
            return std::make_pair((f & 1) != 0, (f & 2) != 0);
        });
    } catch (const Error &e) { err = e.what(); }
    printf("zipped %d %d %s\n", fin.first ? 1 : 0, fin.second ? 1 : 0, err.c_str());
    return 0;
}

// --plain FILE: the reader chain of parse_path without a Parser (no GPU needed): prints the number
// of plain bytes and their FNV-1a hash, or the error.
static int plain(const char *file) {
    try {
        return fastq::with_plain_reader(std::string(file), [](fastq::DynReader &r) {
            std::vector<uint8_t> buf(1 << 16);
            uint64_t n = 0, h = 1469598103934665603ull;
            for (;;) {
                size_t k = r.read(buf.data(), 1 + (n * 7919) % buf.size());  // odd request sizes
                if (!k) break;
                for (size_t i = 0; i < k; ++i) h = (h ^ buf[i]) * 1099511628211ull;
                n += k;
            }
            printf("plain %llu %016llx\n", (unsigned long long)n, (unsigned long long)h);
            return 0;
        });
    } catch (const fastq::Error &e) {
        printf("error %s\n", e.what());
        return 3;
    }
}

int main(int argc, char **argv) {
    if (argc >= 3 && !strcmp(argv[1], "--plain")) return plain(argv[2]);
    if (argc >= 7 && !strcmp(argv[1], "--zip")) return zip(argv[2], argv[3], argv[4], strtoull(argv[5], 0, 0), strtoull(argv[6], 0, 0));
    if (argc >= 4 && !strcmp(argv[1], "--sharded")) {  // each_sharded as the only rank: the C++ mirror over fqh_shard_stream_*
        std::ifstream in(argv[2], std::ios::binary);
        std::string d((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
        const uint32_t lmax = (uint32_t)atoi(argv[3]);
        fqh_ctx *ctx = nullptr;
        if (fqh_create(0, &ctx) != FQH_OK) return 2;
        const uint64_t nw = 9 + (uint64_t)lmax * 264;
        void *dh = nullptr;
        if (fqh_dev_alloc(ctx, nw * 8, &dh) != FQH_OK || fqh_memset(ctx, dh, 0, nw * 8) != FQH_OK) return 2;
        const bool fail_io = argc >= 5 && !strcmp(argv[4], "fail-io");   // the read callback throws for the file's second half
        Options o;
        o.slot_bytes = fail_io ? 1 << 16 : 1 << 20;   // (slots well below half the file: reads in front of the failing bytes succeed)
        try {
            const uint64_t n = each_sharded(ctx, nullptr, 1, 0, [&](uint8_t *dst, uint64_t off, uint64_t k) {
                                                if (fail_io && off + k > d.size() / 2) throw std::runtime_error("injected read failure");
                                                memcpy(dst, d.data() + off, k);
                                            },
                                            d.size(), lmax, (uint64_t *)dh, o);
            std::vector<uint64_t> h(nw);
            if (fqh_memcpy_d2h(ctx, h.data(), dh, nw * 8) != FQH_OK) return 2;
            uint64_t sq = 0, sb = 0;
            for (uint64_t i = 0; i < (uint64_t)lmax * 256; ++i) sq += h[9 + i] * (i % 251 + 1);
            for (uint64_t i = 0; i < (uint64_t)lmax * 8; ++i) sb += h[9 + (uint64_t)lmax * 256 + i] * (i % 13 + 1);
            printf("ok %llu %llu %llu %llu %llu\n", (unsigned long long)n, (unsigned long long)h[1], (unsigned long long)h[2],
                   (unsigned long long)sq, (unsigned long long)sb);
        } catch (const Error &e) {
            printf("err %s\n", e.what());
        }
        fqh_dev_free(ctx, dh);
        fqh_destroy(ctx);
        return 0;
    }
    if (argc >= 6 && !strcmp(argv[1], "--pipe"))
        return pipe_mode(argv[2], (size_t)strtoull(argv[3], 0, 0), strtoull(argv[4], 0, 0), strtoull(argv[5], 0, 0));
    if (argc >= 5 && !strcmp(argv[1], "--dump"))
        return dump(argv[2], atoi(argv[3]), strtoull(argv[4], 0, 0), argc > 5 ? strtoull(argv[5], 0, 0) : (1 << 20));
#define RUN(t) do { t(); printf("ok %s\n", #t); fflush(stdout); } while (0)
    RUN(correct); RUN(empty_id); RUN(missing_lines); RUN(truncated); RUN(second_idline); RUN(windows_lineend);
    RUN(length_mismatch); RUN(huge_incomplete); RUN(bufflen); RUN(refset); RUN(refset_incomplete);
    RUN(refset_huge_incomplete); RUN(doctest_parallel_each); RUN(zipped_and_thread_reader);
    printf("all host tests passed\n");
    return 0;
}
