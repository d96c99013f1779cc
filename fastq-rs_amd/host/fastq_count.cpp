// fastq_count — the GPU-backed twin of the reference's examples/fastq-count.rs:6-24 (and, with
// --threads N, of examples/fastq-count-thread.rs): prints the number of records of a plain FASTQ
// file (path argument, "-" or nothing = stdin); on a malformed file it fails like the reference's
// `.expect("Invalid fastq file")` with the reference's error message.
//   --threads N       parallel_each with N workers (examples/fastq-count-thread.rs)
//   --read-threads N  a regular file is read with N pread()s side by side per ring slot (default: an eighth of the host's hardware
//                     threads, at most 8; 1 = the reference's one reader)
//   --read-ahead      a thread of the parser's own fills the ring while the calling thread parses (for files)
//   --slot-mib M      size of a ring slot (default 32)
//   --repeat K        parse the file K times in this process and print every pass's seconds to stderr ("pass i: S s"): the
//                     first pass pays for the HIP runtime's start-up (a few hundred ms), the later ones are the path itself
//   --stats LMAX      the histogram consumer instead of the count: the per-position quality / base histograms of every record
//                     (the closure over Record::seq()/qual(), src/records.rs:75-90) accumulated on the GPU while the file
//                     streams through the ring (FQH_STREAM_STATS); prints "records bases checksum", checksum = sum over
//                     [scalars | quality | base] of value * (index + 1) mod 2^64
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "fastq.hpp"

// The histogram consumer: the ABI's ring with FQH_STREAM_STATS, fed by the same FileReader parse_path uses.
static int run_stats(const std::optional<std::string> &path, uint32_t lmax, const fastq::Options &opt) {
    FILE *f = stdin;
    bool own = false;
    if (path && *path != "-") {
        f = fopen(path->c_str(), "rb");
        if (!f) { fprintf(stderr, "cannot open %s\n", path->c_str()); return 2; }
        own = true;
    }
    fastq::FileReader file(f, own, opt.read_threads ? opt.read_threads : std::max(1u, std::min(8u, std::thread::hardware_concurrency() / 8u)));
    fastq::detail::Handles h;
    auto die = [&](const char *what) { fprintf(stderr, "%s: %s\n", what, fqh_last_error(h.ctx)); return 2; };
    if (fqh_create(opt.device, &h.ctx) != FQH_OK) return die("fqh_create");
    if (fqh_stream_create(h.ctx, opt.slot_bytes, opt.n_slots, FQH_STREAM_STATS, &h.st) != FQH_OK) return die("fqh_stream_create");
    const uint64_t n_words = FQH_NSCALARS + (uint64_t)lmax * 264;
    void *d = nullptr;
    if (fqh_dev_alloc(h.ctx, n_words * 8, &d) != FQH_OK || fqh_memset(h.ctx, d, 0, n_words * 8) != FQH_OK) return die("fqh_dev_alloc");
    uint64_t *d_sc = (uint64_t *)d, *d_q = d_sc + FQH_NSCALARS, *d_b = d_q + (uint64_t)lmax * 256;
    if (fqh_stream_set_stats(h.st, lmax, d_q, d_b, d_sc) != FQH_OK) return die("fqh_stream_set_stats");
    bool eof = false;
    uint64_t in_flight = 0, records = 0;
    int status = FQH_OK;
    for (;;) {
        while (!eof) {
            uint8_t *dst;
            uint64_t cap;
            const fqh_status st = fqh_stream_acquire(h.st, &dst, &cap);
            if (st == FQH_E_CAPACITY) break;
            if (st != FQH_OK) return die("fqh_stream_acquire");
            uint64_t n = 0;
            while (n < cap) {
                const size_t got = file.read(dst + n, (size_t)(cap - n));
                if (!got) { eof = true; break; }
                n += got;
            }
            if (fqh_stream_submit(h.st, n, eof ? 1 : 0) != FQH_OK) return die("fqh_stream_submit");
            ++in_flight;
        }
        if (!in_flight) break;
        fqh_chunk c;
        if (fqh_stream_collect(h.st, &c) != FQH_OK) return die("fqh_stream_collect");
        --in_flight;
        records += c.n_records;
        status = c.parse_status;
        const bool fin = c.is_final != 0;
        fqh_stream_release(h.st);
        if (status != FQH_OK || fin) break;
    }
    if (status != FQH_OK) {
        fprintf(stderr, "Invalid fastq file: %s\n", fqh_strerror((fqh_status)status));
        return 101;
    }
    std::vector<uint64_t> host(n_words);
    if (fqh_sync(h.ctx) != FQH_OK || fqh_memcpy_d2h(h.ctx, host.data(), d, n_words * 8) != FQH_OK) return die("fqh_memcpy_d2h");
    uint64_t sum = 0;
    for (uint64_t i = 0; i < n_words; ++i) sum += host[i] * (i + 1);
    printf("%llu %llu %llu\n", (unsigned long long)records, (unsigned long long)host[1], (unsigned long long)sum);
    (void)fqh_dev_free(h.ctx, d);
    return 0;
}

int main(int argc, char **argv) {
    std::optional<std::string> path;
    int threads = 0;
    uint32_t stats_lmax = 0;
    int repeat = 1;
    fastq::Options opt;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--threads") && i + 1 < argc) threads = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--read-threads") && i + 1 < argc) opt.read_threads = (unsigned)atoi(argv[++i]);
        else if (!strcmp(argv[i], "--read-ahead")) opt.read_ahead = true;
        else if (!strcmp(argv[i], "--slot-mib") && i + 1 < argc) opt.slot_bytes = (uint64_t)atoi(argv[++i]) << 20;
        else if (!strcmp(argv[i], "--repeat") && i + 1 < argc) repeat = std::max(1, atoi(argv[++i]));
        else if (!strcmp(argv[i], "--stats") && i + 1 < argc) stats_lmax = (uint32_t)atoi(argv[++i]);
        else path = argv[i];
    }
    int rc = 0;
    for (int pass = 0; pass < repeat && rc == 0; ++pass) {
        const auto t0 = std::chrono::steady_clock::now();
        try {
            if (stats_lmax) {
                rc = run_stats(path, stats_lmax, opt);
            } else {
                size_t total = 0;
                fastq::parse_path(path, [&](auto &parser) {
                    if (threads > 0) {
                        auto res = parser.template parallel_each<size_t>((size_t)threads, [](auto next) {
                            size_t t = 0;
                            while (auto set = next()) t += set->len();
                            return t;
                        });
                        for (size_t r : res) total += r;
                    } else {
                        parser.each([&](const fastq::RefRecord &) { ++total; return true; });
                    }
                    return 0;
                }, opt);
                printf("%zu\n", total);
            }
        } catch (const fastq::Error &e) {
            fprintf(stderr, "Invalid fastq file: %s\n", e.what());
            return 101;  // a Rust panic exits with 101
        }
        if (repeat > 1)
            fprintf(stderr, "pass %d: %.6f s\n", pass, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    }
    return rc;
}
