// fastq_count — the GPU-backed twin of the reference's examples/fastq-count.rs:6-24 (and, with
// --threads N, of examples/fastq-count-thread.rs): prints the number of records of a plain FASTQ
// file (path argument, "-" or nothing = stdin); on a malformed file it fails like the reference's
// `.expect("Invalid fastq file")` with the reference's error message.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "fastq.hpp"

int main(int argc, char **argv) {
    std::optional<std::string> path;
    int threads = 0;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--threads") && i + 1 < argc) threads = atoi(argv[++i]);
        else path = argv[i];
    }
    try {
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
        });
        printf("%zu\n", total);
    } catch (const fastq::Error &e) {
        fprintf(stderr, "Invalid fastq file: %s\n", e.what());
        return 101;  // a Rust panic exits with 101
    }
    return 0;
}
