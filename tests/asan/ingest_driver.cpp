// Sanitizer driver for the host side of libidelucs_hip.so (the CPU code under ASan/UBSan).
// Test infrastructure: built by `make -C idelucs_amd/csrc asan` together with host_ingest.cpp + runtime.cpp, CPU code objects
// only (never the GPU ones), run by tests/test_host_ingest_asan.py.  It drives both FASTA readers of the C ABI
// (idl_fasta_open / _sizes / _export / _pack_range, reference idelucs/utils.py:224-260; idl_fasta_parse_pack on host arenas)
// over the files given, then over `fuzz` seeded mutations of them written to `tmpdir`, and cross-checks the two readers
// (names, lengths, packed bytes at the slots reported) so that a silent disagreement is an error too.  Exit 0 = clean.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/idelucs_hip.h"

static std::vector<uint8_t> slurp(const char *path)
{
    std::vector<uint8_t> b;
    FILE *f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(3); }
    uint8_t tmp[1 << 16];
    size_t n;
    while ((n = fread(tmp, 1, sizeof tmp, f)) > 0) b.insert(b.end(), tmp, tmp + n);
    fclose(f);
    return b;
}

static void spill(const std::string &path, const std::vector<uint8_t> &b)
{
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) { fprintf(stderr, "cannot write %s\n", path.c_str()); exit(3); }
    if (!b.empty()) fwrite(b.data(), 1, b.size(), f);
    fclose(f);
}

struct Parsed {
    int rc = 0;
    int64_t n = 0, bases = 0, slots = 0, names_bytes = 0;
    std::vector<uint8_t> names, codes, mask, bytes;
    std::vector<int64_t> name_off, lengths, byte_off, slot_off;
};

// the general reader: open, sizes, export everything, re-pack in two ranges and compare with the export's packing
static Parsed general(const char *path, int check)
{
    Parsed p;
    idl_fasta *h = nullptr;
    p.rc = idl_fasta_open(path, check, &h);
    if (p.rc != IDL_OK) return p;
    if (idl_fasta_sizes(h, &p.n, &p.bases, &p.slots, &p.names_bytes) != IDL_OK) { fprintf(stderr, "sizes failed\n"); exit(4); }
    p.names.resize((size_t)p.names_bytes + 1); p.name_off.resize((size_t)p.n + 1); p.lengths.resize((size_t)p.n + 1);
    p.bytes.resize((size_t)p.bases + 1); p.byte_off.resize((size_t)p.n + 1);
    p.codes.resize((size_t)p.slots * 16 + 16); p.mask.resize((size_t)p.slots * 8 + 8); p.slot_off.resize((size_t)p.n + 1);
    if (idl_fasta_export(h, p.names.data(), p.name_off.data(), p.lengths.data(), p.bytes.data(), p.byte_off.data(), p.codes.data(),
                         p.mask.data(), p.slot_off.data()) != IDL_OK) { fprintf(stderr, "export failed: %s\n", idl_last_error()); exit(4); }
    std::vector<uint8_t> c2(p.codes.size(), 0), m2(p.mask.size(), 0);
    const int64_t mid = p.n / 2;
    if (idl_fasta_pack_range(h, mid, p.n, c2.data(), m2.data()) != IDL_OK || idl_fasta_pack_range(h, 0, mid, c2.data(), m2.data()) != IDL_OK) {
        fprintf(stderr, "pack_range failed: %s\n", idl_last_error()); exit(4);
    }
    if (memcmp(c2.data(), p.codes.data(), (size_t)p.slots * 16) || memcmp(m2.data(), p.mask.data(), (size_t)p.slots * 8)) {
        fprintf(stderr, "%s: pack_range disagrees with export\n", path); exit(5);
    }
    if (idl_fasta_pack_range(h, 0, p.n + 1, c2.data(), m2.data()) == IDL_OK) { fprintf(stderr, "pack_range took an invalid range\n"); exit(5); }
    idl_fasta_close(h);
    return p;
}

// the one-pass reader on host arenas, cross-checked against the general reader's result
static int one_pass(const char *path, const Parsed &g, size_t file_bytes, int threads)
{
    const int64_t cap = (int64_t)file_bytes / 48 + 4096 * (int64_t)threads + 1024;
    std::vector<uint8_t> codes((size_t)cap * 16, 0xAB), mask((size_t)cap * 8, 0xCD);
    idl_fasta *h = nullptr;
    const int rc = idl_fasta_parse_pack(path, codes.data(), mask.data(), cap, nullptr, nullptr, nullptr, &h);
    if (rc == IDL_FALLBACK) return rc;
    if (rc != IDL_OK) {
        if (g.rc != rc) { fprintf(stderr, "%s: one-pass rc %d, general rc %d\n", path, rc, g.rc); exit(5); }
        return rc;
    }
    if (g.rc != IDL_OK) { fprintf(stderr, "%s: the one-pass reader accepted what the general reader refused (%d)\n", path, g.rc); exit(5); }
    int64_t n = 0, bases = 0, slots = 0, nb = 0;
    idl_fasta_sizes(h, &n, &bases, &slots, &nb);
    if (n != g.n || bases != g.bases || nb != g.names_bytes) { fprintf(stderr, "%s: readers disagree on sizes\n", path); exit(5); }
    std::vector<uint8_t> names((size_t)nb + 1);
    std::vector<int64_t> name_off((size_t)n + 1), lengths((size_t)n + 1), so((size_t)n + 1);
    if (idl_fasta_export(h, names.data(), name_off.data(), lengths.data(), nullptr, nullptr, nullptr, nullptr, nullptr) != IDL_OK ||
        idl_fasta_arena_slots(h, so.data()) != IDL_OK) { fprintf(stderr, "one-pass export failed: %s\n", idl_last_error()); exit(4); }
    if (memcmp(names.data(), g.names.data(), (size_t)nb) || memcmp(lengths.data(), g.lengths.data(), (size_t)n * 8)) {
        fprintf(stderr, "%s: readers disagree on names / lengths\n", path); exit(5);
    }
    for (int64_t i = 0; i < n; ++i) {
        const int64_t ns = g.slot_off[(size_t)i + 1] - g.slot_off[(size_t)i], a = so[(size_t)i], b = g.slot_off[(size_t)i];
        if (a < 0 || a + ns > cap) { fprintf(stderr, "%s: record %lld outside the arena\n", path, (long long)i); exit(5); }
        if (memcmp(codes.data() + a * 16, g.codes.data() + b * 16, (size_t)ns * 16) || memcmp(mask.data() + a * 8, g.mask.data() + b * 8, (size_t)ns * 8)) {
            fprintf(stderr, "%s: readers disagree on the packed bytes of record %lld\n", path, (long long)i); exit(5);
        }
    }
    idl_fasta_close(h);
    return rc;
}

static uint64_t rng_state;
static uint32_t rnd()
{
    rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;     // xorshift64
    return (uint32_t)(rng_state >> 16);
}

static void mutate(std::vector<uint8_t> &b)
{
    static const char alphabet[] = "ACGTacgtNnUuRYKMSWBDHV-\n\n\r\t >#!Z\0\xff";
    const int edits = 1 + (int)(rnd() % 12);
    for (int e = 0; e < edits; ++e) {
        const uint32_t kind = rnd() % 6;
        const size_t at = b.empty() ? 0 : rnd() % b.size();
        if (kind == 0 && !b.empty()) b[at] = (uint8_t)alphabet[rnd() % (sizeof alphabet - 1)];
        else if (kind == 1) b.insert(b.begin() + (long)at, (uint8_t)alphabet[rnd() % (sizeof alphabet - 1)]);
        else if (kind == 2 && !b.empty()) b.erase(b.begin() + (long)at, b.begin() + (long)std::min(b.size(), at + 1 + rnd() % 40));
        else if (kind == 3 && !b.empty()) b.resize(at);                                                 // truncate (no trailing newline, cut headers)
        else if (kind == 4) { const char *h = ">\n"; b.insert(b.begin() + (long)at, h, h + 1 + rnd() % 2); }   // empty ids / bare '>'
        else if (!b.empty()) { const size_t n = std::min<size_t>(b.size() - at, 1 + rnd() % 200); std::vector<uint8_t> c(b.begin() + (long)at, b.begin() + (long)(at + n)); b.insert(b.begin() + (long)at, c.begin(), c.end()); }
    }
}

static void drive(const char *path, const int *threads, int n_threads_cfg, long *counts)
{
    const size_t bytes = slurp(path).size();
    for (int t = 0; t < n_threads_cfg; ++t) {
        char env[16];
        snprintf(env, sizeof env, "%d", threads[t]);
        setenv("IDELUCS_THREADS", env, 1);
        for (int check = 1; check >= 0; --check) {
            const Parsed g = general(path, check);
            ++counts[g.rc == IDL_OK ? 0 : 1];
            if (check) {
                const int rc = one_pass(path, g, bytes, threads[t]);
                ++counts[rc == IDL_OK ? 2 : (rc == IDL_FALLBACK ? 3 : 4)];
            }
        }
        idl_ingest_release();
    }
}

int main(int argc, char **argv)
{
    if (argc < 5) { fprintf(stderr, "usage: %s tmpdir fuzz_cases seed file...\n", argv[0]); return 2; }
    const std::string tmpdir = argv[1];
    const int fuzz = atoi(argv[2]);
    rng_state = 0x9E3779B97F4A7C15ull ^ (uint64_t)atoll(argv[3]);
    setenv("IDELUCS_PAR_MIN", "0", 1);               // several threads also on small files
    const int threads[3] = {1, 3, 8};
    long counts[5] = {0, 0, 0, 0, 0};
    std::vector<std::vector<uint8_t>> seeds;
    for (int i = 4; i < argc; ++i) {
        drive(argv[i], threads, 3, counts);
        std::vector<uint8_t> b = slurp(argv[i]);
        if (b.size() > (1u << 16)) b.resize(1u << 16);         // fuzz on the head of large fixtures
        seeds.push_back(b);
    }
    seeds.push_back({});                                         // the empty file (UBSan: memchr on a null mapping)
    const std::string fz = tmpdir + "/fuzz.fas";
    for (int c = 0; c < fuzz; ++c) {
        std::vector<uint8_t> b = seeds[rnd() % seeds.size()];
        if (c > 0) mutate(b);
        else b.clear();
        spill(fz, b);
        const int one[1] = {threads[c % 3]};
        drive(fz.c_str(), one, 1, counts);
    }
    printf("general ok %ld, general refused %ld, one-pass ok %ld, one-pass fallback %ld, one-pass refused %ld\n", counts[0], counts[1],
           counts[2], counts[3], counts[4]);
    return 0;
}
