// Sanitizer driver for the native corpus ingest (pylda_amd/csrc/ingest.cpp): built by
// tests/test_ingest_sanitizers.py with -fsanitize=address,undefined and run on the CPU.
// Feeds pylda_parse_corpus structured and random byte strings (truncated UTF-8 sequences at
// the buffer end, separators at the edges, empty inputs) held in exactly-sized heap buffers so
// that any read past the end is an ASan error, and checks the CSR invariants of every result.
#include "../../include/pylda_hip.h"

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

static uint64_t state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd()
{
    state ^= state << 13; state ^= state >> 7; state ^= state << 17;
    return (uint32_t)(state >> 32);
}

static int check(const std::string& text, int sep, const std::string& vocab, int lowercase)
{
    // exactly-sized copies: ASan red zones sit right behind the last byte
    char* t = (char*)malloc(text.size() ? text.size() : 1);
    char* v = (char*)malloc(vocab.size() ? vocab.size() : 1);
    memcpy(t, text.data(), text.size());
    memcpy(v, vocab.data(), vocab.size());
    int64_t docs = -1, nnz = -1, dropped = -1;
    int rc = pylda_parse_corpus(t, (int64_t)text.size(), sep, v, (int64_t)vocab.size(), lowercase, &docs, &nnz,
                                nullptr, nullptr, nullptr, &dropped);
    if (rc != PYLDA_OK || docs < 0 || nnz < 0) { fprintf(stderr, "sizing pass failed rc=%d\n", rc); return 1; }
    std::vector<int64_t> ptr((size_t)docs + 1, -1);
    std::vector<int32_t> ids((size_t)nnz), cts((size_t)nnz);
    int64_t docs2 = -1, nnz2 = -1;
    rc = pylda_parse_corpus(t, (int64_t)text.size(), sep, v, (int64_t)vocab.size(), lowercase, &docs2, &nnz2, ptr.data(),
                            nnz ? ids.data() : (int32_t*)ptr.data() /* non-NULL */, nnz ? cts.data() : (int32_t*)ptr.data() + 1,
                            nullptr);
    free(t);
    free(v);
    if (rc != PYLDA_OK || docs2 != docs || nnz2 != nnz) { fprintf(stderr, "fill pass disagrees\n"); return 1; }
    if (docs > 0 || nnz > 0) {
        if (ptr[0] != 0 || ptr[(size_t)docs] != nnz) { fprintf(stderr, "doc_ptr ends\n"); return 1; }
        for (int64_t d = 0; d < docs; ++d)
            if (ptr[(size_t)d + 1] <= ptr[(size_t)d]) { fprintf(stderr, "empty document kept\n"); return 1; }
        for (int64_t i = 0; i < nnz; ++i)
            if (ids[(size_t)i] < 0 || cts[(size_t)i] < 1) { fprintf(stderr, "bad entry\n"); return 1; }
    }
    return 0;
}

int main()
{
    int bad = 0;
    const std::string vocab = "alpha\nbeta\n gamma \nalpha\n\xc2\xa0" "delta\xe2\x80\x83\n\n\xe3\x80\x80\nz";
    const char* fixed[] = {
        "", "\n", "alpha", "alpha\n", "\nalpha", "alpha beta\nbeta\tgamma\r\n\n delta  z ", "alpha\xc2", "alpha\xe2\x80",
        "alpha\xe2", "\xe3\x80\x80" "alpha\xe3\x80\x80" "beta\xe3\x80", "\xc2\xa0\xc2\x85\xe1\x9a\x80\xe2\x80\xa8", "ALPHA Beta",
        "alpha\x1c" "beta\x1f" "gamma", "\xff\xfe\xfd", "alpha\0beta"};
    for (const char* f : fixed)
        for (int sep : {(int)'\n', 0, 255})
            for (int lower : {0, 1}) bad += check(std::string(f, f == fixed[14] ? 10 : strlen(f)), sep, vocab, lower);
    bad += check("alpha", '\n', "", 0);
    bad += check("", '\n', "", 0);
    // random byte soup over an alphabet rich in separators, blanks and UTF-8 lead bytes
    const unsigned char alphabet[] = {'a', 'b', 'z', ' ', '\n', '\t', 0, 0xc2, 0xa0, 0x85, 0xe1, 0x9a, 0x80, 0xe2, 0x81, 0x9f,
                                      0xa8, 0xe3, 0xff, 'A', 0x1d, 0x8a};
    for (int iter = 0; iter < 20000; ++iter) {
        std::string t, v;
        const int n = (int)(rnd() % 40), m = (int)(rnd() % 24);
        for (int i = 0; i < n; ++i) t.push_back((char)alphabet[rnd() % sizeof alphabet]);
        for (int i = 0; i < m; ++i) v.push_back((char)alphabet[rnd() % sizeof alphabet]);
        const int seps[] = {'\n', 0, 255, ' '};
        bad += check(t, seps[rnd() % 4], v, (int)(rnd() & 1));
    }
    // argument validation
    int64_t a = 0, b = 0;
    if (pylda_parse_corpus(nullptr, 0, '\n', "", 0, 0, &a, &b, nullptr, nullptr, nullptr, nullptr) != PYLDA_ERR_INVALID) ++bad;
    if (pylda_parse_corpus("", 0, 256, "", 0, 0, &a, &b, nullptr, nullptr, nullptr, nullptr) != PYLDA_ERR_INVALID) ++bad;
    if (pylda_parse_corpus("", -1, '\n', "", 0, 0, &a, &b, nullptr, nullptr, nullptr, nullptr) != PYLDA_ERR_INVALID) ++bad;
    if (bad) { fprintf(stderr, "%d failures\n", bad); return 1; }
    puts("ingest sanitizer run: ok");
    return 0;
}
