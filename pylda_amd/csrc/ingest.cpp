// Native corpus ingest: parse_data (variational_bayes.py:98-130) without the Python interpreter.
// Pure host code (no HIP calls): also compiled on its own with -fsanitize=address,undefined by
// tests/test_ingest_sanitizers.py.
#include "../../include/pylda_hip.h"

#include <cstring>
#include <new>
#include <string>
#include <string_view>
#include <unordered_map>
#include <vector>

namespace {

// Length in bytes of the white-space character at p (UTF-8), 0 if there is none: the set Python's
// str.split() splits on (str.isspace): U+0009-000D, 001C-001F, 0020, 0085, 00A0, 1680, 2000-200A,
// 2028, 2029, 202F, 205F, 3000.
inline int blank_len(const unsigned char* p, const unsigned char* end)
{
    const unsigned c = p[0];
    if (c <= 0x20) return (c == 0x20 || (c >= 0x09 && c <= 0x0d) || (c >= 0x1c && c <= 0x1f)) ? 1 : 0;
    if (c < 0xc2) return 0;
    if (c == 0xc2) return (end - p >= 2 && (p[1] == 0x85 || p[1] == 0xa0)) ? 2 : 0;
    if (end - p < 3) return 0;
    if (c == 0xe1) return (p[1] == 0x9a && p[2] == 0x80) ? 3 : 0;
    if (c == 0xe2) {
        if (p[1] == 0x80) return (p[2] <= 0x8a && p[2] >= 0x80) || p[2] == 0xa8 || p[2] == 0xa9 || p[2] == 0xaf ? 3 : 0;
        return (p[1] == 0x81 && p[2] == 0x9f) ? 3 : 0;
    }
    if (c == 0xe3) return (p[1] == 0x80 && p[2] == 0x80) ? 3 : 0;
    return 0;
}

}  // namespace

extern "C" int pylda_parse_corpus(const char* text, int64_t text_bytes, int doc_separator, const char* vocab,
                                  int64_t vocab_bytes, int lowercase, int64_t* n_docs, int64_t* nnz,
                                  int64_t* doc_ptr, int32_t* term_id, int32_t* term_ct, int64_t* dropped_docs)
{
    if (!text || !vocab || text_bytes < 0 || vocab_bytes < 0 || !n_docs || !nnz) return PYLDA_ERR_INVALID;
    if (doc_separator < 0 || doc_separator > 255) return PYLDA_ERR_INVALID;
    const bool fill = term_id != nullptr;
    if (fill && (!doc_ptr || !term_ct)) return PYLDA_ERR_INVALID;
    try {
        typedef const unsigned char* bytes;
        const unsigned char sep = (unsigned char)doc_separator;
        // vocabulary: one type per line, id = index among the distinct lines (first occurrence wins,
        // as parse_vocabulary); surrounding blanks are not part of the type
        std::unordered_map<std::string_view, int32_t> lookup;
        lookup.reserve((size_t)(vocab_bytes / 6 + 16));
        {
            int32_t next = 0;
            const char *p = vocab, *end = vocab + vocab_bytes;
            while (p < end) {
                const char* eol = static_cast<const char*>(memchr(p, '\n', (size_t)(end - p)));
                if (!eol) eol = end;
                bytes a = (bytes)p, b = (bytes)eol;
                for (int n; a < b && (n = blank_len(a, b)) > 0;) a += n;
                for (;;) {                      // trailing blanks (1-3 bytes each)
                    int cut = 0;
                    for (int n = 1; n <= 3 && b - a >= n && !cut; ++n)
                        if (blank_len(b - n, b) == n) cut = n;
                    if (!cut) break;
                    b -= cut;
                }
                if (b > a && lookup.emplace(std::string_view((const char*)a, (size_t)(b - a)), next).second) ++next;
                p = eol + 1;
            }
        }
        std::string lowered;
        std::vector<int32_t> slot(lookup.size(), -1);   // term id -> position in this document's list
        std::vector<int32_t> ids, cts;
        int64_t docs = 0, entries = 0, dropped = 0;
        if (fill) doc_ptr[0] = 0;
        bytes p = (bytes)text, end = (bytes)text + text_bytes;
        while (p < end) {
            bytes eol = static_cast<bytes>(memchr(p, sep, (size_t)(end - p)));
            if (!eol) eol = end;
            ids.clear();
            cts.clear();
            bytes q = p;
            while (q < eol) {
                for (int n; q < eol && (n = blank_len(q, eol)) > 0;) q += n;
                bytes tok = q;
                while (q < eol && blank_len(q, eol) == 0) ++q;
                if (q == tok) break;
                std::string_view key((const char*)tok, (size_t)(q - tok));
                if (lowercase) {
                    lowered.assign((const char*)tok, (const char*)q);
                    for (char& ch : lowered)
                        if (ch >= 'A' && ch <= 'Z') ch = (char)(ch - 'A' + 'a');
                    key = lowered;
                }
                const auto hit = lookup.find(key);
                if (hit == lookup.end()) continue;                         // :108-109
                int32_t& at = slot[(size_t)hit->second];
                if (at < 0) {
                    at = (int32_t)ids.size();
                    ids.push_back(hit->second);
                    cts.push_back(1);
                } else {
                    cts[(size_t)at] += 1;
                }
            }
            for (int32_t id : ids) slot[(size_t)id] = -1;
            const bool had_text = eol > p || eol < end;                    // a line exists (even if empty)
            if (!ids.empty()) {
                if (fill) {
                    memcpy(term_id + entries, ids.data(), ids.size() * sizeof(int32_t));
                    memcpy(term_ct + entries, cts.data(), cts.size() * sizeof(int32_t));
                    doc_ptr[docs + 1] = entries + (int64_t)ids.size();
                }
                entries += (int64_t)ids.size();
                ++docs;
            } else if (had_text) {
                ++dropped;                                                 // :116-118
            }
            p = eol + 1;
        }
        *n_docs = docs;
        *nnz = entries;
        if (dropped_docs) *dropped_docs = dropped;
    } catch (const std::bad_alloc&) {
        return PYLDA_ERR_OOM;
    }
    return PYLDA_OK;
}
