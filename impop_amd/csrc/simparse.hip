// simparse.hip — native `.sim` TSV ingest (host code only).
// Replaces pica2.read_similarity_file (pica2.py:6-58) / h-fst.read_similarity_file
// (h-fst.py:84-119) for the common, clean file shape: header row, tab-separated, columns
// group.a / group.b / estimated.identity present (extra columns ignored), plain decimal
// numbers.  Anything it is not sure CPython's csv + float() would treat identically (quotes,
// short rows, underscores, hex floats, ...) is reported as IMPOP_E_UNSUPPORTED so that the
// caller falls back to the reference-faithful Python reader — never a silent difference.
// Semantics kept: key = unordered name pair, later rows overwrite earlier ones, self pairs kept,
// every name of either column is an element, row count = parsed data rows.
#include <errno.h>
#include <fcntl.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <string>
#include <string_view>
#include <unordered_map>
#include <vector>

#include "internal.h"

struct impop_sim {
    std::vector<std::string> names;          // sorted (byte order == code-point order for UTF-8)
    std::vector<uint32_t> rank_of_id;        // first-seen id -> sorted rank
    std::vector<uint32_t> row_a, row_b;      // per data row: first-seen ids
    std::vector<double> row_v;
    uint64_t n_rows = 0;
    int64_t bad_line = -1;                   // 1-based line of the first unparsable value (pica2 flavour)
    std::string bad_text;
    uint64_t n_bad = 0;                      // skipped values (h-fst flavour)
};

namespace {

// CPython float(): optional whitespace, [+-], digits with optional '.', optional exponent, or
// inf / infinity / nan (case-insensitive).  Underscores and hex are valid/invalid differently
// from strtod => "unsure".
enum class Tok { Ok, Invalid, Unsure };

Tok parse_float(std::string_view t, double *out) {
    size_t b = 0, e = t.size();
    while (b < e && (t[b] == ' ' || t[b] == '\t' || t[b] == '\f' || t[b] == '\v')) ++b;
    while (e > b && (t[e - 1] == ' ' || t[e - 1] == '\t' || t[e - 1] == '\f' || t[e - 1] == '\v')) --e;
    if (b == e) return Tok::Invalid;
    bool digits = false, weird = false;
    for (size_t i = b; i < e; ++i) {
        const char c = t[i];
        if (c >= '0' && c <= '9') digits = true;
        else if (c == '+' || c == '-' || c == '.' || c == 'e' || c == 'E') {}
        else weird = true;
    }
    if (weird) {
        std::string low(t.substr(b, e - b));
        for (auto &c : low) c = (char)tolower((unsigned char)c);
        size_t o = (low[0] == '+' || low[0] == '-') ? 1 : 0;
        const std::string w = low.substr(o);
        if (w == "inf" || w == "infinity") { *out = low[0] == '-' ? -INFINITY : INFINITY; return Tok::Ok; }
        if (w == "nan") { *out = NAN; return Tok::Ok; }
        for (char c : low)
            if (c == '_' || c == 'x') return Tok::Unsure;  // float("1_0") is valid Python, "0x10" is not: let Python decide
        return Tok::Invalid;
    }
    if (!digits) return Tok::Invalid;
    char buf[128];
    const size_t len = e - b;
    if (len >= sizeof buf) return Tok::Unsure;
    memcpy(buf, t.data() + b, len);
    buf[len] = 0;
    char *endp = nullptr;
    errno = 0;
    const double v = strtod(buf, &endp);  // glibc: correctly rounded, same value as CPython's dtoa
    if (endp != buf + len) return Tok::Invalid;
    *out = v;
    return Tok::Ok;
}

}  // namespace

using namespace impop;

// flavor 0: pica2 (first bad value aborts: line + text recorded); 1: h-fst (bad values skipped and counted)
IMPOP_API int impop_sim_parse(const char *path, int flavor, impop_sim **out) {
    REQUIRE(path && out, "impop_sim_parse: NULL argument");
    *out = nullptr;
    int fd = open(path, O_RDONLY);
    if (fd < 0) {
        set_error("File not found: %s", path);
        return IMPOP_E_INVALID;
    }
    struct stat st;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) {
        close(fd);
        set_error("impop_sim_parse: not a regular file: %s", path);
        return IMPOP_E_UNSUPPORTED;
    }
    const size_t size = (size_t)st.st_size;
    const char *data = size ? (const char *)mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0) : "";
    close(fd);
    if (size && data == MAP_FAILED) {
        set_error("impop_sim_parse: mmap failed for %s", path);
        return IMPOP_E_UNSUPPORTED;
    }
    auto done = [&](int code) {
        if (size) munmap((void *)data, size);
        return code;
    };
    if (memchr(data, '"', size)) {  // csv quoting rules: leave to Python
        set_error("impop_sim_parse: quoted fields present");
        return done(IMPOP_E_UNSUPPORTED);
    }
    size_t pos = 0;
    auto next_line = [&](std::string_view *line) -> bool {  // '\n', '\r\n' and '\r' all end a row (csv module)
        if (pos >= size) return false;
        size_t i = pos;
        while (i < size && data[i] != '\n' && data[i] != '\r') ++i;
        *line = std::string_view(data + pos, i - pos);
        if (i < size && data[i] == '\r' && i + 1 < size && data[i + 1] == '\n') ++i;
        pos = i + 1;
        return true;
    };
    std::string_view line;
    // header: csv.DictReader skips leading rows that are completely empty
    bool have_header = false;
    while (next_line(&line))
        if (!line.empty()) { have_header = true; break; }
    if (!have_header) {
        set_error("empty file or missing header");
        return done(IMPOP_E_UNSUPPORTED);
    }
    int col_a = -1, col_b = -1, col_v = -1, ncol = 0;
    {
        size_t s = 0;
        for (;;) {
            size_t t = line.find('\t', s);
            std::string_view f = line.substr(s, t == std::string_view::npos ? std::string_view::npos : t - s);
            if (f == "group.a" && col_a < 0) col_a = ncol;
            else if (f == "group.b" && col_b < 0) col_b = ncol;
            else if (f == "estimated.identity" && col_v < 0) col_v = ncol;
            else if (f == "group.a" || f == "group.b" || f == "estimated.identity") {  // duplicate header names: DictReader keeps the last
                set_error("impop_sim_parse: duplicate column names");
                return done(IMPOP_E_UNSUPPORTED);
            }
            ++ncol;
            if (t == std::string_view::npos) break;
            s = t + 1;
        }
    }
    if (col_a < 0 || col_b < 0 || col_v < 0) {
        set_error("missing required columns");
        return done(IMPOP_E_UNSUPPORTED);  // the Python reader prints the reference's exact message
    }
    const int need = std::max(col_a, std::max(col_b, col_v)) + 1;
    impop_sim *S = new impop_sim();
    std::unordered_map<std::string_view, uint32_t> ids;
    ids.reserve(4096);
    std::vector<std::string_view> first_seen;
    auto id_of = [&](std::string_view nm) -> uint32_t {
        auto it = ids.find(nm);
        if (it != ids.end()) return it->second;
        const uint32_t id = (uint32_t)first_seen.size();
        ids.emplace(nm, id);
        first_seen.push_back(nm);
        return id;
    };
    uint64_t line_no = 1;
    std::vector<std::string_view> f((size_t)need);
    while (next_line(&line)) {
        ++line_no;
        if (line.empty()) continue;  // csv.DictReader skips empty rows
        int nf = 0;
        size_t s = 0;
        for (;;) {
            size_t t = line.find('\t', s);
            if (nf < need) f[(size_t)nf] = line.substr(s, t == std::string_view::npos ? std::string_view::npos : t - s);
            ++nf;
            if (t == std::string_view::npos) break;
            s = t + 1;
        }
        if (nf < need) {  // short row -> None values in DictReader: reference behaviour differs per script
            delete S;
            set_error("impop_sim_parse: short row at line %llu", (unsigned long long)line_no);
            return done(IMPOP_E_UNSUPPORTED);
        }
        double v;
        const Tok tk = parse_float(f[(size_t)col_v], &v);
        if (tk == Tok::Unsure) {
            delete S;
            set_error("impop_sim_parse: unusual number syntax at line %llu", (unsigned long long)line_no);
            return done(IMPOP_E_UNSUPPORTED);
        }
        if (tk == Tok::Invalid) {
            if (flavor == 0) {  // pica2.py:39-41: message + exit(1); the pair count already includes this row
                S->bad_line = (int64_t)line_no;
                S->bad_text.assign(f[(size_t)col_v]);
                break;
            }
            S->n_bad++;  // h-fst.py:107-109: warn and skip BEFORE touching names
            continue;
        }
        S->n_rows++;
        S->row_a.push_back(id_of(f[(size_t)col_a]));
        S->row_b.push_back(id_of(f[(size_t)col_b]));
        S->row_v.push_back(v);
    }
    // sort names; map first-seen ids to ranks
    const uint32_t n = (uint32_t)first_seen.size();
    std::vector<uint32_t> order(n);
    for (uint32_t i = 0; i < n; ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return first_seen[x] < first_seen[y]; });
    S->rank_of_id.resize(n);
    S->names.resize(n);
    for (uint32_t r = 0; r < n; ++r) {
        S->rank_of_id[order[r]] = r;
        S->names[r].assign(first_seen[order[r]]);
    }
    *out = S;
    return done(IMPOP_OK);
}

IMPOP_API int impop_sim_info(const impop_sim *s, uint32_t *n_names, uint64_t *n_rows, uint64_t *names_bytes,
                             int64_t *bad_line, uint64_t *n_bad) {
    REQUIRE(s, "impop_sim_info: handle is NULL");
    if (n_names) *n_names = (uint32_t)s->names.size();
    if (n_rows) *n_rows = s->n_rows;
    if (names_bytes) {
        uint64_t b = 0;
        for (auto &x : s->names) b += x.size() + 1;
        *names_bytes = b;
    }
    if (bad_line) *bad_line = s->bad_line;
    if (n_bad) *n_bad = s->n_bad;
    return IMPOP_OK;
}

IMPOP_API int impop_sim_names(const impop_sim *s, char *buf) {
    REQUIRE(s && buf, "impop_sim_names: NULL argument");
    for (auto &x : s->names) {
        memcpy(buf, x.data(), x.size());
        buf[x.size()] = 0;
        buf += x.size() + 1;
    }
    return IMPOP_OK;
}

// first_seen_out[k] = sorted rank of the k-th DISTINCT name in file order (group.a before group.b of a row):
// the order in which the reference's reader adds names to its `elements` set (pica2.py:45-46), which decides
// that set's iteration order and with it the seed order of pica2's greedy grouping
IMPOP_API int impop_sim_first_seen(const impop_sim *s, uint32_t *first_seen_out) {
    REQUIRE(s, "impop_sim_first_seen: handle is NULL");
    REQUIRE(s->rank_of_id.empty() || first_seen_out, "impop_sim_first_seen: out is NULL");
    for (size_t k = 0; k < s->rank_of_id.size(); ++k) first_seen_out[k] = s->rank_of_id[k];
    return IMPOP_OK;
}

IMPOP_API int impop_sim_bad_text(const impop_sim *s, char *buf, size_t buflen) {
    REQUIRE(s && buf && buflen, "impop_sim_bad_text: bad arguments");
    snprintf(buf, buflen, "%s", s->bad_text.c_str());
    return IMPOP_OK;
}

// dense n x n identity in sorted-name order; NaN = pair absent; later rows overwrite earlier ones
IMPOP_API int impop_sim_dense(const impop_sim *s, double *out) {
    REQUIRE(s, "impop_sim_dense: handle is NULL");
    const size_t n = s->names.size();
    REQUIRE(n == 0 || out, "impop_sim_dense: out is NULL");
    for (size_t k = 0; k < n * n; ++k) out[k] = NAN;
    for (size_t k = 0; k < s->row_v.size(); ++k) {
        const size_t i = s->rank_of_id[s->row_a[k]], j = s->rank_of_id[s->row_b[k]];
        out[i * n + j] = s->row_v[k];
        out[j * n + i] = s->row_v[k];
    }
    return IMPOP_OK;
}

IMPOP_API int impop_sim_free(impop_sim *s) {
    delete s;
    return IMPOP_OK;
}
