// gfaparse.hip — native GFA ingest (host code only): S / P / W lines of what `impg query -o gfa`
// (run_tajd.sh:126) or `odgi view -g` emit -> the NODE-level haplotype x segment presence matrix, bit-packed, plus
// the segment lengths (site weights) and, given a reference path prefix, every column's reference coordinate.
// Same rules as the Python extractor impop_amd/extract.py:from_gfa(expand_bp=False) — which stays the readable
// definition and the fallback — but one mmap pass and no dense byte matrix: a chromosome graph (10^7 segments x 465
// paths) is 0.6 GB of bits here and 4.6 GB of bytes (plus Python lists of step strings) there.
//   rows    = P paths and W walks (W named sample#hap#seqid[:start-end]), sorted by name (stable)
//   columns = segments, decimal ids first in numeric order, then the others in string order (stable)
#include <fcntl.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <string>
#include <string_view>
#include <thread>
#include <unordered_map>
#include <vector>

#include "internal.h"

struct impop_gfa {
    std::vector<std::string> names;     // row order
    std::vector<uint64_t> bits;         // [n_path][words]
    uint64_t words = 0;
    std::vector<uint32_t> seg_len;      // column order
    std::vector<int64_t> pos;           // column order; empty without a reference path
    int64_t ref_row = -1;
};

namespace {

bool all_digits(std::string_view s) {
    if (s.empty()) return false;
    for (char c : s)
        if (c < '0' || c > '9') return false;
    return true;
}
std::string_view strip_zeros(std::string_view s) {
    size_t i = 0;
    while (i + 1 < s.size() && s[i] == '0') ++i;
    return s.substr(i);
}
// sorted(seg_len, key = (0, int(s)) if s.isdigit() else (1, s)); equal keys keep their insertion order
bool seg_less(std::string_view a, std::string_view b) {
    const bool da = all_digits(a), db = all_digits(b);
    if (da != db) return da;
    if (da) {
        const std::string_view x = strip_zeros(a), y = strip_zeros(b);
        if (x.size() != y.size()) return x.size() < y.size();
        return x < y;
    }
    return a < b;
}

struct PathLine {
    std::string name;
    std::string_view steps;  // P: "id+,id-,..."   W: ">id<id..."
    bool walk;
};

// field k (0-based) of a tab-separated line; false if absent
bool field(std::string_view line, int k, std::string_view *out) {
    size_t s = 0;
    for (int i = 0; i < k; ++i) {
        const size_t t = line.find('\t', s);
        if (t == std::string_view::npos) return false;
        s = t + 1;
    }
    const size_t t = line.find('\t', s);
    *out = line.substr(s, t == std::string_view::npos ? std::string_view::npos : t - s);
    return true;
}

template <typename F>
bool for_each_step(const PathLine &p, F &&f) {  // f(segment id) -> false aborts
    const std::string_view s = p.steps;
    if (!p.walk) {  // [x[:-1] for x in steps.split(",") if x]
        size_t a = 0;
        while (a <= s.size()) {
            size_t b = s.find(',', a);
            if (b == std::string_view::npos) b = s.size();
            if (b > a && !f(s.substr(a, b - a - 1))) return false;
            a = b + 1;
        }
    } else {  // ([<>])([^<>]+)
        size_t a = 0;
        while (a < s.size()) {
            if (s[a] != '<' && s[a] != '>') { ++a; continue; }
            size_t b = a + 1;
            while (b < s.size() && s[b] != '<' && s[b] != '>') ++b;
            if (b > a + 1 && !f(s.substr(a + 1, b - a - 1))) return false;
            a = b;
        }
    }
    return true;
}

}  // namespace

using namespace impop;

IMPOP_API int impop_gfa_parse(const char *path, const char *ref_prefix, impop_gfa **out) {
    REQUIRE(path && out, "impop_gfa_parse: NULL argument");
    *out = nullptr;
    const int fd = open(path, O_RDONLY);
    if (fd < 0) {
        set_error("impop_gfa_parse: cannot open %s", path);
        return IMPOP_E_INVALID;
    }
    struct stat st;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) {
        close(fd);
        set_error("impop_gfa_parse: not a regular file: %s", path);
        return IMPOP_E_UNSUPPORTED;
    }
    const size_t size = (size_t)st.st_size;
    const char *data = size ? (const char *)mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0) : "";
    close(fd);
    if (size && data == MAP_FAILED) {
        set_error("impop_gfa_parse: mmap failed for %s", path);
        return IMPOP_E_UNSUPPORTED;
    }
    auto done = [&](int code) {
        if (size) munmap((void *)data, size);
        return code;
    };
    // ---- one pass: segments (later S lines of the same id overwrite the length, the first one fixes the position in
    // insertion order), path / walk lines remembered by their step field
    std::unordered_map<std::string_view, uint32_t> seg_ix;  // id -> insertion index
    std::vector<std::string_view> seg_id;
    std::vector<int64_t> seg_len_ins;
    std::vector<PathLine> paths;
    seg_ix.reserve(1 << 16);
    size_t pos = 0;
    uint64_t line_no = 0;
    while (pos < size) {
        size_t e = pos;
        while (e < size && data[e] != '\n') ++e;
        std::string_view line(data + pos, e - pos);
        pos = e + 1;
        ++line_no;
        if (!line.empty() && line.back() == '\r') {  // CRLF: the Python reader keeps '\r' in the last field — let it decide
            set_error("impop_gfa_parse: CRLF line endings");
            return done(IMPOP_E_UNSUPPORTED);
        }
        if (line.empty() || (line[0] != 'S' && line[0] != 'P' && line[0] != 'W')) continue;
        std::string_view f0;
        if (!field(line, 0, &f0)) continue;
        if (f0 == "S") {
            std::string_view id, seq;
            if (!field(line, 1, &id) || !field(line, 2, &seq)) {
                set_error("impop_gfa_parse: short S line %llu", (unsigned long long)line_no);
                return done(IMPOP_E_INVALID);
            }
            int64_t L = seq == "*" ? 0 : (int64_t)seq.size();
            if (L > 0xFFFFFFFFll) {
                set_error("impop_gfa_parse: segment longer than 2^32 on line %llu", (unsigned long long)line_no);
                return done(IMPOP_E_UNSUPPORTED);
            }
            for (int k = 3;; ++k) {  // the last LN:i: tag wins
                std::string_view tag;
                if (!field(line, k, &tag)) break;
                if (tag.size() >= 5 && tag.substr(0, 5) == "LN:i:") {
                    if (tag.size() > 5 + 10) {  // more digits than a length below 2^32 has: the Python reader decides (no overflow here)
                        set_error("impop_gfa_parse: unusual LN tag on line %llu", (unsigned long long)line_no);
                        return done(IMPOP_E_UNSUPPORTED);
                    }
                    int64_t v = 0;
                    bool neg = false, okv = true;
                    size_t i = 5;
                    if (i < tag.size() && (tag[i] == '-' || tag[i] == '+')) { neg = tag[i] == '-'; ++i; }
                    if (i >= tag.size()) okv = false;
                    for (; i < tag.size() && okv; ++i) {
                        if (tag[i] < '0' || tag[i] > '9') okv = false;
                        else v = v * 10 + (tag[i] - '0');
                    }
                    if (!okv || v > 0xFFFFFFFFll) {  // anything int() might read differently, or a length the 32-bit weights cannot hold
                        set_error("impop_gfa_parse: unusual LN tag on line %llu", (unsigned long long)line_no);
                        return done(IMPOP_E_UNSUPPORTED);
                    }
                    L = neg ? -v : v;
                }
            }
            auto it = seg_ix.find(id);
            if (it == seg_ix.end()) {
                seg_ix.emplace(id, (uint32_t)seg_id.size());
                seg_id.push_back(id);
                seg_len_ins.push_back(L);
            } else {
                seg_len_ins[it->second] = L;
            }
        } else if (f0 == "P") {
            std::string_view name, steps;
            if (!field(line, 1, &name) || !field(line, 2, &steps)) {
                set_error("impop_gfa_parse: short P line %llu", (unsigned long long)line_no);
                return done(IMPOP_E_INVALID);
            }
            paths.push_back({std::string(name), steps, false});
        } else if (f0 == "W") {
            std::string_view sample, hap, seqid, s0, s1, walk;
            if (!field(line, 1, &sample) || !field(line, 2, &hap) || !field(line, 3, &seqid) || !field(line, 4, &s0) ||
                !field(line, 5, &s1) || !field(line, 6, &walk)) {
                set_error("impop_gfa_parse: short W line %llu", (unsigned long long)line_no);
                return done(IMPOP_E_INVALID);
            }
            std::string nm = std::string(sample) + "#" + std::string(hap) + "#" + std::string(seqid);
            if (s0 != "*" && s1 != "*") nm += ":" + std::string(s0) + "-" + std::string(s1);
            paths.push_back({std::move(nm), walk, true});
        }
    }
    // ---- column order and row order
    const uint64_t n_seg = seg_id.size();
    std::vector<uint32_t> order(n_seg);
    for (uint64_t i = 0; i < n_seg; ++i) order[i] = (uint32_t)i;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return seg_less(seg_id[a], seg_id[b]); });
    std::vector<uint32_t> col_of_ins(n_seg);
    for (uint64_t c = 0; c < n_seg; ++c) col_of_ins[order[c]] = (uint32_t)c;
    std::stable_sort(paths.begin(), paths.end(), [](const PathLine &a, const PathLine &b) { return a.name < b.name; });
    impop_gfa *G = new impop_gfa();
    const uint64_t n_path = paths.size();
    G->words = std::max<uint64_t>((n_seg + 63) / 64, 1);
    G->bits.assign(n_path * G->words, 0ull);
    G->seg_len.resize(n_seg);
    for (uint64_t c = 0; c < n_seg; ++c) G->seg_len[c] = (uint32_t)std::max<int64_t>(seg_len_ins[order[c]], 0);
    G->names.reserve(n_path);
    // odgi / impg graphs number their segments 1..N: a direct table id -> column then replaces a string hash per step
    // (only when every id is a plain decimal without leading zeros, so that number == string identity)
    std::vector<uint32_t> direct;
    {
        uint64_t max_id = 0;
        bool plain = n_seg > 0;
        for (uint64_t i = 0; i < n_seg && plain; ++i) {
            const std::string_view id = seg_id[i];
            plain = all_digits(id) && id.size() <= 9 && (id.size() == 1 || id[0] != '0');
            if (plain) max_id = std::max<uint64_t>(max_id, strtoull(std::string(id).c_str(), nullptr, 10));
        }
        if (plain && max_id <= 16 * n_seg + 1024) {
            direct.assign(max_id + 1, 0xFFFFFFFFu);
            for (uint64_t i = 0; i < n_seg; ++i) direct[strtoull(std::string(seg_id[i]).c_str(), nullptr, 10)] = col_of_ins[i];
        }
    }
    auto column_of = [&](std::string_view id) -> uint32_t {  // 0xFFFFFFFF: no such segment
        if (!direct.empty()) {
            if (id.empty() || id.size() > 9 || (id.size() > 1 && id[0] == '0')) return 0xFFFFFFFFu;
            uint64_t v = 0;
            for (char ch : id) {
                if (ch < '0' || ch > '9') return 0xFFFFFFFFu;
                v = v * 10 + (uint64_t)(ch - '0');
            }
            return v < direct.size() ? direct[v] : 0xFFFFFFFFu;
        }
        auto it = seg_ix.find(id);
        return it == seg_ix.end() ? 0xFFFFFFFFu : col_of_ins[it->second];
    };
    for (uint64_t r = 0; r < n_path; ++r) {
        G->names.push_back(paths[r].name);
        uint64_t *row = G->bits.data() + r * G->words;
        std::string_view missing;
        const bool ok = for_each_step(paths[r], [&](std::string_view id) {
            const uint32_t c = column_of(id);
            if (c == 0xFFFFFFFFu) { missing = id; return false; }
            row[c >> 6] |= 1ull << (c & 63);
            return true;
        });
        if (!ok) {
            set_error("impop_gfa_parse: path %s steps on segment '%.*s' that has no S line", paths[r].name.c_str(), (int)missing.size(),
                      missing.data());
            delete G;
            return done(IMPOP_E_INVALID);
        }
    }
    // ---- reference coordinates: the first path (row order) whose name starts with ref_prefix
    if (ref_prefix) {
        const std::string_view pre(ref_prefix);
        for (uint64_t r = 0; r < n_path && G->ref_row < 0; ++r)
            if (std::string_view(paths[r].name).substr(0, pre.size()) == pre) G->ref_row = (int64_t)r;
        if (G->ref_row < 0) {
            set_error("no path starts with '%s'", ref_prefix);
            delete G;
            return done(IMPOP_E_INVALID);
        }
        const std::string &rn = paths[(size_t)G->ref_row].name;
        int64_t ref_start = 0;
        {  // re.search(r":(\d+)-(\d+)$", name)
            const size_t dash = rn.rfind('-');
            const size_t colon = dash == std::string::npos ? std::string::npos : rn.rfind(':', dash);
            if (dash != std::string::npos && colon != std::string::npos && colon + 1 < dash && dash + 1 < rn.size() &&
                all_digits(std::string_view(rn).substr(colon + 1, dash - colon - 1)) && all_digits(std::string_view(rn).substr(dash + 1)))
                ref_start = strtoll(rn.c_str() + colon + 1, nullptr, 10);
        }
        if (ref_start > (int64_t)1 << 60) {  // coordinates are summed below: keep them far from the end of int64
            set_error("impop_gfa_parse: reference start beyond 2^60");
            delete G;
            return done(IMPOP_E_UNSUPPORTED);
        }
        G->pos.assign(n_seg, -1);
        int64_t off = ref_start;
        for_each_step(paths[(size_t)G->ref_row], [&](std::string_view id) {
            const uint32_t ins = seg_ix.find(id)->second, c = col_of_ins[ins];
            if (G->pos[c] < 0) G->pos[c] = off;
            off += seg_len_ins[ins];
            return true;
        });
        int64_t last = ref_start, run = INT64_MIN;
        for (uint64_t c = 0; c < n_seg; ++c) {  // off-reference columns inherit the preceding reference coordinate ...
            if (G->pos[c] < 0) G->pos[c] = last;
            else last = G->pos[c];
            run = std::max(run, G->pos[c]);  // ... and the coordinates are made non-decreasing (np.maximum.accumulate)
            G->pos[c] = run;
        }
    }
    *out = G;
    return done(IMPOP_OK);
}

// `odgi paths -H` table (scripts/wip/op-afs.py:112 reads the same shape): a header row, three metadata columns (path.name,
// path.length, node.count), then one column per node holding 0 / visit counts.  Presence = the field is neither "0" nor
// empty; one site per node (all lengths 1, no coordinates); rows sorted by name (stable).  Same rules as
// impop_amd/extract.py:from_paths_table, which the caller falls back to on ANY non-zero status.  Rows are independent:
// line starts are found first, then the rows are parsed by a few threads straight into their bit rows (a 465 x 10^6
// table is 0.93 GB of text and 58 MB of bits; the Python reader builds 4.6 x 10^8 Python strings on the way).
IMPOP_API int impop_paths_table_parse(const char *path, impop_gfa **out) {
    REQUIRE(path && out, "impop_paths_table_parse: NULL argument");
    *out = nullptr;
    const int fd = open(path, O_RDONLY);
    if (fd < 0) {
        set_error("impop_paths_table_parse: cannot open %s", path);
        return IMPOP_E_INVALID;
    }
    struct stat st;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) {
        close(fd);
        set_error("impop_paths_table_parse: not a regular file: %s", path);
        return IMPOP_E_UNSUPPORTED;
    }
    const size_t size = (size_t)st.st_size;
    // MAP_POPULATE: the rows are parsed by several threads, and first-touch page faults of one address space serialise
    const char *data = size ? (const char *)mmap(nullptr, size, PROT_READ, MAP_PRIVATE | MAP_POPULATE, fd, 0) : "";
    close(fd);
    if (size && data == MAP_FAILED) {
        set_error("impop_paths_table_parse: mmap failed for %s", path);
        return IMPOP_E_UNSUPPORTED;
    }
    auto done = [&](int code) {
        if (size) munmap((void *)data, size);
        return code;
    };
    // Python reads in text mode: "\r\n" and a lone "\r" end a line there.  Leave such files to it.
    if (size && memchr(data, '\r', size)) {
        set_error("impop_paths_table_parse: carriage returns in %s (left to the Python reader)", path);
        return done(IMPOP_E_UNSUPPORTED);
    }
    // header: n_node = fields - 3
    const char *end = data + size;
    const char *nl = size ? (const char *)memchr(data, '\n', size) : nullptr;
    const char *hdr_end = nl ? nl : end;
    uint64_t hdr_fields = 1;
    for (const char *q = data; q < hdr_end; ++q) hdr_fields += *q == '\t';
    if (hdr_fields < 4) {
        set_error("impop_paths_table_parse: %s: expected >= 4 tab-separated columns", path);
        return done(IMPOP_E_INVALID);
    }
    const uint64_t n_node = hdr_fields - 3;
    // line starts of the non-empty data rows
    std::vector<std::pair<const char *, const char *>> lines;
    for (const char *q = nl ? nl + 1 : end; q < end;) {
        const char *e = (const char *)memchr(q, '\n', (size_t)(end - q));
        if (!e) e = end;
        if (e > q) lines.emplace_back(q, e);
        q = e + 1;
    }
    const size_t n = lines.size();
    auto *G = new impop_gfa();
    G->words = std::max<uint64_t>((n_node + 63) / 64, 1);
    std::vector<std::string> names(n);
    std::vector<uint64_t> bits(n * G->words, 0);
    std::atomic<size_t> next{0};
    std::atomic<int> bad{0};
    auto work = [&]() {
        for (;;) {
            const size_t r = next.fetch_add(1);
            if (r >= n || bad.load(std::memory_order_relaxed)) return;
            const char *q = lines[r].first, *e = lines[r].second;
            // three metadata fields
            const char *f0 = q;
            int meta = 0;
            const char *name_end = nullptr;
            while (q < e && meta < 3) {
                if (*q == '\t') {
                    if (meta == 0) name_end = q;
                    ++meta;
                }
                ++q;
            }
            if (meta < 3) { bad = 1; return; }
            names[r].assign(f0, (size_t)(name_end - f0));
            uint64_t *row = bits.data() + r * G->words;
            uint64_t col = 0;
            // node fields.  The common field is one character ("0" / "1"): recognised by the tab (or line end) right
            // behind it, its presence added without a branch on the DATA (0 / 1 columns are coin flips to a branch
            // predictor: 8 ns per field with one, 1.5 ns without); longer fields are present, empty ones absent
            uint64_t word = 0;
            while (true) {
                if (col >= n_node) { bad = 1; return; }
                uint64_t present;
                if (q < e && *q != '\t' && (q + 1 == e || q[1] == '\t')) {
                    present = *q != '0';
                    q += 1;
                } else {
                    const char *t = q;
                    while (t < e && *t != '\t') ++t;
                    present = t != q;  // length >= 2 (a one-character field went the other way), or empty
                    q = t;
                }
                word |= present << (col & 63);
                ++col;
                if ((col & 63) == 0) { row[(col >> 6) - 1] = word; word = 0; }
                if (q >= e) break;
                ++q;  // the tab
                if (q == e) {  // a trailing tab: one more, empty, field
                    if (col >= n_node) { bad = 1; return; }
                    ++col;
                    if ((col & 63) == 0) { row[(col >> 6) - 1] = word; word = 0; }
                    break;
                }
            }
            if (col & 63) row[col >> 6] = word;
            if (col != n_node) { bad = 1; return; }
        }
    };
    {
        unsigned hw = std::thread::hardware_concurrency();
        const unsigned nt = (unsigned)std::min<size_t>(std::max<size_t>(n, 1), std::min<unsigned>(hw ? hw : 4, 16));
        std::vector<std::thread> th;
        for (unsigned k = 1; k < nt; ++k) th.emplace_back(work);
        work();
        for (auto &t : th) t.join();
    }
    if (bad) {
        delete G;
        set_error("impop_paths_table_parse: %s: a row's field count differs from the header's", path);
        return done(IMPOP_E_INVALID);
    }
    // engine convention: lexicographic name order, stable (sorted(range(n), key = name))
    std::vector<size_t> order(n);
    for (size_t i = 0; i < n; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return names[a] < names[b]; });
    G->names.resize(n);
    G->bits.assign(n * G->words, 0);
    for (size_t i = 0; i < n; ++i) {
        G->names[i] = std::move(names[order[i]]);
        memcpy(G->bits.data() + i * G->words, bits.data() + order[i] * G->words, G->words * 8);
    }
    G->seg_len.assign(n_node, 1u);
    *out = G;
    return done(IMPOP_OK);
}

IMPOP_API int impop_gfa_info(const impop_gfa *g, uint32_t *n_path, uint64_t *n_seg, uint64_t *names_bytes, int64_t *ref_row) {
    REQUIRE(g, "impop_gfa_info: handle is NULL");
    if (n_path) *n_path = (uint32_t)g->names.size();
    if (n_seg) *n_seg = g->seg_len.size();
    if (names_bytes) {
        uint64_t b = 0;
        for (auto &x : g->names) b += x.size() + 1;
        *names_bytes = b;
    }
    if (ref_row) *ref_row = g->ref_row;
    return IMPOP_OK;
}

IMPOP_API int impop_gfa_names(const impop_gfa *g, char *buf) {
    REQUIRE(g && buf, "impop_gfa_names: NULL argument");
    for (auto &x : g->names) {
        memcpy(buf, x.data(), x.size());
        buf[x.size()] = 0;
        buf += x.size() + 1;
    }
    return IMPOP_OK;
}

IMPOP_API int impop_gfa_bits(const impop_gfa *g, uint64_t *bits_hap_major, uint64_t row_stride_words) {
    REQUIRE(g, "impop_gfa_bits: handle is NULL");
    REQUIRE(g->names.empty() || bits_hap_major, "impop_gfa_bits: out is NULL");
    REQUIRE(row_stride_words >= g->words, "impop_gfa_bits: row_stride_words too small");
    for (size_t r = 0; r < g->names.size(); ++r) memcpy(bits_hap_major + r * row_stride_words, g->bits.data() + r * g->words, g->words * 8);
    return IMPOP_OK;
}

IMPOP_API int impop_gfa_lengths(const impop_gfa *g, uint32_t *lengths) {
    REQUIRE(g, "impop_gfa_lengths: handle is NULL");
    REQUIRE(g->seg_len.empty() || lengths, "impop_gfa_lengths: out is NULL");
    if (!g->seg_len.empty()) memcpy(lengths, g->seg_len.data(), g->seg_len.size() * 4);
    return IMPOP_OK;
}

IMPOP_API int impop_gfa_positions(const impop_gfa *g, int64_t *positions) {
    REQUIRE(g, "impop_gfa_positions: handle is NULL");
    REQUIRE(!g->pos.empty() || g->seg_len.empty(), "impop_gfa_positions: parsed without a reference prefix");
    REQUIRE(g->pos.empty() || positions, "impop_gfa_positions: out is NULL");
    if (!g->pos.empty()) memcpy(positions, g->pos.data(), g->pos.size() * 8);
    return IMPOP_OK;
}

IMPOP_API int impop_gfa_free(impop_gfa *g) {
    delete g;
    return IMPOP_OK;
}
