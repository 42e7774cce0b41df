// parsers_fuzz.cc — CPU-only sanitizer + fuzz target for the host-side text parsers of libimpop_hip.so
// (csrc/simparse.hip: the replacement of pica2.read_similarity_file, pica2.py:6-58, and h-fst.read_similarity_file,
// h-fst.py:84-119; csrc/gfaparse.hip: GFA and `odgi paths -H` readers).  tests/test_parser_fuzz.py compiles the two .hip
// files AS C++ (they hold no device code) together with this driver with g++ -fsanitize=address,undefined and runs it.
//
// Contract under test: whatever bytes a file holds, a parse call returns IMPOP_OK or an IMPOP_E_* code (on which the
// Python mirror takes over with the reference's own messages) — never a crash, an out-of-bounds access, a leak or
// undefined behaviour; and after IMPOP_OK every accessor works on buffers of exactly the advertised sizes.
// Inputs: hand-written corner cases (0-byte file, no trailing newline, NUL bytes, truncated rows, duplicate headers,
// huge numeric fields, a 10^6-column row, ...) and deterministic random mutations of valid files.
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <string>
#include <vector>

#include "internal.h"

namespace impop {  // the two symbols of context.hip the parsers use
static thread_local char g_err[1024];
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}
int hip_fail(hipError_t, const char *what, const char *, int) {
    set_error("hip: %s", what);
    return IMPOP_E_HIP;
}
}  // namespace impop

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() {
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return rng_state;
}
static std::string g_path;
static unsigned long g_ok[3], g_err_n[3], g_cases;

static void write_file(const std::string &data) {
    FILE *f = fopen(g_path.c_str(), "wb");
    if (!f) { perror("fopen"); exit(2); }
    if (!data.empty() && fwrite(data.data(), 1, data.size(), f) != data.size()) { perror("fwrite"); exit(2); }
    fclose(f);
}
static bool status_ok(int rc) { return rc == IMPOP_OK || (rc <= IMPOP_E_INVALID && rc >= IMPOP_E_INTERNAL); }

static void run_sim(const std::string &data) {
    write_file(data);
    for (int flavor = 0; flavor < 2; ++flavor) {
        impop_sim *s = nullptr;
        const int rc = impop_sim_parse(g_path.c_str(), flavor, &s);
        if (!status_ok(rc)) { fprintf(stderr, "impop_sim_parse: unexpected status %d\n", rc); exit(1); }
        if (rc != IMPOP_OK) {
            if (s) { fprintf(stderr, "impop_sim_parse: handle returned with an error status\n"); exit(1); }
            ++g_err_n[0];
            continue;
        }
        ++g_ok[0];
        uint32_t n = 0;
        uint64_t rows = 0, nb = 0, n_bad = 0;
        int64_t bad_line = -1;
        if (impop_sim_info(s, &n, &rows, &nb, &bad_line, &n_bad) != IMPOP_OK) exit(1);
        std::vector<char> names(nb ? nb : 1);
        std::vector<uint32_t> fs(n ? n : 1);
        std::vector<double> dense((size_t)n * n ? (size_t)n * n : 1);
        char bad[64];
        impop_sim_names(s, names.data());
        impop_sim_first_seen(s, fs.data());
        impop_sim_bad_text(s, bad, sizeof bad);
        if (n <= 2000) impop_sim_dense(s, dense.data());
        for (uint32_t k = 0; k < n; ++k)
            if (fs[k] >= n) { fprintf(stderr, "first_seen out of range\n"); exit(1); }
        impop_sim_free(s);
    }
}

static void run_gfa_handle(impop_gfa *g, bool with_pos) {
    uint32_t np = 0;
    uint64_t ns = 0, nb = 0;
    int64_t ref_row = -1;
    if (impop_gfa_info(g, &np, &ns, &nb, &ref_row) != IMPOP_OK) exit(1);
    std::vector<char> names(nb ? nb : 1);
    const uint64_t words = (ns + 63) / 64 ? (ns + 63) / 64 : 1;
    std::vector<uint64_t> bits((size_t)np * words ? (size_t)np * words : 1);
    std::vector<uint32_t> len(ns ? ns : 1);
    std::vector<int64_t> pos(ns ? ns : 1);
    impop_gfa_names(g, names.data());
    impop_gfa_bits(g, bits.data(), words);
    impop_gfa_lengths(g, len.data());
    if (with_pos) impop_gfa_positions(g, pos.data());
    impop_gfa_free(g);
}
static void run_gfa(const std::string &data) {
    write_file(data);
    for (int k = 0; k < 2; ++k) {
        impop_gfa *g = nullptr;
        const int rc = impop_gfa_parse(g_path.c_str(), k ? "CHM13#0#" : nullptr, &g);
        if (!status_ok(rc)) { fprintf(stderr, "impop_gfa_parse: unexpected status %d\n", rc); exit(1); }
        if (rc != IMPOP_OK) { if (g) exit(1); ++g_err_n[1]; continue; }
        ++g_ok[1];
        run_gfa_handle(g, k == 1);
    }
}
static void run_table(const std::string &data) {
    write_file(data);
    impop_gfa *g = nullptr;
    const int rc = impop_paths_table_parse(g_path.c_str(), &g);
    if (!status_ok(rc)) { fprintf(stderr, "impop_paths_table_parse: unexpected status %d\n", rc); exit(1); }
    if (rc != IMPOP_OK) { if (g) exit(1); ++g_err_n[2]; return; }
    ++g_ok[2];
    run_gfa_handle(g, false);
}

static std::string mutate(const std::string &src) {
    std::string s = src;
    const int n_mut = 1 + (int)(rnd() % 4);
    static const char *tokens[] = {"\t", "\n", "\r\n", "\"", "0x1p3", "1_0", "nan", "-inf", "1e999", "1e-999", ":", "#", ",", "+", "-", ">", "<",
                                   "*", "LN:i:", "LN:i:-5", "LN:i:99999999999999999999", "P\t", "W\t", "S\t", "L\t", "group.a", "estimated.identity",
                                   "path.name", "18446744073709551616", "-1", "4294967296", "\t\t\t", " "};
    for (int k = 0; k < n_mut; ++k) {
        const size_t len = s.size();
        switch (rnd() % 9) {
            case 0: if (len) s[rnd() % len] ^= (char)(1u << (rnd() % 8)); break;                     // bit flip
            case 1: if (len) s.resize(rnd() % len); break;                                            // truncate
            case 2: if (len) s[rnd() % len] = (char)0; break;                                         // NUL byte
            case 3: s.insert(len ? rnd() % len : 0, tokens[rnd() % (sizeof tokens / sizeof *tokens)]); break;
            case 4: if (len) { const size_t a = rnd() % len, b = rnd() % (len - a) + 1; s.erase(a, b > 64 ? 64 : b); } break;  // delete a run
            case 5: if (len) { const size_t a = rnd() % len, c = rnd() % (len - a) + 1; s.insert(rnd() % len, s.substr(a, c > 200 ? 200 : c)); } break;  // splice
            case 6: if (len) s[rnd() % len] = '\n'; break;
            case 7: if (len) s[rnd() % len] = '\t'; break;
            default: if (len && s.back() == '\n') s.pop_back(); break;                                // missing trailing newline
        }
    }
    return s;
}

int main(int argc, char **argv) {
    const double budget_s = argc > 1 ? atof(argv[1]) : 10.0;
    const unsigned long min_cases = argc > 2 ? strtoul(argv[2], nullptr, 10) : 0;
    char tmpl[] = "/tmp/impop_fuzz_XXXXXX";
    const int fd = mkstemp(tmpl);
    if (fd < 0) { perror("mkstemp"); return 2; }
    close(fd);
    g_path = tmpl;

    // ---- valid seeds -------------------------------------------------------------------------------------------------
    std::string sim = "group.a\tgroup.b\tgroup.a.length\tgroup.b.length\tintersection\testimated.identity\n";
    const char *nm[] = {"S0#1#chr1:0-100", "S0#2#chr1:0-100", "S1#1#chr1:0-100", "S1#2#chr1:0-100", "S2#1#chr1:0-100"};
    for (int i = 0; i < 5; ++i)
        for (int j = 0; j < 5; ++j) {
            char row[256];
            snprintf(row, sizeof row, "%s\t%s\t100\t100\t%d\t%.17g\n", nm[i], nm[j], 90 + i, i == j ? 1.0 : 0.99 + 0.001 * ((i * 7 + j * 3) % 10));
            sim += row;
        }
    const std::string gfa =
        "H\tVN:Z:1.0\nS\t1\tACGT\nS\t2\tA\nS\t3\tG\nS\t4\tTTTTT\nS\t5\t*\tLN:i:3\nL\t1\t+\t2\t+\t0M\n"
        "P\tCHM13#0#chr1:1000-1013\t1+,2+,4+,5+\t*\nP\tHG002#1#ctgA:0-13\t1+,3+,4+,5+\t*\nW\tHG003\t2\tctgB\t0\t9\t>1>2>4\n"
        "P\tHG002#2#ctgC:5-10\t4-,2-\t*\n";
    std::string table = "path.name\tpath.length\tnode.count";
    for (int c = 1; c <= 40; ++c) table += "\tnode." + std::to_string(c);
    table += "\n";
    for (int r = 0; r < 9; ++r) {
        table += "S" + std::to_string(r) + "#1#x\t" + std::to_string(100 + r) + "\t7";
        for (int c = 0; c < 40; ++c) table += ((r * 31 + c * 17) % 5 < 2) ? "\t1" : "\t0";
        table += "\n";
    }

    // ---- hand-written corner cases -----------------------------------------------------------------------------------
    std::vector<std::string> corner = {
        "", "\n", "\n\n\n", "\t", std::string("\0", 1), std::string(4096, '\0'), std::string(70000, '\t'), std::string(70000, '\n'),
        "group.a\tgroup.b\testimated.identity", "group.a\tgroup.b\testimated.identity\n", "group.a\tgroup.b\testimated.identity\na\tb",
        "group.a\tgroup.b\testimated.identity\na\tb\t", "group.a\tgroup.b\testimated.identity\na\tb\t0.5", "group.a\tgroup.b\testimated.identity\na\tb\tzzz\n",
        "group.a\tgroup.b\testimated.identity\na\tb\t1e999\nb\tc\t-1e999\nc\td\t1e-999\n", "group.a\tgroup.b\testimated.identity\na\tb\t" + std::string(5000, '9') + "\n",
        "group.a\tgroup.b\testimated.identity\na\tb\t0." + std::string(100000, '3') + "\n", "group.a\tgroup.a\testimated.identity\na\tb\t0.5\n",
        "group.a\tgroup.b\testimated.identity\ngroup.a\tgroup.b\testimated.identity\na\tb\t0.5\n", "estimated.identity\tgroup.b\tgroup.a\n0.5\tb\ta\n",
        "group.a\tgroup.b\testimated.identity\n\"a\"\tb\t0.5\n", "group.a\tgroup.b\testimated.identity\r\na\tb\t0.5\r\n",
        "group.a\tgroup.b\testimated.identity\na\tb\t0.5\n\n\nc\td\t0.25\n", "group.a\tgroup.b\testimated.identity\n\t\t\n", "group.a\tgroup.b\testimated.identity\n" + std::string(300000, 'a') + "\tb\t0.5\n",
        std::string("group.a\tgroup.b\testimated.identity\na\0c\tb\t0.5\n", 44), "a\tb\tc\nx\ty\t0.5\n",
        "H\n", "S\n", "S\t\n", "S\t1\n", "S\t1\t\n", "S\t1\t*\n", "S\t1\t*\tLN:i:\n", "S\t1\t*\tLN:i:-3\n", "S\t1\t*\tLN:i:99999999999999999999999\n", "S\t1\tA\nP\n", "S\t1\tA\nP\tx\n",
        "S\t1\tA\nP\tx\t\n", "S\t1\tA\nP\tx\t1\n", "S\t1\tA\nP\tx\t1+,\n", "S\t1\tA\nP\tx\t,,,,\n", "S\t1\tA\nP\tx\t2+\n", "S\t1\tA\nP\tx\t1+,1+,1-\t*\nP\tx\t1+\t*\n", "S\t1\tA\nS\t1\tC\nP\tx\t1+\t*\n",
        "S\t1\tA\nW\n", "S\t1\tA\nW\ts\n", "S\t1\tA\nW\ts\t1\tc\t0\t1\n", "S\t1\tA\nW\ts\t1\tc\t0\t1\t\n", "S\t1\tA\nW\ts\t1\tc\t0\t1\t>\n", "S\t1\tA\nW\ts\t1\tc\t0\t1\t>1<1>\n", "S\t1\tA\nW\ts\tx\tc\ty\tz\t>1\n",
        "S\t18446744073709551615\tA\nS\t18446744073709551616\tC\nP\tx\t18446744073709551615+\t*\n", "S\t0\tA\nS\t00\tC\nS\t-1\tG\nP\tx\t0+,00+,-1+\t*\n",
        "S\t1\tA\nP\tCHM13#0#c:9999999999999999999999-5\t1+\t*\n", "S\t1\tA\nP\tCHM13#0#c:-\t1+\t*\n", "S\t1\tA\nP\tCHM13#0#c:5\t1+\t*\n", "S\t1\tA\nP\tCHM13#0#:\t1+\t*\n",
        "path.name\tpath.length\tnode.count\n", "path.name\tpath.length\tnode.count", "path.name\tpath.length\tnode.count\tn1\n", "path.name\tpath.length\tnode.count\tn1\na\t1\t1\t1",
        "path.name\tpath.length\tnode.count\tn1\na\t1\t1\n", "path.name\tpath.length\tnode.count\tn1\na\t1\t1\t1\t1\n", "path.name\tpath.length\tnode.count\tn1\na\t1\t1\t\n",
        "path.name\tpath.length\tnode.count\tn1\tn2\na\t1\t1\t99999999999999999999\t-1\n", "path.name\tpath.length\tnode.count\tn1\na\t1\t1\tx\n", "path.name\tpath.length\tnode.count\tn1\na\t1\t1\t1\na\t1\t1\t0\n",
        "path.name\tpath.length\tnode.count\tn1\n\n\na\t1\t1\t1\n\n", "a\tb\n1\t2\n",
    };
    {   // one row of 10^6 columns (header + one data row), and a 10^6-column .sim header
        std::string wide = "path.name\tpath.length\tnode.count";
        for (int c = 0; c < 1000000; ++c) wide += "\tn";
        wide += "\np\t1\t1";
        for (int c = 0; c < 1000000; ++c) wide += (c % 3) ? "\t0" : "\t1";
        corner.push_back(wide + "\n");
        corner.push_back(wide);                             // ... without the trailing newline
        corner.push_back(wide.substr(0, wide.size() - 7));  // ... truncated inside the row
        std::string wsim = "group.a\tgroup.b";
        for (int c = 0; c < 1000000; ++c) wsim += "\tx";
        wsim += "\testimated.identity\na\tb";
        for (int c = 0; c < 1000000; ++c) wsim += "\t1";
        corner.push_back(wsim + "\t0.5\n");
        corner.push_back(wsim);
    }
    for (const std::string &c : corner) {
        run_sim(c);
        run_gfa(c);
        run_table(c);
        g_cases += 3;
    }
    // every prefix of the valid files (truncation at each byte)
    for (size_t k = 0; k <= sim.size(); k += 3) { run_sim(sim.substr(0, k)); ++g_cases; }
    for (size_t k = 0; k <= gfa.size(); ++k) { run_gfa(gfa.substr(0, k)); ++g_cases; }
    for (size_t k = 0; k <= table.size(); k += 2) { run_table(table.substr(0, k)); ++g_cases; }
    // ---- random mutations until the time budget is used ----------------------------------------------------------------
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (;;) {
        for (int k = 0; k < 50; ++k) {
            run_sim(mutate(sim));
            run_gfa(mutate(gfa));
            run_table(mutate(table));
            g_cases += 3;
        }
        clock_gettime(CLOCK_MONOTONIC, &t1);
        if ((t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec) > budget_s && g_cases >= min_cases) break;
    }
    unlink(g_path.c_str());
    printf("parsers_fuzz ok: %lu cases; sim ok/declined %lu/%lu, gfa %lu/%lu, table %lu/%lu\n", g_cases, g_ok[0], g_err_n[0], g_ok[1], g_err_n[1],
           g_ok[2], g_err_n[2]);
    return 0;
}
