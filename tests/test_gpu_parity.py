"""GPU parity tests (run with -m gpu on an MI355X): the HIP engine, called through the C ABI,
against the CPU oracle on the same seeded inputs and against the goldens captured from the
real reference.  Integers bit-exact; pi / Fst / D within 1e-9 relative (north_star)."""
import random

import numpy as np
import pytest

from conftest import fh, golden_bits, golden_counts, load_golden, rel_close, stat_close
from synth_ref import synth_matrix

pytestmark = pytest.mark.gpu

REL = 1e-9  # tolerance stated by BASELINE.json north_star for pi / Fst / D
INT_KEYS = ("n_sites", "s_all", "s_p", "s_a", "s_b", "sum_p", "sum_a", "sum_b", "sum_ab")
DBL_KEYS = ("pi", "pi_site", "pi_a", "pi_b", "pi_xy", "dxy", "da", "fst", "tajima_d")


@pytest.fixture(scope="module")
def ctx():
    import impop_amd
    c = impop_amd.Context(0)
    assert c.device_name().startswith("gfx950")
    yield c
    c.close()


def founder_matrix(rng, n, W, nf=6, pf=0.01, pp=0.002):
    anc = rng.integers(0, 2, size=W, dtype=np.uint8)
    f = np.repeat(anc[None, :], nf, axis=0) ^ (rng.random((nf, W)) < pf).astype(np.uint8)
    m = f[rng.integers(0, nf, size=n)].copy()
    if pp > 0:
        m ^= (rng.random((n, W)) < pp).astype(np.uint8)
    return m


def check_record(got, want, where):
    for k in INT_KEYS:
        assert int(got[k]) == int(want[k]), (where, k, int(got[k]), int(want[k]))
    for k in DBL_KEYS:
        assert rel_close(float(got[k]), float(want[k]), REL, 1e-300), (where, k, float(got[k]), float(want[k]))


def test_py_round_device_matches_cpython(ctx):
    rnd = random.Random(3)
    vals = [0.999985, 0.999975, 0.99999499999, 0.5, 1.5, 2.5, 0.125, 1e-9, 0.0, 1.0, 0.99995, 0.9995, 2.675,
            1.0000000000000002, 2.093456789e-05, 123456.7890125, -0.5, -2.675, 5e-324, 1e300, 4503599627370497.0]
    for _ in range(60000):
        k = rnd.randint(0, 3)
        if k == 0:
            vals.append(rnd.random())
        elif k == 1:
            vals.append(1.0 - rnd.random() * 1e-3)
        elif k == 2:
            vals.append(rnd.randint(0, 10 ** 6) / 10 ** rnd.randint(1, 7) + rnd.choice([0, 5e-7, 5e-6, 5e-9]))
        else:
            vals.append(rnd.random() * 10 ** rnd.randint(-10, 3))
    x = np.array(vals)
    for nd in (0, 2, 3, 5, 8):
        got = ctx.py_round(x, nd)
        want = np.array([round(float(v), nd) for v in x])
        bad = np.nonzero(~((got == want) | (np.isnan(got) & np.isnan(want))))[0]
        assert bad.size == 0, (nd, x[bad[:5]], got[bad[:5]], want[bad[:5]])


def test_tajima_golden_device(ctx):
    g = load_golden("tajima.json")
    n = [c["n"] for c in g["cases"]]
    S = [fh(c["S"]) for c in g["cases"]]
    pi = [fh(c["pi"]) for c in g["cases"]]
    D, comps = ctx.tajimas_d(n, S, pi, components=True)
    for i, c in enumerate(g["cases"]):
        assert rel_close(float(D[i]), fh(c["D"]), 1e-12), (c, D[i])
        for got, w in zip(comps[i], c["comps"]):
            assert rel_close(float(got), fh(w), 1e-12)
    from impop_amd import tj_d
    for e in g["errors"]:
        with pytest.raises(ValueError) as ei:
            tj_d.tajimas_d(e["n"], e["S"], e["pi"], ctx=ctx)
        assert str(ei.value) == e["error"]
    assert tj_d.tajimas_d(446, 20.0, 0.59146123, ctx=ctx) == pytest.approx(-1.9926482274156396, rel=1e-13)


def test_roundtrip_upload_download(ctx):
    rng = np.random.default_rng(11)
    import impop_amd
    for n, W in ((1, 1), (5, 63), (33, 64), (130, 777), (465, 1000), (600, 130)):
        m = rng.integers(0, 2, size=(n, W), dtype=np.uint8)
        bm = ctx.upload_dense(m)
        assert (impop_amd.unpack_hap_major(bm.download(), W) == m).all()
        a, b = W // 3, W - W // 5
        assert (impop_amd.unpack_hap_major(bm.download(a, b), b - a) == m[:, a:b]).all()
        cnt = bm.site_counts(0, W)
        assert (cnt == m.sum(0)).all()
        bm.free()


def test_scan_vs_oracle_and_golden(ctx, oracle):
    g = load_golden("bitmatrix.json")
    for mrec in g["matrices"]:
        n, W, L = mrec["n"], mrec["W"], mrec["L"]
        bits = golden_bits(mrec)
        bm = ctx.upload(bits, W)
        inA, inB = np.array(mrec["in_a"], np.uint8), np.array(mrec["in_b"], np.uint8)
        ones = np.ones(n, np.uint8)
        wins = [(0, W, L), (0, W // 2, L), (W // 3, W, 0), (5, 5, 10), (W - 1, W, 1)]
        got = bm.scan(wins, None, inA, inB)
        for (s0, s1, sl), r in zip(wins, got):
            want = oracle.window_allpairs(bits, n, s0, s1, oracle.pack_mask(ones), oracle.pack_mask(inA),
                                          oracle.pack_mask(inB), sl)
            check_record(r, want, (mrec["name"], s0, s1))
        # against the real reference's numbers
        r = got[0]
        ref_p = [c for c in mrec["kinds"]["match"]["pica2"] if fh(c["threshold"]) == 1.0 and c["round"] is None and c["L"] == L][0]
        assert rel_close(float(r["pi"]), fh(ref_p["pi"]), REL) and rel_close(float(r["pi_site"]), fh(ref_p["pi_site"]), REL)
        ref_h = [c for c in mrec["kinds"]["match"]["hfst"] if c["L"] == L and c["round"] is None][0]["out"]
        for k in ("fst", "pi_a", "pi_b", "pi_xy", "dxy", "da"):
            assert rel_close(float(r[k]), fh(ref_h[k]), REL, 1e-300), (mrec["name"], k)
        assert int(r["s_all"]) == mrec["S_all"]
        assert rel_close(float(r["tajima_d"]), fh(mrec["tajd_chain"]["D"]), REL)
        # overlapping populations are removed from both (h-fst.py:181-185)
        ov = mrec["kinds"]["match"]["hfst_overlap"]
        inB2 = inB.copy()
        for nm in ov["extra_in_b"]:
            inB2[mrec["names"].index(nm)] = 1
        r2 = bm.scan([(0, W, ov["L"])], None, inA, inB2)[0]
        for k in ("fst", "pi_a", "pi_b", "pi_xy", "dxy", "da"):
            assert rel_close(float(r2[k]), fh(ov["out"][k]), REL, 1e-300)
        bm.free()


@pytest.mark.parametrize("n,W", [(465, 5000), (31, 700), (64, 640), (97, 1300), (513, 900), (1030, 400)])
def test_scan_random_shapes(ctx, oracle, n, W):
    rng = np.random.default_rng(n * 7919 + W)
    m = founder_matrix(rng, n, W)
    bits = oracle.pack_hap_major(m)
    bm = ctx.upload(bits, W)
    inP = (rng.random(n) < 0.8).astype(np.uint8)
    inA = (rng.random(n) < 0.3).astype(np.uint8)
    inB = ((rng.random(n) < 0.3) & (inA == 0)).astype(np.uint8)
    # ragged, overlapping, unaligned, empty and single-site windows in one plan
    wins = [(0, W, W), (1, W - 1, 50000), (63, 65, 2), (64, 128, 64), (100, 100, 7), (W - 1, W, 1), (0, 1, 1),
            (W // 2, W, 0), (W // 4, 3 * W // 4, 123), (W // 4 + 1, 3 * W // 4 + 1, 123)]
    for tile_blocks, d_pi_mode, s_scope in ((0, 0, 0), (1, 1, 1), (3, 2, 0)):
        got = bm.scan(wins, inP, inA, inB, d_pi_mode=d_pi_mode, s_scope=s_scope, tile_blocks=tile_blocks)
        for (s0, s1, sl), r in zip(wins, got):
            want = oracle.window_sitecount(bits, n, s0, s1, oracle.pack_mask(inP), oracle.pack_mask(inA),
                                           oracle.pack_mask(inB), sl, d_pi_mode, s_scope)
            check_record(r, want, (n, W, s0, s1, tile_blocks))
    # permutation invariance of haplotype order (integers exact)
    perm = rng.permutation(n)
    bm2 = ctx.upload(oracle.pack_hap_major(m[perm]), W)
    a = bm.scan(wins[:3], inP, inA, inB)
    b = bm2.scan(wins[:3], inP[perm], inA[perm], inB[perm])
    for k in INT_KEYS:
        assert (a[k] == b[k]).all()
    bm.free(); bm2.free()


def test_scan_allpairs_oracle_n465(ctx, oracle):
    """BASELINE config 2/3 shape at a size the all-pairs oracle finishes in seconds."""
    n, W = 465, 6000
    rng = np.random.default_rng(465)
    m = founder_matrix(rng, n, W, nf=8, pf=1e-3, pp=1e-4)
    bits = oracle.pack_hap_major(m)
    bm = ctx.upload(bits, W)
    inA = np.zeros(n, np.uint8); inA[:140] = 1
    inB = np.zeros(n, np.uint8); inB[140:240] = 1
    wins = [(0, 3000, 3000), (3000, 6000, 3000)]
    got = bm.scan(wins, None, inA, inB)
    ones = oracle.pack_mask(np.ones(n, np.uint8))
    for (s0, s1, sl), r in zip(wins, got):
        want = oracle.window_allpairs(bits, n, s0, s1, ones, oracle.pack_mask(inA), oracle.pack_mask(inB), sl)
        check_record(r, want, (s0, s1))
    bm.free()


def test_sliding_window_additivity(ctx):
    """size-independent property: integer sums of a window equal the sums of its parts."""
    n, W = 465, 64 * 700 + 17
    bm = ctx.synthetic(n, W, seed=5)
    inA = np.zeros(n, np.uint8); inA[:140] = 1
    inB = np.zeros(n, np.uint8); inB[140:240] = 1
    import impop_amd
    slide = impop_amd.fixed_windows(W, 10000, 5000)
    halves = impop_amd.fixed_windows(W, 5000)
    a = bm.scan(slide, None, inA, inB)
    h = bm.scan(halves, None, inA, inB)
    for i, w in enumerate(slide):
        j = int(w["site_begin"]) // 5000
        parts = h[j: j + 2] if int(w["site_end"]) - int(w["site_begin"]) > 5000 else h[j: j + 1]
        for k in INT_KEYS:
            assert int(a[i][k]) == int(parts[k].sum()), (i, k)
    whole = bm.scan([(0, W)], None, inA, inB)[0]
    for k in INT_KEYS:
        assert int(whole[k]) == int(h[k].sum())
    bm.free()


def test_synthetic_generator_matches_numpy(ctx, oracle):
    import impop_amd
    n, W = 465, 3000
    bm = ctx.synthetic(n, W, seed=77, keep_hap_major=True)
    want = synth_matrix(n, 0, W, seed=77)
    got = impop_amd.unpack_hap_major(bm.download(), W)
    assert (got == want).all()
    # a window in the middle of a larger matrix regenerates identically (counter-based)
    bm2 = ctx.synthetic(n, 100000, seed=77)
    got2 = impop_amd.unpack_hap_major(bm2.download(64000 + 13, 64000 + 13 + 500), 500)
    assert (got2 == synth_matrix(n, 64013, 64513, seed=77)).all()
    # statistics look like the SURVEY generator: S around 5 % of sites
    r = bm2.scan([(0, 50000)])[0]
    assert 1500 < int(r["s_all"]) < 4000
    bm.free(); bm2.free()


def test_pairwise_counts_and_identity(ctx, oracle):
    g = load_golden("bitmatrix.json")
    for mrec in g["matrices"]:
        n, W = mrec["n"], mrec["W"]
        bits = golden_bits(mrec)
        bm = ctx.upload(bits, W)
        I = bm.pairwise_counts(0, W)
        assert (I.astype(np.int64) == golden_counts(mrec)).all()
        for s0, s1 in ((0, W), (3, W - 5), (31, 97), (64, 64)):
            assert (bm.pairwise_counts(s0, s1).astype(np.int64) == oracle.pairwise_counts(bits, n, s0, s1)).all()
        for kind, kid in (("match", 0), ("dice", 1)):
            sim = bm.pairwise_identity(0, W, kind)
            assert (sim == oracle.identity(golden_counts(mrec), W, kid)).all()  # one IEEE division each
        bm.free()


def test_pairwise_scan_vs_reference_goldens(ctx, oracle):
    g = load_golden("bitmatrix.json")
    for mrec in g["matrices"]:
        n, W, L = mrec["n"], mrec["W"], mrec["L"]
        bm = ctx.upload(golden_bits(mrec), W)
        inA, inB = np.array(mrec["in_a"], np.uint8), np.array(mrec["in_b"], np.uint8)
        for kind in ("match", "dice"):
            out = mrec["kinds"][kind]
            for c in out["pica2"]:
                if c["L"] is None:
                    continue
                r = bm.pairwise_scan([(0, W, c["L"])], None, inA, inB, kind=kind, threshold=fh(c["threshold"]),
                                     round_digits=c["round"])[0]
                assert rel_close(float(r["pi"]), fh(c["pi"]), REL, 1e-300), (mrec["name"], kind, c)
                assert rel_close(float(r["pi_site"]), fh(c["pi_site"]), REL, 1e-300)
            for c in out["hfst"]:
                r = bm.pairwise_scan([(0, W, c["L"] or 0)], None, inA, inB, kind=kind, threshold=1.0,
                                     round_digits=c["round"])[0]
                for k, v in c["out"].items():
                    assert rel_close(float(r[k]), fh(v), REL, 1e-300), (mrec["name"], kind, k)
        # batch of windows + subset P vs the oracle's dense functions
        rng = np.random.default_rng(3)
        inP = (rng.random(n) < 0.7).astype(np.uint8)
        wins = [(0, W, W), (W // 3, W, 777), (10, W // 2, 0)]
        res = bm.pairwise_scan(wins, inP, inA, inB, kind="match", threshold=0.995, round_digits=4, d_pi_mode=1, s_scope=1)
        bits = golden_bits(mrec)
        for (s0, s1, sl), r in zip(wins, res):
            sim = oracle.identity(oracle.pairwise_counts(bits, n, s0, s1), s1 - s0, 0)
            idx = np.nonzero(inP)[0]
            pi, ps, _, G = oracle.pica2(sim[np.ix_(idx, idx)], 0.995, sl, 4)
            assert rel_close(float(r["pi"]), pi, REL, 1e-300) and rel_close(float(r["pi_site"]), ps, REL, 1e-300)
            assert int(r["n_groups"]) == G
            h, _ = oracle.hfst(sim, inA, inB, sl, 4)
            for k, v in h.items():
                assert rel_close(float(r[k]), v, REL, 1e-300)
        bm.free()


def test_identity_functions_vs_goldens(ctx, oracle):
    """The .sim drop-in path: reference-API mirrors on the reference's own data structures."""
    from impop_amd import af, hfst, pica2
    g = load_golden("six_seq.json")
    rows = [(a, b, fh(v)) for a, b, v in g["rows"]]
    d = {((a, b) if a <= b else (b, a)): v for a, b, v in rows}
    names = sorted({r[0] for r in rows} | {r[1] for r in rows})
    for c in g["pica2"]:
        pi, ps = pica2.analyze_similarity_matrix(dict(d), set(names), len(d), fh(c["threshold"]), c["L"], None, c["round"], ctx=ctx)
        assert rel_close(pi, fh(c["pi"]), REL) and rel_close(ps, fh(c["pi_site"]), REL)
    A = {s for s in names if "popA" in s}
    B = {s for s in names if "popB" in s}
    for c in g["hfst"]:
        r = hfst.calculate_fst(d, set(A), set(B), c["L"], c["round"], ctx=ctx)
        for k, v in c["out"].items():
            assert rel_close(r[k], fh(v), REL), (k, r[k], fh(v))
    for c in g["af"]:
        cl = af.cluster(rows, names, fh(c["threshold"]), ctx=ctx)
        assert [sorted(x) for x in cl] == c["clusters"]
    # ragged table (missing pairs) and degenerate inputs
    rg = load_golden("ragged.json")
    names = rg["names"]
    sim = np.array([[fh(v) for v in row] for row in rg["sim"]])
    d = {(names[i], names[j]): float(sim[i, j]) for i in range(len(names)) for j in range(i, len(names))}
    for a, b in rg["dropped"]:
        d.pop((a, b), None)
    for c in rg["pica2"]:
        pi, ps = pica2.analyze_similarity_matrix(dict(d), set(names), len(d), fh(c["threshold"]), c["L"], None, c["round"], ctx=ctx)
        assert rel_close(pi, fh(c["pi"]), REL) and rel_close(ps, fh(c["pi_site"]), REL)
    for c in rg["hfst"]:
        r = hfst.calculate_fst(d, set(c["a"]), set(c["b"]), c["L"], None, ctx=ctx)
        for k, v in c["out"].items():
            assert rel_close(r[k], fh(v), REL)
    assert pica2.analyze_similarity_matrix({}, set(), 0, 1.0, 100, None, None, ctx=ctx) == (0.0, 0.0)
    assert pica2.analyze_similarity_matrix({("x", "x"): 1.0}, {"x"}, 1, 1.0, 100, None, None, ctx=ctx) == (0.0, 0.0)
    # bit-matrix goldens through the dict API (names, PanSN) incl. af on truncated names
    gm = load_golden("bitmatrix.json")["matrices"][1]
    n, W = gm["n"], gm["W"]
    simm = oracle.identity(golden_counts(gm), W, 0)
    nm = gm["names"]
    dd = {(nm[i], nm[j]): float(simm[i, j]) for i in range(n) for j in range(i, n)}
    for c in gm["kinds"]["match"]["pica2"]:
        pi, ps = pica2.analyze_similarity_matrix(dict(dd), set(nm), len(dd), fh(c["threshold"]), c["L"], None, c["round"], ctx=ctx)
        assert rel_close(pi, fh(c["pi"]), REL, 1e-300) and rel_close(ps, fh(c["pi_site"]), REL, 1e-300)
    rows = [(a.split(":", 1)[0], b.split(":", 1)[0], v) for (a, b), v in dd.items()]
    samples = sorted({a for a, _, _ in rows} | {b for _, b, _ in rows})
    for c in gm["kinds"]["match"]["af"]:
        cl = af.cluster(rows, samples, fh(c["threshold"]), ctx=ctx)
        assert [sorted(x) for x in cl] == c["clusters"]


def test_full_size_window_properties(ctx):
    """BASELINE.json full size (465 haplotypes, 50 kb windows) — size-independent checks:
    determinism, window-split additivity, pi bounds, Fst in range, per-site counts agree."""
    import impop_amd
    n, W = 465, 50000 * 40
    bm = ctx.synthetic(n, W, seed=20251031)
    inA = np.zeros(n, np.uint8); inA[:140] = 1
    inB = np.zeros(n, np.uint8); inB[140:240] = 1
    wins = impop_amd.fixed_windows(W, 50000)
    a = bm.scan(wins, None, inA, inB)
    b = bm.scan(wins, None, inA, inB, tile_blocks=16)
    assert a.tobytes() == b.tobytes()  # bit-identical regardless of tiling
    cnt = bm.site_counts(0, 50000)
    assert int(((cnt > 0) & (cnt < n)).sum()) == int(a[0]["s_all"])
    assert int((cnt.astype(np.int64) * (n - cnt.astype(np.int64))).sum()) == int(a[0]["sum_p"])
    assert ((a["fst"] >= -1) & (a["fst"] <= 1)).all() and (a["pi"] > 0).all() and (a["pi"] < 0.5).all()
    assert (a["tajima_d"] < 0).all()  # excess of rare variants by construction
    bm.free()


def test_config5_shape_4096_haplotypes(ctx, oracle):
    """BASELINE config 5 shape (4096 haplotypes) at a size the oracle finishes in seconds: the
    generic (wps = 128) scan kernel and the K-split int8-MFMA Gram kernel, integers bit-exact."""
    import impop_amd
    n, W = 4096, 6000
    bm = ctx.synthetic(n, W, seed=5, n_founder=16, p_founder=0.05, p_private_word=0.05, keep_hap_major=True)
    bits = bm.download()
    m = impop_amd.unpack_hap_major(bits, W)
    rng = np.random.default_rng(4096)
    inA = (rng.random(n) < 0.3).astype(np.uint8)
    inB = ((rng.random(n) < 0.3) & (inA == 0)).astype(np.uint8)
    wins = [(0, W, W), (100, 4133, 50000), (5999, 6000, 1)]
    got = bm.scan(wins, None, inA, inB)
    ones = oracle.pack_mask(np.ones(n, np.uint8))
    for (s0, s1, sl), r in zip(wins, got):
        want = oracle.window_sitecount(bits, n, s0, s1, ones, oracle.pack_mask(inA), oracle.pack_mask(inB), sl)
        check_record(r, want, ("n4096", s0, s1))
    # Gram: full matrix against numpy int64 (exact), one unaligned window
    I = bm.pairwise_counts(37, 5901)
    mm = m[:, 37:5901].astype(np.int32)
    want = (mm @ mm.T).astype(np.int64)
    assert (I.astype(np.int64) == want).all()
    bm.free()


def test_gram_determinism_and_batch(ctx):
    """K-split partial sums are integer atomics: two runs must be bit-identical; a batch of windows
    (grouped XCD mapping, >= 8 windows) equals the same windows run one by one."""
    n, W = 465, 64 * 300
    bm = ctx.synthetic(n, W, seed=99, keep_hap_major=True)
    a = bm.pairwise_counts(5, W - 3)
    b = bm.pairwise_counts(5, W - 3)
    assert (a == b).all()
    wins = [(i * 1500, i * 1500 + 1400 + i, 1500) for i in range(11)]
    res = bm.pairwise_scan(wins, None, None, None, kind="dice", threshold=0.9995, round_digits=None)
    for w, r in zip(wins, res):
        one = bm.pairwise_scan([w], None, None, None, kind="dice", threshold=0.9995, round_digits=None)[0]
        assert r.tobytes() == one.tobytes()
    bm.free()


def test_scan_multi_equals_pairwise_hfst(ctx, oracle):
    """K populations in one pass == K(K-1)/2 separate h-fst runs (the panel loop of
    run_h_fst_panels.sh:60-71), each checked through the oracle's h-fst restatement."""
    rng = np.random.default_rng(77)
    n, W = 465, 4000
    m = founder_matrix(rng, n, W, nf=8, pf=0.004, pp=0.0008)
    bits = oracle.pack_hap_major(m)
    bm = ctx.upload(bits, W)
    sizes = [140, 88, 100, 60, 72]  # AFR / AMR / EAS / EUR / SAS haplotype counts (doc/where_hprc_data.md:4-10, x2)
    perm = rng.permutation(n)
    pops, o = [], 0
    for s in sizes:
        f = np.zeros(n, np.uint8); f[perm[o: o + s]] = 1; o += s
        pops.append(f)
    wins = [(0, W, W), (100, 1777, 50000), (2000, 2000, 5), (3999, 4000, 0)]
    got = bm.scan_multi(wins, pops)
    assert got.shape == (len(wins), 10)
    ones = oracle.pack_mask(np.ones(n, np.uint8))
    p = 0
    for k in range(5):
        for l in range(k + 1, 5):
            for wi, (s0, s1, sl) in enumerate(wins):
                want = oracle.window_allpairs(bits, n, s0, s1, ones, oracle.pack_mask(pops[k]), oracle.pack_mask(pops[l]), sl)
                for key in ("fst", "pi_a", "pi_b", "pi_xy", "dxy", "da"):
                    assert rel_close(float(got[wi, p][key]), want[key], REL, 1e-300), (k, l, wi, key)
            # and identical to the 2-population scan of the same pair
            two = bm.scan(wins, None, pops[k], pops[l])
            for key in ("fst", "pi_a", "pi_b", "pi_xy", "dxy", "da"):
                assert (got[:, p][key] == two[key]).all() or np.allclose(got[:, p][key], two[key], rtol=1e-15, atol=0, equal_nan=True)
            p += 1
    import impop_amd
    with pytest.raises(impop_amd.ImpopError):
        bm.scan_multi(wins, [pops[0], pops[0]])  # overlapping populations are rejected
    bm.free()


def test_afs_matches_numpy(ctx):
    rng = np.random.default_rng(3)
    n, W = 130, 9000
    m = (rng.random((n, W)) < rng.beta(0.3, 2.0, size=W)[None, :]).astype(np.uint8)
    bm = ctx.upload_dense(m)
    sub = (rng.random(n) < 0.6).astype(np.uint8)
    wins = [(0, W), (17, 5000), (4096, 8192), (8999, 9000), (10, 10)]
    for mask, rows in ((None, np.ones(n, bool)), (sub, sub.astype(bool))):
        got = bm.afs(wins, mask)
        nP = int(rows.sum())
        assert got.shape == (len(wins), nP + 1)
        for (s0, s1), g in zip(wins, got):
            c = m[rows, s0:s1].sum(0)
            assert (g == np.bincount(c, minlength=nP + 1)).all()
            assert int(g.sum()) == s1 - s0
    bm.free()
    # thousands of windows: one workgroup per window stores its histogram directly; columns nobody / everybody carries
    # (the two bins counted per wave) next to ordinary ones, windows that start and end inside 64-site blocks
    m2 = m.copy()
    m2[:, rng.random(W) < 0.3] = 0
    m2[:, rng.random(W) < 0.3] = 1
    bm = ctx.upload_dense(m2)
    starts = np.sort(rng.integers(0, W - 70, size=9000))
    wins = [(int(a), int(a + rng.integers(1, 70))) for a in starts]
    for mask, rows in ((None, np.ones(n, bool)), (sub, sub.astype(bool)), (np.zeros(n, np.uint8), np.zeros(n, bool))):
        got = bm.afs(wins, mask)
        nP = int(rows.sum())
        cs = m2[rows].sum(0) if nP else np.zeros(W, dtype=np.int64)
        for (s0, s1), g in zip(wins[::37], got[::37]):
            assert (g == np.bincount(cs[s0:s1], minlength=nP + 1)).all(), (s0, s1, nP)
        assert (got.sum(axis=1) == np.array([b - a for a, b in wins])).all()
    bm.free()


def test_afs_and_site_counts_vs_reference_op_afs(ctx, tmp_path):
    """(f)2 / (f)4 against the reference instead of numpy: the `odgi paths -H` table of tests/golden/afs_table.json through
    impop_paths_table_parse -> upload -> impop_site_counts / impop_afs, against the per-column counts the real
    scripts/wip/op-afs.py returned (count of the first row's value: c_s or n - c_s) and the vectors its main() histograms
    (counts_d, op-afs.py:116)."""
    import impop_amd
    from impop_amd import extract
    g = load_golden("afs_table.json")
    p = tmp_path / "paths.tsv"
    p.write_text(g["table_text"])
    mf = extract.from_paths_table(str(p))
    n, W = mf.n_hap, mf.n_site
    bm = ctx.upload(mf.bits, W, keep_hap_major=False)
    c = bm.site_counts(0, W)
    for k, col in enumerate(g["columns"]):
        assert (int(c[k]) if col["value"] == 1 else n - int(c[k])) == col["count"], col
    # the spectrum: op-afs.py histograms counts_d[1] (carrier counts where row 0 carries the node) and counts_d[0]
    # (non-carrier counts where it does not); as carrier counts both are one histogram over all node columns
    want = np.zeros(n + 1, dtype=np.int64)
    for v in g["counts_d"].get("1", []):
        want[v] += 1
    for v in g["counts_d"].get("0", []):
        want[n - v] += 1
    got = bm.afs([(0, W, W)])[0]
    assert got.astype(np.int64).tolist() == want.tolist()
    # a sub-window and a subset of paths against the same captured table
    sub = np.zeros(n, np.uint8); sub[::2] = 1
    m = impop_amd.unpack_hap_major(mf.bits, W)
    assert bm.site_counts(5, 40, sub).tolist() == m[sub.astype(bool)][:, 5:40].sum(0).tolist()
    bm.free()


def test_fst_where_dxy_and_pi_xy_cancel_tolerance_policy(ctx):
    """INTEGRATION.md §4 on the GPU paths: a matrix whose haplotype pairs are all equally distant (Fst = Da = 0 exactly; the
    real h-fst.py returns -5e-16 ... -3e-15, tests/golden/fst_cancel.json).  pi / Dxy to 1e-9 relative; Fst and Da to 1e-9
    relative or the absolute floors 1e-12 / 1e-12 * Dxy — through the streaming scan, the all-pairs path and the .sim entry."""
    import base64
    g = load_golden("fst_cancel.json")
    n, W = g["n"], g["W"]
    bits = np.frombuffer(base64.b64decode(g["bits_u64_b64"]), dtype=np.uint64).reshape(n, -1).copy()
    inA, inB = np.array(g["in_a"], np.uint8), np.array(g["in_b"], np.uint8)
    bm = ctx.upload(bits, W)
    for kind in ("match", "dice"):
        sim = bm.pairwise_identity(0, W, kind)
        for c in g["kinds"][kind]:
            want = {k: fh(v) for k, v in c["out"].items()}
            got_pw = bm.pairwise_scan([(0, W, c["L"] or 0)], None, inA, inB, kind=kind, threshold=1.0, round_digits=c["round"], s_scope=2)[0]
            got_sim, _ = ctx.fst_from_identity(sim, inA, inB, c["L"], c["round"])
            routes = [("all-pairs", {k: float(got_pw[k]) for k in want}), (".sim", dict(zip(("fst", "pi_a", "pi_b", "pi_xy", "dxy", "da"), got_sim)))]
            if kind == "match" and c["round"] is None:
                rs = bm.scan([(0, W, c["L"] or 0)], None, inA, inB)[0]
                routes.append(("streaming", {k: float(rs[k]) for k in want}))
            for route, got in routes:
                for k in want:
                    assert stat_close(k, got[k], want[k], want["dxy"]), (route, kind, c["L"], c["round"], k, got[k], want[k])
    bm.free()


def test_matrix_free_refused_while_plan_alive(ctx):
    import impop_amd
    bm = ctx.synthetic(64, 1000, seed=1)
    plan = bm.plan([(0, 1000)])
    with pytest.raises(impop_amd.ImpopError):
        bm.free()
    plan.launch()
    assert int(plan.fetch()[0]["n_sites"]) == 1000
    plan.destroy()
    bm.free()
    # argument validation surfaces as errors, not faults
    bm = ctx.synthetic(64, 1000, seed=1)
    with pytest.raises(impop_amd.ImpopError):
        bm.scan([(0, 2000)])
    with pytest.raises(impop_amd.ImpopError):
        bm.scan([(10, 5)])
    with pytest.raises(impop_amd.ImpopError):
        bm.pairwise_counts(0, 1000)  # created without the haplotype-major copy
    bm.free()


def test_fuzz_scan_and_gram_random_shapes(ctx, oracle):
    """Seeded fuzz: random haplotype counts (incl. 1, 31, 32, 33, 64, 96, 97 ...), site counts below and
    above a 64-site block, random masks (empty / full / overlapping A and B), random ragged windows,
    random tile sizes: integers bit-exact against the oracle, doubles within 1e-9."""
    rng = np.random.default_rng(20251031)
    special_n = [1, 2, 31, 32, 33, 63, 64, 65, 95, 96, 97, 128, 129, 191, 192, 193, 465, 512, 513]
    for it in range(40):
        n = int(special_n[it % len(special_n)] if it < 2 * len(special_n) else rng.integers(1, 700))
        W = int(rng.choice([1, 5, 63, 64, 65, 127, 128, 129, 500, 1000, 2049, 3000]))
        dens = float(rng.choice([0.0, 0.02, 0.3, 0.5, 1.0]))
        m = (rng.random((n, W)) < dens).astype(np.uint8)
        bits = oracle.pack_hap_major(m)
        bm = ctx.upload(bits, W, keep_hap_major=True)
        def rmask():
            k = rng.integers(0, 4)
            if k == 0:
                return np.zeros(n, np.uint8)
            if k == 1:
                return np.ones(n, np.uint8)
            return (rng.random(n) < rng.random()).astype(np.uint8)
        inP, inA, inB = rmask(), rmask(), rmask()
        if not inP.any():
            inP[rng.integers(0, n)] = 1
        wins = []
        for _ in range(6):
            a, b = sorted(int(x) for x in rng.integers(0, W + 1, size=2))
            wins.append((a, b, int(rng.choice([0, 1, b - a if b > a else 3, 50000]))))
        wins.append((0, W, W))
        tb = int(rng.choice([0, 1, 2, 7, 64]))
        dmode, sscope = int(rng.integers(0, 3)), int(rng.integers(0, 2))
        got = bm.scan(wins, inP, inA, inB, d_pi_mode=dmode, s_scope=sscope, tile_blocks=tb)
        for (s0, s1, sl), r in zip(wins, got):
            want = oracle.window_sitecount(bits, n, s0, s1, oracle.pack_mask(inP), oracle.pack_mask(inA), oracle.pack_mask(inB),
                                           sl, dmode, sscope)
            check_record(r, want, ("fuzz", it, n, W, s0, s1, tb))
        # Gram on two random windows (incl. possibly empty)
        for (s0, s1, _) in wins[:2] + [wins[-1]]:
            I = bm.pairwise_counts(s0, s1)
            assert (I.astype(np.int64) == oracle.pairwise_counts(bits, n, s0, s1)).all(), ("gram", it, n, W, s0, s1)
        bm.free()


def test_hud_grouped_fst_vs_reference_goldens(ctx, oracle):
    """scripts/hudson/hud.py method='grouped' (goldens captured from the real hud.py) through the
    reference-API mirror impop_amd.hud.calculate_fst, plus the oracle on a non-golden threshold."""
    from impop_amd import hud
    g = load_golden("bitmatrix.json")
    for mrec in g["matrices"]:
        n, W = mrec["n"], mrec["W"]
        nm = mrec["names"]
        inA, inB = np.array(mrec["in_a"], np.uint8), np.array(mrec["in_b"], np.uint8)
        A = {nm[i] for i in range(n) if inA[i]}
        B = {nm[i] for i in range(n) if inB[i]}
        for kind, kid in (("match", 0), ("dice", 1)):
            sim = oracle.identity(golden_counts(mrec), W, kid)
            d = {(nm[i], nm[j]): float(sim[i, j]) for i in range(n) for j in range(i, n)}
            for c in mrec["kinds"][kind]["hud_grouped"]:
                r = hud.calculate_fst(d, set(A), set(B), c["L"], c["round"], None, "grouped", fh(c["threshold"]), ctx=ctx)
                for k, v in c["out"].items():
                    assert rel_close(r[k], fh(v), REL, 1e-300), (mrec["name"], kind, k, c)
            out, cnt = ctx.fst_grouped_from_identity(sim, inA, inB, 0.9975, 777, 3)
            want, wcnt = oracle.hud_grouped(sim, inA, inB, 0.9975, 777, 3)
            for k, key in enumerate(("fst", "pi_a", "pi_b", "pi_xy", "dxy", "da")):
                assert rel_close(float(out[k]), want[key], REL, 1e-300), (mrec["name"], kind, key)
            assert (cnt == wcnt).all()
        # method='direct' is h-fst
        r = hud.calculate_fst(d, set(A), set(B), mrec["L"], None, None, "direct", ctx=ctx)
        ref_h = [c for c in mrec["kinds"]["dice"]["hfst"] if c["L"] == mrec["L"] and c["round"] is None][0]["out"]
        for k, v in ref_h.items():
            assert rel_close(r[k], fh(v), REL, 1e-300)


def test_ehh_vs_reference_goldens_and_oracle(ctx, oracle):
    """impop_ehh (scripts/wip/ehhgfa.py calc_EHH): bit-for-bit the goldens captured from the real
    reference; the oracle on larger windows with unaligned edges, subsets and both directions."""
    from impop_amd import ehh as ehh_mod
    g = load_golden("ehh.json")
    for c in g["calc"]:
        m01 = np.array([[int(ch) for ch in r] for r in c["rows"]], dtype=np.uint8)
        got = ehh_mod.calc_EHH(m01, ctx=ctx)
        assert got.tolist() == [fh(v) for v in c["fwd"]], (m01.shape,)
        assert ehh_mod.calc_EHH(np.flip(m01, axis=1), ctx=ctx).tolist() == [fh(v) for v in c["rev"]]
        if m01.shape[0] >= 2:
            bm = ctx.upload_dense(m01, keep_hap_major=False)
            assert bm.ehh(0, m01.shape[1], reverse=True).tolist() == [fh(v) for v in c["rev"]]
            bm.free()
    hv = np.array(g["values"]["rows"])
    assert ehh_mod.calc_EHH(hv, ctx=ctx).tolist() == [fh(v) for v in g["values"]["fwd"]]
    assert ehh_mod.calc_EHH(np.flip(hv, axis=1), ctx=ctx).tolist() == [fh(v) for v in g["values"]["rev"]]

    rng = np.random.default_rng(77)
    n, W = 70, 3000
    anc = rng.integers(0, 2, size=W, dtype=np.uint8)
    f = np.repeat(anc[None, :], 4, axis=0) ^ (rng.random((4, W)) < 0.004).astype(np.uint8)
    m01 = f[rng.integers(0, 4, size=n)] ^ (rng.random((n, W)) < 0.0005).astype(np.uint8)
    bits = oracle.pack_hap_major(m01)
    bm = ctx.upload_dense(m01, keep_hap_major=False)
    member = (rng.random(n) < 0.6).astype(np.uint8)
    for s0, s1 in ((0, W), (5, 1999), (64, 128), (63, 65), (130, 131), (1000, 1000), (777, 2999)):
        for mem in (None, member):
            for rev in (False, True):
                got = bm.ehh(s0, s1, mem, reverse=rev)
                want = oracle.ehh(bits, n, s0, s1, mem, rev)
                assert got.tolist() == want.tolist(), (s0, s1, mem is not None, rev)
    one = np.zeros(n, np.uint8); one[3] = 1
    assert bm.ehh(10, 20, one).tolist() == [500.0] * 10
    bm.free()


def test_compacted_matrix_gives_identical_records(ctx, oracle):
    """impop_matrix_compact keeps only the sites variable among all haplotypes; scans of it with the
    ORIGINAL windows must return byte-identical records (n_sites, integer sums, every double), for
    subsets, overlapping / ragged / empty windows and the K-population scan.  Per-site
    entry points refuse a compacted matrix (the all-pairs path: test_all_pairs_path_on_compacted_matrix)."""
    import impop_amd
    from impop_amd import ImpopError
    rng = np.random.default_rng(11)
    n, W = 93, 20000
    anc = rng.integers(0, 2, size=W, dtype=np.uint8)
    f = np.repeat(anc[None, :], 5, axis=0) ^ (rng.random((5, W)) < 0.01).astype(np.uint8)
    m01 = f[rng.integers(0, 5, size=n)] ^ (rng.random((n, W)) < 0.0008).astype(np.uint8)
    m01[:, 5000:5600] = 1  # a fixed-1 stretch and a fixed-0 stretch: whole tiles without a variable site
    m01[:, 9000:9900] = 0
    bm = ctx.upload_dense(m01, keep_hap_major=False)
    cm = bm.compact()
    c = m01.sum(axis=0)
    var = np.nonzero((c > 0) & (c < n))[0]
    assert cm.n_site == var.size and cm.n_hap == n
    assert (cm.positions() == var.astype(np.uint64)).all()
    assert (impop_amd.unpack_hap_major(cm.download(), cm.n_site) == m01[:, var]).all()
    wins = [(0, W, W), (0, 0, 0), (17, 4999, 5000), (5000, 5600, 600), (5100, 5500, 0), (8999, 9901, 902), (9000, 9900, 900),
            (19999, 20000, 1), (123, 19877, 50000)]
    wins += [(s, min(s + 3000, W), 3000) for s in range(0, W, 1500)]  # overlapping (step = size / 2)
    inA = (rng.random(n) < 0.3).astype(np.uint8)
    inB = ((rng.random(n) < 0.4) & (inA == 0)).astype(np.uint8)
    inP = (rng.random(n) < 0.8).astype(np.uint8)
    for mp in (None, inP):
        for mode, scope in ((0, 0), (1, 1), (2, 0)):
            full = bm.scan(wins, mp, inA, inB, d_pi_mode=mode, s_scope=scope)
            comp = cm.scan(wins, mp, inA, inB, d_pi_mode=mode, s_scope=scope)
            assert full.tobytes() == comp.tobytes(), (mp is not None, mode, scope)
    # one window against the oracle directly
    r = cm.scan([(17, 4999, 5000)], None, inA, inB)[0]
    want = oracle.window_sitecount(oracle.pack_hap_major(m01), n, 17, 4999, oracle.pack_mask(np.ones(n, np.uint8)),
                                   oracle.pack_mask(inA), oracle.pack_mask(inB), 5000)
    for k in ("n_sites", "s_all", "s_a", "s_b", "sum_p", "sum_a", "sum_b", "sum_ab"):
        assert int(r[k]) == int(want[k]), k
    for k in ("pi", "pi_site", "fst", "dxy", "tajima_d"):
        assert rel_close(float(r[k]), float(want[k]), REL, 1e-300), k
    pops = [np.zeros(n, np.uint8) for _ in range(3)]
    for i in range(n):
        pops[i % 3][i] = 1
    assert bm.scan_multi(wins, pops).tobytes() == cm.scan_multi(wins, pops).tobytes()
    for call in (lambda: cm.afs(wins[:2]), lambda: cm.site_counts(0, 10), lambda: cm.ehh(0, 10), lambda: cm.compact()):
        with pytest.raises(ImpopError):
            call()
    with pytest.raises(ImpopError):
        cm.scan([(0, W + 1, 0)])  # windows are checked against the ORIGINAL length
    cm.free()
    bm.free()
    # no variable site at all (all haplotypes identical) and a single-haplotype matrix
    for mat in (np.repeat(anc[None, :300], 7, axis=0), anc[None, :300]):
        full_m = ctx.upload_dense(mat, keep_hap_major=False)
        empty = full_m.compact()
        assert empty.n_site == 0
        w2 = [(0, 300, 300), (10, 20, 0), (5, 5, 0)]
        fa = np.zeros(mat.shape[0], np.uint8); fa[:1] = 1
        fb = np.zeros(mat.shape[0], np.uint8); fb[1:3] = 1
        assert full_m.scan(w2, None, fa, fb).tobytes() == empty.scan(w2, None, fa, fb).tobytes()
        empty.free()
        full_m.free()


def test_int8_gram_kernel_still_exact():
    """The int8 + look-up-table Gram kernel is kept for A/B measurements (IMPOP_GRAM_MFMA=i8; the FP4
    bit-plane kernel is the default).  The switch is read once per process, hence the subprocess."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    code = (
        "import numpy as np, impop_amd\n"
        "ctx = impop_amd.Context(0)\n"
        "rng = np.random.default_rng(3)\n"
        "m = (rng.random((131, 3000)) < 0.3).astype(np.uint8)\n"
        "bm = ctx.upload_dense(m, keep_hap_major=True)\n"
        "for s0, s1 in ((0, 3000), (17, 2049), (64, 128), (5, 6)):\n"
        "    I = bm.pairwise_counts(s0, s1).astype(np.int64)\n"
        "    w = m[:, s0:s1].astype(np.int64)\n"
        "    assert (I == w @ w.T).all(), (s0, s1)\n"
        "print('ok')\n")
    env = dict(os.environ, IMPOP_GRAM_MFMA="i8", PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=ROOT)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr


def test_minor_allele_polarity_of_the_operand_is_invisible():
    """The all-pairs operand is stored in minor-allele polarity (sites most haplotypes carry are complemented, the set of such
    sites is one more row of the Gram; gram_unflip_kernel restores I and a where they matter).  Nothing a caller can see may
    depend on it: counts, both identities and the scan records of the default build must be byte-identical to a process run with
    IMPOP_NO_POLARITY=1 — on a matrix with a random ancestral polarity (half the sites flipped), overlapping windows (segment
    sums), a weighted matrix, a compacted one, and a haplotype count that leaves no padding row (96: stored as given)."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    code = r"""
import sys, hashlib
import numpy as np
import impop_amd
ctx = impop_amd.Context(0)
h = hashlib.sha256()
recs = []
rng = np.random.default_rng(11)
for n, W in ((61, 1500), (96, 700), (200, 2100)):
    anc = rng.integers(0, 2, size=W, dtype=np.uint8)
    f = np.repeat(anc[None], 5, axis=0) ^ (rng.random((5, W)) < 0.02).astype(np.uint8)
    m = f[rng.integers(0, 5, size=n)] ^ (rng.random((n, W)) < 0.003).astype(np.uint8)
    inA = (np.arange(n) % 3 == 0).astype(np.uint8); inB = (np.arange(n) % 3 == 1).astype(np.uint8)
    bm = ctx.upload_dense(m, keep_hap_major=True)
    wins = [(0, W, W), (5, W // 2, 700), (W // 3, W - 7, 0), (W // 4, W // 2 + 100, 50000)]
    tiling = [(k * (W // 24), (k + 1) * (W // 24), 100) for k in range(24)]            # >= 8 short windows: Gram tickets are chains
    sliding = [(k * (W // 40), k * (W // 40) + W // 10, 0) for k in range(30)]         # shared segments, chained too
    for mat in (bm, bm.compact()):
        for kind in ("match", "dice"):
            for fm in ("direct", "grouped"):
                r = mat.pairwise_scan(wins, None, inA, inB, kind=kind, threshold=0.99, round_digits=4, fst_method=fm)
                h.update(r.tobytes()); recs.append(r)
            for ww in (tiling, sliding):
                r = mat.pairwise_scan(ww, None, inA, inB, kind=kind, threshold=0.995, round_digits=5)
                h.update(r.tobytes()); recs.append(r)
    h.update(bm.pairwise_counts(3, W - 1).tobytes())
    h.update(bm.pairwise_identity(0, W, "dice").tobytes()); h.update(bm.pairwise_identity(10, W, "match").tobytes())
    bm.set_site_weights(rng.integers(1, 40, size=W).astype(np.uint32))
    h.update(bm.pairwise_counts(0, W).tobytes())
    h.update(bm.pairwise_scan(wins[:2], None, inA, inB, kind="dice", threshold=0.98, round_digits=None).tobytes())
np.save(sys.argv[1], np.concatenate(recs))
print(h.hexdigest())
"""
    # ... and the same for the other storage / scheduling choices of the all-pairs path that a caller must never see: chains of
    # windows per Gram ticket vs one window per ticket; counts as uint16 where they fit vs always int32.  And the two families of
    # epilogue kernels: the window-statistics kernels (stats_small.hip, the default here) and the general ones
    # (IMPOP_EPILOGUE_SMALL=0; with int32 counts; without their compile-time variants, IMPOP_EPILOGUE_FAST=0) — byte-identical
    # inside a family, equal across the two within the tolerance policy (their sums run in another order).
    import tempfile
    from conftest import stat_close
    outs, recs = {}, {}
    with tempfile.TemporaryDirectory() as td:
        for tag, extra in (("default", {}), ("no polarity", {"IMPOP_NO_POLARITY": "1"}), ("no chains", {"IMPOP_GRAM_CHAIN": "1"}),
                           ("long chains", {"IMPOP_GRAM_CHAIN": "8"}), ("int32 counts", {"IMPOP_GRAM_U16": "0"}),
                           ("general", {"IMPOP_EPILOGUE_SMALL": "0"}), ("general, int32 counts", {"IMPOP_EPILOGUE_SMALL": "0", "IMPOP_GRAM_U16": "0"}),
                           ("general, runtime variants", {"IMPOP_EPILOGUE_SMALL": "0", "IMPOP_EPILOGUE_FAST": "0"})):
            env = dict(os.environ, PYTHONPATH=ROOT, **extra)
            path = os.path.join(td, tag.replace(" ", "_").replace(",", "") + ".npy")
            r = subprocess.run([sys.executable, "-c", code, path], capture_output=True, text=True, env=env, cwd=ROOT, timeout=600)
            assert r.returncode == 0, (tag, r.stderr[-2000:])
            outs[tag] = r.stdout.strip()
            recs[tag] = np.load(path)
    small = {outs[t] for t in ("default", "no polarity", "no chains", "long chains", "int32 counts")}
    general = {outs[t] for t in outs if t.startswith("general")}
    assert len(outs["default"]) == 64 and len(small) == 1 and len(general) == 1, outs
    a, b = recs["default"], recs["general"]
    assert a.dtype == b.dtype and a.shape == b.shape
    for name in a.dtype.names:
        x, y = a[name], b[name]
        if x.dtype.kind in "iu":
            assert (x == y).all(), name
            continue
        for i in range(len(x)):
            if np.isnan(x[i]) or np.isnan(y[i]):
                assert np.isnan(x[i]) and np.isnan(y[i]), (name, i, x[i], y[i])
            else:
                assert stat_close(name, float(x[i]), float(y[i]), float(b["dxy"][i]) if "dxy" in a.dtype.names else 0.0), (name, i, x[i], y[i])


def test_gram_exact_beyond_fp32_integer_range(ctx):
    """FP4 MFMAs accumulate in fp32, exact only below 2^24: a window longer than that is K-split so
    that every partial stays exact and the int32 sum is still I_ij to the last unit (all-ones rows:
    I_ij = W = 2^25 + 77; a second matrix with every third site set in half of the rows)."""
    W = (1 << 25) + 77
    words = (W + 63) // 64
    n = 40
    bits = np.full((n, words), np.uint64(0xFFFFFFFFFFFFFFFF), dtype=np.uint64)
    bm = ctx.upload(bits, W, keep_hap_major=True)
    I = bm.pairwise_counts(0, W)
    assert (I == W).all()
    I2 = bm.pairwise_counts(3, W - 5)
    assert (I2 == W - 8).all()
    bm.free()
    pat = np.uint64(0x9249249249249249)  # bits 0, 3, 6, ... of the word; period 3 does not divide 64, so use counts
    bits[::2] = pat
    bm = ctx.upload(bits, W, keep_hap_major=True)
    I = bm.pairwise_counts(0, W)
    a = int(np.unpackbits(bits[0].view(np.uint8), bitorder="little")[:W].sum())
    assert int(I[0, 0]) == a and int(I[0, 2]) == a and int(I[0, 1]) == a and int(I[1, 1]) == W and int(I[1, 3]) == W
    bm.free()


def test_pairwise_scan_grouped_fst(ctx, oracle):
    """impop_pairwise_scan with fst_method='grouped': hud.py's grouped Fst (oracle_hud_grouped, pinned by
    goldens from the real hud.py) per window on identities formed from the Gram counts; the pica2 /
    Tajima fields must not change with the Fst method."""
    rng = np.random.default_rng(21)
    n, W = 57, 2600
    anc = rng.integers(0, 2, size=W, dtype=np.uint8)
    f = np.repeat(anc[None, :], 4, axis=0) ^ (rng.random((4, W)) < 0.02).astype(np.uint8)
    m01 = f[rng.integers(0, 4, size=n)] ^ (rng.random((n, W)) < 0.0006).astype(np.uint8)
    bits = oracle.pack_hap_major(m01)
    bm = ctx.upload_dense(m01, keep_hap_major=True)
    inA = (rng.random(n) < 0.45).astype(np.uint8)
    inB = (rng.random(n) < 0.45).astype(np.uint8)  # overlaps A on purpose (hud.py:186-190)
    wins = [(0, W, W), (100, 1400, 5000), (64, 128, 64), (1999, 2600, 0)]
    for kind, kid in (("match", 0), ("dice", 1)):
        for thr, rd in ((0.999, None), (0.99, 3), (1.0, None), (0.9985, 4)):
            direct = bm.pairwise_scan(wins, None, inA, inB, kind=kind, threshold=thr, round_digits=rd)
            got = bm.pairwise_scan(wins, None, inA, inB, kind=kind, threshold=thr, round_digits=rd, fst_method="grouped")
            for (s0, s1, L), r, d in zip(wins, got, direct):
                sim = oracle.identity(oracle.pairwise_counts(bits, n, s0, s1), s1 - s0, kid)
                want, _ = oracle.hud_grouped(sim, inA, inB, thr, L if L else None, rd)
                for k in ("fst", "pi_a", "pi_b", "pi_xy", "dxy", "da"):
                    assert rel_close(float(r[k]), want[k], REL, 1e-300), (kind, thr, rd, (s0, s1), k, float(r[k]), want[k])
                for k in ("pi", "pi_site", "tajima_d"):
                    a, b = float(r[k]), float(d[k])
                    assert a == b or (a != a and b != b), k
                assert int(r["n_groups"]) == int(d["n_groups"])
    bm.free()


def test_plan_mask_swap_equals_fresh_plans(ctx):
    """impop_scan_plan_set_masks: one plan (one set of tile tables) re-used for many population pairs
    returns, pair after pair, exactly what a fresh plan returns — fixed-WPS and any-n kernels."""
    rng = np.random.default_rng(8)
    for n, W in ((77, 9000), (600, 4000)):
        m01 = (rng.random((n, W)) < rng.random(W) * 0.5).astype(np.uint8)
        bm = ctx.upload_dense(m01, keep_hap_major=False)
        wins = [(0, W, W), (100, 2100, 2000), (2000, 4000, 50000), (3999, 4000, 1)]
        plan = bm.plan(wins, None, None, None)
        pops = [(rng.random(n) < 0.3).astype(np.uint8) for _ in range(4)]
        sub = (rng.random(n) < 0.7).astype(np.uint8)
        for mp in (None, sub):
            for i in range(len(pops)):
                for j in range(i + 1, len(pops)):
                    plan.set_masks(mp, pops[i], pops[j])
                    plan.launch()
                    assert plan.fetch().tobytes() == bm.scan(wins, mp, pops[i], pops[j]).tobytes(), (n, i, j)
        plan.destroy()
        bm.free()


def test_plan_launch_is_graph_capturable():
    """impop_scan_plan_launch does no host synchronisation and no allocation, so a hipGraph can capture
    it on the stream handed to the context (torch is only the capture harness here) and replay it."""
    import torch

    import impop_amd
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        ctx = impop_amd.Context(0, stream=s.cuda_stream)
        n, W, NW = 465, 5000, 64
        bm = ctx.synthetic(n, W * NW, seed=3)
        in_a = np.zeros(n, np.uint8); in_a[:140] = 1
        in_b = np.zeros(n, np.uint8); in_b[140:240] = 1
        plan = bm.plan(impop_amd.fixed_windows(W * NW, W), None, in_a, in_b)
        out = torch.zeros(NW * 128, dtype=torch.uint8, device="cuda")
        plan.launch(out.data_ptr())
        s.synchronize()
        want = out.clone()
        out.zero_()
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            plan.launch(out.data_ptr())
        assert int(out.sum()) == 0  # capture records, it does not run
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, want)
        del g
        plan.destroy()
        bm.free()
        ctx.close()


def test_pairwise_scan_without_site_scan(ctx):
    """s_scope = 2: callers that print neither S nor Tajima's D skip the site scan; every other field is
    unchanged, s_all / s_p are 0 and tajima_d is NaN."""
    rng = np.random.default_rng(2)
    n, W = 45, 3000
    m01 = (rng.random((n, W)) < 0.2).astype(np.uint8)
    bm = ctx.upload_dense(m01, keep_hap_major=True)
    inA = np.zeros(n, np.uint8); inA[:15] = 1
    inB = np.zeros(n, np.uint8); inB[20:40] = 1
    wins = [(0, W, W), (10, 1500, 0), (2000, 2001, 1)]
    a = bm.pairwise_scan(wins, None, inA, inB, threshold=0.9, round_digits=3)
    b = bm.pairwise_scan(wins, None, inA, inB, threshold=0.9, round_digits=3, s_scope=2)
    for k in ("pi", "pi_site", "fst", "pi_a", "pi_b", "pi_xy", "dxy", "da"):
        x, y = a[k], b[k]
        assert ((x == y) | (np.isnan(x) & np.isnan(y))).all(), k
    assert (a["n_groups"] == b["n_groups"]).all() and (a["n_sites"] == b["n_sites"]).all()
    assert (b["s_all"] == 0).all() and (b["s_p"] == 0).all() and np.isnan(b["tajima_d"]).all()
    bm.free()


def test_pairwise_scan_overlapping_windows_share_segments(ctx, oracle):
    """Sliding windows: the all-pairs path contracts every elementary segment once and forms a window as
    the sum of its segments' Gram matrices.  Records must equal those of the same windows scanned one by
    one (no sharing), in the caller's order, for shuffled, nested, duplicate and empty windows."""
    rng = np.random.default_rng(33)
    n, W = 70, 12000
    anc = rng.integers(0, 2, size=W, dtype=np.uint8)
    f = np.repeat(anc[None, :], 4, axis=0) ^ (rng.random((4, W)) < 0.02).astype(np.uint8)
    m01 = f[rng.integers(0, 4, size=n)] ^ (rng.random((n, W)) < 0.001).astype(np.uint8)
    bm = ctx.upload_dense(m01, keep_hap_major=True)
    inA = np.zeros(n, np.uint8); inA[:25] = 1
    inB = np.zeros(n, np.uint8); inB[30:60] = 1
    wins = [(s, min(s + 2000, W), 2000) for s in range(0, W - 500, 500)]          # window = 4 x step
    wins += [(100, 11000, 0), (3000, 3001, 1), (5000, 5000, 0), (2500, 4500, 2000), (2500, 4500, 777), (0, W, W)]
    order = rng.permutation(len(wins))
    wins = [wins[i] for i in order]
    for kw in (dict(threshold=0.995, round_digits=4), dict(threshold=0.99, kind="dice"), dict(threshold=0.995, fst_method="grouped")):
        shared = bm.pairwise_scan(wins, None, inA, inB, **kw)
        for k, w in enumerate(wins):
            single = bm.pairwise_scan([w], None, inA, inB, **kw)[0]
            assert shared[k].tobytes() == single.tobytes(), (kw, w)
    # one window against the oracle directly
    s0, s1, L = 1500, 3500, 2000
    sim = oracle.identity(oracle.pairwise_counts(oracle.pack_hap_major(m01), n, s0, s1), s1 - s0, 0)
    r = bm.pairwise_scan([(s, s + 2000, 2000) for s in range(0, 4000, 500)], None, inA, inB, threshold=0.995, round_digits=4)[3]
    pi, ps, _, G = oracle.pica2(sim, 0.995, L, 4)
    assert rel_close(float(r["pi"]), pi, REL, 1e-300) and int(r["n_groups"]) == G
    bm.free()


def test_window_without_a_gram_segment_among_overlapping_windows(ctx, oracle):
    """Found by tools/soak_gram.py in round 3 (a bug since round 2's segment sharing): among OVERLAPPING windows a window that owns no
    Gram segment — an empty site range, or on a compacted matrix a window without a single variable site — was given the first
    segment of its chunk by the vectorised row readers (join tests of the grouping, Step 2 of pica2) instead of zero counts.  Such a
    window must give the record it gives alone, and the compacted matrix the records of the full one."""
    rng = np.random.default_rng(42)
    n, W = 200, 6000
    anc = rng.integers(0, 2, size=W, dtype=np.uint8)
    f = np.repeat(anc[None], 6, axis=0) ^ (rng.random((6, W)) < 0.02).astype(np.uint8)
    m = f[rng.integers(0, 6, size=n)] ^ (rng.random((n, W)) < 0.004).astype(np.uint8)
    m[:, 2000:3500] = anc[None, 2000:3500]  # a stretch where all haplotypes agree: windows inside it have no variable site
    inA = (np.arange(n) % 3 == 0).astype(np.uint8); inB = (np.arange(n) % 3 == 1).astype(np.uint8)
    bm = ctx.upload_dense(m, keep_hap_major=True)
    cm = bm.compact()
    wins = [(0, 1500, 1500), (1000, 2600, 1600), (2100, 3400, 1300), (2500, 2500, 0), (2200, 3000, 800), (3000, 5000, 2000), (4000, 6000, 0)]
    for kw in (dict(kind="match", threshold=1.0, round_digits=None), dict(kind="match", threshold=0.999, round_digits=5),
               dict(kind="dice", threshold=0.99, round_digits=4), dict(kind="match", threshold=0.99, round_digits=None, fst_method="grouped")):
        got = bm.pairwise_scan(wins, None, inA, inB, **kw)
        for k, w in enumerate(wins):
            one = bm.pairwise_scan([w], None, inA, inB, **kw)[0]
            assert got[k].tobytes() == one.tobytes(), (kw, w)
        assert cm.pairwise_scan(wins, None, inA, inB, **kw).tobytes() == got.tobytes(), kw
        # the windows inside the stretch: every pair identical -> one group below a threshold of 1, pi = 0
        inside = got[4]
        assert int(inside["s_all"]) == 0 and float(inside["pi"]) == 0.0
        assert int(inside["n_groups"]) == (n if kw["threshold"] >= 1.0 else 1)
    cm.free(); bm.free()


def test_weighted_sites_equal_bp_expanded_matrix(ctx, oracle):
    """impop_matrix_set_site_weights: one column per graph node with its length as weight must give the
    records of the bp-expanded matrix (the same integer sums, n_sites, pi, Fst ...), except that the
    segregating-site counts count variable NODES; Tajima's D follows from that S (checked via the oracle)."""
    from impop_amd import ImpopError
    rng = np.random.default_rng(17)
    for n, K in ((37, 900), (600, 300)):  # fixed-WPS and any-n matrices
        nodes = (rng.random((n, K)) < rng.random(K) * 0.6).astype(np.uint8)
        nodes[:, rng.random(K) < 0.3] = 1                       # nodes every haplotype passes through
        length = rng.integers(1, 40, size=K).astype(np.uint32)
        cum = np.concatenate(([0], np.cumsum(length))).astype(np.int64)
        expanded = np.repeat(nodes, length, axis=1)
        bn = ctx.upload_dense(nodes, keep_hap_major=False)
        be = ctx.upload_dense(expanded, keep_hap_major=False)
        bn.set_site_weights(length)
        inA = (rng.random(n) < 0.4).astype(np.uint8); inB = (rng.random(n) < 0.4).astype(np.uint8)
        inP = (rng.random(n) < 0.8).astype(np.uint8)
        wn = [(0, K, 0), (3, 500 if K > 500 else 200, 12345), (K // 2, K // 2 + 1, 7), (10, 10, 0), (K - 64, K, 99)]
        we = [(int(cum[a]), int(cum[b]), L) for a, b, L in wn]
        for mp in (None, inP):
            got = bn.scan(wn, mp, inA, inB, d_pi_mode=1)
            ref = be.scan(we, mp, inA, inB, d_pi_mode=1)
            for k in ("n_sites", "sum_p", "sum_a", "sum_b", "sum_ab", "pi", "pi_site", "pi_a", "pi_b", "pi_xy", "dxy", "da", "fst"):
                x, y = got[k], ref[k]
                assert ((x == y) | ((x != x) & (y != y))).all(), (n, k, x, y)
            c = nodes.sum(axis=0)
            sel = nodes[(np.ones(n, bool) if mp is None else mp.astype(bool))]
            cp = sel.sum(axis=0)
            for (a, b, L), r in zip(wn, got):
                assert int(r["s_all"]) == int(((c[a:b] > 0) & (c[a:b] < n)).sum())
                assert int(r["s_p"]) == int(((cp[a:b] > 0) & (cp[a:b] < sel.shape[0])).sum())
                ps = float(r["pi_site"])
                if sel.shape[0] >= 2 and ps == ps:
                    D, _ = oracle.tajimas_d(sel.shape[0], float(int(r["s_all"])), ps)
                    d = float(r["tajima_d"])
                    assert (d != d and D != D) or rel_close(d, D, REL, 1e-300)
        # the oracle directly on the expanded matrix for one window
        a, b, L = wn[1]
        want = oracle.window_sitecount(oracle.pack_hap_major(expanded), n, int(cum[a]), int(cum[b]), oracle.pack_mask(np.ones(n, np.uint8)),
                                       oracle.pack_mask(inA), oracle.pack_mask(inB), L, 1, 0)
        r = bn.scan([wn[1]], None, inA, inB, d_pi_mode=1)[0]
        for k in ("n_sites", "sum_p", "sum_a", "sum_b", "sum_ab"):
            assert int(r[k]) == int(want[k]), k
        for k in ("pi", "pi_site", "fst", "dxy"):
            assert rel_close(float(r[k]), float(want[k]), REL, 1e-300), k
        # compaction keeps the weights of the kept nodes and the windows' full weight (monomorphic nodes included)
        cw = bn.compact()
        assert cw.n_site < K
        for mp in (None, inP):
            for mode in (0, 1, 2):
                assert cw.scan(wn, mp, inA, inB, d_pi_mode=mode).tobytes() == bn.scan(wn, mp, inA, inB, d_pi_mode=mode).tobytes()
        cw.free()
        # K-population scan with weights == the expanded matrix; the all-pairs path refuses weights
        pops = [inA & ~inB, inB & ~inA, (1 - (inA | inB)).astype(np.uint8)]
        assert bn.scan_multi(wn, pops).tobytes() == be.scan_multi(we, pops).tobytes()
        plan = bn.plan(wn)
        with pytest.raises(ImpopError):
            bn.set_site_weights(length)  # a live plan carries the windows' weights
        plan.destroy()
        bn.set_site_weights(None)  # weights removed: plain node-level scan again
        assert int(bn.scan([wn[0]])[0]["n_sites"]) == K
        bn.free(); be.free()


def _seeded_cases():
    g = load_golden("pica2_seeded.json")
    for t in g["tables"]:
        sim = np.array([[fh(v) for v in row] for row in t["sim"]])
        at = {nm: i for i, nm in enumerate(t["names"])}
        for run in t["runs"]:
            rank = np.zeros(t["n"], dtype=np.uint32)
            for k, nm in enumerate(run["order"]):
                rank[at[nm]] = k
            hud_rank = np.zeros(t["n"], dtype=np.uint32)
            for order in (run["order_a"], run["order_b"]):
                for k, nm in enumerate(order):
                    hud_rank[at[nm]] = k
            yield t, run, sim, rank, hud_rank


def test_pica2_nontransitive_tables_reproduce_reference_seed_order(ctx, oracle):
    """SURVEY §8a-a3 closed: on tables where "> threshold" is NOT transitive the reference's pi depends on the
    iteration order of set(elements).  Given the order each captured process (PYTHONHASHSEED 0..9) iterated, the
    GPU grouping returns THAT process's pi to 1e-9 and its group count; groups equal the oracle's."""
    checked = 0
    for t, run, sim, rank, _ in _seeded_cases():
        for c in run["pica2"]:
            thr = fh(c["threshold"])
            pi, ps, grp, G, (sum2, npairs) = ctx.pi_from_identity(sim, thr, c["round"], t["L"], seed_rank=rank, detail=True)
            assert rel_close(pi, fh(c["pi"]), REL) and rel_close(ps, fh(c["pi_site"]), REL), (t["name"], run["hashseed"], c, pi)
            assert G == c["n_groups"]
            _, _, ogrp, oG = oracle.pica2(sim, thr, t["L"], c["round"], seed_rank=rank)
            assert oG == G and (ogrp == grp).all()
            assert npairs == G * (G - 1) // 2 and rel_close(pi, t["n"] / (t["n"] - 1) * sum2, 1e-15)
            checked += 1
    assert checked >= 100
    # a rank array with a repeated value is refused, not silently tie-broken
    import impop_amd
    with pytest.raises(impop_amd.ImpopError):
        ctx.pi_from_identity(np.eye(3), 0.5, None, None, seed_rank=np.array([0, 1, 1], np.uint32))


def test_default_tajd_chain_all_pairs_path_vs_reference(ctx):
    """The reference's DEFAULT Tajima chain — run_tajd.sh:9-10 (THRESHOLD=0.999, R_VALUE=5), :166 pica2.py -t T -l LENGTH -r R,
    :174 first stdout token ("%.8f"), :180 tj_d.py -n SAMPLE_COUNT -p PI -S S_COUNT — on the all-pairs path, against values
    captured from the real pica2.py + tj_d.py (tests/golden/bitmatrix.json: tajd_chain_default).  d_pi_mode 0 is that wiring:
    pi_site through py_round(.., 8) into D, S over all rows of the window, n = members of the sample list."""
    g = load_golden("bitmatrix.json")
    seen = 0
    for mrec in g["matrices"]:
        n, W = mrec["n"], mrec["W"]
        bm = ctx.upload(golden_bits(mrec), W)
        inA = np.array(mrec["in_a"], np.uint8)
        for kind in ("match", "dice"):
            for label, mask in (("all", None), ("subset_a", inA)):
                c = mrec["tajd_chain_default"][kind][label]
                if c is None:
                    continue
                r = bm.pairwise_scan([(0, W, c["L"])], mask, None, None, kind=kind, threshold=0.999, round_digits=5)[0]
                assert int(r["n_groups"]) == c["n_groups"] and int(r["s_all"]) == c["S"], (mrec["name"], kind, label)
                assert rel_close(float(r["pi"]), fh(c["pi"]), REL, 1e-300) and rel_close(float(r["pi_site"]), fh(c["pi_site"]), REL, 1e-300)
                assert f"{float(r['pi_site']):.8f}" == c["pi_text"]
                want = fh(c["D"])
                got = float(r["tajima_d"])
                assert (got != got and want != want) or rel_close(got, want, REL), (mrec["name"], kind, label, got, want)
                seen += 1
        bm.free()
    assert seen >= 18


def test_default_tajd_chain_on_seeded_tables_per_hash_seed(ctx):
    """... and on tables where the grouping depends on the reference's set order: pi from impop_pi_from_identity with the
    captured order, its "%.8f" text into impop_tajimas_d, against the D the real chain printed under that hash seed."""
    seen = 0
    for t, run, sim, rank, _ in _seeded_cases():
        for c in run["pica2"]:
            pi, ps, _, G = ctx.pi_from_identity(sim, fh(c["threshold"]), c["round"], t["L"], seed_rank=rank)
            assert f"{ps:.8f}" == c["pi_text"] and G == c["n_groups"]
            got = float(ctx.tajimas_d(t["n"], float(c["S"]), float(f"{ps:.8f}"))[0])
            want = fh(c["D"])
            assert (got != got and want != want) or rel_close(got, want, REL), (t["name"], run["hashseed"], c, got)
            seen += c["round"] == 5 and fh(c["threshold"]) == 0.999
    assert seen >= 50


def test_pica2_default_rule_is_a_reference_outcome_and_mirror_uses_set_order(ctx):
    """Without an order the engine seeds with the smallest remaining name: its pi must be one of the values the
    reference produced under the 40 captured hash seeds (chain5).  And the function-level mirror, handed a set,
    follows THIS interpreter's iteration order of it — checked against the engine run on that order."""
    from impop_amd import pica2
    g = load_golden("pica2_seeded.json")
    t = next(x for x in g["tables"] if x["name"] == "chain5")
    sim = np.array([[fh(v) for v in row] for row in t["sim"]])
    c0 = t["runs"][0]["pica2"][0]
    captured = {fh(r["pica2"][0]["pi"]) for r in t["runs"]}
    pi, _, _, _ = ctx.pi_from_identity(sim, fh(c0["threshold"]), c0["round"], t["L"])
    assert any(rel_close(pi, w, REL) for w in captured), (pi, captured)
    for t in g["tables"]:
        sim = np.array([[fh(v) for v in row] for row in t["sim"]])
        names, n = t["names"], t["n"]
        d = {(names[i], names[j]): float(sim[i, j]) for i in range(n) for j in range(i, n)}
        elements = set()
        for k in t["insert_order"]:
            elements.add(names[k])
        rank = pica2.seed_rank_of(elements, names)
        assert sorted(rank.tolist()) == list(range(n))
        for c in t["runs"][0]["pica2"]:
            got = pica2.analyze_similarity_matrix(dict(d), elements, len(d), fh(c["threshold"]), t["L"], None, c["round"], ctx=ctx)
            want = ctx.pi_from_identity(sim, fh(c["threshold"]), c["round"], t["L"], seed_rank=rank)
            assert got[0] == want[0] and got[1] == want[1]


def test_hud_grouped_nontransitive_tables_reproduce_reference_seed_order(ctx):
    checked = 0
    for t, run, sim, _, hud_rank in _seeded_cases():
        inA, inB = np.array(t["in_a"], np.uint8), np.array(t["in_b"], np.uint8)
        for c in run["hud"]:
            out, _ = ctx.fst_grouped_from_identity(sim, inA, inB, fh(c["threshold"]), t["L"], c["round"], seed_rank=hud_rank)
            for k, key in enumerate(("fst", "pi_a", "pi_b", "pi_xy", "dxy", "da")):
                assert rel_close(float(out[k]), fh(c["out"][key]), REL, 1e-300), (t["name"], run["hashseed"], key)
            checked += 1
    assert checked >= 50


def test_scan_sharded_contexts_equal_one_context():
    """impop_scan_sharded: the window list cut into contiguous ranges, one context + one slab per
    range (the contexts share the test box's one device), every pass launched before the first fetch — records
    byte-identical to one context holding the whole matrix, for disjoint and sliding (halo) windows, with a
    subset mask, and for a shard count that leaves a shard empty."""
    import impop_amd
    from impop_amd import engine
    n, W = 465, 64 * 300 + 17
    c0 = impop_amd.Context(0)
    whole = c0.synthetic(n, W, seed=77)
    bits = whole.download()
    rng = np.random.default_rng(5)
    inA = (rng.random(n) < 0.3).astype(np.uint8); inB = (rng.random(n) < 0.3).astype(np.uint8)
    inP = (rng.random(n) < 0.9).astype(np.uint8)
    for size, step, shards, mp in ((1000, None, 2, None), (1000, 500, 3, inP), (4000, 1500, 4, None), (9000, None, 5, None)):
        wins = impop_amd.fixed_windows(W, size, step)
        want = whole.scan(wins, mp, inA, inB)
        ctxs, slabs, begins = [], [], []
        for k in range(shards):
            first, cnt, s0, s1 = engine.shard_windows_c(wins, shards, k)
            lo, hi = impop_amd.distributed.shard_range(len(wins), shards, k)
            assert (first, first + cnt) == (lo, hi)  # the C rule is the Python rule
            w0, w1 = s0 // 64, max((s1 + 63) // 64, s0 // 64 + 1)
            ck = impop_amd.Context(0)
            slabs.append(ck.upload(np.ascontiguousarray(bits[:, w0:w1]), min(64 * w1, W) - 64 * w0, keep_hap_major=False))
            ctxs.append(ck); begins.append(64 * w0)
        got = engine.scan_sharded(slabs, begins, wins, mp, inA, inB)
        assert got.tobytes() == want.tobytes(), (size, step, shards)
        # a slab that does not cover its shard is refused, not read out of bounds
        if shards >= 2:
            with pytest.raises(impop_amd.ImpopError):
                engine.scan_sharded(slabs, [b + 64 for b in begins], wins, mp, inA, inB)
        for s, ck in zip(slabs, ctxs):
            s.free(); ck.close()
    whole.free(); c0.close()


def test_pairwise_scan_sharded_contexts_equal_one_context():
    """impop_pairwise_scan_sharded: the all-pairs mode (thresholded pica2 + h-fst / grouped Fst + S + D) over several
    contexts, each shard on a host thread of its own — records byte-identical to one context holding the whole matrix, for
    disjoint and sliding windows, a subset mask, a shard count that leaves a shard empty; a slab that does not cover its
    shard or one context given twice is refused, and an error inside a shard comes back with its text."""
    import impop_amd
    from impop_amd import engine
    n, W = 200, 64 * 150 + 9
    c0 = impop_amd.Context(0)
    whole = c0.synthetic(n, W, seed=78, keep_hap_major=True)
    bits = whole.download()
    rng = np.random.default_rng(6)
    inA = (rng.random(n) < 0.3).astype(np.uint8); inB = (rng.random(n) < 0.3).astype(np.uint8)
    inP = (rng.random(n) < 0.9).astype(np.uint8)
    for size, step, shards, mp, meth in ((1000, None, 2, None, "direct"), (1000, 500, 3, inP, "direct"), (3000, 1100, 4, None, "grouped"),
                                         (5000, None, 5, None, "direct")):
        wins = impop_amd.fixed_windows(W, size, step)
        want = whole.pairwise_scan(wins, mp, inA, inB, threshold=0.999, round_digits=5, fst_method=meth)
        ctxs, slabs, begins = [], [], []
        for k in range(shards):
            first, cnt, s0, s1 = engine.shard_windows_c(wins, shards, k)
            w0, w1 = s0 // 64, max((s1 + 63) // 64, s0 // 64 + 1)
            ck = impop_amd.Context(0)
            slabs.append(ck.upload(np.ascontiguousarray(bits[:, w0:w1]), min(64 * w1, W) - 64 * w0, keep_hap_major=True))
            ctxs.append(ck); begins.append(64 * w0)
        got = impop_amd.pairwise_scan_sharded(slabs, begins, wins, mp, inA, inB, threshold=0.999, round_digits=5, fst_method=meth)
        assert got.tobytes() == want.tobytes(), (size, step, shards)
        if shards >= 2:
            with pytest.raises(impop_amd.ImpopError, match="does not cover"):
                impop_amd.pairwise_scan_sharded(slabs, [b + 64 for b in begins], wins, mp, inA, inB)
            with pytest.raises(impop_amd.ImpopError, match="are the same"):
                impop_amd.pairwise_scan_sharded([slabs[0]] * shards, begins, wins, mp, inA, inB)
            # an error raised inside a shard's thread (a slab without its hap-major operand) reaches the caller with its text
            bad = ctxs[1].upload(np.ascontiguousarray(bits[:, begins[1] // 64:]), W - begins[1], keep_hap_major=False)
            with pytest.raises(impop_amd.ImpopError, match="shard 1"):
                impop_amd.pairwise_scan_sharded([slabs[0], bad] + slabs[2:], begins, wins, mp, inA, inB)
            bad.free()
        for s, ck in zip(slabs, ctxs):
            s.free(); ck.close()
    whole.free(); c0.close()


def test_rccl_comm_one_rank_gather_and_allreduce():
    """The one-process-per-GPU exchange of the C ABI (impop_comm_* over RCCL, loaded with dlopen) with the one
    rank a one-GPU box allows: all-gather of raw device bytes on the context's stream directly behind the scan
    that writes them, impop_gather_records (padding + global order), and the int64 all-reduce of the K-split Gram."""
    import torch

    import impop_amd
    from impop_amd import engine
    dev = torch.device("cuda", 0)
    s = torch.cuda.Stream(dev)
    ctx = impop_amd.Context(0, stream=s.cuda_stream)
    comm = engine.Comm(ctx, engine.Comm.unique_id(), 1, 0)
    n, NW, Wn = 465, 3000, 5000
    bm = ctx.synthetic(n, NW * Wn, seed=3)
    wins = impop_amd.fixed_windows(NW * Wn, Wn)
    plan = bm.plan(wins, None, np.arange(n) < 100, np.arange(n) >= 300)
    with torch.cuda.stream(s):
        local = torch.zeros(NW * 128, dtype=torch.uint8, device=dev)
        allb = torch.zeros(NW * 128, dtype=torch.uint8, device=dev)
    s.synchronize()
    for _ in range(3):  # scan (about a millisecond) and gather back to back on one stream, no host sync in between
        plan.launch(local.data_ptr())
        comm.gather(local.data_ptr(), NW * 128, allb.data_ptr())
    s.synchronize()
    want = bm.scan(wins, None, np.arange(n) < 100, np.arange(n) >= 300)
    assert allb.cpu().numpy().tobytes() == want.tobytes()
    plan.launch()
    assert comm.gather_records(plan, NW).tobytes() == want.tobytes()
    with torch.cuda.stream(s):
        v = torch.arange(-500, 500, dtype=torch.int64, device=dev) * (1 << 40)
    s.synchronize()
    comm.allreduce_i64(v.data_ptr(), v.numel())
    s.synchronize()
    assert (v.cpu() == torch.arange(-500, 500, dtype=torch.int64) * (1 << 40)).all()
    plan.destroy(); bm.free(); comm.close(); ctx.close()


def test_window_spanning_many_tiles_uses_parallel_finalize(ctx, oracle):
    """Few long windows (BASELINE config 5's single-window shape): the tile partials of a window are summed by a
    wave (> 48 tiles) or a workgroup (> 2048 tiles) instead of one thread — integer sums, so the records cannot
    depend on which; checked against the one-thread path (large tiles) and the oracle."""
    n, W = 200, 64 * 5000 + 11
    bm = ctx.synthetic(n, W, seed=21)
    inA = np.arange(n) % 3 == 0; inB = np.arange(n) % 3 == 1
    wins = [(0, W, W), (64 * 100 + 5, 64 * 4100, 999), (7, 64 * 40, 0)]
    ref = bm.scan(wins, None, inA, inB, tile_blocks=4096)     # <= 2 tiles per window: one thread each
    for tb in (64, 8, 2, 1):                                   # up to 5001 tiles for the first window
        got = bm.scan(wins, None, inA, inB, tile_blocks=tb)
        assert got.tobytes() == ref.tobytes(), tb
    bits = bm.download()
    ones = oracle.pack_mask(np.ones(n, np.uint8))
    want = oracle.window_sitecount(bits, n, wins[1][0], wins[1][1], ones, oracle.pack_mask(inA), oracle.pack_mask(inB), 999)
    check_record(ref[1], want, "long window")
    bm.free()


def test_weighted_window_beyond_u32_is_refused_and_afs_batches(ctx):
    """impop_window_stats.n_sites is 32 bits: a weighted window whose weights add up to >= 2^32 is refused when the
    plan is built (it used to be truncated silently); impop_afs takes any number of windows (it used to stop at
    65 535, the grid's y limit)."""
    import impop_amd
    K = 3000
    rng = np.random.default_rng(8)
    nodes = (rng.random((20, K)) < 0.3).astype(np.uint8)
    bn = ctx.upload_dense(nodes, keep_hap_major=False)
    bn.set_site_weights(np.full(K, 2_000_000, dtype=np.uint32))   # 3000 x 2e6 = 6e9 > 2^32
    with pytest.raises(impop_amd.ImpopError) as ei:
        bn.scan([(0, K, 0)])
    assert "2^32" in str(ei.value)
    ok = bn.scan([(0, 2000, 0)])                                  # 4e9 < 2^32 still fits
    assert int(ok[0]["n_sites"]) == 4_000_000_000
    with pytest.raises(impop_amd.ImpopError):
        bn.scan_multi([(0, K, 0)], [np.arange(20) < 10, np.arange(20) >= 10])
    bn.free()
    n, W = 33, 70_000
    m = (rng.random((n, W)) < 0.2).astype(np.uint8)
    bm = ctx.upload_dense(m, keep_hap_major=False)
    wins = [(s, s + 1, 1) for s in range(W)]                      # 70 000 one-site windows
    afs = bm.afs(wins)
    c = m.sum(axis=0)
    assert afs.shape == (W, n + 1) and (afs.sum(axis=1) == 1).all()
    assert (afs[np.arange(W), c] == 1).all()
    bm.free()


def test_weighted_gram_equals_bp_expanded_matrix(ctx, oracle):
    """The all-pairs path on a NODE-level matrix with node lengths as site weights (what `impg similarity` hands the
    reference is a bp-weighted node-sharing identity, run_pica2_impg.sh:162-175): I_ij = sum_s w_s b_is b_js through
    the weights' bit planes must equal, bit for bit, the Gram matrix of the bp-expanded matrix — and with it the
    identities and the thresholded pica2 / h-fst / grouped-Fst records of the real Tajima pipeline (-t 0.999 -r 5)."""
    rng = np.random.default_rng(23)
    for n, K, wmax in ((61, 700, 40), (465, 400, 300), (130, 260, 70000)):
        founders = (rng.random((5, K)) < 0.5).astype(np.uint8)
        nodes = founders[rng.integers(0, 5, size=n)] ^ (rng.random((n, K)) < 0.01).astype(np.uint8)
        nodes[:, rng.random(K) < 0.3] = 1
        length = rng.integers(1, wmax, size=K).astype(np.uint32)
        if wmax > 65536:
            length[rng.random(K) < 0.9] = 1  # a few very long nodes: 17 bit planes, most of them sparse
        cum = np.concatenate(([0], np.cumsum(length))).astype(np.int64)
        bn = ctx.upload_dense(nodes, keep_hap_major=True)
        bn.set_site_weights(length)
        wn = [(0, K), (3, K - 70), (K // 2, K // 2 + 1), (17, 17), (K - 130, K)]
        expanded = np.repeat(nodes, length, axis=1) if cum[-1] < 200_000 else None
        be = ctx.upload_dense(expanded, keep_hap_major=True) if expanded is not None else None
        for a, b in wn:
            I = bn.pairwise_counts(a, b)
            nb = nodes[:, a:b].astype(np.int64)
            want = (nb * length[a:b].astype(np.int64)) @ nb.T
            assert (I.astype(np.int64) == want).all(), (n, a, b)
            if be is not None:
                assert (I == be.pairwise_counts(int(cum[a]), int(cum[b]))).all()
                for kind in ("match", "dice"):
                    assert bn.pairwise_identity(a, b, kind).tobytes() == be.pairwise_identity(int(cum[a]), int(cum[b]), kind).tobytes()
        if be is not None:
            inA = (rng.random(n) < 0.4).astype(np.uint8); inB = (rng.random(n) < 0.4).astype(np.uint8)
            inP = (rng.random(n) < 0.8).astype(np.uint8)
            wins_n = [(a, b, int(cum[b] - cum[a]) or 1) for a, b in wn] + [(0, K // 2, 777), (K // 4, K, 12345)]  # overlapping
            wins_e = [(int(cum[a]), int(cum[b]), L) for a, b, L in wins_n]
            for kind in ("match", "dice"):
                for thr, rd, meth in ((0.999, 5, "direct"), (0.99, None, "direct"), (0.995, 3, "grouped")):
                    for mp in (None, inP):
                        got = bn.pairwise_scan(wins_n, mp, inA, inB, kind=kind, threshold=thr, round_digits=rd, s_scope=2, fst_method=meth)
                        ref = be.pairwise_scan(wins_e, mp, inA, inB, kind=kind, threshold=thr, round_digits=rd, s_scope=2, fst_method=meth)
                        assert got.tobytes() == ref.tobytes(), (n, kind, thr, rd, meth)
            # with S: node-level S counts variable NODES; everything else equals the expanded matrix
            got = bn.pairwise_scan(wins_n, None, inA, inB, kind="dice", threshold=0.999, round_digits=5, d_pi_mode=1)
            ref = be.pairwise_scan(wins_e, None, inA, inB, kind="dice", threshold=0.999, round_digits=5, d_pi_mode=1)
            for k in ("pi", "pi_site", "fst", "pi_a", "pi_b", "pi_xy", "dxy", "da", "n_groups", "n_sites"):
                assert ((got[k] == ref[k]) | ((got[k] != got[k]) & (ref[k] != ref[k]))).all(), k
            c = nodes.sum(axis=0)
            for (a, b, L), r in zip(wins_n, got):
                assert int(r["s_all"]) == int(((c[a:b] > 0) & (c[a:b] < n)).sum())
            be.free()
        bn.free()
    # a window whose weights reach 2^31 cannot be held in the int32 Gram: refused, not wrapped
    import impop_amd
    big = ctx.upload_dense(np.ones((4, 40), np.uint8), keep_hap_major=True)
    big.set_site_weights(np.full(40, 60_000_000, dtype=np.uint32))
    with pytest.raises(impop_amd.ImpopError):
        big.pairwise_counts(0, 40)
    assert int(big.pairwise_counts(0, 30)[0, 0]) == 30 * 60_000_000
    big.free()


def test_grouping_soak_shape_regression(ctx, oracle):
    """Fixed-seed analogues of the shape tools/soak_groups.py failed on in round 2 (n = 1023, W = 1318, window 401-565,
    dice, -t 0.99 -r 4: 1673 groups reported for 223 — more groups than elements; the free set of greedy_groups_bits went
    between lanes through LDS and a lane's own store was forwarded over other lanes' writes) and its neighbours: mixed
    blocks of singleton candidates and absorbing ones, partial last bit words, both identity kinds, a subset.  Group by
    group against the oracle; since round 3 the exchange is a register exchange (ds_bpermute / v_readlane)."""
    shapes = [(1023, 1318, 401, 565, "dice", 0.99, 4, 0.003), (1023, 1318, 401, 565, "match", 0.99, 4, 0.003),
              (1024, 1318, 400, 566, "dice", 0.99, 4, 0.003), (1025, 1318, 401, 565, "dice", 0.99, None, 0.02),
              (511, 900, 3, 420, "dice", 0.995, 4, 0.003), (1023, 1318, 401, 565, "dice", 0.9, 4, 0.02)]
    for k, (n, W, a, b, kind, thr, rd, pflip) in enumerate(shapes):
        for seed in (1, 2, 3):
            rng = np.random.default_rng(1000 * k + seed)
            nf = int(rng.integers(8, 40))
            f = (rng.random((nf, W)) < 0.5).astype(np.uint8)
            m = f[rng.integers(0, nf, size=n)] ^ (rng.random((n, W)) < pflip).astype(np.uint8)
            bm = ctx.upload_dense(m, keep_hap_major=True)
            sim = oracle.identity(oracle.pairwise_counts(oracle.pack_hap_major(m), n, a, b), b - a, 0 if kind == "match" else 1)
            for inP in (None, (rng.random(n) < 0.9).astype(np.uint8)):
                r = bm.pairwise_scan([(a, b, b - a)], inP, None, None, kind=kind, threshold=thr, round_digits=rd, s_scope=2)[0]
                sel = np.arange(n) if inP is None else np.nonzero(inP)[0]
                pi, ps, grp, G = oracle.pica2(sim[np.ix_(sel, sel)], thr, b - a, rd)
                assert int(r["n_groups"]) == G, (n, W, a, b, kind, thr, rd, seed, int(r["n_groups"]), G)
                assert G <= len(sel)
                assert rel_close(float(r["pi"]), pi, REL, 1e-300), (n, kind, seed, float(r["pi"]), pi)
            gpi, gps, ggrp, gG = ctx.pi_from_identity(sim, thr, rd, b - a)  # the dense-table entry: element by element
            opi, ops, ogrp, oG = oracle.pica2(sim, thr, b - a, rd)
            assert gG == oG and (ggrp == ogrp).all()
            bm.free()


def test_device_error_word_fails_the_call_and_is_cleared(ctx):
    """A device-side consistency check that trips (the grouping's progress bound, stats.hip) ORs a bit into the context's error word;
    the call that launched the kernel must come back as IMPOP_E_INTERNAL — never with partial results — and the word must be
    clear again for the next call.  The bit is raised here by the ABI's test aid, the rest is the product's plumbing."""
    import ctypes as C
    import impop_amd
    from impop_amd import _lib
    lib = _lib.load()
    sim = np.array([[1.0, 0.9995, 0.9], [0.9995, 1.0, 0.9], [0.9, 0.9, 1.0]])
    want = ctx.pi_from_identity(sim, 0.999, None, 100)
    _lib.check(lib.impop_debug_raise_device_error(ctx.handle, 1))
    with pytest.raises(impop_amd.ImpopError) as e:
        ctx.pi_from_identity(sim, 0.999, None, 100)
    assert e.value.code == _lib.E_INTERNAL and "internal device check" in str(e.value)
    got = ctx.pi_from_identity(sim, 0.999, None, 100)  # cleared: the same call works again
    assert got[0] == want[0] and got[3] == want[3]
    m = (np.random.default_rng(1).random((20, 500)) < 0.3).astype(np.uint8)
    bm = ctx.upload_dense(m, keep_hap_major=True)
    ok = bm.pairwise_scan([(0, 500, 500)], None, None, None, threshold=0.9, s_scope=2)
    _lib.check(lib.impop_debug_raise_device_error(ctx.handle, 1))
    with pytest.raises(impop_amd.ImpopError) as e:
        bm.pairwise_scan([(0, 500, 500)], None, None, None, threshold=0.9, s_scope=2)
    assert e.value.code == _lib.E_INTERNAL
    assert bm.pairwise_scan([(0, 500, 500)], None, None, None, threshold=0.9, s_scope=2).tobytes() == ok.tobytes()
    bm.free()


def test_pica2_at_and_beyond_the_lds_limit(ctx, oracle):
    """The grouping state of a problem lives in LDS: 160 KB minus the kernel's static arrays (asked of the runtime).  A dense
    table just inside the limit runs (and agrees with the oracle); one beyond it is refused with an error, not launched."""
    import impop_amd
    rng = np.random.default_rng(3)
    n = 7000
    g = rng.integers(0, 9, size=n)
    sim = np.where(g[:, None] == g[None, :], 0.9995, 0.9).astype(np.float64)
    sim[np.arange(n), np.arange(n)] = 1.0
    pi, ps, grp, G = ctx.pi_from_identity(sim, 0.999, 5, 50000)
    opi, ops, ogrp, oG = oracle.pica2(sim, 0.999, 50000, 5)
    assert G == oG == 9 and (grp == ogrp).all() and rel_close(pi, opi, REL, 1e-300)
    del sim
    big = np.ones((7700, 7700))
    with pytest.raises(impop_amd.ImpopError, match="LDS-resident"):
        ctx.pi_from_identity(big, 0.999, 5, 50000)


def test_weighted_gram_plane_walks(ctx):
    """The two ways the weight planes are walked: inside one task (Horner on the fp32 accumulators; windows lighter than
    2^24) — with unused planes between used ones, a single used plane, only high planes — and one launch per plane with
    (count << k) added by the epilogue (heavier windows).  Both against numpy, on full, unaligned and batched windows."""
    rng = np.random.default_rng(77)
    n, K = 70, 700
    m = (rng.random((n, K)) < rng.random(K)).astype(np.uint8)
    mi = m.astype(np.int64)
    bm = ctx.upload_dense(m, keep_hap_major=True)
    spans = [(0, K), (13, 77), (64, 128), (129, 700), (300, 301)]
    for name, w in (("gaps", rng.choice([1, 16, 17, 4097, 8192], size=K)),      # planes 0, 4, 12, 13
                    ("one plane", np.full(K, 64)),
                    ("high planes", rng.choice([1 << 12, 1 << 13, 3 << 12], size=K)),
                    ("dense bits", rng.integers(1, 1 << 10, size=K)),
                    ("heavy", rng.integers(1 << 15, 1 << 19, size=K))):         # 700 x 2^18 > 2^24: launch per plane
        w = w.astype(np.uint32)
        bm.set_site_weights(w)
        for a, b in spans:
            want = (mi[:, a:b] * w[a:b].astype(np.int64)) @ mi[:, a:b].T
            assert want.max() < 2 ** 31
            got = bm.pairwise_counts(a, b).astype(np.int64)
            assert (got == want).all(), (name, a, b)
        # a batch of windows (different weights per window, light and — in the heavy case — heavy ones mixed)
        cum = np.concatenate(([0], np.cumsum(w.astype(np.int64))))
        wins = [(a, b, int(cum[b] - cum[a])) for a, b in spans]
        recs = bm.pairwise_scan(wins, None, None, None, kind="dice", threshold=0.9, round_digits=None, s_scope=2)
        for (a, b, L), r in zip(wins, recs):
            one = bm.pairwise_scan([(a, b, L)], None, None, None, kind="dice", threshold=0.9, round_digits=None, s_scope=2)[0]
            assert r.tobytes() == one.tobytes(), (name, a, b)
            assert int(r["n_sites"]) == L
    bm.free()
    # one long window: the site axis is split over K-slices (atomic adds of the slices' results), every slice walks the planes
    n, K = 40, 200_000
    m = (rng.random((n, K)) < rng.random(K)).astype(np.uint8)
    w = rng.integers(0, 41, size=K).astype(np.uint32)          # zero weights too: such a column counts for nothing
    bm = ctx.upload_dense(m, keep_hap_major=True)
    bm.set_site_weights(w)
    mi = m.astype(np.int64)
    for a, b in ((0, K), (12345, 187_654)):
        want = (mi[:, a:b] * w[a:b].astype(np.int64)) @ mi[:, a:b].T
        assert want.max() < 2 ** 24
        assert (bm.pairwise_counts(a, b).astype(np.int64) == want).all(), (a, b)
    bm.free()


def test_config5_full_size_gram_and_scan_properties(ctx):
    """BASELINE config 5 at FULL size — 4096 haplotypes x 10^7 sites (5.12 GB) resident, ONE window — on both paths,
    through properties that need no oracle: the K-split FP4 Gram matrix is symmetric, reproducible byte for byte,
    additive over a split of the site axis, has the row popcounts a_i on its diagonal, satisfies the checksum of
    checksums  sum_ij I_ij = sum_s c_s^2  and, on sampled rows,  sum_j I_ij = sum_s b_is c_s  (c_s from
    impop_site_counts); the streaming scan's integer sums equal sum_s c_s (n - c_s) and #{0 < c_s < n}."""
    n, W = 4096, 10_000_000
    bm = ctx.synthetic(n, W, seed=5, n_founder=16, p_founder=0.05, p_private_word=0.05, keep_hap_major=True)
    I = bm.pairwise_counts(0, W)
    assert bm.pairwise_counts(0, W).tobytes() == I.tobytes()
    assert (I == I.T).all()
    c = bm.site_counts(0, W).astype(np.int64)
    assert int(I.astype(np.int64).sum()) == int((c * c).sum())
    cut = 4_321_987
    assert (I == bm.pairwise_counts(0, cut) + bm.pairwise_counts(cut, W)).all()
    bits = bm.download(0, W)                                     # [4096, 156250] uint64
    assert (np.diag(I).astype(np.int64) == np.bitwise_count(bits).sum(axis=1, dtype=np.int64)).all()
    for i in (0, 1, 31, 32, 2047, 2048, 4095):
        b = np.unpackbits(bits[i].view(np.uint8), bitorder="little")[:W]
        assert int(I[i].astype(np.int64).sum()) == int(c[b.astype(bool)].sum()), i
    del bits
    r = bm.scan([(0, W, W)])[0]
    assert int(r["n_sites"]) == W and int(r["sum_p"]) == int((c * (n - c)).sum())
    assert int(r["s_all"]) == int(((c > 0) & (c < n)).sum())
    # the same window through the all-pairs statistics: pica2 at threshold >= 1 is the site-count pi
    p = bm.pairwise_scan([(0, W, W)], None, np.arange(n) < 1000, np.arange(n) >= 3000, kind="match", threshold=1.0)[0]
    s = bm.scan([(0, W, W)], None, np.arange(n) < 1000, np.arange(n) >= 3000)[0]
    for k in ("pi", "pi_site", "fst", "pi_a", "pi_b", "dxy", "tajima_d"):
        assert rel_close(float(p[k]), float(s[k]), REL, 1e-300), (k, float(p[k]), float(s[k]))
    assert int(p["s_all"]) == int(s["s_all"]) and int(p["n_sites"]) == W
    bm.free()


def test_all_pairs_path_on_compacted_matrix(ctx, oracle):
    """impop_matrix_compact of a matrix that kept its hap-major copy serves the all-pairs path: the contraction runs
    over the variable sites of a window only, every dropped site that all haplotypes carry comes back as +1 on every
    I_ij.  Counts, identities (match and dice) and the thresholded pica2 / h-fst / grouped-Fst / D records must be
    byte-identical to the full matrix's, windows given in ORIGINAL coordinates (overlapping ones too)."""
    import impop_amd
    rng = np.random.default_rng(31)
    for n, W in ((61, 3000), (465, 20000), (700, 5000)):
        anc = rng.integers(0, 2, size=W, dtype=np.uint8)                       # about half the sites are all-ones
        f = np.repeat(anc[None], 6, axis=0) ^ (rng.random((6, W)) < 0.004).astype(np.uint8)
        m = f[rng.integers(0, 6, size=n)] ^ (rng.random((n, W)) < 0.0005).astype(np.uint8)
        full = ctx.upload_dense(m, keep_hap_major=True)
        cm = full.compact()
        assert 0 < cm.n_site < W // 2
        c = m.sum(axis=0)
        assert cm.n_site == int(((c > 0) & (c < n)).sum())
        for a, b in ((0, W), (17, W - 33), (W // 2, W // 2 + 1), (100, 100), (W - 64, W)):
            I = cm.pairwise_counts(a, b)
            assert (I == full.pairwise_counts(a, b)).all(), (n, a, b)
            mm = m[:, a:b].astype(np.int64)
            assert (I.astype(np.int64) == mm @ mm.T).all()
            for kind in ("match", "dice"):
                assert cm.pairwise_identity(a, b, kind).tobytes() == full.pairwise_identity(a, b, kind).tobytes()
        inA = (rng.random(n) < 0.4).astype(np.uint8); inB = (rng.random(n) < 0.4).astype(np.uint8)
        inP = (rng.random(n) < 0.8).astype(np.uint8)
        for wins in (impop_amd.fixed_windows(W, 1000), impop_amd.fixed_windows(W, 1500, 500), [(5, 5, 1), (0, W, 12345)]):
            for kind, thr, rd, meth, mp in (("match", 0.999, 5, "direct", None), ("dice", 0.9995, None, "direct", inP),
                                            ("match", 0.998, 4, "grouped", None), ("dice", 1.0, None, "direct", None)):
                got = cm.pairwise_scan(wins, mp, inA, inB, kind=kind, threshold=thr, round_digits=rd, fst_method=meth)
                ref = full.pairwise_scan(wins, mp, inA, inB, kind=kind, threshold=thr, round_digits=rd, fst_method=meth)
                assert got.tobytes() == ref.tobytes(), (n, kind, thr, rd, meth)
        cm.free(); full.free()
    # weighted source (one column per graph node, node lengths as weights): the dropped all-ones columns come back as the
    # SUM OF THEIR WEIGHTS on every I_ij (long shared anchors leave the contraction, and with them the high weight planes)
    n, W = 120, 6000
    anc = (rng.random(W) < 0.5).astype(np.uint8)
    f = np.repeat(anc[None], 5, axis=0) ^ (rng.random((5, W)) < 0.01).astype(np.uint8)
    m = f[rng.integers(0, 5, size=n)] ^ (rng.random((n, W)) < 0.001).astype(np.uint8)
    w = rng.integers(1, 8, size=W).astype(np.uint32)
    c = m.sum(axis=0)
    w[c == n] = rng.integers(100, 3000, size=int((c == n).sum())).astype(np.uint32)   # anchors: long, carried by everybody
    full = ctx.upload_dense(m, keep_hap_major=True)
    full.set_site_weights(w)
    cm = full.compact()
    assert cm.n_site == int(((c > 0) & (c < n)).sum())
    cum = np.concatenate(([0], np.cumsum(w.astype(np.int64))))
    mi = m.astype(np.int64)
    for a, b in ((0, W), (17, W - 33), (3000, 3001), (W - 64, W)):
        I = cm.pairwise_counts(a, b)
        assert (I.astype(np.int64) == (mi[:, a:b] * w[a:b].astype(np.int64)) @ mi[:, a:b].T).all(), (a, b)
        assert (I == full.pairwise_counts(a, b)).all()
        for kind in ("match", "dice"):
            assert cm.pairwise_identity(a, b, kind).tobytes() == full.pairwise_identity(a, b, kind).tobytes()
    inA = (rng.random(n) < 0.4).astype(np.uint8); inB = (rng.random(n) < 0.4).astype(np.uint8)
    inP = (rng.random(n) < 0.8).astype(np.uint8)
    for spans in ([(k, min(k + 400, W)) for k in range(0, W, 400)], [(k, min(k + 600, W)) for k in range(0, W - 200, 200)]):
        wins = [(a, b, int(cum[b] - cum[a])) for a, b in spans]
        for kind, thr, rd, meth, mp in (("match", 0.999, 5, "direct", None), ("dice", 0.9995, None, "direct", inP),
                                        ("match", 0.998, 4, "grouped", None)):
            got = cm.pairwise_scan(wins, mp, inA, inB, kind=kind, threshold=thr, round_digits=rd, fst_method=meth)
            ref = full.pairwise_scan(wins, mp, inA, inB, kind=kind, threshold=thr, round_digits=rd, fst_method=meth)
            assert got.tobytes() == ref.tobytes(), (kind, thr, rd, meth)
    cm.free(); full.free()
    # nothing variable at all: the compaction keeps no column, every I_ij is the constant
    mono = np.zeros((20, 300), np.uint8); mono[:, ::3] = 1
    wm = rng.integers(1, 500, size=300).astype(np.uint32)
    full = ctx.upload_dense(mono, keep_hap_major=True)
    full.set_site_weights(wm)
    cm = full.compact()
    assert cm.n_site == 0
    assert (cm.pairwise_counts(10, 200) == int(wm[10:200][mono[0, 10:200] == 1].sum())).all()
    w1 = [(0, 300, int(wm.sum())), (10, 200, int(wm[10:200].sum()))]
    assert cm.pairwise_scan(w1, None, None, None).tobytes() == full.pairwise_scan(w1, None, None, None).tobytes()
    cm.free(); full.free()
    # a compacted matrix without the operand (source kept no hap-major copy) still refuses, loudly
    src = ctx.upload_dense(m, keep_hap_major=False)
    c2 = src.compact()
    with pytest.raises(impop_amd.ImpopError):
        c2.pairwise_counts(0, 10)
    c2.free(); src.free()


def test_large_problems_split_epilogues(ctx, oracle):
    """>= 1024 elements: pica2's Step 2 runs as row chunks over many workgroups (pica2_rows_kernel + finish) and h-fst's
    member rows are split likewise; both must agree with the oracle — on a Gram problem straight from the bit matrix
    (few groups and all-singleton groups), on a dense `.sim`-style table with missing pairs, and with a seed order."""
    rng = np.random.default_rng(41)
    n, W = 1100, 3000
    f = (rng.random((7, W)) < 0.5).astype(np.uint8)
    who = rng.integers(0, 7, size=n)
    m = f[who] ^ (rng.random((n, W)) < 0.0006).astype(np.uint8)
    bm = ctx.upload_dense(m, keep_hap_major=True)
    bits = oracle.pack_hap_major(m)
    inA = (rng.random(n) < 0.45).astype(np.uint8); inB = (rng.random(n) < 0.45).astype(np.uint8)
    wins = [(0, W, W), (100, 2100, 5000)]
    for kind, kid in (("match", 0), ("dice", 1)):
        for thr, rd in ((0.999, 5), (0.9995, None), (1.0, None)):   # a few dozen groups ... every haplotype its own group
            got = bm.pairwise_scan(wins, None, inA, inB, kind=kind, threshold=thr, round_digits=rd, s_scope=2)
            for (s0, s1, L), r in zip(wins, got):
                sim = oracle.identity(oracle.pairwise_counts(bits, n, s0, s1), s1 - s0, kid)
                pi, ps, _, G = oracle.pica2(sim, thr, L, rd)
                assert int(r["n_groups"]) == G, (kind, thr, rd, int(r["n_groups"]), G)
                assert rel_close(float(r["pi"]), pi, REL, 1e-300) and rel_close(float(r["pi_site"]), ps, REL, 1e-300), (kind, thr, rd)
                h, _ = oracle.hfst(sim, inA, inB, L, rd)
                for k in ("fst", "pi_a", "pi_b", "pi_xy", "dxy", "da"):
                    # random populations: Fst / Da are differences of nearly equal means, so an absolute floor for them
                    assert rel_close(float(r[k]), h[k], REL, 1e-12 if k in ("fst", "da") else 1e-300), (kind, thr, rd, k)
    # dense table with missing pairs and a seed order (the .sim drop-in path at n >= 1024)
    sim = oracle.identity(oracle.pairwise_counts(bits, n, 0, W), W, 0)
    drop = rng.random((n, n)) < 0.01
    sim[np.triu(drop, 1) | np.triu(drop, 1).T] = np.nan
    rank = rng.permutation(n).astype(np.uint32)
    for thr, rd in ((0.9992, 4), (0.99, None)):
        pi, ps, grp, G, (sum2, npairs) = ctx.pi_from_identity(sim, thr, rd, 777, seed_rank=rank, detail=True)
        opi, ops, ogrp, oG = oracle.pica2(sim, thr, 777, rd, seed_rank=rank)
        assert G == oG and (grp == ogrp).all()
        assert rel_close(pi, opi, REL, 1e-300) and rel_close(ps, ops, REL, 1e-300)
        assert rel_close(pi, n / (n - 1) * sum2, 1e-14) and npairs <= G * (G - 1) // 2
        out, cnt = ctx.fst_from_identity(sim, inA, inB, 777, rd)
        want, wcnt = oracle.hfst(sim, inA, inB, 777, rd)
        for k, key in enumerate(("fst", "pi_a", "pi_b", "pi_xy", "dxy", "da")):
            assert rel_close(float(out[k]), want[key], REL, 1e-12 if key in ("fst", "da") else 1e-300), key
        assert (cnt == wcnt).all()
    # seeds in name order (no seed order handed in): Step 1 runs on the bit matrix pica2_adj_kernel leaves — here on a dense
    # table with missing pairs (never joins), group by group against the oracle
    for thr, rd in ((0.9992, 4), (0.99, None), (1.0, None)):
        pi, ps, grp, G = ctx.pi_from_identity(sim, thr, rd, 777)
        opi, ops, ogrp, oG = oracle.pica2(sim, thr, 777, rd)
        assert G == oG and (grp == ogrp).all(), (thr, rd, G, oG)
        assert rel_close(pi, opi, REL, 1e-300) and rel_close(ps, ops, REL, 1e-300)
    # ... and on a population subset of the Gram problem (positions != matrix rows)
    inP = (rng.random(n) < 0.96).astype(np.uint8)
    sel = np.nonzero(inP)[0]
    assert len(sel) >= 1024
    for thr, rd in ((0.999, 5), (1.0, None)):
        r = bm.pairwise_scan([(0, W, W)], inP, inA, inB, threshold=thr, round_digits=rd, s_scope=2)[0]
        simw = oracle.identity(oracle.pairwise_counts(bits, n, 0, W), W, 0)
        pi, ps, _, G = oracle.pica2(simw[np.ix_(sel, sel)], thr, W, rd)
        assert int(r["n_groups"]) == G and rel_close(float(r["pi"]), pi, REL, 1e-300), (thr, rd)
    bm.free()


def test_bit_matrix_grouping_beyond_4096_elements(ctx, oracle):
    """More than 4096 elements: the free set of the bit-matrix grouping spans both register words of a lane, candidate
    blocks are shorter than 64 and the last adjacency word is partial.  Groups element by element against the oracle
    (non-transitive joins included: noise on seven founders)."""
    rng = np.random.default_rng(43)
    n, W = 4199, 700
    f = (rng.random((7, W)) < 0.5).astype(np.uint8)
    m = f[rng.integers(0, 7, size=n)] ^ (rng.random((n, W)) < 0.002).astype(np.uint8)
    bits = oracle.pack_hap_major(m)
    sim = oracle.identity(oracle.pairwise_counts(bits, n, 0, W), W, 0)
    for thr, rd in ((0.995, None), (0.997, 3), (1.0, None)):
        pi, ps, grp, G = ctx.pi_from_identity(sim, thr, rd, W)
        opi, ops, ogrp, oG = oracle.pica2(sim, thr, W, rd)
        assert G == oG and (grp == ogrp).all(), (thr, rd, G, oG)
        assert rel_close(pi, opi, REL, 1e-300)
    bm = ctx.upload_dense(m, keep_hap_major=True)
    r = bm.pairwise_scan([(0, W, W)], None, None, None, threshold=0.995, round_digits=None, s_scope=2)[0]
    assert int(r["n_groups"]) == oracle.pica2(sim, 0.995, W, None)[3]
    bm.free()


def test_window_statistics_kernels_edge_shapes(ctx, oracle):
    """csrc/stats_small.hip (pica2 / h-fst on disjoint and sliding windows of <= 512 haplotypes, `match`, uint16 counts): sizes around the
    64-position words and the 256-member switch of the h-fst rows, element lists (subset P), overlapping / empty / one-member
    populations, thresholds from "one group" (-1) to "nobody joins" (1.5), roundings 0..6 — every field against the oracle
    (tools/soak_small.py is the long-running form)."""
    from conftest import stat_close
    rng = np.random.default_rng(404)
    sizes = [1, 2, 3, 63, 64, 65, 127, 128, 129, 255, 256, 257, 300, 465, 511, 512]
    thrs = [1.5, 1.0, 0.9999, 0.999, 0.99, 0.9, 0.0, -1.0]
    for it, n in enumerate(sizes * 2):
        nwin = (1, 3, 9)[it % 3]
        wlen = int(rng.integers(40, 700))
        step = wlen if it < len(sizes) else max(1, wlen // (2, 3, 5)[it % 3])  # second round: sliding windows (segment sums)
        W = (nwin - 1) * step + wlen + int(rng.integers(0, 50))
        nf = int(rng.integers(1, 30))
        f = (rng.random((nf, W)) < 0.5).astype(np.uint8)
        m = f[rng.integers(0, nf, size=n)] ^ (rng.random((n, W)) < rng.choice([0.0, 0.0005, 0.003, 0.02])).astype(np.uint8)
        bits = oracle.pack_hap_major(m)
        bm = ctx.upload_dense(m, keep_hap_major=True)
        if it % 4 == 3:
            bm = bm.compact()
        wins = [(k * step, k * step + wlen, (0, wlen, 50000)[(it + k) % 3]) for k in range(nwin)]
        thr = thrs[it % len(thrs)]
        rd = None if it % 5 == 0 else it % 7
        inP = None if it % 2 == 0 else (rng.random(n) < (0.05, 0.5, 0.9)[it % 3]).astype(np.uint8)
        inA = (rng.random(n) < (0.0, 0.02, 0.3, 0.5, 1.0)[it % 5]).astype(np.uint8)
        inB = (rng.random(n) < (0.5, 0.3, 1.0, 0.02, 0.0)[it % 5]).astype(np.uint8)
        if it % 3 == 1:
            inB &= ~inA & 1
        res = bm.pairwise_scan(wins, inP, inA, inB, kind="match", threshold=thr, round_digits=rd, s_scope=2)
        sel = np.arange(n) if inP is None else np.nonzero(inP)[0]
        for (a, b, L), r in zip(wins, res):
            sim = oracle.identity(oracle.pairwise_counts(bits, n, a, b), b - a, 0)
            pi, ps, _, G = oracle.pica2(sim[np.ix_(sel, sel)], thr, L if L else None, rd)
            what = (n, W, (a, b, L), thr, rd)
            assert int(r["n_groups"]) == G, what
            for k, v in (("pi", pi), ("pi_site", ps)):
                got = float(r[k])
                assert (got != got and v != v) or rel_close(got, v, REL, 0.0), what + (k, got, v)
            h, _ = oracle.hfst(sim, inA, inB, L if L else None, rd)
            for k, v in h.items():
                assert stat_close(k, float(r[k]), v, h["dxy"]), what + (k, float(r[k]), v)
        bm.free()


def test_pairwise_scan_in_several_chunks(ctx, oracle):
    """A call with more windows than Gram matrices fit one chunk of the scratch (8192): the records must not depend on where the
    chunks were cut — equal, byte for byte, to the same windows asked for in two halves — for disjoint windows and for sliding
    ones (shared segment matrices; a window's segments must not straddle a cut); a few windows against the oracle."""
    rng = np.random.default_rng(77)
    n, wl, nwin = 40, 24, 9000
    for step in (wl, wl // 3):
        W = (nwin - 1) * step + wl
        f = (rng.random((6, W)) < 0.5).astype(np.uint8)
        m = f[rng.integers(0, 6, size=n)] ^ (rng.random((n, W)) < 0.01).astype(np.uint8)
        bm = ctx.upload_dense(m, keep_hap_major=True)
        inA = (np.arange(n) < 15).astype(np.uint8); inB = (np.arange(n) >= 20).astype(np.uint8)
        wins = [(k * step, k * step + wl, wl) for k in range(nwin)]
        kw = dict(kind="match", threshold=0.9, round_digits=3, s_scope=2)
        whole = bm.pairwise_scan(wins, None, inA, inB, **kw)
        halves = np.concatenate([bm.pairwise_scan(wins[:4321], None, inA, inB, **kw), bm.pairwise_scan(wins[4321:], None, inA, inB, **kw)])
        assert whole.tobytes() == halves.tobytes(), step
        bits = oracle.pack_hap_major(m)
        for k in (0, 4095, 4096, 4499, 4500, 8191, 8192, nwin - 1):
            a, b, L = wins[k]
            sim = oracle.identity(oracle.pairwise_counts(bits, n, a, b), b - a, 0)
            pi, ps, _, G = oracle.pica2(sim, 0.9, L, 3)
            assert int(whole[k]["n_groups"]) == G and rel_close(float(whole[k]["pi"]), pi, REL, 0.0), (step, k)
            h, _ = oracle.hfst(sim, inA, inB, L, 3)
            for key, v in h.items():
                assert stat_close(key, float(whole[k][key]), v, h["dxy"]), (step, k, key)
        bm.free()
