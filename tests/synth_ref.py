"""numpy restatement of the device-side synthetic generator (impop_amd/csrc/layout.hip
synth_sb_kernel): test infrastructure used to check that what the GPU generated is what
the oracle is fed."""
import numpy as np

M1 = np.uint64(0xBF58476D1CE4E5B9)
M2 = np.uint64(0x94D049BB133111EB)
G1 = np.uint64(0x9E3779B97F4A7C15)
G2 = np.uint64(0xD1B54A32D192ED03)


def mix64(x):
    x = x.astype(np.uint64)
    with np.errstate(over="ignore"):
        x = x ^ (x >> np.uint64(30)); x = x * M1
        x = x ^ (x >> np.uint64(27)); x = x * M2
        x = x ^ (x >> np.uint64(31))
    return x


def synth_hash(seed, stream, site):
    with np.errstate(over="ignore"):
        return mix64(np.uint64(seed) + G1 * (np.asarray(site, dtype=np.uint64) + np.uint64(1))
                     + G2 * np.uint64(stream + 1))


def synth_matrix(n_hap, site_begin, site_end, seed=20251031, n_founder=8, p_founder=1e-3, p_private_word=3.2e-3):
    """-> uint8 [n_hap, site_end-site_begin] for global site indices [site_begin, site_end)."""
    s = np.arange(site_begin, site_end, dtype=np.uint64)
    W = len(s)
    wps = (n_hap + 31) // 32
    thr_f = np.uint64(int(p_founder * 4294967296.0))
    thr_p = np.uint64(int(p_private_word * 4294967296.0))
    founder_of = (synth_hash(seed, 1000, np.arange(n_hap, dtype=np.uint64)) % np.uint64(n_founder)).astype(np.int64)
    anc = (synth_hash(seed, 1, s) & np.uint64(1)).astype(np.uint8)
    flips = np.zeros((n_founder, W), dtype=np.uint8)
    for f in range(n_founder):
        flips[f] = ((synth_hash(seed, 2 + f, s) >> np.uint64(32)) < thr_f).astype(np.uint8)
    m = anc[None, :] ^ flips[founder_of]
    for k in range(wps):
        h = synth_hash(seed, 64 + k, s)
        hit = (h >> np.uint64(32)) < thr_p
        bit = (h & np.uint64(31)).astype(np.int64)
        hap = 32 * k + bit
        ok = hit & (hap < n_hap)
        m[hap[ok], np.nonzero(ok)[0]] ^= 1
    return m.astype(np.uint8)
