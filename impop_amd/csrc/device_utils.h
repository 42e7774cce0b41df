// device_utils.h — device-side building blocks (gfx950, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace impop {

typedef unsigned __int128 u128;

// ---------------------------------------------------------------------------------
// CPython round(x, ndigits) — pica2.py:83, h-fst.py:150,163 and the "%.8f" text
// round trip of run_tajd.sh:174.  CPython prints the exact binary value with
// `ndigits` decimals (ties-to-even on the exact value) and parses it back.  Done
// here in exact integer arithmetic: x = m * 2^sh, q = rint_even(m * 10^nd * 2^sh),
// result = q / 10^nd (one correctly rounded IEEE division of two exact doubles ==
// the double nearest to the decimal).  Valid for 0 <= nd <= 19.
//
// Fast path (what almost every call takes): t = x * 10^nd in fp64 is off the exact product by less than 2^-23 while
// |t| < 2^30, so unless t sits within 1e-6 of a rounding boundary k + 1/2 the nearest integer of the exact product is
// rint(t), and q / 10^nd is the same correctly rounded division the exact path ends with.  Boundary cases (and huge /
// tiny arguments) fall through to the exact integer arithmetic below.
__host__ __device__ inline double py_round(double x, int nd) {
    if (nd >= 0 && nd <= 15) {
        double p = 1.0;
        for (int i = 0; i < nd; ++i) p *= 10.0;  // exact: 10^15 < 2^53
        const double t = x * p;
        if (t > -1073741824.0 && t < 1073741824.0) {  // also false for NaN
            const double f = t - floor(t);
            if (f < 0.5 - 1e-6 || f > 0.5 + 1e-6) return rint(t) / p;
        }
    }
    union { double d; uint64_t u; } cv;
    cv.d = x;
    const uint64_t bits = cv.u;
    const int e = (int)((bits >> 52) & 0x7FF);
    if (e == 0x7FF) return x;                  // inf / nan
    uint64_t m = bits & 0xFFFFFFFFFFFFFull;
    if (e == 0 && m == 0) return x;            // +-0
    const bool neg = (bits >> 63) != 0;
    int sh;
    if (e == 0) sh = -1074; else { m |= 1ull << 52; sh = e - 1075; }
    if (sh >= 0) return x;                     // integer-valued: unchanged
    if (nd < 0) nd = 0;
    if (nd > 19) nd = 19;
    uint64_t p10 = 1;
    for (int i = 0; i < nd; ++i) p10 *= 10ull;
    const u128 prod = (u128)m * (u128)p10;     // < 2^53 * 10^19 < 2^117
    const int s = -sh;                         // >= 1
    u128 q;
    if (s >= 128) {
        q = 0;                                 // |x| * 10^nd < 2^-11: rounds to 0
    } else {
        q = prod >> s;
        const u128 rem = prod & ((((u128)1) << s) - 1);
        const u128 half = ((u128)1) << (s - 1);
        if (rem > half || (rem == half && (q & 1))) q += 1;
    }
    if (q >= (((u128)1) << 53)) return x;      // spacing of doubles near x >= 10^-nd: x already is the answer
    double r = (double)(uint64_t)q / (double)p10;
    return neg ? -r : r;
}

// ---------------------------------------------------------------------------------
// Tajima's D constants and statistic, operation order of tj_d.py:53-65.
struct TajConsts { double a1, a2, b1, b2, c1, c2, e1, e2; };

__host__ __device__ inline TajConsts tajima_consts(int64_t n) {
    double a1 = 0.0, a2 = 0.0;
    for (int64_t i = 1; i < n; ++i) a1 += 1.0 / (double)i;               // tj_d.py:41-42
    for (int64_t i = 1; i < n; ++i) a2 += 1.0 / ((double)i * (double)i); // tj_d.py:44-45
    const double dn = (double)n;
    TajConsts c;
    c.a1 = a1; c.a2 = a2;
    c.b1 = (dn + 1.0) / (3.0 * (dn - 1.0));                              // :55
    c.b2 = 2.0 * (dn * dn + dn + 3.0) / (9.0 * dn * (dn - 1.0));         // :56
    c.c1 = c.b1 - (1.0 / a1);                                            // :57
    c.c2 = c.b2 - ((dn + 2.0) / (a1 * dn)) + (a2 / (a1 * a1));           // :58
    c.e1 = c.c1 / a1;                                                    // :59
    c.e2 = c.c2 / (a1 * a1 + a2);                                        // :60
    return c;
}

__host__ __device__ inline double tajima_d_from(const TajConsts &c, double S, double pi, double *num_out,
                                                double *den_out) {
    const double num = pi - (S / c.a1);                                                      // :62
    const double nan = __builtin_nan("");
    const double den = (S > 0) ? sqrt(c.e1 * S + c.e2 * S * (S - 1.0)) : nan;                // :63
    // :65 `denominator and not math.isclose(denominator, 0.0)` — false only for den == 0
    const double D = (den != 0.0) ? num / den : nan;
    if (num_out) *num_out = num;
    if (den_out) *den_out = den;
    return D;
}

// ---------------------------------------------------------------------------------
// counter-based hash for the synthetic generator (splitmix64 finaliser)
__host__ __device__ inline uint64_t mix64(uint64_t x) {
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull;
    x ^= x >> 27; x *= 0x94d049bb133111ebull;
    x ^= x >> 31;
    return x;
}
__host__ __device__ inline uint64_t synth_hash(uint64_t seed, uint64_t stream, uint64_t site) {
    return mix64(seed + 0x9E3779B97F4A7C15ull * (site + 1) + 0xD1B54A32D192ED03ull * (stream + 1));
}

#ifdef __HIPCC__
// wave64 sum (all lanes get the total)
__device__ inline uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ inline uint64_t wave_sum_u64(uint64_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ inline double wave_sum_f64(double v) {  // fixed butterfly order: deterministic
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
#endif

}  // namespace impop
