/* Host restatement of mod_m_f64 (csrc/rbf_kernels_lds.h): h mod m through one FP64 fma + one 24-bit multiply-add,
 * for 2^15 <= m < 2^23.  Every step below is the C twin of one gfx950 instruction:
 *   hd  = (double)h                      v_cvt_f64_u32 x2, v_ldexp_f64, v_add_f64  (round to nearest)
 *   t   = fma(hd, -1/m, 1.5 * 2^52)      v_fma_f64        (IEEE, round to nearest even)
 *   r   = (lo32(t) & 0xFFFFFF) * (m & 0xFFFFFF) + lo32(h)   v_mad_u32_u24
 *   rs  = sign-extend the low 24 bits     v_bfe_i32
 *   pos = min(rs, rs + m) unsigned        v_add_u32, v_min_u32
 * and the result must equal h % m for EVERY h: checked here against 64-bit integer arithmetic on random h, on h
 * adjacent to multiples of m (where a wrong quotient estimate shows), on the extremes of the 64-bit range and on
 * every m of a sweep plus the range ends.  Built and run by tests/test_host_cpu.py (no GPU needed); the GPU parity
 * tests then pin the device code to the oracle. */
#include <inttypes.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static uint32_t mod_m_f64(uint64_t h, uint32_t m)
{
    const double hd = (double)h;
    const double ninv = -1.0 / (double)m;
    const double t = fma(hd, ninv, 0x1.8p52);
    uint64_t bits;
    memcpy(&bits, &t, 8);
    const uint32_t nq = (uint32_t)bits;
    const uint32_t r = (uint32_t)((uint64_t)(nq & 0xFFFFFFu) * (uint64_t)(m & 0xFFFFFFu)) + (uint32_t)h;
    const uint32_t rs = (uint32_t)(((int32_t)(r << 8)) >> 8);
    const uint32_t a = rs, b = rs + m;
    return a < b ? a : b;
}

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd(void)
{
    uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

static uint64_t checked = 0;
static int check(uint64_t h, uint32_t m)
{
    ++checked;
    const uint32_t got = mod_m_f64(h, m), want = (uint32_t)(h % m);
    if (got != want) {
        printf("MISMATCH h=%" PRIu64 " m=%u got=%u want=%u\n", h, m, got, want);
        return 1;
    }
    return 0;
}

static int check_m(uint32_t m, int nrandom)
{
    int bad = 0;
    const uint64_t qmax = UINT64_MAX / m;
    for (int i = 0; i < nrandom; ++i) bad += check(rnd(), m);
    for (int i = 0; i < nrandom / 4; ++i) {                      /* around multiples of m, over the whole quotient range */
        const uint64_t q = rnd() % (qmax + 1);
        const uint64_t base = q * m;
        bad += check(base, m);
        if (base) bad += check(base - 1, m);
        if (base + 1 > base) bad += check(base + 1, m);
        if (q < qmax) { bad += check(base + m - 1, m); bad += check(base + m / 2, m); bad += check(base + (m + 1) / 2, m); }
    }
    const uint64_t edge[] = {0, 1, m - 1u, m, m + 1u, UINT64_MAX, UINT64_MAX - 1, UINT64_MAX - m, qmax * m, qmax * m - 1, qmax * m + (UINT64_MAX - qmax * m),
                             1ull << 63, (1ull << 63) - 1, (1ull << 53) + 1, (1ull << 53) - 1, 0xFFFFFFFFull, 0x100000000ull};
    for (size_t i = 0; i < sizeof edge / sizeof edge[0]; ++i) bad += check(edge[i], m);
    return bad;
}

int main(int argc, char **argv)
{
    const int nrandom = argc > 1 ? atoi(argv[1]) : 2000;
    int bad = 0;
    const uint32_t lo = 1u << 15, hi = (1u << 23) - 1u;
    bad += check_m(lo, nrandom * 50);
    bad += check_m(lo + 1, nrandom * 50);
    bad += check_m(hi, nrandom * 50);
    bad += check_m(hi - 1, nrandom * 50);
    bad += check_m(611158, nrandom * 200);                       /* the 1080p, k* = 2.3 filter */
    bad += check_m(2446471, nrandom * 200);                      /* the 2160p one */
    for (uint32_t m = lo; m <= hi && !bad; m += 1 + (uint32_t)(rnd() % 4099)) bad += check_m(m, nrandom / 10 + 8);
    for (int i = 0; i < 3000 && !bad; ++i) bad += check_m(lo + (uint32_t)(rnd() % (hi - lo + 1)), nrandom);
    printf("%s %" PRIu64 " checks\n", bad ? "FAIL" : "OK", checked);
    return bad ? 1 : 0;
}
