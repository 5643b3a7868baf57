/* CPU replay of rt_device.h's normalize3() fast path: sqrt by one rsq seed + Goldschmidt/Newton steps (the
 * sequence the AMDGPU backend emits for f64 sqrt, without its range scaling), the reciprocal of the norm taken
 * from that iteration's own 1/(2g) estimate plus one Newton step and shared by the three quotients, one fma
 * correction per quotient (the backend's f64 division without div_scale/div_fixup).
 * Seeds carry a relative error of up to 2^-24 here, i.e. WORSE than v_rsq_f64 / v_rcp_f64, so agreement with
 * sqrt()/division on every sample is a conservative check.  One sample in four sits at the edges of the guard
 * (components down to 2^-200, |v|^2 up to 2^400).  Built and run by tests/test_algorithms.py. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static uint64_t s[2] = {0x9E3779B97F4A7C15ull, 0xD1B54A32D192ED03ull};
static inline uint64_t rnd(void) { uint64_t a = s[0], b = s[1]; s[0] = b; a ^= a << 23; s[1] = a ^ b ^ (a >> 17) ^ (b >> 26); return s[1] + b; }
static inline double urand(void) { return (double)(rnd() >> 11) * (1.0 / 9007199254740992.0); }

static void normalize_ref(const double v[3], double out[3])
{
    double n = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    out[0] = v[0] / n; out[1] = v[1] / n; out[2] = v[2] / n;
}

static void normalize_fast(const double v[3], double out[3], double *norm)
{
    const double nn = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
    double y = (1.0 / sqrt(nn)) * (1.0 + (urand() * 2 - 1) * 0x1p-24);   /* stand-in for v_rsq_f64 */
    double g = nn * y, h = 0.5 * y;
    double r = fma(-h, g, 0.5);
    g = fma(g, r, g); h = fma(h, r, h);
    double d = fma(-g, g, nn); g = fma(d, h, g);
    d = fma(-g, g, nn); g = fma(d, h, g);
    *norm = g;
    double r0 = 2.0 * h;                                   /* h ~ 1/(2g) from the sqrt iteration: no v_rcp_f64 */
    double e = fma(-g, r0, 1.0); r0 = fma(r0, e, r0);
    for (int c = 0; c < 3; ++c) {
        double q = v[c] * r0;
        double rem = fma(-g, q, v[c]);
        out[c] = fma(rem, r0, q);
    }
}

int main(int argc, char **argv)
{
    long n = argc > 1 ? atol(argv[1]) : 10000000, bad = 0, badsqrt = 0;
    if (argc > 2) { s[0] ^= (uint64_t)atoll(argv[2]) * 0x9E3779B97F4A7C15ull; s[1] += (uint64_t)atoll(argv[2]); }
    for (long it = 0; it < n; ++it) {
        double v[3], a[3], b[3], nr;
        int mode = it & 7;
        double sc = mode == 0 ? 1.0 : exp((urand() - 0.5) * (mode == 1 ? 8 : 60));   /* magnitudes 1e-13 .. 1e13 */
        if (mode == 4) sc = ldexp(1.0, 190 + (int)(rnd() % 9));                      /* |v|^2 just below 2^400 */
        if (mode == 5) sc = ldexp(1.0, -(185 + (int)(rnd() % 14)));                  /* components around 2^-200 */
        for (int c = 0; c < 3; ++c) v[c] = (urand() * 2 - 1) * sc;
        if (mode == 3) v[(int)(rnd() % 3)] *= exp(-urand() * 60);                    /* one component much smaller */
        if (mode == 6) v[(int)(rnd() % 3)] = ldexp(0.5 + urand() * 0.5, -199);        /* one component at the guard, others O(1) */
        {   /* the guard of rt_device.h: outside it the kernel takes the generic path */
            const double cmin = fmin(fmin(fabs(v[0]), fabs(v[1])), fabs(v[2]));
            if (!(cmin >= 0x1p-200 && v[0] * v[0] + v[1] * v[1] + v[2] * v[2] <= 0x1p400)) continue;
        }
        normalize_ref(v, a);
        normalize_fast(v, b, &nr);
        if (nr != sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2])) badsqrt++;
        if (memcmp(a, b, sizeof a) != 0) {
            if (bad++ < 5) fprintf(stderr, "MISMATCH v=(%a,%a,%a) ref=(%a,%a,%a) got=(%a,%a,%a)\n", v[0], v[1], v[2], a[0], a[1], a[2], b[0], b[1], b[2]);
        }
    }
    printf("checked=%ld sqrt_mismatches=%ld mismatches=%ld\n", n, badsqrt, bad);
    return (bad || badsqrt) ? 1 : 0;
}
