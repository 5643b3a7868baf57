/* CPU replay of rt_device.h's renormalize_unit(): the two-fma sqrt/reciprocal + Markstein-quotient shortcut must equal
 * sqrt-and-divide normalisation bit for bit on unit-length inputs.  Built and run by tests/test_algorithms.py. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static inline int64_t bits(double x) { int64_t u; memcpy(&u, &x, 8); return u; }
static inline double from_bits(int64_t u) { double x; memcpy(&x, &u, 8); return x; }

static void normalize_ref(const double v[3], double out[3])
{
    double n = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    out[0] = v[0] / n; out[1] = v[1] / n; out[2] = v[2] / n;
}

/* sqrt and reciprocal of a value within 2^-33 of 1, as rt_device.h evaluates them: no integer work, no select.
 * Returns 0 if nn is outside the accepted range. */
static int unit_sqrt_rcp(double nn, double *nrm_out, double *y_out)
{
    const double e = nn - 1.0;                               /* exact */
    if (!(fabs(e) < 0x1p-33)) return 0;
    const double e2 = fma(-fabs(e), 0x1p-30, e);             /* nudged toward -inf: decides the ties of 1 + e/2 */
    const double nrm = fma(e2, 0.5, 1.0);
    const double dl = nrm - 1.0;                             /* exact */
    const double y = fma(-dl, 1.0 + 0x1p-30, 1.0);
    *nrm_out = nrm; *y_out = y;
    return 1;
}

/* returns 0 if the fast path does not apply */
static int renormalize_unit(const double d[3], double out[3])
{
    const double nn = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
    double nrm, y;
    if (!unit_sqrt_rcp(nn, &nrm, &y)) return 0;
    const double dl = nrm - 1.0;
    for (int c = 0; c < 3; ++c) {
        double q = fma(-d[c], dl, d[c]);
        double r = fma(-q, nrm, d[c]);
        out[c] = fma(r, y, q);
    }
    if (nrm != sqrt(nn) || y != 1.0 / nrm) return -1;
    return 1;
}

/* every double within 2^-33 of 1: RN(sqrt) and RN(1/RN(sqrt)) against libm / IEEE division */
static long sweep(void)
{
    const int64_t ONE = 0x3FF0000000000000ll;
    long bad = 0, n = 0;
    for (int64_t k = -(1ll << 21); k <= (1ll << 20); ++k) {
        const double nn = from_bits(ONE + k);
        double nrm, y;
        if (!unit_sqrt_rcp(nn, &nrm, &y)) continue;
        ++n;
        if (nrm != sqrt(nn) || y != 1.0 / nrm) {
            if (bad++ < 5) fprintf(stderr, "SWEEP MISMATCH k=%lld nrm=%a sqrt=%a y=%a rcp=%a\n", (long long)k, nrm, sqrt(nn), y, 1.0 / nrm);
        }
    }
    printf("sweep=%ld sweep_mismatches=%ld\n", n, bad);
    return bad;
}

static uint64_t s[2] = {0x9E3779B97F4A7C15ull, 0xD1B54A32D192ED03ull};
static inline uint64_t rnd(void) { uint64_t a = s[0], b = s[1]; s[0] = b; a ^= a << 23; s[1] = a ^ b ^ (a >> 17) ^ (b >> 26); return s[1] + b; }
static inline double urand(void) { return (double)(rnd() >> 11) * (1.0 / 9007199254740992.0); }

int main(int argc, char **argv)
{
    long n = argc > 1 ? atol(argv[1]) : 10000000, bad = sweep(), fast = 0, wide = 0;
    for (long it = 0; it < n; ++it) {
        double v[3], d[3], a[3], b[3];
        int mode = it & 7;
        if (mode == 0) {            /* special direction sets: axis-aligned, 3-4-5, halves */
            static const double sp[][3] = {{1,0,0},{0,1,0},{0,0,-1},{0.6,0.8,0},{0.75,0.5,0.4330127018922193},{0.5,0.5,0.7071067811865476},
                                            {0.28,0.96,0},{0.6,0,-0.8},{0.3333333333333333,0.6666666666666666,0.6666666666666666}};
            int j = (int)(rnd() % 9);
            for (int c = 0; c < 3; ++c) v[c] = sp[j][c];
        } else {
            double sc = mode < 4 ? 1.0 : exp((urand() - 0.5) * 20);
            for (int c = 0; c < 3; ++c) v[c] = (urand() * 2 - 1) * sc;
            if (mode == 5) v[(int)(rnd() % 3)] = 0.0;
            if (mode == 6) v[(int)(rnd() % 3)] *= 1e-9;
        }
        normalize_ref(v, d);                         /* the pipeline always feeds normalize() outputs */
        if (mode == 7) { double t[3]; normalize_ref(d, t); memcpy(d, t, sizeof d); }
        if (mode == 3) {                             /* nudge components by a few ulp: |d|^2 further from 1 */
            for (int c = 0; c < 3; ++c) d[c] = from_bits(bits(d[c]) + (int64_t)(rnd() % 9) - 4);
            wide++;
        }
        if (!(d[0] == d[0])) continue;
        normalize_ref(d, a);
        int rc = renormalize_unit(d, b);
        if (rc == 0) continue;
        fast++;
        if (rc < 0 || memcmp(a, b, sizeof a) != 0) {
            if (bad++ < 5) fprintf(stderr, "MISMATCH rc=%d d=(%a,%a,%a) ref=(%a,%a,%a) got=(%a,%a,%a)\n", rc, d[0], d[1], d[2], a[0], a[1], a[2], b[0], b[1], b[2]);
        }
    }
    printf("checked=%ld fast_path=%ld nudged=%ld mismatches=%ld\n", n, fast, wide, bad);
    return bad ? 1 : 0;
}
