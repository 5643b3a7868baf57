/* CPU replay of rt_device.h's div_inrange() and sqrt_inrange(): the AMDGPU backend's f64 division and square root
 * without their range handling (v_div_scale / v_div_fixup, ldexp scaling), which are identities inside the exponent
 * range the scene queries work in.
 *   a / b:    r = rcp(b); twice { e = fma(-b,r,1); r = fma(r,e,r); }  q = a*r; res = fma(fma(-b,q,a), r, q)
 *   sqrt(x):  y = rsq(x); g = x*y; h = y/2; r = fma(-h,g,1/2); g = fma(g,r,g); h = fma(h,r,h);
 *             twice { d = fma(-g,g,x); g = fma(d,h,g); }; x == 0 -> x
 * Seeds carry a relative error of up to 2^-24 here, i.e. WORSE than v_rcp_f64 / v_rsq_f64, so agreement with the
 * correctly rounded / and sqrt() on every sample is a conservative check.  Operand ranges: denominators 1e-3 .. 1e40 and
 * within an ulp of 1 (t = n/a), numerators 0 and 1e-130 .. 1e80, radicands 0 and 1e-130 .. 1e80 — wider than what
 * float32 scenes produce (see rt_device.h).  Built and run by tests/test_algorithms.py. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

static uint64_t s[2] = {0x9E3779B97F4A7C15ull, 0xD1B54A32D192ED03ull};
static inline uint64_t rnd(void) { uint64_t a = s[0], b = s[1]; s[0] = b; a ^= a << 23; s[1] = a ^ b ^ (a >> 17) ^ (b >> 26); return s[1] + b; }
static inline double urand(void) { return (double)(rnd() >> 11) * (1.0 / 9007199254740992.0); }
static inline double noise(void) { return 1.0 + (urand() * 2 - 1) * 0x1p-24; }

static double div_inrange(double a, double b)
{
    double r = (1.0 / b) * noise();                        /* stand-in for v_rcp_f64 */
    double e = fma(-b, r, 1.0); r = fma(r, e, r);
    e = fma(-b, r, 1.0); r = fma(r, e, r);
    const double q = a * r;
    return fma(fma(-b, q, a), r, q);
}

static double sqrt_inrange(double x)
{
    const double y = (x > 0.0 ? 1.0 / sqrt(x) : INFINITY) * noise();   /* stand-in for v_rsq_f64 */
    double g = x * y, h = 0.5 * y;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g); h = fma(h, r, h);
    double d = fma(-g, g, x); g = fma(d, h, g);
    d = fma(-g, g, x); g = fma(d, h, g);
    return x > 0.0 ? g : x;
}

static double magnitude(int wide) { return exp((urand() - 0.5) * (wide ? 480.0 : 40.0)); }   /* 1e-104..1e104 / 2e-9..5e8 */

int main(int argc, char **argv)
{
    long n = argc > 1 ? atol(argv[1]) : 10000000, baddiv = 0, badsqrt = 0;
    if (argc > 2) { s[0] ^= (uint64_t)atoll(argv[2]) * 0x9E3779B97F4A7C15ull; s[1] += (uint64_t)atoll(argv[2]); }
    for (long it = 0; it < n; ++it) {
        const int mode = it & 7;
        double a = (urand() * 2 - 1) * magnitude(mode & 1), b;
        if (mode < 3) b = 1.0 + ((double)(rnd() % 33) - 16.0) * 0x1p-52;            /* t = n / a: a = R.R of a unit vector */
        else {
            b = (urand() < 0.5 ? -1.0 : 1.0) * (0.001 + urand()) * (mode == 7 ? exp(urand() * 92.0) : exp(urand() * 8.0));   /* |den| >= 0.001, up to 1e40 */
        }
        if (mode == 6) a = 0.0;
        if (fabs(a) < 1e-130 && a != 0.0) a = 1e-130;
        if (fabs(a) > 1e80) a = copysign(1e80, a);
        const double want = a / b, got = div_inrange(a, b);
        if (want != got && !(want == 0.0 && got == 0.0)) { if (baddiv++ < 5) fprintf(stderr, "DIV a=%a b=%a want=%a got=%a\n", a, b, want, got); }
        double x = mode == 6 ? 0.0 : fabs(a) * magnitude(0);
        if (x != 0.0 && x < 1e-130) x = 1e-130;
        if (x > 1e80) x = 1e80;
        if (mode == 5) { const double q = floor(urand() * 1e6) + 1.0; x = q * q; }   /* perfect squares */
        const double ws = sqrt(x), gs = sqrt_inrange(x);
        if (ws != gs) { if (badsqrt++ < 5) fprintf(stderr, "SQRT x=%a want=%a got=%a\n", x, ws, gs); }
    }
    printf("checked=%ld div_mismatches=%ld sqrt_mismatches=%ld\n", n, baddiv, badsqrt);
    return (baddiv || badsqrt) ? 1 : 0;
}
