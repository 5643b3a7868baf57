// rt_device.h — the render kernel for gfx950 (CDNA4): one wavefront per 8x8 pixel tile.
//
// What it computes is the reference's `render` kernel and everything it inlines
// (/root/reference/src/ray_tracing/: kernels.py:6-73, trace.py:7-133, common.py, intersections.py),
// in IEEE float64 with separate multiply/add roundings (this file is compiled with
// -ffp-contract=off), so results are bit-identical to the reference's Python arithmetic.
//
// How it is organised is not the reference's one-thread-per-pixel translation:
//   * a 64-lane wavefront owns an 8x8 tile (8 consecutive y per x-row = the contiguous
//     direction of the (3,w,h) frame); 4 waves per workgroup take 4 tiles consecutive in y;
//   * the scene (float64-widened sphere/plane/light/material records, packed by the host) is
//     staged once per workgroup into LDS and read with wave-uniform (broadcast) ds_reads;
//   * every scene query normalises its direction ONCE (the reference re-normalises per sphere,
//     intersections.py:13 — same value every time) and works on the quadratic scaled by 1/4,
//     which is exact in binary floating point (see sphere_*() below);
//   * closest-hit keeps the smallest positive numerator and divides once per query;
//   * shadow queries are any-hit: no sqrt/divide unless a decision is within rounding reach,
//     and the sphere loop exits as soon as a wave ballot says every live lane is occluded;
//   * no MFMA: there is no dense contraction on this path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#pragma clang fp contract(off)

namespace rt {

constexpr int TILE = 8;            // 8x8 pixels per wavefront
constexpr int WAVES_PER_WG = 4;
constexpr int WG_THREADS = 64 * WAVES_PER_WG;
constexpr int SPH_STRIDE = 8;      // doubles per sphere record: cx,cy,cz,r2, R,G,B,pad
constexpr int PL_STRIDE = 16;      // ox,oy,oz,nx,ny,nz, Nx,Ny,Nz, bNx,bNy,bNz, R,G,B,pad
constexpr int LT_STRIDE = 4;       // x,y,z,pad

struct KParams {
    const double *scene;       // packed records: S spheres, then P planes, then L lights
    const double *pixel_loc;   // explicit (3,w,h) grid or nullptr (closed-form ray generation)
    uint8_t *out_u8;           // or nullptr
    float *out_f32;            // or nullptr
    long long plane_stride;    // elements between colour planes of the output
    int w, h, x0, x1;
    int S, P, L, depth;
    int aa, u8_rgb, tiles_y, ntiles;
    double px, y0, dy, z0, dz;
    double cam_o[3];
    double cam_R[9];
    double amb, lamb;
    double refl_pow[16];
};

struct V3 { double x, y, z; };

__device__ __forceinline__ double dot3(const V3 &a, const V3 &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }   // common.py:35-37

// common.py:28-32 — three true divisions, not a multiply by the reciprocal.
__device__ __forceinline__ V3 normalize3(const V3 &v)
{
    double n = __builtin_sqrt(v.x * v.x + v.y * v.y + v.z * v.z);
    return V3{v.x / n, v.y / n, v.z / n};
}

// ---------------------------------------------------------------------------------------------
// normalize() of a vector that is ALREADY unit length (every direction a scene query receives is
// the output of a normalize) — bit-identical to normalize3(), without the generic sqrt and divides.
//
//   nn = fl(x²+y²+z²) lies within a few ulp of 1.  With eps = 2^-52:
//     nn = 1 + k·eps   (k >= 0):  sqrt = 1 + (k/2)eps - (k²/8)eps² ..  ->  RN = 1 + floor(k/2)·eps
//     nn = 1 - j·eps/2 (j >= 1):  sqrt = 1 - (j/4)eps - ..             ->  RN = 1 - ceil(j/2)·eps/2
//   (the second-order term only decides which side of a midpoint the value falls on), and both k and
//   j are differences of the IEEE bit patterns from that of 1.0, so nrm = RN(sqrt(nn)) is integer work.
//   y = RN(1/nrm) follows the same way:  nrm = 1 + m·eps -> y = 1 - m·eps (= 1 - 2m·eps/2);
//   nrm = 1 - i·eps/2 -> y = 1 + ceil(i/2)·eps.
//   Each quotient x/nrm is then  q1 = fma(-x, nrm-1, x)  (faithful: off by |x|·(nrm-1)² at most),
//   r = fma(-q1, nrm, x) (the exact remainder),  q = fma(r, y, q1)  — Markstein's final correction,
//   which returns the correctly rounded quotient RN(x/nrm) given y = RN(1/nrm) and a faithful q1.
// Falls back to the generic path (wave-uniformly) if any live lane's nn is not within 2^-32 of 1.
// tests/test_algorithms.py replays this routine on the CPU against sqrt-and-divide on 10^7 vectors.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ V3 renormalize_unit(const V3 &d)
{
    const double nn = d.x * d.x + d.y * d.y + d.z * d.z;
    constexpr long long ONE = 0x3FF0000000000000ll;
    const long long k = __double_as_longlong(nn) - ONE;
    const bool near1 = (k > -(1ll << 20)) && (k < (1ll << 20));
    if (__ballot(!near1) != 0ull) return normalize3(d);
    long long nb, yb;
    if (k >= 0) { const long long m = k >> 1; nb = ONE + m; yb = ONE - 2 * m; }
    else        { const long long i = (1 - k) >> 1; nb = ONE - i; yb = ONE + ((i + 1) >> 1); }
    const double nrm = __longlong_as_double(nb), y = __longlong_as_double(yb);
    const double dl = nrm - 1.0;                                   // exact
    V3 q{__builtin_fma(-d.x, dl, d.x), __builtin_fma(-d.y, dl, d.y), __builtin_fma(-d.z, dl, d.z)};
    const V3 r{__builtin_fma(-q.x, nrm, d.x), __builtin_fma(-q.y, nrm, d.y), __builtin_fma(-q.z, nrm, d.z)};
    q = V3{__builtin_fma(r.x, y, q.x), __builtin_fma(r.y, y, q.y), __builtin_fma(r.z, y, q.z)};
    return q;
}

enum { HIT_NONE = 0, HIT_SPHERE = 1, HIT_PLANE = 2 };

// ---------------------------------------------------------------------------------------------
// Sphere test, intersections.py:6-38, restated on the quadratic divided by 4.
//   reference:  b = 2s, disc = b*b - (4a)*c, num = -b -/+ sqrt(disc), t = num / (2a)
//   here:       D = s*s - a*c,  q = sqrt(D),  n = -s -/+ q,           t = n / a
// Scaling by powers of two commutes with every rounding involved (no overflow/underflow at
// scene magnitudes): disc = 4D, sqrt(disc) = 2q, num = 2n, t identical bit for bit.
// "Behind" rule: if s >= 0 and c >= 0 then D <= fl(s*s), so q <= s and both numerators are <= 0:
// the reference returns a miss, and so do we without evaluating the sqrt.
// ---------------------------------------------------------------------------------------------

// trace.py:7-41, closest hit.  R = normalize(d) and a = R.R are computed once per query.
__device__ __forceinline__ void closest_hit(const double *__restrict__ lds, int S, int P, const V3 &o, const V3 &d,
                                            double &t_out, int &idx_out, int &type_out)
{
    const V3 R = renormalize_unit(d);                         // == normalize(d), intersections.py:13
    const double a = dot3(R, R);
    double bestn = __builtin_inf();
    int bidx = -1;
    for (int k = 0; k < S; ++k) {
        const double *g = lds + k * SPH_STRIDE;
        const V3 Lv{o.x - g[0], o.y - g[1], o.z - g[2]};     // :16
        const double s = dot3(Lv, R);                         // b/2
        const double cc = dot3(Lv, Lv) - g[3];                // :21 (g[3] = float32 r*r, widened)
        const double D = s * s - a * cc;                      // disc/4
        if (D >= 0.0 && !(s >= 0.0 && cc >= 0.0)) {
            const double q = __builtin_sqrt(D);
            double n = -s - q;                                // :28
            if (!(n > 0.0)) n = -s + q;                       // :33
            if (n > 0.0 && n < bestn) { bestn = n; bidx = k; }   // first smallest wins (strict <), trace.py:26
        }
    }
    double best = 999.0;                                      // trace.py:17
    int idx = -1, type = HIT_NONE;
    if (bidx >= 0) {
        const double t = bestn / a;                           // :31 / :36, once per query
        if (best > t && t > 0.0) { best = t; idx = bidx; type = HIT_SPHERE; }
    }
    const double *pl = lds + S * SPH_STRIDE;
    for (int k = 0; k < P; ++k) {                             // intersections.py:41-68
        const double *g = pl + k * PL_STRIDE;
        const V3 n{g[3], g[4], g[5]};
        const double den = dot3(d, n);                        // :52
        if (!(__builtin_fabs(den) < 0.001)) {                 // :55
            const V3 LP{g[0] - o.x, g[1] - o.y, g[2] - o.z};  // :59
            const double t = dot3(LP, n) / den;               // :61-63
            if (best > t && t > 0.0) { best = t; idx = k; type = HIT_PLANE; }
        }
    }
    t_out = best; idx_out = idx; type_out = type;
}

// trace.py:92-96: the shadow query only asks "does anything report 0 < t < 999" (any hit).
// Called with the lanes that need the answer active; returns true if occluded.
__device__ __forceinline__ bool any_hit(const double *__restrict__ lds, int S, int P, const V3 &o, const V3 &d)
{
    const V3 R = renormalize_unit(d);                         // == normalize(d), intersections.py:13
    const double a = dot3(R, R);
    const bool a_sane = (a > 0.999999 && a < 1.000001);
    bool occ = false;
    for (int k = 0; k < S; ++k) {
        if (__ballot(!occ) == 0ull) break;                    // every live lane already occluded
        if (!occ) {
            const double *g = lds + k * SPH_STRIDE;
            const V3 Lv{o.x - g[0], o.y - g[1], o.z - g[2]};
            const double s = dot3(Lv, R);
            const double cc = dot3(Lv, Lv) - g[3];
            const double D = s * s - a * cc;
            if (D >= 0.0 && !(s >= 0.0 && cc >= 0.0)) {
                // s < 0: the larger numerator n2 = -s + q is positive, so a positive root exists and
                // the reference's t is at most n2/a.  n2 <= 998 (from -s < 499, q <= 499) and a within
                // 1e-6 of 1 give t < 999: occluded, decided without sqrt or divide.
                if (s < 0.0 && -s < 499.0 && D < 249001.0 && a_sane) {
                    occ = true;
                } else {                                      // origin inside the sphere, or a far hit: exact path
                    const double q = __builtin_sqrt(D);
                    double n = -s - q;
                    if (!(n > 0.0)) n = -s + q;
                    if (n > 0.0) {
                        const double t = n / a;
                        if (999.0 > t && t > 0.0) occ = true;
                    }
                }
            }
        }
    }
    const double *pl = lds + S * SPH_STRIDE;
    for (int k = 0; k < P; ++k) {
        if (__ballot(!occ) == 0ull) break;
        if (!occ) {
            const double *g = pl + k * PL_STRIDE;
            const V3 n{g[3], g[4], g[5]};
            const double den = dot3(d, n);
            if (!(__builtin_fabs(den) < 0.001)) {
                const V3 LP{g[0] - o.x, g[1] - o.y, g[2] - o.z};
                const double num = dot3(LP, n);
                const double an = __builtin_fabs(num), ad = __builtin_fabs(den);
                const bool same_sign = (num > 0.0 && den > 0.0) || (num < 0.0 && den < 0.0);
                if (same_sign) {
                    // t = num/den > 0.  t < 999 is certain when |num| < 998|den| and impossible when
                    // |num| > 1000|den|; only in between is the rounded quotient needed.
                    if (an < 998.0 * ad) occ = true;
                    else if (!(an > 1000.0 * ad)) { const double t = num / den; if (999.0 > t && t > 0.0) occ = true; }
                }
            }
        }
    }
    return occ;
}

// trace.py:44-112.  On entry `alive` lanes carry a ray (o,d); on exit `alive` is false for lanes
// that missed (the reference's 404 sentinels), rgb is this bounce's colour, (o,d) the next ray.
__device__ __forceinline__ void trace_bounce(const double *__restrict__ lds, const KParams &p, bool &alive,
                                             V3 &o, V3 &d, V3 &rgb)
{
    const int S = p.S, P = p.P, L = p.L;
    rgb = V3{0.0, 0.0, 0.0};
    double t = 999.0; int idx = -1, type = HIT_NONE;
    if (alive) closest_hit(lds, S, P, o, d, t, idx, type);                    // :53 (idle lanes masked off)
    alive = alive && (type != HIT_NONE);                                      // :56-57
    if (alive) {
        V3 Pt{o.x + t * d.x, o.y + t * d.y, o.z + t * d.z};                   // :60 (1.0*o is exact)
        V3 col, N, bN;
        if (type == HIT_SPHERE) {                                             // :63-66
            const double *g = lds + idx * SPH_STRIDE;
            col = V3{g[4], g[5], g[6]};
            N = normalize3(V3{Pt.x - g[0], Pt.y - g[1], Pt.z - g[2]});        // common.py:94-101
            bN = V3{0.0002 * N.x, 0.0002 * N.y, 0.0002 * N.z};
        } else {                                                              // :68-71
            const double *g = lds + S * SPH_STRIDE + idx * PL_STRIDE;
            N = V3{g[6], g[7], g[8]};                                         // float32-renormalised, host-side
            bN = V3{g[9], g[10], g[11]};                                      // BIAS*N as the reference rounds it
            col = V3{g[12], g[13], g[14]};
        }
        rgb = V3{p.amb * col.x, p.amb * col.y, p.amb * col.z};                // :77 (0 + amb*col)
        Pt = V3{Pt.x + bN.x, Pt.y + bN.y, Pt.z + bN.z};                       // :82-83
        const double *lt = lds + S * SPH_STRIDE + P * PL_STRIDE;
        for (int m = 0; m < L; ++m) {                                         // :86-102
            const double *g = lt + m * LT_STRIDE;
            const V3 Ld = normalize3(V3{g[0] - Pt.x, g[1] - Pt.y, g[2] - Pt.z});   // common.py:84-91
            const bool occluded = any_hit(lds, S, P, Pt, Ld);                 // :92-96
            const double k = p.lamb * dot3(Ld, N);                            // :99
            if (!occluded && k > 0.0) {                                       // :101-102
                rgb = V3{rgb.x + k * col.x, rgb.y + k * col.y, rgb.z + k * col.z};
            }
        }
        const double c2 = -2.0 * dot3(d, N);                                  // common.py:113-120
        const V3 Rd = normalize3(V3{d.x + c2 * N.x, d.y + c2 * N.y, d.z + c2 * N.z});
        o = V3{Pt.x + 0.0002 * Rd.x, Pt.y + 0.0002 * Rd.y, Pt.z + 0.0002 * Rd.z};   // :110
        d = Rd;
    }
}

// trace.py:115-133
__device__ __forceinline__ V3 sample(const double *__restrict__ lds, const KParams &p, bool alive, V3 o, V3 d)
{
    V3 acc{0.0, 0.0, 0.0};
    for (int b = 0; b <= p.depth; ++b) {
        if (__ballot(alive) == 0ull) break;                                   // wave-uniform exit
        V3 rgb;
        trace_bounce(lds, p, alive, o, d, rgb);
        if (b == 0) acc = rgb;                                                // :120
        else {                                                                // :131 (a missed bounce adds pow*0)
            const double wgt = p.refl_pow[b - 1];
            acc = V3{acc.x + wgt * rgb.x, acc.y + wgt * rgb.y, acc.z + wgt * rgb.z};
        }
    }
    return acc;
}

__device__ __forceinline__ V3 pixel_P(const KParams &p, int x, int y)
{
    if (p.pixel_loc) {                                                        // kernels.py:19
        const size_t wh = (size_t)p.w * p.h, o = (size_t)x * p.h + y;
        return V3{p.pixel_loc[o], p.pixel_loc[wh + o], p.pixel_loc[2 * wh + o]};
    }
    return V3{p.px, (double)x * p.dy + p.y0, (double)y * p.dz + p.z0};        // scene/camera.py:18-26
}

__device__ __forceinline__ V3 primary_dir(const KParams &p, const V3 &P)
{
    const V3 v{p.cam_R[0] * P.x + p.cam_R[1] * P.y + p.cam_R[2] * P.z,       // kernels.py:22, common.py:40-49
               p.cam_R[3] * P.x + p.cam_R[4] * P.y + p.cam_R[5] * P.z,
               p.cam_R[6] * P.x + p.cam_R[7] * P.y + p.cam_R[8] * P.z};
    return normalize3(v);                                                     // kernels.py:23
}

// common.py:52-57: min(max(0, int(round(c))), 255), round half to even (v_rndne_f64).
__device__ __forceinline__ uint8_t clip_color(double c)
{
    if (!(c == c)) return 0;
    if (c <= -0.5) return 0;
    if (c >= 255.5) return 255;
    const int i = (int)__builtin_rint(c);
    return (uint8_t)(i < 0 ? 0 : (i > 255 ? 255 : i));
}

__global__ __launch_bounds__(WG_THREADS) void render_kernel(const KParams p)
{
    extern __shared__ double lds[];
    {   // stage the packed scene once per workgroup
        const int n = p.S * SPH_STRIDE + p.P * PL_STRIDE + p.L * LT_STRIDE;
        for (int i = threadIdx.x; i < n; i += WG_THREADS) lds[i] = p.scene[i];
    }
    __syncthreads();

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tile = blockIdx.x * WAVES_PER_WG + wave;
    if (tile >= p.ntiles) return;                                             // whole wave, after the barrier
    const int tx = tile / p.tiles_y, ty = tile - tx * p.tiles_y;
    const int x = p.x0 + tx * TILE + (lane >> 3);
    const int y = ty * TILE + (lane & 7);
    const bool inb = (x < p.x1) && (y < p.h);
    const int xc = inb ? x : p.x0, yc = inb ? y : 0;                          // keep addresses valid for idle lanes

    const V3 o{p.cam_o[0], p.cam_o[1], p.cam_o[2]};                           // kernels.py:16
    const V3 Pp = pixel_P(p, xc, yc);
    V3 c = sample(lds, p, inb, o, primary_dir(p, Pp));                        // kernels.py:26
    double R = c.x, G = c.y, B = c.z;

    if (p.aa) {                                                               // kernels.py:29-65
        const bool interior = inb && x >= 1 && x <= p.w - 2 && y >= 1 && y <= p.h - 2;
        if (__ballot(interior) != 0ull) {
            // neighbour offsets (dx,dy)+1 packed 2 bits each, in the order of kernels.py:53:
            // left, right, top(y+1), bottom(y-1), top-left, top-right, bottom-left, bottom-right
            constexpr unsigned NBX = 0x8858u, NBY = 0x0A25u;
#pragma unroll 1
            for (int k = 0; k < 8; ++k) {
                const int ddx = (int)((NBX >> (2 * k)) & 3u) - 1, ddy = (int)((NBY >> (2 * k)) & 3u) - 1;
                const V3 Pn = pixel_P(p, interior ? x + ddx : xc, interior ? y + ddy : yc);
                const V3 Pt{0.5 * Pp.x + 0.5 * Pn.x, 0.5 * Pp.y + 0.5 * Pn.y, 0.5 * Pp.z + 0.5 * Pn.z};   // :43-50
                const V3 s = sample(lds, p, interior, o, primary_dir(p, Pt));
                if (interior) { R += s.x; G += s.z; B += s.y; }               // :58-60 (G += B_s; B += G_s)
            }
            if (interior) { R = R / 9; G = G / 9; B = B / 9; }                // :63-65
        }
    }

    if (inb) {
        const long long off = (long long)(x - p.x0) * p.h + y;
        if (p.out_u8) {                                                       // kernels.py:69-73, common.py:60-63
            const uint8_t r8 = clip_color(R), g8 = clip_color(G), b8 = clip_color(B);
            p.out_u8[off] = r8;
            p.out_u8[p.plane_stride + off] = p.u8_rgb ? g8 : b8;
            p.out_u8[2 * p.plane_stride + off] = p.u8_rgb ? b8 : g8;
        }
        if (p.out_f32) {
            p.out_f32[off] = (float)R;
            p.out_f32[p.plane_stride + off] = (float)G;
            p.out_f32[2 * p.plane_stride + off] = (float)B;
        }
    }
}

}  // namespace rt
