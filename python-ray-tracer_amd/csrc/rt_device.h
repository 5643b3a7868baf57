// rt_device.h — the render kernel for gfx950 (CDNA4): one wavefront per 8x8 pixel tile.
//
// What it computes is the reference's `render` kernel and everything it inlines
// (/root/reference/src/ray_tracing/: kernels.py:6-73, trace.py:7-133, common.py, intersections.py),
// in IEEE float64 with separate multiply/add roundings (this file is compiled with
// -ffp-contract=off), so results are bit-identical to the reference's Python arithmetic.
//
// How it is organised is not the reference's one-thread-per-pixel translation:
//   * a 64-lane wavefront owns an 8x8 tile (8 consecutive y per x-row = the contiguous
//     direction of the (3,w,h) frame); the 2 or 4 waves of a workgroup take tiles consecutive in y;
//   * the scene (float64-widened sphere/plane/light/material records, packed by the host) and its float32 cull
//     tables (built once per scene/camera by tables_kernel) are copied once per workgroup into LDS and read
//     with wave-uniform (broadcast) ds_reads;
//   * every scene query normalises its direction ONCE (the reference re-normalises per sphere,
//     intersections.py:13 — same value every time), with an exact two-fma shortcut for
//     already-unit vectors, and works on the quadratic scaled by 1/4 (exact in binary FP);
//   * closest-hit keeps the smallest positive numerator and divides once per query;
//   * shadow queries are any-hit: no sqrt/divide unless a decision is within rounding reach,
//     the sphere loop exits as soon as a wave ballot says every live lane is occluded, and
//     lanes whose Lambert term is not positive (result unused, trace.py:101) do not query;
//   * a conservative float32 pre-test ("cull") per sphere decides, for the whole wave, whether the
//     float64 test can change anything; only spheres some lane might hit get the float64 test.
//     The cull never decides a hit and never feeds a value into the result — it only skips
//     float64 evaluations whose outcome (a miss) it has certified with an explicit error margin;
//   * the kernel is bound by VALU instruction issue (0.79 of the bound priced with measured cycles per instruction class on the
//     headline config: float64 4.1-4.2 cycles, 32-bit 2.2, lane masks 4.1-4.2; profiles/r03_valu_prices.json) with the CU's single
//     scalar ALU as the second bound, so both instruction counts are what the code below economises: lane masks
//     come straight from compares, the cull's mask is built with s_cmp + s_addc, table entries are one
//     ds_read_b128 at an immediate offset of a VGPR-pinned base, wave-uniform facts travel in SGPRs;
//   * no MFMA: there is no dense contraction on this path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#pragma clang fp contract(off)

// Build-time knobs (A/B builds for tools/ab_bench.py); the defaults are the measured best on MI355X:
//   RT_PREFILTER        float32 cull in front of the float64 sphere tests            (C2 -23 %, C4 -63 %)
//   RT_FAST_NORMALIZE   shared-reciprocal normalize instead of sqrt + three divisions (C2 -10 %)
//   RT_FAST_DIVSQRT     0 = the backend's full f64 division / sqrt in the scene queries (div_inrange, sqrt_inrange)
//   RT_CLUSTER_MIN      sphere count above which the scene is stored in clusters of 8
//   RT_W_PARK, RT_W_AAPARK  waves/SIMD the LDS-parked variants are compiled for
#ifndef RT_PREFILTER
#define RT_PREFILTER 1
#endif
#ifndef RT_FAST_DIVSQRT
#define RT_FAST_DIVSQRT 1
#endif
#ifndef RT_CULL_PAIRS_WPW2
#define RT_CULL_PAIRS_WPW2 1   // the two-wave kernels too: 16 spheres -3 %, the headline -0.6 %
#endif
#ifndef RT_SHADOW_EXIT_CHECKS
#define RT_SHADOW_EXIT_CHECKS 0
#endif
#ifndef RT_FAST_NORMALIZE
#define RT_FAST_NORMALIZE 1
#endif
#ifndef RT_OPAQUE_ARGS
#define RT_OPAQUE_ARGS 1
#endif
#ifndef RT_OPAQUE_PLANE_CODE
#define RT_OPAQUE_PLANE_CODE 1
#endif
#ifndef RT_LAZY_RENORM
#define RT_LAZY_RENORM 0
#endif

namespace rt {

constexpr int TILE = 8;            // 8x8 pixels per wavefront
// Tiles (wavefronts) per workgroup: a template parameter of the kernel, chosen per scene by the host.  Small
// workgroups start and retire at a finer grain (C2: 2 waves beat 4 by 3 %); every workgroup stages its own copy
// of the scene and its cull tables, so bigger scenes want bigger workgroups (C4: 4 waves beat 2 by 24 %, C5 by 69 %).
constexpr int TABLE_THREADS = 256; // tables_kernel's workgroup
constexpr int SPH_STRIDE = 8;      // doubles per sphere record: cx,cy,cz,r2, R,G,B, caller's index
constexpr int PL_STRIDE = 16;      // ox,oy,oz,nx,ny,nz, Nx,Ny,Nz, bNx,bNy,bNz, R,G,B, axis code (0 general, +-1/2/3 = +-e_x/y/z)
constexpr int LT_STRIDE = 4;       // x,y,z,pad
constexpr int CL_STRIDE = 4;       // cluster bounding sphere: cx,cy,cz,R2 (global memory only; LDS holds the float32 tables)
constexpr int CLUSTER = 8;         // spheres per cluster
#ifndef RT_CLUSTER_MIN
#define RT_CLUSTER_MIN 20   // measured (median-split clusters) against the flat scene: 16 spheres +3 % (the two-wave kernels, flat
                            // scenes only, are faster there), 25 -7 %, 36 -20 %, 49 -17 %, 64 -13 %
#endif
constexpr int CLUSTER_MIN = RT_CLUSTER_MIN;   // scenes with at most this many spheres stay flat
constexpr int SUPER = 8;           // clusters per group of clusters (one more box each: the lane-owned traversal skips whole groups)
constexpr int BOX_STRIDE = 8;      // floats per cluster box: lo.xyz, -, hi.xyz, - (two ds_read_b128)
constexpr int CULL_STRIDE = 4;     // floats per (anchor, sphere) cull entry: Lx,Ly,Lz, tau (one ds_read_b128)
#ifndef RT_MAX_CULL_TABLE_BYTES
#define RT_MAX_CULL_TABLE_BYTES (40 * 1024)
#endif
constexpr int MAX_CULL_TABLE_BYTES = RT_MAX_CULL_TABLE_BYTES;   // anchored cull table budget per workgroup (LDS)

struct KParams {
    const double *scene;       // packed records: S spheres, then P planes, then L lights
    const double *pixel_loc;   // explicit (3,w,h) grid or nullptr (closed-form ray generation)
    uint8_t *out_u8;           // or nullptr
    float *out_f32;            // or nullptr
    unsigned *tile_cycles;     // or nullptr: per-tile wave cycles of this launch (rt_set_tile_stats)
    unsigned *cost;            // or nullptr: scheduler feedback — a measuring launch stores every tile block's cost (wave cycles / 4,
                               // summed over the block's waves); order_kernel turns them into the next launches' dispatch order
    const unsigned *order;     // or nullptr: workgroup -> tile-block permutation from a measured launch's costs (XCD-affine, longest
                               // first inside every XCD); order + bpf: the same XCD assignment in plain tile order, which all but the
                               // last frame of a multi-frame launch use (order_kernel)
    int order_tiles;           // 1 (four-wave kernels): `order` and `cost` are per TILE — workgroup b's wave w renders tile
                               // order[b WPW + w], so that a workgroup's waves can be tiles of equal cost (they end together and
                               // hand their slots back together); 0: per tile block (WPW consecutive tiles)
    int seq_offset;            // bpf (x WPW with order_tiles), or 0 to dispatch every frame of a multi-frame launch longest-first (MI355RT_SEQ_ORDER=0)
    int nframes, bpf;          // frames rendered by this launch (rt_render_sequence) and workgroups per frame: workgroup b renders
                               // block order[b % bpf] of frame b / bpf into the outputs + (b / bpf) * frame_stride elements
    long long frame_stride;
    const float *ftab;         // the float32 cull tables of this scene / camera / depth, built once by tables_kernel
    unsigned long long *ray_counts;   // counting instantiation only (RT_FLAG_COUNT_RAYS): {closest, shadow issued, shadow skipped, hits}
    double *out_f64;           // lattice instantiation: float64 (R,G,B) per lattice sample, [column - x0][row][3]; aa_resolve_kernel reads it
    int lattice;               // 1: the "frame" is the (2w-1) x (2h-1) half-pixel lattice of the pixel grid (w, h, x0, x1 are lattice units)
    int lat_x0, lat_h;         // aa_resolve_kernel: first lattice column in out_f64, lattice rows
    long long plane_stride;    // elements between colour planes of the output
    int w, h, x0, x1;
    int S, P, L, depth;
    int NC;                    // sphere clusters (0 = flat)
    unsigned plane_codes;      // axis codes of planes 0..3, one signed byte each: 0 = general normal, +-(axis+1) = exactly
                               // axis-aligned unit normal.  A kernel argument lives in an SGPR, so the plane tests of
                               // the usual scenes (a floor, a wall) branch on the scalar unit without touching the VALU
    int aa, u8_rgb, tiles_y, ntiles;
    unsigned tiles_y_magic, tiles_y_shift, bpf_magic, bpf_shift;   // div_magic() of tiles_y and bpf (the host fills them)
    int anchors, spp;          // L+1 if the anchored cull table is in use, else 0; samples per pixel (stochastic AA)
    unsigned seed;             // jitter hash seed (stochastic AA)
    int u8_hwc;                // uint8 frame interleaved as [y][x][3] (an image), row pitch = plane_stride pixels
    int lanes_primary;         // MODE 2: lane-owned traversal for the primary rays and their shadow rays too (else from bounce 1 on)
    float extent2, floor_anch; // max squared distance of camera / lights / sphere surfaces from the world origin;
                               // launch-constant floor of the anchored cull (host: 2^-39 (|cam| + 999(depth+1) + extent)²)
    double px, y0, dy, z0, dz;
    double cam_o[3];
    double cam_R[9];
    double amb, lamb;
    double refl_pow[16];
};

// Division of n < 2^31 by a launch constant d without the backend's 20-instruction sequence (v_rcp_iflag_f32 and two
// correction steps, on the VECTOR unit even for wave-uniform operands): q = mulhi(n, M) >> sh with M = floor(2^(31+l) / d) + 1,
// l = ceil(log2 d), sh = l - 1 — exact because n d < 2^(31+l) (Granlund-Montgomery); d = 1 passes n through.  Every wave
// divides its tile index by the tiles per column, and in multi-frame launches its block index by the blocks per frame,
// twice: 31 of the headline kernel's vector instructions (and as many scalar ones) per wave, C2 -1.5 %, C4 -1.7 %.
// tests/test_host_helpers.py checks the formula exhaustively on small ranges and on random operands.
__host__ __device__ inline void div_magic(unsigned d, unsigned &M, unsigned &sh)
{
    if (d <= 1u) { M = 0u; sh = 0u; return; }
    unsigned l = 0;
    while ((1ull << l) < d) ++l;
    M = (unsigned)((1ull << (31 + l)) / d + 1ull);
    sh = l - 1u;
}
__device__ __forceinline__ int div_by(int n, int d, unsigned M, unsigned sh)
{
    return d == 1 ? n : (int)(__umulhi((unsigned)n, M) >> sh);
}

struct V3 { double x, y, z; };
struct F3 { float x, y, z; };

__device__ __forceinline__ double dot3(const V3 &a, const V3 &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }   // common.py:35-37

// common.py:28-32 — sqrt and three true divisions, as the compiler lowers them (correctly rounded).
__device__ __forceinline__ V3 normalize3_generic(const V3 &v)
{
    double n = __builtin_sqrt(v.x * v.x + v.y * v.y + v.z * v.z);
    return V3{v.x / n, v.y / n, v.z / n};
}

// The same normalize with the work the three divisions have in common done once.  The AMDGPU backend lowers
//   sqrt(x):  y = rsq(x); g = x*y; h = y/2; r = fma(-h,g,1/2); g = fma(g,r,g); h = fma(h,r,h);
//             twice { d = fma(-g,g,x); g = fma(d,h,g); }                      (+ range scaling of x)
//   a / b:    r = rcp(b); twice { e = fma(-b,r,1); r = fma(r,e,r); }  q = a*r; res = fma(fma(-b,q,a), r, q)
//                                                                             (+ div_scale / div_fixup)
// both correctly rounded.  For operands well inside the exponent range the scaling steps are identities, so
// the sequences below return bit-identical results while the reciprocal (5 of a division's 11 instructions) is
// shared by x/n, y/n, z/n and the sqrt skips its range handling.  The reciprocal itself needs no v_rcp_f64
// (a quarter-rate instruction): the sqrt iteration's h already approximates 1/(2g) to ~2^-50, so one Newton
// step from 2h lands where the backend's two steps from the rcp seed land — within half an ulp (+2^-47) of
// 1/g, which is what the final fma correction of each quotient requires.  26 instructions instead of 55.  Guard (wave-uniform): every component's magnitude >= 2^-200 (hence nonzero, and |v|² >= 2^-400) and
// |v|² <= 2^400, so no intermediate leaves the normal range and no signed-zero case arises; otherwise the
// generic path.  tests/test_algorithms.py replays this on the CPU against sqrt()/division on 10^8 vectors
// with seeds 16x less accurate than v_rsq_f64 / v_rcp_f64.
__device__ __forceinline__ V3 normalize3(const V3 &v)
{
#if RT_FAST_NORMALIZE == 0
    return normalize3_generic(v);
#else
    const double nn = v.x * v.x + v.y * v.y + v.z * v.z;
    const double cmin = __builtin_fmin(__builtin_fmin(__builtin_fabs(v.x), __builtin_fabs(v.y)), __builtin_fabs(v.z));
    const bool ok = (cmin >= 0x1p-200) && (nn <= 0x1p400);                  // NaNs fail both
    if (__builtin_amdgcn_ballot_w64(!ok) != 0ull) return normalize3_generic(v);
    const double y = __builtin_amdgcn_rsq(nn);
    double g = nn * y, h = 0.5 * y;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g); h = __builtin_fma(h, r, h);
    double d = __builtin_fma(-g, g, nn); g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, nn); g = __builtin_fma(d, h, g);               // g = RN(sqrt(nn))
    double rc = 2.0 * h;                                                    // h ~ 1/(2g), relative error ~2^-50
    const double e = __builtin_fma(-g, rc, 1.0); rc = __builtin_fma(rc, e, rc);   // within 1/2 ulp(1 + 2^-47) of 1/g
    const double qx = v.x * rc, qy = v.y * rc, qz = v.z * rc;
    return V3{__builtin_fma(__builtin_fma(-g, qx, v.x), rc, qx),
              __builtin_fma(__builtin_fma(-g, qy, v.y), rc, qy),
              __builtin_fma(__builtin_fma(-g, qz, v.z), rc, qz)};
#endif
}

// The scene queries' own division and square root (t = num/den of a plane, t = n/a of the closest sphere, sqrt of a sphere's
// discriminant), as the backend lowers them minus the range handling (see normalize3 above): v_div_scale x2 / v_div_fixup
// and the sqrt's ldexp scaling are identities for operands well inside the exponent range, and there they are: scene values
// are float32 (magnitudes 2^-149 .. 2^128 or zero), a plane's denominator is at least 0.001 in magnitude where the quotient is
// formed (intersections.py:55), a is within rounding of 1, numerators and discriminants are sums of a few products of such
// values (zero, or above 2^-600), while the backend scales below 2^-767 (sqrt) / numerators below 2^-969, denominators of
// extreme exponent, exponent differences beyond 2^768 (division).  Outside that range (a camera at 1e300, infinities) both
// forms end in an infinity, a NaN or a value whose comparison with (0, 999) has the same outcome: no hit.  3 of 11
// instructions per division, 7 of 21 per square root.  tests/algo/divsqrt_check.c replays both on the CPU.
__device__ __forceinline__ double div_inrange(double a, double b)
{
#if RT_FAST_DIVSQRT == 0
    return a / b;
#else
    double r = __builtin_amdgcn_rcp(b);
    double e = __builtin_fma(-b, r, 1.0); r = __builtin_fma(r, e, r);
    e = __builtin_fma(-b, r, 1.0); r = __builtin_fma(r, e, r);
    const double q = a * r;
    return __builtin_fma(__builtin_fma(-b, q, a), r, q);
#endif
}
__device__ __forceinline__ double sqrt_inrange(double x)      // x >= 0 (or NaN)
{
#if RT_FAST_DIVSQRT == 0
    return __builtin_sqrt(x);
#else
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g); h = __builtin_fma(h, r, h);
    double d = __builtin_fma(-g, g, x); g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, x); g = __builtin_fma(d, h, g);
    return x > 0.0 ? g : x;                                   // sqrt(+0) = +0 (the iteration yields NaN there: 0 * inf)
#endif
}

// ---------------------------------------------------------------------------------------------
// normalize() of a vector that is ALREADY unit length (every direction a scene query receives is
// the output of a normalize) — bit-identical to normalize3(), without the generic sqrt and divides.
//
//   nn = fl(x²+y²+z²) lies within a few ulp of 1; e = nn - 1 is exact.  With eps = 2^-52:
//     e = k·eps    (k >= 0):  sqrt = 1 + e/2 - e²/8 ..  ->  RN = 1 + floor(k/2)·eps
//     e = -j·eps/2 (j >= 1):  sqrt = 1 + e/2 - e²/8 ..  ->  RN = 1 - ceil(j/2)·eps/2
//   i.e. RN(sqrt(nn)) is 1 + e/2 rounded to nearest with every tie (odd k, odd j) resolved downward, because
//   the second-order term is negative.  One fma rounds exactly once, so nudging e toward -inf by a relative
//   2^-30 first (e2 = e - |e|·2^-30, far below half a grid step, far above e²/8) makes
//   nrm = fma(e2, 1/2, 1) that value: the nudge decides the ties and moves nothing else.
//   y = RN(1/nrm) likewise: with dl = nrm - 1 (exact), 1/nrm = 1 - dl + dl² .., ties (dl = -i·eps/2, i odd)
//   resolved upward: y = fma(-dl, 1 + 2^-30, 1).
//   Each quotient x/nrm is then  q1 = fma(-x, dl, x)  (faithful: off by |x|·dl² at most),
//   r = fma(-q1, nrm, x) (the exact remainder),  q = fma(r, y, q1)  — Markstein's final correction,
//   which returns the correctly rounded quotient RN(x/nrm) given y = RN(1/nrm) and a faithful q1.
// Falls back to the generic path (wave-uniformly) if any live lane's nn is not within 2^-33 of 1.
// tests/test_algorithms.py replays this routine on the CPU: nrm and y against sqrt and division for EVERY double
// within 2^-33 of 1 (1.57 million), the whole routine against sqrt-and-divide on 10^7 vectors.
// (The same values used to be derived from the IEEE bit pattern with 64-bit integer arithmetic: ~25 more
// instructions per call, 13 calls per tile — one in seven of all the kernel's VALU instructions.)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ V3 renormalize_unit(const V3 &d)
{
    const double nn = d.x * d.x + d.y * d.y + d.z * d.z;
    const double e = nn - 1.0;                                     // exact
    if (__builtin_amdgcn_ballot_w64(!(__builtin_fabs(e) < 0x1p-33)) != 0ull) return normalize3_generic(d);
    const double e2 = __builtin_fma(-__builtin_fabs(e), 0x1p-30, e);
    const double nrm = __builtin_fma(e2, 0.5, 1.0);                // RN(sqrt(nn))
    const double dl = nrm - 1.0;                                   // exact
    const double y = __builtin_fma(-dl, 1.0 + 0x1p-30, 1.0);       // RN(1/nrm)
    V3 q{__builtin_fma(-d.x, dl, d.x), __builtin_fma(-d.y, dl, d.y), __builtin_fma(-d.z, dl, d.z)};
    const V3 r{__builtin_fma(-q.x, nrm, d.x), __builtin_fma(-q.y, nrm, d.y), __builtin_fma(-q.z, nrm, d.z)};
    q = V3{__builtin_fma(r.x, y, q.x), __builtin_fma(r.y, y, q.y), __builtin_fma(r.z, y, q.z)};
    return q;
}

enum { HIT_NONE = 0, HIT_SPHERE = 1, HIT_PLANE = 2 };

// A per-lane V3 that lives either in LDS (PARK: slot `slot` of the per-thread area, [component][thread] so
// consecutive lanes hit consecutive banks) or in registers.  Parking long-lived values (the running colour,
// the incoming direction during the light loop, the AA tap sums) is what takes the kernel from 5 to 7
// waves/SIMD; the compiler would otherwise spill them to scratch, i.e. to HBM.  volatile + explicit LDS
// address space: a real ds_read/ds_write round trip through ONE 32-bit address (without volatile the stores
// are forwarded and the values stay in VGPRs; without the address space the accesses become flat).
// Scenes whose LDS image is too large for 6+ workgroups per CU run the register variant (host picks).
typedef __attribute__((address_space(3))) double lds_f64;
// REMAT (MODE 3 = MODE 2 with this; the AA kernels of the large clustered scenes, which run at 128 VGPRs with the most spills,
// take it: config 5 at 4 spp -1.5 %, while the 1 spp kernels lose 0.3 % with it): the slot's address is not kept in a
// register between accesses but re-derived at each one from the wave's index (a scalar) and the lane's number (two v_mbcnt,
// volatile so that the compiler does not hoist and keep them): three instructions per access, one VGPR less across the whole
// bounce.
__device__ __forceinline__ unsigned fresh_lane()
{
    unsigned l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}
template <bool PARK, int WGT, bool REMAT = false> struct Park3 {
    volatile lds_f64 *p;
    volatile lds_f64 *sbase;       // REMAT: the slot's first word of this wave (scalar)
    V3 v;
    __device__ __forceinline__ Park3(double *base, int slot, int wave = 0)
        : p(REMAT ? nullptr : (volatile lds_f64 *)base + slot * 3 * WGT + threadIdx.x),
          sbase(REMAT ? (volatile lds_f64 *)base + slot * 3 * WGT + wave * 64 : nullptr), v{0.0, 0.0, 0.0} {}
    __device__ __forceinline__ volatile lds_f64 *at() const { if constexpr (REMAT) return sbase + fresh_lane(); else return p; }
    __device__ __forceinline__ void set(const V3 &a)
    {
        if constexpr (PARK) { volatile lds_f64 *q = at(); q[0] = a.x; q[WGT] = a.y; q[2 * WGT] = a.z; } else v = a;
    }
    __device__ __forceinline__ V3 get() const
    {
        if constexpr (PARK) { volatile lds_f64 *q = at(); return V3{q[0], q[WGT], q[2 * WGT]}; } else return v;
    }
};

// ---------------------------------------------------------------------------------------------
// Conservative float32 cull.  For a ray (o, R) and sphere (c, r2) the reference computes, in
// float64, D = s² - a·(|L|² - r2) with L = o - c, s = L·R, and reports a miss when D < 0, or when
// s >= 0 and |L|² - r2 >= 0 ("behind", see below).  D is r2 minus the squared distance from c to
// the ray's LINE, so it may be evaluated from any point A of the line: D = (A-c)·R)² - |A-c|² + r2.
//
//  * anchored form — every primary ray passes through the camera and every shadow ray through its
//    light, so with A = camera/light the vector A-c, w = r2-|A-c|² and the error margin are
//    constants per (anchor, sphere): a table built once per workgroup in LDS (from float64), and
//    the per-ray work is s' = (A-c)·R32 (3 ops) and ONE compare, |s'| < tau, with
//    tau = sqrt(|A-c|² - r2 - margin - floor) rounded down (D' = s'² + r2 - |A-c|² < -margin - floor);
//  * origin form — for reflection rays, and for the sphere a shadow ray starts on, L = o32 - c is
//    formed per lane; it also certifies "behind" (s > e_s and |L|²-r2 > e_c).
//
// Error budget, u = 2^-24, Λ = |A-c| or |L|: inputs rounded to float32 (relative u each), three
// roundings in each dot product, one in the final fma:
//     anchored: |D' - D| <= u(13Λ² + 2 r2)               margin used: 32u(Λ² + r2)
//     origin:   |D' - D| <= u(19Λ² + 2|o|² + 2 r2)       margin used: 64u(Λ² + |o|² + r2)
//               |s' - s| <= u(|o| + 5Λ)                  margin used:  8u(1 + Λ² + |o|²)
//               |c' - c| <= u(7Λ² + |o|² + r2)           margin used: 64u(Λ² + |o|² + r2)
// plus, in all of them, floor = 2^-40(|o|² + extent²) >= the float64 rounding of the reference's own
// D (<= 2^-50 of its operands) and the distance by which a float64 direction misses its anchor.  The anchored
// form uses a launch constant for it: every ray origin lies within |cam| + 999(depth+1) of the world origin
// (each segment of a path is shorter than the far limit), so floor_anch = 2^-39(|cam| + 999(depth+1) + extent)²
// bounds it for every query of the launch.
// A sphere is skipped only if EVERY live lane holds a certificate; NaNs certify nothing.
// ---------------------------------------------------------------------------------------------
constexpr float CULL_K_ANCHOR = 0x1p-19f * 1.0001f;
constexpr float CULL_K_ORIGIN = 0x1p-18f;
constexpr float CULL_K_S = 0x1p-21f;
constexpr float CULL_K_FLOOR = 0x1p-40f;

// The workgroup's dynamic LDS.  Its base is a link-time constant: reading the float64 records through this symbol (and
// not through a pointer carried in a struct) lets every record access be a ds_read at an immediate offset of the slot
// index — as a carried pointer the base sat in a scalar register that was spilled and re-read (v_readlane + v_mov)
// 33 times in the bundle kernels.
extern __shared__ double lds_raw[];

struct Lds {
    __device__ __forceinline__ const double *recs() const { return lds_raw; }     // float64 records
    const float *sph32;    // Sp x {cx,cy,cz,r2}
    const float *tab;      // anchors x Sp x CULL_STRIDE
    const float *csph32;   // NCp x {cx,cy,cz,R2}: cluster bounding spheres, origin form
    const float *ctab;     // anchors x NCp x CULL_STRIDE: cluster bounding spheres, anchored form
    const float *cbox;     // NCp x BOX_STRIDE: cluster bounding boxes (rounded outward), for rays without an anchor
    const float *gbox;     // supers(NC) x BOX_STRIDE: boxes around groups of SUPER clusters
    const float *gtab;     // anchors x pad4(supers(NC)) x CULL_STRIDE: the groups' bounding spheres, anchored form
    const float *col32;    // (Sp + planes) x {R,G,B,-}: colours (MODE 1 kernels only; nullptr otherwise)
    int NC;
    int wave;              // this wave's index in its workgroup (scalar)
    bool pairs;            // a compile-time constant per kernel: the wave-uniform cull keeps two table entries in flight (cull4)
    bool groups;           // a compile-time constant per kernel: test a chunk's group of clusters before its clusters (MODE 2 kernels)
#ifdef RT_REGION_STATS
    unsigned *reg;         // measurement build: 32 words per wave — cycles per code region [0, 24), bounce class [30], last stamp [31]
#endif
    double *acc;           // 6 (9 with AA) x workgroup-size doubles, [slot][thread] (consecutive lanes -> consecutive banks):
                           // slots 0-2 the running colour of the current sample, 3-5 the incoming direction
                           // during the light loop, 6-8 (AA kernel only) the tap sums — kept out of VGPRs that would stay
                           // live across every query of every bounce (registers decide occupancy here)
};

#ifdef RT_REGION_STATS
// Measurement build only (hipcc -DRT_REGION_STATS, tools/region_stats.py): where a wave's cycles go.  RT_MARK(id) adds the
// shader-clock cycles since the wave's previous mark to region `id` (+ 12 for bounces 2 and later); all active lanes write
// the same values to the wave's words.  Regions: 0 re-normalise (closest), 1 closest: cluster bounds, 2 closest: cluster
// loop (float32 sphere tests + float64 tests), 3 closest: planes + select, 4 hit point / normal, 5 light direction + Lambert,
// 6 re-normalise (shadow), 7 shadow: cluster bounds, 8 shadow: cluster loop, 9 shadow: planes, 10 reflection, 11 the rest.
__device__ __forceinline__ void region_mark(unsigned *reg, int id)
{
    volatile unsigned *r = reg + (threadIdx.x >> 6) * 32;
    const unsigned now = (unsigned)__builtin_amdgcn_s_memtime();
    const unsigned last = r[31], c = r[30];
    r[id + c] = r[id + c] + (now - last);
    r[31] = now;
}
#define RT_MARK(id) region_mark(lds.reg, id)
#else
#define RT_MARK(id)
#endif

struct RayF {              // float32 shadow of a query, for the cull only
    F3 o, R;
    float mgq, esq;        // per-ray parts of the origin form's margins: K_O |o|² + floor,  K_S (1 + |o|²)
};

// The anchored form needs the direction only; the origin terms are added where the origin form is used.
__device__ __forceinline__ RayF make_rayf_dir(const V3 &R)
{
    RayF q;
    q.o = F3{0.0f, 0.0f, 0.0f};
    q.R = F3{(float)R.x, (float)R.y, (float)R.z};
    q.mgq = 0.0f; q.esq = 0.0f;
    return q;
}
__device__ __forceinline__ void add_origin(RayF &q, const V3 &o, float extent2)
{
    q.o = F3{(float)o.x, (float)o.y, (float)o.z};
    const float oo = __builtin_fmaf(q.o.z, q.o.z, __builtin_fmaf(q.o.y, q.o.y, q.o.x * q.o.x));
    q.mgq = __builtin_fmaf(CULL_K_ORIGIN, oo, CULL_K_FLOOR * (oo + extent2));
    q.esq = __builtin_fmaf(CULL_K_S, oo, CULL_K_S);
}

// tau of one (anchor, sphere) pair: the line through the anchor with unit direction R misses the sphere by more
// than every error the budget above allows when |(A-c).R32| < tau.  ll = |A-c|² (float64).
__device__ __forceinline__ float anchored_tau(double ll, double r2, float floor_anch)
{
    const double W = ((ll - r2) - (ll + r2) * (double)CULL_K_ANCHOR) - (double)floor_anch;
    if (!(W > 0.0)) return 0.0f;                              // anchor inside or too close (or NaN): never certified
    return (float)(__builtin_sqrt(W) * (1.0 - 0x1p-20));      // rounded down: the float32 rounding is within 2^-24
}

// Table entries are read through a 32-bit LDS address that is pinned in a VGPR (the empty asm): the index is
// wave-uniform, and left to itself the compiler forms every entry's address on the scalar unit and moves it
// into a VGPR for its ds_read — one extra VALU instruction per sphere.  With the base in a VGPR the entries of
// a group are immediate offsets of one 16-byte ds_read_b128 each.
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const f4 lds_cf4;
__device__ __forceinline__ lds_cf4 *pin_lds(const float *generic)
{
    unsigned a = (unsigned)(size_t)(__attribute__((address_space(3))) const float *)generic;
    asm volatile("" : "+v"(a));
    return (lds_cf4 *)(size_t)a;
}

// true = this lane holds a certificate that sphere k reports a miss for this ray (anchored form)
__device__ __forceinline__ bool cull_anchored(const f4 e, const RayF &q)
{
    const float s = __builtin_fmaf(e[2], q.R.z, __builtin_fmaf(e[1], q.R.y, e[0] * q.R.x));
    return __builtin_fabsf(s) < e[3];                         // e[3] = tau (0: never, +inf: padding, always)
}

// Phase 1 of a scene query: one bit per sphere of the chunk [k0, k0+n), set when SOME live lane holds
// no certificate.  Straight-line float32 work with no dependence between spheres; the mask is
// wave-uniform (it is built from ballots), so phase 2 — the float64 test — runs only for set bits.
// self: the sphere a shadow ray starts on (-1: none), certified by the origin form's "behind" test.
// origin form: line-miss or behind certificate
__device__ __forceinline__ bool cull_origin(const f4 c, const RayF &q)
{
    const float lx = q.o.x - c[0], ly = q.o.y - c[1], lz = q.o.z - c[2];
    const float s = __builtin_fmaf(lz, q.R.z, __builtin_fmaf(ly, q.R.y, lx * q.R.x));
    const float ll = __builtin_fmaf(lz, lz, __builtin_fmaf(ly, ly, lx * lx));
    const float cc = ll - c[3];
    const float D = __builtin_fmaf(s, s, -cc);
    const float mg = __builtin_fmaf(CULL_K_ORIGIN, ll + c[3], q.mgq);
    const float es = __builtin_fmaf(CULL_K_S, ll, q.esq);
    // no short-circuit: a divergent branch here costs an exec-mask region plus a select and a compare to get the
    // result back into a lane mask
    return (bool)((int)(D < -mg) | ((int)(s > es) & (int)(cc > mg)));
}

// Lane masks of the cull are produced by the compares themselves (inline asm, one VALU instruction each, result in
// an SGPR pair, zero for inactive lanes), combined on the scalar unit and pushed into the accumulator with
// s_cmp + s_addc:  acc = 2 acc + (some live lane holds no certificate).  Going through bool and a ballot instead
// costs a v_cndmask and a v_cmp per sphere whenever the predicate is more than a single compare, and
// compare + select + shift + or is four scalar instructions where two do.  Each CU has ONE scalar ALU for its four
// SIMDs and this kernel keeps it well over half busy, so scalar instructions are not free here.
// All of these are "not ..." compares: an unordered operand (NaN) yields 1 = no certificate.
typedef unsigned long long lanemask;
__device__ __forceinline__ lanemask m_nlt_abs(float a, float b)   // !(|a| < b)
{
    lanemask m; asm volatile("v_cmp_nlt_f32_e64 %0, |%1|, %2" : "=s"(m) : "v"(a), "v"(b)); return m;
}
__device__ __forceinline__ lanemask m_nlt_neg(float a, float b)   // !(a < -b)
{
    lanemask m; asm volatile("v_cmp_nlt_f32_e64 %0, %1, -%2" : "=s"(m) : "v"(a), "v"(b)); return m;
}
__device__ __forceinline__ lanemask m_ngt(float a, float b)       // !(a > b)
{
    lanemask m; asm volatile("v_cmp_ngt_f32_e64 %0, %1, %2" : "=s"(m) : "v"(a), "v"(b)); return m;
}
__device__ __forceinline__ lanemask m_ne(int a, int b)            // a != b
{
    lanemask m; asm volatile("v_cmp_ne_u32_e64 %0, %1, %2" : "=s"(m) : "v"(a), "s"(b)); return m;
}
__device__ __forceinline__ unsigned push_any(unsigned acc, lanemask m)
{
    asm volatile("s_cmp_lg_u64 %1, 0\n\ts_addc_u32 %0, %0, %0" : "+s"(acc) : "s"(m) : "scc");
    return acc;
}
__host__ __device__ inline int pad4(int n) { return (n + 3) & ~3; }
__host__ __device__ inline int supers(int NC) { return (NC + SUPER - 1) / SUPER; }   // groups of SUPER clusters

// logarithmic cost key for the dispatch-order feedback: monotone in c, < 1024
__device__ __forceinline__ int order_bucket(unsigned c)
{
    if (c < 32u) return (int)c;
    const int msb = 31 - __builtin_clz(c);
    return ((msb - 4) << 5) | (int)((c >> (msb - 5)) & 31u);
}
// dispatch-order feedback (order_kernel)
constexpr int ORDER_THREADS = 1024, ORDER_BUCKETS = 1024, ORDER_XCDS = 8;
// sphere slots in the float32 tables: whole clusters when the scene is clustered, else a multiple of 4
__host__ __device__ inline int padS(int S, int NC) { return NC > 0 ? NC * CLUSTER : pad4(S); }


// ---------------------------------------------------------------------------------------------
// Slab test of a ray without an anchor (o, R) against a cluster's bounding box, float32, conservative: a certificate
// that the reference reports a miss for EVERY sphere in the box.
//   exact geometry: if the half-line o + tR, t >= 0, stays at least d outside the box, it stays d outside every sphere
//   in it, so the reference's D = r2 - (distance to the line)^2 < -d^2 (or both roots lie behind the origin by d): its own
//   float64 rounding (<= 2^-40 (|o|^2 + extent^2), the `floor` of the sphere certificates) cannot turn that into a hit
//   once d >= 2^-20 sqrt(|o|^2 + extent^2);
//   float32 ray: o and R rounded to float32 move a point of the ray at parameter t by at most 2^-24 (|o| + t), and every t
//   that matters is below |o| + extent;
//   both are covered by growing the box by m = 2^-18 (extent + |o|_1) on every side — done on the ray instead: the lower
//   faces are tested from o + m, the upper ones from o - m;
//   float32 arithmetic: a face's parameter is t = fma(face, inv, c), inv = rcp(R_i) (1 ulp), c = -(o_i +- m) inv rounded:
//   t carries a relative error below 2^-21 and an absolute one below 2^-23 |c|; c is moved by 2^-21 |c| in the direction
//   that lowers the near parameter and raises the far one, and the final comparison allows 2^-18 relative.
//   R_i = 0 is replaced by +-2^-40 (a change of direction far below the float32 rounding already covered).
// Miss certificate: tfar < max(tnear (1 - 2^-18), 0).  NaN operands never certify (the caller opens everything).
// 17 VALU instructions per box, as many as the bounding-SPHERE test it replaces, for a bound several times tighter around
// flat or elongated clusters.
// ---------------------------------------------------------------------------------------------
struct RayBox { F3 inv, clo, chi; bool sane; };
__device__ __forceinline__ float vmin3(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float vmax3(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float vmin(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vmax(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ void raybox_axis(float o, float R, float m, float &inv, float &clo, float &chi)
{
    const float Rs = __builtin_fabsf(R) < 0x1p-40f ? __builtin_copysignf(0x1p-40f, R) : R;
    inv = __builtin_amdgcn_rcpf(Rs);
    const float a = -((o + m) * inv), b = -((o - m) * inv);     // lower faces from o + m, upper faces from o - m
    const float da = __builtin_fabsf(a) * 0x1p-21f, db = __builtin_fabsf(b) * 0x1p-21f;
    // inv > 0: the lower face is the near one (its parameter goes down, the upper face's up); inv < 0: the other way round
    clo = inv > 0.0f ? a - da : a + da;
    chi = inv > 0.0f ? b + db : b - db;
}
__device__ __forceinline__ RayBox make_raybox(const RayF &q, float extent2)   // q with its origin (add_origin)
{
    RayBox rb;
    const float m = 0x1p-18f * (__builtin_sqrtf(extent2) + (__builtin_fabsf(q.o.x) + __builtin_fabsf(q.o.y) + __builtin_fabsf(q.o.z)));
    raybox_axis(q.o.x, q.R.x, m, rb.inv.x, rb.clo.x, rb.chi.x);
    raybox_axis(q.o.y, q.R.y, m, rb.inv.y, rb.clo.y, rb.chi.y);
    raybox_axis(q.o.z, q.R.z, m, rb.inv.z, rb.clo.z, rb.chi.z);
    const float chk = (q.o.x + q.o.y + q.o.z) + (q.R.x + q.R.y + q.R.z);
    rb.sane = (chk == chk) && __builtin_fabsf(chk) < 0x1p100f;                // no NaN, no infinity among the six
    return rb;
}
__device__ __forceinline__ bool box_open(const f4 lo, const f4 hi, const RayBox &rb)   // true = NO certificate
{
    const float ax = __builtin_fmaf(lo[0], rb.inv.x, rb.clo.x), bx = __builtin_fmaf(hi[0], rb.inv.x, rb.chi.x);
    const float ay = __builtin_fmaf(lo[1], rb.inv.y, rb.clo.y), by = __builtin_fmaf(hi[1], rb.inv.y, rb.chi.y);
    const float az = __builtin_fmaf(lo[2], rb.inv.z, rb.clo.z), bz = __builtin_fmaf(hi[2], rb.inv.z, rb.chi.z);
    const float tn = vmax3(vmin(ax, bx), vmin(ay, by), vmin(az, bz));
    const float tf = vmin3(vmax(ax, bx), vmax(ay, by), vmax(az, bz));
    return !(tf < vmax(tn * (1.0f - 0x1p-18f), 0.0f));
}

// Certificates of 4 consecutive table entries -> 4 mask bits (bit u set = some live lane has no certificate).
// The tables are padded with entries that always certify a miss (tau = +inf, r2 = -inf), so groups of 4 need no
// bounds handling and use immediate LDS offsets.
// Returns 16 acc + bits (entries are visited from the highest to the lowest).
// PAIRS: two entries are read before either is tested — one LDS round trip per two tests instead of one per test, for four more
// live VGPRs.  The wave-uniform kernels take it (config 4 -2.7 %, 36 ... 144 spheres -3 ... -5 %, 16 spheres -3 %, the headline -0.6 %,
// the 9-tap kernels -0.7 ... -2.6 %); the lane-owned kernels, which run
// this cull for their primary rays at 128 VGPRs with spills, lose 0.8 % with it and keep one entry in flight.
template <bool ANCH, bool SELF, bool PAIRS>
__device__ __forceinline__ unsigned cull4(lds_cf4 *base, const RayF &q, int jsel, unsigned acc)
{
    static_assert(CULL_STRIDE == 4, "one 16-byte entry per (anchor, sphere)");
    f4 pending = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int u = 3; u >= 0; --u) {
        f4 e;
        if constexpr (PAIRS) {
            if (u & 1) {
                asm volatile("" ::: "memory");
                e = base[u];
                f4 enext = base[u - 1];
                asm volatile("" : "+v"(e), "+v"(enext));      // both loaded before either is used
                pending = enext;
            } else e = pending;
        } else {
            asm volatile("" ::: "memory");                    // one entry in flight (registers decide occupancy)
            e = base[u];
        }
        lanemask open;                                        // live lanes without a certificate for this entry
        if constexpr (ANCH) {
            const float sd = __builtin_fmaf(e[2], q.R.z, __builtin_fmaf(e[1], q.R.y, e[0] * q.R.x));
            open = m_nlt_abs(sd, e[3]);                       // cull_anchored()
        } else {                                              // cull_origin()
            const float lx = q.o.x - e[0], ly = q.o.y - e[1], lz = q.o.z - e[2];
            const float sd = __builtin_fmaf(lz, q.R.z, __builtin_fmaf(ly, q.R.y, lx * q.R.x));
            const float ll = __builtin_fmaf(lz, lz, __builtin_fmaf(ly, ly, lx * lx));
            const float cc = ll - e[3];
            const float D = __builtin_fmaf(sd, sd, -cc);
            const float mg = __builtin_fmaf(CULL_K_ORIGIN, ll + e[3], q.mgq);
            const float es = __builtin_fmaf(CULL_K_S, ll, q.esq);
            open = m_nlt_neg(D, mg) & (m_ngt(sd, es) | m_ngt(cc, mg));
        }
        if constexpr (SELF) open &= m_ne(jsel, u);            // the sphere this lane's shadow ray starts on
        acc = push_any(acc, open);
    }
    return acc;
}

// Phase 1 of a scene query for the chunk of spheres [k0, k0+n), k0 a multiple of 64: one bit per sphere, set
// when SOME live lane holds no certificate.  Clustered scenes first test the (up to 8) bounding spheres of the
// chunk's clusters and only open the clusters some lane might hit.  Everything here is wave-uniform control
// flow on scalar masks.
template <bool ANCH, bool SELF>
__device__ __forceinline__ unsigned long long cull_mask_t(const Lds &lds, int S, int anchor, int k0, int n,
                                                          const RayF &q, int selfj)
{
    unsigned long long mask = 0ull;
    const int Sp = padS(S, lds.NC);
    lds_cf4 *sbase = pin_lds(ANCH ? lds.tab + ((size_t)anchor * Sp + k0) * CULL_STRIDE : lds.sph32 + 4 * k0);
    if (lds.NC > 0) {
        const int NCp = pad4(lds.NC), c0 = k0 / CLUSTER, nc = (n + CLUSTER - 1) / CLUSTER;
        if (ANCH && lds.groups) {                                             // the chunk's 8 clusters are one group: is any lane's ray near it at all?
                                                                              // (the kernels of the large scenes only: 196 spheres -3 %, 256 -4 %; 36-100 spheres +0.4..+1.6 %)
            static_assert(CLUSTER * SUPER == 64, "a 64-sphere chunk is one group of clusters");
            const f4 ge = *pin_lds(lds.gtab + ((size_t)anchor * pad4(supers(lds.NC)) + (k0 >> 6)) * CULL_STRIDE);
            const float gs = __builtin_fmaf(ge[2], q.R.z, __builtin_fmaf(ge[1], q.R.y, ge[0] * q.R.x));
            if (m_nlt_abs(gs, ge[3]) == 0ull) return 0ull;
        }
        // (the clusters' BOXES, which the lane-owned traversal tests for rays without an anchor, were tried here too: the
        // extra live values cost the wave-uniform kernels more in spills than the tighter bound saves: 36-100 spheres
        // +9..+20 %)
        lds_cf4 *cbase = pin_lds(ANCH ? lds.ctab + ((size_t)anchor * NCp + c0) * CULL_STRIDE : lds.csph32 + 4 * c0);
        unsigned cm = 0;
        for (int j = 0; j < nc; j += 4) cm |= (lds.pairs ? cull4<ANCH, false, true>(cbase + j, q, -1, 0u) : cull4<ANCH, false, false>(cbase + j, q, -1, 0u)) << j;
        while (cm) {                                                          // clusters some lane might hit
            const int c = __builtin_ctz(cm);
            cm &= cm - 1u;
            const int jb = c * CLUSTER;
            const unsigned hi = lds.pairs ? cull4<ANCH, SELF, true>(sbase + jb + 4, q, selfj - jb - 4, 0u) : cull4<ANCH, SELF, false>(sbase + jb + 4, q, selfj - jb - 4, 0u);
            const unsigned lohi = lds.pairs ? cull4<ANCH, SELF, true>(sbase + jb, q, selfj - jb, hi) : cull4<ANCH, SELF, false>(sbase + jb, q, selfj - jb, hi);
            mask |= (unsigned long long)lohi << jb;
        }
    } else {
        const int npad = pad4(n);
        for (int j = 0; j < npad; j += 4) mask |= (unsigned long long)(lds.pairs ? cull4<ANCH, SELF, true>(sbase + j, q, selfj - j, 0u) : cull4<ANCH, SELF, false>(sbase + j, q, selfj - j, 0u)) << j;
    }
    return mask;
}

__device__ __forceinline__ unsigned long long cull_mask(const Lds &lds, int S, int anchor, int k0, int n,
                                                        const V3 &o, const V3 &R, float extent2, int self)
{
    RayF q = make_rayf_dir(R);
    if (anchor >= 0) {
        const bool cand = self >= k0 && self < k0 + n;                        // per lane
        if (__builtin_amdgcn_ballot_w64(cand) != 0ull) {                                         // only shadow rays leaving a sphere
            add_origin(q, o, extent2);
            const float *cs = lds.sph32 + 4 * self;
            const bool self_culled = cand ? cull_origin(f4{cs[0], cs[1], cs[2], cs[3]}, q) : false;
            if (__builtin_amdgcn_ballot_w64(self_culled) != 0ull)
                return cull_mask_t<true, true>(lds, S, anchor, k0, n, q, self_culled ? self - k0 : -1);
        }
        return cull_mask_t<true, false>(lds, S, anchor, k0, n, q, -1);
    }
    add_origin(q, o, extent2);
    return cull_mask_t<false, false>(lds, S, anchor, k0, n, q, -1);
}

// ---------------------------------------------------------------------------------------------
// Sphere test, intersections.py:6-38, restated on the quadratic divided by 4.
//   reference:  b = 2s, disc = b*b - (4a)*c, num = -b -/+ sqrt(disc), t = num / (2a)
//   here:       D = s*s - a*c,  q = sqrt(D),  n = -s -/+ q,           t = n / a
// Scaling by powers of two commutes with every rounding involved (no overflow/underflow at
// scene magnitudes): disc = 4D, sqrt(disc) = 2q, num = 2n, t identical bit for bit.
// "Behind" rule: if s >= 0 and c >= 0 then D <= fl(s*s), so q <= s and both numerators are <= 0:
// the reference returns a miss, and so do we without evaluating the sqrt.
// ---------------------------------------------------------------------------------------------

// intersections.py:52 and :59-61 for one plane record: den = d.n and num = (p0 - o).n.  A plane whose stored
// normal is exactly +-e_i (the reference's ground plane is (0,0,1)) has den = +-d_i and num = +-(p0_i - o_i)
// bit for bit — the other two products are +-0 and add nothing — so 8 of the 13 operations are skipped under a
// wave-uniform branch on the record's axis code (set by the host).
// axis code of plane k from the kernel argument (planes 4.. use the general formula, which is exact for them too)
// A wave-uniform kernel argument made opaque at its point of use: the value stays where it is (a scalar register), but
// what is derived from it (a comparison, a select) is re-derived there with scalar instructions instead of being hoisted
// out of the bounce loop into scalar registers the kernel does not have — hoisted values come back as v_readlane, a VALU
// slot each, at every use (RT_OPAQUE_ARGS = 0 leaves it to the compiler).
__device__ __forceinline__ int opaque(int x)
{
#if RT_OPAQUE_ARGS
    asm volatile("" : "+s"(x));
#endif
    return x;
}

__device__ __forceinline__ int plane_code(const KParams &p, int k)
{
#if RT_OPAQUE_PLANE_CODE
    // the decoded code and the booleans derived from it are loop-invariant, and the compiler hoists them out of the bounce
    // loop — into scalar registers it does not have: they come back as v_readlane (a VALU slot each) at every use.
    // Re-deriving them from the one kernel-argument word costs scalar instructions only.
    unsigned c = p.plane_codes;
    asm volatile("" : "+s"(c));
    return k < 4 ? (int)(signed char)(c >> (8 * k)) : 0;
#else
    return k < 4 ? (int)(signed char)(p.plane_codes >> (8 * k)) : 0;
#endif
}

__device__ __forceinline__ void plane_den_num(const double *__restrict__ g, int code, const V3 &o, const V3 &d, double &den, double &num)
{
    if (code == 0) {
        const V3 n{g[3], g[4], g[5]};
        den = dot3(d, n);
        const V3 LP{g[0] - o.x, g[1] - o.y, g[2] - o.z};
        num = dot3(LP, n);
    } else {
        const int a = code < 0 ? -code : code;
        double dc, lp;
        if (a == 1)      { dc = d.x; lp = g[0] - o.x; }
        else if (a == 2) { dc = d.y; lp = g[1] - o.y; }
        else             { dc = d.z; lp = g[2] - o.z; }
        if (code > 0) { den = dc; num = lp; } else { den = -dc; num = -lp; }
    }
}

// What a sphere test reads of sphere slot k: centre and r*r.  F32 (the kernels of the large clustered scenes, MODE 2): from
// the float32 table the cull already keeps in LDS, widened on use — the scene IS float32 (scene.py:18) and r*r is a float32
// product (intersections.py:21), so the widening is exact; those kernels stage no float64 sphere records at all (config 5:
// 16 KB of LDS, which is what lets their long-lived state be parked in LDS instead of spilling to scratch), and read a hit
// sphere's colour and caller's index from the packed scene in global memory, at the point of use.
struct SphHot { double x, y, z, r2; };
template <bool F32> __device__ __forceinline__ SphHot sphere_hot(const Lds &lds, int k)
{
    if constexpr (F32) {
        const f4 c = *(lds_cf4 *)(size_t)(unsigned)(size_t)(__attribute__((address_space(3))) const float *)(lds.sph32 + 4 * k);
        return SphHot{(double)c[0], (double)c[1], (double)c[2], (double)c[3]};
    } else {
        const double *g = lds.recs() + k * SPH_STRIDE;
        return SphHot{g[0], g[1], g[2], g[3]};
    }
}
template <bool F32> __device__ __forceinline__ double sphere_orig(const Lds &lds, const KParams &p, int k)   // the caller's index of slot k
{
    if constexpr (F32) return p.scene[(size_t)k * SPH_STRIDE + 7]; else return lds.recs()[k * SPH_STRIDE + 7];
}

// intersections.py:6-38 for the sphere in slot k (float64), on the quadratic divided by 4 (see above), feeding the
// closest-hit selection: bestn = numerator of the current winner, bidx its slot.
template <bool F32>
__device__ __forceinline__ void sphere_closest(const Lds &lds, const KParams &p, int k, const V3 &o, const V3 &R, double a,
                                               double &bestn, int &bidx)
{
    const SphHot g = sphere_hot<F32>(lds, k);
    const V3 Lv{o.x - g.x, o.y - g.y, o.z - g.z};            // :16
    const double s = dot3(Lv, R);                             // b/2
    const double cc = dot3(Lv, Lv) - g.r2;                    // :21 (r2 = float32 r*r, widened)
    const double D = s * s - a * cc;                          // disc/4
    if (D >= 0.0 && !(s >= 0.0 && cc >= 0.0)) {
        const double q = sqrt_inrange(D);
        double n = -s - q;                                    // :28
        if (!(n > 0.0)) n = -s + q;                           // :33
        // The reference compares the rounded quotients t = n/a with a strict `best > t`, in ascending caller
        // index (trace.py:26): the smallest t wins and, among equal t, the lowest index.  Division by the
        // common a > 0 and rounding are monotone, so numerators order the quotients — except that two
        // numerators within a couple of ulp of each other may round to the SAME quotient, where the index
        // decides.  Numerators further apart than a relative 2^-50 have different quotients in the same
        // order (the gap is four ulp); for closer ones (equal included) both quotients are formed and the
        // reference's rule is applied literally.  Wave-uniform branch, practically never taken.
        // Slots are visited in any order (clustered scenes permute them): the records keep the caller's index, which is
        // looked up only here, for the two spheres of a tie.
        if (n > 0.0) {
            bool take = n < bestn;
            const bool close = __builtin_fabs(n - bestn) < bestn * 0x1p-50;   // false while bestn = +inf (bidx = -1)
            if (__builtin_amdgcn_ballot_w64(close) != 0ull) {
                if (close) {
                    const double tq = n / a, tb = bestn / a;                  // :31 / :36
                    take = tq < tb || (tq == tb && sphere_orig<F32>(lds, p, k) < sphere_orig<F32>(lds, p, bidx));
                }
            }
            if (take) { bestn = n; bidx = k; }
        }
    }
}

// the same sphere for a shadow (any-hit) query: does it report 0 < t < 999?
template <bool F32>
__device__ __forceinline__ bool sphere_any(const Lds &lds, int k, const V3 &o, const V3 &R, double a, bool a_sane)
{
    const SphHot g = sphere_hot<F32>(lds, k);
    const V3 Lv{o.x - g.x, o.y - g.y, o.z - g.z};
    const double s = dot3(Lv, R);
    const double cc = dot3(Lv, Lv) - g.r2;
    const double D = s * s - a * cc;
    if (D >= 0.0 && !(s >= 0.0 && cc >= 0.0)) {
        // s < 0: the larger numerator n2 = -s + q is positive, so a positive root exists and
        // the reference's t is at most n2/a.  n2 <= 998 (from -s < 499, q <= 499) and a within
        // 1e-6 of 1 give t < 999: occluded, decided without sqrt or divide.
        if (s < 0.0 && -s < 499.0 && D < 249001.0 && a_sane) return true;
        const double q = __builtin_sqrt(D);                   // origin inside the sphere, or a far hit: exact path (rare: the generic forms — with the in-range ones here the headline kernel's register allocation tips, +2.4 %)
        double n = -s - q;
        if (!(n > 0.0)) n = -s + q;
        if (n > 0.0) {
            const double t = n / a;
            if (999.0 > t && t > 0.0) return true;
        }
    }
    return false;
}


// ---------------------------------------------------------------------------------------------
// Lane-owned traversal of clustered scenes (more than CLUSTER_MIN spheres; the LANES instantiations).
// The wave-uniform cull above opens a cluster for the WHOLE wave as soon as one lane's ray might reach it, and gives
// every sphere of an opened cluster its per-ray float32 test and — if one lane lacks a certificate — its float64
// test for all lanes.  That is the right trade while the 64 rays travel together (primary rays, first shadow rays).
// After a few bounces they do not: every lane's ray opens its own 2-4 clusters, the union over the wave approaches
// ALL of them, and a wave with 20 rays left pays for hundreds of sphere tests none of its rays needs (measured on
// config 5: 3x the instructions per wave at bounce 8 compared with bounce 2).
// Here each lane keeps its own candidates: phase 1 tests the cluster bounds in a wave-uniform loop as before but
// leaves one bit per cluster IN THE LANE; then the wave loops while any lane has a cluster left, every lane taking
// ITS next cluster (per-lane LDS addresses: each lane reads the table entries and records of its own spheres),
// running the 8 float32 tests into a per-lane survivor mask, and looping again over its own survivors for the
// float64 test.  Iterations = the MAXIMUM over the lanes of their own counts, not the size of the union.
// Same certificates, same float64 arithmetic, same tie rule (explicit caller's index), any visiting order.
// ---------------------------------------------------------------------------------------------
template <bool ANCH>
__device__ __forceinline__ bool lane_open(const f4 e, const RayF &q)          // true = this lane holds NO certificate
{
    if constexpr (ANCH) {
        const float sd = __builtin_fmaf(e[2], q.R.z, __builtin_fmaf(e[1], q.R.y, e[0] * q.R.x));
        return !(__builtin_fabsf(sd) < e[3]);                                 // cull_anchored()
    } else return !cull_origin(e, q);
}
__device__ __forceinline__ lds_cf4 *lds_f4(const float *generic)             // per-lane LDS address of a table entry
{
    return (lds_cf4 *)(size_t)(unsigned)(size_t)(__attribute__((address_space(3))) const float *)generic;
}

// One bit per cluster bound of the block [cb, cb + nc), nc <= 32: set where THIS lane's ray lacks a certificate.
// Anchored rays: the bounding spheres' table of the anchor (a dot product and a compare each); the others: the boxes.
template <bool ANCH>
__device__ __forceinline__ unsigned lane_cluster_bits(const Lds &lds, int anchor, int cb, int nc, const RayF &q, float extent2)
{
    unsigned cm = 0u;
    if constexpr (ANCH) {
        const int NCp = pad4(lds.NC), NGp = pad4(supers(lds.NC));
        lds_cf4 *base = pin_lds(lds.ctab + ((size_t)anchor * NCp + cb) * CULL_STRIDE);
        lds_cf4 *gbase = pin_lds(lds.gtab + ((size_t)anchor * NGp + cb / SUPER) * CULL_STRIDE);
        for (int c0 = 0; c0 < nc; c0 += SUPER) {                             // a group no lane's ray comes near is skipped whole
            if (__builtin_amdgcn_ballot_w64(lane_open<true>(gbase[c0 / SUPER], q)) == 0ull) continue;
            const int c1 = c0 + SUPER < nc ? c0 + SUPER : nc;
            for (int c = c0; c < c1; c += 4) {                                // tables are padded to a multiple of 4
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    asm volatile("" ::: "memory");
                    const f4 e = base[c + u];
                    cm |= lane_open<true>(e, q) ? (1u << (c + u)) : 0u;
                }
            }
        }
    } else {
        const RayBox rb = make_raybox(q, extent2);
        lds_cf4 *base = pin_lds(lds.cbox + (size_t)cb * BOX_STRIDE);
        lds_cf4 *gbase = pin_lds(lds.gbox + (size_t)(cb / SUPER) * BOX_STRIDE);
        static_assert(SUPER == 8, "a group's clusters are one byte of the mask");
        for (int c0 = 0; c0 < nc; c0 += SUPER) {                             // a group no lane's ray enters is skipped whole
            const bool gopen = box_open(gbase[2 * (c0 / SUPER)], gbase[2 * (c0 / SUPER) + 1], rb) || !rb.sane;
            if (__builtin_amdgcn_ballot_w64(gopen) == 0ull) continue;
            const int c1 = c0 + SUPER < nc ? c0 + SUPER : nc;
            for (int c = c0; c < c1; c += 2) {                                // (an even number of boxes is stored)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    asm volatile("" ::: "memory");
                    const f4 lo = base[2 * (c + u)], hi = base[2 * (c + u) + 1];
                    cm |= box_open(lo, hi, rb) ? (1u << (c + u)) : 0u;
                }
            }
        }
        if (!rb.sane) cm = ~0u;
    }
    return cm & (nc >= 32 ? ~0u : ((1u << nc) - 1u));                         // (padding, and what a NaN ray opened of it)
}

// the survivors among the (up to 8) spheres of the lane's own cluster, first slot kb
template <bool ANCH>
__device__ __forceinline__ unsigned lane_sphere_bits(const Lds &lds, int S, int anchor, int kb, const RayF &q)
{
    const int Sp = padS(S, lds.NC);
    lds_cf4 *sb = lds_f4(ANCH ? lds.tab + ((size_t)anchor * Sp + kb) * CULL_STRIDE : lds.sph32 + 4 * kb);
    unsigned sm = 0u;
#pragma unroll
    for (int j = 0; j < CLUSTER; ++j) sm |= lane_open<ANCH>(sb[j], q) ? (1u << j) : 0u;
    const int nv = S - kb;                                                    // slots past the last sphere are padding
    return nv >= CLUSTER ? sm : (sm & ((1u << nv) - 1u));
}

#ifdef RT_LANE_STATS
// measurement build only: counters behind the per-tile cycles (p.tile_cycles[ntiles + base ..]): calls, sum of the
// per-wave maximum, sum of ceil(total / 64), sum of the totals
__device__ __forceinline__ void lane_stats(const KParams &p, int base, unsigned v)
{
    if (!p.tile_cycles) return;
    unsigned mx = v, sum = v;
    for (int d = 32; d >= 1; d >>= 1) { const unsigned m2 = __shfl_xor(mx, d), s2 = __shfl_xor(sum, d); mx = m2 > mx ? m2 : mx; sum += s2; }
    if ((threadIdx.x & 63) == __builtin_ctzll(__builtin_amdgcn_ballot_w64(true))) {
        unsigned *c = p.tile_cycles + p.ntiles + base;
        atomicAdd(c + 0, 1u); atomicAdd(c + 1, mx); atomicAdd(c + 2, (sum + 63u) >> 6); atomicAdd(c + 3, sum);
    }
}
#endif
template <bool ANCH>
__device__ __forceinline__ void lanes_closest(const Lds &lds, const KParams &p, int anchor, const V3 &o, const V3 &R, double a,
                                              double &bestn, int &bidx)
{
    RayF q = make_rayf_dir(R);
    if constexpr (!ANCH) add_origin(q, o, p.extent2);
    for (int cb = 0; cb < lds.NC; cb += 32) {
        const int nc = lds.NC - cb < 32 ? lds.NC - cb : 32;
        unsigned cm = lane_cluster_bits<ANCH>(lds, anchor, cb, nc, q, p.extent2);
        RT_MARK(1);
#ifdef RT_LANE_STATS
        lane_stats(p, 0, __builtin_popcount(cm));
#endif
        while (__builtin_amdgcn_ballot_w64(cm != 0u) != 0ull) {
#ifdef RT_LANE_STATS
            unsigned smc = 0;
#endif
            if (cm != 0u) {
                const int kb = (cb + __builtin_ctz(cm)) * CLUSTER;
                cm &= cm - 1u;
                unsigned sm = lane_sphere_bits<ANCH>(lds, p.S, anchor, kb, q);
#ifdef RT_LANE_STATS
                smc = __builtin_popcount(sm);
#endif
                while (__builtin_amdgcn_ballot_w64(sm != 0u) != 0ull) {
                    if (sm != 0u) {
                        const int k = kb + __builtin_ctz(sm);
                        sm &= sm - 1u;
                        sphere_closest<true>(lds, p, k, o, R, a, bestn, bidx);
                    }
                }
            }
#ifdef RT_LANE_STATS
            lane_stats(p, 4, smc);
#endif
        }
        RT_MARK(2);
    }
}

// any-hit: self = the sphere slot this lane's shadow ray starts on (-1: none); its own miss is certified by the
// origin form's "behind" test where that holds (as in cull_mask)
template <bool ANCH>
__device__ __forceinline__ bool lanes_any(const Lds &lds, const KParams &p, int anchor, const V3 &o, const V3 &R, double a, int self)
{
    const bool a_sane = (a > 0.999999 && a < 1.000001);
    RayF q = make_rayf_dir(R);
    bool self_culled = false;
    if constexpr (ANCH) {
        if (__builtin_amdgcn_ballot_w64(self >= 0) != 0ull) {
            add_origin(q, o, p.extent2);
            if (self >= 0) { const float *cs = lds.sph32 + 4 * self; self_culled = cull_origin(f4{cs[0], cs[1], cs[2], cs[3]}, q); }
        }
    } else add_origin(q, o, p.extent2);
    bool occ = false;
    for (int cb = 0; cb < lds.NC; cb += 32) {
        if ((RT_SHADOW_EXIT_CHECKS || cb > 0) && __builtin_amdgcn_ballot_w64(!occ) == 0ull) break;   // (nothing is occluded before the first block)
        const int nc = lds.NC - cb < 32 ? lds.NC - cb : 32;
        unsigned cm = lane_cluster_bits<ANCH>(lds, anchor, cb, nc, q, p.extent2);
        RT_MARK(7);
        if (occ) cm = 0u;
#ifdef RT_LANE_STATS
        lane_stats(p, 8, __builtin_popcount(cm));
#endif
        while (__builtin_amdgcn_ballot_w64(cm != 0u) != 0ull) {
#ifdef RT_LANE_STATS
            unsigned smc = 0;
#endif
            if (cm != 0u) {
                const int kb = (cb + __builtin_ctz(cm)) * CLUSTER;
                cm &= cm - 1u;
                unsigned sm = lane_sphere_bits<ANCH>(lds, p.S, anchor, kb, q);
                if (self_culled && self >= kb && self < kb + CLUSTER) sm &= ~(1u << (self - kb));
#ifdef RT_LANE_STATS
                smc = __builtin_popcount(sm);
#endif
                while (__builtin_amdgcn_ballot_w64(sm != 0u) != 0ull) {
                    if (sm != 0u) {
                        const int k = kb + __builtin_ctz(sm);
                        sm &= sm - 1u;
                        if (sphere_any<true>(lds, k, o, R, a, a_sane)) { occ = true; sm = 0u; cm = 0u; }
                    }
                }
            }
#ifdef RT_LANE_STATS
            lane_stats(p, 12, smc);
#endif
        }
        RT_MARK(8);
    }
    return occ;
}

// trace.py:7-41, closest hit.  R = normalize(d) and a = R.R are computed once per query.
// anchor: index into the cull table of a point every live lane's ray passes through (0 = camera),
// or -1 for rays with no common anchor (reflections).
template <int MODE>      // 0: wave-uniform cull; 1: the same without float64 sphere records in LDS (sphere_hot); 2: lane-owned traversal of a clustered scene, also without; 3: 2 with re-derived park addresses (Park3)
__device__ __forceinline__ void closest_hit(const Lds &lds, const KParams &p, const V3 &o, const V3 &d, int anchor,
                                            double &t_out, int &idx_out, int &type_out)
{
    const int P = opaque(p.P);
    const int S = p.S;
    const V3 R = renormalize_unit(d);                         // R == normalize(d), intersections.py:13
    const double a = dot3(R, R);
    RT_MARK(0);
#if RT_PREFILTER
    const int canchor = (opaque(p.anchors) > 0) ? anchor : -1;
#endif
    double bestn = __builtin_inf();
    int bidx = -1;
    constexpr bool F32 = MODE >= 1;                           // sphere records: the float32 LDS table (see sphere_hot)
    // lane-owned traversal for the rays without a common anchor (bounce 1 on), where the wave's rays have parted;
    // the primary rays of a tile travel together: the wave-uniform cull below is cheaper for them
    if (MODE >= 2 && lds.NC > 0 && (canchor < 0 || p.lanes_primary)) {
        if (canchor >= 0) lanes_closest<true>(lds, p, canchor, o, R, a, bestn, bidx);
        else lanes_closest<false>(lds, p, -1, o, R, a, bestn, bidx);
    } else
    for (int k0 = 0; k0 < S; k0 += 64) {
      const int n = (S - k0 < 64) ? S - k0 : 64;
#if RT_PREFILTER
      unsigned long long mask;
      // the float32 ray is rebuilt per chunk so that it is not live during the float64 phase
      mask = cull_mask(lds, S, canchor, k0, n, o, R, p.extent2, -1);
      mask &= (n == 64) ? ~0ull : ((1ull << n) - 1ull);       // padding slots certify themselves, except to a NaN ray
#else
      unsigned long long mask = (n == 64) ? ~0ull : ((1ull << n) - 1ull);
#endif
      RT_MARK(1);
      while (mask) {                                          // spheres some live lane might hit, ascending
        const int k = k0 + __builtin_ctzll(mask);
        mask &= mask - 1ull;
        sphere_closest<F32>(lds, p, k, o, R, a, bestn, bidx);
      }
      RT_MARK(2);
    }
    double best = 999.0;                                      // trace.py:17
    int idx = -1, type = HIT_NONE;
    if (bidx >= 0) {
        const double t = div_inrange(bestn, a);               // :31 / :36, once per query
        if (best > t && t > 0.0) { best = t; idx = bidx; type = HIT_SPHERE; }
    }
    const double *pl = lds.recs() + (F32 ? 0 : opaque(p.S) * SPH_STRIDE);
    for (int k = 0; k < P; ++k) {                             // intersections.py:41-68
        const double *g = pl + k * PL_STRIDE;
        double den, num;
        plane_den_num(g, plane_code(p, k), o, d, den, num);   // :52, :59-61
        if (!(__builtin_fabs(den) < 0.001)) {                 // :55
            const double t = div_inrange(num, den);           // :63
            if (best > t && t > 0.0) { best = t; idx = k; type = HIT_PLANE; }
        }
    }
    t_out = best; idx_out = idx; type_out = type;
    RT_MARK(3);
}

// trace.py:92-96: the shadow query only asks "does anything report 0 < t < 999" (any hit).
// Called with the lanes that need the answer active; returns true if occluded.
// anchor = cull-table index of the light the ray points at; self = index of the sphere the ray
// starts on (-1: a plane), whose miss is certified by the origin-form "behind" test.
template <int MODE>
__device__ __forceinline__ bool any_hit(const Lds &lds, const KParams &p, const V3 &o, const V3 &d, int anchor, int self, bool lanes = true)
{
    const int P = opaque(p.P);
    const int S = p.S;
    V3 R{0.0, 0.0, 0.0};
    double a = 1.0;
    // RT_LAZY_RENORM: the float32 cull runs on d itself (within 2^-52 of R: the same float32 values up to the
    // rounding the margins already budget for) and R, a are formed only if some lane's mask is not empty
    constexpr bool LAZY = RT_LAZY_RENORM && MODE == 0;
    if (!LAZY) { R = renormalize_unit(d); a = dot3(R, R); }  // R == normalize(d), intersections.py:13
    RT_MARK(6);
    bool a_sane = (a > 0.999999 && a < 1.000001);
    bool haveR = !LAZY;
    bool occ = false;
#if RT_PREFILTER
    const int canchor = (opaque(p.anchors) > 0) ? anchor : -1;
#endif
    if (MODE >= 2 && lds.NC > 0 && lanes) {
        occ = (canchor >= 0) ? lanes_any<true>(lds, p, canchor, o, R, a, self) : lanes_any<false>(lds, p, -1, o, R, a, self);
    } else
    for (int k0 = 0; k0 < S; k0 += 64) {
      // "is any live lane still unoccluded" compiles to a v_cndmask + v_cmp pair (the bool lives as a lane mask that may hold
      // stale bits of inactive lanes): asked only where the answer can be no and skipping pays — not before the first chunk,
      // not before the first sphere of the first chunk, not before a plane (RT_SHADOW_EXIT_CHECKS = 1: everywhere, as before)
      if ((RT_SHADOW_EXIT_CHECKS || k0 > 0) && __builtin_amdgcn_ballot_w64(!occ) == 0ull) break;
      const int n = (S - k0 < 64) ? S - k0 : 64;
#if RT_PREFILTER
      unsigned long long mask;
      mask = cull_mask(lds, S, canchor, k0, n, o, LAZY ? d : R, p.extent2, self);
      mask &= (n == 64) ? ~0ull : ((1ull << n) - 1ull);
#else
      unsigned long long mask = (n == 64) ? ~0ull : ((1ull << n) - 1ull);
#endif
      if (LAZY && mask && !haveR) { R = renormalize_unit(d); a = dot3(R, R); a_sane = (a > 0.999999 && a < 1.000001); haveR = true; }
      RT_MARK(7);
      while (mask) {
        if (RT_SHADOW_EXIT_CHECKS && __builtin_amdgcn_ballot_w64(!occ) == 0ull) break;   // every live lane already occluded
        const int k = k0 + __builtin_ctzll(mask);
        mask &= mask - 1ull;
        if (!occ) occ = sphere_any<(MODE >= 1)>(lds, k, o, R, a, a_sane);
        if (!RT_SHADOW_EXIT_CHECKS && mask && __builtin_amdgcn_ballot_w64(!occ) == 0ull) break;   // (only if spheres remain)
      }
      RT_MARK(8);
    }
    const double *pl = lds.recs() + (MODE >= 1 ? 0 : opaque(p.S) * SPH_STRIDE);
    for (int k = 0; k < P; ++k) {
        if (RT_SHADOW_EXIT_CHECKS && __builtin_amdgcn_ballot_w64(!occ) == 0ull) break;   // (an empty exec mask skips the test below anyway)
        if (!occ) {
            const double *g = pl + k * PL_STRIDE;
            double den, num;
            plane_den_num(g, plane_code(p, k), o, d, den, num);
            if (!(__builtin_fabs(den) < 0.001)) {
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
    RT_MARK(9);
    return occ;
}

// Per-lane ray counters of the counting instantiation (rt_get_stats); empty, and every call a no-op, otherwise.
template <bool COUNT> struct RayCount {
    __device__ __forceinline__ void closest(bool) {}
    __device__ __forceinline__ void hit(bool) {}
    __device__ __forceinline__ void shadow(bool, bool) {}
};
template <> struct RayCount<true> {
    unsigned n_closest = 0, n_issued = 0, n_skipped = 0, n_hit = 0;
    __device__ __forceinline__ void closest(bool alive) { n_closest += alive ? 1u : 0u; }      // trace.py:53
    __device__ __forceinline__ void hit(bool alive) { n_hit += alive ? 1u : 0u; }
    // trace.py:92 issues a shadow query per light and hit; the kernel traces it only where its answer is used (k > 0)
    __device__ __forceinline__ void shadow(bool alive, bool traced) { n_issued += (alive && traced) ? 1u : 0u; n_skipped += (alive && !traced) ? 1u : 0u; }
};

// trace.py:44-112.  On entry `alive` lanes carry a ray (o,d); on exit `alive` is false for lanes
// that missed (the reference's 404 sentinels), rgb is this bounce's colour, (o,d) the next ray.
template <bool PARK, int WGT, bool COUNT, int MODE>
__device__ __forceinline__ void trace_bounce(const Lds &lds, const KParams &p, bool &alive, int anchor,
                                             V3 &o, V3 &d, V3 &rgb, RayCount<COUNT> &cnt)
{
    constexpr bool NOREC = MODE >= 1;         // no float64 sphere records in LDS (sphere_hot)
    const int S = p.S, P = p.P, L = opaque(p.L);
    rgb = V3{0.0, 0.0, 0.0};
    double t = 999.0; int idx = -1, type = HIT_NONE;
    cnt.closest(alive);
    RT_MARK(11);
    if (alive) closest_hit<MODE>(lds, p, o, d, anchor, t, idx, type);   // :53 (idle lanes masked off)
    alive = alive && (type != HIT_NONE);                                      // :56-57
    cnt.hit(alive);
    if (alive) {
        V3 Pt{o.x + t * d.x, o.y + t * d.y, o.z + t * d.z};                   // :60 (1.0*o is exact)
        // PARK: the object's colour is re-read from its LDS record where it is used (volatile: at the point of
        // use) instead of being held in 6 VGPRs across the shadow queries
        // MODE 1 / 2 kernels keep no float64 sphere records in LDS (sphere_hot): the planes' and lights' records start at 0, and a
        // hit sphere's colour comes from the packed scene in global memory — through one flat pointer that serves both cases
        const int plb = NOREC ? 0 : opaque(S) * SPH_STRIDE;
        const int coff = (type == HIT_SPHERE) ? idx * SPH_STRIDE + 4 : plb + idx * PL_STRIDE + 12;
        volatile const lds_f64 *colp = (volatile const lds_f64 *)lds.recs() + coff;
        volatile const double *colf = (NOREC && type == HIT_SPHERE) ? p.scene + coff : lds.recs() + coff;
        // MODE 1: one float32 LDS table holds the spheres' and the planes' colours (exact: the scene is float32)
        typedef __attribute__((address_space(3))) float lds_f32;
        volatile const lds_f32 *col1 = nullptr;
        if constexpr (MODE == 1) col1 = (volatile const lds_f32 *)lds.col32 + 4 * ((type == HIT_SPHERE) ? idx : (int)padS(S, lds.NC) + idx);
        V3 colr{0.0, 0.0, 0.0};
        if constexpr (!PARK) colr = (MODE == 1) ? V3{(double)col1[0], (double)col1[1], (double)col1[2]} : V3{colf[0], colf[1], colf[2]};
        auto col = [&](int c) -> double {
            if constexpr (PARK && MODE == 1) return (double)col1[c];
            else if constexpr (PARK && NOREC) return colf[c];
            else if constexpr (PARK) return colp[c];
            else return c == 0 ? colr.x : (c == 1 ? colr.y : colr.z);
        };
        V3 N, bN;
        if (type == HIT_SPHERE) {                                             // :63-66
            const SphHot g = sphere_hot<NOREC>(lds, idx);
            N = normalize3(V3{Pt.x - g.x, Pt.y - g.y, Pt.z - g.z});           // common.py:94-101
            bN = V3{0.0002 * N.x, 0.0002 * N.y, 0.0002 * N.z};
        } else {                                                              // :68-71
            const double *g = lds.recs() + plb + idx * PL_STRIDE;
            N = V3{g[6], g[7], g[8]};                                         // float32-renormalised, host-side
            bN = V3{g[9], g[10], g[11]};                                      // BIAS*N as the reference rounds it
        }
        rgb = V3{p.amb * col(0), p.amb * col(1), p.amb * col(2)};             // :77 (0 + amb*col)
        Pt = V3{Pt.x + bN.x, Pt.y + bN.y, Pt.z + bN.z};                       // :82-83
        const int self = (type == HIT_SPHERE) ? idx : -1;
        Park3<PARK, WGT, MODE == 3> dpark(lds.acc, 1, lds.wave);      // the incoming direction is only needed again for the reflection
        dpark.set(d);
        RT_MARK(4);

        const double *lt = lds.recs() + plb + opaque(P) * PL_STRIDE;
        for (int m = 0; m < L; ++m) {                                         // :86-102
            const double *g = lt + m * LT_STRIDE;
            const V3 Ld = normalize3(V3{g[0] - Pt.x, g[1] - Pt.y, g[2] - Pt.z});   // common.py:84-91
            const double k = p.lamb * dot3(Ld, N);                            // :99
            // :92-102 — the shadow query's answer is only used when k > 0; it has no other effect,
            // so lanes with k <= 0 (light behind the surface) do not ask.
            cnt.shadow(true, k > 0.0);
            RT_MARK(5);
            if (k > 0.0) {
                // (the shadow rays of the primary hits still travel together: wave-uniform cull unless p.lanes_primary)
                const bool occluded = any_hit<MODE>(lds, p, Pt, Ld, 1 + m, self, anchor != 0 || p.lanes_primary);
                if (!occluded) rgb = V3{rgb.x + k * col(0), rgb.y + k * col(1), rgb.z + k * col(2)};
            }
        }
        d = dpark.get();
        const double c2 = -2.0 * dot3(d, N);                                  // common.py:113-120
        // |d - 2(d.N)N| = 1 up to rounding for unit d, N: the exact unit-vector path applies (it falls
        // back to sqrt-and-divide by itself otherwise)
        const V3 Rd = renormalize_unit(V3{d.x + c2 * N.x, d.y + c2 * N.y, d.z + c2 * N.z});
        o = V3{Pt.x + 0.0002 * Rd.x, Pt.y + 0.0002 * Rd.y, Pt.z + 0.0002 * Rd.z};   // :110
        d = Rd;
        RT_MARK(10);
    }
}


// trace.py:115-133.  Bounce 0 rays all start at the camera (cull anchor 0); later bounces have none.
template <bool PARK, int WGT, bool COUNT, int MODE>       // MODE: 0 plain, 1 plain without float64 sphere records, 2 lane-owned traversal
__device__ __forceinline__ V3 sample(const Lds &lds, const KParams &p, bool alive, V3 o, V3 d, RayCount<COUNT> &cnt)
{
    Park3<PARK, WGT, MODE == 3> acc(lds.acc, 0, lds.wave);           // the running colour
    acc.set(V3{0.0, 0.0, 0.0});
    for (int b = 0; b <= p.depth; ++b) {
        if (__builtin_amdgcn_ballot_w64(alive) == 0ull) break;                                   // wave-uniform exit
        if constexpr (COUNT) {                                                // lane utilisation per bounce: waves entering, lanes alive
            const unsigned long long am = __builtin_amdgcn_ballot_w64(alive);
            if ((threadIdx.x & 63u) == 0u && p.ray_counts) {
                atomicAdd(&p.ray_counts[4 + 2 * b], 1ull);
                atomicAdd(&p.ray_counts[5 + 2 * b], (unsigned long long)__builtin_popcountll(am));
            }
        }
        V3 rgb;
#ifdef RT_REGION_STATS
        ((volatile unsigned *)lds.reg)[(threadIdx.x >> 6) * 32 + 30] = b >= 2 ? 12u : 0u;
#endif
        trace_bounce<PARK, WGT, COUNT, MODE>(lds, p, alive, b == 0 ? 0 : -1, o, d, rgb, cnt);
        if (b == 0) acc.set(rgb);                                             // :120
        else {                                                                // :131 (a missed bounce adds pow*0)
            const double wgt = p.refl_pow[b - 1];
            const V3 a = acc.get();
            acc.set(V3{a.x + wgt * rgb.x, a.y + wgt * rgb.y, a.z + wgt * rgb.z});
        }
    }
    return acc.get();
}

__device__ __forceinline__ V3 pixel_P(const KParams &p, int x, int y)
{
    if (p.pixel_loc) {                                                        // kernels.py:19
        const size_t wh = (size_t)p.w * p.h, o = (size_t)x * p.h + y;
        return V3{p.pixel_loc[o], p.pixel_loc[wh + o], p.pixel_loc[2 * wh + o]};
    }
    return V3{p.px, (double)x * p.dy + p.y0, (double)y * p.dz + p.z0};        // scene/camera.py:18-26
}

// Lattice point (i, j) of the half-pixel lattice of the closed-form grid: even indices are pixel centres, odd ones
// the reference's midpoints 0.5 Pa + 0.5 Pb of the two neighbouring centres (kernels.py:43-50: c1*a + c2*b, two
// products and one sum per component; either operand order gives the same bits).  The first component is
// 0.5 px + 0.5 px = px exactly.
__device__ __forceinline__ V3 lattice_P(const KParams &p, int i, int j)
{
    const double ya = (double)(i >> 1) * p.dy + p.y0, yb = (double)((i + 1) >> 1) * p.dy + p.y0;
    const double za = (double)(j >> 1) * p.dz + p.z0, zb = (double)((j + 1) >> 1) * p.dz + p.z0;
    return V3{p.px, (i & 1) ? 0.5 * ya + 0.5 * yb : ya, (j & 1) ? 0.5 * za + 0.5 * zb : za};
}

__device__ __forceinline__ V3 primary_dir(const KParams &p, const V3 &P)
{
    const V3 v{p.cam_R[0] * P.x + p.cam_R[1] * P.y + p.cam_R[2] * P.z,       // kernels.py:22, common.py:40-49
               p.cam_R[3] * P.x + p.cam_R[4] * P.y + p.cam_R[5] * P.z,
               p.cam_R[6] * P.x + p.cam_R[7] * P.y + p.cam_R[8] * P.z};
    return normalize3(v);                                                     // kernels.py:23
}

// Sub-pixel jitter of sample s of pixel (x,y) for the stochastic mode: a counter hash (no state, any launch
// tiling gives the same frame).  u = ((h & 0xFFFF) + 1/2)/65536 - 1/2, v likewise from the high half: exact.
__device__ __forceinline__ unsigned jitter_hash(unsigned x, unsigned y, unsigned s, unsigned seed)
{
    unsigned h = seed ^ 0x9E3779B9u;
    h = (h ^ x) * 0x85EBCA6Bu; h ^= h >> 13;
    h = (h ^ y) * 0xC2B2AE35u; h ^= h >> 16;
    h = (h ^ s) * 0x27D4EB2Fu; h ^= h >> 15;
    h *= 0x165667B1u; h ^= h >> 13;
    return h;
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

// kernels.py:69-73: clip and store one pixel; off = (x - x0) h + y.
// fo = element offset of this launch's frame (rt_render_sequence; 0 for a single frame)
__device__ __forceinline__ void store_pixel(const KParams &p, long long off, long long fo, double R, double G, double B)
{
    if (p.out_u8) {                                                           // common.py:60-63
        const uint8_t r8 = clip_color(R), g8 = clip_color(G), b8 = clip_color(B);
        const uint8_t c1 = p.u8_rgb ? g8 : b8, c2 = p.u8_rgb ? b8 : g8;
        uint8_t *o8 = p.out_u8 + fo;
        if (p.u8_hwc) {                                                       // image layout: row y, column x, 3 bytes
            const long long xr = off / p.h, yy = off - xr * p.h;
            uint8_t *px = o8 + (yy * p.plane_stride + xr) * 3;
            px[0] = r8; px[1] = c1; px[2] = c2;
        } else {
            o8[off] = r8;
            o8[p.plane_stride + off] = c1;
            o8[2 * p.plane_stride + off] = c2;
        }
    }
    if (p.out_f32) {
        float *o32 = p.out_f32 + fo;
        o32[off] = (float)R;
        o32[p.plane_stride + off] = (float)G;
        o32[2 * p.plane_stride + off] = (float)B;
    }
}

// LDS image: [float64 records][per-thread slots 6|9 x 256 doubles][256 int32 pixel offsets][float32 sphere table S x 4][cull table anchors x S x CULL_STRIDE]
__host__ __device__ inline size_t lds_doubles(int S, int P, int L) { return (size_t)S * SPH_STRIDE + (size_t)P * PL_STRIDE + (size_t)L * LT_STRIDE; }
__host__ __device__ inline int lds_slots(bool aa, bool park, bool mode2 = false) { return park ? ((aa && !mode2) ? 9 : 6) : 0; }   // x workgroup-size doubles (MODE 2: the tap sums stay in registers)
__host__ __device__ inline int lds_offset_words(bool park, int wgt) { return park ? wgt : 0; }    // + one int32 per thread: the pixel offset
// The float32 tables of a scene, offsets in floats (every one a multiple of 4):
//   sph32 | anchored table | cluster anchored table | cluster boxes | group boxes | group anchored table | cluster sph32 | colours
// (colours: {R,G,B,-} of the S spheres, padded to Sp entries, then of the P planes — exact, the scene is float32; only
// the MODE 1 kernels, which keep no float64 sphere records, stage and read them: `total_col` floats instead of `total`)
// The lane-owned traversal never reads the clusters' origin-form spheres (it tests boxes), so its kernels stage — and
// reserve LDS for — everything but that last table (`lanes`): config 5's image stays under the 4-workgroups-per-CU line.
struct TableLayout { size_t tab, ctab, cbox, gbox, gtab, csph32, total_lanes, total, col32, total_col; };
__host__ __device__ inline TableLayout table_layout(int S, int NC, int anchors, int P = 0)
{
    const size_t Sp = padS(S, NC), NCp = pad4(NC), NG = supers(NC), NGp = pad4((int)NG);
    TableLayout t;
    size_t o = 4 * Sp;
    t.tab = o;    o += (size_t)anchors * Sp * CULL_STRIDE;
    t.ctab = o;   o += (size_t)anchors * NCp * CULL_STRIDE;
    t.cbox = o;   o += NCp * BOX_STRIDE;
    t.gbox = o;   o += NG * BOX_STRIDE;
    t.gtab = o;   o += (size_t)anchors * NGp * CULL_STRIDE;
    t.total_lanes = o;
    t.csph32 = o; o += 4 * NCp;
    t.total = o;
    t.col32 = o;  o += 4 * (Sp + (size_t)pad4(P));
    t.total_col = o;
    return t;
}
__host__ __device__ inline size_t table_floats(int S, int NC, int anchors, bool lanes = false, bool col = false, int P = 0)
{
    const TableLayout t = table_layout(S, NC, anchors, P);
    return col ? t.total_col : (lanes ? t.total_lanes : t.total);
}
// mode2: the kernels of the large clustered scenes (lane-owned traversal) stage no float64 sphere records (sphere_hot)
__host__ __device__ inline size_t lds_bytes(int S, int P, int L, int NC, int anchors, bool aa, bool park, int wgt, bool lanes = false, bool mode2 = false, bool norec = false)
{
    return (lds_doubles((mode2 || norec) ? 0 : S, P, L) + (size_t)lds_slots(aa, park, mode2) * wgt) * sizeof(double) +
           ((size_t)lds_offset_words(park, wgt) + table_floats(S, NC, anchors, lanes, norec, P)) * sizeof(float) + 16   // + workgroup cost/arrival words
#ifdef RT_REGION_STATS
           + (size_t)(wgt / 64) * 32 * sizeof(unsigned)
#endif
           ;
}

// The float32 cull tables (exact sphere table for the origin form; {A-c, tau} per anchor and sphere; the same two
// for the cluster bounding spheres), in the order and layout the render kernel keeps them in LDS.  They depend on
// the scene, the camera position (anchor 0) and floor_anch only, so the host runs this once per change of those
// (one workgroup, a few microseconds) and every render workgroup copies the result.
__global__ __launch_bounds__(TABLE_THREADS) void tables_kernel(const KParams p, float *__restrict__ out)
{
    const int nrec = (int)lds_doubles(p.S, p.P, p.L);
    const int Sp = padS(p.S, p.NC), NCp = pad4(p.NC);
    const TableLayout tl = table_layout(p.S, p.NC, p.anchors, p.P);
    float *sph32 = out;
    float *tab = out + tl.tab;                         // anchors x Sp entries
    float *csph32 = out + tl.csph32;
    float *ctab = out + tl.ctab;                       // anchors x NCp entries
    const double *rec = p.scene;
    const float NINF = -__builtin_inff();
    for (int k = threadIdx.x; k < Sp; k += TABLE_THREADS) {   // exact: the scene is float32
        const double *g = rec + k * SPH_STRIDE;
        const bool real = k < p.S;
        sph32[4 * k + 0] = real ? (float)g[0] : 0.0f; sph32[4 * k + 1] = real ? (float)g[1] : 0.0f;
        sph32[4 * k + 2] = real ? (float)g[2] : 0.0f; sph32[4 * k + 3] = real ? (float)g[3] : NINF;
    }
    float *col32 = out + tl.col32;
    for (int k = threadIdx.x; k < Sp + pad4(p.P); k += TABLE_THREADS) {   // colours of spheres and planes (MODE 1 kernels)
        const double *g = k < p.S ? rec + k * SPH_STRIDE + 4 : rec + p.S * SPH_STRIDE + (k - Sp) * PL_STRIDE + 12;
        const bool real = k < p.S || (k >= Sp && k - Sp < p.P);
        for (int c = 0; c < 3; ++c) col32[4 * k + c] = real ? (float)g[c] : 0.0f;
        col32[4 * k + 3] = 0.0f;
    }
    const double *lt = rec + p.S * SPH_STRIDE + p.P * PL_STRIDE;
    for (int e = threadIdx.x; e < p.anchors * Sp; e += TABLE_THREADS) {
        const int a = e / Sp, k = e - a * Sp;
        float *t = tab + (size_t)e * CULL_STRIDE;
        if (k >= p.S) { t[0] = t[1] = t[2] = 0.0f; t[3] = -NINF; continue; }   // padding: always culled
        const double *g = rec + k * SPH_STRIDE;
        const double ax = a ? lt[(a - 1) * LT_STRIDE + 0] : p.cam_o[0];
        const double ay = a ? lt[(a - 1) * LT_STRIDE + 1] : p.cam_o[1];
        const double az = a ? lt[(a - 1) * LT_STRIDE + 2] : p.cam_o[2];
        const double lx = ax - g[0], ly = ay - g[1], lz = az - g[2];
        const double ll = lx * lx + ly * ly + lz * lz;
        t[0] = (float)lx; t[1] = (float)ly; t[2] = (float)lz;
        t[3] = anchored_tau(ll, g[3], p.floor_anch);
    }
    // the same two tables for the cluster bounding spheres (float64 records after the lights; rounding the
    // centre to float32 is covered by rounding R2 up)
    const double *cl = rec + nrec;
    for (int c = threadIdx.x; c < NCp; c += TABLE_THREADS) {
        const bool real = c < p.NC;
        const double *g = cl + (real ? c : 0) * CL_STRIDE;
        csph32[4 * c + 0] = real ? (float)g[0] : 0.0f; csph32[4 * c + 1] = real ? (float)g[1] : 0.0f;
        csph32[4 * c + 2] = real ? (float)g[2] : 0.0f;
        csph32[4 * c + 3] = real ? (float)(g[3] * (1.0 + 0x1p-20)) : NINF;
    }
    for (int e = threadIdx.x; e < p.anchors * NCp; e += TABLE_THREADS) {
        const int a = e / NCp, c = e - a * NCp;
        float *t = ctab + (size_t)e * CULL_STRIDE;
        if (c >= p.NC) { t[0] = t[1] = t[2] = 0.0f; t[3] = -NINF; continue; }
        const double *g = cl + c * CL_STRIDE;
        const double ax = a ? lt[(a - 1) * LT_STRIDE + 0] : p.cam_o[0];
        const double ay = a ? lt[(a - 1) * LT_STRIDE + 1] : p.cam_o[1];
        const double az = a ? lt[(a - 1) * LT_STRIDE + 2] : p.cam_o[2];
        const double lx = ax - g[0], ly = ay - g[1], lz = az - g[2];
        const double ll = lx * lx + ly * ly + lz * lz;
        t[0] = (float)lx; t[1] = (float)ly; t[2] = (float)lz;
        t[3] = anchored_tau(ll, g[3], p.floor_anch);
    }
    // bounding boxes of the clusters' spheres (radius = sqrt of the float32 r*r the reference tests against, a little
    // more), every face rounded outward to float32
    float *cbox = out + tl.cbox;
    for (int c = threadIdx.x; c < NCp; c += TABLE_THREADS) {
        double lo[3] = {1e30, 1e30, 1e30}, hi[3] = {-1e30, -1e30, -1e30};
        for (int k = c * CLUSTER; k < (c + 1) * CLUSTER && k < p.S; ++k) {
            const double *g = rec + k * SPH_STRIDE;
            const double rad = __builtin_sqrt(g[3]) * (1.0 + 0x1p-20) + 1e-30;
            for (int i = 0; i < 3; ++i) { lo[i] = __builtin_fmin(lo[i], g[i] - rad); hi[i] = __builtin_fmax(hi[i], g[i] + rad); }
        }
        float *b = cbox + (size_t)c * BOX_STRIDE;
        for (int i = 0; i < 3; ++i) {                  // a float is within 2^-24 |x| of x: step 2^-22 |x| outward first
            b[i] = (float)(lo[i] - (__builtin_fabs(lo[i]) * 0x1p-22 + 1e-30));
            b[4 + i] = (float)(hi[i] + (__builtin_fabs(hi[i]) * 0x1p-22 + 1e-30));
        }
        b[3] = 0.0f; b[7] = 0.0f;
    }
    __syncthreads();
    for (int g = threadIdx.x; g < supers(p.NC); g += TABLE_THREADS) {       // the union of the group's (outward-rounded) boxes
        float *b = out + tl.gbox + (size_t)g * BOX_STRIDE;
        for (int i = 0; i < 3; ++i) { b[i] = __builtin_inff(); b[4 + i] = -__builtin_inff(); }
        for (int c = g * SUPER; c < (g + 1) * SUPER && c < p.NC; ++c)
            for (int i = 0; i < 3; ++i) {
                b[i] = __builtin_fminf(b[i], cbox[(size_t)c * BOX_STRIDE + i]);
                b[4 + i] = __builtin_fmaxf(b[4 + i], cbox[(size_t)c * BOX_STRIDE + 4 + i]);
            }
        b[3] = 0.0f; b[7] = 0.0f;
    }
    // the anchored table of the groups' bounding spheres (float64 records behind the clusters')
    const int NG = supers(p.NC), NGp = pad4(NG);
    const double *gr = cl + (size_t)p.NC * CL_STRIDE;
    for (int e = threadIdx.x; e < p.anchors * NGp; e += TABLE_THREADS) {
        const int a = e / NGp, g = e - a * NGp;
        float *t = out + tl.gtab + (size_t)e * CULL_STRIDE;
        if (g >= NG) { t[0] = t[1] = t[2] = 0.0f; t[3] = -NINF; continue; }
        const double *c = gr + g * CL_STRIDE;
        const double ax = a ? lt[(a - 1) * LT_STRIDE + 0] : p.cam_o[0];
        const double ay = a ? lt[(a - 1) * LT_STRIDE + 1] : p.cam_o[1];
        const double az = a ? lt[(a - 1) * LT_STRIDE + 2] : p.cam_o[2];
        const double lx = ax - c[0], ly = ay - c[1], lz = az - c[2];
        const double ll = lx * lx + ly * ly + lz * lz;
        t[0] = (float)lx; t[1] = (float)ly; t[2] = (float)lz;
        t[3] = anchored_tau(ll, c[3], p.floor_anch);
    }
}

// AA = false: aliasing off — instantiated separately so that the common case does not carry the tap loop's
// live state (registers decide occupancy here).
template <bool AA, bool PARK, int WPW, bool COUNT = false, bool LAT = false, int MODE = 0>
#ifndef RT_W_PARK
#define RT_W_PARK 7
#endif
#ifndef RT_W_AAPARK
#define RT_W_AAPARK 7   // 72 VGPRs with a few spills (76 B/lane of scratch) still beat 5 waves/SIMD without: -9 %
#endif
#ifndef RT_W_LANES
#define RT_W_LANES 4    // lane-owned traversal (clustered scenes: the LDS image bounds the occupancy at about 4 anyway) wants registers
#endif
__global__ __launch_bounds__(64 * WPW, MODE >= 2 ? RT_W_LANES : (AA ? (PARK ? RT_W_AAPARK : 4) : (PARK ? RT_W_PARK : 5))) void render_kernel(const KParams p)
{
    constexpr int WG_THREADS = 64 * WPW, WAVES_PER_WG = WPW;
    constexpr bool M2 = MODE >= 2;
    constexpr bool NOREC = MODE >= 1;
    const int nrec = (int)lds_doubles(NOREC ? 0 : p.S, p.P, p.L);             // MODE 1 / 2: planes and lights only (sphere_hot)
    const double *rec_src = p.scene + (NOREC ? (size_t)p.S * SPH_STRIDE : 0);
    double *accum = lds_raw + nrec;
    int *offw = reinterpret_cast<int *>(accum + lds_slots(AA, PARK, M2) * WG_THREADS);
    float *sph32 = reinterpret_cast<float *>(offw + lds_offset_words(PARK, WG_THREADS));
    const TableLayout tl = table_layout(p.S, p.NC, p.anchors, p.P);
    // the lane-owned kernels leave the clusters' origin-form spheres in global memory: with anchored tables in place the only
    // rays without an anchor are the reflections, and those test boxes
    const bool LANES = MODE >= 2 && p.anchors > 0;
    float *tab = sph32 + tl.tab;                       // anchors x Sp entries
    float *csph32 = LANES ? nullptr : sph32 + tl.csph32;
    float *ctab = sph32 + tl.ctab;                     // anchors x NCp entries
    float *cbox = sph32 + tl.cbox;                     // NCp boxes
    const size_t ntab = MODE == 1 ? tl.total_col : (LANES ? tl.total_lanes : tl.total);
    unsigned *wgstat = reinterpret_cast<unsigned *>(sph32 + ntab);   // {cycles, waves done}
    if (threadIdx.x == 0) { wgstat[0] = 0u; wgstat[1] = 0u; }
    {   // stage the packed scene and its float32 cull tables once per workgroup: two straight copies.  (The tables
        // used to be computed here, by every workgroup: 3 % of the frame's VALU instructions and a second barrier.)
        for (int i = threadIdx.x; i < nrec; i += WG_THREADS) lds_raw[i] = rec_src[i];
#if RT_PREFILTER
        const int nf4 = (int)(ntab / 4);
        const f4 *src = reinterpret_cast<const f4 *>(p.ftab);
        f4 *dst = reinterpret_cast<f4 *>(sph32);
        for (int i = threadIdx.x; i < nf4; i += WG_THREADS) dst[i] = src[i];
#endif
    }
    __syncthreads();
    // two-wave workgroups serve the small flat scenes only (the host sends every clustered scene to workgroups of 4): with
    // NC a constant 0 there, none of the cluster code is compiled into those kernels (the headline kernel sits in a narrow
    // register optimum)
#ifdef RT_REGION_STATS
    const Lds lds{sph32, tab, csph32, ctab, cbox, sph32 + tl.gbox, sph32 + tl.gtab, MODE == 1 ? sph32 + tl.col32 : nullptr, WPW == 2 ? 0 : p.NC, __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), MODE < 2 && (WPW >= 4 || RT_CULL_PAIRS_WPW2), MODE >= 2, wgstat + 4, accum};
    for (int i = threadIdx.x & 63; i < 32; i += 64) lds.reg[(threadIdx.x >> 6) * 32 + i] = 0u;
    if ((threadIdx.x & 63) == 0) lds.reg[(threadIdx.x >> 6) * 32 + 31] = (unsigned)__builtin_amdgcn_s_memtime();
#else
    const Lds lds{sph32, tab, csph32, ctab, cbox, sph32 + tl.gbox, sph32 + tl.gtab, MODE == 1 ? sph32 + tl.col32 : nullptr, WPW == 2 ? 0 : p.NC, __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), MODE < 2 && (WPW >= 4 || RT_CULL_PAIRS_WPW2), MODE >= 2, accum};
#endif

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // Longest-first dispatch: the hardware hands out workgroups in blockIdx order, so blockIdx indexes a
    // permutation of the tile blocks sorted by the cycles they took in the previous launch (order_kernel).
    // Any permutation renders every tile exactly once; only the length of the launch's tail depends on it.
    // A launch may carry several frames of the same scene and camera (rt_render_sequence): workgroups [f bpf, (f+1) bpf)
    // render frame f, each frame in the same order, so one frame's last workgroups run beside the next frame's first.
    // Only the LAST frame of a launch decides how raggedly the launch ends, so only it is dispatched longest-first; the
    // frames before it run every XCD's blocks in tile order, neighbours in y back to back on the XCD that owns their group:
    // the 128-byte lines they share are completed in that L2 within microseconds.
    int bid = (int)blockIdx.x;
    const unsigned *ord = p.order;
    if (p.nframes > 1) {
        const int frame = div_by(bid, p.bpf, p.bpf_magic, p.bpf_shift);
        bid -= frame * p.bpf;
        if (ord && frame < p.nframes - 1) ord += p.seq_offset;
    }
    // Four-wave kernels may dispatch TILE by tile (p.order_tiles): a workgroup holds its LDS image and — by the time three of its
    // waves have ended — three idle wave slots until its last wave ends, so its waves should be tiles of EQUAL cost, not
    // neighbours (measured on config 5: 3.26 of 4 wave slots per SIMD occupied on average with neighbours).
#ifndef RT_TILE_ORDER_MIN_WPW
#define RT_TILE_ORDER_MIN_WPW 4
#endif
    const bool by_tile = WPW >= RT_TILE_ORDER_MIN_WPW && p.order_tiles;
    const int block = by_tile ? bid : (ord ? (int)ord[bid] : bid);
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int tile_v = by_tile ? (ord ? (int)ord[bid * WAVES_PER_WG + wave_u] : bid * WAVES_PER_WG + wave_u) : block * WAVES_PER_WG + wave;
    // (wave-uniform, but loaded through a vector register: as a scalar it does not occupy a VGPR until the cost is recorded
    // at the end — the MODE 1 kernel spilled it, 8 bytes of scratch written per lane; the two-wave kernels keep their register
    // allocation as it is)
    const int tile = WPW >= RT_TILE_ORDER_MIN_WPW ? __builtin_amdgcn_readfirstlane(tile_v) : tile_v;
    if (tile >= p.ntiles) return;                                             // whole wave, after the barriers
    const unsigned long long t_begin = (p.tile_cycles || p.cost) ? __builtin_amdgcn_s_memtime() : 0ull;
    const int tx = div_by(tile, p.tiles_y, p.tiles_y_magic, p.tiles_y_shift), ty = tile - tx * p.tiles_y;
    const int x = p.x0 + tx * TILE + (lane >> 3);
    const int y = ty * TILE + (lane & 7);
    bool inb = (x < p.x1) && (y < p.h);
    // LAT: of the lattice, only pixel centres (even, even) and the points inside the outermost centres are ever summed
    if constexpr (LAT) inb = inb && ((((x | y) & 1) == 0) || (x >= 1 && x <= p.w - 2 && y >= 1 && y <= p.h - 2));
    const int xc = inb ? x : p.x0, yc = inb ? y : 0;                          // keep addresses valid for idle lanes
    // PARK: the pixel's output offset waits in LDS instead of staying live (or being spilled to scratch)
    // across the whole trace; -1 marks lanes outside the frame.
    typedef __attribute__((address_space(3))) int lds_i32;
    // (REMAT kernels re-derive it at the end from the tile, a scalar, and the lane's number instead: see Park3)
    constexpr bool OFF_REMAT = PARK && MODE == 3;
    volatile lds_i32 *offp = OFF_REMAT ? nullptr : (volatile lds_i32 *)offw + threadIdx.x;
    if constexpr (PARK && !OFF_REMAT) *offp = inb ? (x - p.x0) * p.h + y : -1;   // w*h <= 2^31 (checked by the host)

    const V3 o{p.cam_o[0], p.cam_o[1], p.cam_o[2]};                           // kernels.py:16
    RayCount<COUNT> cnt;
    double R, G, B;
    if constexpr (!AA) {
        const V3 c = sample<PARK, WG_THREADS, COUNT, MODE>(lds, p, inb, o, primary_dir(p, LAT ? lattice_P(p, xc, yc) : pixel_P(p, xc, yc)), cnt);   // kernels.py:19-26
        R = c.x; G = c.y; B = c.z;
    } else {
        // kernels.py:26-65 as ONE loop: tap 0 is the centre sample, taps 1-8 the half-pixel neighbours (only
        // if some lane of the wave is interior).  One inlined copy of sample(); the tap sums live in LDS when
        // PARK.  The same loop serves the build-defined stochastic mode (p.aa == 2): p.spp jittered samples.
        const bool stoch = (p.aa == 2);
        const bool interior = !stoch && inb && x >= 1 && x <= p.w - 2 && y >= 1 && y <= p.h - 2;
        const int ntaps = stoch ? p.spp : ((__builtin_amdgcn_ballot_w64(interior) != 0ull) ? 9 : 1);
        // neighbour offsets (dx,dy)+1 packed 2 bits each, in the order of kernels.py:53:
        // left, right, top(y+1), bottom(y-1), top-left, top-right, bottom-left, bottom-right
        constexpr unsigned NBX = 0x8858u, NBY = 0x0A25u;
        Park3<PARK && !M2, WG_THREADS> taps(lds.acc, 2);
#pragma unroll 1
        for (int tap = 0; tap < ntaps; ++tap) {
            const V3 Pp = pixel_P(p, xc, yc);                                 // :19
            V3 Pt = Pp;
            if (stoch) {
                const unsigned hh = jitter_hash((unsigned)xc, (unsigned)yc, (unsigned)tap, p.seed);
                const double u = (double)(hh & 0xFFFFu) * 0x1p-16 + (0x1p-17 - 0.5);
                const double v = (double)(hh >> 16) * 0x1p-16 + (0x1p-17 - 0.5);
                Pt = V3{Pp.x, Pp.y + u * p.dy, Pp.z + v * p.dz};
            } else if (tap) {
                const int k = tap - 1;
                const int ddx = (int)((NBX >> (2 * k)) & 3u) - 1, ddy = (int)((NBY >> (2 * k)) & 3u) - 1;
                const V3 Pn = pixel_P(p, interior ? x + ddx : xc, interior ? y + ddy : yc);
                Pt = V3{0.5 * Pp.x + 0.5 * Pn.x, 0.5 * Pp.y + 0.5 * Pn.y, 0.5 * Pp.z + 0.5 * Pn.z};   // :43-50
            }
            const V3 s = sample<PARK, WG_THREADS, COUNT, MODE>(lds, p, (tap && !stoch) ? interior : inb, o, primary_dir(p, Pt), cnt);   // :26 / :56
            if (tap == 0) taps.set(s);
            else if (stoch) { const V3 a = taps.get(); taps.set(V3{a.x + s.x, a.y + s.y, a.z + s.z}); }
            else if (interior) { const V3 a = taps.get(); taps.set(V3{a.x + s.x, a.y + s.z, a.z + s.y}); }   // :58-60 (G += B_s; B += G_s)
        }
        { const V3 a = taps.get(); R = a.x; G = a.y; B = a.z; }
        if (stoch) { const double n = (double)p.spp; R = R / n; G = G / n; B = B / n; }
        else if (interior) { R = R / 9; G = G / 9; B = B / 9; }               // :63-65
    }

    long long off;
    if constexpr (OFF_REMAT) {
        const int l2 = (int)fresh_lane();
        const int x2 = p.x0 + tx * TILE + (l2 >> 3), y2 = ty * TILE + (l2 & 7);
        bool inb2 = (x2 < p.x1) && (y2 < p.h);
        if constexpr (LAT) inb2 = inb2 && ((((x2 | y2) & 1) == 0) || (x2 >= 1 && x2 <= p.w - 2 && y2 >= 1 && y2 <= p.h - 2));
        off = inb2 ? (long long)((x2 - p.x0) * p.h + y2) : -1ll;
    } else if constexpr (PARK) off = *offp; else off = inb ? (long long)(x - p.x0) * p.h + y : -1ll;   // (int32 -1 sign-extends)
    if (off >= 0) {
        if constexpr (LAT) {                                                  // one float64 (R,G,B) per lattice sample
            double *q = p.out_f64 + off * 3;
            q[0] = R; q[1] = G; q[2] = B;
        } else {
            // the frame index is formed again here (a wave-uniform division) instead of being carried through the trace
            const long long fo = opaque(p.nframes) > 1 ? (long long)div_by((int)blockIdx.x, p.bpf, p.bpf_magic, p.bpf_shift) * p.frame_stride : 0ll;
            store_pixel(p, off, fo, R, G, B);
        }
    }
#ifdef RT_REGION_STATS
    RT_MARK(11);
    if (p.tile_cycles && (threadIdx.x & 63) < 24)                            // behind the per-tile cycles: 24 region counters (cycles / 64)
        atomicAdd(p.tile_cycles + p.ntiles + 16 + (threadIdx.x & 63), lds.reg[(threadIdx.x >> 6) * 32 + (threadIdx.x & 63)] >> 6);
#endif
    if constexpr (COUNT) {                                                    // rt_get_stats: one atomic per wave and counter
        unsigned v[4] = {cnt.n_closest, cnt.n_issued, cnt.n_skipped, cnt.n_hit};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) v[c] += __shfl_down(v[c], d);
            if ((threadIdx.x & 63) == 0 && p.ray_counts) atomicAdd(&p.ray_counts[c], (unsigned long long)v[c]);
        }
    }
    if ((p.tile_cycles || p.cost) && (threadIdx.x & 63) == 0) {               // timing only; never feeds a pixel
        const unsigned cyc = (unsigned)(__builtin_amdgcn_s_memtime() - t_begin);
        if (p.tile_cycles) p.tile_cycles[tile] = cyc;
        if (p.cost) {
            if (by_tile) p.cost[tile] = cyc >> 2;                             // per tile
            else {
                // the block's cost = sum over its waves; the wave that finishes last stores it
                const int expected = (p.ntiles - block * WAVES_PER_WG < WAVES_PER_WG) ? p.ntiles - block * WAVES_PER_WG : WAVES_PER_WG;
                atomicAdd(&wgstat[0], cyc >> 2);
                if ((int)atomicAdd(&wgstat[1], 1u) == expected - 1) p.cost[block] = atomicAdd(&wgstat[0], 0u);
            }
        }
    }
}

// RT_AA_REFERENCE, second half (kernels.py:29-65): pixel (x,y) sums the lattice samples around its centre (2x, 2y) in the
// reference's order — centre, left, right, top (y+1), bottom (y-1), top-left, top-right, bottom-left, bottom-right,
// with G += B_s and B += G_s — and divides by 9; frame-border pixels keep their centre sample.  p describes the PIXEL
// frame (w, h, x0, x1, outputs); p.out_f64 / lat_x0 / lat_h the lattice samples.  One thread per pixel, y fastest.
__global__ __launch_bounds__(256) void aa_resolve_kernel(const KParams p)
{
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)(p.x1 - p.x0) * p.h) return;
    const int xr = (int)(idx / p.h), y = (int)(idx - (long long)xr * p.h), x = p.x0 + xr;
    const double *c = p.out_f64 + ((size_t)(2 * x - p.lat_x0) * p.lat_h + 2 * y) * 3;
    double R = c[0], G = c[1], B = c[2];
    if (x >= 1 && x <= p.w - 2 && y >= 1 && y <= p.h - 2) {
        constexpr unsigned NBX = 0x8858u, NBY = 0x0A25u;                      // (dx,dy)+1, 2 bits each, in the order of kernels.py:53
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int ddx = (int)((NBX >> (2 * k)) & 3u) - 1, ddy = (int)((NBY >> (2 * k)) & 3u) - 1;
            const double *s = c + ((long long)ddx * p.lat_h + ddy) * 3;
            R += s[0]; G += s[2]; B += s[1];                                  // :58-60
        }
        R = R / 9; G = G / 9; B = B / 9;                                      // :63-65
    }
    store_pixel(p, idx, 0ll, R, G, B);
}

// Builds the dispatch order of the following launches from the block costs a measuring launch stored: longest-first WITHIN
// every XCD, and XCD-affine.
//
// The dispatcher hands workgroup i to XCD i mod 8, and every XCD has its own L2.  A two-wave workgroup writes 16-pixel runs
// of the output planes (16 B of a uint8 plane, 64 B of a float32 plane); the rest of each 128-byte line belongs to its
// neighbours in y.  When those run on other XCDs every fragment of the line is written back on its own (round 2: WRITE_SIZE
// 41 MB per 1080p frame for 31 MB of pixels); written by one XCD, the line is completed in that L2 and leaves it once.
// So WHICH XCD renders a block is decided for GROUPS of 2^gshift consecutive blocks (y runs fastest: neighbours in y), and
// WHEN it renders it block by block:
//   1. the groups are sorted by decreasing mean block cost (counting sort, 1024 logarithmic buckets) and dealt to the XCDs
//      in serpentine order — rank r goes to XCD r mod 8 in even rows of eight and 7 - r mod 8 in odd ones — so every XCD
//      gets the same number of groups and nearly the same total cost;
//   2. every XCD's blocks are sorted by decreasing cost of their own (a second counting sort, per XCD), and the XCD's j-th
//      block takes position 8 j + xcd of the order.  Round 2's XCD-affine order sorted whole groups (a coarser order: +6..14 %
//      on one stream); here the order inside an XCD is as fine as the plain longest-first order.
// The groups that do not fill a row of eight and the short group at the end of the frame share the positions behind those
// rows, longest first.  The output is a permutation of [0, nblocks) by construction; any permutation renders the same frame.
//   3. a second permutation, order[nblocks ..], keeps the XCD assignment of step 1 but runs every XCD's blocks in plain tile
//      order (group after group, the blocks of a group back to back).  All but the last frame of a multi-frame launch use
//      it: in the middle of a launch every CU is busy whatever the order, and blocks that share output lines then follow
//      each other on their XCD within microseconds (step 2's order scatters a group's blocks over the XCD's whole list:
//      measured 37 MB of writes per headline frame instead of 41 for 31 MB of pixels; tile order inside the XCD: see
//      profiles/r03_order_group_sweep.txt).
// gtmp: nblocks / 2^gshift + 1 words, btmp: nblocks words of scratch; order: 2 nblocks words.
// wshift > 0: the items are TILES and a workgroup takes 2^wshift consecutive positions of the order (KParams::order_tiles):
// XCD x's t-th item then sits at position ((8 (t >> wshift) + x) << wshift) + (t mod 2^wshift) — its (t >> wshift)-th workgroup —
// and sorting an XCD's tiles by cost makes every workgroup four tiles of (nearly) equal cost.  nvalid: items that exist (the
// last workgroup's missing tiles cost nothing).  gshift counts items (>= wshift).
__global__ __launch_bounds__(ORDER_THREADS) void order_kernel(const unsigned *__restrict__ cost, unsigned *__restrict__ gtmp,
                                                               unsigned *__restrict__ btmp, unsigned *__restrict__ order, int nblocks, int gshift,
                                                               int wshift, int nvalid)
{
    const unsigned wm = (1u << wshift) - 1u;
    auto place = [&](unsigned x, unsigned t, unsigned tail_) -> unsigned {
        return x < (unsigned)ORDER_XCDS ? ((((unsigned)ORDER_XCDS * (t >> wshift) + x) << wshift) | (t & wm)) : tail_ + t;
    };
    auto cost_of = [&](int it) -> unsigned { return it < nvalid ? cost[it] : 0u; };
    __shared__ unsigned hist[(ORDER_XCDS + 1) * ORDER_BUCKETS];     // [class][bucket]; class ORDER_XCDS = the leftover blocks
    __shared__ unsigned scan[ORDER_THREADS / 64];
    static_assert(ORDER_THREADS == ORDER_BUCKETS, "thread i owns bucket 1023 - i");
    const int i = threadIdx.x, lane = i & 63, wv = i >> 6;
    const int gb = 1 << gshift;
    const int full = nblocks >> gshift;                 // groups of gb blocks; a shorter one may follow
    const int rows = full / ORDER_XCDS;                 // rows of eight groups: one group per XCD each
    // in place: hist[c][b] (counts per bucket) -> exclusive offsets in descending bucket order, for classes [0, nc)
    auto scan_classes = [&](int nc) {
        for (int c = 0; c < nc; ++c) {
            __syncthreads();
            const unsigned v = hist[c * ORDER_BUCKETS + ORDER_BUCKETS - 1 - i];
            unsigned incl = v;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const unsigned up = __shfl_up(incl, d);
                if (lane >= d) incl += up;
            }
            if (lane == 63) scan[wv] = incl;            // wave totals
            __syncthreads();
            unsigned base = 0;
            for (int w = 0; w < wv; ++w) base += scan[w];
            hist[c * ORDER_BUCKETS + ORDER_BUCKETS - 1 - i] = base + incl - v;
        }
        __syncthreads();
    };
    for (int k = i; k < (ORDER_XCDS + 1) * ORDER_BUCKETS; k += ORDER_THREADS) hist[k] = 0u;
    __syncthreads();
    // 1. groups by mean block cost
    for (int g = i; g < full; g += ORDER_THREADS) {
        unsigned long long sum = 0;
        for (int k = 0; k < gb; ++k) sum += cost_of((g << gshift) + k);
        const unsigned mean = (unsigned)(sum >> gshift);
        const int bkt = order_bucket(mean);
        gtmp[g] = ((unsigned)bkt << 20) | atomicAdd(&hist[bkt], 1u);
    }
    scan_classes(1);
    for (int g = i; g < full; g += ORDER_THREADS) {
        const unsigned s = gtmp[g];
        const int r = (int)(hist[s >> 20] + (s & 0xFFFFFu));                 // rank among the groups, most expensive first
        const int q = r / ORDER_XCDS, c = r - q * ORDER_XCDS;
        gtmp[g] = q < rows ? (unsigned)((q & 1) ? ORDER_XCDS - 1 - c : c) : (unsigned)ORDER_XCDS;
    }
    __syncthreads();
    const unsigned tail = (unsigned)(ORDER_XCDS * rows * gb);                // first position behind the rows
    {   // 3. the same XCD assignment in tile order: thread i owns the groups [i per, (i + 1) per); for every class an exclusive
        // scan over the threads of how many of its groups they own gives each group its rank within its class
        unsigned *seq = order + nblocks;
        const int per = (full + ORDER_THREADS - 1) / ORDER_THREADS;
        const int g0 = i * per < full ? i * per : full, g1 = g0 + per < full ? g0 + per : full;
        unsigned base[ORDER_XCDS + 1];
        for (int c = 0; c <= ORDER_XCDS; ++c) {
            unsigned v = 0;
            for (int g = g0; g < g1; ++g) v += gtmp[g] == (unsigned)c ? 1u : 0u;
            unsigned incl = v;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const unsigned up = __shfl_up(incl, d);
                if (lane >= d) incl += up;
            }
            __syncthreads();
            if (lane == 63) scan[wv] = incl;
            __syncthreads();
            unsigned b0 = 0;
            for (int w = 0; w < wv; ++w) b0 += scan[w];
            base[c] = b0 + incl - v;
        }
        for (int g = g0; g < g1; ++g) {
            const unsigned c = gtmp[g];
            unsigned jg = 0;
#pragma unroll
            for (int cc = 0; cc <= ORDER_XCDS; ++cc) if (c == (unsigned)cc) jg = base[cc]++;
            for (int k = 0; k < gb; ++k) {
                const unsigned j = jg * (unsigned)gb + (unsigned)k;
                seq[place(c, j, tail)] = (unsigned)((g << gshift) + k);
            }
        }
        for (int b = (full << gshift) + i; b < nblocks; b += ORDER_THREADS) seq[b] = (unsigned)b;    // the short group: last, in place
    }
    __syncthreads();
    for (int k = i; k < (ORDER_XCDS + 1) * ORDER_BUCKETS; k += ORDER_THREADS) hist[k] = 0u;
    __syncthreads();
    // 2. blocks by their own cost, per XCD
    for (int b = i; b < nblocks; b += ORDER_THREADS) {
        const int g = b >> gshift;
        const unsigned x = g < full ? gtmp[g] : (unsigned)ORDER_XCDS;
        const int bkt = order_bucket(cost_of(b));
        btmp[b] = ((unsigned)bkt << 22) | atomicAdd(&hist[x * ORDER_BUCKETS + bkt], 1u);                  // items < 2^22 (host)
    }
    scan_classes(ORDER_XCDS + 1);
    for (int b = i; b < nblocks; b += ORDER_THREADS) {
        const int g = b >> gshift;
        const unsigned s = btmp[b], x = g < full ? gtmp[g] : (unsigned)ORDER_XCDS;
        const unsigned j = hist[x * ORDER_BUCKETS + (s >> 22)] + (s & 0x3FFFFFu);
        order[place(x, j, tail)] = (unsigned)b;
    }
}

}  // namespace rt
