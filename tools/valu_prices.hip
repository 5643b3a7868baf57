// tools/valu_prices.hip — what ONE wave-instruction of each class costs a SIMD of an MI355X (gfx950), measured.
//
//   hipcc -O2 --offload-arch=gfx950 -o tools/valu_prices tools/valu_prices.hip && tools/valu_prices > prices.json
//
// bench.py prices the render kernel's measured op mix (rocprofv3 SQ_INSTS_VALU_* counters) with this table to get the
// time below which the kernel cannot run at its instruction count ("issue bound"); round 2 priced every VALU
// wave-instruction at 4 cycles, which MI355X_MICROARCH.md's table ("v_fma_f32 (wave64): 2 cyc (SIMD-32); one wave
// alone: 4") contradicts for 32-bit operations with co-resident waves.  The work being priced is the reference's
// float64 arithmetic (/root/reference/src/ray_tracing/intersections.py:13-36, common.py:28-37) and the float32 cull
// and mask bookkeeping the kernel wraps around it.
//
// Method: for every class, a kernel whose loop body is 32 independent instructions of that class (8 destinations,
// each instruction depending only on the one 8 instructions earlier), run by workgroups of 256 threads (one wave per
// SIMD) with W workgroups per CU (W = 1, 4, 7 waves per SIMD: LDS-limited, every CU occupied, 256 W workgroups).
// Each wave stamps s_memtime (shader clock) and s_memrealtime (100 MHz) around its loop and records which SIMD it ran
// on (HW_ID / XCC_ID).  cycles per wave-instruction = median over waves of
//      (s_memtime ticks of the loop) / (instructions it issued x waves running on the same SIMD at the same time),
// i.e. the SIMD's time per instruction when it is kept busy by that many waves.  The loop's own scalar instructions
// (add, compare, branch: 3 per 32) issue beside the vector ones.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));

enum Cls {
    FMA_F64, MUL_F64, ADD_F64, FMA_F32, MUL_F32, ADD_F32, CMP_F64_SGPR, CMP_F32_SGPR, CMP_F32_VCC, CNDMASK_B32, CNDMASK_VCC_ONCE, CNDMASK_E64_SGPR, CNDMASK_E64_VCC, CNDMASK_E32_FDATA, CNDMASK_AFTER_VCMP, MIX_FMA64_CND_E32, MIX_FMA64_CND_E64, PAT_CC6F, PAT_CFC5F, PAT_CFFC4F, PAT_VCC6F_E32, PAT_VCC6F_E64, PAT_CCCCC3F, DIV_FMAS_F64, DIV_SCALE_F64, DIV_FIXUP_F64, FMAC_F64, LDEXP_F64, MOV_B64, MOV_B32, READLANE,
    CVT_F32_F64, CVT_F64_F32, RSQ_F64, RCP_F64, SQRT_F64, RNDNE_F64, RCP_F32, MIN3_F32, MAX_F64, AND_B32, ADD_U32, LSHL_B64, MOV_DPP,
    MIX_F64_F32, DS_READ_B128_UNIFORM, DS_READ_B128_LANE, DS_READ_B64_UNIFORM, DS_WRITE_B64_LANE,
    SALU_AND_B64, SALU_CMP_ADDC, EMPTY_LOOP, NCLS
};
static const char *cls_name[NCLS] = {
    "v_fma_f64", "v_mul_f64", "v_add_f64", "v_fma_f32", "v_mul_f32", "v_add_f32", "v_cmp_lt_f64_e64->sgpr", "v_cmp_lt_f32_e64->sgpr",
    "v_cmp_lt_f32_e32 (vcc)", "v_cndmask_b32 (vcc written by s_mov once per 32)", "v_cndmask_b32 (vcc written once, outside the loop)",
    "v_cndmask_b32_e64 (mask in an SGPR pair written outside the loop)", "v_cndmask_b32_e64 (VOP3 encoding, mask = vcc written outside the loop)", "v_cndmask_b32_e32 vcc, operands hold normal float32 values", "v_cmp_lt_f32_e32 vcc + 7 v_cndmask_b32 (per instruction of the 8)",
    "mix: 6 v_fma_f64 + 2 v_cndmask_b32_e32 vcc (per instruction of the 8)", "mix: 6 v_fma_f64 + 2 v_cndmask_b32_e64 sgpr pair (per instruction of the 8)",
    "pattern: 2 v_cndmask_b32_e32 back to back + 6 v_fma_f64 (per instruction of the 8)", "pattern: cndmask_e32, fma, cndmask_e32, 5 fma (per instruction of the 8)",
    "pattern: cndmask_e32, 2 fma, cndmask_e32, 4 fma (per instruction of the 8)", "pattern: v_cmp_gt_f64_e32 vcc, 2 v_cndmask_b32_e32, 5 fma (per instruction of the 8)",
    "pattern: v_cmp_gt_f64_e64 sgpr, 2 v_cndmask_b32_e64, 5 fma (per instruction of the 8)", "pattern: 5 v_cndmask_b32_e32 back to back + 3 fma (per instruction of the 8)",
    "v_div_fmas_f64 (reads vcc)", "v_div_scale_f64 (writes vcc)", "v_div_fixup_f64", "v_fmac_f64_e32", "v_ldexp_f64", "v_mov_b64", "v_mov_b32", "v_readlane_b32", "v_cvt_f32_f64", "v_cvt_f64_f32", "v_rsq_f64", "v_rcp_f64", "v_sqrt_f64", "v_rndne_f64",
    "v_rcp_f32", "v_min3_f32", "v_max_f64", "v_and_b32", "v_add_u32", "v_lshlrev_b64", "v_mov_b32_dpp",
    "mix: v_fma_f64 / v_fma_f32 alternating", "ds_read_b128 (wave-uniform address)", "ds_read_b128 (16 B per lane, consecutive)",
    "ds_read_b64 (wave-uniform address)", "ds_write_b64 (8 B per lane, consecutive)",
    "s_and_b64", "s_cmp_lg_u64 + s_addc_u32 (pair = 2 instructions)", "empty loop (per iteration / 32)"};
static const char *cls_counter[NCLS] = {    // the rocprofv3 SQ counter(s) that count this class
    "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_ADD_F32",
    "uncategorised", "uncategorised", "uncategorised", "uncategorised", "uncategorised", "uncategorised", "uncategorised", "uncategorised", "uncategorised", "-", "-", "-", "-", "-", "-", "-", "-", "uncategorised", "uncategorised", "uncategorised", "SQ_INSTS_VALU_FMA_F64", "uncategorised", "uncategorised", "uncategorised", "uncategorised", "SQ_INSTS_VALU_CVT", "SQ_INSTS_VALU_CVT",
    "SQ_INSTS_VALU_TRANS_F64", "SQ_INSTS_VALU_TRANS_F64", "SQ_INSTS_VALU_TRANS_F64", "uncategorised", "SQ_INSTS_VALU_TRANS_F32", "uncategorised", "uncategorised",
    "SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_INT64", "uncategorised", "-", "SQ_INSTS_LDS", "SQ_INSTS_LDS", "SQ_INSTS_LDS", "SQ_INSTS_LDS",
    "SQ_INSTS_SALU", "SQ_INSTS_SALU", "-"};

struct Rec { unsigned long long t0, t1, r0, r1; unsigned hwid, xcc; };

#define X8(S) S S S S S S S S
// 8 instructions, destination j = operand j
#define I3(OP) OP " %0, %8, %9, %0\n" OP " %1, %8, %9, %1\n" OP " %2, %8, %9, %2\n" OP " %3, %8, %9, %3\n" \
               OP " %4, %8, %9, %4\n" OP " %5, %8, %9, %5\n" OP " %6, %8, %9, %6\n" OP " %7, %8, %9, %7\n"
#define I2(OP) OP " %0, %8, %0\n" OP " %1, %8, %1\n" OP " %2, %8, %2\n" OP " %3, %8, %3\n" \
               OP " %4, %8, %4\n" OP " %5, %8, %5\n" OP " %6, %8, %6\n" OP " %7, %8, %7\n"
#define I1(OP) OP " %0, %8\n" OP " %1, %8\n" OP " %2, %8\n" OP " %3, %8\n" OP " %4, %8\n" OP " %5, %8\n" OP " %6, %8\n" OP " %7, %8\n"
#define D8 "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7])
#define F8 "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7])
#define U8 "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(u[4]), "+v"(u[5]), "+v"(u[6]), "+v"(u[7])
#define Q8 "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]), "+v"(q[6]), "+v"(q[7])
#define V8 "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]), "=v"(v[4]), "=v"(v[5]), "=v"(v[6]), "=v"(v[7])
#define SG "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55"

extern __shared__ char lds_dyn[];

template <int CLS>
__global__ __launch_bounds__(256) void price_kernel(Rec *out, double *sink, int iters, double xin, double yin)
{
    double d[8]; float f[8]; unsigned u[8]; unsigned long long q[8]; f4 v[8];
    const double x = xin, y = yin;                         // 1 + 2^-20, 2^-30: products and sums stay finite and normal
    const float xf = (float)xin, yf = (float)yin;
    for (int j = 0; j < 8; ++j) { d[j] = 1.0 + j; f[j] = 1.0f + j; u[j] = 17u * j + threadIdx.x; q[j] = 1ull + j; v[j] = f4{0, 0, 0, 0}; }
    const unsigned lane = threadIdx.x & 63u;
    // LDS addresses: every wave its own 1 KiB (lane-consecutive) / its own 16 B (uniform)
    const unsigned a_lane = (threadIdx.x >> 6) * 1024u + lane * 16u, a_uni = (threadIdx.x >> 6) * 1024u, a_lane8 = (threadIdx.x >> 6) * 1024u + lane * 8u;
    if (CLS >= DS_READ_B128_UNIFORM && CLS <= DS_WRITE_B64_LANE) { ((float *)lds_dyn)[threadIdx.x] = 1.0f; __syncthreads(); }
    const unsigned usel = 3u, ush = 0u;
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    __syncthreads();
    if constexpr (CLS == CNDMASK_VCC_ONCE || CLS == CNDMASK_E64_VCC || CLS == CNDMASK_E32_FDATA || CLS == MIX_FMA64_CND_E32 || CLS == PAT_CC6F || CLS == PAT_CFC5F || CLS == PAT_CFFC4F || CLS == PAT_CCCCC3F || CLS == DIV_FMAS_F64) asm volatile("s_mov_b64 vcc, 0x55" ::: "vcc");
    if constexpr (CLS == CNDMASK_E64_SGPR || CLS == MIX_FMA64_CND_E64) asm volatile("s_mov_b64 s[40:41], 0x55" ::: "s40", "s41");
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int i = 0; i < iters; ++i) {
        if constexpr (CLS == FMA_F64) { asm volatile(X8("") I3("v_fma_f64") I3("v_fma_f64") I3("v_fma_f64") I3("v_fma_f64") : D8 : "v"(x), "v"(y)); }
        else if constexpr (CLS == MUL_F64) { asm volatile(I2("v_mul_f64") I2("v_mul_f64") I2("v_mul_f64") I2("v_mul_f64") : D8 : "v"(x)); }
        else if constexpr (CLS == ADD_F64) { asm volatile(I2("v_add_f64") I2("v_add_f64") I2("v_add_f64") I2("v_add_f64") : D8 : "v"(y)); }
        else if constexpr (CLS == MAX_F64) { asm volatile(I2("v_max_f64") I2("v_max_f64") I2("v_max_f64") I2("v_max_f64") : D8 : "v"(y)); }
        else if constexpr (CLS == FMA_F32) { asm volatile(I3("v_fma_f32") I3("v_fma_f32") I3("v_fma_f32") I3("v_fma_f32") : F8 : "v"(xf), "v"(yf)); }
        else if constexpr (CLS == MUL_F32) { asm volatile(I2("v_mul_f32") I2("v_mul_f32") I2("v_mul_f32") I2("v_mul_f32") : F8 : "v"(xf)); }
        else if constexpr (CLS == ADD_F32) { asm volatile(I2("v_add_f32") I2("v_add_f32") I2("v_add_f32") I2("v_add_f32") : F8 : "v"(yf)); }
        else if constexpr (CLS == MIN3_F32) { asm volatile(I3("v_min3_f32") I3("v_min3_f32") I3("v_min3_f32") I3("v_min3_f32") : F8 : "v"(xf), "v"(yf)); }
        else if constexpr (CLS == CMP_F64_SGPR) {
#define C64 "v_cmp_lt_f64_e64 s[40:41], %0, %8\nv_cmp_lt_f64_e64 s[42:43], %1, %8\nv_cmp_lt_f64_e64 s[44:45], %2, %8\nv_cmp_lt_f64_e64 s[46:47], %3, %8\n" \
            "v_cmp_lt_f64_e64 s[48:49], %4, %8\nv_cmp_lt_f64_e64 s[50:51], %5, %8\nv_cmp_lt_f64_e64 s[52:53], %6, %8\nv_cmp_lt_f64_e64 s[54:55], %7, %8\n"
            asm volatile(C64 C64 C64 C64 : D8 : "v"(x) : SG);
        } else if constexpr (CLS == CMP_F32_SGPR) {
#define C32 "v_cmp_lt_f32_e64 s[40:41], %0, %8\nv_cmp_lt_f32_e64 s[42:43], %1, %8\nv_cmp_lt_f32_e64 s[44:45], %2, %8\nv_cmp_lt_f32_e64 s[46:47], %3, %8\n" \
            "v_cmp_lt_f32_e64 s[48:49], %4, %8\nv_cmp_lt_f32_e64 s[50:51], %5, %8\nv_cmp_lt_f32_e64 s[52:53], %6, %8\nv_cmp_lt_f32_e64 s[54:55], %7, %8\n"
            asm volatile(C32 C32 C32 C32 : F8 : "v"(xf) : SG);
        } else if constexpr (CLS == CNDMASK_B32) {
            asm volatile("s_mov_b64 vcc, 0x55\n" I2("v_cndmask_b32") I2("v_cndmask_b32") I2("v_cndmask_b32") I2("v_cndmask_b32") : U8 : "v"(usel) : "vcc");
        } else if constexpr (CLS == CMP_F32_VCC) {
#define CV "v_cmp_lt_f32 vcc, %0, %8\nv_cmp_lt_f32 vcc, %1, %8\nv_cmp_lt_f32 vcc, %2, %8\nv_cmp_lt_f32 vcc, %3, %8\nv_cmp_lt_f32 vcc, %4, %8\nv_cmp_lt_f32 vcc, %5, %8\nv_cmp_lt_f32 vcc, %6, %8\nv_cmp_lt_f32 vcc, %7, %8\n"
            asm volatile(CV CV CV CV : F8 : "v"(xf) : "vcc");
        } else if constexpr (CLS == CNDMASK_VCC_ONCE) {
            asm volatile(I2("v_cndmask_b32") I2("v_cndmask_b32") I2("v_cndmask_b32") I2("v_cndmask_b32") : U8 : "v"(usel) : "vcc");
        } else if constexpr (CLS == CNDMASK_E64_SGPR) {
#define CE(j) "v_cndmask_b32_e64 %" #j ", %8, %" #j ", s[40:41]\n"
#define CE8 CE(0) CE(1) CE(2) CE(3) CE(4) CE(5) CE(6) CE(7)
            asm volatile(CE8 CE8 CE8 CE8 : U8 : "v"(usel) : SG);
        } else if constexpr (CLS == CNDMASK_E64_VCC) {
#define CW(j) "v_cndmask_b32_e64 %" #j ", %8, %" #j ", vcc\n"
#define CW8 CW(0) CW(1) CW(2) CW(3) CW(4) CW(5) CW(6) CW(7)
            asm volatile(CW8 CW8 CW8 CW8 : U8 : "v"(usel) : "vcc");
        } else if constexpr (CLS == CNDMASK_E32_FDATA) {
            asm volatile(I2("v_cndmask_b32") I2("v_cndmask_b32") I2("v_cndmask_b32") I2("v_cndmask_b32") : F8 : "v"(xf) : "vcc");
        } else if constexpr (CLS == PAT_CC6F) {
            asm volatile("v_cndmask_b32 %6, %10, %6\nv_cndmask_b32 %7, %10, %7\nv_fma_f64 %0, %8, %9, %0\nv_fma_f64 %1, %8, %9, %1\nv_fma_f64 %2, %8, %9, %2\nv_fma_f64 %3, %8, %9, %3\nv_fma_f64 %4, %8, %9, %4\nv_fma_f64 %5, %8, %9, %5\n" "v_cndmask_b32 %6, %10, %6\nv_cndmask_b32 %7, %10, %7\nv_fma_f64 %0, %8, %9, %0\nv_fma_f64 %1, %8, %9, %1\nv_fma_f64 %2, %8, %9, %2\nv_fma_f64 %3, %8, %9, %3\nv_fma_f64 %4, %8, %9, %4\nv_fma_f64 %5, %8, %9, %5\n" "v_cndmask_b32 %6, %10, %6\nv_cndmask_b32 %7, %10, %7\nv_fma_f64 %0, %8, %9, %0\nv_fma_f64 %1, %8, %9, %1\nv_fma_f64 %2, %8, %9, %2\nv_fma_f64 %3, %8, %9, %3\nv_fma_f64 %4, %8, %9, %4\nv_fma_f64 %5, %8, %9, %5\n" "v_cndmask_b32 %6, %10, %6\nv_cndmask_b32 %7, %10, %7\nv_fma_f64 %0, %8, %9, %0\nv_fma_f64 %1, %8, %9, %1\nv_fma_f64 %2, %8, %9, %2\nv_fma_f64 %3, %8, %9, %3\nv_fma_f64 %4, %8, %9, %4\nv_fma_f64 %5, %8, %9, %5\n" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(u[0]), "+v"(u[1]) : "v"(x), "v"(y), "v"(usel) : "vcc");
        } else if constexpr (CLS == PAT_CFC5F) {
            asm volatile("v_cndmask_b32 %6, %10, %6\nv_fma_f64 %0, %8, %9, %0\nv_cndmask_b32 %7, %10, %7\nv_fma_f64 %1, %8, %9, %1\nv_fma_f64 %2, %8, %9, %2\nv_fma_f64 %3, %8, %9, %3\nv_fma_f64 %4, %8, %9, %4\nv_fma_f64 %5, %8, %9, %5\n" "v_cndmask_b32 %6, %10, %6\nv_fma_f64 %0, %8, %9, %0\nv_cndmask_b32 %7, %10, %7\nv_fma_f64 %1, %8, %9, %1\nv_fma_f64 %2, %8, %9, %2\nv_fma_f64 %3, %8, %9, %3\nv_fma_f64 %4, %8, %9, %4\nv_fma_f64 %5, %8, %9, %5\n" "v_cndmask_b32 %6, %10, %6\nv_fma_f64 %0, %8, %9, %0\nv_cndmask_b32 %7, %10, %7\nv_fma_f64 %1, %8, %9, %1\nv_fma_f64 %2, %8, %9, %2\nv_fma_f64 %3, %8, %9, %3\nv_fma_f64 %4, %8, %9, %4\nv_fma_f64 %5, %8, %9, %5\n" "v_cndmask_b32 %6, %10, %6\nv_fma_f64 %0, %8, %9, %0\nv_cndmask_b32 %7, %10, %7\nv_fma_f64 %1, %8, %9, %1\nv_fma_f64 %2, %8, %9, %2\nv_fma_f64 %3, %8, %9, %3\nv_fma_f64 %4, %8, %9, %4\nv_fma_f64 %5, %8, %9, %5\n" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(u[0]), "+v"(u[1]) : "v"(x), "v"(y), "v"(usel) : "vcc");
        } else if constexpr (CLS == PAT_CFFC4F) {
            asm volatile("v_cndmask_b32 %6, %10, %6\nv_fma_f64 %0, %8, %9, %0\nv_fma_f64 %1, %8, %9, %1\nv_cndmask_b32 %7, %10, %7\nv_fma_f64 %2, %8, %9, %2\nv_fma_f64 %3, %8, %9, %3\nv_fma_f64 %4, %8, %9, %4\nv_fma_f64 %5, %8, %9, %5\n" "v_cndmask_b32 %6, %10, %6\nv_fma_f64 %0, %8, %9, %0\nv_fma_f64 %1, %8, %9, %1\nv_cndmask_b32 %7, %10, %7\nv_fma_f64 %2, %8, %9, %2\nv_fma_f64 %3, %8, %9, %3\nv_fma_f64 %4, %8, %9, %4\nv_fma_f64 %5, %8, %9, %5\n" "v_cndmask_b32 %6, %10, %6\nv_fma_f64 %0, %8, %9, %0\nv_fma_f64 %1, %8, %9, %1\nv_cndmask_b32 %7, %10, %7\nv_fma_f64 %2, %8, %9, %2\nv_fma_f64 %3, %8, %9, %3\nv_fma_f64 %4, %8, %9, %4\nv_fma_f64 %5, %8, %9, %5\n" "v_cndmask_b32 %6, %10, %6\nv_fma_f64 %0, %8, %9, %0\nv_fma_f64 %1, %8, %9, %1\nv_cndmask_b32 %7, %10, %7\nv_fma_f64 %2, %8, %9, %2\nv_fma_f64 %3, %8, %9, %3\nv_fma_f64 %4, %8, %9, %4\nv_fma_f64 %5, %8, %9, %5\n" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(u[0]), "+v"(u[1]) : "v"(x), "v"(y), "v"(usel) : "vcc");
        } else if constexpr (CLS == PAT_VCC6F_E32) {
            asm volatile("v_cmp_gt_f64 vcc, %0, %1\nv_cndmask_b32 %6, %10, %6\nv_cndmask_b32 %7, %10, %7\nv_fma_f64 %0, %8, %9, %0\nv_fma_f64 %1, %8, %9, %1\nv_fma_f64 %2, %8, %9, %2\nv_fma_f64 %3, %8, %9, %3\nv_fma_f64 %4, %8, %9, %4\n" "v_cmp_gt_f64 vcc, %0, %1\nv_cndmask_b32 %6, %10, %6\nv_cndmask_b32 %7, %10, %7\nv_fma_f64 %0, %8, %9, %0\nv_fma_f64 %1, %8, %9, %1\nv_fma_f64 %2, %8, %9, %2\nv_fma_f64 %3, %8, %9, %3\nv_fma_f64 %4, %8, %9, %4\n" "v_cmp_gt_f64 vcc, %0, %1\nv_cndmask_b32 %6, %10, %6\nv_cndmask_b32 %7, %10, %7\nv_fma_f64 %0, %8, %9, %0\nv_fma_f64 %1, %8, %9, %1\nv_fma_f64 %2, %8, %9, %2\nv_fma_f64 %3, %8, %9, %3\nv_fma_f64 %4, %8, %9, %4\n" "v_cmp_gt_f64 vcc, %0, %1\nv_cndmask_b32 %6, %10, %6\nv_cndmask_b32 %7, %10, %7\nv_fma_f64 %0, %8, %9, %0\nv_fma_f64 %1, %8, %9, %1\nv_fma_f64 %2, %8, %9, %2\nv_fma_f64 %3, %8, %9, %3\nv_fma_f64 %4, %8, %9, %4\n" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(u[0]), "+v"(u[1]) : "v"(x), "v"(y), "v"(usel) : "vcc");
        } else if constexpr (CLS == PAT_VCC6F_E64) {
            asm volatile("v_cmp_gt_f64_e64 s[40:41], %0, %1\nv_cndmask_b32_e64 %6, %10, %6, s[40:41]\nv_cndmask_b32_e64 %7, %10, %7, s[40:41]\nv_fma_f64 %0, %8, %9, %0\nv_fma_f64 %1, %8, %9, %1\nv_fma_f64 %2, %8, %9, %2\nv_fma_f64 %3, %8, %9, %3\nv_fma_f64 %4, %8, %9, %4\n" "v_cmp_gt_f64_e64 s[40:41], %0, %1\nv_cndmask_b32_e64 %6, %10, %6, s[40:41]\nv_cndmask_b32_e64 %7, %10, %7, s[40:41]\nv_fma_f64 %0, %8, %9, %0\nv_fma_f64 %1, %8, %9, %1\nv_fma_f64 %2, %8, %9, %2\nv_fma_f64 %3, %8, %9, %3\nv_fma_f64 %4, %8, %9, %4\n" "v_cmp_gt_f64_e64 s[40:41], %0, %1\nv_cndmask_b32_e64 %6, %10, %6, s[40:41]\nv_cndmask_b32_e64 %7, %10, %7, s[40:41]\nv_fma_f64 %0, %8, %9, %0\nv_fma_f64 %1, %8, %9, %1\nv_fma_f64 %2, %8, %9, %2\nv_fma_f64 %3, %8, %9, %3\nv_fma_f64 %4, %8, %9, %4\n" "v_cmp_gt_f64_e64 s[40:41], %0, %1\nv_cndmask_b32_e64 %6, %10, %6, s[40:41]\nv_cndmask_b32_e64 %7, %10, %7, s[40:41]\nv_fma_f64 %0, %8, %9, %0\nv_fma_f64 %1, %8, %9, %1\nv_fma_f64 %2, %8, %9, %2\nv_fma_f64 %3, %8, %9, %3\nv_fma_f64 %4, %8, %9, %4\n" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(u[0]), "+v"(u[1]) : "v"(x), "v"(y), "v"(usel) : SG);
        } else if constexpr (CLS == PAT_CCCCC3F) {
            asm volatile("v_cndmask_b32 %6, %10, %6\nv_cndmask_b32 %7, %10, %7\nv_cndmask_b32 %6, %10, %6\nv_cndmask_b32 %7, %10, %7\nv_cndmask_b32 %6, %10, %6\nv_fma_f64 %0, %8, %9, %0\nv_fma_f64 %1, %8, %9, %1\nv_fma_f64 %2, %8, %9, %2\n" "v_cndmask_b32 %6, %10, %6\nv_cndmask_b32 %7, %10, %7\nv_cndmask_b32 %6, %10, %6\nv_cndmask_b32 %7, %10, %7\nv_cndmask_b32 %6, %10, %6\nv_fma_f64 %0, %8, %9, %0\nv_fma_f64 %1, %8, %9, %1\nv_fma_f64 %2, %8, %9, %2\n" "v_cndmask_b32 %6, %10, %6\nv_cndmask_b32 %7, %10, %7\nv_cndmask_b32 %6, %10, %6\nv_cndmask_b32 %7, %10, %7\nv_cndmask_b32 %6, %10, %6\nv_fma_f64 %0, %8, %9, %0\nv_fma_f64 %1, %8, %9, %1\nv_fma_f64 %2, %8, %9, %2\n" "v_cndmask_b32 %6, %10, %6\nv_cndmask_b32 %7, %10, %7\nv_cndmask_b32 %6, %10, %6\nv_cndmask_b32 %7, %10, %7\nv_cndmask_b32 %6, %10, %6\nv_fma_f64 %0, %8, %9, %0\nv_fma_f64 %1, %8, %9, %1\nv_fma_f64 %2, %8, %9, %2\n" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(u[0]), "+v"(u[1]) : "v"(x), "v"(y), "v"(usel) : "vcc");
        } else if constexpr (CLS == MIX_FMA64_CND_E32) {
#define M32 "v_fma_f64 %0, %8, %9, %0\nv_fma_f64 %1, %8, %9, %1\nv_fma_f64 %2, %8, %9, %2\nv_cndmask_b32 %6, %10, %6\nv_fma_f64 %3, %8, %9, %3\nv_fma_f64 %4, %8, %9, %4\nv_fma_f64 %5, %8, %9, %5\nv_cndmask_b32 %7, %10, %7\n"
            asm volatile(M32 M32 M32 M32 : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(u[0]), "+v"(u[1]) : "v"(x), "v"(y), "v"(usel) : "vcc");
        } else if constexpr (CLS == MIX_FMA64_CND_E64) {
#define M64 "v_fma_f64 %0, %8, %9, %0\nv_fma_f64 %1, %8, %9, %1\nv_fma_f64 %2, %8, %9, %2\nv_cndmask_b32_e64 %6, %10, %6, s[40:41]\nv_fma_f64 %3, %8, %9, %3\nv_fma_f64 %4, %8, %9, %4\nv_fma_f64 %5, %8, %9, %5\nv_cndmask_b32_e64 %7, %10, %7, s[40:41]\n"
            asm volatile(M64 M64 M64 M64 : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(u[0]), "+v"(u[1]) : "v"(x), "v"(y), "v"(usel) : SG);
        } else if constexpr (CLS == DIV_FMAS_F64) { asm volatile(I3("v_div_fmas_f64") I3("v_div_fmas_f64") I3("v_div_fmas_f64") I3("v_div_fmas_f64") : D8 : "v"(x), "v"(y) : "vcc"); }
        else if constexpr (CLS == DIV_SCALE_F64) {
#define DS_(j) "v_div_scale_f64 %" #j ", vcc, %8, %9, %8\n"
#define DS8 DS_(0) DS_(1) DS_(2) DS_(3) DS_(4) DS_(5) DS_(6) DS_(7)
            asm volatile(DS8 DS8 DS8 DS8 : D8 : "v"(x), "v"(y) : "vcc");
        } else if constexpr (CLS == DIV_FIXUP_F64) { asm volatile(I3("v_div_fixup_f64") I3("v_div_fixup_f64") I3("v_div_fixup_f64") I3("v_div_fixup_f64") : D8 : "v"(x), "v"(y)); }
        else if constexpr (CLS == FMAC_F64) {
#define FM(j) "v_fmac_f64_e32 %" #j ", %8, %9\n"
#define FM8 FM(0) FM(1) FM(2) FM(3) FM(4) FM(5) FM(6) FM(7)
            asm volatile(FM8 FM8 FM8 FM8 : D8 : "v"(x), "v"(y));
        } else if constexpr (CLS == LDEXP_F64) {
#define LX(j) "v_ldexp_f64 %" #j ", %" #j ", %8\n"
#define LX8 LX(0) LX(1) LX(2) LX(3) LX(4) LX(5) LX(6) LX(7)
            asm volatile(LX8 LX8 LX8 LX8 : D8 : "v"(ush));
        } else if constexpr (CLS == MOV_B64) { asm volatile(I1("v_mov_b64") I1("v_mov_b64") I1("v_mov_b64") I1("v_mov_b64") : D8 : "v"(x)); }
        else if constexpr (CLS == CNDMASK_AFTER_VCMP) {
#define CA "v_cmp_lt_f32 vcc, %9, %10\n" "v_cndmask_b32 %0, %8, %0\nv_cndmask_b32 %1, %8, %1\nv_cndmask_b32 %2, %8, %2\nv_cndmask_b32 %3, %8, %3\nv_cndmask_b32 %4, %8, %4\nv_cndmask_b32 %5, %8, %5\nv_cndmask_b32 %6, %8, %6\n"
            asm volatile(CA CA CA CA : U8 : "v"(usel), "v"(xf), "v"(yf) : "vcc");
        } else if constexpr (CLS == MOV_B32) { asm volatile(I1("v_mov_b32") I1("v_mov_b32") I1("v_mov_b32") I1("v_mov_b32") : U8 : "v"(usel)); }
        else if constexpr (CLS == MOV_DPP) {
#define DP(j) "v_mov_b32_dpp %" #j ", %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define DP8 DP(0) DP(1) DP(2) DP(3) DP(4) DP(5) DP(6) DP(7)
            asm volatile(DP8 DP8 DP8 DP8 : U8 : "v"(usel));
        } else if constexpr (CLS == READLANE) {
#define RL "v_readlane_b32 s40, %0, 5\nv_readlane_b32 s41, %1, 5\nv_readlane_b32 s42, %2, 5\nv_readlane_b32 s43, %3, 5\n" \
           "v_readlane_b32 s44, %4, 5\nv_readlane_b32 s45, %5, 5\nv_readlane_b32 s46, %6, 5\nv_readlane_b32 s47, %7, 5\n"
            asm volatile(RL RL RL RL : U8 : : SG);
        } else if constexpr (CLS == CVT_F32_F64) { asm volatile(I1("v_cvt_f32_f64") I1("v_cvt_f32_f64") I1("v_cvt_f32_f64") I1("v_cvt_f32_f64") : F8 : "v"(x)); }
        else if constexpr (CLS == CVT_F64_F32) { asm volatile(I1("v_cvt_f64_f32") I1("v_cvt_f64_f32") I1("v_cvt_f64_f32") I1("v_cvt_f64_f32") : D8 : "v"(xf)); }
        else if constexpr (CLS == RSQ_F64) { asm volatile(I1("v_rsq_f64") I1("v_rsq_f64") I1("v_rsq_f64") I1("v_rsq_f64") : D8 : "v"(x)); }
        else if constexpr (CLS == RCP_F64) { asm volatile(I1("v_rcp_f64") I1("v_rcp_f64") I1("v_rcp_f64") I1("v_rcp_f64") : D8 : "v"(x)); }
        else if constexpr (CLS == SQRT_F64) { asm volatile(I1("v_sqrt_f64") I1("v_sqrt_f64") I1("v_sqrt_f64") I1("v_sqrt_f64") : D8 : "v"(x)); }
        else if constexpr (CLS == RNDNE_F64) { asm volatile(I1("v_rndne_f64") I1("v_rndne_f64") I1("v_rndne_f64") I1("v_rndne_f64") : D8 : "v"(x)); }
        else if constexpr (CLS == RCP_F32) { asm volatile(I1("v_rcp_f32") I1("v_rcp_f32") I1("v_rcp_f32") I1("v_rcp_f32") : F8 : "v"(xf)); }
        else if constexpr (CLS == AND_B32) { asm volatile(I2("v_and_b32") I2("v_and_b32") I2("v_and_b32") I2("v_and_b32") : U8 : "v"(usel)); }
        else if constexpr (CLS == ADD_U32) { asm volatile(I2("v_add_u32") I2("v_add_u32") I2("v_add_u32") I2("v_add_u32") : U8 : "v"(usel)); }
        else if constexpr (CLS == LSHL_B64) { asm volatile(I2("v_lshlrev_b64") I2("v_lshlrev_b64") I2("v_lshlrev_b64") I2("v_lshlrev_b64") : Q8 : "v"(ush)); }
        else if constexpr (CLS == MIX_F64_F32) {
#define MX(a, b) "v_fma_f64 %" #a ", %8, %9, %" #a "\nv_fma_f32 %" #b ", %10, %11, %" #b "\n"
#define MX8 MX(0, 4) MX(1, 5) MX(2, 6) MX(3, 7)
            asm volatile(MX8 MX8 MX8 MX8 : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]) : "v"(x), "v"(y), "v"(xf), "v"(yf));
        } else if constexpr (CLS == DS_READ_B128_UNIFORM || CLS == DS_READ_B128_LANE) {
#define R128 "ds_read_b128 %0, %8\nds_read_b128 %1, %8 offset:16\nds_read_b128 %2, %8 offset:32\nds_read_b128 %3, %8 offset:48\n" \
             "ds_read_b128 %4, %8 offset:64\nds_read_b128 %5, %8 offset:80\nds_read_b128 %6, %8 offset:96\nds_read_b128 %7, %8 offset:112\ns_waitcnt lgkmcnt(0)\n"
            asm volatile(R128 R128 R128 R128 : V8 : "v"(CLS == DS_READ_B128_LANE ? a_lane : a_uni) : "memory");
        } else if constexpr (CLS == DS_READ_B64_UNIFORM) {
#define R64 "ds_read_b64 %0, %8\nds_read_b64 %1, %8 offset:8\nds_read_b64 %2, %8 offset:16\nds_read_b64 %3, %8 offset:24\n" \
            "ds_read_b64 %4, %8 offset:32\nds_read_b64 %5, %8 offset:40\nds_read_b64 %6, %8 offset:48\nds_read_b64 %7, %8 offset:56\ns_waitcnt lgkmcnt(0)\n"
            asm volatile(R64 R64 R64 R64 : "=v"(d[0]), "=v"(d[1]), "=v"(d[2]), "=v"(d[3]), "=v"(d[4]), "=v"(d[5]), "=v"(d[6]), "=v"(d[7]) : "v"(a_uni) : "memory");
        } else if constexpr (CLS == DS_WRITE_B64_LANE) {
#define W64 "ds_write_b64 %8, %0\nds_write_b64 %8, %1\nds_write_b64 %8, %2\nds_write_b64 %8, %3\nds_write_b64 %8, %4\nds_write_b64 %8, %5\nds_write_b64 %8, %6\nds_write_b64 %8, %7\ns_waitcnt lgkmcnt(0)\n"
            asm volatile(W64 W64 W64 W64 : D8 : "v"(a_lane8) : "memory");
        } else if constexpr (CLS == SALU_AND_B64) {
#define SA "s_and_b64 s[40:41], s[42:43], s[44:45]\ns_and_b64 s[46:47], s[48:49], s[50:51]\ns_and_b64 s[52:53], s[42:43], s[44:45]\ns_and_b64 s[54:55], s[48:49], s[50:51]\n"
            asm volatile(SA SA SA SA SA SA SA SA : : : SG, "scc");
        } else if constexpr (CLS == SALU_CMP_ADDC) {
#define SC "s_cmp_lg_u64 s[42:43], 0\ns_addc_u32 s40, s40, s40\ns_cmp_lg_u64 s[44:45], 0\ns_addc_u32 s41, s41, s41\n"
            asm volatile(SC SC SC SC SC SC SC SC : : : SG, "scc");
        } else { asm volatile("" ::: "memory"); }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    double acc = 0.0;
    for (int j = 0; j < 8; ++j) acc += d[j] + (double)f[j] + (double)u[j] + (double)q[j] + (double)v[j][0];
    if (acc == 123.456) sink[0] = acc;                      // keeps every destination live
    if (lane == 0) {
        Rec r{t0, t1, r0, r1, hwid, xcc};
        out[blockIdx.x * 4 + (threadIdx.x >> 6)] = r;
    }
}

typedef void (*kern_t)(Rec *, double *, int, double, double);
template <int C> struct Table { static void fill(kern_t *t) { t[C] = price_kernel<C>; Table<C + 1>::fill(t); } };
template <> struct Table<NCLS> { static void fill(kern_t *) {} };

static double median(std::vector<double> v) { std::sort(v.begin(), v.end()); return v.empty() ? 0.0 : v[v.size() / 2]; }

int main(int argc, char **argv)
{
    int iters = argc > 1 ? atoi(argv[1]) : 3000;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int CUS = prop.multiProcessorCount;
    kern_t table[NCLS];
    Table<0>::fill(table);
    const int Ws[3] = {1, 4, 7};
    Rec *d_out; double *d_sink;
    CK(hipMalloc(&d_out, sizeof(Rec) * 4 * CUS * 8));
    CK(hipMalloc(&d_sink, 64));
    printf("{\n \"device\": \"%s\", \"gcn_arch\": \"%s\", \"cus\": %d, \"iterations\": %d, \"instructions_per_iteration\": 32,\n", prop.name, prop.gcnArchName, CUS, iters);
    printf(" \"method\": \"tools/valu_prices.hip: 256-thread workgroups (one wave per SIMD), W workgroups per CU on every CU; cycles = median over waves of "
           "s_memtime ticks / (instructions x waves sharing the SIMD)\",\n");
    printf(" \"guide\": \"/opt/skills/guides/MI355X_MICROARCH.md, Per-instruction cycle constants: v_fma_f32 (wave64) 2 cyc (SIMD-32); one wave alone: 4\",\n \"classes\": [\n");
    for (int c = 0; c < NCLS; ++c) {
        printf("  {\"class\": \"%s\", \"counter\": \"%s\"", cls_name[c], cls_counter[c]);
        for (int wi = 0; wi < 3; ++wi) {
            const int W = Ws[wi];
            const size_t lds = (size_t)(160 * 1024 / (W + 1) + 1024) & ~(size_t)1023;   // W workgroups fit a CU's LDS, W + 1 do not
            CK(hipFuncSetAttribute((const void *)table[c], hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            const int grid = CUS * W;
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            for (int rep = 0; rep < 3; ++rep) {                               // the last repetition is the one read
                CK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(table[c], dim3(grid), dim3(256), lds, 0, d_out, d_sink, iters, 1.0 + 0x1p-20, 0x1p-30);
                CK(hipEventRecord(e1, 0));
                CK(hipGetLastError());
                CK(hipDeviceSynchronize());
            }
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            std::vector<Rec> recs((size_t)grid * 4);
            CK(hipMemcpy(recs.data(), d_out, sizeof(Rec) * recs.size(), hipMemcpyDeviceToHost));
            // waves per SIMD: those whose loops overlap this wave's loop for more than half of it, on the same SIMD
            std::map<unsigned, std::vector<int>> bysimd;
            for (size_t i = 0; i < recs.size(); ++i) bysimd[(recs[i].xcc << 16) | (recs[i].hwid & 0xFF30u)].push_back((int)i);
            std::vector<double> cyc, clk, share;
            const double ninstr = (double)iters * 32.0;
            for (auto &kv : bysimd)
                for (int i : kv.second) {
                    const Rec &a = recs[i];
                    int n = 0;
                    for (int j : kv.second) {
                        const Rec &b = recs[j];
                        const long long lo = (long long)std::max(a.t0, b.t0), hi = (long long)std::min(a.t1, b.t1);
                        if (hi - lo > (long long)(a.t1 - a.t0) / 2) ++n;
                    }
                    share.push_back(n);
                    cyc.push_back((double)(a.t1 - a.t0) / (ninstr * n));
                    if (a.r1 > a.r0) clk.push_back((double)(a.t1 - a.t0) / (double)(a.r1 - a.r0) * 0.1);   // GHz
                }
            // second estimate, per SIMD: its busy span (first start to last end of the waves it ran) / all instructions they issued
            std::vector<double> span;
            size_t wmin = 1u << 30, wmax = 0;
            for (auto &kv : bysimd) {
                unsigned long long lo = ~0ull, hi = 0;
                for (int i : kv.second) { lo = std::min(lo, recs[i].t0); hi = std::max(hi, recs[i].t1); }
                span.push_back((double)(hi - lo) / (ninstr * kv.second.size()));
                wmin = std::min(wmin, kv.second.size()); wmax = std::max(wmax, kv.second.size());
            }
            printf(", \"w%d\": {\"cycles\": %.3f, \"cycles_simd_span\": %.3f, \"concurrent_waves_median\": %.0f, \"waves_per_simd_min_max\": [%zu, %zu], \"simds\": %zu, \"clock_ghz\": %.3f, \"launch_ms\": %.4f}",
                   W, median(cyc), median(span), median(share), wmin, wmax, bysimd.size(), median(clk), ms);
            CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
        }
        printf("}%s\n", c + 1 < NCLS ? "," : "");
        fflush(stdout);
    }
    printf(" ]\n}\n");
    CK(hipFree(d_out)); CK(hipFree(d_sink));
    return 0;
}
