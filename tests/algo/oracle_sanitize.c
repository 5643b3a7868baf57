/* The CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer (sanitizers run on the CPU build
 * only).  Reads one scene dump written by tests/test_algorithms.py, renders it through orc_render (all AA modes,
 * whole frame and an unaligned slab, explicit pixel grid and closed form) and orc_render_pixels, and writes the bytes
 * back; the test compares them with the regular build.  Any sanitizer report aborts the run (exit != 0). */
#include "../../oracle/rt_oracle.c"
#include <stdio.h>

static void *slurp(FILE *f, size_t n) { void *p = malloc(n ? n : 1); if (n && fread(p, 1, n, f) != n) { fprintf(stderr, "short read\n"); exit(2); } return p; }

int main(int argc, char **argv)
{
    if (argc != 3) return 2;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    int hdr[8];   /* w h S L P depth spp seed */
    if (fread(hdr, sizeof hdr, 1, f) != 1) return 2;
    const int w = hdr[0], h = hdr[1], S = hdr[2], L = hdr[3], P = hdr[4], depth = hdr[5], spp = hdr[6];
    double *cam = slurp(f, 12 * sizeof(double));          /* origin[3], rotation[9] */
    double *rg = slurp(f, 5 * sizeof(double));            /* px y0 dy z0 dz */
    double *sc = slurp(f, (3 + depth) * sizeof(double));  /* amb lamb refl, refl_pow[depth] */
    float *sp = slurp(f, 7 * (size_t)S * 4), *li = slurp(f, 3 * (size_t)L * 4), *pl = slurp(f, 9 * (size_t)P * 4);
    fclose(f);
    orc_raygen g = { w, h, NULL, rg[0], rg[1], rg[2], rg[3], rg[4] };
    const size_t n = (size_t)3 * w * h;
    uint8_t *u8 = malloc(n); double *f64 = malloc(n * 8); float *f32 = malloc(n * 4);
    long long counters[3];
    FILE *o = fopen(argv[2], "wb");
    const int modes[3] = { 0, 1, 0x100 | spp };
    for (int m = 0; m < 3; ++m) {
        memset(u8, 0, n);
        if (orc_render(&g, cam, cam + 3, sp, S, li, L, pl, P, sc[0], sc[1], sc + 3, depth, modes[m], 0, 0, w, u8, f64, f32, counters, 2, (uint32_t)hdr[7])) return 3;
        fwrite(u8, 1, n, o); fwrite(f32, 4, n, o);
    }
    /* an unaligned slab with the typed-bias flag; only columns [x0,x1) are written */
    memset(u8, 0, n);
    if (orc_render(&g, cam, cam + 3, sp, S, li, L, pl, P, sc[0], sc[1], sc + 3, depth, 1, ORC_FLAG_TYPED_BIAS, w / 3, w - 2, u8, NULL, NULL, NULL, 1, 0)) return 3;
    fwrite(u8, 1, n, o);
    /* explicit pixel grid */
    double *grid = malloc(n * 8);
    for (int x = 0; x < w; ++x) for (int y = 0; y < h; ++y) {
        grid[(size_t)x * h + y] = rg[0]; grid[(size_t)w * h + (size_t)x * h + y] = x * rg[2] + rg[1]; grid[(size_t)2 * w * h + (size_t)x * h + y] = y * rg[4] + rg[3];
    }
    orc_raygen ge = { w, h, grid, 0, 0, 0, 0, 0 };
    memset(u8, 0, n);
    if (orc_render(&ge, cam, cam + 3, sp, S, li, L, pl, P, sc[0], sc[1], sc + 3, depth, 1, 0, 0, w, u8, NULL, NULL, NULL, 2, 0)) return 3;
    fwrite(u8, 1, n, o);
    /* sparse pixels, incl. the frame's corners */
    int32_t co[8] = { 0, 0, w - 1, h - 1, w / 2, h / 2, w - 1, 0 };
    uint8_t px[12]; double pf[12];
    if (orc_render_pixels(&g, cam, cam + 3, sp, S, li, L, pl, P, sc[0], sc[1], sc + 3, depth, 1, 0, co, 4, px, pf, 1, 0)) return 3;
    fwrite(px, 1, 12, o);
    /* rejected arguments must not touch memory */
    if (orc_render(&g, cam, cam + 3, sp, S, li, L, pl, P, sc[0], sc[1], sc + 3, depth, 0, 0, 5, w + 1, u8, NULL, NULL, NULL, 1, 0) != -1) return 4;
    co[0] = w;
    if (orc_render_pixels(&g, cam, cam + 3, sp, S, li, L, pl, P, sc[0], sc[1], sc + 3, depth, 0, 0, co, 4, px, pf, 1, 0) != -1) return 4;
    fclose(o);
    free(u8); free(f64); free(f32); free(grid); free(cam); free(rg); free(sc); free(sp); free(li); free(pl);
    printf("ok\n");
    return 0;
}
