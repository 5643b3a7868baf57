/* The C ABI without any Python: renders the reference's default scene (scene/scene.py:99-115) at 512x512,
 * depth 2, through libmi355rt.so and writes an image (binary PPM, the device-side image layout
 * RT_FLAG_U8_HWC | RT_FLAG_U8_RGB).
 *
 *   gcc -O2 -I include examples/render_c_abi.c -L python-ray-tracer_amd -lmi355rt -lm \
 *       -Wl,-rpath,$PWD/python-ray-tracer_amd -o /tmp/render_c_abi && /tmp/render_c_abi out.ppm
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mi355rt.h"

#define CHECK(ctx, call)                                                                  \
    do {                                                                                  \
        int st_ = (call);                                                                 \
        if (st_ != RT_OK) { fprintf(stderr, "%s -> %d: %s\n", #call, st_, rt_last_error(ctx)); return 1; } \
    } while (0)

int main(int argc, char **argv)
{
    const int w = 512, h = 512, depth = 2;
    /* (7,S) rows cx,cy,cz,r,R,G,B — the six spheres of Scene.default_scene() */
    enum { S = 6, L = 3, P = 1 };
    const float sph[S][7] = { {2.2f, 0.3f, 1.0f, 1.0f, 255, 70, 70},   {0.6f, 0.7f, 0.4f, 0.4f, 70, 70, 255},
                              {0.6f, -0.8f, 0.5f, 0.5f, 255, 255, 70}, {-1.2f, 0.2f, 0.5f, 0.5f, 139, 0, 139},
                              {-1.7f, -0.5f, 0.3f, 0.3f, 70, 255, 70}, {-2.0f, 1.31f, 1.3f, 1.3f, 255, 70, 70} };
    float spheres[7 * S], lights[3 * L] = { 2.5f, 2.5f, 5.0f, -2.0f, 2.0f, 0.1f, 3.0f, 3.0f, 6.0f };   /* rows x,y,z */
    float planes[9 * P] = { 5, 0, 0, 0, 0, 1, 125, 125, 125 };
    for (int k = 0; k < S; ++k) for (int i = 0; i < 7; ++i) spheres[i * S + k] = sph[k][i];

    /* Camera(position=[-2,0,2], euler=[0,-30,0]): R = Rz(0) Ry(-30deg) Rx(0), scene/rotation.py:18-20 */
    const double th = -30.0 * M_PI / 180.0, c = cos(th), s = sin(th);
    const double origin[3] = { -2.0, 0.0, 2.0 }, rot[9] = { c, 0, -s, 0, 1, 0, s, 0, c };
    /* pixel grid closed form, scene/camera.py:18-26 (AR = int(w/h) = 1, fov 45) */
    const double px = 1.0 / tan((45.0 * M_PI / 180.0) / 2.0), dy = -2.0 / (double)(w - 1), dz = -2.0 / (double)(h - 1);

    rt_ctx *ctx = NULL;
    CHECK(NULL, rt_create(&ctx, 0));
    CHECK(ctx, rt_set_scene(ctx, spheres, S, lights, L, planes, P, 0));
    CHECK(ctx, rt_set_camera(ctx, origin, rot));
    CHECK(ctx, rt_set_raygen(ctx, w, h, px, 1.0, dy, 1.0, dz));

    rt_params prm;
    memset(&prm, 0, sizeof prm);
    prm.amb = 0.0; prm.lamb = 0.6; prm.depth = depth; prm.aa_mode = RT_AA_REFERENCE;
    for (int i = 0; i < depth; ++i) prm.refl_pow[i] = pow(0.3, i + 1);
    prm.flags = RT_FLAG_U8_HWC | RT_FLAG_U8_RGB;

    unsigned char *img = (unsigned char *)malloc((size_t)3 * w * h);
    CHECK(ctx, rt_render(ctx, &prm, 0, w, img, NULL));
    rt_kernel_info info;
    CHECK(ctx, rt_get_kernel_info(ctx, &info));

    /* A sequence of frames with ONE call (main.py:41-47 launches frame after frame): five frames into consecutive device
     * buffers, launches of three frames each, without anti-aliasing; every frame must be the same bytes. */
    enum { NF = 5 };
    const size_t fb = (size_t)3 * w * h;
    void *dseq = NULL;
    unsigned char *seq = (unsigned char *)malloc(NF * fb);
    prm.aa_mode = RT_AA_NONE;
    CHECK(ctx, rt_malloc(ctx, NF * fb, &dseq));
    for (int rep = 0; rep < 3; ++rep)     /* the first calls measure the tile costs, the last one runs whole batches */
        CHECK(ctx, rt_render_sequence(ctx, &prm, 0, w, NF, dseq, NULL, (int64_t)w, (int64_t)fb, NULL, NULL, 0, 3));
    CHECK(ctx, rt_sync(ctx));
    CHECK(ctx, rt_memcpy_d2h(ctx, seq, dseq, NF * fb));
    rt_stats st;
    CHECK(ctx, rt_get_stats(ctx, &st));
    int same = 1;
    for (int i = 1; i < NF; ++i) same = same && memcmp(seq, seq + i * fb, fb) == 0;
    printf("sequence: %d frames %s, %llu launches for %llu frames\n", NF, same ? "identical" : "DIFFER",
           (unsigned long long)st.launches, (unsigned long long)st.frames);
    free(seq);
    CHECK(ctx, rt_free(ctx, dseq));
    if (!same) return 1;
    CHECK(ctx, rt_destroy(ctx));

    const char *path = argc > 1 ? argv[1] : "render_c_abi.ppm";
    FILE *f = fopen(path, "wb");
    if (!f) { perror(path); return 1; }
    fprintf(f, "P6\n%d %d\n255\n", w, h);
    fwrite(img, 1, (size_t)3 * w * h, f);
    fclose(f);
    unsigned long sum = 0;
    for (size_t i = 0; i < (size_t)3 * w * h; ++i) sum += img[i];
    printf("wrote %s (%dx%d, byte sum %lu, kernel %d VGPRs on %d CUs)\n", path, w, h, sum, info.vgprs, info.cu_count);
    free(img);
    return 0;
}
