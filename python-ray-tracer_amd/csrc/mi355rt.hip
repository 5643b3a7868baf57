// mi355rt.hip — host side of libmi355rt.so: the C ABI of include/mi355rt.h over the gfx950
// render kernel in rt_device.h.  HIP only (no torch, no CPU fallback): without a HIP device
// rt_create fails and nothing renders.
#include "../../include/mi355rt.h"
#include "rt_device.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <numeric>
#include <new>
#include <string>
#include <functional>
#include <vector>

#pragma clang fp contract(off)

namespace {

std::string g_create_error;

struct Buf {
    void *p = nullptr;
    size_t cap = 0;
};

}  // namespace

#define RT_FEEDBACK_SLOTS 8
#define RT_SCENE_RING 4
#define RT_COUNT_WORDS (4 + 2 * (RT_MAX_DEPTH + 1))   /* ray counters + per bounce {waves, alive lanes} */
#define RT_RENDER_CHUNKS 8   /* upper bound; the pipeline uses ctx->render_chunks of them */

struct rt_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr, copy_stream = nullptr;   // rt_render's chunk pipeline (created on first use)
    hipEvent_t chunk_ev[RT_RENDER_CHUNKS] = {};
    hipStream_t chunk_stream[RT_RENDER_CHUNKS] = {};
    int chunk_mode = -1;              // -1 = by destination memory type
    int cluster_min = rt::CLUSTER_MIN;            // scenes with more spheres are stored in clusters (MI355RT_CLUSTER_MINS overrides)
    int lanes_primary = 1, lanes_min_spheres = 161;   // MI355RT_LANES_MINS: the lane-owned traversal from that size on; MI355RT_LANES_PRIMARY=0: from bounce 1 on only
                                                      // (primary rays and their shadow rays wave-uniform: +2..3 % since the group level exists)
    int render_chunks = 4;            // MI355RT_CHUNKS overrides (1 = one launch, one copy)
    int order_group = -1;             // MI355RT_ORDER_GROUP: log2 of the blocks per XCD-affine dispatch group (0..6; 0 = every block on its
                                      // own; default -1 = groups of 16 tiles)
    int seq_order = -1;               // MI355RT_SEQ_ORDER: 1 / 0 = all but the last frame of a multi-frame launch in the XCD's tile order / every
                                      // frame longest-first; default -1 = tile order for the two-wave kernels (small flat scenes: headline
                                      // 0.1050 ms either way, writes 31.8 instead of 38.7 MB per frame), longest-first for the four-wave ones
                                      // (config 4: 0.734 against 0.785 ms — runs of cheap sky tiles starve the dispatcher in tile order)
    int f32_records = 1;              // MI355RT_F32_RECORDS: four-wave wave-uniform kernels keep no float64 sphere records in LDS (MODE 1)
    size_t wpw2_max_image = 4608;     // MI355RT_WPW2_MAX_IMAGE: flat scenes whose LDS image is at most this many bytes run two-wave workgroups
    int order_tiles = 1;              // MI355RT_ORDER_TILES=0: the four-wave kernels' dispatch order per block of four neighbouring tiles (A/B)
    int lanes_park = 1;               // MI355RT_LANES_PARK=0: register variants of the lane-owned kernels (A/B; with workgroups of equal-cost tiles the
                                      // parked variants win: config 5 6.88 against 7.16 ms — with neighbouring tiles they lost, 8.20 against 7.95)
    int remeasure = 24;               // MI355RT_REMEASURE: launches a dispatch order measured under an older camera is kept for before
                                      // the tile costs are measured again (a moving camera; any order renders the same frame)
    struct Slot {                     // rt_render_begin / rt_render_end: a frame in flight to host memory
        hipStream_t stream = nullptr;
        Buf u8, f32;
    } slots[RT_RENDER_SLOTS];
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // The packed scene records live in a ring of device buffers: rt_set_scene fills the NEXT one and launches carry the
    // pointer of the one that was current when they were queued, so frames in flight on any stream keep the scene they
    // were launched with (include/mi355rt.h: rt_render_begin).  A buffer comes up for reuse RT_SCENE_RING scene changes
    // later; the streams that launched with it are synchronised then (by that time they have long finished with it).
    Buf scene[RT_SCENE_RING];
    std::vector<hipStream_t> scene_readers[RT_SCENE_RING];
    int scene_cur = 0;
    Buf pixel_loc, u8, f32;
    int S = 0, P = 0, L = 0;
    int NC = 0;                   // sphere clusters (0 = flat scene)
    double scene_extent2 = 0.0;   // max squared distance of lights / sphere surfaces from the world origin
    bool have_scene = false, have_cam = false, have_grid = false, explicit_grid = false;
    double cam_o[3] = {0, 0, 0}, cam_R[9] = {0};
    int w = 0, h = 0;
    double px = 0, y0 = 0, dy = 0, z0 = 0, dz = 0;
    size_t lds_limit_set = 0;
    unsigned plane_codes = 0;         // axis codes of planes 0..3 (rt_device.h: KParams::plane_codes)
    unsigned *tile_stats = nullptr;   // caller-owned device buffer or NULL
    // Scheduler feedback: a MEASURING launch stores its tile blocks' costs; a small kernel behind it (same stream) turns
    // them into a dispatch order (rt::order_kernel).  The order lives in two buffers: launches dispatch in order[cur]
    // while a measuring launch's order kernel writes order[cur ^ 1]; the context switches to the new one when a later
    // launch (on any stream) finds the order kernel's event complete — so no stream ever waits for another stream's
    // measuring launch (round 2: the other streams' first launch in a new order waited for it, and a measuring launch
    // waited for everything the other streams had queued; with a camera that moves every frame that was a pipeline
    // bubble per measurement).  The buffer a measurement overwrites was last read by launches queued before the previous
    // switch; events recorded on their streams AT that switch (complete long before they are waited on) fence them.
    struct Feedback {
        Buf cost, gtmp, btmp, order[2]; // per-block costs of the measuring launch, order_kernel's scratch, the dispatch orders
        struct Key {                  // launch geometry the orders were built for (valid = false: none)
            bool valid = false;
            int x0 = 0, x1 = 0, h = 0, aa = 0, depth = 0, spp = 0, wpw = 0;
            bool operator==(const Key &o) const
            {
                return valid && o.valid && x0 == o.x0 && x1 == o.x1 && h == o.h && aa == o.aa && depth == o.depth &&
                       spp == o.spp && wpw == o.wpw;
            }
        } key;
        int cur = 0;
        bool have = false;            // order[cur] holds a complete order
        bool building = false;        // a measuring launch and its order kernel are in flight, writing order[cur ^ 1]
        unsigned long long epoch = 0; // ctx->epoch the costs behind order[cur] were measured under
        unsigned long long build_epoch = 0;   // ... behind the order being built
        int builds = 0;               // consecutive orders built under `epoch`
        int since = 0;                // launches that used the order under a LATER epoch (moving camera) since the last measurement
        hipEvent_t done = nullptr;    // recorded behind every order kernel
        std::vector<hipStream_t> users;                            // streams that launched in order[cur] since the last switch
        std::vector<std::pair<hipStream_t, hipEvent_t>> fence;     // recorded at the last switch: what may still read order[cur ^ 1]
        std::vector<hipEvent_t> spare;
        unsigned long long stamp = 0; // last use (the least recently used geometry is replaced)
    } fbs[RT_FEEDBACK_SLOTS];         // one per launch geometry in use: slabs, chunks and AA modes do not evict each other
    unsigned long long fb_stamp = 0;
    rt_stats stats = {};              // host-side launch counters (the ray counters live in `counts`)
    Buf counts;                       // 4 x uint64 on the device: ray counters of RT_FLAG_COUNT_RAYS launches
    std::vector<std::pair<hipStream_t, Buf>> lattice;   // per launching stream: float64 lattice samples (RT_AA_REFERENCE)
    int cu_count = 256;
    unsigned long long epoch = 1;     // bumped by every rt_set_*: scene, camera or ray grid changed
    unsigned long long scene_epoch = 1;   // bumped by rt_set_scene only
    // The float32 cull tables (rt::tables_kernel) of the last (scene, camera position, floor) combinations, PER STREAM: a
    // set is built on the stream of the launch that needs it and read only by launches of that stream, so rebuilding a
    // stream's older set is ordered behind its readers by the stream itself — no events, no cross-stream waits.  (Round 2
    // shared three sets among the streams behind events; with a camera that moves every frame the events made every
    // stream wait for the others' newest launches: frames of different streams no longer overlapped, +45 %.)  Frames
    // of a static camera on n streams build n identical sets once (a few microseconds each).
    struct Tables {
        Buf buf;
        bool valid = false;
        unsigned long long scene_epoch = 0, stamp = 0;
        double cam[3] = {0, 0, 0};
        float floor_anch = 0.0f;
        int anchors = -1;
    };
    struct StreamTables { hipStream_t stream = nullptr; Tables sets[2]; };
    std::vector<StreamTables> tables;
    unsigned long long table_stamp = 0;
    std::string err;
};

namespace {

int fail(rt_ctx *ctx, int code, const std::string &msg)
{
    if (ctx) ctx->err = msg; else g_create_error = msg;
    return code;
}

#define RT_HIP(ctx, call)                                                                         \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return fail((ctx), RT_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));    \
    } while (0)

int ensure(rt_ctx *ctx, Buf &b, size_t bytes)
{
    if (bytes <= b.cap) return RT_OK;
    if (b.p) { RT_HIP(ctx, hipFree(b.p)); b.p = nullptr; b.cap = 0; }
    RT_HIP(ctx, hipMalloc(&b.p, bytes));
    b.cap = bytes;
    return RT_OK;
}

// common.py:104-110 on float32 inputs (float32 squares/sum, sqrt of that sum rounded to float32,
// float32 divisions) — the shading normal of a plane, hoisted to scene-upload time.
void plane_normal_f32(const float n[3], float out[3])
{
    const float s = n[0] * n[0] + n[1] * n[1] + n[2] * n[2];
    const float norm = (float)std::sqrt((double)s);
    out[0] = n[0] / norm; out[1] = n[1] / norm; out[2] = n[2] / norm;
}

int check_params(rt_ctx *ctx, const rt_params *p, int x0, int x1)
{
    if (!p) return fail(ctx, RT_ERR_BAD_ARG, "params is NULL");
    if (!ctx->have_scene) return fail(ctx, RT_ERR_STATE, "rt_set_scene has not been called");
    if (!ctx->have_cam) return fail(ctx, RT_ERR_STATE, "rt_set_camera has not been called");
    if (!ctx->have_grid) return fail(ctx, RT_ERR_STATE, "rt_set_raygen / rt_set_pixel_loc has not been called");
    if (p->depth < 0 || p->depth > RT_MAX_DEPTH) return fail(ctx, RT_ERR_BAD_ARG, "depth outside 0..RT_MAX_DEPTH");
    if (p->aa_mode != RT_AA_NONE && p->aa_mode != RT_AA_REFERENCE && p->aa_mode != RT_AA_STOCHASTIC)
        return fail(ctx, RT_ERR_BAD_ARG, "unknown aa_mode");
    if (p->aa_mode == RT_AA_STOCHASTIC) {
        if (p->spp < 1 || p->spp > RT_MAX_SPP) return fail(ctx, RT_ERR_BAD_ARG, "spp outside 1..RT_MAX_SPP");
        if (ctx->explicit_grid) return fail(ctx, RT_ERR_STATE, "RT_AA_STOCHASTIC needs the closed-form ray grid (rt_set_raygen)");
    }
    if (x0 < 0 || x1 > ctx->w || x0 >= x1) return fail(ctx, RT_ERR_BAD_ARG, "column range must satisfy 0 <= x0 < x1 <= w");
    return RT_OK;
}

// instantiations with the lane-owned traversal (MODE 2: clustered scenes from lanes_min_spheres spheres on); workgroups of
// 4.  They keep no float64 sphere records in LDS (rt_device.h: sphere_hot), which leaves room for the parked state.
const void *lanes_variant(bool aa, bool lattice, bool park)
{
    if (lattice) return park ? (const void *)rt::render_kernel<false, true, 4, false, true, 2> : (const void *)rt::render_kernel<false, false, 4, false, true, 2>;
    return aa ? (park ? (const void *)rt::render_kernel<true, true, 4, false, false, 3> : (const void *)rt::render_kernel<true, false, 4, false, false, 2>)
              : (park ? (const void *)rt::render_kernel<false, true, 4, false, false, 2> : (const void *)rt::render_kernel<false, false, 4, false, false, 2>);
}

const void *lattice_variant(bool park, int wpw, bool count = false, bool norec = false)     // the plain kernel over the half-pixel lattice (RT_AA_REFERENCE)
{
    if (count) return (const void *)rt::render_kernel<false, false, 4, true, true>;
    if (norec && wpw == 4) return park ? (const void *)rt::render_kernel<false, true, 4, false, true, 1> : (const void *)rt::render_kernel<false, false, 4, false, true, 1>;
    if (wpw == 2) return park ? (const void *)rt::render_kernel<false, true, 2, false, true> : (const void *)rt::render_kernel<false, false, 2, false, true>;
    return park ? (const void *)rt::render_kernel<false, true, 4, false, true> : (const void *)rt::render_kernel<false, false, 4, false, true>;
}

const void *kernel_variant(bool aa, bool park, int wpw, bool count = false, bool norec = false)
{
    if (count)      // rt_get_stats: the register variant with workgroups of 4 carries the ray counters
        return aa ? (const void *)rt::render_kernel<true, false, 4, true> : (const void *)rt::render_kernel<false, false, 4, true>;
    if (norec && wpw == 4 && !aa)   // MODE 1: four-wave workgroups without float64 sphere records in LDS (rt_device.h: sphere_hot)
        return park ? (const void *)rt::render_kernel<false, true, 4, false, false, 1> : (const void *)rt::render_kernel<false, false, 4, false, false, 1>;
    if (wpw == 2)
        return aa ? (park ? (const void *)rt::render_kernel<true, true, 2> : (const void *)rt::render_kernel<true, false, 2>)
                  : (park ? (const void *)rt::render_kernel<false, true, 2> : (const void *)rt::render_kernel<false, false, 2>);
    return aa ? (park ? (const void *)rt::render_kernel<true, true, 4> : (const void *)rt::render_kernel<true, false, 4>)
              : (park ? (const void *)rt::render_kernel<false, true, 4> : (const void *)rt::render_kernel<false, false, 4>);
}

// The cull tables for this launch's scene / camera position / floor: reuse one of the stream's sets or rebuild its older one.
int acquire_tables(rt_ctx *ctx, const rt::KParams &k, hipStream_t stream, const float **out)
{
    rt_ctx::StreamTables *st = nullptr;
    for (auto &c : ctx->tables) if (c.stream == stream) { st = &c; break; }
    if (!st) {
        try { ctx->tables.emplace_back(); } catch (const std::bad_alloc &) { return fail(ctx, RT_ERR_ALLOC, "out of host memory"); }
        st = &ctx->tables.back();
        st->stream = stream;
    }
    rt_ctx::Tables *victim = &st->sets[0];
    for (auto &t : st->sets) {
        if (t.valid && t.scene_epoch == ctx->scene_epoch && t.anchors == k.anchors && t.floor_anch == k.floor_anch &&
            std::memcmp(t.cam, k.cam_o, sizeof t.cam) == 0) {
            t.stamp = ++ctx->table_stamp;
            *out = (const float *)t.buf.p;
            return RT_OK;
        }
        if ((!t.valid && victim->valid) || (t.valid == victim->valid && t.stamp < victim->stamp)) victim = &t;
    }
    rt_ctx::Tables &t = *victim;
    t.valid = false;
    const size_t bytes = rt::table_floats(k.S, k.NC, k.anchors, false, true, k.P) * sizeof(float);   // (with the colours)
    if (t.buf.cap < (bytes ? bytes : 16)) RT_HIP(ctx, hipStreamSynchronize(stream));       // (growing frees the old buffer)
    int rc = ensure(ctx, t.buf, bytes ? bytes : 16);
    if (rc != RT_OK) return rc;
    hipLaunchKernelGGL(rt::tables_kernel, dim3(1), dim3(rt::TABLE_THREADS), 0, stream, k, (float *)t.buf.p);
    ctx->stats.table_builds++;
    RT_HIP(ctx, hipGetLastError());
    t.scene_epoch = ctx->scene_epoch; t.anchors = k.anchors; t.floor_anch = k.floor_anch;
    std::memcpy(t.cam, k.cam_o, sizeof t.cam);
    t.valid = true;
    t.stamp = ++ctx->table_stamp;
    *out = (const float *)t.buf.p;
    return RT_OK;
}

int dispatch(rt_ctx *ctx, const rt_params *p, rt::KParams &k, bool lattice, hipStream_t stream, int nframes, int64_t frame_stride);

// The lattice buffer of a stream (RT_AA_REFERENCE with the closed-form grid renders the half-pixel lattice once into
// float64 samples, then sums nine of them per pixel): one buffer per launching stream, so that frames in flight on
// different streams keep their samples apart.
int lattice_buffer(rt_ctx *ctx, hipStream_t stream, size_t bytes, double **out)
{
    for (auto &e : ctx->lattice)
        if (e.first == stream) {
            if (e.second.cap < bytes) RT_HIP(ctx, hipStreamSynchronize(stream));
            int rc = ensure(ctx, e.second, bytes);
            *out = (double *)e.second.p;
            return rc;
        }
    ctx->lattice.emplace_back(stream, Buf{});
    int rc = ensure(ctx, ctx->lattice.back().second, bytes);
    *out = (double *)ctx->lattice.back().second.p;
    return rc;
}

// nframes > 1 (rt_render_sequence): that many frames of the current scene and camera, frame f into the outputs +
// f * frame_stride elements — one launch for all of them where the dispatch order is settled.
int launch(rt_ctx *ctx, const rt_params *p, int x0, int x1, void *d_u8, void *d_f32, int64_t plane_stride,
           hipStream_t stream, int nframes = 1, int64_t frame_stride = 0)
{
    rt::KParams k;
    std::memset(&k, 0, sizeof k);
    k.scene = (const double *)ctx->scene[ctx->scene_cur].p;
    {
        auto &rd = ctx->scene_readers[ctx->scene_cur];
        if (std::find(rd.begin(), rd.end(), stream) == rd.end()) rd.push_back(stream);
    }
    k.nframes = 1;
    k.pixel_loc = ctx->explicit_grid ? (const double *)ctx->pixel_loc.p : nullptr;
    k.out_u8 = (uint8_t *)d_u8;
    k.out_f32 = (float *)d_f32;
    k.tile_cycles = ctx->tile_stats;
    k.plane_stride = plane_stride;
    k.w = ctx->w; k.h = ctx->h; k.x0 = x0; k.x1 = x1;
    k.S = ctx->S; k.P = ctx->P; k.L = ctx->L; k.depth = p->depth; k.NC = ctx->NC; k.plane_codes = ctx->plane_codes;
    k.aa = p->aa_mode; k.u8_rgb = (p->flags & RT_FLAG_U8_RGB) ? 1 : 0; k.u8_hwc = (p->flags & RT_FLAG_U8_HWC) ? 1 : 0;
    k.spp = p->spp; k.seed = p->seed;
    k.lanes_primary = ctx->lanes_primary;
    k.tiles_y = (ctx->h + rt::TILE - 1) / rt::TILE;
    const int tiles_x = (x1 - x0 + rt::TILE - 1) / rt::TILE;
    k.ntiles = tiles_x * k.tiles_y;
    k.px = ctx->px; k.y0 = ctx->y0; k.dy = ctx->dy; k.z0 = ctx->z0; k.dz = ctx->dz;
    std::memcpy(k.cam_o, ctx->cam_o, sizeof k.cam_o);
    std::memcpy(k.cam_R, ctx->cam_R, sizeof k.cam_R);
    k.amb = p->amb; k.lamb = p->lamb;
    std::memcpy(k.refl_pow, p->refl_pow, sizeof k.refl_pow);

    // anchored cull table (camera + one anchor per light) if it fits its LDS budget, else origin-form culling only
    const size_t table = (size_t)(ctx->L + 1) * (rt::padS(ctx->S, ctx->NC) + rt::pad4(ctx->NC)) * rt::CULL_STRIDE * sizeof(float);
    k.anchors = (table <= (size_t)rt::MAX_CULL_TABLE_BYTES) ? ctx->L + 1 : 0;
    const double cam2 = ctx->cam_o[0] * ctx->cam_o[0] + ctx->cam_o[1] * ctx->cam_o[1] + ctx->cam_o[2] * ctx->cam_o[2];
    k.extent2 = (float)(1.0001 * (cam2 > ctx->scene_extent2 ? cam2 : ctx->scene_extent2));
    {   // every ray origin of the launch lies within |cam| + 999 (depth + 1) of the world origin
        const double reach = std::sqrt(cam2) + 999.0 * (p->depth + 1) + std::sqrt(ctx->scene_extent2);
        k.floor_anch = (float)(0x1p-39 * reach * reach);
    }
    {
        int rc = acquire_tables(ctx, k, stream, &k.ftab);
        if (rc != RT_OK) return rc;
    }
    // RT_AA_REFERENCE on the closed-form grid: the reference's nine taps of a pixel are the 3x3 neighbourhood of the
    // pixel's centre on the (2w-1) x (2h-1) half-pixel lattice — the tap between two pixels is 0.5 Pa + 0.5 Pb, bit
    // for bit the same from either side (IEEE addition commutes), and with a separable grid both diagonals of a cell
    // cross in the same point.  So every lattice sample is traced ONCE (4 per pixel instead of 9) by the plain kernel
    // over the lattice "frame", stored as float64 (R,G,B), and a second small kernel sums each pixel's nine samples in
    // the reference's order (kernels.py:53-65, including its G/B swap).  Explicit pixel_loc grids are not separable
    // in general and keep the nine-taps-per-pixel kernel.
    const long long LW = 2ll * ctx->w - 1, LH = 2ll * ctx->h - 1;
    if (k.aa == RT_AA_REFERENCE && !ctx->explicit_grid && LW * LH < (1ll << 31) && !(p->flags & RT_FLAG_AA_PER_PIXEL)) {
        const int li0 = std::max(0, 2 * x0 - 1), li1 = (int)std::min(LW - 1, 2ll * x1 - 1);       // lattice columns, inclusive
        double *lat = nullptr;
        int rc = lattice_buffer(ctx, stream, (size_t)(li1 - li0 + 1) * (size_t)LH * 3 * sizeof(double), &lat);
        if (rc != RT_OK) return rc;
        rt::KParams kl = k;
        kl.aa = 0; kl.lattice = 1; kl.out_u8 = nullptr; kl.out_f32 = nullptr; kl.out_f64 = lat; kl.tile_cycles = nullptr;   // (rt_set_tile_stats: pixel launches only)
        kl.w = (int)LW; kl.h = (int)LH; kl.x0 = li0; kl.x1 = li1 + 1; kl.plane_stride = 0;
        kl.tiles_y = ((int)LH + rt::TILE - 1) / rt::TILE;
        kl.ntiles = ((li1 + 1 - li0 + rt::TILE - 1) / rt::TILE) * kl.tiles_y;
        k.out_f64 = lat; k.lat_x0 = li0; k.lat_h = (int)LH;
        const long long npx = (long long)(x1 - x0) * ctx->h;
        for (int f = 0; f < nframes; ++f) {                    // the stream's one lattice buffer serves the frames in turn
            rc = dispatch(ctx, p, kl, true, stream, 1, 0);
            if (rc != RT_OK) return rc;
            rt::KParams kf = k;
            if (kf.out_u8) kf.out_u8 += (size_t)f * frame_stride;
            if (kf.out_f32) kf.out_f32 += (size_t)f * frame_stride;
            hipLaunchKernelGGL(rt::aa_resolve_kernel, dim3((unsigned)((npx + 255) / 256)), dim3(256), 0, stream, kf);
            RT_HIP(ctx, hipGetLastError());
        }
        return RT_OK;
    }
    return dispatch(ctx, p, k, false, stream, nframes, frame_stride);
}

// Chooses the kernel instantiation and the dispatch order for one launch of the render kernel over the tiles k
// describes, and launches it.
int dispatch(rt_ctx *ctx, const rt_params *p, rt::KParams &k, bool lattice, hipStream_t stream, int nframes, int64_t frame_stride)
{
    const int x0 = k.x0, x1 = k.x1;
    // Workgroup size: 2 tiles (wavefronts) for scenes whose LDS image (records + cull tables) is small, 4 otherwise
    // (every workgroup stages its own copy; rt_device.h has the measurements).
    // Kernel variant: state parked in LDS (7 waves/SIMD, no scratch) while at least 24 wavefronts per CU still
    // fit their workgroups' LDS images; otherwise the register variant (its occupancy is then LDS-bound anyway).
    const bool aa = k.aa != 0;
    const bool count = (p->flags & RT_FLAG_COUNT_RAYS) != 0;
    const size_t image = rt::lds_doubles(ctx->S, ctx->P, ctx->L) * sizeof(double) + rt::table_floats(ctx->S, ctx->NC, k.anchors) * sizeof(float);
    // Scenes with more than rt::CLUSTER_MIN spheres are clustered (rt_set_scene) and culled cluster by cluster; from 161
    // spheres on with the lane-owned traversal and its groups of clusters (rt_device.h, MODE 2; compiled for 128 VGPRs,
    // 4 waves/SIMD).  Measured against the plain wave-uniform cull of the same clusters
    // (profiles/r02_variant_thresholds.txt): 256 spheres -25 %, 196 spheres -4 % (depth 3) .. -9 % (depth 8), 169 -8 %,
    // but 144 +6 %, 100 +17 %.
    // (Round 2's bundle pre-cull, MODE 1/3, lost against these clusters at every measured size and was removed in round 3:
    // profiles/r02_variant_thresholds.txt.)
    const bool lanes = ctx->NC > 0 && ctx->S >= ctx->lanes_min_spheres && !count && !(p->flags & RT_FLAG_NO_BUNDLES);
    const int wpw = (image <= ctx->wpw2_max_image && !count && ctx->NC == 0) ? 2 : 4;   // flat scenes only (up to rt::CLUSTER_MIN spheres); measured at 1080p, depth 3 on flat scenes: 2 wins up to 25 spheres (4.1 KB), 4 from 36 (5.4 KB)
    const int wgt = 64 * wpw;
    const bool ltab = lanes && k.anchors > 0;                     // lane-owned kernels with anchored tables leave the clusters' origin-form spheres out of LDS
    // MODE 1 (wave-uniform cull, four-wave workgroups, no float64 sphere records in LDS: sphere_hot widens the float32 table
    // and a hit's colour comes from global memory) where the smaller image lets a CU hold one workgroup more — the parked
    // variant runs 7 per CU if they fit and needs 6, the register variant 5.  Config 4 (64 spheres): 6 -> 7 workgroups, -5 %;
    // 100 spheres: register variant at 5 -> parked at 6, -7 %; where the count stays (36, 49, 144 spheres) it costs 0...2 %
    // (four conversions per sphere test), and the AA kernels lose 1.5 % with it: those keep MODE 0.
    auto per_cu = [&](bool nr) {
        const size_t lp = rt::lds_bytes(ctx->S, ctx->P, ctx->L, ctx->NC, k.anchors, aa, true, wgt, ltab, lanes, nr);
        if (lp * 6 <= 160 * 1024) return (int)std::min<size_t>(7, 160 * 1024 / lp);
        return (int)std::min<size_t>(5, 160 * 1024 / rt::lds_bytes(ctx->S, ctx->P, ctx->L, ctx->NC, k.anchors, aa, false, wgt, ltab, lanes, nr));
    };
    const bool norec = !lanes && !count && !aa && wpw == 4 && ctx->f32_records && per_cu(true) > per_cu(false);
    const size_t lds_park = rt::lds_bytes(ctx->S, ctx->P, ctx->L, ctx->NC, k.anchors, aa, true, wgt, ltab, lanes, norec);
    // (lane-owned kernels are compiled for 4 waves per SIMD = 4 workgroups per CU: parked state while those still fit)
    const bool park = !count && (lanes ? (lds_park * RT_W_LANES <= 160 * 1024 && ctx->lanes_park) : lds_park * (24 / wpw) <= 160 * 1024);
    const size_t lds = park ? lds_park : rt::lds_bytes(ctx->S, ctx->P, ctx->L, ctx->NC, k.anchors, aa, false, wgt, ltab, lanes, norec);
    const void *fn = lanes ? lanes_variant(aa, lattice, park) : (lattice ? lattice_variant(park, wpw, count, norec) : kernel_variant(aa, park, wpw, count, norec));
    if (lds > 48 * 1024 && lds > ctx->lds_limit_set) {
        for (int v = 0; v < 8; ++v)
            RT_HIP(ctx, hipFuncSetAttribute(kernel_variant(v & 1, v & 2, (v & 4) ? 4 : 2), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        for (int v = 0; v < 2; ++v) {
            RT_HIP(ctx, hipFuncSetAttribute(kernel_variant(false, v & 1, 4, false, true), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            RT_HIP(ctx, hipFuncSetAttribute(lattice_variant(v & 1, 4, false, true), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        }
        for (int v = 0; v < 2; ++v)
            RT_HIP(ctx, hipFuncSetAttribute(kernel_variant(v & 1, false, 4, true), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        for (int v = 0; v < 4; ++v)
            RT_HIP(ctx, hipFuncSetAttribute(lattice_variant(v & 1, (v & 2) ? 4 : 2), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        RT_HIP(ctx, hipFuncSetAttribute(lattice_variant(false, 4, true), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        for (int v = 0; v < 6; ++v)
            RT_HIP(ctx, hipFuncSetAttribute(lanes_variant(v % 3 == 1, v % 3 == 2, v >= 3), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        ctx->lds_limit_set = lds;
    }
    if (count) {
        if (!ctx->counts.p) {
            int rc0 = ensure(ctx, ctx->counts, RT_COUNT_WORDS * sizeof(unsigned long long));
            if (rc0 != RT_OK) return rc0;
            RT_HIP(ctx, hipMemsetAsync(ctx->counts.p, 0, RT_COUNT_WORDS * sizeof(unsigned long long), stream));
        }
        k.ray_counts = (unsigned long long *)ctx->counts.p;
    }
    const unsigned grid = (unsigned)((k.ntiles + wpw - 1) / wpw);
    // Scheduler feedback (longest-first dispatch): a launch files its tile blocks by cost and dispatches in the
    // order built from the previous measured launch of the same range, depth and AA mode.
    // RT_FLAG_NO_FEEDBACK renders in plain tile order.  Any order renders every tile exactly once.
    const bool feedback = !(p->flags & RT_FLAG_NO_FEEDBACK) && grid > 1 && grid < (1u << 20);
    // XCD-affine block groups (rt::order_kernel): runs of 2^gshift consecutive blocks (neighbours in y, which share 128-byte
    // lines of the output planes) are rendered by ONE XCD, so that its L2 completes those lines before they leave for HBM;
    // inside every XCD the order is block-level longest-first.  Default: groups of 16 tiles.  MI355RT_ORDER_GROUP overrides
    // (0 = every block on its own: round 2's order, 1.39x the algorithmic write traffic on the headline frame).
    const int gshift = ctx->order_group >= 0 ? ctx->order_group : (wpw == 2 ? 3 : 2);
    // Four-wave kernels: the order's items are TILES, not blocks of four neighbouring tiles (rt_device.h: KParams::order_tiles) —
    // a workgroup's four waves are then tiles of equal cost, end together and free their slots together.
    const bool otiles = wpw >= RT_TILE_ORDER_MIN_WPW && ctx->order_tiles && feedback;
    const int wshift = wpw == 4 ? 2 : 1;
    const unsigned items = otiles ? grid * (unsigned)wpw : grid;          // entries of one permutation
    rt_ctx::Feedback::Key key;
    key.valid = true; key.x0 = x0; key.x1 = x1; key.h = k.h; key.aa = lattice ? 3 : k.aa; key.depth = k.depth;
    key.spp = (k.aa == RT_AA_STOCHASTIC) ? k.spp : 0; key.wpw = wpw + (lanes ? 32 : 0) + (otiles ? 64 : 0);
    rt_ctx::Feedback *fsel = nullptr;
    for (auto &c : ctx->fbs) if (c.key == key) { fsel = &c; break; }
    if (!fsel && feedback) {                                   // a free slot, else the least recently used geometry
        for (auto &c : ctx->fbs) if (!fsel || (!c.key.valid && fsel->key.valid) || (c.key.valid == fsel->key.valid && c.stamp < fsel->stamp)) fsel = &c;
        if (fsel->key.valid) RT_HIP(ctx, hipDeviceSynchronize());   // launches of the evicted geometry may still read its orders (rare: > 8 geometries)
        for (auto &e : fsel->fence) fsel->spare.push_back(e.second);
        fsel->fence.clear(); fsel->users.clear();
        fsel->have = fsel->building = false; fsel->builds = 0; fsel->since = 0; fsel->cur = 0;
        fsel->key = key;
    }
    static rt_ctx::Feedback none;                              // RT_FLAG_NO_FEEDBACK / one-block launches: no order, no measuring
    rt_ctx::Feedback &f = (feedback && fsel) ? *fsel : none;
    if (feedback) f.stamp = ++ctx->fb_stamp;
    if (feedback && !f.done) RT_HIP(ctx, hipEventCreateWithFlags(&f.done, hipEventDisableTiming));
    // the order being built is complete: switch to it.  Launches queued so far on the streams that used the old order may
    // still read it; an event per such stream, recorded now, is what the measurement after next waits for before it
    // overwrites that buffer.
    if (feedback && f.building) {
        const hipError_t q = hipEventQuery(f.done);
        if (q == hipSuccess) {
            f.cur ^= 1;
            f.have = true;
            f.building = false;
            f.builds = (f.build_epoch == f.epoch) ? f.builds + 1 : 1;
            f.epoch = f.build_epoch;
            for (auto &e : f.fence) f.spare.push_back(e.second);
            f.fence.clear();
            for (hipStream_t us : f.users) {
                hipEvent_t ev = nullptr;
                if (!f.spare.empty()) { ev = f.spare.back(); f.spare.pop_back(); }
                else RT_HIP(ctx, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
                RT_HIP(ctx, hipEventRecord(ev, us));
                f.fence.emplace_back(us, ev);
            }
            f.users.clear();
        } else (void)hipGetLastError();                        // hipErrorNotReady is an answer, not a failure
    }
    // An order built for this launch geometry is a valid permutation whatever has happened to scene and camera since: only
    // how well it balances the end of the launch depends on them.
    //  * Nothing that decides a tile's cost has changed since the order was rebuilt twice (once from plain tile order, once
    //    from longest-first order): the costs are the same again, so launches neither measure nor rebuild — they dispatch
    //    in that order, on any stream.
    //  * Something has changed (rt_set_* bumped the epoch — a moving camera does so with every frame): launches still
    //    dispatch in the order there is, and only every `remeasure`-th of them measures its tiles again (under that order)
    //    and rebuilds.  Round 2 measured and rebuilt with every frame of a moving camera: +11 us per frame.
    const bool same_epoch = f.epoch == ctx->epoch;
    const bool settled = feedback && f.have && (same_epoch ? f.builds >= 2 : f.since < ctx->remeasure);
    // A launch of several frames (rt_render_sequence) is ONE launch only in a settled order; until then its frames go
    // through this function one by one (a measuring launch stores the costs of one frame).
    auto frame_of = [&](int fr) { rt::KParams kf = k; if (kf.out_u8) kf.out_u8 += (size_t)fr * frame_stride; if (kf.out_f32) kf.out_f32 += (size_t)fr * frame_stride; return kf; };
    if (nframes > 1 && (long long)grid * nframes >= (1ll << 31)) {            // (a grid beyond 2^31 workgroups: frame by frame)
        for (int fr = 0; fr < nframes; ++fr) {
            rt::KParams kf = frame_of(fr);
            int rc = dispatch(ctx, p, kf, lattice, stream, 1, 0);
            if (rc != RT_OK) return rc;
        }
        return RT_OK;
    }
    if (nframes > 1 && feedback && !settled) {
        // the first frame on its own (it measures, if no measurement is in flight), then — rather than rendering more
        // frames singly while the order is being built — wait for the build (a fraction of a millisecond, twice per new
        // geometry) and hand the rest back: at most two single frames before whole batches go out in the settled order
        rt::KParams kf = frame_of(0);
        int rc = dispatch(ctx, p, kf, lattice, stream, 1, 0);
        if (rc != RT_OK) return rc;
        if (f.building) RT_HIP(ctx, hipEventSynchronize(f.done));
        rt::KParams kr = frame_of(1);
        return dispatch(ctx, p, kr, lattice, stream, nframes - 1, frame_stride);
    }
    const bool measure = feedback && !settled && !f.building;  // one measurement in flight at a time (one cost buffer)
    if (settled && !same_epoch) f.since++;
    if (f.have) {
        k.order = (const unsigned *)f.order[f.cur].p;
        if (std::find(f.users.begin(), f.users.end(), stream) == f.users.end()) f.users.push_back(stream);
    }
    if (measure) {
        const size_t words = (size_t)items * sizeof(unsigned);
        int rc = ensure(ctx, f.cost, words);
        if (rc == RT_OK) rc = ensure(ctx, f.btmp, words);
        if (rc == RT_OK) rc = ensure(ctx, f.order[0], 2 * words);     // longest-first order, then the tile-order one (order_kernel)
        if (rc == RT_OK) rc = ensure(ctx, f.order[1], 2 * words);
        if (rc == RT_OK) rc = ensure(ctx, f.gtmp, words + sizeof(unsigned));
        if (rc != RT_OK) return rc;
        for (auto &e : f.fence) {                              // (recorded at the last switch: complete long ago)
            if (e.first != stream) RT_HIP(ctx, hipStreamWaitEvent(stream, e.second, 0));
            f.spare.push_back(e.second);
        }
        f.fence.clear();
        k.cost = (unsigned *)f.cost.p;
    }
    k.nframes = nframes; k.bpf = (int)grid; k.frame_stride = frame_stride; k.order_tiles = otiles ? 1 : 0;
    rt::div_magic((unsigned)k.bpf, k.bpf_magic, k.bpf_shift);
    rt::div_magic((unsigned)k.tiles_y, k.tiles_y_magic, k.tiles_y_shift);
    k.seq_offset = (ctx->seq_order < 0 ? wpw == 2 : ctx->seq_order != 0) ? (int)items : 0;
    void *args[] = {(void *)&k};
    RT_HIP(ctx, hipLaunchKernel(fn, dim3(grid * (unsigned)nframes), dim3(wgt), args, lds, stream));
    ctx->stats.launches++;
    ctx->stats.frames += (uint64_t)nframes;
    if (settled) ctx->stats.launches_settled++;
    if (measure) ctx->stats.launches_measuring++;
    if (measure) {
        hipLaunchKernelGGL(rt::order_kernel, dim3(1), dim3(rt::ORDER_THREADS), 0, stream, (const unsigned *)f.cost.p,
                           (unsigned *)f.gtmp.p, (unsigned *)f.btmp.p, (unsigned *)f.order[f.cur ^ 1].p, (int)items, otiles ? gshift + wshift : gshift,
                           otiles ? wshift : 0, otiles ? k.ntiles : (int)grid);
        RT_HIP(ctx, hipEventRecord(f.done, stream));
        f.building = true;
        f.build_epoch = ctx->epoch;
        f.since = 0;
    }
    RT_HIP(ctx, hipGetLastError());
    return RT_OK;
}

}  // namespace

extern "C" {

int rt_abi_version(void) { return RT_ABI_VERSION; }

int rt_device_count(int *count)
{
    if (!count) return RT_ERR_BAD_ARG;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; g_create_error = std::string("hipGetDeviceCount: ") + hipGetErrorString(e); return RT_ERR_NO_DEVICE; }
    *count = n;
    return RT_OK;
}

int rt_create(rt_ctx **out, int device)
{
    if (!out) return fail(nullptr, RT_ERR_BAD_ARG, "ctx out-pointer is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(nullptr, RT_ERR_NO_DEVICE, std::string("no HIP device: ") + (e != hipSuccess ? hipGetErrorString(e) : "device count is 0"));
    if (device < 0 || device >= n) return fail(nullptr, RT_ERR_NO_DEVICE, "device index out of range");
    rt_ctx *ctx = new (std::nothrow) rt_ctx;
    if (!ctx) return fail(nullptr, RT_ERR_ALLOC, "out of host memory");
    ctx->device = device;
    if (const char *e = std::getenv("MI355RT_LANES_PRIMARY")) ctx->lanes_primary = std::atoi(e);
    if (const char *e = std::getenv("MI355RT_LANES_MINS")) ctx->lanes_min_spheres = std::atoi(e);
    if (const char *e = std::getenv("MI355RT_CLUSTER_MINS")) ctx->cluster_min = std::max(8, std::atoi(e));
    if (const char *e = std::getenv("MI355RT_CHUNK_MODE")) ctx->chunk_mode = std::atoi(e);
    if (const char *e = std::getenv("MI355RT_ORDER_GROUP")) { const int v = std::atoi(e); if (v >= 0 && v <= 6) ctx->order_group = v; }
    if (const char *e = std::getenv("MI355RT_REMEASURE")) ctx->remeasure = std::max(0, std::atoi(e));
    if (const char *e = std::getenv("MI355RT_F32_RECORDS")) ctx->f32_records = std::atoi(e) != 0;
    if (const char *e = std::getenv("MI355RT_WPW2_MAX_IMAGE")) ctx->wpw2_max_image = (size_t)std::max(0, std::atoi(e));
    if (const char *e = std::getenv("MI355RT_ORDER_TILES")) ctx->order_tiles = std::atoi(e) != 0;
    if (const char *e = std::getenv("MI355RT_LANES_PARK")) ctx->lanes_park = std::atoi(e) != 0;
    if (const char *e = std::getenv("MI355RT_SEQ_ORDER")) ctx->seq_order = std::atoi(e) != 0 ? 1 : 0;
    if (const char *e = std::getenv("MI355RT_CHUNKS")) { const int v = std::atoi(e); if (v >= 1 && v <= RT_RENDER_CHUNKS) ctx->render_chunks = v; }
    hipError_t s;
    if ((s = hipSetDevice(device)) != hipSuccess || (s = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess ||
        (s = hipEventCreate(&ctx->ev0)) != hipSuccess || (s = hipEventCreate(&ctx->ev1)) != hipSuccess) {
        std::string m = std::string("context setup: ") + hipGetErrorString(s);
        delete ctx;
        return fail(nullptr, RT_ERR_HIP, m);
    }
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) ctx->cu_count = cus;
        else (void)hipGetLastError();
    }
    *out = ctx;
    return RT_OK;
}

int rt_destroy(rt_ctx *ctx)
{
    if (!ctx) return RT_OK;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    for (Buf *b : {&ctx->pixel_loc, &ctx->u8, &ctx->f32, &ctx->counts})
        if (b->p) (void)hipFree(b->p);
    for (Buf &b : ctx->scene) if (b.p) (void)hipFree(b.p);
    for (auto &e : ctx->lattice) if (e.second.p) (void)hipFree(e.second.p);
    for (auto &f : ctx->fbs) {
        for (Buf *b : {&f.cost, &f.gtmp, &f.btmp, &f.order[0], &f.order[1]}) if (b->p) (void)hipFree(b->p);
        for (auto &r : f.fence) (void)hipEventDestroy(r.second);
        for (hipEvent_t e : f.spare) (void)hipEventDestroy(e);
        if (f.done) (void)hipEventDestroy(f.done);
    }
    for (auto &st : ctx->tables)
        for (auto &t : st.sets) if (t.buf.p) (void)hipFree(t.buf.p);
    for (hipStream_t st : {ctx->stream2, ctx->copy_stream}) if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
    for (auto &sl : ctx->slots) {
        if (sl.stream) { (void)hipStreamSynchronize(sl.stream); (void)hipStreamDestroy(sl.stream); }
        for (Buf *b : {&sl.u8, &sl.f32}) if (b->p) (void)hipFree(b->p);
    }
    for (hipEvent_t e : ctx->chunk_ev) if (e) (void)hipEventDestroy(e);
    for (hipStream_t st : ctx->chunk_stream) if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return RT_OK;
}

const char *rt_last_error(const rt_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int rt_set_scene(rt_ctx *ctx, const float *spheres, int S, const float *lights, int L, const float *planes, int P, int flags)
{
    if (!ctx) return RT_ERR_BAD_ARG;
    if (S < 0 || S > RT_MAX_SPHERES || L < 0 || L > RT_MAX_LIGHTS || P < 0 || P > RT_MAX_PLANES)
        return fail(ctx, RT_ERR_BAD_ARG, "scene size outside RT_MAX_SPHERES / RT_MAX_LIGHTS / RT_MAX_PLANES");
    if ((S && !spheres) || (L && !lights) || (P && !planes)) return fail(ctx, RT_ERR_BAD_ARG, "NULL scene array with non-zero count");
    int nclusters = 0;
    try {
        // Packed float64 records (layout: rt_device.h).  All float32 sub-expressions of the reference
        // are evaluated here, once, in float32: r*r (intersections.py:21), the plane shading normal
        // (common.py:104-110) and BIAS*N of a plane hit (trace.py:82-83).
        // Scenes with more than rt::CLUSTER_MIN spheres are stored in clusters of rt::CLUSTER spatially close
        // spheres (Morton order of the centres), each with a bounding sphere the kernel culls first.  Slot
        // order is a permutation of the caller's order; every record keeps the caller's index so that the
        // reference's tie rule (the lower index wins an exact tie, trace.py:26) is unaffected.
        std::vector<int> order(S);
        std::iota(order.begin(), order.end(), 0);
        int NC = 0;
        if (S > ctx->cluster_min) {
            // Recursive median split of the centres along the longest axis of their bounding box, the left part always a
            // whole number of clusters: every cluster but the last has exactly rt::CLUSTER spheres and is a compact block
            // of neighbours.  (Until late in round 2: Morton order cut into runs of 8 — first with every axis scaled to its own
            // span, which sorted a flat layer of spheres by radius; then with one scale; the split is tighter still.)
            // Ties are broken by the caller's index, so the order is the same on every host.
            const bool group_aligned = S >= ctx->lanes_min_spheres;
            std::function<void(int, int)> split = [&](int a, int b) {
                const int n = b - a;
                if (n <= rt::CLUSTER) return;
                double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
                for (int j = a; j < b; ++j)
                    for (int i = 0; i < 3; ++i) { const double v = spheres[i * S + order[j]]; lo[i] = std::min(lo[i], v); hi[i] = std::max(hi[i], v); }
                int ax = 0;
                for (int i = 1; i < 3; ++i) if (hi[i] - lo[i] > hi[ax] - lo[ax]) ax = i;
                const int nc = (n + rt::CLUSTER - 1) / rt::CLUSTER;
                // scenes whose kernels test GROUPS of rt::SUPER consecutive clusters first (the lane-owned traversal's): the left
                // part is a whole number of groups as well, so that every group is a subtree of this split
                const int left = (group_aligned && nc > rt::SUPER) ? rt::SUPER * ((nc / rt::SUPER + 1) / 2) : (nc + 1) / 2;
                const int mid = a + left * rt::CLUSTER;
                auto key = [&](int x) { const float v = spheres[ax * S + x]; return v == v ? v : 3.0e38f; };   // (a NaN sorts last)
                std::sort(order.begin() + a, order.begin() + b, [&](int x, int y) {
                    const float vx = key(x), vy = key(y);
                    return vx < vy || (vx == vy && x < y);
                });
                split(a, mid);
                split(mid, b);
            };
            split(0, S);
            NC = (S + rt::CLUSTER - 1) / rt::CLUSTER;
        }
        nclusters = NC;
        std::vector<double> rec((size_t)S * rt::SPH_STRIDE + (size_t)P * rt::PL_STRIDE + (size_t)L * rt::LT_STRIDE +
                                (size_t)(NC + rt::supers(NC)) * rt::CL_STRIDE + 1, 0.0);
        double *sp = rec.data();
        unsigned codes = 0;
        for (int slot = 0; slot < S; ++slot, sp += rt::SPH_STRIDE) {
            const int k = order[slot];
            sp[7] = (double)k;                                   // the caller's index of this sphere
            const float r = spheres[3 * S + k];
            const float r2 = r * r;
            sp[0] = spheres[0 * S + k]; sp[1] = spheres[1 * S + k]; sp[2] = spheres[2 * S + k]; sp[3] = (double)r2;
            sp[4] = spheres[4 * S + k]; sp[5] = spheres[5 * S + k]; sp[6] = spheres[6 * S + k];
        }
        for (int k = 0; k < P; ++k, sp += rt::PL_STRIDE) {
            for (int i = 0; i < 6; ++i) sp[i] = planes[i * P + k];
            const float nraw[3] = {planes[3 * P + k], planes[4 * P + k], planes[5 * P + k]};
            float nf[3];
            plane_normal_f32(nraw, nf);
            const double BIAS = 0.0002;
            const float bf = (float)BIAS;
            {   // axis code: the stored normal is exactly +-e_i (intersection shortcut in rt_device.h:plane_den_num)
                int axis = -1, nonzero = 0;
                for (int i = 0; i < 3; ++i) if (nraw[i] != 0.0f) { ++nonzero; axis = i; }
                sp[15] = (nonzero == 1 && (nraw[axis] == 1.0f || nraw[axis] == -1.0f)) ? (double)(axis + 1) * (double)nraw[axis] : 0.0;
                if (k < 4) codes |= (unsigned)(unsigned char)(signed char)sp[15] << (8 * k);
            }
            for (int i = 0; i < 3; ++i) {
                sp[6 + i] = (double)nf[i];
                sp[9 + i] = (flags & RT_FLAG_TYPED_BIAS) ? BIAS * (double)nf[i] : (double)(bf * nf[i]);
                sp[12 + i] = planes[(6 + i) * P + k];
            }
        }
        for (int k = 0; k < L; ++k, sp += rt::LT_STRIDE) {
            sp[0] = lights[0 * L + k]; sp[1] = lights[1 * L + k]; sp[2] = lights[2 * L + k];
        }
        // bounding sphere (float64, inflated) of the spheres in slots [j0, j1): around the centroid of the centres or the centre of
        // their bounding box, whichever gives the smaller sphere
        auto bound = [&](int j0, int j1, double *out) {
            double Cc[2][3] = {{0, 0, 0}, {0, 0, 0}}, blo[3] = {1e300, 1e300, 1e300}, bhi[3] = {-1e300, -1e300, -1e300};
            for (int j = j0; j < j1; ++j)
                for (int i = 0; i < 3; ++i) {
                    const double v = spheres[i * S + order[j]], rr = std::fabs((double)spheres[3 * S + order[j]]);
                    Cc[0][i] += v; blo[i] = std::min(blo[i], v - rr); bhi[i] = std::max(bhi[i], v + rr);
                }
            for (int i = 0; i < 3; ++i) { Cc[0][i] /= (j1 - j0); Cc[1][i] = 0.5 * (blo[i] + bhi[i]); }
            double C[3] = {0, 0, 0}, R = 1e300;
            for (int t = 0; t < 2; ++t) {
                double Rt = 0;
                for (int j = j0; j < j1; ++j) {
                    const int k = order[j];
                    const double dx = spheres[0 * S + k] - Cc[t][0], dy = spheres[1 * S + k] - Cc[t][1], dz = spheres[2 * S + k] - Cc[t][2];
                    Rt = std::max(Rt, std::sqrt(dx * dx + dy * dy + dz * dz) + std::fabs((double)spheres[3 * S + k]));
                }
                if (Rt < R || t == 0) { R = Rt; for (int i = 0; i < 3; ++i) C[i] = Cc[t][i]; }   // (NaN: keeps the centroid's)
            }
            R = R * (1.0 + 1e-6) + 1e-9;
            out[0] = C[0]; out[1] = C[1]; out[2] = C[2]; out[3] = R * R;
        };
        for (int c = 0; c < NC; ++c, sp += rt::CL_STRIDE)                   // clusters of rt::CLUSTER spheres
            bound(c * rt::CLUSTER, std::min(S, (c + 1) * rt::CLUSTER), sp);
        for (int g = 0; g < rt::supers(NC); ++g, sp += rt::CL_STRIDE)      // groups of rt::SUPER clusters
            bound(g * rt::SUPER * rt::CLUSTER, std::min(S, (g + 1) * rt::SUPER * rt::CLUSTER), sp);
        RT_HIP(ctx, hipSetDevice(ctx->device));
        const size_t bytes = rec.size() * sizeof(double);
        // the next buffer of the ring: launches in flight keep reading the buffers they were queued with.  Whatever
        // launched with THIS buffer did so RT_SCENE_RING scene changes ago; its streams are drained before it is rewritten.
        const int next = (ctx->scene_cur + 1) % RT_SCENE_RING;
        for (hipStream_t st : ctx->scene_readers[next]) RT_HIP(ctx, hipStreamSynchronize(st));
        ctx->scene_readers[next].clear();
        int rc = ensure(ctx, ctx->scene[next], bytes);
        if (rc != RT_OK) return rc;
        RT_HIP(ctx, hipMemcpyAsync(ctx->scene[next].p, rec.data(), bytes, hipMemcpyHostToDevice, ctx->stream));
        RT_HIP(ctx, hipStreamSynchronize(ctx->stream));   // rec is about to go out of scope; other streams may launch at once
        ctx->scene_cur = next;
        ctx->plane_codes = codes;
    } catch (const std::bad_alloc &) {
        return fail(ctx, RT_ERR_ALLOC, "out of host memory");
    }
    double ext2 = 0.0;
    for (int k = 0; k < S; ++k) {
        const double cx = spheres[0 * S + k], cy = spheres[1 * S + k], cz = spheres[2 * S + k], r = std::fabs((double)spheres[3 * S + k]);
        const double e = std::sqrt(cx * cx + cy * cy + cz * cz) + r;
        if (e * e > ext2) ext2 = e * e;
    }
    for (int k = 0; k < L; ++k) {
        const double x = lights[0 * L + k], y = lights[1 * L + k], z = lights[2 * L + k];
        if (x * x + y * y + z * z > ext2) ext2 = x * x + y * y + z * z;
    }
    ctx->scene_extent2 = ext2;
    ctx->S = S; ctx->P = P; ctx->L = L; ctx->NC = nclusters;
    ctx->have_scene = true;
    ctx->epoch++;
    ctx->scene_epoch++;
    return RT_OK;
}

int rt_set_camera(rt_ctx *ctx, const double origin[3], const double rotation[9])
{
    if (!ctx) return RT_ERR_BAD_ARG;
    if (!origin || !rotation) return fail(ctx, RT_ERR_BAD_ARG, "NULL camera array");
    // the same camera again (the reference's driver passes it with every launch, main.py:41-47) changes nothing a
    // tile's cost depends on: the measured dispatch order and the cull tables stay valid
    if (ctx->have_cam && std::memcmp(ctx->cam_o, origin, sizeof ctx->cam_o) == 0 &&
        std::memcmp(ctx->cam_R, rotation, sizeof ctx->cam_R) == 0) return RT_OK;
    std::memcpy(ctx->cam_o, origin, sizeof ctx->cam_o);
    std::memcpy(ctx->cam_R, rotation, sizeof ctx->cam_R);
    ctx->have_cam = true;
    ctx->epoch++;
    return RT_OK;
}

int rt_set_raygen(rt_ctx *ctx, int w, int h, double px, double y0, double dy, double z0, double dz)
{
    if (!ctx) return RT_ERR_BAD_ARG;
    if (w < 1 || h < 1 || (long long)w * h > (1ll << 31)) return fail(ctx, RT_ERR_BAD_ARG, "frame size out of range");
    {
        const double now[5] = {px, y0, dy, z0, dz}, was[5] = {ctx->px, ctx->y0, ctx->dy, ctx->z0, ctx->dz};
        if (ctx->have_grid && !ctx->explicit_grid && ctx->w == w && ctx->h == h && std::memcmp(now, was, sizeof now) == 0)
            return RT_OK;                                       // unchanged: keep the measured dispatch order
    }
    ctx->w = w; ctx->h = h; ctx->px = px; ctx->y0 = y0; ctx->dy = dy; ctx->z0 = z0; ctx->dz = dz;
    ctx->explicit_grid = false;
    ctx->have_grid = true;
    ctx->epoch++;
    return RT_OK;
}

int rt_set_pixel_loc(rt_ctx *ctx, const double *pixel_loc, int w, int h)
{
    if (!ctx) return RT_ERR_BAD_ARG;
    if (!pixel_loc) return fail(ctx, RT_ERR_BAD_ARG, "pixel_loc is NULL");
    if (w < 1 || h < 1 || (long long)w * h > (1ll << 31)) return fail(ctx, RT_ERR_BAD_ARG, "frame size out of range");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    const size_t bytes = (size_t)3 * w * h * sizeof(double);
    // one buffer, read by every launch of an explicit grid on any stream: nothing may be in flight while it is rewritten
    if (ctx->pixel_loc.p) RT_HIP(ctx, hipDeviceSynchronize());
    int rc = ensure(ctx, ctx->pixel_loc, bytes);
    if (rc != RT_OK) return rc;
    RT_HIP(ctx, hipMemcpyAsync(ctx->pixel_loc.p, pixel_loc, bytes, hipMemcpyHostToDevice, ctx->stream));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->w = w; ctx->h = h;
    ctx->explicit_grid = true;
    ctx->have_grid = true;
    ctx->epoch++;
    return RT_OK;
}

int rt_render_device(rt_ctx *ctx, const rt_params *params, int x0, int x1, void *d_u8, void *d_f32, int64_t plane_stride, void *stream)
{
    if (!ctx) return RT_ERR_BAD_ARG;
    int rc = check_params(ctx, params, x0, x1);
    if (rc != RT_OK) return rc;
    if (!d_u8 && !d_f32) return fail(ctx, RT_ERR_BAD_ARG, "both output pointers are NULL");
    if (params->flags & RT_FLAG_U8_HWC) {
        if (d_f32) return fail(ctx, RT_ERR_BAD_ARG, "RT_FLAG_U8_HWC re-uses plane_stride as the image row pitch: render the float32 buffer in a separate call");
        if (plane_stride < (int64_t)(x1 - x0)) return fail(ctx, RT_ERR_BAD_ARG, "row pitch smaller than the slab width");
    } else if (plane_stride < (int64_t)(x1 - x0) * ctx->h) return fail(ctx, RT_ERR_BAD_ARG, "plane_stride smaller than the slab");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    return launch(ctx, params, x0, x1, d_u8, d_f32, plane_stride, stream ? (hipStream_t)stream : ctx->stream);
}

int rt_render_sequence(rt_ctx *ctx, const rt_params *params, int x0, int x1, int n, void *d_u8, void *d_f32, int64_t plane_stride,
                       int64_t frame_stride, const double *cameras, void *const *streams, int n_streams, int frames_per_launch)
{
    if (!ctx) return RT_ERR_BAD_ARG;
    if (n < 0) return fail(ctx, RT_ERR_BAD_ARG, "rt_render_sequence: negative frame count");
    if (n == 0) return RT_OK;
    if (cameras) {                                            // check_params wants a camera: the first frame's
        int rc0 = rt_set_camera(ctx, cameras, cameras + 3);
        if (rc0 != RT_OK) return rc0;
    }
    int rc = check_params(ctx, params, x0, x1);
    if (rc != RT_OK) return rc;
    if (!d_u8 && !d_f32) return fail(ctx, RT_ERR_BAD_ARG, "both output pointers are NULL");
    const bool hwc = (params->flags & RT_FLAG_U8_HWC) != 0;
    if (hwc) {
        if (d_f32) return fail(ctx, RT_ERR_BAD_ARG, "RT_FLAG_U8_HWC re-uses plane_stride as the image row pitch: render the float32 buffer in a separate call");
        if (plane_stride < (int64_t)(x1 - x0)) return fail(ctx, RT_ERR_BAD_ARG, "row pitch smaller than the slab width");
        if (n > 1 && frame_stride < 3 * plane_stride * ctx->h) return fail(ctx, RT_ERR_BAD_ARG, "frame_stride smaller than one image");
    } else {
        if (plane_stride < (int64_t)(x1 - x0) * ctx->h) return fail(ctx, RT_ERR_BAD_ARG, "plane_stride smaller than the slab");
        if (n > 1 && frame_stride < 3 * plane_stride) return fail(ctx, RT_ERR_BAD_ARG, "frame_stride smaller than three planes");
    }
    if (n_streams < 0 || (n_streams > 0 && !streams)) return fail(ctx, RT_ERR_BAD_ARG, "rt_render_sequence: n_streams without a stream array");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    auto stream_of = [&](int i) { return (n_streams > 0 && streams[i % n_streams]) ? (hipStream_t)streams[i % n_streams] : ctx->stream; };
    uint8_t *u8 = (uint8_t *)d_u8;
    float *f32 = (float *)d_f32;
    if (cameras) {                                            // an animation: one launch per frame, each with its own camera
        for (int i = 0; i < n; ++i) {
            rc = rt_set_camera(ctx, cameras + 12 * (size_t)i, cameras + 12 * (size_t)i + 3);
            if (rc == RT_OK) rc = launch(ctx, params, x0, x1, u8 ? u8 + (size_t)i * frame_stride : nullptr,
                                         f32 ? f32 + (size_t)i * frame_stride : nullptr, plane_stride, stream_of(i));
            if (rc != RT_OK) return rc;
        }
        return RT_OK;
    }
    const int fpl = frames_per_launch > 0 ? frames_per_launch : 8;
    for (int i = 0, g = 0; i < n; i += fpl, ++g) {
        const int nf = std::min(fpl, n - i);
        rc = launch(ctx, params, x0, x1, u8 ? u8 + (size_t)i * frame_stride : nullptr, f32 ? f32 + (size_t)i * frame_stride : nullptr,
                    plane_stride, stream_of(g), nf, frame_stride);
        if (rc != RT_OK) return rc;
    }
    return RT_OK;
}

int rt_render(rt_ctx *ctx, const rt_params *params, int x0, int x1, uint8_t *out_u8, float *out_f32)
{
    if (!ctx) return RT_ERR_BAD_ARG;
    int rc = check_params(ctx, params, x0, x1);
    if (rc != RT_OK) return rc;
    if (!out_u8 && !out_f32) return fail(ctx, RT_ERR_BAD_ARG, "both output pointers are NULL");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    const size_t npx = (size_t)(x1 - x0) * ctx->h;
    if (out_u8 && (rc = ensure(ctx, ctx->u8, 3 * npx)) != RT_OK) return rc;
    if (out_f32 && (rc = ensure(ctx, ctx->f32, 3 * npx * sizeof(float))) != RT_OK) return rc;
    const bool hwc = (params->flags & RT_FLAG_U8_HWC) != 0;
    if (hwc && out_f32)
        return fail(ctx, RT_ERR_BAD_ARG, "RT_FLAG_U8_HWC: request the uint8 image and the float32 buffer in separate calls");
    // Large planar frames are rendered in RT_RENDER_CHUNKS column chunks, alternately on two streams (consecutive
    // launches overlap), and every chunk's planes start their way to the host (third stream, behind an
    // event) while the following chunks still render: the copy of a 1080p frame costs about as much as rendering it,
    // and this hides all of it but the last chunk's.  (main.py:41-51: launch, then copy_to_host.)
    const int tiles = (x1 - x0 + rt::TILE - 1) / rt::TILE;
    const int NCH = ctx->render_chunks;
    if (hwc || NCH < 2 || npx < (1u << 19) || tiles < 4 * NCH) {
        rc = launch(ctx, params, x0, x1, out_u8 ? ctx->u8.p : nullptr, out_f32 ? ctx->f32.p : nullptr,
                    hwc ? (int64_t)(x1 - x0) : (int64_t)npx, ctx->stream);
        if (rc != RT_OK) return rc;
        if (out_u8) RT_HIP(ctx, hipMemcpyAsync(out_u8, ctx->u8.p, 3 * npx, hipMemcpyDeviceToHost, ctx->stream));
        if (out_f32) RT_HIP(ctx, hipMemcpyAsync(out_f32, ctx->f32.p, 3 * npx * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
        RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return RT_OK;
    }
    // Page-locked destinations (rt_host_alloc): every chunk on its own stream (descending priority) with its copy queued
    // behind it in the same stream — no cross-stream event between kernel and copy (measured 0.213 ms per 1080p uint8
    // frame against 0.222).  Pageable destinations: the runtime stages those copies on the calling thread, so the
    // chunks alternate on two streams and the copies wait on a third behind events (0.23 ms; in-stream: 0.37).
    // MI355RT_CHUNK_MODE=0/1 forces one scheme.
    bool instream = false;
    {
        hipPointerAttribute_t at;
        void *probe = out_u8 ? (void *)out_u8 : (void *)out_f32;
        if (hipPointerGetAttributes(&at, probe) == hipSuccess) instream = (at.type == hipMemoryTypeHost);
        else (void)hipGetLastError();                           // plain malloc memory: "invalid value", not an error here
        if (ctx->chunk_mode >= 0) instream = ctx->chunk_mode == 1;
    }
    if (!ctx->copy_stream) {
        RT_HIP(ctx, hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking));
        RT_HIP(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
        for (auto &e : ctx->chunk_ev) RT_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        int lo = 0, hi = 0;
        RT_HIP(ctx, hipDeviceGetStreamPriorityRange(&lo, &hi));   // lo = least urgent (numerically largest)
        for (int c = 0; c < RT_RENDER_CHUNKS; ++c)
            RT_HIP(ctx, hipStreamCreateWithPriority(&ctx->chunk_stream[c], hipStreamNonBlocking, std::min(lo, hi + c)));
    }
    int cx[RT_RENDER_CHUNKS + 1];
    // the first and the last chunk are half as wide as the others: the copies start sooner, and the one copy that
    // nothing overlaps (the last chunk's) is short
    for (int c = 0; c <= NCH; ++c) {
        const int num = (c == 0) ? 0 : (c == NCH ? 2 * (NCH - 1) : 2 * c - 1);          // of 2 (NCH - 1) half-units
        cx[c] = std::min(x1, x0 + (int)((long long)tiles * num / (2 * (NCH - 1))) * rt::TILE);
    }
    unsigned *const tile_stats = ctx->tile_stats;
    const int tiles_y = (ctx->h + rt::TILE - 1) / rt::TILE;
    for (int c = 0; c < NCH; ++c) {
        hipStream_t s = instream ? ctx->chunk_stream[c] : ((c & 1) ? ctx->stream2 : ctx->stream);
        const size_t off = (size_t)(cx[c] - x0) * ctx->h, n = (size_t)(cx[c + 1] - cx[c]) * ctx->h;
        // rt_set_tile_stats: a chunk records from its own first tile column on (chunk edges are multiples of the tile size)
        if (tile_stats) ctx->tile_stats = tile_stats + (size_t)((cx[c] - x0) / rt::TILE) * tiles_y;
        rc = launch(ctx, params, cx[c], cx[c + 1], out_u8 ? (uint8_t *)ctx->u8.p + off : nullptr,
                    out_f32 ? (float *)ctx->f32.p + off : nullptr, (int64_t)npx, s);
        ctx->tile_stats = tile_stats;
        if (rc != RT_OK) return rc;
        if (instream) {
            if (out_u8) RT_HIP(ctx, hipMemcpy2DAsync(out_u8 + off, npx, (uint8_t *)ctx->u8.p + off, npx, n, 3, hipMemcpyDeviceToHost, s));
            if (out_f32) RT_HIP(ctx, hipMemcpy2DAsync(out_f32 + off, npx * sizeof(float), (float *)ctx->f32.p + off, npx * sizeof(float),
                                                      n * sizeof(float), 3, hipMemcpyDeviceToHost, s));
        } else RT_HIP(ctx, hipEventRecord(ctx->chunk_ev[c], s));
    }
    if (instream) {
        for (int c = 0; c < NCH; ++c) RT_HIP(ctx, hipStreamSynchronize(ctx->chunk_stream[c]));
        return RT_OK;
    }
    for (int c = 0; c < NCH; ++c) {                             // a chunk's three planes = one 2-D copy (3 rows, pitch = plane)
        const size_t off = (size_t)(cx[c] - x0) * ctx->h, n = (size_t)(cx[c + 1] - cx[c]) * ctx->h;
        RT_HIP(ctx, hipStreamWaitEvent(ctx->copy_stream, ctx->chunk_ev[c], 0));
        if (out_u8) RT_HIP(ctx, hipMemcpy2DAsync(out_u8 + off, npx, (uint8_t *)ctx->u8.p + off, npx, n, 3, hipMemcpyDeviceToHost, ctx->copy_stream));
        if (out_f32) RT_HIP(ctx, hipMemcpy2DAsync(out_f32 + off, npx * sizeof(float), (float *)ctx->f32.p + off, npx * sizeof(float),
                                                  n * sizeof(float), 3, hipMemcpyDeviceToHost, ctx->copy_stream));
    }
    RT_HIP(ctx, hipStreamSynchronize(ctx->copy_stream));
    return RT_OK;
}

int rt_render_begin(rt_ctx *ctx, const rt_params *params, int x0, int x1, uint8_t *out_u8, float *out_f32, int slot)
{
    if (!ctx) return RT_ERR_BAD_ARG;
    int rc = check_params(ctx, params, x0, x1);
    if (rc != RT_OK) return rc;
    if (slot < 0 || slot >= RT_RENDER_SLOTS) return fail(ctx, RT_ERR_BAD_ARG, "rt_render_begin: slot outside 0..RT_RENDER_SLOTS-1");
    if (!out_u8 && !out_f32) return fail(ctx, RT_ERR_BAD_ARG, "both output pointers are NULL");
    const bool hwc = (params->flags & RT_FLAG_U8_HWC) != 0;
    if (hwc && out_f32)
        return fail(ctx, RT_ERR_BAD_ARG, "RT_FLAG_U8_HWC: request the uint8 image and the float32 buffer in separate calls");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    rt_ctx::Slot &sl = ctx->slots[slot];
    if (!sl.stream) RT_HIP(ctx, hipStreamCreateWithFlags(&sl.stream, hipStreamNonBlocking));
    // One launch and one copy per output, queued on the slot's stream: the frame before this one in the same slot is
    // ahead of it in that stream (its staging planes are free by the time this launch writes them), the frames of the
    // other slots render and travel beside it.  The staging planes grow on the first frame of a larger size only
    // (hipFree waits for the device).
    const size_t npx = (size_t)(x1 - x0) * ctx->h;
    if (out_u8 && (rc = ensure(ctx, sl.u8, 3 * npx)) != RT_OK) return rc;
    if (out_f32 && (rc = ensure(ctx, sl.f32, 3 * npx * sizeof(float))) != RT_OK) return rc;
    rc = launch(ctx, params, x0, x1, out_u8 ? sl.u8.p : nullptr, out_f32 ? sl.f32.p : nullptr,
                hwc ? (int64_t)(x1 - x0) : (int64_t)npx, sl.stream);
    if (rc != RT_OK) return rc;
    if (out_u8) RT_HIP(ctx, hipMemcpyAsync(out_u8, sl.u8.p, 3 * npx, hipMemcpyDeviceToHost, sl.stream));
    if (out_f32) RT_HIP(ctx, hipMemcpyAsync(out_f32, sl.f32.p, 3 * npx * sizeof(float), hipMemcpyDeviceToHost, sl.stream));
    return RT_OK;
}

int rt_render_end(rt_ctx *ctx, int slot)
{
    if (!ctx) return RT_ERR_BAD_ARG;
    if (slot < 0 || slot >= RT_RENDER_SLOTS) return fail(ctx, RT_ERR_BAD_ARG, "rt_render_end: slot outside 0..RT_RENDER_SLOTS-1");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    if (ctx->slots[slot].stream) RT_HIP(ctx, hipStreamSynchronize(ctx->slots[slot].stream));
    return RT_OK;
}

int rt_host_alloc(rt_ctx *ctx, size_t bytes, void **hptr)
{
    if (!ctx) return RT_ERR_BAD_ARG;
    if (!hptr || bytes == 0) return fail(ctx, RT_ERR_BAD_ARG, "rt_host_alloc: NULL out-pointer or zero size");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    RT_HIP(ctx, hipHostMalloc(hptr, bytes, hipHostMallocDefault));
    return RT_OK;
}

int rt_host_free(rt_ctx *ctx, void *hptr)
{
    if (!hptr) return RT_OK;
    if (ctx) RT_HIP(ctx, hipSetDevice(ctx->device));           // ctx == NULL: the memory outlived its context (allowed)
    RT_HIP(ctx, hipHostFree(hptr));
    return RT_OK;
}

int rt_malloc(rt_ctx *ctx, size_t bytes, void **dptr)
{
    if (!ctx) return RT_ERR_BAD_ARG;
    if (!dptr || bytes == 0) return fail(ctx, RT_ERR_BAD_ARG, "rt_malloc: NULL out-pointer or zero size");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    RT_HIP(ctx, hipMalloc(dptr, bytes));
    return RT_OK;
}

int rt_free(rt_ctx *ctx, void *dptr)
{
    if (!ctx) return RT_ERR_BAD_ARG;
    if (!dptr) return RT_OK;
    RT_HIP(ctx, hipSetDevice(ctx->device));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    RT_HIP(ctx, hipFree(dptr));
    return RT_OK;
}

int rt_memcpy_h2d(rt_ctx *ctx, void *dst_device, const void *src_host, size_t bytes)
{
    if (!ctx) return RT_ERR_BAD_ARG;
    if (!dst_device || !src_host) return fail(ctx, RT_ERR_BAD_ARG, "rt_memcpy_h2d: NULL pointer");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    RT_HIP(ctx, hipMemcpyAsync(dst_device, src_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

int rt_memcpy_d2h(rt_ctx *ctx, void *dst_host, const void *src_device, size_t bytes)
{
    if (!ctx) return RT_ERR_BAD_ARG;
    if (!dst_host || !src_device) return fail(ctx, RT_ERR_BAD_ARG, "rt_memcpy_d2h: NULL pointer");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    RT_HIP(ctx, hipMemcpyAsync(dst_host, src_device, bytes, hipMemcpyDeviceToHost, ctx->stream));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

int rt_sync(rt_ctx *ctx)
{
    if (!ctx) return RT_ERR_BAD_ARG;
    RT_HIP(ctx, hipSetDevice(ctx->device));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

int rt_stream_create(rt_ctx *ctx, void **stream)
{
    if (!ctx) return RT_ERR_BAD_ARG;
    if (!stream) return fail(ctx, RT_ERR_BAD_ARG, "stream out-pointer is NULL");
    *stream = nullptr;
    RT_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = nullptr;
    RT_HIP(ctx, hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = (void *)s;
    return RT_OK;
}

// The context remembers streams that launched on it (owner / readers of the dispatch order, readers of the cull
// tables) so that it can fence them later: drop every reference to `stream` once its queued work is complete.
static int forget_stream(rt_ctx *ctx, hipStream_t stream)
{
    RT_HIP(ctx, hipStreamSynchronize(stream));
    for (size_t i = 0; i < ctx->lattice.size();) {
        if (ctx->lattice[i].first == stream) { if (ctx->lattice[i].second.p) (void)hipFree(ctx->lattice[i].second.p); ctx->lattice.erase(ctx->lattice.begin() + (long)i); }
        else ++i;
    }
    for (auto &f : ctx->fbs) {                                  // (its work is complete: nothing of it reads an order any more)
        for (size_t i = 0; i < f.fence.size();) {
            if (f.fence[i].first == stream) { f.spare.push_back(f.fence[i].second); f.fence.erase(f.fence.begin() + (long)i); }
            else ++i;
        }
        f.users.erase(std::remove(f.users.begin(), f.users.end(), stream), f.users.end());
    }
    for (auto &rd : ctx->scene_readers) rd.erase(std::remove(rd.begin(), rd.end(), stream), rd.end());
    for (size_t i = 0; i < ctx->tables.size();) {               // the stream's own cull-table sets go with it
        if (ctx->tables[i].stream == stream) {
            for (auto &t : ctx->tables[i].sets) if (t.buf.p) (void)hipFree(t.buf.p);
            ctx->tables.erase(ctx->tables.begin() + (long)i);
        } else ++i;
    }
    return RT_OK;
}

int rt_stream_destroy(rt_ctx *ctx, void *stream)
{
    if (!ctx) return RT_ERR_BAD_ARG;
    if (!stream) return RT_OK;
    RT_HIP(ctx, hipSetDevice(ctx->device));
    int rc = forget_stream(ctx, (hipStream_t)stream);
    if (rc != RT_OK) return rc;
    RT_HIP(ctx, hipStreamDestroy((hipStream_t)stream));
    return RT_OK;
}

int rt_stream_forget(rt_ctx *ctx, void *stream)
{
    if (!ctx) return RT_ERR_BAD_ARG;
    if (!stream) return RT_OK;
    if ((hipStream_t)stream == ctx->stream) return fail(ctx, RT_ERR_BAD_ARG, "rt_stream_forget: the context's own stream");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    return forget_stream(ctx, (hipStream_t)stream);
}

int rt_stream_sync(rt_ctx *ctx, void *stream)
{
    if (!ctx) return RT_ERR_BAD_ARG;
    RT_HIP(ctx, hipSetDevice(ctx->device));
    RT_HIP(ctx, hipStreamSynchronize(stream ? (hipStream_t)stream : ctx->stream));
    return RT_OK;
}

int rt_timer_begin(rt_ctx *ctx, void *stream)
{
    if (!ctx) return RT_ERR_BAD_ARG;
    RT_HIP(ctx, hipSetDevice(ctx->device));
    RT_HIP(ctx, hipEventRecord(ctx->ev0, stream ? (hipStream_t)stream : ctx->stream));
    return RT_OK;
}

int rt_timer_end(rt_ctx *ctx, void *stream, float *ms)
{
    if (!ctx) return RT_ERR_BAD_ARG;
    if (!ms) return fail(ctx, RT_ERR_BAD_ARG, "ms is NULL");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    RT_HIP(ctx, hipEventRecord(ctx->ev1, stream ? (hipStream_t)stream : ctx->stream));
    RT_HIP(ctx, hipEventSynchronize(ctx->ev1));
    RT_HIP(ctx, hipEventElapsedTime(ms, ctx->ev0, ctx->ev1));
    return RT_OK;
}

int rt_set_tile_stats(rt_ctx *ctx, void *d_cycles)
{
    if (!ctx) return RT_ERR_BAD_ARG;
    ctx->tile_stats = (unsigned *)d_cycles;
    return RT_OK;
}

int rt_get_stats(rt_ctx *ctx, rt_stats *out)
{
    if (!ctx) return RT_ERR_BAD_ARG;
    if (!out) return fail(ctx, RT_ERR_BAD_ARG, "stats is NULL");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    unsigned long long c[RT_COUNT_WORDS] = {0};
    if (ctx->counts.p) {                                        // counting launches may be in flight on any stream
        RT_HIP(ctx, hipDeviceSynchronize());
        RT_HIP(ctx, hipMemcpy(c, ctx->counts.p, sizeof c, hipMemcpyDeviceToHost));
    }
    *out = ctx->stats;
    out->closest_queries = c[0]; out->shadow_traced = c[1]; out->shadow_skipped = c[2]; out->hits = c[3];
    for (int b = 0; b <= RT_MAX_DEPTH; ++b) { out->bounce_waves[b] = c[4 + 2 * b]; out->bounce_lanes[b] = c[5 + 2 * b]; }
    return RT_OK;
}

int rt_reset_stats(rt_ctx *ctx)
{
    if (!ctx) return RT_ERR_BAD_ARG;
    RT_HIP(ctx, hipSetDevice(ctx->device));
    if (ctx->counts.p) {
        RT_HIP(ctx, hipDeviceSynchronize());
        RT_HIP(ctx, hipMemset(ctx->counts.p, 0, RT_COUNT_WORDS * sizeof(unsigned long long)));
    }
    ctx->stats = rt_stats{};
    return RT_OK;
}

int rt_get_kernel_info(rt_ctx *ctx, rt_kernel_info *info)
{
    if (!ctx) return RT_ERR_BAD_ARG;
    if (!info) return fail(ctx, RT_ERR_BAD_ARG, "info is NULL");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    hipFuncAttributes a;
    RT_HIP(ctx, hipFuncGetAttributes(&a, (const void *)rt::render_kernel<false, true, 2>));
    hipDeviceProp_t prop;
    RT_HIP(ctx, hipGetDeviceProperties(&prop, ctx->device));
    std::memset(info, 0, sizeof *info);
    info->vgprs = a.numRegs;
    info->lds_static = (int32_t)a.sharedSizeBytes;
    info->max_threads = a.maxThreadsPerBlock;
    info->wave_size = prop.warpSize;
    info->cu_count = prop.multiProcessorCount;
    info->clock_khz = prop.clockRate;
    return RT_OK;
}

}  // extern "C"
