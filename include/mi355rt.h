/*
 * mi355rt.h — C ABI of libmi355rt.so, the MI355X (gfx950) replacement for the per-pixel
 * ray-trace path of peter-seres/python-ray-tracer.
 *
 * The boundary being replaced is the numba kernel launch (reference paths are relative to
 * /root/reference/src):
 *
 *     render[blockspergrid, threadsperblock](pixel_loc, result, camera_origin, camera_rotation,
 *                                            spheres, lights, planes, amb, lamb, refl,
 *                                            refl_depth, aliasing)          main.py:41-42
 *     signature                                                             ray_tracing/kernels.py:7
 *     exported name  `from ray_tracing import render`                       ray_tracing/__init__.py:1
 *
 * plus the data movement around it: `cuda.to_device(...)` x7 (main.py:19-32) and
 * `result.copy_to_host()` (main.py:51).  Each entry point below names the piece of that call it
 * stands for.  Plain pointers and sizes only; no C++ or torch types cross this boundary; no
 * exception crosses it (every function returns an rt_status).
 *
 * Threading: a context is not thread-safe; use one context per host thread / per GPU.
 * Streams: rt_render_device may be called for one context on several streams (frames of a sequence in flight
 * together); the scheduler feedback inside the context is safe under that use.  rt_set_camera, rt_set_raygen and
 * rt_set_scene apply to LATER launches only: camera and ray grid travel with every launch by value, and the scene
 * lives in a ring of device buffers — a launch keeps reading the buffer that was current when it was queued, so frames
 * in flight on any stream finish with the scene they were launched with.  rt_set_pixel_loc (the explicit grid of the
 * literal drop-in) rewrites ONE device buffer: it waits for the whole device before it does.
 * The library has no CPU fallback: without a usable HIP device rt_create fails.
 */
#ifndef MI355RT_H
#define MI355RT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_ABI_VERSION 6
#define RT_MAX_DEPTH 16      /* entries of rt_params.refl_pow (reflection bounces) */
#define RT_MAX_SPHERES 1024  /* scene limits: the packed scene must fit one workgroup's LDS */
#define RT_MAX_PLANES 64
#define RT_MAX_LIGHTS 64

typedef enum rt_status {
    RT_OK = 0,
    RT_ERR_BAD_ARG = -1,   /* NULL pointer, size out of range, x-range outside the frame ... */
    RT_ERR_HIP = -2,       /* a HIP runtime call failed; see rt_last_error */
    RT_ERR_NO_DEVICE = -3, /* no HIP device / device index out of range */
    RT_ERR_STATE = -4,     /* render called before scene / camera / ray grid were set */
    RT_ERR_ALLOC = -5      /* host or device allocation failed */
} rt_status;

/* aa_mode */
#define RT_AA_NONE 0      /* aliasing=False                                        kernels.py:26 only */
#define RT_AA_REFERENCE 1 /* aliasing=True: the reference's 3x3 half-pixel taps incl. its G/B
                             accumulation order (kernels.py:29-65) on 1<=x<=w-2, 1<=y<=h-2; the
                             frame border, where the reference indexes out of bounds, gets one tap.
                             With the closed-form grid (rt_set_raygen) neighbouring pixels' shared taps are
                             traced once: see RT_FLAG_AA_PER_PIXEL */
#define RT_AA_STOCHASTIC 2 /* build-defined (the reference has no such mode; README "anti aliasing" to-do):
                             rt_params.spp samples per pixel at P + u*dy*y^ + v*dz*z^ with (u,v) in [-1/2,1/2)^2
                             from a counter hash of (x, y, sample, seed) — rt_device.h:jitter(); plain mean of the
                             samples' (R,G,B).  Needs the closed-form ray grid (rt_set_raygen).  Each sample is
                             the reference's sample() (trace.py:115-133) on that direction. */
#define RT_MAX_SPP 64

/* flags */
#define RT_FLAG_TYPED_BIAS 1 /* evaluate BIAS*N of a plane hit (trace.py:82-83) in float64 (numba
                                typing) instead of float32 (NumPy>=2 simulator promotion; default,
                                the variant pinned by the golden vectors) */
#define RT_FLAG_U8_RGB 2     /* store the uint8 frame as (R,G,B); default is the reference's
                                (R,B,G) order (common.py:60-63) */
#define RT_FLAG_U8_HWC 8     /* store the uint8 frame as an image, [y][x][3] interleaved (what viewer/image.py:7-19
                                builds on the host from the (3,w,h) frame); plane_stride is then the row pitch in
                                pixels (x1-x0 for a compact slab); uint8 output only */
#define RT_FLAG_NO_FEEDBACK 4 /* dispatch tiles in plain order; by default a launch dispatches its workgroups
                                longest-first using the per-tile cycles the previous launch of the same
                                geometry recorded (same pixels either way).  Once two consecutive launches
                                have measured the same scene, camera, grid, range, depth and AA mode the
                                order is kept and measuring stops until an rt_set_* call or another
                                range/depth/mode/stream starts it again */

#define RT_FLAG_AA_PER_PIXEL 32 /* RT_AA_REFERENCE: trace all nine taps of every pixel (what the reference does, kernels.py:29-65)
                                  instead of tracing each half-pixel lattice sample once and summing nine per pixel — the
                                  default on the closed-form grid, same bytes, 4 instead of 9 samples per pixel */
#define RT_FLAG_NO_BUNDLES 64 /* use the plain wave-uniform cull where the library would pick the lane-owned traversal (clustered scenes
                                with 161 spheres or more; rt_device.h).  Same pixels; for A/B timing.  (The name dates from round 2's
                                bundle pre-cull, which the flag also switched off; that variant was removed in round 3.) */
#define RT_FLAG_COUNT_RAYS 16 /* run the counting instantiation of the kernel (slower: registers instead of LDS-parked
                                state): adds this launch's ray counts to the context's rt_stats.  Same pixels. */

typedef struct rt_ctx rt_ctx;

/* Shader scalars of the launch (main.py:11, kernels.py:7 args amb, lamb, refl, refl_depth, aliasing).
 * refl_pow[i] must hold refl ** (i+1) as the host language evaluates it (trace.py:131). */
typedef struct rt_params {
    double amb;
    double lamb;
    double refl_pow[RT_MAX_DEPTH];
    int32_t depth;   /* refl_depth, 0..RT_MAX_DEPTH */
    int32_t aa_mode; /* RT_AA_* */
    int32_t flags;   /* RT_FLAG_* */
    int32_t spp;     /* RT_AA_STOCHASTIC: samples per pixel, 1..RT_MAX_SPP (ignored otherwise) */
    uint32_t seed;   /* RT_AA_STOCHASTIC: hash seed */
    int32_t reserved;
} rt_params;

/* Static facts about the compiled kernel (for reports). */
typedef struct rt_kernel_info {
    int32_t vgprs;          /* numRegs of the render kernel */
    int32_t sgprs;
    int32_t lds_static;     /* bytes */
    int32_t max_threads;    /* per workgroup */
    int32_t wave_size;
    int32_t cu_count;
    int32_t clock_khz;
    int32_t reserved;
} rt_kernel_info;

/* What the context has done since rt_create / rt_reset_stats (SURVEY.md §5 "metrics": rays counted).
 * The ray counters are those of launches made with RT_FLAG_COUNT_RAYS, in the reference's terms:
 *   closest_queries = calls of get_intersection for a closest hit (trace.py:53), hits = those that hit,
 *   shadow_traced + shadow_skipped = shadow queries the reference issues (trace.py:92, one per light and hit);
 *   skipped ones are those whose answer the reference discards (Lambert term <= 0, trace.py:101): the kernel does
 *   not trace them.  With RT_AA_REFERENCE the kernel traces every lattice sample once where the reference
 *   re-traces shared taps, so its counts are lower than the reference algorithm's. */
typedef struct rt_stats {
    uint64_t launches;            /* render launches */
    uint64_t frames;              /* frames those launches rendered (rt_render_sequence: several per launch) */
    uint64_t launches_measuring;  /* ... that timed their tiles and rebuilt the dispatch order */
    uint64_t launches_settled;    /* ... that dispatched in a settled (kept) order */
    uint64_t table_builds;        /* cull-table sets built (one per new scene / camera position / depth bound) */
    uint64_t closest_queries;
    uint64_t hits;
    uint64_t shadow_traced;
    uint64_t shadow_skipped;
    /* per bounce b (0 = primary rays): wavefronts that traced it and the lanes of those that carried a ray —
     * bounce_lanes[b] / (64 bounce_waves[b]) is the SIMD lane utilisation of that bounce */
    uint64_t bounce_waves[RT_MAX_DEPTH + 1];
    uint64_t bounce_lanes[RT_MAX_DEPTH + 1];
} rt_stats;

int rt_abi_version(void);

/* Number of HIP devices visible to this process. */
int rt_device_count(int *count);

/* Create / destroy a context bound to one GPU (owns a stream, scene/camera buffers, timing events).
 * Replaces numba's implicit CUDA context creation at first `cuda.to_device` (main.py:19). */
int rt_create(rt_ctx **ctx, int device);
int rt_destroy(rt_ctx *ctx);

/* Message for the last error on this context (or, with ctx == NULL, for a failed rt_create). */
const char *rt_last_error(const rt_ctx *ctx);

/* Scene arrays exactly as Scene.generate_scene() returns them (scene/scene.py:69-97):
 *   spheres float32 (7,S) C-order, rows cx,cy,cz,r,R,G,B     replaces cuda.to_device(spheres_host) main.py:19
 *   lights  float32 (3,L) rows x,y,z                          replaces cuda.to_device(light_host)   main.py:20
 *   planes  float32 (9,P) rows ox,oy,oz,nx,ny,nz,R,G,B        replaces cuda.to_device(planes_host)  main.py:21
 * Any of S, L, P may be 0 (the pointer is then ignored). */
int rt_set_scene(rt_ctx *ctx, const float *spheres, int S, const float *lights, int L,
                 const float *planes, int P, int flags);

/* camera_origin float64 (3,) and camera_rotation float64 (3,3) C-order   main.py:27-28 */
int rt_set_camera(rt_ctx *ctx, const double origin[3], const double rotation[9]);

/* The pixel grid of Camera.generate_pixel_locations() (scene/camera.py:18-26) in closed form:
 *   pixel_loc[:, x, y] = (px, x*dy + y0, y*dz + z0)      (one multiply, one add, as np.mgrid does)
 * The kernel generates primary rays from this; no (3,w,h) array is read.   replaces main.py:29 */
int rt_set_raygen(rt_ctx *ctx, int w, int h, double px, double y0, double dy, double z0, double dz);

/* Drop-in alternative: an explicit float64 (3,w,h) C-order pixel_loc array (host pointer), uploaded
 * and read by the kernel as kernels.py:19 does.                          replaces main.py:29 */
int rt_set_pixel_loc(rt_ctx *ctx, const double *pixel_loc, int w, int h);

/* One launch of the render path for frame columns x0 <= x < x1 (x1 - x0 need not be a multiple of
 * the tile size), into HOST buffers; synchronous (includes the D2H copy — main.py:41-42 + :51).
 *   out_u8  : uint8   (3, x1-x0, h) C-order, index [c, x-x0, y]; channel order per flags; or NULL
 *   out_f32 : float32 (3, x1-x0, h) the (R,G,B) handed to clip_color_vector (kernels.py:69),
 *             true channel order, unclamped; or NULL */
int rt_render(rt_ctx *ctx, const rt_params *params, int x0, int x1, uint8_t *out_u8, float *out_f32);
/* (Frames of half a megapixel and more are rendered in four column chunks whose copies to the host overlap the
 * rendering of the chunks behind them; the result is the same bytes.) */

/* Page-locked host memory for rt_render's outputs: the device-to-host copy of a frame then runs at the link's rate
 * (with pageable memory the runtime stages it).  Any host pointer is accepted by rt_render; these are an offer.
 * (The reference's `result.copy_to_host()` allocates its own pageable array, main.py:51.) */
int rt_host_alloc(rt_ctx *ctx, size_t bytes, void **hptr);
int rt_host_free(rt_ctx *ctx, void *hptr);   /* ctx may be NULL: page-locked memory may outlive the context that allocated it */

/* rt_render in two halves, for SEQUENCES of frames into host memory (an animation: main.py:41-51 in a loop).
 * rt_render_begin queues the launch and the copies of one frame on the stream of `slot` (0 <= slot < RT_RENDER_SLOTS)
 * and returns; rt_render_end(slot) returns when everything queued on that slot has arrived in the host buffers.  Frames
 * begun on different slots render and travel side by side — the copy of one frame overlaps the rendering of the next, so
 * with page-locked destinations (rt_host_alloc) a sequence runs at the slower of the two rates instead of their sum.
 * A second frame begun on a slot before its rt_render_end simply queues behind the first.  The camera, the closed-form ray
 * grid and the scene may be changed between two begins (camera and grid travel with the launch by value; the scene of a
 * launch in flight stays in its own buffer of the context's ring, see "Streams" at the top); the host buffers of a slot
 * must stay untouched until its rt_render_end.  Arguments and results as rt_render (same bytes). */
#define RT_RENDER_SLOTS 4
int rt_render_begin(rt_ctx *ctx, const rt_params *params, int x0, int x1, uint8_t *out_u8, float *out_f32, int slot);
int rt_render_end(rt_ctx *ctx, int slot);

/* The same launch into DEVICE buffers, asynchronous on `stream` (a hipStream_t; NULL = the
 * context's own stream).  Element [c, x, y] (x0 <= x < x1) is stored at
 *   base[c * plane_stride + (x - x0) * h + y]
 * so plane_stride = (x1-x0)*h gives a compact slab and, with base pointing at column x0 of a full
 * frame, plane_stride = w*h renders the slab in place.  Either pointer may be NULL. */
int rt_render_device(rt_ctx *ctx, const rt_params *params, int x0, int x1, void *d_u8, void *d_f32,
                     int64_t plane_stride, void *stream);

/* A SEQUENCE of n frames into device buffers with one call (the reference's driver launches frame after frame,
 * main.py:41-47): frame i is stored at d_u8 + i * frame_stride bytes / d_f32 + i * frame_stride floats, each frame laid
 * out as rt_render_device describes (frame_stride >= 3 * plane_stride; for RT_FLAG_U8_HWC >= 3 * row pitch * h).
 *   cameras == NULL: n frames of the context's camera.  They are rendered by launches of `frames_per_launch` frames each
 *     (0 = a default of 8; one launch renders its frames back to back in one grid, so one frame's last workgroups run
 *     beside the next frame's first and the host pays one launch for all of them), launch g on streams[g % n_streams].
 *   cameras != NULL: float64 (n, 12) — origin[3] then rotation[9] per frame (an animation).  One launch per frame, frame i
 *     on streams[i % n_streams]; afterwards the context's camera is the last one.  The dispatch order measured under an
 *     earlier camera is kept and refreshed every few frames (any order renders the same pixels).
 * streams == NULL or n_streams == 0: everything on the context's stream.  Asynchronous like rt_render_device; the same
 * pixels as n calls of it. */
int rt_render_sequence(rt_ctx *ctx, const rt_params *params, int x0, int x1, int n, void *d_u8, void *d_f32,
                       int64_t plane_stride, int64_t frame_stride, const double *cameras, void *const *streams,
                       int n_streams, int frames_per_launch);

/* Device memory owned by the caller (the DeviceNDArray that `cuda.to_device(np.zeros((3,w,h)))`
 * returns, main.py:32, and `result.copy_to_host()`, main.py:51).  The copies are ordered on the
 * context's stream and return when the bytes have arrived. */
int rt_malloc(rt_ctx *ctx, size_t bytes, void **dptr);
int rt_free(rt_ctx *ctx, void *dptr);
int rt_memcpy_h2d(rt_ctx *ctx, void *dst_device, const void *src_host, size_t bytes);
int rt_memcpy_d2h(rt_ctx *ctx, void *dst_host, const void *src_device, size_t bytes);

/* Block until everything queued on the context's stream is done (the implicit sync of
 * copy_to_host, main.py:51). */
int rt_sync(rt_ctx *ctx);

/* Extra streams for callers that do not link HIP themselves: frames of a sequence can be queued alternately on two
 * or more streams (each into its own output buffers) so that one frame's last workgroups overlap the next frame's
 * first — the reference launches one frame and waits (main.py:41-51).  The handle is a hipStream_t; any
 * hipStream_t of the context's device is equally valid wherever this header takes a `stream`. */
int rt_stream_create(rt_ctx *ctx, void **stream);
int rt_stream_destroy(rt_ctx *ctx, void *stream);
int rt_stream_sync(rt_ctx *ctx, void *stream);   /* NULL = context stream */
/* A stream the caller owns (a torch / HIP stream passed to rt_render_device) must be forgotten before its owner
 * destroys it: the context keeps the handles of streams that launched on it, to fence them when the dispatch order
 * or a cull-table set they may still read is rebuilt.  Waits for the stream's queued work, then drops the handle;
 * the stream may be used with the context again afterwards.  rt_stream_destroy does this by itself. */
int rt_stream_forget(rt_ctx *ctx, void *stream);

/* hipEvent pair on `stream` (NULL = context stream) around whatever is launched in between;
 * rt_timer_end synchronises on the second event and returns elapsed milliseconds. */
int rt_timer_begin(rt_ctx *ctx, void *stream);
int rt_timer_end(rt_ctx *ctx, void *stream, float *ms);

int rt_get_kernel_info(rt_ctx *ctx, rt_kernel_info *info);

/* rt_get_stats waits for the device when counting launches have been made (the counters live in device memory). */
int rt_get_stats(rt_ctx *ctx, rt_stats *stats);
int rt_reset_stats(rt_ctx *ctx);

/* Statistics: with a non-NULL device buffer of ceil((x1-x0)/8) * ceil(h/8) uint32, every later rt_render_device /
 * rt_render_sequence launch stores the shader-clock cycles each 8x8 tile's wavefront took, tile index =
 * tile_x * ceil(h/8) + tile_y with tile_x counted from the launch's x0 (what the reference's unused `timed` decorator,
 * viewer/image.py:22-34, gestures at).  Defined for launches of the pixel frame into device buffers: the host-buffer
 * entry points split large frames into column chunks (each chunk records from ITS first column into its own part of the
 * buffer, so the frame's tiles still land at the indices above), and RT_AA_REFERENCE on the closed-form grid traces a
 * half-pixel lattice instead of pixels and records nothing.  NULL turns it off. */
int rt_set_tile_stats(rt_ctx *ctx, void *d_cycles);

#ifdef __cplusplus
}
#endif
#endif /* MI355RT_H */
