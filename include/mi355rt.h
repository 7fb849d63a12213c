/*
 * mi355rt.h -- C ABI of the MI355X-native replacement for the per-ray
 * intersection + shading hot path of DarkAce65/raytracer.
 *
 * The reference has no FFI layer.  The seam this library replaces is the body
 * of RaytracingScene::raytrace_to_image (src/render/raytracing_scene.rs:703-776),
 * i.e. the `indexes.par_iter().for_each(process_pixel)` loop at :734-744 whose
 * unit of work is `screen_raycast(x, y)` (:598-622).  A Rust host would flatten
 * its private RaytracingScene fields (:45-51) into an `rt_scene_desc`, call
 * `rt_scene_create` once after `Scene::build_raytracing_scene`
 * (src/render/scene.rs:76-97) and `rt_render` once per frame instead of
 * `screen_raycast` once per pixel.  See INTEGRATION.md for the binding sketch.
 *
 * Conventions
 *   - every entry point returns 0 on success or a negative RT_ERR_* code and
 *     never unwinds or aborts across the boundary (the reference panics);
 *     `rt_last_error()` returns a thread-local message for the last failure.
 *   - plain pointers and sizes only.  The caller keeps ownership of every
 *     array in `rt_scene_desc`; the library copies what it needs.
 *   - matrices are row-major.  `world` / `inv_world` are 3x4 affine blocks
 *     (12 doubles, rows of [r0 r1 r2 t]).
 *   - primitive order in the descriptor IS the primitive-ID order
 *     (= index in the Vec built by RaytracingScene::from_scene before the
 *     bounded/unbounded partition, src/render/scene.rs:84-87,
 *     src/core/bounds.rs:256-263).
 *   - a handle is used by one host thread at a time.
 */
#ifndef MI355RT_H
#define MI355RT_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_OK 0
#define RT_ERR_INVALID (-1)   /* bad argument / malformed descriptor          */
#define RT_ERR_UNSUPPORTED (-2) /* reference would panic (e.g. max_depth > 6) */
#define RT_ERR_DEVICE (-3)    /* HIP runtime failure                          */
#define RT_ERR_NOMEM (-4)     /* pool budget too small for one camera ray     */
#define RT_ERR_IO (-5)        /* host loader: file / parse error              */

#define RT_PRIM_SPHERE 0u   /* src/primitives/sphere.rs   */
#define RT_PRIM_CUBE 1u     /* src/primitives/cube.rs     */
#define RT_PRIM_PLANE 2u    /* src/primitives/plane.rs    */
#define RT_PRIM_TRIANGLE 3u /* src/primitives/triangle.rs */

#define RT_MAT_PHONG 0u    /* src/core/material.rs:22-47 */
#define RT_MAT_PHYSICAL 1u /* src/core/material.rs:60-89 */

#define RT_SIDE_BOTH 0u /* src/core/material.rs:9-20 (default Front) */
#define RT_SIDE_FRONT 1u
#define RT_SIDE_BACK 2u

#define RT_LIGHT_AMBIENT 0u /* src/lights/ambient.rs */
#define RT_LIGHT_POINT 1u   /* src/lights/point.rs   */

#define RT_NO_HIT 0xFFFFFFFFu

/* RenderOptions, src/render/mod.rs:133-159 (widths widened to u32). */
typedef struct rt_render_options {
    uint32_t width, height;
    uint32_t max_depth;             /* u8 in the reference  */
    uint32_t samples_per_pixel;     /* u16                  */
    uint32_t max_reflected_rays;    /* u16                  */
    uint32_t max_illumination_rays; /* u16                  */
    double max_occlusion_distance;
} rt_render_options;

/* RaytracingCamera, src/render/raytracing_scene.rs:25-42.  cam_to_world is the
 * upper-left 3x3 of look_at_rh(position,target,up).transpose() (row-major). */
typedef struct rt_camera {
    double fov_deg;
    double position[3];
    double cam_to_world[9];
} rt_camera;

/* Material::{Phong,Physical}, src/core/material.rs:22-107. */
typedef struct rt_material {
    uint32_t kind; /* RT_MAT_*  */
    uint32_t side; /* RT_SIDE_* */
    double color[3];
    double emissive[3];
    double specular[3];     /* phong    */
    double reflectivity;    /* phong    */
    double shininess;       /* phong    */
    double opacity;         /* physical */
    double roughness;       /* physical */
    double metalness;       /* physical */
    double refractive_index; /* physical */
    int32_t texture_id;     /* -1: none */
    uint32_t _pad;
} rt_material;

/* Light::{Ambient,Point}, src/lights/{ambient,point}.rs.  position = transform x origin
 * (src/core/transform.rs:11-13). */
typedef struct rt_light {
    uint32_t kind; /* RT_LIGHT_* */
    uint32_t _pad;
    double color[3];
    double intensity;
    double position[3];
} rt_light;

/* Texture (decoded to RGB8, row 0 = top), src/core/texture.rs:8-44. */
typedef struct rt_texture {
    uint32_t width, height;
    const uint8_t *rgb8; /* width*height*3 bytes */
} rt_texture;

typedef struct rt_scene_desc {
    rt_render_options opts;
    rt_camera camera;

    /* flattened primitives, in primitive-ID order */
    uint32_t n_prims;
    uint32_t _pad0;
    const uint8_t *prim_type;       /* [n_prims] RT_PRIM_*                       */
    const uint32_t *prim_material;  /* [n_prims] index into materials            */
    const uint32_t *prim_transform; /* [n_prims] index into world / inv_world    */
    const double *prim_param;       /* [n_prims] sphere radius | cube size | 0   */
    const double *prim_normal;      /* [n_prims*3] plane normal (as deserialised,
                                       NOT normalised), zeros otherwise          */
    const uint32_t *prim_tri;       /* [n_prims] triangle index or RT_NO_HIT     */

    /* world transforms (parent*child already baked, flatten_to_world) */
    uint32_t n_transforms;
    uint32_t _pad1;
    const double *world;     /* [n_transforms*12] */
    const double *inv_world; /* [n_transforms*12] */

    /* triangle vertex data in OBJECT space (src/primitives/triangle.rs:148-152) */
    uint32_t n_tris;
    uint32_t _pad2;
    const double *tri_pos; /* [n_tris*9] p0 p1 p2          */
    const double *tri_nrm; /* [n_tris*9] unit n0 n1 n2     */
    const double *tri_uv;  /* [n_tris*6] uv0 uv1 uv2       */

    uint32_t n_materials;
    uint32_t n_lights;
    const rt_material *materials;
    const rt_light *lights;

    uint32_t n_textures;
    uint32_t _pad3;
    const rt_texture *textures;
} rt_scene_desc;

/* What one call renders.  Pixels of the window [x0,x1) x [y0,y1), samples
 * [s0,s1) of each pixel (global sample indices: sample 0 is the pixel centre,
 * src/render/raytracing_scene.rs:560).  With shard_count > 1 only the
 * tile_size x tile_size tiles assigned to `shard_index` (scattered order) are
 * rendered; the other pixels of the output are left untouched.  All-zero
 * window / sample fields mean "whole image, all samples". */
typedef struct rt_render_params {
    uint64_t seed;
    uint32_t x0, y0, x1, y1;
    uint32_t s0, s1;
    uint32_t tile_size;   /* 0 -> 32 */
    uint32_t shard_index; /* rank    */
    uint32_t shard_count; /* 0/1 -> everything */
    uint32_t flags;       /* RT_FLAG_* */
} rt_render_params;

#define RT_FLAG_COUNT_WORK 1u /* also count BVH nodes visited / primitives tested */

/* Per-call measurements, filled by rt_render* when `stats` is non-NULL.
 * Kernel times are HIP-event times on the stream the kernels ran on. */
typedef struct rt_render_stats {
    uint64_t ray_count;        /* reference counting rule, CastStats.ray_count */
    uint64_t closest_rays;     /* get_color + get_illumination rays            */
    uint64_t shadow_rays;
    uint64_t camera_rays;
    uint64_t nodes_visited_closest, prims_tested_closest; /* RT_FLAG_COUNT_WORK */
    uint64_t nodes_visited_shadow, prims_tested_shadow;
    uint32_t batches;
    uint32_t launches_closest, launches_shadow, launches_other;
    float ms_total;   /* whole call, first launch to last            */
    float ms_raygen;  /* camera-ray generation                       */
    float ms_closest; /* closest-hit traversal kernels               */
    float ms_shadow;  /* any-hit traversal kernels                   */
    float ms_expand;  /* shade/expand (child + shadow ray emission)  */
    float ms_resolve; /* bottom-up shading resolve                   */
    float ms_accum;   /* per-pixel accumulate                        */
    uint32_t retries; /* batches re-run smaller after a pool overflow */
    /* global-illumination rays traced in place (k_trace_gi: drawn, traced and reduced per parent hit, no stored ray);
     * they are part of closest_rays, their kernel time and launches are NOT part of ms_closest / launches_closest */
    uint32_t launches_gi;
    float ms_gi;
    uint32_t gi_two_phase; /* the form of GI query this call used: 0 = one closest-hit query; two-phase on the inner levels (bit 0) / on the last traced level (bit 1) */
    uint32_t gi_waves;     /* and the k_trace_gi build: register budget for 3 or 4 waves per SIMD */
    uint32_t overlapped;   /* 1: the launches of a level ran beside each other on streams of the library (deep trees, fractions of a frame): the per-class times below are spans that overlap, their sum exceeds ms_total */
    uint64_t gi_rays;
    /* illumination rays of the last traced level answered by k_trace_illum_last (two-phase queries; part of closest_rays,
     * their time and launches are not part of ms_closest / ms_gi) */
    uint64_t illum_rays;
    float ms_illum;
    uint32_t launches_illum;
} rt_render_stats;

typedef struct rt_scene rt_scene; /* opaque */

/* Replaces Scene::build_raytracing_scene / KdTreeAccelerator::new
 * (src/render/scene.rs:76-97, src/core/bounds.rs:255-296): validates and copies
 * the descriptor, builds and linearises a BVH on the host, uploads everything
 * to the current HIP device.  Fails with RT_ERR_DEVICE when no GPU is usable. */
int rt_scene_create(const rt_scene_desc *desc, rt_scene **out);
void rt_scene_destroy(rt_scene *scene);

/* Limit the device memory used for wavefront pools (allocated at the first render).  Default: sized from the frame -- 64
 * node slots per camera ray of width x height x samples, at least 256 MiB, at most 24 GiB; a budget set here is taken as it is. */
int rt_scene_set_pool_bytes(rt_scene *scene, uint64_t bytes);

/* Replaces the pixel loop of raytrace_to_image
 * (src/render/raytracing_scene.rs:727-746): accumulates, per rendered pixel,
 * sum over samples of clamp(get_color(sample)) (:605-612) into
 * d_rgb_accum[(y*width+x)*3 + c] (DEVICE pointer, doubles, += semantics) and
 * adds the number of rays cast to *ray_count (HOST pointer, may be NULL).
 * `stream` is a hipStream_t (NULL = the default stream).  Blocking. */
int rt_render_device(rt_scene *scene, const rt_render_params *params, double *d_rgb_accum,
                     uint64_t *ray_count, rt_render_stats *stats, void *stream);

/* rt_render_device for callers that cannot allocate (tight serving loops, harnesses that forbid hipMalloc in the hot
 * path, processes whose device memory is spoken for).  NO COUNTERPART IN THE REFERENCE.  BLOCKING like rt_render_device:
 * the call synchronises the stream after every batch (batch sizes follow counters read back from the device), so it
 * cannot be recorded into a HIP graph -- a caller under stream capture must end the capture first.  Same arithmetic, same kernels and the same results as
 * rt_render_device, but the wavefront pools, counters and tile list live in a static device array of
 * RT_NOALLOC_SCRATCH_BYTES inside the library (one copy per GPU the library runs on, held until the process ends)
 * instead of the scene's hipMalloc'ed pool budget: the call allocates and frees nothing (no hipMalloc / hipFree, no
 * events; the counters come back into pinned host memory allocated once per device at rt_scene_create; no statistics
 * are offered).  Limits: at most RT_NOALLOC_MAX_CAMERA_RAYS camera rays per call (window pixels
 * x samples, after sharding) and 4096 tiles; larger calls fail with RT_ERR_NOMEM and a message -- split the window or
 * the sample range.  Every call on a GPU shares that GPU's array: calls that overlap in time on one GPU (two host
 * threads, two streams) race; serialise them.  The array keeps the last call's contents; the library resets what must
 * start at zero (the batch counters) itself. */
#define RT_NOALLOC_SCRATCH_BYTES (128ull << 20)
#define RT_NOALLOC_MAX_CAMERA_RAYS (1u << 18)
int rt_render_device_noalloc(rt_scene *scene, const rt_render_params *params, double *d_rgb_accum, uint64_t *ray_count,
                             void *stream);

/* As rt_render_device, plus the other two members of the ColorData that screen_raycast returns
 * (src/render/mod.rs:16-20, src/render/raytracing_scene.rs:598-622), which the reference hands to its
 * denoise pass (:645-701):
 *   d_albedo_accum[(y*width+x)*3 + c] += sum over the samples of the clamped albedo (:543, :610); the
 *       final albedo is clamp(accum * (1/spp)) (:616, :619);
 *   d_normal[(y*width+x)*3 + c] = normal of sample 0's ColorData (:603-605): the shading normal of the
 *       first hit, or the normal at the end of its refraction chain (:485), (0,0,1) for a miss
 *       (ColorData::black, mod.rs:39-41).  Written only when the sample range contains sample 0.
 * Both are DEVICE pointers to w*h*3 doubles; either may be NULL. */
int rt_render_aov_device(rt_scene *scene, const rt_render_params *params, double *d_rgb_accum, double *d_albedo_accum,
                         double *d_normal, uint64_t *ray_count, rt_render_stats *stats, void *stream);

/* Host-pointer convenience for the AOVs: zero-initialised accumulation, render, copy back.  rgb_accum, albedo
 * (FINAL: clamp(sum * (1/spp))) and normal are w*h*3 doubles each and may be NULL. */
int rt_render_aov(rt_scene *scene, const rt_render_params *params, double *rgb_accum, double *albedo, double *normal,
                  uint64_t *ray_count, rt_render_stats *stats);

/* Replaces :614-621 + :757-769: colour = accum / spp, clamp, ^(1/2.2),
 * (c*255.0) as u8, alpha 255.  Both pointers are DEVICE pointers. */
int rt_tonemap_device(rt_scene *scene, const double *d_rgb_accum, uint8_t *d_rgba8, void *stream);

/* Host-pointer convenience: zero-initialised accumulation, render, tonemap,
 * copy back.  rgb_accum (w*h*3 doubles) and rgba8 (w*h*4 bytes) may be NULL. */
int rt_render(rt_scene *scene, const rt_render_params *params, double *rgb_accum, uint8_t *rgba8,
              uint64_t *ray_count, rt_render_stats *stats);

/* One host process, several GPUs (or several streams of one): the multi-device form of the pixel loop, SURVEY 8(b)/(e).
 * The reference's only parallelism is rayon's par_iter over the pixels of a frame (src/render/raytracing_scene.rs:734-744);
 * here `scenes[0..n_scenes)` are replicas of ONE scene (rt_scene_create called n_scenes times with the same descriptor,
 * each with the device it shall live on current), and the frame described by `params` (window, sample range, seed;
 * shard_index / shard_count must be 0 -- the library deals the tiles, tile_size 0 -> 8) is rendered by all of them at
 * once: scene i takes the i-th of n_scenes Morton-interleaved tile shards on a host thread of its own, the shards'
 * tiles travel to scenes[0]'s device (hipMemcpyPeerAsync, direct over xGMI where peer access exists; 1/n_scenes of
 * the frame per link) and are added into d_rgb_accum there.  Every pixel has exactly one contributor, so the result is
 * bit-identical to one rt_render_device call.  d_rgb_accum: DEVICE pointer on scenes[0]'s device (+= semantics);
 * stream: a hipStream_t of that device (NULL = default); stats: NULL or n_scenes entries (one per scene).  Blocking. */
int rt_render_multi_device(rt_scene *const *scenes, uint32_t n_scenes, const rt_render_params *params, double *d_rgb_accum,
                           uint64_t *ray_count, rt_render_stats *stats, void *stream);
/* Host-pointer convenience, as rt_render: zero-initialised accumulation on scenes[0]'s device, render, tonemap, copy back. */
int rt_render_multi(rt_scene *const *scenes, uint32_t n_scenes, const rt_render_params *params, double *rgb_accum, uint8_t *rgba8,
                    uint64_t *ray_count, rt_render_stats *stats);

/* Replaces raytrace_to_buffer's progressively filled window buffer (src/render/raytracing_scene.rs:778-875): the
 * frame is rendered in `n_passes` passes; pass k completes (all samples) the 8x8 tiles dealt to it in a scattered order
 * (the reference visits the pixels in a shuffled order, :824-825) and writes `utils::to_argb_u32` (src/utils/mod.rs:14-19)
 * of their colour into argb[y*width+x] (HOST pointer, w*h words; pixels not rendered yet are 0 as in :800-801).  After
 * every pass `on_pass(user, passes_done, n_passes, rays_so_far)` is called (may be NULL) with `argb` up to date -- the
 * place to present the buffer, :865-871; a non-zero return stops the render (the reference stops when its window closes).
 * After the last pass `argb` holds the same pixels as rt_render's rgba8.  rgb_accum (HOST, w*h*3 doubles) may be NULL. */
typedef int (*rt_progress_fn)(void *user, uint32_t passes_done, uint32_t n_passes, uint64_t rays_so_far);
int rt_render_progressive(rt_scene *scene, const rt_render_params *params, uint32_t n_passes, uint32_t *argb,
                          rt_progress_fn on_pass, void *user, double *rgb_accum, uint64_t *ray_count);

/* Parity AOVs: closest hit of the pixel-centre camera ray of every pixel
 * (get_color's raycast + compute_data, :531-532).  HOST pointers, w*h entries
 * (normal: w*h*3); prim_id = RT_NO_HIT on a miss. */
int rt_query_primary(rt_scene *scene, uint32_t *prim_id, double *t, double *normal3);

/* Traversal-only probes used by the parity tests: n world-space rays
 * (HOST pointers, origin/dir n*3 doubles).  shadow == 0: closest hit
 * (KdTreeAccelerator::raycast, bounds.rs:302-308; ray_type Secondary);
 * shadow != 0: any hit within tmax[i] (shadow_cast, bounds.rs:310-316),
 * prim_id[i] = 1 if occluded else 0. */
int rt_trace_rays(rt_scene *scene, uint32_t n, const double *origin, const double *dir,
                  const double *tmax, int shadow, uint32_t *prim_id, double *t);

/* BVH facts for reports: node count, leaf count, max depth, bytes resident,
 * bounded / unbounded primitive counts, node / job pool capacities. */
int rt_scene_info(rt_scene *scene, uint64_t *out8);

/* The acceleration structure as resident on the device, for checks and reports (the counterpart of nothing in the
 * reference: its kd-tree, src/core/bounds.rs:425-450, is private).  nodes32: HOST buffer of node_capacity 32-byte nodes
 * = { u16 lmin[3], lmax[3], rmin[3], rmax[3]; u32 left, right } in breadth-first order, child boxes in the 16-bit frame
 * of the scene, child ref bit 31 = leaf (bits 28..30 = primitives - 1, bits 0..27 = first leaf record);
 * leaf_prim_ids: HOST buffer, the primitive id of every leaf record in leaf order.  Either may be NULL.
 * info4 = { node count, leaf-record count, root ref, 1 if the tree was built on the device }.
 * Scenes of 65,536 bounded primitives and more are built on the GPU (binned SAH, the host builder's split rule);
 * the environment variable MI355RT_BVH=host|device forces either. */
int rt_scene_bvh(rt_scene *scene, void *nodes32, uint64_t node_capacity, uint32_t *leaf_prim_ids, uint64_t leaf_capacity,
                 uint64_t *info4);

/* The tile partition rt_render_device uses for `params` (pure host code, no
 * GPU needed): writes up to `cap` tile origins (x,y pairs) to tile_xy and
 * returns the tile count and tile extent.  Every rank of a multi-GPU job must
 * see the same partition; tiles are dealt to shards in Morton order. */
int rt_shard_tiles(const rt_render_options *opts, const rt_render_params *params, uint32_t *tile_xy,
                   uint32_t cap, uint32_t *n_tiles, uint32_t *tile_w, uint32_t *tile_h);

const char *rt_last_error(void);
const char *rt_version(void);

#ifdef __cplusplus
}
#endif
#endif /* MI355RT_H */
