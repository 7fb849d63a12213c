// rt_internal.h -- device-resident data layout shared by host and kernels.
//
// Everything the wavefront kernels touch lives in HBM as flat arrays:
//   * scene (read-only): BVH nodes (32 B: both children's boxes as 16-bit fixed
//     point + two child references), leaf primitive records in BVH-leaf order
//     (128 B: the primitive's own world AABB + object-space geometry),
//     per-primitive shading records in
//     primitive-ID order, transforms, materials, lights, textures;
//   * wavefront state (per batch): SoA node pool (ray, hit/result, expansion
//     info) and SoA shadow-ray pool, bump-allocated level by level.
#pragma once
#include <stdint.h>

#include "../../include/mi355rt.h"

namespace rt {

constexpr double BIAS = 1e-10;                   // src/render/mod.rs:14
constexpr double GAMMA = 2.2;                    // src/render/mod.rs:13
constexpr double F64_EPS = 2.220446049250313e-16;
constexpr uint32_t MAX_LEVELS = 8;
constexpr uint32_t STACK_DEPTH = 40;             // per-lane LDS traversal stack entries
constexpr uint32_t MAX_LEAF_PRIMS = 8;
constexpr uint32_t LEAF_FLAG = 0x80000000u;

// RNG child slots (DESIGN.md "RNG"; identical constants in oracle/rt_oracle.cpp)
constexpr uint32_t SLOT_REFRACTION = 0x40000000u;
constexpr uint32_t SLOT_GI = 0x80000000u;

// BVH2 node, 32 bytes = two 16-byte loads per visit (the traversal kernels are bound
// by vector-memory instruction issue, not by bytes): both children's boxes as
// 16-bit fixed point in the scene frame (SceneDev::q_origin / q_scale), minima
// rounded down and maxima rounded up from the exact f64 unions of the
// primitives' culling boxes, + child references.  Node boxes only cull
// (conservatively); every hit decision is taken in f64 at the leaves.
// ref: bit31 set -> leaf, bits 28..30 = count-1, bits 0..27 = first leaf record.
struct alignas(16) BvhNode {
    uint16_t lmin[3], lmax[3];
    uint16_t rmin[3], rmax[3];
    uint32_t left, right;
};
static_assert(sizeof(BvhNode) == 32, "BvhNode must be 32 bytes");

// Leaf primitive record (bounded primitives only), BVH-leaf order.
//   meta: bits 0..1 type, bits 2..3 side, bits 4..31 transform index
//   geom: triangle p0, e1 = p1 - p0, e2 = p2 - p0 (object space; the edges are the
//         first thing triangle.rs:196-197 computes) | geom[0] = sphere radius / cube size
//   center / half: the primitive's own world AABB (bounds.rs:40-72) in the form its
//         slab test uses it: (min + max) * 0.5 and (max - min) / 2 (bounds.rs:116-121),
//         evaluated once on the host with the same f64 operations
struct alignas(16) LeafPrim {
    double center[3], half[3];
    double geom[9];
    uint32_t prim_id;
    uint32_t meta;
};
static_assert(sizeof(LeafPrim) == 128, "LeafPrim must be 128 bytes");

// Unbounded primitives (planes), scene order; tested for every ray.
struct alignas(16) PlanePrim {
    double normal[3];
    uint32_t prim_id;
    uint32_t meta;  // as LeafPrim::meta
};

// prim_flags bits
constexpr uint32_t PRIM_ILLUM_TERMINAL = 1u;  // get_emissive_* of this surface = its emissive colour, no child rays
constexpr uint32_t PRIM_EMISSIVE = 2u;        // emissive colour is not all zero

// Shading-side record, primitive-ID order.
struct alignas(16) ShadePrim {
    uint32_t type;
    uint32_t material;
    uint32_t xf;
    uint32_t tri;      // index into tri_nrm / tri_uv
    double param;      // radius / size
    double normal[3];  // plane normal
    double uv_rot[6];  // plane: rows 0 and 2 of rotation_between(normal, +y)
    uint32_t uv_ok;    // plane: rotation exists (reference panics otherwise)
    uint32_t _pad;
};

struct DTexture {
    uint32_t width, height;
    uint64_t offset;  // byte offset into the texel buffer
};

struct DLight {
    uint32_t kind, _pad;
    double color[3];
    double intensity;
    double position[3];
};

struct SceneDev {
    rt_render_options opts;
    // camera
    double cam_pos[3];
    double cam_to_world[9];
    double fov_scale;  // tan(fov/2), computed on the host
    // acceleration structure
    const BvhNode *nodes;
    const LeafPrim *leaf_prims;
    const PlanePrim *planes;
    uint32_t root_ref;    // child-ref encoding; 0xFFFFFFFF = no bounded primitives
    uint32_t n_top_nodes; // nodes [0, n_top_nodes) (breadth-first order: the top of the tree) are cached in LDS by the traversal kernels
    uint32_t n_planes;
    uint32_t n_bounded;
    uint32_t n_prims;
    double scene_scale;   // max |coordinate| of the root box
    float q_origin[3];    // node boxes: coordinate = q_origin + q * q_scale (q = 0..65535)
    float q_scale[3];
    // two-phase GI queries (k_trace_gi): the special primitives have a tree of their own, in the frame of the main one.
    // Which primitives are special depends on the level: where the hit's own children would still be traced, the
    // emissive ones and those that cast child rays (PRIM_* flags); on the last traced level -- children return zero,
    // :495-497 -- the emissive ones only.  A set is usable when it is small (`ok`).
    struct SpecialSet {
        const BvhNode *nodes;
        const LeafPrim *leaf_prims;
        uint32_t root_ref;  // 0xFFFFFFFF: no bounded primitive in the set
        uint32_t n_top;     // nodes of the tree kept in LDS behind the main tree's (set 0, then set 1)
        uint32_t count;     // primitives in the set, planes included
        uint32_t ok;        // the set is small enough for two-phase queries to be considered
    } special[2];           // [0] inner levels, [1] the last traced level
    uint32_t gi_two_phase;  // k_trace_gi: closest special hit + any-hit instead of one closest-hit query; bit 0 inner levels, bit 1 last level (where the set is ok)
    // shading
    const ShadePrim *shade_prims;
    const uint32_t *prim_flags;  // primitive-ID order, PRIM_* bits
    const double *inv_world;  // [n_transforms*12]
    const double *tri_nrm;    // [n_tris*9]
    const double *tri_uv;     // [n_tris*6]
    const rt_material *materials;
    const DLight *lights;
    const DLight *point_lights;  // the point lights among them, in order (shadow-ray light index -> light)
    const DTexture *textures;
    const uint8_t *texels;
    uint32_t n_lights;
    uint32_t n_point_lights;
};

// meta word of a node (ray) record
//   bits 0..2  depth
//   bit  3     kind: 0 = colour node (get_color), 1 = illumination node (get_illumination)
//   bit  4     refraction child (ray carries the parent's material ior)
//   bit  5     inactive (padding pixel outside the render window)
//   bit  6     probe ray of rt_trace_rays (treated as Secondary)
//   bit  7     done: the trace kernel already wrote the node's final result (misses,
//              illumination nodes that hit a surface with nothing to gather)
//   bit  8     traced: the hit fields are already filled in (a GI ray that k_trace_gi found to need a node of its own)
//   bits 9..31 material of the parent hit (valid when bit 4)
constexpr uint32_t META_DEPTH_MASK = 7u;
constexpr uint32_t META_ILLUM = 8u;
constexpr uint32_t META_REFR = 16u;
constexpr uint32_t META_INACTIVE = 32u;
constexpr uint32_t META_DONE = 128u;  // bit 7: result already final (set by k_trace_closest), nothing to expand or resolve
constexpr uint32_t META_TRACED = 256u;  // bit 8
constexpr uint32_t META_MAT_SHIFT = 9u;

// Node pool: structure of arrays of 16-byte elements where a record's fields travel
// together (one 16-byte access per lane is the widest, and a wave's 64 of them are
// 1 KB contiguous), plain words where a field is scanned or rewritten on its own.
// After resolve, (t,bu,bv) hold the node's RGB result and `prim` its flag word
// (bit 0: occluded, illumination nodes; bit 1: RGB is not all zero).
struct alignas(16) Vec2d {
    double x, y;
};
struct alignas(16) ExpandInfo {
    uint32_t first_child;
    uint32_t job;         // the colour node's ShadeJob (its shadow rays and, in place, its GI rays); NO_JOB: none
    uint32_t counts;      // nR | nT << 16 | nG << 17; bit 31: the GI rays are the job's (traced in place)
    uint32_t light_mask;  // point lights that cast a shadow ray
};
constexpr uint32_t NO_JOB = 0xFFFFFFFFu;
// Rays in place.  The rays a colour node's shading casts from its hit point and that end at once are never stored as
// rays: k_expand leaves ONE job per colour node -- the hit point, the normal, the RNG key -- and the trace kernels
// generate, trace and reduce them per job.
//   * GI rays (compute_global_illumination, raytracing_scene.rs:98-133): 3 of 4 closest-hit rays of a frame, and 9 of 10
//     of those end at once (a miss, or a surface whose get_illumination gathers nothing).  k_trace_gi draws the G(d)
//     hemisphere rays of every job.  What a parent needs back: how many rays were unoccluded, and -- in slot order, so
//     that the f64 sum of :121 keeps its order -- the few rays whose emissive contribution is not zero: a terminal hit
//     on an emissive surface (kind 0: primitive id) or a hit that needs shading of its own and got a node (kind 1:
//     node index, resolved one level down before the parent reads it).
//   * shadow rays (:216-225, :428-437): one per point light on whose side the surface lies (`light_mask`), from the
//     light to the hit point.  k_trace_shadow derives each from (light, P) and leaves one bit per light.
struct alignas(16) GiJob {
    double P[3], n[3];  // hit point, shading normal
    uint64_t path;      // the parent's RNG path
    uint32_t root;      // camera-ray item
    uint32_t pixel;
    uint32_t sample;
    uint32_t light_mask;  // point lights (index among the point lights) that get a shadow ray
    uint32_t _pad[2];
};
static_assert(sizeof(GiJob) == 80, "GiJob must be 80 bytes");
// What the trace kernels leave for k_resolve, apart from the job (16 dense bytes per job: k_resolve reads nothing else of
// it, and a 16-byte read out of an 80-byte record costs a cache line all the same).
struct alignas(16) JobResult {
    uint32_t gi_counts;        // k_trace_gi: bits 0..7 unoccluded rays, bits 8..15 special rays
    uint32_t first_special;    // k_trace_gi
    uint32_t shadow_occluded;  // k_trace_shadow: the lights of the job's light_mask whose ray was stopped
    uint32_t _pad;
};
struct GiSpecial {
    uint32_t slot_kind;  // bits 0..30 slot g, bit 31 kind
    uint32_t id;         // kind 0: primitive id (terminal hit, emissive surface); kind 1: node index
};
constexpr uint32_t GI_MAX_INPLACE = 64;  // G(d) above a wave's width: the rays are materialised as nodes (the round-1 path)
constexpr uint32_t job_unoccluded(uint32_t gi_counts) { return gi_counts & 0xFFu; }  // constexpr: host and device code alike
constexpr uint32_t job_n_special(uint32_t gi_counts) { return (gi_counts >> 8) & 0xFFu; }
struct GiPool {
    GiJob *jobs;
    JobResult *results;  // per job
    GiSpecial *specials;
    uint32_t job_capacity, special_capacity;
};

struct NodePool {
    Vec2d *r0, *r1, *r2;  // ray: (o.x, o.y) (o.z, d.x) (d.y, d.z)
    uint64_t *path;
    uint32_t *root;  // camera-ray index within the call's work list
    uint32_t *meta;
    // hit / result
    uint32_t *prim;
    double *t, *bu, *bv;
    ExpandInfo *xinfo;  // written by k_expand for the nodes it expands, read by k_resolve
    // Live lists: the nodes of a level that still need k_expand / k_resolve (not finished by the trace kernel), written
    // by the trace kernel into the level's own index range of this array -- colour nodes from the range's start
    // upwards, illumination nodes from its end downwards (Counters::live_colour / live_illum count them).
    uint32_t *live;
    uint32_t capacity;
};

// Device-side counters of one batch.
struct Counters {
    uint32_t level_start[MAX_LEVELS + 1];
    uint32_t level_count[MAX_LEVELS + 1];
    unsigned long long node_alloc;    // bump pointer into the node pool
    unsigned long long shadow_rays;   // shadow rays traced (k_trace_shadow; statistics)
    unsigned long long job_alloc;      // bump pointer into the shard's GI jobs
    unsigned long long special_alloc;  // bump pointer into the shard's GI specials
    uint32_t job_level_start[MAX_LEVELS + 1];  // jobs left by k_expand(level): [job_level_start[level], job_alloc) until the level is closed
    uint32_t child_end[MAX_LEVELS + 1];        // node_alloc when k_expand(level - 1) had finished: the children it emitted are
                                               // [level_start[level - 1] + level_count[level - 1], child_end[level]) (k_close_expand)
    unsigned long long gi_rays;          // GI rays traced in place
    unsigned long long gi_materialised;  // of which got a node (also counted in level_count)
    unsigned long long illum_last_rays;  // illumination nodes of the last traced level answered by k_trace_illum_last
    uint32_t live_colour[MAX_LEVELS + 1], live_illum[MAX_LEVELS + 1];  // entries of NodePool::live per level
    uint32_t overflow;       // a pool ran out: the batch must be re-run smaller
    uint32_t active_camera;  // camera rays inside the window
    unsigned long long nodes_visited_closest, prims_tested_closest;
    unsigned long long nodes_visited_shadow, prims_tested_shadow;
    // RT_FLAG_COUNT_WORK census per level: [0] colour nodes, [1] illumination nodes,
    // [2] misses, [3] nodes that emit neither children nor shadow rays
    unsigned long long census[MAX_LEVELS][4];
    // RT_FLAG_COUNT_WORK, closest-hit kernel, wave level: node visits, leaf rounds, waves that ran, node visits of the busiest wave
    unsigned long long wave_node_steps, wave_leaf_rounds, wave_iterations, wave_max_steps;
    // leaf rounds of the closest-hit kernel by what the wave held: bit 0 triangles, bit 1 spheres, bit 2 cubes, bit 3 some lane took the exact box test
    unsigned long long leaf_census[16];
    unsigned long long leaf_lanes[4];  // lanes in leaf rounds, lanes taking the exact box test, lanes whose primitive was hit
};

// The wavefront pools are split into `count` shards.  A shard is an independent
// sub-batch: its own bump allocators (Counters), its own slice of the node and
// shadow pools, its own camera rays (granules of `granule` items dealt round-robin,
// so every shard samples the whole image) and its own thread blocks
// (blockIdx.x % count; with count a multiple of 8 a shard stays on one XCD, whose
// L2 then holds what the shard wrote).  One allocator word per shard instead of one
// per batch: a single word saturates at ~90 returning atomics per microsecond.
constexpr uint32_t MAX_SHARDS = 64;
struct Shards {
    uint32_t count;
    uint32_t node_cap;    // nodes per shard
    uint32_t job_cap;     // GI jobs per shard
    uint32_t special_cap; // GI specials per shard
    uint32_t granule;     // camera-ray items per deal (multiple of the samples per pixel)
};

// One call's work list: tiles of tile_w x tile_h pixels, ns samples each.
struct WorkList {
    const uint32_t *tile_xy;  // [n_tiles*2] tile origin (pixels)
    uint32_t n_tiles;
    uint32_t tile_w, tile_h;
    uint32_t row_mul;         // rows of a tile are visited in the order (r * row_mul) % tile_h (coprime with tile_h; 1 = in order)
    uint32_t x0, y0, x1, y1;  // window clip
    uint32_t s0, ns;          // sample range [s0, s0+ns)
    uint64_t seed;
};

}  // namespace rt
