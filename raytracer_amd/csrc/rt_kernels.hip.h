// rt_kernels.hip.h -- wavefront kernels of the hot path (gfx950, wave64).
//
// One batch of camera rays is expanded level by level into a node pool
// (level d = rays of depth d).  Per level: closest-hit traversal -> expand
// (emit child rays + shadow rays).  Then one any-hit launch over every shadow
// ray of the batch, then a bottom-up resolve (the reference's ray tree is not a
// linear throughput product: results are clamped at every get_color return and
// the direct + GI term is scaled by the AO fraction, raytracing_scene.rs:246-258,
// :465-486, :543), then a per-pixel accumulate.
//
//   reference function (raytracing_scene.rs)      kernel
//   build_camera_rays :549-596                    k_raygen
//   raycast :90 -> bounds.rs:302-364              k_trace_closest
//   shadow_cast :94 -> bounds.rs:310-316,366-399  k_trace_shadow
//   get_color / get_illumination :492-547,
//     child-ray construction in :98-133,
//     :135-167, :169-261, :263-337, :339-489      k_expand (rays) + k_resolve (radiance)
//   screen_raycast :598-622 (sum over samples)    k_accumulate
//   gamma + quantise :614-621, :757-769           k_tonemap
#pragma once
#include <hip/hip_runtime.h>

#include "rt_device_math.hip.h"
#include "rt_internal.h"

namespace rt {

constexpr int BLOCK = 256;
constexpr double PI_D = 3.14159265358979323846;
constexpr double HALF_PI_D = 1.57079632679489661923;
constexpr double INV_PI_D = 0.318309886183790671538;

// ------------------------------------------------------------------ helpers
__device__ __forceinline__ void load_ray(const NodePool &np, uint32_t i, d3 *o, d3 *d) {
    const Vec2d a = np.r0[i], b = np.r1[i], c = np.r2[i];
    *o = mk3(a.x, a.y, b.x);
    *d = mk3(b.y, c.x, c.y);
}
__device__ __forceinline__ void store_ray(const NodePool &np, uint32_t i, d3 o, d3 d) {
    np.r0[i] = Vec2d{o.x, o.y};
    np.r1[i] = Vec2d{o.z, d.x};
    np.r2[i] = Vec2d{d.y, d.z};
}
__device__ __forceinline__ void item_to_pixel(const WorkList &wl, uint32_t item, uint32_t *px, uint32_t *py,
                                              uint32_t *sample) {
    uint32_t s_local = item % wl.ns;
    uint32_t pl = item / wl.ns;
    uint32_t per_tile = wl.tile_w * wl.tile_h;
    uint32_t tile = pl / per_tile, pit = pl % per_tile;
    *px = wl.tile_xy[2 * tile] + pit % wl.tile_w;
    *py = wl.tile_xy[2 * tile + 1] + ((pit / wl.tile_w) * wl.row_mul) % wl.tile_h;
    *sample = wl.s0 + s_local;
}

struct ShardCtx {
    Counters *ctr;
    uint32_t node_off;              // first pool index of the shard
    uint32_t job_off, special_off;
    uint32_t bid, nblocks;          // this block's index / the block count within the shard
    uint32_t shard;
};
// gridDim.x is a multiple of sh.count (host guarantees)
__device__ __forceinline__ ShardCtx shard_ctx(Counters *ctrs, const Shards &sh) {
    ShardCtx c;
    c.shard = blockIdx.x % sh.count;
    c.ctr = ctrs + c.shard;
    c.node_off = c.shard * sh.node_cap;
    c.job_off = c.shard * sh.job_cap;
    c.special_off = c.shard * sh.special_cap;
    c.bid = blockIdx.x / sh.count;
    c.nblocks = gridDim.x / sh.count;
    return c;
}
// batch-relative item of a shard's j-th camera ray
__device__ __forceinline__ uint32_t shard_item(const Shards &sh, uint32_t shard, uint32_t j) {
    return ((j / sh.granule) * sh.count + shard) * sh.granule + j % sh.granule;
}

// Work map of one level: the node (or shadow-ray) lists of all shards as ONE list
// of 64-entry groups, so every block of a launch gets the same share of work
// whatever the shards hold (a shard that received the expensive pixels of a small
// batch would otherwise keep its blocks busy long after the others have drained).
// Shards stay the unit of ALLOCATION (a block bump-allocates from the shard
// blockIdx.x % count, wherever the parent lives); records are addressed by
// absolute pool index, so parents and children may sit in different shards.
struct LevelMap {
    uint32_t prefix[MAX_SHARDS + 1];  // groups before shard k
    uint32_t base[MAX_SHARDS];        // pool index of the shard's first entry of this list
    uint32_t count[MAX_SHARDS];
    uint32_t dead;                    // some shard overflowed: the batch will be re-run, do nothing
};
// All threads of the block call it (one __syncthreads inside); returns the number of groups.
// What the entries of a work map are:
//   MAP_NODES      the nodes of `level` (closed by k_advance_level(level - 1))
//   MAP_CHILDREN   the nodes of `level` that k_expand(level - 1) emitted (closed by k_close_expand(level - 1)): what a
//                  closest-hit launch can trace while k_trace_gi(level) is still appending the level's other nodes
//   MAP_JOBS       the jobs of all levels
//   MAP_LEVEL_JOBS the jobs k_expand(level) left (closed by k_close_expand(level))
enum : uint32_t { MAP_NODES = 0, MAP_CHILDREN = 1, MAP_JOBS = 2, MAP_LEVEL_JOBS = 3 };
__device__ __forceinline__ uint32_t level_map_build(LevelMap &m, const Counters *ctrs, const Shards &sh, uint32_t level,
                                                   uint32_t what) {
    if (threadIdx.x < 64u) {
        const uint32_t k = threadIdx.x;
        uint32_t cnt = 0, over = 0;
        if (k < sh.count) {
            over = ctrs[k].overflow;
            if (what == MAP_NODES) {
                cnt = ctrs[k].level_count[level];
                m.base[k] = k * sh.node_cap + ctrs[k].level_start[level];
            } else if (what == MAP_CHILDREN) {
                const uint32_t start = ctrs[k].level_start[level - 1u] + ctrs[k].level_count[level - 1u];
                cnt = over ? 0u : ctrs[k].child_end[level] - start;
                m.base[k] = k * sh.node_cap + start;
            } else if (what == MAP_JOBS) {
                cnt = static_cast<uint32_t>(ctrs[k].job_alloc);
                m.base[k] = k * sh.job_cap;
            } else {
                cnt = over ? 0u : ctrs[k].job_level_start[level + 1u] - ctrs[k].job_level_start[level];
                m.base[k] = k * sh.job_cap + ctrs[k].job_level_start[level];
            }
            m.count[k] = cnt;
        }
        uint32_t x = (cnt + 63u) / 64u;
#pragma unroll
        for (uint32_t off = 1; off < 64; off <<= 1) {
            const uint32_t y = __shfl_up(x, off, 64);
            if (k >= off) x += y;
        }
        if (k < sh.count) m.prefix[k + 1] = x;
        if (k == 0) {
            m.prefix[0] = 0;
        }
        const unsigned long long any_over = __ballot(over != 0);
        if (k == 0) m.dead = any_over ? 1u : 0u;
    }
    __syncthreads();
    return m.dead ? 0u : m.prefix[sh.count];
}
// group -> (pool index of its first entry, valid entries in it)
__device__ __forceinline__ uint32_t level_map_group(const LevelMap &m, const Shards &sh, uint32_t group, uint32_t *n_valid,
                                                   uint32_t *shard = nullptr) {
    uint32_t k = 0;
#pragma unroll
    for (uint32_t step = MAX_SHARDS / 2; step >= 1; step >>= 1)
        if (k + step < sh.count && m.prefix[k + step] <= group) k += step;
    const uint32_t first = (group - m.prefix[k]) * 64u;
    *n_valid = min(64u, m.count[k] - first);
    if (shard) *shard = k;
    return m.base[k] + first;
}

// Work map over the LIVE nodes of a level (NodePool::live, written by k_build_live): per shard the colour list and /
// or the illumination list, again as one list of 64-entry groups.  Built like LevelMap; a group resolves to 64
// consecutive list entries.
struct LiveMap {
    uint32_t prefix[MAX_SHARDS + 1];  // groups before shard k
    uint32_t lo[MAX_SHARDS], hi[MAX_SHARDS];  // the level's index range of shard k in NodePool::live: [lo, hi)
    uint32_t n_colour[MAX_SHARDS], n_illum[MAX_SHARDS];
    uint32_t dead;
};
__device__ __forceinline__ uint32_t live_map_build(LiveMap &m, const Counters *ctrs, const Shards &sh, uint32_t level, bool colour,
                                                  bool illum) {
    if (threadIdx.x < 64u) {
        const uint32_t k = threadIdx.x;
        uint32_t nc = 0, ni = 0, over = 0;
        if (k < sh.count) {
            nc = colour ? ctrs[k].live_colour[level] : 0u;
            ni = illum ? ctrs[k].live_illum[level] : 0u;
            over = ctrs[k].overflow;
            m.lo[k] = k * sh.node_cap + ctrs[k].level_start[level];
            m.hi[k] = m.lo[k] + ctrs[k].level_count[level];
            m.n_colour[k] = nc;
            m.n_illum[k] = ni;
        }
        uint32_t x = (nc + 63u) / 64u + (ni + 63u) / 64u;
#pragma unroll
        for (uint32_t off = 1; off < 64; off <<= 1) {
            const uint32_t y = __shfl_up(x, off, 64);
            if (k >= off) x += y;
        }
        if (k < sh.count) m.prefix[k + 1] = x;
        if (k == 0) m.prefix[0] = 0;
        const unsigned long long any_over = __ballot(over != 0);
        if (k == 0) m.dead = any_over ? 1u : 0u;
    }
    __syncthreads();
    return m.dead ? 0u : m.prefix[sh.count];
}
// group, lane -> node index (0xFFFFFFFF when the lane has no entry)
__device__ __forceinline__ uint32_t live_map_node(const LiveMap &m, const Shards &sh, const uint32_t *__restrict__ live, uint32_t group,
                                                 uint32_t lane) {
    uint32_t k = 0;
#pragma unroll
    for (uint32_t step = MAX_SHARDS / 2; step >= 1; step >>= 1)
        if (k + step < sh.count && m.prefix[k + step] <= group) k += step;
    uint32_t g = group - m.prefix[k];
    const uint32_t gc = (m.n_colour[k] + 63u) / 64u;
    if (g < gc) {  // colour entries grow upwards from lo
        const uint32_t e = g * 64u + lane;
        return e < m.n_colour[k] ? live[m.lo[k] + e] : 0xFFFFFFFFu;
    }
    g -= gc;       // illumination entries grow downwards from hi
    const uint32_t e = g * 64u + lane;
    return e < m.n_illum[k] ? live[m.hi[k] - 1u - e] : 0xFFFFFFFFu;
}

// Wave-aggregated bump allocation: every lane of the wave must call it.
__device__ __forceinline__ uint64_t wave_alloc(unsigned long long *counter, uint32_t n, uint32_t *wave_total) {
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t x = n;
#pragma unroll
    for (uint32_t off = 1; off < 64; off <<= 1) {
        uint32_t y = __shfl_up(x, off, 64);
        if (lane >= off) x += y;
    }
    uint32_t total = __shfl(x, 63, 64);
    unsigned long long base = 0;
    if (lane == 63 && total) base = atomicAdd(counter, static_cast<unsigned long long>(total));
    base = __shfl(base, 63, 64);
    *wave_total = total;
    return base + x - n;
}

// first p in [0,64) with incl[p] > j (incl = inclusive prefix sums of one wave)
__device__ __forceinline__ uint32_t wave_owner(const uint32_t *incl, uint32_t j) {
    uint32_t lo = 0;
#pragma unroll
    for (uint32_t step = 32; step >= 1; step >>= 1)
        if (incl[lo + step - 1] <= j) lo += step;
    return lo;
}

__device__ __forceinline__ uint32_t ipow(uint32_t b, uint32_t e) {
    uint32_t r = 1;
    for (uint32_t i = 0; i < e; ++i) r *= b;
    return r;
}
// raytracing_scene.rs:104-105 and :279-280 / :363-364
__device__ __forceinline__ bool is_special(uint32_t prim_flags) { return (prim_flags & PRIM_EMISSIVE) || !(prim_flags & PRIM_ILLUM_TERMINAL); }
__device__ __forceinline__ uint32_t gi_rays(const SceneDev &sc, uint32_t depth) {
    uint32_t n = sc.opts.max_illumination_rays / ipow(4u, depth);
    return n < 1u ? 1u : n;
}
__device__ __forceinline__ uint32_t refl_rays(const SceneDev &sc, uint32_t depth) {
    uint32_t n = sc.opts.max_reflected_rays / ipow(8u, depth);
    return n < 1u ? 1u : n;
}

struct Surface {
    d3 P, n;      // world hit point, shading normal
    d3 ohp, on;   // object-space hit point / normal (for uv)
};

// Intersection::compute_data (ray_intersection.rs:86-117) + surface_normal of each primitive
__device__ __forceinline__ Surface compute_surface(const SceneDev &sc, const ShadePrim &sh, uint32_t side, d3 o, d3 d,
                                                   double t, double bu, double bv) {
    Surface s;
    const double *inv = sc.inv_world + 12ull * sh.xf;
    s.P = o + d * t;
    s.ohp = xf_point(inv, s.P);
    if (sh.type == RT_PRIM_SPHERE) {
        s.on = unit3(s.ohp);
    } else if (sh.type == RT_PRIM_CUBE) {  // cube.rs:188-210, face code = axis + 4*positive
        const int code = static_cast<int>(bu);
        const int axis = code & 3;
        s.on = mk3(axis == 0 ? 1.0 : 0.0, axis == 1 ? 1.0 : 0.0, axis == 2 ? 1.0 : 0.0);
        if (!(code & 4)) s.on = -s.on;
    } else if (sh.type == RT_PRIM_PLANE) {
        s.on = mk3(sh.normal[0], sh.normal[1], sh.normal[2]);
    } else {
        const double *n = sc.tri_nrm + 9ull * sh.tri;
        double bw = 1.0 - bu - bv;
        s.on = unit3(bw * mk3(n[0], n[1], n[2]) + bu * mk3(n[3], n[4], n[5]) + bv * mk3(n[6], n[7], n[8]));
    }
    d3 n = unit3(xf_vector_t(inv, s.on));
    if (side == RT_SIDE_BOTH) {
        if (dot3(n, d) > 0.0) n = -n;
    } else if (side == RT_SIDE_BACK) {
        n = -n;
    }
    s.n = n;
    return s;
}

// uv of each primitive + Texture::get_color (texture.rs:46-69) + Material::get_color
__device__ __forceinline__ d3 material_color(const SceneDev &sc, const ShadePrim &sh, const rt_material &m,
                                             const Surface &s, double bu, double bv) {
    d3 c = mk3(m.color[0], m.color[1], m.color[2]);
    if (m.texture_id < 0) return c;
    double u, v;
    if (sh.type == RT_PRIM_SPHERE) {
        d3 hp = s.ohp / sh.param;
        u = atan2(hp.x, hp.z) * INV_PI_D * 0.5 + 0.5;
        v = asin(hp.y) * INV_PI_D + 0.5;
    } else if (sh.type == RT_PRIM_CUBE) {
        d3 hp = mk3(s.ohp.x / sh.param + 0.5, s.ohp.y / sh.param + 0.5, s.ohp.z / sh.param + 0.5);
        int code = static_cast<int>(bu);
        int axis = code & 3;
        if (code & 4) {
            if (axis == 0) { u = -hp.z; v = hp.y; }
            else if (axis == 1) { u = hp.x; v = -hp.z; }
            else { u = hp.x; v = hp.y; }
        } else {
            if (axis == 0) { u = hp.z; v = hp.y; }
            else if (axis == 1) { u = hp.x; v = hp.z; }
            else { u = -hp.x; v = hp.y; }
        }
    } else if (sh.type == RT_PRIM_PLANE) {
        if (sh.uv_ok) {
            u = sh.uv_rot[0] * s.ohp.x + sh.uv_rot[1] * s.ohp.y + sh.uv_rot[2] * s.ohp.z;
            v = sh.uv_rot[3] * s.ohp.x + sh.uv_rot[4] * s.ohp.y + sh.uv_rot[5] * s.ohp.z;
        } else {
            u = v = __builtin_nan("");
        }
    } else {
        const double *t = sc.tri_uv + 6ull * sh.tri;
        double bw = 1.0 - bu - bv;
        u = bw * t[0] + bu * t[2] + bv * t[4];
        v = bw * t[1] + bu * t[3] + bv * t[5];
    }
    const DTexture tx = sc.textures[m.texture_id];
    uint32_t w = tx.width - 1u, h = tx.height - 1u;
    double x = fmod(u, 1.0), y = fmod(v, 1.0);
    x = x < 0.0 ? x + 1.0 : x;
    y = y < 0.0 ? y + 1.0 : y;
    x = x * static_cast<double>(w);
    y = (1.0 - y) * static_cast<double>(h);
    uint32_t xi = min(cast_u32(x), w), yi = min(cast_u32(y), h);
    const uint8_t *px = sc.texels + tx.offset + (static_cast<uint64_t>(yi) * tx.width + xi) * 3ull;
    d3 texel = mk3(static_cast<double>(px[0]) / 255.0, static_cast<double>(px[1]) / 255.0,
                   static_cast<double>(px[2]) / 255.0);
    return hadamard(c, texel);
}

// PointLight::get_color (lights/point.rs:32-34)
__device__ __forceinline__ d3 light_radiance(const DLight &l, double dist) {
    d3 c = mk3(l.intensity * l.color[0], l.intensity * l.color[1], l.intensity * l.color[2]) / (dist * dist);
    return sat01(c);
}

// ------------------------------------------------------------------ batch bookkeeping
__global__ void k_begin_batch(Counters *ctrs, Shards sh, uint32_t n_items) {
    const uint32_t k = threadIdx.x;
    if (blockIdx.x != 0 || k >= sh.count) return;
    Counters *ctr = ctrs + k;
    const uint32_t n_gran = (n_items + sh.granule - 1) / sh.granule;
    const uint32_t local = (n_gran / sh.count + (k < n_gran % sh.count ? 1u : 0u)) * sh.granule;
    for (uint32_t i = 0; i <= MAX_LEVELS; ++i) {
        ctr->level_start[i] = 0;
        ctr->level_count[i] = 0;
    }
    for (uint32_t i = 0; i <= MAX_LEVELS; ++i) ctr->live_colour[i] = ctr->live_illum[i] = 0;
    ctr->level_count[0] = local;  // items past n_items are padding (META_INACTIVE)
    ctr->node_alloc = local;
    ctr->shadow_rays = 0;
    ctr->job_alloc = 0;
    ctr->special_alloc = 0;
    for (uint32_t i = 0; i <= MAX_LEVELS; ++i) ctr->job_level_start[i] = ctr->child_end[i] = 0;
    ctr->gi_rays = ctr->gi_materialised = ctr->illum_last_rays = 0;
    ctr->overflow = local > sh.node_cap ? 1u : 0u;
    ctr->active_camera = 0;
    ctr->nodes_visited_closest = ctr->prims_tested_closest = 0;
    ctr->nodes_visited_shadow = ctr->prims_tested_shadow = 0;
    for (uint32_t l = 0; l < MAX_LEVELS; ++l)
        for (uint32_t q = 0; q < 4; ++q) ctr->census[l][q] = 0;
    ctr->wave_node_steps = ctr->wave_leaf_rounds = ctr->wave_iterations = ctr->wave_max_steps = 0;
    for (uint32_t q = 0; q < 16; ++q) ctr->leaf_census[q] = 0;
    for (uint32_t q = 0; q < 4; ++q) ctr->leaf_lanes[q] = 0;
}

// After k_expand(level): what it emitted is final -- its children [.., child_end[level + 1]) and its jobs
// [job_level_start[level], job_level_start[level + 1]) -- while the level's nodes as a whole are not (k_trace_gi(level + 1)
// appends the GI rays that need nodes; k_advance_level closes the level after it).  Lets the closest-hit launch over the
// children and the shadow launch over the jobs run beside k_trace_gi on streams of their own.
__global__ void k_close_expand(Counters *ctrs, Shards sh, uint32_t level) {
    const uint32_t k = threadIdx.x;
    if (blockIdx.x != 0 || k >= sh.count) return;
    Counters *ctr = ctrs + k;
    if (ctr->node_alloc > sh.node_cap || ctr->job_alloc > sh.job_cap) ctr->overflow = 1;
    ctr->child_end[level + 1] = static_cast<uint32_t>(ctr->node_alloc);
    ctr->job_level_start[level + 1] = static_cast<uint32_t>(ctr->job_alloc);
}

__global__ void k_advance_level(Counters *ctrs, Shards sh, uint32_t level) {
    const uint32_t k = threadIdx.x;
    if (blockIdx.x != 0 || k >= sh.count) return;
    Counters *ctr = ctrs + k;
    uint32_t start = ctr->level_start[level] + ctr->level_count[level];
    if (ctr->node_alloc > sh.node_cap || ctr->job_alloc > sh.job_cap || ctr->special_alloc > sh.special_cap)
        ctr->overflow = 1;
    ctr->job_level_start[level + 1] = static_cast<uint32_t>(ctr->job_alloc);  // k_expand(level + 1) appends from here
    ctr->level_start[level + 1] = start;
    ctr->level_count[level + 1] = ctr->overflow ? 0u : static_cast<uint32_t>(ctr->node_alloc) - start;
}

// Live lists of a level, after its closest-hit launch: the nodes k_expand / k_resolve still have to visit (not
// finished by the trace kernel, not padding; 1 in 5) go to the level's own index range of NodePool::live -- colour
// nodes from its start upwards, illumination nodes from its end downwards.  Those kernels then read full groups of 64
// live nodes; before, each of them scanned the meta words of every node of the level (three scans of 108 M words a
// frame, wave-serial).  A wave takes four consecutive groups at a time and bumps the counters once per shard and kind
// (a returning atomic is microseconds; the scan itself is one coalesced load per group).
constexpr uint32_t LIVE_GROUPS = 4;
// N consecutive groups of shard k, `n_entries` nodes from pool index `first_node` on: one atomic per kind for all of them.
template <uint32_t N>
__device__ __forceinline__ void live_scan(const NodePool &np, Counters *ctrs, uint32_t level, uint32_t k, uint32_t lo, uint32_t hi,
                                          uint32_t first_node, uint32_t n_entries, uint32_t lane) {
    unsigned long long m_c[N], m_i[N];
    uint32_t tot_c = 0, tot_i = 0;
#pragma unroll
    for (uint32_t j = 0; j < N; ++j) {
        const uint32_t meta = j * 64u + lane < n_entries ? np.meta[first_node + j * 64u + lane] : META_INACTIVE;
        const bool live = !(meta & (META_INACTIVE | META_DONE));
        m_c[j] = __ballot(live && !(meta & META_ILLUM));
        m_i[j] = __ballot(live && (meta & META_ILLUM));
        tot_c += static_cast<uint32_t>(__popcll(m_c[j]));
        tot_i += static_cast<uint32_t>(__popcll(m_i[j]));
    }
    uint32_t b_c = 0, b_i = 0;
    if (lane == 0) {
        if (tot_c) b_c = atomicAdd(&ctrs[k].live_colour[level], tot_c);
        if (tot_i) b_i = atomicAdd(&ctrs[k].live_illum[level], tot_i);
    }
    b_c = __builtin_amdgcn_readfirstlane(b_c);
    b_i = __builtin_amdgcn_readfirstlane(b_i);
#pragma unroll
    for (uint32_t j = 0; j < N; ++j) {
        const uint32_t node = first_node + j * 64u + lane;
        const uint32_t r_c = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m_c[j] >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m_c[j]), 0u));
        const uint32_t r_i = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m_i[j] >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m_i[j]), 0u));
        if ((m_c[j] >> lane) & 1ull) np.live[lo + b_c + r_c] = node;
        if ((m_i[j] >> lane) & 1ull) np.live[hi - 1u - (b_i + r_i)] = node;
        b_c += static_cast<uint32_t>(__popcll(m_c[j]));
        b_i += static_cast<uint32_t>(__popcll(m_i[j]));
    }
}
__global__ __launch_bounds__(BLOCK) void k_build_live(NodePool np, Counters *ctrs, Shards shards, uint32_t level) {
    __shared__ LevelMap map;
    if (level_map_build(map, ctrs, shards, level, MAP_NODES) == 0) return;
    // Chunks of LIVE_GROUPS groups are dealt shard by shard (chunk c: shard c % K, the shard's c / K-th chunk), so the
    // waves running at any moment bump the counters of all shards, not the one or two whose groups are next in the
    // level's list: a counter word takes ~90 returning atomics a microsecond.
    const uint32_t K = shards.count, per_chunk = 64u * LIVE_GROUPS;
    uint32_t max_chunks = 0;
    for (uint32_t k = 0; k < K; ++k) max_chunks = max(max_chunks, (map.count[k] + per_chunk - 1u) / per_chunk);
    const uint32_t lane = threadIdx.x & 63u, total = max_chunks * K;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6));
    for (uint32_t c = wave; c < total; c += gridDim.x * (BLOCK / 64)) {
        const uint32_t k = c % K, first = (c / K) * per_chunk, count = map.count[k];
        if (first >= count) continue;
        const uint32_t lo = map.base[k];
        live_scan<LIVE_GROUPS>(np, ctrs, level, k, lo, lo + count, lo + first, count - first, lane);
    }
}

// ------------------------------------------------------------------ ray generation
__global__ __launch_bounds__(BLOCK) void k_raygen(SceneDev sc, WorkList wl, NodePool np, Counters *ctrs, Shards shards,
                                                 uint32_t item0, uint32_t n_items) {
    const ShardCtx c = shard_ctx(ctrs, shards);
    Counters *ctr = c.ctr;
    if (ctr->overflow) return;
    const uint32_t n_local = ctr->level_count[0];
    const double width = static_cast<double>(sc.opts.width), height = static_cast<double>(sc.opts.height);
    const double aspect = width / height;
    uint32_t n_active = 0;
    for (uint32_t j = c.bid * BLOCK + threadIdx.x; j < n_local; j += c.nblocks * BLOCK) {
        const uint32_t i = c.node_off + j;
        const uint32_t rel = shard_item(shards, c.shard, j);
        uint32_t item = item0 + rel, px = 0, py = 0, sample = 0;
        bool active = rel < n_items;
        if (active) {
            item_to_pixel(wl, item, &px, &py, &sample);
            active = px >= wl.x0 && px < wl.x1 && py >= wl.y0 && py < wl.y1;
        }
        np.root[i] = item;
        np.path[i] = 0ull;
        np.meta[i] = active ? 0u : META_INACTIVE;
        if (!active) continue;
        n_active++;
        double fx, fy;
        if (sample == 0) {  // raytracing_scene.rs:560
            fx = static_cast<double>(px) + 0.5;
            fy = static_cast<double>(py) + 0.5;
        } else {            // :562-567, thread_rng replaced by the counter-based stream
            double u1, u2;
            rng_pair(wl.seed, py * sc.opts.width + px, sample, 0ull, &u1, &u2);
            fx = static_cast<double>(px) + u1;
            fy = static_cast<double>(py) + u2;
        }
        // remap_value (utils/mod.rs:29-33): (num - d0) * (r1 - r0) / (d1 - d0) + r0
        double nx = (fx - 0.0) * (1.0 - -1.0) / (width - 0.0) + -1.0;
        double ny = (fy - 0.0) * (-1.0 - 1.0) / (height - 0.0) + 1.0;
        if (width < height) nx = nx * aspect;
        else ny = ny / aspect;
        nx = nx * sc.fov_scale;
        ny = ny * sc.fov_scale;
        d3 dir = unit3(mk3(nx, ny, -1.0));
        const double *m = sc.cam_to_world;
        d3 wd = mk3(m[0] * dir.x + m[1] * dir.y + m[2] * dir.z, m[3] * dir.x + m[4] * dir.y + m[5] * dir.z,
                    m[6] * dir.x + m[7] * dir.y + m[8] * dir.z);
        store_ray(np, i, mk3(sc.cam_pos[0], sc.cam_pos[1], sc.cam_pos[2]), wd);
    }
    if (n_active) atomicAdd(&ctr->active_camera, n_active);
}

// ------------------------------------------------------------------ traversal
// One ray per lane.  Per-lane short stack in LDS, `stack_depth` entries of
// BLOCK words each (entry e of thread t at stack[e * BLOCK + t]: conflict-free);
// the host sizes it from the BVH (max leaf depth + 2), so small scenes keep
// LDS use low and occupancy high.  Entry 0 holds a sentinel.
//
// "while-while" form: every lane first descends through interior nodes until it
// holds a leaf (or is done), then the wave tests leaves together -- a diverged
// wave does not pay for the node path and the leaf path in every iteration.
//
// Leaf primitive = BoundedObject::intersect (bounds.rs:172-181): the primitive's
// own world-AABB slab test AND the object-space primitive test.  Both are pure
// predicates, so they are evaluated cheapest-first: the builder makes
// single-primitive leaves, whose box in the parent node IS the primitive's own
// AABB, so the conservative node test has already pre-filtered it (no divisions,
// running t limit applied); then the primitive test, and only for primitive
// hits the reference-order AABB arithmetic (6 divisions).
#ifndef EXPAND_WAVES
#define EXPAND_WAVES 4
#endif
#ifndef RESOLVE_ILLUM_WAVES
#define RESOLVE_ILLUM_WAVES 4
#endif
#ifndef RESOLVE_WAVES
#define RESOLVE_WAVES 2
#endif
#ifndef TRACE_WAVES
#define TRACE_WAVES 4
#endif
constexpr uint32_t STACK_SENTINEL = 0x7FFFFFFFu;  // not a leaf ref, not a node index

struct ObjRayCache {
    uint32_t xf;
    RayD r;
};

// The record's geometry half (80 bytes: five 16-byte loads of one cache line, meta and id included) is requested at
// once before anything is decided: asked for piecemeal -- meta, then the geometry the type needs, then the id of an
// accepted hit -- a leaf costs dependent trips to the vector cache.  The primitive's own box (the line's first 48
// bytes) is fetched only for primitive hits (3-4 % of the tests on meshes) and hits L1: holding it from the start cost
// 12 VGPRs at the kernels' register peak -- spills in every traversal kernel at 128 VGPRs (k_trace_gi 8.6 -> 7.7 ms
// on scene.json without them).
template <bool SHADOW>
__device__ __forceinline__ bool test_leaf_prim(const SceneDev &sc, const LeafPrim *__restrict__ lp, const RayD &ray,
                                               ObjRayCache &cache, double max_distance, HitRec *h, uint32_t *prim_id,
                                               uint32_t *prim_type = nullptr, uint32_t *exact_box = nullptr) {
    const double2 *q = reinterpret_cast<const double2 *>(lp);
    const double2 q3 = q[3], q4 = q[4], q5 = q[5], q6 = q[6];
    const uint4 tail = reinterpret_cast<const uint4 *>(lp)[7];  // geom[8] | prim_id | meta
    const uint32_t meta = tail.w;
    *prim_id = tail.z;
    const uint32_t xf = meta >> 4;
    if (xf != cache.xf) {  // Ray::transform (ray_intersection.rs:33-43), shared by a mesh's faces
        const double *inv = sc.inv_world + 12ull * xf;
        cache.r.o = xf_point(inv, ray.o);
        cache.r.d = xf_vector(inv, ray.d);
        cache.xf = xf;
    }
    const uint32_t type = meta & 3u, side = (meta >> 2) & 3u;
    if (prim_type) *prim_type = type;
    const double geom[9] = {q3.x, q3.y, q4.x, q4.y, q5.x, q5.y, q6.x, q6.y,
                            __hiloint2double(static_cast<int>(tail.y), static_cast<int>(tail.x))};
    double t_probe;
    if (!hit_bounded<SHADOW>(geom, type, side, cache.r, max_distance, h, &t_probe)) return false;
    const double2 q0 = q[0], q1 = q[1], q2 = q[2];  // the primitive's own box: the same cache line again, for primitive hits only
    const double center[3] = {q0.x, q0.y, q1.x}, half[3] = {q1.y, q2.x, q2.y};
    if (prim_box_surely_passes(center, half, ray, t_probe, SHADOW, max_distance)) return true;
    if (exact_box) *exact_box = 1u;
    return prim_box_test(center, half, ray, SHADOW, max_distance);
}

__device__ __forceinline__ double cull_limit(double t) { return t + 1e-9 * (1.0 + fabs(t)); }

// The block's copy of the top of the tree (nodes [0, n_top_nodes), breadth-first
// order) sits in LDS behind the stacks: every ray starts there and most node
// visits never leave it, which takes them off the vector-memory path (two
// divergent 16-byte loads per visit otherwise).  The copy holds the twelve box
// coordinates ALREADY converted to f32 (64 bytes a node: lmin lmax rmin rmax |
// left right - -): the conversion of a 16-bit coordinate is exact, so a visit
// through the copy decides exactly what a visit through memory decides, without
// the twelve v_cvt a visit (a fifth of the node loop's instructions).
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const f32x4 LdsNodeWord;
constexpr uint32_t TOP_NODE_BYTES = 64;

__device__ __forceinline__ void stage_tree_top(const BvhNode *__restrict__ nodes, uint32_t n, float4 *dst) {
    for (uint32_t i = threadIdx.x; i < n; i += BLOCK) {
        const uint4 w0 = reinterpret_cast<const uint4 *>(nodes + i)[0], w1 = reinterpret_cast<const uint4 *>(nodes + i)[1];
        dst[4u * i + 0u] = make_float4(static_cast<float>(w0.x & 0xFFFFu), static_cast<float>(w0.x >> 16), static_cast<float>(w0.y & 0xFFFFu),
                                       static_cast<float>(w0.y >> 16));
        dst[4u * i + 1u] = make_float4(static_cast<float>(w0.z & 0xFFFFu), static_cast<float>(w0.z >> 16), static_cast<float>(w0.w & 0xFFFFu),
                                       static_cast<float>(w0.w >> 16));
        dst[4u * i + 2u] = make_float4(static_cast<float>(w1.x & 0xFFFFu), static_cast<float>(w1.x >> 16), static_cast<float>(w1.y & 0xFFFFu),
                                       static_cast<float>(w1.y >> 16));
        dst[4u * i + 3u] = make_float4(__uint_as_float(w1.z), __uint_as_float(w1.w), 0.0f, 0.0f);
    }
}

__device__ __forceinline__ LdsNodeWord *stage_top_nodes(const SceneDev &sc, uint32_t *stack_lds, uint32_t stack_entries) {
    float4 *top = reinterpret_cast<float4 *>(stack_lds + stack_entries * BLOCK);
    stage_tree_top(sc.nodes, sc.n_top_nodes, top);
    uint32_t at = sc.n_top_nodes;  // the trees of the special primitives behind it
    for (uint32_t q = 0; q < 2; ++q) {
        stage_tree_top(sc.special[q].nodes, sc.special[q].n_top, top + 4u * at);
        at += sc.special[q].n_top;
    }
    __syncthreads();
    return (LdsNodeWord *)top;
}

// One of the scene's trees: all bounded primitives, or one of the two sets of special ones (SceneDev::special).
struct TreeView {
    const BvhNode *nodes;
    const LeafPrim *leaf_prims;
    LdsNodeWord *top;  // LDS copy of nodes [0, n_top), TOP_NODE_BYTES each
    uint32_t n_top;
    uint32_t root_ref;
};
__device__ __forceinline__ TreeView main_tree(const SceneDev &sc, LdsNodeWord *top_nodes) {
    return TreeView{sc.nodes, sc.leaf_prims, top_nodes, sc.n_top_nodes, sc.root_ref};
}
__device__ __forceinline__ TreeView special_tree(const SceneDev &sc, LdsNodeWord *top_nodes, uint32_t set) {
    const SceneDev::SpecialSet &sp = sc.special[set];
    return TreeView{sp.nodes, sp.leaf_prims, top_nodes + 4u * (sc.n_top_nodes + (set ? sc.special[0].n_top : 0u)), sp.n_top, sp.root_ref};
}

// Descend from `cur` through interior nodes until a leaf ref or the sentinel.
template <bool COUNT>
__device__ __forceinline__ uint32_t descend(const TreeView &tv, const CullRay &cr, float limit, uint32_t cur,
                                            uint32_t *stack, uint32_t &sp, unsigned long long &n_nodes) {
    while (!(cur & LEAF_FLAG) && cur != STACK_SENTINEL) {
        float lmin[3], lmax[3], rmin[3], rmax[3];
        uint32_t lref, rref;
        if (cur < tv.n_top) {  // the block's copy: coordinates held as f32
            LdsNodeWord *nd = tv.top + 4u * cur;
            const f32x4 a0 = nd[0], a1 = nd[1], a2 = nd[2];
            const f32x2 a3 = *(__attribute__((address_space(3))) const f32x2 *)(nd + 3);
            lmin[0] = a0.x, lmin[1] = a0.y, lmin[2] = a0.z, lmax[0] = a0.w, lmax[1] = a1.x, lmax[2] = a1.y;
            rmin[0] = a1.z, rmin[1] = a1.w, rmin[2] = a2.x, rmax[0] = a2.y, rmax[1] = a2.z, rmax[2] = a2.w;
            lref = __float_as_uint(a3.x), rref = __float_as_uint(a3.y);
        } else {  // the 32-byte node in memory (lmin lmax rmin | rmax left right): a pair of 16-byte loads
            const uint4 *nd = reinterpret_cast<const uint4 *>(tv.nodes + cur);
            const uint4 w0 = nd[0], w1 = nd[1];
            lmin[0] = static_cast<float>(static_cast<uint16_t>(w0.x)), lmin[1] = static_cast<float>(static_cast<uint16_t>(w0.x >> 16));
            lmin[2] = static_cast<float>(static_cast<uint16_t>(w0.y)), lmax[0] = static_cast<float>(static_cast<uint16_t>(w0.y >> 16));
            lmax[1] = static_cast<float>(static_cast<uint16_t>(w0.z)), lmax[2] = static_cast<float>(static_cast<uint16_t>(w0.z >> 16));
            rmin[0] = static_cast<float>(static_cast<uint16_t>(w0.w)), rmin[1] = static_cast<float>(static_cast<uint16_t>(w0.w >> 16));
            rmin[2] = static_cast<float>(static_cast<uint16_t>(w1.x)), rmax[0] = static_cast<float>(static_cast<uint16_t>(w1.x >> 16));
            rmax[1] = static_cast<float>(static_cast<uint16_t>(w1.y)), rmax[2] = static_cast<float>(static_cast<uint16_t>(w1.y >> 16));
            lref = w1.z, rref = w1.w;
        }
        if (COUNT) {  // low 40 bits: lane-level visits; high bits: wave-level visits (counted by the first active lane)
            n_nodes++;
            if ((threadIdx.x & 63u) == static_cast<uint32_t>(__ffsll(__ballot(true)) - 1)) n_nodes += 1ull << 40;
        }
        float tl, tr;
        const bool hl = box_cull_test(lmin, lmax, cr, limit, &tl);
        const bool hr = box_cull_test(rmin, rmax, cr, limit, &tr);
        if (hl && hr) {
            const bool left_first = !(tr < tl);
            stack[(sp++) * BLOCK] = left_first ? rref : lref;
            cur = left_first ? lref : rref;
        } else if (hl) {
            cur = lref;
        } else if (hr) {
            cur = rref;
        } else {
            cur = stack[(--sp) * BLOCK];
        }
    }
    return cur;
}

// Order of two hits at EXACTLY the same distance (the smaller key comes first).  The reference takes the first minimum
// in iteration order (bounds.rs:303-307, :344-353, :359-362): unbounded objects in scene order, then the tree's hit --
// so an unbounded object precedes a bounded one and the earlier of two unbounded ones wins, which the key keeps.  Among
// bounded primitives its order is the kd-tree's traversal order, which no other accelerator has; the key puts the
// primitive with the HIGHER id first: on the one shipped scene with exact ties (cornell_box: the light's two triangles,
// ids 26 / 27, coplanar with the ceiling's, 20 / 21 -- quirk Q17) that is what the reference's own tree and order give on
// all 2,386 tied pixel centres of the 800x800 frame (tests/test_gpu_parity.py, the CPU restatement of its tree; with the lower id first they all differed and the
// frame's radiance was 5e-2 rms off, the light's emission lost to the ceiling wherever a GI ray met the tie).
__device__ __forceinline__ uint32_t tie_key_bounded(uint32_t prim_id) { return 0x80000000u | (0x7FFFFFFFu - prim_id); }

// KdTreeAccelerator::raycast (bounds.rs:302-308) for one ray, in the traversal-order independent form: the unbounded
// objects, then the BVH; minimum of (t, tie key).  Shared by k_trace_closest and k_trace_gi.
// SPECIAL: only the special primitives (PRIM_* flags: emissive, or an illumination hit needs child rays) -- the planes
// among them and the tree `tv` of the bounded ones; `key_out` then gets the rest of the winner's ordering key
// (tie_key_bounded of a bounded winner, the plane list index of an unbounded one).  `shared_cr`: the ray's CullRay if the caller has it.
template <bool COUNT, bool SPECIAL>
__device__ __forceinline__ void closest_hit(const SceneDev &sc, const TreeView &tv, uint32_t *stack, const RayD &ray, Counters *ctr,
                                            unsigned long long &n_nodes, unsigned long long &n_prims, uint32_t *id_out, double *t_out,
                                            double *bu_out, double *bv_out, const CullRay *shared_cr = nullptr, uint32_t *key_out = nullptr,
                                            bool emissive_only = false) {
    double best_t = __builtin_inf();
    uint32_t best_id = RT_NO_HIT;
    double best_bu = 0.0, best_bv = 0.0;
    bool best_bounded = false;
    uint32_t best_plane = 0;

    for (uint32_t p = 0; p < sc.n_planes; ++p) {  // unbounded objects first (bounds.rs:303-305)
        const PlanePrim pl = sc.planes[p];
        if (SPECIAL && !(emissive_only ? (sc.prim_flags[pl.prim_id] & PRIM_EMISSIVE) != 0 : is_special(sc.prim_flags[pl.prim_id]))) continue;
        const double *inv = sc.inv_world + 12ull * (pl.meta >> 4);
        RayD orr;
        orr.o = xf_point(inv, ray.o);
        orr.d = xf_vector(inv, ray.d);
        HitRec h;
        if (COUNT) n_prims++;
        if (hit_plane(mk3(pl.normal[0], pl.normal[1], pl.normal[2]), (pl.meta >> 2) & 3u, false, orr, false, 0.0,
                      &h) &&
            (best_id == RT_NO_HIT || h.t < best_t)) {
            best_t = h.t;
            best_id = pl.prim_id;
            best_plane = p;
        }
    }

    if (tv.root_ref != 0xFFFFFFFFu) {
        const CullRay cr = shared_cr ? *shared_cr : make_cull_ray(ray, sc.scene_scale, sc.q_origin, sc.q_scale);
        ObjRayCache cache;
        cache.xf = 0xFFFFFFFFu;
        stack[0] = STACK_SENTINEL;
        uint32_t sp = 1;
        uint32_t cur = tv.root_ref;
        while (true) {
            cur = descend<COUNT>(tv, cr, cull_limit_f32(cull_limit(best_t)), cur, stack, sp, n_nodes);
            if (cur == STACK_SENTINEL) break;
            const uint32_t first = cur & 0x0FFFFFFFu, n = ((cur >> 28) & 7u) + 1u;
            if (COUNT && (threadIdx.x & 63u) == static_cast<uint32_t>(__ffsll(__ballot(true)) - 1)) n_prims += 1ull << 40;  // wave-level leaf round
            for (uint32_t k = 0; k < n; ++k) {
                const LeafPrim *lp = tv.leaf_prims + first + k;
                HitRec h;
                if (COUNT) n_prims++;
                uint32_t exact_box = 0, pid, ty = 0;
                const bool leaf_hit = test_leaf_prim<false>(sc, lp, ray, cache, 0.0, &h, &pid, COUNT ? &ty : nullptr, COUNT ? &exact_box : nullptr);
                if (COUNT) {  // census of the wave's leaf round: which primitive types are present, how many lanes, exact box tests
                    const unsigned long long m_all = __ballot(true), m_t = __ballot(ty == RT_PRIM_TRIANGLE), m_s = __ballot(ty == RT_PRIM_SPHERE),
                                             m_c = __ballot(ty == RT_PRIM_CUBE), m_x = __ballot(exact_box != 0), m_h = __ballot(leaf_hit);
                    if ((threadIdx.x & 63u) == static_cast<uint32_t>(__ffsll(m_all) - 1)) {
                        atomicAdd(&ctr->leaf_census[(m_t ? 1 : 0) | (m_s ? 2 : 0) | (m_c ? 4 : 0) | (m_x ? 8 : 0)], 1ull);
                        atomicAdd(&ctr->leaf_lanes[0], static_cast<unsigned long long>(__popcll(m_all)));
                        atomicAdd(&ctr->leaf_lanes[1], static_cast<unsigned long long>(__popcll(m_x)));
                        atomicAdd(&ctr->leaf_lanes[2], static_cast<unsigned long long>(__popcll(m_h)));
                    }
                }
                if (leaf_hit) {
                    bool better = best_id == RT_NO_HIT || h.t < best_t || (h.t == best_t && best_bounded && pid > best_id);
                    if (better) {
                        best_t = h.t;
                        best_id = pid;
                        best_bu = h.bu;
                        best_bv = h.bv;
                        best_bounded = true;
                    }
                }
            }
            cur = stack[(--sp) * BLOCK];
        }
    }
    *id_out = best_id;
    *t_out = best_t;
    *bu_out = best_bu;
    *bv_out = best_bv;
    if (key_out) *key_out = best_bounded ? tie_key_bounded(best_id) : best_plane;
}

// The second half of a two-phase GI query (k_trace_gi): given the closest SPECIAL hit S of the ray (s_exists, s_t,
// s_key as closest_hit<., true> returns them), decide what the closest hit over ALL primitives would be without
// finding it: any primitive that precedes S in the order (t, tie key) makes the closest hit a plain one, and of
// a plain hit get_illumination (:492-520) uses nothing but `distance <= max_occlusion_distance` (D).  So the search
// stops at the first preceding hit within D, and once a preceding hit beyond D is known only hits within D matter.
// A special primitive met again here never precedes S (S is the first of them; the same record gives the same t).
// Returns bit 0: a preceding hit exists (the closest hit is plain); bit 1: the closest hit lies within D.
template <bool COUNT>
__device__ __forceinline__ uint32_t any_hit_before(const SceneDev &sc, const TreeView &tv, uint32_t *stack, const RayD &ray, const CullRay &cr,
                                                   unsigned long long &n_nodes, unsigned long long &n_prims, bool s_exists, double s_t,
                                                   uint32_t s_key, double D) {
    bool found = false, within = false;
    double limit = s_exists ? s_t : D;  // hits beyond it change nothing
    for (uint32_t p = 0; p < sc.n_planes && !within; ++p) {
        const PlanePrim pl = sc.planes[p];
        const double *inv = sc.inv_world + 12ull * (pl.meta >> 4);
        RayD orr;
        orr.o = xf_point(inv, ray.o);
        orr.d = xf_vector(inv, ray.d);
        HitRec h;
        if (COUNT) n_prims++;
        if (!hit_plane(mk3(pl.normal[0], pl.normal[1], pl.normal[2]), (pl.meta >> 2) & 3u, false, orr, false, 0.0, &h)) continue;
        if (s_exists && !(h.t < s_t || (h.t == s_t && p < s_key))) continue;  // an unbounded S has key = its list index; a bounded one tie_key_bounded(id)
        if (h.t <= D) within = true;
        if (s_exists || h.t <= D) found = true;
        if (!within) limit = D;
    }
    if (!within && tv.root_ref != 0xFFFFFFFFu) {
        ObjRayCache cache;
        cache.xf = 0xFFFFFFFFu;
        stack[0] = STACK_SENTINEL;
        uint32_t sp = 1;
        uint32_t cur = tv.root_ref;
        while (true) {
            cur = descend<COUNT>(tv, cr, cull_limit_f32(cull_limit(limit)), cur, stack, sp, n_nodes);
            if (cur == STACK_SENTINEL) break;
            const uint32_t first = cur & 0x0FFFFFFFu, n = ((cur >> 28) & 7u) + 1u;
            for (uint32_t k = 0; k < n && !within; ++k) {
                HitRec h;
                if (COUNT) n_prims++;
                uint32_t pid;
                if (!test_leaf_prim<false>(sc, tv.leaf_prims + first + k, ray, cache, 0.0, &h, &pid)) continue;
                if (s_exists && !(h.t < s_t || (h.t == s_t && tie_key_bounded(pid) < s_key))) continue;
                if (h.t <= D) within = true;
                if (s_exists || h.t <= D) found = true;
                if (!within) limit = D;
            }
            if (within) break;
            cur = stack[(--sp) * BLOCK];
        }
    }
    return (found ? 1u : 0u) | (within ? 2u : 0u);
}

// KdTreeAccelerator::raycast semantics, traversal-order independent form
// (DESIGN.md "canonical hit"): minimum of (t, tie key: unbounded first, then the higher primitive id).
//
// FINALIZE (render path): nodes whose result needs neither child rays nor
// shading are finished here -- misses (zero), and get_illumination hits on a
// surface with nothing to gather (emissive colour + occlusion flag, :492-520
// with :135-167 / :263-337).  They get META_DONE and their (t,bu,bv,prim) hold
// the resolved (RGB, flag); k_expand / k_resolve skip them.  On scene.json that
// is 80 % of all nodes.
template <bool COUNT, bool FINALIZE>
__global__ __launch_bounds__(BLOCK, TRACE_WAVES) void k_trace_closest(SceneDev sc, NodePool np, Counters *ctrs, Shards shards,
                                                        uint32_t level, uint32_t stack_entries, uint32_t skip_illum = 0, uint32_t what = MAP_NODES) {
    extern __shared__ __align__(16) uint32_t stack_lds[];
    uint32_t *stack = stack_lds + threadIdx.x;
    __shared__ LevelMap map;
    Counters *ctr = ctrs + blockIdx.x % shards.count;  // statistics only
    const uint32_t n_groups = level_map_build(map, ctrs, shards, level, what);
    if (blockIdx.x * (BLOCK / 64) >= n_groups) return;  // no group for this block: leave before staging the tree (the small launches of the deep levels)
    LdsNodeWord *top_nodes = stage_top_nodes(sc, stack_lds, stack_entries);
    unsigned long long n_nodes = 0, n_prims = 0;
    for (uint32_t g = blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6); g < n_groups; g += gridDim.x * (BLOCK / 64)) {
        uint32_t n_valid;
        const uint32_t idx = level_map_group(map, shards, g, &n_valid) + (threadIdx.x & 63u);
        if ((threadIdx.x & 63u) >= n_valid) continue;
        const uint32_t meta = np.meta[idx];
        if (meta & META_INACTIVE) {
            np.prim[idx] = RT_NO_HIT;
            continue;
        }
        if (meta & META_TRACED) continue;  // a GI ray k_trace_gi gave a node of its own: hit already stored
        if (skip_illum && (meta & META_ILLUM)) continue;  // the last level's illumination nodes are k_trace_illum_last's
        RayD ray;
        load_ray(np, idx, &ray.o, &ray.d);
        double best_t, best_bu, best_bv;
        uint32_t best_id;
        closest_hit<COUNT, false>(sc, main_tree(sc, top_nodes), stack, ray, ctr, n_nodes, n_prims, &best_id, &best_t, &best_bu, &best_bv);
        if (FINALIZE) {
            bool done = best_id == RT_NO_HIT;
            uint32_t flag = 0;
            d3 rgb = mk3(0.0, 0.0, 0.0);
            if (!done && (meta & META_ILLUM)) {
                const uint32_t pf = sc.prim_flags[best_id];
                if ((pf & PRIM_ILLUM_TERMINAL) || level + 1 >= sc.opts.max_depth) {  // children of the last level return zero
                    done = true;
                    flag = best_t <= sc.opts.max_occlusion_distance ? 1u : 0u;  // :512-516
                    if (pf & PRIM_EMISSIVE) {
                        const rt_material &m = sc.materials[sc.shade_prims[best_id].material];
                        rgb = mk3(m.emissive[0], m.emissive[1], m.emissive[2]);
                    }
                }
            }
            if (done) {  // resolved word: bit 0 = occluded, bit 1 = RGB is not all zero (only then is it stored / read)
                np.meta[idx] = meta | META_DONE;
                if (rgb.x != 0.0 || rgb.y != 0.0 || rgb.z != 0.0) {
                    flag |= 2u;
                    np.t[idx] = rgb.x;
                    np.bu[idx] = rgb.y;
                    np.bv[idx] = rgb.z;
                }
                np.prim[idx] = flag;
                continue;
            }
        }
        np.prim[idx] = best_id;
        np.t[idx] = best_t;
        np.bu[idx] = best_bu;
        np.bv[idx] = best_bv;
    }
    if (COUNT) {
        atomicAdd(&ctr->nodes_visited_closest, n_nodes & ((1ull << 40) - 1ull));
        atomicAdd(&ctr->prims_tested_closest, n_prims & ((1ull << 40) - 1ull));
        unsigned long long ws = n_nodes >> 40, wl = n_prims >> 40;  // wave-level totals
        for (int off = 32; off >= 1; off >>= 1) {
            ws += __shfl_xor(ws, off, 64);
            wl += __shfl_xor(wl, off, 64);
        }
        if ((threadIdx.x & 63u) == 0) {
            atomicAdd(&ctr->wave_node_steps, ws);
            atomicAdd(&ctr->wave_leaf_rounds, wl);
            atomicMax(&ctr->wave_max_steps, ws);
            atomicAdd(&ctr->wave_iterations, 1ull);
        }
    }
}

// compute_global_illumination (:98-133) in place: the G(d) hemisphere rays of every job k_expand(level - 1) left are
// drawn, traced and reduced here; `level` is the depth of the GI rays.  Lanes map to whole jobs (64 / G jobs per wave
// round, lane = job * G + slot), so the per-job reductions are ballots over aligned lane groups:
//   unoccluded     rays whose hit distance is above max_occlusion_distance, and misses (:512-516, :123-125);
//   specials       in slot order, the rays whose emissive contribution may be non-zero: a terminal hit on an emissive
//                  surface (its colour is the material's), or a hit that needs get_emissive_* children (:135-167,
//                  :263-337) -- only that ray gets a node (ray, path, hit; META_TRACED) and joins the level's nodes.
// WAVES: resident waves per SIMD the register budget is cut for -- 4 (128 VGPRs, 2 spilled) or 3 (137, none).  Four are
// faster on every scene measured since the draw became a call and the leaf's box a late load (before: 45 spills at
// four waves, and three were faster wherever the tree sits in LDS / L2); the tuner still tries three (rt_tune.h).
// TWO_PHASE: the level's form of query, the host's choice ((sc.gi_two_phase >> set) & 1 and sc.special[set].ok).
// A GI ray of a job: drawn about the job's normal (the same Philox key and sampler as a stored child would get), :115.
// A call, not inlined: the sampler's constants and temporaries then end at the call's return instead of staying live
// across the traversal that follows (the kernel is cut for 128 VGPRs: 43 of them spilled with the draw inlined, 10
// now), and the four-wave build became the faster one on every scene.
__device__ __noinline__ RayD gi_draw(const GiJob *job, uint64_t seed, uint32_t slot) {
    const d3 P = mk3(job->P[0], job->P[1], job->P[2]), n = mk3(job->n[0], job->n[1], job->n[2]);
    double u1, u2;
    rng_pair(seed, job->pixel, job->sample, path_child(job->path, SLOT_GI + slot), &u1, &u2);
    RayD ray;
    ray.d = sample_direction(SAMPLE_HEMISPHERE, n, 0.0, u1, u2);
    ray.o = P + ray.d * BIAS;  // :115
    return ray;
}
template <bool COUNT, int WAVES, bool TWO_PHASE>
__global__ __launch_bounds__(BLOCK, WAVES) void k_trace_gi(SceneDev sc, WorkList wl, NodePool np, GiPool gp, Counters *ctrs, Shards shards, uint32_t level,
                                                   uint32_t stack_entries) {
    extern __shared__ __align__(16) uint32_t stack_lds[];
    uint32_t *stack = stack_lds + threadIdx.x;
    __shared__ LevelMap map;
    const ShardCtx c = shard_ctx(ctrs, shards);  // the shard this block ALLOCATES nodes and specials from
    Counters *ctr = c.ctr;
    const uint32_t G = gi_rays(sc, level - 1u);                 // rays per job (uniform over the level)
    const uint32_t per_round = 64u / G, used = per_round * G;  // whole jobs per wave round (G <= 64)
    // work map over the jobs of level - 1: "entries" are jobs, a group = per_round jobs
    if (threadIdx.x < 64u) {
        const uint32_t k = threadIdx.x;
        uint32_t cnt = 0, over = 0;
        if (k < shards.count) {
            cnt = static_cast<uint32_t>(ctrs[k].job_alloc) - ctrs[k].job_level_start[level - 1u];
            over = ctrs[k].overflow | (ctrs[k].job_alloc > shards.job_cap ? 1u : 0u);
            map.base[k] = k * shards.job_cap + ctrs[k].job_level_start[level - 1u];
            map.count[k] = cnt;
        }
        uint32_t x = (cnt + per_round - 1u) / per_round;
#pragma unroll
        for (uint32_t off = 1; off < 64; off <<= 1) {
            const uint32_t y = __shfl_up(x, off, 64);
            if (k >= off) x += y;
        }
        if (k < shards.count) map.prefix[k + 1] = x;
        if (k == 0) map.prefix[0] = 0;
        const unsigned long long any_over = __ballot(over != 0);
        if (k == 0) map.dead = any_over ? 1u : 0u;
    }
    __syncthreads();
    const uint32_t n_groups = map.dead ? 0u : map.prefix[shards.count];
    if (blockIdx.x * (BLOCK / 64) >= n_groups) return;
    LdsNodeWord *top_nodes = stage_top_nodes(sc, stack_lds, stack_entries);
    const uint32_t lane = threadIdx.x & 63u;
    const bool next_real = level + 1u < sc.opts.max_depth;  // children of this level's hits are traced (else they return zero, :495-497)
    const uint32_t set = next_real ? 0u : 1u;                // which primitives are special on this level
    constexpr bool two_phase = TWO_PHASE;  // the host's choice for this level: (sc.gi_two_phase >> set) & 1 and sc.special[set].ok
    unsigned long long n_nodes = 0, n_prims = 0;
    uint32_t wave_rays = 0, wave_nodes = 0;  // wave-uniform
    for (uint32_t g = blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6); g < n_groups; g += gridDim.x * (BLOCK / 64)) {
        uint32_t k = 0;
#pragma unroll
        for (uint32_t step = MAX_SHARDS / 2; step >= 1; step >>= 1)
            if (k + step < shards.count && map.prefix[k + step] <= g) k += step;
        const uint32_t first_job = (g - map.prefix[k]) * per_round, jobs_here = min(per_round, map.count[k] - first_job);
        const uint32_t jl = lane / G, slot = lane - jl * G;
        const bool valid = lane < used && jl < jobs_here;
        const uint32_t job_index = map.base[k] + first_job + (valid ? jl : 0u);
        uint32_t hit_id = RT_NO_HIT, plain = 0;  // plain: any_hit_before's answer when the closest hit is a plain primitive
        double hit_t = 0.0, hit_bu = 0.0, hit_bv = 0.0;
        RayD ray;
        if (valid) {
            ray = gi_draw(gp.jobs + job_index, wl.seed, slot);
            if (two_phase) {  // closest special hit, then: does anything precede it / lie within the occlusion distance?
                const CullRay cr = make_cull_ray(ray, sc.scene_scale, sc.q_origin, sc.q_scale);
                uint32_t s_key = 0;
                if (sc.special[set].count)
                    closest_hit<COUNT, true>(sc, special_tree(sc, top_nodes, set), stack, ray, ctr, n_nodes, n_prims, &hit_id, &hit_t, &hit_bu, &hit_bv, &cr, &s_key, set == 1u);
                plain = any_hit_before<COUNT>(sc, main_tree(sc, top_nodes), stack, ray, cr, n_nodes, n_prims, hit_id != RT_NO_HIT, hit_t, s_key,
                                              sc.opts.max_occlusion_distance);
            } else {
                closest_hit<COUNT, false>(sc, main_tree(sc, top_nodes), stack, ray, ctr, n_nodes, n_prims, &hit_id, &hit_t, &hit_bu, &hit_bv);
            }
        }
        wave_rays += static_cast<uint32_t>(__popcll(__ballot(valid)));
        // ---- classify (get_illumination :492-520)
        const bool hit = valid && hit_id != RT_NO_HIT && !(plain & 1u);
        const bool occluded = (plain & 1u) ? (plain & 2u) != 0 : hit && hit_t <= sc.opts.max_occlusion_distance;
        uint32_t pf = 0;
        if (hit) pf = sc.prim_flags[hit_id];
        const bool terminal = hit && ((pf & PRIM_ILLUM_TERMINAL) || !next_real);
        const bool wants_node = hit && !terminal;
        const bool special = wants_node || (terminal && (pf & PRIM_EMISSIVE));
        const unsigned long long m_unocc = __ballot(valid && !occluded), m_special = __ballot(special), m_node = __ballot(wants_node);
        const unsigned long long job_mask = (G == 64u ? ~0ull : ((1ull << G) - 1ull)) << (jl * G);  // this lane's job's lanes
        const unsigned long long below = (1ull << lane) - 1ull;
        // wave-wide allocation of specials and nodes (every lane of the wave reaches this)
        const uint32_t tot_special = static_cast<uint32_t>(__popcll(m_special)), tot_node = static_cast<uint32_t>(__popcll(m_node));
        unsigned long long base_special = 0, base_node = 0;
        if (lane == 0) {
            if (tot_special) base_special = atomicAdd(&ctr->special_alloc, static_cast<unsigned long long>(tot_special));
            if (tot_node) base_node = atomicAdd(&ctr->node_alloc, static_cast<unsigned long long>(tot_node));
        }
        base_special = __shfl(base_special, 0, 64);
        base_node = __shfl(base_node, 0, 64);
        if (base_special + tot_special > shards.special_cap || base_node + tot_node > shards.node_cap) {
            if (lane == 0) atomicOr(&ctr->overflow, 1u);
            break;  // wave-uniform: the batch is re-run smaller
        }
        uint32_t node_index = 0;
        if (wants_node) {
            node_index = c.node_off + static_cast<uint32_t>(base_node) + static_cast<uint32_t>(__popcll(m_node & below));
            const GiJob &job = gp.jobs[job_index];  // path and root are not kept across the traversal: read again
            store_ray(np, node_index, ray.o, ray.d);
            np.path[node_index] = path_child(job.path, SLOT_GI + slot);
            np.root[node_index] = job.root;
            np.meta[node_index] = level | META_ILLUM | META_TRACED;
            np.prim[node_index] = hit_id;
            np.t[node_index] = hit_t;
            np.bu[node_index] = hit_bu;
            np.bv[node_index] = hit_bv;
        }
        wave_nodes += tot_node;
        const uint32_t special_off = c.special_off + static_cast<uint32_t>(base_special);
        if (special)
            gp.specials[special_off + static_cast<uint32_t>(__popcll(m_special & below))] =
                GiSpecial{slot | (wants_node ? 0x80000000u : 0u), wants_node ? node_index : hit_id};
        if (valid && slot == 0) {  // the job's results
            JobResult &res = gp.results[job_index];
            res.gi_counts = static_cast<uint32_t>(__popcll(m_unocc & job_mask)) | (static_cast<uint32_t>(__popcll(m_special & job_mask)) << 8);
            res.first_special = special_off + static_cast<uint32_t>(__popcll(m_special & ((1ull << (jl * G)) - 1ull)));
        }
    }
    // statistics: rays traced here, nodes made here (one atomic pair per wave)
    if (lane == 0 && wave_rays) {
        atomicAdd(&ctr->gi_rays, static_cast<unsigned long long>(wave_rays));
        if (wave_nodes) atomicAdd(&ctr->gi_materialised, static_cast<unsigned long long>(wave_nodes));
    }
    if (COUNT) {
        atomicAdd(&ctr->nodes_visited_closest, n_nodes & ((1ull << 40) - 1ull));
        atomicAdd(&ctr->prims_tested_closest, n_prims & ((1ull << 40) - 1ull));
    }
}

// The illumination nodes of the LAST traced level (get_illumination, :492-520, whose own children return zero, :495-497):
// each is finished by its trace -- emissive colour of the hit, occluded or not -- so, as in k_trace_gi, only an
// emissive primitive needs to be identified; anything else that comes first only has to be known to come first and to
// lie within max_occlusion_distance or not.  Two-phase queries over the emissive set (SceneDev::special[1]); the
// colour nodes of the level stay with k_trace_closest (launched with skip_illum).  physical.json: 164 M of these.
template <bool COUNT, int WAVES>
__global__ __launch_bounds__(BLOCK, WAVES) void k_trace_illum_last(SceneDev sc, NodePool np, Counters *ctrs, Shards shards, uint32_t level,
                                                                 uint32_t stack_entries, uint32_t what) {
    extern __shared__ __align__(16) uint32_t stack_lds[];
    uint32_t *stack = stack_lds + threadIdx.x;
    __shared__ LevelMap map;
    Counters *ctr = ctrs + blockIdx.x % shards.count;  // statistics only
    const uint32_t n_groups = level_map_build(map, ctrs, shards, level, what);
    if (blockIdx.x * (BLOCK / 64) >= n_groups) return;
    LdsNodeWord *top_nodes = stage_top_nodes(sc, stack_lds, stack_entries);
    unsigned long long n_nodes = 0, n_prims = 0;
    uint32_t wave_rays = 0;
    for (uint32_t g = blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6); g < n_groups; g += gridDim.x * (BLOCK / 64)) {
        uint32_t n_valid;
        const uint32_t idx = level_map_group(map, shards, g, &n_valid) + (threadIdx.x & 63u);
        bool mine = (threadIdx.x & 63u) < n_valid;
        uint32_t meta = 0;
        if (mine) {
            meta = np.meta[idx];
            mine = (meta & META_ILLUM) && !(meta & (META_INACTIVE | META_TRACED | META_DONE));
        }
        wave_rays += static_cast<uint32_t>(__popcll(__ballot(mine)));
        if (!mine) continue;
        RayD ray;
        load_ray(np, idx, &ray.o, &ray.d);
        const CullRay cr = make_cull_ray(ray, sc.scene_scale, sc.q_origin, sc.q_scale);
        uint32_t s_id = RT_NO_HIT, s_key = 0;
        double s_t = 0.0, s_bu, s_bv;
        if (sc.special[1].count)
            closest_hit<COUNT, true>(sc, special_tree(sc, top_nodes, 1u), stack, ray, ctr, n_nodes, n_prims, &s_id, &s_t, &s_bu, &s_bv, &cr, &s_key, true);
        const uint32_t plain = any_hit_before<COUNT>(sc, main_tree(sc, top_nodes), stack, ray, cr, n_nodes, n_prims, s_id != RT_NO_HIT, s_t, s_key,
                                                     sc.opts.max_occlusion_distance);
        const bool emissive_hit = s_id != RT_NO_HIT && !(plain & 1u);
        uint32_t flag = (plain & 1u) ? ((plain & 2u) ? 1u : 0u) : (emissive_hit && s_t <= sc.opts.max_occlusion_distance ? 1u : 0u);  // :512-516
        if (emissive_hit) {  // every primitive of the set is emissive
            const rt_material &m = sc.materials[sc.shade_prims[s_id].material];
            flag |= 2u;
            np.t[idx] = m.emissive[0];
            np.bu[idx] = m.emissive[1];
            np.bv[idx] = m.emissive[2];
        }
        np.meta[idx] = meta | META_DONE;
        np.prim[idx] = flag;
    }
    if ((threadIdx.x & 63u) == 0 && wave_rays) atomicAdd(&ctr->illum_last_rays, static_cast<unsigned long long>(wave_rays));
    if (COUNT) {
        atomicAdd(&ctr->nodes_visited_closest, n_nodes & ((1ull << 40) - 1ull));
        atomicAdd(&ctr->prims_tested_closest, n_prims & ((1ull << 40) - 1ull));
    }
}

// KdTreeAccelerator::shadow_cast (bounds.rs:310-316, :366-399) for one ray: OR over the primitives within max_distance,
// the unbounded objects first.
template <bool COUNT>
__device__ __forceinline__ bool any_hit(const SceneDev &sc, const TreeView &tv, uint32_t *stack, const RayD &ray, double tmax,
                                        unsigned long long &n_nodes, unsigned long long &n_prims) {
    bool occluded = false;
    for (uint32_t p = 0; p < sc.n_planes && !occluded; ++p) {
        const PlanePrim pl = sc.planes[p];
        const double *inv = sc.inv_world + 12ull * (pl.meta >> 4);
        RayD orr;
        orr.o = xf_point(inv, ray.o);
        orr.d = xf_vector(inv, ray.d);
        HitRec h;
        if (COUNT) n_prims++;
        if (hit_plane(mk3(pl.normal[0], pl.normal[1], pl.normal[2]), (pl.meta >> 2) & 3u, true, orr, true, tmax, &h) && h.t <= tmax) occluded = true;
    }
    if (!occluded && tv.root_ref != 0xFFFFFFFFu) {
        const CullRay cr = make_cull_ray(ray, sc.scene_scale, sc.q_origin, sc.q_scale);
        const float limit = cull_limit_f32(cull_limit(tmax));
        ObjRayCache cache;
        cache.xf = 0xFFFFFFFFu;
        stack[0] = STACK_SENTINEL;
        uint32_t sp = 1;
        uint32_t cur = tv.root_ref;
        while (true) {
            cur = descend<COUNT>(tv, cr, limit, cur, stack, sp, n_nodes);
            if (cur == STACK_SENTINEL) break;
            const uint32_t first = cur & 0x0FFFFFFFu, n = ((cur >> 28) & 7u) + 1u;
            for (uint32_t k = 0; k < n && !occluded; ++k) {
                HitRec h;
                if (COUNT) n_prims++;
                uint32_t pid;
                if (test_leaf_prim<true>(sc, tv.leaf_prims + first + k, ray, cache, tmax, &h, &pid)) occluded = true;
            }
            if (occluded) break;
            cur = stack[(--sp) * BLOCK];
        }
    }
    return occluded;
}

// Shadow rays in place (get_color_phong :216-225, get_color_physical :428-437: from the light towards the surface,
// shadow_cast(.., distance - BIAS), :94-96).  A shadow ray is (light, hit point): k_expand leaves the hit point and the
// mask of the point lights on whose side the surface lies in the colour node's job, and this kernel derives, traces and
// reduces the rays of ALL levels' jobs in one launch -- no ray record is written or read (before: 34 bytes per ray out
// of k_expand, in here, and a result byte each into k_resolve).  A wave takes 64 jobs a round and deals their rays to
// its lanes densely (prefix sums of the masks' bit counts, the owner of ray j by binary search as in k_expand), so
// every lane starts with a ray whatever the masks hold; a stopped ray sets its light's bit in the job's word through
// LDS, and the job's lane stores the word.
template <bool COUNT>
__global__ __launch_bounds__(BLOCK, TRACE_WAVES) void k_trace_shadow(SceneDev sc, GiPool gp, Counters *ctrs, Shards shards, uint32_t stack_entries, uint32_t what,
                                                                     uint32_t level) {
    extern __shared__ __align__(16) uint32_t stack_lds[];
    uint32_t *stack = stack_lds + threadIdx.x;
    __shared__ LevelMap map;
    __shared__ uint32_t s_incl[BLOCK], s_occ[BLOCK];
    Counters *ctr = ctrs + blockIdx.x % shards.count;  // statistics only
    const uint32_t n_groups = level_map_build(map, ctrs, shards, level, what);  // MAP_JOBS: all levels at once; MAP_LEVEL_JOBS: the jobs of `level`
    if (blockIdx.x * (BLOCK / 64) >= n_groups) return;
    LdsNodeWord *top_nodes = stage_top_nodes(sc, stack_lds, stack_entries);
    const TreeView tree = main_tree(sc, top_nodes);
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wbase = tid & ~63u;
    unsigned long long n_nodes = 0, n_prims = 0;
    uint32_t wave_rays = 0;  // wave-uniform
    for (uint32_t g = blockIdx.x * (BLOCK / 64) + (tid >> 6); g < n_groups; g += gridDim.x * (BLOCK / 64)) {
        uint32_t n_valid;
        const uint32_t first_job = level_map_group(map, shards, g, &n_valid);
        const bool valid = lane < n_valid;
        const uint32_t mask = valid ? gp.jobs[first_job + lane].light_mask : 0u;
        uint32_t incl = static_cast<uint32_t>(__popc(mask));
#pragma unroll
        for (uint32_t off = 1; off < 64; off <<= 1) {
            const uint32_t y = __shfl_up(incl, off, 64);
            if (lane >= off) incl += y;
        }
        const uint32_t total = __shfl(incl, 63, 64);
        if (total == 0) continue;  // wave-uniform
        wave_rays += total;
        s_incl[tid] = incl;
        s_occ[tid] = 0u;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // the LDS stores above -> the reads below (same wave)
        for (uint32_t base = 0; base < total; base += 64u) {
            const uint32_t j = min(base + lane, total - 1u);
            const bool active = base + lane < total;
            const uint32_t p = wave_owner(s_incl + wbase, j);  // the job (lane index) ray j belongs to
            const uint32_t pm = __shfl(mask, p, 64);
            if (!active) continue;
            const uint32_t slot = j - (s_incl[wbase + p] - static_cast<uint32_t>(__popc(pm)));
            uint32_t mm = pm;  // slot-th set bit of the job's light mask -> index among the point lights
            for (uint32_t k = 0; k < slot; ++k) mm &= mm - 1u;
            const uint32_t want = static_cast<uint32_t>(__ffs(mm)) - 1u;
            const DLight &l = sc.point_lights[want];
            const double *Pp = gp.jobs[first_job + p].P;
            RayD ray;
            ray.o = mk3(l.position[0], l.position[1], l.position[2]);
            d3 ld = ray.o - mk3(Pp[0], Pp[1], Pp[2]);
            const double dist = length3(ld);
            ld = unit3(ld);
            ray.d = -ld;
            if (any_hit<COUNT>(sc, tree, stack, ray, dist - BIAS, n_nodes, n_prims)) atomicOr(&s_occ[wbase + p], 1u << want);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // the LDS atomics -> the read below
        if (mask) gp.results[first_job + lane].shadow_occluded = s_occ[tid];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // ... before the next round's stores
    }
    if (lane == 0 && wave_rays) atomicAdd(&ctr->shadow_rays, static_cast<unsigned long long>(wave_rays));
    if (COUNT) {
        atomicAdd(&ctr->nodes_visited_shadow, n_nodes & ((1ull << 40) - 1ull));
        atomicAdd(&ctr->prims_tested_shadow, n_prims);
    }
}

// Any-hit probe rays of rt_trace_rays: rays 0..probe_count of the node pool (single-shard launch), limit in np.t.
__global__ __launch_bounds__(BLOCK, TRACE_WAVES) void k_trace_shadow_probe(SceneDev sc, NodePool np, uint32_t probe_count, uint32_t stack_entries) {
    extern __shared__ __align__(16) uint32_t stack_lds[];
    uint32_t *stack = stack_lds + threadIdx.x;
    const uint32_t n_groups = (probe_count + 63u) / 64u;
    if (blockIdx.x * (BLOCK / 64) >= n_groups) return;
    LdsNodeWord *top_nodes = stage_top_nodes(sc, stack_lds, stack_entries);
    unsigned long long n_nodes = 0, n_prims = 0;
    for (uint32_t g = blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6); g < n_groups; g += gridDim.x * (BLOCK / 64)) {
        const uint32_t i = g * 64u + (threadIdx.x & 63u);
        if (i >= probe_count) continue;
        RayD ray;
        load_ray(np, i, &ray.o, &ray.d);
        np.prim[i] = any_hit<false>(sc, main_tree(sc, top_nodes), stack, ray, np.t[i], n_nodes, n_prims) ? 1u : 0u;
    }
}

// ------------------------------------------------------------------ expand
// Emits the child rays of every node of `level`, and for a colour node the job that stands for its shadow rays and
// GI rays (traced in place by k_trace_shadow / k_trace_gi).
// Child block layout of a node: [reflection x nR][refraction x nT][GI x nG] (GI children only when G(d) > 64).
//
// Two phases per 64 parents (one wave):
//   A  one parent per lane: compute_data, child / shadow-ray counts, wave-wide
//      bump allocation; the parent's shading context goes to LDS (SoA by thread).
//   B  one CHILD per lane: lane j of the wave's contiguous child block finds its
//      parent by binary search over the wave's inclusive prefix sums, draws its
//      own sample and stores its ray -> every store instruction writes 64
//      consecutive records (phase A's per-parent loops wrote 64 strided ones and
//      left lanes of missed rays idle).
// Child order (parent-major, slot-minor) and RNG keys are unchanged.
// The nodes that still need work (not finished by the trace kernel, not padding) come from the level's live lists
// (NodePool::live, LiveMap, k_build_live): every wave takes groups of 64 live nodes, so the heavy path has (nearly)
// all lanes busy.
struct ExpandLds {
    double Px[BLOCK], Py[BLOCK], Pz[BLOCK];     // hit point
    double nx[BLOCK], ny[BLOCK], nz[BLOCK];     // shading normal
    double rx[BLOCK], ry[BLOCK], rz[BLOCK];     // mirror direction
    double tx[BLOCK], ty[BLOCK], tz[BLOCK];     // refraction direction
    double max_angle[BLOCK];                    // cone half-angle (physical)
    uint64_t path[BLOCK];
    uint32_t root[BLOCK], pixel[BLOCK], sample[BLOCK];
    uint32_t counts[BLOCK];                     // nR | nT << 16 | nG << 17 | phong << 31
    uint32_t meta_same[BLOCK], meta_refr[BLOCK];
    uint32_t incl_nodes[BLOCK];
};

template <bool COUNT>
__global__ __launch_bounds__(BLOCK, EXPAND_WAVES) void k_expand(SceneDev sc, WorkList wl, NodePool np, GiPool gp, Counters *ctrs,
                                                 Shards shards, uint32_t level) {
    __shared__ ExpandLds L;
    __shared__ LiveMap map;
    const ShardCtx c = shard_ctx(ctrs, shards);  // the shard this block ALLOCATES children and shadow rays from
    Counters *ctr = c.ctr;
    const uint32_t n_groups = live_map_build(map, ctrs, shards, level, true, true);
    const bool next_real = level + 1 < sc.opts.max_depth;
    const bool gi_in_place = gi_rays(sc, level) <= GI_MAX_INPLACE;  // the level's GI rays get jobs (k_trace_gi), not nodes
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wbase = tid & ~63u;
    const uint32_t child_depth = level + 1;
    const uint32_t wave_stride = gridDim.x * (BLOCK / 64);
    for (uint32_t g = blockIdx.x * (BLOCK / 64) + (tid >> 6); g < n_groups; g += wave_stride) {
        const uint32_t node = live_map_node(map, shards, np.live, g, lane);
        const bool valid = node != 0xFFFFFFFFu;
        const uint32_t idx = valid ? node : 0u;
        uint32_t nR = 0, nT = 0, nG = 0, nS = 0, mask = 0, phong_flag = 0, nJ = 0, gi_job = 0;
        uint32_t prim = RT_NO_HIT, meta = 0;
        d3 job_P = mk3(0.0, 0.0, 0.0), job_n = mk3(0.0, 0.0, 0.0);
        if (valid) {
            meta = np.meta[idx];
            prim = np.prim[idx];
        }
        if (prim != RT_NO_HIT) {
            const bool illum = (meta & META_ILLUM) != 0;
            d3 o, d;
            load_ray(np, idx, &o, &d);
            const ShadePrim &sh = sc.shade_prims[prim];
            const rt_material &m = sc.materials[sh.material];
            const Surface s = compute_surface(sc, sh, m.side, o, d, np.t[idx], np.bu[idx], np.bv[idx]);
            d3 refr_dir = mk3(0.0, 0.0, 0.0);
            bool cast_shadows = false;
            if (m.kind == RT_MAT_PHONG) {
                phong_flag = 0x80000000u;  // its one reflection child is the mirror direction itself
                nR = m.reflectivity > 0.0 ? 1u : 0u;
                if (!illum) {
                    nG = gi_rays(sc, level);
                    cast_shadows = m.reflectivity < 1.0;
                }
            } else {
                nR = sc.opts.max_reflected_rays > 0 ? refl_rays(sc, level) : 0u;
                if (m.opacity < 1.0) {
                    double ray_ior = (meta & META_REFR) ? sc.materials[meta >> META_MAT_SHIFT].refractive_index : 1.0;
                    double eta = ray_ior / m.refractive_index;
                    if (refract_dir(d, s.n, eta, &refr_dir)) nT = 1;
                }
                if (!illum) {
                    nG = gi_rays(sc, level);
                    cast_shadows = true;
                }
            }
            if (!next_real) nR = nT = nG = 0;
            if (gi_in_place && nG) {  // the job instead of nG child nodes
                gi_job = 0x80000000u;
                nG = 0;
            }
            if (cast_shadows) {  // the lights on whose side the surface lies: n . l > 0 with l = normalize(L - P), :216-221 / :428-433
                for (uint32_t bit = 0; bit < sc.n_point_lights; ++bit) {
                    const DLight &l = sc.point_lights[bit];
                    d3 ld = mk3(l.position[0], l.position[1], l.position[2]) - s.P;
                    // The sign of n . l is the sign of n . (L - P) unless that is within rounding of zero: the
                    // reference's value differs from (n . (L - P)) / |L - P| by a few 1e-16 |n|, this product from the
                    // exact one by a few 1e-16 B with B = |n|_1 |L - P|_1 >= |n| |L - P|.  Beyond 1e-10 B the sign is
                    // certain and the square root and three divisions of the normalisation are not needed; inside
                    // (and for NaN or overflow: the comparison is false) the reference's own arithmetic decides.
                    const double s_raw = dot3(s.n, ld);
                    const double bnd = 1e-10 * ((fabs(s.n.x) + fabs(s.n.y) + fabs(s.n.z)) * (fabs(ld.x) + fabs(ld.y) + fabs(ld.z)));
                    bool lit;
                    if (fabs(s_raw) > bnd) {
                        lit = s_raw > 0.0;
                    } else {
                        ld = unit3(ld);
                        lit = dot3(s.n, ld) > 0.0;
                    }
                    if (lit) {
                        mask |= 1u << bit;
                        nS++;
                    }
                }
            }
            if (gi_job || nS) {  // the colour node's job: its shadow rays and (gi_job) its GI rays
                nJ = 1;
                job_P = s.P;
                job_n = s.n;
            }
            // ---- context for phase B
            L.Px[tid] = s.P.x;
            L.Py[tid] = s.P.y;
            L.Pz[tid] = s.P.z;
            L.nx[tid] = s.n.x;
            L.ny[tid] = s.n.y;
            L.nz[tid] = s.n.z;
            if (nR) {
                const d3 rdir = reflect_dir(d, s.n);
                L.rx[tid] = rdir.x;
                L.ry[tid] = rdir.y;
                L.rz[tid] = rdir.z;
                L.max_angle[tid] = HALF_PI_D * m.roughness;
            }
            if (nT) {
                L.tx[tid] = refr_dir.x;
                L.ty[tid] = refr_dir.y;
                L.tz[tid] = refr_dir.z;
            }
            if (nR + nT + nG + nJ) {
                const uint32_t root = np.root[idx];
                uint32_t px, py, sample;
                item_to_pixel(wl, root, &px, &py, &sample);
                L.root[tid] = root;
                L.pixel[tid] = py * sc.opts.width + px;
                L.sample[tid] = sample;
                L.path[tid] = np.path[idx];
                const uint32_t same_kind = child_depth | (illum ? META_ILLUM : 0u);
                L.meta_same[tid] = same_kind;
                L.meta_refr[tid] = same_kind | META_REFR | (sh.material << META_MAT_SHIFT);
            }
        }
        const uint32_t total = nR + nT + nG;
        if (COUNT && valid) {  // census of the nodes that reach the heavy path
            atomicAdd(&ctr->census[level][(meta & META_ILLUM) ? 1 : 0], 1ull);
            if (prim == RT_NO_HIT) atomicAdd(&ctr->census[level][2], 1ull);
            if (total + nS + gi_job == 0) atomicAdd(&ctr->census[level][3], 1ull);
        }
        L.counts[tid] = nR | (nT << 16) | (nG << 17) | phong_flag;
        uint32_t wave_nodes, wave_jobs;
        const uint64_t child64 = wave_alloc(&ctr->node_alloc, total, &wave_nodes);
        const uint64_t job64 = wave_alloc(&ctr->job_alloc, nJ, &wave_jobs);
        // shard-local positions; wave-uniform overflow test (first lane's base + wave total)
        const uint64_t node_local = __shfl(child64, 0, 64), job_local = __shfl(job64, 0, 64);
        if (node_local + wave_nodes > shards.node_cap || job_local + wave_jobs > shards.job_cap) {
            if (lane == 0) atomicOr(&ctr->overflow, 1u);
            return;  // the shard is dead for this batch (wave-uniform exit)
        }
        const uint32_t node_base = c.node_off + static_cast<uint32_t>(node_local);
        const uint32_t first_child = c.node_off + static_cast<uint32_t>(child64);
        L.incl_nodes[tid] = first_child - node_base + total;
        const uint32_t ji = nJ ? c.job_off + static_cast<uint32_t>(job64) : NO_JOB;
        if (valid) np.xinfo[idx] = ExpandInfo{first_child, ji, nR | (nT << 16) | (nG << 17) | gi_job, mask};
        if (nJ) {  // what k_trace_shadow / k_trace_gi need to derive this node's shadow and GI rays
            GiJob job;
            job.P[0] = job_P.x; job.P[1] = job_P.y; job.P[2] = job_P.z;
            job.n[0] = job_n.x; job.n[1] = job_n.y; job.n[2] = job_n.z;
            job.path = L.path[tid];
            job.root = L.root[tid];
            job.pixel = L.pixel[tid];
            job.sample = L.sample[tid];
            job.light_mask = mask;
            job._pad[0] = job._pad[1] = 0;
            gp.jobs[ji] = job;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // phase A's LDS stores -> phase B's reads (same wave)

        // ---- phase B: child rays
        for (uint32_t j = lane; j < wave_nodes; j += 64u) {
            const uint32_t p = wbase + wave_owner(L.incl_nodes + wbase, j);
            const uint32_t cnt = L.counts[p];
            const uint32_t pR = cnt & 0xFFFFu, pT = (cnt >> 16) & 1u, pG = (cnt >> 17) & 0x3FFFu;
            const uint32_t slot = j - (L.incl_nodes[p] - (pR + pT + pG));
            const uint64_t path = L.path[p];
            const d3 P = mk3(L.Px[p], L.Py[p], L.Pz[p]);
            // which sampler, about which axis; the draw and the sampler itself are reached by all lanes together
            d3 axis;
            uint64_t cpath;
            uint32_t cmeta, kind;
            double max_angle = 0.0;
            if (slot < pR) {
                axis = mk3(L.rx[p], L.ry[p], L.rz[p]);
                cpath = path_child(path, slot);
                cmeta = L.meta_same[p];
                kind = (cnt & 0x80000000u) ? SAMPLE_NONE : SAMPLE_CONE;  // phong mirror: the direction itself
                max_angle = L.max_angle[p];
            } else if (slot < pR + pT) {
                axis = mk3(L.tx[p], L.ty[p], L.tz[p]);
                cpath = path_child(path, SLOT_REFRACTION);
                cmeta = L.meta_refr[p];
                kind = SAMPLE_NONE;
            } else {
                axis = mk3(L.nx[p], L.ny[p], L.nz[p]);
                cpath = path_child(path, SLOT_GI + (slot - pR - pT));
                cmeta = child_depth | META_ILLUM;
                kind = SAMPLE_HEMISPHERE;
            }
            double u1 = 0.0, u2 = 0.0;
            if (kind != SAMPLE_NONE) rng_pair(wl.seed, L.pixel[p], L.sample[p], cpath, &u1, &u2);
            const d3 dir = sample_direction(kind, axis, max_angle, u1, u2);
            const d3 org = P + dir * BIAS;
            const uint32_t at = node_base + j;
            store_ray(np, at, org, dir);
            np.path[at] = cpath;
            np.root[at] = L.root[p];
            np.meta[at] = cmeta;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // phase B's reads before the next round's stores
    }
}

// ------------------------------------------------------------------ resolve
// Bottom-up shading of every node of `level`; children of level+1 are resolved.
// Result: (t,bu,bv) <- RGB, prim <- flag (illumination nodes: occluded).
// Two instantiations per level: illumination nodes (get_emissive_*: a few loads, a
// few multiplies) and colour nodes (the full shading; register heavy), so that the
// illumination nodes do not run at the colour path's two waves per SIMD.
template <bool ILLUM>
__global__ __launch_bounds__(BLOCK, ILLUM ? RESOLVE_ILLUM_WAVES : RESOLVE_WAVES) void k_resolve(SceneDev sc, NodePool np, GiPool gp, Counters *ctrs, Shards shards,
                                                  uint32_t level, uint32_t aov) {
    __shared__ LiveMap map;
    const uint32_t n_groups = live_map_build(map, ctrs, shards, level, !ILLUM, ILLUM);
    const bool next_real = level + 1 < sc.opts.max_depth;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave_stride = gridDim.x * (BLOCK / 64);
    for (uint32_t g = blockIdx.x * (BLOCK / 64) + (tid >> 6); g < n_groups; g += wave_stride) {
        const uint32_t idx = live_map_node(map, shards, np.live, g, lane);
        if (idx == 0xFFFFFFFFu) continue;
        const uint32_t meta = np.meta[idx];
        const uint32_t prim = np.prim[idx];
        d3 result = mk3(0.0, 0.0, 0.0);
        uint32_t flag = 0;
        if (prim != RT_NO_HIT) {
            constexpr bool illum = ILLUM;
            d3 o, d;
            load_ray(np, idx, &o, &d);
            const double t = np.t[idx], bu = np.bu[idx], bv = np.bv[idx];
            const ShadePrim &sh = sc.shade_prims[prim];
            const rt_material &m = sc.materials[sh.material];
            const Surface s = compute_surface(sc, sh, m.side, o, d, t, bu, bv);
            const d3 mc = material_color(sc, sh, m, s, bu, bv);
            const d3 emissive = mk3(m.emissive[0], m.emissive[1], m.emissive[2]);
            const ExpandInfo xi = np.xinfo[idx];
            const uint32_t counts = xi.counts;
            // ---- children, one pass in slot order [reflection x nR][refraction x nT][GI x nG] (counts are zero when the
            // next level is cut off).  Their result words are read four at a time (aligned 16-byte loads over the
            // block of children); bit 1 of a word says whether its RGB is non-zero (4 of 5 are black: nothing more to
            // load), bit 0 "occluded".  Each class is summed in slot order, as the reference's loops do.
            const bool gi_job = !ILLUM && (counts >> 31) != 0u;  // this node's GI rays were traced in place
            const uint32_t nR = counts & 0xFFFFu, nT = (counts >> 16) & 1u, nG = gi_job ? 0u : (counts >> 17) & 0x3FFFu;
            const bool clamp_each = !ILLUM && m.kind == RT_MAT_PHYSICAL;  // :380 adds clamped colours; :299 and :196 raw ones
            d3 refl_sum = mk3(0.0, 0.0, 0.0), refr_rgb = mk3(0.0, 0.0, 0.0), gi_sum = mk3(0.0, 0.0, 0.0);
            uint32_t unoccluded = 0;
            {
                const uint32_t first = xi.first_child, end = first + nR + nT + nG;
                for (uint32_t a4 = first & ~3u; a4 < end; a4 += 4u) {
                    const uint4 w4 = *reinterpret_cast<const uint4 *>(np.prim + a4);
                    const uint32_t w[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
                    for (uint32_t j = 0; j < 4u; ++j) {
                        const uint32_t at = a4 + j;
                        if (at < first || at >= end) continue;
                        const uint32_t slot = at - first;
                        d3 rgb = mk3(0.0, 0.0, 0.0);
                        if (w[j] & 2u) rgb = mk3(np.t[at], np.bu[at], np.bv[at]);
                        if (slot < nR) {
                            refl_sum = refl_sum + (clamp_each ? sat01(rgb) : rgb);
                        } else if (slot < nR + nT) {
                            refr_rgb = rgb;
                        } else {
                            if (w[j] & 2u) gi_sum = gi_sum + rgb;
                            if (!(w[j] & 1u)) unoccluded++;
                        }
                    }
                }
            }

            uint32_t shadow_occluded = 0;  // the job's other result: the lights of xi.light_mask whose shadow ray was stopped
            JobResult jr{};
            if (!ILLUM && xi.job != NO_JOB) jr = gp.results[xi.job];  // shadow_occluded is valid where the mask is not empty, the GI fields where gi_job
            if (xi.light_mask) shadow_occluded = jr.shadow_occluded;
            if (gi_job) {  // compute_global_illumination's sums from the job: the non-zero terms in slot order (:121), the count (:123-125)
                unoccluded = job_unoccluded(jr.gi_counts);
                const uint32_t ns = job_n_special(jr.gi_counts), fs = jr.first_special;
                for (uint32_t q = 0; q < ns; ++q) {
                    const GiSpecial sp1 = gp.specials[fs + q];
                    if (sp1.slot_kind >> 31) {  // a node of its own, resolved one level down
                        if (np.prim[sp1.id] & 2u) gi_sum = gi_sum + mk3(np.t[sp1.id], np.bu[sp1.id], np.bv[sp1.id]);
                    } else {                    // terminal hit on an emissive surface: get_emissive_* = the material's colour
                        const rt_material &em = sc.materials[sc.shade_prims[sp1.id].material];
                        gi_sum = gi_sum + mk3(em.emissive[0], em.emissive[1], em.emissive[2]);
                    }
                }
            }

            if (illum) {
                // get_emissive_phong :135-167 / get_emissive_physical :263-337
                if (m.kind == RT_MAT_PHONG) {
                    result = emissive;
                    if (m.reflectivity > 0.0) result = result + hadamard(refl_sum, mc) * m.reflectivity;
                } else {
                    result = emissive;
                    if (sc.opts.max_reflected_rays > 0) {
                        const uint32_t n = refl_rays(sc, level);
                        const d3 acc = refl_sum * (HALF_PI_D / static_cast<double>(n));
                        result = result + hadamard(acc, mc);
                    }
                    bool has_refr;
                    if (next_real) {
                        has_refr = nT != 0;
                    } else {  // child not materialised: redo the TIR test of :307-309
                        has_refr = false;
                        if (m.opacity < 1.0) {
                            double ray_ior =
                                (meta & META_REFR) ? sc.materials[meta >> META_MAT_SHIFT].refractive_index : 1.0;
                            d3 dummy;
                            has_refr = refract_dir(d, s.n, ray_ior / m.refractive_index, &dummy);
                        }
                    }
                    if (has_refr) result = result + refr_rgb * (1.0 - m.opacity);
                }
                flag = t <= sc.opts.max_occlusion_distance ? 1u : 0u;  // :512-516
            } else {
                // ---- children: reflection / refraction
                bool has_refl = false, has_refr = false;
                d3 reflection = mk3(0.0, 0.0, 0.0), refraction = mk3(0.0, 0.0, 0.0);
                d3 f = mk3(0.0, 0.0, 0.0), diffuse = mk3(0.0, 0.0, 0.0);
                double n_dot_v = 0.0, roughness = 0.0;
                if (m.kind == RT_MAT_PHONG) {
                    if (m.reflectivity > 0.0) {  // :184-199
                        has_refl = true;
                        reflection = hadamard(refl_sum, mc);
                    }
                } else {  // :350-411
                    d3 view_dir = unit3(-d);
                    n_dot_v = fmax(dot3(s.n, view_dir), 0.0);
                    roughness = fmax(m.roughness, 0.04);
                    d3 base = mix3(mk3(0.04, 0.04, 0.04), mc, m.metalness);
                    f = base + (mk3(1.0, 1.0, 1.0) - base) * pow(1.0 - n_dot_v, 5.0);
                    d3 k_d = (mk3(1.0, 1.0, 1.0) - f) * (1.0 - m.metalness);
                    diffuse = INV_PI_D * hadamard(k_d, mc);
                    if (sc.opts.max_reflected_rays > 0) {
                        has_refl = true;
                        const uint32_t n = refl_rays(sc, level);
                        reflection = refl_sum * (HALF_PI_D / static_cast<double>(n));
                    }
                    if (next_real) {
                        has_refr = nT != 0;
                    } else if (m.opacity < 1.0) {
                        double ray_ior =
                            (meta & META_REFR) ? sc.materials[meta >> META_MAT_SHIFT].refractive_index : 1.0;
                        d3 dummy;
                        has_refr = refract_dir(d, s.n, ray_ior / m.refractive_index, &dummy);
                    }
                    if (has_refr) refraction = hadamard(refr_rgb, mix3(mk3(1.0, 1.0, 1.0), f, m.opacity));
                }
                // ---- direct lighting
                d3 ambient = mk3(0.0, 0.0, 0.0), irradiance = mk3(0.0, 0.0, 0.0);
                if (m.kind == RT_MAT_PHYSICAL || m.reflectivity < 1.0) {
                    const uint32_t mask = xi.light_mask;
                    uint32_t bit = 0;
                    for (uint32_t li = 0; li < sc.n_lights; ++li) {
                        const DLight &l = sc.lights[li];
                        if (l.kind == RT_LIGHT_AMBIENT) {
                            ambient = ambient + hadamard(mk3(l.color[0], l.color[1], l.color[2]), mc);
                            continue;
                        }
                        const bool lit_side = (mask >> bit) & 1u, occluded = (shadow_occluded >> bit) & 1u;
                        bit++;
                        if (!lit_side) continue;  // n.l <= 0 (decided in k_expand with the same arithmetic)
                        if (occluded) continue;
                        d3 ld = mk3(l.position[0], l.position[1], l.position[2]) - s.P;
                        const double dist = length3(ld);
                        ld = unit3(ld);
                        const double n_dot_l = dot3(s.n, ld);
                        const d3 lc = light_radiance(l, dist);
                        const d3 half_vec = unit3(ld - d);
                        if (m.kind == RT_MAT_PHONG) {  // :226-235
                            irradiance = irradiance + hadamard(lc, mc) * n_dot_l;
                            const double n_dot_h = dot3(s.n, half_vec);
                            if (n_dot_h > 0.0)
                                irradiance = irradiance + hadamard(lc, mk3(m.specular[0], m.specular[1], m.specular[2])) *
                                                              pow(n_dot_h, m.shininess);
                        } else {  // :438-454
                            const double n_dot_h = fmax(dot3(s.n, half_vec), 0.0);
                            const d3 radiance = lc * n_dot_l;
                            // ndf / geometry_function (utils/physical_material_equations.rs)
                            const double a = roughness * roughness, a2 = a * a;
                            double denom = n_dot_h * n_dot_h * (a2 - 1.0) + 1.0;
                            denom = PI_D * denom * denom;
                            const double ndf = a2 / denom;
                            const double rr = roughness + 1.0, kk = rr * rr / 8.0;
                            const double g = (n_dot_v / (n_dot_v * (1.0 - kk) + kk)) * (n_dot_l / (n_dot_l * (1.0 - kk) + kk));
                            d3 ds = diffuse;
                            if (!(n_dot_v == 0.0)) ds = diffuse + ((ndf * g) * f) / (4.0 * n_dot_v * n_dot_l);
                            irradiance = irradiance + hadamard(ds, radiance) * n_dot_l;
                        }
                    }
                }
                // ---- global illumination / ambient occlusion, :98-133
                const uint32_t n_gi = gi_rays(sc, level);  // == nG unless the next level is cut off (then nothing was cast:
                if (!next_real) unoccluded = n_gi;         //    zero light, every ray unoccluded, :495-497)
                d3 incoming = gi_sum;
                incoming = incoming / static_cast<double>(n_gi);
                const double ao = static_cast<double>(unoccluded) / static_cast<double>(n_gi);
                const d3 gi_tint = m.kind == RT_MAT_PHONG ? mc : diffuse;
                d3 color = emissive + (ambient + irradiance + hadamard(incoming, gi_tint)) * ao;
                if (has_refl) {
                    if (m.kind == RT_MAT_PHONG) color = mix3(color, sat01(reflection), m.reflectivity);  // :254-258
                    else color = mk3(lerp1(color.x, reflection.x, f.x), lerp1(color.y, reflection.y, f.y),
                                     lerp1(color.z, reflection.z, f.z));                              // :473-479
                }
                if (has_refr) color = mix3(sat01(refraction), color, m.opacity);                      // :481-486
                result = sat01(color);                                                                  // :543
                // The other two members of this node's ColorData, for the nodes whose ColorData somebody reads: camera
                // rays (screen_raycast) and refraction children (:485 `normal = refraction.normal`).  They go where the
                // ray was (nobody reads it again): (albedo | normal) over r0 r1 r2.
                if (aov && (level == 0 || (meta & META_REFR))) {
                    d3 n_out = s.n;
                    if (has_refr) {
                        n_out = mk3(0.0, 0.0, 1.0);  // ColorData::black of a miss or of the depth cut-off (mod.rs:39-41)
                        if (next_real) {
                            const uint32_t rc = xi.first_child + nR;  // the refraction child follows the reflections
                            if (!(np.meta[rc] & META_DONE)) {         // a colour node is finished by the trace kernel only as a miss
                                const Vec2d b = np.r1[rc], c = np.r2[rc];
                                n_out = mk3(b.y, c.x, c.y);
                            }
                        }
                    }
                    store_ray(np, idx, sat01(mc), n_out);  // :543 clamps the albedo too
                }
            }
        }
        if (!(result.x == 0.0 && result.y == 0.0 && result.z == 0.0)) {  // NaN counts as non-zero
            flag |= 2u;
            np.t[idx] = result.x;
            np.bu[idx] = result.y;
            np.bv[idx] = result.z;
        }
        np.prim[idx] = flag;
    }
}

// ------------------------------------------------------------------ accumulate / tonemap / AOVs
// screen_raycast :605-612: sum of the per-sample clamped colours, in sample order.
__global__ __launch_bounds__(BLOCK) void k_accumulate(SceneDev sc, WorkList wl, NodePool np, Counters *ctrs, Shards shards,
                                                     double *accum, double *albedo_accum, double *normal_out) {
    for (uint32_t k = 0; k < shards.count; ++k)  // += semantics: nothing may land if the batch is going to be re-run
        if (ctrs[k].overflow) return;
    const ShardCtx c = shard_ctx(ctrs, shards);
    const uint32_t n_pix = c.ctr->level_count[0] / wl.ns;
    for (uint32_t p = c.bid * BLOCK + threadIdx.x; p < n_pix; p += c.nblocks * BLOCK) {
        const uint32_t node0 = c.node_off + p * wl.ns;
        if (np.meta[node0] & META_INACTIVE) continue;
        uint32_t px, py, sample;
        item_to_pixel(wl, np.root[node0], &px, &py, &sample);
        d3 sum = mk3(0.0, 0.0, 0.0);
        for (uint32_t s = 0; s < wl.ns; ++s)
            if (np.prim[node0 + s] & 2u) sum = sum + mk3(np.t[node0 + s], np.bu[node0 + s], np.bv[node0 + s]);
        const uint64_t at = (static_cast<uint64_t>(py) * sc.opts.width + px) * 3ull;
        accum[at] += sum.x;
        accum[at + 1] += sum.y;
        accum[at + 2] += sum.z;
        // AOVs (k_resolve left albedo | normal in the camera nodes' ray fields; a node the trace kernel finished is a miss)
        if (albedo_accum) {  // :610, in sample order
            d3 alb = mk3(0.0, 0.0, 0.0);
            for (uint32_t s = 0; s < wl.ns; ++s)
                if (!(np.meta[node0 + s] & META_DONE)) {
                    const Vec2d a = np.r0[node0 + s], b = np.r1[node0 + s];
                    alb = alb + mk3(a.x, a.y, b.x);
                }
            albedo_accum[at] += alb.x;
            albedo_accum[at + 1] += alb.y;
            albedo_accum[at + 2] += alb.z;
        }
        if (normal_out && wl.s0 == 0) {  // the pixel's ColorData starts as sample 0's (:603-605)
            d3 n = mk3(0.0, 0.0, 1.0);
            if (!(np.meta[node0] & META_DONE)) {
                const Vec2d b = np.r1[node0], cc = np.r2[node0];
                n = mk3(b.y, cc.x, cc.y);
            }
            normal_out[at] = n.x;
            normal_out[at + 1] = n.y;
            normal_out[at + 2] = n.z;
        }
    }
}

// :614-621 + :757-769
__global__ __launch_bounds__(BLOCK) void k_tonemap(uint32_t n_pixels, uint32_t spp, const double *accum,
                                                  uint8_t *rgba) {
    for (uint32_t p = blockIdx.x * BLOCK + threadIdx.x; p < n_pixels; p += gridDim.x * BLOCK) {
        d3 c = mk3(accum[3ull * p], accum[3ull * p + 1], accum[3ull * p + 2]);
        if (spp != 1) c = sat01(c * (1.0 / static_cast<double>(spp)));
        c = sat01(c);
        const double e = 1.0 / GAMMA;
        c = mk3(pow(c.x, e), pow(c.y, e), pow(c.z, e));
        auto q = [](double v) -> uint8_t {  // `(c * 255.0) as u8`
            double x = v * 255.0;
            if (!(x > 0.0)) return 0;
            if (x >= 255.0) return 255;
            return static_cast<uint8_t>(x);
        };
        rgba[4ull * p] = q(c.x);
        rgba[4ull * p + 1] = q(c.y);
        rgba[4ull * p + 2] = q(c.z);
        rgba[4ull * p + 3] = 255;
    }
}

// raytrace_to_buffer's `image_buffer[index] = utils::to_argb_u32(color_data.color)` (:816-819, utils/mod.rs:14-19) for
// the pixels of a work list's tiles (inside the window): colour = :614-621 of the accumulated sum.
__global__ __launch_bounds__(BLOCK) void k_tonemap_argb_tiles(SceneDev sc, WorkList wl, uint32_t spp, const double *accum, uint32_t *argb) {
    const uint32_t per_tile = wl.tile_w * wl.tile_h;
    const uint64_t total = static_cast<uint64_t>(wl.n_tiles) * per_tile;
    for (uint64_t i = blockIdx.x * static_cast<uint64_t>(BLOCK) + threadIdx.x; i < total; i += static_cast<uint64_t>(gridDim.x) * BLOCK) {
        const uint32_t tile = static_cast<uint32_t>(i / per_tile), pit = static_cast<uint32_t>(i % per_tile);
        const uint32_t px = wl.tile_xy[2 * tile] + pit % wl.tile_w, py = wl.tile_xy[2 * tile + 1] + pit / wl.tile_w;
        if (px < wl.x0 || px >= wl.x1 || py < wl.y0 || py >= wl.y1) continue;
        const uint64_t p = static_cast<uint64_t>(py) * sc.opts.width + px;
        d3 c = mk3(accum[3ull * p], accum[3ull * p + 1], accum[3ull * p + 2]);
        if (spp != 1) c = sat01(c * (1.0 / static_cast<double>(spp)));
        c = sat01(c);
        const double e = 1.0 / GAMMA;
        c = mk3(pow(c.x, e), pow(c.y, e), pow(c.z, e));
        auto q = [](double v) -> uint32_t {  // `(c * 255.0) as u8`
            const double x = v * 255.0;
            if (!(x > 0.0)) return 0u;
            if (x >= 255.0) return 255u;
            return static_cast<uint32_t>(x);
        };
        argb[p] = (255u << 24) | (q(c.x) << 16) | (q(c.y) << 8) | q(c.z);
    }
}

// Exchange step of rt_render_multi: a shard's tiles leave its frame tile-major (`packed`: tile, row, column, channel) and
// are added into the root device's accumulation buffer.  Every pixel of a frame belongs to exactly one shard, so the sum is
// exact (the other addends are the zeros the caller's buffer starts from, or earlier passes of the same pixels).
__global__ __launch_bounds__(BLOCK) void k_pack_tiles(const uint32_t *tile_xy, uint32_t n_tiles, uint32_t tw, uint32_t th, uint32_t width,
                                                     uint32_t height, const double *frame, double *packed) {
    const uint64_t per_tile = static_cast<uint64_t>(tw) * th * 3u, total = per_tile * n_tiles;
    for (uint64_t i = blockIdx.x * static_cast<uint64_t>(BLOCK) + threadIdx.x; i < total; i += static_cast<uint64_t>(gridDim.x) * BLOCK) {
        const uint32_t tile = static_cast<uint32_t>(i / per_tile), r = static_cast<uint32_t>(i % per_tile);
        const uint32_t c = r % 3u, pit = r / 3u;
        const uint32_t px = tile_xy[2 * tile] + pit % tw, py = tile_xy[2 * tile + 1] + pit / tw;
        packed[i] = (px < width && py < height) ? frame[(static_cast<uint64_t>(py) * width + px) * 3ull + c] : 0.0;
    }
}
__global__ __launch_bounds__(BLOCK) void k_unpack_add_tiles(const uint32_t *tile_xy, uint32_t n_tiles, uint32_t tw, uint32_t th, uint32_t width,
                                                           uint32_t height, const double *packed, double *accum) {
    const uint64_t per_tile = static_cast<uint64_t>(tw) * th * 3u, total = per_tile * n_tiles;
    for (uint64_t i = blockIdx.x * static_cast<uint64_t>(BLOCK) + threadIdx.x; i < total; i += static_cast<uint64_t>(gridDim.x) * BLOCK) {
        const uint32_t tile = static_cast<uint32_t>(i / per_tile), r = static_cast<uint32_t>(i % per_tile);
        const uint32_t c = r % 3u, pit = r / 3u;
        const uint32_t px = tile_xy[2 * tile] + pit % tw, py = tile_xy[2 * tile + 1] + pit / tw;
        if (px < width && py < height) accum[(static_cast<uint64_t>(py) * width + px) * 3ull + c] += packed[i];
    }
}

// primary-hit AOVs of level-0 nodes (one sample per pixel): id, t, shading normal
__global__ __launch_bounds__(BLOCK) void k_primary_aov(SceneDev sc, WorkList wl, NodePool np, uint32_t n_items,
                                                      uint32_t *out_id, double *out_t, double *out_n) {
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < n_items; i += gridDim.x * BLOCK) {  // single-shard launches only
        if (np.meta[i] & META_INACTIVE) continue;
        uint32_t px, py, sample;
        item_to_pixel(wl, np.root[i], &px, &py, &sample);
        const uint64_t at = static_cast<uint64_t>(py) * sc.opts.width + px;
        const uint32_t prim = np.prim[i];
        out_id[at] = prim;
        if (prim == RT_NO_HIT) {
            out_t[at] = 0.0;
            out_n[3 * at] = out_n[3 * at + 1] = out_n[3 * at + 2] = 0.0;
            continue;
        }
        const ShadePrim &sh = sc.shade_prims[prim];
        const rt_material &m = sc.materials[sh.material];
        d3 o, d;
        load_ray(np, i, &o, &d);
        const Surface s = compute_surface(sc, sh, m.side, o, d, np.t[i], np.bu[i], np.bv[i]);
        out_t[at] = np.t[i];
        out_n[3 * at] = s.n.x;
        out_n[3 * at + 1] = s.n.y;
        out_n[3 * at + 2] = s.n.z;
    }
}

}  // namespace rt
