// rt_bvh_device.hip.h -- binned-SAH BVH2 build on the GPU (gfx950).  Same split rule as the host builder (rt_bvh.cpp:
// 16 centroid bins per axis over the segment's centroid bounds, SAH cost area(L) nL + area(R) nR, first minimum in
// axis-major / bin-minor order, single-primitive leaves except pairs of triangles of one mesh, the same depth guard),
// so on ordinary input the tree has the host tree's topology and boxes.  Replaces the "pre-process scene" step of the
// reference (KdTreeAccelerator::new / KdTree::build, src/core/bounds.rs:255-296, 452-610; timed in src/main.rs:48-56).
//
// Level synchronous: the segments (index ranges of `order`) of one tree level are processed together, so nodes come
// out numbered breadth first -- the order the traversal kernels want (the head of the node array = the top of the tree,
// kept in LDS).  Per level:
//   k_bvh_huge_*     segments of more than 8,192 primitives (the top of a large tree): chunks of 2,048 positions, one
//                    workgroup each -- bin into LDS and merge into the segment's global bins, choose the split, scatter
//                    the chunks stably (a chunk's left count is known from its own bin counts), copy back;
//   k_bvh_split<T>   one workgroup of T threads per smaller segment: bin the segment's primitives in LDS (LDS atomics on
//                    order-preserving u64 images of the f64 bounds), pick the split, partition the segment's slice of
//                    `order` stably (block scan), leave the children's boxes / centroid bounds / sizes;
//   exclusive scan   of the "became an interior node" flags (rocPRIM) = breadth-first node numbers;
//   k_bvh_emit       one thread per segment: write the node (child boxes quantised as the host does), patch the parent's
//                    child reference, append the two child segments to the next level.
// f64 throughout (-ffp-contract=off): bins, areas and costs round exactly as on the host.
#pragma once
#include <hip/hip_runtime.h>

#include <rocprim/rocprim.hpp>

#include <chrono>
#include <cstdio>
#include <cstdlib>

#include "rt_bvh.h"
#include "rt_internal.h"

namespace rt {
namespace bvhdev {

constexpr int BINS = 16;
constexpr uint32_t MESH_LEAF = 2;  // rt_bvh.cpp kMeshLeaf
constexpr uint32_t NONE = 0xFFFFFFFFu;
constexpr uint32_t SMALL_SEGMENT = 64;  // segments up to this size get one wave, larger ones a 512-thread workgroup
constexpr uint32_t HUGE_SEGMENT = 8192; // larger segments are binned and partitioned by many workgroups (chunks)
constexpr uint32_t CHUNK = 2048;        // positions per chunk of a huge segment
constexpr int HT = 512;                 // threads of the chunk kernels

struct Seg {            // a range of `order` still to be split
    uint32_t first, count;
    uint32_t parent;    // node index whose child this is (NONE: the root)
    uint32_t side_depth;  // bit 0: right child; bits 1..: depth
    double cmin[3], cmax[3];  // centroid bounds
};
struct Split {          // what k_bvh_split leaves for k_bvh_emit
    uint32_t interior, n_left;
    uint16_t lq[6], rq[6];  // children's boxes, quantised (min down, max up)
    double lcmin[3], lcmax[3], rcmin[3], rcmax[3];
};
struct Frame {
    double origin[3], scale[3];  // the f32 quantisation frame, widened
};
struct Counters {
    uint32_t n_interior, n_small, n_big, n_huge, n_leaves;  // the first four are reset every level
};
// Huge segments (the top of a large tree): one record per huge segment of the level, bins in global memory
struct HugeSeg {
    uint32_t seg;          // index into the level's segment array
    uint32_t chunk_first;  // its first chunk in the level's chunk table
    uint32_t n_chunks;
    int axis, sbin;        // the chosen split (axis < 0: first half | the rest)
    uint32_t n_left;
    uint32_t cnt[3][BINS];
    unsigned long long box[3][BINS][6];
    unsigned long long child[2][6];  // children's centroid bounds (keys)
    unsigned long long cbox[2][6];   // children's boxes (middle split only)
};
struct Chunk {
    uint32_t huge;         // index into the HugeSeg array
    uint32_t offset;       // first position of the chunk inside its segment
    uint32_t left_before;  // lefts of the segment in earlier chunks (filled by k_bvh_huge_select)
    uint32_t cnt[3][BINS]; // per-bin counts of this chunk
};

// order-preserving image of a double in u64 (so that LDS integer min / max atomics order doubles)
__device__ __forceinline__ unsigned long long key_of(double d) {
    const unsigned long long b = static_cast<unsigned long long>(__double_as_longlong(d));
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double double_of(unsigned long long k) {
    const unsigned long long b = (k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
    return __longlong_as_double(static_cast<long long>(b));
}
constexpr unsigned long long KEY_PLUS_INF = 0xFFF0000000000000ull;   // key_of(+inf)
constexpr unsigned long long KEY_MINUS_INF = 0x000FFFFFFFFFFFFFull;  // key_of(-inf)

__device__ __forceinline__ double area6(const double *mn, const double *mx) {  // Builder::area
    const double dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
    if (!(dx >= 0.0) || !(dy >= 0.0) || !(dz >= 0.0)) return 0.0;
    return 2.0 * (dx * dy + dy * dz + dx * dz);
}
__device__ __forceinline__ int bin_of(double c, double lo, double scale) {
    int b = static_cast<int>((c - lo) * scale);
    return min(max(b, 0), BINS - 1);
}
// rt_bvh.cpp q_down / q_up
__device__ __forceinline__ uint16_t q_down(double x, double origin, double scale) {
    double q = floor((x - origin) / scale);
    if (!(q > 0.0)) return 0;
    if (q > 65535.0) q = 65535.0;
    while (q > 0.0 && origin + q * scale > x) q -= 1.0;
    return static_cast<uint16_t>(q);
}
__device__ __forceinline__ uint16_t q_up(double x, double origin, double scale) {
    double q = ceil((x - origin) / scale);
    if (!(q < 65535.0)) return 65535;
    if (q < 0.0) q = 0.0;
    while (q < 65535.0 && origin + q * scale < x) q += 1.0;
    return static_cast<uint16_t>(q);
}
__device__ __forceinline__ uint32_t min_depth_for(uint32_t count) {
    uint32_t d = 0;
    unsigned long long cap = MAX_LEAF_PRIMS;
    while (cap < count) {
        cap *= 2;
        ++d;
    }
    return d;
}

// exclusive prefix of `flag` over the workgroup's threads and the total; all threads call it
template <int T>
__device__ __forceinline__ uint32_t block_rank(bool flag, uint32_t *wave_tot, uint32_t *total) {
    const unsigned long long m = __ballot(flag);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t rank = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0u));
    if (T > 64) {
        __syncthreads();  // wave_tot may still be read from the previous round
        if (lane == 0) wave_tot[wave] = static_cast<uint32_t>(__popcll(m));
        __syncthreads();
        uint32_t before = 0, all = 0;
        for (uint32_t w = 0; w < T / 64; ++w) {
            const uint32_t v = wave_tot[w];
            if (w < wave) before += v;
            all += v;
        }
        *total = all;
        return before + rank;
    }
    *total = static_cast<uint32_t>(__popcll(m));
    return rank;
}

// boxes: [n][6] (min xyz, max xyz) of the primitives' culling boxes; group: leaf-sharing group per primitive
template <int T>
__global__ __launch_bounds__(T) void k_bvh_split(const double *__restrict__ boxes, const uint32_t *__restrict__ group, uint32_t *order, uint32_t *tmp,
                                               const Seg *__restrict__ segs, const uint32_t *__restrict__ list, uint32_t n_list, Split *splits,
                                               uint32_t *flags, Frame frame) {
    __shared__ uint32_t s_cnt[3][BINS];
    __shared__ unsigned long long s_box[3][BINS][6];  // keys: min xyz, max xyz
    __shared__ unsigned long long s_child[2][6];      // children's centroid bounds (keys)
    __shared__ unsigned long long s_cbox[2][6];       // children's boxes (forced middle split only)
    __shared__ double s_cost[3 * (BINS - 1)];
    __shared__ uint32_t s_wave_tot[T / 64 + 1];
    __shared__ int s_axis, s_bin;
    __shared__ uint32_t s_nleft;
    if (blockIdx.x >= n_list) return;
    const uint32_t si = list[blockIdx.x];
    const Seg seg = segs[si];
    const uint32_t first = seg.first, count = seg.count, depth = seg.side_depth >> 1, tid = threadIdx.x;
    Split &out = splits[si];

    // ---- leaves (Builder::leaf_target, the depth guard of Builder::build)
    uint32_t leaf_target = 1;
    if (count == MESH_LEAF) {
        const uint32_t g0 = group[order[first]], g1 = group[order[first + 1]];
        if (g0 != NO_LEAF_GROUP && g0 == g1) leaf_target = MESH_LEAF;
    }
    const uint32_t depth_budget = (STACK_DEPTH - 2) - depth;
    if (count <= leaf_target || (count <= MAX_LEAF_PRIMS && depth_budget == 0)) {
        if (tid == 0) {
            out.interior = 0;
            flags[si] = 0;
        }
        return;
    }
    const bool force_balanced = min_depth_for(count) + 1 >= depth_budget;

    // ---- bins
    for (uint32_t i = tid; i < 3 * BINS; i += T) {
        s_cnt[i / BINS][i % BINS] = 0;
        for (int k = 0; k < 3; ++k) {
            s_box[i / BINS][i % BINS][k] = KEY_PLUS_INF;
            s_box[i / BINS][i % BINS][3 + k] = KEY_MINUS_INF;
        }
    }
    if (tid < 12) {
        s_child[tid / 6][tid % 6] = (tid % 6) < 3 ? KEY_PLUS_INF : KEY_MINUS_INF;
        s_cbox[tid / 6][tid % 6] = (tid % 6) < 3 ? KEY_PLUS_INF : KEY_MINUS_INF;
    }
    if (tid == 0) {
        s_axis = -1;
        s_bin = -1;
        s_nleft = 0;
    }
    __syncthreads();
    double scale[3];
    bool axis_ok[3];
    for (int a = 0; a < 3; ++a) {
        axis_ok[a] = seg.cmax[a] > seg.cmin[a];
        scale[a] = axis_ok[a] ? BINS / (seg.cmax[a] - seg.cmin[a]) : 0.0;
    }
    if (!force_balanced) {
        for (uint32_t i = tid; i < count; i += T) {
            const uint32_t p = order[first + i];
            const double *bx = boxes + 6ull * p;
            double b6[6];
            for (int k = 0; k < 6; ++k) b6[k] = bx[k];
            for (int a = 0; a < 3; ++a) {
                if (!axis_ok[a]) continue;
                const int b = bin_of(0.5 * (b6[a] + b6[3 + a]), seg.cmin[a], scale[a]);
                atomicAdd(&s_cnt[a][b], 1u);
                for (int k = 0; k < 3; ++k) {
                    atomicMin(&s_box[a][b][k], key_of(b6[k]));
                    atomicMax(&s_box[a][b][3 + k], key_of(b6[3 + k]));
                }
            }
        }
        __syncthreads();
        // ---- candidates: thread c = axis * 15 + split bin
        if (tid < 3 * (BINS - 1)) {
            const int a = tid / (BINS - 1), sb = tid % (BINS - 1);
            double cost = __builtin_inf();
            if (axis_ok[a]) {
                double lmn[3] = {__builtin_inf(), __builtin_inf(), __builtin_inf()}, lmx[3] = {-__builtin_inf(), -__builtin_inf(), -__builtin_inf()};
                double rmn[3] = {__builtin_inf(), __builtin_inf(), __builtin_inf()}, rmx[3] = {-__builtin_inf(), -__builtin_inf(), -__builtin_inf()};
                uint32_t nl = 0, nr = 0;
                for (int b = 0; b < BINS; ++b) {
                    const uint32_t c = s_cnt[a][b];
                    if (!c) continue;
                    double *mn = b <= sb ? lmn : rmn, *mx = b <= sb ? lmx : rmx;
                    (b <= sb ? nl : nr) += c;
                    for (int k = 0; k < 3; ++k) {
                        mn[k] = fmin(mn[k], double_of(s_box[a][b][k]));
                        mx[k] = fmax(mx[k], double_of(s_box[a][b][3 + k]));
                    }
                }
                if (nl && nr) cost = area6(lmn, lmx) * nl + area6(rmn, rmx) * nr;
            }
            s_cost[tid] = cost;
        }
        __syncthreads();
        if (tid == 0) {  // first minimum in axis-major, bin-minor order (strict <)
            double best = __builtin_inf();
            int bc = -1;
            for (int c = 0; c < 3 * (BINS - 1); ++c)
                if (s_cost[c] < best) {
                    best = s_cost[c];
                    bc = c;
                }
            if (bc >= 0) {
                s_axis = bc / (BINS - 1);
                s_bin = bc % (BINS - 1);
                uint32_t nl = 0;
                for (int b = 0; b <= s_bin; ++b) nl += s_cnt[s_axis][b];
                s_nleft = nl;
            }
        }
        __syncthreads();
    }
    const int axis = s_axis, sbin = s_bin;
    if (axis < 0 && !force_balanced && count <= MAX_LEAF_PRIMS) {  // identical centroids: nothing to gain from splitting
        if (tid == 0) {
            out.interior = 0;
            flags[si] = 0;
        }
        return;
    }
    // ---- partition (stable); without a SAH split (depth guard / degenerate centroids): the first half | the rest
    const bool middle = axis < 0;
    const uint32_t n_left = middle ? count / 2 : s_nleft;
    uint32_t run_l = 0, run_r = 0;
    for (uint32_t base = 0; base < count; base += T) {
        const uint32_t i = base + tid;
        const bool valid = i < count;
        uint32_t p = 0;
        bool left = false;
        double b6[6];
        if (valid) {
            p = order[first + i];
            const double *bx = boxes + 6ull * p;
            for (int k = 0; k < 6; ++k) b6[k] = bx[k];
            left = middle ? i < n_left : bin_of(0.5 * (b6[axis] + b6[3 + axis]), seg.cmin[axis], scale[axis]) <= sbin;
        }
        uint32_t total_l;
        const uint32_t rank = block_rank<T>(left, s_wave_tot, &total_l);
        const uint32_t n_valid = min(static_cast<uint32_t>(T), count - base);
        if (valid) {
            const uint32_t dest = left ? first + run_l + rank : first + n_left + run_r + (tid - rank);
            tmp[dest] = p;
            const int side = left ? 0 : 1;
            for (int k = 0; k < 3; ++k) {
                const unsigned long long c = key_of(0.5 * (b6[k] + b6[3 + k]));
                atomicMin(&s_child[side][k], c);
                atomicMax(&s_child[side][3 + k], c);
                if (middle) {
                    atomicMin(&s_cbox[side][k], key_of(b6[k]));
                    atomicMax(&s_cbox[side][3 + k], key_of(b6[3 + k]));
                }
            }
        }
        run_l += total_l;
        run_r += n_valid - total_l;
    }
    __syncthreads();
    for (uint32_t i = tid; i < count; i += T) order[first + i] = tmp[first + i];
    // ---- what the children are
    if (tid == 0) {
        double lmn[3], lmx[3], rmn[3], rmx[3];
        if (middle) {
            for (int k = 0; k < 3; ++k) {
                lmn[k] = double_of(s_cbox[0][k]);
                lmx[k] = double_of(s_cbox[0][3 + k]);
                rmn[k] = double_of(s_cbox[1][k]);
                rmx[k] = double_of(s_cbox[1][3 + k]);
            }
        } else {
            for (int k = 0; k < 3; ++k) {
                lmn[k] = rmn[k] = __builtin_inf();
                lmx[k] = rmx[k] = -__builtin_inf();
            }
            for (int b = 0; b < BINS; ++b) {
                if (!s_cnt[axis][b]) continue;
                double *mn = b <= sbin ? lmn : rmn, *mx = b <= sbin ? lmx : rmx;
                for (int k = 0; k < 3; ++k) {
                    mn[k] = fmin(mn[k], double_of(s_box[axis][b][k]));
                    mx[k] = fmax(mx[k], double_of(s_box[axis][b][3 + k]));
                }
            }
        }
        out.interior = 1;
        out.n_left = n_left;
        for (int k = 0; k < 3; ++k) {
            out.lq[k] = q_down(lmn[k], frame.origin[k], frame.scale[k]);
            out.lq[3 + k] = q_up(lmx[k], frame.origin[k], frame.scale[k]);
            out.rq[k] = q_down(rmn[k], frame.origin[k], frame.scale[k]);
            out.rq[3 + k] = q_up(rmx[k], frame.origin[k], frame.scale[k]);
            out.lcmin[k] = double_of(s_child[0][k]);
            out.lcmax[k] = double_of(s_child[0][3 + k]);
            out.rcmin[k] = double_of(s_child[1][k]);
            out.rcmax[k] = double_of(s_child[1][3 + k]);
        }
        flags[si] = 1;
    }
}

// ---------------------------------------------------------------- huge segments: many workgroups per segment
// prepare: one workgroup; chunk table of the level's huge segments, bins reset
__global__ __launch_bounds__(256) void k_bvh_huge_prepare(const Seg *__restrict__ segs, const uint32_t *__restrict__ huge_list, uint32_t n_huge, HugeSeg *hs,
                                                         Chunk *chunks, uint32_t *n_chunks_out) {
    __shared__ uint32_t s_first[1];
    if (threadIdx.x == 0) {
        uint32_t c = 0;
        for (uint32_t h = 0; h < n_huge; ++h) {
            hs[h].seg = huge_list[h];
            hs[h].chunk_first = c;
            hs[h].n_chunks = (segs[huge_list[h]].count + CHUNK - 1) / CHUNK;
            c += hs[h].n_chunks;
        }
        *n_chunks_out = c;
        s_first[0] = c;
    }
    __syncthreads();
    for (uint32_t h = 0; h < n_huge; ++h) {
        HugeSeg &H = hs[h];
        for (uint32_t i = threadIdx.x; i < 3 * BINS; i += 256) {
            H.cnt[i / BINS][i % BINS] = 0;
            for (int k = 0; k < 3; ++k) {
                H.box[i / BINS][i % BINS][k] = KEY_PLUS_INF;
                H.box[i / BINS][i % BINS][3 + k] = KEY_MINUS_INF;
            }
        }
        if (threadIdx.x < 12) {
            H.child[threadIdx.x / 6][threadIdx.x % 6] = (threadIdx.x % 6) < 3 ? KEY_PLUS_INF : KEY_MINUS_INF;
            H.cbox[threadIdx.x / 6][threadIdx.x % 6] = (threadIdx.x % 6) < 3 ? KEY_PLUS_INF : KEY_MINUS_INF;
        }
        for (uint32_t c = threadIdx.x; c < H.n_chunks; c += 256) {
            chunks[H.chunk_first + c].huge = h;
            chunks[H.chunk_first + c].offset = c * CHUNK;
        }
    }
}

// bin: one workgroup per chunk -- LDS bins, then at most 48 x 7 global atomics
__global__ __launch_bounds__(HT) void k_bvh_huge_bin(const double *__restrict__ boxes, const uint32_t *__restrict__ order, const Seg *__restrict__ segs, HugeSeg *hs,
                                                    Chunk *chunks, const uint32_t *__restrict__ n_chunks) {
    __shared__ uint32_t s_cnt[3][BINS];
    __shared__ unsigned long long s_box[3][BINS][6];
    if (blockIdx.x >= *n_chunks) return;
    Chunk &ck = chunks[blockIdx.x];
    HugeSeg &H = hs[ck.huge];
    const Seg seg = segs[H.seg];
    const uint32_t tid = threadIdx.x;
    for (uint32_t i = tid; i < 3 * BINS; i += HT) {
        s_cnt[i / BINS][i % BINS] = 0;
        for (int k = 0; k < 3; ++k) {
            s_box[i / BINS][i % BINS][k] = KEY_PLUS_INF;
            s_box[i / BINS][i % BINS][3 + k] = KEY_MINUS_INF;
        }
    }
    __syncthreads();
    double scale[3];
    bool axis_ok[3];
    for (int a = 0; a < 3; ++a) {
        axis_ok[a] = seg.cmax[a] > seg.cmin[a];
        scale[a] = axis_ok[a] ? BINS / (seg.cmax[a] - seg.cmin[a]) : 0.0;
    }
    const uint32_t end = min(seg.count, ck.offset + CHUNK);
    for (uint32_t i = ck.offset + tid; i < end; i += HT) {
        const uint32_t p = order[seg.first + i];
        const double *bx = boxes + 6ull * p;
        double b6[6];
        for (int k = 0; k < 6; ++k) b6[k] = bx[k];
        for (int a = 0; a < 3; ++a) {
            if (!axis_ok[a]) continue;
            const int b = bin_of(0.5 * (b6[a] + b6[3 + a]), seg.cmin[a], scale[a]);
            atomicAdd(&s_cnt[a][b], 1u);
            for (int k = 0; k < 3; ++k) {
                atomicMin(&s_box[a][b][k], key_of(b6[k]));
                atomicMax(&s_box[a][b][3 + k], key_of(b6[3 + k]));
            }
        }
    }
    __syncthreads();
    for (uint32_t i = tid; i < 3 * BINS; i += HT) {
        const uint32_t a = i / BINS, b = i % BINS, c = s_cnt[a][b];
        ck.cnt[a][b] = c;
        if (!c) continue;
        atomicAdd(&H.cnt[a][b], c);
        for (int k = 0; k < 3; ++k) {
            atomicMin(&H.box[a][b][k], s_box[a][b][k]);
            atomicMax(&H.box[a][b][3 + k], s_box[a][b][3 + k]);
        }
    }
}

// select: one wave per huge segment -- the split from the global bins (as k_bvh_split), the chunks' left offsets
__global__ __launch_bounds__(64) void k_bvh_huge_select(const Seg *__restrict__ segs, HugeSeg *hs, Chunk *chunks, Split *splits, uint32_t *flags, Frame frame) {
    __shared__ double s_cost[3 * (BINS - 1)];
    HugeSeg &H = hs[blockIdx.x];
    const Seg seg = segs[H.seg];
    const uint32_t tid = threadIdx.x, depth = seg.side_depth >> 1;
    const uint32_t depth_budget = (STACK_DEPTH - 2) - depth;
    const bool force_balanced = min_depth_for(seg.count) + 1 >= depth_budget;
    if (tid < 3 * (BINS - 1)) {
        const int a = tid / (BINS - 1), sb = tid % (BINS - 1);
        double cost = __builtin_inf();
        if (!force_balanced && seg.cmax[a] > seg.cmin[a]) {
            double lmn[3] = {__builtin_inf(), __builtin_inf(), __builtin_inf()}, lmx[3] = {-__builtin_inf(), -__builtin_inf(), -__builtin_inf()};
            double rmn[3] = {__builtin_inf(), __builtin_inf(), __builtin_inf()}, rmx[3] = {-__builtin_inf(), -__builtin_inf(), -__builtin_inf()};
            uint32_t nl = 0, nr = 0;
            for (int b = 0; b < BINS; ++b) {
                const uint32_t c = H.cnt[a][b];
                if (!c) continue;
                double *mn = b <= sb ? lmn : rmn, *mx = b <= sb ? lmx : rmx;
                (b <= sb ? nl : nr) += c;
                for (int k = 0; k < 3; ++k) {
                    mn[k] = fmin(mn[k], double_of(H.box[a][b][k]));
                    mx[k] = fmax(mx[k], double_of(H.box[a][b][3 + k]));
                }
            }
            if (nl && nr) cost = area6(lmn, lmx) * nl + area6(rmn, rmx) * nr;
        }
        s_cost[tid] = cost;
    }
    __syncthreads();
    if (tid == 0) {
        double best = __builtin_inf();
        int bc = -1;
        for (int c = 0; c < 3 * (BINS - 1); ++c)
            if (s_cost[c] < best) {
                best = s_cost[c];
                bc = c;
            }
        H.axis = bc >= 0 ? bc / (BINS - 1) : -1;
        H.sbin = bc >= 0 ? bc % (BINS - 1) : -1;
        uint32_t nl = 0;
        if (bc >= 0)
            for (int b = 0; b <= H.sbin; ++b) nl += H.cnt[H.axis][b];
        else nl = seg.count / 2;  // a huge segment is never a leaf: first half | the rest
        H.n_left = nl;
        uint32_t run = 0;
        for (uint32_t c = 0; c < H.n_chunks; ++c) {
            Chunk &ck = chunks[H.chunk_first + c];
            ck.left_before = run;
            if (bc >= 0) {
                for (int b = 0; b <= H.sbin; ++b) run += ck.cnt[H.axis][b];
            } else {
                const uint32_t end = min(seg.count, ck.offset + CHUNK);
                run += end <= nl ? end - ck.offset : (ck.offset < nl ? nl - ck.offset : 0u);
            }
        }
        Split &out = splits[H.seg];
        out.interior = 1;
        out.n_left = nl;
        flags[H.seg] = 1;
        if (bc >= 0) {  // children's boxes from the bins; a middle split's come from the scatter pass (k_bvh_huge_finish)
            double lmn[3], lmx[3], rmn[3], rmx[3];
            for (int k = 0; k < 3; ++k) {
                lmn[k] = rmn[k] = __builtin_inf();
                lmx[k] = rmx[k] = -__builtin_inf();
            }
            for (int b = 0; b < BINS; ++b) {
                if (!H.cnt[H.axis][b]) continue;
                double *mn = b <= H.sbin ? lmn : rmn, *mx = b <= H.sbin ? lmx : rmx;
                for (int k = 0; k < 3; ++k) {
                    mn[k] = fmin(mn[k], double_of(H.box[H.axis][b][k]));
                    mx[k] = fmax(mx[k], double_of(H.box[H.axis][b][3 + k]));
                }
            }
            for (int k = 0; k < 3; ++k) {
                out.lq[k] = q_down(lmn[k], frame.origin[k], frame.scale[k]);
                out.lq[3 + k] = q_up(lmx[k], frame.origin[k], frame.scale[k]);
                out.rq[k] = q_down(rmn[k], frame.origin[k], frame.scale[k]);
                out.rq[3 + k] = q_up(rmx[k], frame.origin[k], frame.scale[k]);
            }
        }
    }
}

// scatter: one workgroup per chunk -- stable partition of the chunk into the segment's slice of tmp
__global__ __launch_bounds__(HT) void k_bvh_huge_scatter(const double *__restrict__ boxes, const uint32_t *__restrict__ order, uint32_t *tmp,
                                                        const Seg *__restrict__ segs, HugeSeg *hs, const Chunk *__restrict__ chunks,
                                                        const uint32_t *__restrict__ n_chunks) {
    __shared__ unsigned long long s_child[2][6], s_cbox[2][6];
    __shared__ uint32_t s_wave_tot[HT / 64 + 1];
    if (blockIdx.x >= *n_chunks) return;
    const Chunk &ck = chunks[blockIdx.x];
    HugeSeg &H = hs[ck.huge];
    const Seg seg = segs[H.seg];
    const uint32_t tid = threadIdx.x, first = seg.first, n_left = H.n_left;
    const int axis = H.axis, sbin = H.sbin;
    const bool middle = axis < 0;
    if (tid < 12) {
        s_child[tid / 6][tid % 6] = (tid % 6) < 3 ? KEY_PLUS_INF : KEY_MINUS_INF;
        s_cbox[tid / 6][tid % 6] = (tid % 6) < 3 ? KEY_PLUS_INF : KEY_MINUS_INF;
    }
    __syncthreads();
    const double scale = middle ? 0.0 : BINS / (seg.cmax[axis] - seg.cmin[axis]);
    const uint32_t end = min(seg.count, ck.offset + CHUNK);
    uint32_t run_l = ck.left_before, run_r = ck.offset - ck.left_before;
    for (uint32_t base = ck.offset; base < end; base += HT) {
        const uint32_t i = base + tid;
        const bool valid = i < end;
        uint32_t p = 0;
        bool left = false;
        double b6[6];
        if (valid) {
            p = order[first + i];
            const double *bx = boxes + 6ull * p;
            for (int k = 0; k < 6; ++k) b6[k] = bx[k];
            left = middle ? i < n_left : bin_of(0.5 * (b6[axis] + b6[3 + axis]), seg.cmin[axis], scale) <= sbin;
        }
        uint32_t total_l;
        const uint32_t rank = block_rank<HT>(left, s_wave_tot, &total_l);
        const uint32_t n_valid = min(static_cast<uint32_t>(HT), end - base);
        if (valid) {
            tmp[left ? first + run_l + rank : first + n_left + run_r + (tid - rank)] = p;
            const int side = left ? 0 : 1;
            for (int k = 0; k < 3; ++k) {
                const unsigned long long c = key_of(0.5 * (b6[k] + b6[3 + k]));
                atomicMin(&s_child[side][k], c);
                atomicMax(&s_child[side][3 + k], c);
                if (middle) {
                    atomicMin(&s_cbox[side][k], key_of(b6[k]));
                    atomicMax(&s_cbox[side][3 + k], key_of(b6[3 + k]));
                }
            }
        }
        run_l += total_l;
        run_r += n_valid - total_l;
    }
    __syncthreads();
    if (tid < 12) {
        const uint32_t sd = tid / 6, k = tid % 6;
        if (k < 3) {
            atomicMin(&H.child[sd][k], s_child[sd][k]);
            if (middle) atomicMin(&H.cbox[sd][k], s_cbox[sd][k]);
        } else {
            atomicMax(&H.child[sd][k], s_child[sd][k]);
            if (middle) atomicMax(&H.cbox[sd][k], s_cbox[sd][k]);
        }
    }
}

// finish: copy the partitioned slices back (one workgroup per chunk); the first chunk of a segment completes its Split
__global__ __launch_bounds__(HT) void k_bvh_huge_finish(uint32_t *order, const uint32_t *__restrict__ tmp, const Seg *__restrict__ segs,
                                                       const HugeSeg *__restrict__ hs, const Chunk *__restrict__ chunks, const uint32_t *__restrict__ n_chunks,
                                                       Split *splits, Frame frame) {
    if (blockIdx.x >= *n_chunks) return;
    const Chunk &ck = chunks[blockIdx.x];
    const HugeSeg &H = hs[ck.huge];
    const Seg seg = segs[H.seg];
    const uint32_t end = min(seg.count, ck.offset + CHUNK);
    for (uint32_t i = ck.offset + threadIdx.x; i < end; i += HT) order[seg.first + i] = tmp[seg.first + i];
    if (ck.offset == 0 && threadIdx.x == 0) {
        Split &out = splits[H.seg];
        for (int k = 0; k < 3; ++k) {
            out.lcmin[k] = double_of(H.child[0][k]);
            out.lcmax[k] = double_of(H.child[0][3 + k]);
            out.rcmin[k] = double_of(H.child[1][k]);
            out.rcmax[k] = double_of(H.child[1][3 + k]);
            if (H.axis < 0) {
                out.lq[k] = q_down(double_of(H.cbox[0][k]), frame.origin[k], frame.scale[k]);
                out.lq[3 + k] = q_up(double_of(H.cbox[0][3 + k]), frame.origin[k], frame.scale[k]);
                out.rq[k] = q_down(double_of(H.cbox[1][k]), frame.origin[k], frame.scale[k]);
                out.rq[3 + k] = q_up(double_of(H.cbox[1][3 + k]), frame.origin[k], frame.scale[k]);
            }
        }
    }
}

// one thread per segment of the level; rank = exclusive scan of flags
__global__ __launch_bounds__(256) void k_bvh_emit(const Seg *__restrict__ segs, uint32_t n_seg, const Split *__restrict__ splits,
                                                 const uint32_t *__restrict__ flags, const uint32_t *__restrict__ rank, uint32_t node_base,
                                                 BvhNode *nodes, uint32_t node_cap, Seg *next, uint32_t *small_list, uint32_t *big_list,
                                                 uint32_t *huge_list, Counters *ctr, uint32_t *root_ref, uint32_t *error) {
    const uint32_t j = blockIdx.x * 256u + threadIdx.x;
    if (j >= n_seg) return;
    const Seg s = segs[j];
    const bool interior = flags[j] != 0;
    uint32_t ref;
    if (interior) {
        const uint32_t r = rank[j], idx = node_base + r;
        if (idx >= node_cap) {
            atomicOr(error, 1u);
            return;
        }
        ref = idx;
        const Split &sp = splits[j];
        BvhNode nd;
        for (int k = 0; k < 3; ++k) {
            nd.lmin[k] = sp.lq[k];
            nd.lmax[k] = sp.lq[3 + k];
            nd.rmin[k] = sp.rq[k];
            nd.rmax[k] = sp.rq[3 + k];
        }
        nd.left = nd.right = 0;  // patched by the children at the next level
        // the refs are patched with plain stores by other threads later: write the boxes only (12 x u16 = 24 bytes)
        uint16_t *dst = reinterpret_cast<uint16_t *>(nodes + idx);
        const uint16_t *src = reinterpret_cast<const uint16_t *>(&nd);
        for (int k = 0; k < 12; ++k) dst[k] = src[k];
        const uint32_t depth = (s.side_depth >> 1) + 1;
        Seg l, rr;
        l.first = s.first;
        l.count = sp.n_left;
        l.parent = idx;
        l.side_depth = (depth << 1);
        rr.first = s.first + sp.n_left;
        rr.count = s.count - sp.n_left;
        rr.parent = idx;
        rr.side_depth = (depth << 1) | 1u;
        for (int k = 0; k < 3; ++k) {
            l.cmin[k] = sp.lcmin[k];
            l.cmax[k] = sp.lcmax[k];
            rr.cmin[k] = sp.rcmin[k];
            rr.cmax[k] = sp.rcmax[k];
        }
        next[2 * r] = l;
        next[2 * r + 1] = rr;
        for (uint32_t c = 0; c < 2; ++c) {
            const uint32_t cnt = c ? rr.count : l.count;
            if (cnt <= SMALL_SEGMENT) small_list[atomicAdd(&ctr->n_small, 1u)] = 2 * r + c;
            else if (cnt <= HUGE_SEGMENT) big_list[atomicAdd(&ctr->n_big, 1u)] = 2 * r + c;
            else huge_list[atomicAdd(&ctr->n_huge, 1u)] = 2 * r + c;
        }
        atomicAdd(&ctr->n_interior, 1u);
    } else {
        ref = LEAF_FLAG | ((s.count - 1u) << 28) | s.first;  // make_leaf_ref
        atomicAdd(&ctr->n_leaves, 1u);
    }
    if (s.parent == NONE) *root_ref = ref;
    else if (s.side_depth & 1u) nodes[s.parent].right = ref;
    else nodes[s.parent].left = ref;
}

__global__ void k_bvh_iota(uint32_t *order, uint32_t n) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) order[i] = i;
}

struct DeviceBuildResult {
    BvhNode *d_nodes = nullptr;  // caller owns (hipFree)
    uint32_t n_nodes = 0;
    std::vector<uint32_t> order;  // leaf-record order -> index into the input boxes
    uint32_t root_ref = 0xFFFFFFFFu, max_depth = 0, n_leaves = 0;
    Aabb root_box{};
    float q_origin[3] = {0, 0, 0}, q_scale[3] = {1, 1, 1};
    double ms_device = 0.0;  // kernels + the per-level read-backs
};

struct DevMem {  // frees what is still held
    std::vector<void *> held;
    ~DevMem() {
        for (void *p : held) hipFree(p);
    }
    template <typename U>
    hipError_t alloc(U **p, size_t count) {
        void *v = nullptr;
        const hipError_t e = hipMalloc(&v, std::max<size_t>(1, count) * sizeof(U));
        if (e == hipSuccess) held.push_back(v);
        *p = static_cast<U *>(v);
        return e;
    }
    void release(void *p) { held.erase(std::remove(held.begin(), held.end(), p), held.end()); }
};

#define BVH_TRY(expr)                                                        \
    do {                                                                     \
        const hipError_t _e = (expr);                                        \
        if (_e != hipSuccess) {                                              \
            *err = std::string(#expr) + ": " + hipGetErrorString(_e);        \
            return false;                                                    \
        }                                                                    \
    } while (0)

// boxes: finite culling boxes of the bounded primitives (host); the frame (root box, quantisation) is computed here on
// the host exactly as rt_bvh.cpp does.  Returns false with *err set when a HIP call fails.
inline bool build_bvh_device(const std::vector<Aabb> &boxes, const std::vector<uint32_t> &leaf_group, DeviceBuildResult *res, std::string *err) {
    const uint32_t n = static_cast<uint32_t>(boxes.size());
    if (n == 0) return true;
    // ---- root box, centroid bounds, quantisation frame (rt_bvh.cpp: build_bvh)
    Aabb root = {{__builtin_inf(), __builtin_inf(), __builtin_inf()}, {-__builtin_inf(), -__builtin_inf(), -__builtin_inf()}};
    Seg seg0{};
    for (int k = 0; k < 3; ++k) {
        seg0.cmin[k] = __builtin_inf();
        seg0.cmax[k] = -__builtin_inf();
    }
    for (uint32_t i = 0; i < n; ++i)
        for (int k = 0; k < 3; ++k) {
            root.mn[k] = std::fmin(root.mn[k], boxes[i].mn[k]);
            root.mx[k] = std::fmax(root.mx[k], boxes[i].mx[k]);
            const double c = 0.5 * (boxes[i].mn[k] + boxes[i].mx[k]);
            seg0.cmin[k] = std::fmin(seg0.cmin[k], c);
            seg0.cmax[k] = std::fmax(seg0.cmax[k], c);
        }
    res->root_box = root;
    Frame frame{};
    for (int k = 0; k < 3; ++k) {
        float f = static_cast<float>(root.mn[k]);
        if (static_cast<double>(f) > root.mn[k]) f = std::nextafterf(f, -std::numeric_limits<float>::infinity());
        res->q_origin[k] = f;
        frame.origin[k] = static_cast<double>(f);
        double extent = root.mx[k] - frame.origin[k];
        if (!(extent > 0.0) || !std::isfinite(extent)) extent = 1.0;
        const double want = extent / 65535.0 * (1.0 + 1e-6);
        float s = static_cast<float>(want);
        if (static_cast<double>(s) < want) s = std::nextafterf(s, std::numeric_limits<float>::infinity());
        res->q_scale[k] = s;
        frame.scale[k] = static_cast<double>(s);
    }
    seg0.first = 0;
    seg0.count = n;
    seg0.parent = NONE;
    seg0.side_depth = 0;

    DevMem mem;
    double *d_boxes = nullptr;
    uint32_t *d_group = nullptr, *d_order = nullptr, *d_tmp = nullptr, *d_flags = nullptr, *d_rank = nullptr, *d_root = nullptr, *d_error = nullptr;
    uint32_t *d_small[2] = {nullptr, nullptr}, *d_big[2] = {nullptr, nullptr}, *d_huge[2] = {nullptr, nullptr}, *d_nchunks = nullptr;
    HugeSeg *d_hs = nullptr;
    Chunk *d_chunks = nullptr;
    const uint32_t huge_cap = n / HUGE_SEGMENT + 2, chunk_cap = n / CHUNK + huge_cap + 2;
    Seg *d_segs[2] = {nullptr, nullptr};
    Split *d_splits = nullptr;
    Counters *d_ctr = nullptr;
    BvhNode *d_nodes = nullptr;
    void *d_scan_tmp = nullptr;
    const uint32_t node_cap = n;  // a binary tree over n primitives has at most n - 1 interior nodes
    static_assert(sizeof(Aabb) == 6 * sizeof(double), "Aabb is six doubles");
    BVH_TRY(mem.alloc(&d_boxes, 6ull * n));
    BVH_TRY(mem.alloc(&d_group, n));
    BVH_TRY(mem.alloc(&d_order, n));
    BVH_TRY(mem.alloc(&d_tmp, n));
    BVH_TRY(mem.alloc(&d_flags, n));
    BVH_TRY(mem.alloc(&d_rank, n));
    BVH_TRY(mem.alloc(&d_root, 1));
    BVH_TRY(mem.alloc(&d_error, 1));
    for (int b = 0; b < 2; ++b) {
        BVH_TRY(mem.alloc(&d_small[b], n));
        BVH_TRY(mem.alloc(&d_big[b], n));
        BVH_TRY(mem.alloc(&d_huge[b], huge_cap));
        BVH_TRY(mem.alloc(&d_segs[b], n));
    }
    BVH_TRY(mem.alloc(&d_splits, n));
    BVH_TRY(mem.alloc(&d_hs, huge_cap));
    BVH_TRY(mem.alloc(&d_chunks, chunk_cap));
    BVH_TRY(mem.alloc(&d_nchunks, 1));
    BVH_TRY(mem.alloc(&d_ctr, 1));
    BVH_TRY(mem.alloc(&d_nodes, node_cap));
    size_t scan_bytes = 0;
    BVH_TRY(rocprim::exclusive_scan(nullptr, scan_bytes, d_flags, d_rank, 0u, static_cast<size_t>(n), rocprim::plus<uint32_t>()));
    BVH_TRY(mem.alloc(reinterpret_cast<char **>(&d_scan_tmp), scan_bytes));
    BVH_TRY(hipMemcpy(d_boxes, boxes.data(), 6ull * n * sizeof(double), hipMemcpyHostToDevice));
    BVH_TRY(hipMemcpy(d_group, leaf_group.data(), static_cast<size_t>(n) * sizeof(uint32_t), hipMemcpyHostToDevice));
    BVH_TRY(hipMemset(d_root, 0xFF, sizeof(uint32_t)));
    BVH_TRY(hipMemset(d_error, 0, sizeof(uint32_t)));
    BVH_TRY(hipMemcpy(d_segs[0], &seg0, sizeof(Seg), hipMemcpyHostToDevice));
    const uint32_t zero = 0;
    BVH_TRY(hipMemcpy(n <= SMALL_SEGMENT ? d_small[0] : (n <= HUGE_SEGMENT ? d_big[0] : d_huge[0]), &zero, sizeof(uint32_t), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_bvh_iota, dim3((n + 255) / 256), dim3(256), 0, nullptr, d_order, n);

    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEventCreate(&ev0);
    hipEventCreate(&ev1);
    hipEventRecord(ev0, nullptr);
    uint32_t n_seg = 1, n_small = n <= SMALL_SEGMENT ? 1u : 0u, n_huge = n > HUGE_SEGMENT ? 1u : 0u, n_big = 1u - n_small - n_huge, node_base = 0, levels = 0,
             n_leaves = 0;
    int cur = 0;
    bool ok = true;
    while (n_seg > 0) {
        if (levels > STACK_DEPTH) {
            *err = "internal: device BVH build exceeded the depth limit";
            ok = false;
            break;
        }
        if (n_small)
            hipLaunchKernelGGL((k_bvh_split<64>), dim3(n_small), dim3(64), 0, nullptr, d_boxes, d_group, d_order, d_tmp, d_segs[cur], d_small[cur], n_small, d_splits,
                               d_flags, frame);
        if (n_big)
            hipLaunchKernelGGL((k_bvh_split<512>), dim3(n_big), dim3(512), 0, nullptr, d_boxes, d_group, d_order, d_tmp, d_segs[cur], d_big[cur], n_big, d_splits, d_flags,
                               frame);
        if (n_huge) {  // the top of a large tree: many workgroups per segment (the grids are upper bounds; spare workgroups exit)
            const uint32_t g = std::min(chunk_cap, n / CHUNK + n_huge + 1);
            hipLaunchKernelGGL(k_bvh_huge_prepare, dim3(1), dim3(256), 0, nullptr, d_segs[cur], d_huge[cur], n_huge, d_hs, d_chunks, d_nchunks);
            hipLaunchKernelGGL(k_bvh_huge_bin, dim3(g), dim3(HT), 0, nullptr, d_boxes, d_order, d_segs[cur], d_hs, d_chunks, d_nchunks);
            hipLaunchKernelGGL(k_bvh_huge_select, dim3(n_huge), dim3(64), 0, nullptr, d_segs[cur], d_hs, d_chunks, d_splits, d_flags, frame);
            hipLaunchKernelGGL(k_bvh_huge_scatter, dim3(g), dim3(HT), 0, nullptr, d_boxes, d_order, d_tmp, d_segs[cur], d_hs, d_chunks, d_nchunks);
            hipLaunchKernelGGL(k_bvh_huge_finish, dim3(g), dim3(HT), 0, nullptr, d_order, d_tmp, d_segs[cur], d_hs, d_chunks, d_nchunks, d_splits, frame);
        }
        if (rocprim::exclusive_scan(d_scan_tmp, scan_bytes, d_flags, d_rank, 0u, static_cast<size_t>(n_seg), rocprim::plus<uint32_t>()) != hipSuccess ||
            hipMemsetAsync(d_ctr, 0, 4 * sizeof(uint32_t), nullptr) != hipSuccess) {  // n_leaves keeps counting
            *err = "device BVH build: scan failed";
            ok = false;
            break;
        }
        if (levels == 0 && hipMemsetAsync(d_ctr, 0, sizeof(Counters), nullptr) != hipSuccess) ok = false;
        hipLaunchKernelGGL(k_bvh_emit, dim3((n_seg + 255) / 256), dim3(256), 0, nullptr, d_segs[cur], n_seg, d_splits, d_flags, d_rank, node_base, d_nodes, node_cap,
                           d_segs[cur ^ 1], d_small[cur ^ 1], d_big[cur ^ 1], d_huge[cur ^ 1], d_ctr, d_root, d_error);
        Counters c{};
        static const bool trace_levels = std::getenv("MI355RT_BVH_TRACE") != nullptr;
        const auto t_level = std::chrono::steady_clock::now();
        if (hipMemcpy(&c, d_ctr, sizeof(c), hipMemcpyDeviceToHost) != hipSuccess) {
            *err = std::string("device BVH build: ") + hipGetErrorString(hipGetLastError());
            ok = false;
            break;
        }
        if (trace_levels)
            std::fprintf(stderr, "[bvh] level %2u: %8u segments (%u small, %u big, %u huge) -> %8u interior, wait %.3f ms\n", levels, n_seg, n_small, n_big, n_huge, c.n_interior,
                         std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_level).count());
        node_base += c.n_interior;
        n_seg = 2 * c.n_interior;
        n_small = c.n_small;
        n_big = c.n_big;
        n_huge = c.n_huge;
        n_leaves = c.n_leaves;
        cur ^= 1;
        ++levels;
    }
    hipEventRecord(ev1, nullptr);
    hipEventSynchronize(ev1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, ev0, ev1);
    hipEventDestroy(ev0);
    hipEventDestroy(ev1);
    res->ms_device = ms;
    if (!ok) return false;
    uint32_t flag = 0;
    BVH_TRY(hipMemcpy(&flag, d_error, sizeof(flag), hipMemcpyDeviceToHost));
    if (flag) {
        *err = "internal: device BVH build ran out of node slots";
        return false;
    }
    BVH_TRY(hipMemcpy(&res->root_ref, d_root, sizeof(uint32_t), hipMemcpyDeviceToHost));
    res->order.resize(n);
    BVH_TRY(hipMemcpy(res->order.data(), d_order, static_cast<size_t>(n) * sizeof(uint32_t), hipMemcpyDeviceToHost));
    res->n_nodes = node_base;
    res->max_depth = levels ? levels - 1 : 0;
    res->n_leaves = n_leaves;
    // the node array stays on the device, trimmed to what is used
    if (node_base) {
        BvhNode *trim = nullptr;
        BVH_TRY(hipMalloc(reinterpret_cast<void **>(&trim), static_cast<size_t>(node_base) * sizeof(BvhNode)));
        const hipError_t e = hipMemcpy(trim, d_nodes, static_cast<size_t>(node_base) * sizeof(BvhNode), hipMemcpyDeviceToDevice);
        if (e != hipSuccess) {
            hipFree(trim);
            *err = std::string("device BVH build: ") + hipGetErrorString(e);
            return false;
        }
        res->d_nodes = trim;
    }
    return true;
}

#undef BVH_TRY

}  // namespace bvhdev
}  // namespace rt
