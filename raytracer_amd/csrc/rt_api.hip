// rt_api.hip -- C ABI of libmi355rt.so (include/mi355rt.h): scene upload,
// batch orchestration of the wavefront kernels, parity probes.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

#include "rt_bvh.h"
#include "rt_tune.h"
#include "rt_bvh_device.hip.h"
#include "rt_internal.h"
#include "rt_kernels.hip.h"

// The library's static device scratch (see "static device scratch" below): every device holds its own copy.
static __device__ __align__(256) unsigned char scratch[RT_NOALLOC_SCRATCH_BYTES];

namespace {

thread_local std::string g_error;

int fail(int code, const std::string &msg) {
    g_error = msg;
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess)                                                                      \
            return fail(RT_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(_e));         \
    } while (0)

using namespace rt;

// BoundingVolume::from_bounds_and_transform (bounds.rs:40-72): AABB of the 8
// transformed corners, corner order x-major, f64::min / f64::max.
Aabb world_aabb(const double mn[3], const double mx[3], const double *w) {
    Aabb out{};
    bool first = true;
    const double xs[2] = {mn[0], mx[0]}, ys[2] = {mn[1], mx[1]}, zs[2] = {mn[2], mx[2]};
    for (double x : xs)
        for (double y : ys)
            for (double z : zs) {
                double p[3] = {w[0] * x + w[1] * y + w[2] * z + w[3], w[4] * x + w[5] * y + w[6] * z + w[7],
                               w[8] * x + w[9] * y + w[10] * z + w[11]};
                for (int k = 0; k < 3; ++k) {
                    if (first) {
                        out.mn[k] = out.mx[k] = p[k];
                    } else {
                        out.mn[k] = std::fmin(out.mn[k], p[k]);
                        out.mx[k] = std::fmax(out.mx[k], p[k]);
                    }
                }
                first = false;
            }
    return out;
}

// Rotation3::rotation_between(normal, +y) rows 0 and 2 (plane.rs:138-141).
// Returns false where the reference's unwrap() panics (normal anti-parallel to +y).
bool plane_uv_rotation(const double n[3], double rows[6]) {
    const double ident[6] = {1, 0, 0, 0, 0, 1};
    double len = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
    if (!(len > 0.0)) {
        std::memcpy(rows, ident, sizeof(ident));
        return true;
    }
    double a[3] = {n[0] / len, n[1] / len, n[2] / len};
    // c = a x (0,1,0)
    double c[3] = {a[1] * 0.0 - a[2] * 1.0, a[2] * 0.0 - a[0] * 0.0, a[0] * 1.0 - a[1] * 0.0};
    double cl = std::sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
    double cosang = a[0] * 0.0 + a[1] * 1.0 + a[2] * 0.0;
    if (cl > F64_EPS) {
        double ax[3] = {c[0] / cl, c[1] / cl, c[2] / cl};
        double angle = std::acos(cosang);
        if (angle == 0.0) {
            std::memcpy(rows, ident, sizeof(ident));
            return true;
        }
        double s = std::sin(angle), co = std::cos(angle), omc = 1.0 - co;
        double sqx = ax[0] * ax[0], sqz = ax[2] * ax[2];
        rows[0] = sqx + (1.0 - sqx) * co;
        rows[1] = ax[0] * ax[1] * omc - ax[2] * s;
        rows[2] = ax[0] * ax[2] * omc + ax[1] * s;
        rows[3] = ax[0] * ax[2] * omc - ax[1] * s;
        rows[4] = ax[1] * ax[2] * omc + ax[0] * s;
        rows[5] = sqz + (1.0 - sqz) * co;
        return true;
    }
    if (cosang < 0.0) return false;
    std::memcpy(rows, ident, sizeof(ident));
    return true;
}

// Host -> device copy of a scene array.  Large arrays (the 1M-triangle scene uploads 380 MB) go through two pinned
// 16 MB staging buffers, the CPU filling one while the DMA engine drains the other: a hipMemcpy straight from pageable
// memory ran at 2 GB/s.
struct Stager {
    static constexpr size_t CHUNK = 16u << 20;
    void *pinned[2] = {nullptr, nullptr};
    hipEvent_t drained[2] = {nullptr, nullptr};
    hipStream_t stream = nullptr;
    double ms_alloc = 0.0, ms_fill = 0.0, ms_wait = 0.0;  // MI355RT_TIMING: hipMalloc, CPU copies into the staging buffers, waiting for the DMA engine
    static double since(std::chrono::steady_clock::time_point t0) {
        return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    ~Stager() {
        for (int i = 0; i < 2; ++i) {
            if (drained[i]) hipEventDestroy(drained[i]);
            if (pinned[i]) hipHostFree(pinned[i]);
        }
        if (stream) hipStreamDestroy(stream);
    }
    int copy(void *dst, const void *src, size_t bytes) {
        if (bytes < CHUNK) {
            HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
            return RT_OK;
        }
        if (!stream) {
            HIP_TRY(hipStreamCreate(&stream));
            for (int i = 0; i < 2; ++i) {
                HIP_TRY(hipHostMalloc(&pinned[i], CHUNK));
                HIP_TRY(hipEventCreateWithFlags(&drained[i], hipEventDisableTiming));
            }
        }
        int i = 0;
        for (size_t off = 0; off < bytes; off += CHUNK, i ^= 1) {
            const size_t n = std::min(CHUNK, bytes - off);
            auto t0 = std::chrono::steady_clock::now();
            HIP_TRY(hipEventSynchronize(drained[i]));  // a fresh event is complete
            ms_wait += since(t0);
            t0 = std::chrono::steady_clock::now();
            std::memcpy(pinned[i], static_cast<const char *>(src) + off, n);
            ms_fill += since(t0);
            HIP_TRY(hipMemcpyAsync(static_cast<char *>(dst) + off, pinned[i], n, hipMemcpyHostToDevice, stream));
            HIP_TRY(hipEventRecord(drained[i], stream));
        }
        const auto t0 = std::chrono::steady_clock::now();
        HIP_TRY(hipStreamSynchronize(stream));
        ms_wait += since(t0);
        return RT_OK;
    }
};

template <typename T>
int upload(Stager &st, const T *host, size_t count, const T **dev, std::vector<void *> &allocs, uint64_t *bytes) {
    *dev = nullptr;
    if (count == 0) return RT_OK;
    void *p = nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    HIP_TRY(hipMalloc(&p, count * sizeof(T)));
    st.ms_alloc += Stager::since(t0);
    allocs.push_back(p);
    const auto t1 = std::chrono::steady_clock::now();
    if (int rc = st.copy(p, host, count * sizeof(T))) return rc;
    if (std::getenv("MI355RT_TIMING")) std::fprintf(stderr, "rt_scene_create:   upload of %zu bytes: hipMalloc %.2f ms, copy %.2f ms\n", count * sizeof(T), std::chrono::duration<double, std::milli>(t1 - t0).count(), Stager::since(t1));
    *dev = static_cast<const T *>(p);
    *bytes += count * sizeof(T);
    return RT_OK;
}
template <typename T>
int upload(Stager &st, const std::vector<T> &host, const T **dev, std::vector<void *> &allocs, uint64_t *bytes) {
    return upload(st, host.data(), host.size(), dev, allocs, bytes);
}

// fn(begin, end) over [0, n) on up to 16 host threads (one for small n)
template <typename F>
void parallel_for(size_t n, F fn) {
    const size_t hw = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    const size_t workers = n < 65536 ? 1 : hw;
    if (workers == 1) {
        fn(size_t(0), n);
        return;
    }
    std::vector<std::thread> pool;
    const size_t per = (n + workers - 1) / workers;
    for (size_t w = 0; w < workers; ++w) {
        const size_t b = w * per, e = std::min(n, b + per);
        if (b < e) pool.emplace_back([=, &fn] { fn(b, e); });
    }
    for (auto &t : pool) t.join();
}

bool is_special_flags(uint32_t prim_flags) { return (prim_flags & PRIM_EMISSIVE) || !(prim_flags & PRIM_ILLUM_TERMINAL); }

uint32_t morton2(uint32_t x, uint32_t y) {
    auto spread = [](uint32_t v) {
        v &= 0xFFFFu;
        v = (v | (v << 8)) & 0x00FF00FFu;
        v = (v | (v << 4)) & 0x0F0F0F0Fu;
        v = (v | (v << 2)) & 0x33333333u;
        v = (v | (v << 1)) & 0x55555555u;
        return v;
    };
    return spread(x) | (spread(y) << 1);
}

struct EventTimer {
    // accumulates elapsed ms per kernel class from (start, stop) events.  Consecutive spans share an event -- the stop of
    // one is the start of the next: every hipEventRecord between two kernels costs ~5 us of idle queue (two of them, as
    // before, 10 us: 0.3 ms of a scene.json frame's 29 class changes)
    struct Span {
        int a, b, cls;
    };
    std::vector<hipEvent_t> evs;
    std::vector<Span> spans;
    bool enabled = false;
    bool chaining = true;  // every launch of the stream lies inside a span (else spans are not adjacent: no sharing)
    bool chain = false;    // the last recorded event is the stop of the previous span
    hipStream_t stream = nullptr;
    EventTimer() = default;
    EventTimer(const EventTimer &) = delete;
    EventTimer &operator=(const EventTimer &) = delete;
    ~EventTimer() {  // an early error return leaves events behind
        for (hipEvent_t e : evs) hipEventDestroy(e);
    }
    int mark() {
        hipEvent_t e = nullptr;
        hipEventCreate(&e);
        hipEventRecord(e, stream);
        evs.push_back(e);
        return static_cast<int>(evs.size()) - 1;
    }
    void begin(int cls) {
        if (!enabled) return;
        const int a = chain && chaining ? static_cast<int>(evs.size()) - 1 : mark();
        spans.push_back(Span{a, -1, cls});
        chain = false;
    }
    void end() {
        if (!enabled) return;
        spans.back().b = mark();
        chain = true;
    }
    void collect(float *per_class, uint32_t *launches, int n_classes) {
        for (const Span &s : spans) {
            float ms = 0.f;
            if (s.b >= 0 && hipEventElapsedTime(&ms, evs[s.a], evs[s.b]) == hipSuccess && s.cls < n_classes) {
                per_class[s.cls] += ms;
                launches[s.cls] += 1;
            }
        }
        for (hipEvent_t e : evs) hipEventDestroy(e);
        evs.clear();
        spans.clear();
        chain = false;
    }
};

enum { CLS_RAYGEN = 0, CLS_CLOSEST, CLS_SHADOW, CLS_EXPAND, CLS_RESOLVE, CLS_ACCUM, CLS_MISC, CLS_GI, CLS_ILLUM, CLS_COUNT };

}  // namespace

// The wavefront state one render call works in: pools, counters, the tile list of the call's plan.  A scene owns one
// carved from its hipMalloc'ed pool budget (ensure_pools); rt_render_device_noalloc uses one carved from the library's
// static device scratch (one per device, fixed size).
struct RenderCtx {
    NodePool np{};
    GiPool gp{};
    Counters *d_ctr = nullptr;
    Counters *h_ctr = nullptr;  // host copy of the counters (pinned for a scene's own context)
    uint32_t *d_tiles = nullptr;
    uint32_t d_tiles_cap = 0;   // tiles
    bool can_grow = true;       // d_tiles may be re-allocated (false: fixed scratch)
    bool plan_valid = false;    // d_tiles holds the tile list of plan_key
    rt_render_params plan_key{};
    WorkList plan_wl{};
    bool plan_scattered = false;  // plan_wl is in scattered order (a call of several batches)
    // node slots per camera ray assumed before anything was measured: 16 for a scene of Phong surfaces (a colour node has
    // at most one reflection child; of its GI rays one in ten gets a node: scene.json 8.2, mesh.json 3), 96 where physical
    // materials are common (32 reflection children at level 0, 4 each at level 1: physical.json 70)
    uint32_t guess_nodes_per_item = 64;
    double est_nodes_per_item = 0.0, est_jobs_per_item = 0.0, est_specials_per_item = 0.0;
    // camera rays one batch can hold, from the largest expansion seen so far (0: nothing seen yet)
    double fit_items() const {
        if (!(est_nodes_per_item > 0.0)) return 0.0;
        double fit = np.capacity / est_nodes_per_item;
        fit = std::min(fit, gp.job_capacity / std::max(est_jobs_per_item, 1e-9));
        fit = std::min(fit, gp.special_capacity / std::max(est_specials_per_item, 1e-9));
        return fit;
    }
    void forget_estimates() { est_nodes_per_item = est_jobs_per_item = est_specials_per_item = 0.0; }
};

struct rt_scene {
    int device = 0;
    SceneDev dev{};
    std::vector<void *> allocs;
    uint64_t scene_bytes = 0;
    uint64_t n_nodes = 0, n_leaves = 0, bvh_depth = 0;
    int n_cus = 256;
    bool bvh_on_device = false;  // the tree was built by rt_bvh_device.hip.h
    GiTune gi_tune;  // rt_tune.h
    // Side streams of a render call (created once, at the scene's first render that uses them): after k_expand(level) three
    // launches are independent -- the level's GI rays (k_trace_gi, caller's stream), the closest-hit rays of the children
    // it emitted (stream 0) and the shadow rays of its jobs (stream 1) -- and each one's tail (a launch cannot end before
    // its longest ray) is filled by the others' blocks.
    struct Side {
        hipStream_t stream[2] = {nullptr, nullptr};
        hipEvent_t expanded[MAX_LEVELS + 1] = {};  // caller's stream: k_close_expand(level) done
        hipEvent_t traced[MAX_LEVELS + 1] = {};    // stream 0: the children of `level` are traced
        hipEvent_t shadows = nullptr;              // stream 1: every shadow launch done
        bool ready = false;
    } side;

    uint64_t pool_bytes = 24ull << 30;  // upper limit of the wavefront pools; 8 % of the 288 GB
    bool pool_explicit = false;         // set by rt_scene_set_pool_bytes: take it as it is
    uint32_t guess_nodes_per_item = 64; // RenderCtx::guess_nodes_per_item of this scene
    void *pool_mem = nullptr;
    uint64_t pool_mem_bytes = 0;
    RenderCtx ctx;
    // rt_render_multi*: what a scene needs to take part as one shard of a frame (allocated at its first such call)
    struct Multi {
        double *frame = nullptr;    // w*h*3 on the scene's device: the shard's pixels in place
        double *packed = nullptr;   // the shard's tiles, tile-major: what travels to the root device
        uint32_t *tiles = nullptr;  // the shard's tile list (image order)
        uint64_t packed_cap = 0;    // doubles
        uint32_t tiles_cap = 0;     // tiles
        hipStream_t stream = nullptr;
        // on the ROOT scene's device (root_device): where the packed tiles arrive, and the tile list again
        int root_device = -1;
        double *inbox = nullptr;
        uint32_t *tiles_root = nullptr;
        uint64_t inbox_cap = 0;
        uint32_t tiles_root_cap = 0;
        bool peer_enabled = false;
        rt_render_params key{};     // the (params, shard) the tile lists were made for
        uint32_t key_index = 0, key_count = 0;
        uint32_t n_tiles = 0, tile_w = 0, tile_h = 0;
        bool key_valid = false;
    } multi;
    void *scratch = nullptr;  // this device's copy of the library's static scratch (hipGetSymbolAddress at create)

    ~rt_scene() {
        int prev = -1;
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != device) hipSetDevice(device);
        for (void *p : allocs) hipFree(p);
        for (hipStream_t st : side.stream)
            if (st) hipStreamDestroy(st);
        for (uint32_t l = 0; l <= MAX_LEVELS; ++l) {
            if (side.expanded[l]) hipEventDestroy(side.expanded[l]);
            if (side.traced[l]) hipEventDestroy(side.traced[l]);
        }
        if (side.shadows) hipEventDestroy(side.shadows);
        if (pool_mem) hipFree(pool_mem);
        if (ctx.d_ctr) hipFree(ctx.d_ctr);
        if (ctx.h_ctr) hipHostFree(ctx.h_ctr);
        if (ctx.d_tiles) hipFree(ctx.d_tiles);
        if (multi.frame) hipFree(multi.frame);
        if (multi.packed) hipFree(multi.packed);
        if (multi.tiles) hipFree(multi.tiles);
        if (multi.stream) hipStreamDestroy(multi.stream);
        if (multi.inbox || multi.tiles_root) {
            if (multi.root_device != device) hipSetDevice(multi.root_device);
            if (multi.inbox) hipFree(multi.inbox);
            if (multi.tiles_root) hipFree(multi.tiles_root);
        }
        if (prev >= 0) hipSetDevice(prev);
    }
};

namespace {

// Pool shares.  With the GI and shadow rays traced in place a frame's nodes are the camera rays, reflection / refraction
// children and the 1-in-10 GI rays that go on (scene.json: 21 M nodes, 16 M jobs -- one per colour node that is lit or
// casts GI rays --, 7 M specials).  Per node: ray 64 + hit 28 + expansion info 16 + live 4 = 112 bytes; three quarters of
// a job of 80 bytes and of its 16 result bytes; one special of 8 bytes (+ 8 of alignment slack).  Batches are sized from the measured use of every pool.
constexpr uint64_t POOL_BYTES_PER_NODE = 112 + 60 + 12 + 8 + 8;

// Carves the node and job pools out of [mem, mem + bytes).  No allocation.
int carve_pools(void *mem, uint64_t bytes, NodePool *np_out, GiPool *gp_out) {
    uint64_t cap = bytes / POOL_BYTES_PER_NODE;
    cap = std::min<uint64_t>(cap, 1ull << 30);
    cap &= ~uint64_t(255);
    if (cap < 4096) return fail(RT_ERR_NOMEM, "pool budget too small");
    char *p = static_cast<char *>(mem);
    auto take = [&](uint64_t n) {
        char *r = p;
        p += (n + 255) & ~uint64_t(255);
        return r;
    };
    NodePool &np = *np_out;
    np.r0 = reinterpret_cast<Vec2d *>(take(cap * 16));
    np.r1 = reinterpret_cast<Vec2d *>(take(cap * 16));
    np.r2 = reinterpret_cast<Vec2d *>(take(cap * 16));
    np.path = reinterpret_cast<uint64_t *>(take(cap * 8));
    np.t = reinterpret_cast<double *>(take(cap * 8));
    np.bu = reinterpret_cast<double *>(take(cap * 8));
    np.bv = reinterpret_cast<double *>(take(cap * 8));
    np.root = reinterpret_cast<uint32_t *>(take(cap * 4));
    np.meta = reinterpret_cast<uint32_t *>(take(cap * 4));
    np.prim = reinterpret_cast<uint32_t *>(take(cap * 4));
    np.xinfo = reinterpret_cast<ExpandInfo *>(take(cap * 16));
    np.live = reinterpret_cast<uint32_t *>(take(cap * 4));
    np.capacity = static_cast<uint32_t>(cap);
    GiPool &gp = *gp_out;
    gp.job_capacity = static_cast<uint32_t>((cap / 4 * 3) & ~uint64_t(255));
    gp.special_capacity = static_cast<uint32_t>(cap);
    gp.jobs = reinterpret_cast<GiJob *>(take(static_cast<uint64_t>(gp.job_capacity) * sizeof(GiJob)));
    gp.results = reinterpret_cast<JobResult *>(take(static_cast<uint64_t>(gp.job_capacity) * sizeof(JobResult)));
    gp.specials = reinterpret_cast<GiSpecial *>(take(static_cast<uint64_t>(gp.special_capacity) * sizeof(GiSpecial)));
    if (static_cast<uint64_t>(p - static_cast<char *>(mem)) > bytes) return fail(RT_ERR_NOMEM, "internal: pool carve exceeds budget");
    return RT_OK;
}

int ensure_pools(rt_scene *s) {
    RenderCtx &ctx = s->ctx;
    if (s->pool_mem && s->pool_mem_bytes == s->pool_bytes) return RT_OK;
    if (s->pool_mem) {
        hipFree(s->pool_mem);
        s->pool_mem = nullptr;
        s->pool_mem_bytes = 0;
    }
    // Sized from the scene: what a whole frame needs at 64 node slots per camera ray (scene.json uses 8, physical.json 70:
    // a frame that needs more takes several batches), at least 256 MiB, at most the budget (default 24 GiB) -- a 200x200
    // frame asks for 2 GiB, not 24.  A budget the caller set is taken as it is.  Never more than 80 % of what is free.
    uint64_t want = s->pool_bytes;
    if (!s->pool_explicit) {
        const uint64_t frame_items = static_cast<uint64_t>(s->dev.opts.width) * s->dev.opts.height * s->dev.opts.samples_per_pixel;
        want = std::min<uint64_t>(want, std::max<uint64_t>(256ull << 20, frame_items * 64u * POOL_BYTES_PER_NODE));
    }
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && want > free_b / 10 * 8) want = free_b / 10 * 8;
    if (want < (1ull << 20)) return fail(RT_ERR_NOMEM, "less than 1 MiB of device memory free for the wavefront pools");
    const auto t_alloc = std::chrono::steady_clock::now();
    HIP_TRY(hipMalloc(&s->pool_mem, want));
    if (std::getenv("MI355RT_TIMING")) std::fprintf(stderr, "[mi355rt] wavefront pools: hipMalloc of %.2f GiB took %.1f ms\n", want / 1073741824.0, Stager::since(t_alloc));
    s->pool_mem_bytes = s->pool_bytes;  // the budget this allocation answers (may be clamped)
    if (int rc = carve_pools(s->pool_mem, want, &ctx.np, &ctx.gp)) return rc;
    ctx.guess_nodes_per_item = s->guess_nodes_per_item;
    if (!ctx.d_ctr) HIP_TRY(hipMalloc(reinterpret_cast<void **>(&ctx.d_ctr), MAX_SHARDS * sizeof(Counters)));
    if (!ctx.h_ctr) HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&ctx.h_ctr), MAX_SHARDS * sizeof(Counters)));
    ctx.forget_estimates();
    ctx.plan_valid = false;
    return RT_OK;
}

int ensure_side_streams(rt_scene *s) {
    rt_scene::Side &sd = s->side;
    if (sd.ready) return RT_OK;
    for (hipStream_t &st : sd.stream)
        if (!st) HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    for (uint32_t l = 0; l <= MAX_LEVELS; ++l) {
        if (!sd.expanded[l]) HIP_TRY(hipEventCreateWithFlags(&sd.expanded[l], hipEventDisableTiming));
        if (!sd.traced[l]) HIP_TRY(hipEventCreateWithFlags(&sd.traced[l], hipEventDisableTiming));
    }
    if (!sd.shadows) HIP_TRY(hipEventCreateWithFlags(&sd.shadows, hipEventDisableTiming));
    sd.ready = true;
    return RT_OK;
}

struct CallPlan {
    WorkList wl{};
    uint64_t total_items = 0;
};

// Tile partition of a call (host only).  Unsharded: one tile = the window.
int plan_tiles(const rt_render_options &o, const rt_render_params *prm, std::vector<uint32_t> *tiles_out, WorkList *wl_out) {
    rt_render_params p{};
    if (prm) p = *prm;
    uint32_t x0 = p.x0, y0 = p.y0, x1 = p.x1, y1 = p.y1, s0 = p.s0, s1 = p.s1;
    if (x1 == 0 && y1 == 0) {
        x0 = y0 = 0;
        x1 = o.width;
        y1 = o.height;
    }
    if (s1 == 0) {
        s0 = 0;
        s1 = o.samples_per_pixel;
    }
    if (x1 > o.width || y1 > o.height || x0 >= x1 || y0 >= y1 || s0 >= s1 || s1 > o.samples_per_pixel)
        return fail(RT_ERR_INVALID, "render window / sample range out of range");
    std::vector<uint32_t> &tiles = *tiles_out;
    tiles.clear();
    WorkList &wl = *wl_out;
    wl.row_mul = 1;
    if (p.shard_count <= 1) {
        tiles = {x0, y0};
        wl.tile_w = x1 - x0;
        wl.tile_h = y1 - y0;
    } else {
        if (p.shard_index >= p.shard_count) return fail(RT_ERR_INVALID, "shard_index >= shard_count");
        const uint32_t ts = p.tile_size ? p.tile_size : 32u;
        const uint32_t tx0 = x0 / ts, ty0 = y0 / ts, tx1 = (x1 + ts - 1) / ts, ty1 = (y1 + ts - 1) / ts;
        std::vector<std::pair<uint32_t, uint32_t>> keyed;  // (morton, packed xy)
        for (uint32_t ty = ty0; ty < ty1; ++ty)
            for (uint32_t tx = tx0; tx < tx1; ++tx) {
                if (tx > 0xFFFF || ty > 0xFFFF) return fail(RT_ERR_INVALID, "too many tiles");
                keyed.emplace_back(morton2(tx, ty), (ty << 16) | tx);
            }
        std::sort(keyed.begin(), keyed.end());
        for (size_t j = 0; j < keyed.size(); ++j)
            if (j % p.shard_count == p.shard_index) {
                tiles.push_back((keyed[j].second & 0xFFFFu) * ts);
                tiles.push_back((keyed[j].second >> 16) * ts);
            }
        wl.tile_w = wl.tile_h = ts;
    }
    wl.n_tiles = static_cast<uint32_t>(tiles.size() / 2);
    wl.x0 = x0;
    wl.y0 = y0;
    wl.x1 = x1;
    wl.y1 = y1;
    wl.s0 = s0;
    wl.ns = s1 - s0;
    wl.seed = p.seed;
    wl.tile_xy = nullptr;
    return RT_OK;
}

// Camera rays of the first batch of a call (later batches are sized from the expansion the earlier ones showed).
// Nothing seen yet (a scene's first call): a guess from scene facts (RenderCtx::guess_nodes_per_item).  A guess that
// overflows costs a re-run of the batch (physical.json's first frame took 64 ms instead of 49 when every scene started
// from 64 slots per camera ray), one that is too cautious costs batches (scene.json's first frame took two: +9 %), and a
// small pilot batch to measure the expansion costs more than either on scenes with deep trees (a 1/32 batch of
// mesh_test.json: 3.5 ms of launch tails in a 10 ms frame).
uint64_t first_batch_items(const RenderCtx &c, uint64_t total_items, uint32_t ns) {
    uint64_t batch = std::min<uint64_t>(total_items, std::max<uint64_t>(ns, c.np.capacity / c.guess_nodes_per_item));
    const double fit = c.fit_items();
    // a call like the last one (the same window or another shard of it: expansions within a few per cent) goes in ONE batch if
    // it fits with a tenth to spare; else batches of 0.7 x what fits, the margin between the parts of an image
    if (fit > 0.0) batch = static_cast<double>(total_items) <= 0.9 * fit ? total_items : std::min<uint64_t>(total_items, std::max<uint64_t>(ns, static_cast<uint64_t>(0.7 * fit)));
    return batch;
}

// A call that needs several batches takes its camera rays in list order, and a batch is sized from the largest
// expansion (nodes per camera ray) any batch has shown: in image order the batches over the costly part of the image
// set the size for all (sky above a height field: pools a quarter full on average, 333 batches for config 5 instead of
// 112).  Such a call gets a scattered order -- rows of the window, or the tiles of a shard, visited with a
// golden-ratio stride -- so that every batch is a sample of the whole window.  A pixel's result does not depend on
// the order (RNG keyed by pixel and sample); a call that fits one batch keeps image order, which traces ~1 % faster.
uint32_t scatter_stride(uint32_t n, uint32_t max_n) {
    if (n < 16 || n > max_n) return 1u;
    uint32_t m = static_cast<uint32_t>(0.6180339887498949 * n);
    while (std::gcd(m, n) != 1u) ++m;
    return m % n;
}
void scatter_order(std::vector<uint32_t> *tiles, WorkList *wl) {
    if (wl->n_tiles == 1) {
        wl->row_mul = scatter_stride(wl->tile_h, 65535u);  // the kernels multiply in 32 bits
        return;
    }
    const uint32_t n = wl->n_tiles, stride = scatter_stride(n, 0x7FFFFFFFu);
    std::vector<uint32_t> out(tiles->size());
    for (uint32_t j = 0; j < n; ++j) {
        const uint32_t from = static_cast<uint32_t>((static_cast<uint64_t>(j) * stride) % n);
        out[2 * j] = (*tiles)[2 * from];
        out[2 * j + 1] = (*tiles)[2 * from + 1];
    }
    tiles->swap(out);
}

int plan_call(rt_scene *s, RenderCtx &c, const rt_render_params *prm, CallPlan *plan, hipStream_t stream) {
    // the tile partition of a call depends on its window / shard fields only: a frame loop
    // (same params every frame) plans, sorts and uploads its tile list once
    rt_render_params key{};
    if (prm) key = *prm;
    key.seed = 0;
    key.flags = 0;
    key.s0 = key.s1 = 0;
    if (c.plan_valid && std::memcmp(&key, &c.plan_key, sizeof(key)) == 0) {
        *plan = CallPlan{};
        plan->wl = c.plan_wl;
        rt_render_params p{};
        if (prm) p = *prm;
        uint32_t s0 = p.s0, s1 = p.s1;
        if (s1 == 0) {
            s0 = 0;
            s1 = s->dev.opts.samples_per_pixel;
        }
        if (s0 >= s1 || s1 > s->dev.opts.samples_per_pixel) return fail(RT_ERR_INVALID, "render window / sample range out of range");
        plan->wl.s0 = s0;
        plan->wl.ns = s1 - s0;
        plan->wl.seed = p.seed;
        c.plan_wl.s0 = s0;  // the cached work list stays the last call's (rt_render_progressive reads it back)
        c.plan_wl.ns = s1 - s0;
        c.plan_wl.seed = p.seed;
        plan->total_items = static_cast<uint64_t>(plan->wl.n_tiles) * plan->wl.tile_w * plan->wl.tile_h * plan->wl.ns;
        if (plan->total_items > 0xFFFFFFFFull) return fail(RT_ERR_INVALID, "more than 2^32 camera rays in one call");
        if (c.plan_scattered == (first_batch_items(c, plan->total_items, plan->wl.ns) < plan->total_items)) return RT_OK;
        // else: the call now needs (or no longer needs) several batches -- plan again in the other order
    }
    c.plan_valid = false;
    std::vector<uint32_t> tiles;
    WorkList &wl = plan->wl;
    int prc = plan_tiles(s->dev.opts, prm, &tiles, &wl);
    if (prc) return prc;
    plan->total_items = static_cast<uint64_t>(wl.n_tiles) * wl.tile_w * wl.tile_h * wl.ns;
    if (plan->total_items > 0xFFFFFFFFull) return fail(RT_ERR_INVALID, "more than 2^32 camera rays in one call");
    const bool scattered = first_batch_items(c, plan->total_items, wl.ns) < plan->total_items;
    if (scattered) scatter_order(&tiles, &wl);
    if (wl.n_tiles > c.d_tiles_cap) {
        if (!c.can_grow) return fail(RT_ERR_NOMEM, "tile list of this call exceeds the fixed scratch (" + std::to_string(c.d_tiles_cap) + " tiles)");
        if (c.d_tiles) hipFree(c.d_tiles);
        c.d_tiles = nullptr;
        c.d_tiles_cap = 0;
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&c.d_tiles), tiles.size() * sizeof(uint32_t)));
        c.d_tiles_cap = wl.n_tiles;
    }
    if (wl.n_tiles) {
        HIP_TRY(hipMemcpyAsync(c.d_tiles, tiles.data(), tiles.size() * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
        HIP_TRY(hipStreamSynchronize(stream));  // `tiles` is a local
    }
    wl.tile_xy = c.d_tiles;
    c.plan_key = key;
    c.plan_wl = wl;
    c.plan_scattered = scattered;
    c.plan_valid = true;
    return RT_OK;
}

// raytracing_scene.rs:104-105 on the host (the kernels' gi_rays)
uint32_t gi_rays_host(const rt_render_options &o, uint32_t depth) {
    uint32_t d = 1;
    for (uint32_t i = 0; i < depth; ++i) d *= 4u;
    const uint32_t n = o.max_illumination_rays / d;
    return n < 1u ? 1u : n;
}

uint32_t grid_for(const rt_scene *s, uint64_t n, uint32_t blocks_per_cu) {
    uint64_t need = (n + BLOCK - 1) / BLOCK;
    uint64_t cap = static_cast<uint64_t>(s->n_cus) * blocks_per_cu;
    return static_cast<uint32_t>(std::max<uint64_t>(1, std::min(need, cap)));
}

// grid of a sharded launch: a multiple of the shard count (kernels derive the shard from blockIdx.x)
uint32_t shard_grid(const rt_scene *s, const Shards &sh, uint64_t n, uint32_t blocks_per_cu) {
    const uint32_t g = grid_for(s, n, blocks_per_cu);
    return ((g + sh.count - 1) / sh.count) * sh.count;
}

// Shard layout of a batch of n camera-ray items (ns samples per pixel).
Shards plan_shards(const RenderCtx &c, uint64_t n, uint32_t ns) {
    Shards sh{};
    sh.granule = 64u * ns;  // 64 consecutive pixels per deal
    const uint64_t n_gran = (n + sh.granule - 1) / sh.granule;
    uint32_t k = static_cast<uint32_t>(std::min<uint64_t>(MAX_SHARDS, std::max<uint64_t>(1, n_gran / 4)));
    if (k >= 8) k &= ~7u;  // whole shards per XCD (blocks are dealt round-robin over the 8 XCDs)
    sh.count = k;
    if (k == 1) sh.granule = ns;  // no padding items in small batches
    sh.node_cap = c.np.capacity / k;
    sh.job_cap = c.gp.job_capacity / k;
    sh.special_cap = c.gp.special_capacity / k;
    return sh;
}

// dynamic LDS of the traversal kernels: per-lane stack of (max leaf depth + 2) entries (sentinel included)
uint32_t stack_entries(const rt_scene *s) { return static_cast<uint32_t>(s->bvh_depth + 2); }
uint32_t stack_lds_bytes(const rt_scene *s) {
    static const uint32_t extra = std::getenv("MI355RT_EXTRA_LDS") ? std::atoi(std::getenv("MI355RT_EXTRA_LDS")) : 0;  // occupancy experiments
    return static_cast<uint32_t>(s->bvh_depth + 2) * BLOCK * sizeof(uint32_t) + (s->dev.n_top_nodes + s->dev.special[0].n_top + s->dev.special[1].n_top) * TOP_NODE_BYTES + extra;
}

Shards single_shard(const RenderCtx &c) {
    Shards sh{};
    sh.count = 1u;
    sh.node_cap = c.np.capacity;
    sh.job_cap = c.gp.job_capacity;
    sh.special_cap = c.gp.special_capacity;
    sh.granule = 1u;
    return sh;
}

// ---------------------------------------------------------------- the library's static device scratch
// rt_render_device_noalloc renders with every pool, counter and tile list in this array instead of hipMalloc'ed memory.
// Each device the library runs on holds its own copy (at its own address) until the process ends.  The array's name
// is a host-side stand-in, not a device address: host code only ever uses the pointer hipGetSymbolAddress returned for
// the device in question (scratch_for_device, called from rt_scene_create with that device current).
constexpr uint64_t SCRATCH_BYTES = RT_NOALLOC_SCRATCH_BYTES;  // also the extent of `scratch` (file scope, top of this file)
constexpr uint32_t SCRATCH_MAX_TILES = 4096;
constexpr uint64_t SCRATCH_CTR_BYTES = (MAX_SHARDS * sizeof(Counters) + 255) & ~uint64_t(255);
constexpr uint64_t SCRATCH_TILE_BYTES = SCRATCH_MAX_TILES * 2 * sizeof(uint32_t);
constexpr uint64_t SCRATCH_POOL_BYTES = SCRATCH_BYTES - SCRATCH_CTR_BYTES - SCRATCH_TILE_BYTES;

constexpr int MAX_DEVICES = 64;
struct FixedCtx {
    void *ptr = nullptr;  // this device's address of `scratch`
    RenderCtx ctx;        // carved from it once
    Counters *h_ctr = nullptr;  // pinned host copy of the counters, allocated once (set-up), kept until the process ends
};
FixedCtx g_fixed[MAX_DEVICES];
std::mutex g_fixed_mutex;

// Once per device, with that device current: the device address of `scratch`, and the render context carved from it.
int scratch_for_device(int device, void **out) {
    if (device < 0 || device >= MAX_DEVICES) return fail(RT_ERR_UNSUPPORTED, "device index beyond 63");
    std::lock_guard<std::mutex> lock(g_fixed_mutex);
    FixedCtx &f = g_fixed[device];
    if (!f.ptr) {
        void *ptr = nullptr;
        HIP_TRY(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
        if (!ptr) return fail(RT_ERR_DEVICE, "hipGetSymbolAddress(scratch) returned a null address");
        char *base = static_cast<char *>(ptr);
        RenderCtx &c = f.ctx;
        c.d_ctr = reinterpret_cast<Counters *>(base);
        c.d_tiles = reinterpret_cast<uint32_t *>(base + SCRATCH_CTR_BYTES);
        c.d_tiles_cap = SCRATCH_MAX_TILES;
        c.can_grow = false;
        if (!f.h_ctr) HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&f.h_ctr), MAX_SHARDS * sizeof(Counters)));
        c.h_ctr = f.h_ctr;
        if (int rc = carve_pools(base + SCRATCH_CTR_BYTES + SCRATCH_TILE_BYTES, SCRATCH_POOL_BYTES, &c.np, &c.gp)) return rc;
        f.ptr = ptr;
    }
    *out = f.ptr;
    return RT_OK;
}

// Entry points never unwind across the boundary, and run with the scene's device current (restored on exit).
struct DeviceScope {
    int prev = -1;
    bool switched = false;
    int enter(int device) {
        HIP_TRY(hipGetDevice(&prev));
        if (prev != device) {
            HIP_TRY(hipSetDevice(device));
            switched = true;
        }
        return RT_OK;
    }
    ~DeviceScope() {
        if (switched) hipSetDevice(prev);
    }
};
template <typename F>
int guarded(F fn) {
    try {
        return fn();
    } catch (const std::bad_alloc &) {
        return fail(RT_ERR_NOMEM, "out of host memory");
    } catch (const std::exception &e) {
        return fail(RT_ERR_INVALID, std::string("internal error: ") + e.what());
    } catch (...) {
        return fail(RT_ERR_INVALID, "internal error");
    }
}
template <typename F>
int guarded_on(rt_scene *s, F fn) {
    if (!s) return fail(RT_ERR_INVALID, "null scene");
    return guarded([&]() -> int {
        DeviceScope scope;
        if (int rc = scope.enter(s->device)) return rc;
        return fn();
    });
}

// RAII for the temporaries of an entry point
struct DevBuf {
    void *p = nullptr;
    ~DevBuf() {
        if (p) hipFree(p);
    }
    int alloc(size_t bytes) {
        HIP_TRY(hipMalloc(&p, bytes));
        return RT_OK;
    }
    template <typename T>
    T *as() const { return static_cast<T *>(p); }
};
struct EventPair {
    hipEvent_t a = nullptr, b = nullptr;
    ~EventPair() {
        if (a) hipEventDestroy(a);
        if (b) hipEventDestroy(b);
    }
};

int scene_create_impl(const rt_scene_desc *d, rt_scene **out) {
    if (!d || !out) return fail(RT_ERR_INVALID, "null argument");
    *out = nullptr;
    // a count above zero needs its array
    if ((d->n_prims && (!d->prim_type || !d->prim_material || !d->prim_transform || !d->prim_param || !d->prim_normal || !d->prim_tri)) ||
        (d->n_transforms && (!d->world || !d->inv_world)) || (d->n_tris && (!d->tri_pos || !d->tri_nrm || !d->tri_uv)) ||
        (d->n_materials && !d->materials) || (d->n_lights && !d->lights) || (d->n_textures && !d->textures))
        return fail(RT_ERR_INVALID, "null array with a non-zero count in rt_scene_desc");
    const rt_render_options &o = d->opts;
    if (o.width == 0 || o.height == 0 || o.samples_per_pixel == 0)
        return fail(RT_ERR_INVALID, "width, height and samples_per_pixel must be positive");
    if (static_cast<uint64_t>(o.width) * o.height > 0x7FFFFFFFull) return fail(RT_ERR_INVALID, "image too large");
    if (o.max_depth > 6)  // 8_u16.pow(depth) overflows in the reference for depth >= 6 (divide by zero)
        return fail(RT_ERR_UNSUPPORTED, "max_depth > 6: 8_u16.pow(depth) overflows in the reference");
    if (o.samples_per_pixel > 65535 || o.max_reflected_rays > 65535 || o.max_illumination_rays > 65535)
        return fail(RT_ERR_INVALID, "u16 option out of range");
    if (o.max_illumination_rays > 16383) return fail(RT_ERR_UNSUPPORTED, "max_illumination_rays > 16383");
    if (d->n_lights > 255) return fail(RT_ERR_UNSUPPORTED, "more than 255 lights");
    if (d->n_materials >= (1u << 23)) return fail(RT_ERR_UNSUPPORTED, "too many materials");
    uint32_t n_point = 0;
    for (uint32_t i = 0; i < d->n_lights; ++i) n_point += d->lights[i].kind == RT_LIGHT_POINT ? 1u : 0u;
    if (n_point > 32) return fail(RT_ERR_UNSUPPORTED, "more than 32 point lights");
    for (uint32_t i = 0; i < d->n_materials; ++i) {
        const rt_material &m = d->materials[i];
        if (m.kind > RT_MAT_PHYSICAL || m.side > RT_SIDE_BACK) return fail(RT_ERR_INVALID, "bad material kind/side");
        if (m.texture_id >= static_cast<int32_t>(d->n_textures)) return fail(RT_ERR_INVALID, "texture_id out of range");
    }

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(RT_ERR_DEVICE, "no HIP device available (this library has no CPU path)");
    auto sc = std::make_unique<rt_scene>();
    HIP_TRY(hipGetDevice(&sc->device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, sc->device));
    sc->n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;

    // MI355RT_TIMING=1: phase times of this call on stderr (the reference prints its "pre-process scene" time, main.rs:48-56)
    const bool timing = std::getenv("MI355RT_TIMING") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!timing) return;
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "rt_scene_create: %-28s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };

    // ---- flatten primitives: a serial pass checks the description and numbers the bounded primitives, a parallel
    // one fills the shading records and the boxes
    std::vector<ShadePrim> shade(d->n_prims);
    std::vector<PlanePrim> planes;
    std::vector<uint32_t> bounded_ids;
    for (uint32_t i = 0; i < d->n_prims; ++i) {
        const uint32_t type = d->prim_type[i];
        if (type > RT_PRIM_TRIANGLE) return fail(RT_ERR_INVALID, "bad primitive type");
        if (d->prim_material[i] >= d->n_materials || d->prim_transform[i] >= d->n_transforms)
            return fail(RT_ERR_INVALID, "primitive references out of range");
        if (d->prim_transform[i] >= (1u << 28)) return fail(RT_ERR_UNSUPPORTED, "too many transforms");
        if (type == RT_PRIM_TRIANGLE && d->prim_tri[i] >= d->n_tris) return fail(RT_ERR_INVALID, "triangle index out of range");
        if (type != RT_PRIM_PLANE) bounded_ids.push_back(i);
    }
    parallel_for(d->n_prims, [&](size_t i0, size_t i1) {
        for (size_t i = i0; i < i1; ++i) {
            ShadePrim &sh = shade[i];
            std::memset(&sh, 0, sizeof(sh));
            sh.type = d->prim_type[i];
            sh.material = d->prim_material[i];
            sh.xf = d->prim_transform[i];
            sh.tri = sh.type == RT_PRIM_TRIANGLE ? d->prim_tri[i] : 0u;
            sh.param = d->prim_param[i];
            sh.uv_ok = 1;
            if (sh.type == RT_PRIM_PLANE) {
                for (int k = 0; k < 3; ++k) sh.normal[k] = d->prim_normal[3ull * i + k];
                sh.uv_ok = plane_uv_rotation(sh.normal, sh.uv_rot) ? 1u : 0u;
            }
        }
    });
    for (uint32_t i = 0; i < d->n_prims; ++i) {
        const ShadePrim &sh = shade[i];
        if (sh.type != RT_PRIM_PLANE) continue;
        PlanePrim pp{};
        for (int k = 0; k < 3; ++k) pp.normal[k] = sh.normal[k];
        pp.prim_id = i;
        pp.meta = sh.type | (d->materials[sh.material].side << 2) | (sh.xf << 4);
        planes.push_back(pp);
    }
    const size_t n_bounded = bounded_ids.size();
    std::vector<Aabb> boxes(n_bounded);       // reference AABBs (part of the hit predicate)
    std::vector<Aabb> cull_boxes(n_bounded);  // what the BVH is built from
    std::vector<uint32_t> leaf_group(n_bounded);  // per bounded primitive: which primitives it may share a BVH leaf with
    parallel_for(n_bounded, [&](size_t b0, size_t b1) {
        for (size_t b = b0; b < b1; ++b) {
            const uint32_t i = bounded_ids[b];
            const ShadePrim &sh = shade[i];
            const double *w = d->world + 12ull * sh.xf;
            double mn[3], mx[3];
            if (sh.type == RT_PRIM_SPHERE) {  // sphere.rs:133-141
                for (int k = 0; k < 3; ++k) {
                    mn[k] = -sh.param;
                    mx[k] = sh.param;
                }
            } else if (sh.type == RT_PRIM_CUBE) {  // cube.rs:177-186
                double h = sh.param / 2.0;
                for (int k = 0; k < 3; ++k) {
                    mn[k] = -h;
                    mx[k] = h;
                }
            } else {  // triangle.rs:235-251
                const double *p = d->tri_pos + 9ull * sh.tri;
                for (int k = 0; k < 3; ++k) {
                    mn[k] = mx[k] = p[k];
                    for (int v = 1; v < 3; ++v) {
                        mn[k] = std::fmin(mn[k], p[3 * v + k]);
                        mx[k] = std::fmax(mx[k], p[3 * v + k]);
                    }
                }
            }
            boxes[b] = world_aabb(mn, mx, w);
            // Box the BVH is built from.  A triangle's reference AABB is the transform of its
            // object-space box (looser than the triangle when the mesh is rotated); every hit
            // point lies on the triangle itself, so the bounds of the world-space vertices are
            // a valid -- and tighter -- culling volume.  The exact predicate keeps the loose box.
            cull_boxes[b] = boxes[b];
            if (sh.type == RT_PRIM_TRIANGLE) {
                const double *p = d->tri_pos + 9ull * sh.tri;
                Aabb tight{};
                for (int v = 0; v < 3; ++v) {
                    const double x = p[3 * v], y = p[3 * v + 1], z = p[3 * v + 2];
                    const double q[3] = {w[0] * x + w[1] * y + w[2] * z + w[3], w[4] * x + w[5] * y + w[6] * z + w[7],
                                         w[8] * x + w[9] * y + w[10] * z + w[11]};
                    for (int k = 0; k < 3; ++k) {
                        tight.mn[k] = v ? std::fmin(tight.mn[k], q[k]) : q[k];
                        tight.mx[k] = v ? std::fmax(tight.mx[k], q[k]) : q[k];
                    }
                }
                bool finite = true;
                for (int k = 0; k < 3; ++k) finite = finite && std::isfinite(tight.mn[k]) && std::isfinite(tight.mx[k]);
                // never larger than the reference box (intersection of two valid volumes)
                for (int k = 0; finite && k < 3; ++k) {
                    tight.mn[k] = std::fmax(tight.mn[k], boxes[b].mn[k]);
                    tight.mx[k] = std::fmin(tight.mx[k], boxes[b].mx[k]);
                    if (tight.mn[k] > tight.mx[k]) finite = false;
                }
                if (finite) cull_boxes[b] = tight;
            }
            leaf_group[b] = sh.type == RT_PRIM_TRIANGLE ? sh.xf : NO_LEAF_GROUP;  // a mesh's faces share its transform
        }
    });
    if (bounded_ids.size() >= (1ull << 28)) return fail(RT_ERR_UNSUPPORTED, "too many bounded primitives");

    lap("flatten primitives, boxes");

    // ---- BVH: on the device for large scenes (same split rule, rt_bvh_device.hip.h), on the host otherwise.
    // MI355RT_BVH=host|device forces one of them (tests build small scenes with the device builder).
    BvhBuildResult bvh;
    const BvhNode *d_built_nodes = nullptr;
    size_t n_built_nodes = 0;
    {
        const char *want = std::getenv("MI355RT_BVH");
        bool on_device = want ? std::strcmp(want, "device") == 0 : n_bounded >= 65536;
        if (on_device) {  // the device builder orders f64 bounds through integer keys: finite boxes only
            for (size_t b = 0; b < n_bounded && on_device; ++b)
                for (int k = 0; k < 3; ++k)
                    if (!std::isfinite(cull_boxes[b].mn[k]) || !std::isfinite(cull_boxes[b].mx[k])) on_device = false;
        }
        bvhdev::DeviceBuildResult dr;
        if (on_device && n_bounded > 0) {
            std::string why;
            if (!bvhdev::build_bvh_device(cull_boxes, leaf_group, &dr, &why)) {
                // a tree deeper than the traversal stack, or a HIP error of the builder: the host builder takes over (a
                // device that is really gone fails at the uploads below)
                (void)hipGetLastError();
                if (timing) std::fprintf(stderr, "rt_scene_create:   device BVH build gave up (%s): host builder\n", why.c_str());
                on_device = false;
            }
        }
        if (on_device && n_bounded > 0) {
            if (dr.d_nodes) sc->allocs.push_back(const_cast<BvhNode *>(dr.d_nodes));
            d_built_nodes = dr.d_nodes;
            n_built_nodes = dr.n_nodes;
            bvh.order = std::move(dr.order);
            bvh.root_ref = dr.root_ref;
            bvh.max_depth = dr.max_depth;
            bvh.n_leaves = dr.n_leaves;
            bvh.root_box = dr.root_box;
            for (int k = 0; k < 3; ++k) {
                bvh.q_origin[k] = dr.q_origin[k];
                bvh.q_scale[k] = dr.q_scale[k];
            }
            sc->bvh_on_device = true;
            sc->scene_bytes += n_built_nodes * sizeof(BvhNode);
            if (timing) std::fprintf(stderr, "rt_scene_create:   device BVH kernels %.1f ms\n", dr.ms_device);
        } else {
            bvh = build_bvh(cull_boxes, &leaf_group);
            n_built_nodes = bvh.nodes.size();
        }
    }
    lap("BVH build");
    auto leaf_record = [&](uint32_t b) {  // b: index into bounded_ids / boxes
        const uint32_t pid = bounded_ids[b];
        LeafPrim lp;
        std::memset(&lp, 0, sizeof(lp));
        for (int k = 0; k < 3; ++k) {  // BoundingVolume::intersect, bounds.rs:116-121
            lp.center[k] = (boxes[b].mn[k] + boxes[b].mx[k]) * 0.5;
            lp.half[k] = (boxes[b].mx[k] - boxes[b].mn[k]) / 2.0;
        }
        const ShadePrim &sh = shade[pid];
        if (sh.type == RT_PRIM_TRIANGLE) {  // triangle.rs:196-197
            const double *p = d->tri_pos + 9ull * sh.tri;
            for (int k = 0; k < 3; ++k) {
                lp.geom[k] = p[k];
                lp.geom[3 + k] = p[3 + k] - p[k];
                lp.geom[6 + k] = p[6 + k] - p[k];
            }
        } else {
            lp.geom[0] = sh.param;
        }
        lp.prim_id = pid;
        lp.meta = sh.type | (d->materials[sh.material].side << 2) | (sh.xf << 4);
        return lp;
    };
    std::vector<LeafPrim> leaf(bvh.order.size());
    parallel_for(bvh.order.size(), [&](size_t j0, size_t j1) {
        for (size_t j = j0; j < j1; ++j) leaf[j] = leaf_record(bvh.order[j]);
    });
    // get_emissive_phong (:135-167) adds a child only if reflectivity > 0; get_emissive_physical (:263-337)
    // only if max_reflected_rays > 0 or opacity < 1: otherwise an illumination ray's result is the emissive colour
    std::vector<uint32_t> prim_flags(d->n_prims);
    uint32_t n_special[2] = {0, 0};  // [0] emissive or casting child rays, [1] emissive (what is special on the last traced level)
    uint32_t n_physical = 0;
    for (uint32_t i = 0; i < d->n_prims; ++i) {
        const rt_material &m = d->materials[shade[i].material];
        n_physical += m.kind == RT_MAT_PHYSICAL ? 1u : 0u;
        const bool terminal = m.kind == RT_MAT_PHONG ? !(m.reflectivity > 0.0) : (o.max_reflected_rays == 0 && !(m.opacity < 1.0));
        const bool emissive = m.emissive[0] != 0.0 || m.emissive[1] != 0.0 || m.emissive[2] != 0.0;
        prim_flags[i] = (terminal ? PRIM_ILLUM_TERMINAL : 0u) | (emissive ? PRIM_EMISSIVE : 0u);
        if (!terminal || emissive) n_special[0]++;
        if (emissive) n_special[1]++;
    }
    // ---- two-phase GI queries (k_trace_gi): worth considering on the levels whose special primitives are few.  Their
    // trees are built on the host in the main tree's quantisation frame, so one CullRay serves all.
    // MI355RT_GI_TWO_PHASE=0|1 forces the choice (1: on every level, whatever the sets' sizes).
    BvhBuildResult bvh_s[2];
    std::vector<LeafPrim> leaf_s[2];
    bool set_ok[2] = {static_cast<uint64_t>(n_special[0]) * 8u <= d->n_prims, static_cast<uint64_t>(n_special[1]) * 8u <= d->n_prims};
    // the starting configuration (what a one-frame process and small renders run with: a rule from scene facts, rt_tune.h)
    // and the switches the tuner may flip
    //   two-phase queries pay when (a) the second phase, an any-hit search within max_occlusion_distance D, is much
    //   shorter than a closest-hit query: D <= extent / 4 (D = 1 in scenes of extent 5 ... 550: GI rays 0.20 -> 0.095 ms per
    //   million on mesh.json), and (b) the first phase, the closest SPECIAL hit, is cheap and seldom finds anything: the
    //   special primitives' boxes together are at most half the root box's area (scene.json: the room, a cube around
    //   everything, is special -- every ray has a special hit far away, the first phase comes on top: +6 %);
    //   the last level's illumination nodes apart (k_trace_illum_last) where the tree is deep enough for its any-hit
    //   form to matter (2,048 bounded primitives: mesh.json +8 %, mesh_test.json +15 %; within +-3 % elsewhere).
    uint32_t start = 0;
    if (n_bounded) {
        auto area = [](const Aabb &b) {
            const double dx = b.mx[0] - b.mn[0], dy = b.mx[1] - b.mn[1], dz = b.mx[2] - b.mn[2];
            return 2.0 * (dx * dy + dy * dz + dz * dx);
        };
        double extent = 0.0, special_area[2] = {0.0, 0.0};
        for (int k = 0; k < 3; ++k) extent = std::fmax(extent, bvh.root_box.mx[k] - bvh.root_box.mn[k]);
        for (size_t b = 0; b < n_bounded; ++b) {
            const uint32_t f = prim_flags[bounded_ids[b]];
            if (is_special_flags(f)) special_area[0] += area(cull_boxes[b]);
            if (f & PRIM_EMISSIVE) special_area[1] += area(cull_boxes[b]);
        }
        const double root_area = area(bvh.root_box);
        const bool short_occlusion = o.max_occlusion_distance * 4.0 <= extent;
        if (short_occlusion && special_area[0] <= 0.5 * root_area) start |= GiTune::GI_INNER;
        if (short_occlusion && special_area[1] <= 0.5 * root_area) start |= GiTune::GI_LAST;
        if ((start & GiTune::GI_LAST) && n_bounded >= 2048) start |= GiTune::ILLUM_LAST;
    }
    sc->gi_tune.configure(set_ok, start, std::getenv("MI355RT_GI_TWO_PHASE"), std::getenv("MI355RT_GI_WAVES"), std::getenv("MI355RT_GI_CFG"));
    for (int q = 0; q < 2; ++q) {
        if (!set_ok[q] || n_special[q] == 0) continue;
        std::vector<uint32_t> sub;  // indices into bounded_ids
        for (size_t b = 0; b < n_bounded; ++b) {
            const uint32_t f = prim_flags[bounded_ids[b]];
            if ((f & PRIM_EMISSIVE) || (q == 0 && !(f & PRIM_ILLUM_TERMINAL))) sub.push_back(static_cast<uint32_t>(b));
        }
        std::vector<Aabb> sub_boxes(sub.size());
        std::vector<uint32_t> sub_group(sub.size());
        for (size_t j = 0; j < sub.size(); ++j) {
            sub_boxes[j] = cull_boxes[sub[j]];
            sub_group[j] = leaf_group[sub[j]];
        }
        bvh_s[q] = build_bvh(sub_boxes, &sub_group, 0, bvh.q_origin, bvh.q_scale);
        leaf_s[q].resize(bvh_s[q].order.size());
        for (size_t j = 0; j < leaf_s[q].size(); ++j) leaf_s[q][j] = leaf_record(sub[bvh_s[q].order[j]]);
    }
    sc->guess_nodes_per_item = (static_cast<uint64_t>(n_physical) * 4u >= d->n_prims && o.max_reflected_rays > 1) ? 96u : 16u;
    lap("special primitives");
    sc->n_nodes = n_built_nodes;
    sc->n_leaves = bvh.n_leaves;
    sc->bvh_depth = std::max(bvh.max_depth, std::max(bvh_s[0].max_depth, bvh_s[1].max_depth));
    lap("leaf records");

    // ---- upload
    SceneDev &dev = sc->dev;
    dev.opts = o;
    for (int k = 0; k < 3; ++k) dev.cam_pos[k] = d->camera.position[k];
    for (int k = 0; k < 9; ++k) dev.cam_to_world[k] = d->camera.cam_to_world[k];
    dev.fov_scale = std::tan(d->camera.fov_deg * (3.14159265358979323846 / 180.0) / 2.0);  // :82-84
    dev.root_ref = bvh.root_ref;
    {   // top of the trees kept in LDS by the traversal kernels (TOP_NODE_BYTES a node): as many nodes as fit next to the
        // per-lane stacks without costing a block per CU (four 256-thread blocks share 160 KB) -- the small trees of
        // the special primitives first (at most 64 nodes each), then the main tree (at most 512)
        const uint32_t stack_bytes = (static_cast<uint32_t>(sc->bvh_depth) + 2u) * BLOCK * static_cast<uint32_t>(sizeof(uint32_t));
        const uint32_t budget = 40u * 1024u - 3072u;  // per block; 3 KB for the kernels' static LDS (work map; k_trace_shadow's prefix sums and result words)
        uint32_t left = stack_bytes < budget ? (budget - stack_bytes) / TOP_NODE_BYTES : 0u;
        if (const char *e = std::getenv("MI355RT_TOP_NODES")) left = std::min<uint32_t>(left, static_cast<uint32_t>(std::atoi(e)));  // experiments; tests/test_lds_top.py
        for (int q = 0; q < 2; ++q) {
            dev.special[q].n_top = static_cast<uint32_t>(std::min<size_t>(bvh_s[q].nodes.size(), std::min(left, 64u)));
            left -= dev.special[q].n_top;
        }
        dev.n_top_nodes = static_cast<uint32_t>(std::min<size_t>(n_built_nodes, std::min(left, 512u)));
    }
    for (int q = 0; q < 2; ++q) {
        dev.special[q].root_ref = bvh_s[q].root_ref;
        dev.special[q].count = n_special[q];
        dev.special[q].ok = set_ok[q] ? 1u : 0u;
    }
    dev.gi_two_phase = sc->gi_tune.pick() & 7u;
    dev.n_planes = static_cast<uint32_t>(planes.size());
    dev.n_bounded = static_cast<uint32_t>(bounded_ids.size());
    dev.n_prims = d->n_prims;
    dev.scene_scale = 0.0;
    for (int k = 0; k < 3; ++k) {
        dev.q_origin[k] = bvh.q_origin[k];
        dev.q_scale[k] = bvh.q_scale[k];
    }
    if (!boxes.empty())
        for (int k = 0; k < 3; ++k)
            dev.scene_scale = std::fmax(dev.scene_scale, std::fmax(std::fabs(bvh.root_box.mn[k]), std::fabs(bvh.root_box.mx[k])));
    dev.n_lights = d->n_lights;
    dev.n_point_lights = n_point;

    std::vector<DLight> lights(d->n_lights);
    for (uint32_t i = 0; i < d->n_lights; ++i) {
        lights[i].kind = d->lights[i].kind;
        lights[i]._pad = 0;
        for (int k = 0; k < 3; ++k) {
            lights[i].color[k] = d->lights[i].color[k];
            lights[i].position[k] = d->lights[i].position[k];
        }
        lights[i].intensity = d->lights[i].intensity;
    }
    std::vector<DTexture> texs(d->n_textures);
    std::vector<uint8_t> texels;
    for (uint32_t i = 0; i < d->n_textures; ++i) {
        const rt_texture &t = d->textures[i];
        if (t.width == 0 || t.height == 0 || !t.rgb8) return fail(RT_ERR_INVALID, "empty texture");
        texs[i].width = t.width;
        texs[i].height = t.height;
        texs[i].offset = texels.size();
        texels.insert(texels.end(), t.rgb8, t.rgb8 + static_cast<size_t>(t.width) * t.height * 3);
    }
    int rc;
    uint64_t *bytes = &sc->scene_bytes;
    Stager st;
    if (d_built_nodes) dev.nodes = d_built_nodes;
    else if ((rc = upload(st, bvh.nodes, &dev.nodes, sc->allocs, bytes))) return rc;
    if ((rc = upload(st, leaf, &dev.leaf_prims, sc->allocs, bytes))) return rc;
    if ((rc = upload(st, planes, &dev.planes, sc->allocs, bytes))) return rc;
    for (int q = 0; q < 2; ++q) {
        if ((rc = upload(st, bvh_s[q].nodes, &dev.special[q].nodes, sc->allocs, bytes))) return rc;
        if ((rc = upload(st, leaf_s[q], &dev.special[q].leaf_prims, sc->allocs, bytes))) return rc;
    }
    if ((rc = upload(st, shade, &dev.shade_prims, sc->allocs, bytes))) return rc;
    if ((rc = upload(st, prim_flags, &dev.prim_flags, sc->allocs, bytes))) return rc;
    if ((rc = upload(st, d->inv_world, 12ull * d->n_transforms, &dev.inv_world, sc->allocs, bytes))) return rc;
    if ((rc = upload(st, d->tri_nrm, 9ull * d->n_tris, &dev.tri_nrm, sc->allocs, bytes))) return rc;
    if ((rc = upload(st, d->tri_uv, 6ull * d->n_tris, &dev.tri_uv, sc->allocs, bytes))) return rc;
    if ((rc = upload(st, d->materials, static_cast<size_t>(d->n_materials), &dev.materials, sc->allocs, bytes))) return rc;
    if ((rc = upload(st, lights, &dev.lights, sc->allocs, bytes))) return rc;
    std::vector<DLight> point_lights;
    for (const DLight &l : lights)
        if (l.kind == RT_LIGHT_POINT) point_lights.push_back(l);
    if ((rc = upload(st, point_lights, &dev.point_lights, sc->allocs, bytes))) return rc;
    if ((rc = upload(st, texs, &dev.textures, sc->allocs, bytes))) return rc;
    if ((rc = upload(st, texels, &dev.texels, sc->allocs, bytes))) return rc;
    lap("upload");
    if (timing) std::fprintf(stderr, "rt_scene_create:   of which hipMalloc %.1f ms, staging fill %.1f ms, DMA wait %.1f ms\n", st.ms_alloc, st.ms_fill, st.ms_wait);
    if ((rc = scratch_for_device(sc->device, &sc->scratch))) return rc;
    *out = sc.release();
    return RT_OK;
}

}  // namespace

extern "C" {

const char *rt_last_error(void) { return g_error.c_str(); }
const char *rt_version(void) { return "mi355rt 0.2 (gfx950, f64 wavefront)"; }

int rt_scene_create(const rt_scene_desc *d, rt_scene **out) {
    return guarded([&]() -> int { return scene_create_impl(d, out); });
}

void rt_scene_destroy(rt_scene *scene) {
    try {
        delete scene;
    } catch (...) {
    }
}

int rt_scene_set_pool_bytes(rt_scene *scene, uint64_t bytes) {
    if (!scene) return fail(RT_ERR_INVALID, "null scene");
    if (bytes < (1ull << 20)) return fail(RT_ERR_INVALID, "pool budget below 1 MiB");
    scene->pool_bytes = bytes;
    scene->pool_explicit = true;
    return RT_OK;
}

int rt_shard_tiles(const rt_render_options *opts, const rt_render_params *params, uint32_t *tile_xy, uint32_t cap,
                   uint32_t *n_tiles, uint32_t *tile_w, uint32_t *tile_h) {
    if (!opts || !n_tiles) return fail(RT_ERR_INVALID, "null argument");
    return guarded([&]() -> int {
        std::vector<uint32_t> tiles;
        WorkList wl{};
        int rc = plan_tiles(*opts, params, &tiles, &wl);
        if (rc) return rc;
        *n_tiles = wl.n_tiles;
        if (tile_w) *tile_w = wl.tile_w;
        if (tile_h) *tile_h = wl.tile_h;
        if (tile_xy)
            for (uint32_t i = 0; i < wl.n_tiles && i < cap; ++i) {
                tile_xy[2 * i] = tiles[2 * i];
                tile_xy[2 * i + 1] = tiles[2 * i + 1];
            }
        return RT_OK;
    });
}

int rt_scene_info(rt_scene *scene, uint64_t *out8) {
    if (!scene || !out8) return fail(RT_ERR_INVALID, "null argument");
    out8[0] = scene->n_nodes;
    out8[1] = scene->n_leaves;
    out8[2] = scene->bvh_depth;
    out8[3] = scene->scene_bytes;
    out8[4] = scene->dev.n_bounded;
    out8[5] = scene->dev.n_planes;
    out8[6] = scene->ctx.np.capacity;
    out8[7] = scene->ctx.gp.job_capacity;
    return RT_OK;
}

}  // extern "C"

namespace {

// One render call in the wavefront state `ctx` (the scene's own pools, or the static scratch): batches of camera rays
// through raygen -> (trace, expand) per level -> shadow -> resolve per level -> accumulate.  The scene's device is
// current.  Allocates nothing itself when `stats` is NULL and the plan fits ctx's tile list (events are only created
// for statistics).
int render_impl(rt_scene *s, RenderCtx &ctx, const rt_render_params *params, double *d_rgb_accum, double *d_albedo_accum,
                double *d_normal, uint64_t *ray_count, rt_render_stats *stats, hipStream_t stream) {
    const uint32_t aov = (d_albedo_accum || d_normal) ? 1u : 0u;
    int rc;
    CallPlan plan;
    if ((rc = plan_call(s, ctx, params, &plan, stream))) return rc;
    const WorkList &wl = plan.wl;
    const bool count_work = params && (params->flags & RT_FLAG_COUNT_WORK);
    const uint32_t max_depth = s->dev.opts.max_depth;
    const NodePool &np = ctx.np;
    const GiPool &gp = ctx.gp;
    Counters *d_ctr = ctx.d_ctr;

    rt_render_stats st{};
    EventTimer timer;
    timer.enabled = stats != nullptr;
    timer.stream = stream;
    // which form of GI query this call uses; while that is still being tried out, k_trace_gi is timed (the scene's own
    // context only: the no-allocation path creates no events and takes the choice as it stands)
    const uint32_t gi_cfg = s->gi_tune.pick();
    const bool gi_waves3 = (gi_cfg & GiTune::WAVES3) != 0;
    s->dev.gi_two_phase = gi_cfg & 7u;
    // a call can only be used by the tuner if it traces at least TUNE_MIN_RAYS GI rays: one that cannot (fewer camera
    // rays than that many hemisphere rays of level 0 need) is not timed -- no events are created for it
    constexpr uint64_t TUNE_MIN_RAYS = 1000000u;
    const bool gi_tuning = s->gi_tune.exploring() && &ctx == &s->ctx;
    const bool gi_can_qualify = plan.total_items * gi_rays_host(s->dev.opts, 0) >= TUNE_MIN_RAYS && !count_work;
    const bool gi_warmup = gi_tuning && s->gi_tune.step < 0;  // the scene's first sizeable call: counted, not timed (a one-frame process runs only this)
    EventTimer gi_timer;
    gi_timer.enabled = gi_tuning && !gi_warmup && gi_can_qualify && !stats;  // with stats the main timer has the figure
    gi_timer.stream = stream;
    gi_timer.chaining = false;
    // Overlap (rt_scene::Side): where launches have tails worth filling -- a tree too large for LDS / L1 (rays wait on memory:
    // mesh_test.json, 41 K triangles, 10.0 -> 8.4 ms; mesh.json -2 %) or a call of a fraction of a frame (a 1/8 shard of
    // scene.json 3.13 -> 2.99 ms).  A whole frame of a small scene is bound by VALU issue in every one of the three launches
    // and gains nothing (scene.json 17.0 ms either way): it stays on the caller's stream, one kernel at a time.  The scene's
    // own context only, and not while the switches are being tried out (the closest-hit kernels are then timed one after the
    // other) -- the scene's first call, which the tuner only counts, does overlap: a one-frame process runs nothing else.  MI355RT_OVERLAP=0 / 1 forces either.
    static const char *overlap_env = std::getenv("MI355RT_OVERLAP");
    const bool overlap_wanted = overlap_env ? std::strcmp(overlap_env, "0") != 0 : (s->dev.n_bounded >= 4096 || plan.total_items <= 655360);
    const bool overlap = overlap_wanted && &ctx == &s->ctx && (!gi_tuning || gi_warmup) && max_depth > 0;
    if (overlap)
        if ((rc = ensure_side_streams(s))) return rc;
    EventTimer timer_side[2], timer_off;  // spans of the launches on the side streams; timer_off: never enabled
    for (int q = 0; q < 2; ++q) {
        timer_side[q].enabled = stats != nullptr && overlap;
        timer_side[q].stream = overlap ? s->side.stream[q] : nullptr;
    }
    double gi_ms = 0.0;
    uint64_t gi_rays_call = 0, tune_rays_call = 0;
    EventPair call_ev;
    if (stats) {
        HIP_TRY(hipEventCreate(&call_ev.a));
        HIP_TRY(hipEventCreate(&call_ev.b));
        HIP_TRY(hipEventRecord(call_ev.a, stream));
    }

    uint64_t item0 = 0;
    double call_max[3] = {0.0, 0.0, 0.0};  // nodes, jobs, GI specials per camera ray: the largest of the call's batches
    // first batch: modest guess, later batches sized from the measured expansion
    uint64_t batch = first_batch_items(ctx, plan.total_items, wl.ns);
    const uint32_t gmax = static_cast<uint32_t>(s->n_cus) * 8u;
    while (item0 < plan.total_items) {
        uint64_t n = std::min<uint64_t>(batch, plan.total_items - item0);
        n = std::min<uint64_t>(n, np.capacity);
        n -= n % wl.ns;
        if (n == 0) return fail(RT_ERR_NOMEM, "pool budget too small for one pixel's samples");
        const uint32_t n32 = static_cast<uint32_t>(n), i0 = static_cast<uint32_t>(item0);
        const Shards sh = plan_shards(ctx, n, wl.ns);
        // traversal: 48 blocks per CU, four resident -- the hardware deals out the rest as blocks finish, which evens out
        // rays of very different cost (4 persistent blocks per CU: scene.json +7 %, the 1M-triangle scene +37 %)
        // (a small batch -- a 1/8 shard of a frame -- gets fewer blocks, down to 12 per CU: a block's start-up, 16 KB of
        // tree into LDS, is then shared by as many rays as in a full frame; k_trace_gi of a scene.json shard 1.53 -> 1.37 ms)
        const uint32_t trace_bpc = static_cast<uint32_t>(std::min<uint64_t>(48, std::max<uint64_t>(12, n * 48 / 2560000)));
        const uint32_t g_level = shard_grid(s, sh, static_cast<uint64_t>(s->n_cus) * trace_bpc * BLOCK, trace_bpc);
        const uint32_t g_wide = shard_grid(s, sh, static_cast<uint64_t>(gmax) * BLOCK, 8);

        timer.begin(CLS_MISC);
        hipLaunchKernelGGL(k_begin_batch, dim3(1), dim3(64), 0, stream, d_ctr, sh, n32);
        timer.end();
        timer.begin(CLS_RAYGEN);
        hipLaunchKernelGGL(k_raygen, dim3(shard_grid(s, sh, n, 8)), dim3(BLOCK), 0, stream, s->dev, wl, np, d_ctr, sh, i0, n32);
        timer.end();
        if (max_depth > 0) {
            // closest-hit launch over the nodes of `level` (`what`: all of them, or the children k_expand(level - 1) emitted) and,
            // where the last traced level's illumination nodes take two-phase queries over the emissive set, their kernel
            auto launch_closest = [&](uint32_t level, uint32_t what, hipStream_t on, EventTimer &t, EventTimer &t_tune) {
                const uint32_t g = level == 0 ? shard_grid(s, sh, n, 48) : g_level;
                const uint32_t illum_apart = level > 0 && level + 1 == max_depth && (s->dev.gi_two_phase & 4u) && s->dev.special[1].ok ? 1u : 0u;
                t.begin(CLS_CLOSEST);
                t_tune.begin(CLS_CLOSEST);
                if (count_work) hipLaunchKernelGGL((k_trace_closest<true, true>), dim3(g), dim3(BLOCK), stack_lds_bytes(s), on, s->dev, np, d_ctr, sh, level, stack_entries(s), illum_apart, what);
                else hipLaunchKernelGGL((k_trace_closest<false, true>), dim3(g), dim3(BLOCK), stack_lds_bytes(s), on, s->dev, np, d_ctr, sh, level, stack_entries(s), illum_apart, what);
                t.end();
                t_tune.end();
                if (illum_apart) {
                    t.begin(CLS_ILLUM);
                    t_tune.begin(CLS_ILLUM);
                    auto *kern = count_work ? (gi_waves3 ? k_trace_illum_last<true, 3> : k_trace_illum_last<true, 4>) : (gi_waves3 ? k_trace_illum_last<false, 3> : k_trace_illum_last<false, 4>);
                    hipLaunchKernelGGL(kern, dim3(g), dim3(BLOCK), stack_lds_bytes(s), on, s->dev, np, d_ctr, sh, level, stack_entries(s), what);
                    t.end();
                    t_tune.end();
                }
            };
            auto launch_shadow = [&](uint32_t what, uint32_t level, hipStream_t on, EventTimer &t) {
                t.begin(CLS_SHADOW);
                if (count_work) hipLaunchKernelGGL(k_trace_shadow<true>, dim3(g_level), dim3(BLOCK), stack_lds_bytes(s), on, s->dev, gp, d_ctr, sh, stack_entries(s), what, level);
                else hipLaunchKernelGGL(k_trace_shadow<false>, dim3(g_level), dim3(BLOCK), stack_lds_bytes(s), on, s->dev, gp, d_ctr, sh, stack_entries(s), what, level);
                t.end();
            };
            for (uint32_t level = 0; level < max_depth; ++level) {
                if (level == 0 || !overlap) launch_closest(level, MAP_NODES, stream, timer, gi_timer);
                else {  // the children were traced beside k_trace_gi(level)
                    HIP_TRY(hipStreamWaitEvent(stream, s->side.traced[level], 0));
                    timer.chain = false;
                }
                timer.begin(CLS_EXPAND);
                hipLaunchKernelGGL(k_build_live, dim3(level == 0 ? shard_grid(s, sh, n / 4 + 1, 8) : g_wide), dim3(BLOCK), 0, stream, np, d_ctr, sh, level);
                const uint32_t ge = level == 0 ? shard_grid(s, sh, n, 8) : g_wide;
                if (count_work) hipLaunchKernelGGL(k_expand<true>, dim3(ge), dim3(BLOCK), 0, stream, s->dev, wl, np, gp, d_ctr, sh, level);
                else hipLaunchKernelGGL(k_expand<false>, dim3(ge), dim3(BLOCK), 0, stream, s->dev, wl, np, gp, d_ctr, sh, level);
                if (overlap) {  // what k_expand emitted is final: its children and its jobs can be traced beside the level's GI rays
                    hipLaunchKernelGGL(k_close_expand, dim3(1), dim3(64), 0, stream, d_ctr, sh, level);
                    timer.end();
                    HIP_TRY(hipEventRecord(s->side.expanded[level], stream));
                    timer.chain = false;  // an event of another kind sits between this span and the next
                    if (level + 1 < max_depth) {
                        HIP_TRY(hipStreamWaitEvent(s->side.stream[0], s->side.expanded[level], 0));
                        launch_closest(level + 1, MAP_CHILDREN, s->side.stream[0], timer_side[0], timer_off);
                        HIP_TRY(hipEventRecord(s->side.traced[level + 1], s->side.stream[0]));
                        timer_side[0].chain = false;
                    }
                    HIP_TRY(hipStreamWaitEvent(s->side.stream[1], s->side.expanded[level], 0));
                    launch_shadow(MAP_LEVEL_JOBS, level, s->side.stream[1], timer_side[1]);
                } else {
                    timer.end();
                }
                if (level + 1 < max_depth && gi_rays_host(s->dev.opts, level) <= GI_MAX_INPLACE) {  // the GI rays of this level's hits, in place
                    timer.begin(CLS_GI);
                    gi_timer.begin(CLS_GI);
                    // the level's form of query is compiled in: a build with one traversal fits its register budget
                    const uint32_t set = level + 2 < max_depth ? 0u : 1u;  // the GI rays' own children are traced (set 0) or cut off (set 1)
                    const bool tp = ((s->dev.gi_two_phase >> set) & 1u) && s->dev.special[set].ok;
                    auto *kern = tp ? (count_work ? (gi_waves3 ? k_trace_gi<true, 3, true> : k_trace_gi<true, 4, true>) : (gi_waves3 ? k_trace_gi<false, 3, true> : k_trace_gi<false, 4, true>))
                                    : (count_work ? (gi_waves3 ? k_trace_gi<true, 3, false> : k_trace_gi<true, 4, false>) : (gi_waves3 ? k_trace_gi<false, 3, false> : k_trace_gi<false, 4, false>));
                    hipLaunchKernelGGL(kern, dim3(g_level), dim3(BLOCK), stack_lds_bytes(s), stream, s->dev, wl, np, gp, d_ctr, sh, level + 1, stack_entries(s));
                    timer.end();
                    gi_timer.end();
                }
                timer.begin(CLS_EXPAND);
                hipLaunchKernelGGL(k_advance_level, dim3(1), dim3(64), 0, stream, d_ctr, sh, level);
                timer.end();
            }
            if (overlap) {
                HIP_TRY(hipEventRecord(s->side.shadows, s->side.stream[1]));
                HIP_TRY(hipStreamWaitEvent(stream, s->side.shadows, 0));
                timer.chain = false;
            } else {
                launch_shadow(MAP_JOBS, 0u, stream, timer);
            }
            for (uint32_t level = max_depth; level-- > 0;) {
                timer.begin(CLS_RESOLVE);
                const dim3 gr(level == 0 ? shard_grid(s, sh, n, 8) : g_wide);
                if (level > 0) hipLaunchKernelGGL(k_resolve<true>, gr, dim3(BLOCK), 0, stream, s->dev, np, gp, d_ctr, sh, level, 0u);  // camera rays are colour nodes
                hipLaunchKernelGGL(k_resolve<false>, gr, dim3(BLOCK), 0, stream, s->dev, np, gp, d_ctr, sh, level, aov);
                timer.end();
            }
            timer.begin(CLS_ACCUM);
            hipLaunchKernelGGL(k_accumulate, dim3(shard_grid(s, sh, n / wl.ns, 8)), dim3(BLOCK), 0, stream, s->dev, wl, np, d_ctr, sh,
                               d_rgb_accum, d_albedo_accum, d_normal);
            timer.end();
        }
        HIP_TRY(hipMemcpyAsync(ctx.h_ctr, d_ctr, sh.count * sizeof(Counters), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        HIP_TRY(hipGetLastError());
        Counters c{};  // sum over shards
        for (uint32_t k = 0; k < sh.count; ++k) {
            const Counters &ck = ctx.h_ctr[k];
            for (uint32_t l = 0; l <= MAX_LEVELS; ++l) c.level_count[l] += ck.level_count[l];
            c.node_alloc += ck.node_alloc;
            c.shadow_rays += ck.shadow_rays;
            c.job_alloc += ck.job_alloc;
            c.special_alloc += ck.special_alloc;
            c.gi_rays += ck.gi_rays;
            c.gi_materialised += ck.gi_materialised;
            c.illum_last_rays += ck.illum_last_rays;
            c.overflow |= ck.overflow;
            c.active_camera += ck.active_camera;
            c.nodes_visited_closest += ck.nodes_visited_closest;
            c.prims_tested_closest += ck.prims_tested_closest;
            c.nodes_visited_shadow += ck.nodes_visited_shadow;
            c.prims_tested_shadow += ck.prims_tested_shadow;
            for (uint32_t l = 0; l < MAX_LEVELS; ++l)
                for (uint32_t q = 0; q < 4; ++q) c.census[l][q] += ck.census[l][q];
            c.wave_node_steps += ck.wave_node_steps;
            c.wave_leaf_rounds += ck.wave_leaf_rounds;
            c.wave_iterations += ck.wave_iterations;
            for (uint32_t q = 0; q < 16; ++q) c.leaf_census[q] += ck.leaf_census[q];
            for (uint32_t q = 0; q < 4; ++q) c.leaf_lanes[q] += ck.leaf_lanes[q];
            c.wave_max_steps = std::max(c.wave_max_steps, ck.wave_max_steps);
        }
        if (count_work && std::getenv("MI355RT_CENSUS")) {
            std::fprintf(stderr, "[mi355rt census] closest kernel, wave level: %llu node visits, %llu leaf rounds over %llu waves (all levels); busiest wave %llu node visits\n",
                         c.wave_node_steps, c.wave_leaf_rounds, c.wave_iterations, c.wave_max_steps);
            std::fprintf(stderr, "[mi355rt census] leaf rounds by content (T S C X=exact box):");
            for (uint32_t q = 0; q < 16; ++q)
                if (c.leaf_census[q]) std::fprintf(stderr, " %s%s%s%s=%llu", q & 1 ? "T" : "", q & 2 ? "S" : "", q & 4 ? "C" : "", q & 8 ? "X" : "", c.leaf_census[q]);
            std::fprintf(stderr, "\n[mi355rt census] leaf-round lanes %llu, exact-box lanes %llu, hit lanes %llu\n", c.leaf_lanes[0], c.leaf_lanes[1], c.leaf_lanes[2]);
            for (uint32_t l = 0; l < max_depth && l < MAX_LEVELS; ++l)
                std::fprintf(stderr, "[mi355rt census] level %u: colour %llu illum %llu miss %llu no-emit %llu\n", l, c.census[l][0],
                             c.census[l][1], c.census[l][2], c.census[l][3]);
        }
        float cls_ms[CLS_COUNT] = {0};
        uint32_t cls_n[CLS_COUNT] = {0};
        timer.collect(cls_ms, cls_n, CLS_COUNT);
        gi_timer.collect(cls_ms, cls_n, CLS_COUNT);  // only one of the two timers runs
        for (EventTimer &t : timer_side) t.collect(cls_ms, cls_n, CLS_COUNT);  // launches that ran beside others: their spans overlap the caller's stream's
        if (c.overflow) {
            if (n <= wl.ns) return fail(RT_ERR_NOMEM, "pool budget too small for one pixel's ray tree");
            batch = std::max<uint64_t>(wl.ns, (n / 2) - (n / 2) % wl.ns);
            ctx.forget_estimates();
            st.retries += 1;
            continue;  // re-run the same range with a smaller batch; nothing was accumulated
        }
        // ---- statistics (reference counting rule: +1 per traced get_color/get_illumination, +1 per shadow ray)
        uint64_t closest = c.active_camera;
        for (uint32_t l = 1; l <= max_depth && l <= MAX_LEVELS; ++l) closest += c.level_count[l];
        closest += c.gi_rays - c.gi_materialised;  // GI rays traced in place; the ones that got a node are in level_count already
        if (max_depth == 0) closest = 0;
        st.closest_rays += closest;
        st.shadow_rays += c.shadow_rays;
        st.camera_rays += c.active_camera;
        st.nodes_visited_closest += c.nodes_visited_closest;
        st.prims_tested_closest += c.prims_tested_closest;
        st.nodes_visited_shadow += c.nodes_visited_shadow;
        st.prims_tested_shadow += c.prims_tested_shadow;
        st.batches += 1;
        st.ms_raygen += cls_ms[CLS_RAYGEN];
        st.ms_closest += cls_ms[CLS_CLOSEST];
        st.ms_gi += cls_ms[CLS_GI];
        st.launches_gi += cls_n[CLS_GI];
        st.gi_rays += c.gi_rays;
        st.ms_illum += cls_ms[CLS_ILLUM];
        st.launches_illum += cls_n[CLS_ILLUM];
        st.illum_rays += c.illum_last_rays;
        gi_ms += cls_ms[CLS_GI] + cls_ms[CLS_ILLUM] + cls_ms[CLS_CLOSEST];  // the tuner compares all closest-hit work: variants move rays between the kernels
        gi_rays_call += c.gi_rays;
        tune_rays_call += closest;
        st.ms_shadow += cls_ms[CLS_SHADOW];
        st.ms_expand += cls_ms[CLS_EXPAND];
        st.ms_resolve += cls_ms[CLS_RESOLVE];
        st.ms_accum += cls_ms[CLS_ACCUM];
        st.launches_closest += cls_n[CLS_CLOSEST];
        st.launches_shadow += cls_n[CLS_SHADOW];
        st.launches_other += cls_n[CLS_RAYGEN] + cls_n[CLS_EXPAND] + cls_n[CLS_RESOLVE] + cls_n[CLS_ACCUM] + cls_n[CLS_MISC];

        // expansion varies over the image: size later batches (and later calls) from the
        // LARGEST expansion seen, not the last one, so a cheap final batch cannot
        // provoke an overflow + re-run at the start of the next frame
        const double dn = static_cast<double>(n);
        call_max[0] = std::max(call_max[0], static_cast<double>(c.node_alloc) / dn);
        call_max[1] = std::max(call_max[1], static_cast<double>(c.job_alloc) / dn);
        call_max[2] = std::max(call_max[2], static_cast<double>(c.special_alloc) / dn);
        ctx.est_nodes_per_item = std::max(ctx.est_nodes_per_item, call_max[0]);
        ctx.est_jobs_per_item = std::max(ctx.est_jobs_per_item, call_max[1]);
        ctx.est_specials_per_item = std::max(ctx.est_specials_per_item, call_max[2]);
        batch = std::max<uint64_t>(wl.ns, static_cast<uint64_t>(0.7 * ctx.fit_items()));
        item0 += n;
    }
    if (call_max[0] > 0.0) {  // forget stale maxima of earlier calls (other windows / shards)
        ctx.est_nodes_per_item = call_max[0];
        ctx.est_jobs_per_item = call_max[1];
        ctx.est_specials_per_item = call_max[2];
    }
    const bool gi_usable = gi_can_qualify && st.retries == 0 && gi_rays_call >= TUNE_MIN_RAYS && (gi_ms > 0.0 || gi_warmup);
    if (gi_tuning && !gi_usable) s->gi_tune.skip();
    if (gi_tuning && gi_usable) {
        GiTune &t = s->gi_tune;
        t.record(gi_cfg, gi_ms / (1e-6 * static_cast<double>(tune_rays_call)));
        if (!t.exploring() && std::getenv("MI355RT_TIMING")) {
            std::fprintf(stderr, "[mi355rt] closest-hit kernels, ms per million closest-hit rays:");
            for (int i = 0; i <= t.n_dims; ++i)
                std::fprintf(stderr, " [%s%s%s%s]=%.4f", t.tried_cfg[i] & 1u ? "I" : "-", t.tried_cfg[i] & 2u ? "L" : "-", t.tried_cfg[i] & 4u ? "N" : "-", t.tried_cfg[i] & 8u ? "3" : "4",
                             t.tried_ms[i]);
            std::fprintf(stderr, " -> [%s%s%s%s]  (I / L: two-phase GI queries on inner / last level, N: last level's illumination nodes apart, 3 / 4: waves per SIMD)\n",
                         t.cfg & 1u ? "I" : "-", t.cfg & 2u ? "L" : "-", t.cfg & 4u ? "N" : "-", t.cfg & 8u ? "3" : "4");
        }
    }
    st.ray_count = st.closest_rays + st.shadow_rays;
    st.gi_two_phase = gi_cfg & 7u;
    st.gi_waves = gi_waves3 ? 3u : 4u;
    st.overlapped = overlap ? 1u : 0u;
    if (stats) {
        HIP_TRY(hipEventRecord(call_ev.b, stream));
        HIP_TRY(hipEventSynchronize(call_ev.b));
        hipEventElapsedTime(&st.ms_total, call_ev.a, call_ev.b);
        *stats = st;
    }
    if (ray_count) *ray_count += st.ray_count;
    return RT_OK;
}

uint32_t grid_1d(uint64_t n, uint32_t cap) { return static_cast<uint32_t>(std::max<uint64_t>(1, std::min<uint64_t>((n + BLOCK - 1) / BLOCK, cap))); }

}  // namespace

extern "C" {

int rt_scene_bvh(rt_scene *s, void *nodes32, uint64_t node_capacity, uint32_t *leaf_prim_ids, uint64_t leaf_capacity, uint64_t *info4) {
    if (!info4) return fail(RT_ERR_INVALID, "null argument");
    return guarded_on(s, [&]() -> int {
        info4[0] = s->n_nodes;
        info4[1] = s->dev.n_bounded;
        info4[2] = s->dev.root_ref;
        info4[3] = s->bvh_on_device ? 1u : 0u;
        if (nodes32) {
            if (node_capacity < s->n_nodes) return fail(RT_ERR_INVALID, "node buffer too small");
            if (s->n_nodes) HIP_TRY(hipMemcpy(nodes32, s->dev.nodes, s->n_nodes * sizeof(BvhNode), hipMemcpyDeviceToHost));
        }
        if (leaf_prim_ids) {
            if (leaf_capacity < s->dev.n_bounded) return fail(RT_ERR_INVALID, "leaf buffer too small");
            std::vector<LeafPrim> leaf(s->dev.n_bounded);
            if (!leaf.empty()) HIP_TRY(hipMemcpy(leaf.data(), s->dev.leaf_prims, leaf.size() * sizeof(LeafPrim), hipMemcpyDeviceToHost));
            for (size_t i = 0; i < leaf.size(); ++i) leaf_prim_ids[i] = leaf[i].prim_id;
        }
        return RT_OK;
    });
}

int rt_render_device(rt_scene *s, const rt_render_params *params, double *d_rgb_accum, uint64_t *ray_count,
                     rt_render_stats *stats, void *stream_v) {
    return rt_render_aov_device(s, params, d_rgb_accum, nullptr, nullptr, ray_count, stats, stream_v);
}

int rt_render_aov_device(rt_scene *s, const rt_render_params *params, double *d_rgb_accum, double *d_albedo_accum,
                         double *d_normal, uint64_t *ray_count, rt_render_stats *stats, void *stream_v) {
    if (!s || !d_rgb_accum) return fail(RT_ERR_INVALID, "null argument");
    return guarded_on(s, [&]() -> int {
        if (int rc = ensure_pools(s)) return rc;
        return render_impl(s, s->ctx, params, d_rgb_accum, d_albedo_accum, d_normal, ray_count, stats, static_cast<hipStream_t>(stream_v));
    });
}

int rt_render_device_noalloc(rt_scene *s, const rt_render_params *params, double *d_rgb_accum, uint64_t *ray_count, void *stream_v) {
    if (!s || !d_rgb_accum) return fail(RT_ERR_INVALID, "null argument");
    return guarded_on(s, [&]() -> int {
        if (!s->scratch) return fail(RT_ERR_DEVICE, "the static scratch of this device was not resolved at rt_scene_create");
        // the size limit: camera rays of the call (window pixels x samples), and its tile list
        const rt_render_options &o = s->dev.opts;
        rt_render_params p{};
        if (params) p = *params;
        const uint64_t wx = (p.x1 == 0 && p.y1 == 0) ? o.width : (p.x1 > p.x0 ? p.x1 - p.x0 : 0);
        const uint64_t wy = (p.x1 == 0 && p.y1 == 0) ? o.height : (p.y1 > p.y0 ? p.y1 - p.y0 : 0);
        const uint64_t ns = p.s1 == 0 ? o.samples_per_pixel : (p.s1 > p.s0 ? p.s1 - p.s0 : 0);
        uint64_t rays = wx * wy * ns;
        if (p.shard_count > 1) rays = rays / p.shard_count + 1;
        if (rays > RT_NOALLOC_MAX_CAMERA_RAYS)
            return fail(RT_ERR_NOMEM, "rt_render_device_noalloc: " + std::to_string(rays) + " camera rays in this call, the limit is " +
                                          std::to_string(static_cast<uint64_t>(RT_NOALLOC_MAX_CAMERA_RAYS)) + " (window pixels x samples); split the window or the sample range");
        if (p.flags & RT_FLAG_COUNT_WORK) return fail(RT_ERR_INVALID, "rt_render_device_noalloc takes no flags");
        FixedCtx &f = g_fixed[s->device];
        if (f.ptr != s->scratch) return fail(RT_ERR_DEVICE, "internal: scratch address mismatch");
        // the scratch is shared by every scene on this device: its plan cache and expansion estimates belong to the last caller
        f.ctx.plan_valid = false;
        f.ctx.forget_estimates();
        f.ctx.guess_nodes_per_item = s->guess_nodes_per_item;
        return render_impl(s, f.ctx, params, d_rgb_accum, nullptr, nullptr, ray_count, nullptr, static_cast<hipStream_t>(stream_v));
    });
}

int rt_render_aov(rt_scene *s, const rt_render_params *params, double *rgb_accum, double *albedo, double *normal,
                  uint64_t *ray_count, rt_render_stats *stats) {
    return guarded_on(s, [&]() -> int {
        const uint64_t n3 = static_cast<uint64_t>(s->dev.opts.width) * s->dev.opts.height * 3;
        const double inv_spp = 1.0 / static_cast<double>(s->dev.opts.samples_per_pixel);
        DevBuf buf;  // [rgb | albedo | normal]
        if (int rc = buf.alloc(3 * n3 * sizeof(double))) return rc;
        double *d_buf = buf.as<double>();
        HIP_TRY(hipMemset(d_buf, 0, 3 * n3 * sizeof(double)));
        if (int rc = ensure_pools(s)) return rc;
        if (int rc = render_impl(s, s->ctx, params, d_buf, albedo ? d_buf + n3 : nullptr, normal ? d_buf + 2 * n3 : nullptr, ray_count, stats, nullptr)) return rc;
        if (rgb_accum) HIP_TRY(hipMemcpy(rgb_accum, d_buf, n3 * sizeof(double), hipMemcpyDeviceToHost));
        if (albedo) HIP_TRY(hipMemcpy(albedo, d_buf + n3, n3 * sizeof(double), hipMemcpyDeviceToHost));
        if (normal) HIP_TRY(hipMemcpy(normal, d_buf + 2 * n3, n3 * sizeof(double), hipMemcpyDeviceToHost));
        if (albedo && s->dev.opts.samples_per_pixel != 1)  // :616 `albedo *= inv_samples`, :619 clamp (one sample: get_color's own clamp)
            for (uint64_t i = 0; i < n3; ++i) {
                const double a = albedo[i] * inv_spp;
                albedo[i] = a < 0.0 ? 0.0 : (a > 1.0 ? 1.0 : a);
            }
        return RT_OK;
    });
}

int rt_tonemap_device(rt_scene *s, const double *d_rgb_accum, uint8_t *d_rgba8, void *stream_v) {
    if (!s || !d_rgb_accum || !d_rgba8) return fail(RT_ERR_INVALID, "null argument");
    return guarded_on(s, [&]() -> int {
        hipStream_t stream = static_cast<hipStream_t>(stream_v);
        const uint32_t n = s->dev.opts.width * s->dev.opts.height;
        hipLaunchKernelGGL(k_tonemap, dim3(grid_for(s, n, 8)), dim3(BLOCK), 0, stream, n, s->dev.opts.samples_per_pixel, d_rgb_accum, d_rgba8);
        HIP_TRY(hipGetLastError());
        return RT_OK;
    });
}

int rt_render(rt_scene *s, const rt_render_params *params, double *rgb_accum, uint8_t *rgba8, uint64_t *ray_count,
              rt_render_stats *stats) {
    return guarded_on(s, [&]() -> int {
        const uint64_t n = static_cast<uint64_t>(s->dev.opts.width) * s->dev.opts.height;
        DevBuf acc, img;
        if (int rc = acc.alloc(n * 3 * sizeof(double))) return rc;
        HIP_TRY(hipMemset(acc.p, 0, n * 3 * sizeof(double)));
        if (int rc = ensure_pools(s)) return rc;
        if (int rc = render_impl(s, s->ctx, params, acc.as<double>(), nullptr, nullptr, ray_count, stats, nullptr)) return rc;
        if (rgb_accum) HIP_TRY(hipMemcpy(rgb_accum, acc.p, n * 3 * sizeof(double), hipMemcpyDeviceToHost));
        if (rgba8) {
            if (int rc = img.alloc(n * 4)) return rc;
            hipLaunchKernelGGL(k_tonemap, dim3(grid_for(s, n, 8)), dim3(BLOCK), 0, nullptr, static_cast<uint32_t>(n), s->dev.opts.samples_per_pixel,
                               acc.as<double>(), img.as<uint8_t>());
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipMemcpy(rgba8, img.p, n * 4, hipMemcpyDeviceToHost));
        }
        return RT_OK;
    });
}

int rt_render_progressive(rt_scene *s, const rt_render_params *params, uint32_t n_passes, uint32_t *argb, rt_progress_fn on_pass,
                          void *user, double *rgb_accum, uint64_t *ray_count) {
    if (!argb) return fail(RT_ERR_INVALID, "null argument");
    return guarded_on(s, [&]() -> int {
        rt_render_params p{};
        if (params) p = *params;
        if (p.shard_count > 1) return fail(RT_ERR_INVALID, "rt_render_progressive deals the tiles itself: shard_count must be 0 or 1");
        if (n_passes == 0) n_passes = 1;
        const uint64_t n = static_cast<uint64_t>(s->dev.opts.width) * s->dev.opts.height;
        DevBuf acc, img;
        if (int rc = acc.alloc(n * 3 * sizeof(double))) return rc;
        if (int rc = img.alloc(n * 4)) return rc;
        HIP_TRY(hipMemset(acc.p, 0, n * 3 * sizeof(double)));
        HIP_TRY(hipMemset(img.p, 0, n * 4));  // the reference's buffer starts as zeros (:800), not as opaque black
        if (int rc = ensure_pools(s)) return rc;
        uint64_t rays = 0;
        for (uint32_t pass = 0; pass < n_passes; ++pass) {
            rt_render_params q = p;
            if (n_passes > 1) {
                q.shard_index = pass;
                q.shard_count = n_passes;
                if (!q.tile_size) q.tile_size = 8;
            }
            if (int rc = render_impl(s, s->ctx, &q, acc.as<double>(), nullptr, nullptr, &rays, nullptr, nullptr)) return rc;
            // the pixels of this pass are complete (all their samples): to_argb_u32 of their colour (:816-819)
            const WorkList &wl = s->ctx.plan_wl;
            const uint64_t npx = static_cast<uint64_t>(wl.n_tiles) * wl.tile_w * wl.tile_h;
            if (npx)
                hipLaunchKernelGGL(k_tonemap_argb_tiles, dim3(grid_1d(npx, 2048)), dim3(BLOCK), 0, nullptr, s->dev, wl, s->dev.opts.samples_per_pixel, acc.as<double>(),
                                   img.as<uint32_t>());  // the divisor is the scene's sample count, as in rt_render's rgba8 (:614), whatever range this call rendered
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipMemcpy(argb, img.p, n * 4, hipMemcpyDeviceToHost));
            if (on_pass && on_pass(user, pass + 1, n_passes, rays) != 0) break;  // the caller closed its window (:865)
        }
        if (rgb_accum) HIP_TRY(hipMemcpy(rgb_accum, acc.p, n * 3 * sizeof(double), hipMemcpyDeviceToHost));
        if (ray_count) *ray_count += rays;
        return RT_OK;
    });
}

int rt_query_primary(rt_scene *s, uint32_t *prim_id, double *t, double *normal3) {
    if (!prim_id || !t || !normal3) return fail(RT_ERR_INVALID, "null argument");
    return guarded_on(s, [&]() -> int {
        if (int rc = ensure_pools(s)) return rc;
        RenderCtx &ctx = s->ctx;
        rt_render_params prm{};
        prm.s0 = 0;
        prm.s1 = 1;
        CallPlan plan;
        if (int rc = plan_call(s, ctx, &prm, &plan, nullptr)) return rc;
        const uint64_t npx = static_cast<uint64_t>(s->dev.opts.width) * s->dev.opts.height;
        DevBuf b_id, b_t, b_n;
        if (int rc = b_id.alloc(npx * 4)) return rc;
        if (int rc = b_t.alloc(npx * 8)) return rc;
        if (int rc = b_n.alloc(npx * 24)) return rc;
        uint64_t item0 = 0;
        while (item0 < plan.total_items) {
            const uint32_t n = static_cast<uint32_t>(std::min<uint64_t>(plan.total_items - item0, ctx.np.capacity));
            const Shards one = single_shard(ctx);
            hipLaunchKernelGGL(k_begin_batch, dim3(1), dim3(64), 0, nullptr, ctx.d_ctr, one, n);
            hipLaunchKernelGGL(k_raygen, dim3(grid_for(s, n, 8)), dim3(BLOCK), 0, nullptr, s->dev, plan.wl, ctx.np, ctx.d_ctr, one, static_cast<uint32_t>(item0), n);
            hipLaunchKernelGGL((k_trace_closest<false, false>), dim3(grid_for(s, n, 4)), dim3(BLOCK), stack_lds_bytes(s), nullptr, s->dev, ctx.np, ctx.d_ctr, one, 0u,
                               stack_entries(s));
            hipLaunchKernelGGL(k_primary_aov, dim3(grid_for(s, n, 8)), dim3(BLOCK), 0, nullptr, s->dev, plan.wl, ctx.np, n, b_id.as<uint32_t>(), b_t.as<double>(),
                               b_n.as<double>());
            item0 += n;
        }
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(prim_id, b_id.p, npx * 4, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(t, b_t.p, npx * 8, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(normal3, b_n.p, npx * 24, hipMemcpyDeviceToHost));
        return RT_OK;
    });
}

int rt_trace_rays(rt_scene *s, uint32_t n, const double *origin, const double *dir, const double *tmax, int shadow,
                  uint32_t *prim_id, double *t) {
    if (!origin || !dir || !prim_id || (shadow && !tmax)) return fail(RT_ERR_INVALID, "null argument");
    return guarded_on(s, [&]() -> int {
        if (int rc = ensure_pools(s)) return rc;
        RenderCtx &ctx = s->ctx;
        const NodePool &np = ctx.np;
        std::vector<double> col;
        std::vector<uint32_t> meta;
        uint32_t done = 0;
        while (done < n) {
            const uint32_t m = std::min<uint32_t>(n - done, np.capacity);
            col.resize(2ull * m);
            // the ray as the node pool holds it: (o.x, o.y) (o.z, d.x) (d.y, d.z)
            auto put = [&](Vec2d *dst, const double *a, int ca, const double *b, int cb) -> hipError_t {
                for (uint32_t i = 0; i < m; ++i) {
                    col[2ull * i] = a[3ull * (done + i) + ca];
                    col[2ull * i + 1] = b[3ull * (done + i) + cb];
                }
                return hipMemcpy(dst, col.data(), 2ull * m * sizeof(double), hipMemcpyHostToDevice);
            };
            HIP_TRY(put(np.r0, origin, 0, origin, 1));
            HIP_TRY(put(np.r1, origin, 2, dir, 0));
            HIP_TRY(put(np.r2, dir, 1, dir, 2));
            meta.assign(m, 1u);  // depth 1 secondary
            HIP_TRY(hipMemcpy(np.meta, meta.data(), m * 4, hipMemcpyHostToDevice));
            const Shards one = single_shard(ctx);
            hipLaunchKernelGGL(k_begin_batch, dim3(1), dim3(64), 0, nullptr, ctx.d_ctr, one, m);
            if (shadow) {
                HIP_TRY(hipMemcpy(np.t, tmax + done, m * sizeof(double), hipMemcpyHostToDevice));
                hipLaunchKernelGGL(k_trace_shadow_probe, dim3(grid_for(s, m, 4)), dim3(BLOCK), stack_lds_bytes(s), nullptr, s->dev, np, m, stack_entries(s));
            } else {
                hipLaunchKernelGGL((k_trace_closest<false, false>), dim3(grid_for(s, m, 4)), dim3(BLOCK), stack_lds_bytes(s), nullptr, s->dev, np, ctx.d_ctr, one, 0u,
                                   stack_entries(s));
            }
            HIP_TRY(hipDeviceSynchronize());
            HIP_TRY(hipMemcpy(prim_id + done, np.prim, m * 4, hipMemcpyDeviceToHost));
            if (t) {
                if (shadow) std::fill(t + done, t + done + m, 0.0);
                else HIP_TRY(hipMemcpy(t + done, np.t, m * sizeof(double), hipMemcpyDeviceToHost));
            }
            done += m;
        }
        if (t && !shadow)
            for (uint32_t i = 0; i < n; ++i)
                if (prim_id[i] == RT_NO_HIT) t[i] = 0.0;
        return RT_OK;
    });
}

}  // extern "C"

namespace {

// (re)allocate a device buffer of at least `need` elements on the CURRENT device
template <typename T, typename N>
int grow(T **buf, N *cap, uint64_t need) {
    if (*buf && *cap >= need) return RT_OK;
    if (*buf) hipFree(*buf);
    *buf = nullptr;
    *cap = 0;
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(buf), need * sizeof(T)));
    *cap = static_cast<N>(need);
    return RT_OK;
}

// One shard of a multi-device frame, on its own host thread: render the tiles dealt to scene `index` of `count` into
// the scene's frame buffer, pack them, send them to the root device.  The root shard (index 0) renders straight into
// the caller's buffer.  Returns with the shard's stream drained.
int multi_shard(rt_scene *s, rt_scene *root, uint32_t index, uint32_t count, const rt_render_params &base, double *d_root_accum,
                hipStream_t root_stream, uint64_t *rays, rt_render_stats *stats) {
    DeviceScope scope;
    if (int rc = scope.enter(s->device)) return rc;
    rt_scene::Multi &m = s->multi;
    const rt_render_options &o = s->dev.opts;
    rt_render_params p = base;
    p.shard_index = index;
    p.shard_count = count;
    if (!p.tile_size) p.tile_size = 8;  // one 64-pixel granule: the finest deal, the best balance
    if (int rc = ensure_pools(s)) return rc;
    if (index == 0) return render_impl(s, s->ctx, &p, d_root_accum, nullptr, nullptr, rays, stats, root_stream);

    if (!m.stream) HIP_TRY(hipStreamCreateWithFlags(&m.stream, hipStreamNonBlocking));
    // tile lists of this shard, here and on the root device (image order: rt_shard_tiles' order)
    rt_render_params key = p;
    key.seed = 0;
    key.flags = 0;
    key.s0 = key.s1 = 0;
    const uint64_t n3 = static_cast<uint64_t>(o.width) * o.height * 3;
    if (!m.frame) HIP_TRY(hipMalloc(reinterpret_cast<void **>(&m.frame), n3 * sizeof(double)));
    if (!m.key_valid || std::memcmp(&key, &m.key, sizeof(key)) != 0 || m.root_device != root->device) {
        m.key_valid = false;
        std::vector<uint32_t> tiles;
        WorkList wl{};
        if (int rc = plan_tiles(o, &p, &tiles, &wl)) return rc;
        m.n_tiles = wl.n_tiles;
        m.tile_w = wl.tile_w;
        m.tile_h = wl.tile_h;
        const uint64_t packed_need = std::max<uint64_t>(1, static_cast<uint64_t>(wl.n_tiles) * wl.tile_w * wl.tile_h * 3);
        if (int rc = grow(&m.tiles, &m.tiles_cap, std::max<uint32_t>(1u, wl.n_tiles) * 2ull)) return rc;
        if (int rc = grow(&m.packed, &m.packed_cap, packed_need)) return rc;
        if (wl.n_tiles) HIP_TRY(hipMemcpy(m.tiles, tiles.data(), tiles.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        {   // the root device's side
            if (m.root_device >= 0 && m.root_device != root->device && (m.inbox || m.tiles_root)) {  // the root moved: start over there
                DeviceScope old;
                if (int rc = old.enter(m.root_device)) return rc;
                if (m.inbox) hipFree(m.inbox);
                if (m.tiles_root) hipFree(m.tiles_root);
                m.inbox = nullptr;
                m.tiles_root = nullptr;
                m.inbox_cap = 0;
                m.tiles_root_cap = 0;
                m.peer_enabled = false;
            }
            DeviceScope on_root;
            if (int rc = on_root.enter(root->device)) return rc;
            m.root_device = root->device;
            if (int rc = grow(&m.tiles_root, &m.tiles_root_cap, std::max<uint32_t>(1u, wl.n_tiles) * 2ull)) return rc;
            if (int rc = grow(&m.inbox, &m.inbox_cap, packed_need)) return rc;
            if (wl.n_tiles) HIP_TRY(hipMemcpy(m.tiles_root, tiles.data(), tiles.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        }
        if (!m.peer_enabled && root->device != s->device) {  // direct xGMI copies; without it the copy is staged through the host
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, s->device, root->device) == hipSuccess && can) {
                const hipError_t e = hipDeviceEnablePeerAccess(root->device, 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
            }
            (void)hipGetLastError();
            m.peer_enabled = true;
        }
        m.key = key;
        m.key_valid = true;
    }
    HIP_TRY(hipMemsetAsync(m.frame, 0, n3 * sizeof(double), m.stream));
    if (int rc = render_impl(s, s->ctx, &p, m.frame, nullptr, nullptr, rays, stats, m.stream)) return rc;
    if (m.n_tiles) {
        const uint64_t total = static_cast<uint64_t>(m.n_tiles) * m.tile_w * m.tile_h * 3;
        hipLaunchKernelGGL(k_pack_tiles, dim3(grid_1d(total, 4096)), dim3(BLOCK), 0, m.stream, m.tiles, m.n_tiles, m.tile_w, m.tile_h, o.width, o.height,
                           m.frame, m.packed);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyPeerAsync(m.inbox, root->device, m.packed, s->device, total * sizeof(double), m.stream));
    }
    HIP_TRY(hipStreamSynchronize(m.stream));
    return RT_OK;
}

int render_multi_impl(rt_scene *const *scenes, uint32_t n, const rt_render_params *params, double *d_rgb_accum, uint64_t *ray_count,
                      rt_render_stats *stats, hipStream_t stream) {
    if (!scenes || n == 0 || !d_rgb_accum) return fail(RT_ERR_INVALID, "null argument");
    if (n > 64) return fail(RT_ERR_UNSUPPORTED, "more than 64 scenes in one rt_render_multi call");
    rt_render_params base{};
    if (params) base = *params;
    if (base.shard_count > 1) return fail(RT_ERR_INVALID, "rt_render_multi deals the tiles itself: shard_count must be 0 or 1");
    rt_scene *root = scenes[0];
    for (uint32_t i = 0; i < n; ++i) {
        if (!scenes[i]) return fail(RT_ERR_INVALID, "null scene");
        for (uint32_t j = 0; j < i; ++j)
            if (scenes[j] == scenes[i]) return fail(RT_ERR_INVALID, "the same scene handle twice in one rt_render_multi call");
        const rt_render_options &a = scenes[i]->dev.opts, &b = root->dev.opts;
        if (a.width != b.width || a.height != b.height || a.samples_per_pixel != b.samples_per_pixel || a.max_depth != b.max_depth ||
            scenes[i]->dev.n_prims != root->dev.n_prims)
            return fail(RT_ERR_INVALID, "rt_render_multi: the scenes are not replicas of one scene (options or primitive counts differ)");
    }
    if (n == 1) {
        DeviceScope scope;
        if (int rc = scope.enter(root->device)) return rc;
        if (int rc = ensure_pools(root)) return rc;
        return render_impl(root, root->ctx, &base, d_rgb_accum, nullptr, nullptr, ray_count, stats, stream);
    }
    // one host thread per scene: each drives its own device (or its own stream of a shared device)
    std::vector<int> rcs(n, RT_OK);
    std::vector<std::string> errs(n);
    std::vector<uint64_t> rays(n, 0);
    std::vector<std::thread> pool;
    pool.reserve(n);
    bool spawn_failed = false;
    for (uint32_t i = 0; i < n && !spawn_failed; ++i) {
        try {
            pool.emplace_back([&, i] {
                rcs[i] = guarded([&]() -> int { return multi_shard(scenes[i], root, i, n, base, d_rgb_accum, stream, &rays[i], stats ? stats + i : nullptr); });
                if (rcs[i]) errs[i] = g_error;
            });
        } catch (...) {  // std::system_error (no more threads) / bad_alloc: the threads already running must be joined before
            spawn_failed = true;  // anything unwinds, or their destructors terminate the process
        }
    }
    for (auto &t : pool) t.join();
    if (spawn_failed) return fail(RT_ERR_NOMEM, "rt_render_multi: could not start a host thread per scene");
    for (uint32_t i = 0; i < n; ++i)
        if (rcs[i]) return fail(rcs[i], "rt_render_multi, scene " + std::to_string(i) + ": " + errs[i]);
    // exchange step, root side: every other shard's tiles are in its inbox on this device
    DeviceScope scope;
    if (int rc = scope.enter(root->device)) return rc;
    const rt_render_options &o = root->dev.opts;
    for (uint32_t i = 1; i < n; ++i) {
        const rt_scene::Multi &m = scenes[i]->multi;
        if (!m.n_tiles) continue;
        const uint64_t total = static_cast<uint64_t>(m.n_tiles) * m.tile_w * m.tile_h * 3;
        hipLaunchKernelGGL(k_unpack_add_tiles, dim3(grid_1d(total, 4096)), dim3(BLOCK), 0, stream, m.tiles_root, m.n_tiles, m.tile_w, m.tile_h, o.width,
                           o.height, m.inbox, d_rgb_accum);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(stream));  // the inboxes are free for the next call
    if (ray_count)
        for (uint32_t i = 0; i < n; ++i) *ray_count += rays[i];
    return RT_OK;
}

}  // namespace

extern "C" {

int rt_render_multi_device(rt_scene *const *scenes, uint32_t n_scenes, const rt_render_params *params, double *d_rgb_accum,
                           uint64_t *ray_count, rt_render_stats *stats, void *stream) {
    return guarded([&]() -> int { return render_multi_impl(scenes, n_scenes, params, d_rgb_accum, ray_count, stats, static_cast<hipStream_t>(stream)); });
}

int rt_render_multi(rt_scene *const *scenes, uint32_t n_scenes, const rt_render_params *params, double *rgb_accum, uint8_t *rgba8,
                    uint64_t *ray_count, rt_render_stats *stats) {
    if (!scenes || n_scenes == 0 || !scenes[0]) return fail(RT_ERR_INVALID, "null argument");
    rt_scene *root = scenes[0];
    return guarded_on(root, [&]() -> int {
        const uint64_t n = static_cast<uint64_t>(root->dev.opts.width) * root->dev.opts.height;
        DevBuf acc, img;
        if (int rc = acc.alloc(n * 3 * sizeof(double))) return rc;
        HIP_TRY(hipMemset(acc.p, 0, n * 3 * sizeof(double)));
        if (int rc = render_multi_impl(scenes, n_scenes, params, acc.as<double>(), ray_count, stats, nullptr)) return rc;
        if (rgb_accum) HIP_TRY(hipMemcpy(rgb_accum, acc.p, n * 3 * sizeof(double), hipMemcpyDeviceToHost));
        if (rgba8) {
            if (int rc = img.alloc(n * 4)) return rc;
            hipLaunchKernelGGL(k_tonemap, dim3(grid_for(root, n, 8)), dim3(BLOCK), 0, nullptr, static_cast<uint32_t>(n), root->dev.opts.samples_per_pixel,
                               acc.as<double>(), img.as<uint8_t>());
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipMemcpy(rgba8, img.p, n * 4, hipMemcpyDeviceToHost));
        }
        return RT_OK;
    });
}

}  // extern "C"
