// Binned-SAH BVH2 builder (host).  The tree only accelerates: every bounded
// primitive keeps its own exact world-AABB slab test in the kernels
// (bounds.rs:172-181), node boxes are exact unions of those AABBs, and the
// kernels test node boxes conservatively, so tree shape never changes results.
#include "rt_bvh.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <thread>
#include <utility>
#include <limits>

namespace rt {
namespace {

constexpr int kBins = 16;
constexpr double kIntersectCost = 3.0;  // one primitive test ~ three node visits (f64, 9 divisions)
// Leaf size.  An analytic primitive (sphere, cube) gets a leaf of its own: the parent's child box IS its own AABB, so
// the node test doubles as the leaf's pre-filter.  Triangles of one mesh share leaves of up to kMeshLeaf: a mesh
// triangle's box is hit 20-30 times more often than the triangle, and with one triangle per leaf the wave went
// through a leaf round for 6-9 of its 64 lanes (mesh.json -12 %, mesh_test.json -17 %, 1M-triangle scene -8 %
// with two per leaf; four per leaf gains more on the 41k-triangle mesh and loses on the 1M-triangle one).
constexpr uint32_t kMeshLeaf = 2;

// f64 -> f32 towards -inf / +inf (node boxes must contain the exact boxes)
float round_down(double x) {
    float f = static_cast<float>(x);
    if (static_cast<double>(f) > x) f = std::nextafterf(f, -std::numeric_limits<float>::infinity());
    return f;
}
float round_up(double x) {
    float f = static_cast<float>(x);
    if (static_cast<double>(f) < x) f = std::nextafterf(f, std::numeric_limits<float>::infinity());
    return f;
}

// Nodes of one subtree, numbered from 0.  The top of the tree and every subtree job build into an arena of their own;
// the arenas are then laid end to end (interior refs shifted) -- and the breadth-first renumbering at the end makes
// the final numbering independent of how the work was cut up, so the tree is the same for any thread count.
struct Arena {
    std::vector<BvhNode> nodes;
    std::vector<std::pair<Aabb, Aabb>> child_boxes;  // exact f64 child boxes per node
    uint32_t max_depth = 0, n_leaves = 0;
};
constexpr uint32_t JOB_FLAG = 0x40000000u;  // temporary child ref of the top arena: bits 0..29 = subtree job (no node index reaches 2^28)

struct Builder {
    const std::vector<Aabb> &boxes;
    const std::vector<uint32_t> *leaf_group = nullptr;
    std::vector<uint32_t> order;
    std::vector<double> cx, cy, cz;
    BvhBuildResult out;
    std::vector<std::pair<Aabb, Aabb>> child_boxes;  // exact f64 child boxes per node (after the arenas are merged)
    struct Job {
        uint32_t first, count, depth;
        Arena arena;
        uint32_t root_ref = 0;
        Aabb box{};
    };
    std::vector<Job> jobs;
    uint32_t max_threads = 0;
    uint32_t job_size = 0xFFFFFFFFu;  // subtrees of at most this many primitives become jobs (top-level build only)

    explicit Builder(const std::vector<Aabb> &b) : boxes(b) {}

    static Aabb empty_box() {
        const double inf = std::numeric_limits<double>::infinity();
        return Aabb{{inf, inf, inf}, {-inf, -inf, -inf}};
    }
    static void grow(Aabb &a, const Aabb &b) {
        for (int k = 0; k < 3; ++k) {
            a.mn[k] = std::fmin(a.mn[k], b.mn[k]);
            a.mx[k] = std::fmax(a.mx[k], b.mx[k]);
        }
    }
    static double area(const Aabb &a) {
        double dx = a.mx[0] - a.mn[0], dy = a.mx[1] - a.mn[1], dz = a.mx[2] - a.mn[2];
        if (!(dx >= 0.0) || !(dy >= 0.0) || !(dz >= 0.0)) return 0.0;
        return 2.0 * (dx * dy + dy * dz + dx * dz);
    }
    double centroid(uint32_t prim, int axis) const { return axis == 0 ? cx[prim] : (axis == 1 ? cy[prim] : cz[prim]); }

    static uint32_t min_depth_for(uint32_t count) {
        uint32_t d = 0;
        uint64_t cap = MAX_LEAF_PRIMS;
        while (cap < count) {
            cap *= 2;
            ++d;
        }
        return d;
    }

    uint32_t worker_count() const { return max_threads ? max_threads : std::max(1u, std::min(16u, std::thread::hardware_concurrency())); }

    // largest leaf the primitives order[first .. first+count) may form
    uint32_t leaf_target(uint32_t first, uint32_t count) const {
        if (count <= 1 || !leaf_group || count > kMeshLeaf) return 1;
        const uint32_t g = (*leaf_group)[order[first]];
        if (g == NO_LEAF_GROUP) return 1;
        for (uint32_t i = first + 1; i < first + count; ++i)
            if ((*leaf_group)[order[i]] != g) return 1;
        return kMeshLeaf;
    }

    // returns child ref (arena-local), fills box.  TOP: building the top of the tree -- ranges of at most job_size
    // primitives are not descended into but queued as jobs (their box is filled in when the job has run).
    template <bool TOP>
    uint32_t build(uint32_t first, uint32_t count, uint32_t depth, Aabb *box_out, Arena &out) {
        if (TOP && count <= job_size && count > MAX_LEAF_PRIMS) {
            jobs.push_back(Job{first, count, depth, Arena{}, 0u, Aabb{}});
            return JOB_FLAG | static_cast<uint32_t>(jobs.size() - 1);
        }
        const uint32_t kLeafTarget = leaf_target(first, count);
        Aabb box = empty_box(), cbox = empty_box();
        for (uint32_t i = first; i < first + count; ++i) {
            uint32_t p = order[i];
            grow(box, boxes[p]);
            for (int k = 0; k < 3; ++k) {
                double c = centroid(p, k);
                cbox.mn[k] = std::fmin(cbox.mn[k], c);
                cbox.mx[k] = std::fmax(cbox.mx[k], c);
            }
        }
        *box_out = box;
        if (depth > out.max_depth) out.max_depth = depth;

        const uint32_t depth_budget = (STACK_DEPTH - 2) - depth;  // levels still available below this one
        bool force_balanced = min_depth_for(count) + 1 >= depth_budget;
        uint32_t mid = 0;
        bool split_found = false;

        if (count <= kLeafTarget || (count <= MAX_LEAF_PRIMS && depth_budget == 0)) {
            out.n_leaves++;
            return make_leaf_ref(first, count);
        }

        if (!force_balanced) {
            double best_cost = std::numeric_limits<double>::infinity();
            int best_axis = -1, best_bin = -1;
            const double parent_area = area(box);
            for (int axis = 0; axis < 3; ++axis) {
                double lo = cbox.mn[axis], hi = cbox.mx[axis];
                if (!(hi > lo)) continue;
                double scale = kBins / (hi - lo);
                uint32_t bin_count[kBins] = {0};
                Aabb bin_box[kBins];
                for (auto &b : bin_box) b = empty_box();
                auto bin_chunk = [&](uint32_t a, uint32_t e, uint32_t *cnt, Aabb *bx) {
                    for (uint32_t i = a; i < e; ++i) {
                        uint32_t p = order[i];
                        int b = static_cast<int>((centroid(p, axis) - lo) * scale);
                        b = std::min(std::max(b, 0), kBins - 1);
                        cnt[b]++;
                        grow(bx[b], boxes[p]);
                    }
                };
                const uint32_t workers = (TOP && count >= 262144) ? worker_count() : 1u;
                if (workers == 1) {
                    bin_chunk(first, first + count, bin_count, bin_box);
                } else {  // top of a large tree (serial otherwise, and 7 levels deep before the subtree jobs start):
                          // chunks binned by the workers, merged -- counts are integers, boxes min/max: same bins
                    struct Part {
                        uint32_t cnt[kBins];
                        Aabb bx[kBins];
                    };
                    std::vector<Part> part(workers);
                    std::vector<std::thread> pool;
                    const uint32_t step = (count + workers - 1) / workers;
                    for (uint32_t t = 0; t < workers; ++t) {
                        for (int b = 0; b < kBins; ++b) {
                            part[t].cnt[b] = 0;
                            part[t].bx[b] = empty_box();
                        }
                        const uint32_t a = first + std::min(count, t * step), e = first + std::min(count, (t + 1) * step);
                        pool.emplace_back([&bin_chunk, &part, t, a, e]() { bin_chunk(a, e, part[t].cnt, part[t].bx); });
                    }
                    for (auto &th : pool) th.join();
                    for (const Part &pt : part)
                        for (int b = 0; b < kBins; ++b) {
                            bin_count[b] += pt.cnt[b];
                            grow(bin_box[b], pt.bx[b]);
                        }
                }
                double right_area[kBins];
                uint32_t right_count[kBins];
                Aabb acc = empty_box();
                uint32_t n = 0;
                for (int b = kBins - 1; b >= 1; --b) {
                    grow(acc, bin_box[b]);
                    n += bin_count[b];
                    right_area[b] = area(acc);
                    right_count[b] = n;
                }
                acc = empty_box();
                n = 0;
                for (int b = 0; b < kBins - 1; ++b) {
                    grow(acc, bin_box[b]);
                    n += bin_count[b];
                    if (n == 0 || right_count[b + 1] == 0) continue;
                    double cost = area(acc) * n + right_area[b + 1] * right_count[b + 1];
                    if (cost < best_cost) {
                        best_cost = cost;
                        best_axis = axis;
                        best_bin = b;
                    }
                }
            }
            if (best_axis >= 0) {
                double split_cost = 1.0 + kIntersectCost * best_cost / (parent_area > 0.0 ? parent_area : 1.0);
                double leaf_cost = kIntersectCost * count;
                if (count <= kLeafTarget && leaf_cost <= split_cost) {
                    out.n_leaves++;
                    return make_leaf_ref(first, count);
                }
                double lo = cbox.mn[best_axis], hi = cbox.mx[best_axis];
                double scale = kBins / (hi - lo);
                auto it = std::partition(order.begin() + first, order.begin() + first + count, [&](uint32_t p) {
                    int b = static_cast<int>((centroid(p, best_axis) - lo) * scale);
                    b = std::min(std::max(b, 0), kBins - 1);
                    return b <= best_bin;
                });
                mid = static_cast<uint32_t>(it - order.begin());
                split_found = mid > first && mid < first + count;
            } else if (count <= MAX_LEAF_PRIMS) {
                out.n_leaves++;  // identical centroids: nothing to gain from splitting
                return make_leaf_ref(first, count);
            }
        }
        if (!split_found) {  // balanced median split (depth guard / degenerate centroids)
            int axis = 0;
            double ext = -1.0;
            for (int k = 0; k < 3; ++k) {
                double e = cbox.mx[k] - cbox.mn[k];
                if (e > ext) {
                    ext = e;
                    axis = k;
                }
            }
            mid = first + count / 2;
            std::nth_element(order.begin() + first, order.begin() + mid, order.begin() + first + count,
                             [&](uint32_t a, uint32_t b) {
                                 double ca = centroid(a, axis), cb = centroid(b, axis);
                                 return ca < cb || (ca == cb && a < b);
                             });
        }

        uint32_t self = static_cast<uint32_t>(out.nodes.size());
        out.nodes.emplace_back();
        out.child_boxes.emplace_back();
        Aabb lb{}, rb{};
        uint32_t l = build<TOP>(first, mid - first, depth + 1, &lb, out);
        uint32_t r = build<TOP>(mid, first + count - mid, depth + 1, &rb, out);
        BvhNode &n = out.nodes[self];
        n.left = l;
        n.right = r;
        out.child_boxes[self] = {lb, rb};  // quantised once the root box (= the frame) is known
        return self;
    }

    // Whole tree: top serially, subtree jobs on `threads` workers, arenas merged into this->out / child_boxes.
    void build_all(uint32_t n, Aabb *root_box) {
        const uint32_t threads = worker_count();
        // the cut into jobs depends on n alone, never on the thread count (small scenes: one arena)
        job_size = n >= 65536 ? std::max<uint32_t>(4096, n / 128) : 0u;
        Arena top;
        uint32_t root = job_size ? build<true>(0, n, 0, root_box, top) : build<false>(0, n, 0, root_box, top);
        if (!jobs.empty()) {
            std::atomic<uint32_t> next{0};
            auto work = [&]() {
                for (uint32_t j; (j = next.fetch_add(1)) < jobs.size();) {
                    Job &job = jobs[j];
                    job.root_ref = build<false>(job.first, job.count, job.depth, &job.box, job.arena);
                }
            };
            std::vector<std::thread> pool;
            for (uint32_t t = 1; t < threads; ++t) pool.emplace_back(work);
            work();
            for (auto &t : pool) t.join();
        }
        // ---- merge: top arena first, then the jobs' arenas in job order
        std::vector<uint32_t> base(jobs.size());
        uint32_t total = static_cast<uint32_t>(top.nodes.size());
        for (size_t j = 0; j < jobs.size(); ++j) {
            base[j] = total;
            total += static_cast<uint32_t>(jobs[j].arena.nodes.size());
        }
        auto job_ref = [&](uint32_t ref) {  // temporary ref -> final ref of the job's root
            const uint32_t j = ref & ~JOB_FLAG;
            const uint32_t r = jobs[j].root_ref;
            return (r & LEAF_FLAG) ? r : r + base[j];
        };
        this->out.nodes.reserve(total);
        child_boxes.reserve(total);
        for (size_t i = 0; i < top.nodes.size(); ++i) {
            BvhNode nd = top.nodes[i];
            std::pair<Aabb, Aabb> cb = top.child_boxes[i];
            if (!(nd.left & LEAF_FLAG) && (nd.left & JOB_FLAG)) {
                cb.first = jobs[nd.left & ~JOB_FLAG].box;
                nd.left = job_ref(nd.left);
            }
            if (!(nd.right & LEAF_FLAG) && (nd.right & JOB_FLAG)) {
                cb.second = jobs[nd.right & ~JOB_FLAG].box;
                nd.right = job_ref(nd.right);
            }
            this->out.nodes.push_back(nd);
            child_boxes.push_back(cb);
        }
        this->out.max_depth = top.max_depth;
        this->out.n_leaves = top.n_leaves;
        for (size_t j = 0; j < jobs.size(); ++j) {
            const Arena &a = jobs[j].arena;
            for (size_t i = 0; i < a.nodes.size(); ++i) {
                BvhNode nd = a.nodes[i];
                if (!(nd.left & LEAF_FLAG)) nd.left += base[j];
                if (!(nd.right & LEAF_FLAG)) nd.right += base[j];
                this->out.nodes.push_back(nd);
                child_boxes.push_back(a.child_boxes[i]);
            }
            this->out.max_depth = std::max(this->out.max_depth, a.max_depth);
            this->out.n_leaves += a.n_leaves;
        }
        if (!(root & LEAF_FLAG) && (root & JOB_FLAG)) {  // the whole scene was one job
            *root_box = jobs[root & ~JOB_FLAG].box;
            root = job_ref(root);
        }
        this->out.root_ref = root;
    }
};

}  // namespace

BvhBuildResult build_bvh(const std::vector<Aabb> &boxes, const std::vector<uint32_t> *leaf_group, uint32_t max_threads,
                         const float *frame_origin, const float *frame_scale) {
    Builder b(boxes);
    b.leaf_group = leaf_group;
    b.max_threads = max_threads;
    const uint32_t n = static_cast<uint32_t>(boxes.size());
    if (n == 0) return std::move(b.out);
    b.order.resize(n);
    b.cx.resize(n);
    b.cy.resize(n);
    b.cz.resize(n);
    for (uint32_t i = 0; i < n; ++i) {
        b.order[i] = i;
        b.cx[i] = 0.5 * (boxes[i].mn[0] + boxes[i].mx[0]);
        b.cy[i] = 0.5 * (boxes[i].mn[1] + boxes[i].mx[1]);
        b.cz[i] = 0.5 * (boxes[i].mn[2] + boxes[i].mx[2]);
    }
    Aabb root{};
    b.build_all(n, &root);
    b.out.root_box = root;
    b.out.order = std::move(b.order);
    // ---- quantisation frame: the root box, 65535 steps per axis
    double origin[3], scale[3];
    for (int k = 0; k < 3; ++k) {
        b.out.q_origin[k] = round_down(root.mn[k]);
        origin[k] = static_cast<double>(b.out.q_origin[k]);
        double extent = root.mx[k] - origin[k];
        if (!(extent > 0.0) || !std::isfinite(extent)) extent = 1.0;
        b.out.q_scale[k] = round_up(extent / 65535.0 * (1.0 + 1e-6));
        scale[k] = static_cast<double>(b.out.q_scale[k]);
        if (frame_origin && frame_scale) {  // the caller's frame: the boxes must lie inside it (a subset of the tree that defined it)
            b.out.q_origin[k] = frame_origin[k];
            b.out.q_scale[k] = frame_scale[k];
            origin[k] = static_cast<double>(frame_origin[k]);
            scale[k] = static_cast<double>(frame_scale[k]);
        }
    }
    // origin + q * scale <= x for minima (>= for maxima), in exact arithmetic on the stored f32 frame
    auto q_down = [&](double x, int k) -> uint16_t {
        double q = std::floor((x - origin[k]) / scale[k]);
        if (!(q > 0.0)) return 0;  // also NaN
        if (q > 65535.0) q = 65535.0;
        while (q > 0.0 && origin[k] + q * scale[k] > x) q -= 1.0;
        return static_cast<uint16_t>(q);
    };
    auto q_up = [&](double x, int k) -> uint16_t {
        double q = std::ceil((x - origin[k]) / scale[k]);
        if (!(q < 65535.0)) return 65535;  // also NaN
        if (q < 0.0) q = 0.0;
        while (q < 65535.0 && origin[k] + q * scale[k] < x) q += 1.0;
        return static_cast<uint16_t>(q);
    };
    for (size_t i = 0; i < b.out.nodes.size(); ++i) {
        BvhNode &n = b.out.nodes[i];
        const Aabb &lb = b.child_boxes[i].first, &rb = b.child_boxes[i].second;
        for (int k = 0; k < 3; ++k) {
            n.lmin[k] = q_down(lb.mn[k], k);
            n.lmax[k] = q_up(lb.mx[k], k);
            n.rmin[k] = q_down(rb.mn[k], k);
            n.rmax[k] = q_up(rb.mx[k], k);
        }
    }
    // ---- breadth-first order: the top of the tree becomes the first nodes of the array, which the
    // traversal kernels keep in LDS (every ray starts there; most visits never leave it)
    if (!(b.out.root_ref & LEAF_FLAG) && !b.out.nodes.empty()) {
        const size_t nn = b.out.nodes.size();
        std::vector<uint32_t> bfs;  // new index -> old index
        bfs.reserve(nn);
        bfs.push_back(b.out.root_ref);
        for (size_t head = 0; head < bfs.size(); ++head) {
            const BvhNode &n = b.out.nodes[bfs[head]];
            if (!(n.left & LEAF_FLAG)) bfs.push_back(n.left);
            if (!(n.right & LEAF_FLAG)) bfs.push_back(n.right);
        }
        std::vector<uint32_t> renum(nn, 0);
        for (size_t i = 0; i < bfs.size(); ++i) renum[bfs[i]] = static_cast<uint32_t>(i);
        std::vector<BvhNode> sorted(bfs.size());
        for (size_t i = 0; i < bfs.size(); ++i) {
            BvhNode n = b.out.nodes[bfs[i]];
            if (!(n.left & LEAF_FLAG)) n.left = renum[n.left];
            if (!(n.right & LEAF_FLAG)) n.right = renum[n.right];
            sorted[i] = n;
        }
        b.out.nodes.swap(sorted);
        b.out.root_ref = 0;
    }
    return std::move(b.out);
}

}  // namespace rt
