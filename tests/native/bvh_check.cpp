// CPU harness for tests/test_bvh_build.py: builds the BVH of N pseudo-random boxes with 1 and with 8 workers and
// checks (a) both trees are byte-identical, (b) every primitive sits in exactly one leaf, (c) every node's
// (dequantised) child box contains the boxes of all primitives below it, (d) depth fits the traversal stack, (e) the
// same containment for a tree over a subset built in the full tree's quantisation frame.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#include "../../raytracer_amd/csrc/rt_bvh.h"

using namespace rt;

int main(int argc, char **argv) {
    const uint32_t n = argc > 1 ? std::atoi(argv[1]) : 200000;
    std::vector<Aabb> boxes(n);
    std::vector<uint32_t> group(n);
    uint64_t s = 0x9E3779B97F4A7C15ull;
    auto rnd = [&]() {
        s ^= s << 13;
        s ^= s >> 7;
        s ^= s << 17;
        return static_cast<double>(s >> 11) * 0x1.0p-53;
    };
    for (uint32_t i = 0; i < n; ++i) {
        for (int k = 0; k < 3; ++k) {
            const double c = rnd() * 100.0 - 50.0, h = rnd() * (i % 97 == 0 ? 5.0 : 0.2);
            boxes[i].mn[k] = c - h;
            boxes[i].mx[k] = c + h;
        }
        group[i] = i % 5 == 0 ? NO_LEAF_GROUP : i % 3;  // analytic primitives and three "meshes"
    }
    BvhBuildResult a = build_bvh(boxes, &group, 1), b = build_bvh(boxes, &group, 8);
    if (a.nodes.size() != b.nodes.size() || a.order != b.order || a.root_ref != b.root_ref || a.max_depth != b.max_depth ||
        a.n_leaves != b.n_leaves || std::memcmp(a.nodes.data(), b.nodes.data(), a.nodes.size() * sizeof(BvhNode)) != 0) {
        std::printf("FAIL trees differ between 1 and 8 workers\n");
        return 1;
    }
    std::vector<uint32_t> seen(n, 0);
    uint64_t leaves = 0;
    bool ok = true;
    auto deq = [&](uint16_t q, int k) { return static_cast<double>(a.q_origin[k]) + static_cast<double>(q) * static_cast<double>(a.q_scale[k]); };
    // returns the exact union box of the subtree
    std::function<Aabb(uint32_t, uint32_t)> walk = [&](uint32_t ref, uint32_t depth) -> Aabb {
        Aabb u{{1e300, 1e300, 1e300}, {-1e300, -1e300, -1e300}};
        if (depth > STACK_DEPTH - 2) ok = false;
        if (ref & LEAF_FLAG) {
            const uint32_t first = ref & 0x0FFFFFFFu, cnt = ((ref >> 28) & 7u) + 1u;
            leaves++;
            uint32_t g0 = group[a.order[first]];
            for (uint32_t i = first; i < first + cnt; ++i) {
                const uint32_t p = a.order[i];
                seen[p]++;
                if (cnt > 1 && (group[p] == NO_LEAF_GROUP || group[p] != g0)) ok = false;  // leaf sharing rule
                for (int k = 0; k < 3; ++k) {
                    u.mn[k] = std::min(u.mn[k], boxes[p].mn[k]);
                    u.mx[k] = std::max(u.mx[k], boxes[p].mx[k]);
                }
            }
            return u;
        }
        const BvhNode &nd = a.nodes[ref];
        const Aabb l = walk(nd.left, depth + 1), r = walk(nd.right, depth + 1);
        for (int k = 0; k < 3; ++k) {
            if (deq(nd.lmin[k], k) > l.mn[k] || deq(nd.lmax[k], k) < l.mx[k]) ok = false;
            if (deq(nd.rmin[k], k) > r.mn[k] || deq(nd.rmax[k], k) < r.mx[k]) ok = false;
            u.mn[k] = std::min(l.mn[k], r.mn[k]);
            u.mx[k] = std::max(l.mx[k], r.mx[k]);
        }
        return u;
    };
    walk(a.root_ref, 0);
    for (uint32_t i = 0; i < n; ++i)
        if (seen[i] != 1) ok = false;
    if (leaves != a.n_leaves) ok = false;
    // breadth-first order: children never precede their parent
    for (size_t i = 0; i < a.nodes.size(); ++i) {
        if (!(a.nodes[i].left & LEAF_FLAG) && a.nodes[i].left <= i) ok = false;
        if (!(a.nodes[i].right & LEAF_FLAG) && a.nodes[i].right <= i) ok = false;
    }
    // (e) a tree over a subset of the boxes, quantised in the full tree's frame (the trees of the special primitives,
    // rt_api.hip: traversed with the main tree's CullRay): its frame is the full tree's, its boxes contain what is below
    {
        std::vector<Aabb> sub;
        for (uint32_t i = 0; i < n; i += 7) sub.push_back(boxes[i]);
        const BvhBuildResult c = build_bvh(sub, nullptr, 1, a.q_origin, a.q_scale);
        for (int k = 0; k < 3; ++k)
            if (c.q_origin[k] != a.q_origin[k] || c.q_scale[k] != a.q_scale[k]) ok = false;
        std::vector<uint32_t> seen_c(sub.size(), 0);
        std::function<Aabb(uint32_t)> walk_c = [&](uint32_t ref) -> Aabb {
            Aabb u{{1e300, 1e300, 1e300}, {-1e300, -1e300, -1e300}};
            if (ref & LEAF_FLAG) {
                const uint32_t first = ref & 0x0FFFFFFFu, cnt = ((ref >> 28) & 7u) + 1u;
                for (uint32_t i = first; i < first + cnt; ++i) {
                    const uint32_t p = c.order[i];
                    seen_c[p]++;
                    for (int k = 0; k < 3; ++k) {
                        u.mn[k] = std::min(u.mn[k], sub[p].mn[k]);
                        u.mx[k] = std::max(u.mx[k], sub[p].mx[k]);
                    }
                }
                return u;
            }
            const BvhNode &nd = c.nodes[ref];
            const Aabb l = walk_c(nd.left), r = walk_c(nd.right);
            for (int k = 0; k < 3; ++k) {
                if (deq(nd.lmin[k], k) > l.mn[k] || deq(nd.lmax[k], k) < l.mx[k]) ok = false;
                if (deq(nd.rmin[k], k) > r.mn[k] || deq(nd.rmax[k], k) < r.mx[k]) ok = false;
                u.mn[k] = std::min(l.mn[k], r.mn[k]);
                u.mx[k] = std::max(l.mx[k], r.mx[k]);
            }
            return u;
        };
        if (!sub.empty() && c.root_ref != 0xFFFFFFFFu) walk_c(c.root_ref);
        for (uint32_t x : seen_c)
            if (x != 1) ok = false;
    }
    std::printf("%s nodes %zu leaves %u depth %u\n", ok ? "OK" : "FAIL", a.nodes.size(), a.n_leaves, a.max_depth);
    return ok ? 0 : 1;
}
