// Host-side BVH build + linearisation (replaces KdTree::build,
// src/core/bounds.rs:452-610, as the acceleration structure of the GPU path).
#pragma once
#include <cstdint>
#include <vector>

#include "rt_internal.h"

namespace rt {

struct Aabb {
    double mn[3], mx[3];
};

struct BvhBuildResult {
    std::vector<BvhNode> nodes;
    std::vector<uint32_t> order;  // leaf-record order -> index into the input boxes
    uint32_t root_ref = 0xFFFFFFFFu;
    uint32_t max_depth = 0;
    uint32_t n_leaves = 0;
    Aabb root_box{};
    float q_origin[3] = {0, 0, 0}, q_scale[3] = {1, 1, 1};  // quantisation frame of the node boxes
};

// boxes: world AABBs of the bounded primitives.  Depth never exceeds
// STACK_DEPTH - 1 so the per-lane LDS stack of the traversal kernels suffices.
// leaf_group (optional, one entry per box): primitives may share a leaf only when they
// carry the same group value other than NO_LEAF_GROUP (triangles of one mesh: same
// object-space ray, same code path); everything else gets a leaf of its own.
constexpr uint32_t NO_LEAF_GROUP = 0xFFFFFFFFu;
// max_threads: workers for the subtree jobs of a large scene (0 = the host's cores, at most 16); the tree does not
// depend on it.
// frame_origin / frame_scale (optional, 3 floats each): quantise the node boxes in this frame instead of the tree's own
// root box (a tree over a subset of another tree's primitives, traversed with the same CullRay).
BvhBuildResult build_bvh(const std::vector<Aabb> &boxes, const std::vector<uint32_t> *leaf_group = nullptr, uint32_t max_threads = 0,
                         const float *frame_origin = nullptr, const float *frame_scale = nullptr);

inline uint32_t make_leaf_ref(uint32_t first, uint32_t count) {
    return LEAF_FLAG | ((count - 1u) << 28) | first;
}

}  // namespace rt
