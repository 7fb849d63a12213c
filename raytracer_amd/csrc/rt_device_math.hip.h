// rt_device_math.hip.h -- device-side f64 math of the hot path (gfx950).
//
// Everything that decides a hit (the primitives' own AABB slab test and the
// four intersect routines) follows the reference's operation order exactly;
// the translation unit is built with -ffp-contract=off so no multiply-add is
// fused and every operation rounds like the reference's f64 code:
//   src/core/bounds.rs:114-152          BoundingVolume::intersect
//   src/primitives/sphere.rs:101-130    + src/utils/mod.rs:35-47 (quadratic)
//   src/primitives/cube.rs:100-174
//   src/primitives/plane.rs:98-117
//   src/primitives/triangle.rs:193-232
//   src/ray_intersection.rs:33-43,86-117  Ray::transform, compute_data
// BVH node culling is NOT part of the hit predicate; it uses a cheaper,
// provably conservative slab test (box_cull_*).
#pragma once
#include <hip/hip_runtime.h>

#include "rt_internal.h"

namespace rt {

struct d3 {
    double x, y, z;
};

__device__ __forceinline__ d3 mk3(double x, double y, double z) { return d3{x, y, z}; }
__device__ __forceinline__ d3 operator+(d3 a, d3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ d3 operator-(d3 a, d3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ d3 operator-(d3 a) { return {-a.x, -a.y, -a.z}; }
__device__ __forceinline__ d3 operator*(d3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ d3 operator*(double s, d3 a) { return {s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ d3 operator/(d3 a, double s) { return {a.x / s, a.y / s, a.z / s}; }
__device__ __forceinline__ d3 hadamard(d3 a, d3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
__device__ __forceinline__ double dot3(d3 a, d3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ d3 cross3(d3 a, d3 b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__device__ __forceinline__ double length3(d3 a) { return sqrt(dot3(a, a)); }
__device__ __forceinline__ d3 unit3(d3 a) { return a / length3(a); }
__device__ __forceinline__ d3 mix3(d3 a, d3 b, double t) { return a * (1.0 - t) + b * t; }  // nalgebra lerp
__device__ __forceinline__ double comp(d3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }

// Rust f64::signum: +-1 by sign bit, NaN stays NaN
__device__ __forceinline__ double rsign(double x) { return isnan(x) ? x : copysign(1.0, x); }
// Rust f64::clamp(0,1): NaN stays NaN
__device__ __forceinline__ double sat01(double c) { return c < 0.0 ? 0.0 : (c > 1.0 ? 1.0 : c); }
__device__ __forceinline__ d3 sat01(d3 a) { return {sat01(a.x), sat01(a.y), sat01(a.z)}; }
__device__ __forceinline__ uint32_t cast_u32(double x) {  // Rust `as u32` (saturating, NaN -> 0)
    if (!(x > 0.0)) return 0u;
    if (x >= 4294967295.0) return 4294967295u;
    return static_cast<uint32_t>(x);
}
__device__ __forceinline__ double lerp1(double x0, double x1, double t) { return x0 - x0 * t + x1 * t; }

// Affine 3x4 row-major block in global memory
__device__ __forceinline__ d3 xf_point(const double *__restrict__ m, d3 p) {
    return {m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3], m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7],
            m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11]};
}
__device__ __forceinline__ d3 xf_vector(const double *__restrict__ m, d3 v) {
    return {m[0] * v.x + m[1] * v.y + m[2] * v.z, m[4] * v.x + m[5] * v.y + m[6] * v.z,
            m[8] * v.x + m[9] * v.y + m[10] * v.z};
}
__device__ __forceinline__ d3 xf_vector_t(const double *__restrict__ m, d3 v) {
    return {m[0] * v.x + m[4] * v.y + m[8] * v.z, m[1] * v.x + m[5] * v.y + m[9] * v.z,
            m[2] * v.x + m[6] * v.y + m[10] * v.z};
}

// ------------------------------------------------------------------ RNG
// Philox4x32-10; stream layout in DESIGN.md "RNG".
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                               uint32_t k1, uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        uint32_t n0 = hi1 ^ c1 ^ k0;
        uint32_t n2 = hi0 ^ c3 ^ k1;
        c0 = n0;
        c1 = lo1;
        c2 = n2;
        c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0;
    out[1] = c1;
    out[2] = c2;
    out[3] = c3;
}

__device__ __forceinline__ uint64_t path_child(uint64_t path, uint32_t slot) {
    uint64_t z = path * 0xD1342543DE82EF95ull + (static_cast<uint64_t>(slot) + 1ull);
    z ^= z >> 32;
    z *= 0xD6E8FEB86659FD93ull;
    z ^= z >> 32;
    z *= 0xD6E8FEB86659FD93ull;
    z ^= z >> 32;
    return z;
}

__device__ __forceinline__ void rng_pair(uint64_t seed, uint32_t pixel, uint32_t sample, uint64_t path, double *u1,
                                         double *u2) {
    uint32_t r[4];
    philox4x32_10(pixel, sample, static_cast<uint32_t>(path), static_cast<uint32_t>(path >> 32),
                  static_cast<uint32_t>(seed), static_cast<uint32_t>(seed >> 32), r);
    uint64_t a = static_cast<uint64_t>(r[0]) | (static_cast<uint64_t>(r[1]) << 32);
    uint64_t b = static_cast<uint64_t>(r[2]) | (static_cast<uint64_t>(r[3]) << 32);
    *u1 = static_cast<double>(a >> 11) * 0x1.0p-53;
    *u2 = static_cast<double>(b >> 11) * 0x1.0p-53;
}

// ------------------------------------------------------------------ sampling (src/utils/sampling.rs)
// cosine_sample_hemisphere (sampling.rs:6-37, with concentric_sample_disk) and
// uniform_sample_cone (:39-70) as ONE routine: a wave of child rays mixes cone
// samples (reflections of physical materials) and hemisphere samples (GI rays),
// and as two routines it ran both, each with part of its lanes.  Here the pieces
// both need -- sincos of the first angle, the final normalisation -- are reached by
// all lanes together; every expression is the one the reference evaluates.
constexpr uint32_t SAMPLE_NONE = 0u, SAMPLE_CONE = 1u, SAMPLE_HEMISPHERE = 2u;
__device__ __forceinline__ d3 sample_direction(uint32_t kind, d3 w, double max_angle, double u1, double u2) {
    constexpr double HALF_PI = 1.57079632679489661923, QUARTER_PI = 0.78539816339744830962, TWO_PI = 6.28318530717958647692;
    if (kind == SAMPLE_CONE && max_angle < F64_EPS) kind = SAMPLE_NONE;  // :43-45
    double theta = 0.0, r = 0.0;
    bool centre = false;  // hemisphere: the disk sample is the origin (:10-12)
    if (kind == SAMPLE_CONE) {
        theta = acos(u1);
        theta = theta * max_angle / HALF_PI;
    } else if (kind == SAMPLE_HEMISPHERE) {
        const double rx = 2.0 * u1 - 1.0, ry = 2.0 * u2 - 1.0;
        if (rx == 0.0 && ry == 0.0) {
            centre = true;
        } else if (fabs(rx) > fabs(ry)) {
            r = rx;
            theta = HALF_PI * (ry / rx);
        } else {
            r = ry;
            theta = HALF_PI - QUARTER_PI * (rx / ry);
        }
    }
    if (kind == SAMPLE_NONE) return w;
    double st, ct;
    sincos(theta, &st, &ct);  // one argument reduction for both
    d3 raw;
    if (kind == SAMPLE_CONE) {
        const double z = ct, radius = st;
        const double phi = u2 * TWO_PI;
        double sp, cp;
        sincos(phi, &sp, &cp);
        d3 u = cross3(w, mk3(0.0, 0.0, 1.0));
        const double mag = length3(u);
        if (mag < F64_EPS) {
            raw = mk3(radius * cp, radius * sp, rsign(w.z) * z);
        } else {
            u = unit3(u);
            const d3 v = unit3(cross3(w, u));
            raw = u * radius * cp + v * radius * sp + w * z;
        }
    } else {
        const double px = centre ? 0.0 : r * ct, py = centre ? 0.0 : r * st;
        const double pz = sqrt(1.0 - px * px - py * py);
        const d3 u = fabs(w.x) > F64_EPS ? cross3(w, mk3(0.0, 1.0, 0.0)) : cross3(w, mk3(1.0, 0.0, 0.0));
        const d3 v = cross3(w, u);
        raw = u * px + v * py + w * pz;
    }
    return unit3(raw);
}

// src/utils/rays.rs
__device__ __forceinline__ d3 reflect_dir(d3 incident, d3 normal) {
    return unit3(incident - (2.0 * dot3(incident, normal)) * normal);
}
__device__ __forceinline__ bool refract_dir(d3 incident, d3 normal, double eta, d3 *out) {
    double n_dot_i = dot3(normal, incident);
    d3 rn = n_dot_i < 0.0 ? normal : -normal;
    eta = n_dot_i < 0.0 ? eta : 1.0 / eta;
    n_dot_i = fabs(n_dot_i);
    double k = 1.0 - eta * eta * (1.0 - n_dot_i * n_dot_i);
    if (k < 0.0) return false;
    *out = unit3(incident * eta - rn * (eta * n_dot_i - sqrt(k)));
    return true;
}

// ------------------------------------------------------------------ hit predicates
struct RayD {
    d3 o, d;
};

// BoundingVolume::intersect (bounds.rs:114-152), reference operation order.
// `center` = (min + max) * 0.5 and `half` = (max - min) / 2.0 come precomputed (host, same f64 ops).
__device__ __forceinline__ bool prim_box_test(const double *__restrict__ center, const double *__restrict__ half,
                                              const RayD &r, bool has_max, double max_distance) {
    double tcx = center[0] - r.o.x, tcy = center[1] - r.o.y, tcz = center[2] - r.o.z;
    double hx = half[0] * rsign(r.d.x);
    double hy = half[1] * rsign(r.d.y);
    double hz = half[2] * rsign(r.d.z);

    double d_near = (tcx - hx) / r.d.x;
    double d_far = (tcx + hx) / r.d.x;
    double dy_near = (tcy - hy) / r.d.y;
    double dy_far = (tcy + hy) / r.d.y;
    if (dy_far < d_near || d_far < dy_near) return false;
    d_near = dy_near > d_near ? dy_near : d_near;
    d_far = d_far > dy_far ? dy_far : d_far;

    double dz_near = (tcz - hz) / r.d.z;
    double dz_far = (tcz + hz) / r.d.z;
    if (dz_far < d_near || d_far < dz_near) return false;
    d_near = dz_near > d_near ? dz_near : d_near;
    d_far = d_far > dz_far ? dz_far : d_far;

    if (d_near < 0.0 && d_far < 0.0) return false;
    if (has_max && max_distance < d_near) return false;
    return true;
}

// Sufficient condition for prim_box_test to pass, without its six divisions: some
// point o + t d of the ray with t >= 0 (and t <= max_distance for shadow rays) lies
// inside the box with a margin of 1e-9 relative on every axis.  Then on every axis
// (near plane - o)/d < t < (far plane - o)/d holds by a margin that exceeds the
// rounding of the reference-order slab arithmetic (a few ulps) a million times
// (d = +-0: the quotients are -inf / +inf), so: no pair of slabs is disjoint
// (every near < t < every far), d_far > t >= 0 rules out "both negative", and
// d_near < t <= max_distance rules out the distance rejection.
// The caller picks t (`t_probe` of hit_bounded): the hit itself for a triangle, the
// middle of the ray's chord through a sphere or cube clipped to t >= 0 -- a hit ON
// a face of an axis-aligned cube lies on the face of its box, where no margin
// exists, but the chord's midpoint is interior.  The test checks the point it is
// given, so a grazing chord or a flat box (half = 0 on an axis) simply fails it and
// takes the exact test.
__device__ __forceinline__ bool prim_box_surely_passes(const double *__restrict__ center, const double *__restrict__ half,
                                                       const RayD &r, double t, bool has_max, double max_distance) {
    if (!(t >= 0.0) || (has_max && !(t <= max_distance))) return false;
    const double px = r.o.x + r.d.x * t, py = r.o.y + r.d.y * t, pz = r.o.z + r.d.z * t;
    const double ex = 1e-9 * (fabs(r.o.x) + fabs(center[0]) + half[0]);
    const double ey = 1e-9 * (fabs(r.o.y) + fabs(center[1]) + half[1]);
    const double ez = 1e-9 * (fabs(r.o.z) + fabs(center[2]) + half[2]);
    return fabs(px - center[0]) <= half[0] - ex && fabs(py - center[1]) <= half[1] - ey && fabs(pz - center[2]) <= half[2] - ez;
}

struct HitRec {
    double t;
    double bu, bv;  // barycentrics (triangle) | bu = face code (cube: axis + 4*positive)
};

// side rule shared by sphere / cube: Both or shadow ray -> two-sided
__device__ __forceinline__ bool is_two_sided(uint32_t side, bool shadow) { return side == RT_SIDE_BOTH || shadow; }

__device__ __forceinline__ bool hit_plane(d3 normal, uint32_t side, bool shadow, const RayD &r, bool has_max,
                                          double max_distance, HitRec *h) {
    double n_dot_v = dot3(normal, -r.d);
    bool reject;
    if (is_two_sided(side, shadow)) reject = fabs(n_dot_v) < F64_EPS;
    else if (side == RT_SIDE_FRONT) reject = n_dot_v < F64_EPS;
    else reject = -n_dot_v < F64_EPS;
    if (reject) return false;
    double dist = dot3(r.o, normal) / n_dot_v;
    if (dist < 0.0 || (has_max && max_distance < dist)) return false;
    h->t = dist;
    h->bu = 0.0;
    h->bv = 0.0;
    return true;
}

// The three bounded-primitive tests as ONE routine whose f64 divisions sit in a
// block every lane of the wave reaches together.  A wave's lanes hold leaves of
// different types; evaluated as three separate routines the wave pays the
// triangle's 3 divisions, the sphere's 2 and the cube's 6 one after the other
// (11 division sequences, the dominant cost of a leaf round).  Here each type
// first forms its (numerator, denominator) pairs, the wave divides once (3
// quotients; the cube's other 3 in a second block only cubes enter), then each
// type takes its decisions.  Every quotient is the IEEE division of the operands
// the reference divides (triangle.rs:193-232 Moeller-Trumbore with the edges
// precomputed on the host by the same f64 subtraction; sphere.rs:101-130 with
// utils/mod.rs:35-47 `quadratic`; cube.rs:100-174) and every decision the same
// comparison; only the early exits between a type's divisions are gone (they are
// pure predicates, so evaluating the later ones too changes nothing).
// *t_probe: a ray parameter inside the primitive, for prim_box_surely_passes.
template <bool SHADOW>
__device__ __forceinline__ bool hit_bounded(const double *__restrict__ g, uint32_t type, uint32_t side, const RayD &r,
                                            double max_distance, HitRec *h, double *t_probe) {
    double n1 = 0.0, n2 = 0.0, n3 = 0.0, q_d1 = 1.0, q_d2 = 1.0, q_d3 = 1.0;
    double sx = 0.0, sy = 0.0, sz = 0.0, half = 0.0;  // cube
    bool one_root = false;                            // sphere, disc == 0
    if (type == RT_PRIM_TRIANGLE) {
        const d3 p0 = mk3(g[0], g[1], g[2]), e1 = mk3(g[3], g[4], g[5]), e2 = mk3(g[6], g[7], g[8]);
        const d3 pv = cross3(r.d, e2);
        const double det = dot3(e1, pv);
        bool reject;
        if (is_two_sided(side, SHADOW)) reject = fabs(det) < F64_EPS;
        else if (side == RT_SIDE_FRONT) reject = det < F64_EPS;
        else reject = -det < F64_EPS;
        if (reject) return false;
        const d3 tv = r.o - p0;
        const d3 qv = cross3(tv, e1);
        n1 = dot3(tv, pv);
        n2 = dot3(r.d, qv);
        n3 = dot3(e2, qv);
        q_d1 = q_d2 = q_d3 = det;
        // Certain misses leave before the divisions (a mesh triangle's box is hit 20-30 times more often than the
        // triangle; when every lane of the wave leaves here the division block is skipped).  With a = n * sign(det)
        // and m = |det|, IEEE division gives: a < 0 beyond underflow => n / det < 0; a1 > m => u > 1; and
        // a1 + a2 > m (1 + 1e-9) with both non-negative => u + v > 1 whatever the three roundings do.  Anything
        // near a boundary (and every NaN: all comparisons false) goes on to the reference's own arithmetic.
        const double m = fabs(det), sgn = det < 0.0 ? -1.0 : 1.0;
        const double a1 = n1 * sgn, a2 = n2 * sgn, a3 = n3 * sgn, floor_ = -1e-300 * m;
        if (a1 < floor_ || a2 < floor_ || a3 < floor_ || a1 > m || a1 + a2 > m * (1.0 + 1e-9)) return false;
    } else if (type == RT_PRIM_SPHERE) {
        const double radius = g[0];
        const double ray_proj = dot3(r.o, r.d);
        const double a = dot3(r.d, r.d);
        const double b = 2.0 * ray_proj;
        const double c = dot3(r.o, r.o) - radius * radius;
        const double disc = b * b - 4.0 * a * c;
        if (disc < 0.0) return false;
        if (disc == 0.0) {
            one_root = true;
            n1 = -0.5 * b;
            q_d1 = a;
        } else {
            const double q = -0.5 * (b + rsign(b) * sqrt(disc));
            n1 = q;
            q_d1 = a;
            n2 = c;
            q_d2 = q;
        }
    } else {
        sx = rsign(r.d.x);
        sy = rsign(r.d.y);
        sz = rsign(r.d.z);
        half = g[0] / 2.0;
        n1 = -r.o.x - sx * half;
        n2 = -r.o.x + sx * half;
        n3 = -r.o.y - sy * half;
        q_d1 = q_d2 = r.d.x;
        q_d3 = r.d.y;
    }
    const double q1 = n1 / q_d1, q2 = n2 / q_d2, q3 = n3 / q_d3;
    if (type == RT_PRIM_TRIANGLE) {
        const double u = q1, v = q2, dist = q3;
        if (!(u >= 0.0 && u <= 1.0)) return false;
        if (v < 0.0 || 1.0 < u + v) return false;
        if (dist < 0.0 || (SHADOW && max_distance < dist)) return false;
        h->t = dist;
        h->bu = u;
        h->bv = v;
        *t_probe = dist;
        return true;
    }
    if (type == RT_PRIM_SPHERE) {
        const double t0 = one_root ? q1 : fmin(q1, q2), t1 = one_root ? q1 : fmax(q1, q2);
        double dist;
        if (is_two_sided(side, SHADOW)) dist = t0 < 0.0 ? t1 : t0;
        else if (side == RT_SIDE_FRONT) dist = t0;
        else dist = t1;
        if (dist < 0.0 || (SHADOW && max_distance < dist)) return false;
        h->t = dist;
        h->bu = 0.0;
        h->bv = 0.0;
        *t_probe = 0.5 * (fmax(t0, 0.0) + t1);
        return true;
    }
    int near_axis = 0, far_axis = 0;
    bool near_pos = sx < 0.0, far_pos = sx > 0.0;
    double d_near = q1, d_far = q2;
    const double dy_near = q3;
    const double dy_far = (-r.o.y + sy * half) / r.d.y;
    const double dz_near = (-r.o.z - sz * half) / r.d.z;
    const double dz_far = (-r.o.z + sz * half) / r.d.z;
    if (dy_far < d_near || d_far < dy_near) return false;
    if (dy_near > d_near) {
        near_axis = 1;
        near_pos = sy < 0.0;
        d_near = dy_near;
    }
    if (d_far > dy_far) {
        far_axis = 1;
        far_pos = sy > 0.0;
        d_far = dy_far;
    }
    if (dz_far < d_near || d_far < dz_near) return false;
    if (dz_near > d_near) {
        near_axis = 2;
        near_pos = sz < 0.0;
        d_near = dz_near;
    }
    if (d_far > dz_far) {
        far_axis = 2;
        far_pos = sz > 0.0;
        d_far = dz_far;
    }
    const bool use_near = is_two_sided(side, SHADOW) ? !(d_near < 0.0) : (side == RT_SIDE_FRONT);
    const double dist = use_near ? d_near : d_far;
    const int axis = use_near ? near_axis : far_axis;
    const bool positive = use_near ? near_pos : far_pos;
    if (dist < 0.0 || (SHADOW && max_distance < dist)) return false;
    h->t = dist;
    h->bu = static_cast<double>(axis + (positive ? 4 : 0));
    h->bv = 0.0;
    *t_probe = 0.5 * (fmax(d_near, 0.0) + d_far);
    return true;
}

// ------------------------------------------------------------------ BVH node culling
// Conservative slab test in f32: never rejects a box that contains a primitive
// whose exact f64 predicate passes within the t limit.
//   * node boxes are the exact f64 unions rounded outward to 16-bit fixed point in
//     the scene frame (rt_bvh.cpp): looser by at most extent/65535 per side;
//   * every slab is widened by P = 2^-20 (|o|_inf + scene scale) on each side,
//     which absorbs the f32 rounding of the origin-side constants (<= 2^-23 |o|)
//     and the rounding of the primitives' reference-order AABB arithmetic in f64;
//   * inv = 1/d is taken in f32 (v_rcp_f32 of the direction rounded to f32: relative
//     error <= 1.5 * 2^-23, the same factor on both constants of an axis, so it scales
//     that axis' distances) and the fma adds 2^-24: a quarter of P (make_cull_ray has
//     the whole budget);
//   * |1/d| is clamped to 1e25, so a ray parallel to a slab (d = +-0) gets
//     entry/exit distances of magnitude >= 1e25 * P with the right signs, no
//     inf - inf; comparisons are written so that a NaN (NaN ray) keeps the node.
struct CullRay {
    float ax, ay, az;              // q_scale / d per axis, |1/d| clamped to 1e25 (keeps every product finite)
    float x1, x2, y1, y2, z1, z2;  // (q_origin - (o + P)) / d and (q_origin - (o - P)) / d per axis
};

__device__ __forceinline__ float cull_inv(double d) {
    const float inv = __builtin_amdgcn_rcpf(static_cast<float>(d));  // +-inf for +-0 (and for what rounds or flushes to it)
    const float lim = 1e25f;     // |t| beyond 1e25 * P is no hit distance of a bounded scene
    return fabsf(inv) < lim ? inv : copysignf(lim, inv);  // NaN direction -> +-1e25, still finite
}

// Slab distances are evaluated as fma(q, a, c) with q the 16-bit box coordinate:
// (q_origin + q q_scale - (o +- P)) / d.  The differences q_origin - (o +- P) are
// taken in f64 (o is near the box: cancellation) and rounded to f32, the products
// with 1/d are f32.  Error budget, in units of a displacement of the slab plane
// along its axis (S = scene scale + |o|, so P = 2^-20 S):
//   a = q_scale * inv, rounded:                 <= 2^-24 of the box's extent      (P / 16)
//   c = f32(difference) * inv, two roundings:   <= 2^-23 S                        (P / 8)
//   inv = rcp(f32(d)) (1 ulp) and the fma's own rounding: a common factor
//   1 + e, |e| <= 2^-22, on the axis' distances -- the plane moved by
//   2^-22 |plane - o| <= 2^-22 S                                                  (P / 4)
// -- under P / 2 in all.  A ray that meets the true box at t* therefore has, on
// every axis, a computed entry <= t* <= computed exit: box_cull_test needs no
// widening of its own beyond P (through round 3 it lowered the entry and raised
// the exit by another 2^-21 relative: four instructions a node visit for nothing).
__device__ __forceinline__ CullRay make_cull_ray(const RayD &r, double scene_scale, const float *q_origin, const float *q_scale) {
    const double m = fmax(fmax(fabs(r.o.x), fabs(r.o.y)), fabs(r.o.z));
    const double P = 0x1.0p-20 * (m + scene_scale);
    const float ix = cull_inv(r.d.x), iy = cull_inv(r.d.y), iz = cull_inv(r.d.z);
    CullRay c;
    c.ax = q_scale[0] * ix;
    c.ay = q_scale[1] * iy;
    c.az = q_scale[2] * iz;
    c.x1 = static_cast<float>(static_cast<double>(q_origin[0]) - (r.o.x + P)) * ix;
    c.x2 = static_cast<float>(static_cast<double>(q_origin[0]) - (r.o.x - P)) * ix;
    c.y1 = static_cast<float>(static_cast<double>(q_origin[1]) - (r.o.y + P)) * iy;
    c.y2 = static_cast<float>(static_cast<double>(q_origin[1]) - (r.o.y - P)) * iy;
    c.z1 = static_cast<float>(static_cast<double>(q_origin[2]) - (r.o.z + P)) * iz;
    c.z2 = static_cast<float>(static_cast<double>(q_origin[2]) - (r.o.z - P)) * iz;
    return c;
}

// t limit of a traversal in f32, rounded up with slack
__device__ __forceinline__ float cull_limit_f32(double t_limit) {
    const float f = static_cast<float>(t_limit);
    return f + fabsf(f) * 0x1.0p-21f;  // inf stays inf
}

// `bmin`/`bmax`: 3 quantised coordinates each, as f32 (exact).  Returns false when the box can be
// skipped; *t_entry is the entry distance, used to order the children.
// Per axis: t1 = (bmin - (o + P)) / d, t2 = (bmax - (o - P)) / d -- whichever of
// the two is the entry is lowered by P/|d|, the exit raised.
__device__ __forceinline__ bool box_cull_test(const float *__restrict__ bmin, const float *__restrict__ bmax, const CullRay &c,
                                              float t_limit, float *t_entry) {
    float t1 = __builtin_fmaf(bmin[0], c.ax, c.x1), t2 = __builtin_fmaf(bmax[0], c.ax, c.x2);
    float tn = fminf(t1, t2), tf = fmaxf(t1, t2);
    t1 = __builtin_fmaf(bmin[1], c.ay, c.y1);
    t2 = __builtin_fmaf(bmax[1], c.ay, c.y2);
    tn = fmaxf(tn, fminf(t1, t2));
    tf = fminf(tf, fmaxf(t1, t2));
    t1 = __builtin_fmaf(bmin[2], c.az, c.z1);
    t2 = __builtin_fmaf(bmax[2], c.az, c.z2);
    tn = fmaxf(tn, fminf(t1, t2));
    tf = fminf(tf, fmaxf(t1, t2));
    *t_entry = tn;
    if (tn > tf) return false;
    if (tf < 0.0f) return false;
    if (tn > t_limit) return false;
    return true;
}

}  // namespace rt
