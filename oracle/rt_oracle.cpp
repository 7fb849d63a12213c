// rt_oracle.cpp -- CPU restatement of the reference hot path.  TEST INFRASTRUCTURE.
//
// This file is the parity oracle for the HIP path.  Only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the
// product (raytracer_amd/, include/) never links, imports or calls it.
//
// What it restates (DarkAce65/raytracer v1.0.0, all f64, recursion kept):
//   src/render/raytracing_scene.rs:31-42, 90-622, 757-769   camera rays, get_color,
//        get_illumination, phong / physical shading, GI, per-pixel average + gamma
//   src/render/mod.rs:13-57            GAMMA, BIAS, ColorData clamp/compute_color
//   src/ray_intersection.rs:24-117     Ray::transform, Intersection::compute_data
//   src/core/bounds.rs:22-611          BoundingVolume, kd-tree build (SAH) + traversal
//   src/primitives/{sphere,cube,plane,triangle}.rs   intersect / surface_normal / uv
//   src/core/{material,texture}.rs     get_color, nearest-texel lookup
//   src/lights/point.rs:32-34, src/utils/*.rs
//
// Third-party arithmetic the reference gets from crates that are not under
// /root/reference (nalgebra 0.30.1, Cargo.lock:643; rand 0.8.5, :932) is
// restated from their published definitions: dot/gemv accumulate left to right,
// normalize divides by the norm, Vector3::lerp(a,b,t) = a*(1-t) + b*t,
// center(a,b) = (a+b)*0.5, Rotation3::rotation_between = axis-angle about
// normalize(a x b).  rand::thread_rng (irreproducible, and the reference also
// shuffles pixel order) is replaced by the counter-based Philox4x32-10 stream
// specified in DESIGN.md; sample 0 stays the pixel centre.
//
// Pinning (tests/test_oracle_*.py): the reference's own known-answer vectors
// (utils/mod.rs:55-83, sampling.rs:79-189), its published ray counts
// (renders/renders.md) and its raw renders (renders/{test,mesh,refraction,
// physical,mesh_test}.png) -- the Rust crate itself cannot be built here
// (no cargo/rustc), so no reference binary backs this file.
//
// Accelerator modes:
//   ORC_MODE_KD        the reference's SAH kd-tree and its exact traversal order
//                      (tie-breaks, max_distance threading) -- the CPU baseline.
//   ORC_MODE_CANONICAL traversal-order independent statement of the same hit
//                      predicate: closest = min over primitives of (t, bounded?, -id),
//                      shadow = OR over primitives; every bounded primitive keeps
//                      its own world-AABB slab test (bounds.rs:172-181).  This is
//                      the bit-exact target for the GPU (DESIGN.md "canonical hit").
//   ORC_MODE_BRUTE     canonical semantics with no accelerator at all (small cases).
#include "../include/mi355rt.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <memory>
#include <string>
#include <thread>
#include <vector>

namespace {

constexpr double GAMMA = 2.2;          // render/mod.rs:13
constexpr double BIAS = 1e-10;         // render/mod.rs:14
constexpr double EPS = 2.220446049250313e-16;  // f64::EPSILON
constexpr double PI = 3.14159265358979323846;
constexpr double FRAC_PI_2 = 1.57079632679489661923;
constexpr double FRAC_PI_4 = 0.78539816339744830962;
constexpr double FRAC_1_PI = 0.318309886183790671538;
constexpr double TAU = 6.28318530717958647692;

enum { MODE_KD = 0, MODE_CANONICAL = 1, MODE_BRUTE = 2 };

thread_local std::string g_error;

// ---------------------------------------------------------------- vectors
struct V3 {
    double x, y, z;
    double operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
inline V3 v3(double x, double y, double z) { return V3{x, y, z}; }
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
inline V3 operator*(V3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 operator*(double s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline V3 operator/(V3 a, double s) { return {a.x / s, a.y / s, a.z / s}; }
inline V3 cmul(V3 a, V3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline double norm_sq(V3 a) { return dot(a, a); }
inline double norm(V3 a) { return std::sqrt(norm_sq(a)); }
inline V3 normalize(V3 a) { return a / norm(a); }
inline V3 vlerp(V3 a, V3 b, double t) { return a * (1.0 - t) + b * t; }  // nalgebra Vector3::lerp

// Rust f64 helpers with their exact edge semantics
inline double signum(double x) { return std::isnan(x) ? x : std::copysign(1.0, x); }
inline double rclamp01(double c) {  // f64::clamp(0.0, 1.0): NaN stays NaN
    if (c < 0.0) return 0.0;
    if (c > 1.0) return 1.0;
    return c;
}
inline V3 clamp01(V3 a) { return {rclamp01(a.x), rclamp01(a.y), rclamp01(a.z)}; }
inline uint32_t sat_u32(double x) {  // `x as u32`
    if (!(x > 0.0)) return 0;  // NaN, negatives
    if (x >= 4294967295.0) return 4294967295u;
    return static_cast<uint32_t>(x);
}
inline uint8_t sat_u8(double x) {  // `x as u8`
    if (!(x > 0.0)) return 0;
    if (x >= 255.0) return 255;
    return static_cast<uint8_t>(x);
}
inline double scalar_lerp(double x0, double x1, double t) { return x0 - x0 * t + x1 * t; }  // utils/mod.rs:25-27

// Row-major 3x4 affine block.
struct Affine {
    double m[12];
    V3 point(V3 p) const {  // linear * p + translation
        return {m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3], m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7],
                m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11]};
    }
    V3 vec(V3 v) const {
        return {m[0] * v.x + m[1] * v.y + m[2] * v.z, m[4] * v.x + m[5] * v.y + m[6] * v.z,
                m[8] * v.x + m[9] * v.y + m[10] * v.z};
    }
    V3 vec_transposed(V3 v) const {  // (linear part)^T * v
        return {m[0] * v.x + m[4] * v.y + m[8] * v.z, m[1] * v.x + m[5] * v.y + m[9] * v.z,
                m[2] * v.x + m[6] * v.y + m[10] * v.z};
    }
};

// ---------------------------------------------------------------- RNG
// Philox4x32-10 (Salmon et al., SC'11).  Stream layout (DESIGN.md "RNG"):
//   key     = (seed lo, seed hi)
//   counter = (pixel index, sample index, path lo, path hi)
//   path    = 0 for the camera jitter; child path = path_mix(parent path, slot)
//   slot    = i (reflection sample i) | 0x40000000 (refraction) | 0x80000000 + j (GI sample j)
// One call yields the two uniforms a ray direction needs.
inline void philox4x32_10(const uint32_t ctr_in[4], const uint32_t key_in[2], uint32_t out[4]) {
    uint32_t c0 = ctr_in[0], c1 = ctr_in[1], c2 = ctr_in[2], c3 = ctr_in[3];
    uint32_t k0 = key_in[0], k1 = key_in[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = static_cast<uint64_t>(0xD2511F53u) * c0;
        uint64_t p1 = static_cast<uint64_t>(0xCD9E8D57u) * c2;
        uint32_t n0 = static_cast<uint32_t>(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = static_cast<uint32_t>(p1);
        uint32_t n2 = static_cast<uint32_t>(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = static_cast<uint32_t>(p0);
        c0 = n0;
        c1 = n1;
        c2 = n2;
        c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0;
    out[1] = c1;
    out[2] = c2;
    out[3] = c3;
}

inline uint64_t path_mix(uint64_t path, uint32_t slot) {
    uint64_t z = path * 0xD1342543DE82EF95ull + (static_cast<uint64_t>(slot) + 1ull);
    z ^= z >> 32;
    z *= 0xD6E8FEB86659FD93ull;
    z ^= z >> 32;
    z *= 0xD6E8FEB86659FD93ull;
    z ^= z >> 32;
    return z;
}

constexpr uint32_t SLOT_REFRACTION = 0x40000000u;
constexpr uint32_t SLOT_GI = 0x80000000u;

struct RngCtx {
    uint64_t seed;
    uint32_t pixel, sample;
};

inline void uniforms2(const RngCtx &c, uint64_t path, double *u1, double *u2) {
    uint32_t ctr[4] = {c.pixel, c.sample, static_cast<uint32_t>(path), static_cast<uint32_t>(path >> 32)};
    uint32_t key[2] = {static_cast<uint32_t>(c.seed), static_cast<uint32_t>(c.seed >> 32)};
    uint32_t r[4];
    philox4x32_10(ctr, key, r);
    uint64_t a = static_cast<uint64_t>(r[0]) | (static_cast<uint64_t>(r[1]) << 32);
    uint64_t b = static_cast<uint64_t>(r[2]) | (static_cast<uint64_t>(r[3]) << 32);
    *u1 = static_cast<double>(a >> 11) * 0x1.0p-53;
    *u2 = static_cast<double>(b >> 11) * 0x1.0p-53;
}

// ---------------------------------------------------------------- utils/*.rs
// utils/mod.rs:35-47
inline bool quadratic(double a, double b, double c, double *t0, double *t1) {
    double discriminant = b * b - 4.0 * a * c;
    if (discriminant < 0.0) return false;
    if (discriminant == 0.0) {
        *t0 = -0.5 * b / a;
        *t1 = -0.5 * b / a;
        return true;
    }
    double q = -0.5 * (b + signum(b) * std::sqrt(discriminant));
    double r0 = q / a;
    double r1 = c / q;
    *t0 = std::fmin(r0, r1);
    *t1 = std::fmax(r0, r1);
    return true;
}

// utils/mod.rs:29-33
inline double remap_value(double num, double d0, double d1, double r0, double r1) {
    return (num - d0) * (r1 - r0) / (d1 - d0) + r0;
}

// utils/rays.rs:3-5
inline V3 reflect(V3 incident, V3 normal) { return normalize(incident - (2.0 * dot(incident, normal)) * normal); }

// utils/rays.rs:7-26
inline bool refract(V3 incident, V3 normal, double eta, V3 *out) {
    double n_dot_i = dot(normal, incident);
    V3 refraction_normal = n_dot_i < 0.0 ? normal : -normal;
    eta = n_dot_i < 0.0 ? eta : 1.0 / eta;
    n_dot_i = std::fabs(n_dot_i);
    double k = 1.0 - eta * eta * (1.0 - n_dot_i * n_dot_i);
    if (k < 0.0) return false;
    *out = normalize(incident * eta - refraction_normal * (eta * n_dot_i - std::sqrt(k)));
    return true;
}

// utils/sampling.rs:6-20 with (u1,u2) = Vector2::new_random()
inline void concentric_sample_disk(double u1, double u2, double *px, double *py) {
    double rx = 2.0 * u1 - 1.0, ry = 2.0 * u2 - 1.0;
    if (rx == 0.0 && ry == 0.0) {
        *px = 0.0;
        *py = 0.0;
        return;
    }
    double r, theta;
    if (std::fabs(rx) > std::fabs(ry)) {
        r = rx;
        theta = FRAC_PI_2 * (ry / rx);
    } else {
        r = ry;
        theta = FRAC_PI_2 - FRAC_PI_4 * (rx / ry);
    }
    *px = r * std::cos(theta);
    *py = r * std::sin(theta);
}

// utils/sampling.rs:23-36 (u is NOT normalised -- kept)
inline V3 cosine_sample_hemisphere(V3 direction, double u1, double u2) {
    double px, py;
    concentric_sample_disk(u1, u2, &px, &py);
    double pz = std::sqrt(1.0 - px * px - py * py);
    V3 w = direction;
    V3 u = std::fabs(w.x) > EPS ? cross(direction, v3(0, 1, 0)) : cross(direction, v3(1, 0, 0));
    V3 v = cross(direction, u);
    return normalize(u * px + v * py + w * pz);
}

// utils/sampling.rs:39-70 (u1 = first rng.gen(), u2 = second)
inline V3 uniform_sample_cone(V3 direction, double max_angle, double u1, double u2) {
    if (max_angle < EPS) return direction;
    double theta = std::acos(u1);
    theta = theta * max_angle / FRAC_PI_2;
    double z = std::cos(theta);
    double radius = std::sin(theta);
    double phi = u2 * TAU;
    V3 u = cross(direction, v3(0, 0, 1));
    double mag = norm(u);
    if (mag < EPS) return normalize(v3(radius * std::cos(phi), radius * std::sin(phi), signum(direction.z) * z));
    V3 w = direction;
    u = normalize(u);
    V3 v = normalize(cross(direction, u));
    return normalize(u * radius * std::cos(phi) + v * radius * std::sin(phi) + w * z);
}

// utils/physical_material_equations.rs
inline double ndf(double n_dot_h, double roughness) {
    double a = roughness * roughness;
    double a2 = a * a;
    double n_dot_h2 = n_dot_h * n_dot_h;
    double denom = n_dot_h2 * (a2 - 1.0) + 1.0;
    denom = PI * denom * denom;
    return a2 / denom;
}
inline double geometry_function(double n_dot_v, double n_dot_l, double roughness) {
    double r = roughness + 1.0;
    double k = r * r / 8.0;
    double ggx1 = n_dot_v / (n_dot_v * (1.0 - k) + k);
    double ggx2 = n_dot_l / (n_dot_l * (1.0 - k) + k);
    return ggx1 * ggx2;
}
inline V3 fresnel(double n_dot_v, V3 base) {
    return base + (v3(1, 1, 1) - base) * std::pow(1.0 - n_dot_v, 5.0);
}

// ---------------------------------------------------------------- scene
enum RayType { RAY_PRIMARY = 0, RAY_SECONDARY = 1, RAY_SHADOW = 2 };

struct Ray {
    RayType type;
    uint32_t depth;  // Secondary(depth); Primary = 0
    V3 origin, direction;
    double refractive_index;
};

struct Prim {
    uint32_t type, id;
    const rt_material *material;
    Affine world, inv;
    double param;
    V3 plane_normal;
    V3 p[3], n[3];
    double uv[3][2];
    double plane_rot[9];  // rotation_between(normal, +y), plane.rs:138
    bool plane_rot_ok;
};

struct Hit {
    const Prim *prim = nullptr;
    double distance = 0.0;
    int kind = 0;  // 0 Empty, 1 CubeHitFace, 2 Barycentric
    int axis = 0;
    bool positive = false;
    double bu = 0, bv = 0, bw = 0;
};

struct Bounds {  // bounds.rs:22-27
    V3 center, bmin, bmax;
};

inline Bounds bounds_from(V3 bmin, V3 bmax) { return {(bmin + bmax) * 0.5, bmin, bmax}; }

// bounds.rs:40-72
Bounds bounds_from_transform(V3 bmin, V3 bmax, const Affine &xf) {
    V3 verts[8];
    int i = 0;
    const double xs[2] = {bmin.x, bmax.x}, ys[2] = {bmin.y, bmax.y}, zs[2] = {bmin.z, bmax.z};
    for (double x : xs)
        for (double y : ys)
            for (double z : zs) verts[i++] = v3(x, y, z);
    V3 mn = xf.point(verts[0]);
    V3 mx = mn;
    for (int k = 1; k < 8; ++k) {
        V3 t = xf.point(verts[k]);
        mn.x = std::fmin(mn.x, t.x);
        mn.y = std::fmin(mn.y, t.y);
        mn.z = std::fmin(mn.z, t.z);
        mx.x = std::fmax(mx.x, t.x);
        mx.y = std::fmax(mx.y, t.y);
        mx.z = std::fmax(mx.z, t.z);
    }
    return bounds_from(mn, mx);
}

// bounds.rs:114-152
inline bool bounds_intersect(const Bounds &b, const Ray &ray, bool has_max, double max_distance) {
    V3 tc = b.center - ray.origin;
    V3 half = (b.bmax - b.bmin) / 2.0;
    half = cmul(half, v3(signum(ray.direction.x), signum(ray.direction.y), signum(ray.direction.z)));

    double d_near = (tc.x - half.x) / ray.direction.x;
    double d_far = (tc.x + half.x) / ray.direction.x;
    double dy_near = (tc.y - half.y) / ray.direction.y;
    double dy_far = (tc.y + half.y) / ray.direction.y;
    if (dy_far < d_near || d_far < dy_near) return false;
    d_near = dy_near > d_near ? dy_near : d_near;
    d_far = d_far > dy_far ? dy_far : d_far;

    double dz_near = (tc.z - half.z) / ray.direction.z;
    double dz_far = (tc.z + half.z) / ray.direction.z;
    if (dz_far < d_near || d_far < dz_near) return false;
    d_near = dz_near > d_near ? dz_near : d_near;
    d_far = d_far > dz_far ? dz_far : d_far;

    if (d_near < 0.0 && d_far < 0.0) return false;
    if (has_max && max_distance < d_near) return false;
    return true;
}

// ray_intersection.rs:33-43 (direction NOT renormalised => t is preserved)
inline Ray ray_transform(const Ray &r, const Affine &xf) {
    Ray o = r;
    o.origin = xf.point(r.origin);
    o.direction = xf.vec(r.direction);
    return o;
}

inline bool two_sided(const Prim &p, const Ray &ray) { return p.material->side == RT_SIDE_BOTH || ray.type == RAY_SHADOW; }

// sphere.rs:101-130
bool sphere_intersect(const Prim &p, const Ray &ray, bool has_max, double max_distance, Hit *hit) {
    V3 hypot = ray.origin;
    double ray_proj = dot(hypot, ray.direction);
    double a = norm_sq(ray.direction);
    double b = 2.0 * ray_proj;
    double c = norm_sq(hypot) - p.param * p.param;
    double t0, t1;
    if (!quadratic(a, b, c, &t0, &t1)) return false;
    double distance;
    if (two_sided(p, ray)) distance = t0 < 0.0 ? t1 : t0;
    else if (p.material->side == RT_SIDE_FRONT) distance = t0;
    else distance = t1;
    if (distance < 0.0 || (has_max && max_distance < distance)) return false;
    hit->prim = &p;
    hit->distance = distance;
    hit->kind = 0;
    return true;
}

// cube.rs:100-174
bool cube_intersect(const Prim &p, const Ray &ray, bool has_max, double max_distance, Hit *hit) {
    V3 s = v3(signum(ray.direction.x), signum(ray.direction.y), signum(ray.direction.z));
    double half = p.param / 2.0;
    int near_axis = 0, far_axis = 0;
    bool near_pos = s.x < 0.0, far_pos = s.x > 0.0;

    double d_near = (-ray.origin.x - s.x * half) / ray.direction.x;
    double d_far = (-ray.origin.x + s.x * half) / ray.direction.x;
    double dy_near = (-ray.origin.y - s.y * half) / ray.direction.y;
    double dy_far = (-ray.origin.y + s.y * half) / ray.direction.y;
    if (dy_far < d_near || d_far < dy_near) return false;
    if (dy_near > d_near) {
        near_axis = 1;
        near_pos = s.y < 0.0;
        d_near = dy_near;
    }
    if (d_far > dy_far) {
        far_axis = 1;
        far_pos = s.y > 0.0;
        d_far = dy_far;
    }
    double dz_near = (-ray.origin.z - s.z * half) / ray.direction.z;
    double dz_far = (-ray.origin.z + s.z * half) / ray.direction.z;
    if (dz_far < d_near || d_far < dz_near) return false;
    if (dz_near > d_near) {
        near_axis = 2;
        near_pos = s.z < 0.0;
        d_near = dz_near;
    }
    if (d_far > dz_far) {
        far_axis = 2;
        far_pos = s.z > 0.0;
        d_far = dz_far;
    }
    double distance;
    int axis;
    bool positive;
    bool use_near;
    if (two_sided(p, ray)) use_near = !(d_near < 0.0);
    else use_near = p.material->side == RT_SIDE_FRONT;
    if (use_near) {
        distance = d_near;
        axis = near_axis;
        positive = near_pos;
    } else {
        distance = d_far;
        axis = far_axis;
        positive = far_pos;
    }
    if (distance < 0.0 || (has_max && max_distance < distance)) return false;
    hit->prim = &p;
    hit->distance = distance;
    hit->kind = 1;
    hit->axis = axis;
    hit->positive = positive;
    return true;
}

// plane.rs:98-117
bool plane_intersect(const Prim &p, const Ray &ray, bool has_max, double max_distance, Hit *hit) {
    double n_dot_v = dot(p.plane_normal, -ray.direction);
    bool reject;
    if (two_sided(p, ray)) reject = std::fabs(n_dot_v) < EPS;
    else if (p.material->side == RT_SIDE_FRONT) reject = n_dot_v < EPS;
    else reject = -n_dot_v < EPS;
    if (reject) return false;
    double distance = dot(ray.origin, p.plane_normal) / n_dot_v;
    if (distance < 0.0 || (has_max && max_distance < distance)) return false;
    hit->prim = &p;
    hit->distance = distance;
    hit->kind = 0;
    return true;
}

// triangle.rs:193-232
bool triangle_intersect(const Prim &p, const Ray &ray, bool has_max, double max_distance, Hit *hit) {
    V3 edge1 = p.p[1] - p.p[0];
    V3 edge2 = p.p[2] - p.p[0];
    V3 p_vec = cross(ray.direction, edge2);
    double det = dot(edge1, p_vec);
    bool reject;
    if (two_sided(p, ray)) reject = std::fabs(det) < EPS;
    else if (p.material->side == RT_SIDE_FRONT) reject = det < EPS;
    else reject = -det < EPS;
    if (reject) return false;
    V3 t_vec = ray.origin - p.p[0];
    double u = dot(t_vec, p_vec) / det;
    if (!(u >= 0.0 && u <= 1.0)) return false;  // !(0.0..=1.0).contains(&u)
    V3 q_vec = cross(t_vec, edge1);
    double v = dot(ray.direction, q_vec) / det;
    if (v < 0.0 || 1.0 < u + v) return false;
    double distance = dot(edge2, q_vec) / det;
    if (distance < 0.0 || (has_max && max_distance < distance)) return false;
    hit->prim = &p;
    hit->distance = distance;
    hit->kind = 2;
    hit->bu = u;
    hit->bv = v;
    hit->bw = 1.0 - u - v;
    return true;
}

inline bool prim_intersect_object_space(const Prim &p, const Ray &oray, bool has_max, double max_distance, Hit *hit) {
    switch (p.type) {
        case RT_PRIM_SPHERE: return sphere_intersect(p, oray, has_max, max_distance, hit);
        case RT_PRIM_CUBE: return cube_intersect(p, oray, has_max, max_distance, hit);
        case RT_PRIM_PLANE: return plane_intersect(p, oray, has_max, max_distance, hit);
        default: return triangle_intersect(p, oray, has_max, max_distance, hit);
    }
}

struct KdNode {  // bounds.rs:421-431
    bool leaf = true;
    int axis = 0;
    double split = 0.0;
    Bounds bv{};
    std::unique_ptr<KdNode> left, right;
    std::vector<uint32_t> indexes;
};

struct BvhNode {  // canonical-mode accelerator (oracle-private, median split)
    V3 bmin, bmax;
    int left = -1, right = -1;
    uint32_t first = 0, count = 0;
};

struct Scene {
    rt_render_options opts;
    rt_camera camera;
    std::vector<rt_material> materials;
    std::vector<rt_light> lights;
    std::vector<rt_texture> textures;
    std::vector<std::vector<uint8_t>> texture_data;
    std::vector<Prim> prims;               // primitive-ID order
    std::vector<uint32_t> unbounded;       // prim ids (planes), scene order
    std::vector<uint32_t> bounded;         // prim ids, scene order (bounded_objects)
    std::vector<Bounds> bounded_bv;        // per bounded object
    std::unique_ptr<KdNode> kd;
    std::vector<BvhNode> bvh;
    std::vector<uint32_t> bvh_order;       // indices into `bounded`
    double pad = 0.0;                      // canonical culling pad
    uint64_t kd_nodes = 0, kd_leaves = 0, kd_refs = 0;
    int kd_depth = 0;
};

// ---- kd-tree build, bounds.rs:452-610 -------------------------------------
struct SplitCandidate {
    bool is_start;
    double split;
    uint32_t index;
};

inline bool split_less(const SplitCandidate &a, const SplitCandidate &b) {  // SplitCandidate::cmp == Less
    if (a.split < b.split) return true;
    if (a.split > b.split) return false;
    if (a.split == b.split) return a.is_start && !b.is_start;
    return false;  // unordered -> Equal
}

std::unique_ptr<KdNode> kd_build(Scene &sc, int max_depth, int max_bad_refines, const Bounds &bv,
                                 std::vector<uint32_t> indexes, int depth) {
    const size_t max_objects = 2;
    const double intersection_cost = 80.0, traversal_cost = 1.0, empty_bonus_opt = 0.5;
    if (indexes.empty()) return nullptr;
    if (depth > sc.kd_depth) sc.kd_depth = depth;
    if (indexes.size() <= max_objects || max_depth == 0) {
        auto leaf = std::make_unique<KdNode>();
        leaf->indexes = std::move(indexes);
        sc.kd_leaves++;
        sc.kd_refs += leaf->indexes.size();
        return leaf;
    }
    // maximum_extent, bounds.rs:88-104
    double dx = bv.bmax.x - bv.bmin.x, dy = bv.bmax.y - bv.bmin.y, dz = bv.bmax.z - bv.bmin.z;
    int split_axis;
    if (dx >= dy) split_axis = dx >= dz ? 0 : 2;
    else split_axis = dy >= dz ? 1 : 2;
    double total_sa = 2.0 * (dx * dy + dy * dz + dx * dz);
    V3 diag = bv.bmax - bv.bmin;
    double old_cost = intersection_cost * static_cast<double>(indexes.size());

    int split_attempts = 0;
    std::vector<SplitCandidate> cands;
    bool have_best = false;
    int best_axis = 0;
    size_t best_index = 0;
    double best_cost = std::numeric_limits<double>::infinity();

    for (int a = 0; a < 3; ++a) {
        int axis = (split_axis + a) % 3;
        cands.clear();
        for (uint32_t index : indexes) {
            const Bounds &ob = sc.bounded_bv[index];
            cands.push_back({true, ob.bmin[axis], index});
            cands.push_back({false, ob.bmax[axis], index});
        }
        std::stable_sort(cands.begin(), cands.end(), split_less);

        size_t below = 0, above = indexes.size();
        for (size_t i = 0; i < cands.size(); ++i) {
            const SplitCandidate &c = cands[i];
            if (!c.is_start) above -= 1;
            double split = c.split;
            if (bv.bmin[axis] < split && split < bv.bmax[axis]) {
                int o0 = (axis + 1) % 3, o1 = (axis + 2) % 3;
                double d = diag[o0] * diag[o1];
                double sa_below = 2.0 * (d + (split - bv.bmin[axis]) * (diag[o0] + diag[o1]));
                double sa_above = 2.0 * (d + (bv.bmax[axis] - split) * (diag[o0] + diag[o1]));
                double area_below = sa_below / total_sa;
                double area_above = sa_above / total_sa;
                double empty_bonus = (above == 0 || below == 0) ? empty_bonus_opt : 0.0;
                double cost = traversal_cost + intersection_cost * (1.0 - empty_bonus) *
                                                   (area_below * static_cast<double>(below) +
                                                    area_above * static_cast<double>(above));
                if (cost < best_cost) {
                    best_cost = cost;
                    have_best = true;
                    best_axis = axis;
                    best_index = i;
                }
            }
            if (c.is_start) below += 1;
        }
        if (!have_best && split_attempts < 2) {
            split_attempts += 1;
            continue;
        }
        if (best_cost > old_cost) max_bad_refines -= 1;
        if (!have_best || max_bad_refines == 0 || (best_cost > 4.0 * old_cost && indexes.size() < 16)) {
            auto leaf = std::make_unique<KdNode>();
            leaf->indexes = std::move(indexes);
            sc.kd_leaves++;
            sc.kd_refs += leaf->indexes.size();
            return leaf;
        }
        break;
    }
    // A best split is always found on the axis examined last (finding one ends the
    // loop), so `cands` is that axis' sorted candidate list (bounds.rs:552-553).
    double split_location = cands[best_index].split;
    std::vector<uint32_t> left, right;
    for (size_t i = 0; i < cands.size(); ++i) {
        if (i < best_index) {
            if (cands[i].is_start) left.push_back(cands[i].index);
        } else if (i > best_index) {
            if (!cands[i].is_start) right.push_back(cands[i].index);
        }
    }
    V3 left_max = bv.bmax;
    V3 right_min = bv.bmin;
    if (best_axis == 0) left_max.x = right_min.x = split_location;
    else if (best_axis == 1) left_max.y = right_min.y = split_location;
    else left_max.z = right_min.z = split_location;
    auto l = kd_build(sc, max_depth - 1, max_bad_refines, bounds_from(bv.bmin, left_max), std::move(left), depth + 1);
    auto r = kd_build(sc, max_depth - 1, max_bad_refines, bounds_from(right_min, bv.bmax), std::move(right), depth + 1);
    if (l && r) {
        auto node = std::make_unique<KdNode>();
        node->leaf = false;
        node->axis = best_axis;
        node->split = split_location;
        node->bv = bv;
        node->left = std::move(l);
        node->right = std::move(r);
        sc.kd_nodes++;
        return node;
    }
    if (l) return l;
    return r;
}

// ---- canonical-mode BVH (oracle-private) ----------------------------------
int bvh_build(Scene &sc, uint32_t first, uint32_t count) {
    BvhNode node;
    const double inf = std::numeric_limits<double>::infinity();
    node.bmin = v3(inf, inf, inf);
    node.bmax = v3(-inf, -inf, -inf);
    V3 cmin = node.bmin, cmax = node.bmax;
    for (uint32_t i = first; i < first + count; ++i) {
        const Bounds &b = sc.bounded_bv[sc.bvh_order[i]];
        node.bmin = v3(std::fmin(node.bmin.x, b.bmin.x), std::fmin(node.bmin.y, b.bmin.y), std::fmin(node.bmin.z, b.bmin.z));
        node.bmax = v3(std::fmax(node.bmax.x, b.bmax.x), std::fmax(node.bmax.y, b.bmax.y), std::fmax(node.bmax.z, b.bmax.z));
        cmin = v3(std::fmin(cmin.x, b.center.x), std::fmin(cmin.y, b.center.y), std::fmin(cmin.z, b.center.z));
        cmax = v3(std::fmax(cmax.x, b.center.x), std::fmax(cmax.y, b.center.y), std::fmax(cmax.z, b.center.z));
    }
    node.first = first;
    node.count = count;
    int self = static_cast<int>(sc.bvh.size());
    sc.bvh.push_back(node);
    if (count <= 4) return self;
    V3 ext = cmax - cmin;
    int axis = ext.x >= ext.y ? (ext.x >= ext.z ? 0 : 2) : (ext.y >= ext.z ? 1 : 2);
    uint32_t mid = first + count / 2;
    std::nth_element(sc.bvh_order.begin() + first, sc.bvh_order.begin() + mid, sc.bvh_order.begin() + first + count,
                     [&](uint32_t a, uint32_t b) {
                         double ca = sc.bounded_bv[a].center[axis], cb = sc.bounded_bv[b].center[axis];
                         return ca < cb || (ca == cb && a < b);
                     });
    int l = bvh_build(sc, first, mid - first);
    int r = bvh_build(sc, mid, first + count - mid);
    sc.bvh[self].left = l;
    sc.bvh[self].right = r;
    sc.bvh[self].count = 0;
    return self;
}

// Conservative slab test on a box padded by `pad` (never culls a primitive whose
// own exact predicate passes; only used to skip work in canonical mode).
inline bool padded_box_hit(V3 bmin, V3 bmax, double pad, const Ray &ray, double tmax) {
    double tn = -std::numeric_limits<double>::infinity(), tf = std::numeric_limits<double>::infinity();
    for (int a = 0; a < 3; ++a) {
        double o = ray.origin[a], d = ray.direction[a];
        double lo = bmin[a] - pad, hi = bmax[a] + pad;
        if (d == 0.0 || std::isnan(d)) {
            if (o < lo || o > hi) return false;
            continue;
        }
        double t1 = (lo - o) / d, t2 = (hi - o) / d;
        if (t1 > t2) std::swap(t1, t2);
        if (t1 > tn) tn = t1;
        if (t2 < tf) tf = t2;
    }
    double slack = 1e-9 * (std::fabs(tn) + std::fabs(tf)) + 1e-300;
    if (tn > tf + slack) return false;
    if (tf < -slack) return false;
    if (tn > tmax + slack) return false;
    return true;
}

struct Tracer {
    const Scene &sc;
    int mode;
    explicit Tracer(const Scene &s, int m) : sc(s), mode(m) {}

    // bounds.rs:158-181
    bool unbounded_intersect(uint32_t pid, const Ray &ray, bool has_max, double max_distance, Hit *hit) const {
        const Prim &p = sc.prims[pid];
        Ray o = ray_transform(ray, p.inv);
        return prim_intersect_object_space(p, o, has_max, max_distance, hit);
    }
    bool bounded_intersect(uint32_t bidx, const Ray &ray, bool has_max, double max_distance, Hit *hit) const {
        if (!bounds_intersect(sc.bounded_bv[bidx], ray, has_max, max_distance)) return false;
        const Prim &p = sc.prims[sc.bounded[bidx]];
        Ray o = ray_transform(ray, p.inv);
        return prim_intersect_object_space(p, o, has_max, max_distance, hit);
    }

    // ---------------- reference traversal, bounds.rs:318-364
    bool kd_raycast(const KdNode *n, const Ray &ray, bool has_max, double max_distance, Hit *out) const {
        if (!n->leaf) {
            if (!bounds_intersect(n->bv, ray, has_max, max_distance)) return false;
            double o = ray.origin[n->axis], d = ray.direction[n->axis];
            bool left_first = o < n->split || (std::fabs(o - n->split) < EPS && d <= 0.0);
            const KdNode *first = left_first ? n->left.get() : n->right.get();
            const KdNode *second = left_first ? n->right.get() : n->left.get();
            Hit close;
            if (!kd_raycast(first, ray, has_max, max_distance, &close)) return kd_raycast(second, ray, has_max, max_distance, out);
            Hit far;
            if (kd_raycast(second, ray, true, close.distance, &far) && far.distance < close.distance) *out = far;
            else *out = close;  // min_by keeps the first of equal elements
            return true;
        }
        bool found = false;
        for (uint32_t idx : n->indexes) {
            Hit h;
            if (bounded_intersect(idx, ray, has_max, max_distance, &h)) {
                if (!found || h.distance < out->distance) *out = h;
                found = true;
            }
        }
        return found;
    }
    // bounds.rs:366-399
    bool kd_shadow(const KdNode *n, const Ray &ray, double max_distance) const {
        if (!n->leaf) {
            if (!bounds_intersect(n->bv, ray, true, max_distance)) return false;
            double o = ray.origin[n->axis], d = ray.direction[n->axis];
            bool left_first = o < n->split || (std::fabs(o - n->split) < EPS && d <= 0.0);
            const KdNode *first = left_first ? n->left.get() : n->right.get();
            const KdNode *second = left_first ? n->right.get() : n->left.get();
            return kd_shadow(first, ray, max_distance) || kd_shadow(second, ray, max_distance);
        }
        for (uint32_t idx : n->indexes) {
            Hit h;
            if (bounded_intersect(idx, ray, true, max_distance, &h)) return true;
        }
        return false;
    }

    // ---------------- canonical semantics
    static bool better(const Hit &h, bool h_bounded, const Hit &best, bool best_bounded) {
        if (h.distance < best.distance) return true;
        if (best.distance < h.distance) return false;
        if (h.distance == best.distance) {
            if (h_bounded != best_bounded) return !h_bounded;
            return h.prim->id > best.prim->id;  // the later primitive first: what MODE_KD gives on cornell_box's coplanar light / ceiling
        }
        return false;  // NaN: keep the earlier one
    }

    bool canonical_raycast(const Ray &ray, Hit *out) const {
        bool found = false, found_bounded = false;
        for (uint32_t pid : sc.unbounded) {
            Hit h;
            if (unbounded_intersect(pid, ray, false, 0.0, &h)) {
                if (!found || better(h, false, *out, false)) *out = h;
                found = true;
            }
        }
        auto consider = [&](uint32_t bidx) {
            Hit h;
            if (bounded_intersect(bidx, ray, false, 0.0, &h)) {
                if (!found || better(h, true, *out, found_bounded)) {
                    *out = h;
                    found_bounded = true;
                }
                found = true;
            }
        };
        if (mode == MODE_BRUTE || sc.bvh.empty()) {
            for (uint32_t b = 0; b < sc.bounded.size(); ++b) consider(b);
            return found;
        }
        int stack[128];
        int sp = 0;
        stack[sp++] = 0;
        while (sp > 0) {
            const BvhNode &n = sc.bvh[stack[--sp]];
            double tmax = found ? out->distance : std::numeric_limits<double>::infinity();
            if (!padded_box_hit(n.bmin, n.bmax, sc.pad, ray, tmax)) continue;
            if (n.left < 0) {
                for (uint32_t i = n.first; i < n.first + n.count; ++i) consider(sc.bvh_order[i]);
            } else {
                stack[sp++] = n.right;
                stack[sp++] = n.left;
            }
        }
        return found;
    }

    bool canonical_shadow(const Ray &ray, double max_distance) const {
        for (uint32_t pid : sc.unbounded) {
            Hit h;
            if (unbounded_intersect(pid, ray, true, max_distance, &h) && h.distance <= max_distance) return true;
        }
        if (mode == MODE_BRUTE || sc.bvh.empty()) {
            for (uint32_t b = 0; b < sc.bounded.size(); ++b) {
                Hit h;
                if (bounded_intersect(b, ray, true, max_distance, &h)) return true;
            }
            return false;
        }
        int stack[128];
        int sp = 0;
        stack[sp++] = 0;
        while (sp > 0) {
            const BvhNode &n = sc.bvh[stack[--sp]];
            if (!padded_box_hit(n.bmin, n.bmax, sc.pad, ray, max_distance)) continue;
            if (n.left < 0) {
                for (uint32_t i = n.first; i < n.first + n.count; ++i) {
                    Hit h;
                    if (bounded_intersect(sc.bvh_order[i], ray, true, max_distance, &h)) return true;
                }
            } else {
                stack[sp++] = n.right;
                stack[sp++] = n.left;
            }
        }
        return false;
    }

    // KdTreeAccelerator::raycast, bounds.rs:302-308
    bool raycast(const Ray &ray, Hit *out) const {
        if (mode != MODE_KD) return canonical_raycast(ray, out);
        bool found = false;
        for (uint32_t pid : sc.unbounded) {
            Hit h;
            if (unbounded_intersect(pid, ray, false, 0.0, &h)) {
                if (!found || h.distance < out->distance) *out = h;
                found = true;
            }
        }
        Hit th;
        if (sc.kd && kd_raycast(sc.kd.get(), ray, false, 0.0, &th)) {
            if (!found || th.distance < out->distance) *out = th;
            found = true;
        }
        return found;
    }

    // KdTreeAccelerator::shadow_cast, bounds.rs:310-316 (+ raytracing_scene.rs:94-96)
    bool shadow_cast(const Ray &ray, double max_distance) const {
        if (mode != MODE_KD) return canonical_shadow(ray, max_distance);
        for (uint32_t pid : sc.unbounded) {
            Hit h;
            if (unbounded_intersect(pid, ray, true, max_distance, &h) && h.distance <= max_distance) return true;
        }
        return sc.kd && kd_shadow(sc.kd.get(), ray, max_distance);
    }
};

// ---------------------------------------------------------------- surface data
struct Surface {
    V3 hit_point, normal;
    double u, v;
};

// core/texture.rs:46-69
V3 texture_color(const rt_texture &t, double u, double v) {
    uint32_t w = t.width - 1, h = t.height - 1;
    double x = std::fmod(u, 1.0), y = std::fmod(v, 1.0);
    x = x < 0.0 ? x + 1.0 : x;
    y = y < 0.0 ? y + 1.0 : y;
    x = x * static_cast<double>(w);
    y = (1.0 - y) * static_cast<double>(h);
    uint32_t xi = std::min(sat_u32(x), w), yi = std::min(sat_u32(y), h);
    const uint8_t *px = t.rgb8 + (static_cast<size_t>(yi) * t.width + xi) * 3;
    return v3(static_cast<double>(px[0]) / 255.0, static_cast<double>(px[1]) / 255.0, static_cast<double>(px[2]) / 255.0);
}

// material.rs:50-57, 92-99
V3 material_color(const Scene &sc, const rt_material &m, const Surface &s) {
    V3 c = v3(m.color[0], m.color[1], m.color[2]);
    if (m.texture_id < 0) return c;
    return cmul(c, texture_color(sc.textures[m.texture_id], s.u, s.v));
}

// ray_intersection.rs:86-117 + the primitives' surface_normal / uv
Surface compute_data(const Hit &hit, const Ray &ray) {
    const Prim &p = *hit.prim;
    Surface s;
    s.hit_point = ray.origin + ray.direction * hit.distance;
    V3 ohp = p.inv.point(s.hit_point);
    V3 on;
    switch (p.type) {
        case RT_PRIM_SPHERE: on = normalize(ohp); break;                     // sphere.rs:143-149
        case RT_PRIM_CUBE: {                                                 // cube.rs:188-210
            on = hit.axis == 0 ? v3(1, 0, 0) : (hit.axis == 1 ? v3(0, 1, 0) : v3(0, 0, 1));
            if (!hit.positive) on = -on;
            break;
        }
        case RT_PRIM_PLANE: on = p.plane_normal; break;                      // plane.rs:124-130
        default:                                                             // triangle.rs:253-268
            on = normalize(hit.bw * p.n[0] + hit.bu * p.n[1] + hit.bv * p.n[2]);
            break;
    }
    V3 n = normalize(p.inv.vec_transposed(on));  // inverse_transpose * object_normal
    switch (p.material->side) {
        case RT_SIDE_BOTH:
            if (dot(n, ray.direction) > 0.0) n = -n;
            break;
        case RT_SIDE_FRONT: break;
        default: n = -n; break;
    }
    s.normal = n;
    switch (p.type) {
        case RT_PRIM_SPHERE: {  // sphere.rs:151-163
            V3 hp = ohp / p.param;
            s.u = std::atan2(hp.x, hp.z) * FRAC_1_PI * 0.5 + 0.5;
            s.v = std::asin(hp.y) * FRAC_1_PI + 0.5;
            break;
        }
        case RT_PRIM_CUBE: {  // cube.rs:212-240
            V3 hp = v3(ohp.x / p.param + 0.5, ohp.y / p.param + 0.5, ohp.z / p.param + 0.5);
            if (hit.positive) {
                if (hit.axis == 0) { s.u = -hp.z; s.v = hp.y; }
                else if (hit.axis == 1) { s.u = hp.x; s.v = -hp.z; }
                else { s.u = hp.x; s.v = hp.y; }
            } else {
                if (hit.axis == 0) { s.u = hp.z; s.v = hp.y; }
                else if (hit.axis == 1) { s.u = hp.x; s.v = hp.z; }
                else { s.u = -hp.x; s.v = hp.y; }
            }
            break;
        }
        case RT_PRIM_PLANE: {  // plane.rs:132-142 (unwrap() panics for normal = -y; u,v = NaN here)
            if (p.plane_rot_ok) {
                const double *r = p.plane_rot;
                s.u = r[0] * ohp.x + r[1] * ohp.y + r[2] * ohp.z;
                s.v = r[6] * ohp.x + r[7] * ohp.y + r[8] * ohp.z;
            } else {
                s.u = s.v = std::numeric_limits<double>::quiet_NaN();
            }
            break;
        }
        default:  // triangle.rs:270-284
            s.u = hit.bw * p.uv[0][0] + hit.bu * p.uv[1][0] + hit.bv * p.uv[2][0];
            s.v = hit.bw * p.uv[0][1] + hit.bu * p.uv[1][1] + hit.bv * p.uv[2][1];
            break;
    }
    return s;
}

// ---------------------------------------------------------------- integrator
struct ColorData {
    V3 color{0, 0, 0}, albedo{0, 0, 0}, normal{0, 0, 1};
};

struct Integrator {
    const Scene &sc;
    Tracer tr;
    RngCtx rng;
    uint64_t ray_count = 0;
    Integrator(const Scene &s, int mode, uint64_t seed) : sc(s), tr(s, mode), rng{seed, 0, 0} {}

    static uint32_t pow_u(uint32_t b, uint32_t e) {
        uint32_t r = 1;
        for (uint32_t i = 0; i < e; ++i) r *= b;
        return r;
    }
    uint32_t illumination_rays(uint32_t depth) const {  // raytracing_scene.rs:104-105
        return std::max(sc.opts.max_illumination_rays / pow_u(4, depth), 1u);
    }
    uint32_t reflected_rays(uint32_t depth) const {  // :279-280, :363-364
        return std::max(sc.opts.max_reflected_rays / pow_u(8, depth), 1u);
    }
    static V3 light_color(const rt_light &l, double distance) {  // lights/point.rs:32-34
        V3 c = v3(l.intensity * l.color[0], l.intensity * l.color[1], l.intensity * l.color[2]) / (distance * distance);
        return clamp01(c);
    }

    // raytracing_scene.rs:98-133
    void compute_global_illumination(const Surface &s, uint32_t depth, uint64_t path, V3 *emissive, double *ao) {
        uint32_t n = illumination_rays(depth);
        V3 incoming = v3(0, 0, 0);
        uint32_t unoccluded = 0;
        for (uint32_t j = 0; j < n; ++j) {
            uint64_t child = path_mix(path, SLOT_GI + j);
            double u1, u2;
            uniforms2(rng, child, &u1, &u2);
            V3 direction = cosine_sample_hemisphere(s.normal, u1, u2);
            Ray r{RAY_SECONDARY, depth + 1, s.hit_point + direction * BIAS, direction, 1.0};
            bool occluded;
            V3 e = get_illumination(r, child, &occluded);
            incoming = incoming + e;
            if (!occluded) unoccluded += 1;
        }
        *emissive = incoming / static_cast<double>(n);
        *ao = static_cast<double>(unoccluded) / static_cast<double>(n);
    }

    // :135-167
    V3 get_emissive_phong(const Ray &ray, const Surface &s, const rt_material &m, uint64_t path) {
        V3 mc = material_color(sc, m, s);
        V3 emissive = v3(m.emissive[0], m.emissive[1], m.emissive[2]);
        if (m.reflectivity > 0.0) {
            V3 dir = reflect(ray.direction, s.normal);
            Ray r{RAY_SECONDARY, ray.depth + 1, s.hit_point + dir * BIAS, dir, 1.0};
            bool occ;
            V3 incoming = get_illumination(r, path_mix(path, 0), &occ);
            emissive = emissive + cmul(incoming, mc) * m.reflectivity;
        }
        return emissive;
    }

    // :169-261
    ColorData get_color_phong(const Ray &ray, const Surface &s, const rt_material &m, uint64_t path) {
        uint32_t depth = ray.depth;
        V3 mc = material_color(sc, m, s);
        bool has_reflection = false;
        ColorData reflection;
        if (m.reflectivity > 0.0) {
            V3 dir = reflect(ray.direction, s.normal);
            Ray r{RAY_SECONDARY, depth + 1, s.hit_point + dir * BIAS, dir, 1.0};
            reflection = get_color(r, path_mix(path, 0));
            reflection.color = cmul(reflection.color, mc);
            has_reflection = true;
        }
        V3 ambient = v3(0, 0, 0), irradiance = v3(0, 0, 0);
        if (m.reflectivity < 1.0) {
            for (const rt_light &l : sc.lights) {
                if (l.kind == RT_LIGHT_AMBIENT) {
                    ambient = ambient + cmul(v3(l.color[0], l.color[1], l.color[2]), mc);
                    continue;
                }
                V3 lp = v3(l.position[0], l.position[1], l.position[2]);
                V3 ld = lp - s.hit_point;
                double ldist = norm(ld);
                ld = normalize(ld);
                double n_dot_l = dot(s.normal, ld);
                if (n_dot_l > 0.0) {
                    Ray sr{RAY_SHADOW, 0, lp, -ld, 1.0};
                    ray_count += 1;
                    if (!tr.shadow_cast(sr, ldist - BIAS)) {
                        V3 lc = light_color(l, ldist);
                        irradiance = irradiance + cmul(lc, mc) * n_dot_l;
                        V3 half_vec = normalize(ld - ray.direction);
                        double n_dot_h = dot(s.normal, half_vec);
                        if (n_dot_h > 0.0)
                            irradiance = irradiance + cmul(lc, v3(m.specular[0], m.specular[1], m.specular[2])) *
                                                          std::pow(n_dot_h, m.shininess);
                    }
                }
            }
        }
        V3 gi;
        double ao;
        compute_global_illumination(s, depth, path, &gi, &ao);
        ColorData cd;
        cd.color = v3(m.emissive[0], m.emissive[1], m.emissive[2]) + (ambient + irradiance + cmul(gi, mc)) * ao;
        cd.albedo = mc;
        cd.normal = s.normal;
        if (has_reflection) cd.color = vlerp(cd.color, clamp01(reflection.color), m.reflectivity);
        return cd;
    }

    // :263-337
    V3 get_emissive_physical(const Ray &ray, const Surface &s, const rt_material &m, uint64_t path) {
        uint32_t depth = ray.depth;
        V3 mc = material_color(sc, m, s);
        bool has_refl = false, has_refr = false;
        V3 reflected = v3(0, 0, 0), refracted = v3(0, 0, 0);
        if (sc.opts.max_reflected_rays > 0) {
            uint32_t n = reflected_rays(depth);
            double max_angle = FRAC_PI_2 * m.roughness;
            V3 rdir = reflect(ray.direction, s.normal);
            V3 acc = v3(0, 0, 0);
            for (uint32_t i = 0; i < n; ++i) {
                uint64_t child = path_mix(path, i);
                double u1, u2;
                uniforms2(rng, child, &u1, &u2);
                V3 dir = uniform_sample_cone(rdir, max_angle, u1, u2);
                Ray r{RAY_SECONDARY, depth + 1, s.hit_point + dir * BIAS, dir, 1.0};
                bool occ;
                acc = acc + get_illumination(r, child, &occ);
            }
            acc = acc * (FRAC_PI_2 / static_cast<double>(n));
            reflected = cmul(acc, mc);
            has_refl = true;
        }
        if (m.opacity < 1.0) {
            double eta = ray.refractive_index / m.refractive_index;
            V3 dir;
            if (refract(ray.direction, s.normal, eta, &dir)) {
                Ray r{RAY_SECONDARY, depth + 1, s.hit_point + dir * BIAS, dir, m.refractive_index};
                bool occ;
                refracted = get_illumination(r, path_mix(path, SLOT_REFRACTION), &occ);
                has_refr = true;
            }
        }
        V3 emissive = v3(m.emissive[0], m.emissive[1], m.emissive[2]);
        if (has_refl) emissive = emissive + reflected;
        if (has_refr) emissive = emissive + refracted * (1.0 - m.opacity);
        return emissive;
    }

    // :339-489
    ColorData get_color_physical(const Ray &ray, const Surface &s, const rt_material &m, uint64_t path) {
        uint32_t depth = ray.depth;
        V3 view_dir = normalize(-ray.direction);
        double n_dot_v = std::fmax(dot(s.normal, view_dir), 0.0);
        V3 mc = material_color(sc, m, s);
        double roughness = std::fmax(m.roughness, 0.04);
        V3 base = vlerp(v3(0.04, 0.04, 0.04), mc, m.metalness);
        V3 f = fresnel(n_dot_v, base);
        V3 k_d = (v3(1, 1, 1) - f) * (1.0 - m.metalness);

        bool has_refl = false, has_refr = false;
        V3 reflection = v3(0, 0, 0);
        ColorData refraction;
        if (sc.opts.max_reflected_rays > 0) {
            uint32_t n = reflected_rays(depth);
            double max_angle = FRAC_PI_2 * m.roughness;
            V3 rdir = reflect(ray.direction, s.normal);
            for (uint32_t i = 0; i < n; ++i) {
                uint64_t child = path_mix(path, i);
                double u1, u2;
                uniforms2(rng, child, &u1, &u2);
                V3 dir = uniform_sample_cone(rdir, max_angle, u1, u2);
                Ray r{RAY_SECONDARY, depth + 1, s.hit_point + dir * BIAS, dir, 1.0};
                ColorData cd = get_color(r, child);
                reflection = reflection + clamp01(cd.color);
            }
            reflection = reflection * (FRAC_PI_2 / static_cast<double>(n));
            has_refl = true;
        }
        if (m.opacity < 1.0) {
            double eta = ray.refractive_index / m.refractive_index;
            V3 dir;
            if (refract(ray.direction, s.normal, eta, &dir)) {
                Ray r{RAY_SECONDARY, depth + 1, s.hit_point + dir * BIAS, dir, m.refractive_index};
                refraction = get_color(r, path_mix(path, SLOT_REFRACTION));
                refraction.color = cmul(refraction.color, vlerp(v3(1, 1, 1), f, m.opacity));
                has_refr = true;
            }
        }
        V3 ambient = v3(0, 0, 0), irradiance = v3(0, 0, 0);
        V3 diffuse = FRAC_1_PI * cmul(k_d, mc);
        for (const rt_light &l : sc.lights) {
            if (l.kind == RT_LIGHT_AMBIENT) {
                ambient = ambient + cmul(v3(l.color[0], l.color[1], l.color[2]), mc);
                continue;
            }
            V3 lp = v3(l.position[0], l.position[1], l.position[2]);
            V3 ld = lp - s.hit_point;
            double ldist = norm(ld);
            ld = normalize(ld);
            double n_dot_l = dot(s.normal, ld);
            if (n_dot_l > 0.0) {
                Ray sr{RAY_SHADOW, 0, lp, -ld, 1.0};
                ray_count += 1;
                if (!tr.shadow_cast(sr, ldist - BIAS)) {
                    V3 half_vec = normalize(ld - ray.direction);
                    double n_dot_h = std::fmax(dot(s.normal, half_vec), 0.0);
                    V3 lc = light_color(l, ldist);
                    V3 radiance = lc * n_dot_l;
                    double d = ndf(n_dot_h, roughness);
                    double g = geometry_function(n_dot_v, n_dot_l, roughness);
                    V3 ds;
                    if (n_dot_v == 0.0) ds = diffuse;
                    else ds = diffuse + (d * g * f) / (4.0 * n_dot_v * n_dot_l);
                    irradiance = irradiance + cmul(ds, radiance) * n_dot_l;
                }
            }
        }
        V3 gi;
        double ao;
        compute_global_illumination(s, depth, path, &gi, &ao);
        ColorData cd;
        cd.color = v3(m.emissive[0], m.emissive[1], m.emissive[2]) + (ambient + irradiance + cmul(gi, diffuse)) * ao;
        cd.albedo = mc;
        cd.normal = s.normal;
        if (has_refl)
            cd.color = v3(scalar_lerp(cd.color.x, reflection.x, f.x), scalar_lerp(cd.color.y, reflection.y, f.y),
                          scalar_lerp(cd.color.z, reflection.z, f.z));
        if (has_refr) {
            cd.color = vlerp(clamp01(refraction.color), cd.color, m.opacity);
            cd.normal = refraction.normal;
        }
        return cd;
    }

    // :492-520
    V3 get_illumination(const Ray &ray, uint64_t path, bool *occluded) {
        *occluded = false;
        if (ray.depth >= sc.opts.max_depth) return v3(0, 0, 0);
        ray_count += 1;
        Hit hit;
        if (!tr.raycast(ray, &hit)) return v3(0, 0, 0);
        Surface s = compute_data(hit, ray);
        const rt_material &m = *hit.prim->material;
        V3 e = m.kind == RT_MAT_PHONG ? get_emissive_phong(ray, s, m, path) : get_emissive_physical(ray, s, m, path);
        *occluded = hit.distance <= sc.opts.max_occlusion_distance;
        return e;
    }

    // :523-547
    ColorData get_color(const Ray &ray, uint64_t path) {
        ColorData black;
        if (ray.depth >= sc.opts.max_depth) return black;
        ray_count += 1;
        Hit hit;
        if (!tr.raycast(ray, &hit)) return black;
        Surface s = compute_data(hit, ray);
        const rt_material &m = *hit.prim->material;
        ColorData cd = m.kind == RT_MAT_PHONG ? get_color_phong(ray, s, m, path) : get_color_physical(ray, s, m, path);
        cd.color = clamp01(cd.color);
        cd.albedo = clamp01(cd.albedo);
        return cd;
    }

    // :549-596 for one sample
    Ray camera_ray(uint32_t x, uint32_t y, uint32_t sample) const {
        double width = static_cast<double>(sc.opts.width), height = static_cast<double>(sc.opts.height);
        double aspect = width / height;
        double fov = std::tan(sc.camera.fov_deg * (PI / 180.0) / 2.0);
        double px, py;
        if (sample == 0) {
            px = static_cast<double>(x) + 0.5;
            py = static_cast<double>(y) + 0.5;
        } else {
            RngCtx c{rng.seed, y * sc.opts.width + x, sample};
            double rx, ry;
            uniforms2(c, 0, &rx, &ry);
            px = static_cast<double>(x) + rx;
            py = static_cast<double>(y) + ry;
        }
        double nx = remap_value(px, 0.0, width, -1.0, 1.0);
        double ny = remap_value(py, 0.0, height, 1.0, -1.0);
        if (width < height) nx = nx * aspect;
        else ny = ny / aspect;
        nx = nx * fov;
        ny = ny * fov;
        V3 d = normalize(v3(nx, ny, -1.0));
        const double *m = sc.camera.cam_to_world;
        V3 wd = v3(m[0] * d.x + m[1] * d.y + m[2] * d.z, m[3] * d.x + m[4] * d.y + m[5] * d.z,
                   m[6] * d.x + m[7] * d.y + m[8] * d.z);
        return Ray{RAY_PRIMARY, 0, v3(sc.camera.position[0], sc.camera.position[1], sc.camera.position[2]), wd, 1.0};
    }

    // Sum over samples [s0,s1) of the clamped colour (:605-612, before the 1/spp); optionally the other two members
    // of ColorData as screen_raycast carries them: the sum of the clamped albedos (:610) and the normal of sample 0
    // (:603-605: the pixel's ColorData starts as the first ray's).
    V3 pixel_sum(uint32_t x, uint32_t y, uint32_t s0, uint32_t s1, V3 *albedo_sum = nullptr, V3 *normal0 = nullptr) {
        V3 sum = v3(0, 0, 0), alb = v3(0, 0, 0);
        for (uint32_t s = s0; s < s1; ++s) {
            rng.pixel = y * sc.opts.width + x;
            rng.sample = s;
            ColorData cd = get_color(camera_ray(x, y, s), 0);
            sum = s == s0 ? cd.color : sum + cd.color;
            alb = s == s0 ? cd.albedo : alb + cd.albedo;
            if (s == 0 && normal0) *normal0 = cd.normal;
        }
        if (albedo_sum) *albedo_sum = alb;
        return sum;
    }
};

// :614-621 + :757-769
inline void tonemap_pixel(V3 sum, uint32_t spp, uint8_t *rgba) {
    V3 c = sum;
    if (spp != 1) c = clamp01(c * (1.0 / static_cast<double>(spp)));
    c = clamp01(c);  // gamma_correct() -> compute_color()
    c = v3(std::pow(c.x, 1.0 / GAMMA), std::pow(c.y, 1.0 / GAMMA), std::pow(c.z, 1.0 / GAMMA));
    rgba[0] = sat_u8(c.x * 255.0);
    rgba[1] = sat_u8(c.y * 255.0);
    rgba[2] = sat_u8(c.z * 255.0);
    rgba[3] = 255;
}

// Rotation3::rotation_between(normal, +y) (plane.rs:138), row-major 3x3.
bool rotation_between_y(V3 a, double out[9]) {
    double na = norm(a);
    if (!(na > 0.0)) {  // try_normalize failed -> identity
        const double id[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        std::memcpy(out, id, sizeof(id));
        return true;
    }
    V3 an = a / na;
    V3 bn = v3(0, 1, 0);
    V3 c = cross(an, bn);
    double cn = norm(c);
    if (cn > EPS) {
        V3 ax = c / cn;
        double angle = std::acos(dot(an, bn));
        double s = std::sin(angle), co = std::cos(angle), omc = 1.0 - co;
        double sqx = ax.x * ax.x, sqy = ax.y * ax.y, sqz = ax.z * ax.z;
        if (angle == 0.0) {
            const double id[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
            std::memcpy(out, id, sizeof(id));
            return true;
        }
        out[0] = sqx + (1.0 - sqx) * co;
        out[1] = ax.x * ax.y * omc - ax.z * s;
        out[2] = ax.x * ax.z * omc + ax.y * s;
        out[3] = ax.x * ax.y * omc + ax.z * s;
        out[4] = sqy + (1.0 - sqy) * co;
        out[5] = ax.y * ax.z * omc - ax.x * s;
        out[6] = ax.x * ax.z * omc - ax.y * s;
        out[7] = ax.y * ax.z * omc + ax.x * s;
        out[8] = sqz + (1.0 - sqz) * co;
        return true;
    }
    if (dot(an, bn) < 0.0) return false;  // PI rotation: None -> unwrap() panics in the reference
    const double id[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    std::memcpy(out, id, sizeof(id));
    return true;
}

}  // namespace

struct orc_scene {
    Scene sc;
};

extern "C" {

const char *orc_last_error(void) { return g_error.c_str(); }

int orc_scene_create(const rt_scene_desc *d, orc_scene **out) {
    if (!d || !out) {
        g_error = "null argument";
        return RT_ERR_INVALID;
    }
    *out = nullptr;
    if (d->opts.max_depth > 6) {
        g_error = "max_depth > 6 overflows 8_u16.pow(depth) in the reference";
        return RT_ERR_UNSUPPORTED;
    }
    auto holder = std::make_unique<orc_scene>();
    Scene &sc = holder->sc;
    sc.opts = d->opts;
    sc.camera = d->camera;
    sc.materials.assign(d->materials, d->materials + d->n_materials);
    sc.lights.assign(d->lights, d->lights + d->n_lights);
    for (uint32_t i = 0; i < d->n_textures; ++i) {
        const rt_texture &t = d->textures[i];
        sc.texture_data.emplace_back(t.rgb8, t.rgb8 + static_cast<size_t>(t.width) * t.height * 3);
        sc.textures.push_back(rt_texture{t.width, t.height, nullptr});
    }
    for (size_t i = 0; i < sc.textures.size(); ++i) sc.textures[i].rgb8 = sc.texture_data[i].data();
    sc.prims.resize(d->n_prims);
    for (uint32_t i = 0; i < d->n_prims; ++i) {
        Prim &p = sc.prims[i];
        p.type = d->prim_type[i];
        p.id = i;
        if (d->prim_material[i] >= d->n_materials || d->prim_transform[i] >= d->n_transforms) {
            g_error = "primitive references out of range";
            return RT_ERR_INVALID;
        }
        p.material = &sc.materials[d->prim_material[i]];
        std::memcpy(p.world.m, d->world + 12 * static_cast<size_t>(d->prim_transform[i]), sizeof(double) * 12);
        std::memcpy(p.inv.m, d->inv_world + 12 * static_cast<size_t>(d->prim_transform[i]), sizeof(double) * 12);
        p.param = d->prim_param[i];
        p.plane_normal = v3(d->prim_normal[3 * i], d->prim_normal[3 * i + 1], d->prim_normal[3 * i + 2]);
        p.plane_rot_ok = true;
        if (p.type == RT_PRIM_PLANE) p.plane_rot_ok = rotation_between_y(p.plane_normal, p.plane_rot);
        if (p.type == RT_PRIM_TRIANGLE) {
            uint32_t t = d->prim_tri[i];
            if (t >= d->n_tris) {
                g_error = "triangle index out of range";
                return RT_ERR_INVALID;
            }
            for (int k = 0; k < 3; ++k) {
                p.p[k] = v3(d->tri_pos[9 * t + 3 * k], d->tri_pos[9 * t + 3 * k + 1], d->tri_pos[9 * t + 3 * k + 2]);
                p.n[k] = v3(d->tri_nrm[9 * t + 3 * k], d->tri_nrm[9 * t + 3 * k + 1], d->tri_nrm[9 * t + 3 * k + 2]);
                p.uv[k][0] = d->tri_uv[6 * t + 2 * k];
                p.uv[k][1] = d->tri_uv[6 * t + 2 * k + 1];
            }
        }
        // into_bounded_object: sphere.rs:133-141, cube.rs:177-186, triangle.rs:235-251, plane.rs:120-122
        if (p.type == RT_PRIM_PLANE) {
            sc.unbounded.push_back(i);
            continue;
        }
        V3 mn, mx;
        if (p.type == RT_PRIM_SPHERE) {
            mn = v3(-p.param, -p.param, -p.param);
            mx = v3(p.param, p.param, p.param);
        } else if (p.type == RT_PRIM_CUBE) {
            double h = p.param / 2.0;
            mn = v3(-h, -h, -h);
            mx = v3(h, h, h);
        } else {
            mn = p.p[0];
            mx = mn;
            for (int k = 1; k < 3; ++k) {
                mn = v3(std::fmin(mn.x, p.p[k].x), std::fmin(mn.y, p.p[k].y), std::fmin(mn.z, p.p[k].z));
                mx = v3(std::fmax(mx.x, p.p[k].x), std::fmax(mx.y, p.p[k].y), std::fmax(mx.z, p.p[k].z));
            }
        }
        sc.bounded.push_back(i);
        sc.bounded_bv.push_back(bounds_from_transform(mn, mx, p.world));
    }
    if (!sc.bounded.empty()) {
        // KdTreeAccelerator::new, bounds.rs:265-289
        std::vector<uint32_t> indexes(sc.bounded.size());
        for (uint32_t i = 0; i < indexes.size(); ++i) indexes[i] = i;
        int max_depth = static_cast<int>(static_cast<uint8_t>(8.0 + 1.3 * std::log2(static_cast<double>(sc.bounded.size()))));
        Bounds root = sc.bounded_bv[0];
        for (size_t i = 1; i < sc.bounded_bv.size(); ++i) {  // BoundingVolume::merge, bounds.rs:74-86
            const Bounds &b = sc.bounded_bv[i];
            root = bounds_from(v3(std::fmin(root.bmin.x, b.bmin.x), std::fmin(root.bmin.y, b.bmin.y), std::fmin(root.bmin.z, b.bmin.z)),
                               v3(std::fmax(root.bmax.x, b.bmax.x), std::fmax(root.bmax.y, b.bmax.y), std::fmax(root.bmax.z, b.bmax.z)));
        }
        sc.kd = kd_build(sc, max_depth, 3, root, std::move(indexes), 0);
        sc.bvh_order.resize(sc.bounded.size());
        for (uint32_t i = 0; i < sc.bvh_order.size(); ++i) sc.bvh_order[i] = i;
        bvh_build(sc, 0, static_cast<uint32_t>(sc.bounded.size()));
        double scale = 0.0;
        for (int a = 0; a < 3; ++a) scale = std::fmax(scale, std::fmax(std::fabs(root.bmin[a]), std::fabs(root.bmax[a])));
        sc.pad = 1e-9 * (1.0 + scale);
    }
    *out = holder.release();
    return RT_OK;
}

void orc_scene_destroy(orc_scene *s) { delete s; }

// kd-tree facts: inner nodes, leaves, primitive references, depth, bounded, unbounded
int orc_scene_info(const orc_scene *s, uint64_t *out6) {
    if (!s || !out6) return RT_ERR_INVALID;
    out6[0] = s->sc.kd_nodes;
    out6[1] = s->sc.kd_leaves;
    out6[2] = s->sc.kd_refs;
    out6[3] = static_cast<uint64_t>(s->sc.kd_depth);
    out6[4] = s->sc.bounded.size();
    out6[5] = s->sc.unbounded.size();
    return RT_OK;
}

// Same contract as rt_render (mi355rt.h) with host buffers; rgb_accum is
// += accumulated, rgba8 is produced from rgb_accum (or the call's own sums when
// rgb_accum is NULL) for the pixels of the window.
int orc_render_aov(const orc_scene *s, int mode, const rt_render_params *params, int n_threads, double *rgb_accum,
                   double *albedo_accum, double *normal, uint8_t *rgba8, uint64_t *ray_count);

int orc_render(const orc_scene *s, int mode, const rt_render_params *params, int n_threads, double *rgb_accum,
               uint8_t *rgba8, uint64_t *ray_count) {
    return orc_render_aov(s, mode, params, n_threads, rgb_accum, nullptr, nullptr, rgba8, ray_count);
}

// As orc_render, plus the AOVs of ColorData (render/mod.rs:16-20): albedo_accum[w*h*3] += sum over the samples of the
// clamped albedo; normal[w*h*3] = normal of sample 0 (written only when the sample range contains sample 0).
int orc_render_aov(const orc_scene *s, int mode, const rt_render_params *params, int n_threads, double *rgb_accum,
                   double *albedo_accum, double *normal, uint8_t *rgba8, uint64_t *ray_count) {
    if (!s || !params) {
        g_error = "null argument";
        return RT_ERR_INVALID;
    }
    const Scene &sc = s->sc;
    uint32_t w = sc.opts.width, h = sc.opts.height, spp = sc.opts.samples_per_pixel;
    uint32_t x0 = params->x0, y0 = params->y0, x1 = params->x1, y1 = params->y1, s0 = params->s0, s1 = params->s1;
    if (x1 == 0 && y1 == 0) {
        x0 = y0 = 0;
        x1 = w;
        y1 = h;
    }
    if (s1 == 0) {
        s0 = 0;
        s1 = spp;
    }
    if (x1 > w || y1 > h || x0 > x1 || y0 > y1 || s0 > s1 || s1 > spp) {
        g_error = "render window out of range";
        return RT_ERR_INVALID;
    }
    if (n_threads < 1) n_threads = 1;
    std::vector<double> local;
    double *acc = rgb_accum;
    if (!acc) {
        local.assign(static_cast<size_t>(w) * h * 3, 0.0);
        acc = local.data();
    }
    std::atomic<uint32_t> next_row{y0};
    std::atomic<uint64_t> total{0};
    auto worker = [&]() {
        Integrator integ(sc, mode, params->seed);
        while (true) {
            uint32_t y = next_row.fetch_add(1);
            if (y >= y1) break;
            for (uint32_t x = x0; x < x1; ++x) {
                V3 alb, nrm = v3(0, 0, 1);
                V3 sum = integ.pixel_sum(x, y, s0, s1, &alb, &nrm);
                size_t o = (static_cast<size_t>(y) * w + x) * 3;
                acc[o] += sum.x;
                acc[o + 1] += sum.y;
                acc[o + 2] += sum.z;
                if (albedo_accum) {
                    albedo_accum[o] += alb.x;
                    albedo_accum[o + 1] += alb.y;
                    albedo_accum[o + 2] += alb.z;
                }
                if (normal && s0 == 0) {
                    normal[o] = nrm.x;
                    normal[o + 1] = nrm.y;
                    normal[o + 2] = nrm.z;
                }
            }
        }
        total += integ.ray_count;
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < n_threads; ++t) pool.emplace_back(worker);
    worker();
    for (auto &t : pool) t.join();
    if (rgba8)
        for (uint32_t y = y0; y < y1; ++y)
            for (uint32_t x = x0; x < x1; ++x) {
                size_t p = static_cast<size_t>(y) * w + x;
                tonemap_pixel(v3(acc[p * 3], acc[p * 3 + 1], acc[p * 3 + 2]), spp, rgba8 + p * 4);
            }
    if (ray_count) *ray_count += total.load();
    return RT_OK;
}

// Pixel-centre primary-hit AOVs (same contract as rt_query_primary).
int orc_query_primary(const orc_scene *s, int mode, uint32_t *prim_id, double *t, double *normal3) {
    if (!s) return RT_ERR_INVALID;
    const Scene &sc = s->sc;
    Integrator integ(sc, mode, 0);
    for (uint32_t y = 0; y < sc.opts.height; ++y)
        for (uint32_t x = 0; x < sc.opts.width; ++x) {
            size_t p = static_cast<size_t>(y) * sc.opts.width + x;
            Ray r = integ.camera_ray(x, y, 0);
            Hit hit;
            if (integ.tr.raycast(r, &hit)) {
                Surface sf = compute_data(hit, r);
                if (prim_id) prim_id[p] = hit.prim->id;
                if (t) t[p] = hit.distance;
                if (normal3) {
                    normal3[3 * p] = sf.normal.x;
                    normal3[3 * p + 1] = sf.normal.y;
                    normal3[3 * p + 2] = sf.normal.z;
                }
            } else {
                if (prim_id) prim_id[p] = RT_NO_HIT;
                if (t) t[p] = 0.0;
                if (normal3) normal3[3 * p] = normal3[3 * p + 1] = normal3[3 * p + 2] = 0.0;
            }
        }
    return RT_OK;
}

// Same contract as rt_trace_rays.
int orc_trace_rays(const orc_scene *s, int mode, uint32_t n, const double *origin, const double *dir, const double *tmax,
                   int shadow, uint32_t *prim_id, double *t) {
    if (!s) return RT_ERR_INVALID;
    Tracer tr(s->sc, mode);
    for (uint32_t i = 0; i < n; ++i) {
        Ray r{shadow ? RAY_SHADOW : RAY_SECONDARY, 1, v3(origin[3 * i], origin[3 * i + 1], origin[3 * i + 2]),
              v3(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]), 1.0};
        if (shadow) {
            prim_id[i] = tr.shadow_cast(r, tmax[i]) ? 1u : 0u;
            if (t) t[i] = 0.0;
        } else {
            Hit hit;
            if (tr.raycast(r, &hit)) {
                prim_id[i] = hit.prim->id;
                if (t) t[i] = hit.distance;
            } else {
                prim_id[i] = RT_NO_HIT;
                if (t) t[i] = 0.0;
            }
        }
    }
    return RT_OK;
}

// ---- known-answer probes for the reference's unit-test vectors --------------
int orc_quadratic(double a, double b, double c, double *t0, double *t1) { return quadratic(a, b, c, t0, t1) ? 1 : 0; }
double orc_remap_value(double num, double d0, double d1, double r0, double r1) { return remap_value(num, d0, d1, r0, r1); }
uint32_t orc_to_argb_u32(double r, double g, double b) {  // utils/mod.rs:14-19
    return (255u << 24) | (sat_u32(r * 255.0) << 16) | (sat_u32(g * 255.0) << 8) | sat_u32(b * 255.0);
}
void orc_philox4x32_10(const uint32_t *ctr4, const uint32_t *key2, uint32_t *out4) { philox4x32_10(ctr4, key2, out4); }
uint64_t orc_path_mix(uint64_t path, uint32_t slot) { return path_mix(path, slot); }
void orc_uniforms2(uint64_t seed, uint32_t pixel, uint32_t sample, uint64_t path, double *u2) {
    RngCtx c{seed, pixel, sample};
    uniforms2(c, path, &u2[0], &u2[1]);
}
void orc_cosine_sample_hemisphere(const double *dir3, double u1, double u2, double *out3) {
    V3 r = cosine_sample_hemisphere(v3(dir3[0], dir3[1], dir3[2]), u1, u2);
    out3[0] = r.x;
    out3[1] = r.y;
    out3[2] = r.z;
}
void orc_uniform_sample_cone(const double *dir3, double max_angle, double u1, double u2, double *out3) {
    V3 r = uniform_sample_cone(v3(dir3[0], dir3[1], dir3[2]), max_angle, u1, u2);
    out3[0] = r.x;
    out3[1] = r.y;
    out3[2] = r.z;
}
void orc_reflect(const double *i3, const double *n3, double *out3) {
    V3 r = reflect(v3(i3[0], i3[1], i3[2]), v3(n3[0], n3[1], n3[2]));
    out3[0] = r.x;
    out3[1] = r.y;
    out3[2] = r.z;
}
int orc_refract(const double *i3, const double *n3, double eta, double *out3) {
    V3 r;
    if (!refract(v3(i3[0], i3[1], i3[2]), v3(n3[0], n3[1], n3[2]), eta, &r)) return 0;
    out3[0] = r.x;
    out3[1] = r.y;
    out3[2] = r.z;
    return 1;
}
void orc_tonemap(const double *sum3, uint32_t spp, uint8_t *rgba4) { tonemap_pixel(v3(sum3[0], sum3[1], sum3[2]), spp, rgba4); }

}  // extern "C"
