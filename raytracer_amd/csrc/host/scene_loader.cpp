// Host-side scene front end: scene JSON + OBJ + decoded textures -> rt_scene_desc.
//
// Plays the role of the reference's Rust host code, which cannot be built in
// this image (no cargo/rustc).  Behaviour follows:
//   src/render/scene.rs:10-97        Scene, load_assets, from_scene (flatten order)
//   src/render/mod.rs:113-159        Camera / RenderOptions defaults
//   src/core/transform.rs:73-138     translate/rotate/scale LEFT-multiply; ordered list
//   src/core/material.rs:22-113      Phong / Physical defaults, default = black phong
//   src/lights/{ambient,point}.rs    light defaults
//   src/primitives/*.rs              object schema, children-before-self flatten
//   src/primitives/mesh.rs:57-151    OBJ -> one Triangle per (fan-triangulated) face
// nalgebra conventions restated (crate not available here, pinned 0.30.1 in
// Cargo.lock:643): Rotation3::from_axis_angle is Rodrigues' formula applied to
// the axis AS GIVEN (serde does not normalise Unit<Vector3>), 4x4 products
// accumulate k = 0..3 left to right, look_at_rh = face_towards(-dir, up)^-1.
#include "../../../include/mi355rt_host.h"
#include "json.hpp"

namespace rth {
bool decode_image(const std::vector<uint8_t> &file, uint32_t *w, uint32_t *h, std::vector<uint8_t> *rgb, std::string *err);
}

#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iterator>
#include <map>
#include <sstream>

namespace {

thread_local std::string g_error;

struct Mat4 {
    double m[16];  // row-major
};

Mat4 mat_identity() {
    Mat4 r{};
    r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.0;
    return r;
}

Mat4 mat_mul(const Mat4 &a, const Mat4 &b) {
    Mat4 c;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            double acc = a.m[i * 4 + 0] * b.m[0 * 4 + j];
            acc = acc + a.m[i * 4 + 1] * b.m[1 * 4 + j];
            acc = acc + a.m[i * 4 + 2] * b.m[2 * 4 + j];
            acc = acc + a.m[i * 4 + 3] * b.m[3 * 4 + j];
            c.m[i * 4 + j] = acc;
        }
    return c;
}

Mat4 mat_transpose(const Mat4 &a) {
    Mat4 t;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) t.m[i * 4 + j] = a.m[j * 4 + i];
    return t;
}

// General 4x4 inverse by cofactors (Affine3::inverse -> Matrix4::try_inverse).
bool mat_inverse(const Mat4 &a, Mat4 *out) {
    const double *m = a.m;
    double inv[16];
    inv[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] +
             m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
    inv[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] -
             m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
    inv[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] +
             m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
    inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] -
              m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
    inv[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] -
             m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
    inv[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] +
             m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
    inv[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] -
             m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
    inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] +
              m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
    inv[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] +
             m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
    inv[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] -
             m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
    inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] +
              m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
    inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] -
              m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
    inv[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] -
             m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
    inv[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] +
             m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
    inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] -
              m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
    inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] +
              m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
    double det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
    if (det == 0.0) return false;
    double inv_det = 1.0 / det;
    for (int i = 0; i < 16; ++i) out->m[i] = inv[i] * inv_det;
    return true;
}

Mat4 mat_translation(const double t[3]) {
    Mat4 r = mat_identity();
    r.m[3] = t[0];
    r.m[7] = t[1];
    r.m[11] = t[2];
    return r;
}

Mat4 mat_scaling(const double s[3]) {
    Mat4 r = mat_identity();
    r.m[0] = s[0];
    r.m[5] = s[1];
    r.m[10] = s[2];
    return r;
}

// Rotation3::from_axis_angle with the axis taken as is (quirk: serde yields an
// un-normalised Unit<Vector3>; verified against renders/test.png).
Mat4 mat_axis_angle(const double axis[3], double angle_rad) {
    Mat4 r = mat_identity();
    if (angle_rad == 0.0) return r;
    double ux = axis[0], uy = axis[1], uz = axis[2];
    double sqx = ux * ux, sqy = uy * uy, sqz = uz * uz;
    double s = std::sin(angle_rad), c = std::cos(angle_rad);
    double omc = 1.0 - c;
    r.m[0] = sqx + (1.0 - sqx) * c;
    r.m[1] = ux * uy * omc - uz * s;
    r.m[2] = ux * uz * omc + uy * s;
    r.m[4] = ux * uy * omc + uz * s;
    r.m[5] = sqy + (1.0 - sqy) * c;
    r.m[6] = uy * uz * omc - ux * s;
    r.m[8] = ux * uz * omc - uy * s;
    r.m[9] = uy * uz * omc + ux * s;
    r.m[10] = sqz + (1.0 - sqz) * c;
    return r;
}

double to_radians(double deg) { return deg * (3.14159265358979323846 / 180.0); }

void vec_normalize(double v[3]) {
    double n = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    v[0] = v[0] / n;
    v[1] = v[1] / n;
    v[2] = v[2] / n;
}
void vec_cross(const double a[3], const double b[3], double out[3]) {
    out[0] = a[1] * b[2] - a[2] * b[1];
    out[1] = a[2] * b[0] - a[0] * b[2];
    out[2] = a[0] * b[1] - a[1] * b[0];
}

using rth::JsonValue;

[[noreturn]] void fail(const std::string &msg) { throw std::runtime_error(msg); }

double as_num(const JsonValue &v, const char *what) {
    if (!v.is_number()) fail(std::string("expected a number for ") + what);
    return v.num;
}
void as_vec3(const JsonValue &v, const char *what, double out[3]) {
    if (!v.is_array() || v.arr.size() != 3) fail(std::string("expected [x,y,z] for ") + what);
    for (int i = 0; i < 3; ++i) out[i] = as_num(v.arr[i], what);
}
uint32_t as_uint(const JsonValue &v, const char *what, double max) {
    double d = as_num(v, what);
    if (d < 0 || d > max || d != std::floor(d)) fail(std::string("value out of range for ") + what);
    return static_cast<uint32_t>(d);
}

// src/core/transform.rs:112-134: each entry is applied AFTER the previous ones.
Mat4 parse_transform(const JsonValue &v) {
    if (!v.is_array()) fail("transform must be a sequence");
    Mat4 m = mat_identity();
    for (const JsonValue &e : v.arr) {
        if (!e.is_object() || e.obj.size() != 1) fail("transform entries must be single-key objects");
        const std::string &k = e.obj[0].first;
        const JsonValue &a = e.obj[0].second;
        if (k == "translate") {
            double t[3];
            as_vec3(a, "translate", t);
            m = mat_mul(mat_translation(t), m);
        } else if (k == "rotate") {
            if (!a.is_array() || a.arr.size() != 2) fail("rotate expects [[axis], degrees]");
            double axis[3];
            as_vec3(a.arr[0], "rotate axis", axis);
            double deg = as_num(a.arr[1], "rotate angle");
            m = mat_mul(mat_axis_angle(axis, to_radians(deg)), m);
        } else if (k == "scale") {
            double s[3];
            as_vec3(a, "scale", s);
            m = mat_mul(mat_scaling(s), m);
        } else {
            fail("unknown variant `" + k + "`, expected one of `translate`, `rotate`, `scale`");
        }
    }
    return m;
}

struct Node {
    std::string type;
    double param = 1.0;              // radius / size
    double normal[3] = {0, 1, 0};    // plane
    bool tri_pnt = false;            // triangle given with normals/texcoords
    double tri_pos[9] = {0}, tri_nrm[9] = {0}, tri_uv[6] = {0};
    std::string file;                // mesh
    Mat4 local = mat_identity();
    int material = -1;               // index into Loader::materials
    std::vector<Node> children;
};

struct Loader {
    std::string asset_base;
    rt_render_options opts{100, 100, 3, 4, 32, 16, 1.0};
    bool skip_denoise_pass = false;
    double cam_fov = 65.0, cam_pos[3] = {0, 0, 1}, cam_target[3] = {0, 0, 0}, cam_up[3] = {0, 1, 0};
    std::vector<rt_light> lights;
    std::vector<rt_material> materials;
    std::vector<std::string> material_texture;  // path per material ("" = none)
    std::vector<Node> roots;

    // flattened output
    std::vector<uint8_t> prim_type;
    std::vector<uint32_t> prim_material, prim_transform, prim_tri;
    std::vector<double> prim_param, prim_normal;
    std::vector<double> world, inv_world;
    std::vector<double> tri_pos, tri_nrm, tri_uv;
    std::vector<std::string> texture_paths;
    std::vector<std::vector<uint8_t>> texture_data;
    std::vector<rt_texture> textures;
    rt_scene_desc desc{};

    static rt_material default_material() {
        rt_material m{};
        m.kind = RT_MAT_PHONG;
        m.side = RT_SIDE_FRONT;
        m.reflectivity = 0.0;
        m.shininess = 30.0;
        m.opacity = 1.0;
        m.roughness = 0.5;
        m.metalness = 0.0;
        m.refractive_index = 1.0;
        m.texture_id = -1;
        return m;
    }

    int parse_material(const JsonValue *v) {
        rt_material m = default_material();
        std::string tex;
        if (v) {
            if (!v->is_object()) fail("material must be an object");
            const JsonValue *t = v->find("type");
            if (!t || !t->is_string()) fail("missing field `type` in material");
            if (t->str == "phong") {
                m.kind = RT_MAT_PHONG;
            } else if (t->str == "physical") {
                m.kind = RT_MAT_PHYSICAL;
            } else {
                fail("unknown variant `" + t->str + "`, expected `phong` or `physical`");
            }
            for (const auto &kv : v->obj) {
                const std::string &k = kv.first;
                const JsonValue &a = kv.second;
                if (k == "type") continue;
                if (k == "side") {
                    if (!a.is_string()) fail("side must be a string");
                    if (a.str == "Both") m.side = RT_SIDE_BOTH;
                    else if (a.str == "Front") m.side = RT_SIDE_FRONT;
                    else if (a.str == "Back") m.side = RT_SIDE_BACK;
                    else fail("unknown variant `" + a.str + "`, expected one of `Both`, `Front`, `Back`");
                } else if (k == "color") {
                    as_vec3(a, "color", m.color);
                } else if (k == "emissive") {
                    as_vec3(a, "emissive", m.emissive);
                } else if (k == "texture") {
                    if (a.is_string()) tex = a.str;
                    else if (!a.is_null()) fail("texture must be a string");
                } else if (m.kind == RT_MAT_PHONG && k == "specular") {
                    as_vec3(a, "specular", m.specular);
                } else if (m.kind == RT_MAT_PHONG && k == "reflectivity") {
                    m.reflectivity = as_num(a, "reflectivity");
                } else if (m.kind == RT_MAT_PHONG && k == "shininess") {
                    m.shininess = as_num(a, "shininess");
                } else if (m.kind == RT_MAT_PHYSICAL && k == "opacity") {
                    m.opacity = as_num(a, "opacity");
                } else if (m.kind == RT_MAT_PHYSICAL && k == "emissive_intensity") {
                    (void)as_num(a, "emissive_intensity");  // parsed, never read (material.rs:67)
                } else if (m.kind == RT_MAT_PHYSICAL && k == "roughness") {
                    m.roughness = as_num(a, "roughness");
                } else if (m.kind == RT_MAT_PHYSICAL && k == "metalness") {
                    m.metalness = as_num(a, "metalness");
                } else if (m.kind == RT_MAT_PHYSICAL && k == "refractive_index") {
                    m.refractive_index = as_num(a, "refractive_index");
                } else {
                    fail("unknown field `" + k + "` in material");
                }
            }
        }
        materials.push_back(m);
        material_texture.push_back(tex);
        return static_cast<int>(materials.size()) - 1;
    }

    void parse_triangle_vertices(const JsonValue &a, Node *n) {
        if (!a.is_array() || a.arr.size() != 3) fail("triangle needs exactly 3 vertices");
        bool pnt = a.arr[0].is_object();
        n->tri_pnt = pnt;
        for (int i = 0; i < 3; ++i) {
            const JsonValue &v = a.arr[i];
            if (pnt) {
                if (!v.is_object()) fail("mixed triangle vertex formats");
                const JsonValue *p = v.find("position"), *nn = v.find("normal"), *t = v.find("texcoords");
                if (!p || !nn || !t) fail("vertex needs position, normal and texcoords");
                for (const auto &kv : v.obj)
                    if (kv.first != "position" && kv.first != "normal" && kv.first != "texcoords")
                        fail("unknown field `" + kv.first + "` in vertex");
                as_vec3(*p, "position", n->tri_pos + 3 * i);
                as_vec3(*nn, "normal", n->tri_nrm + 3 * i);  // Unit<>: not normalised by serde
                if (!t->is_array() || t->arr.size() != 2) fail("texcoords must be [u,v]");
                n->tri_uv[2 * i] = as_num(t->arr[0], "texcoords");
                n->tri_uv[2 * i + 1] = as_num(t->arr[1], "texcoords");
            } else {
                as_vec3(v, "vertex", n->tri_pos + 3 * i);
            }
        }
    }

    Node parse_object(const JsonValue &v) {
        if (!v.is_object()) fail("object must be a JSON object");
        const JsonValue *t = v.find("type");
        if (!t || !t->is_string()) fail("missing field `type` in object");
        Node n;
        n.type = t->str;
        const bool is_group = n.type == "group";
        if (n.type != "cube" && n.type != "sphere" && n.type != "plane" && n.type != "triangle" &&
            n.type != "mesh" && !is_group)
            fail("unknown variant `" + n.type +
                 "`, expected one of `cube`, `plane`, `sphere`, `triangle`, `mesh`, `group`");
        const JsonValue *mat = nullptr;
        bool have_children = false, have_file = false;
        for (const auto &kv : v.obj) {
            const std::string &k = kv.first;
            const JsonValue &a = kv.second;
            if (k == "type") continue;
            if (k == "transform") {
                n.local = parse_transform(a);
            } else if (k == "children") {
                if (a.is_null() && !is_group) continue;
                if (!a.is_array()) fail("children must be a sequence");
                have_children = true;
                for (const JsonValue &c : a.arr) n.children.push_back(parse_object(c));
            } else if (k == "material" && !is_group) {
                mat = &a;
            } else if (k == "size" && n.type == "cube") {
                n.param = as_num(a, "size");
            } else if (k == "radius" && n.type == "sphere") {
                n.param = as_num(a, "radius");
            } else if (k == "normal" && n.type == "plane") {
                as_vec3(a, "normal", n.normal);
            } else if ((k == "vertices" || k == "vertex_data") && n.type == "triangle") {
                parse_triangle_vertices(a, &n);
            } else if (k == "file" && n.type == "mesh") {
                if (!a.is_string()) fail("file must be a string");
                n.file = a.str;
                have_file = true;
            } else {
                fail("unknown field `" + k + "` in " + n.type);
            }
        }
        if (is_group && !have_children) fail("missing field `children` in group");
        if (n.type == "mesh" && !have_file) fail("missing field `file` in mesh");
        if (!is_group) n.material = parse_material(mat);
        if (n.type == "mesh") load_mesh(&n);
        return n;
    }

    // tobj::load_obj(triangulate, single_index, ignore_lines, ignore_points)
    // + Mesh::load_assets (mesh.rs:57-151).  Replaces any JSON children.
    void load_mesh(Node *mesh) {
        std::string path = asset_base.empty() ? mesh->file : asset_base + "/" + mesh->file;
        std::ifstream in(path);
        if (!in) fail("failed to load object at path \"" + path + "\": open failed");
        std::vector<float> vp, vt, vn;
        std::vector<Node> faces;
        std::string line;
        auto resolve = [](long idx, size_t count) -> long {
            if (idx > 0) return idx - 1;
            if (idx < 0) return static_cast<long>(count) + idx;
            return -1;
        };
        while (std::getline(in, line)) {
            if (!line.empty() && line.back() == '\r') line.pop_back();
            const char *s = line.c_str();
            while (*s == ' ' || *s == '\t') ++s;
            if (s[0] == 'v' && (s[1] == ' ' || s[1] == '\t')) {
                char *e = const_cast<char *>(s + 1);
                for (int i = 0; i < 3; ++i) vp.push_back(std::strtof(e, &e));
            } else if (s[0] == 'v' && s[1] == 't' && (s[2] == ' ' || s[2] == '\t')) {
                char *e = const_cast<char *>(s + 2);
                for (int i = 0; i < 2; ++i) vt.push_back(std::strtof(e, &e));
            } else if (s[0] == 'v' && s[1] == 'n' && (s[2] == ' ' || s[2] == '\t')) {
                char *e = const_cast<char *>(s + 2);
                for (int i = 0; i < 3; ++i) vn.push_back(std::strtof(e, &e));
            } else if (s[0] == 'f' && (s[1] == ' ' || s[1] == '\t')) {
                struct Corner {
                    long v, t, n;
                };
                std::vector<Corner> corners;
                const char *p = s + 1;
                while (*p) {
                    while (*p == ' ' || *p == '\t') ++p;
                    if (!*p) break;
                    Corner c{0, 0, 0};
                    char *e = nullptr;
                    c.v = std::strtol(p, &e, 10);
                    if (e == p) fail("bad face element in " + path);
                    p = e;
                    if (*p == '/') {
                        ++p;
                        if (*p != '/') {
                            c.t = std::strtol(p, &e, 10);
                            p = e;
                        }
                        if (*p == '/') {
                            ++p;
                            c.n = std::strtol(p, &e, 10);
                            p = e;
                        }
                    }
                    corners.push_back(c);
                }
                if (corners.size() < 3) continue;  // points / lines ignored
                for (size_t k = 2; k < corners.size(); ++k) {
                    const Corner tri[3] = {corners[0], corners[k - 1], corners[k]};
                    Node f;
                    f.type = "triangle";
                    f.material = mesh->material;  // material cloned per face (mesh.rs:143)
                    bool has_n = true, has_t = true;
                    for (int i = 0; i < 3; ++i) {
                        long vi = resolve(tri[i].v, vp.size() / 3);
                        if (vi < 0 || static_cast<size_t>(vi) * 3 + 2 >= vp.size())
                            fail("face index out of range in " + path);
                        for (int c = 0; c < 3; ++c) f.tri_pos[3 * i + c] = static_cast<double>(vp[vi * 3 + c]);
                        long ni = resolve(tri[i].n, vn.size() / 3);
                        if (tri[i].n == 0 || ni < 0 || static_cast<size_t>(ni) * 3 + 2 >= vn.size()) {
                            has_n = false;
                        } else {
                            for (int c = 0; c < 3; ++c) f.tri_nrm[3 * i + c] = static_cast<double>(vn[ni * 3 + c]);
                            vec_normalize(f.tri_nrm + 3 * i);  // Unit::new_normalize (mesh.rs:94)
                        }
                        long ti = resolve(tri[i].t, vt.size() / 2);
                        if (tri[i].t == 0 || ti < 0 || static_cast<size_t>(ti) * 2 + 1 >= vt.size()) {
                            has_t = false;
                        } else {
                            f.tri_uv[2 * i] = static_cast<double>(vt[ti * 2]);
                            f.tri_uv[2 * i + 1] = static_cast<double>(vt[ti * 2 + 1]);
                        }
                    }
                    f.tri_pnt = has_n;  // false -> flat normal computed at flatten time
                    if (!has_t)
                        for (double &u : f.tri_uv) u = 0.0;
                    faces.push_back(std::move(f));
                }
            }
        }
        mesh->children = std::move(faces);
    }

    int texture_id_for(const std::string &path) {
        for (size_t i = 0; i < texture_paths.size(); ++i)
            if (texture_paths[i] == path) return static_cast<int>(i);
        // Texture::load (texture.rs:35-44): image::open(asset_base.join(path))?.to_rgb8()
        const std::string full = asset_base.empty() ? path : asset_base + "/" + path;
        std::ifstream in(full, std::ios::binary);
        if (!in) fail("failed to load texture at path \"" + path + "\": cannot open " + full);
        std::vector<uint8_t> bytes((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
        uint32_t w = 0, h = 0;
        std::vector<uint8_t> data;
        std::string why;
        if (!rth::decode_image(bytes, &w, &h, &data, &why)) fail("failed to load texture at path \"" + path + "\": " + why);
        texture_paths.push_back(path);
        texture_data.push_back(std::move(data));
        rt_texture t{w, h, nullptr};
        textures.push_back(t);
        return static_cast<int>(texture_paths.size()) - 1;
    }

    uint32_t add_transform(const Mat4 &w) {
        Mat4 inv;
        if (!mat_inverse(w, &inv)) fail("singular object transform");
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 4; ++c) {
                world.push_back(w.m[r * 4 + c]);
                inv_world.push_back(inv.m[r * 4 + c]);
            }
        return static_cast<uint32_t>(world.size() / 12) - 1;
    }

    void emit(uint8_t type, const Node &n, uint32_t xf) {
        prim_type.push_back(type);
        prim_material.push_back(static_cast<uint32_t>(n.material));
        prim_transform.push_back(xf);
        prim_param.push_back(type == RT_PRIM_SPHERE || type == RT_PRIM_CUBE ? n.param : 0.0);
        for (int c = 0; c < 3; ++c) prim_normal.push_back(type == RT_PRIM_PLANE ? n.normal[c] : 0.0);
        if (type == RT_PRIM_TRIANGLE) {
            prim_tri.push_back(static_cast<uint32_t>(tri_pos.size() / 9));
            double nrm[9];
            if (n.tri_pnt) {
                for (int i = 0; i < 9; ++i) nrm[i] = n.tri_nrm[i];
            } else {  // Triangle::compute_normal (triangle.rs:100-105)
                double e1[3], e2[3], fn[3];
                for (int c = 0; c < 3; ++c) {
                    e1[c] = n.tri_pos[3 + c] - n.tri_pos[c];
                    e2[c] = n.tri_pos[6 + c] - n.tri_pos[c];
                }
                vec_cross(e1, e2, fn);
                vec_normalize(fn);
                for (int i = 0; i < 9; ++i) nrm[i] = fn[i % 3];
            }
            for (int i = 0; i < 9; ++i) tri_pos.push_back(n.tri_pos[i]);
            for (int i = 0; i < 9; ++i) tri_nrm.push_back(nrm[i]);
            for (int i = 0; i < 6; ++i) tri_uv.push_back(n.tri_uv[i]);
        } else {
            prim_tri.push_back(RT_NO_HIT);
        }
    }

    // flatten_to_world: world = parent * local; children first, then self.
    // `shared_xf` lets the faces of one mesh share the mesh's world transform
    // (their local transform is the identity, so parent * I == parent exactly).
    void flatten(const Node &n, const Mat4 &parent, int shared_xf) {
        Mat4 w = mat_mul(parent, n.local);
        if (n.type == "mesh") {
            uint32_t xf = n.children.empty() ? 0u : add_transform(w);
            for (const Node &c : n.children) flatten(c, w, static_cast<int>(xf));
            return;
        }
        for (const Node &c : n.children) flatten(c, w, -1);
        if (n.type == "group") return;
        uint32_t xf = shared_xf >= 0 ? static_cast<uint32_t>(shared_xf) : add_transform(w);
        if (n.type == "cube") emit(RT_PRIM_CUBE, n, xf);
        else if (n.type == "sphere") emit(RT_PRIM_SPHERE, n, xf);
        else if (n.type == "plane") emit(RT_PRIM_PLANE, n, xf);
        else if (n.type == "triangle") emit(RT_PRIM_TRIANGLE, n, xf);
    }

    void parse_scene(const JsonValue &root) {
        if (!root.is_object()) fail("scene must be a JSON object");
        for (const auto &kv : root.obj) {
            const std::string &k = kv.first;
            const JsonValue &a = kv.second;
            if (k == "width") opts.width = as_uint(a, "width", 4294967295.0);
            else if (k == "height") opts.height = as_uint(a, "height", 4294967295.0);
            else if (k == "max_depth") opts.max_depth = as_uint(a, "max_depth", 255);
            else if (k == "samples_per_pixel") opts.samples_per_pixel = as_uint(a, "samples_per_pixel", 65535);
            else if (k == "max_reflected_rays") opts.max_reflected_rays = as_uint(a, "max_reflected_rays", 65535);
            else if (k == "max_illumination_rays")
                opts.max_illumination_rays = as_uint(a, "max_illumination_rays", 65535);
            else if (k == "max_occlusion_distance") opts.max_occlusion_distance = as_num(a, "max_occlusion_distance");
            else if (k == "skip_denoise_pass") skip_denoise_pass = a.b;
            else if (k == "loaded") continue;
            else if (k == "camera") {
                if (!a.is_object()) fail("camera must be an object");
                for (const auto &ckv : a.obj) {
                    if (ckv.first == "fov") cam_fov = as_num(ckv.second, "fov");
                    else if (ckv.first == "position") as_vec3(ckv.second, "position", cam_pos);
                    else if (ckv.first == "target") as_vec3(ckv.second, "target", cam_target);
                    else if (ckv.first == "up") as_vec3(ckv.second, "up", cam_up);
                }
            } else if (k == "lights") {
                if (!a.is_array()) fail("lights must be a sequence");
                for (const JsonValue &l : a.arr) parse_light(l);
            } else if (k == "objects") {
                if (!a.is_array()) fail("objects must be a sequence");
                for (const JsonValue &o : a.arr) roots.push_back(parse_object(o));
            } else {
                fail("unknown field `" + k + "` in scene");
            }
        }
    }

    void parse_light(const JsonValue &l) {
        if (!l.is_object()) fail("light must be an object");
        const JsonValue *t = l.find("type");
        if (!t || !t->is_string()) fail("missing field `type` in light");
        rt_light out{};
        Mat4 xf = mat_identity();
        if (t->str == "ambient") {
            out.kind = RT_LIGHT_AMBIENT;
            for (const auto &kv : l.obj) {
                if (kv.first == "type") continue;
                if (kv.first == "color") as_vec3(kv.second, "color", out.color);
                else fail("unknown field `" + kv.first + "` in ambient light");
            }
        } else if (t->str == "point") {
            out.kind = RT_LIGHT_POINT;
            out.color[0] = out.color[1] = out.color[2] = 1.0;
            out.intensity = 10.0;
            for (const auto &kv : l.obj) {
                if (kv.first == "type") continue;
                if (kv.first == "color") as_vec3(kv.second, "color", out.color);
                else if (kv.first == "intensity") out.intensity = as_num(kv.second, "intensity");
                else if (kv.first == "transform") xf = parse_transform(kv.second);
                else fail("unknown field `" + kv.first + "` in point light");
            }
            // Transformed::get_position: matrix * origin (transform.rs:11-13)
            out.position[0] = xf.m[3];
            out.position[1] = xf.m[7];
            out.position[2] = xf.m[11];
        } else {
            fail("unknown variant `" + t->str + "`, expected `ambient` or `point`");
        }
        lights.push_back(out);
    }

    void finish() {
        // textures are registered in load_assets order: object tree DFS, each
        // object's own material before its children (primitives/mod.rs:45-61)
        struct Walk {
            Loader *self;
            void run(const Node &n) {
                if (n.material >= 0) {
                    const std::string &tex = self->material_texture[n.material];
                    if (!tex.empty()) self->materials[n.material].texture_id = self->texture_id_for(tex);
                }
                for (const Node &c : n.children) run(c);
            }
        } walk{this};
        for (const Node &r : roots) walk.run(r);

        Mat4 root = mat_identity();
        for (const Node &r : roots) flatten(r, root, -1);
        for (size_t i = 0; i < textures.size(); ++i) textures[i].rgb8 = texture_data[i].data();
        refresh_desc();
    }

    void refresh_desc() {
        desc.opts = opts;
        desc.camera.fov_deg = cam_fov;
        for (int i = 0; i < 3; ++i) desc.camera.position[i] = cam_pos[i];
        rth_camera_to_world(cam_pos, cam_target, cam_up, desc.camera.cam_to_world);
        desc.n_prims = static_cast<uint32_t>(prim_type.size());
        desc.prim_type = prim_type.data();
        desc.prim_material = prim_material.data();
        desc.prim_transform = prim_transform.data();
        desc.prim_param = prim_param.data();
        desc.prim_normal = prim_normal.data();
        desc.prim_tri = prim_tri.data();
        desc.n_transforms = static_cast<uint32_t>(world.size() / 12);
        desc.world = world.data();
        desc.inv_world = inv_world.data();
        desc.n_tris = static_cast<uint32_t>(tri_pos.size() / 9);
        desc.tri_pos = tri_pos.data();
        desc.tri_nrm = tri_nrm.data();
        desc.tri_uv = tri_uv.data();
        desc.n_materials = static_cast<uint32_t>(materials.size());
        desc.materials = materials.data();
        desc.n_lights = static_cast<uint32_t>(lights.size());
        desc.lights = lights.data();
        desc.n_textures = static_cast<uint32_t>(textures.size());
        desc.textures = textures.data();
    }
};

std::string dirname_of(const std::string &path) {
    size_t p = path.find_last_of('/');
    if (p == std::string::npos) return "";
    if (p == 0) return "/";
    return path.substr(0, p);
}

}  // namespace

struct rth_scene {
    Loader loader;
};

extern "C" {

const char *rth_last_error(void) { return g_error.c_str(); }

int rth_scene_load_string(const char *json, const char *asset_base, rth_scene **out) {
    if (!json || !out) {
        g_error = "null argument";
        return RT_ERR_INVALID;
    }
    *out = nullptr;
    rth_scene *s = nullptr;
    try {
        s = new rth_scene();
        s->loader.asset_base = asset_base ? asset_base : "";
        std::string text(json);
        rth::JsonParser parser(text);
        s->loader.parse_scene(parser.parse());
        s->loader.finish();
    } catch (const std::exception &e) {
        g_error = e.what();
        delete s;
        return RT_ERR_IO;
    }
    *out = s;
    return RT_OK;
}

int rth_scene_load(const char *json_path, rth_scene **out) {
    if (!json_path || !out) {
        g_error = "null argument";
        return RT_ERR_INVALID;
    }
    std::ifstream in(json_path);
    if (!in) {
        g_error = std::string("failed to open file: ") + json_path;
        return RT_ERR_IO;
    }
    std::stringstream ss;
    ss << in.rdbuf();
    std::string base = dirname_of(json_path);
    return rth_scene_load_string(ss.str().c_str(), base.c_str(), out);
}

const rt_scene_desc *rth_scene_desc(const rth_scene *scene) { return scene ? &scene->loader.desc : nullptr; }

int rth_scene_set_option(rth_scene *scene, const char *key, double value) {
    if (!scene || !key) {
        g_error = "null argument";
        return RT_ERR_INVALID;
    }
    rt_render_options &o = scene->loader.opts;
    std::string k(key);
    if (value < 0) {
        g_error = "negative option value";
        return RT_ERR_INVALID;
    }
    if (k == "width") o.width = static_cast<uint32_t>(value);
    else if (k == "height") o.height = static_cast<uint32_t>(value);
    else if (k == "max_depth") o.max_depth = static_cast<uint32_t>(value);
    else if (k == "samples_per_pixel") o.samples_per_pixel = static_cast<uint32_t>(value);
    else if (k == "max_reflected_rays") o.max_reflected_rays = static_cast<uint32_t>(value);
    else if (k == "max_illumination_rays") o.max_illumination_rays = static_cast<uint32_t>(value);
    else if (k == "max_occlusion_distance") o.max_occlusion_distance = value;
    else {
        g_error = "unknown option " + k;
        return RT_ERR_INVALID;
    }
    scene->loader.refresh_desc();
    return RT_OK;
}

void rth_scene_free(rth_scene *scene) { delete scene; }

int rth_transform_from_json(const char *json, double *m16, double *inv16, double *invt16) {
    try {
        std::string text(json);
        rth::JsonParser parser(text);
        Mat4 m = parse_transform(parser.parse());
        Mat4 inv;
        if (!mat_inverse(m, &inv)) fail("singular transform");
        Mat4 invt = mat_transpose(inv);
        for (int i = 0; i < 16; ++i) {
            if (m16) m16[i] = m.m[i];
            if (inv16) inv16[i] = inv.m[i];
            if (invt16) invt16[i] = invt.m[i];
        }
    } catch (const std::exception &e) {
        g_error = e.what();
        return RT_ERR_IO;
    }
    return RT_OK;
}

// Matrix4::look_at_rh(pos, target, up).transpose(), upper-left 3x3:
// columns are the camera's x, y, z axes (Rotation3::face_towards(-(target-pos), up)).
int rth_camera_to_world(const double *pos3, const double *target3, const double *up3, double *m9) {
    double dir[3], z[3], x[3], y[3];
    for (int i = 0; i < 3; ++i) dir[i] = target3[i] - pos3[i];
    for (int i = 0; i < 3; ++i) z[i] = -dir[i];
    vec_normalize(z);
    vec_cross(up3, z, x);
    vec_normalize(x);
    vec_cross(z, x, y);
    vec_normalize(y);
    for (int r = 0; r < 3; ++r) {
        m9[r * 3 + 0] = x[r];
        m9[r * 3 + 1] = y[r];
        m9[r * 3 + 2] = z[r];
    }
    return RT_OK;
}

int rth_image_load(const char *path, uint32_t *width, uint32_t *height, uint8_t *rgb8, uint64_t capacity) {
    if (!path || !width || !height) {
        g_error = "null argument";
        return RT_ERR_INVALID;
    }
    try {
        std::ifstream in(path, std::ios::binary);
        if (!in) {
            g_error = std::string("failed to open file: ") + path;
            return RT_ERR_IO;
        }
        std::vector<uint8_t> bytes((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
        std::vector<uint8_t> rgb;
        std::string why;
        if (!rth::decode_image(bytes, width, height, &rgb, &why)) {
            g_error = std::string(path) + ": " + why;
            return RT_ERR_IO;
        }
        if (rgb8) {
            if (capacity < rgb.size()) {
                g_error = "rgb8 buffer too small";
                return RT_ERR_INVALID;
            }
            std::memcpy(rgb8, rgb.data(), rgb.size());
        }
        return RT_OK;
    } catch (const std::exception &e) {
        g_error = e.what();
        return RT_ERR_IO;
    }
}

}  // extern "C"
