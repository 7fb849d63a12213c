/*
 * mi355rt_host.h -- host-side scene front end (C ABI).
 *
 * The reference's host is Rust (serde_json + tobj + image); no Rust toolchain
 * exists in the build image, so this library plays its role: it parses the
 * scene JSON (schema: src/render/scene.rs:10-22, src/render/mod.rs:113-159,
 * src/primitives/{cube,sphere,plane,triangle,mesh,group}.rs, src/core/{material,transform}.rs, src/lights/),
 * loads OBJ meshes with tobj's semantics (src/primitives/mesh.rs:57-151),
 * bakes world transforms in flatten_to_world order (src/render/scene.rs:82-87)
 * and hands out the flattened `rt_scene_desc` that `rt_scene_create`
 * (mi355rt.h) consumes.  It is outside the accelerated hot path.
 */
#ifndef MI355RT_HOST_H
#define MI355RT_HOST_H

#include "mi355rt.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rth_scene rth_scene; /* opaque; owns every array of its desc */

/* main.rs:34-49: open + serde_json::from_reader + Scene::load_assets(parent dir)
 * + flatten.  Textures are decoded in the library (rth_image_load's decoder). */
int rth_scene_load(const char *json_path, rth_scene **out);
/* same from a JSON string; asset paths resolve against asset_base */
int rth_scene_load_string(const char *json, const char *asset_base, rth_scene **out);
const rt_scene_desc *rth_scene_desc(const rth_scene *scene);
/* override a RenderOptions field by its JSON name (e.g. "samples_per_pixel") */
int rth_scene_set_option(rth_scene *scene, const char *key, double value);
void rth_scene_free(rth_scene *scene);

/* Transform deserialisation (src/core/transform.rs:90-138) for the
 * known-answer tests: row-major 4x4 matrix, inverse, inverse-transpose. */
int rth_transform_from_json(const char *json, double *m16, double *inv16, double *invt16);
/* Camera -> cam_to_world 3x3 (src/render/raytracing_scene.rs:31-42). */
int rth_camera_to_world(const double *pos3, const double *target3, const double *up3, double *m9);

/* Texture::load (src/core/texture.rs:35-44): `image::open(path)?.to_rgb8()` for PNG (non-interlaced) and baseline /
 * extended-sequential JPEG files.  Writes the image size; when rgb8 is non-NULL also the width*height*3 RGB bytes (row 0
 * = top; capacity = bytes available).  RT_ERR_IO for unreadable / unsupported files (message in rth_last_error). */
int rth_image_load(const char *path, uint32_t *width, uint32_t *height, uint8_t *rgb8, uint64_t capacity);

/* Write an RGBA8 buffer as a PNG (image.save, main.rs:73); stored deflate. */
int rth_write_png(const char *path, uint32_t w, uint32_t h, const uint8_t *rgba8);

const char *rth_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
