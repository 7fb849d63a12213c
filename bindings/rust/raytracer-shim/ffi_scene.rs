//! src/render/ffi_scene.rs -- drop-in module for DarkAce65/raytracer: flattens the scene into an
//! `rt_scene_desc`, hands the frame to libmi355rt.so and returns what `raytrace_to_image` returns.
//!
//! NOT compiled in this repository's image (no Rust toolchain).  It uses only items the reference
//! already has, plus one new trait method (`RaytracingObject::ffi_describe`, see README.md here).
//!
//! Order matters: the index of an object in the Vec built by `RaytracingScene::from_scene`
//! (src/render/scene.rs:84-87) IS its primitive id, so the flattening happens there, BEFORE
//! `KdTreeAccelerator::new` partitions the objects (src/core/bounds.rs:256-263) -- and the kd-tree
//! build can be skipped altogether on this path.
use crate::core::{Material, MaterialSide, Texture, Transformed};
use crate::lights::Light;
use crate::primitives::RaytracingObject;
use crate::render::{CastStats, CastTimings, RenderOptions};
use mi355rt_sys as ffi;
use std::collections::HashMap;

/// What a primitive tells the flattener (new trait method on `RaytracingObject`).
pub enum FfiPrimitive {
    Sphere { radius: f64 },
    Cube { size: f64 },
    /// `normal` exactly as deserialised (NOT normalised: serde does not renormalise `Unit<_>`).
    Plane { normal: [f64; 3] },
    /// object-space vertices, unit normals, texture coordinates (src/primitives/triangle.rs:148-152)
    Triangle { pos: [[f64; 3]; 3], nrm: [[f64; 3]; 3], uv: [[f64; 2]; 3] },
}

/// Owns every array the descriptor points to.
#[derive(Default)]
pub struct FlatScene {
    prim_type: Vec<u8>,
    prim_material: Vec<u32>,
    prim_transform: Vec<u32>,
    prim_param: Vec<f64>,
    prim_normal: Vec<f64>,
    prim_tri: Vec<u32>,
    world: Vec<f64>,
    inv_world: Vec<f64>,
    tri_pos: Vec<f64>,
    tri_nrm: Vec<f64>,
    tri_uv: Vec<f64>,
    materials: Vec<ffi::rt_material>,
    lights: Vec<ffi::rt_light>,
    texels: Vec<Vec<u8>>,
    textures: Vec<ffi::rt_texture>,
    opts: ffi::rt_render_options,
    camera: ffi::rt_camera,
}

fn affine_rows(m: &nalgebra::Matrix4<f64>) -> [f64; 12] {
    // row-major 3x4 block [r0 r1 r2 t]
    let mut out = [0.0; 12];
    for r in 0..3 {
        for c in 0..4 {
            out[4 * r + c] = m[(r, c)];
        }
    }
    out
}

impl FlatScene {
    pub fn build(
        objects: &[Box<dyn RaytracingObject>],
        options: &RenderOptions,
        camera: &super::raytracing_scene::RaytracingCamera,
        lights: &[Light],
        textures: &HashMap<String, Texture>,
    ) -> FlatScene {
        let mut f = FlatScene::default();
        let mut texture_ids: HashMap<String, i32> = HashMap::new();
        for object in objects {
            // one transform per object; a mesh's faces repeat theirs, which the library's per-ray cache absorbs
            let t = object.get_transform();
            f.prim_transform.push((f.world.len() / 12) as u32);
            f.world.extend_from_slice(&affine_rows(t.matrix().matrix()));
            f.inv_world.extend_from_slice(&affine_rows(t.inverse().matrix()));
            f.prim_material.push(f.materials.len() as u32);
            f.materials.push(f.flatten_material(object.get_material(), textures, &mut texture_ids));
            let (ty, param, normal, tri) = match object.ffi_describe() {
                FfiPrimitive::Sphere { radius } => (ffi::RT_PRIM_SPHERE, radius, [0.0; 3], ffi::RT_NO_HIT),
                FfiPrimitive::Cube { size } => (ffi::RT_PRIM_CUBE, size, [0.0; 3], ffi::RT_NO_HIT),
                FfiPrimitive::Plane { normal } => (ffi::RT_PRIM_PLANE, 0.0, normal, ffi::RT_NO_HIT),
                FfiPrimitive::Triangle { pos, nrm, uv } => {
                    let index = (f.tri_pos.len() / 9) as u32;
                    for v in 0..3 {
                        f.tri_pos.extend_from_slice(&pos[v]);
                        f.tri_nrm.extend_from_slice(&nrm[v]);
                        f.tri_uv.extend_from_slice(&uv[v]);
                    }
                    (ffi::RT_PRIM_TRIANGLE, 0.0, [0.0; 3], index)
                }
            };
            f.prim_type.push(ty);
            f.prim_param.push(param);
            f.prim_normal.extend_from_slice(&normal);
            f.prim_tri.push(tri);
        }
        for light in lights {
            f.lights.push(match light {
                Light::Ambient(a) => ffi::rt_light { kind: ffi::RT_LIGHT_AMBIENT, color: a.get_color().into(), ..Default::default() },
                Light::Point(p) => ffi::rt_light {
                    kind: ffi::RT_LIGHT_POINT,
                    color: p.color.into(),
                    intensity: p.intensity,
                    position: p.get_position().coords.into(), // transform * origin, src/core/transform.rs:11-13
                    ..Default::default()
                },
            });
        }
        f.opts = ffi::rt_render_options {
            width: options.width,
            height: options.height,
            max_depth: u32::from(options.max_depth),
            samples_per_pixel: u32::from(options.samples_per_pixel),
            max_reflected_rays: u32::from(options.max_reflected_rays),
            max_illumination_rays: u32::from(options.max_illumination_rays),
            max_occlusion_distance: options.max_occlusion_distance,
        };
        let m = &camera.camera_to_world; // look_at_rh(..).transpose(), raytracing_scene.rs:33-34
        f.camera = ffi::rt_camera {
            fov_deg: camera.fov,
            position: camera.position.coords.into(),
            cam_to_world: [m[(0, 0)], m[(0, 1)], m[(0, 2)], m[(1, 0)], m[(1, 1)], m[(1, 2)], m[(2, 0)], m[(2, 1)], m[(2, 2)]],
        };
        f
    }

    fn flatten_material(&mut self, material: &Material, textures: &HashMap<String, Texture>, ids: &mut HashMap<String, i32>) -> ffi::rt_material {
        let side = |s: &MaterialSide| match s {
            MaterialSide::Both => ffi::RT_SIDE_BOTH,
            MaterialSide::Front => ffi::RT_SIDE_FRONT,
            MaterialSide::Back => ffi::RT_SIDE_BACK,
        };
        let mut texture_id = |path: &Option<String>| -> i32 {
            let path = match path {
                Some(p) => p,
                None => return -1,
            };
            if let Some(id) = ids.get(path) {
                return *id;
            }
            let texture = &textures[path]; // decoded RGB8, row 0 = top (src/core/texture.rs:35-44)
            let id = self.textures.len() as i32;
            self.texels.push(texture.as_rgb8_bytes().to_vec());
            self.textures.push(ffi::rt_texture { width: texture.width(), height: texture.height(), rgb8: self.texels.last().unwrap().as_ptr() });
            ids.insert(path.clone(), id);
            id
        };
        match material {
            Material::Phong(m) => ffi::rt_material {
                kind: ffi::RT_MAT_PHONG,
                side: side(&m.side),
                color: m.color.into(),
                emissive: m.emissive.into(),
                specular: m.specular.into(),
                reflectivity: m.reflectivity,
                shininess: m.shininess,
                texture_id: texture_id(&m.texture_path),
                ..Default::default()
            },
            Material::Physical(m) => ffi::rt_material {
                kind: ffi::RT_MAT_PHYSICAL,
                side: side(&m.side),
                color: m.color.into(),
                emissive: m.emissive.into(),
                opacity: m.opacity,
                roughness: m.roughness,
                metalness: m.metalness,
                refractive_index: m.refractive_index,
                texture_id: texture_id(&m.texture_path),
                ..Default::default()
            },
        }
    }

    /// Pointers into `self`: valid while `self` is alive and unmoved (the library copies everything in `Scene::new`).
    pub fn desc(&self) -> ffi::rt_scene_desc {
        ffi::rt_scene_desc {
            opts: self.opts,
            camera: self.camera,
            n_prims: self.prim_type.len() as u32,
            _pad0: 0,
            prim_type: self.prim_type.as_ptr(),
            prim_material: self.prim_material.as_ptr(),
            prim_transform: self.prim_transform.as_ptr(),
            prim_param: self.prim_param.as_ptr(),
            prim_normal: self.prim_normal.as_ptr(),
            prim_tri: self.prim_tri.as_ptr(),
            n_transforms: (self.world.len() / 12) as u32,
            _pad1: 0,
            world: self.world.as_ptr(),
            inv_world: self.inv_world.as_ptr(),
            n_tris: (self.tri_pos.len() / 9) as u32,
            _pad2: 0,
            tri_pos: self.tri_pos.as_ptr(),
            tri_nrm: self.tri_nrm.as_ptr(),
            tri_uv: self.tri_uv.as_ptr(),
            n_materials: self.materials.len() as u32,
            n_lights: self.lights.len() as u32,
            materials: self.materials.as_ptr(),
            lights: self.lights.as_ptr(),
            n_textures: self.textures.len() as u32,
            _pad3: 0,
            textures: self.textures.as_ptr(),
        }
    }
}

/// Body of `RaytracingScene::raytrace_to_image` on the MI355X path (replaces :727-769).
pub fn raytrace_to_image(gpu: &mut ffi::Scene, seed: u64) -> (image::RgbaImage, CastTimings, CastStats) {
    let start = std::time::Instant::now();
    let (rgba, rays) = gpu.render(seed).unwrap_or_else(|e| panic!("{}", e)); // the reference fails by panic
    let mut timings = CastTimings::default();
    timings.ray_casting_duration = start.elapsed();
    let image = image::RgbaImage::from_raw(gpu.width, gpu.height, rgba).expect("w*h*4 bytes");
    (image, timings, CastStats { ray_count: rays })
}
