//! Raw bindings to `libmi355rt.so` -- one to one with `include/mi355rt.h`.
//!
//! The library replaces the pixel loop of `RaytracingScene::raytrace_to_image`
//! (src/render/raytracing_scene.rs:727-746 of DarkAce65/raytracer): one
//! `rt_scene_create` per scene, one `rt_render` per frame.  Every entry point
//! returns 0 or a negative `RT_ERR_*`; nothing unwinds across the boundary.
#![allow(non_camel_case_types)]

use std::os::raw::{c_char, c_int, c_void};

pub const RT_OK: c_int = 0;
pub const RT_ERR_INVALID: c_int = -1;
pub const RT_ERR_UNSUPPORTED: c_int = -2;
pub const RT_ERR_DEVICE: c_int = -3;
pub const RT_ERR_NOMEM: c_int = -4;
pub const RT_ERR_IO: c_int = -5;

pub const RT_PRIM_SPHERE: u8 = 0;
pub const RT_PRIM_CUBE: u8 = 1;
pub const RT_PRIM_PLANE: u8 = 2;
pub const RT_PRIM_TRIANGLE: u8 = 3;
pub const RT_MAT_PHONG: u32 = 0;
pub const RT_MAT_PHYSICAL: u32 = 1;
pub const RT_SIDE_BOTH: u32 = 0;
pub const RT_SIDE_FRONT: u32 = 1;
pub const RT_SIDE_BACK: u32 = 2;
pub const RT_LIGHT_AMBIENT: u32 = 0;
pub const RT_LIGHT_POINT: u32 = 1;
pub const RT_NO_HIT: u32 = 0xFFFF_FFFF;
pub const RT_FLAG_COUNT_WORK: u32 = 1;

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rt_render_options {
    pub width: u32,
    pub height: u32,
    pub max_depth: u32,
    pub samples_per_pixel: u32,
    pub max_reflected_rays: u32,
    pub max_illumination_rays: u32,
    pub max_occlusion_distance: f64,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rt_camera {
    pub fov_deg: f64,
    pub position: [f64; 3],
    pub cam_to_world: [f64; 9],
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rt_material {
    pub kind: u32,
    pub side: u32,
    pub color: [f64; 3],
    pub emissive: [f64; 3],
    pub specular: [f64; 3],
    pub reflectivity: f64,
    pub shininess: f64,
    pub opacity: f64,
    pub roughness: f64,
    pub metalness: f64,
    pub refractive_index: f64,
    pub texture_id: i32,
    pub _pad: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rt_light {
    pub kind: u32,
    pub _pad: u32,
    pub color: [f64; 3],
    pub intensity: f64,
    pub position: [f64; 3],
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct rt_texture {
    pub width: u32,
    pub height: u32,
    pub rgb8: *const u8,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct rt_scene_desc {
    pub opts: rt_render_options,
    pub camera: rt_camera,
    pub n_prims: u32,
    pub _pad0: u32,
    pub prim_type: *const u8,
    pub prim_material: *const u32,
    pub prim_transform: *const u32,
    pub prim_param: *const f64,
    pub prim_normal: *const f64,
    pub prim_tri: *const u32,
    pub n_transforms: u32,
    pub _pad1: u32,
    pub world: *const f64,
    pub inv_world: *const f64,
    pub n_tris: u32,
    pub _pad2: u32,
    pub tri_pos: *const f64,
    pub tri_nrm: *const f64,
    pub tri_uv: *const f64,
    pub n_materials: u32,
    pub n_lights: u32,
    pub materials: *const rt_material,
    pub lights: *const rt_light,
    pub n_textures: u32,
    pub _pad3: u32,
    pub textures: *const rt_texture,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rt_render_params {
    pub seed: u64,
    pub x0: u32,
    pub y0: u32,
    pub x1: u32,
    pub y1: u32,
    pub s0: u32,
    pub s1: u32,
    pub tile_size: u32,
    pub shard_index: u32,
    pub shard_count: u32,
    pub flags: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rt_render_stats {
    pub ray_count: u64,
    pub closest_rays: u64,
    pub shadow_rays: u64,
    pub camera_rays: u64,
    pub nodes_visited_closest: u64,
    pub prims_tested_closest: u64,
    pub nodes_visited_shadow: u64,
    pub prims_tested_shadow: u64,
    pub batches: u32,
    pub launches_closest: u32,
    pub launches_shadow: u32,
    pub launches_other: u32,
    pub ms_total: f32,
    pub ms_raygen: f32,
    pub ms_closest: f32,
    pub ms_shadow: f32,
    pub ms_expand: f32,
    pub ms_resolve: f32,
    pub ms_accum: f32,
    pub retries: u32,
    pub launches_gi: u32,
    pub ms_gi: f32,
    pub gi_two_phase: u32,
    pub gi_waves: u32,
    pub overlapped: u32,
    pub gi_rays: u64,
    pub illum_rays: u64,
    pub ms_illum: f32,
    pub launches_illum: u32,
}

/// Opaque handle owned by the library; one host thread at a time.
#[repr(C)]
pub struct rt_scene {
    _private: [u8; 0],
}

/// `int (*rt_progress_fn)(void *user, uint32_t passes_done, uint32_t n_passes, uint64_t rays_so_far)`; NULL allowed.
pub type rt_progress_fn = Option<unsafe extern "C" fn(user: *mut c_void, passes_done: u32, n_passes: u32, rays_so_far: u64) -> c_int>;

extern "C" {
    pub fn rt_scene_create(desc: *const rt_scene_desc, out: *mut *mut rt_scene) -> c_int;
    pub fn rt_scene_destroy(scene: *mut rt_scene);
    pub fn rt_scene_set_pool_bytes(scene: *mut rt_scene, bytes: u64) -> c_int;
    pub fn rt_render_device(scene: *mut rt_scene, params: *const rt_render_params, d_rgb_accum: *mut f64, ray_count: *mut u64, stats: *mut rt_render_stats, stream: *mut c_void) -> c_int;
    pub fn rt_render_device_noalloc(scene: *mut rt_scene, params: *const rt_render_params, d_rgb_accum: *mut f64, ray_count: *mut u64, stream: *mut c_void) -> c_int;
    pub fn rt_render_progressive(scene: *mut rt_scene, params: *const rt_render_params, n_passes: u32, argb: *mut u32, on_pass: rt_progress_fn, user: *mut c_void, rgb_accum: *mut f64, ray_count: *mut u64) -> c_int;
    pub fn rt_render_multi_device(scenes: *const *mut rt_scene, n_scenes: u32, params: *const rt_render_params, d_rgb_accum: *mut f64, ray_count: *mut u64, stats: *mut rt_render_stats, stream: *mut c_void) -> c_int;
    pub fn rt_render_multi(scenes: *const *mut rt_scene, n_scenes: u32, params: *const rt_render_params, rgb_accum: *mut f64, rgba8: *mut u8, ray_count: *mut u64, stats: *mut rt_render_stats) -> c_int;
    pub fn rt_render_aov_device(scene: *mut rt_scene, params: *const rt_render_params, d_rgb_accum: *mut f64, d_albedo_accum: *mut f64, d_normal: *mut f64, ray_count: *mut u64, stats: *mut rt_render_stats, stream: *mut c_void) -> c_int;
    pub fn rt_render_aov(scene: *mut rt_scene, params: *const rt_render_params, rgb_accum: *mut f64, albedo: *mut f64, normal: *mut f64, ray_count: *mut u64, stats: *mut rt_render_stats) -> c_int;
    pub fn rt_tonemap_device(scene: *mut rt_scene, d_rgb_accum: *const f64, d_rgba8: *mut u8, stream: *mut c_void) -> c_int;
    pub fn rt_render(scene: *mut rt_scene, params: *const rt_render_params, rgb_accum: *mut f64, rgba8: *mut u8, ray_count: *mut u64, stats: *mut rt_render_stats) -> c_int;
    pub fn rt_query_primary(scene: *mut rt_scene, prim_id: *mut u32, t: *mut f64, normal3: *mut f64) -> c_int;
    pub fn rt_trace_rays(scene: *mut rt_scene, n: u32, origin: *const f64, dir: *const f64, tmax: *const f64, shadow: c_int, prim_id: *mut u32, t: *mut f64) -> c_int;
    pub fn rt_scene_bvh(scene: *mut rt_scene, nodes32: *mut c_void, node_capacity: u64, leaf_prim_ids: *mut u32, leaf_capacity: u64, info4: *mut u64) -> c_int;
    pub fn rt_scene_info(scene: *mut rt_scene, out8: *mut u64) -> c_int;
    pub fn rt_shard_tiles(opts: *const rt_render_options, params: *const rt_render_params, tile_xy: *mut u32, cap: u32, n_tiles: *mut u32, tile_w: *mut u32, tile_h: *mut u32) -> c_int;
    pub fn rt_last_error() -> *const c_char;
    pub fn rt_version() -> *const c_char;
}

/// Owning wrapper: `Scene::new(&desc)?.render(..)`; the handle is destroyed on drop.
pub struct Scene {
    raw: *mut rt_scene,
    pub width: u32,
    pub height: u32,
}

/// The library's message for the last failure on this thread.
pub fn last_error() -> String {
    unsafe {
        let p = rt_last_error();
        if p.is_null() {
            String::new()
        } else {
            std::ffi::CStr::from_ptr(p).to_string_lossy().into_owned()
        }
    }
}

fn check(rc: c_int) -> Result<(), String> {
    if rc == RT_OK {
        Ok(())
    } else {
        Err(format!("mi355rt error {}: {}", rc, last_error()))
    }
}

impl Scene {
    /// The library copies everything `desc` points to; the caller's arrays may be freed after this returns.
    pub fn new(desc: &rt_scene_desc) -> Result<Scene, String> {
        let mut raw: *mut rt_scene = std::ptr::null_mut();
        check(unsafe { rt_scene_create(desc, &mut raw) })?;
        Ok(Scene { raw, width: desc.opts.width, height: desc.opts.height })
    }

    /// One frame: RGBA8 (`width*height*4`, the reference's gamma and truncating quantisation) and the ray count.
    pub fn render(&mut self, seed: u64) -> Result<(Vec<u8>, u64), String> {
        let mut rgba = vec![0u8; self.width as usize * self.height as usize * 4];
        let mut rays = 0u64;
        let params = rt_render_params { seed, ..Default::default() };
        check(unsafe { rt_render(self.raw, &params, std::ptr::null_mut(), rgba.as_mut_ptr(), &mut rays, std::ptr::null_mut()) })?;
        Ok((rgba, rays))
    }

    /// Colour sum, final albedo and sample-0 normal per pixel (`width*height*3` doubles each): ColorData's members.
    pub fn render_aov(&mut self, seed: u64) -> Result<(Vec<f64>, Vec<f64>, Vec<f64>, u64), String> {
        let n = self.width as usize * self.height as usize * 3;
        let (mut rgb, mut albedo, mut normal) = (vec![0f64; n], vec![0f64; n], vec![0f64; n]);
        let mut rays = 0u64;
        let params = rt_render_params { seed, ..Default::default() };
        check(unsafe { rt_render_aov(self.raw, &params, rgb.as_mut_ptr(), albedo.as_mut_ptr(), normal.as_mut_ptr(), &mut rays, std::ptr::null_mut()) })?;
        Ok((rgb, albedo, normal, rays))
    }

    pub fn as_ptr(&self) -> *mut rt_scene {
        self.raw
    }
}

impl Drop for Scene {
    fn drop(&mut self) {
        unsafe { rt_scene_destroy(self.raw) }
    }
}
