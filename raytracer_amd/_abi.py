"""ctypes mirrors of the structs in include/mi355rt.h (keep in sync)."""
import ctypes as C

RT_OK = 0
RT_ERR_INVALID = -1
RT_ERR_UNSUPPORTED = -2
RT_ERR_DEVICE = -3
RT_ERR_NOMEM = -4
RT_ERR_IO = -5

RT_PRIM_SPHERE, RT_PRIM_CUBE, RT_PRIM_PLANE, RT_PRIM_TRIANGLE = 0, 1, 2, 3
RT_MAT_PHONG, RT_MAT_PHYSICAL = 0, 1
RT_SIDE_BOTH, RT_SIDE_FRONT, RT_SIDE_BACK = 0, 1, 2
RT_LIGHT_AMBIENT, RT_LIGHT_POINT = 0, 1
RT_NO_HIT = 0xFFFFFFFF
RT_FLAG_COUNT_WORK = 1


class RenderOptions(C.Structure):
    _fields_ = [
        ("width", C.c_uint32),
        ("height", C.c_uint32),
        ("max_depth", C.c_uint32),
        ("samples_per_pixel", C.c_uint32),
        ("max_reflected_rays", C.c_uint32),
        ("max_illumination_rays", C.c_uint32),
        ("max_occlusion_distance", C.c_double),
    ]


class Camera(C.Structure):
    _fields_ = [
        ("fov_deg", C.c_double),
        ("position", C.c_double * 3),
        ("cam_to_world", C.c_double * 9),
    ]


class Material(C.Structure):
    _fields_ = [
        ("kind", C.c_uint32),
        ("side", C.c_uint32),
        ("color", C.c_double * 3),
        ("emissive", C.c_double * 3),
        ("specular", C.c_double * 3),
        ("reflectivity", C.c_double),
        ("shininess", C.c_double),
        ("opacity", C.c_double),
        ("roughness", C.c_double),
        ("metalness", C.c_double),
        ("refractive_index", C.c_double),
        ("texture_id", C.c_int32),
        ("_pad", C.c_uint32),
    ]


class Light(C.Structure):
    _fields_ = [
        ("kind", C.c_uint32),
        ("_pad", C.c_uint32),
        ("color", C.c_double * 3),
        ("intensity", C.c_double),
        ("position", C.c_double * 3),
    ]


class Texture(C.Structure):
    _fields_ = [
        ("width", C.c_uint32),
        ("height", C.c_uint32),
        ("rgb8", C.POINTER(C.c_uint8)),
    ]


class SceneDesc(C.Structure):
    _fields_ = [
        ("opts", RenderOptions),
        ("camera", Camera),
        ("n_prims", C.c_uint32),
        ("_pad0", C.c_uint32),
        ("prim_type", C.POINTER(C.c_uint8)),
        ("prim_material", C.POINTER(C.c_uint32)),
        ("prim_transform", C.POINTER(C.c_uint32)),
        ("prim_param", C.POINTER(C.c_double)),
        ("prim_normal", C.POINTER(C.c_double)),
        ("prim_tri", C.POINTER(C.c_uint32)),
        ("n_transforms", C.c_uint32),
        ("_pad1", C.c_uint32),
        ("world", C.POINTER(C.c_double)),
        ("inv_world", C.POINTER(C.c_double)),
        ("n_tris", C.c_uint32),
        ("_pad2", C.c_uint32),
        ("tri_pos", C.POINTER(C.c_double)),
        ("tri_nrm", C.POINTER(C.c_double)),
        ("tri_uv", C.POINTER(C.c_double)),
        ("n_materials", C.c_uint32),
        ("n_lights", C.c_uint32),
        ("materials", C.POINTER(Material)),
        ("lights", C.POINTER(Light)),
        ("n_textures", C.c_uint32),
        ("_pad3", C.c_uint32),
        ("textures", C.POINTER(Texture)),
    ]


class RenderParams(C.Structure):
    _fields_ = [
        ("seed", C.c_uint64),
        ("x0", C.c_uint32),
        ("y0", C.c_uint32),
        ("x1", C.c_uint32),
        ("y1", C.c_uint32),
        ("s0", C.c_uint32),
        ("s1", C.c_uint32),
        ("tile_size", C.c_uint32),
        ("shard_index", C.c_uint32),
        ("shard_count", C.c_uint32),
        ("flags", C.c_uint32),
    ]


class RenderStats(C.Structure):
    _fields_ = [
        ("ray_count", C.c_uint64),
        ("closest_rays", C.c_uint64),
        ("shadow_rays", C.c_uint64),
        ("camera_rays", C.c_uint64),
        ("nodes_visited_closest", C.c_uint64),
        ("prims_tested_closest", C.c_uint64),
        ("nodes_visited_shadow", C.c_uint64),
        ("prims_tested_shadow", C.c_uint64),
        ("batches", C.c_uint32),
        ("launches_closest", C.c_uint32),
        ("launches_shadow", C.c_uint32),
        ("launches_other", C.c_uint32),
        ("ms_total", C.c_float),
        ("ms_raygen", C.c_float),
        ("ms_closest", C.c_float),
        ("ms_shadow", C.c_float),
        ("ms_expand", C.c_float),
        ("ms_resolve", C.c_float),
        ("ms_accum", C.c_float),
        ("retries", C.c_uint32),
        ("launches_gi", C.c_uint32),
        ("ms_gi", C.c_float),
        ("gi_two_phase", C.c_uint32),
        ("gi_waves", C.c_uint32),
        ("overlapped", C.c_uint32),
        ("gi_rays", C.c_uint64),
        ("illum_rays", C.c_uint64),
        ("ms_illum", C.c_float),
        ("launches_illum", C.c_uint32),
    ]

    def as_dict(self):
        return {name: getattr(self, name) for name, _ in self._fields_ if not name.startswith("_")}
