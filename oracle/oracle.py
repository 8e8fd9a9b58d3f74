"""ctypes binding of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never by the product package.  See the header
of arctic_oracle.cpp for what the oracle restates and how it is pinned
("parity unpinned": the reference holds no golden vectors for this path).
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ARCTIC_ORACLE_LIB") or os.path.join(HERE, "liboracle.so")   # override: the sanitizer build (Makefile: liboracle_asan.so)


class Camera(C.Structure):
    _fields_ = [("eye", C.c_float * 3), ("rotation", C.c_float * 2), ("aspect", C.c_float),
                ("fov_y", C.c_float), ("z_near_far", C.c_float * 2)]


class DirectionalLight(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("rotation", C.c_float * 2), ("color", C.c_float * 3)]


class Scene(C.Structure):
    _fields_ = [("camera", Camera), ("ambient", C.c_float), ("sun", DirectionalLight),
                ("point_lights", C.c_void_p), ("n_point_lights", C.c_uint64),
                ("objects", C.c_void_p), ("n_objects", C.c_uint64)]


class Settings(C.Structure):
    _fields_ = [("tm_method", C.c_int32), ("gamma", C.c_float), ("exposure", C.c_float)]


VERTEX_DTYPE = np.dtype([("position", "<f4", 3), ("normal", "<f4", 3), ("tangent", "<f4", 3),
                         ("bitangent", "<f4", 3), ("tex_coords", "<f4", 2)])
OBJECT_DTYPE = np.dtype([("trs", "<f4", 16), ("mesh_idx", "<u8")])
LIGHT_DTYPE = np.dtype([("position", "<f4", 3), ("padding0", "<u4"), ("color", "<f4", 3), ("padding1", "<u4")])
assert VERTEX_DTYPE.itemsize == 56 and OBJECT_DTYPE.itemsize == 72 and LIGHT_DTYPE.itemsize == 32


def build(force=False):
    """compile liboracle.so with the committed Makefile (g++)."""
    src = os.path.join(HERE, "arctic_oracle.cpp")
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        vp, u32, u64, i32, f32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int32, C.c_float
        L.oracle_create.restype = vp
        L.oracle_create.argtypes = [u32, u32, u32, u32, u32, u32]
        L.oracle_destroy.argtypes = [vp]
        L.oracle_create_material.argtypes = [vp, vp, u32, u32, vp, u32, u32, vp, u32, u32]
        L.oracle_create_mesh.argtypes = [vp, vp, u64, vp, u64, u64]
        L.oracle_update_lights.argtypes = [vp, vp, u64]
        for name in ("oracle_pass_shadow_map", "oracle_pass_gbuffer"):
            getattr(L, name).argtypes = [vp, C.POINTER(Scene)]
        L.oracle_pass_shade.argtypes = [vp, C.POINTER(Scene), C.POINTER(Settings), i32]
        L.oracle_shade_gbuffer.argtypes = [vp, C.POINTER(Scene), C.POINTER(Settings), vp, vp, u32, vp, vp, vp, i32]
        L.oracle_render_frame.argtypes = [vp, C.POINTER(Scene), C.POINTER(Settings), vp, i32]
        L.oracle_read_gbuffer.argtypes = [vp, vp, vp, vp, vp]
        L.oracle_write_gbuffer.argtypes = [vp, vp, vp]
        L.oracle_read_shadow_map.argtypes = [vp, vp]
        L.oracle_write_shadow_map.argtypes = [vp, vp]
        L.oracle_read_output.argtypes = [vp, vp, vp, vp]
        L.oracle_stats.argtypes = [vp, vp, u32]
        L.oracle_frame_constants.argtypes = [C.POINTER(Scene), vp, vp, vp]
        L.oracle_dir_from_rot.argtypes = [vp, vp]
        L.oracle_outgoing_radiance.argtypes = [vp, vp, vp, vp, vp, f32, f32, vp]
        L.oracle_tonemap.argtypes = [i32, f32, f32, vp, vp, vp]
        L.oracle_calculate_shadow.restype = f32
        L.oracle_calculate_shadow.argtypes = [vp, u32, vp]
        L.oracle_fetch_surface.argtypes = [vp, u32, f32, f32, vp, vp]
        L.oracle_to_unorm8.restype = C.c_uint8
        L.oracle_to_unorm8.argtypes = [f32]
        L.oracle_hardware_threads.restype = i32
        L.oracle_set_precision.argtypes = [vp, i32]
        L.oracle_set_sampler_mode.argtypes = [vp, i32]
        L.oracle_set_hdr16.argtypes = [vp, i32]
        L.oracle_create_hdri.argtypes = [vp, vp, u32, u32]
        L.oracle_sky_ray.argtypes = [vp, C.POINTER(Camera), u32, u32, vp]
        L.oracle_sky_ray.restype = None
        L.oracle_sample_environment.argtypes = [vp, vp, vp, vp]
        L.oracle_sample_environment.restype = None
        L.oracle_through_half.restype = f32
        L.oracle_through_half.argtypes = [f32]
        _lib = L
    return _lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class Oracle:
    """mirror of the Renderer surface (reference src/renderer/renderer.hpp:100-125) on the CPU oracle."""

    def __init__(self, width, height, shadow_size=0, max_lights=16, row_begin=0, row_end=0):
        self.L = lib()
        self.width, self.height, self.shadow_size = width, height, shadow_size
        self.row_begin, self.row_end = (row_begin, row_end) if row_end else (0, height)
        self.rows = self.row_end - self.row_begin
        self.h = self.L.oracle_create(width, height, shadow_size, max_lights, row_begin, row_end)
        if not self.h:
            raise ValueError("oracle_create failed")

    def set_precision(self, bits):
        """64 (default): BRDF/tonemap in float64 = the parity arbiter; 32: literal fp32 restatement (CPU baseline)."""
        assert self.L.oracle_set_precision(self.h, bits) == 0
        return self

    def set_sampler_mode(self, mode):
        """variants of what the reference leaves to the D3D12 sampler hardware (arctic_oracle.cpp SAMPLER_*): bit 0 material textures
        with 8-bit filter weights, bit 1 sRGB decoded after filtering, bit 2 the shadow map's taps with 8-bit weights.  0 (default) =
        the semantics the HIP kernels implement.  Used to BOUND the unpinned-parity gap, never as the checker."""
        assert self.L.oracle_set_sampler_mode(self.h, int(mode)) == 0

    def set_hdr16(self, on):
        """route ps_main's colour through binary16 like the reference's RGBA16F target (forward_pass.cpp:149)."""
        assert self.L.oracle_set_hdr16(self.h, int(on)) == 0
        return self

    def create_hdri(self, rgba32f):
        a = _f32(rgba32f)
        assert a.ndim == 3 and a.shape[2] == 4
        assert self.L.oracle_create_hdri(self.h, _ptr(a), a.shape[1], a.shape[0]) == 0

    def sky_ray(self, camera, x, y):
        cam = Camera()
        cam.eye[:] = [float(v) for v in camera["eye"]]
        cam.rotation[:] = [float(v) for v in camera["rotation"]]
        cam.aspect, cam.fov_y = float(camera["aspect"]), float(camera["fov_y"])
        cam.z_near_far[:] = [float(v) for v in camera["z_near_far"]]
        out = np.zeros(3, np.float32)
        self.L.oracle_sky_ray(self.h, C.byref(cam), x, y, _ptr(out))
        return out

    def sample_environment(self, direction):
        d, rgb, uv = _f32(direction), np.zeros(3), np.zeros(2)
        self.L.oracle_sample_environment(self.h, _ptr(d), _ptr(rgb), _ptr(uv))
        return rgb, uv

    def close(self):
        if self.h:
            self.L.oracle_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def create_material(self, diffuse, normal, metal_rough):
        d, n, m = (np.ascontiguousarray(t, dtype=np.uint8) for t in (diffuse, normal, metal_rough))
        r = self.L.oracle_create_material(self.h, _ptr(d), d.shape[1], d.shape[0], _ptr(n), n.shape[1], n.shape[0],
                                          _ptr(m), m.shape[1], m.shape[0])
        if r < 0:
            raise ValueError("oracle_create_material failed")
        return r

    def create_mesh(self, vertices, indices, material_idx):
        v = np.ascontiguousarray(vertices, dtype=VERTEX_DTYPE)
        i = np.ascontiguousarray(indices, dtype=np.uint32).ravel()
        r = self.L.oracle_create_mesh(self.h, _ptr(v), len(v), _ptr(i), len(i), material_idx)
        if r < 0:
            raise ValueError("oracle_create_mesh failed")
        return r

    def update_lights(self, lights):
        l = np.ascontiguousarray(lights, dtype=LIGHT_DTYPE)
        if self.L.oracle_update_lights(self.h, _ptr(l) if len(l) else None, len(l)) < 0:
            raise ValueError("oracle_update_lights failed")

    @staticmethod
    def _scene(desc):
        return desc.fill(Scene())

    @staticmethod
    def _settings(settings):
        tm, gamma, exposure = settings
        return Settings(int(tm), float(gamma), float(exposure))

    def pass_shadow_map(self, desc):
        s = self._scene(desc)
        assert self.L.oracle_pass_shadow_map(self.h, C.byref(s)) == 0

    def pass_gbuffer(self, desc):
        s = self._scene(desc)
        assert self.L.oracle_pass_gbuffer(self.h, C.byref(s)) == 0

    def pass_shade(self, desc, settings, threads=8):
        s, st = self._scene(desc), self._settings(settings)
        assert self.L.oracle_pass_shade(self.h, C.byref(s), C.byref(st), threads) == 0

    def render_frame(self, desc, settings, threads=8):
        s, st = self._scene(desc), self._settings(settings)
        out = np.empty((self.rows, self.width, 4), np.uint8)
        assert self.L.oracle_render_frame(self.h, C.byref(s), C.byref(st), _ptr(out), threads) == 0
        return out

    def shade_gbuffer(self, desc, settings, attrs, matid, threads=8, want=("ldr", "rgba8")):
        """shade an arbitrary row-major G-buffer stripe; returns dict of outputs."""
        s, st = self._scene(desc), self._settings(settings)
        attrs, matid = _f32(attrs), np.ascontiguousarray(matid, dtype=np.uint32)
        rows = matid.shape[0]
        assert attrs.shape == (rows, self.width, 18) and matid.shape == (rows, self.width)
        out = {}
        if "hdr" in want:
            out["hdr"] = np.empty((rows, self.width, 3), np.float32)
        if "ldr" in want:
            out["ldr"] = np.empty((rows, self.width, 3), np.float32)
        if "rgba8" in want:
            out["rgba8"] = np.empty((rows, self.width, 4), np.uint8)
        rc = self.L.oracle_shade_gbuffer(self.h, C.byref(s), C.byref(st), _ptr(attrs), _ptr(matid), rows,
                                         _ptr(out.get("hdr")), _ptr(out.get("ldr")), _ptr(out.get("rgba8")), threads)
        assert rc == 0
        return out

    def read_gbuffer(self):
        n = (self.rows, self.width)
        attrs, mat = np.empty(n + (18,), np.float32), np.empty(n, np.uint32)
        depth, tri = np.empty(n, np.float32), np.empty(n, np.uint32)
        assert self.L.oracle_read_gbuffer(self.h, _ptr(attrs), _ptr(mat), _ptr(depth), _ptr(tri)) == 0
        return attrs, mat, depth, tri

    def write_gbuffer(self, attrs, matid):
        attrs, matid = _f32(attrs), np.ascontiguousarray(matid, dtype=np.uint32)
        assert attrs.shape == (self.rows, self.width, 18)
        assert self.L.oracle_write_gbuffer(self.h, _ptr(attrs), _ptr(matid)) == 0

    def read_shadow_map(self):
        d = np.empty((self.shadow_size, self.shadow_size), np.float32)
        assert self.L.oracle_read_shadow_map(self.h, _ptr(d)) == 0
        return d

    def write_shadow_map(self, d):
        d = _f32(d)
        assert d.shape == (self.shadow_size, self.shadow_size)
        assert self.L.oracle_write_shadow_map(self.h, _ptr(d)) == 0

    def read_output(self):
        n = (self.rows, self.width)
        ldr, hdr, rgba = np.empty(n + (3,), np.float32), np.empty(n + (3,), np.float32), np.empty(n + (4,), np.uint8)
        assert self.L.oracle_read_output(self.h, _ptr(ldr), _ptr(hdr), _ptr(rgba)) == 0
        return ldr, hdr, rgba

    def stats(self):
        s = np.zeros(6, np.uint64)
        self.L.oracle_stats(self.h, _ptr(s), 6)
        return s

    def fetch_surface(self, mat, u, v, tbn=(1, 0, 0, 0, 1, 0, 0, 0, 1)):
        t, out = _f32(tbn), np.empty(11, np.float32)
        assert self.L.oracle_fetch_surface(self.h, mat, u, v, _ptr(t), _ptr(out)) == 0
        return out


def frame_constants(desc):
    s = desc.fill(Scene())
    pv, lpv, sd = np.empty(16, np.float32), np.empty(16, np.float32), np.empty(3, np.float32)
    assert lib().oracle_frame_constants(C.byref(s), _ptr(pv), _ptr(lpv), _ptr(sd)) == 0
    return pv.reshape(4, 4), lpv.reshape(4, 4), sd   # [col][row]


def dir_from_rot(rot):
    r, out = _f32(rot), np.empty(3, np.float32)
    lib().oracle_dir_from_rot(_ptr(r), _ptr(out))
    return out


def outgoing_radiance(n, wo, wi, Li, base, metal, rough):
    a = [_f32(x) for x in (n, wo, wi, Li, base)]
    out = np.empty(3, np.float32)
    lib().oracle_outgoing_radiance(*[_ptr(x) for x in a], metal, rough, _ptr(out))
    return out


def tonemap(method, gamma, exposure, c):
    c, tm, out = _f32(c), np.empty(3, np.float32), np.empty(3, np.float32)
    lib().oracle_tonemap(method, gamma, exposure, _ptr(c), _ptr(tm), _ptr(out))
    return tm, out


def calculate_shadow(shadow_map, ls):
    ls = _f32(ls)
    if shadow_map is None:
        return lib().oracle_calculate_shadow(None, 0, _ptr(ls))
    m = _f32(shadow_map)
    return lib().oracle_calculate_shadow(_ptr(m), m.shape[0], _ptr(ls))


def to_unorm8(x):
    return lib().oracle_to_unorm8(float(x))


def through_half(x):
    return lib().oracle_through_half(float(x))


def hardware_threads():
    return lib().oracle_hardware_threads()
