"""POD scene types of the boundary, as numpy dtypes + ctypes structs.

Byte-compatible with include/arctic_hip.h, i.e. with the reference's
src/renderer/scene.hpp:20-110 (Camera, Vertex, Object, DirectionalLight,
PointLight, Scene, Settings).  Pure host-side data description: nothing here
computes anything on the hot path.
"""
import ctypes as C

import numpy as np

# scene.hpp:40-47 Vertex (56 B)
VERTEX_DTYPE = np.dtype([("position", "<f4", 3), ("normal", "<f4", 3), ("tangent", "<f4", 3),
                         ("bitangent", "<f4", 3), ("tex_coords", "<f4", 2)])
# scene.hpp:69-73 Object {mat4 trs (glm column-major); size_t mesh_idx}
OBJECT_DTYPE = np.dtype([("trs", "<f4", 16), ("mesh_idx", "<u8")])
# scene.hpp:88-94 PointLight (32 B)
LIGHT_DTYPE = np.dtype([("position", "<f4", 3), ("padding0", "<u4"), ("color", "<f4", 3), ("padding1", "<u4")])
assert VERTEX_DTYPE.itemsize == 56 and OBJECT_DTYPE.itemsize == 72 and LIGHT_DTYPE.itemsize == 32

TM_REINHARD, TM_EXPOSURE, TM_ACES = 0, 1, 2


class CCamera(C.Structure):
    _fields_ = [("eye", C.c_float * 3), ("rotation", C.c_float * 2), ("aspect", C.c_float),
                ("fov_y", C.c_float), ("z_near_far", C.c_float * 2)]


class CDirectionalLight(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("rotation", C.c_float * 2), ("color", C.c_float * 3)]


class CScene(C.Structure):
    _fields_ = [("camera", CCamera), ("ambient", C.c_float), ("sun", CDirectionalLight),
                ("point_lights", C.c_void_p), ("n_point_lights", C.c_uint64),
                ("objects", C.c_void_p), ("n_objects", C.c_uint64)]


class CSettings(C.Structure):
    _fields_ = [("tm_method", C.c_int32), ("gamma", C.c_float), ("exposure", C.c_float)]


class CCreateInfo(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("shadow_size", C.c_uint32),
                ("max_lights", C.c_uint32), ("device", C.c_int32), ("row_begin", C.c_uint32),
                ("row_end", C.c_uint32), ("band_rows", C.c_uint32), ("shard_index", C.c_uint32),
                ("shard_count", C.c_uint32)]


class SceneDesc:
    """Scene (scene.hpp:96-103) on the Python side: owns the numpy arrays the C
    struct points into and fills any ctypes struct of the Scene layout."""

    def __init__(self, camera, ambient, sun, objects, point_lights=None):
        # camera = dict(eye, rotation, aspect, fov_y, z_near_far); sun = dict(position, rotation, color)
        self.camera, self.ambient, self.sun = dict(camera), float(ambient), dict(sun)
        self.objects = np.ascontiguousarray(objects, dtype=OBJECT_DTYPE)
        self.point_lights = (np.zeros(0, LIGHT_DTYPE) if point_lights is None
                             else np.ascontiguousarray(point_lights, dtype=LIGHT_DTYPE))

    def fill(self, s):
        s.camera.eye[:] = [float(x) for x in self.camera["eye"]]
        s.camera.rotation[:] = [float(x) for x in self.camera["rotation"]]
        s.camera.aspect = float(self.camera["aspect"])
        s.camera.fov_y = float(self.camera["fov_y"])
        s.camera.z_near_far[:] = [float(x) for x in self.camera["z_near_far"]]
        s.ambient = self.ambient
        s.sun.position[:] = [float(x) for x in self.sun["position"]]
        s.sun.rotation[:] = [float(x) for x in self.sun["rotation"]]
        s.sun.color[:] = [float(x) for x in self.sun["color"]]
        s.point_lights = self.point_lights.ctypes.data if len(self.point_lights) else None
        s.n_point_lights = len(self.point_lights)
        s.objects = self.objects.ctypes.data if len(self.objects) else None
        s.n_objects = len(self.objects)
        return s


def make_objects(items):
    """items: iterable of (trs 4x4 in math (row, col) convention or flat glm order, mesh_idx)."""
    out = np.zeros(len(items), OBJECT_DTYPE)
    for i, (trs, mesh) in enumerate(items):
        t = np.asarray(trs, np.float32)
        # a 4x4 given as math matrix M[row][col] is stored column-major like glm
        out[i]["trs"] = t.T.reshape(16) if t.shape == (4, 4) else t.reshape(16)
        out[i]["mesh_idx"] = mesh
    return out


def make_lights(positions, colors):
    positions, colors = np.asarray(positions, np.float32), np.asarray(colors, np.float32)
    out = np.zeros(len(positions), LIGHT_DTYPE)
    if len(positions):
        out["position"], out["color"] = positions, colors
    return out


def translation(x, y, z):
    m = np.eye(4, dtype=np.float32)
    m[:3, 3] = (x, y, z)
    return m


def scaling(x, y, z):
    return np.diag(np.array([x, y, z, 1], np.float32))


def rotation_y(deg):
    a = np.float32(np.deg2rad(deg))
    c, s = np.float32(np.cos(a)), np.float32(np.sin(a))
    m = np.eye(4, dtype=np.float32)
    m[0, 0], m[0, 2], m[2, 0], m[2, 2] = c, s, -s, c
    return m
