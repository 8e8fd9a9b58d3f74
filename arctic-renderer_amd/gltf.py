"""ctypes binding of the glTF scene-loader stand-in (include/arctic_gltf.h, host/gltf_loader.cpp): what App::load_scene
(reference src/app.cpp:173-385) produces for a glTF file -- materials, meshes, objects -- as numpy arrays, ready for any
object with the Renderer surface (the HIP binding or the CPU oracle).  Host-side data loading only."""
import ctypes as C
import os
import subprocess

import numpy as np

from .scene import OBJECT_DTYPE, VERTEX_DTYPE

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "host", "libarctic_gltf.so")
_lib = None


def build(force=False):
    src = os.path.join(HERE, "host", "gltf_loader.cpp")
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", os.path.join(HERE, "host"), "libarctic_gltf.so"], stdout=subprocess.DEVNULL)
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is not built: run __graft_entry__.build()")
        L = C.CDLL(LIB_PATH)
        vp, u64, u32p, u64p = C.c_void_p, C.c_uint64, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)
        L.arctic_gltf_load.restype, L.arctic_gltf_load.argtypes = vp, [C.c_char_p, C.c_char_p, u64]
        L.arctic_gltf_free.restype, L.arctic_gltf_free.argtypes = None, [vp]
        for f in ("arctic_gltf_material_count", "arctic_gltf_mesh_count", "arctic_gltf_object_count"):
            getattr(L, f).restype, getattr(L, f).argtypes = u64, [vp]
        L.arctic_gltf_material_image.restype = C.c_int
        L.arctic_gltf_material_image.argtypes = [vp, u64, C.c_int, C.POINTER(vp), u32p, u32p]
        L.arctic_gltf_mesh.restype = C.c_int
        L.arctic_gltf_mesh.argtypes = [vp, u64, C.POINTER(vp), u64p, C.POINTER(vp), u64p, u64p]
        L.arctic_gltf_objects.restype, L.arctic_gltf_objects.argtypes = vp, [vp]
        L.arctic_gltf_upload.restype, L.arctic_gltf_upload.argtypes = C.c_int, [vp, vp]
        L.arctic_png_decode.restype = vp
        L.arctic_png_decode.argtypes = [C.c_char_p, u64, u32p, u32p, C.c_char_p, u64]
        L.arctic_png_free.restype, L.arctic_png_free.argtypes = None, [vp]
        _lib = L
    return _lib


class GltfScene:
    """materials: list of (diffuse, normal, metal_rough) uint8 (h, w, 4); meshes: list of (vertices, indices, material);
    objects: OBJECT_DTYPE array -- the same three things scenes.SyntheticScene carries."""

    def __init__(self, materials, meshes, objects):
        self.materials, self.meshes, self.objects = materials, meshes, objects

    def upload(self, renderer):
        for d, n, m in self.materials:
            renderer.create_material(d, n, m)
        for v, i, mat in self.meshes:
            renderer.create_mesh(v, i, mat)
        return renderer


def load(path):
    L = lib()
    err = C.create_string_buffer(512)
    h = L.arctic_gltf_load(os.fsencode(path), err, 512)
    if not h:
        raise ValueError(err.value.decode())
    try:
        materials = []
        for i in range(L.arctic_gltf_material_count(h)):
            imgs = []
            for k in range(3):
                p, w, hh = C.c_void_p(), C.c_uint32(), C.c_uint32()
                assert L.arctic_gltf_material_image(h, i, k, C.byref(p), C.byref(w), C.byref(hh)) == 0
                imgs.append(np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), (hh.value, w.value, 4)).copy())
            materials.append(tuple(imgs))
        meshes = []
        for i in range(L.arctic_gltf_mesh_count(h)):
            pv, pi, nv, ni, mat = C.c_void_p(), C.c_void_p(), C.c_uint64(), C.c_uint64(), C.c_uint64()
            assert L.arctic_gltf_mesh(h, i, C.byref(pv), C.byref(nv), C.byref(pi), C.byref(ni), C.byref(mat)) == 0
            v = np.frombuffer(C.string_at(pv, nv.value * VERTEX_DTYPE.itemsize), dtype=VERTEX_DTYPE).copy()
            ix = np.frombuffer(C.string_at(pi, ni.value * 4), dtype=np.uint32).copy()
            meshes.append((v, ix, int(mat.value)))
        n_obj = L.arctic_gltf_object_count(h)
        objects = (np.frombuffer(C.string_at(L.arctic_gltf_objects(h), n_obj * OBJECT_DTYPE.itemsize), dtype=OBJECT_DTYPE).copy()
                   if n_obj else np.zeros(0, OBJECT_DTYPE))
        return GltfScene(materials, meshes, objects)
    finally:
        L.arctic_gltf_free(h)


def png_decode(data):
    L = lib()
    w, h, err = C.c_uint32(), C.c_uint32(), C.create_string_buffer(256)
    p = L.arctic_png_decode(data, len(data), C.byref(w), C.byref(h), err, 256)
    if not p:
        raise ValueError(err.value.decode())
    try:
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), (h.value, w.value, 4)).copy()
    finally:
        L.arctic_png_free(p)
