"""Renderer: Python host-side mirror of Arctic::Renderer::Renderer over the C-ABI.

Same method names, argument meaning and error behaviour as the reference class
(src/renderer/renderer.hpp:100-125): init-at-construction, create_material,
create_mesh, create_hdri, update_lights, render_frame, resize, flush, cleanup.
The reference returns bool and logs; here a failed call raises ArcticError
carrying the C-ABI code and arctic_last_error().  All compute happens in
csrc/libarctic_hip.so (HIP, gfx950); numpy arrays are only the host buffers.
"""
import ctypes as C

import numpy as np

from . import binding
from .scene import (LIGHT_DTYPE, VERTEX_DTYPE, CCreateInfo, CScene, CSettings)


class ArcticError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"{binding.ERRORS.get(code, code)}: {message}")
        self.code = code


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Renderer:
    def __init__(self, width, height, shadow_size=4000, max_lights=16, device=0, row_begin=0, row_end=0, band_rows=0,
                 shard=(0, 1)):
        """Renderer(window, w, h) + init() (renderer.hpp:94-100); no window.  shadow_size defaults to
        ShadowMapPass::SIZE (shadow_map_pass.hpp:23), max_lights to MAX_NUM_POINT_LIGHTS (renderer.hpp:22)."""
        self.L = binding.lib()
        info = CCreateInfo(width, height, shadow_size, max_lights, device, row_begin, row_end, band_rows, shard[0], shard[1])
        err = C.create_string_buffer(512)
        self.h = self.L.arctic_create(C.byref(info), err, 512)
        if not self.h:
            raise ArcticError(-3 if b"no HIP device" in err.value else -2, err.value.decode())
        self.width, self.height, self.shadow_size, self.max_lights, self.device = width, height, shadow_size, max_lights, device
        self.row_begin, self.row_end = (row_begin, row_end) if row_end else (0, height)
        self.band_rows, self.shard = band_rows, shard

    # ---- helpers -------------------------------------------------------------------------------
    @property
    def rows(self):
        if self.band_rows:
            from .sharding import owned_rows
            return len(owned_rows(self.height, self.shard[0], self.shard[1], self.band_rows))
        return self.row_end - self.row_begin

    def _check(self, rc):
        if rc < 0:
            raise ArcticError(rc, self.L.arctic_last_error(self.h).decode())
        return rc

    @staticmethod
    def _scene(desc):
        return desc.fill(CScene())

    @staticmethod
    def _settings(settings):
        tm, gamma, exposure = settings
        return CSettings(int(tm), float(gamma), float(exposure))

    # ---- the reference surface -------------------------------------------------------------------
    def cleanup(self):
        if getattr(self, "h", None):
            self.L.arctic_destroy(self.h)
            self.h = None

    close = cleanup

    def __del__(self):
        self.cleanup()

    def resize(self, width, height):
        self._check(self.L.arctic_resize(self.h, width, height))
        self.width, self.height, self.row_begin, self.row_end = width, height, 0, height
        self.band_rows, self.shard = 0, (0, 1)

    def flush(self):
        self._check(self.L.arctic_flush(self.h))

    def set_stream(self, hip_stream):
        """enqueue everything on a caller-owned HIP stream given as an int handle, e.g. torch.cuda.current_stream().cuda_stream
        (0 = HIP's default stream, which is what torch normally runs on); None returns to the handle's private stream."""
        if hip_stream is None:
            self._check(self.L.arctic_use_own_stream(self.h))
        else:
            self._check(self.L.arctic_set_stream(self.h, C.c_void_p(hip_stream)))

    # ---- multi-GPU exchange steps (include/arctic_dist.h) -----------------------------------------------------------
    @staticmethod
    def comm_unique_id():
        """128 bytes from ncclGetUniqueId: one process calls it, every rank of the communicator gets the bytes."""
        L = binding.lib()
        buf, err = C.create_string_buffer(128), C.create_string_buffer(512)
        rc = L.arctic_comm_unique_id(buf, err, 512)
        if rc != 0:
            raise ArcticError(rc, err.value.decode())
        return buf.raw

    def comm_init(self, unique_id, rank, world):
        """collective: an RCCL communicator of `world` ranks owned by this handle (ncclCommInitRank) + the shard layouts."""
        assert len(unique_id) == 128
        self._check(self.L.arctic_comm_init(self.h, C.c_char_p(unique_id), rank, world))

    def comm_destroy(self):
        self._check(self.L.arctic_comm_destroy(self.h))

    def gather_frame(self, d_shard_ptr, d_frame_ptr, root=0):
        """collective, asynchronous: this rank's RGBA8 shard (device pointer; None = the handle's own output) to the root's
        row-major frame (device pointer; ignored on the other ranks)."""
        self._check(self.L.arctic_gather_frame(self.h, C.c_void_p(d_shard_ptr) if d_shard_ptr else None,
                                               C.c_void_p(d_frame_ptr) if d_frame_ptr else None, root))

    def assemble_frame(self, d_staging_ptr, d_frame_ptr, world, row_ranges=None):
        """the root's placement step alone: shards back to back in rank order -> the full frame."""
        rr = None if row_ranges is None else np.ascontiguousarray(row_ranges, dtype=np.uint32).reshape(-1)
        assert rr is None or rr.size == 2 * world
        self._check(self.L.arctic_assemble_frame(self.h, C.c_void_p(d_staging_ptr), C.c_void_p(d_frame_ptr), world, _ptr(rr)))

    def create_material(self, diffuse, normal, metal_rough):
        """three (h, w, 4) uint8 images; returns the material index."""
        d, n, m = (np.ascontiguousarray(t, dtype=np.uint8) for t in (diffuse, normal, metal_rough))
        for t in (d, n, m):
            if t.ndim != 3 or t.shape[2] != 4:
                raise ArcticError(-1, "create_material: images must be (h, w, 4) uint8")
        return self._check(self.L.arctic_create_material(self.h, _ptr(d), d.shape[1], d.shape[0], _ptr(n), n.shape[1], n.shape[0],
                                                         _ptr(m), m.shape[1], m.shape[0]))

    def create_mesh(self, vertices, indices, material_idx):
        v = np.ascontiguousarray(vertices, dtype=VERTEX_DTYPE)
        i = np.ascontiguousarray(indices, dtype=np.uint32).ravel()
        return self._check(self.L.arctic_create_mesh(self.h, _ptr(v), len(v), _ptr(i), len(i), int(material_idx)))

    def create_hdri(self, rgba32f):
        a = np.ascontiguousarray(rgba32f, dtype=np.float32)
        return self._check(self.L.arctic_create_hdri(self.h, _ptr(a), a.shape[1], a.shape[0]))

    def update_lights(self, lights):
        l = np.ascontiguousarray(lights, dtype=LIGHT_DTYPE)
        self._check(self.L.arctic_update_lights(self.h, _ptr(l) if len(l) else None, len(l)))

    def render_frame(self, desc, settings, out=None):
        """returns the (rows, width, 4) uint8 frame (this handle's row shard)."""
        s, st = self._scene(desc), self._settings(settings)
        if out is None:
            out = np.empty((self.rows, self.width, 4), np.uint8)
        self._check(self.L.arctic_render_frame(self.h, C.byref(s), C.byref(st), _ptr(out)))
        return out

    def render_frame_device(self, desc, settings, d_out_ptr):
        """frame into caller-owned device memory (int pointer, e.g. torch tensor .data_ptr())."""
        s, st = self._scene(desc), self._settings(settings)
        self._check(self.L.arctic_render_frame_device(self.h, C.byref(s), C.byref(st), C.c_void_p(d_out_ptr)))

    # ---- passes, timing, read-back -----------------------------------------------------------------
    def pass_shadow_map(self, desc):
        s = self._scene(desc)
        self._check(self.L.arctic_pass_shadow_map(self.h, C.byref(s)))

    def pass_gbuffer(self, desc):
        s = self._scene(desc)
        self._check(self.L.arctic_pass_gbuffer(self.h, C.byref(s)))

    def prepared_pass_shade(self, desc, settings):
        """returns shade(d_out_ptr): the same call as pass_shade with the C structs built once (per-frame host cost of a
        tight loop is then one ctypes call)."""
        s, st = self._scene(desc), self._settings(settings)
        fn, h, check, ps, pst = self.L.arctic_pass_shade, self.h, self._check, C.byref(s), C.byref(st)

        def shade(d_out_ptr=None, _keep=(s, st, desc)):
            check(fn(h, ps, pst, C.c_void_p(d_out_ptr) if d_out_ptr else None))
        return shade

    def pass_shade(self, desc, settings, d_out_ptr=None):
        s, st = self._scene(desc), self._settings(settings)
        self._check(self.L.arctic_pass_shade(self.h, C.byref(s), C.byref(st), C.c_void_p(d_out_ptr) if d_out_ptr else None))

    def post_process(self, hdr_rgba, settings, want_ldr=True):
        a = np.ascontiguousarray(hdr_rgba, dtype=np.float32)
        h, w = a.shape[:2]
        st = self._settings(settings)
        out = np.empty((h, w, 4), np.uint8)
        ldr = np.empty((h, w, 3), np.float32) if want_ldr else None
        self._check(self.L.arctic_post_process(self.h, _ptr(a), w, h, C.byref(st), _ptr(out), _ptr(ldr)))
        return out, ldr

    def time_shade(self, desc, settings, warmup=5, iters=20):
        s, st = self._scene(desc), self._settings(settings)
        ms = np.empty(iters, np.float32)
        self._check(self.L.arctic_time_shade(self.h, C.byref(s), C.byref(st), warmup, iters, _ptr(ms)))
        return ms

    def read_gbuffer(self, want=("attrs", "material", "depth", "tri")):
        n = (self.rows, self.width)
        attrs = np.empty(n + (18,), np.float32) if "attrs" in want else None
        mat = np.empty(n, np.uint32) if "material" in want else None
        depth = np.empty(n, np.float32) if "depth" in want else None
        tri = np.empty(n, np.uint32) if "tri" in want else None
        self._check(self.L.arctic_read_gbuffer(self.h, _ptr(attrs), _ptr(mat), _ptr(depth), _ptr(tri)))
        return attrs, mat, depth, tri

    def write_gbuffer(self, attrs, material):
        a = np.ascontiguousarray(attrs, dtype=np.float32)
        m = np.ascontiguousarray(material, dtype=np.uint32)
        if a.shape != (self.rows, self.width, 18) or m.shape != (self.rows, self.width):
            raise ArcticError(-1, f"write_gbuffer: expected {(self.rows, self.width, 18)}, got {a.shape}")
        self._check(self.L.arctic_write_gbuffer(self.h, _ptr(a), _ptr(m)))

    def read_shadow_map(self):
        d = np.empty((self.shadow_size, self.shadow_size), np.float32)
        self._check(self.L.arctic_read_shadow_map(self.h, _ptr(d)))
        return d

    def write_shadow_map(self, depth):
        d = np.ascontiguousarray(depth, dtype=np.float32)
        if d.shape != (self.shadow_size, self.shadow_size):
            raise ArcticError(-1, "write_shadow_map: wrong shape")
        self._check(self.L.arctic_write_shadow_map(self.h, _ptr(d)))

    def read_output(self, want=("ldr", "hdr", "rgba8")):
        n = (self.rows, self.width)
        ldr = np.empty(n + (3,), np.float32) if "ldr" in want else None
        hdr = np.empty(n + (3,), np.float32) if "hdr" in want else None
        rgba = np.empty(n + (4,), np.uint8) if "rgba8" in want else None
        self._check(self.L.arctic_read_output(self.h, _ptr(ldr), _ptr(hdr), _ptr(rgba)))
        return ldr, hdr, rgba

    def stats(self):
        s = np.zeros(16, np.uint64)
        self._check(self.L.arctic_stats(self.h, _ptr(s), 16))
        return s

    def tile_trace(self):
        """(tiles_y, tiles_x, 4) uint64 of the latest shading pass under set_option("tile_trace", 1): start, end (100 MHz
        reference clock), HW_ID | XCC_ID << 32, 1 = fast tile | shader-clock ticks << 8 (arctic_read_tile_trace)."""
        tx, ty = C.c_uint32(0), C.c_uint32(0)
        self._check(self.L.arctic_read_tile_trace(self.h, None, 0, C.byref(tx), C.byref(ty)))
        out = np.zeros((ty.value, tx.value, 4), np.uint64)
        self._check(self.L.arctic_read_tile_trace(self.h, _ptr(out), tx.value * ty.value, C.byref(tx), C.byref(ty)))
        return out

    def tile_order(self):
        """(order, classes): the dispatch order arctic_pass_gbuffer left for the shading pass -- one uint32 per strip of 4 tiles,
        ty << 16 | strip column -- and the (tiles_y, tiles_x) uint8 cost classes it was built from (arctic_read_tile_order)."""
        tx, ty = C.c_uint32(0), C.c_uint32(0)
        self._check(self.L.arctic_read_tile_order(self.h, None, None, 0, C.byref(tx), C.byref(ty)))
        order = np.zeros(((tx.value + 3) // 4) * ty.value, np.uint32)
        classes = np.zeros((ty.value, tx.value), np.uint8)
        self._check(self.L.arctic_read_tile_order(self.h, _ptr(order), _ptr(classes), tx.value * ty.value, C.byref(tx), C.byref(ty)))
        return order, classes

    def bin_counts(self, shadow_pass=False):
        """(blocks_y, blocks_x) uint32: work items per 16x16 block of the latest forward / shadow prepass drawn with block
        owners (arctic_read_bin_counts)."""
        bx, by = C.c_uint32(0), C.c_uint32(0)
        self._check(self.L.arctic_read_bin_counts(self.h, int(shadow_pass), None, 0, C.byref(bx), C.byref(by)))
        out = np.zeros((by.value, bx.value), np.uint32)
        self._check(self.L.arctic_read_bin_counts(self.h, int(shadow_pass), _ptr(out), bx.value * by.value, C.byref(bx), C.byref(by)))
        return out

    def cull_counts(self, shadow_pass=False):
        """(clusters, clusters skipped, vertex blocks, vertex blocks skipped) of the latest prepass that ran under debug bit 10
        (arctic_read_cull_counts)."""
        out = np.zeros(4, np.uint32)
        self._check(self.L.arctic_read_cull_counts(self.h, int(shadow_pass), _ptr(out)))
        return out

    def set_option(self, name, value):
        self._check(self.L.arctic_set_option(self.h, binding.OPTIONS[name], int(value)))


def frame_constants(desc):
    """proj_view, light_proj_view ([col][row]) and sun_dir exactly as the library builds them."""
    s = desc.fill(CScene())
    pv, lpv, sd = np.empty(16, np.float32), np.empty(16, np.float32), np.empty(3, np.float32)
    rc = binding.lib().arctic_frame_constants(C.byref(s), _ptr(pv), _ptr(lpv), _ptr(sd))
    if rc < 0:
        raise ArcticError(rc, "frame_constants")
    return pv.reshape(4, 4), lpv.reshape(4, 4), sd
