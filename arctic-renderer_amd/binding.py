"""ctypes declarations of the C-ABI (include/arctic_hip.h -> csrc/libarctic_hip.so).

This is the binding a Python host would use; the C++ equivalent is host/renderer.hpp.
The library is never built implicitly and there is no fallback: if the shared
object is missing, lib() raises with the build command.
"""
import ctypes as C
import os
import re

from .scene import CCreateInfo, CScene, CSettings

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ARCTIC_HIP_LIBRARY") or os.path.join(HERE, "csrc", "libarctic_hip.so")   # the override is for A/B timing of two builds (tools/experiments)
HEADER_PATH = os.path.join(os.path.dirname(HERE), "include", "arctic_hip.h")

OPTIONS = {"keep_float_output": 1, "count_light_evals": 2, "culling": 3, "debug": 4, "hdr16": 6, "shadow_cache": 9, "visbuffer": 10, "item_table_floor": 11, "light_path": 12, "markers": 13, "shadow_sharded": 14, "frames_in_flight": 15, "tiles_per_wave": 16, "tile_trace": 17, "raster_owner": 18, "tile_order": 19, "order_tail": 20, "sampler": 21, "texture_tiling": 22, "cluster_cull": 23, "small_triangles": 24}
ERRORS = {-1: "ARCTIC_E_INVALID", -2: "ARCTIC_E_DEVICE", -3: "ARCTIC_E_NO_DEVICE", -4: "ARCTIC_E_STATE", -5: "ARCTIC_E_CAPACITY"}

_vp, _u32, _u64, _i32, _i64 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int32, C.c_int64
_scene, _settings = C.POINTER(CScene), C.POINTER(CSettings)

# every entry point of include/arctic_hip.h: name -> (restype, argtypes)
SIGNATURES = {
    "arctic_create": (_vp, [C.POINTER(CCreateInfo), C.c_char_p, _u64]),
    "arctic_destroy": (None, [_vp]),
    "arctic_last_error": (C.c_char_p, [_vp]),
    "arctic_resize": (_i32, [_vp, _u32, _u32]),
    "arctic_flush": (_i32, [_vp]),
    "arctic_set_stream": (_i32, [_vp, _vp]),
    "arctic_use_own_stream": (_i32, [_vp]),
    "arctic_create_material": (_i32, [_vp, _vp, _u32, _u32, _vp, _u32, _u32, _vp, _u32, _u32]),
    "arctic_create_mesh": (_i32, [_vp, _vp, _u64, _vp, _u64, _u64]),
    "arctic_update_lights": (_i32, [_vp, _vp, _u64]),
    "arctic_create_hdri": (_i32, [_vp, _vp, _u32, _u32]),
    "arctic_render_frame": (_i32, [_vp, _scene, _settings, _vp]),
    "arctic_render_frame_device": (_i32, [_vp, _scene, _settings, _vp]),
    "arctic_pass_shadow_map": (_i32, [_vp, _scene]),
    "arctic_pass_gbuffer": (_i32, [_vp, _scene]),
    "arctic_pass_shade": (_i32, [_vp, _scene, _settings, _vp]),
    "arctic_post_process": (_i32, [_vp, _vp, _u32, _u32, _settings, _vp, _vp]),
    "arctic_time_shade": (_i32, [_vp, _scene, _settings, _u32, _u32, _vp]),
    "arctic_read_gbuffer": (_i32, [_vp, _vp, _vp, _vp, _vp]),
    "arctic_write_gbuffer": (_i32, [_vp, _vp, _vp]),
    "arctic_read_shadow_map": (_i32, [_vp, _vp]),
    "arctic_write_shadow_map": (_i32, [_vp, _vp]),
    "arctic_read_output": (_i32, [_vp, _vp, _vp, _vp]),
    "arctic_frame_constants": (_i32, [_scene, _vp, _vp, _vp]),
    "arctic_stats": (_i32, [_vp, _vp, _u32]),
    "arctic_read_tile_trace": (_i32, [_vp, _vp, _u64, _vp, _vp]),
    "arctic_read_tile_order": (_i32, [_vp, _vp, _vp, _u64, _vp, _vp]),
    "arctic_read_bin_counts": (_i32, [_vp, _i32, _vp, _u64, _vp, _vp]),
    "arctic_read_cull_counts": (_i32, [_vp, _i32, _vp]),
    "arctic_owner_grid": (_i32, [_u32, _u32, _u32, _u32, _u32, _u32, _u32, _vp]),
    "arctic_owner_visit": (_i32, [_u32, _u32, _u32, _u32, _u32, _u32, _u32, _u32, _vp, _vp]),
    "arctic_set_option": (_i32, [_vp, _u32, _i64]),
    "arctic_version": (_i32, []),
    # include/arctic_dist.h: the multi-GPU exchange steps
    "arctic_comm_unique_id": (_i32, [_vp, _vp, _u64]),
    "arctic_comm_init": (_i32, [_vp, _vp, _i32, _i32]),
    "arctic_comm_destroy": (_i32, [_vp]),
    "arctic_gather_frame": (_i32, [_vp, _vp, _vp, _i32]),
    "arctic_assemble_frame": (_i32, [_vp, _vp, _vp, _u32, _vp]),
    # ... and their plan as pure host functions (no device needed)
    "arctic_exchange_plan": (_i32, [_u32, _u32, _u32, _u32, _vp, _vp, _vp, _vp]),
    "arctic_exchange_row_source": (_i32, [_u32, _u32, _u32, _u32, _vp, _vp, _vp]),
    "arctic_exchange_transfers": (_i32, [_u32, _u32, _i32, _i32, _vp, _vp, _vp, _u32]),
}


class CTransfer(C.Structure):
    """ArcticTransfer (include/arctic_dist.h)"""
    _fields_ = [("peer", C.c_int32), ("is_send", C.c_int32), ("staging_offset", C.c_uint64), ("bytes", C.c_uint64)]


def header_symbols():
    """names of every function include/arctic_hip.h and include/arctic_dist.h declare."""
    text = open(HEADER_PATH).read() + open(os.path.join(os.path.dirname(HEADER_PATH), "arctic_dist.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(arctic_[a-z0-9_]+)\s*\(", text)))


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as e; e.build()'` "
                               f"(make -C arctic-renderer_amd/csrc). There is no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        missing = [n for n in header_symbols() if not hasattr(L, n)]
        older = os.environ.get("ARCTIC_HIP_LIBRARY_OLDER") == "1"   # A/B tools only: an earlier round's build, which lacks this round's entry points
        if missing and not older:
            raise RuntimeError(f"libarctic_hip.so does not export {missing}")
        for name, (res, args) in SIGNATURES.items():
            if older and name in missing:
                continue
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib
