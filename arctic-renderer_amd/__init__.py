"""arctic-renderer_amd: MI355X-native forward PBR shading path.

Host-side mirror of the reference's Renderer surface (src/renderer/renderer.hpp:100-125)
over the C-ABI of csrc/ (include/arctic_hip.h -> libarctic_hip.so, hand-written HIP
for gfx950).  The directory name has a hyphen, so it is imported through
__graft_entry__.load_package() under the module name `arctic_renderer_amd`.
"""
from . import scene, scenes  # noqa: F401
from .scene import SceneDesc, TM_ACES, TM_EXPOSURE, TM_REINHARD  # noqa: F401


def __getattr__(name):
    # the HIP binding is loaded lazily so that pure-host helpers (scene generator,
    # row partitioning) import without the shared library; using Renderer without
    # it fails loudly -- there is no CPU fallback.
    if name in ("Renderer", "ArcticError", "binding", "renderer"):
        import importlib
        mod = importlib.import_module(".renderer", __name__)
        if name == "renderer":
            return mod
        if name == "binding":
            return importlib.import_module(".binding", __name__)
        return getattr(mod, name)
    raise AttributeError(name)
