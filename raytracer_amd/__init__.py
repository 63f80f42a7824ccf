"""MI355X (gfx950) wavefront path-tracing core: a drop-in for the CPU
integrator of kurono267/Raytracer behind a C-ABI (include/rt_hip.h)."""
from .api import Context, Node, RtError, load_library, unpack_slabs  # noqa: F401
from . import scenes  # noqa: F401
