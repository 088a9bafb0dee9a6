"""mi355pt — MI355X-native spectral path-tracing integrator (the hot path of
MatchaChoco010/toy-cpu-pathtracing behind a C ABI).  The directory name contains a hyphen, so
import it with importlib.import_module("toy-cpu-pathtracing_amd")."""
from . import assets, ffi, scenes  # noqa: F401
from .ffi import Product, make_camera, make_params  # noqa: F401

__version__ = "0.1.0"
