"""MI355X-native Laplacian-solve backend for Circuitscape (hot path only).

The directory is named ``circuitscape.jl_amd`` after the reference repository; because of the dot it is imported
through the shim module ``circuitscape_jl_amd`` at the repository root.
"""
from . import lib  # noqa: F401
