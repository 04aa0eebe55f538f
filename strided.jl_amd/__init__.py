"""strided_jl_amd -- MI355X-native implementation of Strided.jl's fused strided map / map-reduce
hot path (`_mapreduce_fuse!` -> `_mapreduce_kernel!`, /root/reference/src/mapreduce.jl).

The package directory is named ``strided.jl_amd`` (not a Python identifier); import it as
``strided_jl_amd`` through the alias module at the repository root.

Layout:  csrc/ = hand-written HIP kernels for gfx950 + the C ABI (include/strided_hip.h);
the modules here are the host-side mirror of the reference's operator interface
(StridedView, broadcast lowering, map!/mapreduce fronts) that ends in the C ABI.
"""
from . import _lib  # noqa: F401
from ._lib import Plan, Sequence, Stream, StridedHIPError, UnsupportedOnDevice, build, get_option, overlap, set_option  # noqa: F401
from .stridedview import DimensionMismatch, StridedView, isstrided, sreshape, sview  # noqa: F401
from . import fn  # noqa: F401
from .broadcast import (Broadcasted, Ref, broadcast_shape, capturestridedargs, copyto_,  # noqa: F401
                        make_capture, materialize, promoteshape, promoteshape1)
from .mapreduce import (Array, _mapreduce, _mapreduce_fuse_, _mapreducedim_, adjoint_,  # noqa: F401
                        build_problem, conj_, copy, copy_, make_plan, map, map_, mapreduce,
                        mapreducedim_, maximum, minimum, permutedims_, prod, sum, transpose_)
from .linalg import axpby_, axpy_, lmul_, mul_, rmul_  # noqa: F401

__version__ = "0.1.0"
