"""MI355X-native multigrid-preconditioned GCR hot path (SpMV + MG V-cycle + GCR orthogonalisation).

Drop-in for that one path of jing2li/MGPreconditionedGCR behind its Operator / Field / *_Param
interface: hand-written HIP (gfx950) kernels in libmgcr_hip.so behind the C ABI of
include/mgcr.h; this package is the Python mirror of the reference's host interface.
"""
from ._lib import MgcrError, finalize, init, lib  # noqa: F401
from .api import (Dense, DiracOp, Field, GCR, GCR_Param, HierarchicalSparse, MG, MG_Param, Mesh,  # noqa: F401
                  Operator, Sparse, gamma5, legacy_dense_gcr, read_data, set_option, stat, vec_double)
from . import problems  # noqa: F401
from .distributed import Comm, DistHierarchicalSparse, DistSparse, Plan  # noqa: F401
from . import experiments  # noqa: F401

__all__ = ["init", "finalize", "lib", "MgcrError", "Field", "Operator", "Sparse", "DiracOp",
           "HierarchicalSparse", "Dense", "GCR_Param", "GCR", "MG_Param", "MG", "Mesh", "gamma5", "vec_double", "read_data", "problems", "Comm", "Plan", "DistSparse", "DistHierarchicalSparse", "set_option", "stat"]
