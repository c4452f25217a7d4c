"""MI355X-native Hessian-free Newton-step solver (drop-in for the PCG /
curvature-matvec hot path of ltatzel/PyTorchHessianFree)."""

# Importing this package changes nothing in the process.  MIOpen settings that make
# stock fp32 convolutions reference-accurate (no Winograd, no split-K CK weight
# gradient), the solver find step and the shipped find-db are applied by
# ``configure()`` -- called by ``modelprep.prepare_model``, ``bench.py`` and the tests;
# see ``config.py`` for every switch.
from .config import configure, configured  # noqa: F401

from .cg import DampedCurvature, DiagonalPreconditioner, cg, storing_grid  # noqa: F401

__all__ = ["cg", "DampedCurvature", "DiagonalPreconditioner", "storing_grid", "configure"]
from .optimizer import HessianFree  # noqa: F401,E402
from .preconditioners import (  # noqa: F401,E402
    diag_EF_autograd,
    diag_EF_backpack,
    diag_EF_preconditioner,
    diag_to_preconditioner,
)
from .cg_backtracking import cg_backtracking, cg_efficient_backtracking  # noqa: F401,E402
from .linesearch import simple_linesearch  # noqa: F401,E402
from .utils import vector_to_parameter_list, vector_to_trainparams  # noqa: F401,E402

__all__ += ["HessianFree", "diag_EF_autograd", "diag_EF_backpack", "diag_EF_preconditioner",
            "diag_to_preconditioner", "cg_backtracking", "cg_efficient_backtracking",
            "simple_linesearch", "vector_to_parameter_list", "vector_to_trainparams"]
