"""MI355X-native Hessian-free Newton-step solver (drop-in for the PCG /
curvature-matvec hot path of ltatzel/PyTorchHessianFree)."""

from .cg import DampedCurvature, DiagonalPreconditioner, cg, storing_grid  # noqa: F401

__all__ = ["cg", "DampedCurvature", "DiagonalPreconditioner", "storing_grid"]
