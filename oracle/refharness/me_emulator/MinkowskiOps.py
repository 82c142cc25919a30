"""Emulated `MinkowskiOps.to_sparse` (reference call site: models/convnextv2_sparse.py:11-13,199)."""
from MinkowskiEngine import _to_sparse as to_sparse  # noqa: F401
