"""`from flow_matching import CNF` of the reference (flow_matching.py:15) -> uspace_amd.flow_matching.CNF."""
from uspace_amd.flow_matching import CNF, CNFBase  # noqa: F401

__all__ = ["CNF"]
