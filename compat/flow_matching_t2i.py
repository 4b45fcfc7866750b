"""`from flow_matching_t2i import CNF` of the reference (flow_matching_t2i.py:15) -> uspace_amd.flow_matching_t2i.CNF."""
from uspace_amd.flow_matching_t2i import CNF  # noqa: F401

__all__ = ["CNF"]
