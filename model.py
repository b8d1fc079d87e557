"""Drop-in for the reference's ``model.py``: ``from model import Model`` (/root/reference/train.py:13).

Same constructor and ``forward(data)`` contract as /root/reference/model.py:9-45; the implementation
is the MI355X-native HIP path in :mod:`dgcnn_amd` (no PyG, no torch-scatter/sparse, no Triton).
"""
from dgcnn_amd.model import Model  # noqa: F401

__all__ = ["Model"]
