"""dgcnn_amd -- MI355X-native (gfx950) DGCNN forward+backward hot path.

Public surface (mirrors what the reference exposes for this path):
  Model            -- nn.Module drop-in of /root/reference/model.py:9-45
  Batch, collate   -- the input container the model consumes (PyG Batch stand-in)
  Trainer          -- the per-batch step of /root/reference/train.py:27-66 on fused kernels
  optim.Adam       -- torch.optim.Adam drop-in that updates the flat parameter buffer with one kernel (train.py:11,99)
  device_data      -- DeviceDataset / DeviceLoader: dataset resident in HBM, batches assembled on the device
  tudataset        -- TU-format reader + Indegree + fold files + GraphLoader (train.py:81-109, utils.py:18-33)
  cli              -- the 10-fold driver (train.py:69-148), also reachable as ``python train.py``
"""
from .batch import Batch, Graph, collate  # noqa: F401

__all__ = ["Batch", "Graph", "collate", "Model", "Trainer"]


def __getattr__(name):
    # lazy: importing the package must work on a box without the built library (CPU CI)
    if name == "Model":
        from .model import Model
        return Model
    if name == "Trainer":
        from .train import Trainer
        return Trainer
    raise AttributeError(name)
