"""Counterpart of the reference's ``train.py`` (10-fold cross-validation driver) on the HIP path.
See :mod:`dgcnn_amd.cli`; same flags as /root/reference/train.py:17-25, no visdom."""
from dgcnn_amd.cli import main

if __name__ == "__main__":
    main()
