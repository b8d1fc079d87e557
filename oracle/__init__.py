"""ORACLE package -- test infrastructure only (CPU restatements of the reference hot path).

PARITY UNPINNED: the reference's arithmetic lives in PyTorch Geometric, which is absent
here and unpinned upstream; the reference ships no tests/golden vectors.  See
``ref_ops.py`` / ``ref_dense.py`` headers and DESIGN.md.

Import rule: only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import anything from here.  ``dgcnn_amd`` never does.
"""
