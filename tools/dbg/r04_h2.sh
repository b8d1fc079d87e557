#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -m gpu 2>&1 | tail -4
python tools/dbg/dropin_ab.py 2>&1 | tail -3
