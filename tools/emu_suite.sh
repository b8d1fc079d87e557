#!/bin/bash
# The whole `-m gpu` suite on the CPU SIMT emulation (emu/README.md): hours of CPU, no GPU.  -> profiles/rNN_emu_gputest.txt
# usage: bash tools/emu_suite.sh [tag] [pytest args...]
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
T=${1:-r06}; shift
make -C emu -j8 > /dev/null || exit 1
OUT=profiles/${T}_emu_gputest.txt
echo "HEAD=$(git rev-parse --short HEAD)  DGCNN_EMU=1 DGCNN_RACE_B=600 python -m pytest tests -m gpu -q -n $(nproc) --timeout 2400 -rfs $*   (CPU SIMT emulation, emu/; $(nproc) cores)" > $OUT
DGCNN_EMU=1 DGCNN_RACE_B=600 python -m pytest tests -m gpu -q -n $(nproc) --timeout 2400 -rfs "$@" 2>&1 | tail -80 >> $OUT
tail -5 $OUT
