#!/bin/bash
# A/B builds of libdgcnn_hip.so with extra -D flags (measurement only): tools/build_variant.sh <name> "<flags>"
# -> dgcnn_amd/variants/lib_<name>.so ; select at run time with DGCNN_HIP_LIB=<path>
set -e
NAME=$1; FLAGS=$2
R=$(cd "$(dirname "$0")/.." && pwd)
D=$R/dgcnn_amd/variants/obj_$NAME; mkdir -p $D
for f in $(cd $R/dgcnn_amd/csrc && ls *.hip | sed "s/\.hip$//"); do      # every translation unit of the Makefile
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -fno-gpu-rdc $FLAGS \
     -c $R/dgcnn_amd/csrc/$f.hip -o $D/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $D/*.o -o $R/dgcnn_amd/variants/lib_$NAME.so
rm -rf $D
echo built $R/dgcnn_amd/variants/lib_$NAME.so
