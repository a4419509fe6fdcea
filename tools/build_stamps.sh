#!/bin/bash
# diagnostic build: conv3x3.hip with in-kernel cycle stamps (-DDCVIC_STAMPS) -> tools/libdcvic_stamps.so (not the product library)
set -e
cd "$(dirname "$0")/../dc_vic_amd/csrc"
OBJ=/tmp/dcvic_stamps_obj; mkdir -p $OBJ
F="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-value -Wno-unused-result -I../../include -I. -DDCVIC_STAMPS"
for s in conv conv3x3 conv_async conv_async16 conv1x1 gemm attn norm ew swin vq rate train; do /opt/rocm/bin/hipcc $F -x hip -c $s.hip -o $OBJ/$s.o & done
for s in error host_entropy; do /opt/rocm/bin/hipcc $F -c $s.cpp -o $OBJ/$s.o & done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/libdcvic_stamps.so $OBJ/*.o -lpthread
echo built tools/libdcvic_stamps.so
