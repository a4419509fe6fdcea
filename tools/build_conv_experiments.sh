#!/bin/bash
# diagnostic build: conv.hip with DCVIC_FORCE_CLS / DCVIC_FORCE_P overrides of the tile-variant choice (-DDCVIC_CONV_EXPERIMENTS)
# -> tools/libdcvic_conv_exp.so (never the product library).  Use: DCVIC_LIB_PATH=tools/libdcvic_conv_exp.so DCVIC_FORCE_CLS=1 DCVIC_FORCE_P=64 python tools/conv_layer_bench.py 224 128 16 16 32 5
set -e
cd "$(dirname "$0")/../dc_vic_amd/csrc"
OBJ=/tmp/dcvic_convexp_obj; mkdir -p $OBJ
cp _obj/*.o $OBJ/
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-value -Wno-unused-result -I../../include -I. -DDCVIC_CONV_EXPERIMENTS -x hip -c conv.hip -o $OBJ/conv.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/libdcvic_conv_exp.so $OBJ/*.o -lpthread
echo built tools/libdcvic_conv_exp.so
