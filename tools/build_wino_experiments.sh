#!/bin/bash
# diagnostic build: wino.hip with its timing-experiment variants (-DDCVIC_WINO_EXPERIMENTS: DCVIC_WINO_DEBUG selects kernels that
# skip the barrier / transform / DMA / waits and give WRONG results) -> tools/libdcvic_wino_exp.so, never the product library.
# Use: DCVIC_LIB_PATH=tools/libdcvic_wino_exp.so DCVIC_WINO_DEBUG=64 python tools/wino_check.py one 256 256 128 128 32
set -e
cd "$(dirname "$0")/../dc_vic_amd/csrc"
OBJ=/tmp/dcvic_winoexp_obj; mkdir -p $OBJ
F="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-value -Wno-unused-result -I../../include -I."
/opt/rocm/bin/hipcc $F -fno-slp-vectorize -DDCVIC_WINO_EXPERIMENTS -x hip -c wino.hip -o $OBJ/wino.o
cp _obj/*.o $OBJ/ 2>/dev/null || true
/opt/rocm/bin/hipcc $F -fno-slp-vectorize -DDCVIC_WINO_EXPERIMENTS -x hip -c wino.hip -o $OBJ/wino.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/libdcvic_wino_exp.so $OBJ/*.o -lpthread
echo built tools/libdcvic_wino_exp.so
