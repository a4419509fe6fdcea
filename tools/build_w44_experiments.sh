#!/bin/bash
# diagnostic builds of wino44.hip with -DDCVIC_W44_DBG=<bits> (timing experiments, WRONG results) -> tools/libdcvic_w44_dbg<bits>.so,
# never the product library.  Use: DCVIC_LIB_PATH=tools/libdcvic_w44_dbg4.so WINO_CHECK_F44=1 python tools/wino_check.py one 256 256 128 128 32
set -e
cd "$(dirname "$0")/../dc_vic_amd/csrc"
F="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-value -Wno-unused-result -I../../include -I."
for B in "$@"; do
  OBJ=/tmp/dcvic_w44exp_obj_$B; mkdir -p $OBJ
  cp _obj/*.o $OBJ/
  /opt/rocm/bin/hipcc $F -fno-slp-vectorize -DDCVIC_W44_DBG=$B -x hip -c wino44.hip -o $OBJ/wino44.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/libdcvic_w44_dbg$B.so $OBJ/*.o -lpthread
  echo built tools/libdcvic_w44_dbg$B.so
done
