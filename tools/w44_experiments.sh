#!/bin/bash
# timing experiments of the F(4x4) kernel (diagnostic builds, wrong results): one line per DCVIC_W44_DBG variant
cd "$(dirname "$0")/.."
SHAPE="${SHAPE:-256 256 128 128 32}"
echo "product:"; WINO_CHECK_F44=1 python tools/wino_check.py one $SHAPE 2>/dev/null | sed 's/.*| wino/wino/'
for B in "$@"; do
  echo "DBG=$B:"; DCVIC_LIB_PATH=tools/libdcvic_w44_dbg$B.so WINO_CHECK_F44=1 python tools/wino_check.py one $SHAPE 2>/dev/null | sed 's/.*| wino/wino/'
done
