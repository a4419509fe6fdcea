#!/bin/bash
# compile csrc/wino44.hip alone with -save-temps and print register / scratch use (diagnostic)
rm -f /tmp/wino44-hip-amdgcn-amd-amdhsa-gfx950.s; cd /root/repo/dc_vic_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -fPIC -std=c++17 -Wno-unused-value -Wno-unused-result -I../../include -I. -x hip -c wino44.hip -o /tmp/wino44.o -save-temps=obj 2>&1 | grep -E "error|Error" | head -30
grep -E "^; (NumVgprs|NumAgprs|TotalNumVgprs|ScratchSize|Occupancy|codeLenInByte)" /tmp/wino44-hip-amdgcn-amd-amdhsa-gfx950.s | tail -12
