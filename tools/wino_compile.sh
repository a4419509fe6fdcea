#!/bin/bash
# compile csrc/wino.hip alone with -save-temps and print register / scratch use (diagnostic)
cd /root/repo/dc_vic_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -fPIC -std=c++17 -Wno-unused-value -I../../include -I. -x hip -c wino.hip -o /tmp/wino.o -save-temps=obj 2>&1 | grep -v "warning\|^\s*$" | head -30
grep -E "NumVgprs|ScratchSize|Occupancy" /tmp/wino-hip-amdgcn-amd-amdhsa-gfx950.s | tail -4
awk '/^_Z19conv3x3_wino_kernel9ConvKArgs:/,/s_endpgm/' /tmp/wino-hip-amdgcn-amd-amdhsa-gfx950.s > /tmp/wk.s
