#!/bin/bash
# is a grouped (mu + sigma in one launch) CHARM conv worth it?  time(N=64) vs 2 x time(N=32) on the three CHARM conv shapes
for shape in "256 224 16 16 N 5" "224 128 16 16 N 5" "128 32 16 16 N 3"; do
  for n in 32 64; do
    s=${shape/N/$n}
    python tools/conv_layer_bench.py $s 50 | tail -1
  done
done
