#!/bin/bash
# Single-layer conv micro-benchmarks (Cin Cout H W N k reps) for the shapes that matter on the path; run on the GPU box:
#   bash tools/conv_shapes_bench.sh            # default dispatch
#   DCVIC_CONV_DMA=0 bash tools/conv_shapes_bench.sh   # generic kernel only (A/B)
python tools/conv_layer_bench.py 256 256 128 128 32 3 5     # VQGAN decoder 3x3 (DMA tap kernel)
python tools/conv_layer_bench.py 128 128 256 256 32 3 3
python tools/conv_layer_bench.py 512 512 32 32 32 3 10
python tools/conv_layer_bench.py 96 96 128 128 32 3 10      # ELIC 3x3 (96-channel DMA build)
python tools/conv_layer_bench.py 512 1536 32 32 32 1 10     # attention qkv (1x1 DMA GEMM)
python tools/conv_layer_bench.py 96 192 128 128 32 1 10     # ELIC bottleneck end (memory-bound 1x1)
python tools/conv_layer_bench.py 224 128 16 16 32 5 20      # CHARM 5x5 on 16x16 maps (async twins)
python tools/conv_layer_bench.py 128 32 16 16 32 3 20
python tools/conv_layer_bench.py 512 512 32 32 1 3 20       # N = 1 regime
python tools/conv_layer_bench.py 256 256 64 64 1 3 20
