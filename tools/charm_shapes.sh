for d in 0 1; do export DCVIC_CONV_DMA=$d; echo "== dma $d"
python tools/conv_layer_bench.py 96 96 128 128 32 3 10
python tools/conv_layer_bench.py 96 96 64 64 32 3 10
python tools/conv_layer_bench.py 192 192 64 64 32 3 10
done
