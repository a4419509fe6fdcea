for shape in "128 128 256 256 32 3" "256 256 128 128 32 3" "512 512 32 32 32 3" "256 256 64 64 32 3" "192 96 128 128 32 3"; do python tools/conv_layer_bench.py $shape 20 | tail -1; done
