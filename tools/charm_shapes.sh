python tools/conv_layer_bench.py 96 192 128 128 32 1 10
python tools/conv_layer_bench.py 192 96 128 128 32 1 10
python tools/conv_layer_bench.py 512 1536 32 32 32 1 10
python tools/conv_layer_bench.py 512 512 32 32 32 1 10
python tools/conv_layer_bench.py 256 128 256 256 32 1 5
python tools/conv_layer_bench.py 448 256 128 128 32 1 5
