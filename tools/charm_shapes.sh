for a in 0 1; do
export DCVIC_CONV_ASYNC16=$a
echo "== async16 $a"
python tools/conv_layer_bench.py 224 128 16 16 32 5 20
python tools/conv_layer_bench.py 128 224 16 16 32 5 20
python tools/conv_layer_bench.py 128 32 16 16 32 3 20
python tools/conv_layer_bench.py 192 96 16 16 32 1 20
python tools/conv_layer_bench.py 128 128 32 32 32 3 10
python tools/conv_layer_bench.py 512 512 32 32 1 3 20
python tools/conv_layer_bench.py 256 256 64 64 1 3 20
done
