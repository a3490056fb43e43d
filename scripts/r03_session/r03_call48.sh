for d in 0 32 1 33 2 4; do echo "BASIC_CONV_DEBUG=$d"; BASIC_CONV_DEBUG=$d timeout -k 10 120 python scripts/conv_layer_bench.py g_a.1 2>&1 | grep -v amdgpu; done
