timeout 600 python -m pytest tests/test_gpu_shapes.py -x -q -m gpu -k "gemm_f32_big" 2>&1 | tail -5
python tools/gemm_f32_bench.py 14 15
