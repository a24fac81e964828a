set -x
timeout 900 python -m pytest tests/test_gpu_shapes.py -x -q -m gpu -k "gemm_f32_big" 2>&1 | tail -15
timeout 900 python -m pytest tests/test_gpu.py -x -q -m gpu -k "fp32" 2>&1 | tail -15
timeout 600 python tools/bench_rows.py --rows convtasnet_fp32 2>&1 | tail -3
