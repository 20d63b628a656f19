#!/bin/bash
# round 6, call 14: k_supports2 with the adjacency staged once per LARGE tile; the narrow test with its restatement check
python3 tools/run_general.py mc900 20 1 256 1 3 2>/dev/null | tee gpurun_out/r06_poly_families2.txt
python3 tools/run_general.py er200x0.1 20 2 500 1 3 2>/dev/null | tee -a gpurun_out/r06_poly_families2.txt
python3 tools/run_general.py er200x0.1 20 2 64 1 3 2>/dev/null | tee -a gpurun_out/r06_poly_families2.txt
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_general.py tests/test_gpu_api.py -x -q -p no:cacheprovider -k "supports2 or three_support or narrow or chebyshev or cheb or poly or all_models or shape_coverage" 2>&1 | tail -4 | tee gpurun_out/r06_gpu14_tests.txt
