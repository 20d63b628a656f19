#!/bin/bash
# round 6, call 10: where padding a narrow deep stack onto k_big / k_big2 pays (batch size), and the [I, L, L.L] call by kernel family
for shape in "mc900 20 20 32 16" "mc900 20 20 64 16" "mc900 20 20 128 16" "mc900 20 4 64 16" "mc900 20 4 128 16" "mc900 20 20 64 8" "mc900 20 20 256 8" "mc1500 10 20 64 16" "mc1500 10 20 128 16" "mc1500 10 20 256 16" "mc1500 10 4 256 16" "er600x0.02 20 20 512 16" "er600x0.02 20 20 1024 16"; do
  python3 tools/run_general.py $shape 2>/dev/null | grep "kernels per call" | sed 's/^/padded:   /'
  DGCN_OPTIONS="big=0,big2=0" python3 tools/run_general.py $shape 2>/dev/null | grep "kernels per call" | sed 's/^/layered:  /'
done | tee gpurun_out/r06_narrow_batch_sizes.txt
python3 tools/run_general.py mc900 20 1 256 1 3 2>/dev/null | tee gpurun_out/r06_poly_families.txt
python3 tools/run_general.py er200x0.1 20 2 500 1 3 2>/dev/null | tee -a gpurun_out/r06_poly_families.txt
for g in 16 64; do
  echo "iterative mc900 c16 l20, $g graphs, padded:"; python3 tools/run_iterative.py --family mc --n 900 --p 0.03 --graphs $g --layers 20 --hidden 16 --host 0 2>/dev/null | grep '^{"solver' | cut -c1-150
  echo "iterative mc900 c16 l20, $g graphs, layered (big = 0):"; DGCN_OPTIONS="big=0,big2=0" python3 tools/run_iterative.py --family mc --n 900 --p 0.03 --graphs $g --layers 20 --hidden 16 --host 0 2>/dev/null | grep '^{"solver' | cut -c1-150
done | tee gpurun_out/r06_narrow_iterative.txt
echo "iterative mc900 c16 l4, 64 graphs, padded / layered:" | tee -a gpurun_out/r06_narrow_iterative.txt
python3 tools/run_iterative.py --family mc --n 900 --p 0.03 --graphs 64 --layers 4 --hidden 16 --host 0 2>/dev/null | grep '^{"solver' | cut -c1-150 | tee -a gpurun_out/r06_narrow_iterative.txt
DGCN_OPTIONS="big=0,big2=0" python3 tools/run_iterative.py --family mc --n 900 --p 0.03 --graphs 64 --layers 4 --hidden 16 --host 0 2>/dev/null | grep '^{"solver' | cut -c1-150 | tee -a gpurun_out/r06_narrow_iterative.txt
DGCN_FUZZ_CASES=1000 timeout 1800 python -m pytest tests/test_gpu_fuzz.py -q -p no:cacheprovider 2>&1 | tail -4 | tee gpurun_out/r06_fuzz_1000.txt
