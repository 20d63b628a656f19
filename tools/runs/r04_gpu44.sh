#!/bin/bash
# round 4, call 44: weight fragments fetched only by waves that own a tile (lone 1 024-thread workgroups late in a search)
python tools/run_iterative.py --n 500 --p 0.02 --graphs 64 --layers 20 --host 0 2>/dev/null | grep "^{" | cut -c1-150
DGCN_TAIL=0 python tools/run_iterative.py --n 500 --p 0.02 --graphs 64 --layers 20 --host 0 2>/dev/null | grep "^{" | cut -c1-150
python bench.py --cpu-seconds 0 --no-cpu-pool --no-e2e --parity-seconds 0 --no-spmm-probe 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C3', d['ms_per_step'], d['roofline']['avg_launch_us'])"
python bench.py --config C4-share --layers 20 --steps 600 --cpu-seconds 0 --no-e2e --parity-seconds 0 --no-spmm-probe 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C4 share l20', d['ms_per_step'], d['roofline']['avg_launch_us'])"
timeout 1500 python -m pytest tests/test_gpu_api.py tests/test_gpu_kernels.py tests/test_gpu_tail.py -x -q -p no:cacheprovider 2>&1 | tail -2
