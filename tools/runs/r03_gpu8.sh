#!/bin/bash
# round 3, call 8: single-wave shallow kernel; bench --config C5 / C2 / C4-share
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fuzz.py -q -k "shallow or ties or golden or fuzz" 2>&1 | tail -4
python -m pytest tests/test_gpu_api.py -q -k "executed or heuristics or known_answers or test_loop" 2>&1 | tail -3
for cfg in "er100 300 1 500" "ba 300 1 500" "er200 300 1 500" "er100 100 1 4000"; do python tools/run_fused.py $cfg; done
bash tools/build_diag.sh 2>&1 | tail -2
for k in er100 ba; do DGCN_LIB=distgcn_amd/libdgcn_diag.so python tools/stamp_shallow.py $k 500; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_shallow_stamps4.txt
python bench.py --config C5 --cpu-seconds 25 2>&1 | tail -1 > gpurun_out/r03_gpu8_c5.json; python - <<'PY'
import json; d=json.loads(open('gpurun_out/r03_gpu8_c5.json').read()); print({k: d[k] for k in ("metric","value","ms_per_step","roofline","cpu_baseline")}); print(d["config"])
PY
python bench.py --config C2 --no-e2e --no-spmm-probe --no-cpu-pool --cpu-seconds 3 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['parity_full_size']['sets_differing'])"
