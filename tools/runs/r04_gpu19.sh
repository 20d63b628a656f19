#!/bin/bash
# round 4, call 19: k_big with 512 threads (two graphs per CU) for graphs up to 512 vertices: tests; then C3 and the C4 share
# forced down the any-size path, against the fused kernel
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_general.py -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/r04_gpu19.log 2>&1
tail -3 gpurun_out/r04_gpu19.log
run() { local name=$1; shift; python bench.py "$@" --cpu-seconds 0 --no-cpu-pool --no-e2e --no-spmm-probe --parity-seconds 0 2>/dev/null | tail -1 > gpurun_out/r04_blk_$name.json; }
run c3_fused
run c3_any --any-size-path
DGCN_BIG_BLOCK=1024 run c3_any1024 --any-size-path
run c4l20_fused --config C4-share --layers 20 --steps 600
run c4l20_any --config C4-share --layers 20 --steps 600 --any-size-path
DGCN_BIG_BLOCK=1024 run c4l20_any1024 --config C4-share --layers 20 --steps 600 --any-size-path
run c5size_fused --graphs 64 --nodes 500 --p 0.02 --steps 600
run c5size_any --graphs 64 --nodes 500 --p 0.02 --steps 600 --any-size-path
run c5size256_fused --graphs 256 --nodes 500 --p 0.02 --steps 600
run c5size256_any --graphs 256 --nodes 500 --p 0.02 --steps 600 --any-size-path
run er500 --config ER500
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04_blk_*.json")):
    try:
        d=json.load(open(f)); print("%-40s %12.0f graphs/s  %.4f ms/step  %s" % (f.split("r04_blk_")[1], d["value"], d["ms_per_step"], {k: round(v["avg_us"],1) for k,v in d["kernels"].items()}))
    except Exception as e: print(f, "ERR", e)
PY
