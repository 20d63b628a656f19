#!/bin/bash
# round 4, call 24: k_big<BLOCK, TILES>: record stream with a cursor, two groups in flight where two tiles per wave suffice
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_general.py tests/test_gpu_fuzz.py -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/r04_gpu24.log 2>&1
tail -3 gpurun_out/r04_gpu24.log
run() { local name=$1; shift; python bench.py "$@" --cpu-seconds 0 --no-cpu-pool --no-e2e --no-spmm-probe --parity-seconds 0 2>/dev/null | tail -1 > gpurun_out/r04_tiles_$name.json; }
run er500 --config ER500
DGCN_BIG_TILES=4 run er500_t4 --config ER500
run mc900 --config MC900
run c3_any --any-size-path
DGCN_BIG_TILES=4 run c3_any_t4 --any-size-path
run c4l20_any --config C4-share --layers 20 --steps 600 --any-size-path
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04_tiles_*.json")):
    try:
        d=json.load(open(f)); print("%-40s %12.0f graphs/s  %.4f ms/step  %s" % (f.split("r04_tiles_")[1], d["value"], d["ms_per_step"], {k: round(v["avg_us"],1) for k,v in d["kernels"].items()}))
    except Exception as e: print(f, "ERR", e)
PY
