#!/bin/bash
# round 3, call 30: residual-graph kernel with the remaining vertices renumbered (image of na vertices, not N); empty-row support test
python -m pytest tests -m gpu -q -x 2>&1 | tail -4
python tools/run_iterative.py --graphs 64 --host 0 2>&1 | tail -3 | cut -c1-220
python tools/run_iterative.py --graphs 8 --host 0 2>&1 | tail -3 | cut -c1-220
python bench.py --config C5 --cpu-seconds 0 --parity-seconds 0 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C5:', d['value'], d['ms_per_step'])"
python tools/run_wireless.py 2>&1 | tail -5 | cut -c1-200
