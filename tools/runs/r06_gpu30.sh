#!/bin/bash
# round 6, call 30: the whole GPU suite on the library with the folded dispatch order
python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee gpurun_out/r06_suite_fold.txt
