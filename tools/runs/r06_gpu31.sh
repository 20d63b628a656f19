#!/bin/bash
# round 6, call 31: twelve more shapes held against the restatement directly (tests/test_gpu_full_size.py::test_more_shapes_against_the_restatement_on_gpu)
python -m pytest tests/test_gpu_full_size.py -m gpu -q -k more_shapes 2>&1 | tail -40 | tee gpurun_out/r06_more_shapes.txt
