#!/bin/bash
# round 6, call 32: the numbers behind test_more_shapes_against_the_restatement_on_gpu
timeout 300 python tools/parity_more_shapes.py 2>&1 | tee gpurun_out/r06_more_shapes.txt
