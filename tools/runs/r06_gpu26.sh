#!/bin/bash
# round 6, call 26: tools/micro/mix_gather - part of the aggregation's gathers through the vector-memory path instead of the LDS?
cd tools/micro && ./mix_gather | tee ../../gpurun_out/r06_mix_gather.txt
