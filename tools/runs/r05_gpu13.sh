#!/bin/bash
# round 5, call 13: the ceiling of a one-wave launch that moves the SpMM's bytes (tools/micro/stream_ramp.hip); k_fused C3 as the box's clock reference; plain k_big
hipcc --offload-arch=gfx950 -O3 -o tools/micro/stream_ramp tools/micro/stream_ramp.hip 2>/dev/null && ./tools/micro/stream_ramp
python tools/run_fused.py er 300 20 500 2>/dev/null | tail -1
for c in er500 mc900; do python tools/run_general.py $c 300 20 256 2>/dev/null | grep big_solve; done
