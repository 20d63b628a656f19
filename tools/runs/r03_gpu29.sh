#!/bin/bash
# round 3, call 29: evidence pass - kernel-trace stats of every bench config, PMC passes, phase clocks (tools/collect_profiles_r03.sh)
bash tools/collect_profiles_r03.sh 2>&1 | tail -30
