#!/bin/bash
echo "== cluster auto, l=20"; DGCN_LIB=$PWD/distgcn_amd/libdgcn_diag.so python tools/stamp_fused.py er 20 1 2>/dev/null | sed -n 2,30p
