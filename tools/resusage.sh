#!/bin/bash
# Register / spill / scratch summary of every kernel in one source file: tools/resusage.sh fused.hip [extra flags]
cd "$(dirname "$0")/../distgcn_amd/csrc" && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c "$1" -o /tmp/resusage.o "${@:2}" \
  -Rpass-analysis=kernel-resource-usage 2>&1 | awk '
  /Function Name:/ {name=$0; sub(/.*Function Name: /,"",name); sub(/ \[.*/,"",name)}
  /    VGPRs:/ {v=$0; sub(/.*VGPRs: /,"",v); sub(/ \[.*/,"",v)}
  /AGPRs:/ {a=$0; sub(/.*AGPRs: /,"",a); sub(/ \[.*/,"",a)}
  /VGPRs Spill:/ {s=$0; sub(/.*Spill: /,"",s); sub(/ \[.*/,"",s)}
  /ScratchSize/ {c=$0; sub(/.*: /,"",c); sub(/ \[.*/,"",c)}
  /LDS Size/ {printf "%-70s vgpr %4s agpr %3s vspill %3s scratch %5s\n", name, v, a, s, c}'
