#!/bin/bash
# registers / scratch / occupancy of every kernel of one .hip file (device-only compile with the compiler's resource remarks)
# usage: scripts/kernel_resources.sh megagta_amd/csrc/astar.hip [extra hipcc flags]
f=$1; shift
cd "$(dirname "$f")" && /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -Wno-unused-function -Wno-pass-failed \
  --cuda-device-only -c "$(basename "$f")" -o /dev/null -Rpass-analysis=kernel-resource-usage "$@" 2>&1 |
  awk '/Function Name|Name:/{n=$0; sub(/.*Name: /,"",n); sub(/ \[.*/,"",n)} /VGPRs:/{v=$0; sub(/.*VGPRs: /,"",v); sub(/ \[.*/,"",v)} /AGPRs:/{a=$0; sub(/.*AGPRs: /,"",a); sub(/ \[.*/,"",a)}
       /ScratchSize/{s=$0; sub(/.*: /,"",s); sub(/ \[.*/,"",s)} /Occupancy/{o=$0; sub(/.*: /,"",o); sub(/ \[.*/,"",o); printf "%-90s VGPR %s AGPR %s scratch %s occupancy %s\n", n, v, a, s, o}'
