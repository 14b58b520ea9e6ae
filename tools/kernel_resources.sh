#!/bin/bash
# usage: tools/kernel_resources.sh <file.hip under featuredetection_amd/csrc> <kernel-name pattern>
# registers / scratch / LDS / occupancy of the matching kernels as the compiler reports them (no GPU needed)
cd "$(dirname "$0")/../featuredetection_amd/csrc" || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-result -Wno-int-to-pointer-cast -c "$1" -o /tmp/kr_$$.o \
  -Rpass-analysis=kernel-resource-usage 2>&1 | awk -v pat="$2" '
  /Function Name:|remark: Name:|: Name:/ { name=$0; sub(/.*Name: /,"",name); sub(/ \[-Rpass.*/,"",name); show = (name ~ pat) }
  show && /VGPRs:|AGPRs:|ScratchSize|Occupancy|LDS Size|VGPRs Spill/ { v=$0; sub(/.*:[0-9]+:[0-9]+: +/,"",v); sub(/ \[-Rpass.*/,"",v); printf "%s | %s\n", name, v }'
rm -f /tmp/kr_$$.o
