#!/bin/bash
# per-kernel register / spill / LDS report of one HIP source: scripts/kres.sh polars_amd/csrc/fast_path.hip
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Rpass-analysis=kernel-resource-usage -c "$1" -o /tmp/kres.o 2>&1 |
 grep -E "error|Function Name|    VGPRs:|VGPRs Spill|Occupancy|LDS Size" | sed -e 's/.*remark: *//' -e 's/ \[-Rpass.*//' |
 awk '/Function Name/{if(l)print l; l=$3} /VGPRs:/{l=l" vgpr="$2} /Spill/{l=l" spill="$3} /Occupancy/{l=l" occ="$3} /LDS/{l=l" lds="$4} /error/{print} END{print l}'
