#!/bin/bash
# usage: tools/kres.sh <file.hip>  -- per-kernel VGPR/SGPR/LDS/occupancy/spill summary (gfx950)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -c "$1" -o /dev/null \
  -Rpass-analysis=kernel-resource-usage 2>&1 | grep "remark:" | sed -e 's/.*remark: *//' -e 's/ \[-Rpass.*//' | awk '
/^Function Name:/ { if (n) print line; n=1; line=substr($3,1,60); next }
/^(VGPRs:|AGPRs:|TotalSGPRs|ScratchSize|Occupancy|VGPRs Spill|LDS Size)/ { gsub(/ \[[^]]*\]/,""); line=line " | " $0 }
END { print line }'
