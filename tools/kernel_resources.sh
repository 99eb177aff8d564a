#!/bin/bash
# tools/kernel_resources.sh — registers, scratch and occupancy of every kernel of
# the library, from the compiler's resource-usage remarks (no GPU needed).
cd "$(dirname "$0")/.." && make -s isa 2>&1 | awk '
/Function Name:/ {name=$NF}
/ Name: /        {name=$(NF-1)}
/VGPRs:/ && !/Agpr|AGPR/ {v=$(NF-1)}
/ScratchSize/    {s=$(NF-1)}
/SGPRs:/         {sg=$(NF-1)}
/Occupancy/      {o=$(NF-1); cmd="c++filt " name; cmd | getline d; close(cmd); printf "%-4s vgpr %-4s sgpr %-6s scratch-bytes %-3s waves/SIMD  %s\n", v, sg, s, o, substr(d,1,80)}'
