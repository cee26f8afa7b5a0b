#!/bin/bash
# Register / scratch / occupancy table of every kernel in one .hip file (compiler remarks, no GPU needed):
#   scripts/kernel_resources.sh viterbi_spl_amd/csrc/backtrace_lane.hip
cd "$(dirname "$1")" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-honor-nans \
    -I../../include -c -o /dev/null "$(basename "$1")" -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys, re
row = None
for line in sys.stdin:
    m = re.search(r'remark:\s+(.*?) \[-Rpass', line)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith('Function Name'):
        if row: print(row)
        row = t.split(': ')[1][:70].ljust(72)
    elif re.match(r'(VGPRs:|AGPRs|ScratchSize|Occupancy|VGPRs Spill)', t):
        row += t.replace(' [bytes/lane]', '').replace(' [waves/SIMD]', '') + '  '
if row: print(row)
"
