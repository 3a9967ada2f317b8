#!/bin/bash
# VGPRs, scratch bytes and spills of every kernel in libultra_hip.so (name, private_segment_fixed_size, vgpr_count, vgpr_spill_count)
set -e
T=$(mktemp -d); cp "$(dirname "$0")/../ultra_amd/libultra_hip.so" "$T/l.so"; cd "$T"
/opt/rocm/lib/llvm/bin/llvm-objdump --offloading l.so > /dev/null
for f in *gfx950; do /opt/rocm/lib/llvm/bin/llvm-readelf --notes "$f" 2>/dev/null | grep -E "^\s+\.name:|\.private_segment_fixed_size|\.vgpr_count|\.vgpr_spill_count"; done | paste - - - - | awk '{print $2, $4, $6, $8}'
rm -rf "$T"
