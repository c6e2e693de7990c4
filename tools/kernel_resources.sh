#!/bin/bash
# Scratch memory, VGPRs and static LDS of every kernel, from the compiler's own metadata (no GPU needed):
#   tools/kernel_resources.sh [file.hip ...]        default: every .hip under naf_amd/csrc
# A kernel with .private_segment_fixed_size > 0 keeps arrays in scratch memory: every wave of it waits for a scratch allocation, and the
# launches of other streams wait behind it -- on the hot paths this column should read 0.
set -u
root=$(cd "$(dirname "$0")/.." && pwd)
files=("$@"); [ ${#files[@]} -eq 0 ] && files=("$root"/naf_amd/csrc/*.hip)
tmp=$(mktemp -d)
printf '%-8s %-6s %-8s %s\n' scratch vgprs lds kernel
for f in "${files[@]}"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -I"$root/include" --cuda-device-only -S -o "$tmp/k.s" "$f" 2>/dev/null || { echo "cannot compile $f" >&2; continue; }
  grep -E '^\s+\.(group_segment_fixed_size|name|private_segment_fixed_size|vgpr_count):' "$tmp/k.s" | paste - - - - | \
    awk '{ for (i = 1; i <= NF; i++) { if ($i == ".group_segment_fixed_size:") l = $(i+1); if ($i == ".name:") n = $(i+1); if ($i == ".private_segment_fixed_size:") s = $(i+1); if ($i == ".vgpr_count:") v = $(i+1) } printf "%-8s %-6s %-8s %s\n", s, v, l, n }' | sort -k1,1nr -k4,4
done | c++filt | sort -k1,1nr -s
rm -rf "$tmp"
