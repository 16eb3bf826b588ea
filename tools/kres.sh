#!/bin/bash
# Register / LDS / scratch usage of every kernel in one .hip translation unit (cross-compiles for gfx950; no GPU needed).
# usage: tools/kres.sh pilco_amd/csrc/prep.hip [extra hipcc flags]
f=$(realpath $1); shift; cd $(dirname $(realpath $0))/..
out=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -I/opt/rocm/include -mllvm -amdgpu-mfma-vgpr-form "$@" \
    --cuda-device-only -S -o $out/k.s $f 2>/dev/null
awk '/^[ \t]*\.amdhsa_kernel /{name=$2} /\.amdhsa_next_free_vgpr/{v=$2} /\.amdhsa_next_free_sgpr/{s=$2} /\.amdhsa_group_segment_fixed_size/{l=$2} /\.amdhsa_private_segment_fixed_size/{p=$2} /\.amdhsa_accum_offset/{a=$2} /^[ \t]*\.end_amdhsa_kernel/{printf "%-110s vgpr %4s (accum_offset %4s) sgpr %4s lds %6s scratch %5s\n", name, v, a, s, l, p}' $out/k.s | c++filt | cut -c1-200
rm -rf $out
