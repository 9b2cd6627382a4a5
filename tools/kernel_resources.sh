#!/bin/bash
# Dev tool (CPU): register / scratch / LDS figures of the kernels in ddmpc_api.hip's device code, straight from the compiler.
#   bash tools/kernel_resources.sh [name-filter]
cd "$(dirname "$0")/.."
CS=direct_data_driven_mpc_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -simplifycfg-sink-common=false -I$CS --cuda-device-only -S -o /tmp/ddmpc_api_dev.s $CS/ddmpc_api.hip $DDMPC_EXTRA_FLAGS || exit 1
awk -v f="${1:-rr2}" '/^[ \t]*\.amdhsa_kernel /{k=$2} /\.amdhsa_next_free_vgpr|\.amdhsa_accum_offset|\.amdhsa_private_segment_fixed_size|\.amdhsa_group_segment_fixed_size/{ if (k ~ f) printf "%s %s %s\n", k, $1, $2 }' /tmp/ddmpc_api_dev.s | sed 's/_ZN5ddmpc//; s/\.amdhsa_//' | awk '{a[$1]=a[$1] " " $2 "=" $3} END {for (k in a) print k a[k]}' | sort
