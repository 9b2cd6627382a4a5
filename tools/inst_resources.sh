#!/bin/bash
# Dev tool (CPU): register / scratch / LDS figures of one instance of the cold-solve kernel.   bash tools/inst_resources.sh NT W [true|false]
cd "$(dirname "$0")/.."
CS=direct_data_driven_mpc_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -simplifycfg-sink-common=false -I$CS -DDDMPC_INST_NT=$1 -DDDMPC_INST_W=$2 -DDDMPC_INST_REF=${3:-false} --cuda-device-only -S -o /tmp/ddmpc_inst_$1_$2_${3:-false}.s $CS/ddmpc_inst.hip 2>/dev/null || exit 1
grep -E "\.amdhsa_next_free_vgpr|\.amdhsa_accum_offset|\.amdhsa_private_segment_fixed_size|\.amdhsa_group_segment_fixed_size|\.amdhsa_next_free_sgpr" /tmp/ddmpc_inst_$1_$2_${3:-false}.s | sed 's/\.amdhsa_//' | tr -s ' \t' ' ' | tr '\n' ';'; echo
grep -c "v_mfma_f64_16x16x4" /tmp/ddmpc_inst_$1_$2_${3:-false}.s
