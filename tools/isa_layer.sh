#!/bin/bash
# Compile csrc/cwn_layer.hip with -save-temps into /tmp/st and print, per layer_kernel instantiation, the
# register / scratch use and the positions of full waits (vmcnt(0)) and scratch traffic.  Extra flags: $@
set -e
rm -rf /tmp/st && mkdir -p /tmp/st && cd /tmp/st
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -save-temps "$@" -c /root/repo/cwn_amd/csrc/cwn_layer.hip -o l.o 2>&1 | grep -v warning | head -5
S=cwn_layer-hip-amdgcn-amd-amdhsa-gfx950.s
grep -E "^\s+\.(vgpr_count|sgpr_count|vgpr_spill_count|private_segment_fixed_size|name):" $S | awk '{print $2,$3}' | paste - - - - - | grep layer_kernel | sed 's/_ZN12_GLOBAL__N_112layer_kernelILi//; s/EEEvNS_9LayerArgsE//'
for k in 128ELi2 128ELi1 64ELi2 64ELi1; do
  awk "/^_ZN12_GLOBAL__N_112layer_kernelILi${k}EEEvNS_9LayerArgsE:/,/s_endpgm/" $S > k_$k.s
  echo "== $k: $(wc -l < k_$k.s) lines; vmcnt(0) at: $(grep -n 'vmcnt(0)' k_$k.s | cut -d: -f1 | tr '\n' ' '); scratch ops: $(grep -c 'scratch_' k_$k.s); first mfma line $(grep -n v_mfma k_$k.s | head -1 | cut -d: -f1)"
done
