#!/bin/bash
# register / scratch use of every k_gemm3 instantiation (no GPU needed)
cd /root/repo/rltime_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -S --cuda-device-only gemm3.hip -o /tmp/gemm3_new.s 2>&1 | grep -v warning | tail -5
grep -E "^\s+\.(vgpr_count|private_segment_fixed_size|vgpr_spill_count|name):" /tmp/gemm3_new.s | paste - - - - | sed 's/_ZN4mirl//' | awk '{print $2, "scratch", $4, "vgpr", $6, "spills", $8}'
