#!/bin/bash
# dev: what bounds dfx_k_proj256_h3x2<8> (tools/dev/proj_h3_bench.hip with DFX_PH_ABLATE bits: 1 no stores, 2 no LDS refill, 4 no matrix ops, 8 no prefetch loads)
cd "$(dirname "$0")"
for a in 0 1 2 4 8 3 11 15; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 -DDFX_PH_ABLATE=$a -I../../include -I../../deepfilternet_amd/csrc/env_hip -I../../deepfilternet_amd/csrc proj_h3_bench.hip -o /tmp/pb_$a 2>/dev/null && echo "ablate=$a: $(/tmp/pb_$a | head -1)"
done
