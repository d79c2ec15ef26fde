#!/bin/bash
# dev: build libdfx with extra -D defines into tools/dev/_build/libdfx_<tag>.so (load it with DFX_LIBRARY=...)
# usage: tools/dev/build_variant.sh <tag> -DDFX_GH_FR=25 -DDFX_GH_D=8 ...
set -e
TAG=$1; shift
cd "$(dirname "$0")/../.."
B=tools/dev/_build/v_$TAG; mkdir -p $B
for s in dfx_dsp dfx_model dfx_io dfx_mf dfx_capi dfx_onnx; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form=1 -Xclang -target-feature -Xclang -packed-fp32-ops -Iinclude -Ideepfilternet_amd/csrc/env_hip -Ideepfilternet_amd/csrc "$@" -c deepfilternet_amd/csrc/$s.hip -o $B/$s.o 2>&1 | grep -E "error|warning: .*spill" &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC $B/*.o -lz -o tools/dev/_build/libdfx_$TAG.so
rm -rf $B
ls -la tools/dev/_build/libdfx_$TAG.so
