#!/bin/bash
# dev: libdfx with extra -D defines for dfx_dsp.hip only (the other objects are the product build's: run deepfilternet_amd/build.py first)
# usage: tools/dev/build_dsp_variant.sh <tag> -DDFX_STFT_TAB=5 ...   -> tools/dev/_build/libdfx_<tag>.so (load it with DFX_LIBRARY=...)
set -e
TAG=$1; shift
cd "$(dirname "$0")/../.."
B=deepfilternet_amd/csrc/_build
mkdir -p tools/dev/_build
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form=1 -Xclang -target-feature -Xclang -packed-fp32-ops -Iinclude -Ideepfilternet_amd/csrc/env_hip -Ideepfilternet_amd/csrc "$@" -c deepfilternet_amd/csrc/dfx_dsp.hip -o tools/dev/_build/dfx_dsp_$TAG.o
hipcc --offload-arch=gfx950 -shared -fPIC tools/dev/_build/dfx_dsp_$TAG.o $B/dfx_model.hip.o $B/dfx_capi.hip.o $B/dfx_io.hip.o $B/dfx_mf.hip.o $B/dfx_onnx.hip.o -lz -o tools/dev/_build/libdfx_$TAG.so
rm -f tools/dev/_build/dfx_dsp_$TAG.o
ls -la tools/dev/_build/libdfx_$TAG.so
