#!/bin/bash
# builds build/abl_<mask>/libstarkhip_abl.so for each ablation mask of the wave-resident LDE kernel (kernels_lde.hip only; the rest is the default build)
set -e
cd "$(dirname "$0")/../.."
for m in "$@"; do
  mkdir -p build/abl_$m
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Iinclude -DSTARKHIP_LDE_ABLATE=$m -c starky_bls12_381_amd/csrc/kernels_lde.hip -o build/abl_$m/kernels_lde.hip.o &
done
wait
for m in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/abl_$m/libstarkhip_abl.so $(ls build/*.o | grep -v kernels_lde) build/abl_$m/kernels_lde.hip.o -lpthread
done
