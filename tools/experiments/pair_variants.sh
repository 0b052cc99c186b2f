#!/bin/bash
# builds build/pair_<name>/libstarkhip_pair.so for variants of the pair form's generated rounds: usage pair_variants.sh name:ENV=VAL,ENV=VAL ...
set -e
cd "$(dirname "$0")/../.."
for spec in "$@"; do
  name=${spec%%:*}; envs=${spec#*:}
  mkdir -p build/pair_$name
  ( IFS=,; for kv in $envs; do export "$kv"; done; python3 tools/gen_pair_round_asm.py > build/pair_$name/pair_round_asm.inc )
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Iinclude -DSTARKHIP_PAIR_INC="\"$(pwd)/build/pair_$name/pair_round_asm.inc\"" -c starky_bls12_381_amd/csrc/kernels_hash.hip -o build/pair_$name/kernels_hash.hip.o &
done
wait
for spec in "$@"; do
  name=${spec%%:*}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/pair_$name/libstarkhip_pair.so $(ls build/*.o | grep -v kernels_hash) build/pair_$name/kernels_hash.hip.o -lpthread
  grep "Per wave" build/pair_$name/pair_round_asm.inc
done
