#!/bin/bash
# lab build (never shipped): the whole library with extra flags -> tools/lab/lab_<name>.so, for PDAE_LIB A/B runs
#   bash tools/lab/build_variant.sh sc1 "-DR3_SC1"       then on the box:  PDAE_LIB=tools/lab/lab_sc1.so python bench.py ...
set -e
cd "$(dirname "$0")/../.."
NAME=$1; EXTRA=$2
B=/tmp/pdae_variant_$NAME
mkdir -p $B
cd point_dae_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -Wall -Wno-unused-function $EXTRA"
ls *.hip abi.cpp | xargs -P 8 -I{} sh -c "/opt/rocm/bin/hipcc $FLAGS \$( [ {} = abi.cpp ] && echo '-x hip' ) -c {} -o $B/{}.o"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/lab/lab_$NAME.so $B/*.o
ls -la ../../tools/lab/lab_$NAME.so
