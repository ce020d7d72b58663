#!/bin/bash
# Builds libecc_hip.so variants that differ in the Radon kernel's tile capacity / staging registers / occupancy hint
# (ECC_SLAB_TILE_CAP, ECC_SLAB_N_PRE, ECC_RADON_MIN_WAVES) into scripts/experiments/_build/ (git-ignored; travels to the GPU box).
# usage: scripts/experiments/build_radon_variants.sh "5056 26 1" "4032 16 5" ...   then on the GPU box:
#        for so in scripts/experiments/_build/libecc_radon_*.so; do ECC_HIP_LIB=$PWD/$so python scripts/bench_radon.py 50 1024 768 2; done
set -e
R=$(cd $(dirname $0)/../.. && pwd)
C=$R/epipolarconsistency_amd/csrc
B=$R/scripts/experiments/_build
mkdir -p $B
for cfg in "$@"; do
  set -- $cfg
  tag=${1}_${2}_${3}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fvisibility=hidden -fno-slp-vectorize \
    -DECC_SLAB_TILE_CAP=$1 -DECC_SLAB_N_PRE=$2 -DECC_RADON_MIN_WAVES=$3 -c $C/radon_kernel.hip -o $B/radon_$tag.o
  objs=$(ls $C/*.o | grep -v radon_kernel.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs $B/radon_$tag.o -lrt -lpthread -o $B/libecc_radon_$tag.so
  echo built $B/libecc_radon_$tag.so
done
