#!/bin/bash
# usage: scripts/experiments/build_variant.sh <tag> <source.hip> <extra hipcc flags...>  -- libecc_hip.so with ONE translation unit
# rebuilt with extra flags, into scripts/experiments/_build/libecc_<tag>.so (git-ignored; travels to the GPU box; run with
# ECC_HIP_LIB=$PWD/scripts/experiments/_build/libecc_<tag>.so)
set -e
R=$(cd $(dirname $0)/../.. && pwd); C=$R/epipolarconsistency_amd/csrc; B=$R/scripts/experiments/_build
tag=$1; src=$2; shift; shift
mkdir -p $B
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fvisibility=hidden -fno-slp-vectorize "$@" -c $C/$src -o $B/${tag}.o
objs=$(ls $C/*.o | grep -v "/${src%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs $B/${tag}.o -lrt -lpthread -ldl -o $B/libecc_${tag}.so
echo built $B/libecc_${tag}.so
