#!/bin/bash
# The 16x16x32 bf16 kernel (mlp_kernel_bf16v3.hip, design 3) as a tagged variant library: only its own object, nerf_api.cpp
# (-DNERF_BF16_V3=1 routes NERF_MLP_BF16 to it) and the build tag are compiled; everything else is the product's objects.
#   tools/variant_bv3.sh NAME "-D..."  ->  nerf-rs_amd/libnerf_mi355x_NAME.so
set -e
NAME=$1; DEFS=$2
[ -n "$NAME" ] || { echo "usage: $0 NAME \"-D...\""; exit 2; }
cd "$(dirname "$0")/../nerf-rs_amd/csrc"
mkdir -p build/$NAME
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-result --offload-arch=gfx950"
/opt/rocm/bin/hipcc $FLAGS -mllvm -amdgpu-mfma-vgpr-form=1 -DNERF_BF16_V3=1 $DEFS -c mlp_kernel_bf16v3.hip -o build/$NAME/mlp_kernel_bf16v3.o &
/opt/rocm/bin/hipcc $FLAGS -DNERF_BF16_V3=1 $DEFS -x hip -c nerf_api.cpp -o build/$NAME/nerf_api.o &
TAG="-DNERF_BUILD_VARIANT=\"$NAME: bv3 ${DEFS//\"/}\""
/opt/rocm/bin/hipcc $FLAGS "$TAG" -x hip -c nerf_host_api.cpp -o build/$NAME/nerf_host_api.o &
wait
OTHERS=$(ls *.o | grep -v "^nerf_api.o$" | grep -v "^nerf_host_api.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../libnerf_mi355x_$NAME.so build/$NAME/mlp_kernel_bf16v3.o build/$NAME/nerf_api.o build/$NAME/nerf_host_api.o $OTHERS -ldl -lpthread
echo built ../libnerf_mi355x_$NAME.so
