#!/bin/bash
# One-file tuning variant: rebuild ONE kernel file with extra -D flags and link it with the regular objects.
#   tools/variant_one.sh NAME FILE.hip "-DFOO=1 ..."   ->  nerf-rs_amd/libnerf_mi355x_NAME.so   (A/B them with tools/ab.sh)
# Like `make variant` the library carries a build tag (nerf_build_variant(), compiled into nerf_host_api.o): the Python loader refuses it
# without NERF_ALLOW_VARIANT=1, so a timing-only build can never stand in for the product.
set -e
NAME=$1; FILE=$2; DEFS=$3
[ -n "$NAME" ] && [ -n "$FILE" ] || { echo "usage: $0 NAME FILE.hip \"-D...\""; exit 2; }
cd "$(dirname "$0")/../nerf-rs_amd/csrc"
mkdir -p build/$NAME
OBJ=build/$NAME/${FILE%.hip}.o
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-result --offload-arch=gfx950"
EXTRA=""
{ [ "$FILE" = "mlp_kernel_bf16v2.hip" ] || [ "$FILE" = "mlp_kernel_f16v2.hip" ]; } && EXTRA="-mllvm -amdgpu-mfma-vgpr-form=1"   # as in the Makefile
/opt/rocm/bin/hipcc $FLAGS $EXTRA $DEFS -c $FILE -o $OBJ
TAG="-DNERF_BUILD_VARIANT=\"$NAME: ${FILE} ${DEFS//\"/}\""
/opt/rocm/bin/hipcc $FLAGS "$TAG" -x hip -c nerf_host_api.cpp -o build/$NAME/nerf_host_api.o
OTHERS=$(ls *.o | grep -v "^${FILE%.hip}.o$" | grep -v "^nerf_host_api.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../libnerf_mi355x_$NAME.so $OBJ build/$NAME/nerf_host_api.o $OTHERS -ldl -lpthread
echo built ../libnerf_mi355x_$NAME.so
