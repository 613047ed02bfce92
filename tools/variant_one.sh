#!/bin/bash
# One-file tuning variant: rebuild ONE kernel file with extra -D flags and link it with the regular objects.
#   tools/variant_one.sh NAME FILE.hip "-DFOO=1 ..."   ->  nerf-rs_amd/libnerf_mi355x_NAME.so   (A/B them with tools/ab.sh)
set -e
NAME=$1; FILE=$2; DEFS=$3
cd "$(dirname "$0")/../nerf-rs_amd/csrc"
mkdir -p build/$NAME
OBJ=build/$NAME/${FILE%.hip}.o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-result --offload-arch=gfx950 $DEFS -c $FILE -o $OBJ
OTHERS=$(ls *.o | grep -v "^${FILE%.hip}.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../libnerf_mi355x_$NAME.so $OBJ $OTHERS -ldl -lpthread
echo built ../libnerf_mi355x_$NAME.so
