#!/bin/bash
# Builds a variant of the library next to the product one for A/B runs: one source recompiled with extra flags, linked with the product's other
# objects.   usage: tools/build_variant.sh NAME SOURCE.hip "-DFLAG=.. ..."   ->  knp-emi-dg_amd/knpemidg/libknpemi_hip_NAME.so
#            then:  KNP_LIB_PATH=knp-emi-dg_amd/knpemidg/libknpemi_hip_NAME.so python tools/apply_only.py 2 50 2
set -e
cd "$(dirname "$0")/.."
name=$1; src=$2; flags=$3
csrc=knp-emi-dg_amd/csrc
python knp-emi-dg_amd/build.py > /dev/null
obj=/tmp/variant_${name}_${src%.hip}.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value $flags -c $csrc/$src -o $obj
objs=$(ls $csrc/*.o | grep -v "/${src%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o knp-emi-dg_amd/knpemidg/libknpemi_hip_${name}.so $objs $obj -L/opt/rocm/lib -lrccl -lpthread -Wl,-rpath,/opt/rocm/lib
echo knp-emi-dg_amd/knpemidg/libknpemi_hip_${name}.so
