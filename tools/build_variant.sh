#!/bin/bash
# Build a perf / diagnostic variant of libvtaco_hip.so beside the product build (objects under /tmp, the library under
# gpurun_out/variants/, which is scratch and travels to the GPU box only when the call is made from there).
# Usage: bash tools/build_variant.sh NAME "-DFLAG ..."   ->  variants/lib_NAME.so ; select it with VTACO_HIP_LIB.
set -e
NAME=$1; FLAGS=$2
R=$(cd "$(dirname "$0")/.." && pwd)
O=/tmp/vt_variant_$NAME; mkdir -p $O $R/variants
cd $R/vtaco_amd/csrc
for f in *.hip; do
  b=${f%.hip}; extra=""
  case $b in decode|decode_bwd|fusion) extra="-fno-honor-nans";; decode_f16|decode_wide) extra="-fno-honor-nans -fno-slp-vectorize";; esac
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DVT_WAVES_PER_SIMD=4 -I../../include $extra $FLAGS -c $f -o $O/$b.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $O/*.o -o $R/variants/lib_$NAME.so
echo built $R/variants/lib_$NAME.so
