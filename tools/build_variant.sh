#!/bin/bash
# tools only: another build of the library with extra -D flags, for same-box A/Bs through SNEKMER_HIP_LIB.
#   tools/build_variant.sh lb32 "-DSKM_OS_LB=32"   ->  snekmer_amd/libsnekmer_hip_lb32.so   (git-ignored like every .so)
R=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; DEFS=$2
C=$R/snekmer_amd/csrc; O=$C/_obj/v_$NAME
mkdir -p $O
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$R/include -I$C -Wno-unused-function -Wno-unused-value -Wno-unused-result -ffp-contract=off $DEFS"
pids=()
for f in $C/*.hip; do
  b=$(basename $f .hip)
  /opt/rocm/bin/hipcc $FLAGS -c $f -o $O/$b.o &
  pids+=($!)
  if [ ${#pids[@]} -ge 6 ]; then wait ${pids[0]}; pids=("${pids[@]:1}"); fi
done
wait
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -I$R/include -I$C -c $C/skm_host.cpp -o $O/skm_host.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/snekmer_amd/libsnekmer_hip_$NAME.so $O/*.o -ldl -lpthread -lz && echo built libsnekmer_hip_$NAME.so
