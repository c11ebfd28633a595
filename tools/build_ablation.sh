#!/bin/bash
# tools/build_ablation.sh <name> <extra hipcc flags...>  -> gpurun_out/abl_<name>.so (ships with gpurun? no: build into tools/_abl/)
set -e
name=$1; shift
cd "$(dirname "$0")/.."
mkdir -p tools/_abl
objs=""
for f in ln_partition linear_f32 region_attn rmsa_fused crmsa api; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast "$@" -c rrt-mil_amd/csrc/$f.hip -o tools/_abl/${name}_$f.o &
  objs="$objs tools/_abl/${name}_$f.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o tools/_abl/librrt_$name.so
rm -f $objs
echo tools/_abl/librrt_$name.so
