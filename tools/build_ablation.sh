#!/bin/bash
# tools/build_ablation.sh <name> <extra hipcc flags...>  -> tools/_abl/librrt_<name>.so (git-ignored; ships with gpurun)
set -e
name=$1; shift
cd "$(dirname "$0")/.."
mkdir -p tools/_abl
objs=""
for f in $(python -c "import sys; sys.path.insert(0, 'rrt-mil_amd'); import build; print(' '.join(s[:-4] for s in build.SOURCES))"); do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -mllvm -amdgpu-mfma-vgpr-form=1 "$@" -c rrt-mil_amd/csrc/$f.hip -o tools/_abl/${name}_$f.o &
  objs="$objs tools/_abl/${name}_$f.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o tools/_abl/librrt_$name.so
rm -f $objs
echo tools/_abl/librrt_$name.so
