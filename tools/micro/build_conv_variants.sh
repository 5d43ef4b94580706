#!/bin/bash
# usage: tools/micro/build_conv_variants.sh name:"-DFLAG" ...  -> tools/micro/libdcd_<name>.so with conv.hip variants
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
rm -f $R/tools/micro/libdcd_*.so
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics $flags -c $R/dcd_amd/csrc/conv.hip -o /tmp/c_$name.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/micro/libdcd_$name.so /tmp/c_$name.o $R/dcd_amd/csrc/heads.o $R/dcd_amd/csrc/norm.o $R/dcd_amd/csrc/dcn_v2.o $R/dcd_amd/csrc/upsample.o $R/dcd_amd/csrc/stem.o $R/dcd_amd/csrc/targets.o $R/dcd_amd/csrc/spd.o $R/dcd_amd/csrc/loss_rows.o
  echo built $name
done
