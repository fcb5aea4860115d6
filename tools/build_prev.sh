#!/bin/bash
# builds gdr-net_amd/lib/libgdrn_hip_prev.so from the sources of a commit (default HEAD) -- the other side of a same-box A/B (GDRN_HIP_LIB)
set -e
REV=${1:-HEAD}
R=$(cd $(dirname $0)/.. && pwd)
T=$(mktemp -d /tmp/prevsrc.XXXX)
mkdir -p $T/gdr-net_amd/csrc $T/include
for f in $(git -C $R ls-tree --name-only $REV gdr-net_amd/csrc/); do git -C $R show $REV:$f > $T/$f; done
git -C $R show $REV:include/gdrn_hip.h > $T/include/gdrn_hip.h
cd $T/gdr-net_amd/csrc
ls *.hip | xargs -P 8 -I{} /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I $T/include -c {} -o {}.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/gdr-net_amd/lib/libgdrn_hip_prev.so *.o
rm -rf $T
ls -la $R/gdr-net_amd/lib/libgdrn_hip_prev.so
