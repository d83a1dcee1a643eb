#!/bin/bash
# tools/kres.sh [lib.so] [name filter]: VGPR / SGPR / spill / LDS figures of the kernels in a built library (code object notes)
L=${1:-gyroflow_amd/libgfremap.so}; F=${2:-k_}
d=$(mktemp -d); cp "$L" $d/lib.so; (cd $d && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading lib.so > /dev/null)
co=$(ls $d/lib.so.*amdgcn* | head -1)
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $co | awk '/\.name:/{n=$2} /\.vgpr_count:/{v=$2} /\.sgpr_count:/{s=$2} /\.vgpr_spill_count:/{vs=$2} /\.sgpr_spill_count:/{ss=$2} /\.group_segment_fixed_size:/{l=$2} /\.private_segment_fixed_size:/{p=$2} /\.wavefront_size:/{print n, "vgpr", v, "sgpr", s, "vspill", vs, "sspill", ss, "lds", l, "scratch", p}' | c++filt | grep "$F"
rm -rf $d
