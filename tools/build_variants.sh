#!/bin/bash
# usage: tools/build_variants.sh tag1="-DFLAG ..." tag2="..."   -- builds tools/variants/libvqhip_<tag>.so (A/B runs through VQHIP_SO)
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/tools/variants
for spec in "$@"; do
  tag=${spec%%=*}; flags=${spec#*=}
  ( cd $R/vector_quantize_pytorch_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -shared -Wall -Wno-unused-result $flags -o $R/tools/variants/libvqhip_$tag.so vqhip.hip vq_screen.hip 2>&1 | grep -E "error|warning: variable" ; echo "built $tag" ) &
done
wait
