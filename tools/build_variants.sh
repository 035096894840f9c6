#!/bin/bash
# usage: tools/build_variants.sh tag1="-DFLAG ..." tag2="..."   -- builds tools/variants/libvqhip_<tag>.so (A/B runs through VQHIP_SO).
# Only csrc/vq_screen.hip is recompiled with the flags; csrc/vqhip.o (built by make) is linked as is.
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/tools/variants
for spec in "$@"; do
  tag=${spec%%=*}; flags=${spec#*=}
  ( cd $R/vector_quantize_pytorch_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -Wall -Wno-unused-result $flags -c -o $R/tools/variants/vq_screen_$tag.o vq_screen.hip 2>&1 | grep -E "error|warning: variable" ;
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/variants/libvqhip_$tag.so vqhip.o vq_screen_c.o $R/tools/variants/vq_screen_$tag.o && rm -f $R/tools/variants/vq_screen_$tag.o; echo "built $tag" ) &
done
wait
