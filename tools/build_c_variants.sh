#!/bin/bash
# usage: tools/build_c_variants.sh tag1="-DFLAG ..." tag2="..."   -- A/B builds of csrc/vq_screen_c.hip (the persistent screening kernel with
# the cyclic tile stream) into tools/variants/libvqhip_<tag>.so, linked against the in-tree objects; run them through VQHIP_SO (tools/run_var.sh).
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/tools/variants
for spec in "$@"; do
  tag=${spec%%=*}; flags=${spec#*=}
  ( cd $R/vector_quantize_pytorch_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -Wall -Wno-unused-result $flags -c -o $R/tools/variants/vq_screen_c_$tag.o vq_screen_c.hip 2>&1 | grep -E "error|warning: variable" ;
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/variants/libvqhip_$tag.so vqhip.o vq_screen.o $R/tools/variants/vq_screen_c_$tag.o && rm -f $R/tools/variants/vq_screen_c_$tag.o; echo "built $tag" ) &
done
wait
