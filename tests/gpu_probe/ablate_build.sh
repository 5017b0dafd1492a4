#!/bin/bash
# Here (no GPU needed): the timing-only ablation builds tests/gpu_probe/ablate.sh runs on the GPU box.
#   bash tests/gpu_probe/ablate_build.sh 1 2 4 8 16      ->  build/ab/libimk_abl<bits>.so  (imk_conv.hip with -DIMK_ABL=<bits>, the other
#   objects of the in-tree build; run `python -c "import __graft_entry__ as g; g.build()"` first)
cd "$(dirname "$0")/../.."
mkdir -p build/ab
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result -fvisibility=hidden -mllvm -amdgpu-mfma-vgpr-form"
objs=$(ls build/obj/*.o | grep -v imk_conv.hip.o)
for v in "$@"; do
  ( /opt/rocm/bin/hipcc $FLAGS -DIMK_ABL=$v -c inconsistencymasks_amd/csrc/imk_conv.hip -o build/ab/conv_abl$v.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/ab/libimk_abl$v.so build/ab/conv_abl$v.o $objs && echo build/ab/libimk_abl$v.so ) &
done
wait
