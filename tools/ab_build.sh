#!/bin/bash
# Build a variant of libtavhip.so with extra compiler flags into build/ab/<name>.so (git-ignored, travels with gpurun):
#   tools/ab_build.sh base                      -> build/ab/base.so  (current sources, default flags)
#   tools/ab_build.sh nohoist -DTAV_HOIST_BWD=0 -> build/ab/nohoist.so
# then on the GPU box:  TAV_LIB=build/ab/base.so python tools/gpu_ab.py attn ; TAV_LIB=build/ab/nohoist.so python tools/gpu_ab.py attn
set -e
name=$1; shift
root="$(cd "$(dirname "$0")/.." && pwd)"
out="$root/build/ab"; mkdir -p "$out/obj_$name"
cd "$root/multi-modal-emotion_amd/csrc"
for f in gemm attention norm elementwise audio_frontend fp8 collective; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable -I../../include "$@" -c $f.hip -o "$out/obj_$name/$f.o" &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC "$out/obj_$name"/*.o -ldl -o "$out/$name.so"
echo "$out/$name.so"
