#!/bin/bash
# usage: tools/build_tsan.sh  -> csrc/libgs_variant_tsan.so: the library with its HOST code under ThreadSanitizer (-fsanitize=thread applies to the
# host pass of hipcc; the gfx950 code objects are the product's).  Run with GS_SPLAT_LIB=<that file> and the clang TSan runtime preloaded
# (tests/test_stress_gpu.py does: LD_PRELOAD=$(tools/build_tsan.sh --runtime)).
set -e
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.tsan-x86_64.so 2>/dev/null | head -1)
if [ "$1" = "--runtime" ]; then echo $RT; exit 0; fi
R=$(cd "$(dirname "$0")/.." && pwd); C=$R/aframe-gaussian-splatting_amd/csrc
T=$(mktemp -d)
F="-O1 -g -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -fno-fast-math -Wno-unused-function -fsanitize=thread -shared-libsan"
PIDS=""
for f in $(cd $C && ls *.hip | sed s/.hip//); do
  hipcc --offload-arch=gfx950 $F -x hip -c $C/$f.hip -o $T/$f.o & PIDS="$PIDS $!"
done
hipcc $F -x hip -c $C/gs_host.cpp -o $T/gs_host.o & PIDS="$PIDS $!"
# (one wait per job: a plain `wait` returns 0 whatever the jobs did, and a failed compile then shows up as a link error)
for p in $PIDS; do wait $p || { echo "build_tsan.sh: a compile job failed" >&2; rm -rf $T; exit 1; }; done
hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=thread -shared-libsan -o $C/libgs_variant_tsan.so $T/*.o
rm -rf $T; echo built $C/libgs_variant_tsan.so
