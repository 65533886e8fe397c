#!/bin/bash
# Build tools/_ab/liba4r_w4_<variant>[_abl<N>].so for each "variant[:abl]" argument: the objects of `make W4=1` (adapter4rec_amd/csrc/*.w4.o) with a4r_gemm256w4.w4.o replaced by a build
# against a freshly generated K loop (FORMS=0, default: plain epilogue form only).  Run in the build container; the libraries travel to the GPU box.
#   bash tools/w4/w4_variants.sh v1 v2 v2:1 v2:2 ...      then on the GPU:  A4R_LIB_PATH=tools/_ab/liba4r_w4_v2.so python tools/w4/w4_check.py 40448 2 quick
set -e
# (the library's other objects must carry the a4r_gemm_variant(8 / 9) switch: make -C adapter4rec_amd/csrc W4=1 first)
cd "$(dirname "$0")/../.."
R=$PWD
mkdir -p tools/_ab /tmp/w4v
for spec in "$@"; do
  v=${spec%%:*}; abl=0; [[ "$spec" == *:* ]] && abl=${spec##*:}
  tag=$v; [ "$abl" != 0 ] && tag=${v}_abl$abl; [ -n "$STAMP" ] && tag=${tag}_st
  d=/tmp/w4v/$tag; mkdir -p $d
  cp tools/w4/a4r_gemm256w4.hip $d/
  python tools/w4/gen_gemm_w4_loop.py --variant $v --abl $abl --out $d/a4r_gemm256w4_loop.inc > /dev/null
  ( cd $d && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -DA4R_W4_FORMS=${FORMS:-0} ${STAMP:+-DA4R_W4_STAMP=1} ${EXTRA:-} -I$R/adapter4rec_amd/csrc -c a4r_gemm256w4.hip -o w4.o )
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_ab/liba4r_w4_$tag.so $d/w4.o $(ls adapter4rec_amd/csrc/*.w4.o | grep -v a4r_gemm256w4.w4.o)
  echo built tools/_ab/liba4r_w4_$tag.so
done
