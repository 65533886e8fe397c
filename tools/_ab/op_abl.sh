#!/bin/bash
# diagnostic libraries of the one-pass attention backward (build container): tools/_ab/liba4r_op_<tag>.so with EXTRA flags, e.g.
#   TAG=st EXTRA=-DA4R_OP_STAMP=1 bash tools/_ab/op_abl.sh        (in-kernel stamps: tools/op_stamps.py)
set -e
cd "$(dirname "$0")/../.."
TAG=${TAG:-x}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-unused-function ${EXTRA:-} -c adapter4rec_amd/csrc/a4r_attn_long1.hip -o /tmp/a4r_attn_long1_$TAG.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_ab/liba4r_op_$TAG.so /tmp/a4r_attn_long1_$TAG.o $(ls adapter4rec_amd/csrc/*.o | grep -v "a4r_attn_long1.o\|\.w4\.o")
echo built tools/_ab/liba4r_op_$TAG.so
