#!/bin/bash
# build_variant.sh NAME [-DMACRO ...]: libnaf_hip.so with extra macros for render_fused.hip (the only source the scatter / MLP kernels
# live in) -> neuralvolumetricreconstructionformedicalimages_amd/lib/ab/NAME.so; the other objects are taken from lib/obj (build the
# default library first).  For same-box A/B runs with tools/ab_libs.sh.
set -e
NAME=$1; shift
P=/root/repo/neuralvolumetricreconstructionformedicalimages_amd
mkdir -p $P/lib/ab /tmp/naf_var_$NAME
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wall -Wno-unused-function -fno-slp-vectorize "$@" -c $P/csrc/render_fused.hip -o /tmp/naf_var_$NAME/render_fused.o
OBJS=$(ls $P/lib/obj/*.o | grep -v render_fused.o)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $P/lib/ab/$NAME.so /tmp/naf_var_$NAME/render_fused.o $OBJS
echo built $P/lib/ab/$NAME.so
