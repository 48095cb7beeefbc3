#!/bin/bash
# build a variant of the library: tools/build_variant.sh <name> [-DFLAG=..]...   -> repo_amd/variants/lib_<name>.so
# only conv.hip (or SRC=...) is recompiled with the flags (ablation macros such as -DBG_NO_SPLIT; tile choices are edited in the source since round 5); the other objects come from repo_amd/csrc/build
N=$1; shift
D=repo_amd/variants; mkdir -p $D
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result "$@" -c repo_amd/csrc/${SRC:-conv}.hip -o $D/${SRC:-conv}_$N.o || exit 1
OBJS=$(ls repo_amd/csrc/build/*.o | grep -v "/${SRC:-conv}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/lib_$N.so $OBJS $D/${SRC:-conv}_$N.o && echo built $D/lib_$N.so
