#!/bin/bash
# An A/B build of the engine (same ABI) under scripts/_bin/ (git-ignored, travels to the GPU box): never loaded by the product path.
# usage: scripts/build_variant.sh <name> [-DMACRO=VALUE ...]      (SRC=<dir> builds another checkout's sources)
NAME=$1; shift
cd "$(dirname "$0")/.." && mkdir -p scripts/_bin && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -munsafe-fp-atomics -fno-slp-vectorize "$@" \
    -o scripts/_bin/libfe_$NAME.so ${SRC:-.}/fluidlab_amd/csrc/fe_engine.hip && echo built scripts/_bin/libfe_$NAME.so
