#!/bin/bash
# profiling build of the engine with in-kernel phase stamps (scripts/timeline.py); never loaded by the product path
cd "$(dirname "$0")/.." && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -munsafe-fp-atomics -fno-slp-vectorize -DFE_TIMELINE \
    -o fluidlab_amd/csrc/libfluidengine_tl_hip.so fluidlab_amd/csrc/fe_engine.hip
