#!/bin/bash
# Builds another copy of the library with extra compiler flags: scripts/build_variant.sh <name> [flags...]  -> build_exp/lib<name>.so
# (select it at run time with RT355_LIB=build_exp/lib<name>.so)
name=$1; shift
cd "$(dirname "$0")/.." && mkdir -p build_exp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared "$@" -o build_exp/lib$name.so raytracer_amd/csrc/rt355.hip raytracer_amd/csrc/scene_build.cpp raytracer_amd/csrc/rt355_multi.cpp -ldl -lpthread
