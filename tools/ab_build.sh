#!/bin/bash
# dev tool: build a variant of librrt_hip.so with extra hipcc flags into relativisticraytracer_amd/lib/variants/<name>.so
name=$1; shift
mkdir -p relativisticraytracer_amd/lib/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -fno-gpu-rdc "$@" relativisticraytracer_amd/csrc/rrt_hip.hip -o relativisticraytracer_amd/lib/variants/$name.so
