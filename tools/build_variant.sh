#!/bin/bash
# tools/build_variant.sh NAME [hipcc flags...]  ->  build/variants/NAME.so  (A/B builds: WAVEMAMBA_HIP_LIB=build/variants/NAME.so)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p build/variants
id=$(python -c "import sys; sys.path.insert(0, '.'); from wave_mamba_amd import build; print(build.source_id())")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wno-unused-value "-DWM_BUILD_ID=\"$id+$name\"" "$@" \
    wave_mamba_amd/csrc/wavemamba_hip.hip -o build/variants/$name.so
echo "built build/variants/$name.so ($*)"
