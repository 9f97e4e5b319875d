#!/bin/bash
# tools/build_variant.sh NAME [hipcc flags...]  ->  build/variants/NAME.so  (A/B builds: WAVEMAMBA_HIP_LIB=build/variants/NAME.so)
# The library's own flags (wave_mamba_amd/build.py: HIPCC_FLAGS, incl. -fno-slp-vectorize) + the extra ones; "-slp" as an extra flag
# drops -fno-slp-vectorize (the round-4 code generator).
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p build/variants
id=$(python -c "import sys; sys.path.insert(0, '.'); from wave_mamba_amd import build; print(build.source_id())")
flags=$(python -c "import sys; sys.path.insert(0, '.'); from wave_mamba_amd import build; print(' '.join(build.HIPCC_FLAGS))")
extra=()
for a in "$@"; do if [ "$a" = "-slp" ]; then flags=${flags/-fno-slp-vectorize/}; else extra+=("$a"); fi; done
/opt/rocm/bin/hipcc $flags "-DWM_BUILD_ID=\"$id+$name\"" "${extra[@]}" wave_mamba_amd/csrc/wavemamba_hip.hip -o build/variants/$name.so
echo "built build/variants/$name.so ($*)"
