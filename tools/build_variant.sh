#!/bin/bash
# Developer tool: a second build of the library with extra compiler flags (experiment switches), as a file of its own that a run selects through
# IMPACT_VOXEL_HIP_LIB; the product library is not touched. usage: tools/build_variant.sh <name> <flags...>  ->  impact_amd/lib/var_<name>/libimpact_voxel_hip.so
set -eu
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p impact_amd/lib/var_$name
make -C impact_amd/csrc -j8 BUILD=build_var_$name EXTRA="$*" LIB=../lib/var_$name/libimpact_voxel_hip.so FLAVOUR=../lib/var_$name/.flavour 2>&1 | grep -E "error|Error" || true
ls -la impact_amd/lib/var_$name/libimpact_voxel_hip.so
