#!/bin/bash
# Does the oracle stand under the batched contact forms? Builds the library a second time with one ulp added to every contact's depth in the
# device code that the single-object calls and their `_many` twins share (collide.hip / contacts.hip, -DIVX_MUTATION_CHECK), and runs the batched
# forms' tests against it: they must FAIL (the comparisons with the HIP twins alone would pass). The mutated library is a file of its own
# (impact_amd/lib/mut/) selected through IMPACT_VOXEL_HIP_LIB; the product library is not touched.
#   on this container:  tools/mutation_check.sh build      (cross-compiles; the .so travels with the gpurun snapshot)
#   on the GPU box:     tools/mutation_check.sh run        -> gpurun_out/mutation_check.log
set -u
cd "$(dirname "$0")/.."
case "${1:-}" in
build)
    mkdir -p impact_amd/lib/mut
    make -C impact_amd/csrc -j8 BUILD=build_mut EXTRA=-DIVX_MUTATION_CHECK LIB=../lib/mut/libimpact_voxel_hip.so FLAVOUR=../lib/mut/.flavour >/dev/null || exit 1
    ls -la impact_amd/lib/mut/libimpact_voxel_hip.so
    ;;
run)
    mkdir -p gpurun_out
    IMPACT_VOXEL_HIP_LIB=$PWD/impact_amd/lib/mut/libimpact_voxel_hip.so python -m pytest -q -m gpu -p no:cacheprovider \
        "tests/test_gpu_collide.py::test_mutual_contacts_of_many_pairs" "tests/test_gpu_contacts.py::test_many_objects_against_a_collidable_each" \
        "tests/test_gpu_many.py::test_random_batches_of_contacts_pairs_and_probe_syncs" "tests/test_golden.py::test_hip_matches_next_rows_golden" \
        > gpurun_out/mutation_check.log 2>&1
    rc=$?
    tail -8 gpurun_out/mutation_check.log
    if [ $rc -eq 0 ]; then echo "MUTATION SURVIVED: the tests passed against the mutated library" | tee -a gpurun_out/mutation_check.log; exit 1; fi
    echo "mutation caught (pytest exit $rc): the batched forms are held to the oracle" | tee -a gpurun_out/mutation_check.log
    ;;
*) echo "usage: $0 build|run"; exit 2 ;;
esac
