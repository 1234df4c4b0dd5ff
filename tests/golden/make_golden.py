#!/usr/bin/env python3
"""Writes the golden vectors of tests/golden/ from the CPU oracle (oracle/, a restatement of the reference pinned by the
reference's own known-answer tests — tests/test_oracle_voxel.py, tests/test_oracle_physics.py). The reference is Rust and this
image has no cargo/rustc, so it cannot produce vectors itself. Fixtures are data only: digests (sha256 prefixes) of every
output buffer of the voxel path for five small scenes, the exact f64 moments, counts and ranges; rigid-body end states of a
few collision sequences as f32 bit patterns; one scripted scenario over the "next" rows (contact generation against three collidables and
between two objects, collision probes, absorbing sphere / capsule / mutual absorption with the incremental remesh after each).
usage: python tests/golden/make_golden.py     (from the repository root; needs oracle/liboracle.so, built by `make -C oracle`)"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import golden_util as gu  # noqa: E402
import oracle_lib as ol  # noqa: E402


def physics_end_state():
    """BASELINE config 4 at 4^3 bodies: three steps of the sphere pile (the contact list does not depend on the state)"""
    from impact_amd import scenes

    bodies, contacts = scenes.sphere_pile_scene(4)
    o = ol.OraclePhysics(bodies, None, (8, 0.4, 3, 0.2))
    for _ in range(3):
        o.step(contacts, 0.005)
    d, _ = o.bodies()
    return {"scene": "sphere_pile_scene(4)", "steps": 3, "dt": 0.005, "config": [8, 0.4, 3, 0.2], "n_bodies": int(len(d)), "n_contacts": int(len(contacts)),
            "state_f32_bits": gu.sha(np.concatenate([d[f].reshape(len(d), -1) for f in ("position", "orientation", "momentum", "angular_momentum")], axis=1)),
            **{f: [[float(x).hex() for x in row] for row in d[f]] for f in ("position", "orientation", "momentum", "angular_momentum")}}


def main():
    ol.build_oracle()
    vox = {name: gu.oracle_voxel_digest(graph) for name, graph in gu.scenes_small().items()}
    with open(gu.VOXEL_GOLDEN, "w") as f:
        json.dump({"_made_by": "tests/golden/make_golden.py (CPU oracle)", "scenes": vox}, f, indent=1, sort_keys=True)
    phys = {"pile_4": physics_end_state()}
    with open(gu.PHYSICS_GOLDEN, "w") as f:
        json.dump({"_made_by": "tests/golden/make_golden.py (CPU oracle)", "cases": phys}, f, indent=1, sort_keys=True)
    with open(gu.NEXT_GOLDEN, "w") as f:
        json.dump({"_made_by": "tests/golden/make_golden.py (CPU oracle)", "script": gu.next_rows_script(), "digest": gu.oracle_next_rows_digest()}, f, indent=1,
                  sort_keys=True)
    print("wrote", gu.VOXEL_GOLDEN, gu.PHYSICS_GOLDEN, gu.NEXT_GOLDEN)


if __name__ == "__main__":
    main()
