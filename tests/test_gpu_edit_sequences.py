"""Random edit sequences, in the spirit of the reference's fuzz targets (fuzz_test_absorbing_voxels_within_sphere / _capsule,
object/intersection.rs:955-1100): a dozen absorbing spheres and capsules at random places of an irregular body, each followed by the
incremental remesh and the probe sync; after every edit the object (voxels, chunk records, regions), the synced mesh and the probes of the
HIP path equal the oracle's. Seeds are fixed: the sequences are data, not chance."""
import numpy as np
import pytest

import oracle_lib as ol
import parity_util as pu
from impact_amd import scenes
from impact_amd.voxel import VoxelObjectMesh

pytestmark = pytest.mark.gpu


# (105940: found by round 5's sweep — a tiny edit followed by a long capsule, densities not resident: the table uploaded behind the small edit's
# results was read as touched words by the large one, five chunks invalidated for nothing)
@pytest.mark.parametrize("seed", pu.fuzz_seeds([1, 2, 3, 4, 5, 6, 105940]))
def test_random_edit_sequence(ctx, seed):
    rng = np.random.default_rng(seed)
    graph = scenes.asteroid_scene(0.3) if seed % 3 else scenes.box_scene((40.0, 26.0, 33.0))
    o = pu.oracle_from_graph(graph, 1.0)
    g = pu.gpu_from_graph(ctx, graph, 1.0)
    o.update_occupied_voxel_ranges()
    o.compute_all_derived_state()
    g.compute_all_derived_state()
    g.update_occupied_voxel_ranges()
    g.label_regions()
    om, gm = ol.OracleMeshHandle(o), VoxelObjectMesh.create(g)
    op = ol.OracleProbes(om)
    g.collision_probes_recompute()
    occ = np.array(o.info()["occupied_voxel_ranges"], dtype=np.float64)
    lo, hi = occ[:, 0], occ[:, 1]
    emptied = 0
    for step in range(12):
        p = (lo + rng.uniform(-0.1, 1.1, 3) * (hi - lo)).astype(np.float32)
        r = float(np.float32(rng.uniform(2.0, 9.0)))
        if rng.uniform() < 0.5:
            ro, rg = o.absorb_sphere(p, r + 2.0, r), g.absorb_sphere(p, r + 2.0, r)
        else:
            v = (rng.normal(size=3) * rng.uniform(0.0, 25.0)).astype(np.float32)
            if step == 5:
                v[:] = 0.0  # a degenerate capsule now and then
            ro, rg = o.absorb_capsule(p, v, r + 2.0, r), g.absorb_capsule(p, v, r + 2.0, r)
        assert rg["touched_chunks"] == ro["touched_chunks"] and rg["removed_chunks"] == ro["removed_chunks"], step
        np.testing.assert_array_equal(rg["emptied_by_type"], ro["emptied_by_type"])
        np.testing.assert_array_equal(rg["invalidated"], ro["invalidated"])
        emptied += int(ro["emptied_by_type"].sum())
        pu.assert_edited_objects_equal(o, g, with_mesh=False)
        om.sync(ro["invalidated"])
        op.sync(ro["invalidated"])
        gm.sync_with_voxel_object(rg["invalidated"])
        g.collision_probes_sync(rg["invalidated"])
        want = om.get()
        pos, nrm, idx, im, sub = gm.download()
        assert len(sub) == len(want.submeshes) and len(pos) == len(want.positions) and len(idx) == len(want.indices), step
        for col, f in ((3, "index_offset"), (4, "index_count"), (13, "vertex_offset"), (14, "vertex_count")):
            np.testing.assert_array_equal(sub[f], want.submeshes[:, col], err_msg=f"{f} at step {step}")
        for sm in want.submeshes:
            ioff, icnt, voff, vcnt = int(sm[3]), int(sm[4]), int(sm[13]), int(sm[14])
            np.testing.assert_array_equal(pos[voff:voff + vcnt].view(np.uint32), want.positions[voff:voff + vcnt].view(np.uint32))
            np.testing.assert_array_equal(idx[ioff:ioff + icnt], want.indices[ioff:ioff + icnt])
        want_pts, want_ent = op.get()
        got_pts, got_ent = g.collision_probes()
        np.testing.assert_array_equal(got_ent, want_ent)
        for e in want_ent:
            np.testing.assert_array_equal(got_pts[e[3]:e[4]].view(np.uint32), want_pts[e[3]:e[4]].view(np.uint32))
    assert emptied > 1000 or pu.fuzzing()
    g.close()
