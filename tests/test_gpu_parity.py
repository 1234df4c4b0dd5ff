"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on identical inputs.

Bars (BASELINE.json north_star): voxel bytes, flags, chunk state, triangle index buffers, vertex
positions and index materials bit-exact; component labels equal after canonical relabelling; inertia
moments within 1e-5 relative of the f64 oracle.
"""
import numpy as np
import pytest

import oracle_lib as ol
import parity_util as pu
from impact_amd import scenes
from impact_amd.sdf_graph import SDFGraph, SDFNode
from impact_amd.voxel import VoxelObject

pytestmark = pytest.mark.gpu


def full_pipeline(ctx, graph, extent=1.0, expect_regions=None):
    o = pu.oracle_from_graph(graph, extent)
    g = pu.gpu_from_graph(ctx, graph, extent)
    pu.assert_generated_equal(o, g)
    o.update_occupied_voxel_ranges()
    o.compute_all_derived_state()
    g.compute_all_derived_state()
    pu.assert_derived_equal(o, g)
    g.update_occupied_voxel_ranges()
    info = o.info()
    assert g.occupied_voxel_ranges == info["occupied_voxel_ranges"]
    assert g.occupied_chunk_ranges == info["occupied_chunk_ranges"]
    pu.assert_mesh_equal(o, g)
    pu.assert_inertia_equal(o, g)
    n = pu.assert_regions_equal(o, g)
    if expect_regions is not None:
        assert n == expect_regions
    return o, g


def test_config1_box_32(ctx):
    """BASELINE config 1: Box([30,30,30]) -> 32^3, 8 chunks."""
    full_pipeline(ctx, scenes.box_scene(), expect_regions=1)


def test_box_extent_point1(ctx):
    """inertia.rs:857-907 setting: 22x27x19 box, voxel extent 0.1 (non-trivial f32 positions)."""
    full_pipeline(ctx, scenes.box_scene((22.0, 27.0, 19.0)), extent=0.1, expect_regions=1)


def test_small_sphere(ctx):
    full_pipeline(ctx, scenes.sphere_scene(20.0), expect_regions=1)


def test_capsule_rotated_scaled(ctx):
    g = SDFGraph()
    c = g.add_node(SDFNode.new_capsule(20.0, 9.0))
    r = g.add_node(SDFNode.new_rotation_from_axis_angle(c, (1.0, 2.0, 3.0), 0.7))
    s = g.add_node(SDFNode.new_scaling(r, 1.3))
    g.add_node(SDFNode.new_translation(s, (3.25, -1.5, 0.75)))
    full_pipeline(ctx, g, extent=0.5)


def test_intersection_and_smooth_ops(ctx):
    g = SDFGraph()
    a = g.add_node(SDFNode.new_sphere(22.0))
    b = g.add_node(SDFNode.new_box((30.0, 50.0, 30.0)))
    i = g.add_node(SDFNode.new_intersection(a, b, 3.0))
    c = g.add_node(SDFNode.new_sphere(9.0))
    tc = g.add_node(SDFNode.new_translation(c, (14.0, 0.0, 0.0)))
    g.add_node(SDFNode.new_subtraction(i, tc, 2.0))
    full_pipeline(ctx, g)


def test_two_spheres_split(ctx):
    """extraction.rs:2587-2624: two r=25 spheres 60 apart are two regions."""
    o, g = full_pipeline(ctx, scenes.two_spheres_scene(), expect_regions=2)
    r = g.describe_regions()
    assert len(r) == 2 and r["voxel_count"][0] == r["voxel_count"][1]
    assert g.find_two_disconnected_regions() is not None


def test_config2_asteroid_256(ctx):
    """BASELINE config 2: 256^3 asteroid — remesh + inertia."""
    o, g = full_pipeline(ctx, scenes.asteroid_scene(), expect_regions=1)
    assert g.chunk_counts == (16, 16, 16)


def test_config3_fracture_256(ctx):
    """BASELINE config 3: config-2 body cut into exactly 8 components."""
    o, g = full_pipeline(ctx, scenes.fracture_scene(), expect_regions=8)
    d = np.ones(256, dtype=np.float32)
    r = g.describe_regions(d)
    assert len(r) == 8
    _, o64 = o.inertia(d)
    np.testing.assert_allclose(r["moments"].sum(axis=0), o64, rtol=1e-9)
    _, olab = o.region_labels()
    counts = np.bincount(olab[olab != 0xFFFFFFFF])
    np.testing.assert_array_equal(np.sort(r["voxel_count"]), np.sort(counts))


@pytest.mark.parametrize("seed,level", [(0, 0.5), (1, 0.5), (2, 0.5), (3, 0.56), (4, 0.6)])
def test_random_dense_upload(ctx, seed, level):
    """ragged random voxels through the dense upload path: many local regions, mixed faces, the
    empty-voxel outward-flag case, multiple voxel types. `level` 0.5: chunks with thousands of mesh vertices of several materials
    (the mesher's quad phase reads its vertices back from memory); higher: sparse chunks of several materials (it keeps them in LDS)."""
    rng = np.random.default_rng(seed)
    cc = (3, 2, 3)
    n = cc[0] * cc[1] * cc[2] * 4096
    blobs = rng.random((cc[0] * 16, cc[1] * 16, cc[2] * 16))
    # smooth a little so that regions are not single voxels
    for ax in range(3):
        blobs = 0.5 * blobs + 0.25 * (np.roll(blobs, 1, ax) + np.roll(blobs, -1, ax))
    sd = np.where(blobs > level, -128, np.where(blobs > level - 0.03, rng.integers(-60, -1, blobs.shape), rng.integers(0, 127, blobs.shape))).astype(np.int8)
    sd[16:32, :, 0:16] = -128  # a solid chunk (uniform) next to ragged neighbours
    sd[32:48, 16:32, 32:48] = 127  # a void chunk
    ty = rng.integers(0, 5, blobs.shape).astype(np.uint8)
    ty[16:32, :, 0:16] = 3
    sd_t, ty_t = ol.dense_to_tiled(sd), ol.dense_to_tiled(ty)
    o = ol.OracleObject.from_dense(cc, sd_t, ty_t, 0.25)
    g = VoxelObject.from_dense(ctx, cc, sd_t, ty_t, 0.25)
    pu.assert_generated_equal(o, g)
    o.update_occupied_voxel_ranges()
    o.compute_all_derived_state()
    g.compute_all_derived_state()
    pu.assert_derived_equal(o, g)
    pu.assert_mesh_equal(o, g)
    dens = np.linspace(0.5, 3.0, 256).astype(np.float32)
    pu.assert_inertia_equal(o, g, dens)
    pu.assert_regions_equal(o, g)
    assert n == g.n_voxels


def test_empty_and_full_objects(ctx):
    cc = (2, 2, 2)
    n = 8 * 4096
    for fill in (127, -128):
        sd = np.full(n, fill, dtype=np.int8)
        ty = np.zeros(n, dtype=np.uint8)
        o = ol.OracleObject.from_dense(cc, sd, ty, 1.0)
        g = VoxelObject.from_dense(ctx, cc, sd, ty, 1.0)
        o.update_occupied_voxel_ranges()
        o.compute_all_derived_state()
        g.compute_all_derived_state()
        pu.assert_derived_equal(o, g)
        pu.assert_mesh_equal(o, g)
        pu.assert_regions_equal(o, g)
        g.update_occupied_voxel_ranges()
        assert g.occupied_voxel_ranges == o.info()["occupied_voxel_ranges"]


# ---- shapes that stress the list-driven / column-structured kernels ---------------------------------------------------------
def _boxes_along(axis, centres, half=(5.0, 5.0, 5.0), long_half=None, smooth=0.0):
    """union of boxes centred at `centres` along `axis`; `long_half` = half extent along the axis"""
    g = SDFGraph()
    acc = None
    for c in centres:
        h = list(half)
        if long_half is not None:
            h[axis] = long_half
        b = g.add_node(SDFNode.new_box([2.0 * x for x in h]))
        t = [0.0, 0.0, 0.0]
        t[axis] = float(c)
        tb = g.add_node(SDFNode.new_translation(b, t))
        acc = tb if acc is None else g.add_node(SDFNode.new_union(acc, tb, smooth))
    return g


@pytest.mark.parametrize("axis", [0, 1, 2])
def test_long_thin_rod_more_than_64_chunks(ctx, axis):
    """one rod of 1100 voxels: a chunk column of 70 chunks (the column merge works in segments of 64 chunks), grids of
    1 x 1 x 70 chunks (fewer chunks than a workgroup has threads, chunk counts that are not multiples of a super-block)"""
    full_pipeline(ctx, _boxes_along(axis, [0.0], long_half=550.0), expect_regions=1)


def test_three_rods_in_one_column(ctx):
    """three separate bodies stacked along z in the same chunk column, the column longer than 64 chunks: three regions, runs
    that start and stop inside and across the 64-chunk segments"""
    full_pipeline(ctx, _boxes_along(2, [-420.0, 0.0, 420.0], long_half=190.0), expect_regions=3)


def test_row_of_many_small_bodies(ctx):
    """24 bodies in a row (a long program: far bodies are replaced by their folded constants per super-block), every second
    pair joined by a smooth union"""
    g = SDFGraph()
    acc = None
    for i in range(24):
        s = g.add_node(SDFNode.new_sphere(9.0 + (i % 3)))
        t = g.add_node(SDFNode.new_translation(s, (26.0 * i, 3.0 * ((i * 7) % 5 - 2), 2.0 * ((i * 3) % 4))))
        acc = t if acc is None else g.add_node(SDFNode.new_union(acc, t, 6.0 if i % 2 else 0.0))
    full_pipeline(ctx, g)


def test_plate_one_chunk_thick(ctx):
    """a 300 x 300 x 6 plate: every chunk is a surface chunk, no Uniform chunks at all"""
    full_pipeline(ctx, scenes.box_scene((300.0, 300.0, 6.0)), expect_regions=1)


def test_bench_workload_512(ctx):
    """The workload bench.py times at N=1 — the config-2 asteroid under a root Scaling x2.05, 512^3 stored voxels — through the same
    entry point (`ivx_voxel_step(STAGE_ALL)` over the resident program) against the oracle: voxel bytes, chunk records, raw
    chunk-local labels, index buffer, vertex data, region count, moments."""
    from impact_amd import capi
    from impact_amd.voxel import SDFVoxelGenerator

    graph = scenes.asteroid_scene(2.05)
    gen = SDFVoxelGenerator(1.0, graph, 0)
    assert gen.chunk_counts() == (32, 32, 32)
    obj = VoxelObject(ctx, gen.chunk_counts(), 1.0)
    obj.set_sdf_program(gen)
    obj.set_densities(np.ones(256, dtype=np.float32))
    obj.step(capi.STAGE_ALL)
    res = obj.step(capi.STAGE_ALL)  # (the second step runs with the list-sized grids of a steady-state frame)
    o = pu.oracle_from_graph(graph)
    o.update_occupied_voxel_ranges()
    o.compute_all_derived_state()
    p = pu.step_parity(o, obj, res)
    assert p["equal"], p
    info = o.info()
    occ = np.asarray(res["occupied"]).reshape(-1)
    assert [(int(occ[6 + 2 * d]), int(occ[7 + 2 * d])) for d in range(3)] == info["occupied_voxel_ranges"]
    obj.close()


def test_dense_workload_512(ctx):
    """bench.py's all-surface workload at full size — 32 perforated plates, every chunk of the 512^3 grid NonUniform and meshed
    (33 M triangles) — through `ivx_voxel_step(STAGE_ALL)` against the oracle."""
    from impact_amd import capi
    from impact_amd.voxel import SDFVoxelGenerator

    graph = scenes.plates_scene(32)
    gen = SDFVoxelGenerator(1.0, graph, 0)
    assert gen.chunk_counts() == (32, 32, 32)
    obj = VoxelObject(ctx, gen.chunk_counts(), 1.0)
    obj.set_sdf_program(gen)
    obj.set_densities(np.ones(256, dtype=np.float32))
    obj.step(capi.STAGE_ALL)
    res = obj.step(capi.STAGE_ALL)
    o = ol.OracleObject.from_sdf_parallel(graph, 1.0, 0, 8)  # (identical to the sequential oracle: tests/test_oracle_parallel.py)
    p = pu.step_parity(o, obj, res)
    assert p["equal"], p
    assert p["regions"][0] == 32
    obj.close()


def test_program_swap_on_a_resident_grid(ctx):
    """a new program on a resident grid, and the same program stepped repeatedly (the sampler's list counters are rolled over by the
    derive sweep of the step before, its scratch presets ride in the pre-pass): every step must give the object a fresh grid gives"""
    from impact_amd import capi
    from impact_amd.voxel import SDFVoxelGenerator

    def two_bodies(offset):
        g = SDFGraph()
        a = g.add_node(SDFNode.new_sphere(20.0))
        b = g.add_node(SDFNode.new_translation(g.add_node(SDFNode.new_box((9.0, 14.0, 7.0))), (offset, 3.0, -4.0)))
        g.add_node(SDFNode.new_union(a, b, 2.0))
        return g

    gen_a, gen_b = SDFVoxelGenerator(1.0, two_bodies(12.0), 0), SDFVoxelGenerator(1.0, two_bodies(-15.0), 0)
    cc = tuple(max(x, y) for x, y in zip(gen_a.chunk_counts(), gen_b.chunk_counts()))
    dens = np.ones(256, dtype=np.float32)
    obj = VoxelObject(ctx, cc, 1.0)
    obj.set_densities(dens)
    fresh = {}
    for name, gen in (("a", gen_a), ("b", gen_b)):
        f = VoxelObject(ctx, cc, 1.0)
        f.set_densities(dens)
        f.set_sdf_program(gen)
        r = f.step(capi.STAGE_ALL)
        fresh[name] = (f.download(), int(r["mesh"]["n_indices"]), int(r["region_count"]), np.array(r["moments"]["m64"]))
        f.close()
    for name, gen in (("a", gen_a), ("a", gen_a), ("b", gen_b), ("b", gen_b), ("a", gen_a)):
        obj.set_sdf_program(gen) if name != getattr(obj, "_last", None) else None
        obj._last = name
        r = obj.step(capi.STAGE_ALL)
        want = fresh[name]
        got = obj.download()
        for x, y in zip(got[:4], want[0][:4]):
            np.testing.assert_array_equal(x, y)
        assert int(r["mesh"]["n_indices"]) == want[1] and int(r["region_count"]) == want[2]
        np.testing.assert_array_equal(np.array(r["moments"]["m64"]), want[3])
    obj.close()


@pytest.mark.parametrize("scene", ["asteroid", "fracture"])
def test_step_results_with_and_without_stage_events(ctx, scene):
    """The step's record is the same whichever timed slots carry event records (none: what `bench.py` times and what an engine runs), step after
    step, whether it is driven as one call or as enqueue + collect, and the buffers behind it are complete when collect returns. (Folding the
    result gather and the doorbell into the last block of the step's last launch was tried against this test: correct, and 5 us SLOWER per step
    than the 4 us gather launch it removed — every block's device-scope fence costs more than the launch.)"""
    from impact_amd import capi
    from impact_amd.voxel import SDFVoxelGenerator

    graph = scenes.asteroid_scene() if scene == "asteroid" else scenes.fracture_scene()
    gen = SDFVoxelGenerator(1.0, graph, 0)
    obj = VoxelObject(ctx, gen.chunk_counts(), 1.0)
    obj.set_sdf_program(gen)
    obj.set_densities(np.ones(256, dtype=np.float32))
    want = obj.step(capi.STAGE_ALL).copy()
    labels_want = obj.region_labels().copy()  # (global component ids: what the step's last launch writes)
    assert int(want["region_count"]) == (1 if scene == "asteroid" else 8)
    fields = [f for f in want.dtype.names if f != "stage_ms"]
    for timing in (0, 0, 0xFFFFFFFF, 0, 1 << 4, 0):
        obj.set_stage_timing(timing)
        got = obj.step(capi.STAGE_ALL)
        for f in fields:
            np.testing.assert_array_equal(np.asarray(got[f]), np.asarray(want[f]), err_msg=f"{f} (timing mask {timing:#x})")
        np.testing.assert_array_equal(obj.region_labels(), labels_want)
    obj.set_stage_timing(0)
    for _ in range(3):  # enqueue / collect apart, as the bench drives it
        obj.step_enqueue(capi.STAGE_ALL)
        got = obj.step_collect()
        for f in fields:
            np.testing.assert_array_equal(np.asarray(got[f]), np.asarray(want[f]), err_msg=f)


def test_sample_of_another_program_after_a_stepped_object(ctx):
    """`ivx_sdf_sample` runs ANY program through the sampler's launcher; the list lengths the step path remembers belong to the
    resident program (A: one sphere, every chunk in the one-level class) and must not decide which evaluation classes a different
    program (B: nested smooth unions that need three levels) gets launched with."""
    from impact_amd import capi
    from impact_amd.voxel import SDFVoxelGenerator

    def offset_sphere(g, r, t):
        return g.add_node(SDFNode.new_translation(g.add_node(SDFNode.new_sphere(r)), t))

    gb = SDFGraph()
    c = offset_sphere(gb, 20.0, (-6.0, 0.0, 0.0))
    d = offset_sphere(gb, 20.0, (6.0, 0.0, 0.0))
    cd = gb.add_node(SDFNode.new_union(c, d, 6.0))
    e = offset_sphere(gb, 20.0, (0.0, 6.0, 0.0))
    f = offset_sphere(gb, 20.0, (0.0, -6.0, 0.0))
    h = offset_sphere(gb, 20.0, (0.0, 0.0, 6.0))
    fh = gb.add_node(SDFNode.new_union(f, h, 6.0))
    efh = gb.add_node(SDFNode.new_union(e, fh, 6.0))
    gb.add_node(SDFNode.new_union(cd, efh, 6.0))
    gen_b = SDFVoxelGenerator(1.0, gb, 0)
    gen_a = SDFVoxelGenerator(1.0, scenes.sphere_scene(18.0), 0)
    assert all(a <= b for a, b in zip(gen_a.grid_shape(), gen_b.grid_shape()))
    obj = VoxelObject(ctx, gen_b.chunk_counts(), 1.0)
    obj.set_sdf_program(gen_a)
    obj.set_densities(np.ones(256, dtype=np.float32))
    for _ in range(2):  # the lengths of A's lists are known after a collected sample + derive
        obj.step(capi.STAGE_ALL)
    obj.sample(gen_b)
    o = pu.oracle_from_graph(gb)
    pu.assert_generated_equal(o, obj)
    o.compute_all_derived_state()
    obj.compute_all_derived_state()
    pu.assert_derived_equal(o, obj)
    # and the resident program still steps to its own result afterwards
    obj.step(capi.STAGE_ALL)
    oa = ol.OracleObject.from_sdf(scenes.sphere_scene(18.0), 1.0, 0)
    sdf_a = oa.export_dense()[0]
    assert int((obj.download()[0] < 0).sum()) == int((sdf_a < 0).sum())
    obj.close()
