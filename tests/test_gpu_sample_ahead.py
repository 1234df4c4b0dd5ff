"""Sample-ahead (`ivx_grid_set_sample_ahead`): the next sample stage's interval pre-pass rides on the context's second stream behind this
stage's evaluator, into a second set of the sampler's buffers, and the records it settles wait in a shadow array until the next evaluator
launch commits them. Whatever order the steps come in, what a step leaves on the device must be what the oracle computes."""
import numpy as np
import pytest

import parity_util as pu
from impact_amd import capi, scenes
from impact_amd.sdf_graph import SDFGraph, SDFNode
from impact_amd.voxel import SDFVoxelGenerator, VoxelObject

pytestmark = pytest.mark.gpu


def resident_object(ctx, graph, ahead):
    gen = SDFVoxelGenerator(1.0, graph, 0)
    obj = VoxelObject(ctx, gen.chunk_counts(), 1.0)
    obj.set_sdf_program(gen)
    obj.set_densities(np.ones(256, dtype=np.float32))
    obj.set_sample_ahead(ahead)
    return gen, obj


def oracle_of(graph):
    o = pu.oracle_from_graph(graph)
    o.update_occupied_voxel_ranges()
    o.compute_all_derived_state()
    return o


def craters(n=3):
    g = SDFGraph()
    acc = g.add_node(SDFNode.new_sphere(30.0))
    for i in range(n):
        s = g.add_node(SDFNode.new_sphere(9.0))
        t = g.add_node(SDFNode.new_translation(s, (26.0 - 20.0 * i, 8.0 * i, -6.0 * i)))
        acc = g.add_node(SDFNode.new_subtraction(acc, t, 2.0))
    return g


def test_steps_with_the_pre_pass_a_step_ahead_are_the_same_steps(ctx):
    """Config 2's asteroid: five steps in a row (the first runs its own pre-pass, the others find theirs done), each against the oracle;
    a twin without sample-ahead gives the same step results."""
    graph = scenes.asteroid_scene(1.0)
    o = oracle_of(graph)
    _, a = resident_object(ctx, graph, True)
    _, b = resident_object(ctx, graph, False)
    for i in range(5):
        ra, rb = a.step(capi.STAGE_ALL), b.step(capi.STAGE_ALL)
        p = pu.step_parity(o, a, ra)
        assert p["equal"], (i, p)
        assert ra["mesh"] == rb["mesh"] and ra["region_count"] == rb["region_count"]
        np.testing.assert_array_equal(np.asarray(ra["occupied"]), np.asarray(rb["occupied"]))
        np.testing.assert_array_equal(np.asarray(ra["moments"]["m64"]), np.asarray(rb["moments"]["m64"]))
    assert a.stage_counters()["evaluated_chunks"] == b.stage_counters()["evaluated_chunks"]
    a.close()
    b.close()


def test_other_steps_in_between_and_a_new_program(ctx):
    """Between two sampled steps: a step without the sample stage, a remesh alone, then ANOTHER program (the pre-pass under way was made for
    the old one and must be dropped), then the first program again."""
    g1, g2 = craters(3), craters(1)
    gen1 = SDFVoxelGenerator(1.0, g1, 0)
    gen2 = SDFVoxelGenerator(1.0, g2, 0)
    cc = tuple(max(a_, b_) for a_, b_ in zip(gen1.chunk_counts(), gen2.chunk_counts()))
    assert gen1.chunk_counts() == cc and gen2.chunk_counts() == cc  # (a subtraction's domain is its first operand's: both are the sphere's)
    o1, o2 = oracle_of(g1), oracle_of(g2)
    obj = VoxelObject(ctx, cc, 1.0)
    obj.set_densities(np.ones(256, dtype=np.float32))
    obj.set_sample_ahead(True)
    obj.set_sdf_program(gen1)
    r = obj.step(capi.STAGE_ALL)
    assert pu.step_parity(o1, obj, r)["equal"]
    obj.step(capi.STAGE_DERIVE | capi.STAGE_REGIONS)
    obj.step(capi.STAGE_REMESH)
    r = obj.step(capi.STAGE_ALL)
    assert pu.step_parity(o1, obj, r)["equal"]
    obj.set_sdf_program(gen2)
    for _ in range(2):
        r = obj.step(capi.STAGE_ALL)
        assert pu.step_parity(o2, obj, r)["equal"]
    obj.set_sdf_program(gen1)
    for _ in range(2):
        r = obj.step(capi.STAGE_ALL)
        assert pu.step_parity(o1, obj, r)["equal"]
    obj.set_sample_ahead(False)  # (drops the pre-pass under way)
    r = obj.step(capi.STAGE_ALL)
    assert pu.step_parity(o1, obj, r)["equal"]
    obj.close()


def test_an_object_goes_while_its_pre_pass_is_under_way(ctx):
    """close() right behind a step: the grid waits for its pre-pass on the second stream before its buffers go; many times over, and a
    sampled object made afterwards is sound."""
    graph = craters(2)
    for _ in range(20):
        _, obj = resident_object(ctx, graph, True)
        obj.step(capi.STAGE_ALL)
        obj.close()
    o = oracle_of(graph)
    _, obj = resident_object(ctx, graph, True)
    for _ in range(3):
        r = obj.step(capi.STAGE_ALL)
        assert pu.step_parity(o, obj, r)["equal"]
    obj.close()


def test_sample_alone_then_the_rest(ctx):
    """A step of the sample stage alone (no derive sweep rolls the list counters over), then the other stages, with the pre-pass ahead."""
    graph = craters(3)
    o = oracle_of(graph)
    _, obj = resident_object(ctx, graph, True)
    for _ in range(3):
        obj.step(capi.STAGE_SAMPLE)
        r = obj.step(capi.STAGE_ALL & ~capi.STAGE_SAMPLE)
        assert pu.step_parity(o, obj, r)["equal"]
    for _ in range(2):
        r = obj.step(capi.STAGE_ALL)
        assert pu.step_parity(o, obj, r)["equal"]
    obj.close()


@pytest.mark.parametrize("seed", pu.fuzz_seeds([31, 32, 33, 34, 35, 36, 210, 341]))
def test_random_programs_with_the_pre_pass_a_step_ahead(ctx, seed):
    """tests/test_gpu_random_sdf.py's programs as resident programs stepped three times with the pre-pass ahead: the second and third step start
    at their evaluator, whose first launch commits the records the pre-pass parked — the chunk records and voxel bytes of every step against
    the oracle's"""
    from test_gpu_random_sdf import random_tree

    rng = np.random.default_rng(seed)
    g = SDFGraph()
    random_tree(g, rng, int(rng.integers(1, 5)))
    gen = SDFVoxelGenerator(1.0, g, 0)
    if min(gen.chunk_counts()) == 0:
        pytest.skip("degenerate root domain")
    o = oracle_of(g)
    obj = VoxelObject(ctx, gen.chunk_counts(), 1.0)
    obj.set_sdf_program(gen)
    obj.set_densities(np.ones(256, dtype=np.float32))
    obj.set_sample_ahead(True)
    for i in range(3):
        r = obj.step(capi.STAGE_ALL)
        p = pu.step_parity(o, obj, r)
        assert p["equal"], (i, p)
    obj.close()
