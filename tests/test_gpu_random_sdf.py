"""Random SDF programs through the whole generate → derive → remesh → inertia → regions path (fixed seeds): trees of spheres, boxes and
capsules under translations, rotations and scalings, joined by hard and smooth unions, subtractions and intersections. The sampler's
interval pre-pass, its constant-chunk shortcuts and the LDS classes of the evaluator decide per chunk what gets evaluated; every
decision must leave the voxel bytes bit-identical to the oracle's plain per-voxel evaluation."""
import numpy as np
import pytest

import parity_util as pu
from impact_amd.sdf_graph import SDFGraph, SDFNode
from test_gpu_parity import full_pipeline

pytestmark = pytest.mark.gpu


def random_tree(g, rng, depth):
    if depth == 0 or rng.random() < 0.25:
        kind = rng.integers(0, 3)
        if kind == 0:
            node = g.add_node(SDFNode.new_sphere(float(rng.uniform(4.0, 22.0))))
        elif kind == 1:
            node = g.add_node(SDFNode.new_box([float(x) for x in rng.uniform(4.0, 36.0, 3)]))
        else:
            node = g.add_node(SDFNode.new_capsule(float(rng.uniform(2.0, 30.0)), float(rng.uniform(3.0, 12.0))))
    else:
        a = random_tree(g, rng, depth - 1)
        b = random_tree(g, rng, depth - 1)
        smooth = 0.0 if rng.random() < 0.5 else float(rng.uniform(0.5, 6.0))
        op = rng.choice(3, p=[0.6, 0.25, 0.15])
        node = g.add_node((SDFNode.new_union, SDFNode.new_subtraction, SDFNode.new_intersection)[op](a, b, smooth))
    if rng.random() < 0.5:
        node = g.add_node(SDFNode.new_translation(node, [float(x) for x in rng.uniform(-18.0, 18.0, 3)]))
    if rng.random() < 0.35:
        axis = rng.normal(size=3)
        node = g.add_node(SDFNode.new_rotation_from_axis_angle(node, [float(x) for x in axis / np.linalg.norm(axis)], float(rng.uniform(0, 6.28))))
    if rng.random() < 0.25:
        node = g.add_node(SDFNode.new_scaling(node, float(rng.uniform(0.6, 1.5))))
    return node


@pytest.mark.parametrize("seed", pu.fuzz_seeds([31, 32, 33, 34, 35, 36]))
def test_random_sdf_program(ctx, seed):
    rng = np.random.default_rng(seed)
    g = SDFGraph()
    random_tree(g, rng, int(rng.integers(1, 5)))
    extent = [1.0, 0.5, 0.25, 2.0][seed % 4]
    o = pu.oracle_from_graph(g, extent)
    if min(o.chunk_counts) == 0:  # an empty program (e.g. an intersection of disjoint shapes): the host side refuses it
        with pytest.raises(ValueError):
            pu.gpu_from_graph(ctx, g, extent)
        return
    del o
    _, obj = full_pipeline(ctx, g, extent)
    obj.close()
