import sys, time
sys.path.insert(0, '.')
import numpy as np
from impact_amd import scenes, capi
from impact_amd.sdf_graph import SDFGraph, SDFNode
from impact_amd.voxel import Context, SDFVoxelGenerator, VoxelObject
ctx = Context(0)
def run(name, graph):
    gen = SDFVoxelGenerator(1.0, graph, 0)
    obj = VoxelObject(ctx, gen.chunk_counts(), 1.0)
    obj.set_sdf_program(gen); obj.set_densities(np.ones(256, dtype=np.float32))
    for _ in range(3): obj.step(capi.STAGE_SAMPLE)
    ts = [obj.step(capi.STAGE_SAMPLE)["stage_ms"][0] for _ in range(10)]
    print(f"{name:28s} nodes={len(gen.sdf_generator.nodes):3d} stack={gen.sdf_generator.required_forward_stack_size} chunks={obj.n_chunks} sample_ms={np.mean(ts):.4f}")
    obj.close()
run('box 500', scenes.box_scene((500.0, 500.0, 500.0)))
run('sphere r250', scenes.sphere_scene(250.0))
g = SDFGraph(); a = g.add_node(SDFNode.new_sphere(250.0)); b = g.add_node(SDFNode.new_sphere(100.0)); g.add_node(SDFNode.new_union(a, b, 0.0))
run('union 2 spheres', g)
g = SDFGraph(); acc = g.add_node(SDFNode.new_sphere(250.0))
for i in range(14):
    b = g.add_node(SDFNode.new_sphere(10.0)); acc = g.add_node(SDFNode.new_union(acc, b, 0.0))
run('union chain 15 (small)', g)
run('asteroid 2.05', scenes.asteroid_scene(2.05))
ctx.close()
