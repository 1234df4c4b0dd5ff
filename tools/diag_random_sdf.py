"""Diagnostic for a failing seed of tests/test_gpu_random_sdf.py: which chunks differ from the oracle and how (GPU box)."""
import sys

import numpy as np

sys.path.insert(0, "tests")
import oracle_lib as ol  # noqa: E402
import parity_util as pu  # noqa: E402
from impact_amd.sdf_graph import SDFGraph  # noqa: E402
from impact_amd.voxel import Context  # noqa: E402
from test_gpu_random_sdf import random_tree  # noqa: E402

ctx = Context(0)
for seed in [int(s) for s in sys.argv[1:]]:
    rng = np.random.default_rng(seed)
    g = SDFGraph()
    random_tree(g, rng, int(rng.integers(1, 5)))
    extent = [1.0, 0.5, 0.25, 2.0][seed % 4]
    print(f"== seed {seed} extent {extent} root {g.root_node_id}")
    for i, n in enumerate(g._nodes):
        print("  node", i, n if not hasattr(n, "__dict__") else n.__dict__)
    o = pu.oracle_from_graph(g, extent)
    G = pu.gpu_from_graph(ctx, g, extent)
    o_sdf, o_typ, _, _, o_info = o.export_dense()
    g_sdf, g_typ, _, _, g_info = G.download(flags=False, labels=False)
    cc = o.chunk_counts
    od3 = ol.tiled_to_dense(np.asarray(o_sdf).view(np.int8), cc)
    gd3 = ol.tiled_to_dense(np.asarray(g_sdf).view(np.int8), cc)
    bad = np.nonzero(o_info["gen_kind"] != g_info["gen_kind"])[0]
    print("  chunk-kind mismatches:", len(bad), "of", len(o_info))
    for c in bad[:12]:
        ci, cj, ck = c // (cc[1] * cc[2]), (c // cc[2]) % cc[1], c % cc[2]
        ob = od3[ci * 16:ci * 16 + 16, cj * 16:cj * 16 + 16, ck * 16:ck * 16 + 16]
        gb = gd3[ci * 16:ci * 16 + 16, cj * 16:cj * 16 + 16, ck * 16:ck * 16 + 16]
        print(f"   chunk {c} ({ci},{cj},{ck}) gen_kind oracle {o_info['gen_kind'][c]} gpu {g_info['gen_kind'][c]} | oracle sd [{ob.min()},{ob.max()}] gpu sd [{gb.min()},{gb.max()}] differing voxels {(ob != gb).sum()}")
    dv = np.argwhere(od3 != gd3)
    print("  voxel mismatches:", len(dv))
    for v in dv[:8]:
        print("   ", tuple(v), "oracle", od3[tuple(v)], "gpu", gd3[tuple(v)])
    G.close()
