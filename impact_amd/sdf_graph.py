"""Host-side mirror of the reference's atomic SDF graph builder.

Mirrors `SDFGraph` / `SDFNode` of engine/crates/impact_voxel/src/generation/sdf/atomic.rs:55-181,
1019-1148 (same constructor names and argument meaning). The graph is a flat array of 32-byte
`ivx_sdf_node` records (include/impact_voxel_hip.h) that is handed to the C ABI unchanged.
"""
from __future__ import annotations

import math

import numpy as np

SDF_SPHERE, SDF_CAPSULE, SDF_BOX = 0, 1, 2
SDF_TRANSLATION, SDF_ROTATION, SDF_SCALING, SDF_NOISE = 3, 4, 5, 6
SDF_UNION, SDF_SUBTRACTION, SDF_INTERSECTION = 7, 8, 9

NODE_DTYPE = np.dtype(
    [("kind", "<u4"), ("child1", "<u4"), ("child2", "<u4"), ("pad", "<u4"), ("p", "<f4", (4,))]
)
assert NODE_DTYPE.itemsize == 32

PROCESSED_NODE_DTYPE = np.dtype(
    [
        ("kind", "<u4"),
        ("leaf_count", "<u4"),
        ("transform", "<f4", (16,)),
        ("domain_lo", "<f4", (3,)),
        ("domain_hi", "<f4", (3,)),
        ("margin", "<f4"),
        ("a", "<f4"),
        ("b", "<f4"),
        ("c", "<f4"),
        ("reserved", "<u4", (4,)),
    ]
)
assert PROCESSED_NODE_DTYPE.itemsize == 128


class SDFNode:
    """Constructors named as in atomic.rs:1060-1128."""

    @staticmethod
    def _mk(kind, c1=0, c2=0, p=(0.0, 0.0, 0.0, 0.0)):
        rec = np.zeros((), dtype=NODE_DTYPE)
        rec["kind"], rec["child1"], rec["child2"] = kind, c1, c2
        rec["p"] = np.asarray(list(p) + [0.0] * (4 - len(p)), dtype=np.float32)
        return rec

    @staticmethod
    def new_sphere(radius):
        assert radius >= 0.0
        return SDFNode._mk(SDF_SPHERE, p=(radius,))

    @staticmethod
    def new_capsule(segment_length, radius):
        assert segment_length >= 0.0 and radius >= 0.0
        return SDFNode._mk(SDF_CAPSULE, p=(segment_length, radius))

    @staticmethod
    def new_box(extents):
        assert all(e >= 0.0 for e in extents)
        return SDFNode._mk(SDF_BOX, p=tuple(extents))

    @staticmethod
    def new_translation(child_id, translation):
        return SDFNode._mk(SDF_TRANSLATION, child_id, p=tuple(translation))

    @staticmethod
    def new_rotation(child_id, quaternion_xyzw):
        return SDFNode._mk(SDF_ROTATION, child_id, p=tuple(quaternion_xyzw))

    @staticmethod
    def new_rotation_from_axis_angle(child_id, axis, angle):
        ax = np.asarray(axis, dtype=np.float64)
        ax = ax / np.linalg.norm(ax)
        s, c = math.sin(0.5 * angle), math.cos(0.5 * angle)
        return SDFNode.new_rotation(child_id, (ax[0] * s, ax[1] * s, ax[2] * s, c))

    @staticmethod
    def new_scaling(child_id, scaling):
        assert scaling > 0.0
        return SDFNode._mk(SDF_SCALING, child_id, p=(scaling,))

    @staticmethod
    def new_union(child_1_id, child_2_id, smoothness):
        assert smoothness >= 0.0
        return SDFNode._mk(SDF_UNION, child_1_id, child_2_id, (smoothness,))

    @staticmethod
    def new_subtraction(child_1_id, child_2_id, smoothness):
        assert smoothness >= 0.0
        return SDFNode._mk(SDF_SUBTRACTION, child_1_id, child_2_id, (smoothness,))

    @staticmethod
    def new_intersection(child_1_id, child_2_id, smoothness):
        assert smoothness >= 0.0
        return SDFNode._mk(SDF_INTERSECTION, child_1_id, child_2_id, (smoothness,))


class SDFGraph:
    """`SDFGraph` (atomic.rs:1019-1058): `add_node` returns the id and makes the node the root."""

    def __init__(self):
        self._nodes = []
        self.root_node_id = 0

    def add_node(self, node) -> int:
        node_id = len(self._nodes)
        self._nodes.append(node)
        self.root_node_id = node_id
        return node_id

    def set_root_node(self, node_id: int):
        assert node_id < len(self._nodes)
        self.root_node_id = node_id

    def nodes(self) -> np.ndarray:
        if not self._nodes:
            return np.zeros((0,), dtype=NODE_DTYPE)
        return np.array(self._nodes, dtype=NODE_DTYPE)

    def __len__(self):
        return len(self._nodes)
