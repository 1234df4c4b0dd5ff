// ORACLE (test infrastructure only — never linked into the product library).
//
// CPU restatement of the atomic SDF graph generator:
//   SDFGraph / SDFNode                 generation/sdf/atomic.rs:55-181
//   SDFGenerator::new_in               generation/sdf/atomic.rs:228-493
//   determine_transforms_and_margins   generation/sdf/atomic.rs:495-596
//   compute_signed_distances_for_block generation/sdf/atomic.rs:633-875
//   primitives                         generation/sdf/atomic.rs:1151-1292
//   smooth ops                         generation/sdf.rs:47-102
//   SDFVoxelGenerator                  generation.rs:204-371
// MultifractalNoise nodes (simdnoise 3.1.7, not vendored) are NOT supported: parity unpinned.
#pragma once
#include <vector>

#include "orc_voxel.hpp"

namespace orc {

enum SdfKind : uint32_t {
    SDF_SPHERE = 0,
    SDF_CAPSULE = 1,
    SDF_BOX = 2,
    SDF_TRANSLATION = 3,
    SDF_ROTATION = 4,
    SDF_SCALING = 5,
    SDF_NOISE = 6,  // unsupported
    SDF_UNION = 7,
    SDF_SUBTRACTION = 8,
    SDF_INTERSECTION = 9,
};

// Graph node as the user builds it (SDFNode, atomic.rs:62-81). 32 bytes, same layout as ivx_sdf_node.
struct SdfNode {
    uint32_t kind;
    uint32_t child1;
    uint32_t child2;
    uint32_t pad;
    float p[4];  // sphere: r | capsule: segment_length, r | box: extents xyz | translation xyz |
                 // rotation quat xyzw | scaling s | binary: smoothness
};

// ProcessedSDFNode (atomic.rs:83-102) with the primitive parameters precomputed as the
// constructors do (atomic.rs:1154-1262, sdf.rs:14-22).
struct ProcessedNode {
    uint32_t kind;
    uint32_t leaf_count;
    M4 transform;           // root space -> node space
    AABB domain_with_margin;
    float margin;
    float a, b, c;          // sphere: a=r | capsule: a=half_segment, b=r | box: a,b,c=half extents |
                            // scaling: a=s | binary: a=smoothness, b=0.25/smoothness
};

struct SdfGenerator {
    std::vector<ProcessedNode> nodes;
    int stack_size = 0;
    AABB domain{{0, 0, 0}, {0, 0, 0}};
    bool build(const SdfNode* nodes, int n, uint32_t root);  // false on error (cycle / missing)
    // evaluates a 16^3 block whose lower corner voxel centre is `aabb.lo`; out[4096] in (i,j,k) order
    void compute_block(const AABB& block_aabb, std::vector<float>& stack, float* out) const;
};

struct SdfVoxelGenerator : Generator {
    float extent = 1.0f;
    int shape[3] = {0, 0, 0};
    V3 shifted_center{-0.5f, -0.5f, -0.5f};
    SdfGenerator sdf;
    uint8_t voxel_type = 0;  // SameVoxelTypeGenerator (generation/voxel_type.rs:85-96)
    void init(float voxel_extent, uint8_t type);  // after sdf.build(): generation.rs:207-258
    float voxel_extent() const override { return extent; }
    void grid_shape(int out[3]) const override {
        out[0] = shape[0];
        out[1] = shape[1];
        out[2] = shape[2];
    }
    ChunkSparseness generate_chunk(Voxel* voxels, const int origin[3]) const override;
};

float smooth_union(float d1, float d2, float s, float q);

}  // namespace orc
