// ORACLE (test infrastructure only — never linked into the product library).
// Mesh types of the reference: mesh.rs:60-103, surface_nets.rs:39-50.
#pragma once
#include <vector>

#include "orc_voxel.hpp"

namespace orc {

struct VertexMaterials {  // SurfaceNetsVertexMaterials
    uint8_t indices[8];
    uint8_t weights[8];
};
struct IndexMaterials {  // VoxelMeshIndexMaterials (8 bytes)
    uint8_t indices[4];
    uint8_t weights[4];
};
struct Submesh {  // ChunkSubmesh (mesh.rs:94-103) + the chunk's vertex range (mesh.rs:145)
    uint32_t chunk[3];
    uint32_t index_offset;
    uint32_t index_count;
    uint32_t obscured[2][2][2];
    uint32_t vertex_offset;
    uint32_t vertex_count;
};
struct Mesh {
    std::vector<V3> positions, normals;
    std::vector<IndexMaterials> index_materials;
    std::vector<uint32_t> indices;
    std::vector<Submesh> submeshes;
};

void vertex_materials_compute(const bool has_voxel[8], const uint8_t mat[8], VertexMaterials& m);
void index_materials_for_triangle(const VertexMaterials* vm[3], IndexMaterials out[3]);
void mesh_recreate(const VoxelObject& obj, Mesh& mesh);
bool chunk_sdf_if_exposed(const VoxelObject& obj, int ci, int cj, int ck, float* values, uint8_t* types);

}  // namespace orc
