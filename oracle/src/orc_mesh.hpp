// ORACLE (test infrastructure only — never linked into the product library).
// Mesh types of the reference: mesh.rs:60-103, surface_nets.rs:39-50.
#pragma once
#include <map>
#include <unordered_map>
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
// RangeAllocator (impact_containers/src/range_allocator.rs:1-108): free ranges ordered (and identified) by their start
struct RangeAllocator {
    std::map<size_t, size_t> free_ranges;  // start -> end
    void free_range(size_t start, size_t end) {
        if (start < end) free_ranges.emplace(start, end);  // BTreeSet::insert keeps an existing entry with the same start
    }
    void mark_all_ranges_occupied() { free_ranges.clear(); }
    bool allocate_range(size_t required_len, size_t& start);
    void merge_consecutive_ranges();
};
struct Mesh {
    std::vector<V3> positions, normals;
    std::vector<IndexMaterials> index_materials;
    std::vector<uint32_t> indices;
    std::vector<Submesh> submeshes;
    // ChunkSubmeshManager (mesh.rs:699-849): chunk -> submesh slot, free ranges of the vertex and index buffers
    std::unordered_map<uint64_t, size_t> chunk_index;
    RangeAllocator vertex_ranges, index_ranges;
    // VoxelMeshModifications (mesh.rs:113-123): what a renderer has to re-upload; cleared by report_gpu_resources_synchronized
    std::vector<uint32_t> updated_data_ranges;  // 4 per written chunk: vertex start, end, index start, end
    bool chunks_were_removed = false;
};

void vertex_materials_compute(const bool has_voxel[8], const uint8_t mat[8], VertexMaterials& m);
void index_materials_for_triangle(const VertexMaterials* vm[3], IndexMaterials out[3]);
void mesh_recreate(const VoxelObject& obj, Mesh& mesh);
// VoxelObjectMesh::sync_with_voxel_object (mesh.rs:355-456); invalidated = one byte per chunk, visited in chunk-linear order
void mesh_sync(const VoxelObject& obj, Mesh& mesh, const uint8_t* invalidated);
bool chunk_sdf_if_exposed(const VoxelObject& obj, int ci, int cj, int ck, float* values, uint8_t* types);

}  // namespace orc
