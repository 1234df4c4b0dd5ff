// ORACLE (test infrastructure only — never linked into the product library).
// C interface (oracle/include/oracle.h) over the restatement in this directory.
#include <cstring>
#include <memory>

#include "../include/oracle.h"
#include "orc_mesh.hpp"
#include "orc_sdf.hpp"

namespace orc {
void inertia_moments_f32(const VoxelObject& obj, const float* dens, float out32[10]);
void inertia_moments_f64(const VoxelObject& obj, const float* dens, double out[10]);
void inertia_moments_f32_parallel(const VoxelObject& obj, const float* dens, float out32[10], int threads);
void generate_without_derived_state_parallel(VoxelObject& obj, const Generator& gen, int threads);
void compute_all_derived_state_parallel(VoxelObject& obj, int threads);
void mesh_recreate_parallel(const VoxelObject& obj, Mesh& mesh, int threads);
void derive_inertial_properties(const float m[10], float out[22]);
uint32_t canonical_region_labels(const VoxelObject& obj, uint32_t* labels);
int split_off_smallest_region(VoxelObject& parent, VoxelObject& child, int origin[3]);
int clip_polyhedron(VoxelObject& parent, const float* planes, int n_planes, const float aabb[6], int mode, VoxelObject& child, int origin[3]);
int sphere_voxel_object_contacts(const VoxelObject& obj, const float rot[4], const float trans[3], const float center[3], float radius, int cap,
                                 int32_t* indices, float* position, float* normal, float* depth);
int capsule_voxel_object_contacts(const VoxelObject& obj, const float rot[4], const float trans[3], const float seg_start[3], const float seg_vec[3],
                                  float radius, int cap, int32_t* indices, float* position, float* normal, float* depth);
int collision_probes(const VoxelObject& obj, const float* pos, const float* nrm, const uint32_t* idx, const uint32_t* submeshes, uint32_t n_sub,
                     float* points, uint32_t cap, uint32_t* entries, uint32_t* n_entries);
int mutual_contacts(const VoxelObject& A, const float* probes_a, const uint32_t* entries_a, uint32_t n_entries_a, const float com_a[3], const float rot_a[4],
                    const float trans_a[3], const VoxelObject& B, const float* probes_b, const uint32_t* entries_b, uint32_t n_entries_b,
                    const float com_b[3], const float rot_b[4], const float trans_b[3], int cap, int32_t* which_ijk, float* position, float* normal,
                    float* depth);
struct ProbeSet;
ProbeSet* probes_new();
void probes_delete(ProbeSet*);
void probes_recompute(const VoxelObject& obj, const Mesh& mesh, ProbeSet& ps);
void probes_sync(const VoxelObject& obj, const Mesh& mesh, ProbeSet& ps, const uint8_t* invalidated);
uint32_t probes_entries(const ProbeSet& ps, uint32_t* out);
uint32_t probes_points(const ProbeSet& ps, float* out, uint32_t cap);
int plane_voxel_object_contacts(const VoxelObject& obj, const float rot[4], const float trans[3], const float plane_normal[3], float plane_displacement,
                                int cap, int32_t* indices, float* position, float* normal, float* depth);
void absorb_mutual(VoxelObject& A, const float rot_a[4], const float trans_a[3], const float* dens_a, VoxelObject& B, const float rot_b[4],
                   const float trans_b[3], const float* dens_b, float smoothness, double removed_a[10], double removed_b[10], uint8_t* invalidated_a,
                   uint8_t* invalidated_b, uint64_t stats[6]);
int absorb_capsule(VoxelObject& obj, const float segment_start[3], const float segment_vector[3], float influence_radius, float capsule_radius,
                   const float* dens, double removed64[10], uint32_t emptied_by_type[256], uint8_t* invalidated, uint32_t* touched_chunks);
int absorb_sphere(VoxelObject& obj, const float center[3], float influence_radius, float sphere_radius, const float* dens, double removed64[10],
                  uint32_t emptied_by_type[256], uint8_t* invalidated, uint32_t* touched_chunks);

// OffsetBoxVoxelGenerator (object.rs:3387-3504)
struct BoxGenerator : Generator {
    int shape[3], offset[3];
    Voxel voxel;
    float voxel_extent() const override { return 0.25f; }
    void grid_shape(int out[3]) const override {
        for (int d = 0; d < 3; ++d) out[d] = offset[d] + shape[d];
    }
    ChunkSparseness generate_chunk(Voxel* v, const int o[3]) const override {
        bool only_empty = true, is_void = true;
        int idx = 0;
        for (int i = o[0]; i < o[0] + CHUNK; ++i)
            for (int j = o[1]; j < o[1] + CHUNK; ++j)
                for (int k = o[2]; k < o[2] + CHUNK; ++k) {
                    bool in = i >= offset[0] && i < offset[0] + shape[0] && j >= offset[1] && j < offset[1] + shape[1] &&
                              k >= offset[2] && k < offset[2] + shape[2];
                    if (in) {
                        if (!voxel.empty()) only_empty = false;
                        if (voxel.sd != 127) is_void = false;
                        v[idx] = voxel;
                    } else {
                        v[idx] = voxel_max_outside();
                    }
                    idx++;
                }
        return {only_empty, is_void};
    }
};

// ManualVoxelGenerator<N> (object.rs:3393-3561)
struct ManualGenerator : Generator {
    int n;
    std::vector<uint8_t> cells;
    int offset[3];
    float voxel_extent() const override { return 0.25f; }
    void grid_shape(int out[3]) const override {
        for (int d = 0; d < 3; ++d) out[d] = offset[d] + n;
    }
    ChunkSparseness generate_chunk(Voxel* v, const int o[3]) const override {
        bool is_void = true;
        int idx = 0;
        for (int i = o[0]; i < o[0] + CHUNK; ++i)
            for (int j = o[1]; j < o[1] + CHUNK; ++j)
                for (int k = o[2]; k < o[2] + CHUNK; ++k) {
                    bool in = i >= offset[0] && i < offset[0] + n && j >= offset[1] && j < offset[1] + n && k >= offset[2] &&
                              k < offset[2] + n && cells[((size_t)(i - offset[0]) * n + (j - offset[1])) * n + (k - offset[2])] != 0;
                    if (in) {
                        is_void = false;
                        v[idx] = voxel_max_inside(0);
                    } else {
                        v[idx] = voxel_max_outside();
                    }
                    idx++;
                }
        return {is_void, is_void};
    }
};

// Dense chunk-tiled planes presented as a generator (classification as generation.rs:336-355).
struct DenseGenerator : Generator {
    int cc[3];
    float extent;
    const int8_t* sdf;
    const uint8_t* type;
    float voxel_extent() const override { return extent; }
    void grid_shape(int out[3]) const override {
        for (int d = 0; d < 3; ++d) out[d] = cc[d] * CHUNK;
    }
    ChunkSparseness generate_chunk(Voxel* v, const int o[3]) const override {
        size_t c = ((size_t)(o[0] / CHUNK) * cc[1] + (o[1] / CHUNK)) * cc[2] + (o[2] / CHUNK);
        bool only_empty = true, is_void = true;
        for (int idx = 0; idx < CHUNK_VOXELS; ++idx) {
            int8_t sd = sdf[c * CHUNK_VOXELS + idx];
            uint8_t t = type[c * CHUNK_VOXELS + idx];
            if (sd < 0) {
                only_empty = false;
                is_void = false;
                v[idx] = Voxel{t, sd, 0};
            } else {
                if (!sd_is_void(sd)) is_void = false;
                v[idx] = Voxel{t, sd, F_EMPTY};
            }
        }
        return {only_empty, is_void};
    }
};
}  // namespace orc

using namespace orc;

struct orc_object {
    VoxelObject obj;
    int shape[3];
};
struct orc_mesh {
    Mesh mesh;
};

static orc_object* make_object(const Generator& g) {
    orc_object* o = new orc_object();
    g.grid_shape(o->shape);
    generate_without_derived_state(o->obj, g);
    return o;
}

extern "C" {

int orc_sdf_compile(const orc_sdf_node* nodes, int n, uint32_t root, orc_sdf_processed_node* out, int cap, float domain[6],
                    int* stack_size) {
    static_assert(sizeof(orc_sdf_node) == sizeof(SdfNode), "layout");
    SdfGenerator g;
    if (!g.build(reinterpret_cast<const SdfNode*>(nodes), n, root)) return -1;
    if ((int)g.nodes.size() > cap) return -(int)g.nodes.size();
    for (size_t i = 0; i < g.nodes.size(); ++i) {
        const ProcessedNode& p = g.nodes[i];
        orc_sdf_processed_node& q = out[i];
        std::memset(&q, 0, sizeof(q));
        q.kind = p.kind;
        q.leaf_count = p.leaf_count;
        for (int c = 0; c < 4; ++c)
            for (int r = 0; r < 4; ++r) q.transform[c * 4 + r] = p.transform.c[c][r];
        q.domain_lo[0] = p.domain_with_margin.lo.x;
        q.domain_lo[1] = p.domain_with_margin.lo.y;
        q.domain_lo[2] = p.domain_with_margin.lo.z;
        q.domain_hi[0] = p.domain_with_margin.hi.x;
        q.domain_hi[1] = p.domain_with_margin.hi.y;
        q.domain_hi[2] = p.domain_with_margin.hi.z;
        q.margin = p.margin;
        q.a = p.a;
        q.b = p.b;
        q.c = p.c;
    }
    domain[0] = g.domain.lo.x;
    domain[1] = g.domain.lo.y;
    domain[2] = g.domain.lo.z;
    domain[3] = g.domain.hi.x;
    domain[4] = g.domain.hi.y;
    domain[5] = g.domain.hi.z;
    *stack_size = g.stack_size;
    return (int)g.nodes.size();
}

orc_object* orc_object_from_sdf(const orc_sdf_node* nodes, int n, uint32_t root, float voxel_extent, uint8_t voxel_type) {
    SdfVoxelGenerator g;
    if (!g.sdf.build(reinterpret_cast<const SdfNode*>(nodes), n, root)) return nullptr;
    g.init(voxel_extent, voxel_type);
    return make_object(g);
}

// all-cores variants (bench.py's cpu_baseline_all_cores; same results as the sequential entry points)
orc_object* orc_object_from_sdf_parallel(const orc_sdf_node* nodes, int n, uint32_t root, float voxel_extent, uint8_t voxel_type, int threads) {
    SdfVoxelGenerator g;
    if (!g.sdf.build(reinterpret_cast<const SdfNode*>(nodes), n, root)) return nullptr;
    g.init(voxel_extent, voxel_type);
    orc_object* o = new orc_object();
    g.grid_shape(o->shape);
    generate_without_derived_state_parallel(o->obj, g, threads);
    update_occupied_voxel_ranges(o->obj);
    compute_all_derived_state_parallel(o->obj, threads);
    return o;
}
orc_mesh* orc_mesh_recreate_parallel(const orc_object* o, int threads) {
    orc_mesh* m = new orc_mesh();
    mesh_recreate_parallel(o->obj, m->mesh, threads);
    return m;
}
void orc_inertia_parallel(const orc_object* o, const float densities[256], float out32[10], int threads) {
    inertia_moments_f32_parallel(o->obj, densities, out32, threads);
}

orc_object* orc_object_from_box(const int shape[3], const int offset[3], uint8_t type, int8_t sd, uint8_t flags) {
    BoxGenerator g;
    for (int d = 0; d < 3; ++d) {
        g.shape[d] = shape[d];
        g.offset[d] = offset[d];
    }
    g.voxel = Voxel{type, sd, flags};
    return make_object(g);
}

orc_object* orc_object_from_manual(int n, const uint8_t* cells, const int offset[3]) {
    ManualGenerator g;
    g.n = n;
    g.cells.assign(cells, cells + (size_t)n * n * n);
    for (int d = 0; d < 3; ++d) g.offset[d] = offset[d];
    return make_object(g);
}

orc_object* orc_object_from_dense(const int chunk_counts[3], float voxel_extent, const int8_t* sdf, const uint8_t* type) {
    DenseGenerator g;
    for (int d = 0; d < 3; ++d) g.cc[d] = chunk_counts[d];
    g.extent = voxel_extent;
    g.sdf = sdf;
    g.type = type;
    return make_object(g);
}

void orc_object_free(orc_object* o) { delete o; }
void orc_update_occupied_voxel_ranges(orc_object* o) { update_occupied_voxel_ranges(o->obj); }
void orc_compute_all_derived_state(orc_object* o) { compute_all_derived_state(o->obj); }
float orc_object_extent(const orc_object* o) { return o->obj.extent; }

void orc_object_info(const orc_object* o, int32_t out[19]) {
    const VoxelObject& v = o->obj;
    out[0] = v.cc[0];
    out[1] = v.cc[1];
    out[2] = v.cc[2];
    out[3] = (int32_t)(v.voxels.size() >> 12);
    for (int d = 0; d < 3; ++d) {
        out[4 + 2 * d] = v.occ_chunk[d][0];
        out[5 + 2 * d] = v.occ_chunk[d][1];
        out[10 + 2 * d] = v.occ_voxel[d][0];
        out[11 + 2 * d] = v.occ_voxel[d][1];
        out[16 + d] = o->shape[d];
    }
}

void orc_export_dense(const orc_object* o, int8_t* sdf, uint8_t* type, uint8_t* flags, uint8_t* local_labels, orc_chunk_info* info) {
    const VoxelObject& v = o->obj;
    int n = v.n_chunks();
    for (int c = 0; c < n; ++c) {
        const Chunk& ch = v.chunks[c];
        orc_chunk_info ci{};
        ci.kind = ch.kind;
        ci.gen_kind = ch.gen_kind;
        if (ch.kind == K_NONUNIFORM) {
            ci.flags = ch.flags;
            for (int d = 0; d < 3; ++d)
                for (int s = 0; s < 2; ++s) ci.face_dist |= (uint16_t)(ch.face[d][s] << (2 * (2 * d + s)));
            ci.region_count = (uint8_t)ch.region_count;
            ci.boundary_region_count = (uint8_t)ch.boundary_region_count;
        } else if (ch.kind == K_UNIFORM) {
            ci.uniform_type = ch.uniform_voxel.type;
            ci.face_dist = 0x555;
            ci.region_count = 1;
            ci.boundary_region_count = 1;
        }
        if (info) info[c] = ci;
        for (int idx = 0; idx < CHUNK_VOXELS; ++idx) {
            Voxel x;
            uint8_t lab;
            if (ch.kind == K_VOID) {
                x = voxel_max_outside();
                lab = 255;
            } else if (ch.kind == K_UNIFORM) {
                x = ch.uniform_voxel;
                lab = 0;
            } else {
                x = v.voxels[((size_t)ch.data_offset << 12) + idx];
                lab = v.labels[((size_t)ch.data_offset << 12) + idx];
            }
            size_t g = (size_t)c * CHUNK_VOXELS + idx;
            if (sdf) sdf[g] = x.sd;
            if (type) type[g] = x.type;
            if (flags) flags[g] = x.flags;
            if (local_labels) local_labels[g] = lab;
        }
    }
}

void orc_export_sparse(const orc_object* o, int32_t* data_offsets, uint8_t* voxels_aos3, uint8_t* labels) {
    const VoxelObject& v = o->obj;
    int n = v.n_chunks();
    for (int c = 0; c < n; ++c) data_offsets[c] = v.chunks[c].kind == K_NONUNIFORM ? (int32_t)v.chunks[c].data_offset : -1;
    if (voxels_aos3) std::memcpy(voxels_aos3, v.voxels.data(), v.voxels.size() * 3);
    if (labels) std::memcpy(labels, v.labels.data(), v.labels.size());
}

orc_mesh* orc_mesh_recreate(const orc_object* o) {
    orc_mesh* m = new orc_mesh();
    mesh_recreate(o->obj, m->mesh);
    return m;
}
// VoxelObjectMesh::sync_with_voxel_object (mesh.rs:355-456)
void orc_mesh_sync(orc_mesh* m, const orc_object* o, const uint8_t* invalidated_chunks) { mesh_sync(o->obj, m->mesh, invalidated_chunks); }
// RangeAllocator (impact_containers/src/range_allocator.rs) driven by a script: ops = triples (0 free a b | 1 allocate len _ | 2 merge _ _);
// results: two values per op (allocate: start, end or -1, -1; others 0, 0)
void orc_range_allocator_script(const int64_t* ops, int n_ops, int64_t* results) {
    RangeAllocator a;
    for (int i = 0; i < n_ops; ++i) {
        const int64_t* op = ops + 3 * i;
        results[2 * i] = results[2 * i + 1] = 0;
        if (op[0] == 0) a.free_range((size_t)op[1], (size_t)op[2]);
        else if (op[0] == 1) {
            size_t s0;
            if (a.allocate_range((size_t)op[1], s0)) results[2 * i] = (int64_t)s0, results[2 * i + 1] = (int64_t)(s0 + (size_t)op[1]);
            else results[2 * i] = results[2 * i + 1] = -1;
        } else a.merge_consecutive_ranges();
    }
}
// VoxelObjectMesh::mesh_modifications / report_gpu_resources_synchronized (mesh.rs:826-841): returns the number of ranges (4 u32 each)
int orc_mesh_modifications(const orc_mesh* m, uint32_t* ranges, int cap, int* chunks_were_removed) {
    const int n = (int)(m->mesh.updated_data_ranges.size() / 4);
    for (int i = 0; i < n && i < cap; ++i) std::memcpy(ranges + 4 * i, &m->mesh.updated_data_ranges[4 * (size_t)i], 16);
    *chunks_were_removed = m->mesh.chunks_were_removed ? 1 : 0;
    return n;
}
void orc_mesh_report_synchronized(orc_mesh* m) {
    m->mesh.updated_data_ranges.clear();
    m->mesh.chunks_were_removed = false;
}
void orc_mesh_counts(const orc_mesh* m, uint32_t out[3]) {
    out[0] = (uint32_t)m->mesh.positions.size();
    out[1] = (uint32_t)m->mesh.indices.size();
    out[2] = (uint32_t)m->mesh.submeshes.size();
}
void orc_mesh_get(const orc_mesh* m, float* positions, float* normals, uint32_t* indices, uint8_t* index_materials, uint32_t* submeshes) {
    const Mesh& mesh = m->mesh;
    if (positions) std::memcpy(positions, mesh.positions.data(), mesh.positions.size() * 12);
    if (normals) std::memcpy(normals, mesh.normals.data(), mesh.normals.size() * 12);
    if (indices) std::memcpy(indices, mesh.indices.data(), mesh.indices.size() * 4);
    if (index_materials) std::memcpy(index_materials, mesh.index_materials.data(), mesh.index_materials.size() * 8);
    if (submeshes)
        for (size_t s = 0; s < mesh.submeshes.size(); ++s) {
            const Submesh& sm = mesh.submeshes[s];
            uint32_t* q = submeshes + 16 * s;
            q[0] = sm.chunk[0];
            q[1] = sm.chunk[1];
            q[2] = sm.chunk[2];
            q[3] = sm.index_offset;
            q[4] = sm.index_count;
            std::memcpy(q + 5, sm.obscured, 32);
            q[13] = sm.vertex_offset;
            q[14] = sm.vertex_count;
            q[15] = 0;
        }
}
void orc_mesh_free(orc_mesh* m) { delete m; }
int orc_chunk_sdf(const orc_object* o, int ci, int cj, int ck, float* values5832, uint8_t* types5832) {
    return chunk_sdf_if_exposed(o->obj, ci, cj, ck, values5832, types5832) ? 1 : 0;
}

void orc_vertex_materials(const uint8_t has_voxel[8], const uint8_t materials[8], uint8_t out_indices[8], uint8_t out_weights[8]) {
    bool has[8];
    for (int i = 0; i < 8; ++i) has[i] = has_voxel[i] != 0;
    VertexMaterials vm;
    vertex_materials_compute(has, materials, vm);
    std::memcpy(out_indices, vm.indices, 8);
    std::memcpy(out_weights, vm.weights, 8);
}
void orc_index_materials(const uint8_t vm_indices[24], const uint8_t vm_weights[24], uint8_t out[24]) {
    VertexMaterials vm[3];
    for (int v = 0; v < 3; ++v) {
        std::memcpy(vm[v].indices, vm_indices + 8 * v, 8);
        std::memcpy(vm[v].weights, vm_weights + 8 * v, 8);
    }
    const VertexMaterials* p[3] = {&vm[0], &vm[1], &vm[2]};
    IndexMaterials im[3];
    index_materials_for_triangle(p, im);
    std::memcpy(out, im, 24);
}

void orc_inertia(const orc_object* o, const float densities[256], float out32[10], double out64[10]) {
    if (out32) inertia_moments_f32(o->obj, densities, out32);
    if (out64) inertia_moments_f64(o->obj, densities, out64);
}
void orc_derive_inertial_properties(const float moments[10], float out[22]) { derive_inertial_properties(moments, out); }

uint32_t orc_region_labels(const orc_object* o, uint32_t* labels) { return canonical_region_labels(o->obj, labels); }

// extract_any_disconnected_region (object/extraction.rs:78-119): returns the outcome (0 none, 1 extracted, 2 removed
// but discarded); *child is a new object (free with orc_object_free) when the outcome is 1
int orc_split_off_smallest_region(orc_object* parent, orc_object** child, int origin_offset_in_parent[3]) {
    orc_object* c = new orc_object();
    int rc = split_off_smallest_region(parent->obj, c->obj, origin_offset_in_parent);
    if (rc == 1) {
        for (int d = 0; d < 3; ++d) c->shape[d] = c->obj.cc[d] * CHUNK;
        *child = c;
    } else {
        delete c;
        *child = nullptr;
    }
    return rc;
}

// extract_polyhedron (mode 0) / copy_polyhedron (mode 1) (object/extraction.rs:604-1768); same outcome convention
int orc_clip_polyhedron(orc_object* parent, const float* planes4, int n_planes, const float aabb[6], int mode, orc_object** child,
                        int origin_offset_in_parent[3]) {
    orc_object* c = new orc_object();
    int rc = clip_polyhedron(parent->obj, planes4, n_planes, aabb, mode, c->obj, origin_offset_in_parent);
    if (rc == 1) {
        for (int d = 0; d < 3; ++d) c->shape[d] = c->obj.cc[d] * CHUNK;
        *child = c;
    } else {
        delete c;
        *child = nullptr;
    }
    return rc;
}

// apply_sphere_absorption (interaction/absorption.rs:801-844) with the sphere already in the object's normalized space:
// returns the number of chunks that became void
int orc_absorb_sphere(orc_object* o, const float center[3], float influence_radius, float sphere_radius, const float densities[256],
                      double removed64[10], uint32_t emptied_by_type[256], uint8_t* invalidated_chunks, uint32_t* touched_chunks) {
    return absorb_sphere(o->obj, center, influence_radius, sphere_radius, densities, removed64, emptied_by_type, invalidated_chunks, touched_chunks);
}

// apply_mutual_absorption (interaction/absorption.rs:891-1079)
void orc_absorb_mutual(orc_object* a, const float rotation_a[4], const float translation_a[3], const float densities_a[256], orc_object* b,
                       const float rotation_b[4], const float translation_b[3], const float densities_b[256], float smoothness, double removed_a[10],
                       double removed_b[10], uint8_t* invalidated_a, uint8_t* invalidated_b, uint64_t stats[6]) {
    absorb_mutual(a->obj, rotation_a, translation_a, densities_a, b->obj, rotation_b, translation_b, densities_b, smoothness, removed_a, removed_b,
                  invalidated_a, invalidated_b, stats);
}

// apply_capsule_absorption (interaction/absorption.rs:846-889)
int orc_absorb_capsule(orc_object* o, const float segment_start[3], const float segment_vector[3], float influence_radius, float capsule_radius,
                       const float densities[256], double removed64[10], uint32_t emptied_by_type[256], uint8_t* invalidated_chunks, uint32_t* touched_chunks) {
    return absorb_capsule(o->obj, segment_start, segment_vector, influence_radius, capsule_radius, densities, removed64, emptied_by_type, invalidated_chunks,
                          touched_chunks);
}

// for_each_sphere_voxel_object_contact (impact_voxel/src/collidable.rs:1098-1127): contacts in the reference's traversal order
int orc_sphere_voxel_object_contacts(const orc_object* o, const float rotation_xyzw[4], const float translation[3], const float center[3], float radius,
                                     int cap, int32_t* indices, float* position, float* normal, float* depth) {
    return sphere_voxel_object_contacts(o->obj, rotation_xyzw, translation, center, radius, cap, indices, position, normal, depth);
}

// probes as a living set: recompute_for_all_chunks once, then sync_with_voxel_object_and_mesh after every mesh sync (collidable.rs:361-433, 524-612)
orc_probes* orc_probes_recompute(const orc_object* o, const orc_mesh* m) {
    ProbeSet* p = probes_new();
    probes_recompute(o->obj, m->mesh, *p);
    return reinterpret_cast<orc_probes*>(p);
}
void orc_probes_sync(orc_probes* p, const orc_object* o, const orc_mesh* m, const uint8_t* invalidated_chunks) {
    probes_sync(o->obj, m->mesh, *reinterpret_cast<ProbeSet*>(p), invalidated_chunks);
}
/* points: the whole buffer, freed ranges included; entries (5 u32: chunk i, j, k, first, end) sorted by range start; returns the number of points */
uint32_t orc_probes_get(const orc_probes* p, float* points, uint32_t cap_points, uint32_t* entries, uint32_t* n_entries) {
    const ProbeSet& ps = *reinterpret_cast<const ProbeSet*>(p);
    if (entries) *n_entries = probes_entries(ps, entries);
    return probes_points(ps, points, cap_points);
}
void orc_probes_free(orc_probes* p) { probes_delete(reinterpret_cast<ProbeSet*>(p)); }

// VoxelObjectCollisionProbes::recompute_for_all_chunks (impact_voxel/src/collidable.rs:361-392, 473-523, 614-731)
int orc_collision_probes(const orc_object* o, const float* positions, const float* normals, const uint32_t* indices, const uint32_t* submeshes,
                         uint32_t n_submeshes, float* points, uint32_t cap, uint32_t* chunk_entries, uint32_t* n_entries) {
    return collision_probes(o->obj, positions, normals, indices, submeshes, n_submeshes, points, cap, chunk_entries, n_entries);
}

// for_each_mutual_voxel_object_contact (impact_voxel/src/collidable.rs:859-1049)
int orc_mutual_voxel_object_contacts(const orc_object* a, const float* probes_a, const uint32_t* entries_a, uint32_t n_entries_a, const float com_a[3],
                                     const float rotation_a[4], const float translation_a[3], const orc_object* b, const float* probes_b,
                                     const uint32_t* entries_b, uint32_t n_entries_b, const float com_b[3], const float rotation_b[4],
                                     const float translation_b[3], int cap, int32_t* which_ijk, float* position, float* normal, float* depth) {
    return mutual_contacts(a->obj, probes_a, entries_a, n_entries_a, com_a, rotation_a, translation_a, b->obj, probes_b, entries_b, n_entries_b, com_b, rotation_b,
                           translation_b, cap, which_ijk, position, normal, depth);
}

// for_each_capsule_voxel_object_contact (impact_voxel/src/collidable.rs:1257-1286)
int orc_capsule_voxel_object_contacts(const orc_object* o, const float rotation_xyzw[4], const float translation[3], const float segment_start[3],
                                      const float segment_vector[3], float radius, int cap, int32_t* indices, float* position, float* normal,
                                      float* depth) {
    return capsule_voxel_object_contacts(o->obj, rotation_xyzw, translation, segment_start, segment_vector, radius, cap, indices, position, normal, depth);
}

// for_each_voxel_object_plane_contact (impact_voxel/src/collidable.rs:1176-1208)
int orc_plane_voxel_object_contacts(const orc_object* o, const float rotation_xyzw[4], const float translation[3], const float plane_normal[3],
                                    float plane_displacement, int cap, int32_t* indices, float* position, float* normal, float* depth) {
    return plane_voxel_object_contacts(o->obj, rotation_xyzw, translation, plane_normal, plane_displacement, cap, indices, position, normal, depth);
}

int8_t orc_sd_from_f32(float v) { return sd_from_f32(v); }
float orc_sd_to_f32(int8_t e) { return sd_to_f32(e); }

}  // extern "C"
