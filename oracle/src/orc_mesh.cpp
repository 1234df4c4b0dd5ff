// ORACLE (test infrastructure only — never linked into the product library).
//
// CPU restatement of the chunk SDF padding fill, Surface Nets mesher and full mesh rebuild:
//   fill_sdf_for_chunk_if_exposed + padding fills   object/sdf.rs:156-508
//   gradient from corner samples                    object/sdf.rs:603-633
//   compute_surface_nets_mesh                       object/sdf/surface_nets.rs:131-637
//   CUBE_CORNERS / CUBE_EDGES                       object/sdf/surface_nets.rs:639-674
//   VoxelObjectMesh::recreate                       mesh.rs:286-354, 559-577
//   ChunkSubmesh obscuredness table                 mesh.rs:611-635
#include <memory>

#include "orc_mesh.hpp"

#include <algorithm>
#include <cstdint>
#include <cstring>

namespace orc {

static const int G = 18;        // SDF_GRID_SIZE (object/sdf.rs:35)
static const int G2 = G * G;
static const int GCELLS = G * G * G;

struct ChunkSdf {
    float values[GCELLS];
    uint8_t types[GCELLS];
    bool adj_non_uniform[3][2];
};

static inline int gidx(int i, int j, int k) { return i * G2 + j * G + k; }

// object/sdf.rs:478-508 applied per padded cell. The reference copies faces/edges/corners with
// positional Loop3 pairs; every padded cell (a,b,c) maps to object voxel (16*ci + a - 1, ...), so a
// per-cell gather is equivalent. Void (or outside-grid) neighbours leave `types` untouched (stale),
// which is unobservable: only negative-distance corners contribute materials (surface_nets.rs:215,471).
static void fill_sdf(const VoxelObject& obj, int ci, int cj, int ck, ChunkSdf& sdf) {
    for (int a = 0; a < G; ++a)
        for (int b = 0; b < G; ++b)
            for (int c = 0; c < G; ++c) {
                int oi = ci * CHUNK + a - 1, oj = cj * CHUNK + b - 1, ok = ck * CHUNK + c - 1;
                const Chunk* ch = (oi < 0 || oj < 0 || ok < 0) ? nullptr : obj.chunk_at(oi >> 4, oj >> 4, ok >> 4);
                int g = gidx(a, b, c);
                if (!ch || ch->kind == K_VOID) {
                    sdf.values[g] = sd_to_f32(127);
                } else if (ch->kind == K_UNIFORM) {
                    sdf.values[g] = sd_to_f32(-128);
                    sdf.types[g] = ch->uniform_voxel.type;
                } else {
                    const Voxel& v = obj.voxels[((size_t)ch->data_offset << 12) + (((oi & 15) << 8) | ((oj & 15) << 4) | (ok & 15))];
                    sdf.values[g] = sd_to_f32(v.sd);
                    sdf.types[g] = v.type;
                }
            }
    const int d[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int dim = 0; dim < 3; ++dim)
        for (int side = 0; side < 2; ++side) {
            int s = side ? 1 : -1;
            const Chunk* ch = obj.chunk_at(ci + s * d[dim][0], cj + s * d[dim][1], ck + s * d[dim][2]);
            sdf.adj_non_uniform[dim][side] = ch && ch->kind == K_NONUNIFORM;
        }
}

static const int CUBE_CORNERS[8][3] = {{0, 0, 0}, {0, 0, 1}, {0, 1, 0}, {0, 1, 1}, {1, 0, 0}, {1, 0, 1}, {1, 1, 0}, {1, 1, 1}};
static const int CUBE_EDGES[12][2] = {{0, 1}, {0, 2}, {0, 4}, {1, 3}, {1, 5}, {2, 3}, {2, 6}, {3, 7}, {4, 5}, {4, 6}, {5, 7}, {6, 7}};

// surface_nets.rs:384-418
static V3 centroid_of_edge_intersections(const float* d) {
    int count = 0;
    V3 sum{0, 0, 0};
    for (int e = 0; e < 12; ++e) {
        int c1 = CUBE_EDGES[e][0], c2 = CUBE_EDGES[e][1];
        float d1 = d[c1], d2 = d[c2];
        if (sign_neg(d1) != sign_neg(d2)) {
            count += 1;
            float interp1 = d1 / (d1 - d2);
            float interp2 = 1.0f - interp1;
            V3 p1 = v3((float)CUBE_CORNERS[c1][0], (float)CUBE_CORNERS[c1][1], (float)CUBE_CORNERS[c1][2]);
            V3 p2 = v3((float)CUBE_CORNERS[c2][0], (float)CUBE_CORNERS[c2][1], (float)CUBE_CORNERS[c2][2]);
            V3 p = interp2 * p1 + interp1 * p2;
            sum = sum + p;
        }
    }
    return div_recip(sum, (float)count);
}

// object/sdf.rs:603-633
static V3 gradient_from_corners(const float* d, V3 o) {
    V3 p00 = v3(d[4], d[2], d[1]), n00 = v3(d[0], d[0], d[0]);
    V3 p01 = v3(d[5], d[6], d[3]), n01 = v3(d[1], d[4], d[2]);
    V3 p10 = v3(d[6], d[3], d[5]), n10 = v3(d[2], d[1], d[4]);
    V3 p11 = v3(d[7], d[7], d[7]), n11 = v3(d[3], d[5], d[6]);
    V3 d00 = p00 - n00, d01 = p01 - n01, d10 = p10 - n10, d11 = p11 - n11;
    V3 r = v3s(1.0f) - o;
    auto yzx = [](V3 v) { return v3(v.y, v.z, v.x); };
    auto zxy = [](V3 v) { return v3(v.z, v.x, v.y); };
    return ((cmul(cmul(yzx(r), zxy(r)), d00) + cmul(cmul(yzx(r), zxy(o)), d01)) + cmul(cmul(yzx(o), zxy(r)), d10)) +
           cmul(cmul(yzx(o), zxy(o)), d11);
}

// surface_nets.rs:428-522
void vertex_materials_compute(const bool has_voxel[8], const uint8_t mat[8], VertexMaterials& m) {
    std::memset(&m, 0, sizeof(m));
    uint8_t map[256];
    std::memset(map, 255, sizeof(map));
    int count = 0;
    for (int c = 0; c < 8; ++c)
        if (has_voxel[c]) {
            uint8_t idx = map[mat[c]];
            if (idx == 255) {
                m.indices[count] = mat[c];
                m.weights[count] = 1;
                map[mat[c]] = (uint8_t)count;
                count += 1;
            } else {
                m.weights[idx] += 1;
            }
        }
    m.indices[7] = (uint8_t)count;
    // sorting_network_7 (surface_nets.rs:428-446), 17 compare-and-swaps in source order
    static const int NET[17][2] = {{0, 6}, {1, 5}, {2, 4}, {0, 3}, {1, 2}, {4, 5}, {0, 1}, {2, 3}, {4, 6},
                                   {5, 6}, {1, 4}, {3, 5}, {1, 2}, {3, 4}, {5, 6}, {2, 3}, {4, 5}};
    for (int s = 0; s < 17; ++s) {
        int i = NET[s][0], j = NET[s][1];
        if (m.weights[i] < m.weights[j]) {
            std::swap(m.indices[i], m.indices[j]);
            std::swap(m.weights[i], m.weights[j]);
        }
    }
}

// surface_nets.rs:558-637
void index_materials_for_triangle(const VertexMaterials* vm[3], IndexMaterials out[3]) {
    auto mcount = [](const VertexMaterials* v) { return (int)v->indices[7]; };
    if (mcount(vm[0]) == 1 && mcount(vm[1]) == 1 && mcount(vm[2]) == 1) {
        uint8_t index = vm[0]->indices[0];
        if (vm[1]->indices[0] == index && vm[2]->indices[0] == index) {
            IndexMaterials im{{index, 0, 0, 0}, {1, 0, 0, 0}};
            out[0] = out[1] = out[2] = im;
            return;
        }
    }
    uint8_t top[4] = {0, 0, 0, 0};
    int n_top = 0;
    bool is_top[256];
    std::memset(is_top, 0, sizeof(is_top));
    int off[3] = {0, 0, 0};
    for (int t = 0; t < 4; ++t) {
        uint8_t w[3];
        for (int i = 0; i < 3; ++i) w[i] = vm[i]->weights[off[i]];
        int mx = (w[0] >= w[1]) ? ((w[0] >= w[2]) ? 0 : 2) : ((w[1] >= w[2]) ? 1 : 2);
        if (w[mx] == 0) break;
        top[t] = vm[mx]->indices[off[mx]];
        n_top += 1;
        is_top[top[t]] = true;
        for (int i = 0; i < 3; ++i)
            while (off[i] < mcount(vm[i]) && is_top[vm[i]->indices[off[i]]]) off[i] += 1;
    }
    for (int v = 0; v < 3; ++v) {
        IndexMaterials im{{top[0], top[1], top[2], top[3]}, {0, 0, 0, 0}};
        for (int i = 0; i < n_top; ++i)
            for (int j = 0; j < mcount(vm[v]); ++j)
                if (vm[v]->indices[j] == top[i]) {
                    im.weights[i] = vm[v]->weights[j];
                    break;
                }
        out[v] = im;
    }
}

struct SurfaceNetsBuffer {
    std::vector<V3> positions, normals;
    std::vector<VertexMaterials> vmats;
    std::vector<IndexMaterials> imats;
    std::vector<uint16_t> indices;
    std::vector<uint32_t> surf;  // packed i | j<<8 | k<<16
    std::vector<uint16_t> surf_lin;
    uint16_t map[GCELLS];
};

// surface_nets.rs:336-381
static void maybe_make_quad(const ChunkSdf& sdf, const SurfaceNetsBuffer& buf, int p1, int p2, int axis_b, int axis_c,
                            std::vector<uint16_t>& indices) {
    float d1 = sdf.values[p1], d2 = sdf.values[p2];
    bool n1 = sign_neg(d1), n2 = sign_neg(d2);
    bool negative_face;
    if (n1 && !n2) negative_face = false;
    else if (!n1 && n2) negative_face = true;
    else return;
    uint16_t v1 = buf.map[p1], v2 = buf.map[p1 - axis_b], v3_ = buf.map[p1 - axis_c], v4 = buf.map[p1 - axis_b - axis_c];
    V3 q1 = buf.positions[v1], q2 = buf.positions[v2], q3 = buf.positions[v3_], q4 = buf.positions[v4];
    uint16_t quad[6];
    if (length(q1 - q4) < length(q2 - q3)) {
        if (negative_face) { uint16_t q[6] = {v1, v4, v2, v1, v3_, v4}; std::memcpy(quad, q, sizeof(q)); }
        else { uint16_t q[6] = {v1, v2, v4, v1, v4, v3_}; std::memcpy(quad, q, sizeof(q)); }
    } else if (negative_face) { uint16_t q[6] = {v2, v3_, v4, v2, v1, v3_}; std::memcpy(quad, q, sizeof(q)); }
    else { uint16_t q[6] = {v2, v4, v3_, v2, v3_, v1}; std::memcpy(quad, q, sizeof(q)); }
    indices.insert(indices.end(), quad, quad + 6);
}

// surface_nets.rs:131-303
static void compute_surface_nets_mesh(const ChunkSdf& sdf, float extent, V3 offset, SurfaceNetsBuffer& buf) {
    buf.positions.clear();
    buf.normals.clear();
    buf.vmats.clear();
    buf.imats.clear();
    buf.indices.clear();
    buf.surf.clear();
    buf.surf_lin.clear();
    for (int i = 0; i < GCELLS; ++i) buf.map[i] = 0xFFFF;
    for (int i = 0; i < G - 1; ++i)
        for (int j = 0; j < G - 1; ++j)
            for (int k = 0; k < G - 1; ++k) {
                int lin = gidx(i, j, k);
                float cd[8];
                bool has[8];
                int num_neg = 0;
                for (int c = 0; c < 8; ++c) {
                    cd[c] = sdf.values[lin + gidx(CUBE_CORNERS[c][0], CUBE_CORNERS[c][1], CUBE_CORNERS[c][2])];
                    has[c] = sign_neg(cd[c]);
                    if (has[c]) num_neg += 1;
                }
                if (num_neg == 0 || num_neg == 8) continue;
                uint8_t mats[8];
                for (int c = 0; c < 8; ++c) mats[c] = sdf.types[lin + gidx(CUBE_CORNERS[c][0], CUBE_CORNERS[c][1], CUBE_CORNERS[c][2])];
                V3 centroid = centroid_of_edge_intersections(cd);
                V3 gradient = gradient_from_corners(cd, centroid);
                V3 normal = normalize(gradient);
                VertexMaterials vm;
                vertex_materials_compute(has, mats, vm);
                V3 position = extent * (centroid + v3((float)i, (float)j, (float)k)) + offset;
                buf.map[lin] = (uint16_t)buf.positions.size();
                buf.surf.push_back((uint32_t)i | ((uint32_t)j << 8) | ((uint32_t)k << 16));
                buf.surf_lin.push_back((uint16_t)lin);
                buf.positions.push_back(position);
                buf.normals.push_back(normal);
                buf.vmats.push_back(vm);
            }
    int upper[3] = {G - 1, G - 1, G - 1};
    for (int d = 0; d < 3; ++d)
        if (sdf.adj_non_uniform[d][1]) upper[d] -= 1;
    for (size_t s = 0; s < buf.surf.size(); ++s) {
        int i = buf.surf[s] & 255, j = (buf.surf[s] >> 8) & 255, k = (buf.surf[s] >> 16) & 255;
        int p = buf.surf_lin[s];
        if (j != 0 && k != 0 && i < upper[0]) maybe_make_quad(sdf, buf, p, p + G2, G, 1, buf.indices);
        if (i != 0 && k != 0 && j < upper[1]) maybe_make_quad(sdf, buf, p, p + G, 1, G2, buf.indices);
        if (i != 0 && j != 0 && k < upper[2]) maybe_make_quad(sdf, buf, p, p + 1, G2, G, buf.indices);
    }
    buf.imats.resize(buf.indices.size());
    for (size_t t = 0; t + 2 < buf.indices.size(); t += 3) {
        const VertexMaterials* vm[3] = {&buf.vmats[buf.indices[t]], &buf.vmats[buf.indices[t + 1]], &buf.vmats[buf.indices[t + 2]]};
        index_materials_for_triangle(vm, &buf.imats[t]);
    }
}

// object/sdf.rs:181-213 for one chunk (used by the validate_sdf restatement in tests/)
bool chunk_sdf_if_exposed(const VoxelObject& obj, int ci, int cj, int ck, float* values, uint8_t* types) {
    const Chunk* c = obj.chunk_at(ci, cj, ck);
    if (!c || !(c->kind == K_NONUNIFORM && (c->flags & CF_FULLY_OBSCURED) != CF_FULLY_OBSCURED)) return false;
    static thread_local ChunkSdf sdf;
    std::memset(sdf.types, TYPE_DUMMY, sizeof(sdf.types));
    fill_sdf(obj, ci, cj, ck, sdf);
    std::memcpy(values, sdf.values, sizeof(sdf.values));
    std::memcpy(types, sdf.types, sizeof(sdf.types));
    return true;
}

bool RangeAllocator::allocate_range(size_t required_len, size_t& start) {  // range_allocator.rs:43-72: the smallest free range that fits, first on ties
    auto best = free_ranges.end();
    size_t best_len = SIZE_MAX;
    for (auto it = free_ranges.begin(); it != free_ranges.end(); ++it) {
        const size_t len = it->second - it->first;
        if (len < best_len && len >= required_len) {
            best = it;
            best_len = len;
        }
    }
    if (best == free_ranges.end()) return false;
    const size_t s0 = best->first, e0 = best->second;
    free_ranges.erase(best);
    if (s0 + required_len < e0) free_ranges.emplace(s0 + required_len, e0);
    start = s0;
    return true;
}

void RangeAllocator::merge_consecutive_ranges() {  // range_allocator.rs:74-108: every run of touching ranges becomes one range
    if (free_ranges.size() < 2) return;
    std::map<size_t, size_t> merged;
    auto it = free_ranges.begin();
    size_t s0 = it->first, e0 = it->second;
    for (++it; it != free_ranges.end(); ++it) {
        if (it->first == e0) {
            e0 = it->second;
        } else {
            merged.emplace(s0, e0);
            s0 = it->first;
            e0 = it->second;
        }
    }
    merged.emplace(s0, e0);
    free_ranges.swap(merged);
}

static inline uint64_t chunk_key(uint32_t i, uint32_t j, uint32_t k) { return ((uint64_t)i << 42) | ((uint64_t)j << 21) | (uint64_t)k; }

static void obscured_table(uint8_t flags, uint32_t out[2][2][2]) {  // ChunkSubmesh::new (mesh.rs:611-635)
    const uint8_t OX[2] = {CF_OBSC_X_DN, CF_OBSC_X_UP}, OY[2] = {CF_OBSC_Y_DN, CF_OBSC_Y_UP}, OZ[2] = {CF_OBSC_Z_DN, CF_OBSC_Z_UP};
    for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 2; ++b)
            for (int cc = 0; cc < 2; ++cc) out[a][b][cc] = ((flags & OX[a]) && (flags & OY[b]) && (flags & OZ[cc])) ? 1u : 0u;
}

// mesh.rs:286-354
void mesh_recreate(const VoxelObject& obj, Mesh& mesh) {
    mesh = Mesh{};
    static thread_local ChunkSdf sdf;
    static thread_local SurfaceNetsBuffer buf;
    std::memset(sdf.types, TYPE_DUMMY, sizeof(sdf.types));
    for (int i = 0; i < GCELLS; ++i) sdf.values[i] = 0.0f;
    float chunk_extent = (float)CHUNK * obj.extent;
    for (int ci = 0; ci < obj.cc[0]; ++ci)
        for (int cj = 0; cj < obj.cc[1]; ++cj)
            for (int ck = 0; ck < obj.cc[2]; ++ck) {
                const Chunk& c = obj.chunks[obj.cidx(ci, cj, ck)];
                if (!(c.kind == K_NONUNIFORM && (c.flags & CF_FULLY_OBSCURED) != CF_FULLY_OBSCURED)) continue;
                fill_sdf(obj, ci, cj, ck, sdf);
                V3 offset = v3((float)ci * chunk_extent - 0.5f * obj.extent, (float)cj * chunk_extent - 0.5f * obj.extent,
                               (float)ck * chunk_extent - 0.5f * obj.extent);
                compute_surface_nets_mesh(sdf, obj.extent, offset, buf);
                if (buf.indices.empty()) continue;
                uint32_t voff = (uint32_t)mesh.positions.size();
                Submesh sm{};
                sm.chunk[0] = (uint32_t)ci;
                sm.chunk[1] = (uint32_t)cj;
                sm.chunk[2] = (uint32_t)ck;
                sm.index_offset = (uint32_t)mesh.indices.size();
                sm.index_count = (uint32_t)buf.indices.size();
                sm.vertex_offset = voff;
                sm.vertex_count = (uint32_t)buf.positions.size();
                obscured_table(c.flags, sm.obscured);
                mesh.chunk_index[chunk_key(sm.chunk[0], sm.chunk[1], sm.chunk[2])] = mesh.submeshes.size();  // push_chunk
                mesh.submeshes.push_back(sm);
                mesh.positions.insert(mesh.positions.end(), buf.positions.begin(), buf.positions.end());
                mesh.normals.insert(mesh.normals.end(), buf.normals.begin(), buf.normals.end());
                mesh.index_materials.insert(mesh.index_materials.end(), buf.imats.begin(), buf.imats.end());
                for (uint16_t ix : buf.indices) mesh.indices.push_back(voff + (uint32_t)ix);
            }
}

// all-cores variant of mesh_recreate (bench.py's `cpu_baseline_all_cores`): the chunk meshes are computed by a pool of threads and
// concatenated in chunk-linear order afterwards — the same buffers as the sequential function (checked by tests/test_oracle_parallel.py).
void mesh_recreate_parallel(const VoxelObject& obj, Mesh& mesh, int threads) {
    mesh = Mesh{};
    const int n = obj.n_chunks();
    const float chunk_extent = (float)CHUNK * obj.extent;
    std::vector<int> exposed;
    for (int c = 0; c < n; ++c) {
        const Chunk& ch = obj.chunks[c];
        if (ch.kind == K_NONUNIFORM && (ch.flags & CF_FULLY_OBSCURED) != CF_FULLY_OBSCURED) exposed.push_back(c);
    }
    std::vector<SurfaceNetsBuffer> parts(exposed.size());
#pragma omp parallel num_threads(threads)
    {
        std::unique_ptr<ChunkSdf> sdf(new ChunkSdf());
        std::memset(sdf->types, TYPE_DUMMY, sizeof(sdf->types));
        for (int i = 0; i < GCELLS; ++i) sdf->values[i] = 0.0f;
#pragma omp for schedule(dynamic, 8)
        for (size_t e = 0; e < exposed.size(); ++e) {
            const int c = exposed[e];
            const int ci = c / (obj.cc[2] * obj.cc[1]), cj = (c / obj.cc[2]) % obj.cc[1], ck = c % obj.cc[2];
            fill_sdf(obj, ci, cj, ck, *sdf);
            V3 offset = v3((float)ci * chunk_extent - 0.5f * obj.extent, (float)cj * chunk_extent - 0.5f * obj.extent,
                           (float)ck * chunk_extent - 0.5f * obj.extent);
            compute_surface_nets_mesh(*sdf, obj.extent, offset, parts[e]);
            parts[e].surf.clear();
            parts[e].surf.shrink_to_fit();
            parts[e].surf_lin.clear();
            parts[e].surf_lin.shrink_to_fit();
        }
    }
    size_t nv = 0, ni = 0;
    for (const auto& b : parts) {
        nv += b.positions.size();
        ni += b.indices.size();
    }
    mesh.positions.reserve(nv);
    mesh.normals.reserve(nv);
    mesh.indices.reserve(ni);
    mesh.index_materials.reserve(ni);
    for (size_t e = 0; e < exposed.size(); ++e) {
        const SurfaceNetsBuffer& buf = parts[e];
        if (buf.indices.empty()) continue;
        const int c = exposed[e];
        const Chunk& ch = obj.chunks[c];
        uint32_t voff = (uint32_t)mesh.positions.size();
        Submesh sm{};
        sm.chunk[0] = (uint32_t)(c / (obj.cc[2] * obj.cc[1]));
        sm.chunk[1] = (uint32_t)((c / obj.cc[2]) % obj.cc[1]);
        sm.chunk[2] = (uint32_t)(c % obj.cc[2]);
        sm.index_offset = (uint32_t)mesh.indices.size();
        sm.index_count = (uint32_t)buf.indices.size();
        sm.vertex_offset = voff;
        sm.vertex_count = (uint32_t)buf.positions.size();
        obscured_table(ch.flags, sm.obscured);
        mesh.chunk_index[chunk_key(sm.chunk[0], sm.chunk[1], sm.chunk[2])] = mesh.submeshes.size();
        mesh.submeshes.push_back(sm);
        mesh.positions.insert(mesh.positions.end(), buf.positions.begin(), buf.positions.end());
        mesh.normals.insert(mesh.normals.end(), buf.normals.begin(), buf.normals.end());
        mesh.index_materials.insert(mesh.index_materials.end(), buf.imats.begin(), buf.imats.end());
        for (uint16_t ix : buf.indices) mesh.indices.push_back(voff + (uint32_t)ix);
    }
}

static void remove_chunk_if_present(Mesh& mesh, uint64_t key) {  // mesh.rs:811-824 (swap_remove keeps the tables dense)
    auto it = mesh.chunk_index.find(key);
    if (it == mesh.chunk_index.end()) return;
    const size_t idx = it->second;
    mesh.chunk_index.erase(it);
    const Submesh gone = mesh.submeshes[idx];
    if (idx + 1 != mesh.submeshes.size()) {
        mesh.submeshes[idx] = mesh.submeshes.back();
        const Submesh& moved = mesh.submeshes[idx];
        mesh.chunk_index[chunk_key(moved.chunk[0], moved.chunk[1], moved.chunk[2])] = idx;
    }
    mesh.submeshes.pop_back();
    mesh.chunks_were_removed = true;
    mesh.vertex_ranges.free_range(gone.vertex_offset, (size_t)gone.vertex_offset + gone.vertex_count);
    mesh.index_ranges.free_range(gone.index_offset, (size_t)gone.index_offset + gone.index_count);
}

// mesh.rs:355-456 with ChunkSubmeshManager::write_chunk (mesh.rs:751-809). ORDER: the reference walks a hash set of chunk indices (unpinned
// iteration order, it decides which freed range a chunk's data lands in); here the invalidated chunks are visited in chunk-linear order.
void mesh_sync(const VoxelObject& obj, Mesh& mesh, const uint8_t* invalidated) {
    static thread_local ChunkSdf sdf;
    static thread_local SurfaceNetsBuffer buf;
    std::memset(sdf.types, TYPE_DUMMY, sizeof(sdf.types));
    for (int i = 0; i < GCELLS; ++i) sdf.values[i] = 0.0f;
    const float chunk_extent = (float)CHUNK * obj.extent;
    for (int ci = 0; ci < obj.cc[0]; ++ci)
        for (int cj = 0; cj < obj.cc[1]; ++cj)
            for (int ck = 0; ck < obj.cc[2]; ++ck) {
                if (!invalidated[obj.cidx(ci, cj, ck)]) continue;
                const Chunk& c = obj.chunks[obj.cidx(ci, cj, ck)];
                const uint64_t key = chunk_key((uint32_t)ci, (uint32_t)cj, (uint32_t)ck);
                if (!(c.kind == K_NONUNIFORM && (c.flags & CF_FULLY_OBSCURED) != CF_FULLY_OBSCURED)) {
                    remove_chunk_if_present(mesh, key);
                    continue;
                }
                fill_sdf(obj, ci, cj, ck, sdf);
                const V3 offset = v3((float)ci * chunk_extent - 0.5f * obj.extent, (float)cj * chunk_extent - 0.5f * obj.extent,
                                     (float)ck * chunk_extent - 0.5f * obj.extent);
                compute_surface_nets_mesh(sdf, obj.extent, offset, buf);
                if (buf.indices.empty()) {
                    remove_chunk_if_present(mesh, key);
                    continue;
                }
                const size_t total_v = mesh.positions.size(), total_i = mesh.indices.size();
                const size_t nv = buf.positions.size(), ni = buf.indices.size();
                // write_chunk
                auto it = mesh.chunk_index.find(key);
                if (it != mesh.chunk_index.end()) {
                    const Submesh& old = mesh.submeshes[it->second];
                    mesh.vertex_ranges.free_range(old.vertex_offset, (size_t)old.vertex_offset + old.vertex_count);
                    mesh.index_ranges.free_range(old.index_offset, (size_t)old.index_offset + old.index_count);
                }
                size_t v0, i0;
                if (!mesh.vertex_ranges.allocate_range(nv, v0)) v0 = total_v;
                if (!mesh.index_ranges.allocate_range(ni, i0)) i0 = total_i;
                Submesh sm{};
                sm.chunk[0] = (uint32_t)ci, sm.chunk[1] = (uint32_t)cj, sm.chunk[2] = (uint32_t)ck;
                sm.index_offset = (uint32_t)i0;
                sm.index_count = (uint32_t)ni;
                sm.vertex_offset = (uint32_t)v0;
                sm.vertex_count = (uint32_t)nv;
                obscured_table(c.flags, sm.obscured);
                mesh.updated_data_ranges.insert(mesh.updated_data_ranges.end(), {(uint32_t)v0, (uint32_t)(v0 + nv), (uint32_t)i0, (uint32_t)(i0 + ni)});
                if (it != mesh.chunk_index.end()) {
                    mesh.submeshes[it->second] = sm;
                } else {
                    mesh.chunk_index[key] = mesh.submeshes.size();
                    mesh.submeshes.push_back(sm);
                }
                if (v0 == total_v) {
                    mesh.positions.insert(mesh.positions.end(), buf.positions.begin(), buf.positions.end());
                    mesh.normals.insert(mesh.normals.end(), buf.normals.begin(), buf.normals.end());
                } else {
                    std::copy(buf.positions.begin(), buf.positions.end(), mesh.positions.begin() + (std::ptrdiff_t)v0);
                    std::copy(buf.normals.begin(), buf.normals.end(), mesh.normals.begin() + (std::ptrdiff_t)v0);
                }
                if (i0 == total_i) {
                    mesh.index_materials.insert(mesh.index_materials.end(), buf.imats.begin(), buf.imats.end());
                    for (uint16_t ix : buf.indices) mesh.indices.push_back((uint32_t)v0 + (uint32_t)ix);
                } else {
                    std::copy(buf.imats.begin(), buf.imats.end(), mesh.index_materials.begin() + (std::ptrdiff_t)i0);
                    for (size_t q = 0; q < ni; ++q) mesh.indices[i0 + q] = (uint32_t)v0 + (uint32_t)buf.indices[q];
                }
            }
    mesh.vertex_ranges.merge_consecutive_ranges();  // perform_maintainance
    mesh.index_ranges.merge_consecutive_ranges();
}

}  // namespace orc
