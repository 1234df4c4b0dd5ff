// ORACLE (test infrastructure only — never linked into the product library).
//
// Sequential CPU restatement of the reference's chunked voxel object generation and derived
// state, following the reference's own order of operations (not the closed-form rule the HIP
// kernels use — that is the point: the two must agree).
//   generate_without_derived_state        object.rs:307-404
//   create_for_generated_voxels           object.rs:1890-1964
//   analyze_and_initialize_chunks         object.rs:560-617
//   update_occupied_voxel_ranges          object.rs:1187-1280
//   update_internal_adjacencies           object.rs:2673-2756
//   update_all_chunk_boundary_adjacencies object.rs:1659-1785
//   update_mutual_face_adjacencies        object.rs:2077-2528
//   convert_to_non_uniform_if_uniform     object.rs:2530-2550, split_detection.rs:614-639
//   local connected regions               split_detection.rs:662-891, 1776-1882
#include <algorithm>
#include <cassert>
#include <climits>

#include "orc_voxel.hpp"

namespace orc {

static inline int lin(int i, int j, int k) { return (i << 8) | (j << 4) | k; }

// object.rs:1890-1964
static Chunk create_for_generated_voxels(const Voxel* v, ChunkSparseness sp) {
    Chunk c;
    if (sp.is_void) {
        c.kind = c.gen_kind = K_VOID;
        return c;
    }
    if (sp.only_empty) {
        c.kind = c.gen_kind = K_NONUNIFORM;
        c.flags = CF_ONLY_EMPTY;
        return c;  // all face distributions Empty
    }
    Voxel first = v[0];
    bool uniform = true;
    int empty_counts[3][2] = {{0, 0}, {0, 0}, {0, 0}};
    for (int i = 0; i < CHUNK; ++i)
        for (int j = 0; j < CHUNK; ++j)
            for (int k = 0; k < CHUNK; ++k) {
                const Voxel& x = v[lin(i, j, k)];
                if (uniform && (!(x.type == first.type && x.flags == first.flags) || x.sd != -128)) uniform = false;
                if (x.empty()) {
                    if (i == 0) empty_counts[0][0]++;
                    else if (i == CHUNK - 1) empty_counts[0][1]++;
                    if (j == 0) empty_counts[1][0]++;
                    else if (j == CHUNK - 1) empty_counts[1][1]++;
                    if (k == 0) empty_counts[2][0]++;
                    else if (k == CHUNK - 1) empty_counts[2][1]++;
                }
            }
    if (uniform) {
        first.flags |= F_FULL_ADJ;
        c.kind = c.gen_kind = K_UNIFORM;
        c.uniform_voxel = first;
    } else {
        c.kind = c.gen_kind = K_NONUNIFORM;
        for (int d = 0; d < 3; ++d)
            for (int s = 0; s < 2; ++s)
                c.face[d][s] = empty_counts[d][s] == 256 ? FD_EMPTY : (empty_counts[d][s] == 0 ? FD_FULL : FD_MIXED);
    }
    return c;
}

void generate_without_derived_state(VoxelObject& obj, const Generator& gen) {
    obj.extent = gen.voxel_extent();
    int shape[3];
    gen.grid_shape(shape);
    for (int d = 0; d < 3; ++d) obj.cc[d] = (shape[d] + CHUNK - 1) / CHUNK;
    int n = obj.n_chunks();
    obj.chunks.assign(n, Chunk{});
    obj.voxels.clear();
    obj.labels.clear();
    // object.rs:384-403
    for (int ci = 0; ci < n; ++ci) {
        int i = ci / (obj.cc[2] * obj.cc[1]);
        int j = (ci / obj.cc[2]) % obj.cc[1];
        int k = ci % obj.cc[2];
        int origin[3] = {i * CHUNK, j * CHUNK, k * CHUNK};
        size_t old = obj.voxels.size();
        obj.voxels.resize(old + CHUNK_VOXELS, Voxel{0, 0, 0});
        ChunkSparseness sp = gen.generate_chunk(&obj.voxels[old], origin);
        obj.chunks[ci] = create_for_generated_voxels(&obj.voxels[old], sp);
        if (obj.chunks[ci].kind != K_NONUNIFORM) obj.voxels.resize(old);
    }
    // analyze_and_initialize_chunks (object.rs:560-617)
    int lo[3] = {INT_MAX, INT_MAX, INT_MAX}, hi[3] = {0, 0, 0};
    bool any = false;
    uint32_t nu = 0;
    for (int ci = 0; ci < n; ++ci) {
        Chunk& c = obj.chunks[ci];
        if (c.kind == K_NONUNIFORM) c.data_offset = nu++;
        bool only_empty = c.kind == K_VOID || (c.kind == K_NONUNIFORM && (c.flags & CF_ONLY_EMPTY));
        if (!only_empty) {
            int idx[3] = {ci / (obj.cc[2] * obj.cc[1]), (ci / obj.cc[2]) % obj.cc[1], ci % obj.cc[2]};
            for (int d = 0; d < 3; ++d) {
                lo[d] = std::min(lo[d], idx[d]);
                hi[d] = std::max(hi[d], idx[d] + 1);
            }
            any = true;
        }
    }
    for (int d = 0; d < 3; ++d) {
        obj.occ_chunk[d][0] = any ? lo[d] : 0;
        obj.occ_chunk[d][1] = any ? hi[d] : 0;
        obj.occ_voxel[d][0] = obj.occ_chunk[d][0] * CHUNK;
        obj.occ_voxel[d][1] = obj.occ_chunk[d][1] * CHUNK;
    }
    obj.labels.assign(obj.voxels.size(), 0);  // SplitDetector::new (split_detection.rs:589)
}

// object.rs:1256-1280 (Loop3::over_all_from_side + short-circuit)
static int find_voxel_bound_in_chunk(const Voxel* v, int dim, int side) {
    for (int a = 0; a < CHUNK; ++a) {
        int p = side == 0 ? a : CHUNK - 1 - a;
        for (int b = 0; b < CHUNK; ++b)
            for (int c = 0; c < CHUNK; ++c) {
                int i, j, k;
                if (dim == 0) { i = p; j = b; k = c; }
                else if (dim == 1) { j = p; i = b; k = c; }
                else { k = p; i = b; j = c; }
                if (!v[lin(i, j, k)].empty()) return p;
            }
    }
    return -1;
}

// object.rs:1187-1253
void update_occupied_voxel_ranges(VoxelObject& obj) {
    for (int d = 0; d < 3; ++d)
        if (obj.occ_chunk[d][0] >= obj.occ_chunk[d][1]) {
            for (int e = 0; e < 3; ++e) obj.occ_voxel[e][0] = obj.occ_voxel[e][1] = 0;
            return;
        }
    int res[3][2];
    for (int dim = 0; dim < 3; ++dim) {
        int o0 = dim == 0 ? 1 : 0, o1 = dim == 2 ? 1 : 2;
        for (int side = 0; side < 2; ++side) {
            int chunk_l = side == 0 ? obj.occ_chunk[dim][0] : obj.occ_chunk[dim][1] - 1;
            long bound = side == 0 ? LONG_MAX : 0;
            int start = chunk_l * CHUNK;
            bool done = false;
            for (int m = obj.occ_chunk[o0][0]; m < obj.occ_chunk[o0][1] && !done; ++m)
                for (int n = obj.occ_chunk[o1][0]; n < obj.occ_chunk[o1][1] && !done; ++n) {
                    int ci[3];
                    ci[dim] = chunk_l;
                    ci[o0] = m;
                    ci[o1] = n;
                    const Chunk& c = obj.chunks[obj.cidx(ci[0], ci[1], ci[2])];
                    if (c.kind == K_UNIFORM) {
                        bound = side == 0 ? start : start + CHUNK - 1;
                        done = true;
                    } else if (c.kind == K_NONUNIFORM) {
                        int b = find_voxel_bound_in_chunk(&obj.voxels[(size_t)c.data_offset << 12], dim, side);
                        if (b >= 0) bound = side == 0 ? std::min<long>(bound, start + b) : std::max<long>(bound, start + b);
                    }
                }
            res[dim][side] = (int)bound;
        }
    }
    for (int d = 0; d < 3; ++d) {
        obj.occ_voxel[d][0] = res[d][0];
        obj.occ_voxel[d][1] = res[d][1] + 1;
    }
}

// object.rs:2673-2756
static void update_internal_adjacencies(Voxel* cv) {
    for (int i = 0; i < CHUNK; ++i)
        for (int j = 0; j < CHUNK; ++j)
            for (int k = 0; k < CHUNK; ++k) {
                int idx = lin(i, j, k);
                Voxel voxel = cv[idx];
                const int adj[3][3] = {{i + 1, j, k}, {i, j + 1, k}, {i, j, k + 1}};
                const uint8_t up[3] = {F_X_UP, F_Y_UP, F_Z_UP};
                const uint8_t dn[3] = {F_X_DN, F_Y_DN, F_Z_DN};
                if (voxel.empty()) {
                    for (int d = 0; d < 3; ++d)
                        if (adj[d][d] < CHUNK) cv[lin(adj[d][0], adj[d][1], adj[d][2])].flags &= (uint8_t)~dn[d];
                } else {
                    uint8_t flags = voxel.flags;
                    for (int d = 0; d < 3; ++d)
                        if (adj[d][d] < CHUNK) {
                            Voxel& a = cv[lin(adj[d][0], adj[d][1], adj[d][2])];
                            if (a.empty()) flags &= (uint8_t)~up[d];
                            else {
                                flags |= up[d];
                                a.flags |= dn[d];
                            }
                        }
                    cv[idx].flags = flags;
                }
            }
}

// split_detection.rs:1776-1802 (checked variant; the `unchecked` variant yields identical roots,
// SURVEY.md §9.13)
static int find_root(uint16_t* parents, int idx) {
    int p = parents[idx];
    if (p == idx) return p;
    int r = find_root(parents, p);
    parents[idx] = (uint16_t)r;
    return r;
}
static int find_root_nc(const uint16_t* parents, int idx) {
    while (parents[idx] != idx) idx = parents[idx];
    return idx;
}
static void assign_parent(uint16_t* parents, int voxel, int parent) {
    int r = find_root(parents, voxel);
    if (r != parent) parents[r] = (uint16_t)parent;
}

// split_detection.rs:662-891 with occupied ranges = full chunk (644-656)
void update_local_connected_regions_for_chunk(VoxelObject& obj, int chunk_idx) {
    Chunk& c = obj.chunks[chunk_idx];
    assert(c.kind == K_NONUNIFORM);
    const Voxel* v = &obj.voxels[(size_t)c.data_offset << 12];
    uint8_t* labels = &obj.labels[(size_t)c.data_offset << 12];
    uint16_t parents[CHUNK_VOXELS];
    for (int i = 0; i < CHUNK_VOXELS; ++i) parents[i] = (uint16_t)i;
    if (!(c.flags & CF_ONLY_EMPTY)) {
        for (int i = 0; i < CHUNK; ++i)
            for (int j = 0; j < CHUNK; ++j)
                for (int k = 0; k < CHUNK; ++k) {
                    int idx = lin(i, j, k);
                    const Voxel& x = v[idx];
                    if (!x.empty()) {
                        int root = find_root(parents, idx);
                        if (i < CHUNK - 1 && (x.flags & F_X_UP)) assign_parent(parents, lin(i + 1, j, k), root);
                        if (j < CHUNK - 1 && (x.flags & F_Y_UP)) assign_parent(parents, lin(i, j + 1, k), root);
                        if (k < CHUNK - 1 && (x.flags & F_Z_UP)) assign_parent(parents, lin(i, j, k + 1), root);
                    }
                }
    }
    int current = 0;
    const int MAX_BOUNDARY_LABEL = 255;  // CHUNK_MAX_BOUNDARY_REGIONS - 1
    auto visit = [&](int i, int j, int k) {
        int idx = lin(i, j, k);
        if (!v[idx].empty()) {
            int set_id = find_root(parents, idx);
            int si = set_id >> 8, sj = (set_id >> 4) & 15, sk = set_id & 15;
            bool root_interior = si > 0 && si < CHUNK - 1 && sj > 0 && sj < CHUNK - 1 && sk > 0 && sk < CHUNK - 1;
            if (set_id == idx) {
                labels[idx] = (uint8_t)current;
                current = std::min(std::min(current + 1, 255), MAX_BOUNDARY_LABEL);
            } else if (root_interior) {
                parents[set_id] = (uint16_t)idx;  // make_voxel_root (1877-1882)
                parents[idx] = (uint16_t)idx;
                labels[idx] = (uint8_t)current;
                current = std::min(std::min(current + 1, 255), MAX_BOUNDARY_LABEL);
            }
        } else {
            labels[idx] = 255;
        }
    };
    // Loop3::over_full_boundary (utils.rs:247-322)
    for (int side = 0; side < 2; ++side) {  // X-, X+: i fixed, j then k
        int i = side ? CHUNK - 1 : 0;
        for (int j = 0; j < CHUNK; ++j)
            for (int k = 0; k < CHUNK; ++k) visit(i, j, k);
    }
    for (int side = 0; side < 2; ++side) {  // Y-, Y+: j outer, i in 1..15, k
        int j = side ? CHUNK - 1 : 0;
        for (int i = 1; i < CHUNK - 1; ++i)
            for (int k = 0; k < CHUNK; ++k) visit(i, j, k);
    }
    for (int side = 0; side < 2; ++side) {  // Z-, Z+: k outer, i, j in 1..15
        int k = side ? CHUNK - 1 : 0;
        for (int i = 1; i < CHUNK - 1; ++i)
            for (int j = 1; j < CHUNK - 1; ++j) visit(i, j, k);
    }
    assert(current < MAX_BOUNDARY_LABEL);
    c.boundary_region_count = (uint16_t)current;
    const int MAX_LABEL = 255;
    for (int i = 1; i < CHUNK - 1; ++i)
        for (int j = 1; j < CHUNK - 1; ++j)
            for (int k = 1; k < CHUNK - 1; ++k) {
                int idx = lin(i, j, k);
                if (parents[idx] == idx) {
                    if (!v[idx].empty()) {
                        labels[idx] = (uint8_t)current;
                        current = std::min(std::min(current + 1, 255), MAX_LABEL);
                    } else {
                        labels[idx] = 255;
                    }
                }
            }
    assert(current < MAX_LABEL);
    c.region_count = (uint16_t)current;
    for (int idx = 0; idx < CHUNK_VOXELS; ++idx)
        if (!v[idx].empty()) {
            int set_id = find_root_nc(parents, idx);
            if (set_id != idx) labels[idx] = labels[set_id];
        }
}

// object.rs:2530-2550 + split_detection.rs:614-639
static void convert_to_non_uniform_if_uniform(VoxelObject& obj, int ci) {
    Chunk& c = obj.chunks[ci];
    if (c.kind != K_UNIFORM) return;
    size_t start = obj.voxels.size();
    obj.voxels.resize(start + CHUNK_VOXELS, c.uniform_voxel);
    obj.labels.resize(start + CHUNK_VOXELS, 0);
    c.kind = K_NONUNIFORM;
    c.data_offset = (uint32_t)(start >> 12);
    for (int d = 0; d < 3; ++d) c.face[d][0] = c.face[d][1] = FD_FULL;
    c.flags = CF_FULLY_OBSCURED;
    c.region_count = 1;
    c.boundary_region_count = 1;
}

// Loop3::over_face(dim, side) visits (object.rs:2582-2600)
template <class F>
static void for_face(int dim, int side, F f) {
    int p = side ? CHUNK - 1 : 0;
    for (int a = 0; a < CHUNK; ++a)
        for (int b = 0; b < CHUNK; ++b) {
            if (dim == 0) f(p, a, b);
            else if (dim == 1) f(a, p, b);
            else f(a, b, p);
        }
}
static void set_all_outward(VoxelObject& obj, uint32_t off, int dim, int side, bool add) {
    Voxel* cv = &obj.voxels[(size_t)off << 12];
    uint8_t flag = adjacency_flag_for_face(dim, side);
    for_face(dim, side, [&](int i, int j, int k) {
        if (add) cv[lin(i, j, k)].flags |= flag;
        else cv[lin(i, j, k)].flags &= (uint8_t)~flag;
    });
}
// object.rs:2602-2652
static void update_outward_with_non_uniform(VoxelObject& obj, uint32_t cur_off, uint32_t adj_off, int dim, int side) {
    Voxel* cur = &obj.voxels[(size_t)cur_off << 12];
    const Voxel* adj = &obj.voxels[(size_t)adj_off << 12];
    uint8_t flag = adjacency_flag_for_face(dim, side);
    int q = side ? 0 : CHUNK - 1;  // opposite face position in the adjacent chunk
    for_face(dim, side, [&](int i, int j, int k) {
        int ai = i, aj = j, ak = k;
        if (dim == 0) ai = q;
        else if (dim == 1) aj = q;
        else ak = q;
        Voxel& x = cur[lin(i, j, k)];
        if (!x.empty()) {
            if (adj[lin(ai, aj, ak)].empty()) x.flags &= (uint8_t)~flag;
            else x.flags |= flag;
        }
    });
}
static inline void mark_face(Chunk& c, int dim, int side, bool obscured) {
    if (c.kind != K_NONUNIFORM) return;  // object.rs:2028-2075 (Void/Uniform ignored)
    uint8_t bit = (uint8_t)(1u << (side * 3 + dim));
    if (obscured) c.flags |= bit;
    else c.flags &= (uint8_t)~bit;
}

// object.rs:2077-2528 (split-detector connection bookkeeping omitted: not observable at the boundary)
static void update_mutual_face_adjacencies(VoxelObject& obj, int lower_idx, int upper_idx, int dim) {
    Chunk void_chunk;
    Chunk lower = lower_idx >= 0 ? obj.chunks[lower_idx] : void_chunk;  // copies, as in the reference
    Chunk upper = upper_idx >= 0 ? obj.chunks[upper_idx] : void_chunk;
    auto L = [&]() -> Chunk& { return obj.chunks[lower_idx]; };
    auto U = [&]() -> Chunk& { return obj.chunks[upper_idx]; };
    if (lower.kind == K_VOID && upper.kind == K_VOID) return;
    if (lower.kind == K_UNIFORM && upper.kind == K_UNIFORM) return;
    if (lower.kind == K_UNIFORM && upper.kind == K_VOID) {
        convert_to_non_uniform_if_uniform(obj, lower_idx);
        set_all_outward(obj, L().data_offset, dim, 1, false);
        mark_face(L(), dim, 1, false);
        return;
    }
    if (lower.kind == K_VOID && upper.kind == K_UNIFORM) {
        convert_to_non_uniform_if_uniform(obj, upper_idx);
        set_all_outward(obj, U().data_offset, dim, 0, false);
        mark_face(U(), dim, 0, false);
        return;
    }
    if (lower.kind == K_NONUNIFORM && upper.kind == K_VOID) {
        if (lower.face[dim][1] != FD_EMPTY) set_all_outward(obj, lower.data_offset, dim, 1, false);
        mark_face(L(), dim, 1, false);
        return;
    }
    if (lower.kind == K_VOID && upper.kind == K_NONUNIFORM) {
        if (upper.face[dim][0] != FD_EMPTY) set_all_outward(obj, upper.data_offset, dim, 0, false);
        mark_face(U(), dim, 0, false);
        return;
    }
    if (lower.kind == K_NONUNIFORM && upper.kind == K_UNIFORM) {
        uint8_t fd = lower.face[dim][1];
        if (fd != FD_EMPTY) set_all_outward(obj, lower.data_offset, dim, 1, true);
        mark_face(L(), dim, 1, true);
        if (fd == FD_EMPTY) {
            convert_to_non_uniform_if_uniform(obj, upper_idx);
            set_all_outward(obj, U().data_offset, dim, 0, false);
            mark_face(U(), dim, 0, false);
        } else if (fd == FD_MIXED) {
            convert_to_non_uniform_if_uniform(obj, upper_idx);
            update_outward_with_non_uniform(obj, U().data_offset, lower.data_offset, dim, 0);
            mark_face(U(), dim, 0, false);
        }
        return;
    }
    if (lower.kind == K_UNIFORM && upper.kind == K_NONUNIFORM) {
        uint8_t fd = upper.face[dim][0];
        if (fd != FD_EMPTY) set_all_outward(obj, upper.data_offset, dim, 0, true);
        mark_face(U(), dim, 0, true);
        if (fd == FD_EMPTY) {
            convert_to_non_uniform_if_uniform(obj, lower_idx);
            set_all_outward(obj, L().data_offset, dim, 1, false);
            mark_face(L(), dim, 1, false);
        } else if (fd == FD_MIXED) {
            convert_to_non_uniform_if_uniform(obj, lower_idx);
            update_outward_with_non_uniform(obj, L().data_offset, upper.data_offset, dim, 1);
            mark_face(L(), dim, 1, false);
        }
        return;
    }
    // both non-uniform
    uint8_t lf = lower.face[dim][1], uf = upper.face[dim][0];
    if (lf != FD_EMPTY) {
        if (uf == FD_EMPTY) set_all_outward(obj, lower.data_offset, dim, 1, false);
        else if (uf == FD_FULL) set_all_outward(obj, lower.data_offset, dim, 1, true);
        else update_outward_with_non_uniform(obj, lower.data_offset, upper.data_offset, dim, 1);
    }
    if (uf != FD_EMPTY) {
        if (lf == FD_EMPTY) set_all_outward(obj, upper.data_offset, dim, 0, false);
        else if (lf == FD_FULL) set_all_outward(obj, upper.data_offset, dim, 0, true);
        else update_outward_with_non_uniform(obj, upper.data_offset, lower.data_offset, dim, 0);
    }
    mark_face(L(), dim, 1, uf == FD_FULL);
    mark_face(U(), dim, 0, lf == FD_FULL);
}

// object.rs:1136-1145
void compute_all_derived_state(VoxelObject& obj) {
    int n = obj.n_chunks();
    for (int ci = 0; ci < n; ++ci)
        if (obj.chunks[ci].kind == K_NONUNIFORM) update_internal_adjacencies(&obj.voxels[(size_t)obj.chunks[ci].data_offset << 12]);
    for (int ci = 0; ci < n; ++ci)
        if (obj.chunks[ci].kind == K_NONUNIFORM) update_local_connected_regions_for_chunk(obj, ci);
    // object.rs:1673-1709
    for (int i = 0; i < obj.cc[0]; ++i)
        for (int j = 0; j < obj.cc[1]; ++j)
            for (int k = 0; k < obj.cc[2]; ++k) {
                int ci = obj.cidx(i, j, k);
                int adj[3][3] = {{i + 1, j, k}, {i, j + 1, k}, {i, j, k + 1}};
                for (int d = 0; d < 3; ++d) {
                    int up = adj[d][d] < obj.cc[d] ? obj.cidx(adj[d][0], adj[d][1], adj[d][2]) : -1;
                    update_mutual_face_adjacencies(obj, ci, up, d);
                }
            }
    // object.rs:1745-1785
    for (int j = 0; j < obj.cc[1]; ++j)
        for (int k = 0; k < obj.cc[2]; ++k) update_mutual_face_adjacencies(obj, -1, obj.cidx(0, j, k), 0);
    for (int i = 0; i < obj.cc[0]; ++i)
        for (int k = 0; k < obj.cc[2]; ++k) update_mutual_face_adjacencies(obj, -1, obj.cidx(i, 0, k), 1);
    for (int i = 0; i < obj.cc[0]; ++i)
        for (int j = 0; j < obj.cc[1]; ++j) update_mutual_face_adjacencies(obj, -1, obj.cidx(i, j, 0), 2);
}

}  // namespace orc
