// ORACLE (test infrastructure only — never linked into the product library).
//
// CPU restatement of splitting a disconnected region off a voxel object
// (paths relative to /root/reference/engine/crates/impact_voxel/src):
//   find_two_disconnected_regions                          object/split_detection.rs:193-248
//   extract_smallest_region_with_property_transferrer      object/extraction.rs:121-295
//   extract_disconnected_region (voxel / chunk moves)      object/extraction.rs:297-596
//   complete_extracted_voxel_object (discard < 8 voxels)   object/extraction.rs:1901-1970
//   single-chunk repack of <= 2x2x2-chunk objects          object/extraction.rs:1972-2123
//   update_all_internal_state_and_determine_sparseness     object.rs:2758-2874
// The reference updates the derived state of both objects incrementally; its own validators show that
// this equals the from-scratch state (fuzz targets `split_off_*`), so after moving the voxels exactly as
// the reference does (including which chunks change kind) the derived state is recomputed with
// compute_all_derived_state. Adjacency bits of EMPTY voxels are history artefacts in the reference
// (never rewritten on faces whose distribution is Empty) and are not part of the comparison.
#include <algorithm>
#include <climits>
#include <cmath>
#include <cstring>
#include <vector>

#include "../include/oracle.h"
#include "orc_voxel.hpp"

namespace orc {
uint32_t canonical_region_labels(const VoxelObject& obj, uint32_t* labels);

static inline int lin(int i, int j, int k) { return (i << 8) | (j << 4) | k; }

// object.rs:2758-2874 (face distributions, internal adjacencies, HAS_ONLY_EMPTY_VOXELS)
static void update_all_internal_state(Chunk& c, Voxel* cv) {
    int empty_counts[3][2] = {{0, 0}, {0, 0}, {0, 0}};
    bool only_empty = true;
    const uint8_t up[3] = {F_X_UP, F_Y_UP, F_Z_UP}, dn[3] = {F_X_DN, F_Y_DN, F_Z_DN};
    for (int i = 0; i < CHUNK; ++i)
        for (int j = 0; j < CHUNK; ++j)
            for (int k = 0; k < CHUNK; ++k) {
                const int idx = lin(i, j, k);
                const Voxel voxel = cv[idx];
                const int adj[3][3] = {{i + 1, j, k}, {i, j + 1, k}, {i, j, k + 1}};
                if (voxel.empty()) {
                    if (i == 0) empty_counts[0][0]++;
                    else if (i == CHUNK - 1) empty_counts[0][1]++;
                    if (j == 0) empty_counts[1][0]++;
                    else if (j == CHUNK - 1) empty_counts[1][1]++;
                    if (k == 0) empty_counts[2][0]++;
                    else if (k == CHUNK - 1) empty_counts[2][1]++;
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
                    only_empty = false;
                }
            }
    for (int d = 0; d < 3; ++d)
        for (int s = 0; s < 2; ++s) c.face[d][s] = empty_counts[d][s] == 256 ? FD_EMPTY : (empty_counts[d][s] == 0 ? FD_FULL : FD_MIXED);
    if (only_empty) c.flags |= CF_ONLY_EMPTY;
    else c.flags &= (uint8_t)~CF_ONLY_EMPTY;
}

static void reset_occupied_chunk_ranges(VoxelObject& obj) {  // object.rs:1149-1185
    int lo[3] = {INT_MAX, INT_MAX, INT_MAX}, hi[3] = {0, 0, 0};
    bool any = false;
    for (int i = 0; i < obj.cc[0]; ++i)
        for (int j = 0; j < obj.cc[1]; ++j)
            for (int k = 0; k < obj.cc[2]; ++k) {
                const Chunk& c = obj.chunks[obj.cidx(i, j, k)];
                bool only_empty = c.kind == K_VOID || (c.kind == K_NONUNIFORM && (c.flags & CF_ONLY_EMPTY));
                if (only_empty) continue;
                int idx[3] = {i, j, k};
                for (int d = 0; d < 3; ++d) {
                    lo[d] = std::min(lo[d], idx[d]);
                    hi[d] = std::max(hi[d], idx[d] + 1);
                }
                any = true;
            }
    for (int d = 0; d < 3; ++d) {
        obj.occ_chunk[d][0] = any ? lo[d] : 0;
        obj.occ_chunk[d][1] = any ? hi[d] : 0;
        obj.occ_voxel[d][0] = obj.occ_chunk[d][0] * CHUNK;
        obj.occ_voxel[d][1] = obj.occ_chunk[d][1] * CHUNK;
    }
    update_occupied_voxel_ranges(obj);
}

// (shared with orc_edit.cpp)
void edit_update_all_internal_state(Chunk& c, Voxel* cv) { update_all_internal_state(c, cv); }
void edit_reset_occupied_chunk_ranges(VoxelObject& obj) { reset_occupied_chunk_ranges(obj); }

// outcome: 0 nothing to split, 1 extracted into `child` (origin offset in voxels), 2 region removed but
// discarded (fewer than NON_EMPTY_VOXEL_THRESHOLD = 8 voxels, object.rs:203)
int split_off_smallest_region(VoxelObject& parent, VoxelObject& child, int origin[3]) {
    const int nx = parent.cc[0] * CHUNK, ny = parent.cc[1] * CHUNK, nz = parent.cc[2] * CHUNK;
    std::vector<uint32_t> comp((size_t)nx * ny * nz);
    const uint32_t n_comp = canonical_region_labels(parent, comp.data());
    if (n_comp < 2) return 0;
    auto comp_at = [&](int i, int j, int k) { return comp[((size_t)i * ny + j) * nz + k]; };
    const int n_chunks = parent.n_chunks();
    // component of every (chunk, local region); uniform chunks have the single region 0
    std::vector<std::vector<uint32_t>> region_comp(n_chunks);
    for (int ci = 0; ci < n_chunks; ++ci) {
        const Chunk& c = parent.chunks[ci];
        const int I = ci / (parent.cc[2] * parent.cc[1]), J = (ci / parent.cc[2]) % parent.cc[1], K = ci % parent.cc[2];
        if (c.kind == K_UNIFORM) region_comp[ci] = {comp_at(I * CHUNK, J * CHUNK, K * CHUNK)};
        else if (c.kind == K_NONUNIFORM) {
            region_comp[ci].assign(c.region_count, 0xFFFFFFFFu);
            const uint8_t* lab = &parent.labels[(size_t)c.data_offset << 12];
            for (int idx = 0; idx < CHUNK_VOXELS; ++idx)
                if (lab[idx] != 255 && lab[idx] < c.region_count && region_comp[ci][lab[idx]] == 0xFFFFFFFFu)
                    region_comp[ci][lab[idx]] = comp_at(I * CHUNK + (idx >> 8), J * CHUNK + ((idx >> 4) & 15), K * CHUNK + (idx & 15));
        }
    }
    // find_two_disconnected_regions: the first two distinct components in (chunk, region) scan order
    uint32_t two[2] = {0xFFFFFFFFu, 0xFFFFFFFFu};
    for (int ci = 0; ci < n_chunks && two[1] == 0xFFFFFFFFu; ++ci)
        for (uint32_t r : region_comp[ci]) {
            if (r == 0xFFFFFFFFu) continue;
            if (two[0] == 0xFFFFFFFFu) two[0] = r;
            else if (r != two[0]) {
                two[1] = r;
                break;
            }
        }
    if (two[1] == 0xFFFFFFFFu) return 0;
    // extraction.rs:121-271: chunk lists, counts and chunk boxes of both regions; smallest wins
    std::vector<int> chunks_of[2];
    int nu_count[2] = {0, 0};
    int lo[2][3] = {{INT_MAX, INT_MAX, INT_MAX}, {INT_MAX, INT_MAX, INT_MAX}}, hi[2][3] = {{0, 0, 0}, {0, 0, 0}};
    for (int ci = 0; ci < n_chunks; ++ci) {
        const Chunk& c = parent.chunks[ci];
        const int idx3[3] = {ci / (parent.cc[2] * parent.cc[1]), (ci / parent.cc[2]) % parent.cc[1], ci % parent.cc[2]};
        for (int q = 0; q < 2; ++q) {
            bool found = false;
            for (uint32_t r : region_comp[ci]) found = found || r == two[q];
            if (!found) continue;
            chunks_of[q].push_back(ci);
            if (c.kind == K_NONUNIFORM) nu_count[q] += 1;
            for (int d = 0; d < 3; ++d) {
                lo[q][d] = std::min(lo[q][d], idx3[d]);
                hi[q][d] = std::max(hi[q][d], idx3[d]);
            }
        }
    }
    int pick;
    if (nu_count[0] < nu_count[1]) pick = 0;
    else if (nu_count[0] > nu_count[1]) pick = 1;
    else pick = chunks_of[0].size() < chunks_of[1].size() ? 0 : 1;
    const uint32_t target = two[pick];
    const std::vector<int>& region_chunks = chunks_of[pick];
    int range_lo[3], range_hi[3], ccounts[3];
    for (int d = 0; d < 3; ++d) {
        range_lo[d] = lo[pick][d];
        range_hi[d] = hi[pick][d] + 1;
        ccounts[d] = range_hi[d] - range_lo[d];
    }
    // extract_disconnected_region (extraction.rs:297-546)
    child = VoxelObject{};
    child.extent = parent.extent;
    for (int d = 0; d < 3; ++d) child.cc[d] = ccounts[d];
    size_t cursor = 0;
    int uniform_count = 0, nonuniform_count = 0;
    for (int I = range_lo[0]; I < range_hi[0]; ++I)
        for (int J = range_lo[1]; J < range_hi[1]; ++J)
            for (int K = range_lo[2]; K < range_hi[2]; ++K) {
                const int ci = parent.cidx(I, J, K);
                Chunk rc;  // void by default
                if (cursor < region_chunks.size() && region_chunks[cursor] == ci) {
                    Chunk& pc = parent.chunks[ci];
                    if (pc.kind == K_NONUNIFORM) {
                        bool mixed = false;
                        for (uint32_t r : region_comp[ci]) mixed = mixed || r != target;
                        Voxel* pv = &parent.voxels[(size_t)pc.data_offset << 12];
                        const uint8_t* lab = &parent.labels[(size_t)pc.data_offset << 12];
                        const size_t start = child.voxels.size();
                        rc.kind = rc.gen_kind = K_NONUNIFORM;
                        rc.data_offset = (uint32_t)(start >> 12);
                        if (mixed) {
                            child.voxels.resize(start + CHUNK_VOXELS);
                            for (int idx = 0; idx < CHUNK_VOXELS; ++idx) {
                                Voxel& v = pv[idx];
                                Voxel out;
                                if (v.empty()) out = v;
                                else if (region_comp[ci][lab[idx]] == target) {
                                    out = v;
                                    v = voxel_max_outside();
                                } else out = voxel_max_outside();
                                child.voxels[start + idx] = out;
                            }
                            update_all_internal_state(rc, &child.voxels[start]);
                            pc.flags = 0;
                            update_all_internal_state(pc, pv);
                        } else {
                            child.voxels.insert(child.voxels.end(), pv, pv + CHUNK_VOXELS);
                            for (int idx = 0; idx < CHUNK_VOXELS; ++idx) pv[idx] = voxel_max_outside();
                            for (int d = 0; d < 3; ++d)
                                for (int s = 0; s < 2; ++s) rc.face[d][s] = pc.face[d][s];
                            pc = Chunk{};  // Void
                        }
                        nonuniform_count += 1;
                    } else {  // Uniform
                        rc = pc;
                        rc.gen_kind = K_UNIFORM;
                        parent.chunks[ci] = Chunk{};
                        uniform_count += 1;
                    }
                    cursor += 1;
                }
                child.chunks.push_back(rc);
            }
    child.labels.assign(child.voxels.size(), 0);
    // the parent after the removal: ranges, boundary adjacencies, regions — recomputed from scratch
    for (Chunk& c : parent.chunks)
        if (c.kind == K_NONUNIFORM) c.flags &= CF_ONLY_EMPTY;  // obscuredness is rebuilt below
    compute_all_derived_state(parent);
    reset_occupied_chunk_ranges(parent);
    // complete_extracted_voxel_object (extraction.rs:1901-1970)
    if (uniform_count == 0) {
        int non_empty = 0;
        for (const Voxel& v : child.voxels) non_empty += v.empty() ? 0 : 1;
        if (non_empty < 8) return 2;
    }
    for (int d = 0; d < 3; ++d) origin[d] = range_lo[d] * CHUNK;
    // single-chunk repack (extraction.rs:1972-2123)
    if (ccounts[0] <= 2 && ccounts[1] <= 2 && ccounts[2] <= 2 && uniform_count == 0 && ccounts[0] * ccounts[1] * ccounts[2] > 1) {
        int olo[3] = {INT_MAX, INT_MAX, INT_MAX}, ohi[3] = {0, 0, 0};
        for (int I = 0; I < ccounts[0]; ++I)
            for (int J = 0; J < ccounts[1]; ++J)
                for (int K = 0; K < ccounts[2]; ++K) {
                    const Chunk& c = child.chunks[child.cidx(I, J, K)];
                    if (c.kind != K_NONUNIFORM) continue;
                    const Voxel* v = &child.voxels[(size_t)c.data_offset << 12];
                    for (int idx = 0; idx < CHUNK_VOXELS; ++idx)
                        if (!v[idx].empty()) {
                            const int p[3] = {I * CHUNK + (idx >> 8), J * CHUNK + ((idx >> 4) & 15), K * CHUNK + (idx & 15)};
                            for (int d = 0; d < 3; ++d) {
                                olo[d] = std::min(olo[d], p[d]);
                                ohi[d] = std::max(ohi[d], p[d] + 1);
                            }
                        }
                }
        if (ohi[0] - olo[0] <= CHUNK - 2 && ohi[1] - olo[1] <= CHUNK - 2 && ohi[2] - olo[2] <= CHUNK - 2) {
            int off[3];
            for (int d = 0; d < 3; ++d) off[d] = olo[d] > 0 ? olo[d] - 1 : 0;
            VoxelObject single;
            single.extent = child.extent;
            single.cc[0] = single.cc[1] = single.cc[2] = 1;
            single.voxels.assign(CHUNK_VOXELS, voxel_max_outside());
            for (int i = 0; i < CHUNK; ++i)
                for (int j = 0; j < CHUNK; ++j)
                    for (int k = 0; k < CHUNK; ++k) {
                        const int s[3] = {off[0] + i, off[1] + j, off[2] + k};
                        if (s[0] >= ccounts[0] * CHUNK || s[1] >= ccounts[1] * CHUNK || s[2] >= ccounts[2] * CHUNK) continue;
                        const Chunk& c = child.chunks[child.cidx(s[0] >> 4, s[1] >> 4, s[2] >> 4)];
                        if (c.kind != K_NONUNIFORM) continue;
                        single.voxels[lin(i, j, k)] = child.voxels[((size_t)c.data_offset << 12) + lin(s[0] & 15, s[1] & 15, s[2] & 15)];
                    }
            Chunk sc;
            sc.kind = sc.gen_kind = K_NONUNIFORM;
            sc.data_offset = 0;
            update_all_internal_state(sc, single.voxels.data());
            single.chunks.push_back(sc);
            single.labels.assign(CHUNK_VOXELS, 0);
            for (int d = 0; d < 3; ++d) origin[d] += off[d];
            child = single;
        }
    }
    compute_all_derived_state(child);
    reset_occupied_chunk_ranges(child);
    return 1;
}

}  // namespace orc

// ---------------------------------------------------------------------------------------------
// Polyhedron clip: extract_polyhedron_with_property_transferrer (object/extraction.rs:639-1270) and
// copy_polyhedron_with_property_computer (1301-1768). Planes are (unit normal, displacement) in normalized
// model space (voxel units, grid corner at the origin); aabb = lower xyz, upper xyz.
namespace orc {

static inline float plane_sd(const float* p, float x, float y, float z) { return ((p[0] * x + p[1] * y) + p[2] * z) - p[3]; }
static inline bool sign_bit(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    return (u >> 31) != 0;
}
// VoxelSignedDistance::from_f32_array (lib.rs:207-216)
static inline int8_t sd_from_f32_clamped(float v) {
    float s = v * 50.0f;
    s = s < -128.0f ? -128.0f : (s > 127.0f ? 127.0f : s);  // f32::clamp (NaN propagates; not reachable here)
    return (int8_t)(int)s;
}
static inline int8_t sd_complement(int8_t e) {  // lib.rs:266-268: saturating_add(1).saturating_neg()
    int a = e == 127 ? 127 : e + 1;
    int n = a == -128 ? 127 : -a;
    return (int8_t)n;
}

// mode 0 = extract (parent loses the polyhedron), 1 = copy. outcome as split_off_smallest_region.
int clip_polyhedron(VoxelObject& parent, const float* planes, int n_planes, const float aabb[6], int mode, VoxelObject& child, int origin[3]) {
    const float EXT = SD_MAX_F32, INT = -SD_MIN_F32;
    int vlo[3], vhi[3], clo[3], chi[3], ccounts[3];
    for (int d = 0; d < 3; ++d) {
        const float lo = aabb[d] - EXT, hi = aabb[3 + d] + EXT;
        const float fl = std::floor(lo);
        const long s = (long)(fl > 0.0f ? fl : 0.0f), e = (long)std::ceil(hi);
        vlo[d] = std::max<long>(parent.occ_voxel[d][0], s);
        vhi[d] = std::min<long>(parent.occ_voxel[d][1], std::max<long>(e, 0));
        if (vlo[d] >= vhi[d]) return 0;
        clo[d] = vlo[d] / CHUNK;
        chi[d] = (vhi[d] + CHUNK - 1) / CHUNK;
        ccounts[d] = chi[d] - clo[d];
    }
    child = VoxelObject{};
    child.extent = parent.extent;
    for (int d = 0; d < 3; ++d) child.cc[d] = ccounts[d];
    int uniform_count = 0;
    std::vector<int> isect(n_planes);
    for (int I = clo[0]; I < chi[0]; ++I)
        for (int J = clo[1]; J < chi[1]; ++J)
            for (int K = clo[2]; K < chi[2]; ++K) {
                const int ci = parent.cidx(I, J, K);
                Chunk rc;  // void
                Chunk& pc = parent.chunks[ci];
                if (pc.kind == K_VOID) {
                    child.chunks.push_back(rc);
                    continue;
                }
                const float blo[3] = {(float)(I * CHUNK), (float)(J * CHUNK), (float)(K * CHUNK)};
                const float bhi[3] = {(float)((I + 1) * CHUNK), (float)((J + 1) * CHUNK), (float)((K + 1) * CHUNK)};
                bool outside = false;
                int n_isect = 0;
                for (int p = 0; p < n_planes; ++p) {
                    const float* pl = planes + 4 * p;
                    // minimum / maximum corner along the normal (axis_aligned_box.rs:494-507): a negative
                    // component picks the upper coordinate for the minimum corner
                    float mn[3], mx[3];
                    for (int d = 0; d < 3; ++d) {
                        const bool neg = sign_bit(pl[d]);
                        mn[d] = neg ? bhi[d] : blo[d];
                        mx[d] = neg ? blo[d] : bhi[d];
                    }
                    const float outer[4] = {pl[0], pl[1], pl[2], pl[3] + EXT}, inner[4] = {pl[0], pl[1], pl[2], pl[3] - INT};
                    if (plane_sd(outer, mn[0], mn[1], mn[2]) > 0.0f) outside = true;
                    if (!(plane_sd(inner, mx[0], mx[1], mx[2]) < 0.0f)) isect[n_isect++] = p;
                }
                if (outside) {
                    child.chunks.push_back(rc);
                    continue;
                }
                if (n_isect == 0) {  // fully inside
                    if (pc.kind == K_NONUNIFORM) {
                        Voxel* pv = &parent.voxels[(size_t)pc.data_offset << 12];
                        rc.kind = rc.gen_kind = K_NONUNIFORM;
                        rc.data_offset = (uint32_t)(child.voxels.size() >> 12);
                        rc.flags = pc.flags & CF_ONLY_EMPTY;
                        for (int d = 0; d < 3; ++d)
                            for (int s = 0; s < 2; ++s) rc.face[d][s] = pc.face[d][s];
                        child.voxels.insert(child.voxels.end(), pv, pv + CHUNK_VOXELS);
                        if (mode == 0) {
                            for (int idx = 0; idx < CHUNK_VOXELS; ++idx) pv[idx] = voxel_max_outside();
                            pc = Chunk{};
                        }
                    } else {
                        rc = pc;
                        rc.gen_kind = K_UNIFORM;
                        uniform_count += 1;
                        if (mode == 0) pc = Chunk{};
                    }
                    child.chunks.push_back(rc);
                    continue;
                }
                // intersecting chunk
                std::vector<Voxel> tmp(CHUNK_VOXELS);
                Voxel* pv = nullptr;
                if (pc.kind == K_UNIFORM) {
                    if (mode == 0) {  // convert_to_non_uniform_if_uniform (object.rs:2530-2550)
                        const size_t start = parent.voxels.size();
                        parent.voxels.resize(start + CHUNK_VOXELS, pc.uniform_voxel);
                        parent.labels.resize(start + CHUNK_VOXELS, 0);
                        pc.kind = K_NONUNIFORM;
                        pc.data_offset = (uint32_t)(start >> 12);
                        pv = &parent.voxels[start];
                        std::copy(pv, pv + CHUNK_VOXELS, tmp.begin());
                    } else {
                        std::fill(tmp.begin(), tmp.end(), pc.uniform_voxel);
                    }
                } else {
                    pv = &parent.voxels[(size_t)pc.data_offset << 12];
                    std::copy(pv, pv + CHUNK_VOXELS, tmp.begin());
                    if (mode == 1) pv = nullptr;
                }
                const float lvp[3] = {blo[0] + 0.5f, blo[1] + 0.5f, blo[2] + 0.5f};
                bool parent_only_empty = true, poly_only_empty = true, parent_all_void = true, poly_all_void = true;
                for (int i = 0; i < CHUNK; ++i)
                    for (int j = 0; j < CHUNK; ++j) {
                        const float rx = lvp[0] + (float)i, ry = lvp[1] + (float)j, rz = lvp[2] + 0.0f;
                        float md[CHUNK];
                        for (int q = 0; q < n_isect; ++q) {
                            const float* pl = planes + 4 * isect[q];
                            const float base = plane_sd(pl, rx, ry, rz), step = pl[2];
                            for (int k = 0; k < CHUNK; ++k) {
                                const float v = base + step * (float)k;
                                md[k] = q == 0 ? v : (v > md[k] ? v : (md[k] != md[k] ? v : md[k]));  // f32::max
                            }
                        }
                        for (int k = 0; k < CHUNK; ++k) {
                            const int idx = lin(i, j, k);
                            const int8_t d = sd_from_f32_clamped(md[k]);
                            Voxel& poly = tmp[idx];
                            const int8_t orig = poly.sd;
                            poly.sd = std::max(orig, d);
                            if (poly.sd < 0) {
                                poly.flags &= (uint8_t)~F_EMPTY;
                                poly_only_empty = false;
                            } else {
                                poly.flags |= F_EMPTY;
                            }
                            if (!sd_is_void(poly.sd)) poly_all_void = false;
                            if (pv) {
                                Voxel& par = pv[idx];
                                par.sd = std::max(orig, sd_complement(d));
                                if (poly.sd < 0) par.flags |= F_EMPTY;
                                if (par.sd < 0) parent_only_empty = false;
                                if (!sd_is_void(par.sd)) parent_all_void = false;
                            }
                        }
                    }
                if (pv) {
                    if (parent_only_empty && parent_all_void) {
                        for (int idx = 0; idx < CHUNK_VOXELS; ++idx) pv[idx] = voxel_max_outside();
                        pc = Chunk{};
                    } else {
                        pc.flags = 0;
                        update_all_internal_state(pc, pv);
                    }
                }
                if (!(poly_only_empty && poly_all_void)) {
                    rc.kind = rc.gen_kind = K_NONUNIFORM;
                    rc.data_offset = (uint32_t)(child.voxels.size() >> 12);
                    child.voxels.insert(child.voxels.end(), tmp.begin(), tmp.end());
                    update_all_internal_state(rc, &child.voxels[(size_t)rc.data_offset << 12]);
                }
                child.chunks.push_back(rc);
            }
    child.labels.assign(child.voxels.size(), 0);
    if (mode == 0) {
        for (Chunk& c : parent.chunks)
            if (c.kind == K_NONUNIFORM) c.flags &= CF_ONLY_EMPTY;
        compute_all_derived_state(parent);
        reset_occupied_chunk_ranges(parent);
    }
    if (uniform_count == 0) {
        int non_empty = 0;
        for (const Voxel& v : child.voxels) non_empty += v.empty() ? 0 : 1;
        if (non_empty < 8) return 2;
    }
    for (int d = 0; d < 3; ++d) origin[d] = clo[d] * CHUNK;
    if (ccounts[0] <= 2 && ccounts[1] <= 2 && ccounts[2] <= 2 && uniform_count == 0 && ccounts[0] * ccounts[1] * ccounts[2] > 1) {
        int olo[3] = {INT_MAX, INT_MAX, INT_MAX}, ohi[3] = {0, 0, 0};
        for (int I = 0; I < ccounts[0]; ++I)
            for (int J = 0; J < ccounts[1]; ++J)
                for (int K = 0; K < ccounts[2]; ++K) {
                    const Chunk& c = child.chunks[child.cidx(I, J, K)];
                    if (c.kind != K_NONUNIFORM) continue;
                    const Voxel* v = &child.voxels[(size_t)c.data_offset << 12];
                    for (int idx = 0; idx < CHUNK_VOXELS; ++idx)
                        if (!v[idx].empty()) {
                            const int p[3] = {I * CHUNK + (idx >> 8), J * CHUNK + ((idx >> 4) & 15), K * CHUNK + (idx & 15)};
                            for (int d = 0; d < 3; ++d) {
                                olo[d] = std::min(olo[d], p[d]);
                                ohi[d] = std::max(ohi[d], p[d] + 1);
                            }
                        }
                }
        if (ohi[0] - olo[0] <= CHUNK - 2 && ohi[1] - olo[1] <= CHUNK - 2 && ohi[2] - olo[2] <= CHUNK - 2) {
            int off[3];
            for (int d = 0; d < 3; ++d) off[d] = olo[d] > 0 ? olo[d] - 1 : 0;
            VoxelObject single;
            single.extent = child.extent;
            single.cc[0] = single.cc[1] = single.cc[2] = 1;
            single.voxels.assign(CHUNK_VOXELS, voxel_max_outside());
            for (int i = 0; i < CHUNK; ++i)
                for (int j = 0; j < CHUNK; ++j)
                    for (int k = 0; k < CHUNK; ++k) {
                        const int s[3] = {off[0] + i, off[1] + j, off[2] + k};
                        if (s[0] >= ccounts[0] * CHUNK || s[1] >= ccounts[1] * CHUNK || s[2] >= ccounts[2] * CHUNK) continue;
                        const Chunk& c = child.chunks[child.cidx(s[0] >> 4, s[1] >> 4, s[2] >> 4)];
                        if (c.kind != K_NONUNIFORM) continue;
                        single.voxels[lin(i, j, k)] = child.voxels[((size_t)c.data_offset << 12) + lin(s[0] & 15, s[1] & 15, s[2] & 15)];
                    }
            Chunk sc;
            sc.kind = sc.gen_kind = K_NONUNIFORM;
            sc.data_offset = 0;
            update_all_internal_state(sc, single.voxels.data());
            single.chunks.push_back(sc);
            single.labels.assign(CHUNK_VOXELS, 0);
            for (int d = 0; d < 3; ++d) origin[d] += off[d];
            child = single;
        }
    }
    for (Chunk& c : child.chunks)
        if (c.kind == K_NONUNIFORM) c.flags &= CF_ONLY_EMPTY;
    compute_all_derived_state(child);
    reset_occupied_chunk_ranges(child);
    return 1;
}

}  // namespace orc
