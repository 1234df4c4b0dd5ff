// ORACLE (test infrastructure only — never linked into the product library).
//
// CPU restatement of the per-frame voxel edit that feeds the remesh: an absorbing sphere eating into a voxel object
// (paths relative to /root/reference/engine/crates/impact_voxel/src):
//   apply_sphere_absorption (influence sphere, closure)      interaction/absorption.rs:801-844
//   VoxelAbsorbingSphere::compute_new_signed_distance        interaction/absorption.rs:170-180
//   hard_sdf_subtraction                                      generation/sdf.rs:79-81
//   Voxel::set_signed_distance                                lib.rs:451-461
//   modify_voxels_within_sphere                               object/intersection.rs:273-395
//   voxel_ranges_touching_aab                                 object/intersection.rs:766-782
//   handle_chunk_voxels_modified (mesh invalidation rule)     object/intersection.rs:532-598
//   VoxelObjectInertialPropertyUpdater::remove_voxel          object/inertia.rs:377-394
// The reference then patches adjacencies around the touched chunks incrementally; as in orc_split.cpp the derived state is
// recomputed from scratch after the voxels and chunk kinds have been changed exactly as the reference changes them (its own
// validators show the two agree). Occupied ranges are refreshed only when a chunk became void, as in the reference.
#include <algorithm>
#include <climits>
#include <cmath>
#include <cstring>
#include <vector>

#include "../include/oracle.h"
#include "orc_math.hpp"
#include "orc_voxel.hpp"

namespace orc {
void edit_update_all_internal_state(Chunk& c, Voxel* cv);  // orc_split.cpp
void edit_reset_occupied_chunk_ranges(VoxelObject& obj);   // orc_split.cpp

// returns the number of chunks that became void; removed64 = integer-form moments of the emptied voxels scaled like
// inertia_moments_f64 (mass, first moments, moments and products of inertia about the grid origin)
// mode 0: sphere (center, influence radius); mode 1: capsule (segment start = center, segment vector = seg, influence radius) —
// modify_voxels_within_capsule (object/intersection.rs:397-530): the segment is trimmed per chunk against the chunk box grown by the
// radius (Capsule::trim_segment_outside_aab, impact_geometry/src/capsule.rs:144-164; AxisAlignedBox::find_contained_subsegment,
// axis_aligned_box.rs:385-415), the voxel ranges of a chunk come from the trimmed capsule's box, and a voxel is inside when its
// squared distance to the UNtrimmed segment is <= r^2 (CapsulePointContainmentTester, capsule.rs:166-250).
static int absorb_shape(VoxelObject& obj, int mode, const float center[3], const float seg[3], float influence_radius, float shape_radius, const float* dens,
                        double removed64[10], uint32_t emptied_by_type[256], uint8_t* invalidated /* [n_chunks] or null */, uint32_t* touched_chunks) {
    for (int q = 0; q < 10; ++q) removed64[q] = 0.0;
    if (emptied_by_type) std::memset(emptied_by_type, 0, 256 * sizeof(uint32_t));
    if (invalidated) std::memset(invalidated, 0, (size_t)obj.n_chunks());
    *touched_chunks = 0;
    long vlo[3], vhi[3];
    int clo[3], chi[3];
    for (int d = 0; d < 3; ++d) {
        float lo = center[d] - influence_radius, hi = center[d] + influence_radius;  // Sphere::compute_aabb
        if (mode == 1) {  // Capsule::compute_aabb: the pair of end-sphere boxes (capsule.rs:132-137)
            const float end = center[d] + seg[d];
            lo = fmin_rs(lo, end - influence_radius);
            hi = fmax_rs(hi, end + influence_radius);
        }
        const float fl = std::floor(lo);
        const float ce = std::ceil(hi);
        const long s = (long)(fl > 0.0f ? fl : 0.0f), e = ce > 0.0f ? (long)ce : 0;  // `as usize` saturates at 0
        vlo[d] = std::max<long>(obj.occ_voxel[d][0], s);
        vhi[d] = std::min<long>(obj.occ_voxel[d][1], e);
        if (vlo[d] >= vhi[d]) return 0;
        clo[d] = (int)(vlo[d] / CHUNK);
        chi[d] = (int)((vhi[d] + CHUNK - 1) / CHUNK);
    }
    const float r2 = influence_radius * influence_radius;
    // CapsulePointContainmentTester
    const V3 seg_start{center[0], center[1], center[2]}, seg_vec{seg ? seg[0] : 0.0f, seg ? seg[1] : 0.0f, seg ? seg[2] : 0.0f};
    const float seg_len2 = dot(seg_vec, seg_vec);
    const V3 seg_over_len2 = seg_len2 > 1e-8f ? div_recip(seg_vec, seg_len2) : V3{0.0f, 0.0f, 0.0f};
    const double ext = (double)obj.extent;
    double s[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    int removed_chunks = 0;
    for (int I = clo[0]; I < chi[0]; ++I)
        for (int J = clo[1]; J < chi[1]; ++J)
            for (int K = clo[2]; K < chi[2]; ++K) {
                const int ci = obj.cidx(I, J, K);
                Chunk& ch = obj.chunks[ci];
                const long base[3] = {(long)I * CHUNK, (long)J * CHUNK, (long)K * CHUNK};
                long rlo[3], rhi[3];
                if (mode == 1) {
                    // trim_segment_outside_aab(normalized_chunk_aabb) -> voxel_ranges_touching_aab(chunk ranges, trimmed.compute_aabb())
                    float t_min = 0.0f, t_max = 1.0f;
                    bool none = false;
                    const float sv[3] = {seg_vec.x, seg_vec.y, seg_vec.z};
                    for (int d = 0; d < 3 && !none; ++d) {
                        const float blo = (float)base[d] - influence_radius, bhi = (float)(base[d] + CHUNK) + influence_radius;
                        if (std::fabs(sv[d]) > 1e-8f) {
                            const float recip = 1.0f / sv[d];
                            const float t1 = (blo - center[d]) * recip, t2 = (bhi - center[d]) * recip;
                            const float te = t1 < t2 ? t1 : t2, tx = t1 < t2 ? t2 : t1;
                            t_min = fmax_rs(t_min, te);
                            t_max = fmin_rs(t_max, tx);
                        } else if (center[d] < blo || center[d] > bhi) {
                            none = true;
                        }
                    }
                    if (none || !(t_min <= t_max)) continue;
                    bool empty = false;
                    for (int d = 0; d < 3; ++d) {
                        const float ts = center[d] + sv[d] * t_min;
                        const float tv = sv[d] * (t_max - t_min);
                        const float te = ts + tv;
                        const float lo = fmin_rs(ts - influence_radius, te - influence_radius), hi = fmax_rs(ts + influence_radius, te + influence_radius);
                        const float fl = std::floor(lo), ce = std::ceil(hi);
                        rlo[d] = std::max<long>(base[d], (long)(fl > 0.0f ? fl : 0.0f));
                        rhi[d] = std::min<long>(base[d] + CHUNK, ce > 0.0f ? (long)ce : 0);
                        if (rlo[d] >= rhi[d]) empty = true;
                    }
                    if (empty) continue;
                }
                if (ch.kind == K_VOID) continue;
                if (ch.kind == K_UNIFORM) {  // convert_to_non_uniform_if_uniform (object.rs:2530-2550)
                    const size_t start = obj.voxels.size();
                    obj.voxels.resize(start + CHUNK_VOXELS, ch.uniform_voxel);
                    obj.labels.resize(start + CHUNK_VOXELS, 0);
                    ch.kind = K_NONUNIFORM;
                    ch.data_offset = (uint32_t)(start >> 12);
                    for (int d = 0; d < 3; ++d) ch.face[d][0] = ch.face[d][1] = FD_FULL;
                    ch.flags = CF_FULLY_OBSCURED;
                    ch.region_count = 1;
                    ch.boundary_region_count = 1;
                }
                Voxel* cv = &obj.voxels[(size_t)ch.data_offset << 12];
                if (mode == 0)
                    for (int d = 0; d < 3; ++d) {
                        rlo[d] = std::max(base[d], vlo[d]);
                        rhi[d] = std::min(base[d] + CHUNK, vhi[d]);
                    }
                bool touched = false;
                for (long i = rlo[0]; i < rhi[0]; ++i)
                    for (long j = rlo[1]; j < rhi[1]; ++j)
                        for (long k = rlo[2]; k < rhi[2]; ++k) {
                            const V3 p{(float)i + 0.5f, (float)j + 0.5f, (float)k + 0.5f};
                            float d2;
                            if (mode == 0) {
                                const V3 dv{p.x - center[0], p.y - center[1], p.z - center[2]};
                                d2 = dot(dv, dv);
                                if (!(d2 < r2)) continue;
                            } else {
                                const V3 sp = p - seg_start;
                                float t = dot(sp, seg_over_len2);
                                t = t < 0.0f ? 0.0f : (t > 1.0f ? 1.0f : t);  // f32::clamp
                                const V3 closest = seg_start + seg_vec * t;
                                const V3 dv = p - closest;
                                d2 = dot(dv, dv);
                                if (!(d2 <= r2)) continue;
                            }
                            Voxel& v = cv[((i - base[0]) << 8) | ((j - base[1]) << 4) | (k - base[2])];
                            const bool was_empty = v.empty();
                            const float sphere_sd = std::sqrt(d2) - shape_radius;
                            const float nv = fmax_rs(sd_to_f32(v.sd), -sphere_sd);  // hard_sdf_subtraction
                            v.sd = sd_from_f32(nv);
                            if (!(v.sd < 0)) {
                                v.flags |= F_EMPTY;
                                if (!was_empty) {
                                    const double dd = (double)dens[v.type], X = (double)i, Y = (double)j, Z = (double)k;
                                    const double qx = 2 * X + 1, qy = 2 * Y + 1, qz = 2 * Z + 1;
                                    const double cx = 3 * X * X + 3 * X + 1, cy = 3 * Y * Y + 3 * Y + 1, cz = 3 * Z * Z + 3 * Z + 1;
                                    s[0] += dd;
                                    s[1] += dd * qx;
                                    s[2] += dd * qy;
                                    s[3] += dd * qz;
                                    s[4] += dd * (cy + cz);
                                    s[5] += dd * (cx + cz);
                                    s[6] += dd * (cx + cy);
                                    s[7] += dd * qx * qy;
                                    s[8] += dd * qy * qz;
                                    s[9] += dd * qx * qz;
                                    if (emptied_by_type) emptied_by_type[v.type] += 1;
                                }
                            }
                            touched = true;
                        }
                if (!touched) continue;
                *touched_chunks += 1;
                // handle_chunk_voxels_modified
                bool only_empty = true, all_void = true;
                for (int idx = 0; idx < CHUNK_VOXELS; ++idx) {
                    if (!cv[idx].empty()) only_empty = false;
                    else if (!sd_is_void(cv[idx].sd)) all_void = false;
                }
                if (only_empty && all_void) {
                    for (int idx = 0; idx < CHUNK_VOXELS; ++idx) cv[idx] = voxel_max_outside();
                    ch = Chunk{};
                    removed_chunks += 1;
                } else {
                    ch.flags = 0;
                    edit_update_all_internal_state(ch, cv);
                }
                if (invalidated) {
                    invalidated[ci] = 1;
                    const int cidx3[3] = {I, J, K};
                    for (int d = 0; d < 3; ++d) {
                        if (cidx3[d] > 0 && rlo[d] - base[d] < 2) {
                            int a[3] = {I, J, K};
                            a[d] -= 1;
                            invalidated[obj.cidx(a[0], a[1], a[2])] = 1;
                        }
                        if (cidx3[d] + 1 < obj.cc[d] && base[d] + CHUNK - rhi[d] < 2) {
                            int a[3] = {I, J, K};
                            a[d] += 1;
                            invalidated[obj.cidx(a[0], a[1], a[2])] = 1;
                        }
                    }
                }
            }
    const double e3 = ext * ext * ext, e4 = e3 * ext, e5 = e4 * ext;
    const double f[10] = {e3, 0.5 * e4, 0.5 * e4, 0.5 * e4, e5 / 3.0, e5 / 3.0, e5 / 3.0, 0.25 * e5, 0.25 * e5, 0.25 * e5};
    for (int q = 0; q < 10; ++q) removed64[q] = s[q] * f[q];
    for (Chunk& c : obj.chunks)
        if (c.kind == K_NONUNIFORM) c.flags &= CF_ONLY_EMPTY;
    compute_all_derived_state(obj);
    if (removed_chunks) edit_reset_occupied_chunk_ranges(obj);
    return removed_chunks;
}

int absorb_sphere(VoxelObject& obj, const float center[3], float influence_radius, float sphere_radius, const float* dens, double removed64[10],
                  uint32_t emptied_by_type[256], uint8_t* invalidated, uint32_t* touched_chunks) {
    return absorb_shape(obj, 0, center, nullptr, influence_radius, sphere_radius, dens, removed64, emptied_by_type, invalidated, touched_chunks);
}
// apply_capsule_absorption (interaction/absorption.rs:846-889) with the capsule in the object's normalized space
int absorb_capsule(VoxelObject& obj, const float segment_start[3], const float segment_vector[3], float influence_radius, float capsule_radius,
                   const float* dens, double removed64[10], uint32_t emptied_by_type[256], uint8_t* invalidated, uint32_t* touched_chunks) {
    return absorb_shape(obj, 1, segment_start, segment_vector, influence_radius, capsule_radius, dens, removed64, emptied_by_type, invalidated, touched_chunks);
}

// ---- mutual absorption --------------------------------------------------------------------------------------------------------------
//   apply_mutual_absorption + compute_subtracted_signed_distance   interaction/absorption.rs:891-1094
//   modify_voxels_within_ranges                                     object/intersection.rs:167-261
//   sample_voxel_object_sdf / evaluate_sdf_from_corner_samples      object/sdf.rs:579-597, 636-675
//   sdf_subtraction, smooth_sdf_union, Smoothness                   generation/sdf.rs:8-37, 52-102
bool intersection_voxel_ranges(const VoxelObject& A, const float rot_a[4], const float trans_a[3], const VoxelObject& B, const float rot_b[4],
                               const float trans_b[3], long ra_lo[3], long ra_hi[3], long rb_lo[3], long rb_hi[3], float q_ba_out[4], float t_ba_out[3]);
float sample_voxel_object_sdf(const VoxelObject& o, V3 p);
float smooth_union(float d1, float d2, float s, float q);  // orc_sdf.cpp

namespace {
struct EditStats {
    double s[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t emptied = 0;
    uint32_t touched_chunks = 0;
    int removed_chunks = 0;
};

// modify_voxels_within_ranges: Void chunks are skipped, Uniform chunks of the chunk range become NonUniform, `fn` sees every voxel of the
// ranges and says whether it counts as touched; touched chunks go through handle_chunk_voxels_modified
template <class F>
void modify_within_ranges(VoxelObject& obj, const long lo[3], const long hi[3], const float* dens, EditStats& st, uint8_t* invalidated, F&& fn) {
    for (int d = 0; d < 3; ++d)
        if (lo[d] >= hi[d]) return;
    for (long I = lo[0] / CHUNK; I < (hi[0] + CHUNK - 1) / CHUNK; ++I)
        for (long J = lo[1] / CHUNK; J < (hi[1] + CHUNK - 1) / CHUNK; ++J)
            for (long K = lo[2] / CHUNK; K < (hi[2] + CHUNK - 1) / CHUNK; ++K) {
                const int ci = obj.cidx((int)I, (int)J, (int)K);
                Chunk& ch = obj.chunks[ci];
                if (ch.kind == K_VOID) continue;
                if (ch.kind == K_UNIFORM) {  // convert_to_non_uniform_if_uniform (object.rs:2530-2550)
                    const size_t start = obj.voxels.size();
                    obj.voxels.resize(start + CHUNK_VOXELS, ch.uniform_voxel);
                    obj.labels.resize(start + CHUNK_VOXELS, 0);
                    ch.kind = K_NONUNIFORM;
                    ch.data_offset = (uint32_t)(start >> 12);
                    for (int d = 0; d < 3; ++d) ch.face[d][0] = ch.face[d][1] = FD_FULL;
                    ch.flags = CF_FULLY_OBSCURED;
                    ch.region_count = 1;
                    ch.boundary_region_count = 1;
                }
                Voxel* cv = &obj.voxels[(size_t)ch.data_offset << 12];
                const long base[3] = {I * CHUNK, J * CHUNK, K * CHUNK};
                long rlo[3], rhi[3];
                for (int d = 0; d < 3; ++d) {
                    rlo[d] = std::max(base[d], lo[d]);
                    rhi[d] = std::min(base[d] + CHUNK, hi[d]);
                }
                bool touched = false;
                for (long i = rlo[0]; i < rhi[0]; ++i)
                    for (long j = rlo[1]; j < rhi[1]; ++j)
                        for (long k = rlo[2]; k < rhi[2]; ++k) {
                            Voxel& v = cv[((i - base[0]) << 8) | ((j - base[1]) << 4) | (k - base[2])];
                            const bool was_empty = v.empty();
                            float nv;
                            if (!fn(i, j, k, v, nv)) continue;
                            touched = true;
                            v.sd = sd_from_f32(nv);  // Voxel::set_signed_distance
                            if (!(v.sd < 0)) {
                                v.flags |= F_EMPTY;
                                if (!was_empty) {
                                    const double dd = (double)dens[v.type], X = (double)i, Y = (double)j, Z = (double)k;
                                    const double qx = 2 * X + 1, qy = 2 * Y + 1, qz = 2 * Z + 1;
                                    const double cx = 3 * X * X + 3 * X + 1, cy = 3 * Y * Y + 3 * Y + 1, cz = 3 * Z * Z + 3 * Z + 1;
                                    st.s[0] += dd;
                                    st.s[1] += dd * qx;
                                    st.s[2] += dd * qy;
                                    st.s[3] += dd * qz;
                                    st.s[4] += dd * (cy + cz);
                                    st.s[5] += dd * (cx + cz);
                                    st.s[6] += dd * (cx + cy);
                                    st.s[7] += dd * qx * qy;
                                    st.s[8] += dd * qy * qz;
                                    st.s[9] += dd * qx * qz;
                                    st.emptied += 1;
                                }
                            }
                        }
                if (!touched) continue;
                st.touched_chunks += 1;
                bool only_empty = true, all_void = true;  // handle_chunk_voxels_modified
                for (int idx = 0; idx < CHUNK_VOXELS; ++idx) {
                    if (!cv[idx].empty()) only_empty = false;
                    else if (!sd_is_void(cv[idx].sd)) all_void = false;
                }
                if (only_empty && all_void) {
                    for (int idx = 0; idx < CHUNK_VOXELS; ++idx) cv[idx] = voxel_max_outside();
                    ch = Chunk{};
                    st.removed_chunks += 1;
                } else {
                    ch.flags = 0;
                    edit_update_all_internal_state(ch, cv);
                }
                if (invalidated) {
                    invalidated[ci] = 1;
                    const long cidx3[3] = {I, J, K};
                    for (int d = 0; d < 3; ++d) {
                        if (cidx3[d] > 0 && rlo[d] - base[d] < 2) {
                            long a[3] = {I, J, K};
                            a[d] -= 1;
                            invalidated[obj.cidx((int)a[0], (int)a[1], (int)a[2])] = 1;
                        }
                        if (cidx3[d] + 1 < obj.cc[d] && base[d] + CHUNK - rhi[d] < 2) {
                            long a[3] = {I, J, K};
                            a[d] += 1;
                            invalidated[obj.cidx((int)a[0], (int)a[1], (int)a[2])] = 1;
                        }
                    }
                }
            }
}

void finish_edit(VoxelObject& obj, const EditStats& st, double removed64[10]) {
    const double ext = (double)obj.extent, e3 = ext * ext * ext, e4 = e3 * ext, e5 = e4 * ext;
    const double f[10] = {e3, 0.5 * e4, 0.5 * e4, 0.5 * e4, e5 / 3.0, e5 / 3.0, e5 / 3.0, 0.25 * e5, 0.25 * e5, 0.25 * e5};
    for (int q = 0; q < 10; ++q) removed64[q] = st.s[q] * f[q];
    for (Chunk& c : obj.chunks)
        if (c.kind == K_NONUNIFORM) c.flags &= CF_ONLY_EMPTY;
    compute_all_derived_state(obj);
    if (st.removed_chunks) edit_reset_occupied_chunk_ranges(obj);
}

inline V3 qrot_e(Quat q, V3 v) {  // glam Quat::mul_vec3a
    V3 b{q.x, q.y, q.z};
    float b2 = dot(b, b);
    return (v * (q.w * q.w - b2) + b * (dot(v, b) * 2.0f)) + cross(b, v) * (q.w * 2.0f);
}
inline float subtracted(float sd, float inside_other, float smooth, float quarter_inv) {  // compute_subtracted_signed_distance
    const float inter = fmax_rs(sd, inside_other);
    return smooth == 0.0f ? fmax_rs(sd, -inter) : -smooth_union(-sd, inter, smooth, quarter_inv);
}
}  // namespace

// stats: [0] emptied voxels of A, [1] touched chunks of A, [2] removed chunks of A, [3..5] the same for B
void absorb_mutual(VoxelObject& A, const float rot_a[4], const float trans_a[3], const float* dens_a, VoxelObject& B, const float rot_b[4],
                   const float trans_b[3], const float* dens_b, float smoothness, double removed_a[10], double removed_b[10], uint8_t* invalidated_a,
                   uint8_t* invalidated_b, uint64_t stats[6]) {
    for (int q = 0; q < 10; ++q) removed_a[q] = removed_b[q] = 0.0;
    for (int q = 0; q < 6; ++q) stats[q] = 0;
    if (invalidated_a) std::memset(invalidated_a, 0, (size_t)A.n_chunks());
    if (invalidated_b) std::memset(invalidated_b, 0, (size_t)B.n_chunks());
    long ra_lo[3], ra_hi[3], rb_lo[3], rb_hi[3];
    float qv[4], tv[3];
    if (!intersection_voxel_ranges(A, rot_a, trans_a, B, rot_b, trans_b, ra_lo, ra_hi, rb_lo, rb_hi, qv, tv)) return;
    const Quat q_ba{qv[0], qv[1], qv[2], qv[3]};
    const V3 t_ba{tv[0], tv[1], tv[2]};
    const float inv_a = 1.0f / A.extent, inv_b = 1.0f / B.extent;
    const float b_dist_to_a = B.extent * inv_a, a_dist_to_b = A.extent * inv_b;
    const float quarter_inv = 0.25f / smoothness;
    const long pad = (long)std::ceil(b_dist_to_a);
    long s_lo[3], s_hi[3];
    size_t count = 1;
    for (int d = 0; d < 3; ++d) {
        s_lo[d] = std::max<long>(0, ra_lo[d] - pad);
        s_hi[d] = std::min<long>(ra_hi[d] + pad, (long)A.cc[d] * 16);
        count *= (size_t)std::max<long>(0, s_hi[d] - s_lo[d]);
    }
    std::vector<float> snapshot(count, sd_to_f32(127));
    auto snap = [&](long i, long j, long k) -> float& { return snapshot[(size_t)(((i - s_lo[0]) * (s_hi[1] - s_lo[1]) + (j - s_lo[1])) * (s_hi[2] - s_lo[2]) + (k - s_lo[2]))]; };
    EditStats sa, sb;
    modify_within_ranges(A, s_lo, s_hi, dens_a, sa, invalidated_a, [&](long i, long j, long k, Voxel& v, float& nv) {
        if (v.sd == 127) return false;
        const float sd = sd_to_f32(v.sd);
        snap(i, j, k) = sd;
        const V3 c_a{((float)i + 0.5f) * A.extent, ((float)j + 0.5f) * A.extent, ((float)k + 0.5f) * A.extent};
        const V3 c_b = inv_b * qrot_e(conj(q_ba), c_a - t_ba);  // inverse_transform_point
        const float inside_b = sample_voxel_object_sdf(B, c_b) * b_dist_to_a;
        nv = subtracted(sd, inside_b, smoothness, quarter_inv);
        return true;
    });
    modify_within_ranges(B, rb_lo, rb_hi, dens_b, sb, invalidated_b, [&](long i, long j, long k, Voxel& v, float& nv) {
        if (v.sd == 127) return false;
        const V3 c_b{((float)i + 0.5f) * B.extent, ((float)j + 0.5f) * B.extent, ((float)k + 0.5f) * B.extent};
        const V3 c_a = inv_a * (qrot_e(q_ba, c_b) + t_ba);
        const V3 lc = c_a - V3{0.5f, 0.5f, 0.5f};
        const V3 fl{std::floor(lc.x), std::floor(lc.y), std::floor(lc.z)};
        const V3 off = lc - fl;
        const float flv[3] = {fl.x, fl.y, fl.z};
        long l[3];
        for (int d = 0; d < 3; ++d) {  // `as isize` (saturating)
            l[d] = flv[d] >= 9.0e18f ? LONG_MAX - 1 : (flv[d] <= -9.0e18f ? LONG_MIN : (long)flv[d]);
            if (l[d] < s_lo[d] || l[d] + 1 >= s_hi[d]) return false;
        }
        float dd[8];
        for (int c = 0; c < 8; ++c) dd[c] = snap(l[0] + ((c >> 2) & 1), l[1] + ((c >> 1) & 1), l[2] + (c & 1));
        const V3 rev = V3{1.0f, 1.0f, 1.0f} - off;
        const float d00 = dd[0] * rev.x + dd[4] * off.x, d01 = dd[1] * rev.x + dd[5] * off.x, d10 = dd[2] * rev.x + dd[6] * off.x,
                    d11 = dd[3] * rev.x + dd[7] * off.x;
        const float d0 = d00 * rev.y + d10 * off.y, d1 = d01 * rev.y + d11 * off.y;
        const float inside_a = (d0 * rev.z + d1 * off.z) * a_dist_to_b;
        nv = subtracted(sd_to_f32(v.sd), inside_a, smoothness, quarter_inv);
        return true;
    });
    finish_edit(A, sa, removed_a);
    finish_edit(B, sb, removed_b);
    stats[0] = sa.emptied, stats[1] = sa.touched_chunks, stats[2] = (uint64_t)sa.removed_chunks;
    stats[3] = sb.emptied, stats[4] = sb.touched_chunks, stats[5] = (uint64_t)sb.removed_chunks;
}

}  // namespace orc
