// ORACLE (test infrastructure only — never linked into the product library).
//
// CPU restatement of contact generation between a sphere collidable and a voxel object, the producer of the contacts the
// constraint solver consumes (paths relative to /root/reference/engine/crates):
//   for_each_sphere_voxel_object_contact                    impact_voxel/src/collidable.rs:1098-1127
//   for_each_surface_voxel_maybe_intersecting_sphere        impact_voxel/src/object/intersection.rs:51-60
//   for_each_surface_voxel_in_voxel_ranges                  impact_voxel/src/object/intersection.rs:97-151
//   voxel_ranges_touching_aab                               impact_voxel/src/object/intersection.rs:766-782
//   VoxelFlags::placement (Surface = fewer than 6 neighbours) impact_voxel/src/lib.rs:330-342
//   compute_voxel_radius                                    impact_voxel/src/collidable.rs:1453-1455
//   Isometry3::transform_point / inverse_transform_point    impact_math/src/transform/isometry.rs:146-172
//   determine_sphere_sphere_contact_geometry                impact_physics/src/collision/collidable/sphere.rs:105-136
#include <algorithm>
#include <cmath>
#include <vector>

#include "../include/oracle.h"
#include "orc_math.hpp"
#include "orc_voxel.hpp"

extern "C" int orc_sphere_plane_contact(const float c[3], float r, const float plane_normal[3], float plane_displacement, float position[3], float normal[3],
                                        float* depth);
extern "C" int orc_sphere_sphere_contact(const float ca[3], float ra, const float cb[3], float rb, float position[3], float normal[3], float* depth);

namespace orc {

// glam Quat::mul_vec3a (as in orc_physics.cpp)
static inline V3 qrot(Quat q, V3 v) {
    V3 b{q.x, q.y, q.z};
    float b2 = dot(b, b);
    return (v * (q.w * q.w - b2) + b * (dot(v, b) * 2.0f)) + cross(b, v) * (q.w * 2.0f);
}

// indices [n][3], position/normal [n][3], depth [n]; returns the number of contacts (all of them, also beyond `cap`)
int sphere_voxel_object_contacts(const VoxelObject& obj, const float rot[4], const float trans[3], const float center[3], float radius, int cap,
                                 int32_t* indices, float* position, float* normal, float* depth) {
    const Quat q{rot[0], rot[1], rot[2], rot[3]};
    const V3 t{trans[0], trans[1], trans[2]}, c{center[0], center[1], center[2]};
    const V3 c_obj = qrot(q, c) + t;  // sphere.iso_transformed(transform_to_object_space)
    const float inv = 1.0f / obj.extent;
    const V3 cn = c_obj * inv;  // .scaled(inverse_voxel_extent)
    const float rn = inv * radius;
    long vlo[3], vhi[3];
    const float cnv[3] = {cn.x, cn.y, cn.z};
    for (int d = 0; d < 3; ++d) {
        const float lo = cnv[d] - rn, hi = cnv[d] + rn;
        const float fl = std::floor(lo), ce = std::ceil(hi);
        const long s = (long)(fl > 0.0f ? fl : 0.0f), e = ce > 0.0f ? (long)ce : 0;
        vlo[d] = std::max<long>(obj.occ_voxel[d][0], s);
        vhi[d] = std::min<long>(obj.occ_voxel[d][1], e);
        if (vlo[d] >= vhi[d]) return 0;
    }
    int n = 0;
    const Quat qi = conj(q);
    for (long I = vlo[0] / CHUNK; I < (vhi[0] + CHUNK - 1) / CHUNK; ++I)
        for (long J = vlo[1] / CHUNK; J < (vhi[1] + CHUNK - 1) / CHUNK; ++J)
            for (long K = vlo[2] / CHUNK; K < (vhi[2] + CHUNK - 1) / CHUNK; ++K) {
                const Chunk& ch = obj.chunks[obj.cidx((int)I, (int)J, (int)K)];
                if (ch.kind != K_NONUNIFORM) continue;  // only non-uniform chunks can have surface voxels
                const Voxel* cv = &obj.voxels[(size_t)ch.data_offset << 12];
                const long base[3] = {I * CHUNK, J * CHUNK, K * CHUNK};
                for (long i = std::max(base[0], vlo[0]); i < std::min(base[0] + CHUNK, vhi[0]); ++i)
                    for (long j = std::max(base[1], vlo[1]); j < std::min(base[1] + CHUNK, vhi[1]); ++j)
                        for (long k = std::max(base[2], vlo[2]); k < std::min(base[2] + CHUNK, vhi[2]); ++k) {
                            const Voxel& v = cv[((i - base[0]) << 8) | ((j - base[1]) << 4) | (k - base[2])];
                            if (v.empty()) continue;
                            if (__builtin_popcount(v.flags & 0xFCu) == 6) continue;  // Interior
                            const V3 p_obj{((float)i + 0.5f) * obj.extent, ((float)j + 0.5f) * obj.extent, ((float)k + 0.5f) * obj.extent};
                            const V3 p = qrot(qi, p_obj - t);  // inverse_transform_point
                            const float vr = -sd_to_f32(v.sd) * obj.extent;
                            const float pa[3] = {p.x, p.y, p.z};
                            float pos[3], nrm[3], dep;
                            if (!orc_sphere_sphere_contact(center, radius, pa, vr, pos, nrm, &dep)) continue;
                            if (n < cap) {
                                indices[3 * n] = (int32_t)i, indices[3 * n + 1] = (int32_t)j, indices[3 * n + 2] = (int32_t)k;
                                for (int d = 0; d < 3; ++d) position[3 * n + d] = pos[d], normal[3 * n + d] = nrm[d];
                                depth[n] = dep;
                            }
                            n += 1;
                        }
            }
    return n;
}

// for_each_voxel_object_plane_contact (impact_voxel/src/collidable.rs:1176-1208): only Corner voxels (at most three neighbours) of the
// voxel ranges inside the plane's negative halfspace (for_each_surface_voxel_maybe_intersecting_negative_halfspace_of_plane,
// object/intersection.rs:30-38; voxel_ranges_within_plane, 751-761; AxisAlignedBox::projected_onto_negative_halfspace,
// impact_geometry/src/axis_aligned_box.rs:460-488; Plane::iso_transformed / scaled, impact_geometry/src/plane.rs:170-203)
int plane_voxel_object_contacts(const VoxelObject& obj, const float rot[4], const float trans[3], const float plane_normal[3], float plane_displacement,
                                int cap, int32_t* indices, float* position, float* normal, float* depth) {
    const Quat q{rot[0], rot[1], rot[2], rot[3]};
    const V3 t{trans[0], trans[1], trans[2]}, n{plane_normal[0], plane_normal[1], plane_normal[2]};
    // plane.iso_transformed(transform_to_object_space).scaled(inverse_voxel_extent)
    const V3 point = n * plane_displacement;
    const V3 tp = qrot(q, point) + t;
    const V3 tn = qrot(q, n);
    const float inv = 1.0f / obj.extent;
    const float disp = dot(tn, tp) * inv;
    // normalized_aabb_from_voxel_ranges(occupied) projected onto the negative halfspace
    float lo[3], hi[3];
    for (int d = 0; d < 3; ++d) {
        lo[d] = (float)obj.occ_voxel[d][0];
        hi[d] = (float)obj.occ_voxel[d][1];
    }
    float flo[3] = {lo[0], lo[1], lo[2]}, fhi[3] = {hi[0], hi[1], hi[2]};
    const float nv[3] = {tn.x, tn.y, tn.z};
    const int perm[3][3] = {{0, 1, 2}, {1, 2, 0}, {2, 0, 1}};
    for (int r = 0; r < 3; ++r) {
        const int i = perm[r][0], j = perm[r][1], k = perm[r][2];
        if (std::fabs(nv[k]) > 1e-8f) {
            const float a = nv[i] * lo[i] + nv[j] * lo[j], b = nv[i] * lo[i] + nv[j] * hi[j], c = nv[i] * hi[i] + nv[j] * lo[j],
                        d = nv[i] * hi[i] + nv[j] * hi[j];
            const float mn = fmin_rs(fmin_rs(fmin_rs(a, b), c), d);
            const float extremal = (disp - mn) / nv[k];
            if (!std::signbit(nv[k])) {
                flo[k] = fmin_rs(flo[k], extremal);
                fhi[k] = fmin_rs(fhi[k], extremal);
            } else {
                flo[k] = fmax_rs(flo[k], extremal);
                fhi[k] = fmax_rs(fhi[k], extremal);
            }
        }
    }
    long vlo[3], vhi[3];
    for (int d = 0; d < 3; ++d) {
        const float fl = std::floor(flo[d]), ce = std::ceil(fhi[d]);
        const long s = (long)(fl > 0.0f ? fl : 0.0f), e = ce > 0.0f ? (long)ce : 0;
        vlo[d] = std::max<long>(obj.occ_voxel[d][0], s);
        vhi[d] = std::min<long>(obj.occ_voxel[d][1], e);
        if (vlo[d] >= vhi[d]) return 0;
    }
    int cnt = 0;
    const Quat qi = conj(q);
    for (long I = vlo[0] / CHUNK; I < (vhi[0] + CHUNK - 1) / CHUNK; ++I)
        for (long J = vlo[1] / CHUNK; J < (vhi[1] + CHUNK - 1) / CHUNK; ++J)
            for (long K = vlo[2] / CHUNK; K < (vhi[2] + CHUNK - 1) / CHUNK; ++K) {
                const Chunk& ch = obj.chunks[obj.cidx((int)I, (int)J, (int)K)];
                if (ch.kind != K_NONUNIFORM) continue;
                const Voxel* cv = &obj.voxels[(size_t)ch.data_offset << 12];
                const long base[3] = {I * CHUNK, J * CHUNK, K * CHUNK};
                for (long i = std::max(base[0], vlo[0]); i < std::min(base[0] + CHUNK, vhi[0]); ++i)
                    for (long j = std::max(base[1], vlo[1]); j < std::min(base[1] + CHUNK, vhi[1]); ++j)
                        for (long k = std::max(base[2], vlo[2]); k < std::min(base[2] + CHUNK, vhi[2]); ++k) {
                            const Voxel& v = cv[((i - base[0]) << 8) | ((j - base[1]) << 4) | (k - base[2])];
                            if (v.empty()) continue;
                            if (__builtin_popcount(v.flags & 0xFCu) > 3) continue;  // Corner placement only (lib.rs:330-342)
                            const V3 p_obj{((float)i + 0.5f) * obj.extent, ((float)j + 0.5f) * obj.extent, ((float)k + 0.5f) * obj.extent};
                            const V3 pw = qrot(qi, p_obj - t);
                            const float vr = -sd_to_f32(v.sd) * obj.extent;
                            const float pa[3] = {pw.x, pw.y, pw.z};
                            float pos[3], nrm[3], dep;
                            if (!orc_sphere_plane_contact(pa, vr, plane_normal, plane_displacement, pos, nrm, &dep)) continue;
                            if (cnt < cap) {
                                indices[3 * cnt] = (int32_t)i, indices[3 * cnt + 1] = (int32_t)j, indices[3 * cnt + 2] = (int32_t)k;
                                for (int d = 0; d < 3; ++d) position[3 * cnt + d] = pos[d], normal[3 * cnt + d] = nrm[d];
                                depth[cnt] = dep;
                            }
                            cnt += 1;
                        }
            }
    return cnt;
}

// determine_capsule_sphere_contact_geometry (impact_physics/src/collision/collidable/capsule.rs:212-270) with
// parameter_of_closest_point_on_line_segment_to_point (impact_geometry/src/line.rs:26-45). The branch for a sphere centre ON the
// segment uses glam's Vec3A::any_orthogonal_vector and Vec3A / f32 (glam 0.30.10, not under /root/reference: restated from its
// published source — |x| > |y| ? (-z, 0, x) : (0, z, -y); component-wise division) — parity unpinned for that branch.
static bool capsule_sphere_contact(V3 seg_start, V3 seg_vec, float cap_radius, V3 sc, float sr, V3& position, V3& normal, float& depth) {
    const float len2 = dot(seg_vec, seg_vec);
    float param = 0.0f;
    if (!(len2 <= 1e-8f)) {
        const V3 sp = sc - seg_start;
        param = dot(seg_vec, sp) / len2;
        param = param < 0.0f ? 0.0f : (param > 1.0f ? 1.0f : param);
    }
    const V3 closest = seg_start + param * seg_vec;
    const V3 disp = sc - closest;
    const float d2 = dot(disp, disp);
    const float max_d = sr + cap_radius;
    if (d2 > max_d * max_d) return false;
    const float dist = std::sqrt(d2);
    V3 cap_normal;
    if (dist > 1e-8f) {
        cap_normal = div_recip(disp, dist);
        depth = fmax_rs(0.0f, max_d - dist);
    } else {
        const V3 o = std::fabs(seg_vec.x) > std::fabs(seg_vec.y) ? V3{-seg_vec.z, 0.0f, seg_vec.x} : V3{0.0f, seg_vec.z, -seg_vec.y};
        const float n2 = dot(o, o);
        if (n2 > 1e-8f * 1e-8f) cap_normal = div_elem(o, std::sqrt(n2));
        else cap_normal = V3{0.0f, 0.0f, 1.0f};
        depth = fmax_rs(0.0f, max_d);
    }
    normal = -cap_normal;
    position = sc + sr * normal;
    return true;
}

// for_each_capsule_voxel_object_contact (impact_voxel/src/collidable.rs:1257-1286) over
// for_each_surface_voxel_maybe_intersecting_capsule (object/intersection.rs:73-82); Capsule::iso_transformed / scaled / compute_aabb
// (impact_geometry/src/capsule.rs:100-137)
int capsule_voxel_object_contacts(const VoxelObject& obj, const float rot[4], const float trans[3], const float seg_start[3], const float seg_vec[3],
                                  float radius, int cap, int32_t* indices, float* position, float* normal, float* depth) {
    const Quat q{rot[0], rot[1], rot[2], rot[3]};
    const V3 t{trans[0], trans[1], trans[2]}, a{seg_start[0], seg_start[1], seg_start[2]}, v{seg_vec[0], seg_vec[1], seg_vec[2]};
    const V3 a_obj = qrot(q, a) + t, v_obj = qrot(q, v);
    const float inv = 1.0f / obj.extent;
    const V3 an = a_obj * inv, vn = v_obj * inv;
    const float rn = inv * radius;
    const V3 en = an + vn;  // segment_end
    const float lo_a[3] = {an.x - rn, an.y - rn, an.z - rn}, hi_a[3] = {an.x + rn, an.y + rn, an.z + rn};
    const float lo_e[3] = {en.x - rn, en.y - rn, en.z - rn}, hi_e[3] = {en.x + rn, en.y + rn, en.z + rn};
    long vlo[3], vhi[3];
    for (int d = 0; d < 3; ++d) {
        const float lo = fmin_rs(lo_a[d], lo_e[d]), hi = fmax_rs(hi_a[d], hi_e[d]);
        const float fl = std::floor(lo), ce = std::ceil(hi);
        const long s = (long)(fl > 0.0f ? fl : 0.0f), e = ce > 0.0f ? (long)ce : 0;
        vlo[d] = std::max<long>(obj.occ_voxel[d][0], s);
        vhi[d] = std::min<long>(obj.occ_voxel[d][1], e);
        if (vlo[d] >= vhi[d]) return 0;
    }
    int n = 0;
    const Quat qi = conj(q);
    for (long I = vlo[0] / CHUNK; I < (vhi[0] + CHUNK - 1) / CHUNK; ++I)
        for (long J = vlo[1] / CHUNK; J < (vhi[1] + CHUNK - 1) / CHUNK; ++J)
            for (long K = vlo[2] / CHUNK; K < (vhi[2] + CHUNK - 1) / CHUNK; ++K) {
                const Chunk& ch = obj.chunks[obj.cidx((int)I, (int)J, (int)K)];
                if (ch.kind != K_NONUNIFORM) continue;
                const Voxel* cv = &obj.voxels[(size_t)ch.data_offset << 12];
                const long base[3] = {I * CHUNK, J * CHUNK, K * CHUNK};
                for (long i = std::max(base[0], vlo[0]); i < std::min(base[0] + CHUNK, vhi[0]); ++i)
                    for (long j = std::max(base[1], vlo[1]); j < std::min(base[1] + CHUNK, vhi[1]); ++j)
                        for (long k = std::max(base[2], vlo[2]); k < std::min(base[2] + CHUNK, vhi[2]); ++k) {
                            const Voxel& vx = cv[((i - base[0]) << 8) | ((j - base[1]) << 4) | (k - base[2])];
                            if (vx.empty()) continue;
                            if (__builtin_popcount(vx.flags & 0xFCu) == 6) continue;  // Interior
                            const V3 p_obj{((float)i + 0.5f) * obj.extent, ((float)j + 0.5f) * obj.extent, ((float)k + 0.5f) * obj.extent};
                            const V3 p = qrot(qi, p_obj - t);
                            const float vr = -sd_to_f32(vx.sd) * obj.extent;
                            V3 pos, nrm;
                            float dep;
                            if (!capsule_sphere_contact(a, v, radius, p, vr, pos, nrm, dep)) continue;
                            if (n < cap) {
                                indices[3 * n] = (int32_t)i, indices[3 * n + 1] = (int32_t)j, indices[3 * n + 2] = (int32_t)k;
                                position[3 * n] = pos.x, position[3 * n + 1] = pos.y, position[3 * n + 2] = pos.z;
                                normal[3 * n] = nrm.x, normal[3 * n + 1] = nrm.y, normal[3 * n + 2] = nrm.z;
                                depth[n] = dep;
                            }
                            n += 1;
                        }
            }
    return n;
}

}  // namespace orc
