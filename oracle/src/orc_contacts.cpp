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

}  // namespace orc
