// Rigid bodies after voxels were removed from a voxel object — host arithmetic above the voxel and physics entry
// points, O(1) per object / fragment.
//
// Reference (engine/crates):
//   handle_voxel_object_after_removing_voxels                   impact_voxel/src/interaction.rs:224-403
//   apply_updated_inertial_properties_to_rigid_body             impact_voxel/src/interaction.rs:405-458
//   apply_updated_inertial_properties_to_rigid_body_preserving_momentum   impact_voxel/src/interaction.rs:460-487
//   determine_extracted_voxel_object_dynamics                   impact_voxel/src/interaction.rs:503-585
//   VoxelObjectInertialPropertyManager::{offset_reference_point_by, derive_inertial_properties}   impact_voxel/src/object/inertia.rs:156-169, 257-326
//   InertiaTensor parallel-axis deltas                           impact_physics/src/inertia.rs:511-587
//   DynamicRigidBody::{new, compute_velocity, compute_angular_velocity, synchronize_*}   impact_physics/src/rigid_body.rs:413-441, 481-493, 687-702
//   VoxelObject::is_effectively_empty                            impact_voxel/src/object.rs:803-845
// The reference keeps the managers' moments in f32 and moves them voxel by voxel; here the moments are the f64 sums the device
// returns (ivx_moments.m64, ivx_region_desc.moments) and the O(1) arithmetic runs in f64, rounded once into the f32 body record.
#include <cmath>
#include <cstring>
#include <vector>

#include "ivx_internal.hpp"

namespace {

struct D3 {
    double x, y, z;
};
inline D3 operator+(D3 a, D3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline D3 operator-(D3 a, D3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline D3 operator*(D3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
inline D3 crossd(D3 a, D3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
struct DM {
    double m[3][3];  // row, column
};
inline D3 mulv(const DM& a, D3 v) {
    return {a.m[0][0] * v.x + a.m[0][1] * v.y + a.m[0][2] * v.z, a.m[1][0] * v.x + a.m[1][1] * v.y + a.m[1][2] * v.z,
            a.m[2][0] * v.x + a.m[2][1] * v.y + a.m[2][2] * v.z};
}
inline DM mulm(const DM& a, const DM& b) {
    DM r;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) r.m[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j];
    return r;
}
inline DM transposed(const DM& a) {
    DM r;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) r.m[i][j] = a.m[j][i];
    return r;
}
DM rotation_of(const float q[4]) {  // unit quaternion (x, y, z, w) -> rotation matrix
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    DM r;
    r.m[0][0] = 1 - 2 * (y * y + z * z), r.m[0][1] = 2 * (x * y - w * z), r.m[0][2] = 2 * (x * z + w * y);
    r.m[1][0] = 2 * (x * y + w * z), r.m[1][1] = 1 - 2 * (x * x + z * z), r.m[1][2] = 2 * (y * z - w * x);
    r.m[2][0] = 2 * (x * z - w * y), r.m[2][1] = 2 * (y * z + w * x), r.m[2][2] = 1 - 2 * (x * x + y * y);
    return r;
}
DM from_column_major(const float m[9]) {
    DM r;
    for (int c = 0; c < 3; ++c)
        for (int rr = 0; rr < 3; ++rr) r.m[rr][c] = m[3 * c + rr];
    return r;
}
void to_column_major(const DM& a, float m[9]) {
    for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 3; ++r) m[3 * c + r] = (float)a.m[r][c];
}
bool invert(const DM& a, DM* out) {
    const double c00 = a.m[1][1] * a.m[2][2] - a.m[1][2] * a.m[2][1], c01 = a.m[1][2] * a.m[2][0] - a.m[1][0] * a.m[2][2],
                 c02 = a.m[1][0] * a.m[2][1] - a.m[1][1] * a.m[2][0];
    const double det = a.m[0][0] * c00 + a.m[0][1] * c01 + a.m[0][2] * c02;
    if (det == 0.0 || !std::isfinite(det)) return false;
    const double id = 1.0 / det;
    out->m[0][0] = c00 * id, out->m[1][0] = c01 * id, out->m[2][0] = c02 * id;
    out->m[0][1] = (a.m[0][2] * a.m[2][1] - a.m[0][1] * a.m[2][2]) * id;
    out->m[1][1] = (a.m[0][0] * a.m[2][2] - a.m[0][2] * a.m[2][0]) * id;
    out->m[2][1] = (a.m[0][1] * a.m[2][0] - a.m[0][0] * a.m[2][1]) * id;
    out->m[0][2] = (a.m[0][1] * a.m[1][2] - a.m[0][2] * a.m[1][1]) * id;
    out->m[1][2] = (a.m[0][2] * a.m[1][0] - a.m[0][0] * a.m[1][2]) * id;
    out->m[2][2] = (a.m[0][0] * a.m[1][1] - a.m[0][1] * a.m[1][0]) * id;
    return true;
}

struct Derived {
    double mass;
    D3 com;
    DM inertia, inverse;  // about the centre of mass
};
// compute_inertial_properties_from_moments (object/inertia.rs:288-326)
bool derive(const double m[10], Derived* out) {
    if (!(m[0] > 0.0)) return false;
    out->mass = m[0];
    out->com = D3{m[1], m[2], m[3]} * (1.0 / m[0]);
    const D3 c = out->com;
    DM j = {{{m[4], -m[7], -m[9]}, {-m[7], m[5], -m[8]}, {-m[9], -m[8], m[6]}}};
    const DM delta = {{{-m[0] * (c.y * c.y + c.z * c.z), m[0] * c.x * c.y, m[0] * c.z * c.x},
                       {m[0] * c.x * c.y, -m[0] * (c.z * c.z + c.x * c.x), m[0] * c.y * c.z},
                       {m[0] * c.z * c.x, m[0] * c.y * c.z, -m[0] * (c.x * c.x + c.y * c.y)}}};
    for (int i = 0; i < 3; ++i)
        for (int k = 0; k < 3; ++k) out->inertia.m[i][k] = j.m[i][k] + delta.m[i][k];
    return invert(out->inertia, &out->inverse);
}
// offset_reference_point_by (object/inertia.rs:257-267): the same moments about the point `off`
void offset_reference(double m[10], D3 off) {
    const double mass = m[0];
    const D3 com = D3{m[1], m[2], m[3]} * (1.0 / mass), d = off - com;
    // to the centre of mass (-mass * ...), then out to the point (+mass * ...)
    const double moi[3] = {mass * ((d.y * d.y + d.z * d.z) - (com.y * com.y + com.z * com.z)), mass * ((d.z * d.z + d.x * d.x) - (com.z * com.z + com.x * com.x)),
                           mass * ((d.x * d.x + d.y * d.y) - (com.x * com.x + com.y * com.y))};
    const double poi[3] = {mass * (d.x * d.y - com.x * com.y), mass * (d.y * d.z - com.y * com.z), mass * (d.z * d.x - com.z * com.x)};
    m[1] -= off.x * mass, m[2] -= off.y * mass, m[3] -= off.z * mass;
    for (int i = 0; i < 3; ++i) m[4 + i] += moi[i], m[7 + i] += poi[i];
}

struct Motion {
    D3 position, velocity, angular_velocity;
    DM rot;
};
Motion motion_of(const ivx_rigid_body& b) {  // compute_velocity / compute_angular_velocity = R I^-1 R^T L
    Motion s;
    s.position = {b.position[0], b.position[1], b.position[2]};
    s.velocity = D3{b.momentum[0], b.momentum[1], b.momentum[2]} * (1.0 / (double)b.mass);
    s.rot = rotation_of(b.orientation);
    const DM inv_world = mulm(mulm(s.rot, from_column_major(b.inv_inertia)), transposed(s.rot));
    s.angular_velocity = mulv(inv_world, D3{b.angular_momentum[0], b.angular_momentum[1], b.angular_momentum[2]});
    return s;
}
void set_inertial(ivx_rigid_body* b, const Derived& d) {
    b->mass = (float)d.mass;
    to_column_major(d.inertia, b->inertia);
    to_column_major(d.inverse, b->inv_inertia);
}
void store3(float* p, D3 v) { p[0] = (float)v.x, p[1] = (float)v.y, p[2] = (float)v.z; }
void sync_momenta(ivx_rigid_body* b, const Derived& d, const Motion& s, D3 velocity) {  // synchronize_momentum / synchronize_angular_momentum
    store3(b->momentum, velocity * d.mass);
    store3(b->angular_momentum, mulv(mulm(mulm(s.rot, d.inertia), transposed(s.rot)), s.angular_velocity));
}

}  // namespace

extern "C" {

int ivx_offset_reference_point(double moments[10], const float offset[3]) {
    IVX_REQUIRE(moments && offset, IVX_ERR_INVALID, "ivx_offset_reference_point: null argument");
    IVX_REQUIRE(moments[0] > 0.0, IVX_ERR_INVALID, "ivx_offset_reference_point: the moments describe no mass");
    offset_reference(moments, D3{offset[0], offset[1], offset[2]});
    return IVX_OK;
}

int ivx_apply_updated_inertial_properties(ivx_rigid_body* body, const double moments[10], const float original_local_center_of_mass[3], int preserve_momentum,
                                          float new_local_center_of_mass[3]) {
    IVX_REQUIRE(body && moments && original_local_center_of_mass && new_local_center_of_mass, IVX_ERR_INVALID,
                "ivx_apply_updated_inertial_properties: null argument");
    Derived d;
    IVX_REQUIRE(derive(moments, &d), IVX_ERR_INVALID, "ivx_apply_updated_inertial_properties: the moments describe no mass or a singular inertia tensor");
    const Motion s = motion_of(*body);
    const D3 local = d.com - D3{original_local_center_of_mass[0], original_local_center_of_mass[1], original_local_center_of_mass[2]};
    const D3 world = mulv(s.rot, local);
    set_inertial(body, d);
    store3(body->position, s.position + world);
    if (!preserve_momentum) sync_momenta(body, d, s, s.velocity + crossd(s.angular_velocity, world));
    store3(new_local_center_of_mass, d.com);
    return IVX_OK;
}

int ivx_extracted_object_dynamics(double moments[10], const uint32_t origin_offset_in_parent[3], float voxel_extent, const float original_local_center_of_mass[3],
                                  const ivx_rigid_body* parent_body, ivx_rigid_body* fragment_body, float new_local_center_of_mass[3]) {
    IVX_REQUIRE(moments && origin_offset_in_parent && original_local_center_of_mass && parent_body && fragment_body && new_local_center_of_mass,
                IVX_ERR_INVALID, "ivx_extracted_object_dynamics: null argument");
    IVX_REQUIRE(moments[0] > 0.0, IVX_ERR_INVALID, "ivx_extracted_object_dynamics: the fragment has no mass");
    const Motion s = motion_of(*parent_body);
    const D3 com_in_parent = D3{moments[1], moments[2], moments[3]} * (1.0 / moments[0]);
    const D3 local = com_in_parent - D3{original_local_center_of_mass[0], original_local_center_of_mass[1], original_local_center_of_mass[2]};
    const D3 world = mulv(s.rot, local);
    offset_reference(moments, D3{(double)((float)origin_offset_in_parent[0] * voxel_extent), (double)((float)origin_offset_in_parent[1] * voxel_extent),
                                 (double)((float)origin_offset_in_parent[2] * voxel_extent)});
    Derived d;
    IVX_REQUIRE(derive(moments, &d), IVX_ERR_INVALID, "ivx_extracted_object_dynamics: singular inertia tensor");
    memset(fragment_body, 0, sizeof(*fragment_body));
    set_inertial(fragment_body, d);
    store3(fragment_body->position, s.position + world);
    memcpy(fragment_body->orientation, parent_body->orientation, sizeof(fragment_body->orientation));
    sync_momenta(fragment_body, d, s, s.velocity + crossd(s.angular_velocity, world));
    store3(new_local_center_of_mass, d.com);
    return IVX_OK;
}

int ivx_handle_voxel_object_after_removing_voxels(ivx_grid* g, const float densities[256], double moments[10], ivx_rigid_body* body,
                                                  const float original_local_center_of_mass[3], int removed_mass_destroyed, ivx_extracted_object* out,
                                                  size_t cap, size_t* n_out, int* original_object_empty, float new_local_center_of_mass[3]) {
    const char* who = "ivx_handle_voxel_object_after_removing_voxels";
    IVX_REQUIRE(g && densities && moments && body && original_local_center_of_mass && n_out && original_object_empty && new_local_center_of_mass &&
                    (out || cap == 0),
                IVX_ERR_INVALID, "%s: null argument", who);
    *n_out = 0;
    *original_object_empty = 0;
    for (int d = 0; d < 3; ++d) new_local_center_of_mass[d] = original_local_center_of_mass[d];
    int rc;
    // (the split-offs below take the moments that leave the object from the object's resident density table: this one)
    if ((rc = ivx_grid_set_densities(g, densities))) return rc;
    uint32_t n_regions = 0;
    if ((rc = ivx_label_regions(g, &n_regions))) return rc;
    std::vector<ivx_region_desc> desc(n_regions ? n_regions : 1);
    auto effectively_empty = [&](bool* empty) -> int {  // fewer than NON_EMPTY_VOXEL_THRESHOLD = 8 non-empty voxels
        size_t n = 0;
        desc.resize(n_regions ? n_regions : 1);
        int r = ivx_regions_describe(g, densities, desc.data(), desc.size(), &n);
        if (r) return r;
        uint64_t voxels = 0;
        for (size_t i = 0; i < n; ++i) voxels += desc[i].voxel_count;
        *empty = voxels < 8;
        return IVX_OK;
    };
    bool empty = false;
    if ((rc = effectively_empty(&empty))) return rc;
    if (empty) {
        *original_object_empty = 1;
        return IVX_OK;
    }
    IVX_REQUIRE(n_regions <= cap + 1, IVX_ERR_CAPACITY, "%s: %u regions can give %u fragments, capacity %zu", who, n_regions, n_regions - 1, cap);
    bool had_disconnected = false;
    while (n_regions >= 2) {  // while let Some(..) = find_two_disconnected_regions()
        had_disconnected = true;
        ivx_grid* child = nullptr;
        uint32_t origin[3] = {0, 0, 0};
        int outcome = 0;
        ivx_region_desc moved;
        memset(&moved, 0, sizeof(moved));
        if ((rc = ivx_split_off_smallest_region(g, &child, origin, &outcome, &moved))) return rc;
        if (outcome == 0) break;
        for (int q = 0; q < 10; ++q) moments[q] -= moved.moments[q];  // the property transferrer takes them out of the parent's manager either way
        if (outcome == 1) {
            ivx_extracted_object& e = out[*n_out];
            memset(&e, 0, sizeof(e));
            e.grid = child;
            for (int d = 0; d < 3; ++d) e.origin_offset_in_parent[d] = origin[d];
            memcpy(e.moments, moved.moments, sizeof(e.moments));
            if ((rc = ivx_extracted_object_dynamics(e.moments, origin, g->extent, original_local_center_of_mass, body, &e.body, e.local_center_of_mass))) {
                ivx_grid_destroy(child);
                return rc;
            }
            *n_out += 1;
        }
        if ((rc = ivx_label_regions(g, &n_regions))) return rc;
    }
    if (*n_out == 0) {
        bool now_empty = false;
        if (had_disconnected && (rc = effectively_empty(&now_empty))) return rc;
        *original_object_empty = now_empty ? 1 : 0;
        if (now_empty) return IVX_OK;
        return ivx_apply_updated_inertial_properties(body, moments, original_local_center_of_mass, removed_mass_destroyed ? 1 : 0, new_local_center_of_mass);
    }
    if ((rc = effectively_empty(&empty))) return rc;
    *original_object_empty = empty ? 1 : 0;
    if (empty) return IVX_OK;
    return ivx_apply_updated_inertial_properties(body, moments, original_local_center_of_mass, 0, new_local_center_of_mass);
}
}
