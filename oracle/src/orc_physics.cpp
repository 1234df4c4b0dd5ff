// ORACLE (test infrastructure only — never linked into the product library).
//
// CPU restatement of the reference's rigid-body step and sequential-impulses contact solver
// (paths relative to /root/reference/engine/crates/impact_physics/src):
//   DynamicRigidBody / KinematicRigidBody            rigid_body.rs:94-117, 411-762, 1013-1034
//   angular velocity / momentum, orientation d/dt    quantities.rs:160-172, 372-399
//   rotated / inverse rotated inertia matrix         inertia.rs:401-432
//   ConstrainedBody                                  constraint.rs:137-150, 476-523
//   Contact::prepare, impulses, clamp, apply,        constraint/contact.rs:230-517, 788-843
//     positional correction
//   interlock detection + separating contact         constraint/contact.rs:610-780
//   ConstraintCache (warm start, swap_remove order)  constraint/solver.rs:386-452
//   warm start + sequential sweeps + corrections     constraint/solver.rs:242-289, 481-541
//   write-back                                       constraint/solver.rs:571-602
//   perform_physics_step order                       lib.rs:31-110
//   sphere-sphere / sphere-plane contact geometry    collision/collidable/sphere.rs:105-160
//   (only to drive the reference's own sphere-collision tests, tests/constraint.rs:339-576)
// Vector/quaternion arithmetic follows orc_math.hpp's restatement of glam 0.30.10 (not vendored:
// bit-level behaviour "parity unpinned"; the reference's tests pin results to 1e-6 absolute).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <unordered_map>
#include <vector>

#include "../include/oracle.h"
#include "orc_math.hpp"

namespace orc {

static inline V3 ld3(const float* p) { return v3(p[0], p[1], p[2]); }
static inline void st3(float* p, V3 v) {
    p[0] = v.x;
    p[1] = v.y;
    p[2] = v.z;
}
static inline Quat ldq(const float* p) { return Quat{p[0], p[1], p[2], p[3]}; }
static inline void stq(float* p, Quat q) {
    p[0] = q.x;
    p[1] = q.y;
    p[2] = q.z;
    p[3] = q.w;
}
static inline M3 ldm(const float* p) { return M3{ld3(p), ld3(p + 3), ld3(p + 6)}; }
static inline void stm(float* p, const M3& m) {
    st3(p, m.c0);
    st3(p + 3, m.c1);
    st3(p + 6, m.c2);
}
// glam Quat::mul_vec3a
static inline V3 qrot(Quat q, V3 v) {
    V3 b = v3(q.x, q.y, q.z);
    float b2 = dot(b, b);
    return (v * (q.w * q.w - b2) + b * (dot(v, b) * 2.0f)) + cross(b, v) * (q.w * 2.0f);
}
static const float F32_EPS = 1.1920929e-07f;

// AngularVelocity::from_vector(..).as_vector() (quantities.rs:160-172): axis * speed, zero below EPSILON
struct AngVel {
    V3 axis;
    float speed;
};
static inline AngVel angvel_from_vector(V3 w) {
    float n2 = dot(w, w);
    if (n2 > F32_EPS * F32_EPS) {
        float n = std::sqrt(n2);
        return {div_elem(w, n), n};
    }
    return {v3(0, 1, 0), 0.0f};
}
static inline V3 angvel_vector(AngVel a) { return a.axis * a.speed; }

static inline M3 rotated(const M3& m, Quat q) {  // R * M * R^T (inertia.rs:401-404, 429-432)
    M3 r = m3_from_quat(q);
    return mul(mul(r, m), transpose(r));
}

struct CBody {  // ConstrainedBody (constraint.rs:137-150)
    float inv_mass;
    M3 inv_inertia;
    V3 position;
    Quat orientation;
    V3 velocity, angular_velocity;
};

static inline V3 body_velocity(const orc_rigid_body& b) { return div_recip(ld3(b.momentum), b.mass); }
static inline AngVel body_angular_velocity(const orc_rigid_body& b) {
    return angvel_from_vector(mul(rotated(ldm(b.inv_inertia), ldq(b.orientation)), ld3(b.angular_momentum)));
}

static CBody cbody_from_dynamic(const orc_rigid_body& b) {
    CBody c;
    c.inv_mass = 1.0f / b.mass;
    c.inv_inertia = rotated(ldm(b.inv_inertia), ldq(b.orientation));
    c.position = ld3(b.position);
    c.orientation = ldq(b.orientation);
    c.velocity = body_velocity(b);
    c.angular_velocity = angvel_vector(body_angular_velocity(b));
    return c;
}
static CBody cbody_from_kinematic(const orc_kinematic_body& b) {
    CBody c;
    c.inv_mass = 0.0f;
    c.inv_inertia = M3{v3s(0), v3s(0), v3s(0)};
    c.position = ld3(b.position);
    c.orientation = ldq(b.orientation);
    c.velocity = ld3(b.velocity);
    c.angular_velocity = ld3(b.angular_axis) * b.angular_speed;
    return c;
}
static inline V3 to_world(const CBody& b, V3 p) { return qrot(b.orientation, p) + b.position; }
static inline V3 to_body(const CBody& b, V3 p) { return qrot(conj(b.orientation), p - b.position); }
static inline V3 point_velocity(const CBody& b, V3 disp) { return b.velocity + cross(b.angular_velocity, disp); }
static inline float effective_mass(const CBody& a, const CBody& b, V3 da, V3 db, V3 dir) {
    V3 ca = cross(da, dir), cb = cross(db, dir);
    return 1.0f / (((a.inv_mass + b.inv_mass) + dot(ca, mul(a.inv_inertia, ca))) + dot(cb, mul(b.inv_inertia, cb)));
}
static inline void tangents(V3 n, V3& t1, V3& t2) {  // contact.rs:813-832
    const float INV_SQRT_THREE = 0.57735f;
    t1 = normalize(std::fabs(n.x) < INV_SQRT_THREE ? v3(0.0f, n.z, -n.y) : v3(n.y, -n.x, 0.0f));
    t2 = cross(n, t1);
}

struct Prepared {  // PreparedContact (contact.rs:63-86)
    V3 local_a, local_b, normal, tangent, bitangent;
    float m_n, m_t, m_b, friction, target;
};
struct Impulses {
    float n, t, b;
};

static Prepared prepare_contact(const orc_contact& c, const CBody& a, const CBody& b) {  // contact.rs:233-310
    const V3 pos = ld3(c.position), normal = ld3(c.normal);
    Prepared p;
    p.local_a = to_body(a, pos - normal * c.depth);  // position_on_a = position - depth * normal
    p.local_b = to_body(b, pos);
    const V3 da = pos - a.position, db = pos - b.position;
    V3 t1, t2;
    tangents(normal, t1, t2);
    p.normal = normal;
    p.tangent = t1;
    p.bitangent = t2;
    p.m_n = effective_mass(a, b, da, db, normal);
    p.m_t = effective_mass(a, b, da, db, t1);
    p.m_b = effective_mass(a, b, da, db, t2);
    const V3 rel = point_velocity(a, da) - point_velocity(b, db);
    const float sep = dot(normal, rel);
    p.target = std::fabs(sep) >= 0.4f ? -c.restitution * sep : 0.0f;
    const float d1 = dot(rel, t1), d2 = dot(rel, t2);
    const float slip2 = d1 * d1 + d2 * d2;
    p.friction = slip2 >= 1e-4f ? c.dynamic_friction : c.static_friction;
    return p;
}

static Impulses compute_impulses(const Prepared& p, const CBody& a, const CBody& b) {  // contact.rs:329-368
    const V3 pb = to_world(b, p.local_b);
    const V3 da = pb - a.position, db = pb - b.position;
    const V3 rel = point_velocity(a, da) - point_velocity(b, db);
    const float sep = dot(p.normal, rel);
    return {-p.m_n * (sep - p.target), -p.m_t * dot(p.tangent, rel), -p.m_b * dot(p.bitangent, rel)};
}
static Impulses clamp_impulses(const Prepared& p, Impulses i) {  // contact.rs:370-397
    const float n = fmax_rs(0.0f, i.n);
    const float max_t = p.friction * n;
    const float mag = std::sqrt(i.t * i.t + i.b * i.b);
    const float s = mag > max_t ? max_t / mag : 1.0f;
    return {n, i.t * s, i.b * s};
}
static void apply_impulses(const Prepared& p, CBody& a, CBody& b, Impulses i) {  // contact.rs:399-438
    const V3 dp = (p.normal * i.n + p.tangent * i.t) + p.bitangent * i.b;
    const V3 pb = to_world(b, p.local_b);
    const V3 da = pb - a.position, db = pb - b.position;
    a.velocity = a.velocity + dp * a.inv_mass;
    b.velocity = b.velocity - dp * b.inv_mass;
    a.angular_velocity = a.angular_velocity + mul(a.inv_inertia, cross(da, dp));
    b.angular_velocity = b.angular_velocity - mul(b.inv_inertia, cross(db, dp));
}
static inline Quat pseudo_advanced(Quat q, V3 w) {  // contact.rs:835-843, quantities.rs:372-378
    V3 h = w * 0.5f;
    Quat d = qmul(Quat{h.x, h.y, h.z, 0.0f}, q);
    return qnormalize(Quat{q.x + d.x, q.y + d.y, q.z + d.z, q.w + d.w});
}
static void positional_correction(const Prepared& p, CBody& a, CBody& b, float factor) {  // contact.rs:440-517
    const V3 pa = to_world(a, p.local_a), pb = to_world(b, p.local_b);
    const float depth = dot(p.normal, pb - pa);
    if (depth <= 0.0f) return;
    const V3 da = pb - a.position, db = pb - b.position;
    const float m = effective_mass(a, b, da, db, p.normal);
    const float pseudo = m * factor * depth;
    const V3 dp = p.normal * pseudo;
    const V3 va = dp * a.inv_mass, wa = mul(a.inv_inertia, cross(da, dp));
    const V3 vb = dp * (-b.inv_mass), wb = mul(M3{-b.inv_inertia.c0, -b.inv_inertia.c1, -b.inv_inertia.c2}, cross(db, dp));
    a.position = a.position + va;
    b.position = b.position + vb;
    a.orientation = pseudo_advanced(a.orientation, wa);
    b.orientation = pseudo_advanced(b.orientation, wb);
}

// impact_math/src/random/splitmix.rs:4-15 (synthetic contact ids, contact.rs:180-199, 770-772)
static inline uint64_t splitmix_two(uint64_t a, uint64_t b) {
    auto mix = [](uint64_t z) {
        z += 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    };
    return mix(a ^ mix(b));  // random_u64_from_two_states (random/splitmix.rs:13-15)
}

// contact.rs:610-780
static bool interlocked(const orc_contact* c, int n) {
    float abs_sum = 0.0f;
    V3 vec = v3s(0);
    for (int i = 0; i < n; ++i) {
        if (c[i].depth <= 0.0f) continue;
        abs_sum += c[i].depth;
        vec = vec + ld3(c[i].normal) * c[i].depth;
    }
    if (abs_sum < 1e-6f) return false;
    return dot(vec, vec) / (abs_sum * abs_sum) < 0.1f;
}
template <class F>
static V3 max_displacement(const orc_contact* c, int n, F map) {
    float best = -INFINITY;
    int bi = -1, bj = -1;
    for (int i = 0; i + 1 < n; ++i)
        for (int j = i + 1; j < n; ++j) {
            V3 d = map(ld3(c[i].position)) - map(ld3(c[j].position));
            float s = dot(d, d);
            if (s > best) {
                best = s;
                bi = i;
                bj = j;
            }
        }
    if (bi < 0) return v3s(0);
    return map(ld3(c[bj].position)) - map(ld3(c[bi].position));
}
static bool normalized_if_above(V3 v, float min_norm, V3& out) {
    float n2 = dot(v, v);
    if (!(n2 > min_norm * min_norm)) return false;
    out = div_elem(v, std::sqrt(n2));
    return true;
}
static bool separate_along(const CBody& a, const CBody& b, const orc_contact* c, int n, V3 axis, orc_contact& out) {
    if (dot(axis, a.position - b.position) < 0.0f) axis = -axis;
    float lo = INFINITY, hi = -INFINITY;
    int i0 = -1, i1 = -1;
    for (int i = 0; i < n; ++i) {
        float d = dot(ld3(c[i].position), axis);
        if (d < lo) {
            lo = d;
            i0 = i;
        }
        if (d > hi) {
            hi = d;
            i1 = i;
        }
    }
    if (i0 == i1) return false;
    out = c[i0];
    st3(out.normal, axis);
    out.depth = hi - lo;
    out.restitution = 0.0f;
    out.static_friction = INFINITY;
    out.dynamic_friction = INFINITY;
    out.id = splitmix_two(c[i0].id, c[i1].id);
    return true;
}
static bool separating_contact(const CBody& a, const CBody& b, const orc_contact* c, int n, orc_contact& out) {
    if (n == 0) return false;
    V3 major, middle, minor;
    if (!normalized_if_above(max_displacement(c, n, [](V3 p) { return p; }), 1e-6f, major)) return false;
    V3 md = max_displacement(c, n, [&](V3 p) { return p - major * dot(p, major); });
    if (!normalized_if_above(md, 1e-6f, middle)) return separate_along(a, b, c, n, major, out);
    if (!normalized_if_above(cross(major, middle), 1e-4f, minor)) return separate_along(a, b, c, n, major, out);
    if (separate_along(a, b, c, n, minor, out)) return true;
    return separate_along(a, b, c, n, middle, out);
}

}  // namespace orc

using namespace orc;

struct orc_physics {
    std::vector<orc_rigid_body> dyn;
    std::vector<orc_kinematic_body> kin;
    // ConstraintCache<ContactID, PreparedContact> (solver.rs:60-81, 386-452)
    struct Entry {
        uint32_t a, b;  // constrained body indices
        Prepared p;
        Impulses acc;
        bool was_prepared;
    };
    std::vector<Entry> cache;
    std::vector<uint64_t> keys;  // KeyIndexMapper: key at idx
    std::unordered_map<uint64_t, uint32_t> index_of;
    // ConstrainedBodyManager for the current step
    std::vector<CBody> bodies;
    std::vector<uint32_t> body_ref;  // rigid body reference (bit31 = kinematic) per constrained body
    std::unordered_map<uint32_t, uint32_t> body_index;
    orc_solver_config cfg{8, 0.4f, 3, 0.2f};
    std::vector<uint32_t> joint_refs;  // SphericalJoint anchors' bodies, two per joint (constraint.rs:36, 183-190)
};

static uint32_t add_body(orc_physics* w, uint32_t ref) {  // constraint.rs:442-467
    auto it = w->body_index.find(ref);
    if (it != w->body_index.end()) return it->second;
    CBody c = (ref & 0x80000000u) ? cbody_from_kinematic(w->kin[ref & 0x7FFFFFFFu]) : cbody_from_dynamic(w->dyn[ref]);
    uint32_t idx = (uint32_t)w->bodies.size();
    w->bodies.push_back(c);
    w->body_ref.push_back(ref);
    w->body_index[ref] = idx;
    return idx;
}

static void register_contact(orc_physics* w, const orc_contact& c, uint32_t ia, uint32_t ib) {  // solver.rs:160-176, 406-432
    orc_physics::Entry e;
    e.a = ia;
    e.b = ib;
    e.p = prepare_contact(c, w->bodies[ia], w->bodies[ib]);
    e.acc = {0, 0, 0};
    e.was_prepared = true;
    auto it = w->index_of.find(c.id);
    if (it != w->index_of.end()) {
        const orc_physics::Entry& old = w->cache[it->second];
        // can_use_warm_impulses_from (contact.rs:313-327)
        if (dot(e.p.normal, old.p.normal) > 1.0f - 1e-2f && dot(e.p.tangent, old.p.tangent) > 1.0f - 1e-2f) {
            const float wgt = w->cfg.old_impulse_weight;
            e.acc = {old.acc.n * wgt, old.acc.t * wgt, old.acc.b * wgt};
        }
        w->cache[it->second] = e;
    } else {
        w->index_of[c.id] = (uint32_t)w->cache.size();
        w->cache.push_back(e);
        w->keys.push_back(c.id);
    }
}

extern "C" {

orc_physics* orc_physics_create(void) { return new orc_physics(); }
void orc_physics_free(orc_physics* w) { delete w; }
void orc_physics_set_config(orc_physics* w, const orc_solver_config* c) { w->cfg = *c; }
void orc_physics_set_bodies(orc_physics* w, const orc_rigid_body* dyn, int n_dyn, const orc_kinematic_body* kin, int n_kin) {
    w->dyn.assign(dyn, dyn + n_dyn);
    w->kin.assign(kin, kin + n_kin);
}
void orc_physics_get_bodies(const orc_physics* w, orc_rigid_body* dyn, orc_kinematic_body* kin) {
    if (dyn) std::memcpy(dyn, w->dyn.data(), w->dyn.size() * sizeof(orc_rigid_body));
    if (kin) std::memcpy(kin, w->kin.data(), w->kin.size() * sizeof(orc_kinematic_body));
}

// DynamicRigidBody::new (rigid_body.rs:411-441)
void orc_rigid_body_new(orc_rigid_body* out, float mass, const float inertia[9], const float inv_inertia[9], const float position[3],
                        const float orientation[4], const float velocity[3], const float angular_velocity[3]) {
    std::memset(out, 0, sizeof(*out));
    out->mass = mass;
    std::memcpy(out->inertia, inertia, 36);
    std::memcpy(out->inv_inertia, inv_inertia, 36);
    std::memcpy(out->position, position, 12);
    std::memcpy(out->orientation, orientation, 16);
    st3(out->momentum, ld3(velocity) * mass);
    AngVel av = angvel_from_vector(ld3(angular_velocity));  // AngularVelocityC::from_vector
    st3(out->angular_momentum, mul(rotated(ldm(inertia), ldq(orientation)), angvel_vector(av)));
}
// compute_velocity / compute_angular_velocity().as_vector()
void orc_rigid_body_motion(const orc_rigid_body* b, float velocity[3], float angular_velocity[3]) {
    st3(velocity, body_velocity(*b));
    st3(angular_velocity, angvel_vector(body_angular_velocity(*b)));
}

// ---- a14: re-seating rigid bodies after voxels were removed (impact_voxel/src/interaction.rs:405-602) ---------------------------------
// VoxelObjectInertialPropertyManager::offset_reference_point_by (object/inertia.rs:257-267) with
// InertiaTensor::compute_delta_to_moments_and_products_of_inertia_defined_relative_to_point (impact_physics/src/inertia.rs:531-587)
void orc_offset_reference_point(float m[10], const float offset[3]) {
    const float mass = m[0];
    const V3 off = ld3(offset);
    const V3 com = div_recip(v3(m[1], m[2], m[3]), mass);  // derive_center_of_mass: Vector3 / f32
    auto to_com = [&](V3 d, V3& moi, V3& poi) {
        const V3 sq = cmul(d, d);
        moi = (-mass) * (v3(sq.y, sq.z, sq.x) + v3(sq.z, sq.x, sq.y));
        poi = (-mass) * cmul(d, v3(d.y, d.z, d.x));
    };
    V3 moi1, poi1, moi2, poi2;
    to_com(com, moi1, poi1);
    to_com(off - com, moi2, poi2);  // compute_delta_from_com_...: the negated deltas
    const V3 moi = moi1 + (-moi2), poi = poi1 + (-poi2);
    const V3 mom = v3(m[1], m[2], m[3]) - off * mass;
    m[1] = mom.x, m[2] = mom.y, m[3] = mom.z;
    m[4] += moi.x, m[5] += moi.y, m[6] += moi.z;
    m[7] += poi.x, m[8] += poi.y, m[9] += poi.z;
}

// apply_updated_inertial_properties_to_rigid_body (interaction.rs:405-458) / ..._preserving_momentum (460-487): moments = the object's
// inertial property manager (about its grid origin) after the removal
void orc_apply_updated_inertial_properties(orc_rigid_body* b, const float moments[10], const float original_local_com[3], int preserve_momentum,
                                           float new_local_com[3]) {
    const V3 pos = ld3(b->position);
    const Quat q = ldq(b->orientation);
    const V3 vel = body_velocity(*b);
    const AngVel av = body_angular_velocity(*b);
    float ip[22];
    orc_derive_inertial_properties(moments, ip);  // object/inertia.rs:288-326
    const V3 com = v3(ip[1], ip[2], ip[3]);
    const V3 local_disp = com - ld3(original_local_com);
    const V3 world_disp = qrot(q, local_disp);
    b->mass = ip[0];
    std::memcpy(b->inertia, ip + 4, 36);
    std::memcpy(b->inv_inertia, ip + 13, 36);
    st3(b->position, pos + world_disp);
    if (!preserve_momentum) {
        const V3 dv = cross(angvel_vector(av), world_disp);
        st3(b->momentum, (vel + dv) * b->mass);  // synchronize_momentum
        st3(b->angular_momentum, mul(rotated(ldm(b->inertia), q), angvel_vector(av)));  // synchronize_angular_momentum
    }
    st3(new_local_com, com);
}

// determine_extracted_voxel_object_dynamics (interaction.rs:503-585): moments = the fragment's inertial property manager in the PARENT's grid
// frame (what the property transferrer filled); on return they are about the fragment's own grid origin
void orc_extracted_object_dynamics(float moments[10], const int origin_offset_in_parent[3], float voxel_extent, const float original_local_com[3],
                                   const orc_rigid_body* parent, orc_rigid_body* fragment, float new_local_com[3]) {
    const V3 pos = ld3(parent->position);
    const Quat q = ldq(parent->orientation);
    const V3 vel = body_velocity(*parent);
    const AngVel av = body_angular_velocity(*parent);
    const V3 local_disp = div_recip(v3(moments[1], moments[2], moments[3]), moments[0]) - ld3(original_local_com);
    const V3 world_disp = qrot(q, local_disp);
    const V3 dv = cross(angvel_vector(av), world_disp);
    const float off[3] = {(float)origin_offset_in_parent[0] * voxel_extent, (float)origin_offset_in_parent[1] * voxel_extent,
                          (float)origin_offset_in_parent[2] * voxel_extent};
    orc_offset_reference_point(moments, off);
    float ip[22];
    orc_derive_inertial_properties(moments, ip);  // object/inertia.rs:288-326
    std::memset(fragment, 0, sizeof(*fragment));
    fragment->mass = ip[0];  // DynamicRigidBody::new (rigid_body.rs:413-441)
    std::memcpy(fragment->inertia, ip + 4, 36);
    std::memcpy(fragment->inv_inertia, ip + 13, 36);
    st3(fragment->position, pos + world_disp);
    stq(fragment->orientation, q);
    st3(fragment->momentum, (vel + dv) * fragment->mass);
    st3(fragment->angular_momentum, mul(rotated(ldm(fragment->inertia), q), angvel_vector(av)));
    st3(new_local_com, v3(ip[1], ip[2], ip[3]));
}

// collision/collidable/sphere.rs:105-160. Returns 1 and fills position/normal/depth on contact.
int orc_sphere_sphere_contact(const float ca[3], float ra, const float cb[3], float rb, float position[3], float normal[3], float* depth) {
    V3 d = ld3(ca) - ld3(cb);
    float d2 = dot(d, d), maxd = ra + rb;
    if (d2 > maxd * maxd) return 0;
    float dist = std::sqrt(d2);
    V3 n = dist > 1e-8f ? div_recip(d, dist) : v3(0, 0, 1);
    st3(position, ld3(cb) + n * rb);
    st3(normal, n);
    *depth = fmax_rs(0.0f, maxd - dist);
    return 1;
}
int orc_sphere_plane_contact(const float c[3], float r, const float plane_normal[3], float plane_displacement, float position[3],
                             float normal[3], float* depth) {
    V3 n = ld3(plane_normal);
    float sd = dot(n, ld3(c)) - plane_displacement;  // impact_geometry plane.rs:123-125
    float pen = r - sd;
    if (pen < 0.0f) return 0;
    st3(position, ld3(c) - n * sd);
    st3(normal, n);
    *depth = pen;
    return 1;
}

// ConstraintManager::prepare_constraints over an explicit contact list (constraint.rs:193-287). Contacts
// with flag bit 0 start a new manifold (one collision); a manifold judged interlocked is replaced by its
// separating contact (contact.rs:610-689). Returns the number of prepared contacts.
int orc_physics_prepare(orc_physics* w, const orc_contact* contacts, int n) {
    w->bodies.clear();
    w->body_ref.clear();
    w->body_index.clear();
    int i = 0;
    while (i < n) {
        int j = i + 1;
        while (j < n && !(contacts[j].flags & 1u)) ++j;
        const orc_contact* m = contacts + i;
        const int cnt = j - i;
        uint32_t ia = add_body(w, m[0].body_a), ib = add_body(w, m[0].body_b);
        orc_contact sep;
        if (interlocked(m, cnt) && separating_contact(w->bodies[ia], w->bodies[ib], m, cnt, sep)) {
            register_contact(w, sep, ia, ib);
        } else {
            for (int k = 0; k < cnt; ++k) register_contact(w, m[k], ia, ib);
        }
        i = j;
    }
    // the spherical joints (constraint.rs:252-255 -> solver.rs:182-215): prepare_spherical_joint makes the pair constrained bodies and
    // registers a PreparedSphericalJoint whose impulses and corrections are all zero / empty (constraint/spherical_joint.rs:62-88):
    // nothing to solve, but the bodies take part in the velocity synchronisation and write-back
    for (size_t j = 0; j + 1 < w->joint_refs.size(); j += 2) {
        add_body(w, w->joint_refs[j]);
        add_body(w, w->joint_refs[j + 1]);
    }
    // remove_unprepared_constraints_and_reset_flags (solver.rs:434-452)
    size_t idx = 0, len = w->cache.size();
    while (idx < len) {
        if (w->cache[idx].was_prepared) {
            w->cache[idx].was_prepared = false;
            ++idx;
        } else {
            w->index_of.erase(w->keys[idx]);
            w->cache[idx] = w->cache[len - 1];
            w->keys[idx] = w->keys[len - 1];
            w->cache.pop_back();
            w->keys.pop_back();
            --len;
            if (idx < len) w->index_of[w->keys[idx]] = (uint32_t)idx;
        }
    }
    return (int)w->cache.size();
}
void orc_physics_set_joints(orc_physics* w, const uint32_t* body_pairs, int n_joints) { w->joint_refs.assign(body_pairs, body_pairs + 2 * (size_t)(n_joints > 0 ? n_joints : 0)); }
int orc_physics_prepared_body_count(const orc_physics* w) { return (int)w->bodies.size(); }
// ids of the cached contacts in solve order
void orc_physics_contact_order(const orc_physics* w, uint64_t* ids) { std::memcpy(ids, w->keys.data(), w->keys.size() * 8); }
void orc_physics_accumulated_impulses(const orc_physics* w, float* out3n) {
    for (size_t i = 0; i < w->cache.size(); ++i) {
        out3n[3 * i] = w->cache[i].acc.n;
        out3n[3 * i + 1] = w->cache[i].acc.t;
        out3n[3 * i + 2] = w->cache[i].acc.b;
    }
}

// rigid_body.rs:373-378, 708-721
void orc_physics_advance_momenta(orc_physics* w, float dt) {
    for (auto& b : w->dyn) {
        st3(b.momentum, ld3(b.momentum) + ld3(b.total_force) * dt);
        st3(b.angular_momentum, ld3(b.angular_momentum) + ld3(b.total_torque) * dt);
    }
}

// compute_and_apply_constrained_state (constraint.rs:296-309)
void orc_physics_solve(orc_physics* w) {
    // synchronize_prepared_constrained_body_velocities (solver.rs:217-228, 543-569)
    for (size_t i = 0; i < w->bodies.size(); ++i) {
        uint32_t ref = w->body_ref[i];
        if (ref & 0x80000000u) {
            const orc_kinematic_body& k = w->kin[ref & 0x7FFFFFFFu];
            w->bodies[i].velocity = ld3(k.velocity);
            w->bodies[i].angular_velocity = ld3(k.angular_axis) * k.angular_speed;
        } else {
            w->bodies[i].velocity = body_velocity(w->dyn[ref]);
            w->bodies[i].angular_velocity = angvel_vector(body_angular_velocity(w->dyn[ref]));
        }
    }
    // compute_constrained_velocities (solver.rs:242-262, 481-528)
    for (auto& e : w->cache) apply_impulses(e.p, w->bodies[e.a], w->bodies[e.b], e.acc);
    for (uint32_t it = 0; it < w->cfg.n_iterations; ++it)
        for (auto& e : w->cache) {
            CBody &a = w->bodies[e.a], &b = w->bodies[e.b];
            Impulses c = compute_impulses(e.p, a, b);
            Impulses nw = clamp_impulses(e.p, {e.acc.n + c.n, e.acc.t + c.t, e.acc.b + c.b});
            Impulses d = {nw.n - e.acc.n, nw.t - e.acc.t, nw.b - e.acc.b};
            e.acc = nw;
            apply_impulses(e.p, a, b, d);
        }
    // compute_corrected_configurations (solver.rs:276-289, 530-541)
    for (uint32_t it = 0; it < w->cfg.n_positional_correction_iterations; ++it)
        for (auto& e : w->cache) positional_correction(e.p, w->bodies[e.a], w->bodies[e.b], w->cfg.positional_correction_factor);
    // apply_constrained_velocities_and_corrected_configurations (solver.rs:571-602)
    for (size_t i = 0; i < w->bodies.size(); ++i) {
        uint32_t ref = w->body_ref[i];
        const CBody& c = w->bodies[i];
        if (ref & 0x80000000u) {
            orc_kinematic_body& k = w->kin[ref & 0x7FFFFFFFu];
            st3(k.position, c.position);
            stq(k.orientation, c.orientation);
            st3(k.velocity, c.velocity);
            AngVel av = angvel_from_vector(c.angular_velocity);
            st3(k.angular_axis, av.axis);
            k.angular_speed = av.speed;
        } else {
            orc_rigid_body& b = w->dyn[ref];
            st3(b.position, c.position);
            stq(b.orientation, c.orientation);
            st3(b.momentum, c.velocity * b.mass);
            AngVel av = angvel_from_vector(c.angular_velocity);
            st3(b.angular_momentum, mul(rotated(ldm(b.inertia), ldq(b.orientation)), angvel_vector(av)));
        }
    }
}

// rigid_body.rs:381-395, 723-742, 1013-1034 (+ kinematic bodies, 864-878)
void orc_physics_advance_configurations(orc_physics* w, float dt) {
    for (auto& b : w->dyn) {
        st3(b.position, ld3(b.position) + body_velocity(b) * dt);
        AngVel av = body_angular_velocity(b);
        float angle = av.speed * dt;
        float s = std::sin(0.5f * angle), c = std::cos(0.5f * angle);
        V3 im = av.axis * s;
        stq(b.orientation, qnormalize(qmul(Quat{im.x, im.y, im.z, c}, ldq(b.orientation))));
    }
    for (auto& k : w->kin) {
        st3(k.position, ld3(k.position) + ld3(k.velocity) * dt);
        float angle = k.angular_speed * dt;
        float s = std::sin(0.5f * angle), c = std::cos(0.5f * angle);
        V3 im = ld3(k.angular_axis) * s;
        stq(k.orientation, qnormalize(qmul(Quat{im.x, im.y, im.z, c}, ldq(k.orientation))));
    }
}

// perform_physics_step (lib.rs:31-110) for an explicit contact list; force generators are out of scope
// (total_force / total_torque are inputs and stay as they are).
int orc_physics_step(orc_physics* w, const orc_contact* contacts, int n, float dt) {
    int prepared = orc_physics_prepare(w, contacts, n);
    orc_physics_advance_momenta(w, dt);
    orc_physics_solve(w);
    orc_physics_advance_configurations(w, dt);
    return prepared;
}

}  // extern "C"
