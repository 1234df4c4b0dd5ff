// a15-a19 — rigid-body step and sequential-impulses contact solve on the GPU.
//
// Reference behaviour reproduced (engine/crates/impact_physics/src):
//   ConstrainedBody::from_dynamic/kinematic_rigid_body      constraint.rs:476-523
//   Contact::prepare (effective masses, tangents, targets)  constraint/contact.rs:233-310, 788-832
//   can_use_warm_impulses_from + old_impulse_weight         contact.rs:313-327, solver.rs:406-432
//   advance_momentum / angular momentum                     rigid_body.rs:373-378, 708-721
//   synchronize_prepared_constrained_body_velocities        solver.rs:217-228, 543-569
//   warm impulses, sequential sweeps, clamp, apply          solver.rs:242-262, 481-528; contact.rs:329-438
//   positional correction                                   solver.rs:276-289; contact.rs:440-517, 835-843
//   write-back (synchronize_momentum / angular momentum)    solver.rs:571-602; rigid_body.rs:687-702
//   advance_position / advance_orientation                  rigid_body.rs:723-742, 1013-1034
//
// The reference's solver is a Gauss–Seidel sweep whose result depends on the order of the contacts.
// That order is kept EXACTLY: the host turns the sequence (warm pass, n velocity sweeps | m positional
// sweeps) x (contacts in cache order) into dependency levels — an item's level is one more than the
// latest earlier item that touches one of its two dynamic bodies — and this kernel executes one level
// at a time with all items of a level in parallel. Items of a level share no dynamic body, so the
// result is what the sequential loop produces, operation for operation; successive sweeps overlap
// wherever the contact graph allows (a wavefront), which is where the parallelism comes from.
// Kinematic bodies are never changed by an impulse (inverse mass 0), so they are read-only here and
// create no dependencies.
//
// One workgroup of 1024 threads walks the levels (a level is ~10^2 contacts; the chain of levels is
// the critical path, so more workgroups would only add grid-wide barriers). f32 throughout, no FMA
// contraction, IEEE sqrt/div — same operation order as the oracle.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "ivx_internal.hpp"
#include "physics_internal.hpp"

namespace {

struct V3 {
    float x, y, z;
};
struct Q4 {
    float x, y, z, w;
};
struct M3 {
    V3 c0, c1, c2;
};
__device__ __forceinline__ V3 mk(float x, float y, float z) { return V3{x, y, z}; }
__device__ __forceinline__ V3 ld3(const float* p) { return V3{p[0], p[1], p[2]}; }
__device__ __forceinline__ void st3(float* p, V3 v) {
    p[0] = v.x;
    p[1] = v.y;
    p[2] = v.z;
}
__device__ __forceinline__ Q4 ldq(const float* p) { return Q4{p[0], p[1], p[2], p[3]}; }
__device__ __forceinline__ void stq(float* p, Q4 q) {
    p[0] = q.x;
    p[1] = q.y;
    p[2] = q.z;
    p[3] = q.w;
}
__device__ __forceinline__ M3 ldm(const float* p) { return M3{ld3(p), ld3(p + 3), ld3(p + 6)}; }
__device__ __forceinline__ void stm(float* p, const M3& m) {
    st3(p, m.c0);
    st3(p + 3, m.c1);
    st3(p + 6, m.c2);
}
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
__device__ __forceinline__ V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) { return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
__device__ __forceinline__ V3 div_elem(V3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
__device__ __forceinline__ V3 div_recip(V3 a, float s) {
    const float r = 1.0f / s;
    return {a.x * r, a.y * r, a.z * r};
}
__device__ __forceinline__ V3 mul(const M3& m, V3 v) { return (m.c0 * v.x + m.c1 * v.y) + m.c2 * v.z; }
__device__ __forceinline__ M3 transpose(const M3& m) {
    return {{m.c0.x, m.c1.x, m.c2.x}, {m.c0.y, m.c1.y, m.c2.y}, {m.c0.z, m.c1.z, m.c2.z}};
}
__device__ __forceinline__ M3 mul(const M3& a, const M3& b) { return {mul(a, b.c0), mul(a, b.c1), mul(a, b.c2)}; }
// glam Mat3A::from_quat
__device__ __forceinline__ M3 m3_from_quat(Q4 q) {
    const float x2 = q.x + q.x, y2 = q.y + q.y, z2 = q.z + q.z;
    const float xx = q.x * x2, xy = q.x * y2, xz = q.x * z2;
    const float yy = q.y * y2, yz = q.y * z2, zz = q.z * z2;
    const float wx = q.w * x2, wy = q.w * y2, wz = q.w * z2;
    return {{1.0f - (yy + zz), xy + wz, xz - wy}, {xy - wz, 1.0f - (xx + zz), yz + wx}, {xz + wy, yz - wx, 1.0f - (xx + yy)}};
}
__device__ __forceinline__ M3 rotated(const M3& m, Q4 q) {  // R M R^T (inertia.rs:401-404, 429-432)
    const M3 r = m3_from_quat(q);
    return mul(mul(r, m), transpose(r));
}
__device__ __forceinline__ Q4 conj(Q4 q) { return {-q.x, -q.y, -q.z, q.w}; }
__device__ __forceinline__ Q4 qmul(Q4 a, Q4 b) {
    return {a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y, a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x,
            a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z};
}
__device__ __forceinline__ Q4 qnormalize(Q4 q) {
    const float l = sqrtf(((q.x * q.x + q.y * q.y) + q.z * q.z) + q.w * q.w);
    return {q.x / l, q.y / l, q.z / l, q.w / l};
}
// glam Quat::mul_vec3a
__device__ __forceinline__ V3 qrot(Q4 q, V3 v) {
    const V3 b = mk(q.x, q.y, q.z);
    const float b2 = dot(b, b);
    return (v * (q.w * q.w - b2) + b * (dot(v, b) * 2.0f)) + cross(b, v) * (q.w * 2.0f);
}
__device__ __forceinline__ float max_rs(float a, float b) { return (b > a) ? b : a; }
// f32::sin / f32::cos of the reference are libm's sinf / cosf, which are the correctly rounded values in all but rare cases; the device
// library's single-precision versions are one ulp off now and then (found by a random contact graph whose 1-ulp orientation grew past the
// 1e-5 bar in two frames). The double-precision functions rounded once agree with libm on every value the tests have met; it is two calls
// per body and step.
__device__ __forceinline__ float sin_rn(float x) { return (float)sin((double)x); }
__device__ __forceinline__ float cos_rn(float x) { return (float)cos((double)x); }

// AngularVelocity::from_vector (quantities.rs:160-172): unit axis + speed, zero at or below f32::EPSILON
struct AngVel {
    V3 axis;
    float speed;
};
__device__ __forceinline__ AngVel angvel_from_vector(V3 w) {
    const float n2 = dot(w, w);
    const float eps = 1.1920929e-07f;
    if (n2 > eps * eps) {
        const float n = sqrtf(n2);
        return {div_elem(w, n), n};
    }
    return {mk(0.0f, 1.0f, 0.0f), 0.0f};
}
__device__ __forceinline__ V3 angvel_vector(AngVel a) { return a.axis * a.speed; }

__device__ __forceinline__ V3 body_velocity(const ivx_rigid_body& b) { return div_recip(ld3(b.momentum), b.mass); }
__device__ __forceinline__ AngVel body_angular_velocity(const ivx_rigid_body& b) {
    return angvel_from_vector(mul(rotated(ldm(b.inv_inertia), ldq(b.orientation)), ld3(b.angular_momentum)));
}

// ---- per-body kernels ------------------------------------------------------------------------
__device__ __forceinline__ void prepare_body(uint32_t i, uint32_t n_dyn, uint32_t n_kin, const ivx_rigid_body* dyn, const ivx_kinematic_body* kin,
                                             PhysBody* cb, uint8_t* touched) {
    if (i >= n_dyn + n_kin) return;
    PhysBody c;
    if (i < n_dyn) {
        const ivx_rigid_body b = dyn[i];
        c.inv_mass = 1.0f / b.mass;
        stm(c.inv_inertia, rotated(ldm(b.inv_inertia), ldq(b.orientation)));
        st3(c.pos, ld3(b.position));
        stq(c.q, ldq(b.orientation));
        st3(c.v, body_velocity(b));
        st3(c.w, angvel_vector(body_angular_velocity(b)));
    } else {
        const ivx_kinematic_body k = kin[i - n_dyn];
        c.inv_mass = 0.0f;
        for (int e = 0; e < 9; ++e) c.inv_inertia[e] = 0.0f;
        st3(c.pos, ld3(k.position));
        stq(c.q, ldq(k.orientation));
        st3(c.v, ld3(k.velocity));
        st3(c.w, ld3(k.angular_axis) * k.angular_speed);
    }
    touched[i] = 0;
    c.pad = 0.0f;  // (a kinematic body: the count of positional corrections applied to it, see post_solve_body)
    cb[i] = c;
}
__global__ __launch_bounds__(256) void k_prepare_bodies(uint32_t n_dyn, uint32_t n_kin, const ivx_rigid_body* __restrict__ dyn,
                                                        const ivx_kinematic_body* __restrict__ kin, PhysBody* __restrict__ cb,
                                                        uint8_t* __restrict__ touched) {
    prepare_body(blockIdx.x * 256u + threadIdx.x, n_dyn, n_kin, dyn, kin, cb, touched);
}

__device__ __forceinline__ V3 to_world(const float* pos, const float* q, V3 p) { return qrot(ldq(q), p) + ld3(pos); }
__device__ __forceinline__ V3 to_body(const float* pos, const float* q, V3 p) { return qrot(conj(ldq(q)), p - ld3(pos)); }
__device__ __forceinline__ V3 point_velocity(V3 v, V3 w, V3 disp) { return v + cross(w, disp); }
__device__ __forceinline__ float effective_mass(float ima, const M3& iia, float imb, const M3& iib, V3 da, V3 db, V3 dir) {
    const V3 ca = cross(da, dir), cb = cross(db, dir);
    return 1.0f / (((ima + imb) + dot(ca, mul(iia, ca))) + dot(cb, mul(iib, cb)));
}

__global__ __launch_bounds__(256) void k_prepare_contacts(uint32_t n, uint32_t n_dyn, const ivx_contact* __restrict__ contacts,
                                                          const int32_t* __restrict__ prev_slot, const PhysBody* __restrict__ cb,
                                                          const PhysContact* __restrict__ prev_pc, const float4* __restrict__ prev_acc,
                                                          uint32_t n_prev, float old_impulse_weight, PhysContact* __restrict__ pc,
                                                          float4* __restrict__ acc, uint8_t* __restrict__ touched) {
    const uint32_t s = blockIdx.x * 256u + threadIdx.x;
    if (s >= n) return;
    const ivx_contact c = contacts[s];
    const uint32_t ia = (c.body_a & IVX_KINEMATIC_BODY) ? n_dyn + (c.body_a & 0x7FFFFFFFu) : c.body_a;
    const uint32_t ib = (c.body_b & IVX_KINEMATIC_BODY) ? n_dyn + (c.body_b & 0x7FFFFFFFu) : c.body_b;
    const PhysBody a = cb[ia], b = cb[ib];
    const V3 pos = ld3(c.position), normal = ld3(c.normal);
    PhysContact p;
    st3(p.local_a, to_body(a.pos, a.q, pos - normal * c.depth));
    st3(p.local_b, to_body(b.pos, b.q, pos));
    const V3 da = pos - ld3(a.pos), db = pos - ld3(b.pos);
    // construct_tangent_vectors (contact.rs:813-832)
    const V3 traw = fabsf(normal.x) < 0.57735f ? mk(0.0f, normal.z, -normal.y) : mk(normal.y, -normal.x, 0.0f);
    const V3 t1 = div_elem(traw, sqrtf(dot(traw, traw)));
    const V3 t2 = cross(normal, t1);
    st3(p.normal, normal);
    st3(p.tangent, t1);
    st3(p.bitangent, t2);
    const M3 iia = ldm(a.inv_inertia), iib = ldm(b.inv_inertia);
    p.m_n = effective_mass(a.inv_mass, iia, b.inv_mass, iib, da, db, normal);
    p.m_t = effective_mass(a.inv_mass, iia, b.inv_mass, iib, da, db, t1);
    p.m_b = effective_mass(a.inv_mass, iia, b.inv_mass, iib, da, db, t2);
    const V3 rel = point_velocity(ld3(a.v), ld3(a.w), da) - point_velocity(ld3(b.v), ld3(b.w), db);
    const float sep = dot(normal, rel);
    p.target = fabsf(sep) >= 0.4f ? -c.restitution * sep : 0.0f;
    const float d1 = dot(rel, t1), d2 = dot(rel, t2);
    p.friction = (d1 * d1 + d2 * d2) >= 1e-4f ? c.dynamic_friction : c.static_friction;
    st3(p.world_b, qrot(ldq(b.q), ld3(p.local_b)) + ld3(b.pos));  // (the very operations the velocity sweeps used to repeat per contact and sweep)
    p.pad = 0;
    float4 a4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    const int32_t ps = prev_slot ? prev_slot[s] : (s < n_prev ? (int32_t)s : -1);
    if (ps >= 0) {
        const PhysContact o = prev_pc[ps];
        if (dot(normal, ld3(o.normal)) > 1.0f - 1e-2f && dot(t1, ld3(o.tangent)) > 1.0f - 1e-2f) {
            const float4 oa = prev_acc[ps];
            a4 = make_float4(oa.x * old_impulse_weight, oa.y * old_impulse_weight, oa.z * old_impulse_weight, 0.0f);
        }
    }
    pc[s] = p;
    acc[s] = a4;
    touched[ia] = 1;  // (kinematic bodies too: the reference writes every constrained body back after the solve, post_solve_body)
    touched[ib] = 1;
}

__device__ __forceinline__ void pre_solve_body(uint32_t i, uint32_t n_dyn, float dt, ivx_rigid_body* dyn, PhysBody* cb) {
    if (i >= n_dyn) return;
    ivx_rigid_body b = dyn[i];
    st3(b.momentum, ld3(b.momentum) + ld3(b.total_force) * dt);
    st3(b.angular_momentum, ld3(b.angular_momentum) + ld3(b.total_torque) * dt);
    st3(dyn[i].momentum, ld3(b.momentum));
    st3(dyn[i].angular_momentum, ld3(b.angular_momentum));
    // the constrained copy keeps the configuration and world inverse inertia from prepare time
    st3(cb[i].v, body_velocity(b));
    st3(cb[i].w, angvel_vector(body_angular_velocity(b)));
}
__global__ __launch_bounds__(256) void k_pre_solve(uint32_t n_dyn, float dt, ivx_rigid_body* __restrict__ dyn, PhysBody* __restrict__ cb) {
    pre_solve_body(blockIdx.x * 256u + threadIdx.x, n_dyn, dt, dyn, cb);
}

__device__ __forceinline__ void post_solve_body(uint32_t i, uint32_t n_dyn, uint32_t n_kin, float dt, int write_back, int advance, const PhysBody* cb,
                                                const uint8_t* touched, ivx_rigid_body* dyn, ivx_kinematic_body* kin) {
    if (i >= n_dyn + n_kin) return;
    if (i < n_dyn) {
        ivx_rigid_body b = dyn[i];
        if (write_back && touched[i]) {
            const PhysBody c = cb[i];
            st3(b.position, ld3(c.pos));
            stq(b.orientation, ldq(c.q));
            st3(b.momentum, ld3(c.v) * b.mass);
            st3(b.angular_momentum, mul(rotated(ldm(b.inertia), ldq(b.orientation)), angvel_vector(angvel_from_vector(ld3(c.w)))));
        }
        if (advance) {
            st3(b.position, ld3(b.position) + body_velocity(b) * dt);
            const AngVel av = body_angular_velocity(b);
            const float angle = av.speed * dt;
            const float s = sin_rn(0.5f * angle), co = cos_rn(0.5f * angle);
            const V3 im = av.axis * s;
            stq(b.orientation, qnormalize(qmul(Q4{im.x, im.y, im.z, co}, ldq(b.orientation))));
        }
        dyn[i] = b;
    } else if (advance || (write_back && touched[i])) {
        ivx_kinematic_body k = kin[i - n_dyn];
        if (write_back && touched[i]) {
            // apply_constrained_velocities_and_corrected_configurations (solver.rs:571-602) writes kinematic constrained bodies back too.
            // Impulses and corrections leave them where they were (zero inverse mass and inertia), but two things change in the last
            // bits: the angular velocity goes through vector form (axis * speed -> AngularVelocity::from_vector), and the orientation
            // has been re-normalised once per positional correction applied to the body (ReplayView; a normalised quaternion is a
            // fixed point of the re-normalisation only two times in three).
            const AngVel av = angvel_from_vector(ld3(k.angular_axis) * k.angular_speed);
            st3(k.angular_axis, av.axis);
            k.angular_speed = av.speed;
            stq(k.orientation, ldq(cb[i].q));  // (k_kin_prefix left the re-normalised orientation there; untouched otherwise)
        }
        if (!advance) {
            kin[i - n_dyn] = k;
            return;
        }
        st3(k.position, ld3(k.position) + ld3(k.velocity) * dt);
        const float angle = k.angular_speed * dt;
        const float s = sin_rn(0.5f * angle), co = cos_rn(0.5f * angle);
        const V3 im = ld3(k.angular_axis) * s;
        stq(k.orientation, qnormalize(qmul(Q4{im.x, im.y, im.z, co}, ldq(k.orientation))));
        kin[i - n_dyn] = k;
    }
}
__global__ __launch_bounds__(256) void k_post_solve(uint32_t n_dyn, uint32_t n_kin, float dt, int write_back, int advance,
                                                    const PhysBody* __restrict__ cb, const uint8_t* __restrict__ touched,
                                                    ivx_rigid_body* __restrict__ dyn, ivx_kinematic_body* __restrict__ kin) {
    post_solve_body(blockIdx.x * 256u + threadIdx.x, n_dyn, n_kin, dt, write_back, advance, cb, touched, dyn, kin);
}

// A step without contacts (a free body, e.g. a voxel object's own rigid body between collisions) is three element-wise passes
// over the bodies: one launch runs them back to back per body (each thread reads back only what it wrote itself).
__global__ __launch_bounds__(256) void k_free_step(uint32_t n_dyn, uint32_t n_kin, float dt, ivx_rigid_body* dyn, ivx_kinematic_body* kin, PhysBody* cb,
                                                   uint8_t* touched) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    prepare_body(i, n_dyn, n_kin, dyn, kin, cb, touched);
    pre_solve_body(i, n_dyn, dt, dyn, cb);
    post_solve_body(i, n_dyn, n_kin, dt, 1, 1, cb, touched, dyn, kin);
}

// ---- the solve -----------------------------------------------------------------------------------
// An item is a CHAIN: up to 15 contacts that follow each other in the solve order and act on the same pair of bodies
// (the contact points of one manifold). No other constraint can come between them, so one thread runs them back to back
// with the pair's state in registers — a quarter of the levels for four-point manifolds, and no barrier inside a manifold.
// item word: bits 0-23 first contact slot, 24-27 chain length, 28-31 type.
struct PairState {  // the part of the two bodies a phase changes
    V3 va, wa, vb, wb;  // velocity phase
    V3 pa, pb;          // positional phase
    Q4 qa, qb;
    uint32_t applied = 0u;  // positional phase: corrections this chain applied (ReplayView)
};

__device__ __forceinline__ Q4 pseudo_advanced(Q4 q, V3 w) {  // contact.rs:835-843, quantities.rs:372-378
    const V3 h = w * 0.5f;
    const Q4 d = qmul(Q4{h.x, h.y, h.z, 0.0f}, q);
    return qnormalize(Q4{q.x + d.x, q.y + d.y, q.z + d.z, q.w + d.w});
}

struct PairStatic {
    float ima, imb;
    M3 iia, iib;
    V3 pos_a, pos_b;  // velocity phase only (configuration is fixed there)
    bool dyn_a, dyn_b;
};

// The positional phase and kinematic bodies. A kinematic body takes no correction (zero inverse mass and inertia), but the reference runs the
// same arithmetic on it: every correction applied to a pair re-normalises the kinematic partner's orientation (pseudo_advanced with a zero
// rotation), the corrections that follow — anywhere in the sweep order — see the result, and the body is written back after the solve
// (solver.rs:571-602). A normalised f32 quaternion is a fixed point of that only two times in three. The schedule keeps kinematic bodies out
// of the dependency levels, so their orientation during the phase is REPLAYED: pass 1 runs with the orientation the phase starts from and
// records how many corrections every chain applied; k_kin_prefix turns that, per kinematic body and in solve order, into the number of
// re-normalisations each chain starts from, and tabulates the orientation after 0, 1, 2 ... of them; where a body's orientation actually
// moves, pass 2 runs the phase again from the saved state with every chain starting from its own table entry. (Inside a chain the orientation
// evolves in the thread, by the reference's own arithmetic on the zeros.)
struct ReplayView {
    const float4* qstart = nullptr;  // [2 * item + side]: the orientation a kinematic body a / b has when the chain starts (pass 2); null in pass 1
    uint32_t* applied = nullptr;     // [item]: corrections the chain applied (written in pass 1 for chains with a kinematic body)
};

// a prepared contact as the velocity phase uses it: the pair's lever arms to the contact point (world_b - position: the configuration is
// fixed during the phase) in place of the point itself
struct ContactV {
    V3 n, t, b;
    float m_n, m_t, m_b, friction, target;
    V3 da, db;
};
__device__ __forceinline__ ContactV contact_v(const PhysContact& p, V3 pos_a, V3 pos_b) {
    const V3 pb = ld3(p.world_b);
    return ContactV{ld3(p.normal), ld3(p.tangent), ld3(p.bitangent), p.m_n, p.m_t, p.m_b, p.friction, p.target, pb - pos_a, pb - pos_b};
}
__device__ __forceinline__ void apply_pair(const ContactV& c, const PairStatic& st, PairState& x, float in, float it, float ib_) {
    const V3 dp = (c.n * in + c.t * it) + c.b * ib_;
    if (st.dyn_a) {
        x.va = x.va + dp * st.ima;
        x.wa = x.wa + mul(st.iia, cross(c.da, dp));
    }
    if (st.dyn_b) {
        x.vb = x.vb - dp * st.imb;
        x.wb = x.wb - mul(st.iib, cross(c.db, dp));
    }
}
// one contact of a warm-start or velocity item. `acc` in: the accumulated impulses of the contact; out: what a velocity item leaves there
__device__ __forceinline__ void run_contact_v(uint32_t type, const ContactV& c, const PairStatic& st, PairState& x, float4& acc) {
    if (type == PHYS_ITEM_WARM) {
        apply_pair(c, st, x, acc.x, acc.y, acc.z);
        return;
    }
    // compute_impulses -> clamp -> apply the difference (solver.rs:496-528)
    const V3 rel = point_velocity(x.va, x.wa, c.da) - point_velocity(x.vb, x.wb, c.db);
    const float sep = dot(c.n, rel);
    const float cn = -c.m_n * (sep - c.target), ct = -c.m_t * dot(c.t, rel), cbi = -c.m_b * dot(c.b, rel);
    const float un = acc.x + cn, ut = acc.y + ct, ub = acc.z + cbi;
    const float nn = max_rs(0.0f, un);
    const float max_t = c.friction * nn;
    const float mag = sqrtf(ut * ut + ub * ub);
    const float sc = mag > max_t ? max_t / mag : 1.0f;
    const float nt = ut * sc, nb = ub * sc;
    apply_pair(c, st, x, nn - acc.x, nt - acc.y, nb - acc.z);
    acc = make_float4(nn, nt, nb, 0.0f);
}
// one contact of a positional item
__device__ __forceinline__ void run_contact_p(V3 n, V3 local_a, V3 local_b, const PairStatic& st, PairState& x, float factor) {
    const V3 pa = qrot(x.qa, local_a) + x.pa, pb = qrot(x.qb, local_b) + x.pb;
    const float depth = dot(n, pb - pa);
    if (depth <= 0.0f) return;
    const V3 da = pb - x.pa, db = pb - x.pb;
    const float m = effective_mass(st.ima, st.iia, st.imb, st.iib, da, db, n);
    const V3 dp = n * (m * factor * depth);
    // (both bodies, dynamic or not: on a kinematic body's zeros this leaves the position and re-normalises the orientation, as in the
    // reference; only dynamic bodies are ever stored)
    x.pa = x.pa + dp * st.ima;
    x.qa = pseudo_advanced(x.qa, mul(st.iia, cross(da, dp)));
    x.pb = x.pb + dp * (-st.imb);
    x.qb = pseudo_advanced(x.qb, mul(M3{-st.iib.c0, -st.iib.c1, -st.iib.c2}, cross(db, dp)));
    x.applied += 1u;
}

// one contact of a chain; `type` is uniform over the chain
// `acc` in: the accumulated impulses of the contact; out: what a velocity item leaves there (the caller stores it — after the whole chain: on this
// part a store counts in vmcnt like a load, and waiting for the next contact's data would also wait for the store's acknowledgement, ~1.5 us each)
__device__ __forceinline__ void run_contact(uint32_t type, const PhysContact& p, const PairStatic& st, PairState& x, float factor, float4& acc) {
    if (type == PHYS_ITEM_POSITIONAL) run_contact_p(ld3(p.normal), ld3(p.local_a), ld3(p.local_b), st, x, factor);
    else run_contact_v(type, contact_v(p, st.pos_a, st.pos_b), st, x, acc);
}

// LDS = true: the phase's mutable state of the dynamic bodies lives in s_dyn (velocity: 6 floats v,w; positional: 7 floats
// position, orientation); false: in the PhysBody array in HBM. The prepared contacts (and accumulated impulses) of a chain are
// fetched four at a time so that their HBM latencies overlap each other and the loads of the pair's static data.
template <bool LDS, int PHASE>
__device__ __forceinline__ void run_chain(uint32_t item, uint2 bodies, uint32_t n_dyn, float factor, const PhysContact* __restrict__ pcs,
                                          float4* __restrict__ accs, PhysBody* __restrict__ cb, float* s_dyn, uint32_t item_index, const ReplayView& rv) {
    const uint32_t s0 = item & 0x00FFFFFFu, len = (item >> 24) & 15u, type = item >> 28;
    const uint32_t ia = bodies.x, ib = bodies.y;
    // the prepared contacts (and accumulated impulses) come two at a time: contact c + 2 is requested as soon as contact c has been consumed, so
    // its latency hides behind contact c + 1 — and the kernel keeps to the VGPR budget of 768 threads per workgroup (3 waves per SIMD)
    PhysContact pa, pb2;
    float4 aca = make_float4(0.0f, 0.0f, 0.0f, 0.0f), acb = aca;
    const bool with_acc = type != PHYS_ITEM_POSITIONAL, store_acc = type == PHYS_ITEM_VELOCITY;
    pa = pcs[s0];
    if (with_acc) aca = accs[s0];
    if (len > 1u) {
        pb2 = pcs[s0 + 1u];
        if (with_acc) acb = accs[s0 + 1u];
    }
    const PhysBody& A = cb[ia];
    const PhysBody& B = cb[ib];
    PairStatic st;
    st.ima = A.inv_mass;
    st.imb = B.inv_mass;
    st.iia = ldm(A.inv_inertia);
    st.iib = ldm(B.inv_inertia);
    st.dyn_a = ia < n_dyn;
    st.dyn_b = ib < n_dyn;
    PairState x;
    if (PHASE == 0) {
        st.pos_a = ld3(A.pos);
        st.pos_b = ld3(B.pos);
        if (LDS && st.dyn_a) {
            x.va = ld3(s_dyn + 6 * ia);
            x.wa = ld3(s_dyn + 6 * ia + 3);
        } else {
            x.va = ld3(A.v);
            x.wa = ld3(A.w);
        }
        if (LDS && st.dyn_b) {
            x.vb = ld3(s_dyn + 6 * ib);
            x.wb = ld3(s_dyn + 6 * ib + 3);
        } else {
            x.vb = ld3(B.v);
            x.wb = ld3(B.w);
        }
    } else {
        if (LDS && st.dyn_a) {
            x.pa = ld3(s_dyn + 7 * ia);
            x.qa = ldq(s_dyn + 7 * ia + 3);
        } else {
            x.pa = ld3(A.pos);
            x.qa = ldq(A.q);
        }
        if (LDS && st.dyn_b) {
            x.pb = ld3(s_dyn + 7 * ib);
            x.qb = ldq(s_dyn + 7 * ib + 3);
        } else {
            x.pb = ld3(B.pos);
            x.qb = ldq(B.q);
        }
        if (rv.qstart) {  // pass 2: a kinematic body's orientation as the chains before this one left it
            if (!st.dyn_a) {
                const float4 t = rv.qstart[2u * item_index];
                x.qa = Q4{t.x, t.y, t.z, t.w};
            }
            if (!st.dyn_b) {
                const float4 t = rv.qstart[2u * item_index + 1u];
                x.qb = Q4{t.x, t.y, t.z, t.w};
            }
        }
    }
    run_contact(type, pa, st, x, factor, aca);
    if (store_acc) accs[s0] = aca;
    if (len > 2u) {
        pa = pcs[s0 + 2u];
        if (with_acc) aca = accs[s0 + 2u];
    }
    if (len > 1u) {
        run_contact(type, pb2, st, x, factor, acb);
        if (store_acc) accs[s0 + 1u] = acb;
    }
    if (len > 3u) {
        pb2 = pcs[s0 + 3u];
        if (with_acc) acb = accs[s0 + 3u];
    }
    if (len > 2u) {
        run_contact(type, pa, st, x, factor, aca);
        if (store_acc) accs[s0 + 2u] = aca;
    }
    if (len > 3u) {
        run_contact(type, pb2, st, x, factor, acb);
        if (store_acc) accs[s0 + 3u] = acb;
    }
    for (uint32_t c = 4; c < len; ++c) {  // manifolds with more than four points
        const PhysContact q = pcs[s0 + c];
        float4 a = with_acc ? accs[s0 + c] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        run_contact(type, q, st, x, factor, a);
        if (store_acc) accs[s0 + c] = a;
    }
    if (PHASE == 1 && rv.applied && !(st.dyn_a && st.dyn_b)) rv.applied[item_index] = x.applied;
    if (PHASE == 0) {
        if (st.dyn_a) {
            float* d = LDS ? s_dyn + 6 * ia : cb[ia].v;
            st3(d, x.va);
            st3(LDS ? d + 3 : cb[ia].w, x.wa);
        }
        if (st.dyn_b) {
            float* d = LDS ? s_dyn + 6 * ib : cb[ib].v;
            st3(d, x.vb);
            st3(LDS ? d + 3 : cb[ib].w, x.wb);
        }
    } else {
        if (st.dyn_a) {
            st3(LDS ? s_dyn + 7 * ia : cb[ia].pos, x.pa);
            stq(LDS ? s_dyn + 7 * ia + 3 : cb[ia].q, x.qa);
        }
        if (st.dyn_b) {
            st3(LDS ? s_dyn + 7 * ib : cb[ib].pos, x.pb);
            stq(LDS ? s_dyn + 7 * ib + 3 : cb[ib].q, x.qb);
        }
    }
}

// 512 threads: a chain keeps ~150 values live (pair state, statics, the prepared contact); 1024 threads would cap the
// kernel at 128 VGPRs and spill a third of them to scratch
constexpr uint32_t SOLVE_THREADS = 768u;
template <bool LDS, int PHASE>
__global__ __launch_bounds__(SOLVE_THREADS) void k_solve(uint32_t n_dyn, float factor, const PhysContact* __restrict__ pcs, float4* __restrict__ accs,
                                                PhysBody* __restrict__ cb, const uint32_t* __restrict__ items, const uint2* __restrict__ item_bodies,
                                                const uint32_t* __restrict__ level_start, uint32_t n_levels, ReplayView rv,
                                                const uint32_t* __restrict__ replay_flag) {
    extern __shared__ float s_dyn[];
    __shared__ uint32_t s_start[PHYS_LEVEL_TILE + 1];
    const uint32_t tid = threadIdx.x;
    if (replay_flag && *replay_flag == 0u) return;  // (pass 2 of the positional phase: no kinematic orientation moved)
    if (LDS)
        for (uint32_t i = tid; i < n_dyn; i += SOLVE_THREADS) {
            const PhysBody& b = cb[i];
            if (PHASE == 0) {
                st3(s_dyn + 6 * i, ld3(b.v));
                st3(s_dyn + 6 * i + 3, ld3(b.w));
            } else {
                st3(s_dyn + 7 * i, ld3(b.pos));
                stq(s_dyn + 7 * i + 3, ldq(b.q));
            }
        }
    for (uint32_t l0 = 0; l0 < n_levels; l0 += PHYS_LEVEL_TILE) {
        const uint32_t cnt = min((uint32_t)PHYS_LEVEL_TILE, n_levels - l0);
        __syncthreads();
        for (uint32_t i = tid; i <= cnt; i += SOLVE_THREADS) s_start[i] = level_start[l0 + i];
        __syncthreads();
        // the item word and body pair of this thread's first chain of the next level are fetched one level ahead
        uint32_t nxt_item = 0;
        uint2 nxt_bodies = make_uint2(0u, 0u);
        if (s_start[0] + tid < s_start[1]) {
            nxt_item = items[s_start[0] + tid];
            nxt_bodies = item_bodies[s_start[0] + tid];
        }
        for (uint32_t l = 0; l < cnt; ++l) {
            const uint32_t b = s_start[l], e = s_start[l + 1];
            const uint32_t cur_item = nxt_item;
            const uint2 cur_bodies = nxt_bodies;
            if (l + 1 < cnt && e + tid < s_start[l + 2]) {
                nxt_item = items[e + tid];
                nxt_bodies = item_bodies[e + tid];
            }
            if (b + tid < e) run_chain<LDS, PHASE>(cur_item, cur_bodies, n_dyn, factor, pcs, accs, cb, s_dyn, b + tid, rv);
            for (uint32_t i = b + tid + SOLVE_THREADS; i < e; i += SOLVE_THREADS)
                run_chain<LDS, PHASE>(items[i], item_bodies[i], n_dyn, factor, pcs, accs, cb, s_dyn, i, rv);
            __syncthreads();  // workgroup-scope release/acquire of the body state before the next level
        }
    }
    if (LDS)
        for (uint32_t i = tid; i < n_dyn; i += SOLVE_THREADS) {
            PhysBody& b = cb[i];
            if (PHASE == 0) {
                st3(b.v, ld3(s_dyn + 6 * i));
                st3(b.w, ld3(s_dyn + 6 * i + 3));
            } else {
                st3(b.pos, ld3(s_dyn + 7 * i));
                stq(b.q, ldq(s_dyn + 7 * i + 3));
            }
        }
}


// ---- the solve on several workgroups ------------------------------------------------------------------------------------------------
// One workgroup runs a level of ~570 chains as 9 waves on 4 SIMDs, three to a SIMD, and a chain is ~1000 dependent f32 instructions:
// each wave gets a third of its SIMD's issue slots, so the level's critical path is three times what the chain itself costs. Spread
// over G workgroups of 256 threads (one wave per SIMD, G CUs) the chains run at full issue rate; the price is the bodies' mutable state in
// memory instead of LDS and a hand-off between workgroups. Every word another workgroup may read — the phase's body state (tagged 16-byte
// records per body in `dynst`: v, w | pos, q.xyz, q.w) and the accumulated impulses — is stored write-through (sc1; plain when all working
// workgroups share one XCD) and loaded past the L1 (sc1); what an item waits for are version tags in those records (run_chain_mg), two grid
// barriers per launch stand behind the set-up and before the write-back (a monotonic agent-scope counter, polled with sc1 loads:
// MI355X_MICROARCH.md, "Hand-offs measured with sc1 loads in place of the acquire", first row; all G workgroups are resident: G <= 16 blocks
// of 256 threads on 256 CUs). The schedule, the chain arithmetic and hence the results are those of the single-workgroup kernel, operation
// for operation.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld16_sc1(__amdgpu_buffer_rsrc_t rs, uint32_t byte_off) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)byte_off, 0, 16);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ void st16_sc1(__amdgpu_buffer_rsrc_t rs, uint32_t byte_off, float4 f) {
    u32x4 v;
    v.x = __float_as_uint(f.x), v.y = __float_as_uint(f.y), v.z = __float_as_uint(f.z), v.w = __float_as_uint(f.w);
    __builtin_amdgcn_raw_buffer_store_b128(v, rs, (int)byte_off, 0, 16);
}
// The hand-off store of the multi-workgroup solve. A write-through (sc1) store is what another workgroup ANYWHERE on the chip can read back with
// an sc1 load — but it also drops the line from the XCD's L2, so the reader's load goes to the memory side: ~2 us of every level. When all
// working workgroups sit on ONE XCD (k_solve_mg places them so and CHECKS it, `one_xcd`), their common L2 is the point of coherence: a plain
// store (acknowledged by that L2: the storing wave's s_waitcnt vmcnt(0) still stands before the barrier) and an L1-bypassing load (sc1) meet
// there. Placement decides the form, never the result.
__device__ __forceinline__ void st16_shared(__amdgpu_buffer_rsrc_t rs, uint32_t byte_off, float4 f, bool one_xcd) {
    u32x4 v;
    v.x = __float_as_uint(f.x), v.y = __float_as_uint(f.y), v.z = __float_as_uint(f.z), v.w = __float_as_uint(f.w);
    if (one_xcd) __builtin_amdgcn_raw_buffer_store_b128(v, rs, (int)byte_off, 0, 0);
    else __builtin_amdgcn_raw_buffer_store_b128(v, rs, (int)byte_off, 0, 16);
}

// What a chain reads that does not change during the solve — its (up to) four prepared contacts and the two bodies' constants — comes from a
// PACKED copy made once per step (k_pack_items): one record per item of the schedule, stored tile by tile of 64 consecutive items of a level,
// field-major inside the tile, so that lane l's j-th 16 bytes lie next to lane l+1's. Gathered from the contact and body arrays every lane
// reads lines of its own: ~34 load instructions x 64 lines per wave and level, and the cycle counters put the larger half of a velocity
// level there. A chain appears in 9 velocity and 3 positional items, hence as many copies (~70 MB for the 46 080-contact pile).
template <int PHASE>
struct Packed {
    static constexpr uint32_t CJ = PHASE == 0 ? 5u : 3u;  // float4s per contact
    static constexpr uint32_t BJ = PHASE == 0 ? 5u : 3u;  // float4s per body
    static constexpr uint32_t NJ = 4u * CJ + 2u * BJ;
};
template <int PHASE>
__device__ __forceinline__ void pack_contact(const PhysContact& p, float4* out /* stride 64 */) {
    if (PHASE == 0) {
        out[0] = make_float4(p.normal[0], p.normal[1], p.normal[2], p.tangent[0]);
        out[64] = make_float4(p.tangent[1], p.tangent[2], p.bitangent[0], p.bitangent[1]);
        out[128] = make_float4(p.bitangent[2], p.m_n, p.m_t, p.m_b);
        out[192] = make_float4(p.friction, p.target, p.world_b[0], p.world_b[1]);
        out[256] = make_float4(p.world_b[2], 0.0f, 0.0f, 0.0f);
    } else {
        out[0] = make_float4(p.normal[0], p.normal[1], p.normal[2], p.local_a[0]);
        out[64] = make_float4(p.local_a[1], p.local_a[2], p.local_b[0], p.local_b[1]);
        out[128] = make_float4(p.local_b[2], 0.0f, 0.0f, 0.0f);
    }
}
template <int PHASE>
__device__ __forceinline__ PhysContact unpack_contact(const float4* in /* the contact's CJ words of the item's record, in registers */) {
    PhysContact p = {};
    const float4 a = in[0], b = in[1], c = in[2];
    p.normal[0] = a.x, p.normal[1] = a.y, p.normal[2] = a.z;
    if (PHASE == 0) {
        const float4 d = in[3], e = in[4];
        p.tangent[0] = a.w, p.tangent[1] = b.x, p.tangent[2] = b.y;
        p.bitangent[0] = b.z, p.bitangent[1] = b.w, p.bitangent[2] = c.x;
        p.m_n = c.y, p.m_t = c.z, p.m_b = c.w;
        p.friction = d.x, p.target = d.y;
        p.world_b[0] = d.z, p.world_b[1] = d.w, p.world_b[2] = e.x;
    } else {
        p.local_a[0] = a.w, p.local_a[1] = b.x, p.local_a[2] = b.y;
        p.local_b[0] = b.z, p.local_b[1] = b.w, p.local_b[2] = c.x;
    }
    return p;
}
// an item's packed record into registers: NJ fully coalesced 16-byte loads (lane l's j-th word lies next to lane l+1's)
template <int PHASE>
__device__ __forceinline__ void load_packed(const float4* __restrict__ pk /* the tile's base + the lane */, float4 (&rec)[Packed<PHASE>::NJ]) {
#pragma unroll
    for (uint32_t j = 0; j < Packed<PHASE>::NJ; ++j) rec[j] = pk[(size_t)j * 64u];
}
// (`version`: the hand-off tag the item waits for on this body's shared record, run_chain_mg — a schedule constant riding in a free slot)
template <int PHASE>
__device__ __forceinline__ void pack_body(const PhysBody& b, float4* out, uint32_t version) {
    out[0] = make_float4(b.inv_mass, b.inv_inertia[0], b.inv_inertia[1], b.inv_inertia[2]);
    out[64] = make_float4(b.inv_inertia[3], b.inv_inertia[4], b.inv_inertia[5], b.inv_inertia[6]);
    if (PHASE == 0) {
        out[128] = make_float4(b.inv_inertia[7], b.inv_inertia[8], b.pos[0], b.pos[1]);
        out[192] = make_float4(b.pos[2], b.v[0], b.v[1], b.v[2]);
        out[256] = make_float4(b.w[0], b.w[1], b.w[2], __uint_as_float(version));
    } else {
        out[128] = make_float4(b.inv_inertia[7], b.inv_inertia[8], __uint_as_float(version), 0.0f);
    }
}
// one thread per item, one block per tile of 64 items (tile_first: index of the tile's first item | (items in the tile - 1) << 26)
template <int PHASE>
__global__ __launch_bounds__(64) void k_pack_items(const uint32_t* __restrict__ tile_first, const uint32_t* __restrict__ items,
                                                   const uint2* __restrict__ item_bodies, const uint4* __restrict__ item_tags,
                                                   const PhysContact* __restrict__ pcs, const PhysBody* __restrict__ cb, float4* __restrict__ packed) {
    const uint32_t tf = tile_first[blockIdx.x], first = tf & 0x03FFFFFFu, cnt = (tf >> 26) + 1u, lane = threadIdx.x;
    if (lane >= cnt) return;
    const uint32_t item = items[first + lane];
    const uint2 bo = item_bodies[first + lane];
    const uint4 tg = item_tags[first + lane];
    const uint32_t s0 = item & 0x00FFFFFFu, len = (item >> 24) & 15u;
    float4* out = packed + (size_t)blockIdx.x * Packed<PHASE>::NJ * 64u + lane;
#pragma unroll
    for (uint32_t c = 0; c < 4u; ++c) pack_contact<PHASE>(pcs[s0 + (c < len ? c : len - 1u)], out + (size_t)c * Packed<PHASE>::CJ * 64u);
    if (PHASE == 0) {  // the sweep tags of the accumulated impulses (found, left) beside the first contact's last word
        float4 e = out[256];
        e.y = __uint_as_float(tg.z), e.z = __uint_as_float(tg.w);
        out[256] = e;
    }
    pack_body<PHASE>(cb[bo.x], out + (size_t)4u * Packed<PHASE>::CJ * 64u, tg.x);
    pack_body<PHASE>(cb[bo.y], out + (size_t)(4u * Packed<PHASE>::CJ + Packed<PHASE>::BJ) * 64u, tg.y);
}

constexpr uint32_t MG_THREADS = 256u;
constexpr uint32_t MG_SPIN_LIMIT = 1u << 22;  // a poll loop that never sees its count gives up and flags the launch (every spin is bounded)

// `rec`: the item's packed record, already in registers (load_packed: fetched a level ahead, behind the grid barrier's arrival)
// Returns true when a poll gave up (the launch is flagged; the caller stops taking tiles: after a producer that never stored no tag can match
// again, and every later tile would spin the full limit too).
template <int PHASE>
__device__ __forceinline__ bool run_chain_mg(uint32_t item, uint2 bodies, uint32_t n_dyn, float factor, const PhysContact* __restrict__ pcs,
                                             __amdgpu_buffer_rsrc_t rs_acc, const PhysBody* __restrict__ cb, __amdgpu_buffer_rsrc_t rs_dyn,
                                             const float4 (&rec)[Packed<PHASE>::NJ], uint32_t item_index, const ReplayView& rv, bool one_xcd, bool live,
                                             bool no_wait, bool dry_chain, uint32_t* error) {
    const uint32_t s0 = item & 0x00FFFFFFu, len = (item >> 24) & 15u, type = item >> 28;
    const uint32_t ia = bodies.x, ib = bodies.y;
    const bool with_acc = type != PHYS_ITEM_POSITIONAL, store_acc = type == PHYS_ITEM_VELOCITY;
    PairStatic st;
    st.dyn_a = ia < n_dyn;
    st.dyn_b = ib < n_dyn;
    // THE HAND-OFF. Every shared 16-byte record carries a tag in a slot its data do not use: a body's records the number of items that have
    // touched the body this phase (the launch's set-up writes 0), a contact's accumulated impulses the number of velocity sweeps that have
    // written them (0: as prepared). The item knows from the schedule which tags it must find (in the packed record: pack_body, k_pack_items)
    // and simply loads its operands, past the L1, until all of them carry those tags: the load that sees the producer's store IS the load of
    // the data. No completion word per tile, no wait for the stores' acknowledgement before it, no second trip for the data behind it —
    // a third of a level's time in the tile-flag form. Every dependency of an item is an earlier item touching one of its bodies (the
    // accumulated impulses are its own chain's, one sweep back, and that item touched the same bodies), so what is waited for here is exactly
    // the schedule's partial order; lanes of a wave (one level, mutually independent items) wait together.
    constexpr uint32_t CJ0 = Packed<PHASE>::CJ, BJ0 = Packed<PHASE>::BJ;
    const uint32_t DST = PHASE == 0 ? 32u : 48u;  // bytes of a body's shared records
    const uint32_t ver_a = __float_as_uint(PHASE == 0 ? rec[4u * CJ0 + 4u].w : rec[4u * CJ0 + 2u].z);
    const uint32_t ver_b = __float_as_uint(PHASE == 0 ? rec[4u * CJ0 + BJ0 + 4u].w : rec[4u * CJ0 + BJ0 + 2u].z);
    const uint32_t acc_in = PHASE == 0 ? __float_as_uint(rec[4].y) : 0u, acc_out = PHASE == 0 ? __float_as_uint(rec[4].z) : 0u;
    float4 a0 = make_float4(0, 0, 0, 0), a1 = a0, a2 = a0, b0 = a0, b1 = a0, b2 = a0;
    float4 c0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), c1 = c0, c2 = c0, c3 = c0;
    for (uint32_t spins = 0;; ++spins) {
        bool ok = true;
        if (live && st.dyn_a) {
            a0 = ld16_sc1(rs_dyn, ia * DST);
            a1 = ld16_sc1(rs_dyn, ia * DST + 16u);
            ok = ok && __float_as_uint(a0.w) == ver_a && __float_as_uint(a1.w) == ver_a;
            if (PHASE == 1) {
                a2 = ld16_sc1(rs_dyn, ia * DST + 32u);
                ok = ok && __float_as_uint(a2.w) == ver_a;
            }
        }
        if (live && st.dyn_b) {
            b0 = ld16_sc1(rs_dyn, ib * DST);
            b1 = ld16_sc1(rs_dyn, ib * DST + 16u);
            ok = ok && __float_as_uint(b0.w) == ver_b && __float_as_uint(b1.w) == ver_b;
            if (PHASE == 1) {
                b2 = ld16_sc1(rs_dyn, ib * DST + 32u);
                ok = ok && __float_as_uint(b2.w) == ver_b;
            }
        }
        if (live && with_acc) {
            c0 = ld16_sc1(rs_acc, s0 * 16u);
            ok = ok && __float_as_uint(c0.w) == acc_in;
            if (len > 1u) c1 = ld16_sc1(rs_acc, (s0 + 1u) * 16u), ok = ok && __float_as_uint(c1.w) == acc_in;
            if (len > 2u) c2 = ld16_sc1(rs_acc, (s0 + 2u) * 16u), ok = ok && __float_as_uint(c2.w) == acc_in;
            if (len > 3u) c3 = ld16_sc1(rs_acc, (s0 + 3u) * 16u), ok = ok && __float_as_uint(c3.w) == acc_in;
        }
        if (no_wait || __builtin_amdgcn_ballot_w64(!ok) == 0ull) break;
        __builtin_amdgcn_s_sleep(1);
        if (spins > MG_SPIN_LIMIT) {
            if ((threadIdx.x & 63u) == 0u) __hip_atomic_store(error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // (host-mapped: ivx_world_check_solve)
            return true;
        }
    }
    if (!live || dry_chain) return false;
    constexpr uint32_t CJ = Packed<PHASE>::CJ, BJ = Packed<PHASE>::BJ;
    const PhysContact p0 = unpack_contact<PHASE>(rec), p1 = unpack_contact<PHASE>(rec + CJ), p2 = unpack_contact<PHASE>(rec + 2u * CJ),
                      p3 = unpack_contact<PHASE>(rec + 3u * CJ);
    const float4* ba = rec + 4u * CJ;
    const float4* bb = ba + BJ;
    const float4 ak0 = ba[0], ak1 = ba[1], ak2 = ba[2], bk0 = bb[0], bk1 = bb[1], bk2 = bb[2];
    st.ima = ak0.x;
    st.imb = bk0.x;
    st.iia = M3{mk(ak0.y, ak0.z, ak0.w), mk(ak1.x, ak1.y, ak1.z), mk(ak1.w, ak2.x, ak2.y)};
    st.iib = M3{mk(bk0.y, bk0.z, bk0.w), mk(bk1.x, bk1.y, bk1.z), mk(bk1.w, bk2.x, bk2.y)};
    PairState x;
    if (PHASE == 0) {
        const float4 ak3 = ba[3], ak4 = ba[4], bk3 = bb[3], bk4 = bb[4];
        st.pos_a = mk(ak2.z, ak2.w, ak3.x);
        st.pos_b = mk(bk2.z, bk2.w, bk3.x);
        // (what a kinematic body moves with; unused for dynamic ones)
        const V3 kva = mk(ak3.y, ak3.z, ak3.w), kwa = mk(ak4.x, ak4.y, ak4.z), kvb = mk(bk3.y, bk3.z, bk3.w), kwb = mk(bk4.x, bk4.y, bk4.z);
        x.va = st.dyn_a ? mk(a0.x, a0.y, a0.z) : kva;
        x.wa = st.dyn_a ? mk(a1.x, a1.y, a1.z) : kwa;
        x.vb = st.dyn_b ? mk(b0.x, b0.y, b0.z) : kvb;
        x.wb = st.dyn_b ? mk(b1.x, b1.y, b1.z) : kwb;
    } else {
        x.pa = st.dyn_a ? mk(a0.x, a0.y, a0.z) : ld3(cb[ia].pos);
        x.qa = st.dyn_a ? Q4{a1.x, a1.y, a1.z, a2.x} : ldq(cb[ia].q);
        x.pb = st.dyn_b ? mk(b0.x, b0.y, b0.z) : ld3(cb[ib].pos);
        x.qb = st.dyn_b ? Q4{b1.x, b1.y, b1.z, b2.x} : ldq(cb[ib].q);
        if (rv.qstart) {  // pass 2: a kinematic body's orientation as the chains before this one left it (ReplayView)
            if (!st.dyn_a) {
                const float4 t = rv.qstart[2u * item_index];
                x.qa = Q4{t.x, t.y, t.z, t.w};
            }
            if (!st.dyn_b) {
                const float4 t = rv.qstart[2u * item_index + 1u];
                x.qb = Q4{t.x, t.y, t.z, t.w};
            }
        }
    }
    run_contact(type, p0, st, x, factor, c0);
    if (len > 1u) run_contact(type, p1, st, x, factor, c1);
    if (len > 2u) run_contact(type, p2, st, x, factor, c2);
    if (len > 3u) run_contact(type, p3, st, x, factor, c3);
    const float tag_acc = __uint_as_float(acc_out);
    for (uint32_t c = 4; c < len; ++c) {  // manifolds with more than four points (their accumulated impulses: the same wait, contact by contact)
        const PhysContact q = pcs[s0 + c];
        float4 a = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (with_acc)
            for (uint32_t spins = 0;; ++spins) {
                a = ld16_sc1(rs_acc, (s0 + c) * 16u);
                if (no_wait || __float_as_uint(a.w) == acc_in) break;
                __builtin_amdgcn_s_sleep(1);
                if (spins > MG_SPIN_LIMIT) {  // like the other bounded spins: give up, flag the launch (ivx_world_check_solve)
                    __hip_atomic_store(error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    return true;
                }
            }
        run_contact(type, q, st, x, factor, a);
        if (store_acc) {
            a.w = tag_acc;
            st16_shared(rs_acc, (s0 + c) * 16u, a, one_xcd);
        }
    }
    if (store_acc) {
        c0.w = c1.w = c2.w = c3.w = tag_acc;  // (run_contact leaves 0 there)
        st16_shared(rs_acc, s0 * 16u, c0, one_xcd);
        if (len > 1u) st16_shared(rs_acc, (s0 + 1u) * 16u, c1, one_xcd);
        if (len > 2u) st16_shared(rs_acc, (s0 + 2u) * 16u, c2, one_xcd);
        if (len > 3u) st16_shared(rs_acc, (s0 + 3u) * 16u, c3, one_xcd);
    }
    if (PHASE == 1 && rv.applied && !(st.dyn_a && st.dyn_b)) rv.applied[item_index] = x.applied;
    // the bodies' records, one version on (the accumulated impulses above may land after them: whoever needs those waits for their own tags)
    const float na = __uint_as_float(ver_a + 1u), nb = __uint_as_float(ver_b + 1u);
    if (PHASE == 0) {
        if (st.dyn_a) {
            st16_shared(rs_dyn, ia * DST, make_float4(x.va.x, x.va.y, x.va.z, na), one_xcd);
            st16_shared(rs_dyn, ia * DST + 16u, make_float4(x.wa.x, x.wa.y, x.wa.z, na), one_xcd);
        }
        if (st.dyn_b) {
            st16_shared(rs_dyn, ib * DST, make_float4(x.vb.x, x.vb.y, x.vb.z, nb), one_xcd);
            st16_shared(rs_dyn, ib * DST + 16u, make_float4(x.wb.x, x.wb.y, x.wb.z, nb), one_xcd);
        }
    } else {
        if (st.dyn_a) {
            st16_shared(rs_dyn, ia * DST, make_float4(x.pa.x, x.pa.y, x.pa.z, na), one_xcd);
            st16_shared(rs_dyn, ia * DST + 16u, make_float4(x.qa.x, x.qa.y, x.qa.z, na), one_xcd);
            st16_shared(rs_dyn, ia * DST + 32u, make_float4(x.qa.w, 0.0f, 0.0f, na), one_xcd);
        }
        if (st.dyn_b) {
            st16_shared(rs_dyn, ib * DST, make_float4(x.pb.x, x.pb.y, x.pb.z, nb), one_xcd);
            st16_shared(rs_dyn, ib * DST + 16u, make_float4(x.qb.x, x.qb.y, x.qb.z, nb), one_xcd);
            st16_shared(rs_dyn, ib * DST + 32u, make_float4(x.qb.w, 0.0f, 0.0f, nb), one_xcd);
        }
    }
    return false;
}


// grid-wide barrier of the G resident workgroups: `target` arrivals on the monotonic counter. In two halves, so that loads which do not
// depend on the other workgroups (the next level's packed records) can be issued between a workgroup's arrival and its wait: their trip to
// memory then runs beside the wait instead of after it.
__device__ __forceinline__ void mg_arrive(uint32_t* counter) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's write-through stores have landed
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void mg_wait(uint32_t* counter, uint32_t target, uint32_t* error) {
    if (threadIdx.x == 0) {
        uint32_t spins = 0;
        while ((int32_t)(__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > MG_SPIN_LIMIT) {
                __hip_atomic_store(error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // (a host-mapped word: ivx_world_check_solve)
                break;
            }
        }
    }
    __syncthreads();
}
__device__ __forceinline__ void mg_barrier(uint32_t* counter, uint32_t target, uint32_t* error) {
    mg_arrive(counter);
    mg_wait(counter, target, error);
}

template <int PHASE>
__global__ __launch_bounds__(MG_THREADS) void k_solve_mg(uint32_t n_dyn, float factor, const PhysContact* __restrict__ pcs, float4* __restrict__ accs,
                                                         uint32_t n_contacts, PhysBody* __restrict__ cb, float4* __restrict__ dynst,
                                                         const uint32_t* __restrict__ items, const uint2* __restrict__ item_bodies,
                                                         const uint32_t* __restrict__ level_start, const uint32_t* __restrict__ tile_base,
                                                         const float4* __restrict__ packed, uint32_t n_levels, uint32_t* __restrict__ counter,
                                                         uint32_t counter_base, uint32_t* __restrict__ error, uint32_t dry, uint32_t spread, ReplayView rv,
                                                         const uint32_t* __restrict__ replay_flag, uint32_t* __restrict__ xcc_table,
                                                         const uint32_t* __restrict__ tile_first, uint32_t n_tiles) {
    // `spread` = 8: the launch holds 8 G blocks of which every eighth works — workgroups are handed to the XCDs round robin, so the G that
    // work share one XCD (its L2, its fabric port); 1: G blocks, all working. Where they land changes times only, never results.
    if (spread > 1u && (blockIdx.x % spread) != 0u) return;
    if (replay_flag && *replay_flag == 0u) {  // pass 2 of the positional phase with no kinematic orientation that moved: nothing to do
        // (uniform over the launch; the host has already counted this launch's arrivals at the grid barrier: make them)
        if (threadIdx.x == 0u) __hip_atomic_fetch_add(counter, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    const uint32_t blk = blockIdx.x / spread;
    const uint32_t tid = threadIdx.x, G = gridDim.x / spread, slot = blk * MG_THREADS + tid, stride = G * MG_THREADS;
    const uint32_t DST = PHASE == 0 ? 32u : 48u;  // bytes of a body's shared records: (v | tag)(w | tag) — (p | tag)(q.xyz | tag)(q.w, -, - | tag)
    const __amdgpu_buffer_rsrc_t rs_dyn = __builtin_amdgcn_make_buffer_rsrc(dynst, 0, n_dyn * DST, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_acc = __builtin_amdgcn_make_buffer_rsrc(accs, 0, n_contacts * 16u, 0x00020000);
    // the phase's mutable state of the dynamic bodies into the shared records
    for (uint32_t i = slot; i < n_dyn; i += stride) {
        const PhysBody& b = cb[i];
        if (PHASE == 0) {  // (tag 0 in every record's last word: no item has touched the body yet, run_chain_mg)
            st16_sc1(rs_dyn, i * DST, make_float4(b.v[0], b.v[1], b.v[2], 0.0f));
            st16_sc1(rs_dyn, i * DST + 16u, make_float4(b.w[0], b.w[1], b.w[2], 0.0f));
        } else {
            st16_sc1(rs_dyn, i * DST, make_float4(b.pos[0], b.pos[1], b.pos[2], 0.0f));
            st16_sc1(rs_dyn, i * DST + 16u, make_float4(b.q[0], b.q[1], b.q[2], 0.0f));
            st16_sc1(rs_dyn, i * DST + 32u, make_float4(b.q[3], 0.0f, 0.0f, 0.0f));
        }
    }
    // census: which XCD is every working workgroup on? (table entry = this launch's counter base | XCC_ID; read back behind the first barrier)
    const uint32_t xcc_tag = (counter_base << 4) | (__builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xFu);  // hwreg(HW_REG_XCC_ID, 0, 4)
    if (tid == 0u) __hip_atomic_store(xcc_table + blk, xcc_tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t arrivals = counter_base + G;
    mg_barrier(counter, arrivals, error);
    bool one_xcd = G <= 16u;
    for (uint32_t q = 0; q < G && q < 16u; ++q) one_xcd = one_xcd && __hip_atomic_load(xcc_table + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == xcc_tag;
    if (dry & 4u) one_xcd = false;  // (developer switch: the write-through form wherever the workgroups sit)
    // ---- the schedule, TILE by tile. A tile is 64 consecutive items of a level (mutually independent chains) and the unit a wave runs; wave w
    // of the launch's W = 4 G waves runs the tiles w, w + W, ... in the schedule's order. What a tile waits for is in the data itself: every
    // shared record carries a version tag and an item loads its operands until they carry the tags the schedule says (run_chain_mg) — no grid
    // barrier per level (it made every wave wait for the slowest chain of the whole level), and no completion words per tile either (they cost
    // the producer a wait for its stores' acknowledgement and the consumer a second trip for the data). Every dependency points to an earlier
    // item and every wave takes its tiles in order, so the earliest unfinished tile can always run: no deadlock while the launch's workgroups
    // are resident (checked by the host, bounded polls besides). The item words, the body pairs and the PACKED RECORD of a wave's next tile are
    // fetched while it finishes the current one (they never change during the solve).
    const uint32_t lane = tid & 63u, W = G * (MG_THREADS / 64u), wv = blk * (MG_THREADS / 64u) + (tid >> 6);
    constexpr uint32_t NONE = 0xFFFFFFFFu;
    uint32_t t = wv;
    uint32_t nxt_item = NONE, nxt_index = 0u;
    uint2 nxt_bodies = make_uint2(0u, 0u);
    float4 rec[Packed<PHASE>::NJ];
#pragma unroll
    for (uint32_t j = 0; j < Packed<PHASE>::NJ; ++j) rec[j] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    auto fetch_tile = [&](uint32_t tt) {
        nxt_item = NONE;
        if (tt >= n_tiles) return;
        const uint32_t tf = tile_first[tt], first = tf & 0x03FFFFFFu, cnt = (tf >> 26) + 1u;
        if (lane < cnt) {
            nxt_item = items[first + lane];
            nxt_bodies = item_bodies[first + lane];
            nxt_index = first + lane;
            load_packed<PHASE>(packed + (size_t)tt * (Packed<PHASE>::NJ * 64u) + lane, rec);
        }
    };
    fetch_tile(t);
    for (; t < n_tiles; t += W) {
        const uint32_t cur_item = nxt_item, cur_index = nxt_index;
        const uint2 cur_bodies = nxt_bodies;
        const bool gave_up = run_chain_mg<PHASE>(cur_item, cur_bodies, n_dyn, factor, pcs, rs_acc, cb, rs_dyn, rec, cur_index, rv, one_xcd, cur_item != NONE,
                                                 (dry & 3u) != 0u, (dry & 1u) != 0u, error);
        if (__builtin_amdgcn_ballot_w64(gave_up) != 0ull) break;  // (the wave still makes the closing barrier's arrival below)
        fetch_tile(t + W);
    }
    // every tile of the launch done (one grid barrier) before the shared records go back to the body array
    arrivals += G;
    mg_barrier(counter, arrivals, error);
    for (uint32_t i = slot; i < n_dyn; i += stride) {
        PhysBody& b = cb[i];
        const float4 r0 = ld16_sc1(rs_dyn, i * DST), r1 = ld16_sc1(rs_dyn, i * DST + 16u);
        if (PHASE == 0) {
            st3(b.v, mk(r0.x, r0.y, r0.z));
            st3(b.w, mk(r1.x, r1.y, r1.z));
        } else {
            const float4 r2 = ld16_sc1(rs_dyn, i * DST + 32u);
            st3(b.pos, mk(r0.x, r0.y, r0.z));
            stq(b.q, Q4{r1.x, r1.y, r1.z, r2.x});
        }
    }
}

// ---- the chain-stationary solve ------------------------------------------------------------------------------------------------------
// k_solve_mg hands tiles of the level schedule to whichever wave is next: every item fetches its packed record (480 bytes), its accumulated
// impulses and its two bodies, and leaves impulses and bodies behind — and a level costs ~2.6 us of work + ~2 us of hand-off, 183 times.
// Here a chain STAYS where it is for the whole phase (host: build_stationary): a PAIR OF LANES of one wave, the even lane body A's side,
// the odd lane body B's. What never changes during the solve — the chain's prepared contacts, the own body's inverse mass and inertia — is
// gathered once into the lane's registers, the accumulated impulses live there from the warm start to the last sweep, and the only thing that
// travels between items is what has to: a body's mutable state, through the same version-tagged 16-byte records as k_solve_mg's (a body's
// record carries the number of items that have touched it; sweep s of a chain finds s * degree + rank there, two per-lane constants). Each
// lane waits for, carries and stores its OWN body only; what an item needs of the other side — the contact point's velocity or position,
// one term of the effective mass — crosses to the neighbouring lane by a DPP swap, and the two halves of apply_pair / the positional
// correction, which are independent of each other, run side by side: ~130 instructions per velocity contact on the critical lane instead
// of 165, ~230 per positional contact instead of 420. Every float operation is one the sequential code performs, on the same operands in
// the same order (B's half uses -x where the sequential code subtracts x: a + (-x) = a - x, (-x) y = -(x y) and the cross product of a
// negated vector is the negated cross product, all exactly), so the result is the sequential loop's bit for bit.
// A wave walks its tile's ROUNDS — the distinct levels its items lie on, a lane mask each — in level order.
// One launch runs BOTH phases: blocks with blockIdx % spread == 0 are the velocity phase's workgroups, == 1 the positional phase's (under the
// observed round-robin placement two XCDs with a phase each; the census decides the form of the stores, never the result). All working
// workgroups must be resident (one per CU, at most PHYS_CS_MAX_GROUPS per phase); every poll is bounded and flags the launch.
struct CsPhase {
    const uint32_t* item;         // [tile * 64 + lane] first contact | length << 24, ~0: no chain (both lanes of a pair hold the same word)
    const uint2* bodies;          // the lane's own body and the other one of the pair (constrained-body indices)
    const uint32_t* vers;         // degree | rank << 16 of the chain on the lane's own body
    const uint32_t* round_start;  // [tile] first round of the tile (n_tiles + 1 entries)
    const uint64_t* round_mask;   // lanes of the round
    const uint32_t* round_level;  // level its items lie on (from 1)
    float4* dynst;                // the phase's shared body records
    uint32_t* counter;            // grid barrier of the phase's workgroups (monotonic)
    uint32_t* xcc_table;
    const uint32_t* slot_of;      // positional phase with kinematic bodies: [(tile * 32 + pair) * n_passes + sweep] -> the index ReplayView goes by
    const uint32_t* replay_flag;
    unsigned long long* trace;    // developer switch IVX_SOLVER_TRACE: per round 4 stamps of the 100 MHz clock (round begins, operands there, chain run, stores issued)
    ReplayView rv;
    uint32_t n_tiles, n_groups, n_first, counter_base, n_passes;
    uint32_t nap;  // tenths of a level a wave starts polling ahead of its round's expected time (0: polls from the end of its last round)
};
constexpr uint32_t CS_THREADS = PHYS_CS_WAVES * 64u;

// the neighbouring lane's value (lanes 2k and 2k + 1 swap): DPP quad_perm [1, 0, 3, 2]
__device__ __forceinline__ float swap1(float x) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xF, 0xF, false)); }
__device__ __forceinline__ V3 swap1(V3 v) { return {swap1(v.x), swap1(v.y), swap1(v.z)}; }
// x on A's lane, -x on B's (`flip`: 0 / the sign bit)
__device__ __forceinline__ float flipf(float x, uint32_t flip) { return __uint_as_float(__float_as_uint(x) ^ flip); }
__device__ __forceinline__ V3 flipv(V3 v, uint32_t flip) { return {flipf(v.x, flip), flipf(v.y, flip), flipf(v.z, flip)}; }

// a prepared contact as one side's lane keeps it during the velocity phase (`d`: the own body's lever arm, world_b - position)
struct LaneV {
    V3 n, t, b;
    float m_n, m_t, m_b, friction, target;
    V3 d;
};
// the own body's half of apply_pair
__device__ __forceinline__ void lane_apply(const LaneV& c, float im, const M3& ii, uint32_t flip, bool dyn, V3& v, V3& w, float in, float it, float ib_) {
    const V3 dp = flipv((c.n * in + c.t * it) + c.b * ib_, flip);
    if (dyn) {
        v = v + dp * im;
        w = w + mul(ii, cross(c.d, dp));
    }
}
// one contact of a warm-start or velocity item on a pair of lanes (run_contact_v's operations; both lanes end with the same `acc`)
__device__ __forceinline__ void lane_contact_v(uint32_t type, const LaneV& c, float im, const M3& ii, uint32_t flip, bool dyn, V3& v, V3& w, float4& acc) {
    if (type == PHYS_ITEM_WARM) {
        lane_apply(c, im, ii, flip, dyn, v, w, acc.x, acc.y, acc.z);
        return;
    }
    const V3 pv = point_velocity(v, w, c.d);
    const V3 rel = flipv(pv - swap1(pv), flip);  // A: own - other; B: -(own - other) = other - own: point velocity of A minus that of B on both
    const float sep = dot(c.n, rel);
    const float cn = -c.m_n * (sep - c.target), ct = -c.m_t * dot(c.t, rel), cbi = -c.m_b * dot(c.b, rel);
    const float un = acc.x + cn, ut = acc.y + ct, ub = acc.z + cbi;
    const float nn = max_rs(0.0f, un);
    const float max_t = c.friction * nn;
    const float mag = sqrtf(ut * ut + ub * ub);
    const float sc = mag > max_t ? max_t / mag : 1.0f;
    const float nt = ut * sc, nb = ub * sc;
    lane_apply(c, im, ii, flip, dyn, v, w, nn - acc.x, nt - acc.y, nb - acc.z);
    acc = make_float4(nn, nt, nb, 0.0f);
}
// one contact of a positional item on a pair of lanes (run_contact_p's operations). `local`: the contact point in the own body's frame, `im`:
// the own body's inverse mass, `im_sum`: ima + imb, `ii`: the own body's inverse inertia (world space)
__device__ __forceinline__ void lane_contact_p(V3 n, V3 local, float im, float im_sum, const M3& ii, uint32_t flip, bool side_b, V3& p, Q4& q, float factor,
                                               uint32_t& applied) {
    const V3 pw = qrot(q, local) + p, po = swap1(pw);
    const float depth = dot(n, flipv(po - pw, flip));  // A: pb - pa; B: -(pa - pb)
    if (depth <= 0.0f) return;  // (the same decision on both lanes)
    const V3 pb = side_b ? pw : po;
    const V3 d = pb - p;
    const V3 c = cross(d, n);
    const float t_own = dot(c, mul(ii, c)), t_oth = swap1(t_own);
    const float m = 1.0f / ((im_sum + (side_b ? t_oth : t_own)) + (side_b ? t_own : t_oth));
    const V3 dp = n * (m * factor * depth);
    // A: pa + dp ima, rotation iia (da x dp); B: pb + dp (-imb), rotation (-iib) (db x dp) = -(iib (db x dp))
    p = p + dp * flipf(im, flip);
    q = pseudo_advanced(q, flipv(mul(ii, cross(d, dp)), flip));
    applied += 1u;
}

template <int PHASE>
__device__ __forceinline__ void solve_cs_phase(const uint32_t wg, const uint32_t n_dyn, const float factor, const PhysContact* __restrict__ pcs,
                                               float4* __restrict__ accs, const uint32_t n_contacts, const PhysBody* __restrict__ cb, const CsPhase& ph,
                                               uint32_t* __restrict__ error, const uint32_t dry) {
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const bool side_b = (lane & 1u) != 0u;
    const uint32_t flip = side_b ? 0x80000000u : 0u;
    const uint32_t tile = __builtin_amdgcn_readfirstlane(wg * PHYS_CS_WAVES + (tid >> 6));
    const uint32_t G = ph.n_groups;
    if (ph.replay_flag && *ph.replay_flag == 0u) {  // pass 2 of the positional phase with no kinematic orientation that moved (uniform over the launch)
        if (tid == 0u) __hip_atomic_fetch_add(ph.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (the host has counted this launch's arrivals)
        return;
    }
    constexpr uint32_t DST = PHASE == 0 ? 32u : 48u;
    const __amdgpu_buffer_rsrc_t rs_dyn = __builtin_amdgcn_make_buffer_rsrc(ph.dynst, 0, n_dyn * DST, 0x00020000);
    // census (as in k_solve_mg): which XCD is every working workgroup of the phase on?
    const uint32_t xcc_tag = (ph.counter_base << 4) | (__builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xFu);
    if (tid == 0u) __hip_atomic_store(ph.xcc_table + wg, xcc_tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    mg_arrive(ph.counter);
    // ---- the lane's side of its chain, once: prepared contacts, the own body's constants, the accumulated impulses (their trips to memory run
    // beside the barrier)
    const bool in_tile = tile < ph.n_tiles;
    const uint32_t slot = tile * 64u + lane;
    const uint32_t item = in_tile ? ph.item[slot] : 0xFFFFFFFFu;
    const bool has = item != 0xFFFFFFFFu;
    const uint32_t s0 = has ? (item & 0x00FFFFFFu) : 0u, len = has ? ((item >> 24) & 15u) : 0u;
    const uint2 bodies = has ? ph.bodies[slot] : make_uint2(0u, 0u);
    const uint32_t vr = has ? ph.vers[slot] : 0u;
    const uint32_t own = bodies.x;
    const uint32_t deg = vr & 0xFFFFu, rank = vr >> 16;
    const bool dyn = has && own < n_dyn, both_dyn = dyn && bodies.y < n_dyn;
    LaneV cv0 = {}, cv1 = {}, cv2 = {}, cv3 = {};  // velocity phase: the chain's prepared contacts, this side's view
    struct LaneP {
        V3 n, local;
    } cp0 = {}, cp1 = {}, cp2 = {}, cp3 = {};  // positional phase
    float4 ac0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), ac1 = ac0, ac2 = ac0, ac3 = ac0;
    float im = 0.0f, im_sum = 0.0f;
    M3 ii = {};
    if (has) {
        const PhysBody& O = cb[own];
        im = O.inv_mass;
        ii = ldm(O.inv_inertia);
        if (PHASE == 0) {
            const V3 pos = ld3(O.pos);
            auto cv = [&](uint32_t c) {
                const PhysContact& p = pcs[s0 + c];
                return LaneV{ld3(p.normal), ld3(p.tangent), ld3(p.bitangent), p.m_n, p.m_t, p.m_b, p.friction, p.target, ld3(p.world_b) - pos};
            };
            cv0 = cv(0u), ac0 = accs[s0];
            if (len > 1u) cv1 = cv(1u), ac1 = accs[s0 + 1u];
            if (len > 2u) cv2 = cv(2u), ac2 = accs[s0 + 2u];
            if (len > 3u) cv3 = cv(3u), ac3 = accs[s0 + 3u];
        } else {
            const float im_a = side_b ? cb[bodies.y].inv_mass : im, im_b = side_b ? im : cb[bodies.y].inv_mass;
            im_sum = im_a + im_b;
            auto cp = [&](uint32_t c) {
                const PhysContact& p = pcs[s0 + c];
                return LaneP{ld3(p.normal), side_b ? ld3(p.local_b) : ld3(p.local_a)};
            };
            cp0 = cp(0u);
            if (len > 1u) cp1 = cp(1u);
            if (len > 2u) cp2 = cp(2u);
            if (len > 3u) cp3 = cp(3u);
        }
    }
    mg_wait(ph.counter, ph.counter_base + G, error);
    bool one_xcd = G <= PHYS_CS_MAX_GROUPS;
    for (uint32_t q = 0; q < G && q < PHYS_CS_MAX_GROUPS; ++q)
        one_xcd = one_xcd && __hip_atomic_load(ph.xcc_table + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == xcc_tag;
    if (dry & 4u) one_xcd = false;  // (developer switch: the write-through form wherever the workgroups sit)
    if (!in_tile) return;
    // ---- the rounds
    // (the masks through the scalar cache: everything that addresses them is uniform over the wave; a round's mask is fetched a round ahead)
    const uint32_t r0 = __builtin_amdgcn_readfirstlane(ph.round_start[tile]), r1 = __builtin_amdgcn_readfirstlane(ph.round_start[tile + 1u]);
    uint32_t sweep = 0u;
    typedef const uint64_t __attribute__((address_space(4))) * scalar_u64_ptr;  // (constant address space: s_load, counted apart from the vector loads)
    const scalar_u64_ptr masks = (scalar_u64_ptr)(uintptr_t)ph.round_mask;
    typedef const uint32_t __attribute__((address_space(4))) * scalar_u32_ptr;
    const scalar_u32_ptr levels = (scalar_u32_ptr)(uintptr_t)ph.round_level;
    uint64_t mask_next = r0 < r1 ? masks[r0] : 0ull;
    uint32_t level_next = r0 < r1 ? levels[r0] : 0u;
    // NAPS. A chain runs once per sweep, a wave a handful of rounds per sweep: most of the time a wave's next round is levels away, and hundreds
    // of waves polling all that time fill the L2's request queues in front of the few whose operands are about to arrive (measured: 5.4 us per
    // level with everyone polling, 4.2 with naps). Levels pass at a steady pace for all waves, so a wave sleeps until `nap` tenths of a level
    // before its round is due, counted from the end of its last round at nine tenths of the pace so far by its own clock, (now - start) /
    // level, never taken below what a level of one-contact chains costs. A wrong guess costs time, never correctness: the poll below decides.
    const unsigned long long t_start = wall_clock64();  // (100 MHz)
    unsigned long long t_last = t_start;
    uint32_t level_last = 0u;
    constexpr uint32_t PACE_MIN = 40u;  // 10 ns ticks per level: below what a one-contact chain's arithmetic costs
    uint32_t pace = PACE_MIN;
    for (uint32_t r = r0; r < r1; ++r) {
        const uint64_t mask = mask_next;
        const uint32_t level = level_next;
        mask_next = masks[r + 1u < r1 ? r + 1u : r];
        level_next = levels[r + 1u < r1 ? r + 1u : r];
        if (ph.nap) {
            const uint32_t gap10 = (level - level_last) * 10u;
            const unsigned long long due = t_last + (unsigned long long)(gap10 > ph.nap ? gap10 - ph.nap : 0u) * pace / 10ull;
            while (wall_clock64() < due) __builtin_amdgcn_s_sleep(8);
        }
        if (ph.trace && lane == 0u) ph.trace[4u * r] = wall_clock64();
        const bool active = has && ((mask >> lane) & 1ull) != 0ull;
        const uint32_t type = PHASE == 1 ? PHYS_ITEM_POSITIONAL : (sweep < ph.n_first ? PHYS_ITEM_WARM : PHYS_ITEM_VELOCITY);
        const uint32_t ver = sweep * deg + rank;
        // a kinematic own body: what it moves with / where it stands comes from the body array every round — constant during the solve, so
        // these are plain loads that the L1 serves after the first, issued ahead of the wait for the dynamic bodies
        V3 v = {}, w = {}, p = {};
        Q4 q = {};
        uint32_t rv_index = 0u;
        if (active && !dyn) {
            if (PHASE == 0) v = ld3(cb[own].v), w = ld3(cb[own].w);
            else {
                p = ld3(cb[own].pos), q = ldq(cb[own].q);
                if (ph.rv.qstart) {  // pass 2: the orientation as the chains before this one left it (ReplayView)
                    rv_index = ph.slot_of[(slot >> 1) * ph.n_passes + sweep];
                    const float4 t = ph.rv.qstart[2u * rv_index + (side_b ? 1u : 0u)];
                    q = Q4{t.x, t.y, t.z, t.w};
                }
            }
        }
        // THE HAND-OFF: the own body's records, past the L1, until they carry the version this sweep of the chain starts from. The loads of a
        // poll are issued back to back by every lane (a lane with nothing to wait for reads record 0 and ignores it): one trip per poll.
        const bool need = active && dyn;
        const uint32_t off = need ? own * DST : 0u;
        float4 a0, a1, a2 = make_float4(0, 0, 0, 0);
        for (uint32_t spins = 0;; ++spins) {
            a0 = ld16_sc1(rs_dyn, off);
            a1 = ld16_sc1(rs_dyn, off + 16u);
            uint32_t ok = (uint32_t)(__float_as_uint(a0.w) == ver) & (uint32_t)(__float_as_uint(a1.w) == ver);
            if (PHASE == 1) {
                a2 = ld16_sc1(rs_dyn, off + 32u);
                ok = ok & (uint32_t)(__float_as_uint(a2.w) == ver);
            }
            if ((dry & 2u) || __builtin_amdgcn_ballot_w64(need && !ok) == 0ull) break;
            __builtin_amdgcn_s_sleep(1);
            if (spins > MG_SPIN_LIMIT) {
                // A producer that never stored (a working workgroup that is not resident): the versions can never match again, so every later
                // round of this wave would spin the full limit too — hundreds of rounds, minutes. Flag the launch and leave the phase (there
                // is no barrier below this loop): every wave gives up within one spin limit of its own and ivx_world_check_solve reports.
                if (lane == 0u) __hip_atomic_store(error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // (host-mapped: ivx_world_check_solve)
                return;
            }
        }
        if (ph.trace && lane == 0u) ph.trace[4u * r + 1u] = wall_clock64();
        if (active) {
            const float nv = __uint_as_float(ver + 1u);
            if (PHASE == 0) {
                if (dyn) v = mk(a0.x, a0.y, a0.z), w = mk(a1.x, a1.y, a1.z);
                lane_contact_v(type, cv0, im, ii, flip, dyn, v, w, ac0);
                if (len > 1u) lane_contact_v(type, cv1, im, ii, flip, dyn, v, w, ac1);
                if (len > 2u) lane_contact_v(type, cv2, im, ii, flip, dyn, v, w, ac2);
                if (len > 3u) lane_contact_v(type, cv3, im, ii, flip, dyn, v, w, ac3);
                if (ph.trace && __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0u) ph.trace[4u * r + 2u] = wall_clock64();
                if (dyn) {
                    st16_shared(rs_dyn, own * DST, make_float4(v.x, v.y, v.z, nv), one_xcd);
                    st16_shared(rs_dyn, own * DST + 16u, make_float4(w.x, w.y, w.z, nv), one_xcd);
                }
            } else {
                if (dyn) p = mk(a0.x, a0.y, a0.z), q = Q4{a1.x, a1.y, a1.z, a2.x};
                uint32_t applied = 0u;
                lane_contact_p(cp0.n, cp0.local, im, im_sum, ii, flip, side_b, p, q, factor, applied);
                if (len > 1u) lane_contact_p(cp1.n, cp1.local, im, im_sum, ii, flip, side_b, p, q, factor, applied);
                if (len > 2u) lane_contact_p(cp2.n, cp2.local, im, im_sum, ii, flip, side_b, p, q, factor, applied);
                if (len > 3u) lane_contact_p(cp3.n, cp3.local, im, im_sum, ii, flip, side_b, p, q, factor, applied);
                if (ph.trace && __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0u) ph.trace[4u * r + 2u] = wall_clock64();
                if (ph.rv.applied && !both_dyn && !side_b)  // pass 1: corrections a chain with a kinematic body applied (A's lane reports)
                    ph.rv.applied[ph.slot_of[(slot >> 1) * ph.n_passes + sweep]] = applied;
                if (dyn) {
                    st16_shared(rs_dyn, own * DST, make_float4(p.x, p.y, p.z, nv), one_xcd);
                    st16_shared(rs_dyn, own * DST + 16u, make_float4(q.x, q.y, q.z, nv), one_xcd);
                    st16_shared(rs_dyn, own * DST + 32u, make_float4(q.w, 0.0f, 0.0f, nv), one_xcd);
                }
            }
            if (ph.trace && __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0u) ph.trace[4u * r + 3u] = wall_clock64();
            sweep += 1u;
        }
        // the pace so far, by this wave's clock
        t_last = wall_clock64();
        level_last = level;
        const uint32_t measured = __builtin_amdgcn_readfirstlane((uint32_t)(t_last - t_start)) * 9u / (level * 10u);
        pace = measured > PACE_MIN ? measured : PACE_MIN;
    }
    // the accumulated impulses as the last sweep left them (next frame's warm start, ivx_world_contact_state): A's lane stores them
    if (PHASE == 0 && has && !side_b && sweep > ph.n_first) {
        ac0.w = ac1.w = ac2.w = ac3.w = 0.0f;
        accs[s0] = ac0;
        if (len > 1u) accs[s0 + 1u] = ac1;
        if (len > 2u) accs[s0 + 2u] = ac2;
        if (len > 3u) accs[s0 + 3u] = ac3;
    }
}

__global__ __launch_bounds__(CS_THREADS) void k_solve_cs(uint32_t n_dyn, float factor, const PhysContact* __restrict__ pcs, float4* __restrict__ accs,
                                                         uint32_t n_contacts, const PhysBody* __restrict__ cb, CsPhase vel, CsPhase pos,
                                                         uint32_t* __restrict__ error, uint32_t dry, uint32_t spread) {
    // spread > 1: block b is workgroup b / spread of the velocity phase when b % spread == 0, of the positional phase when b % spread == 1, and
    // nothing otherwise; spread == 1 (developer switch): the velocity phase's workgroups first, then the positional phase's
    uint32_t phase, wg;
    if (spread > 1u) {
        phase = blockIdx.x % spread;
        wg = blockIdx.x / spread;
    } else {
        phase = blockIdx.x < vel.n_groups ? 0u : 1u;
        wg = phase ? blockIdx.x - vel.n_groups : blockIdx.x;
    }
    if (phase == 0u && wg < vel.n_groups) solve_cs_phase<0>(wg, n_dyn, factor, pcs, accs, n_contacts, cb, vel, error, dry);
    else if (phase == 1u && wg < pos.n_groups) solve_cs_phase<1>(wg, n_dyn, factor, pcs, accs, n_contacts, cb, pos, error, dry);
}

// the phases' shared body records before the solve (tag 0: no item has touched the body) ...
__global__ __launch_bounds__(256) void k_cs_init(uint32_t n_dyn, const PhysBody* __restrict__ cb, float4* __restrict__ dyn_vel, float4* __restrict__ dyn_pos,
                                                 const uint32_t* __restrict__ flag) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n_dyn || (flag && *flag == 0u)) return;
    const PhysBody& b = cb[i];
    if (dyn_vel) {
        dyn_vel[2u * i] = make_float4(b.v[0], b.v[1], b.v[2], 0.0f);
        dyn_vel[2u * i + 1u] = make_float4(b.w[0], b.w[1], b.w[2], 0.0f);
    }
    if (dyn_pos) {
        dyn_pos[3u * i] = make_float4(b.pos[0], b.pos[1], b.pos[2], 0.0f);
        dyn_pos[3u * i + 1u] = make_float4(b.q[0], b.q[1], b.q[2], 0.0f);
        dyn_pos[3u * i + 2u] = make_float4(b.q[3], 0.0f, 0.0f, 0.0f);
    }
}
// ... and back into the constrained bodies after it (the array the write-back reads)
__global__ __launch_bounds__(256) void k_cs_finish(uint32_t n_dyn, PhysBody* __restrict__ cb, const float4* __restrict__ dyn_vel, const float4* __restrict__ dyn_pos) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n_dyn) return;
    PhysBody& b = cb[i];
    if (dyn_vel) {
        const float4 r0 = dyn_vel[2u * i], r1 = dyn_vel[2u * i + 1u];
        st3(b.v, mk(r0.x, r0.y, r0.z));
        st3(b.w, mk(r1.x, r1.y, r1.z));
    }
    if (dyn_pos) {
        const float4 r0 = dyn_pos[3u * i], r1 = dyn_pos[3u * i + 1u], r2 = dyn_pos[3u * i + 2u];
        st3(b.pos, mk(r0.x, r0.y, r0.z));
        stq(b.q, Q4{r1.x, r1.y, r1.z, r2.x});
    }
}

// ---- kinematic orientations in the positional phase (ReplayView) --------------------------------------------------------------------
// the dynamic bodies' configuration as the phase finds it, for pass 2 to start from again
__global__ __launch_bounds__(256) void k_kin_snapshot(uint32_t n_dyn, const PhysBody* __restrict__ cb, float4* __restrict__ snap) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n_dyn) return;
    snap[2u * i] = make_float4(cb[i].pos[0], cb[i].pos[1], cb[i].pos[2], 0.0f);
    snap[2u * i + 1u] = make_float4(cb[i].q[0], cb[i].q[1], cb[i].q[2], cb[i].q[3]);
}
__global__ __launch_bounds__(256) void k_kin_restore(uint32_t n_dyn, PhysBody* __restrict__ cb, const float4* __restrict__ snap, const uint32_t* __restrict__ replay_flag) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n_dyn || *replay_flag == 0u) return;
    const float4 a = snap[2u * i], b = snap[2u * i + 1u];
    st3(cb[i].pos, mk(a.x, a.y, a.z));
    stq(cb[i].q, Q4{b.x, b.y, b.z, b.w});
}
// one WAVE per kinematic body: walks its positional chains in solve order (kin_list: item | side << 31, CSR by kin_offsets) with the
// body's orientation in hand — every chain gets the orientation it starts from, then the orientation takes the re-normalisations the chain
// applied in pass 1. 64 chains are fetched at a time; while the orientation still moves they are taken one by one (every lane follows the
// same orientation), and once it has reached a fixed point — two times in three it is one from the start, else after a step or two as a
// rule; sequences that keep moving for hundreds of steps exist and are followed to the end — the remaining chains all start from it and
// are written 64 at a time (a ground plane under a pile has thousands of chains). What is left is the orientation the reference writes
// back (post_solve_body reads it from the body's record); the flag says whether any orientation moved at all (else pass 2 changes nothing).
__global__ __launch_bounds__(64) void k_kin_prefix(uint32_t n_kin, uint32_t n_dyn, const uint32_t* __restrict__ kin_offsets, const uint32_t* __restrict__ kin_list,
                                                   const uint32_t* __restrict__ applied, float4* __restrict__ qstart, PhysBody* __restrict__ cb,
                                                   uint32_t* __restrict__ replay_flag) {
    const uint32_t k = blockIdx.x, lane = threadIdx.x;
    if (k >= n_kin) return;
    Q4 q = ldq(cb[n_dyn + k].q);
    bool fixed = false, moved = false;
    const uint32_t j0 = kin_offsets[k], j1 = kin_offsets[k + 1u];
    for (uint32_t base = j0; base < j1; base += 64u) {
        const uint32_t j = base + lane;
        const bool live = j < j1;
        const uint32_t e = live ? kin_list[j] : 0u;
        const uint32_t slot = 2u * (e & 0x7FFFFFFFu) + (e >> 31);
        const uint32_t n_app = live ? applied[e & 0x7FFFFFFFu] : 0u;
        const uint32_t cnt = min(64u, j1 - base);
        uint32_t l = 0;
        for (; l < cnt && !fixed; ++l) {  // (uniform: every lane carries the same q)
            if (lane == l) qstart[slot] = make_float4(q.x, q.y, q.z, q.w);
            for (uint32_t n = (uint32_t)__builtin_amdgcn_readlane((int)n_app, (int)l); n > 0u && !fixed; --n) {
                const Q4 r = qnormalize(q);
                fixed = __float_as_uint(r.x) == __float_as_uint(q.x) && __float_as_uint(r.y) == __float_as_uint(q.y) && __float_as_uint(r.z) == __float_as_uint(q.z) &&
                        __float_as_uint(r.w) == __float_as_uint(q.w);
                moved = moved || !fixed;
                q = r;
            }
        }
        if (live && lane >= l) qstart[slot] = make_float4(q.x, q.y, q.z, q.w);  // (the orientation no longer moves)
    }
    if (lane == 0u) {
        stq(cb[n_dyn + k].q, q);
        if (moved) atomicOr(replay_flag, 1u);
    }
}

// dynamic bodies a joint is anchored to are constrained bodies of the step (prepare_spherical_joint -> add_body_pair, solver.rs:182-215)
// (both kinds bounds-checked here too: the references outlive the body set they were validated against only until ivx_world_set_bodies,
// which drops them when a count shrinks below one of them)
__global__ __launch_bounds__(256) void k_mark_bodies(uint32_t n, const uint32_t* __restrict__ refs, uint32_t n_dyn, uint32_t n_kin, uint8_t* __restrict__ touched) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint32_t r = refs[i];
    if (!(r & IVX_KINEMATIC_BODY) && r < n_dyn) touched[r] = 1;
    if ((r & IVX_KINEMATIC_BODY) && (r & 0x7FFFFFFFu) < n_kin) touched[n_dyn + (r & 0x7FFFFFFFu)] = 1;  // (kinematic anchors are constrained bodies as well)
}

}  // namespace

int ivx_launch_phys_prepare_bodies(ivx_world* w) {
    const uint32_t n = w->n_dyn + w->n_kin;
    if (n == 0) return IVX_OK;
    IVX_KLAUNCH(k_prepare_bodies, dim3((n + 255u) / 256u), dim3(256), 0, w->ctx->stream, w->n_dyn, w->n_kin, w->dyn, w->kin, w->cb, w->touched);
    IVX_HIP_CHECK(hipGetLastError());
    return IVX_OK;
}

int ivx_launch_phys_prepare_contacts(ivx_world* w, const int32_t* d_prev_slot) {
    if (w->n_contacts == 0) return IVX_OK;
    IVX_KLAUNCH(k_prepare_contacts, dim3((w->n_contacts + 255u) / 256u), dim3(256), 0, w->ctx->stream, w->n_contacts, w->n_dyn, w->contacts,
                       d_prev_slot, w->cb, w->pc[w->cur ^ 1], reinterpret_cast<const float4*>(w->acc[w->cur ^ 1]), w->n_prev,
                       w->cfg.old_impulse_weight, w->pc[w->cur], reinterpret_cast<float4*>(w->acc[w->cur]), w->touched);
    IVX_HIP_CHECK(hipGetLastError());
    return IVX_OK;
}

int ivx_launch_phys_mark_joint_bodies(ivx_world* w) {
    if (w->n_joint_refs == 0) return IVX_OK;
    IVX_KLAUNCH(k_mark_bodies, dim3((w->n_joint_refs + 255u) / 256u), dim3(256), 0, w->ctx->stream, w->n_joint_refs, w->joint_refs, w->n_dyn, w->n_kin, w->touched);
    IVX_HIP_CHECK(hipGetLastError());
    return IVX_OK;
}

int ivx_launch_phys_pre_solve(ivx_world* w, float dt) {
    if (w->n_dyn == 0) return IVX_OK;
    IVX_KLAUNCH(k_pre_solve, dim3((w->n_dyn + 255u) / 256u), dim3(256), 0, w->ctx->stream, w->n_dyn, dt, w->dyn, w->cb);
    IVX_HIP_CHECK(hipGetLastError());
    return IVX_OK;
}

template <bool LDS, int PHASE>
static int launch_solve(ivx_world* w, size_t lds, ReplayView rv = ReplayView(), const uint32_t* replay_flag = nullptr) {
    if (LDS) IVX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_solve<LDS, PHASE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    IVX_KLAUNCH((k_solve<LDS, PHASE>), dim3(1), dim3(SOLVE_THREADS), LDS ? lds : 0, w->ctx->stream, w->n_dyn, w->cfg.positional_correction_factor, w->pc[w->cur],
                       reinterpret_cast<float4*>(w->acc[w->cur]), w->cb, w->items + w->item_offset[PHASE],
                       reinterpret_cast<const uint2*>(w->item_bodies) + w->item_offset[PHASE], w->level_start + w->level_offset[PHASE],
                       w->n_levels[PHASE], rv, replay_flag);
    IVX_HIP_CHECK(hipGetLastError());
    return IVX_OK;
}

// number of workgroups the solve is spread over: as many as the widest level fills with one chain per thread, at most 16 (all
// resident: 256 CUs); 1 = the single-workgroup kernel with the bodies in LDS. ivx_world_set_solver_groups overrides (tests force
// either path on small scenes).
// (co-residency of the G working workgroups is what the grid barrier rests on: G x spread blocks must fit the device beside nothing else of
// this launch — checked against the occupancy query; a world whose barrier ever timed out stays on one workgroup)
static uint32_t solver_groups(const ivx_world* w) {
    if (w->mg_disabled) return 1u;
    static const int resident = [] {
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_solve_mg<0>, (int)MG_THREADS, 0) != hipSuccess) per_cu = 0;
        int per_cu1 = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu1, k_solve_mg<1>, (int)MG_THREADS, 0) != hipSuccess) per_cu1 = 0;
        return per_cu < per_cu1 ? per_cu : per_cu1;
    }();
    const uint32_t fit = (uint32_t)(resident > 0 ? resident : 0) * (uint32_t)w->ctx->n_cu / 8u;  // (/ 8: every eighth block works, see k_solve_mg)
    if (fit < 2u) return 1u;
    if (w->solver_groups_forced && w->solver_groups_forced <= 16u) return w->solver_groups_forced < fit ? w->solver_groups_forced : fit;
    const uint32_t widest = w->max_level_items[0] > w->max_level_items[1] ? w->max_level_items[0] : w->max_level_items[1];
    if (widest <= 256u) return 1u;  // (a level that fits one workgroup at one wave per SIMD gains nothing from more)
    uint32_t g = (widest + 159u) / 160u;  // (a few more waves than the widest level has tiles: measured on the 4096-body pile, 8 workgroups 0.85 ms, 5: 0.89, 16: 0.86)
    g = g < 16u ? g : 16u;
    return g < fit ? g : fit;
}

// developer switch (never set in production): IVX_SOLVER_DRY bit 0 = walk the levels without running the chains, bit 1 = without the grid
// barrier — how tools/time_pile.py splits a level's time into arithmetic and synchronisation (results are garbage with either)
static uint32_t ivx_solver_dry() {
    static const uint32_t v = [] {
        const char* e = getenv("IVX_SOLVER_DRY");
        return e ? (uint32_t)atoi(e) : 0u;
    }();
    return v;
}

// this step's prepared contacts and body constants into the schedule's packed records (run_chain_mg; the positional phase's second pass reads
// the first's). Reads the body array only: both phases' records are packed before either phase runs.
template <int PHASE>
static int pack_items_mg(ivx_world* w, hipStream_t stream) {
    const size_t need = (size_t)w->n_tiles[PHASE] * Packed<PHASE>::NJ * 64u * sizeof(float4);
    if (need > w->packed_cap[PHASE]) {
        IVX_HIP_CHECK(ivx_stream_sync(w->ctx->stream));
        if (w->side_stream) IVX_HIP_CHECK(ivx_stream_sync(w->side_stream));
        if (w->packed[PHASE]) (void)hipFree(w->packed[PHASE]);
        w->packed[PHASE] = nullptr;
        w->packed_cap[PHASE] = 0;
        IVX_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&w->packed[PHASE]), need + need / 8));
        w->packed_cap[PHASE] = need + need / 8;
    }
    IVX_KLAUNCH((k_pack_items<PHASE>), dim3(w->n_tiles[PHASE]), dim3(64), 0, stream, w->tile_first + w->tile_offset[PHASE], w->items + w->item_offset[PHASE],
                       reinterpret_cast<const uint2*>(w->item_bodies) + w->item_offset[PHASE], reinterpret_cast<const uint4*>(w->item_tags) + w->item_offset[PHASE],
                       w->pc[w->cur], w->cb, reinterpret_cast<float4*>(w->packed[PHASE]));
    IVX_HIP_CHECK(hipGetLastError());
    return IVX_OK;
}

// One phase's solve on `stream`. Everything a launch shares between its workgroups is the phase's own — barrier counter and census table
// (barrier_words[PHASE], + 4 + 20 PHASE), shared body records (dynst, second half for the positional phase) — because the two phases run side
// by side (ivx_launch_phys_solve).
template <int PHASE>
static int launch_solve_mg(ivx_world* w, uint32_t groups, hipStream_t stream, ReplayView rv = ReplayView(), const uint32_t* replay_flag = nullptr) {
    uint32_t& count = PHASE ? w->barrier_count1 : w->barrier_count;
    const uint32_t base = count;
    count += groups * 2u;  // (two grid barriers per launch: behind the set-up and census, and before the write-back)
    static const uint32_t spread = [] {  // (developer switch; 8 = the working workgroups share one XCD, 1 = consecutive blocks; see k_solve_mg)
        const char* e = getenv("IVX_SOLVER_SPREAD");
        const int v = e ? atoi(e) : 8;
        return (uint32_t)(v < 1 ? 1 : (v > 8 ? 8 : v));
    }();
    IVX_KLAUNCH((k_solve_mg<PHASE>), dim3(groups * spread), dim3(MG_THREADS), 0, stream, w->n_dyn, w->cfg.positional_correction_factor, w->pc[w->cur],
                       reinterpret_cast<float4*>(w->acc[w->cur]), w->n_contacts, w->cb, reinterpret_cast<float4*>(w->dynst + (PHASE ? w->body_cap * 16 : 0)),
                       w->items + w->item_offset[PHASE],
                       reinterpret_cast<const uint2*>(w->item_bodies) + w->item_offset[PHASE], w->level_start + w->level_offset[PHASE],
                       w->tile_base + w->level_offset[PHASE], reinterpret_cast<const float4*>(w->packed[PHASE]),
                       w->n_levels[PHASE], w->barrier_words + PHASE, base, w->mg_err_dev, ivx_solver_dry(), spread, rv, replay_flag, w->barrier_words + 4 + 20 * PHASE,
                       w->tile_first + w->tile_offset[PHASE], w->n_tiles[PHASE]);
    IVX_HIP_CHECK(hipGetLastError());
    return IVX_OK;
}

// The chain-stationary solve: can this world's schedule run on it, and should it? Needs every phase that has levels in stationary form
// (build_stationary), one workgroup of CS_THREADS per CU and an eighth of the CUs (one XCD's worth) per phase for its workgroups.
static bool solver_stationary(const ivx_world* w) {
    if (w->mg_disabled || w->solver_groups_forced == 1u || (w->solver_groups_forced >= 2u && w->solver_groups_forced <= 16u)) return false;
    for (int p = 0; p < 2; ++p)
        if (w->n_levels[p] && !w->cs_feasible[p]) return false;
    if (!w->n_levels[0] && !w->n_levels[1]) return false;
    static const int per_cu = [] {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_solve_cs, (int)CS_THREADS, 0) != hipSuccess) n = 0;
        return n;
    }();
    if (per_cu < 1) return false;
    for (int p = 0; p < 2; ++p)
        if (w->n_levels[p] && (((uint32_t)w->chain_start.size() - 1u + 31u) / 32u + PHYS_CS_WAVES - 1u) / PHYS_CS_WAVES > (uint32_t)w->ctx->n_cu / 8u) return false;
    if (w->solver_groups_forced == PHYS_SOLVER_STATIONARY) return true;
    // One workgroup with the bodies in LDS (k_solve) walks a level in ~4-5 us whatever its width — a barrier and the contacts' trip from memory —,
    // the chain-stationary solve in ~2.3 (velocity) / ~3.5 (positional) plus ~25 us of launches and census around it: it wins wherever the chain of
    // levels is long, wide or not (81 bodies of 48 contacts each on a ground plane: 297 + 99 levels of at most 79 chains, 0.55 -> 0.21 ms;
    // tools/time_world_frames.py). A handful of levels stays with the one workgroup.
    const uint32_t widest = w->max_level_items[0] > w->max_level_items[1] ? w->max_level_items[0] : w->max_level_items[1];
    return widest > 256u || w->n_levels[0] + w->n_levels[1] >= 48u;
}

static int launch_solve_cs(ivx_world* w) {
    hipStream_t s = w->ctx->stream;
    const bool replay = w->n_levels[1] && w->n_kin_items > 0;
    uint32_t* flag = w->barrier_words + 2;
    float4* dyn_vel = reinterpret_cast<float4*>(w->dynst);
    float4* dyn_pos = reinterpret_cast<float4*>(w->dynst + w->body_cap * 16);
    static const uint32_t spread = [] {  // (developer switch; 8 = a phase's workgroups share one XCD under round-robin placement, 1 = consecutive blocks)
        const char* e = getenv("IVX_SOLVER_SPREAD");
        const int v = e ? atoi(e) : 8;
        return (uint32_t)(v < 2 ? 1 : 8);
    }();
    // developer switch: IVX_SOLVER_TRACE=<file> — every round's four clock stamps of the first solve launch of each step, dumped as
    // [phase][round][4] u64 behind a header of {rounds of phase 0, rounds of phase 1} (tools/solver_trace.py reads it); the step then waits
    static const char* trace_path = getenv("IVX_SOLVER_TRACE");
    static unsigned long long* trace_dev = nullptr;
    static size_t trace_cap = 0;
    const size_t n_rounds[2] = {w->n_levels[0] ? w->cs_round_start_host[w->cs[0].round_start_offset + w->cs[0].n_tiles] : 0u,
                                w->n_levels[1] ? w->cs_round_start_host[w->cs[1].round_start_offset + w->cs[1].n_tiles] : 0u};
    if (trace_path && 4 * (n_rounds[0] + n_rounds[1]) > trace_cap) {
        if (trace_dev) (void)hipFree(trace_dev);
        trace_cap = 4 * (n_rounds[0] + n_rounds[1]);
        IVX_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&trace_dev), trace_cap * 8));
    }
    if (trace_path) IVX_HIP_CHECK(ivx_memset_async(trace_dev, 0, trace_cap * 8, s));
    static const uint32_t nap = [] {  // (developer switch: IVX_SOLVER_NAP = tenths of a level a wave wakes ahead of its round; 0 = never sleeps)
        const char* e = getenv("IVX_SOLVER_NAP");
        return (uint32_t)(e ? atoi(e) : 20);
    }();
    auto phase_args = [&](int p, bool on, const ReplayView& rv, const uint32_t* replay_flag) {
        CsPhase a = {};
        if (!on || !w->n_levels[p]) return a;
        const ivx_world::CsSchedule& cs = w->cs[p];
        a.item = w->cs_item + cs.slot_offset;
        a.bodies = reinterpret_cast<const uint2*>(w->cs_bodies) + cs.slot_offset;
        a.vers = w->cs_vers + cs.slot_offset;
        a.round_start = w->cs_round_start + cs.round_start_offset;
        a.round_mask = w->cs_round_mask + cs.round_offset;
        a.round_level = w->cs_round_level + cs.round_offset;
        a.nap = nap;
        a.dynst = p ? dyn_pos : dyn_vel;
        a.counter = w->barrier_words + p;
        a.xcc_table = w->barrier_words + 64 + 32 * p;
        a.slot_of = w->cs_slot_of;
        a.replay_flag = replay_flag;
        a.trace = trace_path && !replay_flag ? trace_dev + (p ? 4 * n_rounds[0] : 0) : nullptr;
        a.rv = rv;
        a.n_tiles = cs.n_tiles;
        a.n_groups = (cs.n_tiles + PHYS_CS_WAVES - 1u) / PHYS_CS_WAVES;
        a.n_first = p ? 0u : 1u;
        a.n_passes = p ? w->cfg.n_positional_correction_iterations : w->cfg.n_iterations + 1u;
        uint32_t& count = p ? w->barrier_count1 : w->barrier_count;
        a.counter_base = count;
        count += a.n_groups;  // (one grid barrier per launch: behind the census)
        return a;
    };
    auto launch = [&](const CsPhase& v, const CsPhase& p) -> int {
        const uint32_t blocks = spread > 1u ? spread * (v.n_groups > p.n_groups ? v.n_groups : p.n_groups) : v.n_groups + p.n_groups;
        if (!blocks) return IVX_OK;
        IVX_KLAUNCH(k_solve_cs, dim3(blocks), dim3(CS_THREADS), 0, s, w->n_dyn, w->cfg.positional_correction_factor, w->pc[w->cur],
                    reinterpret_cast<float4*>(w->acc[w->cur]), w->n_contacts, w->cb, v, p, w->mg_err_dev, ivx_solver_dry(), spread);
        IVX_HIP_CHECK(hipGetLastError());
        return IVX_OK;
    };
    const uint32_t body_blocks = (w->n_dyn + 255u) / 256u;
    ReplayView pass1, pass2;
    if (replay) {
        pass1.applied = w->kin_applied;
        pass2.qstart = reinterpret_cast<const float4*>(w->kin_qstart);
        IVX_HIP_CHECK(ivx_memset_async(flag, 0, sizeof(uint32_t), s));
    }
    if (body_blocks) {
        IVX_KLAUNCH(k_cs_init, dim3(body_blocks), dim3(256), 0, s, w->n_dyn, w->cb, w->n_levels[0] ? dyn_vel : nullptr, w->n_levels[1] ? dyn_pos : nullptr, nullptr);
        IVX_HIP_CHECK(hipGetLastError());
    }
    const CsPhase v = phase_args(0, true, ReplayView(), nullptr), p1 = phase_args(1, true, pass1, nullptr);
    w->solver_groups_used = v.n_groups + p1.n_groups;
    int rc;
    if ((rc = launch(v, p1))) return rc;
    if (replay) {  // the positional phase again where a kinematic orientation moved (ReplayView): its bodies' records from the untouched body array
        IVX_KLAUNCH(k_kin_prefix, dim3(w->n_kin), dim3(64), 0, s, w->n_kin, w->n_dyn, w->kin_offsets, w->kin_list, w->kin_applied,
                    reinterpret_cast<float4*>(w->kin_qstart), w->cb, flag);
        if (body_blocks) IVX_KLAUNCH(k_cs_init, dim3(body_blocks), dim3(256), 0, s, w->n_dyn, w->cb, nullptr, dyn_pos, flag);
        IVX_HIP_CHECK(hipGetLastError());
        const CsPhase none = phase_args(0, false, ReplayView(), nullptr), p2 = phase_args(1, true, pass2, flag);
        if ((rc = launch(none, p2))) return rc;
    }
    if (trace_path) {
        IVX_HIP_CHECK(ivx_stream_sync(s));
        std::vector<unsigned long long> host(trace_cap + 2);
        host[0] = n_rounds[0], host[1] = n_rounds[1];
        IVX_HIP_CHECK(ivx_memcpy_sync(host.data() + 2, trace_dev, trace_cap * 8, hipMemcpyDeviceToHost));
        if (FILE* f = fopen(trace_path, "wb")) {
            fwrite(host.data(), 8, 2 + 4 * (n_rounds[0] + n_rounds[1]), f);
            for (int p = 0; p < 2; ++p)
                if (n_rounds[p]) fwrite(w->cs_round_level_host.data() + w->cs[p].round_offset, 4, n_rounds[p], f);
            fclose(f);
        }
    }
    if (body_blocks) {
        IVX_KLAUNCH(k_cs_finish, dim3(body_blocks), dim3(256), 0, s, w->n_dyn, w->cb, w->n_levels[0] ? dyn_vel : nullptr, w->n_levels[1] ? dyn_pos : nullptr);
        IVX_HIP_CHECK(hipGetLastError());
    }
    return IVX_OK;
}

int ivx_launch_phys_solve(ivx_world* w) {
    if (w->n_contacts == 0) return IVX_OK;
    int rc;
    if (solver_stationary(w)) {
        if ((rc = ivx_world_ensure_form(w, 2u))) return rc;
        w->solver_kind_used = 2u;
        return launch_solve_cs(w);
    }
    if ((rc = ivx_world_ensure_form(w, 1u))) return rc;
    const uint32_t groups = solver_groups(w);
    w->solver_groups_used = groups;
    w->solver_kind_used = groups > 1u ? 1u : 0u;
    // the positional phase runs twice when kinematic bodies take part in it (ReplayView): counts, then — where an orientation moves — again
    const bool replay = w->n_levels[1] && w->n_kin_items > 0;
    uint32_t* flag = w->barrier_words + 2;
    ReplayView pass1, pass2;
    if (replay) {
        pass1.applied = w->kin_applied;
        pass2.qstart = reinterpret_cast<const float4*>(w->kin_qstart);
    }
    auto before_positional = [&](hipStream_t st) -> int {
        if (!replay) return IVX_OK;
        IVX_HIP_CHECK(ivx_memset_async(flag, 0, sizeof(uint32_t), st));
        IVX_KLAUNCH(k_kin_snapshot, dim3((w->n_dyn + 255u) / 256u + 1u), dim3(256), 0, st, w->n_dyn, w->cb, reinterpret_cast<float4*>(w->kin_snap));
        IVX_HIP_CHECK(hipGetLastError());
        return IVX_OK;
    };
    auto between_passes = [&](hipStream_t st) -> int {
        IVX_KLAUNCH(k_kin_prefix, dim3(w->n_kin), dim3(64), 0, st, w->n_kin, w->n_dyn, w->kin_offsets, w->kin_list, w->kin_applied,
                           reinterpret_cast<float4*>(w->kin_qstart), w->cb, flag);
        IVX_KLAUNCH(k_kin_restore, dim3((w->n_dyn + 255u) / 256u + 1u), dim3(256), 0, st, w->n_dyn, w->cb, reinterpret_cast<const float4*>(w->kin_snap), flag);
        IVX_HIP_CHECK(hipGetLastError());
        return IVX_OK;
    };
    if (groups > 1u) {
        // THE TWO PHASES SIDE BY SIDE. The velocity phase reads and writes velocities (and the accumulated impulses); the positional phase reads
        // and writes positions and orientations; what either reads of the other's — the configuration the velocity items are linearised
        // about, inverse masses and inertia — is in its packed records, taken from the body array before either starts (the reference runs the
        // positional correction on the configuration the step began with, solver.rs:496-602, and integrates afterwards). Each phase is a chain
        // of dependent tiles a handful of waves deep — 183 and 147 levels on the 4096-body pile, of which 135 are the fill of one sweep over
        // the lattice —, so one after the other they cost the sum, on two streams the longer one.
        hipStream_t s0 = w->ctx->stream, s1 = s0;
        static const bool serial = [] {
            const char* e = getenv("IVX_SOLVER_SERIAL");  // (developer switch: the phases one after the other on the context's stream)
            return e && atoi(e) != 0;
        }();
        const bool both = w->n_levels[0] && w->n_levels[1] && !serial;
        if (both) {
            if (!w->side_stream) {
                IVX_HIP_CHECK(hipStreamCreateWithFlags(&w->side_stream, hipStreamNonBlocking));
                IVX_HIP_CHECK(hipEventCreateWithFlags(&w->ev_fork, hipEventDisableTiming));
                IVX_HIP_CHECK(hipEventCreateWithFlags(&w->ev_join, hipEventDisableTiming));
            }
            s1 = w->side_stream;
        }
        if (w->n_levels[0] && (rc = pack_items_mg<0>(w, s0))) return rc;
        if (w->n_levels[1] && (rc = pack_items_mg<1>(w, s0))) return rc;
        if (both) {
            IVX_HIP_CHECK(ivx_event_record(w->ev_fork, s0));
            IVX_HIP_CHECK(hipStreamWaitEvent(s1, w->ev_fork, 0));
        }
        if (w->n_levels[1] && both) {  // (the side stream first: its launches are in flight while the host enqueues the velocity phase)
            if ((rc = before_positional(s1))) return rc;
            if ((rc = launch_solve_mg<1>(w, groups, s1, pass1))) return rc;
            if (replay) {
                if ((rc = between_passes(s1))) return rc;
                if ((rc = launch_solve_mg<1>(w, groups, s1, pass2, flag))) return rc;
            }
            IVX_HIP_CHECK(ivx_event_record(w->ev_join, s1));
        }
        if (w->n_levels[0] && (rc = launch_solve_mg<0>(w, groups, s0))) return rc;
        if (w->n_levels[1] && !both) {
            if ((rc = before_positional(s0))) return rc;
            if ((rc = launch_solve_mg<1>(w, groups, s0, pass1))) return rc;
            if (replay) {
                if ((rc = between_passes(s0))) return rc;
                if ((rc = launch_solve_mg<1>(w, groups, s0, pass2, flag))) return rc;
            }
        }
        if (both) IVX_HIP_CHECK(hipStreamWaitEvent(s0, w->ev_join, 0));
        return IVX_OK;
    }
    // the phase's mutable body state (24 / 28 bytes per dynamic body) goes to LDS when it fits one CU
    const size_t lds0 = (size_t)w->n_dyn * 24, lds1 = (size_t)w->n_dyn * 28;
    if (w->n_levels[0]) {
        if (lds0 <= 140 * 1024) rc = launch_solve<true, 0>(w, lds0);
        else rc = launch_solve<false, 0>(w, 0);
        if (rc) return rc;
    }
    if (w->n_levels[1]) {
        if ((rc = before_positional(w->ctx->stream))) return rc;
        if (lds1 <= 140 * 1024) rc = launch_solve<true, 1>(w, lds1, pass1);
        else rc = launch_solve<false, 1>(w, 0, pass1);
        if (rc) return rc;
        if (replay) {
            if ((rc = between_passes(w->ctx->stream))) return rc;
            if (lds1 <= 140 * 1024) rc = launch_solve<true, 1>(w, lds1, pass2, flag);
            else rc = launch_solve<false, 1>(w, 0, pass2, flag);
            if (rc) return rc;
        }
    }
    return IVX_OK;
}

int ivx_launch_phys_free_step(ivx_world* w, float dt) {
    const uint32_t n = w->n_dyn + w->n_kin;
    if (n == 0) return IVX_OK;
    IVX_KLAUNCH(k_free_step, dim3((n + 255u) / 256u), dim3(256), 0, w->ctx->stream, w->n_dyn, w->n_kin, dt, w->dyn, w->kin, w->cb, w->touched);
    IVX_HIP_CHECK(hipGetLastError());
    return IVX_OK;
}

int ivx_launch_phys_post_solve(ivx_world* w, float dt, int write_back, int advance) {
    const uint32_t n = w->n_dyn + w->n_kin;
    if (n == 0) return IVX_OK;
    IVX_KLAUNCH(k_post_solve, dim3((n + 255u) / 256u), dim3(256), 0, w->ctx->stream, w->n_dyn, w->n_kin, dt, write_back, advance, w->cb, w->touched,
                       w->dyn, w->kin);
    IVX_HIP_CHECK(hipGetLastError());
    return IVX_OK;
}
