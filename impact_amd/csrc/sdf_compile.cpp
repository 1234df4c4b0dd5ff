// Host-side compile of an atomic SDF graph into the flat, post-ordered node program the sampling
// kernel executes. Product-side counterpart of the reference's SDFGenerator::new_in and
// determine_transforms_and_margins (engine/crates/impact_voxel/src/generation/sdf/atomic.rs:228-596)
// and SDFVoxelGenerator::new (generation.rs:207-258). Pure host code (no GPU work): it is set-up time
// logic that the reference also runs once per object on the CPU.
//
// f32 arithmetic follows glam's operation order (see DESIGN.md "third-party arithmetic"); built with
// -ffp-contract=off.
#include <cmath>
#include <cstring>
#include <limits>
#include <vector>

#include "ivx_internal.hpp"

namespace {

struct F3 {
    float v[3];
};
struct Bounds {
    F3 lo, hi;
};

inline F3 f3(float a, float b, float c) { return F3{{a, b, c}}; }
inline F3 splat(float s) { return f3(s, s, s); }
template <class Op>
inline F3 zip(const F3& a, const F3& b, Op op) {
    return f3(op(a.v[0], b.v[0]), op(a.v[1], b.v[1]), op(a.v[2], b.v[2]));
}
inline F3 plus(const F3& a, const F3& b) { return zip(a, b, [](float x, float y) { return x + y; }); }
inline F3 minus(const F3& a, const F3& b) { return zip(a, b, [](float x, float y) { return x - y; }); }
inline F3 times(const F3& a, float s) { return f3(a.v[0] * s, a.v[1] * s, a.v[2] * s); }
inline F3 lower(const F3& a, const F3& b) { return zip(a, b, [](float x, float y) { return y < x ? y : x; }); }
inline F3 upper(const F3& a, const F3& b) { return zip(a, b, [](float x, float y) { return y > x ? y : x; }); }
inline bool any_sign_negative(const F3& a) { return std::signbit(a.v[0]) || std::signbit(a.v[1]) || std::signbit(a.v[2]); }
inline Bounds grown(const Bounds& b, float m) { return {minus(b.lo, splat(m)), plus(b.hi, splat(m))}; }
inline float dot3(const F3& a, const F3& b) { return (a.v[0] * b.v[0] + a.v[1] * b.v[1]) + a.v[2] * b.v[2]; }
inline F3 cross3(const F3& a, const F3& b) {
    return f3(a.v[1] * b.v[2] - b.v[1] * a.v[2], a.v[2] * b.v[0] - b.v[2] * a.v[0], a.v[0] * b.v[1] - b.v[0] * a.v[1]);
}

// glam Quat::mul_vec3a (SSE2): v (w^2 - b.b) + b (2 v.b) + (w (b x v)) 2
F3 rotate(const float q[4], const F3& v) {
    F3 b = f3(q[0], q[1], q[2]);
    float w = q[3];
    F3 t0 = times(v, w * w - dot3(b, b));
    F3 t1 = times(b, dot3(v, b) * 2.0f);
    F3 t2 = times(times(cross3(b, v), w), 2.0f);
    return plus(plus(t0, t1), t2);
}

// atomic.rs:1590-1598
float soft_padding(float smoothness, uint32_t leaves) { return (0.25f * smoothness) * std::log2((float)leaves); }

struct Compiler {
    const ivx_sdf_node* in;
    size_t n;
    std::vector<Bounds> domain;
    std::vector<uint32_t> leaves;
    std::vector<float> padding;
    std::vector<uint8_t> state;  // 0 new, 1 on the current path, 2 done
    std::vector<ivx_sdf_processed_node> out;
    std::vector<uint32_t> source;
    int depth = 0, max_depth = 0;
    bool ok = true;

    void determine(uint32_t id) {
        const ivx_sdf_node& nd = in[id];
        Bounds zero{splat(0.0f), splat(0.0f)};
        switch (nd.kind) {
            case 0: domain[id] = {splat(-nd.p[0]), splat(nd.p[0])}; leaves[id] = 1; break;
            case 1: {
                F3 h = splat(nd.p[1]);
                h.v[1] += 0.5f * nd.p[0];
                domain[id] = {times(h, -1.0f), h};
                leaves[id] = 1;
                break;
            }
            case 2: {
                F3 h = times(f3(nd.p[0], nd.p[1], nd.p[2]), 0.5f);
                domain[id] = {times(h, -1.0f), h};
                leaves[id] = 1;
                break;
            }
            case 3: {
                F3 t = f3(nd.p[0], nd.p[1], nd.p[2]);
                domain[id] = {plus(domain[nd.child1].lo, t), plus(domain[nd.child1].hi, t)};
                leaves[id] = leaves[nd.child1];
                padding[id] = padding[nd.child1];
                break;
            }
            case 4: {  // bounding box of the rotated child box (oriented_box.rs:62-214)
                const Bounds& c = domain[nd.child1];
                F3 centre = rotate(nd.p, times(plus(c.lo, c.hi), 0.5f));
                F3 he = times(minus(c.hi, c.lo), 0.5f);
                F3 ax[3] = {times(rotate(nd.p, f3(1, 0, 0)), he.v[0]), times(rotate(nd.p, f3(0, 1, 0)), he.v[1]),
                            times(rotate(nd.p, f3(0, 0, 1)), he.v[2])};
                Bounds r{};
                for (int corner = 0; corner < 8; ++corner) {
                    F3 p = centre;
                    for (int a = 0; a < 3; ++a) p = ((corner >> (2 - a)) & 1) ? plus(p, ax[a]) : minus(p, ax[a]);
                    if (corner == 0) r = {p, p};
                    else r = {lower(r.lo, p), upper(r.hi, p)};
                }
                domain[id] = r;
                leaves[id] = leaves[nd.child1];
                padding[id] = padding[nd.child1];
                break;
            }
            case 5:
                domain[id] = {times(domain[nd.child1].lo, nd.p[0]), times(domain[nd.child1].hi, nd.p[0])};
                leaves[id] = leaves[nd.child1];
                padding[id] = padding[nd.child1];
                break;
            case 7:
                domain[id] = {lower(domain[nd.child1].lo, domain[nd.child2].lo), upper(domain[nd.child1].hi, domain[nd.child2].hi)};
                break;
            case 8: domain[id] = domain[nd.child1]; break;
            case 9: {
                F3 lo = upper(domain[nd.child1].lo, domain[nd.child2].lo), hi = lower(domain[nd.child1].hi, domain[nd.child2].hi);
                domain[id] = any_sign_negative(minus(hi, lo)) ? zero : Bounds{lo, hi};
                break;
            }
            default: ok = false;
        }
        if (nd.kind >= 7 && nd.kind <= 9) {
            leaves[id] = leaves[nd.child1] + leaves[nd.child2];
            padding[id] = soft_padding(nd.p[0], leaves[id]);
        }
    }

    void visit(uint32_t id) {
        if (!ok) return;
        if (id >= n || state[id] == 1) {  // missing node or cycle (atomic.rs:261-269)
            ok = false;
            return;
        }
        const ivx_sdf_node& nd = in[id];
        const bool first = state[id] == 0;
        if (first) state[id] = 1;
        if (nd.kind >= 3 && nd.kind <= 5) visit(nd.child1);
        else if (nd.kind >= 7 && nd.kind <= 9) {
            visit(nd.child1);
            visit(nd.child2);
        } else if (nd.kind > 2) ok = false;
        if (!ok) return;
        if (state[id] != 2) {
            state[id] = 2;
            determine(id);
        }
        ivx_sdf_processed_node p;
        std::memset(&p, 0, sizeof(p));
        p.kind = nd.kind;
        p.leaf_count = leaves[id];
        Bounds padded = grown(domain[id], padding[id]);
        for (int d = 0; d < 3; ++d) {
            p.domain_lo[d] = padded.lo.v[d];
            p.domain_hi[d] = padded.hi.v[d];
        }
        switch (nd.kind) {
            case 0: p.a = nd.p[0]; break;
            case 1: p.a = 0.5f * nd.p[0]; p.b = nd.p[1]; break;
            case 2: p.a = 0.5f * nd.p[0]; p.b = 0.5f * nd.p[1]; p.c = 0.5f * nd.p[2]; break;
            case 3: p.a = nd.p[0]; p.b = nd.p[1]; p.c = nd.p[2]; break;
            case 5: p.a = nd.p[0]; break;
            case 7: case 8: case 9: p.a = nd.p[0]; p.b = 0.25f / nd.p[0]; break;
            default: break;
        }
        out.push_back(p);
        source.push_back(id);
        if (nd.kind <= 2) {
            depth += 1;
            if (depth > max_depth) max_depth = depth;
        } else if (nd.kind >= 7) {
            depth -= 1;
        }
    }
};

using Mat = float[16];  // column-major

void mat_identity(float* m) {
    std::memset(m, 0, 16 * sizeof(float));
    m[0] = m[5] = m[10] = m[15] = 1.0f;
}
// glam Mat4::from_quat of the conjugate, then Mat4 * Mat4 (columns: ((a0 x + a1 y) + a2 z) + a3 w)
void premultiply_inverse_rotation(const float q_in[4], float* m) {
    const float q[4] = {-q_in[0], -q_in[1], -q_in[2], q_in[3]};
    const float x2 = q[0] + q[0], y2 = q[1] + q[1], z2 = q[2] + q[2];
    const float xx = q[0] * x2, xy = q[0] * y2, xz = q[0] * z2, yy = q[1] * y2, yz = q[1] * z2, zz = q[2] * z2;
    const float wx = q[3] * x2, wy = q[3] * y2, wz = q[3] * z2;
    float r[16];
    mat_identity(r);
    r[0] = 1.0f - (yy + zz); r[1] = xy + wz; r[2] = xz - wy;
    r[4] = xy - wz; r[5] = 1.0f - (xx + zz); r[6] = yz + wx;
    r[8] = xz + wy; r[9] = yz - wx; r[10] = 1.0f - (xx + yy);
    float res[16];
    for (int col = 0; col < 4; ++col)
        for (int row = 0; row < 4; ++row) {
            float s = r[row] * m[col * 4];
            s = s + r[4 + row] * m[col * 4 + 1];
            s = s + r[8 + row] * m[col * 4 + 2];
            s = s + r[12 + row] * m[col * 4 + 3];
            res[col * 4 + row] = s;
        }
    std::memcpy(m, res, sizeof(res));
}

}  // namespace

extern "C" int ivx_sdf_compile(const ivx_sdf_node* nodes, size_t n_nodes, uint32_t root, ivx_sdf_processed_node* out, size_t cap,
                               size_t* n_out, float domain[6], uint32_t* stack_size) {
    IVX_REQUIRE(n_out && domain && stack_size, IVX_ERR_INVALID, "ivx_sdf_compile: null output pointer");
    *n_out = 0;
    *stack_size = 0;
    for (int d = 0; d < 6; ++d) domain[d] = 0.0f;
    if (n_nodes == 0) return IVX_OK;  // SDFGenerator::empty_in (atomic.rs:220-226)
    IVX_REQUIRE(nodes, IVX_ERR_INVALID, "ivx_sdf_compile: null node array");
    Compiler c;
    c.in = nodes;
    c.n = n_nodes;
    c.domain.assign(n_nodes, Bounds{splat(0.0f), splat(0.0f)});
    c.leaves.assign(n_nodes, 0);
    c.padding.assign(n_nodes, 0.0f);
    c.state.assign(n_nodes, 0);
    c.visit(root);
    IVX_REQUIRE(c.ok, IVX_ERR_INVALID, "ivx_sdf_compile: missing node, cycle or unsupported node kind in SDF graph");
    IVX_REQUIRE(c.out.size() <= cap, IVX_ERR_CAPACITY, "ivx_sdf_compile: %zu processed nodes exceed capacity %zu", c.out.size(), cap);

    // determine_transforms_and_margins (atomic.rs:495-596): walk parents before children
    const size_t m = c.out.size();
    std::vector<float> tstack(16 * m, 0.0f), mstack(m, 0.0f);
    size_t top = 0;
    mat_identity(&tstack[0]);
    mstack[0] = 0.02f * 127.0f;  // VoxelSignedDistance::MAX_F32 (lib.rs:158)
    for (size_t r = m; r-- > 0;) {
        ivx_sdf_processed_node& p = c.out[r];
        const ivx_sdf_node& src = nodes[c.source[r]];
        float transform[16];
        std::memcpy(transform, &tstack[16 * top], sizeof(transform));
        const float margin = mstack[top];
        std::memcpy(p.transform, transform, sizeof(transform));
        p.margin = margin;
        for (int d = 0; d < 3; ++d) {
            p.domain_lo[d] = p.domain_lo[d] - margin;
            p.domain_hi[d] = p.domain_hi[d] + margin;
        }
        float* cur = &tstack[16 * top];
        if (p.kind <= 2) {
            if (top > 0) top -= 1;
        } else if (p.kind == 3) {
            cur[12] += -src.p[0];
            cur[13] += -src.p[1];
            cur[14] += -src.p[2];
            cur[15] += 0.0f;
        } else if (p.kind == 4) {
            premultiply_inverse_rotation(src.p, cur);
        } else if (p.kind == 5) {
            const float inv = 1.0f / src.p[0];
            for (int col = 0; col < 4; ++col)
                for (int row = 0; row < 3; ++row) cur[col * 4 + row] = inv * cur[col * 4 + row];
            mstack[top] = margin / src.p[0];
        } else {
            std::memcpy(&tstack[16 * (top + 1)], transform, sizeof(transform));
            const float mc = margin + 2.5f * soft_padding(src.p[0], p.leaf_count);
            mstack[top] = mc;
            mstack[top + 1] = mc;
            top += 1;
        }
    }
    for (size_t i = 0; i < m; ++i) out[i] = c.out[i];
    *n_out = m;
    *stack_size = (uint32_t)c.max_depth;
    Bounds rootd = grown(c.domain[root], c.padding[root]);
    for (int d = 0; d < 3; ++d) {
        domain[d] = rootd.lo.v[d];
        domain[3 + d] = rootd.hi.v[d];
    }
    return IVX_OK;
}


// ---- annotations for the sampler's pre-pass (written into the uploaded copy of the program, reserved[0..4)) ---------------
// The program is in postfix order, so the subtree of node r is the contiguous range [first(r), r]. For every node:
//   reserved[0]  the value of the subtree when every node of it takes its "domain lies outside the block" early-out
//                (leaf -> +margin, atomic.rs:654-668; combination applied only if the folded value fails `>= margin`,
//                atomic.rs:788-806): the constant a far-away body contributes, as the pre-pass would fold it node by node
//   reserved[1]  for a leaf: the OUTERMOST subtree root whose range starts at this leaf (itself if none)
//   reserved[2]  for a non-leaf: the root of the next smaller subtree starting at the same leaf (first operand / the child)
//   reserved[3]  first(r)
// With these the pre-pass can replace a whole far body by one constant instead of walking its nodes.
static inline float rs_min(float a, float b) { return (b < a) ? b : a; }
static inline float rs_max(float a, float b) { return (b > a) ? b : a; }
static inline float host_smooth_union(float d1, float d2, float s, float q) {  // generation/sdf.rs:89-92
    const float h = rs_max(s - std::fabs(d1 - d2), 0.0f);
    return rs_min(d1, d2) - (h * h) * q;
}
static inline float host_combine(uint32_t kind, float a, float b, float s, float q) {
    if (kind == 7u) return s == 0.0f ? rs_min(a, b) : host_smooth_union(a, b, s, q);
    if (kind == 8u) return s == 0.0f ? rs_max(a, -b) : -host_smooth_union(-a, b, s, q);
    return s == 0.0f ? rs_max(a, b) : -host_smooth_union(-a, -b, s, q);
}

void ivx_sdf_annotate_host(ivx_sdf_processed_node* nodes, size_t n) {
    std::vector<uint32_t> root_stack;  // roots of the subtrees on the evaluation stack
    for (size_t i = 0; i < n; ++i) {
        ivx_sdf_processed_node& nd = nodes[i];
        const uint32_t kind = nd.kind;
        float f;
        uint32_t first;
        if (kind <= 2u) {
            f = nd.margin;
            first = (uint32_t)i;
            nd.reserved[2] = (uint32_t)i;
            root_stack.push_back((uint32_t)i);
        } else if (kind >= 7u) {
            const uint32_t r2 = root_stack.back();
            root_stack.pop_back();
            const uint32_t r1 = root_stack.back();
            float f1, f2;
            std::memcpy(&f1, &nodes[r1].reserved[0], 4);
            std::memcpy(&f2, &nodes[r2].reserved[0], 4);
            const float r = host_combine(kind, f1, f2, nd.a, nd.b);
            f = (r >= nd.margin) ? f1 : r;  // the combination's own domain is outside too: applied only if the test fails
            first = nodes[r1].reserved[3];
            nd.reserved[2] = r1;
            root_stack.back() = (uint32_t)i;
        } else {  // translation / rotation (no-ops at evaluation time) and scaling: one operand
            const uint32_t r1 = root_stack.back();
            float f1;
            std::memcpy(&f1, &nodes[r1].reserved[0], 4);
            f = kind == 5u ? f1 * nd.a : f1;
            first = nodes[r1].reserved[3];
            nd.reserved[2] = r1;
            root_stack.back() = (uint32_t)i;
        }
        std::memcpy(&nd.reserved[0], &f, 4);
        nd.reserved[3] = first;
        nd.reserved[1] = (uint32_t)i;
        nodes[first].reserved[1] = (uint32_t)i;  // later (outer) roots overwrite earlier ones
    }
    // Saturation classes, root to leaves. The root's value is stored as (v * 50) as i8 (lib.rs:197-201): every value >= 2.54 is the byte +127,
    // every value <= -2.56 the byte -128. So at the root two values are INTERCHANGEABLE when they are equal, both >= U or both <= L (U = 2.54f
    // — the reference's own MAX_F32, what it fills a far leaf with —, L = -2.56f: 2.54f * 50 and -2.56f * 50 round to 127 and -128 exactly, and
    // the product is monotone). Every node inherits such a pair (U, L) for
    // its own result from its parent — the thresholds below are those for which "operands interchangeable => results interchangeable" holds
    // for the parent's formula (generation/sdf.rs:47-102; a smooth form moves its result by at most s / 4 and only where the operands are
    // within s of each other, hence the 1.25 s):
    //   scaling by a > 0:         U / a, L / a
    //   union r = su(x, y):       U + 1.25 s, L            (both operands)
    //   intersection:             U, L - 1.25 s            (both operands)
    //   subtraction r = x - y:    x: U, L - 1.25 s;  y: -(L - 1.25 s), -U
    // and with them an operand that lies in ONE class throughout a chunk can be dropped from the chunk's program when that leaves the other
    // operand's value (the pre-pass: k_sdf_prepass). What the pre-pass needs is one number per combination, kept in the field `c` (unused by
    // binary nodes) of this uploaded copy: union — an operand everywhere >= c goes; intersection — an operand everywhere <= c goes;
    // subtraction — a second operand everywhere >= c goes. A scaling that is not positive switches the rule off beneath it (infinite thresholds).
    {
        const float inf = std::numeric_limits<float>::infinity();
        std::vector<float> up(n, inf), lo(n, -inf);
        if (n) up[n - 1] = 0.02f * 127.0f, lo[n - 1] = -2.56f;
        // children of node i in postfix order: unary -> i - 1; binary -> second operand's root i - 1, first operand's root = reserved[2]
        for (size_t i = n; i-- > 0;) {
            ivx_sdf_processed_node& nd = nodes[i];
            const uint32_t kind = nd.kind;
            const float U = up[i], L = lo[i];
            if (kind <= 2u) continue;
            if (kind < 7u) {
                float cu = U, cl = L;
                if (kind == 5u) {
                    if (nd.a > 0.0f) cu = U / nd.a * 1.00001f + 1e-6f, cl = L / nd.a * 1.00001f - 1e-6f;  // (U > 0 > L)
                    else cu = inf, cl = -inf;
                }
                up[i - 1] = cu, lo[i - 1] = cl;
                continue;
            }
            const size_t r2 = i - 1, r1 = nd.reserved[2];
            const float w = 1.25f * nd.a * 1.00001f + (nd.a != 0.0f ? 1e-6f : 0.0f);
            if (kind == 7u) {
                up[r1] = up[r2] = U + w;
                lo[r1] = lo[r2] = L;
                nd.c = U + w;
            } else if (kind == 9u) {
                up[r1] = up[r2] = U;
                lo[r1] = lo[r2] = L - w;
                nd.c = L - w;
            } else {
                up[r1] = U, lo[r1] = L - w;
                up[r2] = -(L - w), lo[r2] = -U;
                nd.c = -(L - w);
            }
        }
    }
}

extern "C" int ivx_sdf_grid_shape(const float domain[6], uint32_t grid_shape[3], float shifted_grid_center[3]) {
    IVX_REQUIRE(domain && grid_shape && shifted_grid_center, IVX_ERR_INVALID, "ivx_sdf_grid_shape: null pointer");
    float ext[3];
    bool degenerate = false;
    for (int d = 0; d < 3; ++d) {
        ext[d] = domain[3 + d] - domain[d];
        if (ext[d] == 0.0f) degenerate = true;
    }
    if (degenerate) {  // generation.rs:217-225
        for (int d = 0; d < 3; ++d) {
            grid_shape[d] = 0;
            shifted_grid_center[d] = -0.5f;
        }
        return IVX_OK;
    }
    for (int d = 0; d < 3; ++d) {
        grid_shape[d] = (uint32_t)std::ceil(ext[d]) + 2u;
        const float centre_from_lower = 0.5f * (float)grid_shape[d];
        const float domain_centre = 0.5f * (domain[d] + domain[3 + d]);
        shifted_grid_center[d] = (centre_from_lower - domain_centre) - 0.5f;
    }
    return IVX_OK;
}
