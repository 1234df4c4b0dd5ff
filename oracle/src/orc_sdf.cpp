// ORACLE (test infrastructure only — never linked into the product library).
// See orc_sdf.hpp for the reference file:line map.
#include "orc_sdf.hpp"

#include <cmath>

namespace orc {

static const float FRAC_1_SQRT_3 = 0.57735026f;  // impact_math/src/consts/f32.rs:7

// generation/sdf.rs:89-92
float smooth_union(float d1, float d2, float s, float q) {
    float h = fmax_rs(s - std::fabs(d1 - d2), 0.0f);
    return fmin_rs(d1, d2) - (h * h) * q;
}
// generation/sdf.rs:47-102
static inline float sdf_union(float a, float b, float s, float q) { return s == 0.0f ? fmin_rs(a, b) : smooth_union(a, b, s, q); }
static inline float sdf_subtraction(float a, float b, float s, float q) {
    return s == 0.0f ? fmax_rs(a, -b) : -smooth_union(-a, b, s, q);
}
static inline float sdf_intersection(float a, float b, float s, float q) {
    return s == 0.0f ? fmax_rs(a, b) : -smooth_union(-a, -b, s, q);
}

// atomic.rs:1590-1598
static float soft_combine_domain_padding(float smoothness, uint32_t leaf_count) {
    float local = 0.25f * smoothness;
    return local * std::log2((float)leaf_count);
}

// glam Quat::mul_vec3a (SSE2 form): v*(w^2 - b.b) + b*(2 (v.b)) + (w*(b x v))*2 — parity unpinned
static V3 quat_rotate(Quat q, V3 v) {
    V3 b{q.x, q.y, q.z};
    float b2 = dot(b, b);
    V3 t0 = v * (q.w * q.w - b2);
    V3 t1 = b * (dot(v, b) * 2.0f);
    V3 t2 = (cross(b, v) * q.w) * 2.0f;
    return (t0 + t1) + t2;
}

static AABB domain_of_primitive(const SdfNode& n) {
    if (n.kind == SDF_SPHERE) return {v3s(-n.p[0]), v3s(n.p[0])};
    if (n.kind == SDF_CAPSULE) {
        V3 h = v3s(n.p[1]);
        h.y += 0.5f * n.p[0];
        return {-h, h};
    }
    V3 h = 0.5f * v3(n.p[0], n.p[1], n.p[2]);
    return {-h, h};
}

bool SdfGenerator::build(const SdfNode* in, int n, uint32_t root) {
    nodes.clear();
    stack_size = 0;
    domain = AABB{{0, 0, 0}, {0, 0, 0}};
    if (n == 0) return true;
    AABB zero{{0, 0, 0}, {0, 0, 0}};
    std::vector<AABB> domains(n, zero);
    std::vector<uint32_t> leaf(n, 0);
    std::vector<float> pad(n, 0.0f);
    std::vector<int> state(n, 0);  // 0 unvisited, 1 children being visited, 2 domain determined
    struct Op {
        int process;
        uint32_t id;
    };
    std::vector<Op> ops;
    ops.push_back({0, root});
    int top = 0, max_top = 0;
    std::vector<uint32_t> src;  // source node id of every processed (possibly duplicated) node
    while (!ops.empty()) {
        Op op = ops.back();
        ops.pop_back();
        if (op.id >= (uint32_t)n) return false;
        const SdfNode& node = in[op.id];
        if (!op.process) {
            if (state[op.id] == 1) return false;  // cycle
            if (state[op.id] == 0) state[op.id] = 1;
            ops.push_back({1, op.id});
            switch (node.kind) {
                case SDF_SPHERE: case SDF_CAPSULE: case SDF_BOX: break;
                case SDF_TRANSLATION: case SDF_ROTATION: case SDF_SCALING: ops.push_back({0, node.child1}); break;
                case SDF_UNION: case SDF_SUBTRACTION: case SDF_INTERSECTION:
                    ops.push_back({0, node.child2});
                    ops.push_back({0, node.child1});
                    break;
                default: return false;  // noise unsupported
            }
            continue;
        }
        uint32_t id = op.id;
        if (state[id] != 2) {
            state[id] = 2;
            switch (node.kind) {
                case SDF_SPHERE: case SDF_CAPSULE: case SDF_BOX:
                    domains[id] = domain_of_primitive(node);
                    leaf[id] = 1;
                    break;
                case SDF_TRANSLATION: {
                    V3 t = v3(node.p[0], node.p[1], node.p[2]);
                    domains[id] = {domains[node.child1].lo + t, domains[node.child1].hi + t};
                    leaf[id] = leaf[node.child1];
                    pad[id] = pad[node.child1];
                    break;
                }
                case SDF_ROTATION: {
                    // OrientedBox::from_axis_aligned_box(..).rotated(..).compute_corners() (oriented_box.rs:62-214)
                    Quat q{node.p[0], node.p[1], node.p[2], node.p[3]};
                    const AABB& cd = domains[node.child1];
                    V3 c = quat_rotate(q, center(cd));
                    V3 he = 0.5f * extents(cd);
                    V3 hw = he.x * quat_rotate(q, v3(1, 0, 0));
                    V3 hh = he.y * quat_rotate(q, v3(0, 1, 0));
                    V3 hd = he.z * quat_rotate(q, v3(0, 0, 1));
                    V3 corners[8] = {c - hw - hh - hd, c - hw - hh + hd, c - hw + hh - hd, c - hw + hh + hd,
                                     c + hw - hh - hd, c + hw - hh + hd, c + hw + hh - hd, c + hw + hh + hd};
                    AABB r{corners[0], corners[0]};
                    for (int i = 1; i < 8; ++i) {
                        r.lo = vmin(r.lo, corners[i]);
                        r.hi = vmax(r.hi, corners[i]);
                    }
                    domains[id] = r;
                    leaf[id] = leaf[node.child1];
                    pad[id] = pad[node.child1];
                    break;
                }
                case SDF_SCALING:
                    domains[id] = {node.p[0] * domains[node.child1].lo, node.p[0] * domains[node.child1].hi};
                    leaf[id] = leaf[node.child1];
                    pad[id] = pad[node.child1];
                    break;
                case SDF_UNION:
                    domains[id] = {vmin(domains[node.child1].lo, domains[node.child2].lo),
                                   vmax(domains[node.child1].hi, domains[node.child2].hi)};
                    leaf[id] = leaf[node.child1] + leaf[node.child2];
                    pad[id] = soft_combine_domain_padding(node.p[0], leaf[id]);
                    break;
                case SDF_SUBTRACTION:
                    domains[id] = domains[node.child1];
                    leaf[id] = leaf[node.child1] + leaf[node.child2];
                    pad[id] = soft_combine_domain_padding(node.p[0], leaf[id]);
                    break;
                case SDF_INTERSECTION: {
                    V3 lo = vmax(domains[node.child1].lo, domains[node.child2].lo);
                    V3 hi = vmin(domains[node.child1].hi, domains[node.child2].hi);
                    domains[id] = neg_mask(hi - lo) != 0 ? zero : AABB{lo, hi};
                    leaf[id] = leaf[node.child1] + leaf[node.child2];
                    pad[id] = soft_combine_domain_padding(node.p[0], leaf[id]);
                    break;
                }
                default: return false;
            }
        }
        ProcessedNode pn{};
        pn.kind = node.kind;
        pn.leaf_count = leaf[id];
        pn.transform = m4_identity();
        pn.domain_with_margin = expanded(domains[id], pad[id]);
        pn.margin = 0.0f;
        switch (node.kind) {
            case SDF_SPHERE: pn.a = node.p[0]; break;
            case SDF_CAPSULE: pn.a = 0.5f * node.p[0]; pn.b = node.p[1]; break;
            case SDF_BOX: pn.a = 0.5f * node.p[0]; pn.b = 0.5f * node.p[1]; pn.c = 0.5f * node.p[2]; break;
            case SDF_TRANSLATION: pn.a = node.p[0]; pn.b = node.p[1]; pn.c = node.p[2]; break;
            case SDF_SCALING: pn.a = node.p[0]; break;
            case SDF_UNION: case SDF_SUBTRACTION: case SDF_INTERSECTION:
                pn.a = node.p[0];
                pn.b = 0.25f / node.p[0];
                break;
            default: break;
        }
        nodes.push_back(pn);
        src.push_back(id);
        if (node.kind <= SDF_BOX) {
            top += 1;
            if (top > max_top) max_top = top;
        } else if (node.kind >= SDF_UNION) {
            top -= 1;
        }
    }
    // determine_transforms_and_margins (atomic.rs:495-596)
    {
        std::vector<M4> tstack(nodes.size(), M4{});
        std::vector<float> mstack(nodes.size(), 0.0f);
        int st = 0;
        tstack[0] = m4_identity();
        mstack[0] = SD_MAX_F32;
        for (int idx = (int)nodes.size() - 1; idx >= 0; --idx) {
            ProcessedNode& pn = nodes[idx];
            const SdfNode& sn = in[src[idx]];
            M4 transform = tstack[st];
            float margin = mstack[st];
            pn.transform = transform;
            pn.margin = margin;
            pn.domain_with_margin = expanded(pn.domain_with_margin, margin);
            switch (pn.kind) {
                case SDF_SPHERE: case SDF_CAPSULE: case SDF_BOX: st = st > 0 ? st - 1 : 0; break;
                case SDF_TRANSLATION:
                    tstack[st].c[3][0] += -sn.p[0];
                    tstack[st].c[3][1] += -sn.p[1];
                    tstack[st].c[3][2] += -sn.p[2];
                    tstack[st].c[3][3] += 0.0f;
                    break;
                case SDF_ROTATION: {
                    Quat q{sn.p[0], sn.p[1], sn.p[2], sn.p[3]};
                    tstack[st] = mul(m4_from_quat(conj(q)), transform);
                    break;
                }
                case SDF_SCALING: {
                    float r = 1.0f / sn.p[0];
                    for (int cidx = 0; cidx < 4; ++cidx)
                        for (int row = 0; row < 3; ++row) tstack[st].c[cidx][row] = r * tstack[st].c[cidx][row];
                    mstack[st] = margin / sn.p[0];
                    break;
                }
                default: {  // binary
                    tstack[st + 1] = transform;
                    float m = margin + 2.5f * soft_combine_domain_padding(sn.p[0], pn.leaf_count);
                    mstack[st] = m;
                    mstack[st + 1] = m;
                    st += 1;
                }
            }
        }
    }
    stack_size = max_top;
    domain = expanded(domains[root], pad[root]);
    return true;
}

// atomic.rs:1683-1797: the distinct flat indices of the 26 test positions (corners + face centres;
// the 12 "edge midpoint" entries reuse corner indices).
static const int TEST_IDX[14] = {
    0,
    15 * 256,
    15 * 16,
    15,
    15 * 256 + 15 * 16,
    15 * 256 + 15,
    15 * 16 + 15,
    15 * 256 + 15 * 16 + 15,
    8 * 16 + 8,
    15 * 256 + 8 * 16 + 8,
    8 * 256 + 8,
    8 * 256 + 15 * 16 + 8,
    8 * 256 + 8 * 16,
    8 * 256 + 8 * 16 + 15,
};

void SdfGenerator::compute_block(const AABB& block, std::vector<float>& stack, float* out) const {
    const int N = CHUNK_VOXELS;
    if (nodes.empty()) {
        for (int i = 0; i < N; ++i) out[i] = SD_MAX_F32;
        return;
    }
    stack.resize((size_t)(stack_size + 1) * N);
    V3 block_origin = block.lo;
    int top = 0;
    for (const ProcessedNode& node : nodes) {
        if (node.kind <= SDF_BOX) {
            float* d = &stack[(size_t)top * N];
            AABB bn = aabb_of_transformed(block, node.transform);
            AABB interior;
            if (node.kind == SDF_SPHERE) {
                V3 h = v3s(node.a * FRAC_1_SQRT_3 + (-node.margin));
                interior = {-h, h};
            } else if (node.kind == SDF_CAPSULE) {
                V3 h = v3s(node.b * FRAC_1_SQRT_3 + (-node.margin));
                h.y += node.a;
                interior = {-h, h};
            } else {
                V3 h = v3(node.a, node.b, node.c) + v3s(-node.margin);
                interior = {-h, h};
            }
            if (box_lies_outside(node.domain_with_margin, bn)) {
                for (int i = 0; i < N; ++i) d[i] = node.margin;
            } else if (contains_box(interior, bn)) {
                for (int i = 0; i < N; ++i) d[i] = -node.margin;
            } else {
                // update_signed_distances_for_block[_packed] (atomic.rs:1601-1658)
                V3 origin = transform_point(node.transform, block_origin);
                V3 dx = col3(node.transform, 0), dy = col3(node.transform, 1), dz = col3(node.transform, 2);
                int idx = 0;
                for (int i = 0; i < CHUNK; ++i) {
                    V3 opx = origin + (float)i * dx;
                    for (int j = 0; j < CHUNK; ++j) {
                        V3 pos = opx + (float)j * dy;
                        for (int k = 0; k < CHUNK; ++k) {
                            float v;
                            if (node.kind == SDF_SPHERE) {
                                v = length(pos) - node.a;
                            } else if (node.kind == SDF_CAPSULE) {
                                V3 p = pos;
                                float c = p.y;
                                if (c < -node.a) c = -node.a;
                                if (c > node.a) c = node.a;
                                p.y -= c;
                                v = length(p) - node.b;
                            } else {
                                V3 q = vabs(pos) - v3(node.a, node.b, node.c);
                                v = length(vmax(q, v3s(0.0f))) + fmin_rs(max_component(q), 0.0f);
                            }
                            d[idx++] = v;
                            pos = pos + dz;
                        }
                    }
                }
            }
            top += 1;
        } else if (node.kind == SDF_TRANSLATION || node.kind == SDF_ROTATION) {
        } else if (node.kind == SDF_SCALING) {
            float* d = &stack[(size_t)(top - 1) * N];
            for (int i = 0; i < N; ++i) d[i] *= node.a;
        } else {
            top -= 1;
            AABB bn = aabb_of_transformed(block, node.transform);
            float* d1 = &stack[(size_t)(top - 1) * N];
            const float* d2 = &stack[(size_t)top * N];
            float s = node.a, q = node.b;
            auto op = [&](float a, float b) {
                return node.kind == SDF_UNION ? sdf_union(a, b, s, q)
                       : node.kind == SDF_SUBTRACTION ? sdf_subtraction(a, b, s, q)
                                                      : sdf_intersection(a, b, s, q);
            };
            bool apply = !box_lies_outside(node.domain_with_margin, bn);
            if (!apply) {
                bool all_pass = true;
                for (int t = 0; t < 14 && all_pass; ++t)
                    if (!(op(d1[TEST_IDX[t]], d2[TEST_IDX[t]]) >= node.margin)) all_pass = false;
                apply = !all_pass;
            }
            if (apply)
                for (int i = 0; i < N; ++i) d1[i] = op(d1[i], d2[i]);
        }
    }
    for (int i = 0; i < N; ++i) out[i] = stack[i];
}

// generation.rs:207-258
void SdfVoxelGenerator::init(float voxel_extent, uint8_t type) {
    extent = voxel_extent;
    voxel_type = type;
    V3 e = extents(sdf.domain);
    if (e.x == 0.0f || e.y == 0.0f || e.z == 0.0f) {
        shape[0] = shape[1] = shape[2] = 0;
        shifted_center = v3s(-0.5f);
        return;
    }
    float ee[3] = {e.x, e.y, e.z};
    for (int d = 0; d < 3; ++d) shape[d] = (int)std::ceil(ee[d]) + 2;
    V3 c_rel_lower = v3(0.5f * (float)shape[0], 0.5f * (float)shape[1], 0.5f * (float)shape[2]);
    V3 c_rel_origin = c_rel_lower - center(sdf.domain);
    shifted_center = c_rel_origin - v3s(0.5f);
}

// generation.rs:293-371
ChunkSparseness SdfVoxelGenerator::generate_chunk(Voxel* voxels, const int origin[3]) const {
    if (sdf.nodes.empty() || origin[0] >= shape[0] || origin[1] >= shape[1] || origin[2] >= shape[2]) {
        for (int i = 0; i < CHUNK_VOXELS; ++i) voxels[i] = voxel_max_outside();
        return {true, true};
    }
    V3 o = v3((float)origin[0], (float)origin[1], (float)origin[2]) - shifted_center;
    AABB aabb{o, o + v3s((float)CHUNK)};
    static thread_local std::vector<float> stack;
    static thread_local std::vector<float> dist(CHUNK_VOXELS);
    sdf.compute_block(aabb, stack, dist.data());
    bool only_empty = true, is_void = true;
    int idx = 0;
    for (int ic = 0; ic < CHUNK; ++ic)
        for (int jc = 0; jc < CHUNK; ++jc)
            for (int kc = 0; kc < CHUNK; ++kc, ++idx) {
                int i = origin[0] + ic, j = origin[1] + jc, k = origin[2] + kc;
                if (i >= shape[0] || j >= shape[1] || k >= shape[2]) {
                    voxels[idx] = voxel_max_outside();
                } else {
                    int8_t sd = sd_from_f32(dist[idx]);
                    if (sd < 0) {
                        only_empty = false;
                        is_void = false;
                        voxels[idx] = Voxel{TYPE_DUMMY, sd, 0};
                    } else {
                        if (!sd_is_void(sd)) is_void = false;
                        voxels[idx] = Voxel{TYPE_DUMMY, sd, F_EMPTY};
                    }
                }
            }
    if (!only_empty)
        for (int t = 0; t < CHUNK_VOXELS; ++t) voxels[t].type = voxel_type;  // voxel_type.rs:88-96
    return {only_empty, is_void};
}

}  // namespace orc
