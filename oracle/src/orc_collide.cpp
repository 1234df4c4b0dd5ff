// ORACLE (test infrastructure only — never linked into the product library).
//
// CPU restatement of the contact generation between two voxel objects (paths relative to
// /root/reference/engine/crates):
//   VoxelObjectCollisionProbes::recompute_for_all_chunks      impact_voxel/src/collidable.rs:361-392, 451-523
//   add_points_for_vertices_in_blocks                          impact_voxel/src/collidable.rs:614-731 (block index helpers 733-789)
//   for_each_mutual_voxel_object_contact                       impact_voxel/src/collidable.rs:859-1049
//   determine_sdf_value_and_normal_at_point_if_intersecting    impact_voxel/src/collidable.rs:1288-1440
//   evaluate_sdf_from_corner_samples / compute_sdf_gradient_from_corner_samples   impact_voxel/src/object/sdf.rs:579-633
//   VoxelObject::determine_voxel_ranges_encompassing_intersection   impact_voxel/src/object/intersection.rs:706-746
//   compute_box_intersection_bounds, OrientedBox               impact_geometry/src/oriented_box.rs:50-60, 130-190, 315-431
//   AxisAlignedBox::{find_contained_subsegment, corner, contains_point, expanded_about_center}
//                                                              impact_geometry/src/axis_aligned_box.rs:152-203, 332-335, 385-415
//   Isometry3 product / inverted                               impact_math/src/transform/isometry.rs:128-134, 200-205
//
// ORDER: the reference walks the chunks holding probes in the iteration order of a hashbrown 0.16 map keyed by the chunk indices
// with rustc-hash 2.1 (impact_containers, engine/Cargo.lock) — an order that is a property of those crates' internals and that no
// reference test pins. Here (and in the HIP path) chunks are walked in submesh order = chunk-linear order, the order in which
// recompute_for_all_chunks inserts them. The SET of contacts and every contact's id and geometry follow the reference; their order
// within the manifold is "parity unpinned".
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <unordered_map>
#include <vector>

#include "../include/oracle.h"
#include "orc_math.hpp"
#include "orc_voxel.hpp"

namespace orc {

static inline V3 qrot(Quat q, V3 v) {  // glam Quat::mul_vec3a
    V3 b{q.x, q.y, q.z};
    float b2 = dot(b, b);
    return (v * (q.w * q.w - b2) + b * (dot(v, b) * 2.0f)) + cross(b, v) * (q.w * 2.0f);
}
static inline long as_usize(float f) { return f > 0.0f ? (f < 9.0e18f ? (long)f : std::numeric_limits<long>::max()) : 0; }  // `as usize` (NaN -> 0)

// ---- probes ----------------------------------------------------------------------------------------------------------------------
// entries: 5 u32 per chunk with probes (ci, cj, ck, start, end); returns the number of points (all of them, also beyond cap)
int collision_probes(const VoxelObject& obj, const float* pos, const float* nrm, const uint32_t* idx, const uint32_t* submeshes, uint32_t n_sub,
                     float* points, uint32_t cap, uint32_t* entries, uint32_t* n_entries) {
    long min_extent = std::numeric_limits<long>::max();
    for (int d = 0; d < 3; ++d) min_extent = std::min<long>(min_extent, std::max(0, obj.occ_voxel[d][1] - obj.occ_voxel[d][0]));
    const int log2_bs = min_extent >= 16 ? 3 : (min_extent >= 8 ? 2 : (min_extent >= 4 ? 1 : 0));
    const int log2_cb = 4 - log2_bs;  // log2 of the chunk size in blocks
    const int n_blocks = 1 << (3 * log2_cb);
    const float inv = 1.0f / obj.extent;
    uint32_t n_pts = 0;
    *n_entries = 0;
    std::vector<float> curv;
    struct Best {
        float x, y, z, w;
    };
    std::vector<Best> best((size_t)n_blocks);
    for (uint32_t s = 0; s < n_sub; ++s) {
        const uint32_t* sm = submeshes + 16 * (size_t)s;
        const uint32_t ioff = sm[3], icnt = sm[4], voff = sm[13], vcnt = sm[14];
        curv.assign(2 * (size_t)vcnt, 0.0f);
        const float* P = pos + 3 * (size_t)voff;
        const float* N = nrm + 3 * (size_t)voff;
        for (uint32_t t = 0; t + 2 < icnt; t += 3) {
            const uint32_t i0 = idx[ioff + t] - voff, i1 = idx[ioff + t + 1] - voff, i2 = idx[ioff + t + 2] - voff;
            const V3 v0{P[3 * i0], P[3 * i0 + 1], P[3 * i0 + 2]}, v1{P[3 * i1], P[3 * i1 + 1], P[3 * i1 + 2]}, v2{P[3 * i2], P[3 * i2 + 1], P[3 * i2 + 2]};
            const V3 n0{N[3 * i0], N[3 * i0 + 1], N[3 * i0 + 2]}, n1{N[3 * i1], N[3 * i1 + 1], N[3 * i1 + 2]}, n2{N[3 * i2], N[3 * i2 + 1], N[3 * i2 + 2]};
            const V3 e01 = v1 - v0, e12 = v2 - v1, e20 = v0 - v2;
            curv[2 * i0] += dot(n0, e01) - dot(n0, e20);
            curv[2 * i0 + 1] += 2.0f;
            curv[2 * i1] += dot(n1, e12) - dot(n1, e01);
            curv[2 * i1 + 1] += 2.0f;
            curv[2 * i2] += dot(n2, e20) - dot(n2, e12);
            curv[2 * i2 + 1] += 2.0f;
        }
        const float clo[3] = {(float)(sm[0] * 16u), (float)(sm[1] * 16u), (float)(sm[2] * 16u)};
        const float chi[3] = {(float)((sm[0] + 1u) * 16u), (float)((sm[1] + 1u) * 16u), (float)((sm[2] + 1u) * 16u)};
        const float inf = std::numeric_limits<float>::infinity();
        for (auto& b : best) b = Best{inf, inf, inf, inf};
        for (uint32_t v = 0; v < vcnt; ++v) {
            if (curv[2 * v + 1] == 0.0f) continue;
            const float p[3] = {P[3 * v], P[3 * v + 1], P[3 * v + 2]};
            long vi[3];
            for (int d = 0; d < 3; ++d) {
                const float np_ = p[d] * inv;
                const float cl = fmin_rs(fmax_rs(np_, clo[d]), chi[d]);
                vi[d] = as_usize(cl);
            }
            const long bi = (vi[0] & 15) >> log2_bs, bj = (vi[1] & 15) >> log2_bs, bk = (vi[2] & 15) >> log2_bs;
            const size_t block = (size_t)((bi << (2 * log2_cb)) + (bj << log2_cb) + bk);
            const float c = curv[2 * v] / curv[2 * v + 1];
            if (c < best[block].w) best[block] = Best{p[0], p[1], p[2], c};
        }
        const uint32_t start = n_pts;
        for (const Best& b : best)
            if (b.w != inf) {
                if (n_pts < cap) points[3 * (size_t)n_pts] = b.x, points[3 * (size_t)n_pts + 1] = b.y, points[3 * (size_t)n_pts + 2] = b.z;
                n_pts += 1;
            }
        if (n_pts == start) continue;
        uint32_t* e = entries + 5 * (size_t)(*n_entries);
        e[0] = sm[0], e[1] = sm[1], e[2] = sm[2], e[3] = start, e[4] = n_pts;
        *n_entries += 1;
    }
    return (int)n_pts;
}

// ---- probes kept in step with an incrementally re-meshed object ---------------------------------------------------------------------
//   VoxelObjectCollisionProbes::sync_with_voxel_object_and_mesh / update_for_chunk     impact_voxel/src/collidable.rs:394-433, 524-612
// The point buffer only grows; a chunk's points live in a range handed out by a RangeAllocator (best fit over the ranges freed before,
// else appended). The reference walks the invalidated chunks in hash-set order (unpinned); here chunk-linear order.
}  // namespace orc
#include "orc_mesh.hpp"
namespace orc {
struct ProbeSet {
    std::vector<float> points;                                         // 3 per point, holes included
    std::unordered_map<uint64_t, std::pair<uint32_t, uint32_t>> ranges;  // chunk -> [start, end)
    RangeAllocator allocator;
};
static inline uint64_t probe_key(uint32_t i, uint32_t j, uint32_t k) { return ((uint64_t)i << 42) | ((uint64_t)j << 21) | (uint64_t)k; }
static void mesh_arrays(const Mesh& m, std::vector<uint32_t>& sub16) {
    sub16.assign(16 * m.submeshes.size(), 0u);
    for (size_t s = 0; s < m.submeshes.size(); ++s) {
        const Submesh& sm = m.submeshes[s];
        uint32_t* q = &sub16[16 * s];
        q[0] = sm.chunk[0], q[1] = sm.chunk[1], q[2] = sm.chunk[2], q[3] = sm.index_offset, q[4] = sm.index_count, q[13] = sm.vertex_offset, q[14] = sm.vertex_count;
    }
}
void probes_recompute(const VoxelObject& obj, const Mesh& mesh, ProbeSet& ps) {
    ps = ProbeSet{};
    std::vector<uint32_t> sub16;
    mesh_arrays(mesh, sub16);
    const uint32_t ns = (uint32_t)mesh.submeshes.size();
    std::vector<uint32_t> entries(5 * (size_t)std::max<uint32_t>(ns, 1));
    uint32_t ne = 0;
    const float* pos = mesh.positions.empty() ? nullptr : &mesh.positions[0].x;
    const float* nrm = mesh.normals.empty() ? nullptr : &mesh.normals[0].x;
    int n = collision_probes(obj, pos, nrm, mesh.indices.data(), sub16.data(), ns, nullptr, 0, entries.data(), &ne);
    ps.points.assign(3 * (size_t)n, 0.0f);
    collision_probes(obj, pos, nrm, mesh.indices.data(), sub16.data(), ns, ps.points.data(), (uint32_t)n, entries.data(), &ne);
    for (uint32_t e = 0; e < ne; ++e) ps.ranges[probe_key(entries[5 * e], entries[5 * e + 1], entries[5 * e + 2])] = {entries[5 * e + 3], entries[5 * e + 4]};
}
void probes_sync(const VoxelObject& obj, const Mesh& mesh, ProbeSet& ps, const uint8_t* invalidated) {
    std::vector<float> buf;
    std::vector<uint32_t> entry(5);
    for (int ci = 0; ci < obj.cc[0]; ++ci)
        for (int cj = 0; cj < obj.cc[1]; ++cj)
            for (int ck = 0; ck < obj.cc[2]; ++ck) {
                if (!invalidated[obj.cidx(ci, cj, ck)]) continue;
                const uint64_t key = probe_key((uint32_t)ci, (uint32_t)cj, (uint32_t)ck);
                auto old = ps.ranges.find(key);
                auto sm = mesh.chunk_index.find(key);  // (the two keys are built the same way)
                uint32_t n = 0;
                if (sm != mesh.chunk_index.end()) {
                    const Submesh& m = mesh.submeshes[sm->second];
                    uint32_t one[16] = {m.chunk[0], m.chunk[1], m.chunk[2], m.index_offset, m.index_count, 0, 0, 0, 0, 0, 0, 0, 0, m.vertex_offset, m.vertex_count, 0};
                    buf.assign(3 * 4096, 0.0f);
                    uint32_t ne = 0;
                    n = (uint32_t)collision_probes(obj, &mesh.positions[0].x, &mesh.normals[0].x, mesh.indices.data(), one, 1, buf.data(), 4096, entry.data(), &ne);
                }
                if (n == 0) {  // no mesh, or no points: the chunk leaves the set and its range is freed
                    if (old != ps.ranges.end()) {
                        ps.allocator.free_range(old->second.first, old->second.second);
                        ps.ranges.erase(old);
                    }
                    continue;
                }
                if (old != ps.ranges.end()) ps.allocator.free_range(old->second.first, old->second.second);
                size_t start;
                if (!ps.allocator.allocate_range(n, start)) {
                    start = ps.points.size() / 3;
                    ps.points.resize(ps.points.size() + 3 * (size_t)n);
                }
                ps.ranges[key] = {(uint32_t)start, (uint32_t)(start + n)};
                std::copy(buf.begin(), buf.begin() + 3 * (size_t)n, ps.points.begin() + 3 * (std::ptrdiff_t)start);
            }
    ps.allocator.merge_consecutive_ranges();
}
ProbeSet* probes_new() { return new ProbeSet(); }
void probes_delete(ProbeSet* p) { delete p; }
uint32_t probes_points(const ProbeSet& ps, float* out, uint32_t cap) {
    const uint32_t n = (uint32_t)(ps.points.size() / 3);
    if (out) std::memcpy(out, ps.points.data(), 12 * (size_t)std::min(n, cap));
    return n;
}
// entries sorted by range start: 5 u32 each (ci, cj, ck, start, end); returns their number
uint32_t probes_entries(const ProbeSet& ps, uint32_t* out) {
    std::vector<std::pair<uint32_t, uint64_t>> order;
    for (const auto& kv : ps.ranges) order.push_back({kv.second.first, kv.first});
    std::sort(order.begin(), order.end());
    uint32_t n = 0;
    for (const auto& o : order) {
        const auto& r = ps.ranges.at(o.second);
        uint32_t* e = out + 5 * (size_t)n++;
        e[0] = (uint32_t)(o.second >> 42), e[1] = (uint32_t)((o.second >> 21) & 0x1FFFFFu), e[2] = (uint32_t)(o.second & 0x1FFFFFu), e[3] = r.first, e[4] = r.second;
    }
    return n;
}

// ---- boxes -----------------------------------------------------------------------------------------------------------------------
struct Aab {
    V3 lo, hi;
};
struct Obb {
    V3 center;
    Quat q;
    V3 half;
};
static bool contained_subsegment(const Aab& b, V3 start, V3 off, float& t_min, float& t_max) {
    t_min = 0.0f;
    t_max = 1.0f;
    const float s[3] = {start.x, start.y, start.z}, o[3] = {off.x, off.y, off.z}, lo[3] = {b.lo.x, b.lo.y, b.lo.z}, hi[3] = {b.hi.x, b.hi.y, b.hi.z};
    for (int d = 0; d < 3; ++d) {
        if (std::fabs(o[d]) > 1e-8f) {
            const float recip = 1.0f / o[d];
            const float t1 = (lo[d] - s[d]) * recip, t2 = (hi[d] - s[d]) * recip;
            const float te = t1 < t2 ? t1 : t2, tx = t1 < t2 ? t2 : t1;
            t_min = fmax_rs(t_min, te);
            t_max = fmin_rs(t_max, tx);
        } else if (s[d] < lo[d] || s[d] > hi[d]) {
            return false;
        }
    }
    return t_min <= t_max;
}
static V3 to_box_frame(const Obb& b, V3 p) { return qrot(conj(b.q), p - b.center); }
static V3 from_box_frame(const Obb& b, V3 p) { return b.center + qrot(b.q, p); }

static bool box_intersection_bounds(const Aab& a, const Obb& b, Aab& in_a, Aab& in_b) {
    static const int EDGES[12][2] = {{0, 1}, {2, 3}, {4, 5}, {6, 7}, {0, 2}, {1, 3}, {4, 6}, {5, 7}, {0, 4}, {1, 5}, {2, 6}, {3, 7}};
    const float inf = std::numeric_limits<float>::infinity();
    in_a = Aab{{inf, inf, inf}, {-inf, -inf, -inf}};
    in_b = in_a;
    bool hit = false;
    auto expand = [&](V3 pa, V3 pb) {
        in_a.lo = vmin(in_a.lo, pa);
        in_a.hi = vmax(in_a.hi, pa);
        in_b.lo = vmin(in_b.lo, pb);
        in_b.hi = vmax(in_b.hi, pb);
        hit = true;
    };
    const M3 r = m3_from_quat(b.q);
    const V3 hw = b.half.x * r.c0, hh = b.half.y * r.c1, hd = b.half.z * r.c2;
    const V3 bc[8] = {((b.center - hw) - hh) - hd, ((b.center - hw) - hh) + hd, ((b.center - hw) + hh) - hd, ((b.center - hw) + hh) + hd,
                      ((b.center + hw) - hh) - hd, ((b.center + hw) - hh) + hd, ((b.center + hw) + hh) - hd, ((b.center + hw) + hh) + hd};
    for (const auto& e : EDGES) {
        const V3 s = bc[e[0]], v = bc[e[1]] - s;
        float t0, t1;
        if (contained_subsegment(a, s, v, t0, t1)) {
            const V3 p0 = s + v * t0, p1 = s + v * t1;
            expand(p0, to_box_frame(b, p0));
            expand(p1, to_box_frame(b, p1));
        }
    }
    V3 ac[8];
    for (int c = 0; c < 8; ++c) ac[c] = to_box_frame(b, V3{(c & 4) ? a.hi.x : a.lo.x, (c & 2) ? a.hi.y : a.lo.y, (c & 1) ? a.hi.z : a.lo.z});
    const Aab bb{-b.half, b.half};
    for (const auto& e : EDGES) {
        const V3 s = ac[e[0]], v = ac[e[1]] - s;
        float t0, t1;
        if (contained_subsegment(bb, s, v, t0, t1)) {
            const V3 p0 = s + v * t0, p1 = s + v * t1;
            expand(from_box_frame(b, p0), p0);
            expand(from_box_frame(b, p1), p1);
        }
    }
    return hit;
}

static void ranges_touching(const VoxelObject& o, const Aab& norm, long lo[3], long hi[3]) {  // voxel_ranges_touching_aab on the occupied ranges
    const float l[3] = {norm.lo.x, norm.lo.y, norm.lo.z}, h[3] = {norm.hi.x, norm.hi.y, norm.hi.z};
    for (int d = 0; d < 3; ++d) {
        const float fl = std::floor(l[d]);
        lo[d] = std::max<long>(o.occ_voxel[d][0], as_usize(fl > 0.0f ? fl : 0.0f));
        hi[d] = std::min<long>(o.occ_voxel[d][1], as_usize(std::ceil(h[d])));
    }
}
static Aab occupied_aabb(const VoxelObject& o) {
    const V3 lo{(float)o.occ_voxel[0][0], (float)o.occ_voxel[1][0], (float)o.occ_voxel[2][0]};
    const V3 hi{(float)o.occ_voxel[0][1], (float)o.occ_voxel[1][1], (float)o.occ_voxel[2][1]};
    return {o.extent * lo, o.extent * hi};
}

// ---- SDF probe -------------------------------------------------------------------------------------------------------------------
static bool deep_inside(V3 center, V3 p, float& sd, V3& n) {
    sd = sd_to_f32(-128);
    const V3 d = p - center;
    const float n2 = dot(d, d);
    if (!(n2 > 1e-8f * 1e-8f)) return false;
    n = div_elem(d, std::sqrt(n2));
    return true;
}
static bool sdf_and_normal_if_intersecting(const VoxelObject& o, V3 center, V3 p, float& sd, V3& n) {
    const float HALF_DIAG = 0.5f * 1.7320508075688772f;
    const V3 lp = p - V3{0.5f, 0.5f, 0.5f};
    if (std::signbit(lp.x) || std::signbit(lp.y) || std::signbit(lp.z)) return false;
    const long li = as_usize(lp.x), lj = as_usize(lp.y), lk = as_usize(lp.z);
    if ((li + 1 >= (long)o.cc[0] * 16) | (lj + 1 >= (long)o.cc[1] * 16) | (lk + 1 >= (long)o.cc[2] * 16)) return false;
    const long ci = as_usize(p.x), cj = as_usize(p.y), ck = as_usize(p.z);
    const Chunk& ch = o.chunks[o.cidx((int)(ci >> 4), (int)(cj >> 4), (int)(ck >> 4))];
    if (ch.kind == K_UNIFORM) return deep_inside(center, p, sd, n);
    if (ch.kind == K_VOID) return false;
    const float containing = sd_to_f32(o.voxels[((size_t)ch.data_offset << 12) + (((ci & 15) << 8) | ((cj & 15) << 4) | (ck & 15))].sd);
    if (containing > HALF_DIAG) return false;
    float d[8];
    for (int c = 0; c < 8; ++c) d[c] = sd_to_f32(o.voxel_at((int)(li + ((c >> 2) & 1)), (int)(lj + ((c >> 1) & 1)), (int)(lk + (c & 1))).sd);
    const V3 off{lp.x - std::floor(lp.x), lp.y - std::floor(lp.y), lp.z - std::floor(lp.z)};
    const V3 rev = V3{1.0f, 1.0f, 1.0f} - off;
    {  // evaluate_sdf_from_corner_samples
        const float d00 = d[0] * rev.x + d[4] * off.x, d01 = d[1] * rev.x + d[5] * off.x, d10 = d[2] * rev.x + d[6] * off.x, d11 = d[3] * rev.x + d[7] * off.x;
        const float d0 = d00 * rev.y + d10 * off.y, d1 = d01 * rev.y + d11 * off.y;
        sd = d0 * rev.z + d1 * off.z;
    }
    if (sd > 0.0f) return false;
    if (std::fabs(sd - sd_to_f32(-128)) < 1e-3f) return deep_inside(center, p, sd, n);
    // compute_sdf_gradient_from_corner_samples
    const V3 p00{d[4], d[2], d[1]}, n00{d[0], d[0], d[0]}, p01{d[5], d[6], d[3]}, n01{d[1], d[4], d[2]}, p10{d[6], d[3], d[5]}, n10{d[2], d[1], d[4]},
        p11{d[7], d[7], d[7]}, n11{d[3], d[5], d[6]};
    const V3 e00 = p00 - n00, e01 = p01 - n01, e10 = p10 - n10, e11 = p11 - n11;
    auto yzx = [](V3 v) { return V3{v.y, v.z, v.x}; };
    auto zxy = [](V3 v) { return V3{v.z, v.x, v.y}; };
    const V3 g = ((cmul(cmul(yzx(rev), zxy(rev)), e00) + cmul(cmul(yzx(rev), zxy(off)), e01)) + cmul(cmul(yzx(off), zxy(rev)), e10)) +
                 cmul(cmul(yzx(off), zxy(off)), e11);
    const float g2 = dot(g, g);
    if (!(g2 > 1e-8f * 1e-8f)) return false;
    n = div_elem(g, std::sqrt(g2));
    return true;
}

// ---- mutual contacts -------------------------------------------------------------------------------------------------------------
struct Probes {
    const float* points;
    const uint32_t* entries;
    uint32_t n_entries;
};
static int mutual_contacts_impl(const VoxelObject& A, const Probes& pa, const float com_a[3], const float rot_a[4], const float trans_a[3], const VoxelObject& B,
                                const Probes& pb, const float com_b[3], const float rot_b[4], const float trans_b[3], int cap, int32_t* which_ijk,
                                float* position, float* normal, float* depth);
int mutual_contacts(const VoxelObject& A, const float* probes_a, const uint32_t* entries_a, uint32_t n_entries_a, const float com_a[3], const float rot_a[4],
                    const float trans_a[3], const VoxelObject& B, const float* probes_b, const uint32_t* entries_b, uint32_t n_entries_b,
                    const float com_b[3], const float rot_b[4], const float trans_b[3], int cap, int32_t* which_ijk, float* position, float* normal,
                    float* depth) {
    return mutual_contacts_impl(A, Probes{probes_a, entries_a, n_entries_a}, com_a, rot_a, trans_a, B, Probes{probes_b, entries_b, n_entries_b}, com_b, rot_b,
                                trans_b, cap, which_ijk, position, normal, depth);
}

// which_ijk [n][4]: 0 = probe of A against B / 1 = probe of B against A, then the probing object's voxel indices (the id hashes
// [0, i, j, k] either way); returns the number of contacts (all of them, also beyond cap)
// VoxelObject::determine_voxel_ranges_encompassing_intersection (object/intersection.rs:706-746) for world -> object transforms of A and B;
// also hands back transform_from_b_to_a. Ranges are NOT checked for emptiness (the reference does not either).
bool intersection_voxel_ranges(const VoxelObject& A, const float rot_a[4], const float trans_a[3], const VoxelObject& B, const float rot_b[4],
                               const float trans_b[3], long ra_lo[3], long ra_hi[3], long rb_lo[3], long rb_hi[3], float q_ba_out[4], float t_ba_out[3]) {
    const Quat qa{rot_a[0], rot_a[1], rot_a[2], rot_a[3]}, qb{rot_b[0], rot_b[1], rot_b[2], rot_b[3]};
    const V3 ta{trans_a[0], trans_a[1], trans_a[2]}, tb{trans_b[0], trans_b[1], trans_b[2]};
    // transform_from_b_to_a = world_to_a * world_to_b.inverted()
    const Quat qbi = conj(qb);
    const V3 tbi = -qrot(qbi, tb);
    const Quat q_ba = qmul(qa, qbi);
    const V3 t_ba = qrot(qa, tbi) + ta;
    q_ba_out[0] = q_ba.x, q_ba_out[1] = q_ba.y, q_ba_out[2] = q_ba.z, q_ba_out[3] = q_ba.w;
    t_ba_out[0] = t_ba.x, t_ba_out[1] = t_ba.y, t_ba_out[2] = t_ba.z;
    const Aab aabb_a = occupied_aabb(A), aabb_b = occupied_aabb(B);
    const V3 b_center = 0.5f * (aabb_b.lo + aabb_b.hi);
    const Obb b_in_a{qrot(q_ba, b_center) + t_ba, qmul(q_ba, Quat{0.0f, 0.0f, 0.0f, 1.0f}), 0.5f * (aabb_b.hi - aabb_b.lo)};
    Aab in_a, in_b_rel;
    if (!box_intersection_bounds(aabb_a, b_in_a, in_a, in_b_rel)) return false;
    const Aab in_b{in_b_rel.lo + b_center, in_b_rel.hi + b_center};
    const float inv_a = 1.0f / A.extent, inv_b = 1.0f / B.extent;
    ranges_touching(A, Aab{inv_a * in_a.lo, inv_a * in_a.hi}, ra_lo, ra_hi);
    ranges_touching(B, Aab{inv_b * in_b.lo, inv_b * in_b.hi}, rb_lo, rb_hi);
    return true;
}

// sample_voxel_object_sdf (object/sdf.rs:636-675): trilinear SDF of the object at a normalized position, +2.54 outside the grid
float sample_voxel_object_sdf(const VoxelObject& o, V3 p) {
    const V3 lc = p - V3{0.5f, 0.5f, 0.5f};
    const V3 fl{std::floor(lc.x), std::floor(lc.y), std::floor(lc.z)};
    const V3 off = lc - fl;
    if (std::signbit(fl.x) || std::signbit(fl.y) || std::signbit(fl.z)) return sd_to_f32(127);
    const long i = as_usize(fl.x), j = as_usize(fl.y), k = as_usize(fl.z);
    if (i + 1 >= (long)o.cc[0] * 16 || j + 1 >= (long)o.cc[1] * 16 || k + 1 >= (long)o.cc[2] * 16) return sd_to_f32(127);
    float d[8];
    for (int c = 0; c < 8; ++c) d[c] = sd_to_f32(o.voxel_at((int)(i + ((c >> 2) & 1)), (int)(j + ((c >> 1) & 1)), (int)(k + (c & 1))).sd);
    const V3 rev = V3{1.0f, 1.0f, 1.0f} - off;
    const float d00 = d[0] * rev.x + d[4] * off.x, d01 = d[1] * rev.x + d[5] * off.x, d10 = d[2] * rev.x + d[6] * off.x, d11 = d[3] * rev.x + d[7] * off.x;
    const float d0 = d00 * rev.y + d10 * off.y, d1 = d01 * rev.y + d11 * off.y;
    return d0 * rev.z + d1 * off.z;
}

static int mutual_contacts_impl(const VoxelObject& A, const Probes& pa, const float com_a[3], const float rot_a[4], const float trans_a[3], const VoxelObject& B,
                                const Probes& pb, const float com_b[3], const float rot_b[4], const float trans_b[3], int cap, int32_t* which_ijk,
                                float* position, float* normal, float* depth) {
    const Quat qa{rot_a[0], rot_a[1], rot_a[2], rot_a[3]}, qb{rot_b[0], rot_b[1], rot_b[2], rot_b[3]};
    const V3 ta{trans_a[0], trans_a[1], trans_a[2]}, tb{trans_b[0], trans_b[1], trans_b[2]};
    long ra_lo[3], ra_hi[3], rb_lo[3], rb_hi[3];
    float q_ba[4], t_ba[3];
    if (!intersection_voxel_ranges(A, rot_a, trans_a, B, rot_b, trans_b, ra_lo, ra_hi, rb_lo, rb_hi, q_ba, t_ba)) return 0;
    const float inv_a = 1.0f / A.extent, inv_b = 1.0f / B.extent;
    int n = 0;
    for (int pass = 0; pass < 2; ++pass) {
        const VoxelObject& P = pass == 0 ? A : B;  // the probing object
        const VoxelObject& S = pass == 0 ? B : A;  // the sampled one
        const Probes& pr = pass == 0 ? pa : pb;
        const long* rlo = pass == 0 ? ra_lo : rb_lo;
        const long* rhi = pass == 0 ? ra_hi : rb_hi;
        const Quat qp = pass == 0 ? qa : qb, qs = pass == 0 ? qb : qa;
        const V3 tp = pass == 0 ? ta : tb, ts = pass == 0 ? tb : ta;
        const float* com_s = pass == 0 ? com_b : com_a;
        const float inv_p = pass == 0 ? inv_a : inv_b, inv_s = pass == 0 ? inv_b : inv_a;
        const V3 center_s = V3{com_s[0], com_s[1], com_s[2]} * inv_s;
        // aabb_from_voxel_ranges(extent of P, ranges).expanded_about_center(object_a.voxel_extent()) — A's extent in both passes
        const V3 lo_f{(float)rlo[0], (float)rlo[1], (float)rlo[2]}, hi_f{(float)rhi[0], (float)rhi[1], (float)rhi[2]};
        const V3 margin{A.extent, A.extent, A.extent};
        const Aab box{P.extent * lo_f - margin, P.extent * hi_f + margin};
        long clo[3], chi[3];
        for (int d = 0; d < 3; ++d) {
            clo[d] = rlo[d] / 16;
            chi[d] = (rhi[d] + 15) / 16;
        }
        for (uint32_t e = 0; e < pr.n_entries; ++e) {
            const uint32_t* en = pr.entries + 5 * (size_t)e;
            bool in = true;
            for (int d = 0; d < 3; ++d) in = in && (long)en[d] >= clo[d] && (long)en[d] < chi[d];
            if (!in) continue;
            for (uint32_t k = en[3]; k < en[4]; ++k) {
                const V3 pp{pr.points[3 * (size_t)k], pr.points[3 * (size_t)k + 1], pr.points[3 * (size_t)k + 2]};
                const V3 dl = pp - box.lo, dh = box.hi - pp;
                if (std::signbit(dl.x) || std::signbit(dl.y) || std::signbit(dl.z) || std::signbit(dh.x) || std::signbit(dh.y) || std::signbit(dh.z)) continue;
                const V3 point = qrot(conj(qp), pp - tp);  // inverse_transform_point
                const V3 np_s = (qrot(qs, point) + ts) * inv_s;
                float sd;
                V3 nn;
                if (!sdf_and_normal_if_intersecting(S, center_s, np_s, sd, nn)) continue;
                V3 sn = qrot(conj(qs), nn);
                if (pass == 1) sn = -sn;
                const float dep = -sd * S.extent;
                const V3 np_p = pp * inv_p;
                if (n < cap) {
                    which_ijk[4 * n] = pass;
                    which_ijk[4 * n + 1] = (int32_t)as_usize(np_p.x), which_ijk[4 * n + 2] = (int32_t)as_usize(np_p.y), which_ijk[4 * n + 3] = (int32_t)as_usize(np_p.z);
                    position[3 * n] = point.x, position[3 * n + 1] = point.y, position[3 * n + 2] = point.z;
                    normal[3 * n] = sn.x, normal[3 * n + 1] = sn.y, normal[3 * n + 2] = sn.z;
                    depth[n] = dep;
                }
                n += 1;
            }
        }
    }
    return n;
}

}  // namespace orc

