// §8f-3 / a13 — what turns an impact into fragment plane sets: fracture-point sampling, Delaunay tetrahedralisation, Voronoi cells.
// Host code (<= a few hundred points per impact; the reference runs it on the CPU too); its output — convex plane sets with
// bounding boxes — feeds ivx_clip_polyhedron / ivx_copy_polyhedra, where the voxels are.
//
// Reference (engine/crates/):
//   generate_impact_fracture_points / ..._region_boundary_points   impact_voxel/src/interaction/fracturing.rs:1710-2015
//   VoxelImpactFracturingConfig defaults                             fracturing.rs:855-871;  FracturingProperties fracturing.rs:61-86
//   DelaunayTetrahedralization::construct                            impact_tesselation/src/delaunay.rs:99-455
//     vertex numbering: four ad-hoc bounding vertices first, then the points in input order without near-coincident ones
//     (MIN_RELATIVE_POINT_SEPARATION, delaunay.rs:33-35, 168-191); compute_aabb 500-505; compute_boundary_face_planes 509-541;
//     compute_circumcenter 1724-1772
//   VoronoiPolyhedron::extract_from_delaunay_tetrahedra              impact_tesselation/src/voronoi.rs:75-252
//     compute_bounded_aabb 254-318, compute_plane_containing_three_points 337-351, orient_face_planes_outward 355-364
//   fragment = copy of the cell shrunk by 0.1 voxel                  fracturing.rs:1190-1240; region = hull of the boundary points 1537-1632
//
// The reference builds the tetrahedralisation by incremental insertion with Lawson flips and the `robust` crate's adaptive
// predicates; here it is Bowyer–Watson with the predicates evaluated in binary128 (the orientation test, degree 3, is exact for inputs whose
// coordinate differences fit ~35 bits and the in-sphere test, degree 5, for ~21 bits — lattice-like inputs; otherwise the rounding is 2^-113
// relative, far below f32 resolution, but not a guaranteed sign for exactly co-spherical points, where `robust::insphere` returns 0) and a cavity that is widened across faces the new point lies
// on, so no flat tetrahedron is ever created. For points in general position the Delaunay tetrahedralisation is unique, so both
// constructions give the same tetrahedra; for degenerate inputs (regular grids) both give A Delaunay tetrahedralisation, not
// necessarily the same one — and the order of tetrahedra, of a cell's planes and vertices is this implementation's own (the
// clip takes max over the planes: order-free). RNG: fastrand 2.3.0 (wyrand) restated from its published source — the crate is not
// vendored under /root/reference and no reference test pins its stream: PARITY UNPINNED for the sampled points.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <unordered_map>
#include <vector>

#include "ivx_internal.hpp"

namespace {

typedef __float128 q128;

struct P3 {
    float x, y, z;
};
inline P3 operator+(P3 a, P3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline P3 operator-(P3 a, P3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline P3 operator*(P3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline float dot(P3 a, P3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline P3 cross(P3 a, P3 b) { return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
inline P3 normalized(P3 a) {  // UnitVector3::normalized_from: v * (1 / |v|)
    const float r = 1.0f / std::sqrt(dot(a, a));
    return a * r;
}
inline float det3(P3 c0, P3 c1, P3 c2) {  // Matrix3::from_columns(..).determinant(): glam Mat3::determinant = c2 . (c0 x c1)
    return dot(c2, cross(c0, c1));
}

// > 0: d on the positive side of the plane through a, b, c (the side the normal (b - a) x (c - a) points to)
inline int orient(const P3& a, const P3& b, const P3& c, const P3& d) {
    const q128 ax = (q128)a.x - d.x, ay = (q128)a.y - d.y, az = (q128)a.z - d.z;
    const q128 bx = (q128)b.x - d.x, by = (q128)b.y - d.y, bz = (q128)b.z - d.z;
    const q128 cx = (q128)c.x - d.x, cy = (q128)c.y - d.y, cz = (q128)c.z - d.z;
    const q128 det = ax * (by * cz - bz * cy) - ay * (bx * cz - bz * cx) + az * (bx * cy - by * cx);
    return det > 0 ? -1 : (det < 0 ? 1 : 0);
}
// > 0: e strictly inside the circumsphere of the positively oriented tetrahedron a, b, c, d (orient(a, b, c, d) > 0)
inline int insphere(const P3& a, const P3& b, const P3& c, const P3& d, const P3& e) {
    q128 m[4][4];
    const P3* p[4] = {&a, &b, &c, &d};
    for (int i = 0; i < 4; ++i) {
        m[i][0] = (q128)p[i]->x - e.x;
        m[i][1] = (q128)p[i]->y - e.y;
        m[i][2] = (q128)p[i]->z - e.z;
        m[i][3] = m[i][0] * m[i][0] + m[i][1] * m[i][1] + m[i][2] * m[i][2];
    }
    auto d3 = [&](int r0, int r1, int r2, int c0, int c1, int c2) {
        return m[r0][c0] * (m[r1][c1] * m[r2][c2] - m[r1][c2] * m[r2][c1]) - m[r0][c1] * (m[r1][c0] * m[r2][c2] - m[r1][c2] * m[r2][c0]) +
               m[r0][c2] * (m[r1][c0] * m[r2][c1] - m[r1][c1] * m[r2][c0]);
    };
    const q128 det = -m[0][3] * d3(1, 2, 3, 0, 1, 2) + m[1][3] * d3(0, 2, 3, 0, 1, 2) - m[2][3] * d3(0, 1, 3, 0, 1, 2) + m[3][3] * d3(0, 1, 2, 0, 1, 2);
    // (the determinant is positive inside for det[a - d; b - d; c - d] > 0, i.e. for orient(a, b, c, d) < 0 in the convention above: with the
    // orientation make_positive keeps, orient > 0, inside is negative)
    return det < 0 ? 1 : (det > 0 ? -1 : 0);
}

struct Tet {
    uint32_t v[4];
    uint32_t nb[4];  // neighbour across the face opposite v[i] (IVX_NO_TETRAHEDRON: none)
    bool alive;
};

}  // namespace

struct ivx_delaunay {
    std::vector<P3> vertices;  // [0, 4): the ad-hoc bounding tetrahedron
    std::vector<Tet> tets;     // final: only tetrahedra of real points, neighbours linked
};

namespace {

constexpr uint32_t NONE = 0xFFFFFFFFu;

inline uint64_t face_key(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t s[3] = {a, b, c};
    std::sort(s, s + 3);
    return ((uint64_t)s[0] << 42) | ((uint64_t)s[1] << 21) | (uint64_t)s[2];
}

void make_positive(Tet& t, const std::vector<P3>& v) {
    if (orient(v[t.v[0]], v[t.v[1]], v[t.v[2]], v[t.v[3]]) < 0) std::swap(t.v[0], t.v[1]);
}

int construct(ivx_delaunay& D, const float* pts, size_t n) {
    D.vertices.clear();
    D.tets.clear();
    if (n < 4) return IVX_OK;  // delaunay.rs:161-163
    P3 lo = {pts[0], pts[1], pts[2]}, hi = lo;
    for (size_t i = 1; i < n; ++i) {
        lo = {std::min(lo.x, pts[3 * i]), std::min(lo.y, pts[3 * i + 1]), std::min(lo.z, pts[3 * i + 2])};
        hi = {std::max(hi.x, pts[3 * i]), std::max(hi.y, pts[3 * i + 1]), std::max(hi.z, pts[3 * i + 2])};
    }
    const P3 centre = (lo + hi) * 0.5f;
    const float radius = 0.5f * std::sqrt(dot(hi - lo, hi - lo));  // bounding sphere of the box
    if (!(radius > 0.0f)) return IVX_OK;
    const float min_sep2 = (1e-9f * radius) * (1e-9f * radius);
    // ad-hoc bounding tetrahedron, generous (the hull of the points must not feel it)
    const float s = 64.0f * radius + 1.0f;
    D.vertices = {centre + P3{s, s, s}, centre + P3{s, -s, -s}, centre + P3{-s, s, -s}, centre + P3{-s, -s, s}};
    std::vector<Tet>& T = D.tets;
    {
        Tet t{{0, 1, 2, 3}, {NONE, NONE, NONE, NONE}, true};
        make_positive(t, D.vertices);
        T.push_back(t);
    }
    std::vector<uint32_t> bad;
    std::vector<uint8_t> in_cavity;
    for (size_t pi = 0; pi < n; ++pi) {
        const P3 p = {pts[3 * pi], pts[3 * pi + 1], pts[3 * pi + 2]};
        // tetrahedra whose circumsphere strictly contains p (linear scan: a few hundred points per impact)
        bad.clear();
        in_cavity.assign(T.size(), 0);
        bool coincident = false;
        for (uint32_t t = 0; t < T.size() && !coincident; ++t) {
            if (!T[t].alive) continue;
            const Tet& tt = T[t];
            if (insphere(D.vertices[tt.v[0]], D.vertices[tt.v[1]], D.vertices[tt.v[2]], D.vertices[tt.v[3]], p) > 0) {
                for (int c = 0; c < 4; ++c) {
                    const P3 d = D.vertices[tt.v[c]] - p;
                    if (dot(d, d) < min_sep2) coincident = true;
                }
                bad.push_back(t);
                in_cavity[t] = 1;
            }
        }
        if (coincident || bad.empty()) {
            // (a point that duplicates a vertex lies ON circumspheres only; one that coincides within the separation is skipped
            // as the reference skips it)
            if (!coincident) {
                for (const P3& q : D.vertices)
                    if (dot(q - p, q - p) < min_sep2) coincident = true;
                if (!coincident) return IVX_ERR_INVALID;  // cannot happen inside the bounding tetrahedron
            }
            continue;
        }
        // neighbour lookup for the cavity boundary: faces of alive tetrahedra
        // (rebuilt per insertion from a hash of the alive tetrahedra's faces: simple, and n is small)
        std::unordered_map<uint64_t, std::pair<uint32_t, uint32_t>> owners;
        owners.reserve(T.size() * 4);
        for (uint32_t t = 0; t < T.size(); ++t) {
            if (!T[t].alive) continue;
            for (int c = 0; c < 4; ++c) {
                const uint64_t k = face_key(T[t].v[(c + 1) & 3], T[t].v[(c + 2) & 3], T[t].v[(c + 3) & 3]);
                auto it = owners.find(k);
                if (it == owners.end()) owners.emplace(k, std::make_pair(t, NONE));
                else it->second.second = t;
            }
        }
        auto across = [&](uint32_t t, int c) -> uint32_t {
            const auto& o = owners[face_key(T[t].v[(c + 1) & 3], T[t].v[(c + 2) & 3], T[t].v[(c + 3) & 3])];
            return o.first == t ? o.second : o.first;
        };
        // widen the cavity across boundary faces p is not strictly in front of (p on the face plane: the neighbour is cospherical
        // with p; taking it in keeps every new tetrahedron non-flat)
        for (size_t qi = 0; qi < bad.size(); ++qi) {
            const uint32_t t = bad[qi];
            for (int c = 0; c < 4; ++c) {
                const uint32_t nb = across(t, c);
                if (nb == NONE || in_cavity[nb]) continue;
                // face opposite corner c, seen from inside tetrahedron t: corner c is on its positive side
                const uint32_t f0 = T[t].v[(c + 1) & 3], f1 = T[t].v[(c + 2) & 3], f2 = T[t].v[(c + 3) & 3];
                const int side_c = orient(D.vertices[f0], D.vertices[f1], D.vertices[f2], D.vertices[T[t].v[c]]);
                const int side_p = orient(D.vertices[f0], D.vertices[f1], D.vertices[f2], p);
                if (side_p == 0 || side_p != side_c) {
                    in_cavity[nb] = 1;
                    bad.push_back(nb);
                }
            }
        }
        const uint32_t pv = (uint32_t)D.vertices.size();
        D.vertices.push_back(p);
        for (const uint32_t t : bad) {
            for (int c = 0; c < 4; ++c) {
                const uint32_t nb = across(t, c);
                if (nb != NONE && in_cavity[nb]) continue;
                Tet nt{{T[t].v[(c + 1) & 3], T[t].v[(c + 2) & 3], T[t].v[(c + 3) & 3], pv}, {NONE, NONE, NONE, NONE}, true};
                make_positive(nt, D.vertices);
                T.push_back(nt);
            }
        }
        for (const uint32_t t : bad) T[t].alive = false;
        // compact now and then
        if (T.size() > 4096 && T.size() > 8 * (D.vertices.size() + 8)) {
            std::vector<Tet> keep;
            for (const Tet& t : T)
                if (t.alive) keep.push_back(t);
            T.swap(keep);
        }
    }
    // remove_boundary_tetrahedra (delaunay.rs:794-884): drop everything that touches an ad-hoc vertex, link the rest
    std::vector<Tet> keep;
    for (const Tet& t : T)
        if (t.alive && t.v[0] >= 4 && t.v[1] >= 4 && t.v[2] >= 4 && t.v[3] >= 4) keep.push_back(t);
    T.swap(keep);
    std::unordered_map<uint64_t, std::pair<uint32_t, int>> open;
    open.reserve(T.size() * 4);
    for (uint32_t t = 0; t < T.size(); ++t)
        for (int c = 0; c < 4; ++c) {
            const uint64_t k = face_key(T[t].v[(c + 1) & 3], T[t].v[(c + 2) & 3], T[t].v[(c + 3) & 3]);
            auto it = open.find(k);
            if (it == open.end()) open.emplace(k, std::make_pair(t, c));
            else {
                T[t].nb[c] = it->second.first;
                T[it->second.first].nb[it->second.second] = t;
                open.erase(it);
            }
        }
    return IVX_OK;
}

// delaunay.rs:1724-1772 (f32, glam operation order)
P3 circumcenter(const P3& a, const P3& b, const P3& c, const P3& d) {
    const P3 da = a - d, db = b - d, dc = c - d;
    const float da2 = dot(da, da), db2 = dot(db, db), dc2 = dot(dc, dc);
    const float det_r = det3(da, db, dc);
    const float det_x = det3({da2, da.y, da.z}, {db2, db.y, db.z}, {dc2, dc.y, dc.z});
    const float det_y = det3({da2, da.x, da.z}, {db2, db.x, db.z}, {dc2, dc.x, dc.z});
    const float det_z = det3({da2, da.x, da.y}, {db2, db.x, db.y}, {dc2, dc.x, dc.y});
    const float scale = 1.0f / (2.0f * det_r);
    return d + P3{scale * det_x, -scale * det_y, scale * det_z};
}

inline bool rel_eq(float a, float b) {  // approx::relative_eq!(epsilon = 1e-5, max_relative = 1e-5)
    if (a == b) return true;
    const float diff = std::fabs(a - b);
    if (diff <= 1e-5f) return true;
    return diff <= std::max(std::fabs(a), std::fabs(b)) * 1e-5f;
}
inline bool rel_eq(const P3& a, const P3& b) { return rel_eq(a.x, b.x) && rel_eq(a.y, b.y) && rel_eq(a.z, b.z); }

}  // namespace

extern "C" {

int ivx_delaunay_construct(const float* points3, size_t n_points, ivx_delaunay** out) {
    IVX_REQUIRE(out && (points3 || n_points == 0), IVX_ERR_INVALID, "ivx_delaunay_construct: null argument");
    IVX_REQUIRE(n_points + 4 < (1u << 21), IVX_ERR_CAPACITY, "ivx_delaunay_construct: too many points");
    ivx_delaunay* d = new (std::nothrow) ivx_delaunay();
    IVX_REQUIRE(d, IVX_ERR_CAPACITY, "ivx_delaunay_construct: out of memory");
    const int rc = construct(*d, points3, n_points);
    if (rc) {
        delete d;
        ivx_set_error("ivx_delaunay_construct: a point fell outside the bounding tetrahedron (non-finite coordinates?)");
        return rc;
    }
    *out = d;
    return IVX_OK;
}

void ivx_delaunay_destroy(ivx_delaunay* d) { delete d; }

// counts[0] vertices (the four ad-hoc ones included: real points are 4..), counts[1] tetrahedra
int ivx_delaunay_counts(const ivx_delaunay* d, uint32_t counts[2]) {
    IVX_REQUIRE(d && counts, IVX_ERR_INVALID, "ivx_delaunay_counts: null argument");
    counts[0] = (uint32_t)d->vertices.size();
    counts[1] = (uint32_t)d->tets.size();
    return IVX_OK;
}

int ivx_delaunay_download(const ivx_delaunay* d, float* vertices3, uint32_t* tet_vertices4, uint32_t* tet_neighbors4) {
    IVX_REQUIRE(d, IVX_ERR_INVALID, "ivx_delaunay_download: null argument");
    if (vertices3) memcpy(vertices3, d->vertices.data(), d->vertices.size() * sizeof(P3));
    for (size_t t = 0; t < d->tets.size(); ++t)
        for (int c = 0; c < 4; ++c) {
            if (tet_vertices4) tet_vertices4[4 * t + c] = d->tets[t].v[c];
            if (tet_neighbors4) tet_neighbors4[4 * t + c] = d->tets[t].nb[c];
        }
    return IVX_OK;
}

// compute_aabb (delaunay.rs:500-505): of the real points; returns IVX_ERR_STATE when there are none
// DelaunayTetrahedralization::displace_vertices (delaunay.rs; FracturingProcess::offset_tetrahedralization_to_fracture_region_object,
// fracturing.rs:996-1002): every vertex, the four ad-hoc ones included, moves by `offset` in f32 — AFTER the construction, which has seen the
// points where they were given (their bounding sphere, the minimum point separation, every predicate).
int ivx_delaunay_displace_vertices(ivx_delaunay* d, const float offset[3]) {
    IVX_REQUIRE(d && offset, IVX_ERR_INVALID, "ivx_delaunay_displace_vertices: null argument");
    for (P3& v : d->vertices) v = v + P3{offset[0], offset[1], offset[2]};
    return IVX_OK;
}

int ivx_delaunay_aabb(const ivx_delaunay* d, float aabb[6]) {
    IVX_REQUIRE(d && aabb, IVX_ERR_INVALID, "ivx_delaunay_aabb: null argument");
    IVX_REQUIRE(d->vertices.size() > 4, IVX_ERR_STATE, "ivx_delaunay_aabb: empty tetrahedralization");
    P3 lo = d->vertices[4], hi = lo;
    for (size_t i = 5; i < d->vertices.size(); ++i) {
        const P3& p = d->vertices[i];
        lo = {std::min(lo.x, p.x), std::min(lo.y, p.y), std::min(lo.z, p.z)};
        hi = {std::max(hi.x, p.x), std::max(hi.y, p.y), std::max(hi.z, p.z)};
    }
    aabb[0] = lo.x, aabb[1] = lo.y, aabb[2] = lo.z, aabb[3] = hi.x, aabb[4] = hi.y, aabb[5] = hi.z;
    return IVX_OK;
}

// compute_boundary_face_planes (delaunay.rs:509-541): outward unit normal + displacement of every hull face
int ivx_delaunay_boundary_face_planes(const ivx_delaunay* d, float* planes4, size_t cap, size_t* n_out) {
    IVX_REQUIRE(d && n_out, IVX_ERR_INVALID, "ivx_delaunay_boundary_face_planes: null argument");
    size_t n = 0;
    for (const Tet& t : d->tets)
        for (int c = 0; c < 4; ++c) {
            if (t.nb[c] != NONE) continue;
            if (planes4 && n < cap) {
                // the face's vertices ordered so that the normal points away from the opposite corner
                const P3& v1 = d->vertices[t.v[(c + 1) & 3]];
                P3 v2 = d->vertices[t.v[(c + 2) & 3]], v3 = d->vertices[t.v[(c + 3) & 3]];
                if (orient(v1, v2, v3, d->vertices[t.v[c]]) > 0) std::swap(v2, v3);
                const P3 nrm = normalized(cross(v2 - v1, v3 - v1));
                planes4[4 * n] = nrm.x, planes4[4 * n + 1] = nrm.y, planes4[4 * n + 2] = nrm.z, planes4[4 * n + 3] = dot(nrm, v1);
            }
            n += 1;
        }
    *n_out = n;
    IVX_REQUIRE(!planes4 || n <= cap, IVX_ERR_CAPACITY, "ivx_delaunay_boundary_face_planes: %zu planes exceed capacity %zu", n, cap);
    return IVX_OK;
}

// VoronoiPolyhedron::extract_from_delaunay_tetrahedra (voronoi.rs:75-252) for the vertex `vertex` (>= 4): the cell's vertices
// (circumcentres of the incident tetrahedra), its rays (cells on the hull are open: outward directions at hull faces) and its face
// planes, oriented outward. Counts come back in n_out[3] (vertices, rays, planes); buffers may be null to ask for the counts.
int ivx_voronoi_polyhedron(const ivx_delaunay* d, uint32_t vertex, float* vertices3, size_t cap_v, float* rays6, size_t cap_r, float* planes4, size_t cap_p,
                           size_t n_out[3]) {
    IVX_REQUIRE(d && n_out, IVX_ERR_INVALID, "ivx_voronoi_polyhedron: null argument");
    n_out[0] = n_out[1] = n_out[2] = 0;
    if (vertex >= d->vertices.size() || vertex < 4) return IVX_OK;
    const P3 site = d->vertices[vertex];
    struct Partial {
        uint32_t end;
        P3 pts[3];
        uint32_t count;
    };
    std::vector<Partial> partial;
    std::vector<uint8_t> completed(d->vertices.size(), 0);
    std::vector<float> V, R, PL;
    auto push_plane = [&](P3 nrm, float disp) {
        PL.push_back(nrm.x), PL.push_back(nrm.y), PL.push_back(nrm.z), PL.push_back(disp);
    };
    for (const Tet& t : d->tets) {
        int corner = -1;
        for (int c = 0; c < 4; ++c)
            if (t.v[c] == vertex) corner = c;
        if (corner < 0) continue;
        const P3 cc = circumcenter(d->vertices[t.v[0]], d->vertices[t.v[1]], d->vertices[t.v[2]], d->vertices[t.v[3]]);
        V.push_back(cc.x), V.push_back(cc.y), V.push_back(cc.z);
        // the three faces that contain the site: a face without a neighbour is on the hull
        for (int k = 1; k < 4; ++k) {
            const int opp = (corner + k) & 3;  // the face opposite corner `opp` contains the site
            if (t.nb[opp] != NONE) continue;
            uint32_t e[2];
            int ne = 0;
            for (int c = 0; c < 4; ++c)
                if (c != opp && c != corner) e[ne++] = t.v[c];
            P3 p1 = d->vertices[e[0]], p2 = d->vertices[e[1]];
            // order the two so that (p2 - site) x (p1 - site) points away from the tetrahedron (outward at the hull face)
            if (orient(site, p1, p2, d->vertices[t.v[opp]]) < 0) {
                std::swap(p1, p2);
                std::swap(e[0], e[1]);
            }
            const P3 e1 = p1 - site, e2 = p2 - site;
            const P3 dir = normalized(cross(e2, e1));
            R.push_back(cc.x), R.push_back(cc.y), R.push_back(cc.z), R.push_back(dir.x), R.push_back(dir.y), R.push_back(dir.z);
            const P3 edges[2] = {e1, e2};
            for (int q = 0; q < 2; ++q)
                if (!completed[e[q]]) {
                    completed[e[q]] = 1;
                    const P3 nrm = normalized(edges[q]);
                    push_plane(nrm, dot(nrm, cc));  // the bisector plane passes through the circumcentre
                }
        }
        for (int c = 0; c < 4; ++c) {
            const uint32_t end = t.v[c];
            if (end == vertex || completed[end]) continue;
            Partial* pp = nullptr;
            for (Partial& q : partial)
                if (q.end == end) pp = &q;
            if (!pp) {
                partial.push_back(Partial{end, {cc, cc, cc}, 1});
                continue;
            }
            bool dup = false;
            for (uint32_t i = 0; i < pp->count; ++i) dup = dup || rel_eq(cc, pp->pts[i]);
            if (dup) continue;
            pp->pts[pp->count++] = cc;
            if (pp->count == 3) {
                const P3 nrm = normalized(cross(pp->pts[1] - pp->pts[0], pp->pts[2] - pp->pts[0]));
                push_plane(nrm, dot(nrm, pp->pts[0]));
                completed[end] = 1;
                *pp = partial.back();
                partial.pop_back();
            }
        }
    }
    // orient_face_planes_outward: the site is inside
    for (size_t i = 0; i < PL.size(); i += 4) {
        const float sd = ((PL[i] * site.x + PL[i + 1] * site.y) + PL[i + 2] * site.z) - PL[i + 3];
        if (!std::signbit(sd)) {
            PL[i] = -PL[i], PL[i + 1] = -PL[i + 1], PL[i + 2] = -PL[i + 2], PL[i + 3] = -PL[i + 3];
        }
    }
    n_out[0] = V.size() / 3, n_out[1] = R.size() / 6, n_out[2] = PL.size() / 4;
    IVX_REQUIRE((!vertices3 || n_out[0] <= cap_v) && (!rays6 || n_out[1] <= cap_r) && (!planes4 || n_out[2] <= cap_p), IVX_ERR_CAPACITY,
                "ivx_voronoi_polyhedron: %zu vertices / %zu rays / %zu planes exceed the capacities", n_out[0], n_out[1], n_out[2]);
    if (vertices3) memcpy(vertices3, V.data(), V.size() * 4);
    if (rays6) memcpy(rays6, R.data(), R.size() * 4);
    if (planes4) memcpy(planes4, PL.data(), PL.size() * 4);
    return IVX_OK;
}

// compute_bounded_aabb (voronoi.rs:254-318): the cell's box inside `bounding`; *has = 0 when they do not overlap (or no vertices)
int ivx_voronoi_bounded_aabb(const float* vertices3, size_t n_vertices, const float* rays6, size_t n_rays, const float bounding[6], float out[6], int* has) {
    IVX_REQUIRE(bounding && out && has && (vertices3 || n_vertices == 0) && (rays6 || n_rays == 0), IVX_ERR_INVALID, "ivx_voronoi_bounded_aabb: null argument");
    *has = 0;
    if (n_vertices == 0) return IVX_OK;
    float lo[3] = {vertices3[0], vertices3[1], vertices3[2]}, hi[3] = {lo[0], lo[1], lo[2]};
    auto expand = [&](const float* p) {
        for (int q = 0; q < 3; ++q) {
            lo[q] = std::min(lo[q], p[q]);
            hi[q] = std::max(hi[q], p[q]);
        }
    };
    for (size_t i = 1; i < n_vertices; ++i) expand(vertices3 + 3 * i);
    for (size_t r = 0; r < n_rays; ++r) {
        const float* v = rays6 + 6 * r;
        const float* dir = v + 3;
        float dist = INFINITY;
        for (int q = 0; q < 3; ++q) {
            const float rec = 1.0f / dir[q];
            const float lower = -((v[q] - bounding[q]) * rec), upper = (bounding[3 + q] - v[q]) * rec;
            float nd = -1.0f;
            if (dir[q] > 0.0f) nd = upper;
            else if (dir[q] < 0.0f) nd = lower;
            else continue;
            if (nd >= 0.0f) dist = std::min(dist, nd);
        }
        if (std::isfinite(dist)) {
            const float p[3] = {v[0] + dist * dir[0], v[1] + dist * dir[1], v[2] + dist * dir[2]};
            expand(p);
        }
    }
    for (int q = 0; q < 3; ++q) {  // compute_overlap_with
        out[q] = std::max(lo[q], bounding[q]);
        out[3 + q] = std::min(hi[q], bounding[3 + q]);
        if (out[3 + q] - out[q] < 0.0f) return IVX_OK;  // has_negative_component: a box of zero extent still counts
    }
    *has = 1;
    return IVX_OK;
}

}  // extern "C"

// ---- a13: fracture points ---------------------------------------------------------------------------------------------------------
namespace {

// fastrand 2.3.0 (Cargo.lock:922; wyrand with the wyhash v4.2 constants the 2.1 release moved to), restated from the crate's published
// source (src/lib.rs: gen_u64, gen_mod_u64 = Lemire's method, f32 from the upper 23 bits). The crate is not vendored under /root/reference
// and no reference test holds a value of the stream: PARITY UNPINNED for the random draws (everything downstream of them is pinned).
struct Wyrand {
    uint64_t s;
    uint64_t u64() {
        s += 0x2d358dccaa6c78a5ull;
        const unsigned __int128 t = (unsigned __int128)s * (unsigned __int128)(s ^ 0x8bb84b93962eacc9ull);
        return (uint64_t)(t >> 64) ^ (uint64_t)t;
    }
    uint32_t u32() { return (uint32_t)u64(); }
    uint64_t below(uint64_t n) {
        uint64_t r = u64();
        unsigned __int128 m = (unsigned __int128)r * n;
        uint64_t lo = (uint64_t)m;
        if (lo < n) {
            const uint64_t t = (0 - n) % n;
            while (lo < t) {
                r = u64();
                m = (unsigned __int128)r * n;
                lo = (uint64_t)m;
            }
        }
        return (uint64_t)(m >> 64);
    }
    float f32() {
        const uint32_t bits = (1u << 30) - (1u << 23) + (u32() >> 9);  // [1, 2)
        float f;
        memcpy(&f, &bits, 4);
        return f - 1.0f;
    }
};

struct Iso {
    float q[4];  // xyzw
    P3 t;
};
inline P3 qrot(const float q[4], P3 v) {  // glam Quat::mul_vec3a
    const P3 b = {q[0], q[1], q[2]};
    const float w = q[3], b2 = dot(b, b);
    return (v * (w * w - b2) + b * (dot(v, b) * 2.0f)) + cross(b, v) * (w * 2.0f);
}
inline void qmul(const float a[4], const float b[4], float o[4]) {
    o[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    o[1] = a[3] * b[1] - a[0] * b[2] + a[1] * b[3] + a[2] * b[0];
    o[2] = a[3] * b[2] + a[0] * b[1] - a[1] * b[0] + a[2] * b[3];
    o[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
}
inline P3 xform(const Iso& i, P3 p) { return qrot(i.q, p) + i.t; }
inline Iso compose(const Iso& a, const Iso& b) {  // a * b: apply b, then a
    Iso o;
    qmul(a.q, b.q, o.q);
    o.t = qrot(a.q, b.t) + a.t;
    return o;
}
// UnitQuaternion::rotation_between_axes(z, dir) = glam Quat::from_rotation_arc (impact_math/src/quaternion.rs:345-350; glam 0.30 is not
// vendored: restated from its published source — identity above 1 - 2 eps, half a turn about from.any_orthonormal_vector() = +y below
// the negative of that, else normalize(cross, 1 + dot) with the SSE2 path's (x^2 + z^2) + (y^2 + w^2) and a division)
void rotation_from_z(const P3 to, float q[4]) {
    const P3 from = {0.0f, 0.0f, 1.0f};
    const float d = dot(from, to);
    const float one_minus_eps = 1.0f - 2.0f * 1.1920929e-07f;
    if (d > one_minus_eps) {
        q[0] = q[1] = q[2] = 0.0f, q[3] = 1.0f;
    } else if (d < -one_minus_eps) {
        const float half = 3.14159274f * 0.5f;
        q[0] = 0.0f, q[1] = std::sin(half), q[2] = 0.0f, q[3] = std::cos(half);
    } else {
        const P3 c = cross(from, to);
        const float qq[4] = {c.x, c.y, c.z, 1.0f + d};
        const float l = std::sqrt((qq[0] * qq[0] + qq[2] * qq[2]) + (qq[1] * qq[1] + qq[3] * qq[3]));
        for (int i = 0; i < 4; ++i) q[i] = qq[i] / l;
    }
}

}  // namespace

extern "C" {

void ivx_impact_fracturing_config_default(ivx_impact_fracturing_config* c) {
    if (!c) return;
    c->boundary_polar_grid_size = 3;
    c->boundary_azimuthal_grid_size = 6;
    c->boundary_angular_jitter = 0.8f;
    c->boundary_radial_jitter = 0.2f;
    c->max_fragment_count = 512;
    c->radial_falloff_power = 2.0f;
    c->angular_falloff_power = 0.5f;
    c->radial_grid_size = 128;
    c->angular_grid_size = 128;
    c->max_position_rejections_per_sample = 128;
    c->seed = 0;
}

// generate_impact_fracture_points (fracturing.rs:1710-1941) + generate_impact_fracture_region_boundary_points (1945-2015). Points come back
// in the object's normalized space (voxel units). `rng_state`: in/out (Rng::with_seed(config.seed) = the seed itself for the first impact
// of a frame; the reference threads one generator through all impacts of a frame). Counts may come back 0 (force below the threshold,
// region too small).
int ivx_generate_impact_fracture_points(const ivx_impact_fracturing_config* cfg, const ivx_fracturing_properties* props, float inverse_voxel_extent,
                                        const float world_to_object_rotation[4], const float world_to_object_translation[3], const float aabb[6],
                                        const float force_position[3], const float force_direction[3], float force_magnitude, uint64_t* rng_state,
                                        float* boundary_points3, size_t cap_boundary, size_t* n_boundary, float* fracture_points3, size_t cap_fracture,
                                        size_t* n_fracture) {
    IVX_REQUIRE(cfg && props && world_to_object_rotation && world_to_object_translation && aabb && force_position && force_direction && rng_state && n_boundary &&
                    n_fracture,
                IVX_ERR_INVALID, "ivx_generate_impact_fracture_points: null argument");
    IVX_REQUIRE(cfg->radial_grid_size >= 2 && cfg->angular_grid_size >= 2, IVX_ERR_INVALID, "ivx_generate_impact_fracture_points: grid sizes must be at least 2");
    *n_boundary = *n_fracture = 0;
    Wyrand rng{*rng_state};
    Iso force_to_world;
    rotation_from_z({force_direction[0], force_direction[1], force_direction[2]}, force_to_world.q);
    force_to_world.t = {force_position[0], force_position[1], force_position[2]};
    Iso w2o;
    memcpy(w2o.q, world_to_object_rotation, 16);
    w2o.t = {world_to_object_translation[0], world_to_object_translation[1], world_to_object_translation[2]};
    const Iso f2o = compose(w2o, force_to_world);
    const float volume = ((aabb[3] - aabb[0]) * (aabb[4] - aabb[1])) * (aabb[5] - aabb[2]);
    const float object_extent = std::cbrt(volume);
    IVX_REQUIRE(object_extent > 0.0f, IVX_ERR_INVALID, "ivx_generate_impact_fracture_points: empty bounding box");
    const float relative_force = force_magnitude / props->fracturing_force;
    if (relative_force <= 1.0f) return IVX_OK;
    const float shattering_force = props->shattering_pressure * (object_extent * object_extent);
    const float fragment_scale = props->fragment_scale * object_extent;
    const float min_fragment_extent = props->min_fragment_extent * std::sqrt(object_extent);
    const float max_fragment_extent = props->max_fragment_extent * object_extent;
    const float radial_power = cfg->radial_falloff_power, angular_power = cfg->angular_falloff_power;
    float contact_extent = object_extent / std::max(std::pow(shattering_force / props->fracturing_force, 1.0f / radial_power) - 1.0f, 0.0f);
    contact_extent = std::min(contact_extent, object_extent);
    const float fracture_region_extent = std::max(contact_extent * (std::pow(relative_force, 1.0f / radial_power) - 1.0f), 0.0f);
    if (fracture_region_extent < min_fragment_extent) return IVX_OK;
    const float radial_scale = 1.0f / contact_extent;
    const float min_relative_load = fragment_scale / max_fragment_extent + 1.0f;
    const uint32_t nr = cfg->radial_grid_size, nu = cfg->angular_grid_size;
    const float dr = fracture_region_extent / (float)(nr - 1), du = 1.0f / (float)(nu - 1);
    const float TWO_PI = 6.28318530717958647692f;
    std::vector<float> n_dv((size_t)nr * nu), extent_grid((size_t)nr * nu);
    for (uint32_t ui = 0; ui < nu; ++ui) {
        const float u = du * (float)ui;
        for (uint32_t ri = 0; ri < nr; ++ri) {
            const float r = dr * (float)ri;
            const float load = (relative_force * std::pow(r * radial_scale + 1.0f, -radial_power)) * std::pow(u, angular_power);
            float fe = fragment_scale / (std::max(load, min_relative_load) - 1.0f);
            fe = std::max(fe, min_fragment_extent);
            const float nd = 1.0f / ((fe * fe) * fe);
            n_dv[(size_t)ui * nr + ri] = (TWO_PI * (r * r)) * nd;
            extent_grid[(size_t)ui * nr + ri] = fe;
        }
    }
    float max_n_dv = -INFINITY;
    for (uint32_t ri = 0; ri < nr; ++ri) max_n_dv = std::max(max_n_dv, n_dv[(size_t)(nu - 1) * nr + ri]);
    float sum = 0.0f;
    for (const float v : n_dv) sum += v;
    const float integrated = (sum * dr) * du;
    const uint64_t max_samples = std::min<uint64_t>((uint64_t)std::max(std::floor(integrated), 1.0f), cfg->max_fragment_count);
    const uint64_t max_rejections = (uint64_t)cfg->max_position_rejections_per_sample * max_samples;
    uint64_t samples = 0, rejections = 0;
    std::vector<P3> pts;
    while (samples < max_samples && rejections < max_rejections) {
        const uint32_t ri = (uint32_t)rng.below(nr), ui = (uint32_t)rng.below(nu);
        const size_t idx = (size_t)ui * nr + ri;
        const float frac = rng.f32();
        if (frac * max_n_dv > n_dv[idx]) continue;
        const float fe = extent_grid[idx];
        const float r = dr * (float)ri;
        if (fracture_region_extent - r < 0.5f * fe) {
            rejections += 1;
            continue;
        }
        const float phi = TWO_PI * rng.f32();
        const float sin_phi = std::sin(phi), cos_phi = std::cos(phi);
        const float cos_theta = du * (float)ui;
        const float sin_theta = std::sqrt(std::max(1.0f - cos_theta * cos_theta, 0.0f));
        const P3 fp = {(r * sin_theta) * cos_phi, (r * sin_theta) * sin_phi, r * cos_theta};
        const P3 op = xform(f2o, fp);
        if (!(op.x >= aabb[0] && op.y >= aabb[1] && op.z >= aabb[2] && op.x <= aabb[3] && op.y <= aabb[4] && op.z <= aabb[5])) {
            rejections += 1;
            continue;
        }
        const P3 sp = op * inverse_voxel_extent;
        const float md = fe * inverse_voxel_extent, min_d2 = md * md;
        bool close = false;
        for (const P3& q : pts)
            if (dot(sp - q, sp - q) < min_d2) {
                close = true;
                break;
            }
        if (close) {
            rejections += 1;
            continue;
        }
        pts.push_back(sp);
        samples += 1;
    }
    // boundary points: stratified on a hemisphere of radius fracture_region_extent, plus the apex of the opposite hemisphere
    std::vector<P3> bnd;
    {
        const uint32_t bu = cfg->boundary_polar_grid_size, bphi = cfg->boundary_azimuthal_grid_size;
        const float du_b = 1.0f / (float)bu, dphi = TWO_PI / (float)bphi;
        const float u_jit = du_b * cfg->boundary_angular_jitter, phi_jit = dphi * cfg->boundary_angular_jitter, r_jit = fracture_region_extent * cfg->boundary_radial_jitter;
        for (uint32_t ui = 0; ui < bu; ++ui) {
            const float uc = du_b * ((float)ui + 0.5f);
            for (uint32_t pi = 0; pi < bphi; ++pi) {
                const float pc = dphi * ((float)pi + 0.5f);
                const float u = uc + u_jit * (0.5f - rng.f32());
                const float phi = pc + phi_jit * (0.5f - rng.f32());
                const float r = fracture_region_extent + r_jit * (0.5f - rng.f32());
                const float cos_theta = std::min(std::max(u, 0.0f), 1.0f);
                const float sin_theta = std::sqrt(std::max(1.0f - u * u, 0.0f));
                const float sin_phi = std::sin(phi), cos_phi = std::cos(phi);
                bnd.push_back(xform(f2o, {(r * sin_theta) * cos_phi, (r * sin_theta) * sin_phi, r * cos_theta}) * inverse_voxel_extent);
            }
        }
        bnd.push_back(xform(f2o, {0.0f, 0.0f, -fracture_region_extent}) * inverse_voxel_extent);
    }
    *rng_state = rng.s;
    *n_boundary = bnd.size();
    *n_fracture = pts.size();
    IVX_REQUIRE((!boundary_points3 || bnd.size() <= cap_boundary) && (!fracture_points3 || pts.size() <= cap_fracture), IVX_ERR_CAPACITY,
                "ivx_generate_impact_fracture_points: %zu boundary / %zu fracture points exceed the capacities", bnd.size(), pts.size());
    if (boundary_points3) memcpy(boundary_points3, bnd.data(), bnd.size() * sizeof(P3));
    if (fracture_points3) memcpy(fracture_points3, pts.data(), pts.size() * sizeof(P3));
    return IVX_OK;
}

}  // extern "C"
