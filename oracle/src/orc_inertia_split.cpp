// ORACLE (test infrastructure only — never linked into the product library).
//
// Mass / inertia moments:
//   compute_moments_for_non_uniform_chunk   object/inertia.rs:615-699 (f32 running coordinates)
//   compute_moments_for_uniform_chunk       object/inertia.rs:703-754
//   compute_inertial_property_moments_...   object/inertia.rs:756-790
//   compute_inertial_properties_from_moments object/inertia.rs:288-326
//   parallel-axis deltas                    impact_physics/src/inertia.rs:511-546
// plus an exact f64 evaluation of the same cube integrals (the reference's f32 running sums carry
// ~1e-3 relative error at 256^3, SURVEY.md §7 "hard parts"; the GPU kernel is compared to the f64 form).
//
// Connected regions: brute-force union-find over all voxels exactly as the reference's own validator
// count_regions_brute_force (object/split_detection.rs:498-562), producing the partition that
// count_regions()/find_two_disconnected_regions() (193-301) must agree with.
#include <cstring>
#include <numeric>

#include "orc_voxel.hpp"

namespace orc {

struct Moments32 {
    float mass;
    V3 moments, moi, poi;
};

static void chunk_moments_non_uniform(float e, const Voxel* v, const float* dens, const int ci[3], Moments32& out) {
    float mass = 0.0f;
    V3 mo{0, 0, 0}, mi{0, 0, 0}, pi{0, 0, 0};
    float x0 = ((float)(ci[0] * CHUNK)) * e, y0 = ((float)(ci[1] * CHUNK)) * e, z0 = ((float)(ci[2] * CHUNK)) * e;
    int idx = 0;
    float xl = x0, xh = xl + e;
    for (int i = 0; i < CHUNK; ++i) {
        float yl = y0, yh = yl + e;
        float xl2 = xl * xl, xh2 = xh * xh, xl3 = xl2 * xl, xh3 = xh2 * xh;
        float dx2 = xh2 - xl2, dx3 = xh3 - xl3;
        for (int j = 0; j < CHUNK; ++j) {
            float zl = z0, zh = zl + e;
            float yl2 = yl * yl, yh2 = yh * yh, yl3 = yl2 * yl, yh3 = yh2 * yh;
            float dy2 = yh2 - yl2, dy3 = yh3 - yl3;
            for (int k = 0; k < CHUNK; ++k) {
                const Voxel& x = v[idx];
                if (!x.empty()) {
                    float zl2 = zl * zl, zh2 = zh * zh, zl3 = zl2 * zl, zh3 = zh2 * zh;
                    float dz2 = zh2 - zl2, dz3 = zh3 - zl3;
                    float d = dens[x.type];
                    V3 h2 = v3(dx2, dy2, dz2), h3 = v3(dx3, dy3, dz3);
                    mass += d;
                    mo = mo + d * h2;
                    mi = mi + d * (v3(h3.y, h3.x, h3.x) + v3(h3.z, h3.z, h3.y));
                    pi = pi + d * cmul(h2, v3(h2.y, h2.z, h2.x));
                }
                idx += 1;
                zl = zh;
                zh += e;
            }
            yl = yh;
            yh += e;
        }
        xl = xh;
        xh += e;
    }
    float e2 = e * e;  // powi(2)
    float e3 = e2 * e;
    out.mass = mass * e3;
    out.moments = mo * (0.5f * e2);
    out.moi = mi * ((1.0f / 3.0f) * e2);
    out.poi = pi * (0.25f * e);
}

static void chunk_moments_uniform(float e, const float* dens, uint8_t type, const int ci[3], Moments32& out) {
    float density = dens[type];
    float ce = ((float)CHUNK) * e;
    float l[3], h[3], d2[3], d3[3];
    for (int d = 0; d < 3; ++d) {
        l[d] = ((float)ci[d]) * ce;
        h[d] = l[d] + ce;
        float l2 = l[d] * l[d], h2 = h[d] * h[d], l3 = l2 * l[d], h3 = h2 * h[d];
        d2[d] = h2 - l2;
        d3[d] = h3 - l3;
    }
    V3 h2 = v3(d2[0], d2[1], d2[2]), h3 = v3(d3[0], d3[1], d3[2]);
    float ce2 = ce * ce, ce3 = ce2 * ce;
    out.mass = ce3 * density;
    out.moments = (0.5f * ce2 * density) * h2;
    out.moi = ((1.0f / 3.0f) * ce2 * density) * (v3(h3.y, h3.x, h3.x) + v3(h3.z, h3.z, h3.y));
    out.poi = (0.25f * ce * density) * cmul(h2, v3(h2.y, h2.z, h2.x));
}

// out32[10] = mass, moments xyz, moments of inertia xyz, products of inertia (xy, yz, zx)
void inertia_moments_f32(const VoxelObject& obj, const float* dens, float out32[10]) {
    float mass = 0.0f;
    V3 mo{0, 0, 0}, mi{0, 0, 0}, pi{0, 0, 0};
    for (int i = obj.occ_chunk[0][0]; i < obj.occ_chunk[0][1]; ++i)
        for (int j = obj.occ_chunk[1][0]; j < obj.occ_chunk[1][1]; ++j)
            for (int k = obj.occ_chunk[2][0]; k < obj.occ_chunk[2][1]; ++k) {
                const Chunk& c = obj.chunks[obj.cidx(i, j, k)];
                int ci[3] = {i, j, k};
                Moments32 m;
                if (c.kind == K_NONUNIFORM) chunk_moments_non_uniform(obj.extent, &obj.voxels[(size_t)c.data_offset << 12], dens, ci, m);
                else if (c.kind == K_UNIFORM) chunk_moments_uniform(obj.extent, dens, c.uniform_voxel.type, ci, m);
                else continue;
                mass += m.mass;
                mo = mo + m.moments;
                mi = mi + m.moi;
                pi = pi + m.poi;
            }
    float r[10] = {mass, mo.x, mo.y, mo.z, mi.x, mi.y, mi.z, pi.x, pi.y, pi.z};
    std::memcpy(out32, r, sizeof(r));
}

// all-cores variant: per-chunk moments in parallel, summed in the sequential function's chunk order (bit-identical f32 result)
void inertia_moments_f32_parallel(const VoxelObject& obj, const float* dens, float out32[10], int threads) {
    const int n = obj.n_chunks();
    std::vector<Moments32> per(n);
    std::vector<uint8_t> has(n, 0);
#pragma omp parallel for schedule(dynamic, 32) num_threads(threads)
    for (int c = 0; c < n; ++c) {
        const Chunk& ch = obj.chunks[c];
        int ci[3] = {c / (obj.cc[2] * obj.cc[1]), (c / obj.cc[2]) % obj.cc[1], c % obj.cc[2]};
        if (ch.kind == K_NONUNIFORM) chunk_moments_non_uniform(obj.extent, &obj.voxels[(size_t)ch.data_offset << 12], dens, ci, per[c]);
        else if (ch.kind == K_UNIFORM) chunk_moments_uniform(obj.extent, dens, ch.uniform_voxel.type, ci, per[c]);
        else continue;
        has[c] = 1;
    }
    float mass = 0.0f;
    V3 mo{0, 0, 0}, mi{0, 0, 0}, pi{0, 0, 0};
    for (int i = obj.occ_chunk[0][0]; i < obj.occ_chunk[0][1]; ++i)
        for (int j = obj.occ_chunk[1][0]; j < obj.occ_chunk[1][1]; ++j)
            for (int k = obj.occ_chunk[2][0]; k < obj.occ_chunk[2][1]; ++k) {
                const int c = obj.cidx(i, j, k);
                if (!has[c]) continue;
                mass += per[c].mass;
                mo = mo + per[c].moments;
                mi = mi + per[c].moi;
                pi = pi + per[c].poi;
            }
    float r[10] = {mass, mo.x, mo.y, mo.z, mi.x, mi.y, mi.z, pi.x, pi.y, pi.z};
    std::memcpy(out32, r, sizeof(r));
}

// Exact cube integrals in f64: for voxel (I,J,K), x in [I e,(I+1) e]:
//   xh^2-xl^2 = e^2 (2I+1),  xh^3-xl^3 = e^3 (3I^2+3I+1)
void inertia_moments_f64(const VoxelObject& obj, const float* dens, double out[10]) {
    double e = (double)obj.extent;
    double s[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int ci = 0; ci < obj.cc[0]; ++ci)
        for (int cj = 0; cj < obj.cc[1]; ++cj)
            for (int ck = 0; ck < obj.cc[2]; ++ck) {
                const Chunk& c = obj.chunks[obj.cidx(ci, cj, ck)];
                if (c.kind == K_VOID) continue;
                for (int a = 0; a < CHUNK; ++a)
                    for (int b = 0; b < CHUNK; ++b)
                        for (int cc = 0; cc < CHUNK; ++cc) {
                            Voxel v = c.kind == K_UNIFORM ? c.uniform_voxel
                                                          : obj.voxels[((size_t)c.data_offset << 12) + ((a << 8) | (b << 4) | cc)];
                            if (v.empty()) continue;
                            double d = (double)dens[v.type];
                            double I = ci * CHUNK + a, J = cj * CHUNK + b, K = ck * CHUNK + cc;
                            double qx = 2 * I + 1, qy = 2 * J + 1, qz = 2 * K + 1;
                            double cx = 3 * I * I + 3 * I + 1, cy = 3 * J * J + 3 * J + 1, cz = 3 * K * K + 3 * K + 1;
                            s[0] += d;
                            s[1] += d * qx;
                            s[2] += d * qy;
                            s[3] += d * qz;
                            s[4] += d * (cy + cz);
                            s[5] += d * (cx + cz);
                            s[6] += d * (cx + cy);
                            s[7] += d * qx * qy;
                            s[8] += d * qy * qz;
                            s[9] += d * qz * qx;
                        }
            }
    double e2 = e * e, e3 = e2 * e, e4 = e2 * e2, e5 = e4 * e;
    out[0] = s[0] * e3;
    for (int i = 1; i <= 3; ++i) out[i] = s[i] * 0.5 * e4;
    for (int i = 4; i <= 6; ++i) out[i] = s[i] * (1.0 / 3.0) * e5;
    for (int i = 7; i <= 9; ++i) out[i] = s[i] * 0.25 * e5;
}

// object/inertia.rs:288-326 — out: mass, com[3], inertia matrix (column-major 9), inverse (9)
void derive_inertial_properties(const float m[10], float out[22]) {
    float mass = m[0];
    float inv_mass = 1.0f / mass;
    V3 com = v3(m[1], m[2], m[3]) * inv_mass;
    M3 J{{m[4], -m[7], -m[9]}, {-m[7], m[5], -m[8]}, {-m[9], -m[8], m[6]}};
    // compute_delta_to_com_inertia_matrix (impact_physics/src/inertia.rs:511-546)
    V3 sq = cmul(com, com);
    V3 moi_d = (-mass) * (v3(sq.y, sq.z, sq.x) + v3(sq.z, sq.x, sq.y));
    V3 poi_d = (-mass) * cmul(com, v3(com.y, com.z, com.x));
    V3 np = -poi_d;
    M3 delta{{moi_d.x, np.x, np.z}, {np.x, moi_d.y, np.y}, {np.z, np.y, moi_d.z}};
    M3 Jc = J + delta;
    M3 scaled = Jc * inv_mass;
    M3 inv_scaled = inverse(scaled);
    M3 Jinv = inv_scaled * inv_mass;
    out[0] = mass;
    out[1] = com.x;
    out[2] = com.y;
    out[3] = com.z;
    const V3 cs[6] = {Jc.c0, Jc.c1, Jc.c2, Jinv.c0, Jinv.c1, Jinv.c2};
    for (int c = 0; c < 6; ++c) {
        out[4 + 3 * c + 0] = cs[c].x;
        out[4 + 3 * c + 1] = cs[c].y;
        out[4 + 3 * c + 2] = cs[c].z;
    }
}

// ---------------------------------------------------------------------------------------------
// Connected regions (brute force, split_detection.rs:498-562). `labels` is a dense (nx,ny,nz)
// x-major array; empty voxels get 0xFFFFFFFF, others the canonical label = rank of the component
// by its first voxel in (i,j,k) scan order. Returns the number of components.
static uint32_t uf_find(std::vector<uint32_t>& p, uint32_t i) {
    uint32_t r = i;
    while (p[r] != r) r = p[r];
    while (p[i] != r) {
        uint32_t n = p[i];
        p[i] = r;
        i = n;
    }
    return r;
}

uint32_t canonical_region_labels(const VoxelObject& obj, uint32_t* labels) {
    int nx = obj.cc[0] * CHUNK, ny = obj.cc[1] * CHUNK, nz = obj.cc[2] * CHUNK;
    size_t n = (size_t)nx * ny * nz;
    std::vector<uint8_t> occ(n);
    for (int i = 0; i < nx; ++i)
        for (int j = 0; j < ny; ++j)
            for (int k = 0; k < nz; ++k) occ[((size_t)i * ny + j) * nz + k] = obj.voxel_at(i, j, k).empty() ? 0 : 1;
    std::vector<uint32_t> parent(n);
    std::iota(parent.begin(), parent.end(), 0u);
    auto L = [&](int i, int j, int k) { return (uint32_t)(((size_t)i * ny + j) * nz + k); };
    auto join = [&](uint32_t a, uint32_t b) {
        uint32_t ra = uf_find(parent, a), rb = uf_find(parent, b);
        if (ra != rb) parent[rb] = ra;
    };
    for (int i = 0; i < nx; ++i)
        for (int j = 0; j < ny; ++j)
            for (int k = 0; k < nz; ++k) {
                uint32_t a = L(i, j, k);
                if (!occ[a]) continue;
                if (i < nx - 1 && occ[L(i + 1, j, k)]) join(a, L(i + 1, j, k));
                if (j < ny - 1 && occ[L(i, j + 1, k)]) join(a, L(i, j + 1, k));
                if (k < nz - 1 && occ[L(i, j, k + 1)]) join(a, L(i, j, k + 1));
            }
    std::vector<uint32_t> root_label(n, 0xFFFFFFFFu);
    uint32_t count = 0;
    for (size_t a = 0; a < n; ++a) {
        if (!occ[a]) {
            if (labels) labels[a] = 0xFFFFFFFFu;
            continue;
        }
        uint32_t r = uf_find(parent, (uint32_t)a);
        if (root_label[r] == 0xFFFFFFFFu) root_label[r] = count++;
        if (labels) labels[a] = root_label[r];
    }
    return count;
}

}  // namespace orc
