// C ABI of the rigid-body / contact-solver part (include/impact_voxel_hip.h, "a15-a19") and its host-side
// bookkeeping: everything that is inherently sequential and tiny per item in the reference —
//   * interlock analysis of a manifold and its separating contact     constraint/contact.rs:610-780
//   * ConstraintCache order + warm-start source of every contact      constraint/solver.rs:386-452
//   * the dependency schedule of the sweeps (see physics.hip)
// — runs here once per contact set; all arithmetic on body state runs in the kernels of physics.hip.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include "physics_internal.hpp"

namespace {

template <class T>
int grow(T** p, size_t* cap, size_t need, hipStream_t s) {
    if (need <= *cap) return IVX_OK;
    size_t ncap = std::max(need, *cap * 2);
    IVX_HIP_CHECK(ivx_stream_sync(s));
    if (*p) (void)hipFree(*p);
    *p = nullptr;
    *cap = 0;
    IVX_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(p), ncap * sizeof(T)));
    *cap = ncap;
    return IVX_OK;
}

struct H3 {
    float x, y, z;
};
inline H3 h3(const float* p) { return {p[0], p[1], p[2]}; }
inline H3 operator+(H3 a, H3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline H3 operator-(H3 a, H3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline H3 operator-(H3 a) { return {-a.x, -a.y, -a.z}; }
inline H3 operator*(H3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline float hdot(H3 a, H3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline H3 hcross(H3 a, H3 b) { return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }

// impact_math/src/random/splitmix.rs:4-15
inline uint64_t splitmix(uint64_t state) {
    state += 0x9E3779B97F4A7C15ull;
    uint64_t z = state;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
inline uint64_t splitmix2(uint64_t a, uint64_t b) { return splitmix(a ^ splitmix(b)); }

// objects_in_contact_are_interlocked (contact.rs:610-636)
bool manifold_interlocked(const ivx_contact* c, size_t n) {
    float abs_sum = 0.0f;
    H3 vec{0, 0, 0};
    for (size_t i = 0; i < n; ++i) {
        if (c[i].depth <= 0.0f) continue;
        abs_sum += c[i].depth;
        vec = vec + h3(c[i].normal) * c[i].depth;
    }
    if (abs_sum < 1e-6f) return false;
    return hdot(vec, vec) / (abs_sum * abs_sum) < 0.1f;
}

template <class F>
H3 max_displacement(const ivx_contact* c, size_t n, F map) {  // contact.rs:691-712
    float best = -INFINITY;
    size_t bi = n, bj = n;
    for (size_t i = 0; i + 1 < n; ++i)
        for (size_t j = i + 1; j < n; ++j) {
            const H3 d = map(h3(c[i].position)) - map(h3(c[j].position));
            const float s = hdot(d, d);
            if (s > best) {
                best = s;
                bi = i;
                bj = j;
            }
        }
    if (bi == n) return {0, 0, 0};
    return map(h3(c[bj].position)) - map(h3(c[bi].position));
}
bool unit_if_above(H3 v, float min_norm, H3& out) {
    const float n2 = hdot(v, v);
    if (!(n2 > min_norm * min_norm)) return false;
    const float n = std::sqrt(n2);
    out = {v.x / n, v.y / n, v.z / n};
    return true;
}
// create_contact_separating_along_axis (contact.rs:714-780)
bool separate_along(H3 com_a_minus_b, const ivx_contact* c, size_t n, H3 axis, ivx_contact& out) {
    if (hdot(axis, com_a_minus_b) < 0.0f) axis = -axis;
    float lo = INFINITY, hi = -INFINITY;
    size_t i0 = n, i1 = n;
    for (size_t i = 0; i < n; ++i) {
        const float d = hdot(h3(c[i].position), axis);
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
    out.normal[0] = axis.x;
    out.normal[1] = axis.y;
    out.normal[2] = axis.z;
    out.depth = hi - lo;
    out.restitution = 0.0f;
    out.static_friction = INFINITY;
    out.dynamic_friction = INFINITY;
    out.id = splitmix2(c[i0].id, c[i1].id);
    return true;
}
// create_separating_contact_for_interlocked_objects (contact.rs:638-689)
bool separating_contact(H3 com_a_minus_b, const ivx_contact* c, size_t n, ivx_contact& out) {
    if (n == 0) return false;
    H3 major, middle, minor;
    if (!unit_if_above(max_displacement(c, n, [](H3 p) { return p; }), 1e-6f, major)) return false;
    const H3 md = max_displacement(c, n, [&](H3 p) { return p - major * hdot(p, major); });
    if (!unit_if_above(md, 1e-6f, middle)) return separate_along(com_a_minus_b, c, n, major, out);
    if (!unit_if_above(hcross(major, middle), 1e-4f, minor)) return separate_along(com_a_minus_b, c, n, major, out);
    if (separate_along(com_a_minus_b, c, n, minor, out)) return true;
    return separate_along(com_a_minus_b, c, n, middle, out);
}

int fetch_body_position(ivx_world* w, uint32_t ref, float out[3]) {
    IVX_HIP_CHECK(ivx_stream_sync(w->ctx->stream));
    const void* src = (ref & IVX_KINEMATIC_BODY) ? static_cast<const void*>(w->kin[ref & 0x7FFFFFFFu].position)
                                                 : static_cast<const void*>(w->dyn[ref].position);
    IVX_HIP_CHECK(ivx_memcpy_sync(out, src, 12, hipMemcpyDeviceToHost));
    return IVX_OK;
}

// Chains: maximal runs (<= PHYS_CHAIN_MAX) of contacts that are consecutive in the solve order and act on the same (body_a, body_b).
void build_chains(ivx_world* w) {
    w->chain_start.clear();
    w->chain_bodies.clear();
    const uint32_t n = w->n_contacts;
    uint32_t s = 0;
    while (s < n) {
        uint32_t e = s + 1;
        while (e < n && e - s < PHYS_CHAIN_MAX && w->slot_bodies[2 * (size_t)e] == w->slot_bodies[2 * (size_t)s] &&
               w->slot_bodies[2 * (size_t)e + 1] == w->slot_bodies[2 * (size_t)s + 1])
            ++e;
        w->chain_start.push_back(s);
        w->chain_bodies.push_back(w->slot_bodies[2 * (size_t)s]);
        w->chain_bodies.push_back(w->slot_bodies[2 * (size_t)s + 1]);
        s = e;
    }
    w->chain_start.push_back(n);
}

// The chain-stationary form of a phase's schedule (physics.hip, k_solve_cs): every chain gets a pair of lanes of one wave for the whole phase
// (even lane: body A's side, odd lane: body B's). Tiles are 32 consecutive chains in the order (level of the chain's first item, solve
// order) — the chains of a level are mutually independent and tend to stay so in every sweep —, a workgroup is PHYS_CS_WAVES consecutive
// tiles. A tile's rounds are the distinct levels its items lie on, in level order, each a mask of the lanes whose next item it is: a round's
// items are of one level, so they depend on lower levels only and every wave walks its rounds in level order — the lowest unfinished level
// can always run while all workgroups are resident. `lvl`: level (from 1) of item pass * nch + chain.
#define IVX_SLAP(what)                                                                                                             \
    do {                                                                                                                            \
        static const bool tr_ = getenv("IVX_WORLD_TRACE") && atoi(getenv("IVX_WORLD_TRACE")) >= 2;                                  \
        if (tr_) {                                                                                                                  \
            const auto t1_ = std::chrono::steady_clock::now();                                                                      \
            fprintf(stderr, "[ivx world]     %s: %.1f us\n", what, 1e-3 * (double)std::chrono::duration_cast<std::chrono::nanoseconds>(t1_ - slap_t0).count()); \
            slap_t0 = t1_;                                                                                                          \
        }                                                                                                                           \
    } while (0)
// can this phase run chain-stationary at all? (its tiles fit the working workgroups, no body with more than 65535 chains)
bool stationary_feasible(ivx_world* w, uint32_t nch, uint32_t passes) {
    const uint32_t n_tiles = (nch + 31u) / 32u;
    if (nch == 0 || passes == 0 || n_tiles > PHYS_CS_WAVES * PHYS_CS_MAX_GROUPS) return false;
    std::vector<uint32_t>& deg = w->scratch_count;
    deg.assign(w->n_dyn, 0u);
    for (uint32_t ch = 0; ch < nch; ++ch)
        for (int side = 0; side < 2; ++side) {
            const uint32_t b = w->chain_bodies[2 * (size_t)ch + side];
            if (!(b & IVX_KINEMATIC_BODY)) deg[b] += 1u;
        }
    for (uint32_t b = 0; b < w->n_dyn; ++b)
        if (deg[b] > 0xFFFFu) return false;
    return true;
}
void build_stationary(ivx_world* w, int phase, uint32_t nch, uint32_t passes, const std::vector<uint32_t>& lvl) {
    auto slap_t0 = std::chrono::steady_clock::now();
    ivx_world::CsSchedule& cs = w->cs[phase];
    cs = ivx_world::CsSchedule();
    cs.slot_offset = (uint32_t)w->cs_item_host.size();
    cs.round_start_offset = (uint32_t)w->cs_round_start_host.size();
    cs.round_offset = (uint32_t)w->cs_round_mask_host.size();
    const uint32_t n_tiles = (nch + 31u) / 32u;
    if (!w->cs_feasible[phase]) return;
    // degree of every dynamic body (chains that touch it) and a chain's rank among them: before sweep s of the chain the body's record has
    // been written s * degree + rank times (every sweep walks the chains in the same order)
    std::vector<uint32_t>& deg = w->scratch_count;
    deg.assign(w->n_dyn, 0u);
    std::vector<uint32_t>& rank = w->scratch_rank;
    rank.resize(2 * (size_t)nch);
    for (uint32_t ch = 0; ch < nch; ++ch)
        for (int side = 0; side < 2; ++side) {
            const uint32_t b = w->chain_bodies[2 * (size_t)ch + side];
            rank[2 * (size_t)ch + side] = (b & IVX_KINEMATIC_BODY) ? 0u : deg[b]++;
        }
    IVX_SLAP("stationary: degrees");
    // the chains by the level of their first item, solve order inside a level: a counting sort (levels are small integers)
    const uint32_t max_level = w->n_levels[phase];
    std::vector<uint32_t>& order = w->scratch_order;
    order.resize(nch);
    {
        std::vector<uint32_t>& first = w->scratch_first;
        first.assign(max_level + 2u, 0u);
        for (uint32_t ch = 0; ch < nch; ++ch) first[lvl[ch] + 1u] += 1u;
        for (uint32_t l = 1; l <= max_level + 1u; ++l) first[l] += first[l - 1u];
        for (uint32_t ch = 0; ch < nch; ++ch) order[first[lvl[ch]]++] = ch;
    }
    IVX_SLAP("stationary: order");
    const size_t slot0 = cs.slot_offset;
    w->cs_item_host.resize(slot0 + (size_t)n_tiles * 64u, 0xFFFFFFFFu);
    w->cs_bodies_host.resize(2 * (slot0 + (size_t)n_tiles * 64u), 0u);
    w->cs_vers_host.resize(slot0 + (size_t)n_tiles * 64u, 0u);
    // a tile's rounds: the lanes of every level its items lie on — a table of lane masks over the levels and a bitmap of the levels touched,
    // walked in ascending order afterwards (no sort: the levels are small integers)
    std::vector<uint64_t>& mask_of = w->scratch_mask;
    mask_of.assign(max_level + 1u, 0ull);
    std::vector<uint64_t>& bits = w->scratch_bits;
    bits.assign((max_level + 64u) / 64u, 0ull);
    w->cs_round_mask_host.reserve(w->cs_round_mask_host.size() + (size_t)n_tiles * 64u);
    w->cs_round_level_host.reserve(w->cs_round_level_host.size() + (size_t)n_tiles * 64u);
    for (uint32_t t = 0; t < n_tiles; ++t) {
        const uint32_t first = t * 32u, cnt = std::min(32u, nch - first);
        uint32_t lo_w = 0xFFFFFFFFu, hi_w = 0u;
        for (uint32_t l = 0; l < cnt; ++l) {
            const uint32_t ch = order[first + l];
            const uint32_t s0 = w->chain_start[ch], len = w->chain_start[ch + 1] - s0;
            uint32_t body[2];
            for (int side = 0; side < 2; ++side) {
                const uint32_t b = w->chain_bodies[2 * (size_t)ch + side];
                body[side] = (b & IVX_KINEMATIC_BODY) ? w->n_dyn + (b & 0x7FFFFFFFu) : b;
            }
            for (int side = 0; side < 2; ++side) {
                const size_t slot = slot0 + (size_t)t * 64u + 2u * l + side;
                const uint32_t b = w->chain_bodies[2 * (size_t)ch + side];
                w->cs_item_host[slot] = s0 | (len << 24);
                w->cs_bodies_host[2 * slot] = body[side];
                w->cs_bodies_host[2 * slot + 1] = body[side ^ 1];
                w->cs_vers_host[slot] = (b & IVX_KINEMATIC_BODY) ? 0u : (deg[b] | (rank[2 * (size_t)ch + side] << 16));
            }
            const uint64_t lanes = 3ull << (2u * l);
            for (uint32_t p = 0; p < passes; ++p) {
                const uint32_t lv = lvl[(size_t)p * nch + ch];
                mask_of[lv] |= lanes;
                bits[lv >> 6] |= 1ull << (lv & 63u);
                lo_w = std::min(lo_w, lv >> 6);
                hi_w = std::max(hi_w, lv >> 6);
            }
        }
        w->cs_round_start_host.push_back((uint32_t)(w->cs_round_mask_host.size() - cs.round_offset));
        for (uint32_t wd = lo_w; wd <= hi_w && lo_w != 0xFFFFFFFFu; ++wd) {
            uint64_t m = bits[wd];
            bits[wd] = 0ull;
            while (m) {
                const uint32_t lv = wd * 64u + (uint32_t)__builtin_ctzll(m);
                m &= m - 1ull;
                w->cs_round_mask_host.push_back(mask_of[lv]);
                w->cs_round_level_host.push_back(lv);
                mask_of[lv] = 0ull;
            }
        }
    }
    w->cs_round_start_host.push_back((uint32_t)(w->cs_round_mask_host.size() - cs.round_offset));
    IVX_SLAP("stationary: tiles");
    if (phase == 1 && !w->cs_slot_of_level.empty()) {  // ReplayView's item indices, by pair of lanes and sweep instead of by pass and chain
        w->cs_slot_of_host.assign((size_t)n_tiles * 32u * passes, 0u);
        for (uint32_t i = 0; i < nch; ++i)
            for (uint32_t p = 0; p < passes; ++p) w->cs_slot_of_host[(size_t)i * passes + p] = w->cs_slot_of_level[(size_t)p * nch + order[i]];
    }
    cs.n_tiles = n_tiles;
}

// Dependency levels of the item sequence (pass-major, chains in cache order inside a pass); items of one level touch pairwise different
// dynamic bodies. This is all ivx_world_set_contacts builds of a phase's schedule: the level of every item (`lvl_host`), the levels' extents
// (`level_start_host`) and what the launcher needs to choose a kernel (levels, widest level, whether the chain-stationary form exists). The
// two forms the kernels read — the level-ordered item list with its hand-off tags and tiles (k_solve, k_solve_mg) and the chain-stationary
// tiles and rounds (k_solve_cs) — are built and uploaded by the solve's launcher, the one it is about to use only (ivx_world_ensure_form):
// a frame whose contact set changed used to build and upload both, 1.4 of its 2.6 ms on the host at 41 472 contacts.
void build_levels(ivx_world* w, uint32_t n_first, uint32_t n_passes, int phase) {
    auto slap_t0 = std::chrono::steady_clock::now();
    const uint32_t nch = (uint32_t)w->chain_start.size() - 1u, nb = w->n_dyn;
    const uint32_t total_passes = n_first + n_passes;
    const size_t total = (size_t)total_passes * nch;
    w->phase_items[phase] = (uint32_t)total;
    w->item_offset[phase] = phase ? w->phase_items[0] : 0u;
    w->level_offset[phase] = (uint32_t)w->level_start_host.size();
    w->n_levels[phase] = 0;
    w->max_level_items[phase] = 0;
    w->cs_feasible[phase] = 0;
    std::vector<uint32_t>& lvl = w->lvl_host[phase];
    lvl.clear();
    if (phase == 1) {
        w->kin_offsets_host.assign((size_t)w->n_kin + 1, 0u);
        w->kin_list_host.clear();
        w->cs_slot_of_level.clear();
        w->n_kin_items = 0;
    }
    if (total == 0) {
        w->level_start_host.push_back(0);
        return;
    }
    std::vector<uint32_t>& last = w->scratch_last;
    lvl.resize(total);
    last.assign(nb, 0u);
    uint32_t max_level = 0;
    size_t k = 0;
    for (uint32_t pass = 0; pass < total_passes; ++pass)
        for (uint32_t ch = 0; ch < nch; ++ch, ++k) {
            const uint32_t ba = w->chain_bodies[2 * (size_t)ch], bb = w->chain_bodies[2 * (size_t)ch + 1];
            uint32_t l = 0;
            if (!(ba & IVX_KINEMATIC_BODY)) l = std::max(l, last[ba]);
            if (!(bb & IVX_KINEMATIC_BODY)) l = std::max(l, last[bb]);
            l += 1;
            if (!(ba & IVX_KINEMATIC_BODY)) last[ba] = l;
            if (!(bb & IVX_KINEMATIC_BODY)) last[bb] = l;
            lvl[k] = l;
            max_level = std::max(max_level, l);
        }
    IVX_SLAP("schedule: levels");
    // counting sort by level (stable): level l (from 1) holds the items [start[l - 1], start[l])
    const size_t ls0 = w->level_start_host.size();
    w->level_start_host.resize(ls0 + max_level + 1, 0u);
    uint32_t* start = w->level_start_host.data() + ls0;
    for (size_t i = 0; i < total; ++i) start[lvl[i]] += 1;  // start[l] = count of level l (l >= 1), start[0] = 0
    uint32_t run = 0, widest = 0;
    for (uint32_t l = 1; l <= max_level; ++l) {
        const uint32_t c = start[l];
        widest = std::max(widest, c);
        run += c;
        start[l] = run;  // end of level l = begin of level l + 1
    }
    w->n_levels[phase] = max_level;
    w->max_level_items[phase] = widest;
    w->cs_feasible[phase] = stationary_feasible(w, nch, total_passes) ? 1 : 0;
    if (phase == 1 && w->n_kin) {
        // positional phase with kinematic bodies (ReplayView): every kinematic body's chains in solve order, by the items' places in the level order
        std::vector<uint32_t> cursor(start, start + max_level);
        std::vector<std::vector<uint32_t>> kin_items(w->n_kin);
        w->cs_slot_of_level.assign(total, 0u);
        k = 0;
        for (uint32_t pass = 0; pass < total_passes; ++pass)
            for (uint32_t ch = 0; ch < nch; ++ch, ++k) {
                const uint32_t slot = cursor[lvl[k] - 1]++;
                w->cs_slot_of_level[k] = slot;
                const uint32_t ba = w->chain_bodies[2 * (size_t)ch], bb = w->chain_bodies[2 * (size_t)ch + 1];
                if (ba & IVX_KINEMATIC_BODY) kin_items[ba & 0x7FFFFFFFu].push_back(slot);
                if (bb & IVX_KINEMATIC_BODY) kin_items[bb & 0x7FFFFFFFu].push_back(slot | 0x80000000u);
            }
        w->kin_offsets_host.assign(1, 0u);
        for (const auto& v : kin_items) {
            w->kin_list_host.insert(w->kin_list_host.end(), v.begin(), v.end());
            w->kin_offsets_host.push_back((uint32_t)w->kin_list_host.size());
        }
        w->n_kin_items = (uint32_t)w->kin_list_host.size();
        if (w->n_kin_items == 0) w->cs_slot_of_level.clear();
    }
    IVX_SLAP("schedule: extents");
}

// the level-ordered item list of a phase, its hand-off tags and tiles (k_solve, k_solve_mg); appends to the host arrays (phase 0 first)
void build_items(ivx_world* w, uint32_t first_type, uint32_t n_first, uint32_t type, uint32_t n_passes, int phase) {
    auto slap_t0 = std::chrono::steady_clock::now();
    const uint32_t nch = (uint32_t)w->chain_start.size() - 1u, nb = w->n_dyn;
    const uint32_t total_passes = n_first + n_passes;
    const size_t total = (size_t)total_passes * nch;
    const std::vector<uint32_t>& lvl = w->lvl_host[phase];
    const uint32_t max_level = w->n_levels[phase];
    const size_t ls0 = w->level_offset[phase];
    w->tile_offset[phase] = (uint32_t)w->tile_first_host.size();
    w->n_tiles[phase] = 0;
    w->tile_base_host.resize(ls0 + max_level + 1, 0u);
    if (total == 0) return;
    const uint32_t* start = w->level_start_host.data() + ls0;
    const size_t it0 = w->items_host.size();
    w->items_host.resize(it0 + total);
    w->item_bodies_host.resize(2 * (it0 + total));
    w->item_tags_host.resize(4 * (it0 + total));
    std::vector<uint32_t>& seen = w->scratch_count;  // per dynamic body: items of this phase that have touched it so far
    seen.assign(nb, 0u);
    std::vector<uint32_t> cursor(start, start + max_level);
    size_t k = 0;
    for (uint32_t pass = 0; pass < total_passes; ++pass) {
        const uint32_t ty = pass < n_first ? first_type : type;
        for (uint32_t ch = 0; ch < nch; ++ch, ++k) {
            const uint32_t s0 = w->chain_start[ch], len = w->chain_start[ch + 1] - s0;
            const size_t slot = it0 + cursor[lvl[k] - 1]++;
            w->items_host[slot] = s0 | (len << 24) | (ty << 28);
            const uint32_t ba = w->chain_bodies[2 * (size_t)ch], bb = w->chain_bodies[2 * (size_t)ch + 1];
            w->item_bodies_host[2 * slot] = (ba & IVX_KINEMATIC_BODY) ? w->n_dyn + (ba & 0x7FFFFFFFu) : ba;
            w->item_bodies_host[2 * slot + 1] = (bb & IVX_KINEMATIC_BODY) ? w->n_dyn + (bb & 0x7FFFFFFFu) : bb;
            // the hand-off tags (k_solve_mg): the versions of the two bodies' records this item waits for (it leaves them one higher), and the
            // sweep tag on the accumulated impulses — written by the velocity sweeps only: sweep q finds q (0: as prepared) and leaves q + 1
            uint32_t* tg = &w->item_tags_host[4 * slot];
            tg[0] = (ba & IVX_KINEMATIC_BODY) ? 0u : seen[ba]++;
            tg[1] = (bb & IVX_KINEMATIC_BODY) ? 0u : seen[bb]++;
            const uint32_t sweep = pass < n_first ? 0u : pass - n_first;
            tg[2] = sweep;
            tg[3] = sweep + 1u;
        }
    }
    IVX_SLAP("schedule: items");
    // tiles of 64 consecutive items of a level (the packed records of the multi-workgroup solve); tile_base runs parallel to level_start
    uint32_t n_tiles = 0;
    for (uint32_t l = 0; l < max_level; ++l) {
        w->tile_base_host[ls0 + l] = n_tiles;
        for (uint32_t i = start[l]; i < start[l + 1]; i += 64u) {
            w->tile_first_host.push_back(i | ((std::min(64u, start[l + 1] - i) - 1u) << 26));
            n_tiles += 1;
        }
    }
    w->tile_base_host[ls0 + max_level] = n_tiles;
    w->n_tiles[phase] = n_tiles;
    IVX_SLAP("schedule: tiles");
}

// id -> slot of the host-side ConstraintCache: open addressing, linear probing, deletion by backward shift. (The std::unordered_map it replaces
// cost 0.7-1.0 ms per frame at 41 472 contacts of which a tenth had changed: a node allocation per insert, a pointer chase per find.)
inline uint32_t id_hash(uint64_t id) { return (uint32_t)((id * 0x9E3779B97F4A7C15ull) >> 32); }
void id_table_reset(ivx_world* w, size_t want) {
    size_t cap = 1024;
    while (cap < 2 * want) cap *= 2;
    w->id_keys.assign(cap, 0ull);
    w->id_vals.assign(cap, 0xFFFFFFFFu);
    w->id_used = 0;
    for (uint32_t s = 0; s < w->cache.size(); ++s) {
        size_t h = id_hash(w->cache[s].id) & (cap - 1);
        while (w->id_vals[h] != 0xFFFFFFFFu) h = (h + 1) & (cap - 1);
        w->id_keys[h] = w->cache[s].id;
        w->id_vals[h] = s;
        w->id_used += 1;
    }
}
inline size_t id_find_pos(const ivx_world* w, uint64_t id) {  // position of the id's entry, or of the empty entry its probe ends at
    const size_t mask = w->id_keys.size() - 1;
    size_t h = id_hash(id) & mask;
    while (w->id_vals[h] != 0xFFFFFFFFu && w->id_keys[h] != id) h = (h + 1) & mask;
    return h;
}
inline void id_erase(ivx_world* w, uint64_t id) {
    const size_t mask = w->id_keys.size() - 1;
    size_t h = id_find_pos(w, id);
    if (w->id_vals[h] == 0xFFFFFFFFu) return;
    size_t hole = h;
    for (size_t j = (h + 1) & mask; w->id_vals[j] != 0xFFFFFFFFu; j = (j + 1) & mask) {
        const size_t home = id_hash(w->id_keys[j]) & mask;
        // the entry at j may move into the hole unless its home lies (cyclically) in (hole, j]
        const bool stays = hole <= j ? (home > hole && home <= j) : (home > hole || home <= j);
        if (!stays) {
            w->id_keys[hole] = w->id_keys[j];
            w->id_vals[hole] = w->id_vals[j];
            hole = j;
        }
    }
    w->id_vals[hole] = 0xFFFFFFFFu;
    w->id_used -= 1;
}

}  // namespace

// The multi-workgroup solve's grid barrier is hand-rolled and bounded: a workgroup that never sees the others arrive gives up, sets this word
// and the phase goes on unsynchronised — the bodies of that step are wrong. Every entry point that waits on the world's stream looks at the
// word afterwards (it is host-mapped: no copy), reports the step as failed, clears the word, and keeps the world on the single-workgroup
// kernel from then on.
static int ivx_world_check_solve(ivx_world* w, const char* who) {
    if (!w->mg_err_host || w->mg_err_host[0] == 0u) return IVX_OK;
    w->mg_err_host[0] = 0u;
    w->mg_disabled = 1;
    ivx_set_error("%s: the solver's grid barrier timed out in an earlier step of this world (a workgroup of the solve was not resident): that step's "
                  "bodies are invalid; the world now solves on one workgroup", who);
    return IVX_ERR_HIP;
}

// Host arrays on their way to the device, all through ONE pinned staging block of the world and asynchronous copies on its stream (the block is
// free again when the event behind the last copy has passed).
struct StagedUploads {
    ivx_world* w;
    struct Item {
        void* dst;
        const void* src;
        size_t bytes, off;
    };
    std::vector<Item> items;
    size_t total = 0;
    explicit StagedUploads(ivx_world* world) : w(world) {}
    void add(void* dst, const void* src, size_t bytes) {
        if (!bytes) return;
        items.push_back(Item{dst, src, bytes, total});
        total += (bytes + 63) & ~(size_t)63;
    }
    int flush() {
        if (items.empty()) return IVX_OK;
        hipStream_t s = w->ctx->stream;
        if (w->stage_sched_busy) {
            IVX_HIP_CHECK(hipEventSynchronize(w->stage_sched_ev));
            w->stage_sched_busy = 0;
        }
        if (total > w->stage_sched_cap) {
            if (w->stage_sched) (void)hipHostFree(w->stage_sched);
            w->stage_sched = nullptr;
            w->stage_sched_cap = 0;
            const size_t cap = total + total / 2 + 4096;
            IVX_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&w->stage_sched), cap, hipHostMallocDefault));
            w->stage_sched_cap = cap;
        }
        if (!w->stage_sched_ev_ready) {
            IVX_HIP_CHECK(hipEventCreateWithFlags(&w->stage_sched_ev, hipEventDisableTiming));
            w->stage_sched_ev_ready = 1;
        }
        for (const Item& it : items) memcpy(w->stage_sched + it.off, it.src, it.bytes);
        for (const Item& it : items) IVX_HIP_CHECK(ivx_memcpy_async(it.dst, w->stage_sched + it.off, it.bytes, hipMemcpyHostToDevice, s));
        IVX_HIP_CHECK(ivx_event_record(w->stage_sched_ev, s));
        w->stage_sched_busy = 1;
        items.clear();
        total = 0;
        return IVX_OK;
    }
};

// The schedule form the solve is about to use — 1: the level-ordered item list, tags and tiles (k_solve, k_solve_mg), 2: the chain-stationary
// tiles and rounds (k_solve_cs) — built from the levels (build_levels) and uploaded if this contact structure has not had it yet.
int ivx_world_ensure_form(ivx_world* w, uint32_t form) {
    if (w->forms_built & form) return IVX_OK;
    hipStream_t s = w->ctx->stream;
    int rc;
    StagedUploads up(w);
    if (form == 1u) {
        w->items_host.clear();
        w->item_bodies_host.clear();
        w->item_tags_host.clear();
        w->tile_base_host.clear();
        w->tile_first_host.clear();
        build_items(w, PHYS_ITEM_WARM, 1u, PHYS_ITEM_VELOCITY, w->cfg.n_iterations, 0);
        build_items(w, PHYS_ITEM_POSITIONAL, 0u, PHYS_ITEM_POSITIONAL, w->cfg.n_positional_correction_iterations, 1);
        if ((rc = grow(&w->items, &w->item_cap, w->items_host.size(), s))) return rc;
        if ((rc = grow(&w->item_bodies, &w->item_bodies_cap, w->item_bodies_host.size(), s))) return rc;
        if ((rc = grow(&w->item_tags, &w->item_tags_cap, w->item_tags_host.size(), s))) return rc;
        if ((rc = grow(&w->level_start, &w->level_cap, w->level_start_host.size(), s))) return rc;
        if ((rc = grow(&w->tile_base, &w->tile_base_cap, w->tile_base_host.size(), s))) return rc;
        if ((rc = grow(&w->tile_first, &w->tile_first_cap, w->tile_first_host.size(), s))) return rc;
        if (!w->items_host.empty()) {
            up.add(w->items, w->items_host.data(), w->items_host.size() * 4);
            up.add(w->item_bodies, w->item_bodies_host.data(), w->item_bodies_host.size() * 4);
            up.add(w->item_tags, w->item_tags_host.data(), w->item_tags_host.size() * 4);
        }
        up.add(w->level_start, w->level_start_host.data(), w->level_start_host.size() * 4);
        up.add(w->tile_base, w->tile_base_host.data(), w->tile_base_host.size() * 4);
        up.add(w->tile_first, w->tile_first_host.data(), w->tile_first_host.size() * 4);
    } else {
        w->cs_item_host.clear();
        w->cs_bodies_host.clear();
        w->cs_vers_host.clear();
        w->cs_round_start_host.clear();
        w->cs_round_mask_host.clear();
        w->cs_round_level_host.clear();
        w->cs_slot_of_host.clear();
        const uint32_t nch = (uint32_t)w->chain_start.size() - 1u;
        build_stationary(w, 0, nch, w->cfg.n_iterations + 1u, w->lvl_host[0]);
        build_stationary(w, 1, nch, w->cfg.n_positional_correction_iterations, w->lvl_host[1]);
        if ((rc = grow(&w->cs_item, &w->cs_item_cap, w->cs_item_host.size(), s))) return rc;
        if ((rc = grow(&w->cs_bodies, &w->cs_bodies_cap, w->cs_bodies_host.size(), s))) return rc;
        if ((rc = grow(&w->cs_vers, &w->cs_vers_cap, w->cs_vers_host.size(), s))) return rc;
        if ((rc = grow(&w->cs_round_start, &w->cs_round_start_cap, w->cs_round_start_host.size(), s))) return rc;
        if ((rc = grow(&w->cs_round_mask, &w->cs_round_mask_cap, w->cs_round_mask_host.size(), s))) return rc;
        if ((rc = grow(&w->cs_round_level, &w->cs_round_level_cap, w->cs_round_level_host.size(), s))) return rc;
        if (w->n_kin_items && (rc = grow(&w->cs_slot_of, &w->cs_slot_of_cap, w->cs_slot_of_host.size(), s))) return rc;
        if (!w->cs_item_host.empty()) {
            up.add(w->cs_item, w->cs_item_host.data(), w->cs_item_host.size() * 4);
            up.add(w->cs_bodies, w->cs_bodies_host.data(), w->cs_bodies_host.size() * 4);
            up.add(w->cs_vers, w->cs_vers_host.data(), w->cs_vers_host.size() * 4);
            up.add(w->cs_round_start, w->cs_round_start_host.data(), w->cs_round_start_host.size() * 4);
            up.add(w->cs_round_mask, w->cs_round_mask_host.data(), w->cs_round_mask_host.size() * 8);
            up.add(w->cs_round_level, w->cs_round_level_host.data(), w->cs_round_level_host.size() * 4);
        }
        if (w->n_kin_items && !w->cs_slot_of_host.empty()) up.add(w->cs_slot_of, w->cs_slot_of_host.data(), w->cs_slot_of_host.size() * 4);
    }
    if ((rc = up.flush())) return rc;
    w->forms_built |= form;
    return IVX_OK;
}

extern "C" {

int ivx_world_create(ivx_ctx* c, const ivx_solver_config* cfg, ivx_world** out) {
    IVX_REQUIRE(c && out, IVX_ERR_INVALID, "ivx_world_create: null argument");
    *out = nullptr;
    ivx_world* w = new (std::nothrow) ivx_world();
    IVX_REQUIRE(w, IVX_ERR_CAPACITY, "ivx_world_create: out of host memory");
    w->ctx = c;
    if (cfg) w->cfg = *cfg;
    else w->cfg = ivx_solver_config{8u, 0.4f, 3u, 0.2f};  // ConstraintSolverConfig::default (solver.rs:374-384)
    IVX_REQUIRE(w->cfg.n_iterations + w->cfg.n_positional_correction_iterations < 4096, IVX_ERR_INVALID, "ivx_world_create: too many iterations");
    if (hipMalloc(reinterpret_cast<void**>(&w->barrier_words), 128 * sizeof(uint32_t)) != hipSuccess ||
        ivx_memset_async(w->barrier_words, 0, 128 * sizeof(uint32_t), c->stream) != hipSuccess) {
        ivx_set_error("ivx_world_create: device allocation failed");
        delete w;
        return IVX_ERR_HIP;
    }
    if (hipHostMalloc(reinterpret_cast<void**>(&w->mg_err_host), 64, hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer(reinterpret_cast<void**>(&w->mg_err_dev), w->mg_err_host, 0) != hipSuccess) {
        ivx_set_error("ivx_world_create: host-mapped allocation failed");
        if (w->mg_err_host) (void)hipHostFree(w->mg_err_host);
        (void)hipFree(w->barrier_words);
        delete w;
        return IVX_ERR_HIP;
    }
    w->mg_err_host[0] = 0u;
    w->mg_disabled = 0;
    *out = w;
    return IVX_OK;
}

void ivx_world_destroy(ivx_world* w) {
    if (!w) return;
    (void)ivx_stream_sync(w->ctx->stream);
    if (w->side_stream) {  // the positional phase's stream and its fork / join events (created on the first multi-workgroup solve)
        (void)ivx_stream_sync(w->side_stream);
        (void)hipEventDestroy(w->ev_fork);
        (void)hipEventDestroy(w->ev_join);
        (void)hipStreamDestroy(w->side_stream);
    }
    if (w->stage_sched) (void)hipHostFree(w->stage_sched);  // pinned staging of the general path's uploads
    if (w->stage_sched_ev_ready) (void)hipEventDestroy(w->stage_sched_ev);
    if (w->stage_contacts) (void)hipHostFree(w->stage_contacts);  // pinned staging of the set_contacts fast path
    if (w->stage_ev_ready) (void)hipEventDestroy(w->stage_ev);
    void* ptrs[] = {w->dyn, w->kin, w->cb, w->touched, w->contacts, w->prev_slot, w->pc[0], w->pc[1], w->acc[0], w->acc[1], w->items, w->item_bodies, w->item_tags, w->level_start,
                    w->dynst, w->barrier_words, w->joint_refs, w->tile_base, w->tile_first, w->packed[0], w->packed[1], w->kin_offsets, w->kin_list,
                    w->kin_applied, w->kin_qstart, w->kin_snap, w->cs_item, w->cs_bodies, w->cs_vers, w->cs_round_start, w->cs_round_mask, w->cs_round_level, w->cs_slot_of};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    if (w->mg_err_host) (void)hipHostFree(w->mg_err_host);
    if (w->ev_ready)
        for (int i = 0; i < 5; ++i) (void)hipEventDestroy(w->ev[i]);
    delete w;
}

// ConstraintManager::add_spherical_joint + the joints' share of prepare_constraints (constraint.rs:183-190, 252-255). The reference's
// PreparedSphericalJoint is a placeholder: `compute_impulses` returns 0, `apply_impulses_to_body_pair` and
// `apply_positional_correction_to_body_pair` are empty (constraint/spherical_joint.rs:62-88), so a joint adds no item to the sweeps.
// What it does do is make its two bodies constrained bodies of the step (`prepare_spherical_joint` -> `add_body_pair`): their velocities
// are synchronised before the solve and written back after it — momentum = mass x (momentum / mass) — like those of bodies in contact.
int ivx_world_set_spherical_joints(ivx_world* w, const uint32_t* body_pairs, size_t n_joints) {
    IVX_REQUIRE(w && (body_pairs || n_joints == 0), IVX_ERR_INVALID, "ivx_world_set_spherical_joints: null argument");
    IVX_REQUIRE(n_joints < (1u << 24), IVX_ERR_CAPACITY, "ivx_world_set_spherical_joints: too many joints");
    for (size_t i = 0; i < 2 * n_joints; ++i) {
        const uint32_t r = body_pairs[i];
        IVX_REQUIRE((r & 0x7FFFFFFFu) < ((r & IVX_KINEMATIC_BODY) ? w->n_kin : w->n_dyn), IVX_ERR_INVALID,
                    "ivx_world_set_spherical_joints: anchor %zu refers to a missing body", i);
    }
    IVX_HIP_CHECK(ivx_stream_sync(w->ctx->stream));
    if (w->joint_refs) (void)hipFree(w->joint_refs);
    w->joint_refs = nullptr;
    w->n_joint_refs = 0;
    w->joint_refs_host.clear();
    w->n_bodies_stat_valid = 0;
    if (n_joints) {
        IVX_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&w->joint_refs), 2 * n_joints * sizeof(uint32_t)));
        IVX_HIP_CHECK(ivx_memcpy_sync(w->joint_refs, body_pairs, 2 * n_joints * sizeof(uint32_t), hipMemcpyHostToDevice));
        w->n_joint_refs = (uint32_t)(2 * n_joints);
        w->joint_refs_host.assign(body_pairs, body_pairs + 2 * n_joints);
    }
    return IVX_OK;
}

int ivx_world_set_bodies(ivx_world* w, const ivx_rigid_body* dyn, size_t n_dyn, const ivx_kinematic_body* kin, size_t n_kin) {
    IVX_REQUIRE(w && (dyn || n_dyn == 0) && (kin || n_kin == 0), IVX_ERR_INVALID, "ivx_world_set_bodies: null argument");
    IVX_REQUIRE(n_dyn < IVX_KINEMATIC_BODY && n_kin < IVX_KINEMATIC_BODY, IVX_ERR_CAPACITY, "ivx_world_set_bodies: too many bodies");
    for (size_t i = 0; i < n_dyn; ++i) IVX_REQUIRE(dyn[i].mass > 0.0f, IVX_ERR_INVALID, "ivx_world_set_bodies: body %zu has non-positive mass", i);
    hipStream_t s = w->ctx->stream;
    IVX_HIP_CHECK(ivx_stream_sync(s));
    const size_t need = n_dyn + n_kin;
    if (need > w->body_cap) {
        void* ptrs[] = {w->dyn, w->kin, w->cb, w->touched, w->dynst};
        for (void* p : ptrs)
            if (p) (void)hipFree(p);
        w->dyn = nullptr, w->kin = nullptr, w->cb = nullptr, w->touched = nullptr, w->dynst = nullptr;
        w->body_cap = 0;
        const size_t cap = std::max<size_t>(need, 64);
        IVX_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&w->dyn), cap * sizeof(ivx_rigid_body)));
        IVX_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&w->kin), cap * sizeof(ivx_kinematic_body)));
        IVX_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&w->cb), cap * sizeof(PhysBody)));
        IVX_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&w->touched), cap));
        IVX_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&w->dynst), 2 * cap * 64));  // (the shared records of the velocity phase: 32 bytes per body; then, cap * 64 bytes in, of the positional phase: 48)
        w->body_cap = cap;
    }
    if (n_dyn) IVX_HIP_CHECK(ivx_memcpy_sync(w->dyn, dyn, n_dyn * sizeof(ivx_rigid_body), hipMemcpyHostToDevice));
    if (n_kin) IVX_HIP_CHECK(ivx_memcpy_sync(w->kin, kin, n_kin * sizeof(ivx_kinematic_body), hipMemcpyHostToDevice));
    const bool resized = w->n_dyn != n_dyn || w->n_kin != n_kin;
    if ((n_dyn < w->n_dyn || n_kin < w->n_kin) && w->n_joint_refs) {
        // joints name bodies by index and were validated against the old counts: a set that shrank drops them (the caller sets them again)
        (void)hipFree(w->joint_refs);
        w->joint_refs = nullptr;
        w->n_joint_refs = 0;
        w->joint_refs_host.clear();
    }
    w->n_dyn = (uint32_t)n_dyn;
    w->n_kin = (uint32_t)n_kin;
    w->n_bodies_stat_valid = 0;
    if (resized) w->schedule_valid = 0;
    w->prepared_fresh = 0;
    return IVX_OK;
}

int ivx_world_get_bodies(ivx_world* w, ivx_rigid_body* dyn, ivx_kinematic_body* kin) {
    IVX_REQUIRE(w, IVX_ERR_INVALID, "ivx_world_get_bodies: null world");
    IVX_HIP_CHECK(ivx_stream_sync(w->ctx->stream));
    if (int rc = ivx_world_check_solve(w, "ivx_world_get_bodies")) return rc;
    if (dyn && w->n_dyn) IVX_HIP_CHECK(ivx_memcpy_sync(dyn, w->dyn, w->n_dyn * sizeof(ivx_rigid_body), hipMemcpyDeviceToHost));
    if (kin && w->n_kin) IVX_HIP_CHECK(ivx_memcpy_sync(kin, w->kin, w->n_kin * sizeof(ivx_kinematic_body), hipMemcpyDeviceToHost));
    return IVX_OK;
}

int ivx_world_set_contacts(ivx_world* w, const ivx_contact* contacts, size_t n, size_t* n_prepared) {
    IVX_REQUIRE(w && (contacts || n == 0), IVX_ERR_INVALID, "ivx_world_set_contacts: null argument");
    if (int rc = ivx_world_check_solve(w, "ivx_world_set_contacts")) return rc;  // (no wait here: a flag that is already up)
    IVX_REQUIRE(n < (1u << 24), IVX_ERR_CAPACITY, "ivx_world_set_contacts: more than 2^24 contacts");
    // 0. The usual frame: the contacts of the frame before with new geometry — the same ids in the same order on the same body pairs, no manifold
    // interlocked. Then everything the steps below would compute is what they computed last time: every id keeps its slot (warm start from the
    // same slot: prev_slot = identity), the chains and the dependency schedule stand. One pass over the input decides that and copies it into
    // a pinned staging buffer; the upload is an asynchronous copy and nothing here waits for the GPU (46 080 contacts: 0.1 ms instead of 0.45).
    if (w->schedule_valid && n > 0 && n == w->n_contacts && n == w->cache.size() && 2 * n == w->slot_bodies.size() && w->stage_contacts_cap >= n) {
        if (w->stage_busy) {  // the previous frame's copy out of the staging buffer
            IVX_HIP_CHECK(hipEventSynchronize(w->stage_ev));
            w->stage_busy = 0;
        }
        bool same = true;
        size_t i = 0;
        while (i < n && same) {
            size_t j = i + 1;
            while (j < n && !(contacts[j].flags & IVX_CONTACT_MANIFOLD_START)) ++j;
            if (manifold_interlocked(contacts + i, j - i)) same = false;
            for (size_t k = i; k < j && same; ++k) {
                const ivx_contact& c = contacts[k];
                // (the resident body references were validated when they came in)
                same = c.id == w->cache[k].id && c.body_a == w->slot_bodies[2 * k] && c.body_b == w->slot_bodies[2 * k + 1];
                w->stage_contacts[k] = c;
            }
            i = j;
        }
        if (same) {
            hipStream_t s = w->ctx->stream;
            w->n_prev = w->n_contacts;
            IVX_HIP_CHECK(ivx_memcpy_async(w->contacts, w->stage_contacts, n * sizeof(ivx_contact), hipMemcpyHostToDevice, s));
            IVX_HIP_CHECK(ivx_event_record(w->stage_ev, s));
            w->stage_busy = 1;
            int rc;
            w->cur ^= 1;
            if ((rc = ivx_launch_phys_prepare_bodies(w))) return rc;
            if ((rc = ivx_launch_phys_prepare_contacts(w, nullptr))) return rc;  // (null: every contact's previous slot is its own)
            if ((rc = ivx_launch_phys_mark_joint_bodies(w))) return rc;
            w->prepared_fresh = 1;
            if (n_prepared) *n_prepared = n;
            return IVX_OK;
        }
    }
    static const bool trace_laps = getenv("IVX_WORLD_TRACE") != nullptr;
    auto lap_t0 = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!trace_laps) return;
        const auto t1 = std::chrono::steady_clock::now();
        fprintf(stderr, "[ivx world]   %s: %.1f us\n", what, 1e-3 * (double)std::chrono::duration_cast<std::chrono::nanoseconds>(t1 - lap_t0).count());
        lap_t0 = t1;
    };
    {  // developer aid (IVX_WORLD_TRACE=1): a frame that leaves the one-pass path says why
        static const bool trace = getenv("IVX_WORLD_TRACE") != nullptr;
        if (trace)
            fprintf(stderr, "[ivx world] set_contacts: general path (schedule_valid %d, n %zu, resident %u, cache %zu, staging cap %zu)\n", w->schedule_valid, n, w->n_contacts,
                    w->cache.size(), w->stage_contacts_cap);
    }
    int rc;
    // 1. one pass over the input: the body references, and is any manifold interlocked (constraint.rs:237-249)? Only then is a copy of the
    // list made, with those manifolds replaced by their separating contact (needs the bodies' positions from the device: rare, and slow)
    bool any_interlocked = false;
    {
        size_t i = 0;
        while (i < n) {
            size_t j = i;
            do {
                const ivx_contact& c = contacts[j];
                const uint32_t la = (c.body_a & IVX_KINEMATIC_BODY) ? w->n_kin : w->n_dyn, lb = (c.body_b & IVX_KINEMATIC_BODY) ? w->n_kin : w->n_dyn;
                IVX_REQUIRE((c.body_a & 0x7FFFFFFFu) < la && (c.body_b & 0x7FFFFFFFu) < lb, IVX_ERR_INVALID, "ivx_world_set_contacts: contact %zu refers to a missing body", j);
                IVX_REQUIRE(c.body_a != c.body_b, IVX_ERR_INVALID, "ivx_world_set_contacts: contact %zu joins a body with itself", j);
                ++j;
            } while (j < n && !(contacts[j].flags & IVX_CONTACT_MANIFOLD_START));
            if (manifold_interlocked(contacts + i, j - i)) any_interlocked = true;
            i = j;
        }
    }
    const ivx_contact* eff = contacts;
    size_t n_eff = n;
    if (any_interlocked) {
        w->effective.clear();
        size_t i = 0;
        while (i < n) {
            size_t j = i + 1;
            while (j < n && !(contacts[j].flags & IVX_CONTACT_MANIFOLD_START)) ++j;
            const ivx_contact* m = contacts + i;
            const size_t cnt = j - i;
            bool replaced = false;
            if (manifold_interlocked(m, cnt)) {
                float pa[3], pb[3];
                if ((rc = fetch_body_position(w, m[0].body_a, pa))) return rc;
                if ((rc = fetch_body_position(w, m[0].body_b, pb))) return rc;
                ivx_contact sep;
                if (separating_contact(h3(pa) - h3(pb), m, cnt, sep)) {
                    sep.body_a = m[0].body_a;
                    sep.body_b = m[0].body_b;
                    w->effective.push_back(sep);
                    replaced = true;
                }
            }
            if (!replaced) w->effective.insert(w->effective.end(), m, m + cnt);
            i = j;
        }
        eff = w->effective.data();
        n_eff = w->effective.size();
    }
    lap("validate + interlock");
    // 2. ConstraintCache::register_prepared_constraint + remove_unprepared_constraints (solver.rs:406-452): every id keeps its slot, new ids
    // are appended, ids that did not come are swap-removed in slot order. An id is looked for at the slot behind the one its predecessor in
    // the input had first (a run of contacts that stayed together costs a comparison each), then in the id table.
    if (w->id_keys.empty() || 2 * (w->cache.size() + n_eff) > w->id_keys.size()) id_table_reset(w, w->cache.size() + n_eff);
    {
        const size_t old_len = w->cache.size();
        size_t cursor = 0;
        w->cache.reserve(old_len + n_eff);
        for (uint32_t e = 0; e < n_eff; ++e) {
            const uint64_t id = eff[e].id;
            if (cursor < old_len && w->cache[cursor].id == id) {
                w->cache[cursor].src = e;
                w->cache[cursor].prepared = true;
                cursor += 1;
                continue;
            }
            const size_t h = id_find_pos(w, id);
            if (w->id_vals[h] != 0xFFFFFFFFu) {
                ivx_world::Entry& en = w->cache[w->id_vals[h]];
                en.src = e;
                en.prepared = true;
                cursor = (size_t)w->id_vals[h] + 1;
            } else {
                w->id_keys[h] = id;
                w->id_vals[h] = (uint32_t)w->cache.size();
                w->id_used += 1;
                w->cache.push_back(ivx_world::Entry{id, -1, e, true});
            }
        }
        size_t idx = 0, len = w->cache.size();
        while (idx < len) {
            if (w->cache[idx].prepared) {
                ++idx;
            } else {
                id_erase(w, w->cache[idx].id);
                w->cache[idx] = w->cache[len - 1];
                w->cache.pop_back();
                --len;
                if (idx < len) w->id_vals[id_find_pos(w, w->cache[idx].id)] = (uint32_t)idx;
            }
        }
    }
    const uint32_t nc = (uint32_t)w->cache.size();
    w->n_bodies_stat_valid = 0;
    // the contacts in slot order go straight into the pinned block their upload is made from (one copy of each contact on the host)
    if (w->stage_busy) {  // the previous frame's copy out of the staging block
        IVX_HIP_CHECK(hipEventSynchronize(w->stage_ev));
        w->stage_busy = 0;
    }
    if (nc > w->stage_contacts_cap) {
        if (w->stage_contacts) (void)hipHostFree(w->stage_contacts);
        w->stage_contacts = nullptr;
        w->stage_contacts_cap = 0;
        const size_t cap2 = (size_t)nc + nc / 4 + 64;
        IVX_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&w->stage_contacts), cap2 * sizeof(ivx_contact), hipHostMallocDefault));
        w->stage_contacts_cap = cap2;
    }
    if (!w->stage_ev_ready) {
        IVX_HIP_CHECK(hipEventCreateWithFlags(&w->stage_ev, hipEventDisableTiming));
        w->stage_ev_ready = 1;
    }
    w->prev_slot_host.resize(nc);
    w->slot_bodies.resize(2 * (size_t)nc);
    for (uint32_t sl = 0; sl < nc; ++sl) {
        ivx_world::Entry& en = w->cache[sl];
        const ivx_contact& c = eff[en.src];
        w->stage_contacts[sl] = c;
        w->slot_bodies[2 * (size_t)sl] = c.body_a;
        w->slot_bodies[2 * (size_t)sl + 1] = c.body_b;
        w->prev_slot_host[sl] = en.prev_slot;
        en.prev_slot = (int32_t)sl;
        en.prepared = false;
    }
    w->n_prev = w->n_contacts;  // size of the state arrays written by the previous prepare
    w->n_contacts = nc;
    lap("constraint cache");
    // 3. chains and the levels of the two phases' items: (warm pass + velocity sweeps) and (positional sweeps)
    build_chains(w);
    IVX_REQUIRE((uint64_t)(w->chain_start.size() - 1u) * (std::max(w->cfg.n_iterations + 1u, w->cfg.n_positional_correction_iterations)) < (1ull << 26), IVX_ERR_CAPACITY,
                "ivx_world_set_contacts: more than 2^26 schedule items in one phase");
    // the schedule depends on the chains and their body pairs only: an unchanged contact structure keeps last frame's (host and device copies)
    const bool same_schedule = w->schedule_valid && w->chain_start == w->prev_chain_start && w->chain_bodies == w->prev_chain_bodies;
    if (!same_schedule) {
        w->level_start_host.clear();
        build_levels(w, 1u, w->cfg.n_iterations, 0);
        build_levels(w, 0u, w->cfg.n_positional_correction_iterations, 1);
        w->forms_built = 0;
        w->prev_chain_start = w->chain_start;
        w->prev_chain_bodies = w->chain_bodies;
    }
    lap("chains + levels");
    // 4. upload
    hipStream_t s = w->ctx->stream;
    size_t cap = w->contact_cap;
    if (nc > cap) {
        IVX_HIP_CHECK(ivx_stream_sync(s));
        const size_t ncap = std::max<size_t>(nc, std::max<size_t>(cap * 2, 256));
        // the previous state (pc/acc of the last solve) must survive the growth
        ivx_contact* nc_buf = nullptr;
        int32_t* np = nullptr;
        PhysContact* npc[2] = {nullptr, nullptr};
        float* nacc[2] = {nullptr, nullptr};
        IVX_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&nc_buf), ncap * sizeof(ivx_contact)));
        IVX_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&np), ncap * sizeof(int32_t)));
        for (int b = 0; b < 2; ++b) {
            IVX_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&npc[b]), ncap * sizeof(PhysContact)));
            IVX_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&nacc[b]), ncap * 4 * sizeof(float)));
            if (w->pc[b] && w->n_prev) {
                IVX_HIP_CHECK(ivx_memcpy_sync(npc[b], w->pc[b], w->n_prev * sizeof(PhysContact), hipMemcpyDeviceToDevice));
                IVX_HIP_CHECK(ivx_memcpy_sync(nacc[b], w->acc[b], w->n_prev * 4 * sizeof(float), hipMemcpyDeviceToDevice));
            }
            if (w->pc[b]) (void)hipFree(w->pc[b]);
            if (w->acc[b]) (void)hipFree(w->acc[b]);
            w->pc[b] = npc[b];
            w->acc[b] = nacc[b];
        }
        if (w->contacts) (void)hipFree(w->contacts);
        if (w->prev_slot) (void)hipFree(w->prev_slot);
        w->contacts = nc_buf;
        w->prev_slot = np;
        w->contact_cap = ncap;
    }
    if (w->n_kin_items) {
        const size_t n_pos_items = w->phase_items[1];
        if ((rc = grow(&w->kin_offsets, &w->kin_offsets_cap, w->kin_offsets_host.size(), s))) return rc;
        if ((rc = grow(&w->kin_list, &w->kin_list_cap, w->kin_list_host.size(), s))) return rc;
        if ((rc = grow(&w->kin_applied, &w->kin_applied_cap, n_pos_items, s))) return rc;
        if ((rc = grow(&w->kin_qstart, &w->kin_qstart_cap, 8 * n_pos_items, s))) return rc;  // float4 per (item, side)
        if ((rc = grow(&w->kin_snap, &w->kin_snap_cap, (size_t)std::max<uint32_t>(w->n_dyn, 1u) * 8, s))) return rc;
    }
    // (no wait here: the copies below are stream-ordered behind whatever still reads these buffers; a buffer that grew was waited for in grow())
    if (nc) {
        IVX_HIP_CHECK(ivx_memcpy_async(w->contacts, w->stage_contacts, (size_t)nc * sizeof(ivx_contact), hipMemcpyHostToDevice, s));
        IVX_HIP_CHECK(ivx_event_record(w->stage_ev, s));
        w->stage_busy = 1;
    }
    StagedUploads up(w);
    if (nc) up.add(w->prev_slot, w->prev_slot_host.data(), nc * sizeof(int32_t));
    if (!same_schedule && w->n_kin_items) {
        up.add(w->kin_offsets, w->kin_offsets_host.data(), w->kin_offsets_host.size() * 4);
        up.add(w->kin_list, w->kin_list_host.data(), w->kin_list_host.size() * 4);
    }
    lap("buffers");
    if ((rc = up.flush())) return rc;
    lap("staged uploads");
    w->schedule_valid = 1;
    // 5. device part of prepare_constraints: gather bodies, prepare every contact, warm-start bookkeeping
    w->cur ^= 1;
    if ((rc = ivx_launch_phys_prepare_bodies(w))) return rc;
    if ((rc = ivx_launch_phys_prepare_contacts(w, w->prev_slot))) return rc;
    if ((rc = ivx_launch_phys_mark_joint_bodies(w))) return rc;
    lap("prepare launches");
    w->prepared_fresh = 1;
    if (n_prepared) *n_prepared = nc;
    return IVX_OK;
}

// prepare_constraints again over the resident contact set: same ids, same order (every id is "known from
// the previous solve"), warm impulses from the last solve
int ivx_world_prepare(ivx_world* w) {
    IVX_REQUIRE(w, IVX_ERR_INVALID, "ivx_world_prepare: null world");
    IVX_REQUIRE(w->schedule_valid || w->n_contacts == 0, IVX_ERR_STATE, "ivx_world_prepare: call ivx_world_set_contacts after changing the bodies");
    int rc;
    w->cur ^= 1;
    w->n_prev = w->n_contacts;
    if ((rc = ivx_launch_phys_prepare_bodies(w))) return rc;
    if ((rc = ivx_launch_phys_prepare_contacts(w, nullptr))) return rc;
    if ((rc = ivx_launch_phys_mark_joint_bodies(w))) return rc;
    w->prepared_fresh = 1;
    return IVX_OK;
}

int ivx_world_advance_momenta(ivx_world* w, float dt) {
    IVX_REQUIRE(w, IVX_ERR_INVALID, "ivx_world_advance_momenta: null world");
    return ivx_launch_phys_pre_solve(w, dt);
}

int ivx_world_solve(ivx_world* w) {
    IVX_REQUIRE(w, IVX_ERR_INVALID, "ivx_world_solve: null world");
    IVX_REQUIRE(w->prepared_fresh, IVX_ERR_STATE, "ivx_world_solve: no prepared constraints (ivx_world_set_contacts / ivx_world_prepare)");
    int rc;
    if ((rc = ivx_launch_phys_solve(w))) return rc;
    if ((rc = ivx_launch_phys_post_solve(w, 0.0f, 1, 0))) return rc;
    w->prepared_fresh = 0;
    return IVX_OK;
}

int ivx_world_advance_configurations(ivx_world* w, float dt) {
    IVX_REQUIRE(w, IVX_ERR_INVALID, "ivx_world_advance_configurations: null world");
    return ivx_launch_phys_post_solve(w, dt, 0, 1);
}

// `timed`: bracket the phases with event records for stage_ms. Only the synchronous ivx_world_step asks for them: an event record
// is a packet of its own on the queue (~2-3 us each), and a frame that enqueues this step between other work has no use for them.
static int world_step_enqueue(ivx_world* w, float dt, bool timed) {
    hipStream_t s = w->ctx->stream;
    if (timed && !w->ev_ready) {
        for (int i = 0; i < 5; ++i) IVX_HIP_CHECK(hipEventCreate(&w->ev[i]));
        w->ev_ready = 1;
    }
#define EV(i)                                              \
    do {                                                   \
        if (timed) IVX_HIP_CHECK(ivx_event_record(w->ev[i], s)); \
    } while (0)
    int rc;
    if (w->n_contacts == 0 && !w->prepared_fresh && w->n_joint_refs == 0) {
        // no constraints this step: prepare, advance momenta and advance configurations are element-wise per body — one launch
        w->cur ^= 1;
        w->n_prev = 0;
        for (int i = 0; i < 4; ++i) EV(i);
        if ((rc = ivx_launch_phys_free_step(w, dt))) return rc;
        EV(4);
        return IVX_OK;
    }
    EV(0);
    if (!w->prepared_fresh)
        if ((rc = ivx_world_prepare(w))) return rc;
    EV(1);
    if ((rc = ivx_launch_phys_pre_solve(w, dt))) return rc;
    EV(2);
    if ((rc = ivx_launch_phys_solve(w))) return rc;
    EV(3);
    if ((rc = ivx_launch_phys_post_solve(w, dt, 1, 1))) return rc;
    EV(4);
#undef EV
    w->prepared_fresh = 0;
    return IVX_OK;
}

int ivx_world_step_enqueue(ivx_world* w, float dt) {
    IVX_REQUIRE(w, IVX_ERR_INVALID, "ivx_world_step_enqueue: null world");
    if (int rc = ivx_world_check_solve(w, "ivx_world_step_enqueue")) return rc;  // (an earlier step's flag, if it is up by now)
    return world_step_enqueue(w, dt, false);
}

int ivx_world_step(ivx_world* w, float dt, ivx_physics_result* out) {
    IVX_REQUIRE(w, IVX_ERR_INVALID, "ivx_world_step: null world");
    int rc = world_step_enqueue(w, dt, out != nullptr);
    if (rc) return rc;
    IVX_HIP_CHECK(ivx_stream_sync(w->ctx->stream));
    if ((rc = ivx_world_check_solve(w, "ivx_world_step"))) return rc;
    if (out) {
        memset(out, 0, sizeof(*out));
        out->n_contacts = w->n_contacts;
        out->n_levels[0] = w->n_levels[0];
        out->n_levels[1] = w->n_levels[1];
        if (!w->n_bodies_stat_valid) {
            // the step's constrained bodies: the dynamic bodies the device marks (k_prepare_contacts: both bodies of every contact; k_mark_bodies:
            // the joints' dynamic anchors) and the kinematic bodies of the contacts — counted from the host's copies of the same lists, once per
            // contact list (a read-back of the marks and a walk over 46 080 contacts every step were 60 us of a 1 ms step)
            std::vector<uint8_t> t((size_t)w->n_dyn + w->n_kin, 0);
            for (size_t sl = 0; sl + 1 < w->slot_bodies.size(); sl += 2) {
                const uint32_t a = w->slot_bodies[sl], b = w->slot_bodies[sl + 1];
                if (a & IVX_KINEMATIC_BODY) t[w->n_dyn + (a & 0x7FFFFFFFu)] = 1;
                else if (a < w->n_dyn) t[a] = 1;
                if (b & IVX_KINEMATIC_BODY) t[w->n_dyn + (b & 0x7FFFFFFFu)] = 1;
                else if (b < w->n_dyn) t[b] = 1;
            }
            for (uint32_t r : w->joint_refs_host)
                if (!(r & IVX_KINEMATIC_BODY) && r < w->n_dyn) t[r] = 1;
            uint32_t nb = 0;
            for (uint8_t x : t) nb += x;
            w->n_bodies_stat = nb;
            w->n_bodies_stat_valid = 1;
        }
        out->n_bodies = w->n_bodies_stat;
        for (int i = 0; i < 4; ++i)
            if (hipEventElapsedTime(&out->stage_ms[i], w->ev[i], w->ev[i + 1]) != hipSuccess) out->stage_ms[i] = 0.0f;
        if (hipEventElapsedTime(&out->stage_ms[4], w->ev[0], w->ev[4]) != hipSuccess) out->stage_ms[4] = 0.0f;
    }
    return IVX_OK;
}

int ivx_world_set_solver_groups(ivx_world* w, uint32_t groups) {
    IVX_REQUIRE(w && (groups <= 16u || groups == PHYS_SOLVER_STATIONARY), IVX_ERR_INVALID,
                "ivx_world_set_solver_groups: at most 16 workgroups (or 255: the chain-stationary solve)");
    w->solver_groups_forced = groups;
    return IVX_OK;
}

int ivx_world_solver_info(ivx_world* w, uint32_t out[8]) {
    IVX_REQUIRE(w && out, IVX_ERR_INVALID, "ivx_world_solver_info: null argument");
    out[0] = w->solver_groups_used;
    out[1] = w->n_levels[0];
    out[2] = w->n_levels[1];
    out[3] = w->max_level_items[0];
    out[4] = w->max_level_items[1];
    out[5] = (uint32_t)w->chain_start.size() > 0 ? (uint32_t)w->chain_start.size() - 1u : 0u;
    out[6] = w->n_contacts;
    out[7] = w->solver_kind_used;
    return IVX_OK;
}

int ivx_world_contact_state(ivx_world* w, uint64_t* ids, float* impulses3, size_t cap, size_t* n_out) {
    IVX_REQUIRE(w && n_out, IVX_ERR_INVALID, "ivx_world_contact_state: null argument");
    *n_out = w->n_contacts;
    IVX_REQUIRE(w->n_contacts <= cap, IVX_ERR_CAPACITY, "ivx_world_contact_state: %u contacts exceed capacity %zu", w->n_contacts, cap);
    if (ids)
        for (uint32_t s = 0; s < w->n_contacts; ++s) ids[s] = w->cache[s].id;
    if (impulses3 && w->n_contacts) {
        IVX_HIP_CHECK(ivx_stream_sync(w->ctx->stream));
        if (int rc = ivx_world_check_solve(w, "ivx_world_contact_state")) return rc;
        std::vector<float> a((size_t)w->n_contacts * 4);
        IVX_HIP_CHECK(ivx_memcpy_sync(a.data(), w->acc[w->cur], a.size() * 4, hipMemcpyDeviceToHost));
        for (uint32_t s = 0; s < w->n_contacts; ++s)
            for (int q = 0; q < 3; ++q) impulses3[3 * (size_t)s + q] = a[4 * (size_t)s + q];
    }
    return IVX_OK;
}

}  // extern "C"
