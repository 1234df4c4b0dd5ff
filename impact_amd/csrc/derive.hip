// a4 — derived voxel/chunk state in one sweep: six-neighbour adjacency flags, face distributions,
// chunk obscuredness, uniform-chunk demotion. One workgroup per chunk, one thread per (i,j) row.
//
// The reference computes this sequentially and statefully (engine/crates/impact_voxel/src/object.rs):
//   update_internal_adjacencies               object.rs:2673-2756
//   update_mutual_face_adjacencies            object.rs:2077-2528
//   convert_to_non_uniform_if_uniform         object.rs:2530-2550
//   face distributions / obscuredness         object.rs:2889-3040
// Its fixed point is a pure function of voxel emptiness, which is what this kernel evaluates:
//   * non-empty voxel: HAS_ADJACENT_<dir> = neighbour voxel (possibly in the adjacent chunk) is non-empty;
//   * empty voxel: IS_EMPTY only — except on a chunk face whose own distribution is Mixed while the
//     adjoining face of the neighbour chunk is Full, where `add_all_outward_adjacencies_for_face`
//     (object.rs:2552-2600) sets the outward flag on every voxel of the face, empty ones included;
//   * face distribution = Empty/Full/Mixed by the number of non-empty voxels on the face;
//   * IS_OBSCURED_<dir> = adjoining face of the neighbour chunk is Full;
//   * a chunk generated Uniform stays Uniform iff all six adjoining neighbour faces are Full.
// The oracle replays the reference's sequential procedure; tests require bit-equal flags.
//
// Traffic: reads 1 B/voxel (sdf sign) + six neighbour faces, writes 1 B/voxel (flags). Emptiness of the
// 16 voxels of a row is a 16-bit mask; the 18x18 halo of masks sits in LDS.
#include "chunk_passes.hpp"
#include "table_roles.hpp"
#include "many.hpp"

// Waves per SIMD the derive sweep is compiled for, i.e. its register budget: 7 -> 72 VGPRs (13 spilled per lane), 6 -> 80 (4 spilled), 5 -> 96
// (none). Seven was round 3's choice for the residency (LDS allows seven workgroups per CU) — but every spilled dword is a store to memory
// (stores leave the L2 on this part), per lane and chunk: with all of the sweep's own store groups switched off the headline launch still wrote
// 52 of its 91 MB (tools/derive_budget.sh). Measured (tools/derive_waves.sh, profiles/round5): headline 39.5 us / 103.7 MB written at seven,
// 37.6 us / 45.6 MB at six, 38.3 us / 32.8 MB at five; all-surface 188 us / 378 MB, 181 us / 302 MB, 187 us / 270 MB. Six it is.
#ifndef IVX_DERIVE_WAVES
#define IVX_DERIVE_WAVES 6
#endif

namespace {

struct DeriveParams {
    uint32_t cx, cy, cz;
};

__device__ __forceinline__ uint32_t row_mask(uint4 s) {
    // bit k set <=> voxel k non-empty <=> sd < 0 (sign bit of byte k)
    // (the four sign bits of a word, moved to bits 0, 8, 16, 24, meet in bits 21..24 of a product with 2^21 + 2^14 + 2^7 + 1: byte i's bit
    // lands on 8 i + 21 - 7 j, the wanted ones (i = j) side by side, the other twelve on distinct bits elsewhere, so nothing carries)
    uint32_t w[4] = {s.x, s.y, s.z, s.w};
    uint32_t m = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) m |= ((((w[q] >> 7) & 0x01010101u) * 0x00204081u >> 21) & 0xFu) << (4 * q);
    return m;
}

// non-empty mask of one 16-voxel row of chunk `c` (row offset `off` inside the chunk). Chunks generated Void / Uniform
// are known from their record; their planes are not read (and need not hold data: compact planes).
// (the row is fetched whatever the record says — the plane exists, only its content may be stale — so that the two loads
// overlap instead of forming a dependent chain; these kernels are bound by such chains, not by bandwidth)
__device__ __forceinline__ uint32_t nbr_row_mask(const GridView& g, const ivx_chunk_info* info, size_t c, uint32_t off) {
    const uint32_t gen = info[c].gen_kind;
    const uint32_t rm = row_mask(*reinterpret_cast<const uint4*>(g.sdf + c * IVX_CHUNK_VOXELS + off));
    return gen == KIND_NONUNIFORM ? rm : (gen == KIND_UNIFORM ? 0xFFFFu : 0u);
}

// One THREAD per chunk: settles every chunk whose derived state follows from the chunk records alone and lists the others
// (the "active" chunks) for the workgroup-per-chunk kernels. A Void chunk has no voxels; a chunk generated Uniform whose six
// neighbours were generated Uniform too is solid all round and stays Uniform. Neither has planes (compact planes), so all of
// their per-step state is the record, the occupied sub-box, one region and empty mesh counts.
struct ChunkPreArgs {
    GridView g;
    ivx_chunk_info* info;
    uint32_t* bbox;
    uint32_t* mesh_counts;
    uint8_t* chunk_class;
    uint8_t* touch;
    uint32_t* rparent;
    uint32_t* work_counts;
    uint32_t* next_work_count;
    uint32_t* active_list;
    ivx_roles::PresetArgs preset;
    // (slab protocol, the sweep split around the arrival of the ghost layers) 1: a Uniform chunk of a face plane that only the ghost layer's
    // record can settle is left PENDING (class 2: neither settled nor listed) for k_face_settle, which runs behind the wait
    uint32_t defer_ghost, pad_;
};
// the record of a chunk that needs no sweep: Void, or Uniform among Uniform neighbours
__device__ __forceinline__ void chunk_settle(const GridView& g, const ChunkPreArgs& a, uint32_t chunk, const ivx_chunk_info& own) {
    const uint32_t gen = own.gen_kind;
    const bool solid = gen == KIND_UNIFORM;
    ivx_chunk_info rec = own;
    rec.kind = (uint8_t)gen;
    rec.flags = 0;
    rec.face_dist = solid ? 0x555 : 0;
    // a chunk demoted by an earlier pass has its planes written out and its record's type cleared
    rec.uniform_type = solid ? (own.kind == KIND_NONUNIFORM ? g.type[(size_t)chunk * IVX_CHUNK_VOXELS] : own.uniform_type) : (uint8_t)0;
    rec.region_count = solid ? 1 : 0;
    rec.boundary_region_count = solid ? 1 : 0;
    a.info[chunk] = rec;
    a.bbox[chunk] = solid ? (0x80000000u | (15u << 4) | (15u << 12) | (15u << 20)) : 0u;
    a.mesh_counts[2 * chunk] = 0;
    a.mesh_counts[2 * chunk + 1] = 0;
    a.touch[chunk] = solid ? 7 : 0;  // a settled solid chunk touches its three upper neighbours (all solid)
    if (solid) a.rparent[(size_t)chunk * 256] = chunk * 256u;  // its one region: its own node until the merge pass links it
}
__device__ __forceinline__ void chunk_pre_body(const ChunkPreArgs& a, uint32_t bid, uint32_t) {
    const GridView& g = a.g;
    ivx_chunk_info* __restrict__ info = a.info;
    uint32_t* __restrict__ bbox = a.bbox;
    uint32_t* __restrict__ mesh_counts = a.mesh_counts;
    uint8_t* __restrict__ chunk_class = a.chunk_class;
    uint8_t* __restrict__ touch = a.touch;
    uint32_t* __restrict__ rparent = a.rparent;
    uint32_t* __restrict__ work_counts = a.work_counts;
    uint32_t* __restrict__ next_work_count = a.next_work_count;
    uint32_t* __restrict__ active_list = a.active_list;
    __shared__ uint32_t s_w[4];
    __shared__ uint32_t s_base;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    // this launch is the first of the derive stages: it also presets the scratch words the later stages of the call start from,
    // and zeroes the counter the NEXT sweep will append under (the two counters alternate)
    ivx_roles::role_preset(a.preset, bid * 256u + tid);
    if (bid == 0 && tid == 0) next_work_count[0] = 0u;
    const uint32_t n_chunks = g.cx * g.cy * g.cz;
    const uint32_t chunk = bid * 256u + tid;
    const bool live = chunk < n_chunks;
    bool settled = false, pending = false;
    if (live) {
        const ivx_chunk_info own = info[chunk];
        const uint32_t gen = own.gen_kind;
        settled = gen == KIND_VOID;
        if (gen == KIND_UNIFORM) {
            const int ck = chunk % g.cz, cj = (chunk / g.cz) % g.cy, ci = chunk / (g.cz * g.cy);
            if (cj > 0 && ck > 0 && cj + 1 < (int)g.cy && ck + 1 < (int)g.cz) {
                const uint32_t sx = g.cy * g.cz, sy = g.cz;
                // across a slab face the neighbour is the ghost layer's chunk record (a slab is a few chunk planes thick: without
                // this a quarter of a solid body's chunks would go the long way); at the end of the grid there is none
                const bool ghost_lo = ci == 0 && g.ghost_info[0] != nullptr, ghost_hi = ci + 1 == (int)g.cx && g.ghost_info[1] != nullptr;
                const bool defer = a.defer_ghost != 0u && (ghost_lo || ghost_hi);
                const bool x_lo = ci > 0 ? info[chunk - sx].gen_kind == KIND_UNIFORM : (ghost_lo && (defer || g.ghost_info[0][cj * g.cz + ck].gen_kind == KIND_UNIFORM));
                const bool x_hi =
                    ci + 1 < (int)g.cx ? info[chunk + sx].gen_kind == KIND_UNIFORM : (ghost_hi && (defer || g.ghost_info[1][cj * g.cz + ck].gen_kind == KIND_UNIFORM));
                settled = x_lo && x_hi && info[chunk - sy].gen_kind == KIND_UNIFORM && info[chunk + sy].gen_kind == KIND_UNIFORM &&
                          info[chunk - 1].gen_kind == KIND_UNIFORM && info[chunk + 1].gen_kind == KIND_UNIFORM;
                if (settled && defer) settled = false, pending = true;  // (all but the ghost layer's word: k_face_settle asks for that behind the wait)
            }
        }
        if (settled) chunk_settle(g, a, chunk, own);
        chunk_class[chunk] = pending ? 2 : (settled ? 1 : 0);
    }
    // ordered append of this block's active chunks (one atomic per block; the list stays nearly sorted, so neighbouring
    // list entries are neighbouring chunks)
    const bool active = live && !settled && !pending;
    const unsigned long long bal = __ballot(active);
    if (lane == 0) s_w[wave] = (uint32_t)__popcll(bal);
    __syncthreads();
    if (tid == 0) s_base = atomicAdd(&work_counts[0], (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]));
    __syncthreads();
    if (active) {
        uint32_t off = s_base + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
        for (uint32_t w = 0; w < wave; ++w) off += s_w[w];
        active_list[off] = chunk;
    }
}
__global__ __launch_bounds__(256) void k_chunk_pre(ChunkPreArgs a) { chunk_pre_body(a, blockIdx.x, gridDim.x); }
IVX_MANY_TWIN(k_chunk_pre_many, ChunkPreArgs, chunk_pre_body, __launch_bounds__(256))
IVX_MANY_LAUNCHER(many_chunk_pre, k_chunk_pre_many, ChunkPreArgs, 256)

// (slab protocol) the chunks k_chunk_pre left pending, once the ghost layers are in: one thread per chunk of the two face planes. A pending chunk
// whose ghost neighbour is Uniform is settled as k_chunk_pre would have settled it; any other joins the end of the active list — ahead of the
// sweep's second part, which reads the list's length when it starts.
__global__ __launch_bounds__(256) void k_face_settle(ChunkPreArgs a) {
    const GridView& g = a.g;
    const uint32_t cols = g.cy * g.cz, t = blockIdx.x * 256u + threadIdx.x;
    if (t >= 2u * cols) return;
    const uint32_t side = t / cols, col = t % cols;
    if (side == 1u && g.cx == 1u) return;  // (one plane: both its faces are thread `col` of side 0)
    const uint32_t chunk = (side ? g.cx - 1u : 0u) * cols + col;
    if (a.chunk_class[chunk] != 2) return;
    const ivx_chunk_info own = a.info[chunk];
    bool uniform = true;
    if (chunk < cols && g.ghost_info[0]) uniform = uniform && g.ghost_info[0][col].gen_kind == KIND_UNIFORM;
    if (chunk >= (g.cx - 1u) * cols && g.ghost_info[1]) uniform = uniform && g.ghost_info[1][col].gen_kind == KIND_UNIFORM;
    if (uniform) {
        chunk_settle(g, a, chunk, own);
        a.chunk_class[chunk] = 1;
    } else {
        a.chunk_class[chunk] = 0;
        a.active_list[atomicAdd(&a.work_counts[0], 1u)] = chunk;
    }
}

// The global loads of a chunk's first phase, in flight: the chunk's own record and rows; across each z face one byte per thread; (threads
// 0..63) one 16-byte row across an x or y face; the three neighbours' generated kinds. Addresses are clamped to valid ones and the
// records select afterwards — written as "record, then the byte if the neighbour is dense" the compiler sinks each load under the branch
// that consumes it, and the phase is seven dependent memory round trips instead of one (measured on the mesher's tile load). Byte loads
// come as the aligned word's load and a shift at the use: the `load + extend` pair of a byte load is one the compiler keeps together,
// i.e. a wait where the load was issued.
struct DeriveLoads {
    uint2 own_rec;
    uint4 own_sd, own_types, nrow;
    uint32_t gen_lo, gen_hi, by_lo, by_hi, ngen, type0;
};
// SIGNS: the sampler left the sign rows of every chunk that has planes (ivx_grid::signs_current): the chunk's own row and its neighbours' face
// rows are 2 bytes each in `g.signs` — a row's bit 0 / bit 15 are its voxels on the k faces — and no voxel plane is read (`own_sd.x`,
// `by_lo`, `by_hi`, `nrow.x` then hold sign rows; a ghost layer's face rows still come from its planes).
template <bool SIGNS>
__device__ __forceinline__ void derive_issue(const GridView& g, const ivx_chunk_info* __restrict__ info, uint32_t chunk, uint32_t tid, DeriveLoads& L) {
    const int ck = chunk % g.cz, cj = (chunk / g.cz) % g.cy, ci = chunk / (g.cz * g.cy);
    const size_t base = (size_t)chunk * IVX_CHUNK_VOXELS;
    const bool has_zlo = ck > 0, has_zhi = ck + 1 < (int)g.cz;
    const size_t c_lo = has_zlo ? (size_t)chunk - 1 : (size_t)chunk, c_hi = has_zhi ? (size_t)chunk + 1 : (size_t)chunk;
    const int nf = (tid >> 4) & 3, nr = tid & 15;
    bool n_present, n_ghost = false;
    size_t nc;
    uint32_t noff;
    {
        const size_t plane = (size_t)g.cy * g.cz;
        const int8_t* gp = nf == 0 ? g.ghost_sdf[0] : g.ghost_sdf[1];
        if (nf == 0) n_present = ci > 0, nc = (size_t)chunk - plane, noff = 15 * 256 + nr * 16, n_ghost = !n_present && gp != nullptr;
        else if (nf == 1) n_present = ci + 1 < (int)g.cx, nc = (size_t)chunk + plane, noff = nr * 16, n_ghost = !n_present && gp != nullptr;
        else if (nf == 2) n_present = cj > 0, nc = (size_t)chunk - g.cz, noff = nr * 256 + 15 * 16;
        else n_present = cj + 1 < (int)g.cy, nc = (size_t)chunk + g.cz, noff = nr * 256;
        if (!n_present) nc = chunk;
    }
    const int8_t* nrp = g.sdf + (nc << 12) + noff;
    if (n_ghost) nrp = (nf == 0 ? g.ghost_sdf[0] : g.ghost_sdf[1]) + ((size_t)(cj * g.cz + ck) * 256 + nr * 16);
    L.own_rec = reinterpret_cast<const uint2*>(info)[chunk];
    // (gen_kind is the second byte of a record's first word; the face bytes are the last / first of the row's 16)
    L.gen_lo = reinterpret_cast<const uint32_t*>(info)[2 * c_lo];
    L.gen_hi = reinterpret_cast<const uint32_t*>(info)[2 * c_hi];
    L.ngen = reinterpret_cast<const uint32_t*>(info)[2 * nc];
    L.type0 = *reinterpret_cast<const uint32_t*>(g.type + base);  // (voxel 0's type: the type of a Uniform chunk that was demoted before)
    if (SIGNS) {
        // (sign rows as the aligned words around them, the half taken at the use: see the byte loads of the other form)
        const uint32_t* sw = reinterpret_cast<const uint32_t*>(g.signs);
        L.own_sd = make_uint4(sw[((size_t)chunk * 256 + tid) >> 1], 0u, 0u, 0u);
        L.own_types = make_uint4(0u, 0u, 0u, 0u);
        L.by_lo = sw[(c_lo * 256 + tid) >> 1];
        L.by_hi = sw[(c_hi * 256 + tid) >> 1];
        if (n_ghost) L.nrow = *reinterpret_cast<const uint4*>(nrp);
        else L.nrow = make_uint4(sw[(nc * 256 + (noff >> 4)) >> 1], 0u, 0u, 0u);  // (noff >> 4: the face row's index in its chunk)
    } else {
        L.own_sd = *reinterpret_cast<const uint4*>(g.sdf + base + (size_t)tid * 16);      // used only if the chunk has planes
        L.own_types = *reinterpret_cast<const uint4*>(g.type + base + (size_t)tid * 16);  // likewise
        L.by_lo = *reinterpret_cast<const uint32_t*>(g.sdf + (c_lo << 12) + tid * 16 + 12);
        L.by_hi = *reinterpret_cast<const uint32_t*>(g.sdf + (c_hi << 12) + tid * 16);
        L.nrow = *reinterpret_cast<const uint4*>(nrp);
    }
}

// What the later per-chunk passes need is in this kernel's registers already, so it can run them in the same sweep (`parts`):
// the chunk-local region labelling needs only the chunk's own non-empty masks, and so do its moments (plus the type row).
struct DeriveFused {
    uint32_t parts;             // IVX_PART_*
    uint32_t x_off;             // slab offset in chunks (moments are about the global origin)
    uint8_t* labels;            // regions
    uint32_t* rparent;
    uint32_t* rscalar;
    uint32_t* multi_list;
    const float* dens;          // moments
    double* chunk_moments;
};

// BOX MODE (the sweep after an edit: ivx_launch_derive_box). The reference patches derived state around the chunks an edit touched
// (handle_chunk_voxels_modified + update_upper_boundary_adjacencies_for_chunks_in_ranges, object/intersection.rs:255-262, 532-598); derived state
// being a pure function of the voxels and the generated kinds, the same result comes from sweeping the box of touched chunks grown by one
// chunk each way — a few hundred chunks instead of the object's whole active list. `n` > 0: entry li of the walk is chunk li of the box
// (any kind: a Void chunk is settled on the spot), its list entry goes to `out_list`. The launch's blocks beyond `derive_blocks` reset the
// region forest of every chunk OUTSIDE the box (the global resolve that follows starts from one root per local region everywhere) and list
// the chunks with several regions among them (role_region_reset).
struct DeriveBox {
    uint32_t n, derive_blocks;
    uint32_t lo[3], cc[3];
    uint32_t* out_list;
};
__device__ __forceinline__ void role_region_reset(uint32_t bid, const GridView& g, const DeriveBox& box, const ivx_chunk_info* __restrict__ info,
                                                  uint32_t* __restrict__ rparent, uint32_t* __restrict__ rscalar, uint32_t* __restrict__ multi_list) {
    const uint32_t chunk = bid * 256u + threadIdx.x;
    if (chunk >= g.cx * g.cy * g.cz) return;
    const uint32_t ck = chunk % g.cz, cj = (chunk / g.cz) % g.cy, ci = chunk / (g.cz * g.cy);
    if (ci - box.lo[0] < box.cc[0] && cj - box.lo[1] < box.cc[1] && ck - box.lo[2] < box.cc[2]) return;  // (unsigned: inside the box)
    const uint32_t rc = info[chunk].region_count;
    for (uint32_t r = 0; r < rc; ++r) rparent[(size_t)chunk * 256 + r] = chunk * 256u + r;
    if (rc > 1u) multi_list[atomicAdd(&rscalar[2], 1u)] = chunk;  // (as ccl_local_chunk lists them: role_ccl_merge_multi joins their regions)
}

// (amdgpu_waves_per_eu(7): seven workgroups per CU is what the 20 KB of LDS allow; the workgroups are latency-bound, so residency is
// throughput, and the register budget is set to match. A software pipeline over a resident set of workgroups — the next chunk's loads in
// flight across the region and moment passes, as the mesher does — was tried: its 119 registers leave four workgroups per CU, and four
// pipelined ones were no faster than seven plain ones; what did help is the grid's size, see ivx_launch_derive.)
struct DeriveArgs {
    GridView g;
    int8_t* sdf_rw;
    uint8_t* type_rw;
    uint8_t* flags_out;
    ivx_chunk_info* info;
    uint32_t* bbox;
    uint8_t* touch;
    uint16_t* signs;
    uint8_t* kface_out;
    const uint32_t* work_counts;
    uint32_t* active_list;
    const uint32_t* list_in;
    DeriveFused fz;
    uint32_t signs_type, x_part;  // (x_part: IVX_XPART_*, the slab protocol's split around the arrival of the ghost layers)
    DeriveBox box;
};
template <bool SIGNS>
__device__ __forceinline__ void derive_body(const DeriveArgs& a_, uint32_t bid_, uint32_t nb_) {
    const GridView& g = a_.g;
    int8_t* __restrict__ sdf_rw = a_.sdf_rw;
    uint8_t* __restrict__ type_rw = a_.type_rw;
    uint8_t* __restrict__ flags_out = a_.flags_out;
    ivx_chunk_info* __restrict__ info = a_.info;
    uint32_t* __restrict__ bbox = a_.bbox;
    uint8_t* __restrict__ touch = a_.touch;
    uint16_t* __restrict__ signs = a_.signs;
    uint8_t* __restrict__ kface_out = a_.kface_out;
    const uint32_t* __restrict__ work_counts = a_.work_counts;
    uint32_t* __restrict__ active_list = a_.active_list;
    const uint32_t* __restrict__ list_in = a_.list_in;
    const DeriveFused& fz = a_.fz;
    const uint32_t signs_type = a_.signs_type;
    const DeriveBox& box = a_.box;
    __shared__ uint32_t occ[18][18];  // non-empty masks of rows (i+1, j+1); halo rows from neighbour chunks
    __shared__ uint32_t cnt[13];      // own face non-empty counts [0..6), neighbour face non-empty counts [6..12), [12] touch bits
    __shared__ CclShared s_ccl;
    __shared__ float s_dens[256];
    __shared__ uint16_t s_mtab[256];  // moments_table_entry of every byte value (moments_row_sums_one_type)
    // (the moment pass's 16 x 10 row totals borrow the union-find's node array: the region pass is over when they are written, and the
    // next chunk's starts behind the loop's barrier — an array of their own would be the kilobyte that takes the seventh workgroup off a CU)
    double (*s_red)[10] = reinterpret_cast<double (*)[10]>(s_ccl.par);
    const uint32_t tid = threadIdx.x;
    if (!SIGNS && box.n && bid_ >= box.derive_blocks) {  // (box mode: the blocks behind the sweep's)
        role_region_reset(bid_ - box.derive_blocks, g, box, info, fz.rparent, fz.rscalar, fz.multi_list);
        return;
    }
    if (fz.parts & IVX_PART_MOMENTS) s_dens[tid] = fz.dens[tid], s_mtab[tid] = moments_table_entry(tid);  // (the first barrier of the loop publishes them)
    const int ti = tid >> 4, tj = tid & 15;
    const bool in_box = !SIGNS && box.n != 0u;
    const uint32_t n_active = in_box ? box.n : work_counts[0];
    const uint32_t n_walk = in_box ? box.derive_blocks : nb_;
    // bounded walk over the active list (virtual block ids give each XCD a contiguous stretch of it). (`list_in` IS `active_list`, through a
    // read-only pointer so that the entry comes by a scalar load: a workgroup reads only entries it alone rewrites — later, and never the chunk
    // index in their low bits)
    for (uint32_t li = ivx_xcd_remap(bid_, n_walk); li < n_active; li += n_walk) {
    __syncthreads();  // the previous chunk's LDS use is over
    IVX_T(g, li, 0);
    uint32_t chunk;
    if (in_box) {
        const uint32_t bk = li % box.cc[2], bj = (li / box.cc[2]) % box.cc[1], bi = li / (box.cc[2] * box.cc[1]);
        chunk = ((box.lo[0] + bi) * g.cy + (box.lo[1] + bj)) * g.cz + (box.lo[2] + bk);
    } else {
        chunk = IVX_LIST_CHUNK(list_in[li]);
        if (ivx_xpart_skip(g, a_.x_part, chunk / (g.cz * g.cy))) continue;  // (the other part's chunk: workgroup-uniform)
    }
    DeriveLoads L;
    derive_issue<SIGNS>(g, info, chunk, tid, L);
    const int ck = chunk % g.cz, cj = (chunk / g.cz) % g.cy, ci = chunk / (g.cz * g.cy);
    const size_t base = (size_t)chunk * IVX_CHUNK_VOXELS;
    const bool has_zlo = ck > 0, has_zhi = ck + 1 < (int)g.cz;
    // x / y faces: f: 0 x-, 1 x+, 2 y-, 3 y+ (threads >= 64 repeat thread (tid & 63)'s loads and drop them)
    const int nf = (tid >> 4) & 3, nr = tid & 15;
    bool n_present, n_ghost = false;
    {
        const int8_t* gp = nf == 0 ? g.ghost_sdf[0] : g.ghost_sdf[1];  // (selected, not indexed: the view lives in the kernel arguments)
        if (nf == 0) n_present = ci > 0, n_ghost = !n_present && gp != nullptr;
        else if (nf == 1) n_present = ci + 1 < (int)g.cx, n_ghost = !n_present && gp != nullptr;
        else if (nf == 2) n_present = cj > 0;
        else n_present = cj + 1 < (int)g.cy;
    }
    uint2 own_rec = L.own_rec;
    uint4 own_sd = L.own_sd, own_types = L.own_types, nrow = L.nrow;
    uint32_t gen_lo = L.gen_lo, gen_hi = L.gen_hi, by_lo = L.by_lo, by_hi = L.by_hi, ngen = L.ngen, type0 = L.type0;
    asm volatile("" : "+v"(own_rec.x), "+v"(own_rec.y), "+v"(own_sd.x), "+v"(own_sd.y), "+v"(own_sd.z), "+v"(own_sd.w));
    asm volatile("" : "+v"(own_types.x), "+v"(own_types.y), "+v"(own_types.z), "+v"(own_types.w));
    asm volatile("" : "+v"(gen_lo), "+v"(gen_hi), "+v"(by_lo), "+v"(by_hi), "+v"(ngen), "+v"(nrow.x), "+v"(nrow.y), "+v"(nrow.z), "+v"(nrow.w), "+v"(type0));
    gen_lo = (gen_lo >> 8) & 0xFFu, gen_hi = (gen_hi >> 8) & 0xFFu, ngen = (ngen >> 8) & 0xFFu;
    if (SIGNS) {  // the row's half of the word it came in; the face voxel's bit where the byte's sign bit is expected below (bit 7)
        by_lo = (((by_lo >> (16u * (tid & 1u))) >> 15) & 1u) << 7;
        by_hi = ((by_hi >> (16u * (tid & 1u))) & 1u) << 7;
    } else {
        by_lo >>= 24, by_hi &= 0xFFu;
    }
    ivx_chunk_info own_info;
    own_info.kind = (uint8_t)(own_rec.x & 0xFFu);
    own_info.gen_kind = (uint8_t)((own_rec.x >> 8) & 0xFFu);
    own_info.flags = (uint8_t)((own_rec.x >> 16) & 0xFFu);
    own_info.uniform_type = (uint8_t)(own_rec.x >> 24);
    own_info.face_dist = (uint16_t)(own_rec.y & 0xFFFFu);
    own_info.region_count = (uint8_t)((own_rec.y >> 16) & 0xFFu);
    own_info.boundary_region_count = (uint8_t)(own_rec.y >> 24);
    const uint32_t own_row_mask = SIGNS ? ((own_sd.x >> (16u * (tid & 1u))) & 0xFFFFu) : row_mask(own_sd);
    const bool own_uniform = own_info.gen_kind == KIND_UNIFORM;
    if (!SIGNS && IVX_DBG_KEEP(128u)) {
        // the row's bytes on the two k faces, rows side by side, for the mesher's halo (GridView::kface)
        uint8_t* kf = kface_out + (size_t)chunk * 1024 + tid;
        kf[0] = (uint8_t)(own_sd.x & 0xFFu);
        kf[256] = (uint8_t)(own_sd.w >> 24);
        kf[512] = (uint8_t)(own_types.x & 0xFFu);
        kf[768] = (uint8_t)(own_types.w >> 24);
    }
    if (tid < 13) cnt[tid] = 0;
    // (a Void chunk reaches the sweep in box mode only: no voxels whatever its stale planes hold)
    const uint32_t m = own_uniform ? 0xFFFFu : (own_info.gen_kind == KIND_VOID ? 0u : own_row_mask);
    occ[ti + 1][tj + 1] = m;
    if (!SIGNS && IVX_DBG_KEEP(128u)) signs[(size_t)chunk * 256 + tid] = (uint16_t)m;  // for the mesher's count pass (SIGNS: the sampler's, and a demoted chunk's below)

    uint32_t zlo = 0, zhi = 0;  // neighbour voxel across the z faces for this (i,j)
    if (has_zlo) zlo = gen_lo == KIND_NONUNIFORM ? ((by_lo >> 7) & 1u) : (gen_lo == KIND_UNIFORM ? 1u : 0u);
    if (has_zhi) zhi = gen_hi == KIND_NONUNIFORM ? ((by_hi >> 7) & 1u) : (gen_hi == KIND_UNIFORM ? 1u : 0u);
    if (tid < 64) {
        // (SIGNS: the face row's sign mask itself — row index noff >> 4 = 240 + nr, nr, 16 nr + 15, 16 nr by face: odd exactly for f = 2 and for
        // odd nr on the x faces —, a ghost layer's row still from its plane)
        const uint32_t nrow_odd = nf == 2 ? 1u : (nf == 3 ? 0u : ((uint32_t)nr & 1u));
        const uint32_t rm = (SIGNS && !n_ghost) ? ((nrow.x >> (16u * nrow_odd)) & 0xFFFFu) : row_mask(nrow);
        uint32_t nm = 0;
        if (n_present) nm = ngen == KIND_NONUNIFORM ? rm : (ngen == KIND_UNIFORM ? 0xFFFFu : 0u);
        else if (n_ghost) nm = rm;
        if (nf == 0) occ[0][nr + 1] = nm;
        else if (nf == 1) occ[17][nr + 1] = nm;
        else if (nf == 2) occ[nr + 1][0] = nm;
        else occ[nr + 1][17] = nm;
    }
    __syncthreads();

    // occupied sub-box of the chunk (for update_occupied_ranges, object.rs:1149-1280): lanes 0..15 of
    // wave 0 OR the row masks along j (-> i occupancy + k bits) and along i (-> j occupancy)
    if (tid < 16) {
        uint32_t a = 0, b = 0;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            a |= occ[tid + 1][t + 1];
            b |= occ[t + 1][tid + 1];
        }
        const uint32_t bi = (uint32_t)__ballot(a != 0) & 0xFFFFu, bj = (uint32_t)__ballot(b != 0) & 0xFFFFu;
        uint32_t ku = a;
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) ku |= __shfl_xor(ku, o, 16);
        if (tid == 0) {
            uint32_t packed = 0;
            if (bi) {
                packed = 0x80000000u | (uint32_t)(__ffs(bi) - 1) | ((uint32_t)(31 - __clz(bi)) << 4) | ((uint32_t)(__ffs(bj) - 1) << 8) |
                         ((uint32_t)(31 - __clz(bj)) << 12) | ((uint32_t)(__ffs(ku) - 1) << 16) | ((uint32_t)(31 - __clz(ku)) << 20);
            }
            if (IVX_DBG_KEEP(64u)) bbox[chunk] = packed;
        }
    }

    IVX_T(g, li, 1);  // rows loaded, sub-box done
    // face populations: own faces and adjoining neighbour faces
    {
        // All counts are reduced in registers (DPP rows of 16 lanes = one i each, ballots, scalar lane reads); one lane per
        // wave then adds the wave's part to LDS.
        const uint32_t wave = tid >> 6, lane = tid & 63u;
        const uint32_t pc = __popc(m);
        const uint32_t rs = ivx_row16_sum(pc);  // lane 15 of a row: voxels of the rows with this i
        const uint32_t x_lo = (uint32_t)__builtin_amdgcn_readlane((int)rs, 15), x_hi = (uint32_t)__builtin_amdgcn_readlane((int)rs, 63);
        // y faces: j = 0 / j = 15 are lanes 0,16,32,48 / 15,31,47,63 of every wave
        const uint32_t y_lo = ((uint32_t)__builtin_amdgcn_readlane((int)pc, 0) + (uint32_t)__builtin_amdgcn_readlane((int)pc, 16)) +
                              ((uint32_t)__builtin_amdgcn_readlane((int)pc, 32) + (uint32_t)__builtin_amdgcn_readlane((int)pc, 48));
        const uint32_t y_hi = ((uint32_t)__builtin_amdgcn_readlane((int)pc, 15) + (uint32_t)__builtin_amdgcn_readlane((int)pc, 31)) +
                              ((uint32_t)__builtin_amdgcn_readlane((int)pc, 47) + (uint32_t)__builtin_amdgcn_readlane((int)pc, 63));
        const uint32_t z0 = (uint32_t)__popcll(__ballot((m & 1u) != 0)), z1 = (uint32_t)__popcll(__ballot(((m >> 15) & 1u) != 0));
        const uint32_t nz0 = (uint32_t)__popcll(__ballot(zlo != 0)), nz1 = (uint32_t)__popcll(__ballot(zhi != 0));
        // does a non-empty voxel of the +x / +y / +z face meet a non-empty voxel of the neighbour chunk? (what joins the
        // regions of two single-region chunks, k_ccl_merge_columns)
        const uint32_t tb = (__ballot(ti == 15 && (m & occ[17][tj + 1]) != 0) ? 1u : 0u) | (__ballot(tj == 15 && (m & occ[ti + 1][17]) != 0) ? 2u : 0u) |
                            (__ballot((((m >> 15) & 1u) & zhi) != 0) ? 4u : 0u);
        // neighbour faces: 16 halo rows each, summed by the first DPP row of wave 0
        uint32_t nf[4] = {0, 0, 0, 0};
        if (wave == 0) {
            const uint32_t t = lane & 15u;
            nf[0] = ivx_row16_sum(__popc(occ[0][t + 1]));
            nf[1] = ivx_row16_sum(__popc(occ[17][t + 1]));
            nf[2] = ivx_row16_sum(__popc(occ[t + 1][0]));
            nf[3] = ivx_row16_sum(__popc(occ[t + 1][17]));
        }
        if (lane == 15u && wave == 0) {
            cnt[0] = x_lo;  // i = 0 lives in wave 0, i = 15 in wave 3: single writers
            cnt[6] = nf[0];
            cnt[7] = nf[1];
            cnt[8] = nf[2];
            cnt[9] = nf[3];
        }
        if (lane == 15u && wave == 3) cnt[1] = x_hi;
        if (lane == 0) {
            atomicAdd(&cnt[2], y_lo);
            atomicAdd(&cnt[3], y_hi);
            atomicAdd(&cnt[4], z0);
            atomicAdd(&cnt[5], z1);
            atomicAdd(&cnt[10], nz0);
            atomicAdd(&cnt[11], nz1);
            if (tb) atomicOr(&cnt[12], tb);
        }
    }
    __syncthreads();

    IVX_T(g, li, 2);  // face counts done
    uint32_t own_fd[6], nbr_full = 0, own_mixed = 0;
#pragma unroll
    for (int f = 0; f < 6; ++f) {
        uint32_t c = cnt[f];
        own_fd[f] = c == 0 ? FD_EMPTY : (c == 256 ? FD_FULL : FD_MIXED);
        if (own_fd[f] == FD_MIXED) own_mixed |= 1u << f;
        if (cnt[6 + f] == 256) nbr_full |= 1u << f;
    }

    // flags for the 16 voxels of this row
    const uint32_t xdn = occ[ti][tj + 1], xup = occ[ti + 2][tj + 1];
    const uint32_t ydn = occ[ti + 1][tj], yup = occ[ti + 1][tj + 2];
    const uint32_t zdn = (m << 1) | zlo, zup = (m >> 1) | (zhi << 15);
    // outward flag for EMPTY voxels on a Mixed face whose neighbour face is Full (object.rs:2439-2454)
    const uint32_t quirk = own_mixed & nbr_full;
    // Per direction, the 16-bit mask of voxels that carry its flag: non-empty voxels whose neighbour is non-empty, plus (the
    // quirk) the empty voxels of a face. A 4-bit nibble is spread to one bit per byte with a multiply (bit i -> bit 8i:
    // the partial products i, i+7, i+14, i+21 never collide), so a flags word costs 7 spreads instead of 4 x 7 tests.
    const uint32_t e = ~m & 0xFFFFu;
    uint32_t dm[6];
    dm[0] = (m & xdn) | ((ti == 0 && (quirk & 1u)) ? e : 0u);
    dm[1] = (m & ydn) | ((tj == 0 && (quirk & 4u)) ? e : 0u);
    dm[2] = (m & zdn) | ((quirk & 16u) ? (e & 1u) : 0u);
    dm[3] = (m & xup) | ((ti == 15 && (quirk & 2u)) ? e : 0u);
    dm[4] = (m & yup) | ((tj == 15 && (quirk & 8u)) ? e : 0u);
    dm[5] = (m & zup) | ((quirk & 32u) ? (e & 0x8000u) : 0u);
    // The flags byte of voxel k is column k of an 8 x 16 bit matrix whose rows are the masks (row 0: VF_EMPTY, row 1: none, rows 2..7:
    // VF_X_DN = 1 << 2 ... VF_Z_UP = 1 << 7): two 8 x 8 transposes — low and high byte of the masks — of three rounds of masked swaps each
    // (7, 14 and 28 bits apart; the first two stay inside a 32-bit half), ~60 instructions where spreading 7 x 4 nibbles by multiplication
    // took 110.
    uint32_t w[4];
    {
        const uint32_t p01 = dm[0] | (dm[1] << 16), p23 = dm[2] | (dm[3] << 16), p45 = dm[4] | (dm[5] << 16);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            // bytes of the 64-bit matrix, row r in byte r: (e, 0, dm0, dm1 | dm2, dm3, dm4, dm5), byte h of each mask
            uint32_t lo = __builtin_amdgcn_perm(p01, e, h ? 0x07050C01u : 0x06040C00u);  // selector bytes 0-3: second operand's, 4-7: first's, 0x0C: zero
            uint32_t hi = __builtin_amdgcn_perm(p45, p23, h ? 0x07050301u : 0x06040200u);
            uint32_t t;
            t = (lo ^ (lo >> 7)) & 0x00AA00AAu, lo = lo ^ t ^ (t << 7);
            t = (hi ^ (hi >> 7)) & 0x00AA00AAu, hi = hi ^ t ^ (t << 7);
            t = (lo ^ (lo >> 14)) & 0x0000CCCCu, lo = lo ^ t ^ (t << 14);
            t = (hi ^ (hi >> 14)) & 0x0000CCCCu, hi = hi ^ t ^ (t << 14);
            t = (lo ^ ((hi << 4) | (lo >> 28))) & 0xF0F0F0F0u, lo ^= t, hi ^= t >> 4;
            w[2 * h] = lo, w[2 * h + 1] = hi;
        }
    }
    // (cnt[] and the record's gen_kind are the same for every thread, so is `kind`)
    const uint32_t gen = own_info.gen_kind;
    uint32_t kind = gen;
    if (gen == KIND_UNIFORM && nbr_full != 0x3Fu) kind = KIND_NONUNIFORM;
    // the planes of a Uniform chunk that was demoted before hold its voxels already; its record no longer has the type
    const uint32_t utype = own_uniform ? (own_info.kind == KIND_NONUNIFORM ? (type0 & 0xFFu) : (uint32_t)own_info.uniform_type) : 0u;
    if (kind == KIND_NONUNIFORM) {
        if (IVX_DBG_KEEP(1u)) *reinterpret_cast<uint4*>(flags_out + base + (size_t)tid * 16) = make_uint4(w[0], w[1], w[2], w[3]);
        if (own_uniform && own_info.kind != KIND_NONUNIFORM && IVX_DBG_KEEP(2u)) {
            // convert_to_non_uniform_if_uniform (object.rs:2530-2550): the demoted chunk gets its 4096 voxels
            const uint32_t t4 = utype * 0x01010101u;
            *reinterpret_cast<uint4*>(sdf_rw + base + (size_t)tid * 16) = make_uint4(0x80808080u, 0x80808080u, 0x80808080u, 0x80808080u);
            *reinterpret_cast<uint4*>(type_rw + base + (size_t)tid * 16) = make_uint4(t4, t4, t4, t4);
        }
        if (SIGNS && own_uniform && IVX_DBG_KEEP(4u)) {  // a chunk that is NonUniform by demotion (now or earlier): the sampler left it no sign rows or face bytes
            signs[(size_t)chunk * 256 + tid] = (uint16_t)0xFFFFu;
            uint8_t* kf = kface_out + (size_t)chunk * 1024 + tid;
            kf[0] = (uint8_t)0x80u;
            kf[256] = (uint8_t)0x80u;
            kf[512] = (uint8_t)utype;
            kf[768] = (uint8_t)utype;
        }
    }

    IVX_T(g, li, 3);  // flags written
    // ---- fused passes over the same chunk (uniform branches: `parts` and `kind` are the same for the whole workgroup)
    uint32_t rc = own_info.region_count, brc = own_info.boundary_region_count;
    if (fz.parts & IVX_PART_REGIONS) ccl_local_chunk(s_ccl, tid, chunk, kind, gen, m, fz.labels, fz.rparent, fz.rscalar, fz.multi_list, rc, brc);
    IVX_T(g, li, 4);  // regions labelled
    if ((fz.parts & IVX_PART_MOMENTS) && kind == KIND_NONUNIFORM && IVX_DBG_KEEP(16u)) {
        // (a chunk demoted in this pass has no type plane yet: its voxels all have the record's type)
        const uint32_t ut = utype * 0x01010101u;
        const bool fresh = own_uniform && own_info.kind != KIND_NONUNIFORM;
        // (SIGNS: every voxel of a chunk the sampler gave planes has the generator's one type)
        const uint32_t st = signs_type * 0x01010101u;
        const uint32_t tw[4] = {fresh ? ut : (SIGNS ? st : own_types.x), fresh ? ut : (SIGNS ? st : own_types.y), fresh ? ut : (SIGNS ? st : own_types.z),
                                fresh ? ut : (SIGNS ? st : own_types.w)};
        chunk_moments_rows_tab(tid, m, tw, s_dens, s_mtab, s_red, (ci + (int)fz.x_off) * 16 + ti, cj * 16 + tj, ck * 16, fz.chunk_moments + (size_t)chunk * 10);
    }

    if (tid == 0 && IVX_DBG_KEEP(32u)) {
        touch[chunk] = (uint8_t)cnt[12];
        {  // the list entry carries what the later stages need from the record
            const bool obscured = kind == KIND_NONUNIFORM && nbr_full == 0x3Fu;
            const uint32_t entry = chunk | (kind << 24) | (gen << 26) | ((kind == KIND_NONUNIFORM && !obscured) ? (1u << 28) : 0u);
            if (!in_box) active_list[li] = entry;
            else if (box.out_list) box.out_list[li] = entry;
        }
        ivx_chunk_info ci_ = own_info;
        ci_.kind = (uint8_t)kind;
        ci_.region_count = (uint8_t)rc;
        ci_.boundary_region_count = (uint8_t)brc;
        if (kind == KIND_NONUNIFORM) {
            // bit layout: X_DN,Y_DN,Z_DN,X_UP,Y_UP,Z_UP <- faces f = 2*dim+side
            uint32_t ob = 0;
#pragma unroll
            for (int f = 0; f < 6; ++f)
                if ((nbr_full >> f) & 1u) ob |= 1u << ((f & 1) * 3 + (f >> 1));
            const bool only_empty = (cnt[0] | cnt[1] | cnt[2] | cnt[3] | cnt[4] | cnt[5]) == 0 && (ci_.flags & CF_ONLY_EMPTY);
            ci_.flags = (uint8_t)(ob | (only_empty ? CF_ONLY_EMPTY : 0u));
            uint32_t fd = 0;
#pragma unroll
            for (int f = 0; f < 6; ++f) fd |= own_fd[f] << (2 * f);
            ci_.face_dist = (uint16_t)fd;
            ci_.uniform_type = 0;
        } else if (kind == KIND_UNIFORM) {
            ci_.flags = 0;
            ci_.face_dist = 0x555;
            ci_.uniform_type = (uint8_t)utype;
        } else {
            ci_.flags = 0;
            ci_.face_dist = 0;
        }
        info[chunk] = ci_;
    }
    IVX_T(g, li, 5);
    }
}
template <bool SIGNS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(IVX_DERIVE_WAVES, 8))) void k_derive(DeriveArgs a) {
    derive_body<SIGNS>(a, blockIdx.x, gridDim.x);
}
IVX_MANY_TWIN(k_derive_planes_many, DeriveArgs, derive_body<false>, __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(IVX_DERIVE_WAVES, 8))))
IVX_MANY_TWIN(k_derive_signs_many, DeriveArgs, derive_body<true>, __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(IVX_DERIVE_WAVES, 8))))
IVX_MANY_LAUNCHER(many_derive_planes, k_derive_planes_many, DeriveArgs, 256)
IVX_MANY_LAUNCHER(many_derive_signs, k_derive_signs_many, DeriveArgs, 256)

// The active list anew from the chunk records alone (after a box sweep changed kinds: ivx_ensure_active_list): k_chunk_pre's rule for what is
// settled — Void, or generated Uniform among six chunks generated Uniform —, entries with the kinds and the exposure bit the derive sweep would
// have added, in chunk order. Writes nothing but the list and its counter: every chunk's per-step words are current (the box sweep kept them so).
struct ListRebuildArgs {
    GridView g;
    const ivx_chunk_info* info;
    uint32_t* work_counts;
    uint32_t* next_work_count;
    uint32_t* active_list;
    uint32_t* mesh_counts;
};
__device__ __forceinline__ void list_rebuild_body(const ListRebuildArgs& a, uint32_t bid, uint32_t) {
    const GridView& g = a.g;
    const ivx_chunk_info* __restrict__ info = a.info;
    uint32_t* __restrict__ work_counts = a.work_counts;
    uint32_t* __restrict__ next_work_count = a.next_work_count;
    uint32_t* __restrict__ active_list = a.active_list;
    uint32_t* __restrict__ mesh_counts = a.mesh_counts;
    __shared__ uint32_t s_w[4];
    __shared__ uint32_t s_base;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    if (bid == 0 && tid == 0) next_work_count[0] = 0u;
    const uint32_t n_chunks = g.cx * g.cy * g.cz;
    const uint32_t chunk = bid * 256u + tid;
    const bool live = chunk < n_chunks;
    bool settled = false;
    uint32_t entry = 0;
    if (live) {
        const ivx_chunk_info own = info[chunk];
        const uint32_t gen = own.gen_kind;
        settled = gen == KIND_VOID;
        if (gen == KIND_UNIFORM) {
            const int ck = chunk % g.cz, cj = (chunk / g.cz) % g.cy, ci = chunk / (g.cz * g.cy);
            if (cj > 0 && ck > 0 && cj + 1 < (int)g.cy && ck + 1 < (int)g.cz) {
                const uint32_t sx = g.cy * g.cz, sy = g.cz;
                const bool x_lo = ci > 0 ? info[chunk - sx].gen_kind == KIND_UNIFORM
                                         : (g.ghost_info[0] != nullptr && g.ghost_info[0][cj * g.cz + ck].gen_kind == KIND_UNIFORM);
                const bool x_hi = ci + 1 < (int)g.cx ? info[chunk + sx].gen_kind == KIND_UNIFORM
                                                     : (g.ghost_info[1] != nullptr && g.ghost_info[1][cj * g.cz + ck].gen_kind == KIND_UNIFORM);
                settled = x_lo && x_hi && info[chunk - sy].gen_kind == KIND_UNIFORM && info[chunk + sy].gen_kind == KIND_UNIFORM &&
                          info[chunk - 1].gen_kind == KIND_UNIFORM && info[chunk + 1].gen_kind == KIND_UNIFORM;
            }
        }
        const uint32_t kind = own.kind;
        const bool exposed = kind == KIND_NONUNIFORM && (own.flags & CF_FULLY_OBSCURED) != CF_FULLY_OBSCURED;
        entry = chunk | (kind << 24) | (gen << 26) | (exposed ? (1u << 28) : 0u);
        if (settled) {  // (as k_chunk_pre: a chunk off the list has no mesh — one that an edit emptied still has its old counts here)
            mesh_counts[2 * chunk] = 0;
            mesh_counts[2 * chunk + 1] = 0;
        }
    }
    const bool active = live && !settled;
    const unsigned long long bal = __ballot(active);
    if (lane == 0) s_w[wave] = (uint32_t)__popcll(bal);
    __syncthreads();
    if (tid == 0) s_base = atomicAdd(&work_counts[0], (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]));
    __syncthreads();
    if (active) {
        uint32_t off = s_base + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
        for (uint32_t w = 0; w < wave; ++w) off += s_w[w];
        active_list[off] = entry;
    }
}
__global__ __launch_bounds__(256) void k_list_rebuild(ListRebuildArgs a) { list_rebuild_body(a, blockIdx.x, gridDim.x); }
IVX_MANY_TWIN(k_list_rebuild_many, ListRebuildArgs, list_rebuild_body, __launch_bounds__(256))
IVX_MANY_LAUNCHER(many_list_rebuild, k_list_rebuild_many, ListRebuildArgs, 256)

// update_occupied_ranges (object.rs:1149-1280): tight [lo,hi) ranges of non-empty chunks and voxels from
// the per-chunk boxes written by k_derive. raw[0..6) = minima (chunk lo xyz, voxel lo xyz), raw[6..12) = maxima
// (chunk hi xyz, voxel hi xyz); the launcher presets them to 0xFFFFFFFF / 0.
__global__ __launch_bounds__(256) void k_occupied_reduce(uint32_t cx, uint32_t cy, uint32_t cz, const uint32_t* __restrict__ bbox,
                                                          uint32_t* __restrict__ raw) {
    __shared__ uint32_t red[12];
    const uint32_t tid = threadIdx.x;
    if (tid < 12) red[tid] = tid < 6 ? 0xFFFFFFFFu : 0u;
    __syncthreads();
    const uint32_t n = cx * cy * cz;
    const uint32_t c = blockIdx.x * 256u + tid;
    uint32_t p = c < n ? bbox[c] : 0u;
    if (__syncthreads_or((int)(p >> 31))) {
        uint32_t v[12];
#pragma unroll
        for (int q = 0; q < 12; ++q) v[q] = q < 6 ? 0xFFFFFFFFu : 0u;
        if (p & 0x80000000u) {
            const uint32_t ck = c % cz, cj = (c / cz) % cy, ci = c / (cz * cy);
            const uint32_t cc[3] = {ci, cj, ck};
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                v[d] = cc[d];
                v[6 + d] = cc[d] + 1;
                v[3 + d] = cc[d] * 16u + ((p >> (8 * d)) & 15u);
                v[9 + d] = cc[d] * 16u + ((p >> (8 * d + 4)) & 15u) + 1u;
            }
        }
        // wave-level min/max first: the 12 LDS words would otherwise serialise every lane's atomics
#pragma unroll
        for (int q = 0; q < 12; ++q) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const uint32_t t = __shfl_xor(v[q], o, 64);
                v[q] = q < 6 ? min(v[q], t) : max(v[q], t);
            }
        }
        if ((tid & 63u) == 0) {
#pragma unroll
            for (int q = 0; q < 12; ++q) {
                if (q < 6) atomicMin(&red[q], v[q]);
                else atomicMax(&red[q], v[q]);
            }
        }
        __syncthreads();
        if (tid < 6) atomicMin(&raw[tid], red[tid]);
        else if (tid < 12) atomicMax(&raw[tid], red[tid]);
    }
}

// The preset of the scratch word groups as a launch of its own: only for a step call that starts with neither the sampler nor the
// derive sweep (their first kernels host this role otherwise).
__global__ __launch_bounds__(256) void k_step_preset(ivx_roles::PresetArgs preset) { ivx_roles::role_preset(preset, blockIdx.x * 256u + threadIdx.x); }

// writes out the planes of the chunks that are only a record (Void / Uniform), for callers that want whole planes
__global__ __launch_bounds__(256) void k_materialize(uint32_t n_chunks, const ivx_chunk_info* __restrict__ info, int8_t* __restrict__ sdf,
                                                     uint8_t* __restrict__ type, uint8_t* __restrict__ flags, uint8_t* __restrict__ labels) {
    const uint32_t chunk = ivx_xcd_remap(blockIdx.x, n_chunks);
    const ivx_chunk_info ci = info[chunk];
    if (ci.kind == KIND_NONUNIFORM) return;
    const size_t o = (size_t)chunk * IVX_CHUNK_VOXELS + (size_t)threadIdx.x * 16;
    const uint32_t s = ivx_uniform_sdf(ci.kind) * 0x01010101u, t = ivx_uniform_type(ci) * 0x01010101u;
    const uint32_t f = ivx_uniform_flags(ci.kind) * 0x01010101u, l = ivx_uniform_label(ci.kind) * 0x01010101u;
    *reinterpret_cast<uint4*>(sdf + o) = make_uint4(s, s, s, s);
    *reinterpret_cast<uint4*>(type + o) = make_uint4(t, t, t, t);
    *reinterpret_cast<uint4*>(flags + o) = make_uint4(f, f, f, f);
    *reinterpret_cast<uint4*>(labels + o) = make_uint4(l, l, l, l);
}

static DeriveArgs derive_args(ivx_grid* g, const GridView& v, const DeriveFused& fz) {
    DeriveArgs da;
    memset(&da, 0, sizeof(da));
    da.g = v, da.sdf_rw = g->sdf, da.type_rw = g->type, da.flags_out = g->flags, da.info = g->info, da.bbox = g->chunk_bbox, da.touch = g->chunk_touch;
    da.signs = g->chunk_signs, da.kface_out = g->kface, da.work_counts = ivx_wc(g), da.active_list = g->active_list, da.list_in = g->active_list, da.fz = fz;
    return da;
}
static_assert(sizeof(DeriveArgs) % 8 == 0 && sizeof(ChunkPreArgs) % 8 == 0 && sizeof(ListRebuildArgs) % 8 == 0, "argument blocks travel as 8-byte words");
static const int s_many_registered_derive =
    (ivx_many_register(IVX_MK_DERIVE_PLANES, many_derive_planes, sizeof(DeriveArgs)), ivx_many_register(IVX_MK_DERIVE_SIGNS, many_derive_signs, sizeof(DeriveArgs)),
     ivx_many_register(IVX_MK_CHUNK_PRE, many_chunk_pre, sizeof(ChunkPreArgs)), ivx_many_register(IVX_MK_LIST_REBUILD, many_list_rebuild, sizeof(ListRebuildArgs)), 0);

}  // namespace

int ivx_ensure_dense(ivx_grid* g) {
    if (!g->planes_compact) return IVX_OK;
    IVX_KLAUNCH(k_materialize, dim3(g->n_chunks), dim3(256), 0, g->ctx->stream, g->n_chunks, g->info, g->sdf, g->type, g->flags, g->llabel);
    IVX_HIP_CHECK(hipGetLastError());
    g->planes_compact = 0;
    return IVX_OK;
}

ivx_roles::PresetArgs ivx_preset_args(ivx_grid* g, uint32_t groups) {
    const uint32_t n_groups = (g->n_chunks + 255u) / 256u;
    ivx_roles::PresetArgs a;
    a.groups = groups;
    a.rscalar = g->rscalar;
    a.sn_sums = g->group_sums + n_groups;
    a.n_sn = IVX_SN_GROUP_WORDS * n_groups;  // (role_preset also clears the nine words in use behind them: the general pass's counter, the main pass's list cursors)
    a.eval_count = g->samp_len ? g->samp_len + g->n_chunks : nullptr;
    return a;
}

int ivx_launch_step_preset(ivx_grid* g, uint32_t groups) {
    if (!groups) return IVX_OK;
    const ivx_roles::PresetArgs a = ivx_preset_args(g, groups);
    IVX_KLAUNCH(k_step_preset, dim3((a.n_sn + 255u) / 256u), dim3(256), 0, g->ctx->stream, a);
    IVX_HIP_CHECK(hipGetLastError());
    g->scratch_dirty &= ~groups;
    return IVX_OK;
}

// `preset_groups`: scratch word groups k_chunk_pre presets on the way (the fused step path; 0 elsewhere: stand-alone callers
// memset what they need themselves)
#ifdef IVX_DERIVE_DEBUG
static void derive_debug_mask() {
    static bool done = false;
    if (done) return;
    done = true;
    const char* e = getenv("IVX_DERIVE_SKIP");
    const uint32_t v = e ? (uint32_t)strtoul(e, nullptr, 0) : 0u;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(ivx_dbg_derive_skip), &v, sizeof(v));
}
#endif
int ivx_launch_derive(ivx_grid* g, uint32_t parts, uint32_t preset_groups) {
#ifdef IVX_DERIVE_DEBUG
    derive_debug_mask();
#endif
    GridView v = ivx_view(g);
    DeriveFused fz;
    fz.parts = parts;
    fz.x_off = g->x_off;
    fz.labels = g->llabel;
    fz.rparent = g->rparent;
    fz.rscalar = g->rscalar;
    fz.multi_list = g->ccl_scratch;  // as ivx_launch_ccl_local
    fz.dens = g->dens_dev;
    fz.chunk_moments = g->chunk_moments;
    // (the region scalars start from zero: preset by this launch's k_chunk_pre, or by the step's first kernel — the sampler's — in which case
    // nothing has used them since; only a caller outside a step finds them dirty)
    if ((parts & IVX_PART_REGIONS) && !(preset_groups & IVX_SCRATCH_REGIONS) && (g->scratch_dirty & IVX_SCRATCH_REGIONS))
        IVX_HIP_CHECK(ivx_memset_async(g->rscalar, 0, 16 * sizeof(uint32_t), g->ctx->stream));
    if (parts & IVX_PART_REGIONS) g->scratch_dirty |= IVX_SCRATCH_REGIONS;
    g->bbox_valid = 1;
    uint32_t* next_count = ivx_wc(g);  // the counter of the sweep before: zeroed by this one for the sweep after
    g->wc_cur ^= 1u;
    // (the sampler's list counters have had their last reader by now: rolled over on the way, see role_preset)
    const uint32_t roll = ((g->scratch_dirty & IVX_SCRATCH_EVAL) && g->samp_len) ? IVX_SCRATCH_EVAL_ROLL : 0u;
    ChunkPreArgs face_settle;  // (the arguments of k_face_settle: k_chunk_pre's)
    {
        ChunkPreArgs pa;
        memset(&pa, 0, sizeof(pa));
        pa.g = v, pa.info = g->info, pa.bbox = g->chunk_bbox, pa.mesh_counts = g->chunk_counts, pa.chunk_class = g->chunk_class, pa.touch = g->chunk_touch;
        pa.rparent = g->rparent, pa.work_counts = ivx_wc(g), pa.next_work_count = next_count, pa.active_list = g->active_list;
        pa.preset = ivx_preset_args(g, preset_groups | roll);
        // (a slab whose ghost layers are still on their way, the sweep split around their arrival: this kernel runs ahead of the wait and must
        // not look at the ghost layers' chunk records — a Uniform chunk of a face plane that only such a record can settle is left pending)
        if (g->ghost_event && g->ghost_split && ivx_has_interior_planes(g)) pa.defer_ghost = 1u;
        const uint32_t blocks = (g->n_chunks + 255u) / 256u;
        if (!ivx_many_try(g->ctx, g, IVX_MK_CHUNK_PRE, blocks, pa)) IVX_KLAUNCH(k_chunk_pre, dim3(blocks), dim3(256), 0, g->ctx->stream, pa);
        face_settle = pa;
    }
    g->scratch_dirty &= ~preset_groups;
    if (roll) g->scratch_dirty &= ~IVX_SCRATCH_EVAL;
    // The grid: a whole number of list entries per workgroup. Workgroups start in waves of seven per CU and a wave that is not full costs as
    // much as a full one: measured on 512^3, 5 388 entries ran in 53 us with the 6 125 workgroups of ivx_list_grid (an eighth of slack
    // against a growing list) and in 47 with 5 376; 32 416 entries in 261 us one to a workgroup, 242 with 8 192 workgroups of four entries
    // each, 259 with 7 168 (4.5 each). So: as many entries each as brings the grid nearest to 32 workgroups per CU, and exactly that many
    // workgroups; a list that has grown since the last step is covered by the walk's stride.
    uint32_t derive_grid = ivx_list_grid(g);
    if (g->last_active) {
        const uint32_t per_round = (uint32_t)g->ctx->n_cu * 32u;
        const uint32_t each = (g->last_active + per_round / 2u) / per_round > 1u ? (g->last_active + per_round / 2u) / per_round : 1u;
        derive_grid = (g->last_active + each - 1u) / each;
        const uint32_t lo = (g->n_chunks + 255u) / 256u;  // (ivx_list_grid's floor)
        if (derive_grid < lo) derive_grid = lo;
    }
    // (two forms of the sweep: from the sign rows the sampler left, while nothing else has rewritten voxels — no plane is read —, else from the planes)
    DeriveArgs da = derive_args(g, v, fz);
    if (g->signs_current) da.signs_type = (uint32_t)g->signs_type;
    auto sweep = [&](uint32_t x_part) {
        da.x_part = x_part;
        if (g->signs_current) {
            if (!ivx_many_try(g->ctx, g, IVX_MK_DERIVE_SIGNS, derive_grid, da)) IVX_KLAUNCH(k_derive<true>, dim3(derive_grid), dim3(256), 0, g->ctx->stream, da);
        } else {
            if (!ivx_many_try(g->ctx, g, IVX_MK_DERIVE_PLANES, derive_grid, da)) IVX_KLAUNCH(k_derive<false>, dim3(derive_grid), dim3(256), 0, g->ctx->stream, da);
        }
    };
    if (g->ghost_event) {
        // a slab whose ghost layers are still on their way (slab_comm.cpp: the exchange runs on the communicator's stream): the chunk planes
        // that read nothing of them first, the stream then waits for the arrival, the planes beside the ghost layers last
        const bool interior = g->ghost_split && ivx_has_interior_planes(g);
        if (interior) sweep(IVX_XPART_INTERIOR);
        (void)ivx_many_break();
        // (IVX_DEBUG_SKIP_GHOST_WAIT: developer switch — the wait left out, to show that the slab tests notice, tools/slab_overlap_check.sh)
        static const bool skip_wait = getenv("IVX_DEBUG_SKIP_GHOST_WAIT") && atoi(getenv("IVX_DEBUG_SKIP_GHOST_WAIT")) == 1;
        if (!skip_wait) IVX_HIP_CHECK(hipStreamWaitEvent(g->ctx->stream, static_cast<hipEvent_t>(g->ghost_event), 0));
        g->ghost_event = nullptr;
        if (face_settle.defer_ghost) IVX_KLAUNCH(k_face_settle, dim3((2u * g->cc[1] * g->cc[2] + 255u) / 256u), dim3(256), 0, g->ctx->stream, face_settle);
        sweep(interior ? IVX_XPART_FACES : IVX_XPART_ALL);
    } else {
        sweep(IVX_XPART_ALL);
    }
    g->active_list_stale = 0;
    g->planes_compact = 1;
    IVX_HIP_CHECK(hipGetLastError());
    return IVX_OK;
}

// The sweep over a box of chunks after an edit (DeriveBox above): derived state, chunk-local regions and (parts) chunk moments of the box's
// chunks from the planes, list entries to `d_out_list` (box order; may be null), the region forest of every other chunk reset for the resolve
// that follows. The caller zeroes g->rscalar first (region scalars; [2] counts the listed multi-region chunks). The object's active list is
// not touched — and is stale afterwards wherever a chunk of the box changed its kind (ivx_grid::active_list_stale).
int ivx_launch_derive_box(ivx_grid* g, uint32_t parts, const uint32_t lo[3], const uint32_t cc[3], uint32_t* d_out_list) {
    GridView v = ivx_view(g);
    DeriveFused fz;
    fz.parts = parts;
    fz.x_off = g->x_off;
    fz.labels = g->llabel;
    fz.rparent = g->rparent;
    fz.rscalar = g->rscalar;
    fz.multi_list = g->ccl_scratch;
    fz.dens = g->dens_dev;
    fz.chunk_moments = g->chunk_moments;
    DeriveBox box;
    box.n = cc[0] * cc[1] * cc[2];
    box.derive_blocks = box.n;
    for (int d = 0; d < 3; ++d) box.lo[d] = lo[d], box.cc[d] = cc[d];
    box.out_list = d_out_list;
    if (box.n == 0) return IVX_OK;
    g->scratch_dirty |= IVX_SCRATCH_REGIONS;
    g->active_list_stale = 1;
    g->planes_compact = 1;  // (a chunk the edit emptied is its record from now on: its planes are stale until ivx_ensure_dense)
    DeriveArgs da = derive_args(g, v, fz);
    da.box = box;
    const uint32_t blocks = box.derive_blocks + (g->n_chunks + 255u) / 256u;
    if (!ivx_many_try(g->ctx, g, IVX_MK_DERIVE_PLANES, blocks, da)) IVX_KLAUNCH(k_derive<false>, dim3(blocks), dim3(256), 0, g->ctx->stream, da);
    IVX_HIP_CHECK(hipGetLastError());
    return IVX_OK;
}

int ivx_ensure_active_list(ivx_grid* g) {
    if (!g->active_list_stale) return IVX_OK;
    uint32_t* next_count = ivx_wc(g);  // (the counters alternate as in ivx_launch_derive: this sweep zeroes the old one for the sweep after)
    g->wc_cur ^= 1u;
    ListRebuildArgs la;
    memset(&la, 0, sizeof(la));
    la.g = ivx_view(g), la.info = g->info, la.work_counts = ivx_wc(g), la.next_work_count = next_count, la.active_list = g->active_list, la.mesh_counts = g->chunk_counts;
    const uint32_t blocks = (g->n_chunks + 255u) / 256u;
    if (!ivx_many_try(g->ctx, g, IVX_MK_LIST_REBUILD, blocks, la)) IVX_KLAUNCH(k_list_rebuild, dim3(blocks), dim3(256), 0, g->ctx->stream, la);
    IVX_HIP_CHECK(hipGetLastError());
    g->active_list_stale = 0;
    return IVX_OK;
}

int ivx_launch_occupied(ivx_grid* g, uint32_t* d_raw) {
    IVX_HIP_CHECK(ivx_memset_async(d_raw, 0xFF, 6 * sizeof(uint32_t), g->ctx->stream));
    IVX_HIP_CHECK(ivx_memset_async(d_raw + 6, 0, 6 * sizeof(uint32_t), g->ctx->stream));
    IVX_KLAUNCH(k_occupied_reduce, dim3((g->n_chunks + 255u) / 256u), dim3(256), 0, g->ctx->stream, g->cc[0], g->cc[1], g->cc[2], g->chunk_bbox, d_raw);
    IVX_HIP_CHECK(hipGetLastError());
    return IVX_OK;
}

// raw minima/maxima -> the public layout: chunk lo/hi x3, voxel lo/hi x3 (all zero for an empty object)
void ivx_occupied_from_raw(const ivx_grid* g, const uint32_t raw[12], uint32_t out[12]) {
    if (raw[6] == 0) {  // no non-empty voxel (object.rs:1177-1190)
        for (int i = 0; i < 12; ++i) out[i] = 0;
        return;
    }
    for (int d = 0; d < 3; ++d) {
        out[2 * d] = raw[d];
        out[2 * d + 1] = raw[6 + d];
        out[6 + 2 * d] = raw[3 + d];
        out[7 + 2 * d] = raw[9 + d];
    }
    out[0] += g->x_off;
    out[1] += g->x_off;
    out[6] += g->x_off * 16u;
    out[7] += g->x_off * 16u;
}
