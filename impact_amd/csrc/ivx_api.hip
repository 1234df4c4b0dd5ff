// C ABI entry points of libimpact_voxel_hip.so (see include/impact_voxel_hip.h for the contract and
// the reference interfaces each function replaces). Host-side orchestration only: every compute call
// launches the HIP kernels in this directory on the context's stream.
#include <algorithm>
#include <array>
#include <functional>
#include <cmath>
#include <atomic>
#include <chrono>
#include <cstring>
#include <limits>
#include <map>
#include <unordered_map>
#include <new>
#include <vector>

#include <type_traits>

#include <dlfcn.h>

#include "ivx_internal.hpp"

static thread_local char g_error[512] = "";

void ivx_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
}

namespace {

template <class T>
int dev_alloc(T** p, size_t count) {
    *p = nullptr;
    if (count == 0) return IVX_OK;
    IVX_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(p), count * sizeof(T)));
    return IVX_OK;
}

int ensure_host_scratch(ivx_grid* g, size_t bytes) {
    if (g->host_scratch_bytes >= bytes) return IVX_OK;
    IVX_HIP_CHECK(ivx_stream_sync(g->ctx->stream));
    if (g->host_scratch) (void)hipHostFree(g->host_scratch);
    g->host_scratch = nullptr;
    g->host_scratch_bytes = 0;
    if (bytes < (64u << 10)) bytes = 64u << 10;
    IVX_HIP_CHECK(hipHostMalloc(&g->host_scratch, bytes, hipHostMallocDefault));
    g->host_scratch_bytes = bytes;
    return IVX_OK;
}

// Host<->device copies. Small ones (the scalars and tables the entry points hand back: up to a few hundred KB) go through the
// grid's pinned staging buffer as ONE stream-ordered copy and ONE wait (a blocking copy from pageable memory costs a wait for
// the stream, a staging copy inside the runtime and a second wait); large ones (whole planes) stay plain blocking copies.
constexpr size_t STAGED_COPY_MAX = 1u << 20;
int d2h(ivx_grid* g, void* dst, const void* src, size_t bytes) {
    if (bytes == 0) return IVX_OK;
    if (bytes <= STAGED_COPY_MAX) {
        int rc = ensure_host_scratch(g, bytes);
        if (rc) return rc;
        IVX_HIP_CHECK(ivx_memcpy_async(g->host_scratch, src, bytes, hipMemcpyDeviceToHost, g->ctx->stream));
        IVX_HIP_CHECK(ivx_stream_sync(g->ctx->stream));
        memcpy(dst, g->host_scratch, bytes);
        return IVX_OK;
    }
    IVX_HIP_CHECK(ivx_stream_sync(g->ctx->stream));
    IVX_HIP_CHECK(ivx_memcpy_sync(dst, src, bytes, hipMemcpyDeviceToHost));
    return IVX_OK;
}
int h2d(ivx_grid* g, void* dst, const void* src, size_t bytes) {
    if (bytes == 0) return IVX_OK;
    if (bytes <= STAGED_COPY_MAX) {
        int rc = ensure_host_scratch(g, bytes);
        if (rc) return rc;
        IVX_HIP_CHECK(ivx_stream_sync(g->ctx->stream));  // the staging buffer may still feed an earlier copy
        memcpy(g->host_scratch, src, bytes);
        IVX_HIP_CHECK(ivx_memcpy_async(dst, g->host_scratch, bytes, hipMemcpyHostToDevice, g->ctx->stream));
        IVX_HIP_CHECK(ivx_stream_sync(g->ctx->stream));
        return IVX_OK;
    }
    IVX_HIP_CHECK(ivx_stream_sync(g->ctx->stream));
    IVX_HIP_CHECK(ivx_memcpy_sync(dst, src, bytes, hipMemcpyHostToDevice));
    return IVX_OK;
}
int ensure_dev_scratch(ivx_grid* g, size_t bytes) {
    if (g->dev_scratch_bytes >= bytes) return IVX_OK;
    IVX_HIP_CHECK(ivx_stream_sync(g->ctx->stream));
    if (g->dev_scratch) (void)hipFree(g->dev_scratch);
    g->dev_scratch = nullptr;
    const size_t had = g->dev_scratch_bytes;
    g->dev_scratch_bytes = 0;
    bytes = std::max<size_t>(std::max<size_t>(bytes, 2 * had), 64u << 10);  // (a growth is a wait on the stream: rare by construction)
    IVX_HIP_CHECK(hipMalloc(&g->dev_scratch, bytes));
    g->dev_scratch_bytes = bytes;
    return IVX_OK;
}

// Shared allocations (ivx_block, ivx_internal.hpp): released by the last grid that holds a part.
void block_release(ivx_block* b) {
    if (!b || --b->refs > 0) return;
    if (b->dev) (void)hipFree(b->dev);
    if (b->pinned) (void)hipHostFree(b->pinned);
    delete b;
}
// the mesh arrays of a group (1: positions, normals, vertex scratch; 2: indices, index materials; 4: submeshes) given up: freed, or — parts of a
// shared block — just let go, the block with its last part
void mesh_group_free(ivx_grid* g, uint32_t group) {
    const bool pooled = (g->mesh_pooled & group) != 0u;
    auto drop = [&](auto*& p) {
        if (p && !pooled) (void)hipFree(p);
        p = nullptr;
    };
    if (group == 1u) drop(g->positions), drop(g->normals), drop(g->vertex_materials);
    if (group == 2u) drop(g->indices), drop(g->index_materials);
    if (group == 4u) drop(g->submeshes);
    if (pooled) {
        g->mesh_pooled &= ~group;
        if (!g->mesh_pooled) {
            block_release(g->mesh_block);
            g->mesh_block = nullptr;
        }
    }
}
// Mesh arrays for several grids from ONE device allocation (the fragments of an impact meet their first mesh together): capacities
// `vcaps` / `icaps` / `scaps` per grid; whatever the grids held before is given up. Every array starts on a 256-byte boundary.
int mesh_pool_assign(ivx_grid* const* grids, size_t n, const size_t* vcaps, const size_t* icaps, const size_t* scaps) {
    if (n == 0) return IVX_OK;
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    size_t total = 0;
    for (size_t i = 0; i < n; ++i) total += 2 * up(vcaps[i] * 12) + up(vcaps[i] * 16) + up(icaps[i] * 4) + up(icaps[i] * 8) + up(scaps[i] * sizeof(ivx_submesh));
    ivx_block* blk = new (std::nothrow) ivx_block();
    IVX_REQUIRE(blk, IVX_ERR_CAPACITY, "mesh buffers: out of host memory");
    blk->dev = blk->pinned = nullptr;
    blk->refs = 0;
    if (hipMalloc(&blk->dev, total) != hipSuccess) {
        delete blk;
        ivx_set_error("mesh buffers: device allocation of %zu bytes for %zu objects failed", total, n);
        return IVX_ERR_HIP;
    }
    char* b = static_cast<char*>(blk->dev);
    for (size_t i = 0; i < n; ++i) {
        ivx_grid* g = grids[i];
        mesh_group_free(g, 1u), mesh_group_free(g, 2u), mesh_group_free(g, 4u);
        auto take = [&](auto*& p, size_t bytes) {
            p = reinterpret_cast<std::remove_reference_t<decltype(p)>>(b);
            b += up(bytes);
        };
        take(g->positions, vcaps[i] * 12), take(g->normals, vcaps[i] * 12), take(g->vertex_materials, vcaps[i] * 16);
        take(g->indices, icaps[i] * 4), take(g->index_materials, icaps[i] * 8), take(g->submeshes, scaps[i] * sizeof(ivx_submesh));
        g->vcap = vcaps[i], g->icap = icaps[i], g->scap = scaps[i];
        g->mesh_block = blk;
        g->mesh_pooled = 7u;
        g->mesh_generation += 1;
        blk->refs += 1;
    }
    return IVX_OK;
}

// Developer experiment (IVX_MESH_ARENA=<mode>, IVX_MESH_PHASE=<bytes>): the mesher's four output arrays (+ the vertex scratch) carved from ONE
// allocation at chosen relative offsets — array k starts `k * phase` bytes past its 2 MiB-aligned place. mode 1: hipMalloc; 2: one physically
// contiguous block (hipExtMallocWithFlags(hipDeviceMallocContiguous)). What the mesher's run-to-run modes do NOT depend on.
static int mesh_arena_mode() {
    static const int m = [] {
        const char* e = getenv("IVX_MESH_ARENA");
        return e ? atoi(e) : 0;
    }();
    return m;
}
static int ensure_mesh_block(ivx_grid* g, size_t nv, size_t ni) {
    if (nv <= g->vcap && ni <= g->icap) return IVX_OK;
    IVX_HIP_CHECK(ivx_stream_sync(g->ctx->stream));
    static const size_t phase = [] {
        const char* e = getenv("IVX_MESH_PHASE");
        return (size_t)(e ? strtoull(e, nullptr, 0) : 0ull);
    }();
    const size_t vcap = std::max(2 * nv + 4096, g->vcap * 2), icap = std::max(2 * ni + 24576, g->icap * 2);
    const size_t sizes[5] = {vcap * 12, vcap * 12, icap * 4, icap * 8, vcap * 16};
    size_t off[5], total = 0;
    for (int k = 0; k < 5; ++k) {
        total = (total + (2u << 20) - 1) & ~(size_t)((2u << 20) - 1);
        off[k] = total + (size_t)k * phase;
        total = off[k] + sizes[k];
    }
    mesh_group_free(g, 1u), mesh_group_free(g, 2u);
    g->vcap = g->icap = 0;
    ivx_block* blk = new (std::nothrow) ivx_block();
    IVX_REQUIRE(blk, IVX_ERR_CAPACITY, "mesh buffers: out of host memory");
    blk->dev = blk->pinned = nullptr;
    blk->refs = 1;
    if ((mesh_arena_mode() == 2 ? hipExtMallocWithFlags(&blk->dev, total, hipDeviceMallocContiguous) : hipMalloc(&blk->dev, total)) != hipSuccess) {
        delete blk;
        ivx_set_error("mesh buffers: device allocation of %zu bytes failed", total);
        return IVX_ERR_HIP;
    }
    char* b = static_cast<char*>(blk->dev);
    g->positions = reinterpret_cast<float*>(b + off[0]);
    g->normals = reinterpret_cast<float*>(b + off[1]);
    g->indices = reinterpret_cast<uint32_t*>(b + off[2]);
    g->index_materials = reinterpret_cast<uint8_t*>(b + off[3]);
    g->vertex_materials = reinterpret_cast<uint8_t*>(b + off[4]);
    g->vcap = vcap, g->icap = icap;
    if (g->mesh_block) block_release(g->mesh_block);  // (only the submeshes of an earlier shared block can still be there: they move out below)
    g->mesh_block = blk;
    g->mesh_pooled = 3u;
    g->mesh_generation += 1;
    return IVX_OK;
}

int ensure_mesh_capacity(ivx_grid* g, size_t nv, size_t ni, size_t ns) {
    if (mesh_arena_mode())
        if (int rc = ensure_mesh_block(g, nv, ni)) return rc;
    if (nv > g->vcap || ni > g->icap || ns > g->scap) g->mesh_generation += 1;
    if (nv > g->vcap) {
        size_t cap = std::max(2 * nv + 4096, g->vcap * 2);  // (slack: under the reference's range allocator an edited mesh's buffers grow — a re-meshed chunk that needs a little more than it had goes to the end — by about as much again before freed ranges start to be reused; 2 x 48 B per vertex is nothing next to 288 GB)
        mesh_group_free(g, 1u);
        g->vcap = 0;
        int rc;
        if ((rc = dev_alloc(&g->positions, cap * 3))) return rc;
        if ((rc = dev_alloc(&g->normals, cap * 3))) return rc;
        if ((rc = dev_alloc(&g->vertex_materials, cap * 16))) return rc;
        g->vcap = cap;
    }
    if (ni > g->icap) {
        size_t cap = std::max(2 * ni + 24576, g->icap * 2);
        mesh_group_free(g, 2u);
        g->icap = 0;
        int rc;
        if ((rc = dev_alloc(&g->indices, cap))) return rc;
        if ((rc = dev_alloc(&g->index_materials, cap * 8))) return rc;
        g->icap = cap;
    }
    if (ns > g->scap) {
        size_t cap = std::max(2 * ns + 64, g->scap * 2);
        mesh_group_free(g, 4u);
        g->scap = 0;
        int rc;
        if ((rc = dev_alloc(&g->submeshes, cap))) return rc;
        g->scap = cap;
    }
    return IVX_OK;
}

__global__ __launch_bounds__(256) void k_halo_pack(GridView g, uint32_t side, int8_t* __restrict__ out_sdf, uint8_t* __restrict__ out_type,
                                                   ivx_chunk_info* __restrict__ out_info) {
    const uint32_t col = blockIdx.x;  // cj*cz + ck
    const uint32_t tid = threadIdx.x;
    const uint32_t ci = side ? g.cx - 1 : 0u;
    const uint32_t chunk = ci * g.cy * g.cz + col;
    const size_t src = (size_t)chunk * IVX_CHUNK_VOXELS + ((side ? 15u : 0u) << 8) + tid;
    const ivx_chunk_info rec = g.info[chunk];
    const bool dense = rec.kind == KIND_NONUNIFORM;  // else the chunk is its record (compact planes)
    out_sdf[(size_t)col * 256 + tid] = dense ? g.sdf[src] : (int8_t)ivx_uniform_sdf(rec.kind);
    out_type[(size_t)col * 256 + tid] = dense ? g.type[src] : (uint8_t)ivx_uniform_type(rec);
    if (tid == 0) out_info[col] = rec;
}

}  // namespace

int ivx_launch_halo_pack(ivx_grid* g, int side, void* buf) {
    const size_t cols = (size_t)g->cc[1] * g->cc[2];
    int8_t* o_sdf = static_cast<int8_t*>(buf);
    uint8_t* o_type = static_cast<uint8_t*>(buf) + cols * 256;
    ivx_chunk_info* o_info = reinterpret_cast<ivx_chunk_info*>(static_cast<uint8_t*>(buf) + cols * 512);
    GridView v = ivx_view(g);
    IVX_KLAUNCH(k_halo_pack, dim3((uint32_t)cols), dim3(256), 0, g->ctx->stream, v, (uint32_t)side, o_sdf, o_type, o_info);
    IVX_HIP_CHECK(hipGetLastError());
    return IVX_OK;
}

// what the incremental remesh needs to know about the listed chunks: vertex count, index count, kind | flags << 8
__global__ __launch_bounds__(256) void k_chunk_mesh_needs(uint32_t n, const uint32_t* __restrict__ list, const uint32_t* __restrict__ counts,
                                                          const ivx_chunk_info* __restrict__ info, uint32_t* __restrict__ out) {
    const uint32_t e = blockIdx.x * 256u + threadIdx.x;
    if (e >= n) return;
    const uint32_t c = list[e];
    out[3 * e] = counts[2 * (size_t)c];
    out[3 * e + 1] = counts[2 * (size_t)c + 1];
    out[3 * e + 2] = (uint32_t)info[c].kind | ((uint32_t)info[c].flags << 8);
}

// ---- incremental remesh (row a7: VoxelObjectMesh::sync_with_voxel_object, mesh.rs:355-456) ---------------------------------------------
// Host mirror of the ChunkSubmeshManager (mesh.rs:699-849) with its two RangeAllocators (impact_containers/src/range_allocator.rs): which slot
// of the submesh table a chunk owns and which ranges of the vertex / index buffers are free. The mesh data stays in HBM.
// the occupied ranges in the reference's sense (see ivx_grid::occ_ref): cached, recomputed only when something invalidated them
static int reference_occupied(ivx_grid* g, uint32_t occ[12]) {
    if (!g->occ_ref_valid) {
        uint32_t* d_occ = g->rscalar + 16;
        int rc;
        if ((rc = ivx_launch_occupied(g, d_occ))) return rc;
        uint32_t raw[12];
        if ((rc = d2h(g, raw, d_occ, sizeof(raw)))) return rc;
        ivx_occupied_from_raw(g, raw, g->occ_ref);
        g->occ_ref_valid = 1;
    }
    memcpy(occ, g->occ_ref, sizeof(g->occ_ref));
    return IVX_OK;
}

struct ivx_range_allocator {
    std::map<size_t, size_t> free_ranges;  // start -> end; a second range with the same start is dropped, as BTreeSet::insert does
    void free_range(size_t a, size_t b) {
        if (a < b) free_ranges.emplace(a, b);
    }
    bool allocate(size_t len, size_t* start) {  // the smallest free range that fits, the first of equals
        auto best = free_ranges.end();
        size_t best_len = std::numeric_limits<size_t>::max();
        for (auto it = free_ranges.begin(); it != free_ranges.end(); ++it) {
            const size_t l = it->second - it->first;
            if (l < best_len && l >= len) best = it, best_len = l;
        }
        if (best == free_ranges.end()) return false;
        const size_t a = best->first, b = best->second;
        free_ranges.erase(best);
        if (a + len < b) free_ranges.emplace(a + len, b);
        *start = a;
        return true;
    }
    void merge_consecutive() {  // (in place: a range that starts where the one before it ends is folded into that one)
        if (free_ranges.size() < 2) return;
        auto prev = free_ranges.begin();
        for (auto it = std::next(prev); it != free_ranges.end();) {
            if (it->first == prev->second) {
                prev->second = it->second;
                it = free_ranges.erase(it);
            } else {
                prev = it;
                ++it;
            }
        }
    }
};
struct ivx_submesh_manager {
    std::vector<ivx_submesh> table;                   // slot order = the reference's chunk_submeshes order
    std::unordered_map<uint32_t, uint32_t> slot_of;   // linear chunk index -> slot
    ivx_range_allocator vertices, indices;
    size_t total_vertices = 0, total_indices = 0;     // buffer lengths (freed ranges inside them stay counted)
    std::vector<ivx_submesh_data_ranges> updated;     // VoxelMeshModifications (mesh.rs:113-123) since the last report
    bool chunks_were_removed = false;
    uint64_t serial = 0;                              // the mesh_serial this state describes
};
void ivx_submesh_manager_free(ivx_submesh_manager* m) { delete m; }
// VoxelObjectCollisionProbes' bookkeeping (collidable.rs:97-101): chunk -> range of the point buffer, free ranges
struct ivx_probe_manager {
    std::unordered_map<uint32_t, std::pair<uint32_t, uint32_t>> range_of;  // linear chunk index -> [start, end)
    ivx_range_allocator points;
    size_t total = 0;  // length of the point buffer, freed ranges included
    bool built = true;  // false right after a recompute: the map is filled from the device entries when somebody needs it
};
void ivx_probe_manager_free(ivx_probe_manager* m) { delete m; }
static int probe_manager_build(ivx_grid* g);

namespace {
uint32_t linear_chunk(const ivx_grid* g, const uint32_t c[3]) { return (c[0] * g->cc[1] + c[1]) * g->cc[2] + c[2]; }
void manager_remove(ivx_grid* g, ivx_submesh_manager* m, uint32_t chunk) {  // remove_chunk_if_present (mesh.rs:811-824)
    auto it = m->slot_of.find(chunk);
    if (it == m->slot_of.end()) return;
    const uint32_t slot = it->second;
    m->slot_of.erase(it);
    const ivx_submesh gone = m->table[slot];
    if (slot + 1 != m->table.size()) {
        m->table[slot] = m->table.back();
        m->slot_of[linear_chunk(g, m->table[slot].chunk_indices)] = slot;
    }
    m->table.pop_back();
    m->chunks_were_removed = true;
    m->vertices.free_range(gone.vertex_offset, (size_t)gone.vertex_offset + gone.vertex_count);
    m->indices.free_range(gone.index_offset, (size_t)gone.index_offset + gone.index_count);
}
// grow the mesh buffers keeping what they hold (the full remesh may simply reallocate, a sync may not): every array that has to grow gets its
// new block and its copy on the stream, then ONE wait, then the old blocks go (a wait per array was most of what a growth cost)
struct GrowKeep {
    void** slot;
    void* fresh;
    uint32_t group;  // the mesh group the array belongs to (mesh_group_free)
};
template <class T>
int grow_keep_enqueue(ivx_grid* g, T** buf, size_t old_count, size_t new_count, std::vector<GrowKeep>& pending, uint32_t group) {
    T* fresh = nullptr;
    int rc = dev_alloc(&fresh, new_count);
    if (rc) return rc;
    if (*buf && old_count) IVX_HIP_CHECK(ivx_memcpy_async(fresh, *buf, old_count * sizeof(T), hipMemcpyDeviceToDevice, g->ctx->stream));
    pending.push_back(GrowKeep{reinterpret_cast<void**>(buf), fresh, group});
    return IVX_OK;
}
int ensure_mesh_capacity_keep(ivx_grid* g, size_t nv, size_t ni, size_t ns) {
    int rc;
    if (!(nv > g->vcap || ni > g->icap || ns > g->scap)) return IVX_OK;
    g->mesh_generation += 1;
    // (growth by copy costs allocations, device copies and a wait — the price of hundreds of small objects' syncs when each of them outgrows
    // its buffers by a few vertices per frame: double, and never by less than a few thousand elements; memory is not what this part is short of)
    std::vector<GrowKeep> pending;
    struct FreeOnError {  // (a failed allocation further down must not leak the blocks made so far)
        std::vector<GrowKeep>& p;
        bool armed = true;
        ~FreeOnError() {
            if (armed)
                for (GrowKeep& k : p) (void)hipFree(k.fresh);
        }
    } guard{pending};
    size_t vcap = g->vcap, icap = g->icap, scap = g->scap;
    if (nv > g->vcap) {
        vcap = std::max(nv + nv / 2 + 4096, 2 * g->vcap);
        if ((rc = grow_keep_enqueue(g, &g->positions, g->vcap * 3, vcap * 3, pending, 1u))) return rc;
        if ((rc = grow_keep_enqueue(g, &g->normals, g->vcap * 3, vcap * 3, pending, 1u))) return rc;
        if ((rc = grow_keep_enqueue(g, &g->vertex_materials, (size_t)0, vcap * 16, pending, 1u))) return rc;  // scratch of the emit kernel
    }
    if (ni > g->icap) {
        icap = std::max(ni + ni / 2 + 24576, 2 * g->icap);
        if ((rc = grow_keep_enqueue(g, &g->indices, g->icap, icap, pending, 2u))) return rc;
        if ((rc = grow_keep_enqueue(g, &g->index_materials, g->icap * 8, icap * 8, pending, 2u))) return rc;
    }
    if (ns > g->scap) {
        scap = std::max(ns + ns / 2 + 64, 2 * g->scap);
        if ((rc = grow_keep_enqueue(g, &g->submeshes, g->scap, scap, pending, 4u))) return rc;
    }
    IVX_HIP_CHECK(ivx_stream_sync(g->ctx->stream));
    guard.armed = false;
    uint32_t groups = 0;
    for (GrowKeep& k : pending) groups |= k.group;
    for (uint32_t grp = 1u; grp <= 4u; grp <<= 1)
        if (groups & grp) mesh_group_free(g, grp);  // (the old arrays: freed, or let go of as parts of a shared block)
    for (GrowKeep& k : pending) *k.slot = k.fresh;
    g->vcap = vcap, g->icap = icap, g->scap = scap;
    return IVX_OK;
}
}  // namespace

extern "C" {

const char* ivx_last_error(void) { return g_error; }

int ivx_init(int device_id, void* stream, ivx_ctx** out) {
    IVX_REQUIRE(out, IVX_ERR_INVALID, "ivx_init: null output pointer");
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0) {
        ivx_set_error("ivx_init: no HIP device available (%s); this library has no CPU fallback", e == hipSuccess ? "0 devices" : hipGetErrorString(e));
        return IVX_ERR_HIP;
    }
    IVX_REQUIRE(device_id >= 0 && device_id < count, IVX_ERR_INVALID, "ivx_init: device %d out of range (0..%d)", device_id, count - 1);
    IVX_HIP_CHECK(hipSetDevice(device_id));
    hipDeviceProp_t prop;
    IVX_HIP_CHECK(hipGetDeviceProperties(&prop, device_id));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        ivx_set_error("ivx_init: device %d is %s; this library is built for gfx950 (MI355X) only", device_id, prop.gcnArchName);
        return IVX_ERR_HIP;
    }
    ivx_ctx* c = new (std::nothrow) ivx_ctx();
    IVX_REQUIRE(c, IVX_ERR_CAPACITY, "ivx_init: out of host memory");
    c->device = device_id;
    c->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (stream) {
        c->stream = static_cast<hipStream_t>(stream);
        c->own_stream = false;
    } else {
        hipError_t se = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (se != hipSuccess) {
            ivx_set_error("ivx_init: hipStreamCreate failed: %s", hipGetErrorString(se));
            delete c;
            return IVX_ERR_HIP;
        }
        c->own_stream = true;
    }
    *out = c;
    return IVX_OK;
}

void ivx_shutdown(ivx_ctx* c) {
    if (!c) return;
    (void)ivx_stream_sync(c->stream);
    ivx_many_release(c);  // (the launch recorder of the many-object calls and its staging ring)
    if (c->pinned_scratch) (void)hipHostFree(c->pinned_scratch);
    if (c->dev_scratch) (void)hipFree(c->dev_scratch);
    if (c->aux_stream) {
        (void)hipStreamSynchronize(c->aux_stream);
        (void)hipStreamDestroy(c->aux_stream);
    }
    if (c->own_stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int ivx_synchronize(ivx_ctx* c) {
    IVX_REQUIRE(c, IVX_ERR_INVALID, "ivx_synchronize: null context");
    IVX_HIP_CHECK(ivx_stream_sync(c->stream));
    return IVX_OK;
}

void* ivx_stream(ivx_ctx* c) { return c ? static_cast<void*>(c->stream) : nullptr; }

// ONE device allocation for everything of a grid whose size follows from the chunk counts, carved at 256-byte boundaries: a grid used to
// cost ~45 hipMalloc calls, which was most of the price of creating the small grids that split-off and polyhedron clips make (one per
// fragment). `arena` null: sizes only. Returns the arena's bytes.
static size_t grid_carve(ivx_grid* g, char* arena) {
    size_t arena_bytes = 0;
    const size_t cols = (size_t)g->cc[1] * g->cc[2];
    auto carve = [&](auto** p, size_t count) {
        using T = std::remove_pointer_t<std::remove_pointer_t<decltype(p)>>;
        const size_t bytes = (count * sizeof(T) + 255) & ~(size_t)255;
        if (arena) *p = reinterpret_cast<T*>(arena + arena_bytes);
        arena_bytes += bytes;
    };
    carve(&g->sdf, g->n_vox);
    carve(&g->type, g->n_vox);
    carve(&g->flags, g->n_vox);
    carve(&g->llabel, g->n_vox);
    carve(&g->info, (size_t)g->n_chunks);
    carve(&g->chunk_bbox, (size_t)g->n_chunks);
    for (int sd = 0; sd < 2; ++sd) {
        carve(&g->ghost_sdf[sd], cols * 256);
        carve(&g->ghost_type[sd], cols * 256);
        carve(&g->ghost_info[sd], cols);
    }
    carve(&g->chunk_counts, (size_t)g->n_chunks * 2);
    carve(&g->chunk_offsets, (size_t)g->n_chunks * 3 + 8);
    carve(&g->partials, g->partial_blocks * 10 + 16);
    carve(&g->rparent, (size_t)g->n_chunks * 256);
    carve(&g->rcompid, (size_t)g->n_chunks * 256);
    carve(&g->rscalar, (size_t)64);
    carve(&g->ccl_scratch, (size_t)g->n_chunks * 2);
    carve(&g->sn_list, (size_t)g->n_chunks * 4);  // one uint4 record per meshed chunk
    carve(&g->group_sums, (size_t)((g->n_chunks + 255u) / 256u) * (1u + IVX_SN_GROUP_WORDS) + IVX_SN_TAIL_WORDS);
    carve(&g->sn_hard, (size_t)g->n_chunks);
    carve(&g->sn_walk, (size_t)g->n_chunks * 5 + 2);  // the main pass's walk order: a uint4 record and a list index per meshed chunk, the two class counts (role_sn_scan)
    carve(&g->dens_dev, (size_t)256);
    carve(&g->work_counts, (size_t)8);
    carve(&g->occ_part, (size_t)((g->n_chunks + 255u) / 256u) * 12 + 12);
    carve(&g->active_list, (size_t)g->n_chunks);
    carve(&g->chunk_class, (size_t)g->n_chunks);
    carve(&g->chunk_touch, (size_t)g->n_chunks);
    carve(&g->chunk_signs, (size_t)g->n_chunks * 256);
    carve(&g->kface, (size_t)g->n_chunks * 1024);
    carve(&g->chunk_moments, (size_t)g->n_chunks * 10);
    return arena_bytes;
}
// the host side of a new grid (no device memory yet)
static int grid_new(ivx_ctx* c, const uint32_t cc[3], float voxel_extent, uint32_t x_chunk_offset, uint32_t global_x_chunks, ivx_grid** out) {
    IVX_REQUIRE(c && cc && out, IVX_ERR_INVALID, "ivx_grid_create: null argument");
    *out = nullptr;
    IVX_REQUIRE(cc[0] > 0 && cc[1] > 0 && cc[2] > 0, IVX_ERR_INVALID, "ivx_grid_create: empty chunk grid");
    IVX_REQUIRE(voxel_extent > 0.0f, IVX_ERR_INVALID, "ivx_grid_create: voxel extent must be positive");
    const uint64_t n64 = (uint64_t)cc[0] * cc[1] * cc[2];
    // GlobalRegionLabel packs the chunk index in 24 bits (split_detection.rs:1539-1571)
    IVX_REQUIRE(n64 <= (1u << 24), IVX_ERR_CAPACITY, "ivx_grid_create: more than 2^24 chunks");
    ivx_grid* g = new (std::nothrow) ivx_grid();
    IVX_REQUIRE(g, IVX_ERR_CAPACITY, "ivx_grid_create: out of host memory");
    memset(g, 0, sizeof(*g));
    g->scratch_dirty = IVX_SCRATCH_REGIONS;  // (the pool is not cleared: the region scalars' first user outside a step must zero them, ivx_launch_derive)
    g->ctx = c;
    for (int d = 0; d < 3; ++d) g->cc[d] = cc[d];
    g->n_chunks = (uint32_t)n64;
    g->n_vox = (size_t)n64 * IVX_CHUNK_VOXELS;
    g->extent = voxel_extent;
    g->x_off = x_chunk_offset;
    g->gx = global_x_chunks ? global_x_chunks : cc[0];
    g->partial_blocks = 2048;
    *out = g;
    return IVX_OK;
}

static void ivx_block_release(ivx_block* b) { block_release(b); }

int ivx_grid_create(ivx_ctx* c, const uint32_t cc[3], float voxel_extent, uint32_t x_chunk_offset, uint32_t global_x_chunks, ivx_grid** out) {
    ivx_grid* g = nullptr;
    int rc = grid_new(c, cc, voxel_extent, x_chunk_offset, global_x_chunks, &g);
    if (rc) return rc;
    IVX_HIP_CHECK(hipSetDevice(c->device));
    const size_t arena_bytes = grid_carve(g, nullptr);
    char* arena = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&arena), arena_bytes) != hipSuccess) {
        ivx_set_error("ivx_grid_create: device allocation of %zu bytes failed", arena_bytes);
        delete g;
        return IVX_ERR_HIP;
    }
    g->arena = arena;
    (void)grid_carve(g, arena);
    if (ivx_memset_async(g->info, 0, sizeof(ivx_chunk_info) * g->n_chunks, c->stream) != hipSuccess ||
        ivx_memset_async(g->work_counts, 0, 8 * sizeof(uint32_t), c->stream) != hipSuccess) {
        ivx_set_error("ivx_grid_create: memset failed");
        ivx_grid_destroy(g);
        return IVX_ERR_HIP;
    }
    *out = g;
    return IVX_OK;
}

// Grids that come into being together (the fragments of an impact, fracturing.rs:1047-1189; docs/voxel_gpu_buffer_pooling.md:44-66): their
// arenas from ONE device allocation and their host-mapped result blocks from ONE pinned allocation, both released with the last of them.
// The chunk records and work counters start zeroed by ONE fill over the whole block. `ccs`: 3 chunk counts per grid.
static int grid_create_pooled(ivx_ctx* c, const uint32_t* ccs, size_t n, float voxel_extent, ivx_grid** out) {
    for (size_t i = 0; i < n; ++i) out[i] = nullptr;
    if (n == 0) return IVX_OK;
    IVX_HIP_CHECK(hipSetDevice(c->device));
    auto fail = [&](int rc) {
        for (size_t i = 0; i < n; ++i) {
            if (out[i]) {  // (nothing of the blocks is theirs yet)
                out[i]->arena = nullptr, out[i]->arena_block = nullptr, out[i]->result_host = nullptr, out[i]->host_block = nullptr;
                delete out[i];
            }
            out[i] = nullptr;
        }
        return rc;
    };
    std::vector<size_t> off(n + 1, 0);
    for (size_t i = 0; i < n; ++i) {
        int rc = grid_new(c, ccs + 3 * i, voxel_extent, 0, 0, &out[i]);
        if (rc) return fail(rc);
        off[i + 1] = off[i] + ((grid_carve(out[i], nullptr) + 4095) & ~(size_t)4095);
    }
    ivx_block* blk = new (std::nothrow) ivx_block();
    if (!blk) return fail(IVX_ERR_CAPACITY);
    blk->dev = blk->pinned = nullptr;
    blk->refs = 0;
    if (hipMalloc(&blk->dev, off[n]) != hipSuccess || hipHostMalloc(&blk->pinned, n * 256, hipHostMallocMapped) != hipSuccess) {
        ivx_set_error("ivx_copy_polyhedra: allocation of %zu bytes for %zu grids failed", off[n], n);
        blk->refs = 1;
        ivx_block_release(blk);
        return fail(IVX_ERR_HIP);
    }
    memset(blk->pinned, 0, n * 256);
    void* pinned_dev = nullptr;
    if (hipHostGetDevicePointer(&pinned_dev, blk->pinned, 0) != hipSuccess) {
        blk->refs = 1;
        ivx_block_release(blk);
        return fail(IVX_ERR_HIP);
    }
    for (size_t i = 0; i < n; ++i) {
        ivx_grid* g = out[i];
        g->arena = static_cast<char*>(blk->dev) + off[i];
        (void)grid_carve(g, g->arena);
        g->arena_block = blk;
        g->result_host = reinterpret_cast<uint32_t*>(static_cast<char*>(blk->pinned) + i * 256);
        g->result_host_dev = reinterpret_cast<uint32_t*>(static_cast<char*>(pinned_dev) + i * 256);
        g->host_block = blk;
        blk->refs += 2;
    }
    return IVX_OK;
}

static void ivx_edit_state_free(struct ivx_edit_state* e);
void ivx_grid_destroy(ivx_grid* g) {
    if (!g) return;
    (void)ivx_stream_sync(g->ctx->stream);
    ivx_sampler_ahead_free(g);  // (a pre-pass running ahead on the second stream is waited for)
    ivx_edit_state_free(g->edit);
    g->edit = nullptr;
    ivx_submesh_manager_free(g->submesh_manager);  // (host mirrors of the incremental remesh and of the probes' ranges)
    ivx_probe_manager_free(g->probe_manager);
    g->submesh_manager = nullptr, g->probe_manager = nullptr;
    // (everything sized by the chunk counts lives in the arena; the rest grew on demand)
    if (g->dens_call) (void)hipFree(g->dens_call);
    mesh_group_free(g, 1u), mesh_group_free(g, 2u), mesh_group_free(g, 4u);  // (allocations of their own, or parts of a shared block)
    if (g->arena_block) {  // (the arena is a part of a block shared with the grids that came into being with this one)
        ivx_block_release(g->arena_block);
        g->arena = nullptr;
    }
    if (g->host_block) {
        ivx_block_release(g->host_block);
        g->result_host = nullptr;
    }
    void* ptrs[] = {g->arena, g->positions, g->normals, g->indices, g->index_materials, g->vertex_materials, g->submeshes, g->dev_scratch, g->prog_nodes,
                    g->samp_len, g->samp_ops, g->pairs_dev, g->samp_super, g->probe_points, g->probe_chunk, g->probe_entries};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    if (g->host_scratch) (void)hipHostFree(g->host_scratch);
    if (g->result_host) (void)hipHostFree(g->result_host);
    if (g->ev_ready)
        for (int i = 0; i < 2 * IVX_N_TIMED_STAGES; ++i) (void)hipEventDestroy(g->ev[i]);
    delete g;
}

int ivx_grid_upload_dense(ivx_grid* g, const int8_t* sdf, const uint8_t* type, size_t n_voxels) {
    IVX_REQUIRE(g && sdf && type, IVX_ERR_INVALID, "ivx_grid_upload_dense: null argument");
    IVX_REQUIRE(n_voxels == g->n_vox, IVX_ERR_INVALID, "ivx_grid_upload_dense: expected %zu voxels, got %zu", g->n_vox, n_voxels);
    int rc;
    if ((rc = h2d(g, g->sdf, sdf, g->n_vox))) return rc;
    if ((rc = h2d(g, g->type, type, g->n_vox))) return rc;
    if ((rc = ivx_launch_classify(g))) return rc;
    g->mesh_valid = 0;
    g->mesh_built = 0;
    g->occ_ref_valid = 0;
    g->bbox_valid = 0;
    g->regions_valid = 0;
    IVX_HIP_CHECK(ivx_stream_sync(g->ctx->stream));
    return IVX_OK;
}

int ivx_grid_download_dense(ivx_grid* g, int8_t* sdf, uint8_t* type, uint8_t* flags, uint8_t* local_labels, ivx_chunk_info* info, size_t n_voxels) {
    IVX_REQUIRE(g, IVX_ERR_INVALID, "ivx_grid_download_dense: null grid");
    IVX_REQUIRE(n_voxels == g->n_vox, IVX_ERR_INVALID, "ivx_grid_download_dense: expected %zu voxels, got %zu", g->n_vox, n_voxels);
    int rc;
    if ((sdf || type || flags || local_labels) && (rc = ivx_ensure_dense(g))) return rc;  // Void / Uniform chunks are written out on demand
    if (sdf && (rc = d2h(g, sdf, g->sdf, g->n_vox))) return rc;
    if (type && (rc = d2h(g, type, g->type, g->n_vox))) return rc;
    if (flags && (rc = d2h(g, flags, g->flags, g->n_vox))) return rc;
    if (local_labels && (rc = d2h(g, local_labels, g->llabel, g->n_vox))) return rc;
    if (info && (rc = d2h(g, info, g->info, sizeof(ivx_chunk_info) * g->n_chunks))) return rc;
    return IVX_OK;
}

int ivx_grid_chunk_counts(ivx_grid* g, uint32_t out[3]) {
    IVX_REQUIRE(g && out, IVX_ERR_INVALID, "ivx_grid_chunk_counts: null argument");
    for (int d = 0; d < 3; ++d) out[d] = g->cc[d];
    return IVX_OK;
}

int ivx_grid_stage_counters(ivx_grid* g, uint32_t out[4]) {
    IVX_REQUIRE(g && out, IVX_ERR_INVALID, "ivx_grid_stage_counters: null argument");
    for (int i = 0; i < 4; ++i) out[i] = 0;
    int rc;
    if (g->samp_len) {  // chunks evaluated per voxel by the sampler (three lists by LDS class)
        uint32_t ev[3];
        // (the live counters until the derive sweep has rolled them over into their statistics words, role_preset)
        if ((rc = d2h(g, ev, g->samp_len + g->n_chunks + ((g->scratch_dirty & IVX_SCRATCH_EVAL) ? 0 : 8), sizeof(ev)))) return rc;
        out[0] = ev[0] + ev[1] + ev[2];
    }
    if ((rc = d2h(g, &out[1], g->rscalar + 2, sizeof(uint32_t)))) return rc;                            // chunks with several local regions
    if ((rc = d2h(g, &out[2], g->chunk_offsets + 2 * (size_t)g->n_chunks + 2, sizeof(uint32_t)))) return rc;  // chunks that emitted a mesh
    out[3] = g->n_chunks;
    return IVX_OK;
}

void* ivx_grid_device_ptr(ivx_grid* g, int which) {
    if (!g) return nullptr;
    if (which >= 0 && which < 4 && ivx_ensure_dense(g) != IVX_OK) return nullptr;  // whole planes for the caller (stream-ordered)
    if (which == 0 || which == 1 || which == 4) ivx_planes_touched(g);  // (the caller may write through the pointer)
    switch (which) {
        case 0: return g->sdf;
        case 1: return g->type;
        case 2: return g->flags;
        case 3: return g->llabel;
        case 4: return g->info;
        case 5: return g->rparent;
#ifdef IVX_WG_TRACE
        case 6: return g->chunk_moments;
#endif
        case 7: return g->samp_len;  // developer tools (tools/prog_stats.py): per-chunk compact program lengths, then the three list counters and lists
        case 8: return g->samp_ops;  // ... and the programs, OP_CAP (128) uint2 per chunk
        case 10: return g->sn_hard;  // ... and which (entries of the emit list)
        case 9: return ivx_sn_hard_count(g);  // developer tools: [0] how many chunks the mesher's last main pass handed to the general pass
        default: return nullptr;
    }
}

int ivx_sdf_sample(ivx_grid* g, const ivx_sdf_processed_node* nodes, size_t n_nodes, uint32_t stack_size, const uint32_t grid_shape[3],
                   const float shifted_grid_center[3], uint8_t voxel_type) {
    IVX_REQUIRE(g && grid_shape && shifted_grid_center, IVX_ERR_INVALID, "ivx_sdf_sample: null argument");
    IVX_REQUIRE(n_nodes == 0 || nodes, IVX_ERR_INVALID, "ivx_sdf_sample: null node array");
    for (int d = 0; d < 3; ++d) {
        const uint32_t cap = (d == 0 ? g->gx : g->cc[d]) * 16u;
        IVX_REQUIRE(grid_shape[d] <= cap, IVX_ERR_INVALID, "ivx_sdf_sample: grid shape %u exceeds the chunk grid (%u voxels) along axis %d", grid_shape[d], cap, d);
    }
    // validate the node program against the stack the kernel will use
    int depth = 0, max_depth = 0;
    for (size_t i = 0; i < n_nodes; ++i) {
        const uint32_t k = nodes[i].kind;
        IVX_REQUIRE(k <= 9 && k != 6, IVX_ERR_INVALID, "ivx_sdf_sample: unsupported node kind %u", k);
        if (k <= 2) max_depth = std::max(max_depth, ++depth);
        else if (k >= 7) {
            IVX_REQUIRE(depth >= 2, IVX_ERR_INVALID, "ivx_sdf_sample: malformed node program (combination without two operands)");
            --depth;
        } else if (k == 5) {
            IVX_REQUIRE(depth >= 1, IVX_ERR_INVALID, "ivx_sdf_sample: malformed node program (scaling without operand)");
        }
    }
    IVX_REQUIRE(n_nodes == 0 || depth == 1, IVX_ERR_INVALID, "ivx_sdf_sample: malformed node program (final stack depth %d)", depth);
    IVX_REQUIRE((uint32_t)max_depth <= stack_size || n_nodes == 0, IVX_ERR_INVALID, "ivx_sdf_sample: stack_size %u < required %d", stack_size, max_depth);
    int rc;
    if ((rc = ensure_dev_scratch(g, std::max<size_t>(n_nodes, 1) * sizeof(ivx_sdf_processed_node)))) return rc;
    {
        std::vector<ivx_sdf_processed_node> annotated(nodes, nodes + n_nodes);
        ivx_sdf_annotate_host(annotated.data(), n_nodes);
        if ((rc = h2d(g, g->dev_scratch, annotated.data(), n_nodes * sizeof(ivx_sdf_processed_node)))) return rc;
    }
    rc = ivx_launch_sdf_sample(g, static_cast<const ivx_sdf_processed_node*>(g->dev_scratch), (uint32_t)n_nodes, (uint32_t)max_depth, grid_shape,
                               shifted_grid_center, voxel_type);
    if (rc) return rc;
    g->mesh_valid = 0;
    g->mesh_built = 0;
    g->occ_ref_valid = 0;
    g->bbox_valid = 0;
    g->regions_valid = 0;
    IVX_HIP_CHECK(ivx_stream_sync(g->ctx->stream));
    return IVX_OK;
}

int ivx_derive_state(ivx_grid* g) {
    IVX_REQUIRE(g, IVX_ERR_INVALID, "ivx_derive_state: null grid");
    int rc = ivx_launch_derive(g, 0);
    if (rc) return rc;
    g->mesh_valid = 0;
    g->regions_valid = 0;
    IVX_HIP_CHECK(ivx_stream_sync(g->ctx->stream));
    return IVX_OK;
}

int ivx_occupied_ranges(ivx_grid* g, uint32_t out[12]) {
    IVX_REQUIRE(g && out, IVX_ERR_INVALID, "ivx_occupied_ranges: null argument");
    IVX_REQUIRE(g->bbox_valid, IVX_ERR_STATE, "ivx_occupied_ranges: the chunk boxes come from the derive sweep (call ivx_derive_state first)");
    uint32_t* d = g->rscalar + 16;
    int rc;
    if ((rc = ivx_launch_occupied(g, d))) return rc;
    uint32_t raw[12];
    if ((rc = d2h(g, raw, d, 12 * sizeof(uint32_t)))) return rc;
    ivx_occupied_from_raw(g, raw, out);
    memcpy(g->occ_ref, out, sizeof(g->occ_ref));  // VoxelObject::update_occupied_ranges: this is what the object holds from now on
    g->occ_ref_valid = 1;
    return IVX_OK;
}

int ivx_remesh(ivx_grid* g, ivx_mesh_counts* out) {
    IVX_REQUIRE(g && out, IVX_ERR_INVALID, "ivx_remesh: null argument");
    int rc;
    if ((rc = ivx_launch_sn_count(g))) return rc;
    if ((rc = ivx_launch_sn_scan(g))) return rc;
    uint32_t totals[3];
    if ((rc = d2h(g, totals, g->chunk_offsets + 2 * (size_t)g->n_chunks, sizeof(totals)))) return rc;
    if ((rc = ensure_mesh_capacity(g, totals[0], totals[1], totals[2]))) return rc;
    if (totals[1] > 0 && (rc = ivx_launch_sn_emit(g))) return rc;
    g->mesh_counts.n_vertices = totals[0];
    g->mesh_counts.n_indices = totals[1];
    g->mesh_counts.n_submeshes = totals[2];
    g->mesh_counts.reserved = 0;
    g->mesh_valid = 1;
    g->mesh_built = 1;
    g->mesh_serial += 1;
    *out = g->mesh_counts;
    IVX_HIP_CHECK(ivx_stream_sync(g->ctx->stream));
    return IVX_OK;
}

static void ivx_edit_state_free(struct ivx_edit_state* e);
struct ivx_edit_state;
static ivx_edit_state* edit_state(ivx_grid* g);
static bool edit_needs_lookup(ivx_grid* g, uint32_t chunk, uint32_t out3[3]);
static int edit_sync_upload(ivx_grid* g, const void* src, size_t bytes, void* d_dst);
static void edit_sync_mark(ivx_grid* g, int pending);
static int edit_sync_pending(ivx_grid* g);
static bool edit_sync_drained_take(ivx_grid* g);  // (and clears it)

// VoxelObjectMesh::sync_with_voxel_object (mesh.rs:355-456) in two halves. ENQUEUE: the sizes the invalidated chunks' meshes need come from the
// last edit's result block when the set is that edit's (ivx_absorb_collect: no count pass, no read-back) — else from a count over just these
// chunks and one read-back —; the host mirror of the ChunkSubmeshManager places them; records and slots go up from pinned memory and the
// emit pass for the listed chunks follows on the stream. COLLECT: the wait. (Round 3 counted the whole object, read the sizes back, uploaded
// through a blocking copy and waited again: two round trips and ~25 us of counting for ~100 chunks.)
// what the edit in flight invalidates and what those meshes need, from the records its count role delivers early (ivx_edit_state::early_*):
// waits for the role's bell — a third of the way down the edit's chain —, not for the edit
static int edit_early_needs(ivx_grid* g, const char* who, std::vector<uint32_t>& list);
static int edit_sync_stage(ivx_grid* g, const void* src, size_t bytes, void** dev_view);
static int edit_sync_stage_done(ivx_grid* g);
static int mesh_sync_enqueue(ivx_grid* g, const uint8_t* invalidated_chunks, const char* who) {
    IVX_REQUIRE(g, IVX_ERR_INVALID, "%s: null argument", who);
    ivx_many_other_context other_(g->ctx);
    IVX_REQUIRE(!edit_sync_pending(g), IVX_ERR_STATE, "%s: a sync of this object is in flight (ivx_mesh_sync_collect first)", who);
    IVX_REQUIRE(g->mesh_built, IVX_ERR_STATE, "ivx_mesh_sync: there is no mesh to synchronise (call ivx_remesh first)");
    IVX_REQUIRE(g->regions_valid, IVX_ERR_STATE, "ivx_mesh_sync: derived state must be current (the edit ops leave it so)");
    IVX_REQUIRE(g->x_off == 0 && g->gx == g->cc[0] && !g->has_ghost[0] && !g->has_ghost[1], IVX_ERR_STATE,
                "ivx_mesh_sync: not available on a slab of a decomposed grid");
    int rc;
    static const bool trace_ = getenv("IVX_MANY_TRACE") != nullptr;
    static thread_local double acc_[6];
    static thread_local int calls_ = 0;
    auto tp_ = std::chrono::steady_clock::now();
    auto lap_ = [&](int slot) {
        if (!trace_) return;
        const auto t = std::chrono::steady_clock::now();
        acc_[slot] += 1e-3 * (double)std::chrono::duration_cast<std::chrono::nanoseconds>(t - tp_).count();
        tp_ = t;
    };
    if (!g->submesh_manager) g->submesh_manager = new (std::nothrow) ivx_submesh_manager();
    IVX_REQUIRE(g->submesh_manager, IVX_ERR_HIP, "ivx_mesh_sync: out of host memory");
    ivx_submesh_manager* m = g->submesh_manager;
    if (m->serial != g->mesh_serial) {  // the mesh was rebuilt in full since: push_chunk for every submesh, no free ranges (mesh.rs:731-749)
        m->table.assign(g->mesh_counts.n_submeshes, ivx_submesh{});
        if ((rc = d2h(g, m->table.data(), g->submeshes, m->table.size() * sizeof(ivx_submesh)))) return rc;
        m->slot_of.clear();
        for (uint32_t s = 0; s < m->table.size(); ++s) m->slot_of[linear_chunk(g, m->table[s].chunk_indices)] = s;
        m->vertices.free_ranges.clear();
        m->indices.free_ranges.clear();
        m->total_vertices = g->mesh_counts.n_vertices;
        m->total_indices = g->mesh_counts.n_indices;
        m->updated.clear();  // (recreate -> ChunkSubmeshManager::clear, mesh.rs:843-851)
        m->chunks_were_removed = false;
        m->serial = g->mesh_serial;
    }
    lap_(0);
    // what the invalidated chunks' meshes need now and their records (exposure, obscuredness flags)
    // (the call's work lists live on across calls: with a hundred objects per frame their allocations were a third of the host's time here)
    static thread_local std::vector<uint32_t> list, needs, dirty_slots, rec_chunk, slots;
    static thread_local std::vector<char> stage;
    list.clear();
    if (!invalidated_chunks) {  // the set of the edit in flight
        if ((rc = edit_early_needs(g, who, list))) return rc;
    } else {   // chunk-linear order (the reference walks a hash set: unpinned); eight flags a look — nearly all of them are zero
        uint32_t c = 0;
        for (; c + 8u <= g->n_chunks; c += 8u) {
            uint64_t w;
            memcpy(&w, invalidated_chunks + c, 8);
            if (!w) continue;
            for (uint32_t b = 0; b < 8u; ++b)
                if (invalidated_chunks[c + b]) list.push_back(c + b);
        }
        for (; c < g->n_chunks; ++c)
            if (invalidated_chunks[c]) list.push_back(c);
    }
    needs.assign(3 * list.size(), 0u);
    bool cached = g->needs_current != 0;
    for (size_t e = 0; e < list.size() && cached; ++e) cached = edit_needs_lookup(g, list[e], &needs[3 * e]);
    if (!list.empty() && !cached) {
        const size_t off_out = (list.size() * 4 + 15) & ~(size_t)15;
        if ((rc = ensure_dev_scratch(g, off_out + list.size() * 16))) return rc;
        char* base = static_cast<char*>(g->dev_scratch);
        if ((rc = edit_sync_upload(g, list.data(), list.size() * 4, base))) return rc;
        if ((rc = ivx_launch_list_mesh_needs(g, (uint32_t)list.size(), reinterpret_cast<const uint32_t*>(base), reinterpret_cast<uint32_t*>(base + off_out)))) return rc;
        std::vector<uint32_t> raw(4 * list.size());
        if ((rc = d2h(g, raw.data(), base + off_out, raw.size() * 4))) return rc;
        for (size_t e = 0; e < list.size(); ++e) needs[3 * e] = raw[4 * e + 1], needs[3 * e + 1] = raw[4 * e + 2], needs[3 * e + 2] = raw[4 * e] & 0xFFFFu;
    }
    lap_(1);
    dirty_slots.clear();  // slots of the device table that a removal rewrote (the emit pass writes the others)
    const size_t table_before = m->table.size();
    struct Rec {
        uint32_t chunk, voff, ioff, packed;
    };
    static thread_local std::vector<Rec> recs;
    recs.clear();
    rec_chunk.clear();
    for (size_t e = 0; e < list.size(); ++e) {
        const uint32_t c = list[e];
        const uint32_t nv = needs[3 * e], ni = needs[3 * e + 1], kind = needs[3 * e + 2] & 0xFFu, flags = (needs[3 * e + 2] >> 8) & 0xFFu;
        const bool exposed = kind == KIND_NONUNIFORM && (flags & CF_FULLY_OBSCURED) != CF_FULLY_OBSCURED;
        if (!exposed || ni == 0) {  // no longer exposed, or an empty mesh (mesh.rs:375-379, 447-451)
            auto gone = m->slot_of.find(c);
            if (gone != m->slot_of.end()) dirty_slots.push_back(gone->second);  // (the last entry moves here)
            manager_remove(g, m, c);
            continue;
        }
        // write_chunk (mesh.rs:751-809)
        auto it = m->slot_of.find(c);
        if (it != m->slot_of.end()) {
            const ivx_submesh& old = m->table[it->second];
            m->vertices.free_range(old.vertex_offset, (size_t)old.vertex_offset + old.vertex_count);
            m->indices.free_range(old.index_offset, (size_t)old.index_offset + old.index_count);
        }
        size_t v0, i0;
        if (!m->vertices.allocate(nv, &v0)) v0 = m->total_vertices, m->total_vertices += nv;
        if (!m->indices.allocate(ni, &i0)) i0 = m->total_indices, m->total_indices += ni;
        IVX_REQUIRE(m->total_vertices < 0xFFFFFFFFull && m->total_indices < 0xFFFFFFFFull, IVX_ERR_CAPACITY, "ivx_mesh_sync: mesh buffers exceed 2^32 elements");
        ivx_submesh sm;
        memset(&sm, 0, sizeof(sm));
        sm.chunk_indices[0] = c / (g->cc[1] * g->cc[2]);
        sm.chunk_indices[1] = (c / g->cc[2]) % g->cc[1];
        sm.chunk_indices[2] = c % g->cc[2];
        sm.index_offset = (uint32_t)i0;
        sm.index_count = ni;
        sm.vertex_offset = (uint32_t)v0;
        sm.vertex_count = nv;
        for (int a = 0; a < 2; ++a)  // ChunkSubmesh::new (mesh.rs:611-635): bits X_DN 0, Y_DN 1, Z_DN 2, X_UP 3, Y_UP 4, Z_UP 5
            for (int b = 0; b < 2; ++b)
                for (int d = 0; d < 2; ++d)
                    sm.is_obscured_from_direction[a][b][d] =
                        (((flags >> (3 * a)) & 1u) && ((flags >> (3 * b + 1)) & 1u) && ((flags >> (3 * d + 2)) & 1u)) ? 1u : 0u;
        if (it != m->slot_of.end()) {
            m->table[it->second] = sm;
        } else {
            m->slot_of[c] = (uint32_t)m->table.size();
            m->table.push_back(sm);
        }
        m->updated.push_back(ivx_submesh_data_ranges{(uint32_t)v0, (uint32_t)(v0 + nv), (uint32_t)i0, (uint32_t)(i0 + ni)});
        recs.push_back(Rec{c, (uint32_t)v0, (uint32_t)i0, nv | ((ni / 6u) << 16)});
        rec_chunk.push_back(c);
    }
    lap_(2);
    m->vertices.merge_consecutive();  // perform_maintainance
    m->indices.merge_consecutive();
    if ((rc = ensure_mesh_capacity_keep(g, m->total_vertices, m->total_indices, m->table.size()))) return rc;
    lap_(3);
    // the device table: the emit pass writes the re-meshed chunks' entries; entries a removal moved are patched from the host mirror — by the
    // emit pass's first workgroup when there is one (they ride in its upload), else by a small copy each
    size_t n_patch = 0;
    for (uint32_t slot : dirty_slots) n_patch += slot < m->table.size() ? 1u : 0u;
    if (!recs.empty()) {
        // (slots are final only now: a removal after a write may have moved the written entry)
        slots.resize(recs.size());
        for (size_t r = 0; r < recs.size(); ++r) slots[r] = m->slot_of.at(rec_chunk[r]);
        const size_t off_slots = 16 + recs.size() * sizeof(Rec), off_pslots = off_slots + recs.size() * 4;
        const size_t off_pent = (off_pslots + n_patch * 4 + 15) & ~(size_t)15, total = off_pent + n_patch * sizeof(ivx_submesh);
        if ((rc = ensure_dev_scratch(g, total))) return rc;
        stage.assign(total, 0);
        const uint32_t n = (uint32_t)recs.size();
        memcpy(stage.data(), &n, 4);
        memcpy(stage.data() + 16, recs.data(), recs.size() * sizeof(Rec));
        memcpy(stage.data() + off_slots, slots.data(), slots.size() * 4);
        size_t e = 0;
        for (uint32_t slot : dirty_slots)
            if (slot < m->table.size()) {
                memcpy(stage.data() + off_pslots + 4 * e, &slot, 4);
                memcpy(stage.data() + off_pent + e * sizeof(ivx_submesh), &m->table[slot], sizeof(ivx_submesh));
                e += 1;
            }
        // (records, slots and patches are a few KB the passes only read: from host-mapped memory in place — a copy ahead of the passes is a
        // stream operation of 4 us and a gap; a recorded batch carries them in its one staging copy instead)
        char* base = static_cast<char*>(g->dev_scratch);
        void* view = nullptr;
        if ((rc = edit_sync_stage(g, stage.data(), total, &view))) return rc;
        if (view) base = static_cast<char*>(view);
        else if ((rc = edit_sync_upload(g, stage.data(), total, base))) return rc;
        if ((rc = ivx_launch_sn_emit_list(g, n, reinterpret_cast<const uint32_t*>(base), base + 16, reinterpret_cast<const uint32_t*>(base + off_slots), (uint32_t)n_patch,
                                          base + off_pent, reinterpret_cast<const uint32_t*>(base + off_pslots))))
            return rc;
        if (view && (rc = edit_sync_stage_done(g))) return rc;
    }
    lap_(4);
    (void)table_before;
    if (recs.empty())
        for (uint32_t slot : dirty_slots)
            if (slot < m->table.size() && !ivx_many_upload(g->ctx, g, g->submeshes + slot, &m->table[slot], sizeof(ivx_submesh)))
                IVX_HIP_CHECK(ivx_memcpy_async(g->submeshes + slot, &m->table[slot], sizeof(ivx_submesh), hipMemcpyHostToDevice, g->ctx->stream));
    g->mesh_counts.n_vertices = (uint32_t)m->total_vertices;
    g->mesh_counts.n_indices = (uint32_t)m->total_indices;
    g->mesh_counts.n_submeshes = (uint32_t)m->table.size();
    g->mesh_serial += 1;  // collision probes picked from the old mesh are stale
    m->serial = g->mesh_serial;
    edit_sync_mark(g, 1);
    (void)edit_sync_drained_take(g);
    lap_(5);
    static const int period_ = trace_ && atoi(getenv("IVX_MANY_TRACE")) > 1 ? atoi(getenv("IVX_MANY_TRACE")) : 81;
    if (trace_ && ++calls_ % period_ == 0) {
        fprintf(stderr, "[ivx many]   sync enqueue laps (us per period): setup %.1f list+needs %.1f allocator %.1f capacity %.1f upload+emit %.1f tail %.1f\n", acc_[0], acc_[1], acc_[2],
                acc_[3], acc_[4], acc_[5]);
        for (double& a : acc_) a = 0.0;
    }
    return IVX_OK;
}

static int mesh_sync_collect(ivx_grid* g, ivx_mesh_counts* out, const char* who, bool stream_is_drained = false) {
    IVX_REQUIRE(g && out, IVX_ERR_INVALID, "%s: null argument", who);
    IVX_REQUIRE(edit_sync_pending(g), IVX_ERR_STATE, "%s: no sync of this object is in flight", who);
    edit_sync_mark(g, 0);
    if (edit_sync_drained_take(g)) stream_is_drained = true;  // (the edit's collect waited behind this sync's launches)
    if (!stream_is_drained) IVX_HIP_CHECK(ivx_stream_sync(g->ctx->stream));
    g->mesh_valid = 1;
    *out = g->mesh_counts;
    return IVX_OK;
}

int ivx_mesh_sync_enqueue(ivx_grid* g, const uint8_t* invalidated_chunks) { return mesh_sync_enqueue(g, invalidated_chunks, "ivx_mesh_sync_enqueue"); }
int ivx_grid_set_early_mesh_needs(ivx_grid* g, int on) {
    IVX_REQUIRE(g, IVX_ERR_INVALID, "ivx_grid_set_early_mesh_needs: null grid");
    g->early_needs_on = on ? 1 : 0;
    return IVX_OK;
}
int ivx_mesh_sync_collect(ivx_grid* g, ivx_mesh_counts* out) { return mesh_sync_collect(g, out, "ivx_mesh_sync_collect"); }
int ivx_mesh_sync(ivx_grid* g, const uint8_t* invalidated_chunks, ivx_mesh_counts* out) {
    IVX_REQUIRE(out, IVX_ERR_INVALID, "ivx_mesh_sync: null argument");
    const int rc = mesh_sync_enqueue(g, invalidated_chunks, "ivx_mesh_sync");
    return rc ? rc : mesh_sync_collect(g, out, "ivx_mesh_sync");
}

int ivx_mesh_modifications(ivx_grid* g, ivx_submesh_data_ranges* out, size_t cap, size_t* n_out, int* chunks_were_removed) {
    IVX_REQUIRE(g && n_out && chunks_were_removed && (out || cap == 0), IVX_ERR_INVALID, "ivx_mesh_modifications: null argument");
    const ivx_submesh_manager* m = g->submesh_manager;
    const bool current = m && m->serial == g->mesh_serial;  // (a full remesh since the last sync: nothing pending, the whole mesh is new)
    *n_out = current ? m->updated.size() : 0;
    *chunks_were_removed = current && m->chunks_were_removed ? 1 : 0;
    IVX_REQUIRE(*n_out <= cap || !out, IVX_ERR_CAPACITY, "ivx_mesh_modifications: %zu ranges exceed the capacity %zu", *n_out, cap);
    if (out && *n_out) memcpy(out, m->updated.data(), *n_out * sizeof(ivx_submesh_data_ranges));
    return IVX_OK;
}

int ivx_mesh_report_synchronized(ivx_grid* g) {
    IVX_REQUIRE(g, IVX_ERR_INVALID, "ivx_mesh_report_synchronized: null grid");
    if (g->submesh_manager) {
        g->submesh_manager->updated.clear();
        g->submesh_manager->chunks_were_removed = false;
    }
    return IVX_OK;
}

int ivx_mesh_download(ivx_grid* g, float* positions, float* normals, uint32_t* indices, uint8_t* index_materials, ivx_submesh* submeshes) {
    IVX_REQUIRE(g, IVX_ERR_INVALID, "ivx_mesh_download: null grid");
    IVX_REQUIRE(g->mesh_valid, IVX_ERR_STATE, "ivx_mesh_download: call ivx_remesh first");
    const size_t nv = g->mesh_counts.n_vertices, ni = g->mesh_counts.n_indices, ns = g->mesh_counts.n_submeshes;
    int rc;
    if (positions && (rc = d2h(g, positions, g->positions, nv * 12))) return rc;
    if (normals && (rc = d2h(g, normals, g->normals, nv * 12))) return rc;
    if (indices && (rc = d2h(g, indices, g->indices, ni * 4))) return rc;
    if (index_materials && (rc = d2h(g, index_materials, g->index_materials, ni * 8))) return rc;
    if (submeshes && (rc = d2h(g, submeshes, g->submeshes, ns * sizeof(ivx_submesh)))) return rc;
    return IVX_OK;
}

void* ivx_mesh_device_ptr(ivx_grid* g, int which) {
    if (!g || !g->mesh_valid) return nullptr;
    switch (which) {
        case 0: return g->positions;
        case 1: return g->normals;
        case 2: return g->indices;
        case 3: return g->index_materials;
        case 4: return g->submeshes;
        default: return nullptr;
    }
}

// §8f-4: handles of a mesh buffer another process or another API can import (gpu_resource.rs:498-530, 729-907 fill wgpu buffers from these
// arrays; mesh.rs:94-123). The buffers are plain hipMalloc allocations of their own, so the IPC handle names exactly the buffer.
int ivx_mesh_export(ivx_grid* g, int which, ivx_mesh_export_info* out) {
    IVX_REQUIRE(g && out && which >= 0 && which <= 4, IVX_ERR_INVALID, "ivx_mesh_export: bad argument");
    IVX_REQUIRE(g->mesh_valid || g->mesh_built, IVX_ERR_STATE, "ivx_mesh_export: the grid has no mesh");
    memset(out, 0, sizeof(*out));
    out->dmabuf_fd = -1;
    void* p = nullptr;
    size_t elem = 0, count = 0, cap = 0;
    switch (which) {
        case 0: p = g->positions, elem = 12, count = g->mesh_counts.n_vertices, cap = g->vcap; break;
        case 1: p = g->normals, elem = 12, count = g->mesh_counts.n_vertices, cap = g->vcap; break;
        case 2: p = g->indices, elem = 4, count = g->mesh_counts.n_indices, cap = g->icap; break;
        case 3: p = g->index_materials, elem = 8, count = g->mesh_counts.n_indices, cap = g->icap; break;
        default: p = g->submeshes, elem = sizeof(ivx_submesh), count = g->mesh_counts.n_submeshes, cap = g->scap; break;
    }
    if (g->submesh_manager) {  // after an incremental sync the live ranges are scattered over the buffers: everything up to the capacity may be in use
        count = cap;
    }
    IVX_REQUIRE(p && cap, IVX_ERR_STATE, "ivx_mesh_export: the buffer is empty");
    out->bytes = (uint64_t)count * elem;
    out->capacity_bytes = (uint64_t)cap * elem;
    out->element_bytes = (uint32_t)elem;
    out->generation = g->mesh_generation;
    out->device_ptr = (uint64_t)(uintptr_t)p;
    hipIpcMemHandle_t h;
    static_assert(sizeof(h) == sizeof(out->ipc_handle), "hipIpcMemHandle_t is 64 bytes");
    IVX_HIP_CHECK(hipIpcGetMemHandle(&h, p));
    memcpy(out->ipc_handle, &h, sizeof(h));
    // a dma-buf file descriptor for APIs that import those (Vulkan VK_EXT_external_memory_dma_buf under wgpu): optional, -1 where the
    // runtime cannot make one; the caller owns the descriptor (close it)
    int fd = -1;
    const size_t page = 4096;
    // The HSA runtime's export names the offset of the range inside the dma-buf object (the HIP call drops it): taken from there when the
    // symbol is in the process (it is wherever libamdhip64 is: libhsa-runtime64 is its dependency), page-aligned range around the buffer.
    typedef int (*export_fn)(const void*, size_t, int*, uint64_t*);
    static export_fn hsa_export = reinterpret_cast<export_fn>(dlsym(RTLD_DEFAULT, "hsa_amd_portable_export_dmabuf"));
    const uintptr_t lo = (uintptr_t)p & ~(uintptr_t)(page - 1), hi = ((uintptr_t)p + (size_t)out->capacity_bytes + page - 1) & ~(uintptr_t)(page - 1);
    uint64_t off = 0;
    if (hsa_export && hsa_export(reinterpret_cast<const void*>(lo), hi - lo, &fd, &off) == 0 && fd >= 0) {
        out->dmabuf_fd = fd;
        out->dmabuf_offset = off + ((uintptr_t)p - lo);
        out->dmabuf_bytes = hi - lo;
    } else if (hipMemGetHandleForAddressRange(&fd, reinterpret_cast<void*>(lo), hi - lo, hipMemRangeHandleTypeDmaBufFd, 0) == hipSuccess && fd >= 0) {
        out->dmabuf_fd = fd;
        out->dmabuf_offset = (uintptr_t)p - lo;  // (the range's own start: this call does not say where the range lies in the object)
        out->dmabuf_bytes = hi - lo;
    } else {
        const hipError_t e = hipGetLastError();
        ivx_set_error("ivx_mesh_export: no dma-buf descriptor for the buffer (%s); the hipIpc handle is valid", hipGetErrorString(e));
    }
    return IVX_OK;
}

// the importing side for a HIP process: the 64-byte handle of ivx_mesh_export -> a device pointer in THIS process (no context needed beyond
// the device being usable); ivx_mesh_import_close gives it back
int ivx_mesh_import_open(const uint8_t ipc_handle[64], int device, void** device_ptr) {
    IVX_REQUIRE(ipc_handle && device_ptr, IVX_ERR_INVALID, "ivx_mesh_import_open: null argument");
    IVX_HIP_CHECK(hipSetDevice(device));
    hipIpcMemHandle_t h;
    memcpy(&h, ipc_handle, sizeof(h));
    IVX_HIP_CHECK(hipIpcOpenMemHandle(device_ptr, h, hipIpcMemLazyEnablePeerAccess));
    return IVX_OK;
}
int ivx_mesh_import_close(void* device_ptr) {
    IVX_REQUIRE(device_ptr, IVX_ERR_INVALID, "ivx_mesh_import_close: null argument");
    IVX_HIP_CHECK(hipIpcCloseMemHandle(device_ptr));
    return IVX_OK;
}

int ivx_mesh_generation(ivx_grid* g, uint64_t* generation) {
    IVX_REQUIRE(g && generation, IVX_ERR_INVALID, "ivx_mesh_generation: null argument");
    *generation = g->mesh_generation;
    return IVX_OK;
}

// a density table for ONE call: the resident copy when it is the resident table (ivx_grid_set_densities), else a copy of its own behind the
// resident one — the object's resident table is what its steps and edits use and is never replaced on the side (round 3 overwrote the device
// copy here and left the host mirror saying otherwise: an edit handed the resident table again then ran on this call's)
static int densities_on_device(ivx_grid* g, const float densities[256], const float** d_out) {
    if (g->has_dens && memcmp(g->dens_host, densities, sizeof(g->dens_host)) == 0) {
        *d_out = g->dens_dev;
        return IVX_OK;
    }
    if (!g->dens_call) IVX_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&g->dens_call), 256 * sizeof(float)));
    int rc = h2d(g, g->dens_call, densities, 256 * sizeof(float));
    if (rc) return rc;
    *d_out = g->dens_call;
    return IVX_OK;
}

int ivx_inertia(ivx_grid* g, const float densities[256], ivx_moments* out) {
    IVX_REQUIRE(g && densities && out, IVX_ERR_INVALID, "ivx_inertia: null argument");
    double* out_dev = g->partials + g->partial_blocks * 10;
    int rc;
    const float* d_dens = nullptr;
    if ((rc = densities_on_device(g, densities, &d_dens))) return rc;
    if ((rc = ivx_launch_inertia(g, d_dens, out_dev, 0))) return rc;
    if ((rc = d2h(g, out->m64, out_dev, 10 * sizeof(double)))) return rc;
    for (int i = 0; i < 10; ++i) out->m32[i] = (float)out->m64[i];
    out->reserved[0] = out->reserved[1] = 0;
    return IVX_OK;
}

int ivx_label_regions(ivx_grid* g, uint32_t* region_count) {
    IVX_REQUIRE(g && region_count, IVX_ERR_INVALID, "ivx_label_regions: null argument");
    int rc;
    if ((rc = ivx_launch_ccl_local(g, 0))) return rc;
    if ((rc = ivx_launch_ccl_merge(g))) return rc;
    if ((rc = ivx_launch_ccl_resolve(g))) return rc;
    uint32_t sc[2];
    if ((rc = d2h(g, sc, g->rscalar, sizeof(sc)))) return rc;
    IVX_REQUIRE((sc[1] & 1u) == 0, IVX_ERR_CAPACITY,
                "ivx_label_regions: a chunk has more than 254 local regions (the reference's CHUNK_MAX_REGIONS limit, split_detection.rs:145-157)");
    g->region_count = sc[0];
    g->regions_valid = 1;
    *region_count = sc[0];
    return IVX_OK;
}

int ivx_region_labels_download(ivx_grid* g, uint32_t* labels, size_t n_voxels) {
    IVX_REQUIRE(g && labels, IVX_ERR_INVALID, "ivx_region_labels_download: null argument");
    IVX_REQUIRE(g->regions_valid, IVX_ERR_STATE, "ivx_region_labels_download: call ivx_label_regions first");
    IVX_REQUIRE(n_voxels == g->n_vox, IVX_ERR_INVALID, "ivx_region_labels_download: expected %zu voxels, got %zu", g->n_vox, n_voxels);
    int rc;
    if ((rc = ensure_dev_scratch(g, g->n_vox * sizeof(uint32_t)))) return rc;
    if ((rc = ivx_launch_ccl_dense_labels(g, static_cast<uint32_t*>(g->dev_scratch)))) return rc;
    return d2h(g, labels, g->dev_scratch, g->n_vox * sizeof(uint32_t));
}

static int describe_regions_internal(ivx_grid* g, const float* d_dens, std::vector<ivx_region_desc>& out) {
    const uint32_t n = g->region_count;
    out.assign(n, ivx_region_desc{});
    if (n == 0) return IVX_OK;
    const size_t bytes = ivx_region_stats_bytes(n);
    int rc;
    if ((rc = ensure_dev_scratch(g, bytes))) return rc;
    if ((rc = ivx_launch_region_stats(g, d_dens, g->dev_scratch, n))) return rc;
    std::vector<char> h(bytes);
    if ((rc = d2h(g, h.data(), g->dev_scratch, bytes))) return rc;
    const char* p = h.data();
    const unsigned long long* count = reinterpret_cast<const unsigned long long*>(p);
    p += 8 * (size_t)n;
    const double* mom = reinterpret_cast<const double*>(p);
    p += 80 * (size_t)n;
    const uint32_t* lo = reinterpret_cast<const uint32_t*>(p);
    p += 12 * (size_t)n;
    const uint32_t* hi = reinterpret_cast<const uint32_t*>(p);
    p += 12 * (size_t)n;
    const uint32_t* nu = reinterpret_cast<const uint32_t*>(p);
    p += 4 * (size_t)n;
    const uint32_t* ch = reinterpret_cast<const uint32_t*>(p);
    p += 4 * (size_t)n;
    const uint32_t* root = reinterpret_cast<const uint32_t*>(p);
    const double e = (double)g->extent, e2 = e * e, e3 = e2 * e, e4 = e2 * e2, e5 = e4 * e;
    for (uint32_t r = 0; r < n; ++r) {
        ivx_region_desc& o = out[r];
        o.root_chunk = root[r] >> 8;
        o.root_region = root[r] & 255u;
        o.voxel_count = count[r];
        for (int d3 = 0; d3 < 3; ++d3) {
            o.lo[d3] = lo[3 * r + d3];
            o.hi[d3] = hi[3 * r + d3];
        }
        o.non_uniform_chunk_count = nu[r];
        o.chunk_count = ch[r];
        for (int q = 0; q < 10; ++q) {
            const double f = q == 0 ? e3 : (q <= 3 ? 0.5 * e4 : (q <= 6 ? (1.0 / 3.0) * e5 : 0.25 * e5));
            o.moments[q] = mom[10 * (size_t)r + q] * f;
        }
    }
    return IVX_OK;
}

int ivx_regions_describe(ivx_grid* g, const float densities[256], ivx_region_desc* out, size_t cap, size_t* n_out) {
    IVX_REQUIRE(g && densities && out && n_out, IVX_ERR_INVALID, "ivx_regions_describe: null argument");
    IVX_REQUIRE(g->regions_valid, IVX_ERR_STATE, "ivx_regions_describe: call ivx_label_regions first");
    const uint32_t n = g->region_count;
    *n_out = n;
    IVX_REQUIRE(n <= cap, IVX_ERR_CAPACITY, "ivx_regions_describe: %u regions exceed capacity %zu", n, cap);
    int rc;
    const float* d_dens = nullptr;
    if ((rc = densities_on_device(g, densities, &d_dens))) return rc;
    std::vector<ivx_region_desc> d;
    if ((rc = describe_regions_internal(g, d_dens, d))) return rc;
    for (uint32_t r = 0; r < n; ++r) out[r] = d[r];
    return IVX_OK;
}

// Derived state + regions of an object whose voxels changed, through the fused step path (ivx_voxel_step_enqueue: five launches and the
// results block instead of the stand-alone passes' ten launches and a blocking copy). `rederive_enqueue` only puts the work on the
// stream — a caller with results of its own still in flight waits for both with the one `rederive_collect`.
int ivx_voxel_step_enqueue(ivx_grid* g, uint32_t stages);
int ivx_voxel_step_collect(ivx_grid* g, ivx_step_result* out);
static int rederive_enqueue(ivx_grid* g) {
    const uint32_t keep = g->stage_timing_off;
    g->stage_timing_off = 0xFFFFFFFFu;  // (no event records around the slots: nobody reads this call's stage times)
    const int rc = ivx_voxel_step_enqueue(g, IVX_STAGE_DERIVE | IVX_STAGE_REGIONS);
    g->stage_timing_off = keep;
    return rc;
}
static int rederive_collect(ivx_grid* g) {
    ivx_step_result res;
    const int rc = ivx_voxel_step_collect(g, &res);
    if (rc) return rc;
    g->mesh_valid = 0;
    return IVX_OK;
}
static int rederive(ivx_grid* g) {
    const int rc = rederive_enqueue(g);
    return rc ? rc : rederive_collect(g);
}

int ivx_split_off_smallest_region(ivx_grid* parent, ivx_grid** child, uint32_t origin_offset_in_parent[3], int* outcome, ivx_region_desc* moved) {
    IVX_REQUIRE(parent && child && origin_offset_in_parent && outcome, IVX_ERR_INVALID, "ivx_split_off_smallest_region: null argument");
    *child = nullptr;
    *outcome = 0;
    IVX_REQUIRE(parent->regions_valid, IVX_ERR_STATE, "ivx_split_off_smallest_region: call ivx_label_regions first");
    IVX_REQUIRE(parent->x_off == 0 && parent->gx == parent->cc[0] && !parent->has_ghost[0] && !parent->has_ghost[1], IVX_ERR_STATE,
                "ivx_split_off_smallest_region: not available on a slab of a decomposed grid");
    if (parent->region_count < 2) return IVX_OK;
    int rc;
    if (!parent->has_dens) {
        float ones[256];
        for (float& x : ones) x = 1.0f;
        if ((rc = h2d(parent, parent->dens_dev, ones, sizeof(ones)))) return rc;
        memcpy(parent->dens_host, ones, sizeof(ones));
        parent->has_dens = 1;
    }
    std::vector<ivx_region_desc> d;
    if ((rc = describe_regions_internal(parent, parent->dens_dev, d))) return rc;
    // the first two regions in scan order; the one with fewer non-uniform chunks goes, ties by chunk count,
    // then the second (extraction.rs:255-271)
    uint32_t pick;
    if (d[0].non_uniform_chunk_count != d[1].non_uniform_chunk_count) pick = d[0].non_uniform_chunk_count < d[1].non_uniform_chunk_count ? 0u : 1u;
    else pick = d[0].chunk_count < d[1].chunk_count ? 0u : 1u;
    const ivx_region_desc& r = d[pick];
    if (moved) *moved = r;
    uint32_t lo[3], cc[3];
    for (int q = 0; q < 3; ++q) {
        lo[q] = r.lo[q] >> 4;
        cc[q] = ((r.hi[q] - 1u) >> 4) - lo[q] + 1u;
    }
    const uint32_t uniform_count = r.chunk_count - r.non_uniform_chunk_count;
    const bool discard = uniform_count == 0 && r.voxel_count < 8;  // NON_EMPTY_VOXEL_THRESHOLD (object.rs:203)
    ivx_grid* c = nullptr;
    if (!discard && (rc = ivx_grid_create(parent->ctx, cc, parent->extent, 0, 0, &c))) return rc;
    if ((rc = ivx_launch_split_move(parent, c, lo, cc, pick))) {
        ivx_grid_destroy(c);
        return rc;
    }
    for (int q = 0; q < 3; ++q) origin_offset_in_parent[q] = lo[q] * 16u;
    if (c && cc[0] <= 2 && cc[1] <= 2 && cc[2] <= 2 && uniform_count == 0 && cc[0] * cc[1] * cc[2] > 1 && r.hi[0] - r.lo[0] <= 14 &&
        r.hi[1] - r.lo[1] <= 14 && r.hi[2] - r.lo[2] <= 14) {
        uint32_t off[3];
        for (int q = 0; q < 3; ++q) {
            const uint32_t rel = r.lo[q] - lo[q] * 16u;
            off[q] = rel > 0 ? rel - 1u : 0u;
        }
        const uint32_t one[3] = {1, 1, 1};
        ivx_grid* single = nullptr;
        if ((rc = ivx_grid_create(parent->ctx, one, parent->extent, 0, 0, &single))) {
            ivx_grid_destroy(c);
            return rc;
        }
        if ((rc = ivx_launch_split_repack(c, single, off))) {
            ivx_grid_destroy(c);
            ivx_grid_destroy(single);
            return rc;
        }
        ivx_grid_destroy(c);
        c = single;
        for (int q = 0; q < 3; ++q) origin_offset_in_parent[q] += off[q];
    }
    if ((parent->occ_ref_valid = 0, rc = rederive(parent))) {
        ivx_grid_destroy(c);
        return rc;
    }
    if (c && (rc = rederive(c))) {
        ivx_grid_destroy(c);
        return rc;
    }
    *child = c;
    *outcome = c ? 1 : 2;
    return IVX_OK;
}

static int step_enqueue(ivx_grid* g, uint32_t stages, const uint16_t* slab_nbr_ids, void* slab_record, uint32_t part = 3u);
static int ivx_step_collect_launch(ivx_grid* g);
// The reference's split-off LOOP in one call (interaction.rs:256: `while let Some(..) = find_two_disconnected_regions` ->
// extract the smaller of the FIRST TWO regions in scan order, extraction.rs:255-271): the regions of the object are described once; what a
// region is — voxels, box, chunk counts — does not change when another region leaves (regions share no voxel, and a chunk that holds two of
// them stays NonUniform for the one that remains), nor does their scan order, so the loop's picks follow from the one description: the host
// plays the loop over the descriptors, every region that goes is moved out by its own launch into a grid from one shared block, the parent
// is re-derived ONCE and the children together (recorded, many.hpp). `children` / `origins3` / `outcomes` / `moved` in the order the loop
// extracts them (outcome 1: a child object, 2: discarded as a crumb); *n_out = number of split-offs (regions - 1).
int ivx_split_off_all(ivx_grid* parent, size_t cap, ivx_grid** children, uint32_t* origins3, int* outcomes, ivx_region_desc* moved, size_t* n_out) {
    IVX_REQUIRE(parent && n_out && (cap == 0 || (children && origins3 && outcomes)), IVX_ERR_INVALID, "ivx_split_off_all: null argument");
    *n_out = 0;
    IVX_REQUIRE(parent->regions_valid, IVX_ERR_STATE, "ivx_split_off_all: call ivx_label_regions first");
    IVX_REQUIRE(parent->x_off == 0 && parent->gx == parent->cc[0] && !parent->has_ghost[0] && !parent->has_ghost[1], IVX_ERR_STATE,
                "ivx_split_off_all: not available on a slab of a decomposed grid");
    if (parent->region_count < 2) return IVX_OK;
    const size_t n = parent->region_count - 1u;
    *n_out = n;
    IVX_REQUIRE(n <= cap, IVX_ERR_CAPACITY, "ivx_split_off_all: %zu split-offs exceed capacity %zu", n, cap);
    int rc;
    hipStream_t s = parent->ctx->stream;
    float ones[256];
    for (float& x : ones) x = 1.0f;
    if (!parent->has_dens) {
        if ((rc = h2d(parent, parent->dens_dev, ones, sizeof(ones)))) return rc;
        memcpy(parent->dens_host, ones, sizeof(ones));
        parent->has_dens = 1;
    }
    std::vector<ivx_region_desc> d;
    if ((rc = describe_regions_internal(parent, parent->dens_dev, d))) return rc;
    // the loop over the descriptors: `front` is the first region still there, `second` the next one
    std::vector<uint32_t> order;
    order.reserve(n);
    {
        uint32_t front = 0;
        for (uint32_t second = 1; second < (uint32_t)d.size(); ++second) {
            const ivx_region_desc &a = d[front], &b = d[second];
            bool take_first;
            if (a.non_uniform_chunk_count != b.non_uniform_chunk_count) take_first = a.non_uniform_chunk_count < b.non_uniform_chunk_count;
            else take_first = a.chunk_count < b.chunk_count;
            if (take_first) {
                order.push_back(front);
                front = second;
            } else {
                order.push_back(second);
            }
        }
    }
    // the children's boxes and grids (crumbs get none: their voxels are just emptied)
    std::vector<uint32_t> ccs, los;
    std::vector<size_t> slot(n, (size_t)-1);
    std::vector<char> repack(n, 0);
    for (size_t k = 0; k < n; ++k) {
        const ivx_region_desc& r = d[order[k]];
        if (moved) moved[k] = r;
        children[k] = nullptr;
        const uint32_t uniform_count = r.chunk_count - r.non_uniform_chunk_count;
        uint32_t lo[3], cc[3];
        for (int q = 0; q < 3; ++q) {
            lo[q] = r.lo[q] >> 4;
            cc[q] = ((r.hi[q] - 1u) >> 4) - lo[q] + 1u;
            origins3[3 * k + q] = lo[q] * 16u;
        }
        const bool discard = uniform_count == 0 && r.voxel_count < 8;  // NON_EMPTY_VOXEL_THRESHOLD (object.rs:203)
        outcomes[k] = discard ? 2 : 1;
        if (!discard) {
            slot[k] = ccs.size() / 3;
            repack[k] = cc[0] <= 2 && cc[1] <= 2 && cc[2] <= 2 && uniform_count == 0 && cc[0] * cc[1] * cc[2] > 1 && r.hi[0] - r.lo[0] <= 14 && r.hi[1] - r.lo[1] <= 14 &&
                        r.hi[2] - r.lo[2] <= 14;
        }
        for (int q = 0; q < 3; ++q) {
            if (!discard) ccs.push_back(cc[q]);
            los.push_back(lo[q]);
        }
    }
    const size_t n_kids = ccs.size() / 3;
    std::vector<ivx_grid*> kids(n_kids, nullptr);
    if ((rc = grid_create_pooled(parent->ctx, ccs.data(), n_kids, parent->extent, kids.data()))) return rc;
    auto fail = [&](int code) {
        (void)ivx_stream_sync(s);
        for (ivx_grid*& c : kids)
            if (c) {
                c->pending_stages = 0, c->gather_launched = 0;
                ivx_grid_destroy(c);
                c = nullptr;
            }
        for (size_t k = 0; k < n; ++k) children[k] = nullptr;
        parent->regions_valid = 0;  // (voxels may have left: the caller derives the object again)
        return code;
    };
    // every region that goes, by its own launch (they read the labelling the parent has now; none of them changes it)
    for (size_t k = 0; k < n; ++k) {
        ivx_grid* c = slot[k] == (size_t)-1 ? nullptr : kids[slot[k]];
        uint32_t cc[3];
        if (c) memcpy(cc, c->cc, sizeof(cc));
        else {
            const ivx_region_desc& r = d[order[k]];
            for (int q = 0; q < 3; ++q) cc[q] = ((r.hi[q] - 1u) >> 4) - los[3 * k + q] + 1u;
        }
        if ((rc = ivx_launch_split_move(parent, c, &los[3 * k], cc, order[k]))) return fail(rc);
    }
    // small children into one chunk (complete_extracted_voxel_object, extraction.rs:1902-2142)
    for (size_t k = 0; k < n; ++k) {
        if (!repack[k]) continue;
        ivx_grid*& c = kids[slot[k]];
        const ivx_region_desc& r = d[order[k]];
        uint32_t off[3];
        for (int q = 0; q < 3; ++q) {
            const uint32_t rel = r.lo[q] - los[3 * k + q] * 16u;
            off[q] = rel > 0 ? rel - 1u : 0u;
        }
        const uint32_t one[3] = {1, 1, 1};
        ivx_grid* single = nullptr;
        if ((rc = ivx_grid_create(parent->ctx, one, parent->extent, 0, 0, &single))) return fail(rc);
        if ((rc = ivx_launch_split_repack(c, single, off))) {
            ivx_grid_destroy(single);
            return fail(rc);
        }
        ivx_grid_destroy(c);  // (waits for the stream: the repack has read its source)
        c = single;
        for (int q = 0; q < 3; ++q) origins3[3 * k + q] += off[q];
    }
    // derived state and regions: the parent and every child, recorded and issued together; one wait
    parent->occ_ref_valid = 0;
    std::vector<ivx_grid*> all(kids);
    all.push_back(parent);
    auto enqueue_one = [&](size_t i) -> int {
        ivx_grid* g = all[i];
        if (g != parent && !g->arena_block) return IVX_OK;  // (a repacked child: own allocation, zeroed at creation)
        if (g != parent && !ivx_many_zero(g->ctx, g, g->work_counts, 8 * sizeof(uint32_t))) IVX_HIP_CHECK(ivx_memset_async(g->work_counts, 0, 8 * sizeof(uint32_t), s));
        return IVX_OK;
    };
    auto derive_one = [&](size_t i) -> int {
        int r = enqueue_one(i);
        if (r) return r;
        if ((r = rederive_enqueue(all[i]))) return r;
        return ivx_step_collect_launch(all[i]);
    };
    if (ivx_many_recording()) {
        (void)ivx_many_break();
        for (size_t i = 0; i < all.size(); ++i)
            if ((rc = derive_one(i))) return fail(rc);
    } else {
        if ((rc = ivx_many_begin(parent->ctx))) return fail(rc);
        int first = IVX_OK;
        for (size_t i = 0; i < all.size() && !first; ++i) {
            ivx_many_object((uint32_t)i);
            first = derive_one(i);
        }
        rc = ivx_many_flush(parent->ctx);
        if (first || rc) return fail(first ? first : rc);
    }
    for (ivx_grid* g : all)
        if ((rc = rederive_collect(g))) return fail(rc);
    for (size_t k = 0; k < n; ++k)
        if (slot[k] != (size_t)-1) children[k] = kids[slot[k]];
    return IVX_OK;
}

// complete_extracted_voxel_object (extraction.rs:1901-2123) for a freshly filled child grid: discard rule, single-chunk
// repack, derived state. On return *pc is the final child (or nullptr when discarded).
static int complete_extracted(ivx_grid* parent, ivx_grid** pc, uint32_t origin[3]) {
    ivx_grid* c = *pc;
    int rc;
    std::vector<ivx_chunk_info> info(c->n_chunks);
    if ((rc = d2h(c, info.data(), c->info, sizeof(ivx_chunk_info) * c->n_chunks))) return rc;
    uint32_t uniform_count = 0;
    for (const ivx_chunk_info& i : info) uniform_count += i.gen_kind == KIND_UNIFORM;
    // non-empty voxel count and tight voxel box of the child: derive (flags + per-chunk boxes), unit-density mass
    if ((rc = ivx_launch_derive(c, 0))) return rc;
    uint32_t* d_occ = c->rscalar + 16;
    if ((rc = ivx_launch_occupied(c, d_occ))) return rc;
    uint32_t occ[12], occ_raw[12];
    if ((rc = d2h(c, occ_raw, d_occ, sizeof(occ_raw)))) return rc;
    ivx_occupied_from_raw(c, occ_raw, occ);
    float ones[256];
    for (float& x : ones) x = 1.0f;
    if ((rc = h2d(c, c->dens_dev, ones, sizeof(ones)))) return rc;
    memcpy(c->dens_host, ones, sizeof(ones));
    c->has_dens = 1;
    double* out_dev = c->partials + c->partial_blocks * 10;
    if ((rc = ivx_launch_inertia(c, c->dens_dev, out_dev, 0))) return rc;
    double m0 = 0.0;
    if ((rc = d2h(c, &m0, out_dev, sizeof(double)))) return rc;
    const double e = (double)c->extent;
    const unsigned long long non_empty = (unsigned long long)(m0 / (e * e * e) + 0.5);
    if (uniform_count == 0 && non_empty < 8) {  // NON_EMPTY_VOXEL_THRESHOLD (object.rs:203)
        ivx_grid_destroy(c);
        *pc = nullptr;
        return IVX_OK;
    }
    if (c->cc[0] <= 2 && c->cc[1] <= 2 && c->cc[2] <= 2 && uniform_count == 0 && c->n_chunks > 1 && occ[1] != 0 && occ[7] - occ[6] <= 14 &&
        occ[9] - occ[8] <= 14 && occ[11] - occ[10] <= 14) {
        uint32_t off[3];
        for (int q = 0; q < 3; ++q) off[q] = occ[6 + 2 * q] > 0 ? occ[6 + 2 * q] - 1u : 0u;
        const uint32_t one[3] = {1, 1, 1};
        ivx_grid* single = nullptr;
        if ((rc = ivx_grid_create(parent->ctx, one, parent->extent, 0, 0, &single))) return rc;
        if ((rc = ivx_launch_split_repack(c, single, off))) {
            ivx_grid_destroy(single);
            return rc;
        }
        ivx_grid_destroy(c);
        c = single;
        *pc = c;
        for (int q = 0; q < 3; ++q) origin[q] += off[q];
    }
    return rederive(c);
}

int ivx_clip_polyhedron(ivx_grid* parent, const float* planes4, size_t n_planes, const float aabb[6], int copy, ivx_grid** child,
                        uint32_t origin_offset_in_parent[3], int* outcome) {
    IVX_REQUIRE(parent && planes4 && aabb && child && origin_offset_in_parent && outcome, IVX_ERR_INVALID, "ivx_clip_polyhedron: null argument");
    *child = nullptr;
    *outcome = 0;
    IVX_REQUIRE(n_planes >= 1 && n_planes <= 64, IVX_ERR_CAPACITY, "ivx_clip_polyhedron: 1..64 planes supported, got %zu", n_planes);
    IVX_REQUIRE(parent->regions_valid, IVX_ERR_STATE, "ivx_clip_polyhedron: the object needs its derived state (ivx_derive_state + ivx_label_regions)");
    IVX_REQUIRE(parent->x_off == 0 && parent->gx == parent->cc[0] && !parent->has_ghost[0] && !parent->has_ghost[1], IVX_ERR_STATE,
                "ivx_clip_polyhedron: not available on a slab of a decomposed grid");
    int rc;
    // voxel_ranges_in_object_touching_aab (object/intersection.rs:693-782) of the AABB expanded by 2.54
    uint32_t* d_occ = parent->rscalar + 16;
    if ((rc = ivx_launch_occupied(parent, d_occ))) return rc;
    uint32_t occ[12], occ_raw[12];
    if ((rc = d2h(parent, occ_raw, d_occ, sizeof(occ_raw)))) return rc;
    ivx_occupied_from_raw(parent, occ_raw, occ);
    if (occ[1] == 0) return IVX_OK;
    uint32_t lo[3], cc[3];
    for (int q = 0; q < 3; ++q) {
        const float l = aabb[q] - 2.54f, h = aabb[3 + q] + 2.54f;
        const float fl = floorf(l);
        const long s = (long)(fl > 0.0f ? fl : 0.0f), e = (long)ceilf(h);
        const long vlo = std::max<long>((long)occ[6 + 2 * q], s), vhi = std::min<long>((long)occ[7 + 2 * q], std::max<long>(e, 0));
        if (vlo >= vhi) return IVX_OK;
        lo[q] = (uint32_t)(vlo / 16);
        cc[q] = (uint32_t)((vhi + 15) / 16) - lo[q];
    }
    ivx_grid* c = nullptr;
    if ((rc = ivx_grid_create(parent->ctx, cc, parent->extent, 0, 0, &c))) return rc;
    if ((rc = ivx_launch_clip(parent, c, lo, cc, planes4, (uint32_t)n_planes, copy ? 0 : 1))) {
        ivx_grid_destroy(c);
        return rc;
    }
    for (int q = 0; q < 3; ++q) origin_offset_in_parent[q] = lo[q] * 16u;
    if (!copy && (parent->occ_ref_valid = 0, rc = rederive(parent))) {
        ivx_grid_destroy(c);
        return rc;
    }
    if ((rc = complete_extracted(parent, &c, origin_offset_in_parent))) {
        if (c) ivx_grid_destroy(c);
        return rc;
    }
    *child = c;
    *outcome = c ? 1 : 2;
    return IVX_OK;
}

// Batched polyhedron COPY: every fragment of one impact in one call (FracturingProcess::execute_in_parallel, fracturing.rs:1047-1189, runs
// copy_polyhedron_with_property_computer, extraction.rs:1301-1768, for all Voronoi cells of an impact over a thread pool; the object itself
// is not changed). The looped form pays per fragment: an occupied-range reduction with a host read, the child's record download, three more
// host reads for its voxel count / box / regions. Here the parent's ranges are reduced once, all clip kernels and all children's derive /
// range / voxel-count passes are enqueued back to back and read with ONE wait, the discard / repack decisions are taken on the host, then
// all region passes follow with a second wait. Per fragment the results are those of ivx_clip_polyhedron(copy = 1).
int ivx_copy_polyhedra(ivx_grid* parent, const float* planes4, const uint32_t* plane_counts, const float* aabbs6, size_t n_sets, ivx_grid** children,
                       uint32_t* origins3, int* outcomes) {
    IVX_REQUIRE(parent && planes4 && plane_counts && aabbs6 && children && origins3 && outcomes, IVX_ERR_INVALID, "ivx_copy_polyhedra: null argument");
    IVX_REQUIRE(parent->regions_valid, IVX_ERR_STATE, "ivx_copy_polyhedra: the object needs its derived state (ivx_derive_state + ivx_label_regions)");
    IVX_REQUIRE(parent->x_off == 0 && parent->gx == parent->cc[0] && !parent->has_ghost[0] && !parent->has_ghost[1], IVX_ERR_STATE,
                "ivx_copy_polyhedra: not available on a slab of a decomposed grid");
    for (size_t f = 0; f < n_sets; ++f) {
        children[f] = nullptr;
        outcomes[f] = 0;
        IVX_REQUIRE(plane_counts[f] >= 1 && plane_counts[f] <= 64, IVX_ERR_CAPACITY, "ivx_copy_polyhedra: 1..64 planes per polyhedron, set %zu has %u", f, plane_counts[f]);
    }
    if (n_sets == 0) return IVX_OK;
    int rc;
    hipStream_t s = parent->ctx->stream;
    uint32_t* d_occ = parent->rscalar + 16;
    if ((rc = ivx_launch_occupied(parent, d_occ))) return rc;
    uint32_t occ[12], occ_raw[12];
    if ((rc = d2h(parent, occ_raw, d_occ, sizeof(occ_raw)))) return rc;
    ivx_occupied_from_raw(parent, occ_raw, occ);
    if (occ[1] == 0) return IVX_OK;
    // 1. the children's chunk boxes; their grids from ONE device block and ONE pinned block (grid_create_pooled)
    std::vector<uint32_t> live, ccs, los;
    size_t plane_off = 0;
    std::vector<size_t> plane_offs(n_sets);
    for (size_t f = 0; f < n_sets; plane_off += plane_counts[f], ++f) {
        plane_offs[f] = plane_off;
        const float* aabb = aabbs6 + 6 * f;
        uint32_t lo[3], cc[3];
        bool hit = true;
        for (int q = 0; q < 3 && hit; ++q) {
            const float l = aabb[q] - 2.54f, h = aabb[3 + q] + 2.54f;
            const float fl = floorf(l);
            const long st = (long)(fl > 0.0f ? fl : 0.0f), e = (long)ceilf(h);
            const long vlo = std::max<long>((long)occ[6 + 2 * q], st), vhi = std::min<long>((long)occ[7 + 2 * q], std::max<long>(e, 0));
            if (vlo >= vhi) hit = false;
            else {
                lo[q] = (uint32_t)(vlo / 16);
                cc[q] = (uint32_t)((vhi + 15) / 16) - lo[q];
            }
        }
        if (!hit) continue;
        live.push_back((uint32_t)f);
        for (int q = 0; q < 3; ++q) ccs.push_back(cc[q]), los.push_back(lo[q]), origins3[3 * f + q] = lo[q] * 16u;
    }
    const size_t n_live = live.size();
    if (n_live == 0) return IVX_OK;
    std::vector<ivx_grid*> kids(n_live, nullptr);
    if ((rc = grid_create_pooled(parent->ctx, ccs.data(), n_live, parent->extent, kids.data()))) return rc;
    auto fail = [&](int code) {
        (void)ivx_stream_sync(s);
        for (ivx_grid*& c : kids)
            if (c) {
                c->pending_stages = 0, c->gather_launched = 0;
                ivx_grid_destroy(c);
                c = nullptr;
            }
        for (size_t f = 0; f < n_sets; ++f) children[f] = nullptr, outcomes[f] = 0;
        return code;
    };
    if ((rc = ivx_ensure_dense(parent))) return fail(rc);  // (what the clips read; ahead of the recording: it may launch)
    // 2. per child, RECORDED (many.hpp) and issued as one launch per chain position for all of them: the clip, then a step of the child without
    // the sample and remesh stages — derived state, regions, occupied ranges, unit-density mass (= voxel count) — and the gather of its small
    // results into its host-mapped block. One wait for all.
    float ones[256];
    for (float& x : ones) x = 1.0f;
    const uint32_t child_stages = IVX_STAGE_DERIVE | IVX_STAGE_REGIONS | IVX_STAGE_OCCUPIED | IVX_STAGE_INERTIA;
    auto enqueue_child = [&](size_t i) -> int {
        ivx_grid* c = kids[i];
        const size_t f = live[i];
        int r;
        if (!ivx_many_zero(c->ctx, c, c->work_counts, 8 * sizeof(uint32_t))) IVX_HIP_CHECK(ivx_memset_async(c->work_counts, 0, 8 * sizeof(uint32_t), s));
        if ((r = ivx_launch_clip(parent, c, &los[3 * i], &ccs[3 * i], planes4 + 4 * plane_offs[f], plane_counts[f], 0))) return r;
        if (!ivx_many_upload(c->ctx, c, c->dens_dev, ones, sizeof(ones))) IVX_HIP_CHECK(ivx_memcpy_async(c->dens_dev, ones, sizeof(ones), hipMemcpyHostToDevice, s));
        memcpy(c->dens_host, ones, sizeof(ones));
        c->has_dens = 1;
        const uint32_t keep = c->stage_timing_off;
        c->stage_timing_off = 0xFFFFFFFFu;  // (no event records: they would cut the merged launches between every two children)
        r = step_enqueue(c, child_stages, nullptr, nullptr);
        c->stage_timing_off = keep;
        if (r) return r;
        return ivx_step_collect_launch(c);
    };
    if (ivx_many_recording()) {  // (inside somebody else's bracket: in order on the stream, unmerged)
        (void)ivx_many_break();
        for (size_t i = 0; i < n_live; ++i)
            if ((rc = enqueue_child(i))) return fail(rc);
    } else {
        if ((rc = ivx_many_begin(parent->ctx))) return fail(rc);
        int first = IVX_OK;
        for (size_t i = 0; i < n_live && !first; ++i) {
            ivx_many_object((uint32_t)i);
            first = enqueue_child(i);
        }
        rc = ivx_many_flush(parent->ctx);
        if (first || rc) return fail(first ? first : rc);
    }
    std::vector<ivx_step_result> res(n_live);
    for (size_t i = 0; i < n_live; ++i)
        if ((rc = ivx_voxel_step_collect(kids[i], &res[i]))) return fail(rc);
    // 3. discard crumbs, repack small children into one chunk (complete_extracted_voxel_object, extraction.rs:1902-2142)
    for (size_t i = 0; i < n_live; ++i) {
        ivx_grid* c = kids[i];
        const size_t f = live[i];
        const uint32_t* cocc = res[i].occupied;
        const double e = (double)c->extent;
        const unsigned long long non_empty = (unsigned long long)(res[i].moments.m64[0] / (e * e * e) + 0.5);
        // (a chunk filled with one type — gen_kind Uniform — holds 4096 voxels and spans 16 along every axis: neither test below can pass with
        // one, which is what the reference's `uniform_chunk_count == 0` conditions say)
        if (non_empty < 8) {  // NON_EMPTY_VOXEL_THRESHOLD (object.rs:203)
            ivx_grid_destroy(c);
            kids[i] = nullptr;
            outcomes[f] = 2;
            continue;
        }
        if (c->cc[0] <= 2 && c->cc[1] <= 2 && c->cc[2] <= 2 && c->n_chunks > 1 && cocc[1] != 0 && cocc[7] - cocc[6] <= 14 && cocc[9] - cocc[8] <= 14 &&
            cocc[11] - cocc[10] <= 14) {
            uint32_t off[3];
            for (int q = 0; q < 3; ++q) off[q] = cocc[6 + 2 * q] > 0 ? cocc[6 + 2 * q] - 1u : 0u;
            const uint32_t one[3] = {1, 1, 1};
            ivx_grid* single = nullptr;
            if ((rc = ivx_grid_create(parent->ctx, one, parent->extent, 0, 0, &single))) return fail(rc);
            if ((rc = ivx_launch_split_repack(c, single, off))) {
                ivx_grid_destroy(single);
                return fail(rc);
            }
            ivx_grid_destroy(c);  // (waits for the stream: the repack has read its source)
            c = kids[i] = single;
            for (int q = 0; q < 3; ++q) origins3[3 * f + q] += off[q];
            ivx_step_result again;
            if ((rc = ivx_grid_set_densities(single, ones)) || (rc = ivx_voxel_step(single, child_stages, &again))) return fail(rc);
        }
        c->mesh_valid = 0;
        children[f] = c;
        outcomes[f] = 1;
    }
    return IVX_OK;
}

size_t ivx_region_face_bytes(ivx_grid* g) { return g ? (size_t)g->cc[1] * g->cc[2] * 256 * sizeof(uint16_t) : 0; }

int ivx_region_face_labels(ivx_grid* g, int side, void* device_buf) {
    IVX_REQUIRE(g && device_buf && (side == 0 || side == 1), IVX_ERR_INVALID, "ivx_region_face_labels: bad argument");
    IVX_REQUIRE(g->regions_valid, IVX_ERR_STATE, "ivx_region_face_labels: call ivx_label_regions first");
    int rc = ivx_launch_face_ids(g, side, static_cast<uint16_t*>(device_buf));
    if (rc) return rc;
    IVX_HIP_CHECK(ivx_stream_sync(g->ctx->stream));
    return IVX_OK;
}

int ivx_region_face_pairs(ivx_grid* g, int side, const void* neighbour_face_labels, uint32_t* pairs, size_t cap, size_t* n_out) {
    IVX_REQUIRE(g && neighbour_face_labels && n_out && (side == 0 || side == 1) && (pairs || cap == 0), IVX_ERR_INVALID,
                "ivx_region_face_pairs: bad argument");
    IVX_REQUIRE(g->regions_valid, IVX_ERR_STATE, "ivx_region_face_pairs: call ivx_label_regions first");
    const size_t face = (size_t)g->cc[1] * g->cc[2] * 256;
    int rc;
    if ((rc = ensure_dev_scratch(g, 16 + face * 8))) return rc;
    uint32_t* d_count = static_cast<uint32_t*>(g->dev_scratch);
    void* d_pairs = static_cast<char*>(g->dev_scratch) + 16;
    if ((rc = ivx_launch_face_pairs(g, side, static_cast<const uint16_t*>(neighbour_face_labels), d_count, d_pairs, (uint32_t)face, nullptr))) return rc;
    uint32_t n = 0;
    if ((rc = d2h(g, &n, d_count, sizeof(n)))) return rc;
    std::vector<uint64_t> h(n);
    if ((rc = d2h(g, h.data(), d_pairs, (size_t)n * 8))) return rc;
    // uint2{a,b} little-endian -> key a (low word) | b (high word); order by (a, b)
    for (auto& k : h) k = (k << 32) | (k >> 32);
    std::sort(h.begin(), h.end());
    h.erase(std::unique(h.begin(), h.end()), h.end());
    *n_out = h.size();
    IVX_REQUIRE(h.size() <= cap, IVX_ERR_CAPACITY, "ivx_region_face_pairs: %zu pairs exceed capacity %zu", h.size(), cap);
    for (size_t i = 0; i < h.size(); ++i) {
        pairs[2 * i] = (uint32_t)(h[i] >> 32);
        pairs[2 * i + 1] = (uint32_t)(h[i] & 0xFFFFFFFFu);
    }
    return IVX_OK;
}

// ---- the edit path (§8f-2): ONE wait per edit, and only the chunks the edit can have changed are swept -------------------------------------
// ivx_absorb_*_enqueue puts on the stream: the edit kernel over the touched chunk box; the derive sweep over that box grown by one chunk
// each way (ivx_launch_derive_box: what the reference patches chunk by chunk, object/intersection.rs:255-262, 532-598) with the region
// forest of all other chunks reset; the global region resolve (interaction/absorption.rs:631); the count pass of the remesh for the chunks
// whose meshes the edit invalidates (ivx_launch_box_mesh_needs); one copy of all small results into pinned memory. ivx_absorb_collect waits
// for the doorbell behind them. Round 3 swept the whole object's active list (5 388 chunks of the 512^3 body for a bite that touches 80)
// and ivx_mesh_sync counted the whole object again and read its sizes back.
struct ivx_edit_state {
    // the edit in flight
    int pending = 0, nothing = 0, staged = 0;
    uint32_t lo[3] = {0, 0, 0}, cc[3] = {0, 0, 0}, blo[3] = {0, 0, 0}, bcc[3] = {0, 0, 0};
    size_t off_type = 0, off_cnt = 0, off_touch = 0, off_needs = 0, total = 0;
    void* pinned = nullptr;  // results of the edit in flight: host-mapped, written by the step's gather launch ahead of the doorbell
    void* pinned_dev = nullptr;
    size_t pinned_bytes = 0;
    char* d_results = nullptr;  // the edit's accumulators on the device (its own allocation: zero between edits — cleared behind every collect)
    size_t d_results_bytes = 0;
    // what the meshes of the last edit's invalidated chunks need (chunk -> vertices, indices, kind | flags << 8): valid until voxels change again
    // (ascending by chunk — the grown box is walked in chunk-linear order —, looked up by binary search: a hash map's insertions were most of an
    // edit's collect for the small objects of a many-object frame)
    std::vector<std::array<uint32_t, 4>> needs;  // chunk, vertices, indices, kind | flags << 8
    // early delivery of what the invalidated meshes need (ivx_mesh_sync_enqueue with a null set while the edit is in flight): the role that
    // counts them writes its records into the pinned block behind the results and rings a bell there
    int early_armed = 0, early_taken = 0;
    uint32_t early_seq = 0;
    size_t off_early = 0, off_bell = 0;
    // the sync in flight
    int sync_pending = 0;
    void* pinned_up = nullptr;
    void* pinned_up_dev = nullptr;  // the device's view of it (host-mapped: the sync's kernels read their small lists in place)
    size_t pinned_up_bytes = 0;
    int sync_drained = 0;  // an edit's collect has waited for a doorbell that was rung behind this sync's launches: nothing left to wait for
    hipEvent_t up_done = nullptr;  // the last upload from pinned_up has been read
    int up_busy = 0;
};
static void ivx_edit_state_free(ivx_edit_state* e) {
    if (!e) return;
    if (e->pinned) (void)hipHostFree(e->pinned);
    if (e->d_results) (void)hipFree(e->d_results);
    if (e->pinned_up) (void)hipHostFree(e->pinned_up);
    if (e->up_done) (void)hipEventDestroy(e->up_done);
    delete e;
}
static ivx_edit_state* edit_state(ivx_grid* g) {
    if (!g->edit) g->edit = new (std::nothrow) ivx_edit_state();
    return g->edit;
}
static bool edit_needs_lookup(ivx_grid* g, uint32_t chunk, uint32_t out3[3]) {
    if (!g->edit) return false;
    const auto& v = g->edit->needs;
    const auto it = std::lower_bound(v.begin(), v.end(), chunk, [](const std::array<uint32_t, 4>& a, uint32_t c) { return a[0] < c; });
    if (it == v.end() || (*it)[0] != chunk) return false;
    out3[0] = (*it)[1], out3[1] = (*it)[2], out3[2] = (*it)[3];
    return true;
}
static void edit_sync_mark(ivx_grid* g, int pending) {
    if (ivx_edit_state* e = edit_state(g)) e->sync_pending = pending;
}
static int edit_sync_pending(ivx_grid* g) { return g->edit ? g->edit->sync_pending : 0; }
static bool edit_sync_drained_take(ivx_grid* g) {
    if (!g->edit) return false;
    const bool d = g->edit->sync_drained != 0;
    g->edit->sync_drained = 0;
    return d;
}
static int ensure_pinned(void** p, size_t* have, size_t bytes);
// host -> device from the sync's own pinned block, asynchronously (the block is free again once the event behind the copy has passed)
static int edit_sync_upload(ivx_grid* g, const void* src, size_t bytes, void* d_dst) {
    if (ivx_many_upload(g->ctx, g, d_dst, src, (bytes + 3) & ~(size_t)3)) return IVX_OK;  // (a batch is being recorded: the words ride in its staging copy)
    ivx_edit_state* e = edit_state(g);
    IVX_REQUIRE(e, IVX_ERR_CAPACITY, "ivx_mesh_sync: out of host memory");
    if (!e->up_done) IVX_HIP_CHECK(hipEventCreateWithFlags(&e->up_done, hipEventDisableTiming));
    if (e->up_busy) {
        IVX_HIP_CHECK(hipEventSynchronize(e->up_done));
        e->up_busy = 0;
    }
    int rc = ensure_pinned(&e->pinned_up, &e->pinned_up_bytes, bytes);
    if (rc) return rc;
    memcpy(e->pinned_up, src, bytes);
    IVX_HIP_CHECK(ivx_memcpy_async(d_dst, e->pinned_up, bytes, hipMemcpyHostToDevice, g->ctx->stream));
    IVX_HIP_CHECK(ivx_event_record(e->up_done, g->ctx->stream));
    e->up_busy = 1;
    return IVX_OK;
}
static int ensure_pinned(void** p, size_t* have, size_t bytes) {
    if (*have >= bytes) return IVX_OK;
    if (*p) (void)hipHostFree(*p);
    *p = nullptr;
    *have = 0;
    const size_t cap = std::max<size_t>(bytes, 1 << 16);
    IVX_HIP_CHECK(hipHostMalloc(p, cap, hipHostMallocMapped));
    *have = cap;
    return IVX_OK;
}
// The sync's small lists where its kernels can read them WITHOUT a copy on the stream: into the pinned block, *dev_view = the device's address
// of it. Null view: a batch is being recorded (the caller uploads as before). edit_sync_stage_done goes behind the last reader's launch.
static int edit_sync_stage(ivx_grid* g, const void* src, size_t bytes, void** dev_view) {
    *dev_view = nullptr;
    if (ivx_many_recording()) return IVX_OK;
    ivx_edit_state* e = edit_state(g);
    IVX_REQUIRE(e, IVX_ERR_CAPACITY, "ivx_mesh_sync: out of host memory");
    if (!e->up_done) IVX_HIP_CHECK(hipEventCreateWithFlags(&e->up_done, hipEventDisableTiming));
    if (e->up_busy) {
        IVX_HIP_CHECK(hipEventSynchronize(e->up_done));
        e->up_busy = 0;
    }
    const size_t had = e->pinned_up_bytes;
    int rc = ensure_pinned(&e->pinned_up, &e->pinned_up_bytes, bytes);
    if (rc) return rc;
    if (e->pinned_up_bytes != had || !e->pinned_up_dev) IVX_HIP_CHECK(hipHostGetDevicePointer(&e->pinned_up_dev, e->pinned_up, 0));
    memcpy(e->pinned_up, src, bytes);
    *dev_view = e->pinned_up_dev;
    return IVX_OK;
}
static int edit_sync_stage_done(ivx_grid* g) {
    // (no event behind the readers: the block is next written by the object's next sync, which cannot be enqueued before this one's collect has
    // waited for them — an event record here is a stream operation between the sync's launches and whatever follows them)
    (void)g;
    return IVX_OK;
}

static int absorb_enqueue(ivx_grid* g, const char* who, int capsule, const float center[3], const float seg[3], float influence_radius, float shape_radius,
                          const float densities[256]) {
    IVX_REQUIRE(g && center && densities && (seg || !capsule), IVX_ERR_INVALID, "%s: null argument", who);
    ivx_many_other_context other_(g->ctx);
    IVX_REQUIRE(influence_radius >= 0.0f && shape_radius >= 0.0f, IVX_ERR_INVALID, "%s: negative radius", who);
    IVX_REQUIRE(g->regions_valid, IVX_ERR_STATE, "%s: derived state and regions must be current (ivx_derive_state + ivx_label_regions)", who);
    IVX_REQUIRE(g->x_off == 0 && g->gx == g->cc[0] && !g->has_ghost[0] && !g->has_ghost[1], IVX_ERR_STATE, "%s: not available on a slab of a decomposed grid",
                who);
    ivx_edit_state* e = edit_state(g);
    IVX_REQUIRE(e, IVX_ERR_CAPACITY, "%s: out of host memory", who);
    IVX_REQUIRE(!e->pending, IVX_ERR_STATE, "%s: an edit of this object is in flight (ivx_absorb_collect first)", who);
    int rc;
    // the touched voxel ranges start from the object's occupied ranges (voxel_ranges_touching_aab, intersection.rs:766-782)
    uint32_t occ[12];
    if ((rc = reference_occupied(g, occ))) return rc;
    int32_t vlo[3], vhi[3];
    uint32_t lo[3], cc[3];
    e->nothing = 0;
    for (int d = 0; d < 3; ++d) {
        float a = center[d] - influence_radius, b = center[d] + influence_radius;  // Sphere::compute_aabb
        if (capsule) {  // Capsule::compute_aabb: the boxes of the two end spheres (capsule.rs:132-137)
            const float end = center[d] + seg[d];
            const float a1 = end - influence_radius, b1 = end + influence_radius;
            a = a1 < a ? a1 : a;
            b = b1 > b ? b1 : b;
        }
        const float fl = std::floor(a), ce = std::ceil(b);
        // `as usize` saturates at 0; the occupied ranges bound the other side
        const long s_ = fl > 0.0f ? (fl < 2.0e9f ? (long)fl : 2000000000L) : 0, e_ = ce > 0.0f ? (ce < 2.0e9f ? (long)ce : 2000000000L) : 0;
        vlo[d] = (int32_t)std::max<long>((long)occ[6 + 2 * d], s_);
        vhi[d] = (int32_t)std::min<long>((long)occ[7 + 2 * d], e_);
        if (vlo[d] >= vhi[d]) {
            e->nothing = 1;  // nothing touched: the collect reports zeros
            e->pending = 1;
            return IVX_OK;
        }
        lo[d] = (uint32_t)vlo[d] / 16u;
        cc[d] = ((uint32_t)vhi[d] + 15u) / 16u - lo[d];
    }
    uint32_t blo[3], bcc[3];  // the box the derive sweep goes over: one chunk more each way
    for (int d = 0; d < 3; ++d) {
        blo[d] = lo[d] > 0u ? lo[d] - 1u : 0u;
        const uint32_t bhi = std::min(lo[d] + cc[d] + 1u, g->cc[d]);
        bcc[d] = bhi - blo[d];
        e->lo[d] = lo[d], e->cc[d] = cc[d], e->blo[d] = blo[d], e->bcc[d] = bcc[d];
    }
    // scratch: [10 f64 removed moments][256 u32 by type][2 u32 counters][pad][u32 touched ranges of the box's chunks][uint4 needs of the grown box's]
    const size_t box_chunks = (size_t)cc[0] * cc[1] * cc[2], grown = (size_t)bcc[0] * bcc[1] * bcc[2];
    e->off_type = 80, e->off_cnt = e->off_type + 1024, e->off_touch = e->off_cnt + 16;
    e->off_needs = (e->off_touch + box_chunks * 4 + 15) & ~(size_t)15;
    e->total = e->off_needs + grown * 16;
    hipStream_t s = g->ctx->stream;
    // (a density table other than the resident one rides in the block's LAST kilobyte — not right behind this edit's results: nothing clears it
    // there, and a later, larger edit would read its floats as the touched words of its box: 1.0f is "touched, voxels 0..0, 0..0, 0..8" — five
    // chunks invalidated for nothing in one of 3 600 random edit sequences, profiles/round5/README.md)
    const size_t need_bytes = ((e->total + 255) & ~(size_t)255) + 1024;
    if (e->d_results_bytes < need_bytes) {  // (grown on demand; a fresh block starts zeroed, later ones are cleared behind every collect)
        IVX_HIP_CHECK(ivx_stream_sync(s));
        if (e->d_results) (void)hipFree(e->d_results);
        e->d_results = nullptr, e->d_results_bytes = 0;
        const size_t cap = std::max<size_t>(2 * need_bytes, 1 << 16);
        IVX_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&e->d_results), cap));
        IVX_HIP_CHECK(ivx_memset_async(e->d_results, 0, cap, s));
        e->d_results_bytes = cap;
    }
    char* base = e->d_results;
    const size_t off_dens = e->d_results_bytes - 1024;
    const float* d_dens = g->dens_dev;
    if (!(g->has_dens && memcmp(g->dens_host, densities, sizeof(g->dens_host)) == 0)) {  // another table than the resident one
        if ((rc = h2d(g, base + off_dens, densities, 1024))) return rc;
        d_dens = reinterpret_cast<const float*>(base + off_dens);
    }
    // the edit; its first block also zeroes the region scalars the sweep behind it starts from
    if ((rc = ivx_launch_absorb(g, capsule, lo, cc, vlo, vhi, center, seg, influence_radius, shape_radius, d_dens, reinterpret_cast<double*>(base),
                                reinterpret_cast<uint32_t*>(base + e->off_type), reinterpret_cast<uint32_t*>(base + e->off_cnt),
                                reinterpret_cast<uint32_t*>(base + e->off_touch), g->rscalar)))
        return rc;
    e->needs.clear();  // (voxels change: what an earlier edit's chunks needed is history)
    // derived state and chunk-local regions of the grown box, the other chunks' region nodes reset; then the global resolve, whose first
    // launch also counts what the invalidated chunks' meshes need, and whose last (the gather of ivx_absorb_collect) copies the edit's small
    // results to the host ahead of the doorbell: seven launches, no copy or fill operation on the stream
    if ((rc = ivx_launch_derive_box(g, IVX_PART_REGIONS, blo, bcc, nullptr))) return rc;
    e->staged = e->total <= STAGED_COPY_MAX;
    e->off_early = (e->total + 63) & ~(size_t)63;
    e->off_bell = e->off_early + grown * 16;
    const size_t pinned_need = e->off_bell + 64;
    if (e->staged && e->pinned_bytes < pinned_need) {
        IVX_HIP_CHECK(ivx_stream_sync(s));
        if (e->pinned) (void)hipHostFree(e->pinned);
        e->pinned = e->pinned_dev = nullptr, e->pinned_bytes = 0;
        const size_t cap = std::max<size_t>(2 * pinned_need, 1 << 16);
        IVX_HIP_CHECK(hipHostMalloc(&e->pinned, cap, hipHostMallocMapped));
        IVX_HIP_CHECK(hipHostGetDevicePointer(&e->pinned_dev, e->pinned, 0));
        memset(e->pinned, 0, cap);
        e->pinned_bytes = cap;
    }
    for (int d = 0; d < 3; ++d) g->post1_needs_box[d] = lo[d], g->post1_needs_box[3 + d] = cc[d], g->post1_needs_box[6 + d] = blo[d], g->post1_needs_box[9 + d] = bcc[d];
    g->post1_needs_touched = reinterpret_cast<const uint32_t*>(base + e->off_touch);
    g->post1_needs_out = reinterpret_cast<uint32_t*>(base + e->off_needs);
    e->early_armed = 0, e->early_taken = 0;
    if (e->staged && g->early_needs_on) {  // (the words [2], [3] behind the two counters are the block's spare: zero between edits like the rest)
        e->early_seq += 1u;
        g->post1_needs_early = reinterpret_cast<uint32_t*>(static_cast<char*>(e->pinned_dev) + e->off_early);
        g->post1_needs_counter = reinterpret_cast<uint32_t*>(base + e->off_cnt) + 2;
        g->post1_needs_bell = reinterpret_cast<uint32_t*>(static_cast<char*>(e->pinned_dev) + e->off_bell);
        g->post1_needs_seq = e->early_seq;
        // (the bell's place moves with the size of the box: whatever an earlier, larger edit left there must not read as this edit's number)
        *reinterpret_cast<volatile uint32_t*>(static_cast<char*>(e->pinned) + e->off_bell) = 0u;
        e->early_armed = 1;
    }
    {
        const uint32_t keep = g->stage_timing_off;
        g->stage_timing_off = 0xFFFFFFFFu;       // (no event records around the slots: nobody reads this call's stage times)
        g->preset_fresh |= IVX_SCRATCH_REGIONS;  // (the region scalars were zeroed by the edit kernel, before the box sweep listed its multi-region chunks)
        g->regions_labelled_locally = 1;         // (... and the sweep labelled the box: no stand-alone local pass in front of the resolve)
        rc = ivx_voxel_step_enqueue(g, IVX_STAGE_REGIONS);
        g->regions_labelled_locally = 0;
        g->stage_timing_off = keep;
        if (rc) return rc;
    }
    if (e->staged) {
        g->gather_copy_src = reinterpret_cast<const uint32_t*>(base);
        g->gather_copy_dst = static_cast<uint32_t*>(e->pinned_dev);
        // (with early delivery the needs records are in the host-mapped block already: the gather copies the words in front of them only)
        g->gather_copy_words = (uint32_t)(((e->early_armed ? e->off_needs : e->total) + 3) / 4);
    }
    e->pending = 1;
    return IVX_OK;
}

static int absorb_collect(ivx_grid* g, const char* who, ivx_absorb_result* out, uint32_t* emptied_by_type, uint8_t* invalidated_chunks) {
    IVX_REQUIRE(g && out, IVX_ERR_INVALID, "%s: null argument", who);
    ivx_many_other_context other_(g->ctx);
    ivx_edit_state* e = g->edit;
    IVX_REQUIRE(e && e->pending, IVX_ERR_STATE, "%s: no edit of this object is in flight", who);
    e->pending = 0;
    memset(out, 0, sizeof(*out));
    if (emptied_by_type) memset(emptied_by_type, 0, 256 * sizeof(uint32_t));
    if (invalidated_chunks) memset(invalidated_chunks, 0, g->n_chunks);
    if (e->nothing) return IVX_OK;
    int rc;
    if ((rc = rederive_collect(g))) return rc;  // (the doorbell behind everything enqueued: one wait)
    if (e->sync_pending) e->sync_drained = 1;   // (... a sync enqueued while this edit was in flight included)
    std::vector<char> hostbuf;
    const char* hb;
    if (e->staged) {
        hb = static_cast<const char*>(e->pinned);
    } else {
        hostbuf.resize(e->total);
        if ((rc = d2h(g, hostbuf.data(), e->d_results, e->total))) return rc;
        hb = hostbuf.data();
    }
    // (the accumulators and the touched words start the next edit from zero — and that edit's box may be larger than this one's: everything this
    // edit wrote is cleared now, off the next edit's path; the block beyond has never been written)
    if (!ivx_many_zero(g->ctx, g, e->d_results, (e->total + 3) & ~(size_t)3)) IVX_HIP_CHECK(ivx_memset_async(e->d_results, 0, e->total, g->ctx->stream));
    const double* rem = reinterpret_cast<const double*>(hb);
    const double ex = (double)g->extent, e3 = ex * ex * ex, e4 = e3 * ex, e5 = e4 * ex;
    const double f[10] = {e3, 0.5 * e4, 0.5 * e4, 0.5 * e4, e5 / 3.0, e5 / 3.0, e5 / 3.0, 0.25 * e5, 0.25 * e5, 0.25 * e5};
    for (int q = 0; q < 10; ++q) out->removed_moments[q] = rem[q] * f[q];
    const uint32_t* by_type = reinterpret_cast<const uint32_t*>(hb + e->off_type);
    uint64_t emptied = 0;
    for (int t = 0; t < 256; ++t) {
        emptied += by_type[t];
        if (emptied_by_type) emptied_by_type[t] = by_type[t];
    }
    out->emptied_voxels = emptied;
    const uint32_t* cnt = reinterpret_cast<const uint32_t*>(hb + e->off_cnt);
    out->touched_chunks = cnt[0];
    out->removed_chunks = cnt[1];
    if (cnt[1]) g->occ_ref_valid = 0;  // `if removed_chunks { self.update_occupied_ranges() }` (intersection.rs:384-386, 520-522)
    // handle_chunk_voxels_modified (intersection.rs:560-598): the touched chunk, and a neighbour when the touched voxel range of the chunk
    // comes within two voxels of the face they share — decided on the device per chunk of the grown box, with what its mesh needs now
    const uint32_t* needs = reinterpret_cast<const uint32_t*>(e->early_armed ? static_cast<const char*>(e->pinned) + e->off_early : hb + e->off_needs);
    const size_t grown = (size_t)e->bcc[0] * e->bcc[1] * e->bcc[2];
    e->needs.clear();  // (an early sync may have filled them from the same records)
    for (size_t b = 0; b < grown; ++b) {
        const uint32_t w = needs[4 * b];
        if (!(w & 0x80000000u)) continue;
        const uint32_t c = needs[4 * b + 3];
        if (invalidated_chunks) invalidated_chunks[c] = 1;
        e->needs.push_back({c, needs[4 * b + 1], needs[4 * b + 2], w & 0xFFFFu});
    }
    if (!std::is_sorted(e->needs.begin(), e->needs.end(), [](const std::array<uint32_t, 4>& a, const std::array<uint32_t, 4>& b) { return a[0] < b[0]; }))
        std::sort(e->needs.begin(), e->needs.end(), [](const std::array<uint32_t, 4>& a, const std::array<uint32_t, 4>& b) { return a[0] < b[0]; });
    g->needs_current = 1;
    return IVX_OK;
}

static int edit_early_needs(ivx_grid* g, const char* who, std::vector<uint32_t>& list) {
    ivx_edit_state* e = g->edit;
    IVX_REQUIRE(e && e->pending, IVX_ERR_STATE, "%s: a null set stands for the chunks of an edit in flight, and none is", who);
    if (e->nothing) return IVX_OK;
    IVX_REQUIRE(e->early_armed, IVX_ERR_STATE,
                "%s: the edit in flight does not deliver its mesh needs early (ivx_grid_set_early_mesh_needs before the edit; or its results do not fit the "
                "host-mapped block): collect it and pass its set",
                who);
    if (!e->early_taken) {
        const volatile uint32_t* bell = reinterpret_cast<const volatile uint32_t*>(static_cast<const char*>(e->pinned) + e->off_bell);
        (void)ivx_many_break();
        bool rung = false;
        const auto t0 = std::chrono::steady_clock::now();
        for (uint32_t it = 0;; ++it) {
            if (*bell == e->early_seq) {
                rung = true;
                break;
            }
            __builtin_ia32_pause();
            if ((it & 255u) == 255u && std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() > 2000) break;
        }
        if (!rung) IVX_HIP_CHECK(ivx_stream_sync(g->ctx->stream));
        std::atomic_thread_fence(std::memory_order_acquire);
        IVX_REQUIRE(*bell == e->early_seq, IVX_ERR_HIP, "%s: the edit's mesh needs never arrived", who);
        const uint32_t* needs = reinterpret_cast<const uint32_t*>(static_cast<const char*>(e->pinned) + e->off_early);
        const size_t grown = (size_t)e->bcc[0] * e->bcc[1] * e->bcc[2];
        e->needs.clear();
        for (size_t b = 0; b < grown; ++b) {
            const uint32_t w = needs[4 * b];
            if (!(w & 0x80000000u)) continue;
            e->needs.push_back({needs[4 * b + 3], needs[4 * b + 1], needs[4 * b + 2], w & 0xFFFFu});
        }
        if (!std::is_sorted(e->needs.begin(), e->needs.end(), [](const std::array<uint32_t, 4>& a, const std::array<uint32_t, 4>& b) { return a[0] < b[0]; }))
            std::sort(e->needs.begin(), e->needs.end(), [](const std::array<uint32_t, 4>& a, const std::array<uint32_t, 4>& b) { return a[0] < b[0]; });
        g->needs_current = 1;
        e->early_taken = 1;
    }
    for (const auto& n : e->needs) list.push_back(n[0]);
    return IVX_OK;
}

extern "C" {
int ivx_absorb_sphere_enqueue(ivx_grid* g, const float center[3], float influence_radius, float sphere_radius, const float densities[256]) {
    return absorb_enqueue(g, "ivx_absorb_sphere_enqueue", 0, center, nullptr, influence_radius, sphere_radius, densities);
}
int ivx_absorb_capsule_enqueue(ivx_grid* g, const float segment_start[3], const float segment_vector[3], float influence_radius, float capsule_radius,
                               const float densities[256]) {
    return absorb_enqueue(g, "ivx_absorb_capsule_enqueue", 1, segment_start, segment_vector, influence_radius, capsule_radius, densities);
}
int ivx_absorb_collect(ivx_grid* g, ivx_absorb_result* out, uint32_t* emptied_by_type, uint8_t* invalidated_chunks) {
    return absorb_collect(g, "ivx_absorb_collect", out, emptied_by_type, invalidated_chunks);
}

int ivx_absorb_sphere(ivx_grid* g, const float center[3], float influence_radius, float sphere_radius, const float densities[256], ivx_absorb_result* out,
                      uint32_t* emptied_by_type, uint8_t* invalidated_chunks) {
    IVX_REQUIRE(out, IVX_ERR_INVALID, "ivx_absorb_sphere: null argument");
    const int rc = absorb_enqueue(g, "ivx_absorb_sphere", 0, center, nullptr, influence_radius, sphere_radius, densities);
    return rc ? rc : absorb_collect(g, "ivx_absorb_sphere", out, emptied_by_type, invalidated_chunks);
}

int ivx_absorb_capsule(ivx_grid* g, const float segment_start[3], const float segment_vector[3], float influence_radius, float capsule_radius,
                       const float densities[256], ivx_absorb_result* out, uint32_t* emptied_by_type, uint8_t* invalidated_chunks) {
    IVX_REQUIRE(out, IVX_ERR_INVALID, "ivx_absorb_capsule: null argument");
    const int rc = absorb_enqueue(g, "ivx_absorb_capsule", 1, segment_start, segment_vector, influence_radius, capsule_radius, densities);
    return rc ? rc : absorb_collect(g, "ivx_absorb_capsule", out, emptied_by_type, invalidated_chunks);
}
}  // extern "C"

// rotate a vector by a quaternion the way glam's Quat::mul_vec3a does (host side of Isometry3::transform_point)
static void host_qrot(const float q[4], const float v[3], float out[3]) {
    const float qx = q[0], qy = q[1], qz = q[2], qw = q[3], vx = v[0], vy = v[1], vz = v[2];
    const float b2 = (qx * qx + qy * qy) + qz * qz, s1 = qw * qw - b2, s2 = ((vx * qx + vy * qy) + vz * qz) * 2.0f, s3 = qw * 2.0f;
    const float cxp = qy * vz - vy * qz, cyp = qz * vx - vz * qx, czp = qx * vy - vx * qy;  // cross(b, v)
    out[0] = (vx * s1 + qx * s2) + cxp * s3;
    out[1] = (vy * s1 + qy * s2) + cyp * s3;
    out[2] = (vz * s1 + qz * s2) + czp * s3;
}

// voxel_ranges_touching_aab (intersection.rs:766-782) on the occupied ranges; false when a range is empty
static bool touched_ranges(const uint32_t occ[12], const float lo_f[3], const float hi_f[3], int32_t vlo[3], int32_t vhi[3], uint32_t lo[3], uint32_t cc[3]) {
    for (int d = 0; d < 3; ++d) {
        const float fl = std::floor(lo_f[d]), ce = std::ceil(hi_f[d]);
        const long s = (long)(fl > 0.0f ? fl : 0.0f), e = ce > 0.0f ? (long)ce : 0;  // `as usize` saturates at 0
        vlo[d] = (int32_t)std::max<long>((long)occ[6 + 2 * d], s);
        vhi[d] = (int32_t)std::min<long>((long)occ[7 + 2 * d], e);
        if (vlo[d] >= vhi[d]) return false;
        lo[d] = (uint32_t)vlo[d] / 16u;
        cc[d] = ((uint32_t)vhi[d] + 15u) / 16u - lo[d];
    }
    return true;
}

// mode 0: sphere (shape3 = centre, shape1 = radius); 1: plane (unit normal, displacement); 2: capsule (segment start, radius; shape3b = segment vector)
static int many_phase(ivx_grid* const* grids, size_t n, const std::function<int(size_t)>& f);
static int many_check(ivx_grid* const* grids, size_t n, const char* who);
static int many_fail(ivx_grid* const* grids, size_t n, int rc);
// the chunk box and voxel ranges a collidable touches of an object (mode 0 sphere: centre shape3, radius shape1; 1 plane: unit normal shape3,
// displacement shape1; 2 capsule: segment start shape3, segment vector shape3b, radius shape1); false: nothing touched
static bool contacts_box(const ivx_grid* g, int mode, const float rotation_xyzw[4], const float translation[3], const float shape3[3], const float shape3b[3],
                         float shape1, const uint32_t occ[12], int32_t vlo[3], int32_t vhi[3], uint32_t lo[3], uint32_t cc[3]) {
    const int plane = mode == 1;
    const float inv = 1.0f / g->extent;
    float lo_f[3], hi_f[3];
    if (mode == 2) {
        // capsule.iso_transformed(transform_to_object_space).scaled(inverse_voxel_extent).compute_aabb() (impact_geometry/src/capsule.rs:100-137;
        // intersection.rs:73-82)
        float ra[3], rv[3];
        host_qrot(rotation_xyzw, shape3, ra);
        host_qrot(rotation_xyzw, shape3b, rv);
        const float rn = inv * shape1;
        for (int d = 0; d < 3; ++d) {
            const float an = (ra[d] + translation[d]) * inv, vn = rv[d] * inv;
            const float en = an + vn;
            const float la = an - rn, le = en - rn, ha = an + rn, he = en + rn;
            lo_f[d] = le < la ? le : la;
            hi_f[d] = he > ha ? he : ha;
        }
    } else if (!plane) {
        // sphere.iso_transformed(transform_to_object_space).scaled(inverse_voxel_extent) and its box (intersection.rs:51-60)
        float rc3[3];
        host_qrot(rotation_xyzw, shape3, rc3);
        const float rn = inv * shape1;
        for (int d = 0; d < 3; ++d) {
            const float cn = (rc3[d] + translation[d]) * inv;
            lo_f[d] = cn - rn;
            hi_f[d] = cn + rn;
        }
    } else {
        // plane.iso_transformed(..).scaled(..) (impact_geometry/src/plane.rs:170-203), then the occupied box projected onto its negative
        // halfspace (voxel_ranges_within_plane, intersection.rs:751-761; axis_aligned_box.rs:460-488)
        const float point[3] = {shape3[0] * shape1, shape3[1] * shape1, shape3[2] * shape1};
        float tp[3], tn[3];
        host_qrot(rotation_xyzw, point, tp);
        for (int d = 0; d < 3; ++d) tp[d] += translation[d];
        host_qrot(rotation_xyzw, shape3, tn);
        const float disp = ((tn[0] * tp[0] + tn[1] * tp[1]) + tn[2] * tp[2]) * inv;
        float blo[3], bhi[3];
        for (int d = 0; d < 3; ++d) {
            blo[d] = lo_f[d] = (float)occ[6 + 2 * d];
            bhi[d] = hi_f[d] = (float)occ[7 + 2 * d];
        }
        const int perm[3][3] = {{0, 1, 2}, {1, 2, 0}, {2, 0, 1}};
        auto mn2 = [](float x, float y) { return (y < x) ? y : x; };
        auto mx2 = [](float x, float y) { return (y > x) ? y : x; };
        for (int r = 0; r < 3; ++r) {
            const int i = perm[r][0], j = perm[r][1], k = perm[r][2];
            if (std::fabs(tn[k]) > 1e-8f) {
                const float a0 = tn[i] * blo[i] + tn[j] * blo[j], b0 = tn[i] * blo[i] + tn[j] * bhi[j], c0 = tn[i] * bhi[i] + tn[j] * blo[j],
                            d0 = tn[i] * bhi[i] + tn[j] * bhi[j];
                const float extremal = (disp - mn2(mn2(mn2(a0, b0), c0), d0)) / tn[k];
                if (!std::signbit(tn[k])) {
                    lo_f[k] = mn2(lo_f[k], extremal);
                    hi_f[k] = mn2(hi_f[k], extremal);
                } else {
                    lo_f[k] = mx2(lo_f[k], extremal);
                    hi_f[k] = mx2(hi_f[k], extremal);
                }
            }
        }
    }
    return touched_ranges(occ, lo_f, hi_f, vlo, vhi, lo, cc);
}

static int voxel_object_contacts(ivx_grid* g, const char* who, int mode, const float rotation_xyzw[4], const float translation[3], const float shape3[3],
                                 const float shape3b[3], float shape1, uint64_t id_a, uint64_t id_b, uint32_t body_a, uint32_t body_b, const float response[3], ivx_contact* out,
                                 size_t cap, size_t* n_out) {
    IVX_REQUIRE(g && rotation_xyzw && translation && shape3 && (shape3b || mode != 2) && response && n_out && (out || cap == 0), IVX_ERR_INVALID,
                "%s: null argument", who);
    IVX_REQUIRE(g->regions_valid, IVX_ERR_STATE, "%s: derived state must be current (ivx_derive_state + ivx_label_regions)", who);
    IVX_REQUIRE(g->x_off == 0 && g->gx == g->cc[0] && !g->has_ghost[0] && !g->has_ghost[1], IVX_ERR_STATE, "%s: not available on a slab of a decomposed grid",
                who);
    *n_out = 0;
    int rc;
    uint32_t occ[12];
    if ((rc = reference_occupied(g, occ))) return rc;
    int32_t vlo[3], vhi[3];
    uint32_t lo[3], cc[3];
    if (!contacts_box(g, mode, rotation_xyzw, translation, shape3, shape3b, shape1, occ, vlo, vhi, lo, cc)) return IVX_OK;
    const size_t n_box = (size_t)cc[0] * cc[1] * cc[2];
    const size_t off_offsets = n_box * 4, off_total = 2 * n_box * 4, off_out = (off_total + 16 + 63) & ~(size_t)63;
    if ((rc = ensure_dev_scratch(g, off_out + cap * sizeof(ivx_contact)))) return rc;
    char* base = static_cast<char*>(g->dev_scratch);
    uint32_t* d_counts = reinterpret_cast<uint32_t*>(base);
    uint32_t* d_offsets = reinterpret_cast<uint32_t*>(base + off_offsets);
    uint32_t* d_total = reinterpret_cast<uint32_t*>(base + off_total);
    ivx_contact* d_out = reinterpret_cast<ivx_contact*>(base + off_out);
    for (int pass = 0; pass < 2; ++pass)
        if ((rc = ivx_launch_sphere_contacts(g, lo, cc, vlo, vhi, rotation_xyzw, translation, shape3, shape3b, shape1, id_a, id_b, body_a, body_b, response,
                                             d_counts, d_offsets, d_total, d_out, (uint32_t)std::min<size_t>(cap, 0xFFFFFFFFu), pass, mode)))
            return rc;
    uint32_t total = 0;
    if ((rc = d2h(g, &total, d_total, sizeof(total)))) return rc;
    *n_out = total;
    IVX_REQUIRE(total <= cap, IVX_ERR_CAPACITY, "%s: %u contacts exceed the capacity %zu", who, total, cap);
    if (total && (rc = d2h(g, out, d_out, (size_t)total * sizeof(ivx_contact)))) return rc;
    return IVX_OK;
}

// the context's pinned, device-visible scratch (ivx_ctx::pinned_scratch): at least `bytes`
static int ctx_pinned_scratch(ivx_ctx* c, size_t bytes) {
    if (c->pinned_scratch_bytes >= bytes) return IVX_OK;
    IVX_HIP_CHECK(ivx_stream_sync(c->stream));
    if (c->pinned_scratch) (void)hipHostFree(c->pinned_scratch);
    c->pinned_scratch = c->pinned_scratch_dev = nullptr;
    c->pinned_scratch_bytes = 0;
    const size_t want = std::max<size_t>(2 * bytes, 1 << 20);
    IVX_HIP_CHECK(hipHostMalloc(&c->pinned_scratch, want, hipHostMallocMapped));
    IVX_HIP_CHECK(hipHostGetDevicePointer(&c->pinned_scratch_dev, c->pinned_scratch, 0));
    c->pinned_scratch_bytes = want;
    return IVX_OK;
}

// One collidable per object, N objects, in the launches of one (many.hpp): the reference's collision pass walks every voxel object of the
// scene against the collidables near it (impact_voxel/src/collidable.rs:1051-1286: the per-pair dispatch) — here the pairs (object i,
// collidable i) of one call. Two recorded phases, two waits for ALL objects where the single-object call has two per object: (1) count + scan
// per object, the totals written by the scan straight into host-mapped memory; (2) the emit passes, each object's contacts at its offset of one
// host-mapped buffer, copied to `out` by the host. out_offsets[i] .. out_offsets[i + 1]: object i's contacts, in the order the single-object
// call returns them (a manifold each).
int ivx_voxel_object_contacts_many(ivx_grid* const* grids, size_t n, const ivx_collidable_query* queries, ivx_contact* out, size_t cap, uint32_t* out_offsets) {
    const char* who = "ivx_voxel_object_contacts_many";
    IVX_REQUIRE(out_offsets, IVX_ERR_INVALID, "%s: null argument", who);
    out_offsets[0] = 0;
    if (n == 0) return IVX_OK;
    IVX_REQUIRE(grids && queries && (out || cap == 0), IVX_ERR_INVALID, "%s: null argument", who);
    ivx_ctx* c = grids[0] ? grids[0]->ctx : nullptr;
    for (size_t i = 0; i < n; ++i) {
        ivx_grid* g = grids[i];
        IVX_REQUIRE(g && g->ctx == c, IVX_ERR_INVALID, "%s: object %zu is null or belongs to another context", who, i);
        IVX_REQUIRE(queries[i].mode >= 0 && queries[i].mode <= 2, IVX_ERR_INVALID, "%s: query %zu: mode %d", who, i, queries[i].mode);
        IVX_REQUIRE(g->regions_valid, IVX_ERR_STATE, "%s: derived state of object %zu must be current (ivx_derive_state + ivx_label_regions)", who, i);
        IVX_REQUIRE(g->x_off == 0 && g->gx == g->cc[0] && !g->has_ghost[0] && !g->has_ghost[1], IVX_ERR_STATE, "%s: not available on a slab of a decomposed grid", who);
    }
    IVX_REQUIRE(!ivx_many_recording(), IVX_ERR_STATE, "%s: not inside an ivx_many_begin bracket (the call waits for its own phases)", who);
    int rc;
    hipStream_t s = c->stream;
    struct Box {
        int32_t vlo[3], vhi[3];
        uint32_t lo[3], cc[3];
        bool hit;
        size_t off;  // where this query's counts and offsets start in its object's scratch
    };
    static thread_local std::vector<Box> box;
    box.assign(n, Box{});
    // (an object may appear more than once — near a sphere AND the ground plane, the reference's collision pass visits every collidable near an
    // object —: each query gets a range of its own in the object's scratch, the recorded chains of one object must not share counts)
    static thread_local std::vector<std::pair<ivx_grid*, size_t>> scratch_need;
    scratch_need.clear();
    // what may wait or allocate, ahead of the recording: occupied ranges, the objects' scratch for counts and offsets, the pinned block
    for (size_t i = 0; i < n; ++i) {
        const ivx_collidable_query& q = queries[i];
        uint32_t occ[12];
        if ((rc = reference_occupied(grids[i], occ))) return rc;
        Box& b = box[i];
        b.hit = contacts_box(grids[i], q.mode, q.rotation_xyzw, q.translation, q.shape3, q.shape3b, q.shape1, occ, b.vlo, b.vhi, b.lo, b.cc);
        if (!b.hit) continue;
        const size_t need = (2 * (size_t)b.cc[0] * b.cc[1] * b.cc[2] * 4 + 64 + 255) & ~(size_t)255;
        auto it = std::find_if(scratch_need.begin(), scratch_need.end(), [&](const std::pair<ivx_grid*, size_t>& e) { return e.first == grids[i]; });
        if (it == scratch_need.end()) {
            scratch_need.emplace_back(grids[i], (size_t)0);
            it = scratch_need.end() - 1;
        }
        b.off = it->second;
        it->second += need;
    }
    for (const auto& e : scratch_need)
        if ((rc = ensure_dev_scratch(e.first, e.second))) return rc;
    const size_t totals_bytes = (n * 4 + 63) & ~(size_t)63;
    if ((rc = ctx_pinned_scratch(c, totals_bytes + 4096))) return rc;
    uint32_t* totals = static_cast<uint32_t*>(c->pinned_scratch);
    memset(totals, 0, n * 4);
    auto launch = [&](size_t i, int pass, uint32_t* d_total, ivx_contact* d_out, uint32_t cap_i) -> int {
        const ivx_collidable_query& q = queries[i];
        const Box& b = box[i];
        ivx_grid* g = grids[i];
        const size_t n_box = (size_t)b.cc[0] * b.cc[1] * b.cc[2];
        char* base = static_cast<char*>(g->dev_scratch) + b.off;
        return ivx_launch_sphere_contacts(g, b.lo, b.cc, b.vlo, b.vhi, q.rotation_xyzw, q.translation, q.shape3, q.shape3b, q.shape1, q.collidable_id_a, q.collidable_id_b,
                                          q.body_a, q.body_b, q.response, reinterpret_cast<uint32_t*>(base), reinterpret_cast<uint32_t*>(base + n_box * 4), d_total, d_out,
                                          cap_i, pass, q.mode);
    };
    uint32_t* totals_dev = static_cast<uint32_t*>(c->pinned_scratch_dev);
    if ((rc = many_phase(grids, n, [&](size_t i) -> int { return box[i].hit ? launch(i, 0, totals_dev + i, nullptr, 0u) : IVX_OK; }))) return rc;
    IVX_HIP_CHECK(ivx_stream_sync(s));
    size_t run = 0;
    for (size_t i = 0; i < n; ++i) {
        out_offsets[i] = (uint32_t)run;
        run += totals[i];
    }
    out_offsets[n] = (uint32_t)run;
    IVX_REQUIRE(run <= cap, IVX_ERR_CAPACITY, "%s: %zu contacts exceed the capacity %zu", who, run, cap);
    if (run == 0) return IVX_OK;
    static thread_local std::vector<uint32_t> counts;
    counts.assign(totals, totals + n);  // (the pinned block may move when it grows)
    if ((rc = ctx_pinned_scratch(c, run * sizeof(ivx_contact)))) return rc;
    ivx_contact* list_dev = static_cast<ivx_contact*>(c->pinned_scratch_dev);
    if ((rc = many_phase(grids, n, [&](size_t i) -> int { return counts[i] ? launch(i, 1, nullptr, list_dev + out_offsets[i], counts[i]) : IVX_OK; }))) return rc;
    IVX_HIP_CHECK(ivx_stream_sync(s));
    memcpy(out, c->pinned_scratch, run * sizeof(ivx_contact));
    return IVX_OK;
}

int ivx_sphere_voxel_object_contacts(ivx_grid* g, const float rotation_xyzw[4], const float translation[3], const float sphere_center[3], float sphere_radius,
                                     uint64_t collidable_id_a, uint64_t collidable_id_b, uint32_t body_a, uint32_t body_b, const float response[3],
                                     ivx_contact* out, size_t cap, size_t* n_out) {
    return voxel_object_contacts(g, "ivx_sphere_voxel_object_contacts", 0, rotation_xyzw, translation, sphere_center, nullptr, sphere_radius, collidable_id_a,
                                 collidable_id_b, body_a, body_b, response, out, cap, n_out);
}

int ivx_plane_voxel_object_contacts(ivx_grid* g, const float rotation_xyzw[4], const float translation[3], const float plane_unit_normal[3],
                                    float plane_displacement, uint64_t collidable_id_a, uint64_t collidable_id_b, uint32_t body_a, uint32_t body_b,
                                    const float response[3], ivx_contact* out, size_t cap, size_t* n_out) {
    return voxel_object_contacts(g, "ivx_plane_voxel_object_contacts", 1, rotation_xyzw, translation, plane_unit_normal, nullptr, plane_displacement,
                                 collidable_id_a, collidable_id_b, body_a, body_b, response, out, cap, n_out);
}

int ivx_capsule_voxel_object_contacts(ivx_grid* g, const float rotation_xyzw[4], const float translation[3], const float segment_start[3],
                                      const float segment_vector[3], float capsule_radius, uint64_t collidable_id_a, uint64_t collidable_id_b,
                                      uint32_t body_a, uint32_t body_b, const float response[3], ivx_contact* out, size_t cap, size_t* n_out) {
    return voxel_object_contacts(g, "ivx_capsule_voxel_object_contacts", 2, rotation_xyzw, translation, segment_start, segment_vector, capsule_radius,
                                 collidable_id_a, collidable_id_b, body_a, body_b, response, out, cap, n_out);
}

// ---- collision probes + mutual contacts (second part) ---------------------------------------------------------
int ivx_collision_probes_recompute(ivx_grid* g, size_t* n_points) {
    IVX_REQUIRE(g && n_points, IVX_ERR_INVALID, "ivx_collision_probes_recompute: null argument");
    IVX_REQUIRE(g->mesh_valid, IVX_ERR_STATE, "ivx_collision_probes_recompute: call ivx_remesh first");
    IVX_REQUIRE(g->x_off == 0 && g->gx == g->cc[0] && !g->has_ghost[0] && !g->has_ghost[1], IVX_ERR_STATE,
                "ivx_collision_probes_recompute: not available on a slab of a decomposed grid");
    IVX_REQUIRE(g->cc[0] <= 1024 && g->cc[1] <= 1024 && g->cc[2] <= 1024, IVX_ERR_INVALID, "ivx_collision_probes_recompute: more than 1024 chunks along an axis");
    *n_points = 0;
    int rc;
    uint32_t occ[12];
    if ((rc = reference_occupied(g, occ))) return rc;
    // determine_log2_block_size_for_object (collidable.rs:451-471)
    uint32_t min_extent = 0xFFFFFFFFu;
    for (int d = 0; d < 3; ++d) min_extent = std::min(min_extent, occ[7 + 2 * d] > occ[6 + 2 * d] ? occ[7 + 2 * d] - occ[6 + 2 * d] : 0u);
    const uint32_t log2_bs = min_extent >= 16 ? 3 : (min_extent >= 8 ? 2 : (min_extent >= 4 ? 1 : 0));
    const uint32_t n_blocks = 1u << (3u * (4u - log2_bs));
    const uint32_t n_sub = g->mesh_counts.n_submeshes;
    g->n_probe_points = 0;
    g->n_probe_sub = n_sub;
    g->probes_serial = g->mesh_serial;
    if (!g->probe_manager) g->probe_manager = new (std::nothrow) ivx_probe_manager();
    IVX_REQUIRE(g->probe_manager, IVX_ERR_HIP, "ivx_collision_probes_recompute: out of host memory");
    ivx_probe_manager* pm = g->probe_manager;
    pm->range_of.clear();
    pm->points.free_ranges.clear();
    pm->total = 0;
    if (n_sub == 0) return IVX_OK;
    if (n_sub > g->probe_entry_cap) {
        if (g->probe_entries) (void)hipFree(g->probe_entries);
        g->probe_entries = nullptr;
        g->probe_entry_cap = 0;
        if ((rc = dev_alloc(&g->probe_entries, (size_t)n_sub * 5))) return rc;
        g->probe_entry_cap = n_sub;
    }
    // scratch: [corner lists: one u32 per index][selected vertices: n_sub * n_blocks][counts n_sub][offsets n_sub + 1][error word]
    const size_t ni = g->mesh_counts.n_indices;
    const size_t off_sel = ni * 4, off_counts = off_sel + (size_t)n_sub * n_blocks * 4, off_offsets = off_counts + (size_t)n_sub * 4,
                 off_err = off_offsets + ((size_t)n_sub + 1) * 4, total = off_err + 4;
    if ((rc = ensure_dev_scratch(g, total))) return rc;
    char* base = static_cast<char*>(g->dev_scratch);
    uint32_t* d_counts = reinterpret_cast<uint32_t*>(base + off_counts);
    uint32_t* d_offsets = reinterpret_cast<uint32_t*>(base + off_offsets);
    uint32_t* d_err = reinterpret_cast<uint32_t*>(base + off_err);
    IVX_HIP_CHECK(ivx_memset_async(d_err, 0, 4, g->ctx->stream));
    if ((rc = ivx_launch_probe_select(g, n_sub, log2_bs, reinterpret_cast<uint32_t*>(base), reinterpret_cast<uint32_t*>(base + off_sel), d_counts, d_offsets,
                                      d_err, nullptr)))
        return rc;
    uint32_t tail[2];  // offsets[n_sub] = total, error word
    if ((rc = d2h(g, tail, d_offsets + n_sub, sizeof(tail)))) return rc;
    IVX_REQUIRE(tail[1] == 0, IVX_ERR_CAPACITY, "ivx_collision_probes_recompute: a chunk submesh holds more vertices than a Surface Nets chunk can");
    const uint32_t n_pts = tail[0];
    if (n_pts > g->probe_point_cap) {
        const size_t cap = std::max<size_t>(n_pts, g->probe_point_cap * 2);
        if (g->probe_points) (void)hipFree(g->probe_points);
        if (g->probe_chunk) (void)hipFree(g->probe_chunk);
        g->probe_points = nullptr;
        g->probe_chunk = nullptr;
        g->probe_point_cap = 0;
        if ((rc = dev_alloc(&g->probe_points, cap * 3))) return rc;
        if ((rc = dev_alloc(&g->probe_chunk, cap))) return rc;
        g->probe_point_cap = cap;
    }
    if ((rc = ivx_launch_probe_gather(g, n_sub, log2_bs, reinterpret_cast<uint32_t*>(base + off_sel), d_counts, d_offsets, g->probe_entries, nullptr))) return rc;
    pm->total = n_pts;
    pm->built = false;  // (the entries stay on the device until a sync or a download asks for them)
    IVX_HIP_CHECK(ivx_stream_sync(g->ctx->stream));
    g->n_probe_points = n_pts;
    *n_points = n_pts;
    return IVX_OK;
}

int ivx_collision_probes_download(ivx_grid* g, float* points, size_t cap_points, uint32_t* chunk_entries, size_t cap_entries, size_t* n_points,
                                  size_t* n_entries) {
    IVX_REQUIRE(g && n_points && n_entries, IVX_ERR_INVALID, "ivx_collision_probes_download: null argument");
    IVX_REQUIRE(g->mesh_valid && g->probes_serial == g->mesh_serial, IVX_ERR_STATE, "ivx_collision_probes_download: call ivx_collision_probes_recompute first");
    *n_points = g->n_probe_points;
    *n_entries = 0;
    int rc;
    size_t ne = 0;  // the live entries in the order of their ranges (= submesh order right after a recompute)
    if ((rc = probe_manager_build(g))) return rc;
    if (g->probe_manager) {
        std::vector<std::pair<uint32_t, uint32_t>> order;
        for (const auto& kv : g->probe_manager->range_of) order.push_back({kv.second.first, kv.first});
        std::sort(order.begin(), order.end());
        for (const auto& o : order) {
            if (chunk_entries && ne < cap_entries) {
                const uint32_t c = o.second;
                const auto& r = g->probe_manager->range_of.at(c);
                uint32_t* e = chunk_entries + 5 * ne;
                e[0] = c / (g->cc[1] * g->cc[2]), e[1] = (c / g->cc[2]) % g->cc[1], e[2] = c % g->cc[2], e[3] = r.first, e[4] = r.second;
            }
            ne += 1;
        }
    }
    *n_entries = ne;
    IVX_REQUIRE(!points || g->n_probe_points <= cap_points, IVX_ERR_CAPACITY, "ivx_collision_probes_download: %u points exceed the capacity %zu",
                g->n_probe_points, cap_points);
    IVX_REQUIRE(!chunk_entries || ne <= cap_entries, IVX_ERR_CAPACITY, "ivx_collision_probes_download: %zu entries exceed the capacity %zu", ne, cap_entries);
    if (points && g->n_probe_points && (rc = d2h(g, points, g->probe_points, (size_t)g->n_probe_points * 12))) return rc;
    return IVX_OK;
}

}  // extern "C"
// the host mirror of chunk_point_ranges after a recompute (clear(): no free ranges), from the entries the gather pass left on the device
static int probe_manager_build(ivx_grid* g) {
    ivx_probe_manager* pm = g->probe_manager;
    if (!pm || pm->built) return IVX_OK;
    std::vector<uint32_t> e((size_t)g->n_probe_sub * 5);
    int rc;
    if (!e.empty() && (rc = d2h(g, e.data(), g->probe_entries, e.size() * 4))) return rc;
    pm->range_of.clear();
    pm->range_of.reserve(g->n_probe_sub);
    pm->points.free_ranges.clear();
    for (uint32_t sidx = 0; sidx < g->n_probe_sub; ++sidx)
        if (e[5 * (size_t)sidx + 4] > e[5 * (size_t)sidx + 3]) {
            const uint32_t c3[3] = {e[5 * (size_t)sidx], e[5 * (size_t)sidx + 1], e[5 * (size_t)sidx + 2]};
            pm->range_of[linear_chunk(g, c3)] = {e[5 * (size_t)sidx + 3], e[5 * (size_t)sidx + 4]};
        }
    pm->built = true;
    return IVX_OK;
}
extern "C" {
}  // extern "C"
// VoxelObjectCollisionProbes::sync_with_voxel_object_and_mesh (collidable.rs:394-433, 524-612) in the stages the single-object call and the
// many-objects call share: prepare (host: which chunks, which submesh slots, scratch) | select (device: the points of the listed submeshes
// picked again, their counts) | allocate (host: update_for_chunk chunk by chunk, the RangeAllocator) | gather (device: freed ranges marked,
// the picked points copied to their ranges).
namespace {
struct ProbeSyncJob {
    ivx_grid* g = nullptr;
    uint32_t log2_bs = 0, n_blocks = 0, n_rec = 0;
    std::vector<uint32_t> list, slots, dst;
    std::vector<int32_t> rec_index;
    std::vector<std::pair<uint32_t, uint32_t>> freed;
    size_t off_sel = 0, off_counts = 0, off_dst = 0, off_slots = 0, off_err = 0;
    char* base = nullptr;
};
}  // namespace
static int probe_sync_prepare(ivx_grid* g, const uint8_t* invalidated_chunks, const char* who, ProbeSyncJob& j) {
    IVX_REQUIRE(g && invalidated_chunks, IVX_ERR_INVALID, "%s: null argument", who);
    ivx_submesh_manager* m = g->submesh_manager;
    ivx_probe_manager* pm = g->probe_manager;
    IVX_REQUIRE(g->mesh_valid && m && m->serial == g->mesh_serial, IVX_ERR_STATE, "%s: call ivx_mesh_sync first", who);
    IVX_REQUIRE(pm && (g->probes_serial + 1 == g->mesh_serial || g->probes_serial == g->mesh_serial), IVX_ERR_STATE,
                "%s: the probes must be those of the mesh before the last ivx_mesh_sync (ivx_collision_probes_recompute, or a sync per mesh sync)", who);
    int rc;
    if ((rc = probe_manager_build(g))) return rc;
    uint32_t occ[12];
    if ((rc = reference_occupied(g, occ))) return rc;
    uint32_t min_extent = 0xFFFFFFFFu;
    for (int d = 0; d < 3; ++d) min_extent = std::min(min_extent, occ[7 + 2 * d] > occ[6 + 2 * d] ? occ[7 + 2 * d] - occ[6 + 2 * d] : 0u);
    j.g = g;
    j.log2_bs = min_extent >= 16 ? 3 : (min_extent >= 8 ? 2 : (min_extent >= 4 ? 1 : 0));
    j.n_blocks = 1u << (3u * (4u - j.log2_bs));
    // the invalidated chunks in chunk-linear order (the reference walks a hash set: unpinned); those that have a submesh get their points picked
    j.list.clear(), j.slots.clear(), j.freed.clear();
    for (uint32_t c = 0; c < g->n_chunks; ++c)
        if (invalidated_chunks[c]) j.list.push_back(c);
    j.rec_index.assign(j.list.size(), -1);
    for (size_t e = 0; e < j.list.size(); ++e) {
        auto it = m->slot_of.find(j.list[e]);
        if (it != m->slot_of.end()) {
            j.rec_index[e] = (int32_t)j.slots.size();
            j.slots.push_back(it->second);
        }
    }
    j.n_rec = (uint32_t)j.slots.size();
    j.dst.assign(j.n_rec, 0u);
    // scratch: [corner lists: one u32 per index][selected vertices: n_rec * n_blocks][counts][dst offsets][slots][error word]
    const size_t ni = m->total_indices, n_rec = j.n_rec;
    j.off_sel = ni * 4, j.off_counts = j.off_sel + n_rec * j.n_blocks * 4, j.off_dst = j.off_counts + n_rec * 4, j.off_slots = j.off_dst + n_rec * 4;
    j.off_err = j.off_slots + n_rec * 4;
    j.base = nullptr;
    if (n_rec) {
        if ((rc = ensure_dev_scratch(g, j.off_err + 4))) return rc;
        j.base = static_cast<char*>(g->dev_scratch);
    }
    return IVX_OK;
}
// d_err: the error word the select pass sets (zeroed by the caller); d_counts_host: optional host-mapped copy of the counts
static int probe_sync_select(ProbeSyncJob& j, uint32_t* d_err, uint32_t* d_counts_host) {
    if (!j.n_rec) return IVX_OK;
    ivx_grid* g = j.g;
    int rc;
    if (!ivx_many_upload(g->ctx, g, j.base + j.off_slots, j.slots.data(), (size_t)j.n_rec * 4) && (rc = h2d(g, j.base + j.off_slots, j.slots.data(), (size_t)j.n_rec * 4)))
        return rc;
    return ivx_launch_probe_select(g, j.n_rec, j.log2_bs, reinterpret_cast<uint32_t*>(j.base), reinterpret_cast<uint32_t*>(j.base + j.off_sel),
                                   reinterpret_cast<uint32_t*>(j.base + j.off_counts), nullptr, d_err, reinterpret_cast<const uint32_t*>(j.base + j.off_slots),
                                   d_counts_host);
}
// update_for_chunk, chunk by chunk (collidable.rs:524-612); *grow: the point buffers must hold pm->total points before the gather
static int probe_sync_allocate(ProbeSyncJob& j, const uint32_t* counts, const char* who, bool* grow) {
    ivx_grid* g = j.g;
    ivx_probe_manager* pm = g->probe_manager;
    for (size_t e = 0; e < j.list.size(); ++e) {
        const uint32_t c = j.list[e];
        const uint32_t n = j.rec_index[e] >= 0 ? counts[(size_t)j.rec_index[e]] : 0u;
        auto old = pm->range_of.find(c);
        if (n == 0) {  // no mesh, or no points
            if (old != pm->range_of.end()) {
                pm->points.free_range(old->second.first, old->second.second);
                j.freed.push_back(old->second);
                pm->range_of.erase(old);
            }
            continue;
        }
        if (old != pm->range_of.end()) {
            pm->points.free_range(old->second.first, old->second.second);
            j.freed.push_back(old->second);
        }
        size_t start;
        if (!pm->points.allocate(n, &start)) start = pm->total, pm->total += n;
        IVX_REQUIRE(pm->total < 0xFFFFFFF0ull, IVX_ERR_CAPACITY, "%s: more than 2^32 probe points", who);
        pm->range_of[c] = {(uint32_t)start, (uint32_t)(start + n)};
        j.dst[(size_t)j.rec_index[e]] = (uint32_t)start;
    }
    pm->points.merge_consecutive();
    *grow = pm->total > g->probe_point_cap;
    return IVX_OK;
}
// grow the point buffers of the listed jobs, keeping what is there: the copies of all, one wait, then the old buffers go
static int probe_sync_grow(ProbeSyncJob* const* jobs, size_t n) {
    if (n == 0) return IVX_OK;
    std::vector<GrowKeep> pending;
    std::vector<size_t> caps(n);
    int rc;
    for (size_t i = 0; i < n; ++i) {
        ivx_grid* g = jobs[i]->g;
        const size_t total = g->probe_manager->total;
        caps[i] = std::max<size_t>(total + total / 2 + 4096, 2 * g->probe_point_cap);
        if ((rc = grow_keep_enqueue(g, &g->probe_points, g->probe_point_cap * 3, caps[i] * 3, pending, 0u))) return rc;
        if ((rc = grow_keep_enqueue(g, &g->probe_chunk, g->probe_point_cap, caps[i], pending, 0u))) return rc;
    }
    IVX_HIP_CHECK(ivx_stream_sync(jobs[0]->g->ctx->stream));
    for (GrowKeep& k : pending) {
        if (*k.slot) (void)hipFree(*k.slot);
        *k.slot = k.fresh;
    }
    for (size_t i = 0; i < n; ++i) jobs[i]->g->probe_point_cap = caps[i];
    return IVX_OK;
}
static int probe_sync_gather(ProbeSyncJob& j) {
    ivx_grid* g = j.g;
    int rc;
    for (const auto& r : j.freed) {  // holes read as "no probe" until a later chunk takes them (the gather below overwrites what was taken now)
        const size_t bytes = (size_t)(r.second - r.first) * 4;
        if (!ivx_many_fill(g->ctx, g, g->probe_chunk + r.first, 0xFFFFFFFFu, bytes)) IVX_HIP_CHECK(ivx_memset_async(g->probe_chunk + r.first, 0xFF, bytes, g->ctx->stream));
    }
    if (j.n_rec) {
        if (!ivx_many_upload(g->ctx, g, j.base + j.off_dst, j.dst.data(), (size_t)j.n_rec * 4) && (rc = h2d(g, j.base + j.off_dst, j.dst.data(), (size_t)j.n_rec * 4)))
            return rc;
        if ((rc = ivx_launch_probe_gather(g, j.n_rec, j.log2_bs, reinterpret_cast<uint32_t*>(j.base + j.off_sel), reinterpret_cast<uint32_t*>(j.base + j.off_counts),
                                          reinterpret_cast<uint32_t*>(j.base + j.off_dst), nullptr, reinterpret_cast<const uint32_t*>(j.base + j.off_slots))))
            return rc;
    }
    return IVX_OK;
}
extern "C" {
int ivx_collision_probes_sync(ivx_grid* g, const uint8_t* invalidated_chunks, size_t* n_points) {
    const char* who = "ivx_collision_probes_sync";
    IVX_REQUIRE(g && invalidated_chunks && n_points, IVX_ERR_INVALID, "%s: null argument", who);
    IVX_REQUIRE(!ivx_many_recording(), IVX_ERR_STATE, "%s: not inside an ivx_many_begin bracket (the call waits for the device twice)", who);
    ProbeSyncJob j;
    int rc;
    if ((rc = probe_sync_prepare(g, invalidated_chunks, who, j))) return rc;
    std::vector<uint32_t> counts(j.n_rec);
    if (j.n_rec) {
        uint32_t* d_err = reinterpret_cast<uint32_t*>(j.base + j.off_err);
        IVX_HIP_CHECK(ivx_memset_async(d_err, 0, 4, g->ctx->stream));
        if ((rc = probe_sync_select(j, d_err, nullptr))) return rc;
        if ((rc = d2h(g, counts.data(), j.base + j.off_counts, (size_t)j.n_rec * 4))) return rc;
        uint32_t err = 0;
        if ((rc = d2h(g, &err, d_err, 4))) return rc;
        IVX_REQUIRE(err == 0, IVX_ERR_CAPACITY, "%s: a chunk submesh holds more vertices than a Surface Nets chunk can", who);
    }
    bool grow = false;
    if ((rc = probe_sync_allocate(j, counts.data(), who, &grow))) return rc;
    ProbeSyncJob* one = &j;
    if (grow && (rc = probe_sync_grow(&one, 1))) return rc;
    if ((rc = probe_sync_gather(j))) return rc;
    IVX_HIP_CHECK(ivx_stream_sync(g->ctx->stream));
    g->n_probe_points = (uint32_t)g->probe_manager->total;
    g->probes_serial = g->mesh_serial;
    *n_points = g->probe_manager->total;
    return IVX_OK;
}

// The same for N objects of one context in the launches of one (many.hpp) — every voxel object's probes follow its mesh each frame
// (impact_voxel/src/lib.rs:729-733 with collidable.rs:394-433) —: the select passes of all objects merged, their counts and error words written
// into host-mapped memory, ONE wait; the allocators of all objects on the host; the fills and gathers of all objects merged, ONE wait.
int ivx_collision_probes_sync_many(ivx_grid* const* grids, size_t n, const uint8_t* const* invalidated_chunks, size_t* n_points) {
    const char* who = "ivx_collision_probes_sync_many";
    if (n == 0) return IVX_OK;
    IVX_REQUIRE(grids && invalidated_chunks && n_points, IVX_ERR_INVALID, "%s: null argument", who);
    int rc = many_check(grids, n, who);
    if (rc) return rc;
    IVX_REQUIRE(!ivx_many_recording(), IVX_ERR_STATE, "%s: not inside an ivx_many_begin bracket (the call waits for its own phases)", who);
    ivx_ctx* c = grids[0]->ctx;
    static thread_local std::vector<ProbeSyncJob> jobs;
    if (jobs.size() < n) jobs.resize(n);
    size_t words = 0;
    static thread_local std::vector<size_t> off;
    off.assign(n, 0);
    for (size_t i = 0; i < n; ++i) {
        if ((rc = probe_sync_prepare(grids[i], invalidated_chunks[i], who, jobs[i]))) return rc;
        off[i] = n + words;  // [error word per object][counts of every object]
        words += jobs[i].n_rec;
    }
    if ((rc = ctx_pinned_scratch(c, (n + words) * 4 + 64))) return rc;
    uint32_t* host = static_cast<uint32_t*>(c->pinned_scratch);
    uint32_t* host_dev = static_cast<uint32_t*>(c->pinned_scratch_dev);
    memset(host, 0, (n + words) * 4);
    if (words) {
        if ((rc = many_phase(grids, n, [&](size_t i) -> int { return probe_sync_select(jobs[i], host_dev + i, host_dev + off[i]); }))) return many_fail(grids, n, rc);
        IVX_HIP_CHECK(ivx_stream_sync(c->stream));
    }
    for (size_t i = 0; i < n; ++i)
        IVX_REQUIRE(host[i] == 0, IVX_ERR_CAPACITY, "%s: object %zu: a chunk submesh holds more vertices than a Surface Nets chunk can", who, i);
    static thread_local std::vector<ProbeSyncJob*> growing;
    growing.clear();
    for (size_t i = 0; i < n; ++i) {
        bool grow = false;
        if ((rc = probe_sync_allocate(jobs[i], host + off[i], who, &grow))) return rc;
        if (grow) growing.push_back(&jobs[i]);
    }
    if ((rc = probe_sync_grow(growing.data(), growing.size()))) return rc;
    if ((rc = many_phase(grids, n, [&](size_t i) -> int { return probe_sync_gather(jobs[i]); }))) return many_fail(grids, n, rc);
    IVX_HIP_CHECK(ivx_stream_sync(c->stream));
    for (size_t i = 0; i < n; ++i) {
        ivx_grid* g = grids[i];
        g->n_probe_points = (uint32_t)g->probe_manager->total;
        g->probes_serial = g->mesh_serial;
        n_points[i] = g->probe_manager->total;
    }
    return IVX_OK;
}

namespace {
struct HBox {
    float lo[3], hi[3];
};
// AxisAlignedBox::find_contained_subsegment (impact_geometry/src/axis_aligned_box.rs:385-415)
bool host_subsegment(const HBox& b, const float s[3], const float v[3], float* t0, float* t1) {
    float a = 0.0f, z = 1.0f;
    for (int d = 0; d < 3; ++d) {
        if (std::fabs(v[d]) > 1e-8f) {
            const float r = 1.0f / v[d];
            const float u1 = (b.lo[d] - s[d]) * r, u2 = (b.hi[d] - s[d]) * r;
            const float en = u1 < u2 ? u1 : u2, ex = u1 < u2 ? u2 : u1;
            a = en > a ? en : a;
            z = ex < z ? ex : z;
        } else if (s[d] < b.lo[d] || s[d] > b.hi[d]) {
            return false;
        }
    }
    *t0 = a;
    *t1 = z;
    return a <= z;
}
void host_qmul(const float a[4], const float b[4], float o[4]) {  // glam Quat::mul_quat (xyzw)
    o[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    o[1] = a[3] * b[1] - a[0] * b[2] + a[1] * b[3] + a[2] * b[0];
    o[2] = a[3] * b[2] + a[0] * b[1] - a[1] * b[0] + a[2] * b[3];
    o[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
}
// compute_box_intersection_bounds (impact_geometry/src/oriented_box.rs:315-431): box A axis-aligned, box B = (centre, orientation, half
// extents) in A's frame; bounds of the overlap in A's frame and in B's own frame (relative to its centre)
bool host_box_bounds(const HBox& a, const float bc[3], const float bq[4], const float bh[3], HBox* in_a, HBox* in_b) {
    static const int E[12][2] = {{0, 1}, {2, 3}, {4, 5}, {6, 7}, {0, 2}, {1, 3}, {4, 6}, {5, 7}, {0, 4}, {1, 5}, {2, 6}, {3, 7}};
    const float inf = std::numeric_limits<float>::infinity();
    for (int d = 0; d < 3; ++d) in_a->lo[d] = in_b->lo[d] = inf, in_a->hi[d] = in_b->hi[d] = -inf;
    bool any = false;
    auto grow = [&](const float pa[3], const float pb[3]) {
        for (int d = 0; d < 3; ++d) {
            in_a->lo[d] = pa[d] < in_a->lo[d] ? pa[d] : in_a->lo[d];
            in_a->hi[d] = pa[d] > in_a->hi[d] ? pa[d] : in_a->hi[d];
            in_b->lo[d] = pb[d] < in_b->lo[d] ? pb[d] : in_b->lo[d];
            in_b->hi[d] = pb[d] > in_b->hi[d] ? pb[d] : in_b->hi[d];
        }
        any = true;
    };
    const float bqi[4] = {-bq[0], -bq[1], -bq[2], bq[3]};
    auto to_b = [&](const float p[3], float o[3]) {  // OrientedBox::transform_point_to_box_frame
        const float r[3] = {p[0] - bc[0], p[1] - bc[1], p[2] - bc[2]};
        host_qrot(bqi, r, o);
    };
    auto from_b = [&](const float p[3], float o[3]) {
        host_qrot(bq, p, o);
        for (int d = 0; d < 3; ++d) o[d] = bc[d] + o[d];
    };
    // corners of B: centre -/+ half width -/+ half height -/+ half depth along the columns of Mat3A::from_quat
    float ax[3][3];
    {
        const float x = bq[0], y = bq[1], z = bq[2], w = bq[3];
        const float x2 = x + x, y2 = y + y, z2 = z + z, xx = x * x2, xy = x * y2, xz = x * z2, yy = y * y2, yz = y * z2, zz = z * z2, wx = w * x2, wy = w * y2,
                    wz = w * z2;
        ax[0][0] = 1.0f - (yy + zz), ax[0][1] = xy + wz, ax[0][2] = xz - wy;
        ax[1][0] = xy - wz, ax[1][1] = 1.0f - (xx + zz), ax[1][2] = yz + wx;
        ax[2][0] = xz + wy, ax[2][1] = yz - wx, ax[2][2] = 1.0f - (xx + yy);
    }
    float corner[8][3];
    for (int c = 0; c < 8; ++c)
        for (int d = 0; d < 3; ++d) {
            const float hw = bh[0] * ax[0][d], hh = bh[1] * ax[1][d], hd = bh[2] * ax[2][d];
            float v = (c & 4) ? bc[d] + hw : bc[d] - hw;
            v = (c & 2) ? v + hh : v - hh;
            corner[c][d] = (c & 1) ? v + hd : v - hd;
        }
    for (const auto& e : E) {
        const float* s = corner[e[0]];
        const float v[3] = {corner[e[1]][0] - s[0], corner[e[1]][1] - s[1], corner[e[1]][2] - s[2]};
        float t0, t1;
        if (!host_subsegment(a, s, v, &t0, &t1)) continue;
        const float p0[3] = {s[0] + v[0] * t0, s[1] + v[1] * t0, s[2] + v[2] * t0}, p1[3] = {s[0] + v[0] * t1, s[1] + v[1] * t1, s[2] + v[2] * t1};
        float q0[3], q1[3];
        to_b(p0, q0);
        to_b(p1, q1);
        grow(p0, q0);
        grow(p1, q1);
    }
    float acorner[8][3];
    for (int c = 0; c < 8; ++c) {
        const float p[3] = {(c & 4) ? a.hi[0] : a.lo[0], (c & 2) ? a.hi[1] : a.lo[1], (c & 1) ? a.hi[2] : a.lo[2]};
        to_b(p, acorner[c]);
    }
    HBox self;
    for (int d = 0; d < 3; ++d) self.lo[d] = -bh[d], self.hi[d] = bh[d];
    for (const auto& e : E) {
        const float* s = acorner[e[0]];
        const float v[3] = {acorner[e[1]][0] - s[0], acorner[e[1]][1] - s[1], acorner[e[1]][2] - s[2]};
        float t0, t1;
        if (!host_subsegment(self, s, v, &t0, &t1)) continue;
        const float q0[3] = {s[0] + v[0] * t0, s[1] + v[1] * t0, s[2] + v[2] * t0}, q1[3] = {s[0] + v[0] * t1, s[1] + v[1] * t1, s[2] + v[2] * t1};
        float p0[3], p1[3];
        from_b(q0, p0);
        from_b(q1, p1);
        grow(p0, q0);
        grow(p1, q1);
    }
    return any;
}
// voxel_ranges_touching_aab on the occupied ranges, without the emptiness check the callers of `touched_ranges` want
void host_ranges(const uint32_t occ[12], const float lo_f[3], const float hi_f[3], long lo[3], long hi[3]) {
    for (int d = 0; d < 3; ++d) {
        const float fl = std::floor(lo_f[d]), ce = std::ceil(hi_f[d]);
        const long s = fl > 0.0f ? (fl < 2.0e9f ? (long)fl : 2000000000L) : 0, e = ce > 0.0f ? (ce < 2.0e9f ? (long)ce : 2000000000L) : 0;
        lo[d] = std::max<long>((long)occ[6 + 2 * d], s);
        hi[d] = std::min<long>((long)occ[7 + 2 * d], e);
    }
}
}  // namespace

// determine_voxel_ranges_encompassing_intersection (object/intersection.rs:706-746) from the two objects' occupied ranges and world -> object
// transforms; also transform_from_b_to_a = world_to_a * world_to_b.inverted() (impact_math/src/transform/isometry.rs:128-134, 200-205).
// The ranges are not checked for emptiness (the reference does not either). false: the occupied boxes do not meet.
static bool host_intersection_ranges(const ivx_grid* a, const uint32_t occ_a[12], const float rotation_a[4], const float translation_a[3], const ivx_grid* b,
                                     const uint32_t occ_b[12], const float rotation_b[4], const float translation_b[3], long ra_lo[3], long ra_hi[3],
                                     long rb_lo[3], long rb_hi[3], float q_ba[4], float t_ba[3]) {
    const float qbi[4] = {-rotation_b[0], -rotation_b[1], -rotation_b[2], rotation_b[3]};
    float tbi[3];
    host_qrot(qbi, translation_b, tbi);
    for (int d = 0; d < 3; ++d) tbi[d] = -tbi[d];
    host_qmul(rotation_a, qbi, q_ba);
    host_qrot(rotation_a, tbi, t_ba);
    for (int d = 0; d < 3; ++d) t_ba[d] += translation_a[d];
    HBox box_a, box_b;
    for (int d = 0; d < 3; ++d) {
        box_a.lo[d] = a->extent * (float)occ_a[6 + 2 * d];
        box_a.hi[d] = a->extent * (float)occ_a[7 + 2 * d];
        box_b.lo[d] = b->extent * (float)occ_b[6 + 2 * d];
        box_b.hi[d] = b->extent * (float)occ_b[7 + 2 * d];
    }
    float b_center[3], b_half[3], bc_in_a[3], bq_in_a[4];
    for (int d = 0; d < 3; ++d) {
        b_center[d] = 0.5f * (box_b.lo[d] + box_b.hi[d]);
        b_half[d] = 0.5f * (box_b.hi[d] - box_b.lo[d]);
    }
    host_qrot(q_ba, b_center, bc_in_a);
    for (int d = 0; d < 3; ++d) bc_in_a[d] += t_ba[d];
    const float ident[4] = {0.0f, 0.0f, 0.0f, 1.0f};
    host_qmul(q_ba, ident, bq_in_a);
    HBox in_a, in_b;
    if (!host_box_bounds(box_a, bc_in_a, bq_in_a, b_half, &in_a, &in_b)) return false;
    const float inv_a = 1.0f / a->extent, inv_b = 1.0f / b->extent;
    float na_lo[3], na_hi[3], nb_lo[3], nb_hi[3];
    for (int d = 0; d < 3; ++d) {
        na_lo[d] = inv_a * in_a.lo[d];
        na_hi[d] = inv_a * in_a.hi[d];
        nb_lo[d] = inv_b * (in_b.lo[d] + b_center[d]);
        nb_hi[d] = inv_b * (in_b.hi[d] + b_center[d]);
    }
    host_ranges(occ_a, na_lo, na_hi, ra_lo, ra_hi);
    host_ranges(occ_b, nb_lo, nb_hi, rb_lo, rb_hi);
    return true;
}

// what the two passes of a pair's mutual contacts need (for_each_mutual_voxel_object_contact, collidable.rs:859-1049): the intersection ranges of
// the two objects' occupied boxes in each other's frames, the probers' chunk ranges, the id prefix. *hit false: the boxes do not meet.
static int mutual_prepare(ivx_grid* a, const float rotation_a[4], const float translation_a[3], const float center_of_mass_a[3], ivx_grid* b,
                          const float rotation_b[4], const float translation_b[3], const float center_of_mass_b[3], uint64_t collidable_id_a,
                          uint64_t collidable_id_b, uint32_t body_a, uint32_t body_b, const float response[3], ivx_mutual_pass pass[2], bool* hit) {
    int rc;
    *hit = true;
    uint32_t occ_a[12], occ_b[12];
    if ((rc = reference_occupied(a, occ_a))) return rc;
    if ((rc = reference_occupied(b, occ_b))) return rc;
    long ra_lo[3], ra_hi[3], rb_lo[3], rb_hi[3];
    float q_ba[4], t_ba[3];
    if (!host_intersection_ranges(a, occ_a, rotation_a, translation_a, b, occ_b, rotation_b, translation_b, ra_lo, ra_hi, rb_lo, rb_hi, q_ba, t_ba)) {
        *hit = false;
        return IVX_OK;
    }
    {  // ContactID::from_two_u64_and_n_indices: the part that does not depend on the probe
        auto mix = [](uint64_t state) {
            state += 0x9E3779B97F4A7C15ull;
            uint64_t z = state;
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
            return z ^ (z >> 31);
        };
        pass[0].id_ab = pass[1].id_ab = mix(collidable_id_a ^ mix(collidable_id_b));
    }
    for (int w = 0; w < 2; ++w) {
        ivx_mutual_pass& p = pass[w];
        ivx_grid* prober = w ? b : a;
        ivx_grid* sampled = w ? a : b;
        const long* rlo = w ? rb_lo : ra_lo;
        const long* rhi = w ? rb_hi : ra_hi;
        const float* com_s = w ? center_of_mass_a : center_of_mass_b;
        const float inv_s = 1.0f / sampled->extent;
        for (int d = 0; d < 3; ++d) {
            p.center_s[d] = com_s[d] * inv_s;
            p.q_s[d] = (w ? rotation_a : rotation_b)[d];
            p.q_p[d] = (w ? rotation_b : rotation_a)[d];
            p.t_s[d] = (w ? translation_a : translation_b)[d];
            p.t_p[d] = (w ? translation_b : translation_a)[d];
            // aabb_from_voxel_ranges(prober's extent, ranges).expanded_about_center(object_a.voxel_extent()) — A's extent in both passes
            p.box_lo[d] = prober->extent * (float)rlo[d] - a->extent;
            p.box_hi[d] = prober->extent * (float)rhi[d] + a->extent;
            // chunk_range_encompassing_voxel_range (object.rs:3236-3240)
            p.clo[d] = (uint32_t)(rlo[d] / 16);
            p.chi[d] = (uint32_t)((rhi[d] + 15) / 16);
            p.response[d] = response[d];
        }
        p.q_s[3] = (w ? rotation_a : rotation_b)[3];
        p.q_p[3] = (w ? rotation_b : rotation_a)[3];
        p.negate = w;
        p.body_a = body_a;
        p.body_b = body_b;
    }
    return IVX_OK;
}

int ivx_mutual_voxel_object_contacts(ivx_grid* a, const float rotation_a[4], const float translation_a[3], const float center_of_mass_a[3], ivx_grid* b,
                                     const float rotation_b[4], const float translation_b[3], const float center_of_mass_b[3], uint64_t collidable_id_a,
                                     uint64_t collidable_id_b, uint32_t body_a, uint32_t body_b, const float response[3], ivx_contact* out, size_t cap,
                                     size_t* n_out) {
    const char* who = "ivx_mutual_voxel_object_contacts";
    IVX_REQUIRE(a && b && rotation_a && translation_a && center_of_mass_a && rotation_b && translation_b && center_of_mass_b && response && n_out &&
                    (out || cap == 0),
                IVX_ERR_INVALID, "%s: null argument", who);
    IVX_REQUIRE(a != b && a->ctx == b->ctx, IVX_ERR_INVALID, "%s: two different objects of one context are needed", who);
    for (ivx_grid* g : {a, b}) {
        IVX_REQUIRE(g->regions_valid, IVX_ERR_STATE, "%s: derived state must be current (ivx_derive_state + ivx_label_regions)", who);
        IVX_REQUIRE(g->mesh_valid && g->probes_serial == g->mesh_serial, IVX_ERR_STATE, "%s: collision probes must be current (ivx_collision_probes_recompute)",
                    who);
        IVX_REQUIRE(g->x_off == 0 && g->gx == g->cc[0] && !g->has_ghost[0] && !g->has_ghost[1], IVX_ERR_STATE,
                    "%s: not available on a slab of a decomposed grid", who);
    }
    *n_out = 0;
    int rc;
    ivx_mutual_pass pass[2];
    bool hit = false;
    if ((rc = mutual_prepare(a, rotation_a, translation_a, center_of_mass_a, b, rotation_b, translation_b, center_of_mass_b, collidable_id_a, collidable_id_b, body_a,
                             body_b, response, pass, &hit)))
        return rc;
    if (!hit) return IVX_OK;
    const uint32_t wg_a = (a->n_probe_points + 255u) / 256u, wg_b = (b->n_probe_points + 255u) / 256u, n_wg = wg_a + wg_b;
    if (n_wg == 0) return IVX_OK;
    // scratch (object A's): [counts n_wg][offsets n_wg + 1][contacts]
    const size_t off_offsets = (size_t)n_wg * 4, off_out = (off_offsets + ((size_t)n_wg + 1) * 4 + 63) & ~(size_t)63;
    if ((rc = ensure_dev_scratch(a, off_out + cap * sizeof(ivx_contact)))) return rc;
    char* base = static_cast<char*>(a->dev_scratch);
    uint32_t* d_counts = reinterpret_cast<uint32_t*>(base);
    uint32_t* d_offsets = reinterpret_cast<uint32_t*>(base + off_offsets);
    ivx_contact* d_out = reinterpret_cast<ivx_contact*>(base + off_out);
    const uint32_t cap32 = (uint32_t)std::min<size_t>(cap, 0xFFFFFFFFu);
    if ((rc = ivx_launch_mutual_pass(a, b, &pass[0], d_counts, nullptr, nullptr, cap32, 0))) return rc;
    if ((rc = ivx_launch_mutual_pass(b, a, &pass[1], d_counts + wg_a, nullptr, nullptr, cap32, 0))) return rc;
    if ((rc = ivx_launch_scan_counts(a->ctx, n_wg, d_counts, d_offsets))) return rc;
    if ((rc = ivx_launch_mutual_pass(a, b, &pass[0], nullptr, d_offsets, d_out, cap32, 1))) return rc;
    if ((rc = ivx_launch_mutual_pass(b, a, &pass[1], nullptr, d_offsets + wg_a, d_out, cap32, 1))) return rc;
    uint32_t total = 0;
    if ((rc = d2h(a, &total, d_offsets + n_wg, sizeof(total)))) return rc;
    *n_out = total;
    IVX_REQUIRE(total <= cap, IVX_ERR_CAPACITY, "%s: %u contacts exceed the capacity %zu", who, total, cap);
    if (total && (rc = d2h(a, out, d_out, (size_t)total * sizeof(ivx_contact)))) return rc;
    return IVX_OK;
}

// The same for a LIST of pairs in the launches of one (many.hpp) — the reference's narrow phase visits every pair of voxel objects the broad
// phase hands it (collidable.rs:859-1049 per pair) —: per pair count (A's probes in B's field) | count (B's in A's) | scan, recorded for all
// pairs and issued merged, the totals written by the scans into host-mapped memory; one wait; then the emit passes of all pairs into one
// host-mapped list; one wait. out_offsets[i] .. out_offsets[i + 1]: pair i's manifold, the list the single-pair call returns.
int ivx_mutual_voxel_object_contacts_many(const ivx_mutual_query* queries, size_t n, ivx_contact* out, size_t cap, uint32_t* out_offsets) {
    const char* who = "ivx_mutual_voxel_object_contacts_many";
    IVX_REQUIRE(out_offsets, IVX_ERR_INVALID, "%s: null argument", who);
    out_offsets[0] = 0;
    if (n == 0) return IVX_OK;
    IVX_REQUIRE(queries && (out || cap == 0), IVX_ERR_INVALID, "%s: null argument", who);
    ivx_ctx* c = queries[0].a ? queries[0].a->ctx : nullptr;
    for (size_t i = 0; i < n; ++i) {
        const ivx_mutual_query& q = queries[i];
        IVX_REQUIRE(q.a && q.b && q.a != q.b && q.a->ctx == c && q.b->ctx == c, IVX_ERR_INVALID, "%s: pair %zu: two different objects of the call's context are needed", who, i);
        for (ivx_grid* g : {q.a, q.b}) {
            IVX_REQUIRE(g->regions_valid, IVX_ERR_STATE, "%s: pair %zu: derived state must be current (ivx_derive_state + ivx_label_regions)", who, i);
            IVX_REQUIRE(g->mesh_valid && g->probes_serial == g->mesh_serial, IVX_ERR_STATE, "%s: pair %zu: collision probes must be current (ivx_collision_probes_recompute)", who, i);
            IVX_REQUIRE(g->x_off == 0 && g->gx == g->cc[0] && !g->has_ghost[0] && !g->has_ghost[1], IVX_ERR_STATE, "%s: not available on a slab of a decomposed grid", who);
        }
    }
    IVX_REQUIRE(!ivx_many_recording(), IVX_ERR_STATE, "%s: not inside an ivx_many_begin bracket (the call waits for its own phases)", who);
    int rc;
    hipStream_t s = c->stream;
    struct Pair {
        ivx_mutual_pass pass[2];
        uint32_t wg_a, wg_b;
        size_t off;  // of the pair's counts / offsets in the context's scratch (u32 words)
        bool hit;
    };
    static thread_local std::vector<Pair> pr;
    pr.assign(n, Pair{});
    size_t words = 0;
    for (size_t i = 0; i < n; ++i) {
        const ivx_mutual_query& q = queries[i];
        Pair& p = pr[i];
        if ((rc = mutual_prepare(q.a, q.rotation_a, q.translation_a, q.center_of_mass_a, q.b, q.rotation_b, q.translation_b, q.center_of_mass_b, q.collidable_id_a,
                                 q.collidable_id_b, q.body_a, q.body_b, q.response, p.pass, &p.hit)))
            return rc;
        p.wg_a = (q.a->n_probe_points + 255u) / 256u, p.wg_b = (q.b->n_probe_points + 255u) / 256u;
        if (p.wg_a + p.wg_b == 0) p.hit = false;
        p.off = words;
        if (p.hit) words += 2 * (size_t)(p.wg_a + p.wg_b) + 16;
    }
    // counts and offsets of all pairs: one block of the context's device scratch; totals and contacts: its pinned block
    if (c->dev_scratch_bytes < words * 4 + 64) {
        IVX_HIP_CHECK(ivx_stream_sync(s));
        if (c->dev_scratch) (void)hipFree(c->dev_scratch);
        c->dev_scratch = nullptr;
        c->dev_scratch_bytes = 0;
        const size_t want = std::max<size_t>(2 * (words * 4 + 64), 1 << 20);
        IVX_HIP_CHECK(hipMalloc(&c->dev_scratch, want));
        c->dev_scratch_bytes = want;
    }
    if ((rc = ctx_pinned_scratch(c, ((n * 4 + 63) & ~(size_t)63) + 4096))) return rc;
    uint32_t* totals = static_cast<uint32_t*>(c->pinned_scratch);
    memset(totals, 0, n * 4);
    uint32_t* totals_dev = static_cast<uint32_t*>(c->pinned_scratch_dev);
    uint32_t* base = static_cast<uint32_t*>(c->dev_scratch);
    std::vector<ivx_grid*> chain(n);  // (the recorder's chains go by pair: the first object of each stands for it)
    for (size_t i = 0; i < n; ++i) chain[i] = queries[i].a;
    if ((rc = many_phase(chain.data(), n, [&](size_t i) -> int {
             const Pair& p = pr[i];
             if (!p.hit) return IVX_OK;
             const ivx_mutual_query& q = queries[i];
             const uint32_t n_wg = p.wg_a + p.wg_b;
             uint32_t* d_counts = base + p.off;
             uint32_t* d_offsets = d_counts + n_wg;
             int r;
             if ((r = ivx_launch_mutual_pass(q.a, q.b, &p.pass[0], d_counts, nullptr, nullptr, 0u, 0))) return r;
             if ((r = ivx_launch_mutual_pass(q.b, q.a, &p.pass[1], d_counts + p.wg_a, nullptr, nullptr, 0u, 0))) return r;
             return ivx_launch_scan_counts(c, n_wg, d_counts, d_offsets, totals_dev + i, q.a);
         })))
        return rc;
    IVX_HIP_CHECK(ivx_stream_sync(s));
    size_t run = 0;
    for (size_t i = 0; i < n; ++i) {
        out_offsets[i] = (uint32_t)run;
        run += totals[i];
    }
    out_offsets[n] = (uint32_t)run;
    IVX_REQUIRE(run <= cap, IVX_ERR_CAPACITY, "%s: %zu contacts exceed the capacity %zu", who, run, cap);
    if (run == 0) return IVX_OK;
    static thread_local std::vector<uint32_t> counts;
    counts.assign(totals, totals + n);
    if ((rc = ctx_pinned_scratch(c, run * sizeof(ivx_contact)))) return rc;
    ivx_contact* list_dev = static_cast<ivx_contact*>(c->pinned_scratch_dev);
    if ((rc = many_phase(chain.data(), n, [&](size_t i) -> int {
             const Pair& p = pr[i];
             if (!counts[i]) return IVX_OK;
             const ivx_mutual_query& q = queries[i];
             const uint32_t* d_offsets = base + p.off + (p.wg_a + p.wg_b);
             int r;
             if ((r = ivx_launch_mutual_pass(q.a, q.b, &p.pass[0], nullptr, d_offsets, list_dev + out_offsets[i], counts[i], 1))) return r;
             return ivx_launch_mutual_pass(q.b, q.a, &p.pass[1], nullptr, d_offsets + p.wg_a, list_dev + out_offsets[i], counts[i], 1);
         })))
        return rc;
    IVX_HIP_CHECK(ivx_stream_sync(s));
    memcpy(out, c->pinned_scratch, run * sizeof(ivx_contact));
    return IVX_OK;
}

// apply_mutual_absorption (interaction/absorption.rs:891-1079)
int ivx_absorb_mutual(ivx_grid* a, const float rotation_a[4], const float translation_a[3], const float densities_a[256], ivx_grid* b, const float rotation_b[4],
                      const float translation_b[3], const float densities_b[256], float smoothness, ivx_absorb_result* out_a, ivx_absorb_result* out_b,
                      uint8_t* invalidated_chunks_a, uint8_t* invalidated_chunks_b) {
    const char* who = "ivx_absorb_mutual";
    IVX_REQUIRE(a && b && rotation_a && translation_a && densities_a && rotation_b && translation_b && densities_b && out_a && out_b, IVX_ERR_INVALID,
                "%s: null argument", who);
    IVX_REQUIRE(a != b && a->ctx == b->ctx, IVX_ERR_INVALID, "%s: two different objects of one context are needed", who);
    IVX_REQUIRE(smoothness >= 0.0f, IVX_ERR_INVALID, "%s: negative smoothness", who);
    for (ivx_grid* g : {a, b}) {
        IVX_REQUIRE(g->regions_valid, IVX_ERR_STATE, "%s: derived state and regions must be current (ivx_derive_state + ivx_label_regions)", who);
        IVX_REQUIRE(g->x_off == 0 && g->gx == g->cc[0] && !g->has_ghost[0] && !g->has_ghost[1], IVX_ERR_STATE,
                    "%s: not available on a slab of a decomposed grid", who);
    }
    memset(out_a, 0, sizeof(*out_a));
    memset(out_b, 0, sizeof(*out_b));
    if (invalidated_chunks_a) memset(invalidated_chunks_a, 0, a->n_chunks);
    if (invalidated_chunks_b) memset(invalidated_chunks_b, 0, b->n_chunks);
    int rc;
    uint32_t occ_a[12], occ_b[12];
    if ((rc = reference_occupied(a, occ_a))) return rc;
    if ((rc = reference_occupied(b, occ_b))) return rc;
    long ra_lo[3], ra_hi[3], rb_lo[3], rb_hi[3];
    float q_ba[4], t_ba[3];
    if (!host_intersection_ranges(a, occ_a, rotation_a, translation_a, b, occ_b, rotation_b, translation_b, ra_lo, ra_hi, rb_lo, rb_hi, q_ba, t_ba)) return IVX_OK;
    // the snapshot of A's distances covers A's overlap ranges padded by one B voxel (in A voxels), inside A's grid
    const long pad = (long)std::ceil(b->extent * (1.0f / a->extent));
    int32_t s_lo[3], s_hi[3], vb_lo[3], vb_hi[3];
    bool a_runs = true, b_runs = true;
    size_t snap_bytes = 1;
    for (int d = 0; d < 3; ++d) {
        s_lo[d] = (int32_t)std::max<long>(0, ra_lo[d] - pad);
        s_hi[d] = (int32_t)std::min<long>(ra_hi[d] + pad, (long)a->cc[d] * 16);
        vb_lo[d] = (int32_t)rb_lo[d];
        vb_hi[d] = (int32_t)rb_hi[d];
        a_runs = a_runs && s_lo[d] < s_hi[d];
        b_runs = b_runs && vb_lo[d] < vb_hi[d];
        snap_bytes *= (size_t)std::max(0, s_hi[d] - s_lo[d]);
    }
    if (!a_runs) snap_bytes = 0;
    if (!a_runs && !b_runs) return IVX_OK;
    // scratch (A's): per object [10 f64 removed moments][256 u32 by type][2 u32 counters][pad][n_chunks u32 touched ranges], then the two density
    // tables, then the snapshot
    const size_t off_type = 80, off_cnt = off_type + 1024, off_touch = off_cnt + 16;
    // (the touched ranges are indexed by the chunk's position in the object's chunk box; the boxes are known further down, a whole grid bounds them)
    const size_t blk_a = (off_touch + (size_t)a->n_chunks * 4 + 255) & ~(size_t)255, blk_b = (off_touch + (size_t)b->n_chunks * 4 + 255) & ~(size_t)255;
    const size_t off_dens = blk_a + blk_b, off_snap = off_dens + 2048;
    if ((rc = ensure_dev_scratch(a, off_snap + snap_bytes + 256))) return rc;
    char* base = static_cast<char*>(a->dev_scratch);
    IVX_HIP_CHECK(ivx_memset_async(base, 0, off_dens, a->ctx->stream));
    if ((rc = h2d(a, base + off_dens, densities_a, 1024))) return rc;
    if ((rc = h2d(a, base + off_dens + 1024, densities_b, 1024))) return rc;
    int8_t* d_snap = reinterpret_cast<int8_t*>(base + off_snap);
    uint32_t lo_a[3] = {0, 0, 0}, cc_a[3] = {0, 0, 0}, lo_b[3] = {0, 0, 0}, cc_b[3] = {0, 0, 0};
    if (a_runs) {
        for (int d = 0; d < 3; ++d) {
            lo_a[d] = (uint32_t)s_lo[d] / 16u;
            cc_a[d] = ((uint32_t)s_hi[d] + 15u) / 16u - lo_a[d];
        }
        if ((rc = ivx_launch_sdf_snapshot(a, s_lo, s_hi, d_snap))) return rc;
        if ((rc = ivx_launch_absorb_mutual(a, 0, lo_a, cc_a, s_lo, s_hi, b, nullptr, s_lo, s_hi, q_ba, t_ba, smoothness, reinterpret_cast<float*>(base + off_dens),
                                           reinterpret_cast<double*>(base), reinterpret_cast<uint32_t*>(base + off_type),
                                           reinterpret_cast<uint32_t*>(base + off_cnt), reinterpret_cast<uint32_t*>(base + off_touch))))
            return rc;
    }
    if (b_runs) {
        for (int d = 0; d < 3; ++d) {
            lo_b[d] = (uint32_t)vb_lo[d] / 16u;
            cc_b[d] = ((uint32_t)vb_hi[d] + 15u) / 16u - lo_b[d];
        }
        char* bb = base + blk_a;
        if ((rc = ivx_launch_absorb_mutual(b, 1, lo_b, cc_b, vb_lo, vb_hi, a, d_snap, s_lo, s_hi, q_ba, t_ba, smoothness,
                                           reinterpret_cast<float*>(base + off_dens + 1024), reinterpret_cast<double*>(bb),
                                           reinterpret_cast<uint32_t*>(bb + off_type), reinterpret_cast<uint32_t*>(bb + off_cnt),
                                           reinterpret_cast<uint32_t*>(bb + off_touch))))
            return rc;
    }
    std::vector<char> hostbuf(off_dens);
    if ((rc = d2h(a, hostbuf.data(), base, off_dens))) return rc;
    if (a_runs && (rc = rederive(a))) return rc;
    if (b_runs && (rc = rederive(b))) return rc;
    for (int w = 0; w < 2; ++w) {
        ivx_grid* g = w ? b : a;
        if (!(w ? b_runs : a_runs)) continue;
        const char* hb = hostbuf.data() + (w ? blk_a : 0);
        ivx_absorb_result* out = w ? out_b : out_a;
        uint8_t* inval = w ? invalidated_chunks_b : invalidated_chunks_a;
        const uint32_t* lo = w ? lo_b : lo_a;
        const uint32_t* cc = w ? cc_b : cc_a;
        const double* rem = reinterpret_cast<const double*>(hb);
        const double e = (double)g->extent, e3 = e * e * e, e4 = e3 * e, e5 = e4 * e;
        const double f[10] = {e3, 0.5 * e4, 0.5 * e4, 0.5 * e4, e5 / 3.0, e5 / 3.0, e5 / 3.0, 0.25 * e5, 0.25 * e5, 0.25 * e5};
        for (int q = 0; q < 10; ++q) out->removed_moments[q] = rem[q] * f[q];
        const uint32_t* by_type = reinterpret_cast<const uint32_t*>(hb + off_type);
        uint64_t emptied = 0;
        for (int t = 0; t < 256; ++t) emptied += by_type[t];
        out->emptied_voxels = emptied;
        const uint32_t* cnt = reinterpret_cast<const uint32_t*>(hb + off_cnt);
        out->touched_chunks = cnt[0];
        out->removed_chunks = cnt[1];
        if (cnt[1]) g->occ_ref_valid = 0;  // (intersection.rs:255-257)
        if (!inval) continue;
        const uint32_t* touched = reinterpret_cast<const uint32_t*>(hb + off_touch);  // handle_chunk_voxels_modified (intersection.rs:560-598)
        for (uint32_t i = lo[0]; i < lo[0] + cc[0]; ++i)
            for (uint32_t j = lo[1]; j < lo[1] + cc[1]; ++j)
                for (uint32_t k = lo[2]; k < lo[2] + cc[2]; ++k) {
                    const uint32_t c = (i * g->cc[1] + j) * g->cc[2] + k;
                    const uint32_t wd = touched[((i - lo[0]) * cc[1] + (j - lo[1])) * cc[2] + (k - lo[2])];
                    if (!wd) continue;
                    inval[c] = 1;
                    const uint32_t idx[3] = {i, j, k};
                    for (int d = 0; d < 3; ++d) {
                        const uint32_t rlo = (wd >> (4 * d)) & 15u, rhi = ((wd >> (12 + 4 * d)) & 15u) + 1u;
                        uint32_t n3[3] = {i, j, k};
                        if (idx[d] > 0 && rlo < 2) {
                            n3[d] = idx[d] - 1;
                            inval[(n3[0] * g->cc[1] + n3[1]) * g->cc[2] + n3[2]] = 1;
                        }
                        if (idx[d] + 1 < g->cc[d] && 16u - rhi < 2) {
                            n3[d] = idx[d] + 1;
                            inval[(n3[0] * g->cc[1] + n3[1]) * g->cc[2] + n3[2]] = 1;
                        }
                    }
                }
    }
    return IVX_OK;
}

int ivx_grid_set_sdf_program(ivx_grid* g, const ivx_sdf_processed_node* nodes, size_t n_nodes, uint32_t stack_size, const uint32_t grid_shape[3],
                             const float shifted_grid_center[3], uint8_t voxel_type) {
    IVX_REQUIRE(g && grid_shape && shifted_grid_center && (n_nodes == 0 || nodes), IVX_ERR_INVALID, "ivx_grid_set_sdf_program: null argument");
    int depth = 0, max_depth = 0;
    for (size_t i = 0; i < n_nodes; ++i) {
        const uint32_t k = nodes[i].kind;
        IVX_REQUIRE(k <= 9 && k != 6, IVX_ERR_INVALID, "ivx_grid_set_sdf_program: unsupported node kind %u", k);
        if (k <= 2) max_depth = std::max(max_depth, ++depth);
        else if (k >= 7) {
            IVX_REQUIRE(depth >= 2, IVX_ERR_INVALID, "ivx_grid_set_sdf_program: malformed node program");
            --depth;
        } else if (k == 5) {
            IVX_REQUIRE(depth >= 1, IVX_ERR_INVALID, "ivx_grid_set_sdf_program: malformed node program");
        }
    }
    IVX_REQUIRE(n_nodes == 0 || depth == 1, IVX_ERR_INVALID, "ivx_grid_set_sdf_program: malformed node program (final stack depth %d)", depth);
    IVX_REQUIRE(n_nodes == 0 || (uint32_t)max_depth <= stack_size, IVX_ERR_INVALID, "ivx_grid_set_sdf_program: stack_size too small");
    for (int d = 0; d < 3; ++d) {
        const uint32_t cap = (d == 0 ? g->gx : g->cc[d]) * 16u;
        IVX_REQUIRE(grid_shape[d] <= cap, IVX_ERR_INVALID, "ivx_grid_set_sdf_program: grid shape exceeds the chunk grid along axis %d", d);
    }
    {  // a pre-pass that runs ahead reads the resident program: wait for it, drop what it wrote
        const int rc_a = ivx_sampler_ahead_cancel(g);
        if (rc_a) return rc_a;
    }
    if (n_nodes > g->prog_cap) {
        IVX_HIP_CHECK(ivx_stream_sync(g->ctx->stream));
        if (g->prog_nodes) (void)hipFree(g->prog_nodes);
        g->prog_nodes = nullptr;
        g->prog_cap = 0;
        int rc = dev_alloc(&g->prog_nodes, n_nodes);
        if (rc) return rc;
        g->prog_cap = (uint32_t)n_nodes;
    }
    std::vector<ivx_sdf_processed_node> annotated(nodes, nodes + n_nodes);
    ivx_sdf_annotate_host(annotated.data(), n_nodes);
    int rc = h2d(g, g->prog_nodes, annotated.data(), n_nodes * sizeof(ivx_sdf_processed_node));
    if (rc) return rc;
    g->prog_n = (uint32_t)n_nodes;
    g->prog_stack = (uint32_t)max_depth;
    for (int d = 0; d < 3; ++d) {
        g->prog_shape[d] = grid_shape[d];
        g->prog_center[d] = shifted_grid_center[d];
    }
    g->prog_type = voxel_type;
    g->eval_len_valid = 0;
    g->eval_len_pending = 0;
    return IVX_OK;
}

int ivx_grid_set_densities(ivx_grid* g, const float densities[256]) {
    IVX_REQUIRE(g && densities, IVX_ERR_INVALID, "ivx_grid_set_densities: null argument");
    // (stream-ordered, no wait: the copy reads the grid's own host copy of the table, which lives as long as the grid — a table set again
    // before the copy has run may reach the device twice, in the right order. Setting the tables of the fragments of an impact used to cost a
    // pinned allocation and two waits each.)
    memcpy(g->dens_host, densities, sizeof(g->dens_host));
    if (!ivx_many_upload(g->ctx, g, g->dens_dev, g->dens_host, sizeof(g->dens_host)))
        IVX_HIP_CHECK(ivx_memcpy_async(g->dens_dev, g->dens_host, sizeof(g->dens_host), hipMemcpyHostToDevice, g->ctx->stream));
    g->has_dens = 1;
    return IVX_OK;
}

// Timed slots of a step (ivx_step_result::stage_ms): 0 sample (k_sdf_super, k_sdf_prepass, k_sdf_eval), 1 derive (k_chunk_pre, k_derive:
// flags, chunk state, chunk-local regions, chunk moments), 2 k_step_post1 (mesher count | region merge by columns | occupied slots
// | moment partial sums), 3 k_step_post2 (exact local numbering -> multi-region merge | mesher scan | moments and occupied ranges
// final), 4 k_step_emit (region forest flatten | mesher emit), 5 k_step_assign (component ids); 6..9 unused. Stage timing costs two
// event records per slot on the stream; ivx_grid_set_stage_timing(g, 0) turns it off.
static int ensure_pairs(ivx_grid* g);
// `slab_nbr_ids` / `slab_record`: the slab protocol's remesh phase (ivx_slab_remesh_enqueue) — the pass over the neighbour's face ids and the
// slab's record ride in the phase's own launches instead of taking two more
// `part` (slab protocol, ivx_voxel_step_enqueue_part): bit 0 = the call's sample and derive sweeps, bit 1 = everything behind them; the two
// halves of ONE call enqueued apart, so that the slab's face planes can be packed and sent between them.
static int step_enqueue(ivx_grid* g, uint32_t stages, const uint16_t* slab_nbr_ids, void* slab_record, uint32_t part) {
    IVX_REQUIRE(g, IVX_ERR_INVALID, "ivx_voxel_step_enqueue: null grid");
    const bool front = (part & 1u) != 0u, back = (part & 2u) != 0u;
    ivx_many_other_context other_(g->ctx);
    IVX_REQUIRE(!(stages & IVX_STAGE_SAMPLE) || g->prog_n > 0, IVX_ERR_STATE, "ivx_voxel_step: no SDF program resident (ivx_grid_set_sdf_program)");
    IVX_REQUIRE(!(stages & IVX_STAGE_INERTIA) || g->has_dens, IVX_ERR_STATE, "ivx_voxel_step: no densities resident (ivx_grid_set_densities)");
    hipStream_t s = g->ctx->stream;
    g->ahead_unordered_ok = g->pending_stages == 0;  // (sample-ahead: whatever was enqueued before has been collected)
    // stages enqueued after a record change the step's results: the block the record role wrote is stale (this call sets the flag again
    // further down when it carries the slab record itself)
    g->results_in_block = 0;
    if (!g->ev_ready) {
        for (int i = 0; i < 2 * IVX_N_TIMED_STAGES; ++i) IVX_HIP_CHECK(hipEventCreate(&g->ev[i]));
        g->ev_ready = 1;
    }
    if (!g->result_host) {
        IVX_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&g->result_host), 64 * sizeof(uint32_t), hipHostMallocMapped));
        memset(g->result_host, 0, 64 * sizeof(uint32_t));
        IVX_HIP_CHECK(hipHostGetDevicePointer(reinterpret_cast<void**>(&g->result_host_dev), g->result_host, 0));
    }
    int rc;
    if (stages & IVX_STAGE_SAMPLE)
        if ((rc = ivx_sampler_buffers(g))) return rc;
    // Scratch word groups the stages of this call start from. They are preset by the call's first kernel (k_sdf_super or
    // k_chunk_pre host that role); a call that starts with neither gets a preset launch of its own. Only the requested stages'
    // groups are touched: a phase of the multi-GPU protocol must not wipe what an earlier phase of the same step left.
    uint32_t preset_in_sample = 0, preset_in_derive = 0;
    if (front) {  // (the second half of a call enqueued in two parts finds its groups preset by the first half's kernels)
        uint32_t need = 0;
        if (stages & IVX_STAGE_SAMPLE) need |= IVX_SCRATCH_EVAL;
        if (stages & IVX_STAGE_REGIONS) need |= IVX_SCRATCH_REGIONS;
        if (stages & IVX_STAGE_REMESH) need |= IVX_SCRATCH_SN;
        // (ivx_step_preset_ahead: groups of the NEXT call, preset by this call's first kernel so that a call without one of its own — the
        // remesh phase of the slab protocol — needs no preset launch)
        const uint32_t fresh = g->preset_fresh;
        g->preset_fresh = 0;
        uint32_t ahead = 0;
        if (stages & (IVX_STAGE_SAMPLE | IVX_STAGE_DERIVE)) {
            ahead = g->preset_ahead & ~need;
            g->preset_ahead = 0;
        }
        if (stages & IVX_STAGE_SAMPLE) preset_in_sample = need | ahead;
        else if (stages & IVX_STAGE_DERIVE) preset_in_derive = need | ahead;
        else if ((rc = ivx_launch_step_preset(g, need & ~fresh))) return rc;
        g->preset_fresh = ahead;
    }
    // a slot's duration runs from the stop event of the slot enqueued just before it, when there is one
    const uint32_t timing = ~g->stage_timing_off;  // slots with event records
    hipEvent_t* last_stop = nullptr;
#define T0(i)                                                      \
    if (!((timing >> (i)) & 1u)) last_stop = nullptr;              \
    else {                                                         \
        if (last_stop) g->ev_start_ref[i] = last_stop;             \
        else {                                                     \
            IVX_HIP_CHECK(ivx_event_record(g->ev[2 * (i)], s));      \
            g->ev_start_ref[i] = &g->ev[2 * (i)];                  \
        }                                                          \
    }
#define T1(i)                                                      \
    if ((timing >> (i)) & 1u) {                                    \
        IVX_HIP_CHECK(ivx_event_record(g->ev[2 * (i) + 1], s));      \
        last_stop = &g->ev[2 * (i) + 1];                           \
        g->timed_mask |= 1u << (i);                                \
    }
    // derive runs the chunk-local region labelling and the chunk moments in the same sweep when those stages are part of this call
    const uint32_t fused_parts = (stages & IVX_STAGE_DERIVE) ? (((stages & IVX_STAGE_REGIONS) ? IVX_PART_REGIONS : 0u) |
                                                                 ((stages & IVX_STAGE_INERTIA) ? IVX_PART_MOMENTS : 0u))
                                                              : 0u;
    if (front && (stages & IVX_STAGE_SAMPLE)) {
        T0(0);
        if ((rc = ivx_launch_sdf_sample(g, g->prog_nodes, g->prog_n, g->prog_stack, g->prog_shape, g->prog_center, g->prog_type, preset_in_sample, true))) return rc;
        T1(0);
        g->eval_len_pending = 1;  // (the list lengths reach the result block once a derive sweep has rolled the counters over)
        g->occ_ref_valid = 0;
        g->bbox_valid = 0;
        g->mesh_valid = 0;
        g->mesh_built = 0;  // a newly sampled object: whatever mesh the buffers hold is not a stale version of this one
        g->regions_valid = 0;
    }
    if (front && (stages & IVX_STAGE_DERIVE)) {
        T0(1);
        if ((rc = ivx_launch_derive(g, fused_parts, preset_in_derive))) return rc;
        T1(1);
        if (g->eval_len_pending == 1) g->eval_len_pending = 2;
    }
    const uint32_t post = back ? stages & (IVX_STAGE_OCCUPIED | IVX_STAGE_REGIONS | IVX_STAGE_REMESH | IVX_STAGE_INERTIA) : 0u;
    if (post) {
        // stages that were not swept inside k_derive get their stand-alone per-chunk kernels first (a call without the derive stage)
        if ((stages & IVX_STAGE_REGIONS) && !(fused_parts & IVX_PART_REGIONS)) {
            // (level 1 over the active list; the exact numbering of multi-region chunks leads k_step_post2 below, so
            // only the list-driven labelling kernel is launched here: ivx_launch_ccl_local would run both)
            if (!g->regions_labelled_locally && (rc = ivx_launch_ccl_local_only(g))) return rc;
        }
        if ((stages & IVX_STAGE_INERTIA) && !(fused_parts & IVX_PART_MOMENTS))
            if ((rc = ivx_launch_inertia_dense(g))) return rc;
        const bool fused_assign = ivx_step_assign_fits(g);
        T0(2);
        if ((rc = ivx_launch_step_post1(g, post))) return rc;
        T1(2);
        T0(3);
        if (slab_record) {
            if ((rc = ensure_pairs(g))) return rc;
            if (slab_nbr_ids && !g->pairs_zeroed) IVX_HIP_CHECK(ivx_memset_async(g->pairs_dev, 0, (4 + 128) * sizeof(uint32_t), s));
            g->pairs_zeroed = 0;
        }
        if ((rc = ivx_launch_step_post2(g, post, slab_nbr_ids))) return rc;
        T1(3);
        // no host round trip for the mesh sizes: the emit pass writes into the buffers of the previous step and skips what
        // does not fit; ivx_voxel_step_collect grows the buffers and repeats the pass in that (rare) case
        if (post & (IVX_STAGE_REGIONS | IVX_STAGE_REMESH)) {
            T0(4);
            if (fused_assign) {
                if ((rc = ivx_launch_step_emit(g, post, (post & IVX_STAGE_REGIONS) != 0, slab_record, slab_nbr_ids != nullptr))) return rc;
                if (slab_record) g->results_in_block = 1;
            } else {  // more than 524 288 chunks: the region resolve takes its stand-alone path
                if ((post & IVX_STAGE_REMESH) && (rc = ivx_launch_step_emit(g, IVX_STAGE_REMESH))) return rc;
                if ((post & IVX_STAGE_REGIONS) && (rc = ivx_launch_ccl_resolve(g))) return rc;
            }
            T1(4);
        }
        // (sample-ahead: the next sample stage's pre-pass goes behind the step's last big launch, beside the small ones that follow)
        if (!g->ahead_unordered_ok && (rc = ivx_sampler_launch_ahead(g, true))) return rc;
        if ((post & IVX_STAGE_REGIONS) && fused_assign) {
            T0(5);
            if ((rc = ivx_launch_step_assign(g, (post & IVX_STAGE_REMESH) != 0))) return rc;
            T1(5);
        }
        if (post & IVX_STAGE_REMESH) {
            g->mesh_valid = 0;
            g->scratch_dirty |= IVX_SCRATCH_SN;
        }
        if (post & IVX_STAGE_REGIONS) g->scratch_dirty |= IVX_SCRATCH_REGIONS;
    }
#undef T0
#undef T1
    if (back && (rc = ivx_sampler_launch_ahead(g, !g->ahead_unordered_ok))) return rc;  // (a call without the stages behind the derive sweep)
    g->pending_stages |= stages;
    return IVX_OK;
}

int ivx_voxel_step_enqueue(ivx_grid* g, uint32_t stages) { return step_enqueue(g, stages, nullptr, nullptr); }
}  // extern "C"
// (slab_comm.cpp) one call's launches in two parts: 1 = its sample and derive sweeps, 2 = what follows them
int ivx_voxel_step_enqueue_part(ivx_grid* g, uint32_t stages, uint32_t part) { return step_enqueue(g, stages, nullptr, nullptr, part); }
extern "C" {

// The remesh phase of the slab protocol in the step's own launches: count | — then scan | the component pairs across the upper x face (from
// `neighbour_face_ids`, the ids behind the neighbour's face planes of the second exchange; null for the last slab) — then emit | the slab's
// record into `device_record` and the step's small results into the grid's host-mapped block — then the mesher's general pass. Four launches
// where face pairs, the remesh stage and the record took six; ivx_voxel_step_collect then has its results without a launch of its own (the
// caller waits for the stream first: the slab protocol's doorbell).
int ivx_slab_remesh_enqueue(ivx_grid* g, const void* neighbour_face_ids, void* device_record) {
    IVX_REQUIRE(g && device_record, IVX_ERR_INVALID, "ivx_slab_remesh_enqueue: null argument");
    if (!ivx_step_assign_fits(g)) {  // (grids beyond the fused launches' reach: the separate passes)
        int rc;
        if (neighbour_face_ids && (rc = ivx_region_face_pairs_enqueue(g, 1, neighbour_face_ids))) return rc;
        if ((rc = step_enqueue(g, IVX_STAGE_REMESH, nullptr, nullptr))) return rc;
        if ((rc = ivx_step_record_enqueue(g, device_record))) return rc;
        if (g->record_head_copy)
            IVX_HIP_CHECK(ivx_memcpy_async(g->record_head_copy, device_record, (size_t)g->record_head_words * 8, hipMemcpyDeviceToDevice, g->ctx->stream));
        return IVX_OK;
    }
    g->pairs_enqueued = 0;
    return step_enqueue(g, IVX_STAGE_REMESH, static_cast<const uint16_t*>(neighbour_face_ids), device_record);
}

// Sample-ahead: with `on`, every sample stage under the resident program also enqueues the NEXT sample stage's pre-pass — on the context's
// second stream, behind its own evaluator —, and the next sample stage starts at its evaluator. For callers that sample the same program
// step after step (the bench's headline step); the results are the same bytes either way. Off (the default): the pre-pass is the stage's
// first kernel. Turning it off drops a pre-pass that is under way.
int ivx_grid_set_sample_ahead(ivx_grid* g, int on) {
    IVX_REQUIRE(g, IVX_ERR_INVALID, "ivx_grid_set_sample_ahead: null grid");
    g->ahead_on = on ? 1 : 0;
    if (!on) return ivx_sampler_ahead_cancel(g);
    return IVX_OK;
}

int ivx_grid_set_stage_timing(ivx_grid* g, uint32_t slot_mask) {
    IVX_REQUIRE(g, IVX_ERR_INVALID, "ivx_grid_set_stage_timing: null grid");
    g->stage_timing_off = ~slot_mask;
    return IVX_OK;
}

// How long ivx_voxel_step_collect polls the doorbell before it falls back to hipStreamSynchronize (IVX_COLLECT_SPIN_US, default 2000;
// 0 = never poll).
static uint64_t collect_spin_ns() {
    static const uint64_t ns = [] {
        const char* e = getenv("IVX_COLLECT_SPIN_US");
        const long us = e ? strtol(e, nullptr, 10) : 2000;
        return (uint64_t)(us < 0 ? 0 : us) * 1000ull;
    }();
    return ns;
}

// the launch half of ivx_voxel_step_collect: the gather of the step's results (and whatever rides on it) goes on the stream — or into the
// batch being recorded —, the wait is left to the collect
static int ivx_step_collect_launch(ivx_grid* g) {
    if (g->results_in_block || g->gather_launched) return IVX_OK;
    if (!g->result_host) {
        IVX_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&g->result_host), 64 * sizeof(uint32_t), hipHostMallocMapped));
        memset(g->result_host, 0, 64 * sizeof(uint32_t));
        IVX_HIP_CHECK(hipHostGetDevicePointer(reinterpret_cast<void**>(&g->result_host_dev), g->result_host, 0));
    }
    g->gather_flush_id = ivx_many_flush_count();
    const int rc = ivx_launch_step_gather(g);
    if (!rc) g->gather_launched = 1;
    return rc;
}

int ivx_voxel_step_collect(ivx_grid* g, ivx_step_result* out) {
    IVX_REQUIRE(g && out, IVX_ERR_INVALID, "ivx_voxel_step_collect: null argument");
    ivx_many_other_context other_(g->ctx);
    hipStream_t s = g->ctx->stream;
    memset(out, 0, sizeof(*out));
    const uint32_t stages = g->pending_stages;
    // the small results of every stage (region scalars + occupied minima/maxima, mesh totals, moments) arrive in one
    // host-mapped block written by a last tiny kernel: one wait, no copies
    if (!g->result_host) {
        IVX_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&g->result_host), 64 * sizeof(uint32_t), hipHostMallocMapped));
        memset(g->result_host, 0, 64 * sizeof(uint32_t));
        IVX_HIP_CHECK(hipHostGetDevicePointer(reinterpret_cast<void**>(&g->result_host_dev), g->result_host, 0));
    }
    const bool have_results = g->results_in_block != 0;  // (ivx_slab_remesh_enqueue: the block is written, the caller has waited for the stream)
    g->results_in_block = 0;
    if (!have_results) {
        if (!g->gather_launched) {  // (else ivx_step_collect_launch has put it on the stream: the many-object calls launch all objects' gathers as one)
            int rc = ivx_launch_step_gather(g);
            if (rc) return rc;
            (void)ivx_many_break();
        }
        // (a gather that was only recorded has to be on the stream before anybody polls its doorbell: flushed now unless a flush has gone by since)
        if (g->gather_launched && ivx_many_recording() && g->gather_flush_id == ivx_many_flush_count()) (void)ivx_many_break();
        g->gather_launched = 0;
    } else if (hipStreamQuery(s) != hipSuccess) {
        IVX_HIP_CHECK(ivx_stream_sync(s));
    }
    // A short step is over before the runtime's blocking wait has gone to sleep and been woken again: poll the doorbell word the
    // gather kernel writes last (in-order stream: everything enqueued before it is complete too) for a bounded time first.
    if (!have_results) {
        const volatile uint32_t* bell = g->result_host + 63;
        const uint32_t want = g->result_seq;
        const uint64_t budget_ns = collect_spin_ns();
        bool rung = false;
        if (budget_ns) {
            const auto t0 = std::chrono::steady_clock::now();
            for (uint32_t it = 0;; ++it) {
                if (*bell == want) {
                    rung = true;
                    break;
                }
                __builtin_ia32_pause();
                if ((it & 255u) == 255u && (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count() > budget_ns) break;
            }
            std::atomic_thread_fence(std::memory_order_acquire);
        }
        if (!rung) {
            IVX_HIP_CHECK(ivx_stream_sync(s));
            // the stream is empty and the doorbell has not rung: the gather never reached the stream (a flush of recorded launches failed and
            // dropped it, many.hpp) — what the block holds is an earlier step's
            std::atomic_thread_fence(std::memory_order_acquire);
            if (*bell != want) {
                const int dropped = ivx_many_error(g->ctx, true);
                g->pending_stages = 0;
                ivx_set_error("ivx_voxel_step_collect: the step's results never arrived (%s)", dropped ? "a flush of recorded launches failed" : "the gather did not run");
                return IVX_ERR_HIP;
            }
        }
    }
    const uint32_t* sc = g->result_host;
    if (sc[31]) g->last_active = sc[31];
    if (g->eval_len_pending == 2 && g->samp_len) {  // a sample stage and a derive sweep behind it have run under the resident program
        for (int c = 0; c < 3; ++c) g->eval_len[c] = sc[52 + c];
        g->eval_len_valid = 1;
    }
    g->eval_len_pending = 0;
    if (stages & IVX_STAGE_REGIONS) {
        IVX_REQUIRE((sc[1] & 1u) == 0, IVX_ERR_CAPACITY, "ivx_voxel_step: a chunk has more than 254 local regions");
        IVX_REQUIRE((sc[1] & 8u) == 0, IVX_ERR_HIP, "ivx_voxel_step: the region merge gave up waiting for the numbering of the multi-region chunks (k_step_post2)");
        g->region_count = sc[0];
        g->regions_valid = 1;
        out->region_count = sc[0];
    }
    if (stages & IVX_STAGE_OCCUPIED) {
        ivx_occupied_from_raw(g, sc + 16, out->occupied);
        memcpy(g->occ_ref, out->occupied, sizeof(g->occ_ref));
        g->occ_ref_valid = 1;
    }
    if (stages & IVX_STAGE_REMESH) {
        const uint32_t totals[3] = {sc[28], sc[29], sc[30]};
        bool grown_later = false;
        if (totals[0] > g->vcap || totals[1] > g->icap || totals[2] > g->scap) {
            if (g->defer_mesh_growth) {
                grown_later = true;  // (ivx_voxel_step_many: the buffers of all objects that outgrew theirs from one allocation, their emit passes as one launch)
            } else {
                int rc;
                if ((rc = ensure_mesh_capacity(g, totals[0], totals[1], totals[2]))) return rc;
                if ((rc = ivx_launch_sn_emit(g))) return rc;
                IVX_HIP_CHECK(ivx_stream_sync(s));
            }
        }
        g->mesh_counts.n_vertices = totals[0];
        g->mesh_counts.n_indices = totals[1];
        g->mesh_counts.n_submeshes = totals[2];
        g->mesh_counts.reserved = 0;
        g->mesh_growth_pending = grown_later ? 1 : 0;
        if (!grown_later) {
            g->mesh_valid = 1;
            g->mesh_built = 1;
            g->mesh_serial += 1;
        }
    }
    if (stages & IVX_STAGE_INERTIA) {
        memcpy(out->moments.m64, sc + 32, 10 * sizeof(double));
        for (int i = 0; i < 10; ++i) out->moments.m32[i] = (float)out->moments.m64[i];
    }
    out->mesh = g->mesh_counts;
    for (int i = 0; i < IVX_N_TIMED_STAGES; ++i) {
        float ms = 0.0f;
        if ((g->timed_mask >> i) & 1u)
            if (hipEventElapsedTime(&ms, *g->ev_start_ref[i], g->ev[2 * i + 1]) != hipSuccess) ms = 0.0f;
        out->stage_ms[i] = ms;
    }
    g->pending_stages = 0;
    g->timed_mask = 0;
    return IVX_OK;
}

int ivx_voxel_step(ivx_grid* g, uint32_t stages, ivx_step_result* out) {
    IVX_REQUIRE(g && out, IVX_ERR_INVALID, "ivx_voxel_step: null argument");
    int rc = ivx_voxel_step_enqueue(g, stages);
    if (rc) return rc;
    return ivx_voxel_step_collect(g, out);
}

size_t ivx_halo_bytes(ivx_grid* g) {
    if (!g) return 0;
    const size_t cols = (size_t)g->cc[1] * g->cc[2];
    return cols * (256 + 256 + sizeof(ivx_chunk_info));
}

int ivx_halo_pack(ivx_grid* g, int side, void* device_buf) {
    IVX_REQUIRE(g && device_buf && (side == 0 || side == 1), IVX_ERR_INVALID, "ivx_halo_pack: bad argument");
    int rc = ivx_launch_halo_pack(g, side, device_buf);
    if (rc) return rc;
    IVX_HIP_CHECK(ivx_stream_sync(g->ctx->stream));
    return IVX_OK;
}

int ivx_halo_unpack(ivx_grid* g, int side, const void* device_buf) {
    IVX_REQUIRE(g && device_buf && (side == 0 || side == 1), IVX_ERR_INVALID, "ivx_halo_unpack: bad argument");
    const size_t cols = (size_t)g->cc[1] * g->cc[2];
    hipStream_t s = g->ctx->stream;
    const uint8_t* b = static_cast<const uint8_t*>(device_buf);
    IVX_HIP_CHECK(ivx_memcpy_async(g->ghost_sdf[side], b, cols * 256, hipMemcpyDeviceToDevice, s));
    IVX_HIP_CHECK(ivx_memcpy_async(g->ghost_type[side], b + cols * 256, cols * 256, hipMemcpyDeviceToDevice, s));
    IVX_HIP_CHECK(ivx_memcpy_async(g->ghost_info[side], b + cols * 512, cols * sizeof(ivx_chunk_info), hipMemcpyDeviceToDevice, s));
    g->has_ghost[side] = 1;
    g->ghost_ext[side] = nullptr;
    IVX_HIP_CHECK(ivx_stream_sync(s));
    return IVX_OK;
}

int ivx_halo_pack_enqueue(ivx_grid* g, int side, void* device_buf) {
    IVX_REQUIRE(g && device_buf && (side == 0 || side == 1), IVX_ERR_INVALID, "ivx_halo_pack_enqueue: bad argument");
    return ivx_launch_halo_pack(g, side, device_buf);
}

int ivx_halo_unpack_enqueue(ivx_grid* g, int side, const void* device_buf) {
    IVX_REQUIRE(g && device_buf && (side == 0 || side == 1), IVX_ERR_INVALID, "ivx_halo_unpack_enqueue: bad argument");
    IVX_REQUIRE((reinterpret_cast<uintptr_t>(device_buf) & 15u) == 0, IVX_ERR_INVALID, "ivx_halo_unpack_enqueue: the buffer must be 16-byte aligned");
    // no copy: the kernels read the ghost layer in place from the receive buffer (three copies per side and exchange were a
    // sixth of a multi-GPU step's launches); the caller keeps it untouched until it installs the next one
    g->ghost_ext[side] = static_cast<const uint8_t*>(device_buf);
    g->has_ghost[side] = 1;
    return IVX_OK;
}

int ivx_halo_pack_both_enqueue(ivx_grid* g, void* lower_buf, void* upper_buf, int with_face_labels) {
    IVX_REQUIRE(g && (lower_buf || upper_buf), IVX_ERR_INVALID, "ivx_halo_pack_both_enqueue: bad argument");
    return ivx_launch_halo_pack_both(g, lower_buf, upper_buf, with_face_labels);
}

int ivx_region_face_labels_enqueue(ivx_grid* g, int side, void* device_buf) {
    IVX_REQUIRE(g && device_buf && (side == 0 || side == 1), IVX_ERR_INVALID, "ivx_region_face_labels_enqueue: bad argument");
    return ivx_launch_face_ids(g, side, static_cast<uint16_t*>(device_buf));
}

static int ensure_pairs(ivx_grid* g) {
    if (g->pairs_dev) return IVX_OK;
    IVX_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&g->pairs_dev), sizeof(uint32_t) * (4 + 128 + 2 * (size_t)IVX_MAX_FACE_PAIRS)));
    IVX_HIP_CHECK(ivx_memset_async(g->pairs_dev, 0, sizeof(uint32_t) * (4 + 128), g->ctx->stream));
    return IVX_OK;
}

int ivx_region_face_pairs_enqueue(ivx_grid* g, int side, const void* neighbour_face_labels) {
    IVX_REQUIRE(g && neighbour_face_labels && (side == 0 || side == 1), IVX_ERR_INVALID, "ivx_region_face_pairs_enqueue: bad argument");
    int rc = ensure_pairs(g);
    if (rc) return rc;
    rc = ivx_launch_face_pairs(g, side, static_cast<const uint16_t*>(neighbour_face_labels), g->pairs_dev, g->pairs_dev + 4 + 128, IVX_MAX_FACE_PAIRS,
                               g->pairs_dev + 4);
    if (rc) return rc;
    g->pairs_enqueued = 1;
    return IVX_OK;
}

size_t ivx_step_record_words(void) { return 28 + 2 * (size_t)IVX_MAX_FACE_PAIRS; }

int ivx_step_record_enqueue(ivx_grid* g, void* device_record) {
    IVX_REQUIRE(g && device_record, IVX_ERR_INVALID, "ivx_step_record_enqueue: null argument");
    int rc = ensure_pairs(g);
    if (rc) return rc;
    // (no face-pair pass since the last record: the record lists none — a null count, not a fill of the counter)
    rc = ivx_launch_step_record(g, g->pairs_enqueued ? g->pairs_dev : nullptr, g->pairs_dev + 4 + 128, IVX_MAX_FACE_PAIRS, device_record, g->result_host_dev != nullptr);
    if (!rc && g->result_host_dev) g->results_in_block = 1;
    g->pairs_enqueued = 0;
    return rc;
}

// ---- many objects per call (many.hpp) ---------------------------------------------------------------------------------------------------
// The per-object host code runs object after object while the launches are recorded; one flush issues a launch per chain position for all of
// them; the results come back through every object's own host-mapped block, their gathers launched as one.
int ivx_many_begin(ivx_ctx* c);
int ivx_many_flush(ivx_ctx* c);
static int many_check(ivx_grid* const* grids, size_t n, const char* who) {
    IVX_REQUIRE(grids && grids[0], IVX_ERR_INVALID, "%s: null object list", who);
    static thread_local std::vector<const ivx_grid*> seen;  // (a thousand objects a call: no quadratic search for the duplicate)
    seen.assign(grids, grids + n);
    std::sort(seen.begin(), seen.end());
    for (size_t i = 0; i < n; ++i) {
        IVX_REQUIRE(grids[i] && grids[i]->ctx == grids[0]->ctx, IVX_ERR_INVALID, "%s: object %zu is null or belongs to another context", who, i);
        IVX_REQUIRE(i == 0 || seen[i] != seen[i - 1], IVX_ERR_INVALID, "%s: an object is listed twice", who);
    }
    return IVX_OK;
}
// A many-object call that fails half way — an object's enqueue refused, a flush failed — must not leave the objects before it "in flight":
// the stream is drained, every object's collect half runs with its results discarded, and whatever flag still says "pending" is cleared, so
// that the next call on any of these objects starts clean (their derived state is what the launches that did run left: step them again).
static int many_fail(ivx_grid* const* grids, size_t n, int rc) {
    if (!rc || !n) return rc;
    char keep[512];
    snprintf(keep, sizeof(keep), "%s", ivx_last_error());  // (the collects below may set messages of their own: the caller gets the first failure's)
    (void)ivx_stream_sync(grids[0]->ctx->stream);
    (void)ivx_many_error(grids[0]->ctx, true);
    for (size_t i = 0; i < n; ++i) {
        ivx_grid* g = grids[i];
        if (!g) continue;
        if (g->edit && g->edit->pending) {
            ivx_absorb_result tmp;
            (void)absorb_collect(g, "ivx_*_many (drain)", &tmp, nullptr, nullptr);
        }
        if (g->edit && g->edit->sync_pending) {
            ivx_mesh_counts mc;
            (void)mesh_sync_collect(g, &mc, "ivx_*_many (drain)", true);
        }
        if (g->pending_stages || g->gather_launched) {
            ivx_step_result tmp;
            (void)ivx_voxel_step_collect(g, &tmp);
        }
        if (g->edit) g->edit->pending = 0, g->edit->sync_pending = 0;
        g->pending_stages = 0;
        g->gather_launched = 0;
        g->results_in_block = 0;
        g->defer_mesh_growth = 0;
        if (g->mesh_growth_pending) g->mesh_growth_pending = 0, g->mesh_valid = 0;
    }
    ivx_set_error("%s", keep);
    return rc;
}
// runs `f(i)` for every object under the recorder and flushes; the first error ends the batch (what was recorded still goes out; the caller
// drains: many_fail)
static int many_phase(ivx_grid* const* grids, size_t n, const std::function<int(size_t)>& f) {
    ivx_ctx* c = grids[0]->ctx;
    int rc = ivx_many_begin(c);
    if (rc) return rc;
    int first = IVX_OK;
    const auto t0 = std::chrono::steady_clock::now();
    for (size_t i = 0; i < n && !first; ++i) {
        ivx_many_object((uint32_t)i);
        first = f(i);
    }
    const auto t1 = std::chrono::steady_clock::now();
    rc = ivx_many_flush(c);
    static const bool trace_phase_ = getenv("IVX_MANY_TRACE") != nullptr;
    if (trace_phase_)
        fprintf(stderr, "[ivx many]   phase: objects %.1f us, flush %.1f us\n", 1e-3 * (double)std::chrono::duration_cast<std::chrono::nanoseconds>(t1 - t0).count(),
                1e-3 * (double)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t1).count());
    return first ? first : rc;
}

int ivx_voxel_step_many(ivx_grid* const* grids, size_t n, uint32_t stages, ivx_step_result* out) {
    if (n == 0) return IVX_OK;  // (a manager without voxel objects)
    int rc = many_check(grids, n, "ivx_voxel_step_many");
    if (rc) return rc;
    IVX_REQUIRE(out, IVX_ERR_INVALID, "ivx_voxel_step_many: null result array");
    // (no event records around the stage slots: an event is a stream operation of its own and would end the merging after every object;
    // `stage_ms` of a merged step is zero — the launches are shared, their times are not an object's)
    if ((rc = many_phase(grids, n, [&](size_t i) {
             const uint32_t keep = grids[i]->stage_timing_off;
             grids[i]->stage_timing_off = 0xFFFFFFFFu;
             const int r = step_enqueue(grids[i], stages, nullptr, nullptr);
             grids[i]->stage_timing_off = keep;
             return r;
         })))
        return many_fail(grids, n, rc);
    if ((rc = many_phase(grids, n, [&](size_t i) { return ivx_step_collect_launch(grids[i]); }))) return many_fail(grids, n, rc);
    for (size_t i = 0; i < n; ++i) {
        grids[i]->defer_mesh_growth = 1;
        rc = ivx_voxel_step_collect(grids[i], &out[i]);
        grids[i]->defer_mesh_growth = 0;
        if (rc) return many_fail(grids, n, rc);
    }
    // Objects whose meshes outgrew their buffers (all of them, when the fragments of an impact are stepped for the first time): the new buffers
    // of all from ONE allocation at twice what each needs (docs/voxel_gpu_buffer_pooling.md:44-66), their emit passes again as one launch
    // each, one wait — where the single-object collect pays six allocations, a launch and a wait per object.
    static thread_local std::vector<ivx_grid*> grow;
    static thread_local std::vector<size_t> caps;
    grow.clear(), caps.clear();
    for (size_t i = 0; i < n; ++i)
        if (grids[i]->mesh_growth_pending) grow.push_back(grids[i]);
    if (!grow.empty()) {
        const size_t m = grow.size();
        caps.resize(3 * m);
        for (size_t k = 0; k < m; ++k) {
            const ivx_mesh_counts& mc = grow[k]->mesh_counts;
            caps[k] = std::max<size_t>(2 * (size_t)mc.n_vertices + 4096, grow[k]->vcap * 2);
            caps[m + k] = std::max<size_t>(2 * (size_t)mc.n_indices + 24576, grow[k]->icap * 2);
            caps[2 * m + k] = std::max<size_t>(2 * (size_t)mc.n_submeshes + 64, grow[k]->scap * 2);
        }
        if ((rc = mesh_pool_assign(grow.data(), m, caps.data(), caps.data() + m, caps.data() + 2 * m))) return many_fail(grids, n, rc);
        if ((rc = many_phase(grow.data(), m, [&](size_t k) -> int {
                 ivx_grid* g = grow[k];
                 if (!ivx_many_zero(g->ctx, g, ivx_sn_hard_count(g), IVX_SN_TAIL_WORDS * sizeof(uint32_t)))
                     IVX_HIP_CHECK(ivx_memset_async(ivx_sn_hard_count(g), 0, IVX_SN_TAIL_WORDS * sizeof(uint32_t), g->ctx->stream));
                 int r = ivx_launch_step_emit(g, IVX_STAGE_REMESH, true, nullptr, false);
                 if (r) return r;
                 return ivx_launch_step_assign(g, true, false);
             })))
            return many_fail(grids, n, rc);
        IVX_HIP_CHECK(ivx_stream_sync(grids[0]->ctx->stream));
        for (ivx_grid* g : grow) {
            g->mesh_growth_pending = 0;
            g->mesh_valid = 1;
            g->mesh_built = 1;
            g->mesh_serial += 1;
        }
    }
    return IVX_OK;
}

// developer aid: IVX_MANY_TRACE=1 prints the host time of every phase of the many-object calls
static bool many_trace() {
    static const bool on = getenv("IVX_MANY_TRACE") != nullptr;
    return on;
}
struct ManyClock {
    const char* what;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    explicit ManyClock(const char* w) : what(w) {}
    void lap(const char* phase) {
        if (!many_trace()) return;
        const auto t1 = std::chrono::steady_clock::now();
        fprintf(stderr, "[ivx many] %s %s: %.1f us\n", what, phase, 1e-3 * (double)std::chrono::duration_cast<std::chrono::nanoseconds>(t1 - t0).count());
        t0 = t1;
    }
};

// one absorber per object: spheres (segments3 == NULL) or capsules (segment vectors, 3 floats per object)
static int absorb_many(ivx_grid* const* grids, size_t n, const char* who, const float* points3, const float* segments3, const float* influence_radii,
                       const float* shape_radii, const float densities[256], ivx_absorb_result* out, uint8_t* const* invalidated_chunks) {
    if (n == 0) return IVX_OK;  // (a manager without voxel objects)
    int rc = many_check(grids, n, who);
    if (rc) return rc;
    ManyClock clk("absorb");
    IVX_REQUIRE(points3 && influence_radii && shape_radii && densities && out, IVX_ERR_INVALID, "%s: null argument", who);
    // (what an object's enqueue may have to wait for — its occupied ranges, a density table that is not the resident one — before the recording starts)
    for (size_t i = 0; i < n; ++i) {
        uint32_t occ[12];
        if (grids[i]->regions_valid && (rc = reference_occupied(grids[i], occ))) return rc;
        // (... and the object's edit buffers, which its first edit would otherwise allocate — with a wait on the stream — in the middle of the batch)
        ivx_edit_state* e = edit_state(grids[i]);
        IVX_REQUIRE(e, IVX_ERR_CAPACITY, "%s: out of host memory", who);
        if (!e->d_results) {
            const size_t cap = 1 << 16;
            IVX_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&e->d_results), cap));
            IVX_HIP_CHECK(ivx_memset_async(e->d_results, 0, cap, grids[i]->ctx->stream));
            e->d_results_bytes = cap;
        }
        if (!e->pinned) {
            const size_t cap = 1 << 16;
            IVX_HIP_CHECK(hipHostMalloc(&e->pinned, cap, hipHostMallocMapped));
            IVX_HIP_CHECK(hipHostGetDevicePointer(&e->pinned_dev, e->pinned, 0));
            e->pinned_bytes = cap;
        }
    }
    clk.lap("prepare");
    if ((rc = many_phase(grids, n, [&](size_t i) {
             return absorb_enqueue(grids[i], who, segments3 ? 1 : 0, points3 + 3 * i, segments3 ? segments3 + 3 * i : nullptr, influence_radii[i], shape_radii[i], densities);
         })))
        return many_fail(grids, n, rc);
    clk.lap("enqueue + flush");
    if ((rc = many_phase(grids, n, [&](size_t i) { return (grids[i]->edit && grids[i]->edit->pending && !grids[i]->edit->nothing) ? ivx_step_collect_launch(grids[i]) : IVX_OK; })))
        return many_fail(grids, n, rc);
    clk.lap("gathers");
    rc = many_phase(grids, n, [&](size_t i) { return absorb_collect(grids[i], who, &out[i], nullptr, invalidated_chunks ? invalidated_chunks[i] : nullptr); });
    clk.lap("collect");
    return many_fail(grids, n, rc);
}
int ivx_absorb_sphere_many(ivx_grid* const* grids, size_t n, const float* centers3, const float* influence_radii, const float* sphere_radii,
                           const float densities[256], ivx_absorb_result* out, uint8_t* const* invalidated_chunks) {
    return absorb_many(grids, n, "ivx_absorb_sphere_many", centers3, nullptr, influence_radii, sphere_radii, densities, out, invalidated_chunks);
}
int ivx_absorb_capsule_many(ivx_grid* const* grids, size_t n, const float* segment_starts3, const float* segment_vectors3, const float* influence_radii,
                            const float* capsule_radii, const float densities[256], ivx_absorb_result* out, uint8_t* const* invalidated_chunks) {
    IVX_REQUIRE(n == 0 || segment_vectors3, IVX_ERR_INVALID, "ivx_absorb_capsule_many: null argument");
    return absorb_many(grids, n, "ivx_absorb_capsule_many", segment_starts3, segment_vectors3, influence_radii, capsule_radii, densities, out, invalidated_chunks);
}

int ivx_mesh_sync_many(ivx_grid* const* grids, size_t n, const uint8_t* const* invalidated_chunks, ivx_mesh_counts* out) {
    if (n == 0) return IVX_OK;  // (a manager without voxel objects)
    int rc = many_check(grids, n, "ivx_mesh_sync_many");
    if (rc) return rc;
    IVX_REQUIRE(invalidated_chunks && out, IVX_ERR_INVALID, "ivx_mesh_sync_many: null argument");
    ManyClock clk("sync");
    if ((rc = many_phase(grids, n, [&](size_t i) { return mesh_sync_enqueue(grids[i], invalidated_chunks[i], "ivx_mesh_sync_many"); }))) return many_fail(grids, n, rc);
    clk.lap("enqueue + flush");
    IVX_HIP_CHECK(ivx_stream_sync(grids[0]->ctx->stream));  // (one wait for all)
    clk.lap("wait");
    for (size_t i = 0; i < n; ++i)
        if ((rc = mesh_sync_collect(grids[i], &out[i], "ivx_mesh_sync_many", true))) return many_fail(grids, n, rc);
    return IVX_OK;
}

int ivx_halo_clear(ivx_grid* g, int side) {
    IVX_REQUIRE(g && (side == 0 || side == 1), IVX_ERR_INVALID, "ivx_halo_clear: bad argument");
    g->has_ghost[side] = 0;
    g->ghost_ext[side] = nullptr;
    return IVX_OK;
}

}  // extern "C"
