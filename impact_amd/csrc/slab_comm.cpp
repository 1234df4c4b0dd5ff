// (e) multi-GPU — the per-step protocol of the x-slab decomposition behind the C ABI (no reference counterpart:
// lars-frogner/Impact runs this path in one process). One process per GPU owns one slab (ivx_grid with an x chunk offset); per step
//   1. sample                                 -> one-voxel x-face planes (sdf, type) + the face layer's chunk records to both neighbours
//   2. derive (+ regions, moments, occupied)  -> the planes again (post-demotion chunk kinds) + the faces' slab-local component ids
//   3. remesh, the slab's record              -> ONE all-gather of the records' heads; every rank finishes the same union-find
// Kernels, packing and the one small ncclAllGather go on the context's stream; the two neighbour exchanges (grouped ncclSend / ncclRecv over
// the two xGMI links to the neighbours) go on a stream of the communicator's, behind an event that follows the packing — and the slab's next
// sweep is split around their arrival: the derive sweep (after exchange 1) and the mesher's count (after exchange 2) first take the chunk
// planes that read nothing of a ghost layer, the context's stream then waits for the exchange's event, the two face planes follow
// (ivx_grid::ghost_event; derive.hip, step_fused.hip). Interior work so overlaps the link. That overlapped form is OPT-IN for RCCL ranks
// (IVX_SLAB_OVERLAP=1; ivx_comm_init says why): by default the exchanges run on the context's stream. The host waits once per step. Ghost layers are read in place from the receive buffers (ivx_halo_unpack_enqueue).
//
// Three transports behind one driver:
//   * RCCL (ivx_comm_init): librccl is opened at run time (the copy already loaded in the process — e.g. the one PyTorch bundles —
//     else librccl.so from the loader path / IVX_RCCL_LIB); the library has no link-time dependency on it, and a missing RCCL fails
//     loudly when a communicator is asked for;
//   * in-process (ivx_comm_init_local): all ranks are slabs of ONE process on ONE GPU, neighbour exchange and all-gather are
//     device-to-device copies on the same stream. This is how the decomposition is checked bit for bit against the oracle on the
//     single GPU the test box has (tests/test_gpu_slabs.py), through exactly the driver code the RCCL ranks run;
//   * shared device, several PROCESSES (ivx_comm_init_ipc): one process per rank as under RCCL, all on one GPU (RCCL refuses two ranks on
//     one device). A rank writes its face planes straight into its neighbour's receive buffer (hipIpcGetMemHandle / hipIpcOpenMemHandle),
//     arrival and buffer reuse are sequence numbers in a POSIX shared-memory block, the records are gathered through that block. The host
//     waits at every exchange — a transport for checking the driver's sequencing as separate processes where there is one GPU, not for speed.
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include "ivx_internal.hpp"

namespace {

constexpr size_t HEAD_PAIRS = 64;  // pairs that ride in the first all-gather (a slab boundary is crossed by a handful of components)
constexpr size_t HEAD_WORDS = 28 + 2 * HEAD_PAIRS;

// the part of the RCCL API the protocol uses (types as in rccl.h; resolved with dlsym)
typedef struct ncclComm* ncclComm_t;
struct ncclUniqueId_ {
    char internal[128];
};
typedef int ncclResult_t;
enum { NCCL_INT8 = 0, NCCL_UINT8 = 1, NCCL_INT64 = 4 };
struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId_*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId_, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
RcclApi g_rccl;

int load_rccl() {
    if (g_rccl.handle) return IVX_OK;
    void* h = nullptr;
    if (const char* e = getenv("IVX_RCCL_LIB")) h = dlopen(e, RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);    // a copy the process has loaded already (PyTorch's)
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
    IVX_REQUIRE(h, IVX_ERR_HIP, "RCCL is not available: %s", dlerror());
    RcclApi a;
    a.handle = h;
#define SYM(field, name)                                                              \
    *reinterpret_cast<void**>(&a.field) = dlsym(h, name);                             \
    IVX_REQUIRE(a.field, IVX_ERR_HIP, "RCCL symbol %s not found", name)
    SYM(GetUniqueId, "ncclGetUniqueId");
    SYM(CommInitRank, "ncclCommInitRank");
    SYM(CommDestroy, "ncclCommDestroy");
    SYM(CommCount, "ncclCommCount");
    SYM(CommUserRank, "ncclCommUserRank");
    SYM(Send, "ncclSend");
    SYM(Recv, "ncclRecv");
    SYM(AllGather, "ncclAllGather");
    SYM(GroupStart, "ncclGroupStart");
    SYM(GroupEnd, "ncclGroupEnd");
    SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    g_rccl = a;
    return IVX_OK;
}

#define IVX_NCCL_CHECK(expr)                                                                                   \
    do {                                                                                                       \
        const ncclResult_t _r = (expr);                                                                        \
        if (_r != 0) {                                                                                         \
            ivx_set_error("%s failed: %s (%s:%d)", #expr, g_rccl.GetErrorString(_r), __FILE__, __LINE__);     \
            return IVX_ERR_HIP;                                                                                \
        }                                                                                                      \
    } while (0)

}  // namespace

constexpr int IPC_MAX_RANKS = 16;
constexpr size_t IPC_MAX_REC_WORDS = 28 + 2 * (size_t)IVX_MAX_FACE_PAIRS;
// the rendezvous block of the shared-device transport (POSIX shared memory; created and zeroed by rank 0)
struct IpcRank {
    hipIpcMemHandle_t recv[2];               // this rank's receive buffers, for its neighbours to open
    unsigned long long pid, recv_ptr[2];     // ... and as plain pointers for a neighbour rank living in the same process (a handle is not opened by its maker)
    std::atomic<unsigned long long> posted;  // slabs created: handles are valid
    std::atomic<unsigned long long> sent[2];      // messages that have fully arrived in recv[side]
    std::atomic<unsigned long long> consumed[2];  // messages of recv[side] whose readers have finished (the buffer may be overwritten)
    std::atomic<unsigned long long> rec_seq;      // records published
    std::atomic<unsigned long long> rec_read;     // gathers this rank has finished reading (the others may overwrite their slots)
    unsigned long long record[IPC_MAX_REC_WORDS];
    // rendezvous handshake: a rank writes a fresh random token into `hello`; rank 0 — the creator of THIS block — echoes it into
    // `hello_ack`. A stale block of the same name left by a crashed run never echoes a fresh token: the rank re-opens the name until it sits
    // on the block rank 0 made for this run.
    std::atomic<unsigned long long> hello, hello_ack;
    std::atomic<unsigned long long> opened;  // handle imports of this rank's `posted` generations that its neighbours have finished
};
struct IpcShared {
    std::atomic<unsigned long long> magic;  // set last by rank 0
    IpcRank ranks[IPC_MAX_RANKS];
};
constexpr unsigned long long IPC_MAGIC = 0x4956585F49504331ull;

struct ivx_comm {
    ivx_ctx* ctx;
    int nranks, rank;  // rank = -1: in-process communicator (every rank lives here)
    ncclComm_t nccl;
    // shared-device transport
    IpcShared* ipc;
    char ipc_name[96];
    unsigned long long ipc_recv_seq[2], ipc_send_seq[2], ipc_rec_seq;  // messages expected in / put from this rank, records gathered
    void* ipc_peer_recv[2];  // neighbour rank - 1's recv[1], neighbour rank + 1's recv[0], opened in this process
    int ipc_peer_same_process[2];     // ipc_peer_recv[s] is the neighbour's own pointer (same process): nothing to close
    unsigned long long ipc_slab_gen;  // slabs created on this communicator (ivx_slab_create is collective: every rank counts alike)
    std::thread* ipc_acker;           // rank 0: answers the other ranks' hellos (they may arrive at any time before their first exchange)
    std::atomic<int> ipc_acker_stop;
    // exchanges beside the compute stream: a stream of the communicator's, per exchange of a step an event behind the packing (recorded on the
    // context's stream) and one behind the arrival (recorded on this stream). `overlap`: in use (RCCL with neighbours; the in-process
    // transport when it is told to move its messages by copies, ivx_comm_set_local_copies — the same choreography on the one GPU of a test box)
    hipStream_t comm_stream;
    hipEvent_t ev_packed[3], ev_arrived[3];  // (exchange 1; exchange 2: the face planes, the face ids)
    int overlap, local_copies;
};

struct ivx_slab {
    ivx_comm* comm;
    ivx_grid* grid;
    int rank;
    size_t halo_bytes, face_bytes, rec_words;
    uint8_t* send[2];
    uint8_t* recv[2];
    // in-process transport: no copies — a slab reads its ghost layers straight from the neighbour's send buffer, so the second exchange of a step
    // packs into a second pair (the first pair is still being read by slabs later in the stream's order)
    uint8_t* send2[2];
    const uint8_t* ghost[2];  // where this slab's ghost layers of the last exchange are (recv[side], or a neighbour's send buffer)
    unsigned long long* record;    // this slab's record (device)
    unsigned long long* gathered;  // nranks records (device)
    unsigned long long* host_head;      // host-mapped pinned block: the nranks record heads + one doorbell word (the keeper's only)
    unsigned long long* host_head_dev;  // its device-side address
    unsigned long long head_seq;        // sequence number of the last publish
    std::vector<unsigned long long> host_records;
    size_t map_offset;                  // where the region map starts in host_records (0: no collected step)
    int local_err;                      // first error of this rank's last enqueue (reported by collect, after the step's collectives)
    bool has_lo, has_hi;
    int enqueued;
};

namespace {

// The gathered record heads into the host-mapped block, the sequence number behind them: the step's completion doorbell (the host polls
// it instead of a copy + blocking wait; see ivx_voxel_step_collect). One block; every wave drains its stores before the barrier.
__global__ __launch_bounds__(256) void k_slab_publish(const unsigned long long* __restrict__ gathered, unsigned long long* __restrict__ host, uint32_t n_words,
                                                      unsigned long long seq) {
    for (uint32_t i = threadIdx.x; i < n_words; i += 256u) host[i] = gathered[i];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0u) __hip_atomic_store(host + n_words, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// bounded wait on a word of the rendezvous block (another PROCESS advances it): a peer that died must not hang this one for ever
template <typename F>
bool ipc_wait(F&& done, const char* what) {
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t it = 0;; ++it) {
        if (done()) return true;
        if ((it & 63u) == 63u) {
            usleep(20);
            if (std::chrono::duration_cast<std::chrono::seconds>(std::chrono::steady_clock::now() - t0).count() > 60) {
                ivx_set_error("shared-device transport: waited 60 s for %s", what);
                return false;
            }
        } else {
            __builtin_ia32_pause();
        }
    }
}

// neighbour exchange of the shared-device transport: this rank's stream is drained (the kernels that read the receive buffers in place are
// through: the buffers are free), each send buffer is copied into the neighbour's receive buffer once the neighbour has said the same of
// its own, and the copies are waited for before the arrival counters move
int ipc_exchange(ivx_slab* sl, size_t nbytes) {
    ivx_comm* c = sl->comm;
    hipStream_t s = c->ctx->stream;
    IpcShared* sh = c->ipc;
    IVX_HIP_CHECK(ivx_stream_sync(s));
    for (int side = 0; side < 2; ++side) sh->ranks[c->rank].consumed[side].store(c->ipc_recv_seq[side], std::memory_order_release);
    for (int side = 0; side < 2; ++side) {
        if (!(side ? sl->has_hi : sl->has_lo)) continue;
        const int peer = side ? c->rank + 1 : c->rank - 1, peer_side = side ? 0 : 1;
        const unsigned long long want = c->ipc_send_seq[side];
        if (!ipc_wait([&] { return sh->ranks[peer].consumed[peer_side].load(std::memory_order_acquire) >= want; }, "a neighbour to release its receive buffer"))
            return IVX_ERR_STATE;
        IVX_HIP_CHECK(ivx_memcpy_async(c->ipc_peer_recv[side], sl->send[side], nbytes, hipMemcpyDeviceToDevice, s));
    }
    IVX_HIP_CHECK(ivx_stream_sync(s));
    for (int side = 0; side < 2; ++side) {
        if (!(side ? sl->has_hi : sl->has_lo)) continue;
        const int peer = side ? c->rank + 1 : c->rank - 1, peer_side = side ? 0 : 1;
        c->ipc_send_seq[side] += 1;
        sh->ranks[peer].sent[peer_side].store(c->ipc_send_seq[side], std::memory_order_release);
    }
    for (int side = 0; side < 2; ++side) {
        if (!(side ? sl->has_hi : sl->has_lo)) continue;
        const unsigned long long want = c->ipc_recv_seq[side] + 1;
        if (!ipc_wait([&] { return sh->ranks[c->rank].sent[side].load(std::memory_order_acquire) >= want; }, "a neighbour's face planes")) return IVX_ERR_STATE;
        c->ipc_recv_seq[side] = want;
    }
    return IVX_OK;
}

// record gather of the shared-device transport: through the host and the rendezvous block
int ipc_all_gather(ivx_slab* sl, size_t words) {
    ivx_comm* c = sl->comm;
    hipStream_t s = c->ctx->stream;
    IpcShared* sh = c->ipc;
    IVX_REQUIRE(words <= IPC_MAX_REC_WORDS, IVX_ERR_CAPACITY, "shared-device transport: record of %zu words", words);
    IpcRank& me = sh->ranks[c->rank];
    // (the slot is rewritten only after every rank has copied the previous record out of it)
    const unsigned long long last = c->ipc_rec_seq;
    if (!ipc_wait([&] {
            for (int r = 0; r < c->nranks; ++r)
                if (sh->ranks[r].rec_read.load(std::memory_order_acquire) < last) return false;
            return true;
        }, "the other ranks to read the previous record")) return IVX_ERR_STATE;
    IVX_HIP_CHECK(ivx_memcpy_async(me.record, sl->record, words * 8, hipMemcpyDeviceToHost, s));
    IVX_HIP_CHECK(ivx_stream_sync(s));
    c->ipc_rec_seq += 1;
    me.rec_seq.store(c->ipc_rec_seq, std::memory_order_release);
    const unsigned long long want = c->ipc_rec_seq;
    if (!ipc_wait([&] {
            for (int r = 0; r < c->nranks; ++r)
                if (sh->ranks[r].rec_seq.load(std::memory_order_acquire) < want) return false;
            return true;
        }, "the other ranks' record")) return IVX_ERR_STATE;
    for (int r = 0; r < c->nranks; ++r) IVX_HIP_CHECK(ivx_memcpy_async(sl->gathered + (size_t)r * words, sh->ranks[r].record, words * 8, hipMemcpyHostToDevice, s));
    // (the copies above read the block asynchronously: waited for before this rank reports the gather as read)
    IVX_HIP_CHECK(ivx_stream_sync(s));
    me.rec_read.store(want, std::memory_order_release);
    return IVX_OK;
}

// (test mode of the in-process transport) a link that takes its time: the copies of an exchange wait behind this on the communicator's stream,
// so that a sweep which does not wait for the arrival event is sure to read the receive buffers before the message is in them
__global__ void k_slow_link(unsigned long long ticks_100mhz) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks_100mhz) __builtin_amdgcn_s_sleep(32);
}

// the communicator's own stream and the events of its exchanges, made when first needed
int comm_overlap_ready(ivx_comm* c) {
    if (c->comm_stream) return IVX_OK;
    IVX_HIP_CHECK(hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking));
    for (int k = 0; k < 3; ++k) {
        IVX_HIP_CHECK(hipEventCreateWithFlags(&c->ev_packed[k], hipEventDisableTiming));
        IVX_HIP_CHECK(hipEventCreateWithFlags(&c->ev_arrived[k], hipEventDisableTiming));
    }
    return IVX_OK;
}

// (`second`: the step's second exchange — the in-process transport packs it into its second pair of send buffers, see pack_bufs)
// With `overlap` the messages move on the communicator's stream behind the packing; the slabs' grids are handed the arrival event and make
// the context's stream wait for it where they first read what arrived (ivx_grid::ghost_event / face_ids_event). The second exchange then
// goes in two messages: `what` 1 = the face planes + chunk records (bytes [0, halo_bytes) of the buffers), 2 = the face ids behind them,
// 3 = both as one.
int exchange(ivx_slab** slabs, size_t n, bool second, uint32_t what = 3u) {
    ivx_comm* c = slabs[0]->comm;
    hipStream_t s = c->ctx->stream;
    for (size_t i = 0; i < n; ++i) slabs[i]->ghost[0] = slabs[i]->recv[0], slabs[i]->ghost[1] = slabs[i]->recv[1];
    const size_t halo = slabs[0]->halo_bytes, face = slabs[0]->face_bytes;
    const size_t off = what == 2u ? halo : 0u;
    const size_t nbytes = !second ? halo : (what == 3u ? halo + face : (what == 1u ? halo : face));
    if (c->ipc) return ipc_exchange(slabs[0], nbytes);
    const int k = !second ? 0 : (what == 2u ? 2 : 1);
    hipStream_t xs = s;  // the stream the messages move on
    const bool moves = c->rank >= 0 ? (slabs[0]->has_lo || slabs[0]->has_hi) : (c->local_copies && n > 1);
    if (c->overlap && moves) {
        int rc = comm_overlap_ready(c);
        if (rc) return rc;
        IVX_HIP_CHECK(ivx_event_record(c->ev_packed[k], s));
        IVX_HIP_CHECK(hipStreamWaitEvent(c->comm_stream, c->ev_packed[k], 0));
        xs = c->comm_stream;
    }
    if (c->rank >= 0) {  // RCCL: one slab per process, its two neighbours
        ivx_slab* sl = slabs[0];
        if (!sl->has_lo && !sl->has_hi) return IVX_OK;
        IVX_NCCL_CHECK(g_rccl.GroupStart());
        if (sl->has_lo) {
            IVX_NCCL_CHECK(g_rccl.Send(sl->send[0] + off, nbytes, NCCL_UINT8, sl->rank - 1, c->nccl, xs));
            IVX_NCCL_CHECK(g_rccl.Recv(sl->recv[0] + off, nbytes, NCCL_UINT8, sl->rank - 1, c->nccl, xs));
        }
        if (sl->has_hi) {
            IVX_NCCL_CHECK(g_rccl.Send(sl->send[1] + off, nbytes, NCCL_UINT8, sl->rank + 1, c->nccl, xs));
            IVX_NCCL_CHECK(g_rccl.Recv(sl->recv[1] + off, nbytes, NCCL_UINT8, sl->rank + 1, c->nccl, xs));
        }
        IVX_NCCL_CHECK(g_rccl.GroupEnd());
    } else if (c->local_copies) {
        // in-process, messages moved as a rank's would be: every send buffer into the neighbour's receive buffer (in the overlapped test mode
        // behind 150 us of nothing: a link slower than any sweep that might run ahead of it)
        if (xs != s) {
            static const unsigned long long link_us = [] {
                const char* e = getenv("IVX_SLAB_TEST_LINK_US");  // (0: no delay — for timing the choreography itself, tools/slab_timeline.py)
                return e ? strtoull(e, nullptr, 10) : 150ull;
            }();
            if (link_us) hipLaunchKernelGGL(k_slow_link, dim3(1), dim3(1), 0, xs, link_us * 100ull);
        }
        for (size_t i = 0; i + 1 < n; ++i) {
            IVX_HIP_CHECK(hipMemcpyAsync(slabs[i]->recv[1] + off, (second ? slabs[i + 1]->send2 : slabs[i + 1]->send)[0] + off, nbytes, hipMemcpyDeviceToDevice, xs));
            IVX_HIP_CHECK(hipMemcpyAsync(slabs[i + 1]->recv[0] + off, (second ? slabs[i]->send2 : slabs[i]->send)[1] + off, nbytes, hipMemcpyDeviceToDevice, xs));
        }
    } else {
        // in-process: nothing moves. Every slab's kernels are on the one stream, the packs of this exchange ahead of its readers; the buffers of
        // the step's first exchange are packed again in the next step's first phase, those of the second in its second — after their last reader.
        for (size_t i = 0; i + 1 < n; ++i) {
            slabs[i]->ghost[1] = (second ? slabs[i + 1]->send2 : slabs[i + 1]->send)[0];
            slabs[i + 1]->ghost[0] = (second ? slabs[i]->send2 : slabs[i]->send)[1];
        }
        return IVX_OK;
    }
    if (xs != s) {
        IVX_HIP_CHECK(hipEventRecord(c->ev_arrived[k], xs));
        for (size_t i = 0; i < n; ++i) {
            ivx_grid* g = slabs[i]->grid;
            if (what == 2u) {
                g->face_ids_event = c->ev_arrived[k];
            } else {
                g->ghost_event = c->ev_arrived[k];
                // the sweep that reads the planes first: split around the wait where the message has had no time to arrive (the derive sweep
                // right behind exchange 1); the planes of exchange 2 travel behind the region stages — a plain wait ahead of the mesher's count
                g->ghost_split = second ? 0 : 1;
                if (what == 3u) g->face_ids_event = nullptr;
            }
        }
    }
    return IVX_OK;
}
// the buffers a slab packs its faces into for the step's first / second exchange
uint8_t** pack_bufs(ivx_slab* sl, bool second) { return (second && sl->comm->rank < 0) ? sl->send2 : sl->send; }

int all_gather(ivx_slab** slabs, size_t n, size_t words) {
    ivx_comm* c = slabs[0]->comm;
    hipStream_t s = c->ctx->stream;
    if (c->nranks == 1) return IVX_OK;  // (the one record was written in place, ivx_slabs_step_enqueue)
    if (c->ipc) return ipc_all_gather(slabs[0], words);
    if (c->rank >= 0) {
        ivx_slab* sl = slabs[0];
        if (c->nranks == 1) {
            IVX_HIP_CHECK(ivx_memcpy_async(sl->gathered, sl->record, words * 8, hipMemcpyDeviceToDevice, s));
            return IVX_OK;
        }
        IVX_NCCL_CHECK(g_rccl.AllGather(sl->record, sl->gathered, words, NCCL_INT64, c->nccl, s));
        return IVX_OK;
    }
    if (words == HEAD_WORDS) return IVX_OK;  // (in-process: the record roles wrote the heads in place, ivx_slabs_step_enqueue)
    for (size_t i = 0; i < n; ++i)  // every rank's view is the same: one gathered block, kept by rank 0's slab
        IVX_HIP_CHECK(ivx_memcpy_async(slabs[0]->gathered + i * words, slabs[i]->record, words * 8, hipMemcpyDeviceToDevice, s));
    return IVX_OK;
}

void install_ghosts(ivx_slab* sl) {
    for (int side = 0; side < 2; ++side) {
        const bool has = side ? sl->has_hi : sl->has_lo;
        if (has) (void)ivx_halo_unpack_enqueue(sl->grid, side, sl->ghost[side]);
        else (void)ivx_halo_clear(sl->grid, side);
    }
}

}  // namespace

extern "C" {

int ivx_comm_unique_id(void* out128) {
    IVX_REQUIRE(out128, IVX_ERR_INVALID, "ivx_comm_unique_id: null argument");
    int rc = load_rccl();
    if (rc) return rc;
    ncclUniqueId_ id;
    IVX_NCCL_CHECK(g_rccl.GetUniqueId(&id));
    memcpy(out128, &id, sizeof(id));
    return IVX_OK;
}

int ivx_comm_init(ivx_ctx* c, int nranks, int rank, const void* unique_id128, ivx_comm** out) {
    IVX_REQUIRE(c && out && nranks >= 1 && rank >= 0 && rank < nranks && (unique_id128 || nranks == 1), IVX_ERR_INVALID, "ivx_comm_init: bad argument");
    *out = nullptr;
    ivx_comm* m = new (std::nothrow) ivx_comm();
    IVX_REQUIRE(m, IVX_ERR_CAPACITY, "ivx_comm_init: out of host memory");
    m->ctx = c;
    m->nranks = nranks;
    m->rank = rank;
    m->nccl = nullptr;
    m->ipc = nullptr;
    {
        // Opt-in: the overlapped choreography (exchanges on a second stream, the second exchange in two messages) has run between the in-process
        // ranks of one GPU and as a one-rank RCCL self-test only — no box of this pool has had two GPUs. Until a run with two or more ranks is on
        // record the serial form is what RCCL ranks get; IVX_SLAB_OVERLAP=1 turns the overlap on (tests/test_gpu_bench_multi.py runs both).
        const char* e = getenv("IVX_SLAB_OVERLAP");
        m->overlap = (e && e[0] == '1') ? 1 : 0;
    }
    if (nranks > 1) {
        int rc = load_rccl();
        if (rc) {
            delete m;
            return rc;
        }
        ncclUniqueId_ id;
        memcpy(&id, unique_id128, sizeof(id));
        if (hipSetDevice(c->device) != hipSuccess || g_rccl.CommInitRank(&m->nccl, nranks, id, rank) != 0) {
            ivx_set_error("ivx_comm_init: ncclCommInitRank failed (rank %d of %d)", rank, nranks);
            delete m;
            return IVX_ERR_HIP;
        }
    }
    *out = m;
    return IVX_OK;
}

int ivx_comm_init_local(ivx_ctx* c, int nranks, ivx_comm** out) {
    IVX_REQUIRE(c && out && nranks >= 1, IVX_ERR_INVALID, "ivx_comm_init_local: bad argument");
    ivx_comm* m = new (std::nothrow) ivx_comm();
    IVX_REQUIRE(m, IVX_ERR_CAPACITY, "ivx_comm_init_local: out of host memory");
    m->ctx = c;
    m->nranks = nranks;
    m->rank = -1;
    m->nccl = nullptr;
    m->ipc = nullptr;
    *out = m;
    return IVX_OK;
}

// One process per rank on ONE device: `name` is a POSIX shared-memory name ("/something") all ranks pass; rank 0 creates the block.
int ivx_comm_init_ipc(ivx_ctx* c, int nranks, int rank, const char* name, ivx_comm** out) {
    IVX_REQUIRE(c && out && name && name[0] == '/' && strlen(name) < 90 && nranks >= 1 && nranks <= IPC_MAX_RANKS && rank >= 0 && rank < nranks, IVX_ERR_INVALID,
                "ivx_comm_init_ipc: bad argument");
    *out = nullptr;
    ivx_comm* m = new (std::nothrow) ivx_comm();
    IVX_REQUIRE(m, IVX_ERR_CAPACITY, "ivx_comm_init_ipc: out of host memory");
    m->ctx = c;
    m->nranks = nranks;
    m->rank = rank;
    m->nccl = nullptr;
    m->ipc = nullptr;
    snprintf(m->ipc_name, sizeof(m->ipc_name), "%s", name);
    for (int s = 0; s < 2; ++s) m->ipc_recv_seq[s] = m->ipc_send_seq[s] = 0, m->ipc_peer_recv[s] = nullptr, m->ipc_peer_same_process[s] = 0;
    m->ipc_rec_seq = 0;
    m->ipc_slab_gen = 0;
    m->ipc_acker = nullptr;
    m->ipc_acker_stop.store(0);
    auto map_block = [&](int fd) -> IpcShared* {
        void* p = mmap(nullptr, sizeof(IpcShared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        return p == MAP_FAILED ? nullptr : static_cast<IpcShared*>(p);
    };
    if (rank == 0) {
        // the block is made under a private name, initialised, and only then given the name the others open: nobody ever maps a
        // half-made block, and a stale one of the same name is replaced atomically
        char tmp[128];
        snprintf(tmp, sizeof(tmp), "%s.%ld.tmp", name, (long)getpid());
        (void)shm_unlink(tmp);
        int fd = shm_open(tmp, O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd >= 0 && ftruncate(fd, (off_t)sizeof(IpcShared)) != 0) {
            close(fd);
            fd = -1;
        }
        IpcShared* sh = fd >= 0 ? map_block(fd) : nullptr;
        if (fd >= 0) close(fd);
        if (!sh) {
            (void)shm_unlink(tmp);
            ivx_set_error("ivx_comm_init_ipc: shared memory %s not available (rank 0)", name);
            delete m;
            return IVX_ERR_STATE;
        }
        memset(static_cast<void*>(sh), 0, sizeof(IpcShared));  // (all-zero is the initial state of every counter)
        sh->magic.store(IPC_MAGIC, std::memory_order_release);
        char from[160], to[160];
        snprintf(from, sizeof(from), "/dev/shm%s", tmp);
        snprintf(to, sizeof(to), "/dev/shm%s", name);
        if (rename(from, to) != 0) {  // (POSIX shared memory lives under /dev/shm on Linux; rename replaces a stale block in one step)
            munmap(sh, sizeof(IpcShared));
            (void)shm_unlink(tmp);
            ivx_set_error("ivx_comm_init_ipc: could not publish the rendezvous block %s", name);
            delete m;
            return IVX_ERR_STATE;
        }
        m->ipc = sh;
        if (nranks > 1) {
            m->ipc_acker = new (std::nothrow) std::thread([m, nranks] {
                int acked = 1;
                while (!m->ipc_acker_stop.load(std::memory_order_acquire) && acked < nranks) {
                    acked = 1;
                    for (int r = 1; r < nranks; ++r) {
                        const unsigned long long h = m->ipc->ranks[r].hello.load(std::memory_order_acquire);
                        if (h && m->ipc->ranks[r].hello_ack.load(std::memory_order_relaxed) != h) m->ipc->ranks[r].hello_ack.store(h, std::memory_order_release);
                        if (h) acked += 1;
                    }
                    usleep(50);
                }
            });
        }
    } else {
        // open by name, say hello with a token no earlier run can have used, wait for rank 0's echo; no echo within a moment = this may be a
        // stale block: look the name up again
        const auto t0 = std::chrono::steady_clock::now();
        unsigned long long token = ((unsigned long long)getpid() << 32) ^ (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count();
        token |= 1ull;
        IpcShared* sh = nullptr;
        ino_t ino = 0;
        for (;;) {
            if (std::chrono::duration_cast<std::chrono::seconds>(std::chrono::steady_clock::now() - t0).count() > 60) break;
            int fd = shm_open(name, O_RDWR, 0600);
            struct stat st;
            if (fd >= 0 && (fstat(fd, &st) != 0 || (size_t)st.st_size < sizeof(IpcShared))) {
                close(fd);
                fd = -1;
            }
            if (fd < 0) {
                usleep(1000);
                continue;
            }
            if (sh && st.st_ino == ino) {  // still the block we are waiting on
                close(fd);
            } else {
                if (sh) munmap(sh, sizeof(IpcShared));
                sh = map_block(fd);
                close(fd);
                if (!sh) break;
                ino = st.st_ino;
                if (sh->magic.load(std::memory_order_acquire) == IPC_MAGIC) sh->ranks[rank].hello.store(token, std::memory_order_release);
            }
            bool ok = false;
            for (int it = 0; it < 200 && !ok; ++it) {  // ~20 ms on this block, then check the name again
                if (sh->magic.load(std::memory_order_acquire) == IPC_MAGIC) {
                    if (sh->ranks[rank].hello.load(std::memory_order_relaxed) != token) sh->ranks[rank].hello.store(token, std::memory_order_release);
                    ok = sh->ranks[rank].hello_ack.load(std::memory_order_acquire) == token;
                }
                if (!ok) usleep(100);
            }
            if (ok) {
                m->ipc = sh;
                break;
            }
        }
        if (!m->ipc) {
            if (sh) munmap(sh, sizeof(IpcShared));
            ivx_set_error("ivx_comm_init_ipc: rank 0 did not answer on shared memory %s within 60 s (rank %d)", name, rank);
            delete m;
            return IVX_ERR_STATE;
        }
    }
    *out = m;
    return IVX_OK;
}

int ivx_comm_selftest(ivx_ctx* c) {
    IVX_REQUIRE(c, IVX_ERR_INVALID, "ivx_comm_selftest: null context");
    int rc = load_rccl();
    if (rc) return rc;
    IVX_HIP_CHECK(hipSetDevice(c->device));
    ncclUniqueId_ id;
    IVX_NCCL_CHECK(g_rccl.GetUniqueId(&id));
    ncclComm_t comm = nullptr;
    IVX_NCCL_CHECK(g_rccl.CommInitRank(&comm, 1, id, 0));
    constexpr size_t N = 4096;  // bytes per message
    uint8_t* dev = nullptr;     // [send | recv | gathered]
    if (hipMalloc(reinterpret_cast<void**>(&dev), 3 * N) != hipSuccess) {
        (void)g_rccl.CommDestroy(comm);
        ivx_set_error("ivx_comm_selftest: device allocation failed");
        return IVX_ERR_HIP;
    }
    std::vector<uint8_t> host(3 * N, 0);
    for (size_t i = 0; i < N; ++i) host[i] = (uint8_t)(i * 31u + 7u);
    hipStream_t s = c->stream;
    int result = IVX_OK;
    do {
        if (ivx_memcpy_async(dev, host.data(), 3 * N, hipMemcpyHostToDevice, s) != hipSuccess) {
            result = IVX_ERR_HIP;
            break;
        }
        // the neighbour exchange's call pattern, with this rank as its own neighbour
        bool group_ok = g_rccl.GroupStart() == 0;
        if (group_ok) {  // (a group that was opened is always closed, whatever the calls inside it returned)
            const bool sent = g_rccl.Send(dev, N, NCCL_UINT8, 0, comm, s) == 0 && g_rccl.Recv(dev + N, N, NCCL_UINT8, 0, comm, s) == 0;
            group_ok = (g_rccl.GroupEnd() == 0) && sent;
        }
        if (!group_ok) {
            ivx_set_error("ivx_comm_selftest: grouped ncclSend / ncclRecv failed");
            result = IVX_ERR_HIP;
            break;
        }
        // the same exchange the way the protocol runs it: on a second stream, behind an event recorded on the context's stream, an event behind
        // the arrival that the context's stream waits for (second halves of the buffers; the all-gather below reads what arrived)
        {
            hipStream_t xs = nullptr;
            hipEvent_t packed = nullptr, arrived = nullptr;
            bool ok = hipStreamCreateWithFlags(&xs, hipStreamNonBlocking) == hipSuccess && hipEventCreateWithFlags(&packed, hipEventDisableTiming) == hipSuccess &&
                      hipEventCreateWithFlags(&arrived, hipEventDisableTiming) == hipSuccess;
            ok = ok && hipEventRecord(packed, s) == hipSuccess && hipStreamWaitEvent(xs, packed, 0) == hipSuccess;
            if (ok && g_rccl.GroupStart() == 0) {
                const bool sent = g_rccl.Send(dev + N / 2, N / 2, NCCL_UINT8, 0, comm, xs) == 0 && g_rccl.Recv(dev + N + N / 2, N / 2, NCCL_UINT8, 0, comm, xs) == 0;
                ok = (g_rccl.GroupEnd() == 0) && sent;
            } else {
                ok = false;
            }
            ok = ok && hipEventRecord(arrived, xs) == hipSuccess && hipStreamWaitEvent(s, arrived, 0) == hipSuccess;
            if (xs) (void)hipStreamSynchronize(xs);
            if (packed) (void)hipEventDestroy(packed);
            if (arrived) (void)hipEventDestroy(arrived);
            if (xs) (void)hipStreamDestroy(xs);
            if (!ok) {
                ivx_set_error("ivx_comm_selftest: grouped ncclSend / ncclRecv on a second stream between two events failed");
                result = IVX_ERR_HIP;
                break;
            }
        }
        // the record all-gather's (64-bit words)
        if (g_rccl.AllGather(dev + N, dev + 2 * N, N / 8, NCCL_INT64, comm, s) != 0) {
            ivx_set_error("ivx_comm_selftest: ncclAllGather failed");
            result = IVX_ERR_HIP;
            break;
        }
        if (ivx_memcpy_async(host.data(), dev, 3 * N, hipMemcpyDeviceToHost, s) != hipSuccess || ivx_stream_sync(s) != hipSuccess) {
            result = IVX_ERR_HIP;
            break;
        }
        for (size_t i = 0; i < N && result == IVX_OK; ++i)
            if (host[N + i] != host[i] || host[2 * N + i] != host[i]) {
                ivx_set_error("ivx_comm_selftest: byte %zu came back as %u / %u, sent %u", i, (unsigned)host[N + i], (unsigned)host[2 * N + i], (unsigned)host[i]);
                result = IVX_ERR_STATE;
            }
    } while (false);
    (void)ivx_stream_sync(s);
    (void)hipFree(dev);
    (void)g_rccl.CommDestroy(comm);
    return result;
}

// what the communicator itself says it is: transport (0 RCCL, 1 in-process, 2 shared device), ranks and this process's rank — for RCCL as
// ncclCommCount / ncclCommUserRank report them (a one-rank RCCL communicator makes no library communicator: 1 and 0)
int ivx_comm_info(ivx_comm* m, int* transport, int* nranks, int* rank) {
    IVX_REQUIRE(m && transport && nranks && rank, IVX_ERR_INVALID, "ivx_comm_info: null argument");
    *transport = m->ipc ? 2 : (m->rank < 0 ? 1 : 0);
    *nranks = m->nranks;
    *rank = m->rank;
    if (m->nccl) {
        IVX_NCCL_CHECK(g_rccl.CommCount(m->nccl, nranks));
        IVX_NCCL_CHECK(g_rccl.CommUserRank(m->nccl, rank));
    }
    return IVX_OK;
}

// (test boxes have one GPU) the in-process transport moves its messages as a rank's would be moved — device copies on the communicator's
// stream behind the packing, the slabs' sweeps split around their arrival — instead of reading the neighbour's send buffer in place: the
// choreography of the RCCL transport (events, second stream, split sweeps) on one device, checked bit for bit by tests/test_gpu_slabs.py
int ivx_comm_set_local_copies(ivx_comm* m, int on) {
    IVX_REQUIRE(m && m->rank < 0 && !m->ipc, IVX_ERR_INVALID, "ivx_comm_set_local_copies: an in-process communicator is needed");
    (void)ivx_stream_sync(m->ctx->stream);
    if (m->comm_stream) (void)hipStreamSynchronize(m->comm_stream);
    m->local_copies = on ? 1 : 0;
    m->overlap = on == 1 ? 1 : 0;  // (2: copies on the context's own stream, unsplit sweeps)
    return IVX_OK;
}

void ivx_comm_destroy(ivx_comm* m) {
    if (!m) return;
    if (m->comm_stream) {
        (void)hipStreamSynchronize(m->comm_stream);
        for (int k = 0; k < 3; ++k) {
            (void)hipEventDestroy(m->ev_packed[k]);
            (void)hipEventDestroy(m->ev_arrived[k]);
        }
        (void)hipStreamDestroy(m->comm_stream);
    }
    if (m->nccl) (void)g_rccl.CommDestroy(m->nccl);
    if (m->ipc) {
        if (m->ipc_acker) {
            m->ipc_acker_stop.store(1, std::memory_order_release);
            m->ipc_acker->join();
            delete m->ipc_acker;
        }
        for (int s = 0; s < 2; ++s)
            if (m->ipc_peer_recv[s] && !m->ipc_peer_same_process[s]) (void)hipIpcCloseMemHandle(m->ipc_peer_recv[s]);
        munmap(m->ipc, sizeof(IpcShared));
        if (m->rank == 0) (void)shm_unlink(m->ipc_name);
    }
    delete m;
}

int ivx_slab_create(ivx_comm* m, ivx_grid* g, int rank, ivx_slab** out) {
    IVX_REQUIRE(m && g && out && rank >= 0 && rank < m->nranks && (m->rank < 0 || rank == m->rank), IVX_ERR_INVALID, "ivx_slab_create: bad argument");
    IVX_REQUIRE(g->ctx == m->ctx, IVX_ERR_INVALID, "ivx_slab_create: grid and communicator belong to different contexts");
    *out = nullptr;
    ivx_slab* sl = new (std::nothrow) ivx_slab();
    IVX_REQUIRE(sl, IVX_ERR_CAPACITY, "ivx_slab_create: out of host memory");
    sl->comm = m;
    sl->grid = g;
    sl->rank = rank;
    sl->halo_bytes = ivx_halo_bytes(g);
    sl->face_bytes = ivx_region_face_bytes(g);
    sl->rec_words = ivx_step_record_words();
    sl->has_lo = rank > 0;
    sl->has_hi = rank + 1 < m->nranks;
    sl->enqueued = 0;
    sl->map_offset = 0;
    sl->local_err = IVX_OK;
    const size_t msg = (sl->halo_bytes + sl->face_bytes + 255) & ~(size_t)255;
    bool ok = true;
    for (int s = 0; s < 2; ++s) {
        ok = ok && hipMalloc(reinterpret_cast<void**>(&sl->send[s]), msg) == hipSuccess;
        ok = ok && hipMalloc(reinterpret_cast<void**>(&sl->recv[s]), msg) == hipSuccess;
        sl->send2[s] = nullptr;
        if (m->rank < 0) ok = ok && hipMalloc(reinterpret_cast<void**>(&sl->send2[s]), msg) == hipSuccess;
        sl->ghost[s] = sl->recv[s];
    }
    ok = ok && hipMalloc(reinterpret_cast<void**>(&sl->record), sl->rec_words * 8) == hipSuccess;
    ok = ok && hipMalloc(reinterpret_cast<void**>(&sl->gathered), sl->rec_words * 8 * (size_t)m->nranks) == hipSuccess;
    sl->host_head = sl->host_head_dev = nullptr;
    sl->head_seq = 0;
    {
        const size_t head_bytes = ((size_t)m->nranks * HEAD_WORDS + 1) * 8;
        ok = ok && hipHostMalloc(reinterpret_cast<void**>(&sl->host_head), head_bytes, hipHostMallocMapped) == hipSuccess;
        if (ok) memset(sl->host_head, 0, head_bytes);
        ok = ok && hipHostGetDevicePointer(reinterpret_cast<void**>(&sl->host_head_dev), sl->host_head, 0) == hipSuccess;
    }
    if (!ok) {
        ivx_set_error("ivx_slab_create: device allocation failed");
        ivx_slab_destroy(sl);
        return IVX_ERR_HIP;
    }
    if (m->ipc) {  // shared-device transport (collective: every rank creates its slab): publish the receive buffers, open the neighbours'
        IpcRank& me = m->ipc->ranks[rank];
        // a further slab on the same communicator: the handles of the last one must have been imported by every neighbour before they are
        // replaced, and the neighbours' imports of that generation are closed here
        const unsigned long long gen = ++m->ipc_slab_gen;
        const unsigned long long n_nbr = (sl->has_lo ? 1u : 0u) + (sl->has_hi ? 1u : 0u);
        if (!ipc_wait([&] { return me.opened.load(std::memory_order_acquire) >= n_nbr * (gen - 1); }, "the neighbours to import the previous slab's buffers")) {
            ivx_slab_destroy(sl);
            return IVX_ERR_STATE;
        }
        for (int s = 0; s < 2; ++s)
            if (m->ipc_peer_recv[s]) {
                if (!m->ipc_peer_same_process[s]) (void)hipIpcCloseMemHandle(m->ipc_peer_recv[s]);
                m->ipc_peer_recv[s] = nullptr;
                m->ipc_peer_same_process[s] = 0;
            }
        for (int s = 0; s < 2; ++s)
            if (hipIpcGetMemHandle(&me.recv[s], sl->recv[s]) != hipSuccess) {
                ivx_set_error("ivx_slab_create: hipIpcGetMemHandle failed (HSA_ENABLE_IPC_MODE_LEGACY=0 must be set on this pool)");
                ivx_slab_destroy(sl);
                return IVX_ERR_HIP;
            }
        me.pid = (unsigned long long)getpid();
        for (int s = 0; s < 2; ++s) me.recv_ptr[s] = (unsigned long long)(uintptr_t)sl->recv[s];
        me.posted.store(gen, std::memory_order_release);
        for (int s = 0; s < 2; ++s) {
            if (!(s ? sl->has_hi : sl->has_lo)) continue;
            const int peer = s ? rank + 1 : rank - 1;
            if (!ipc_wait([&] { return m->ipc->ranks[peer].posted.load(std::memory_order_acquire) >= gen; }, "a neighbour's slab")) {
                ivx_slab_destroy(sl);
                return IVX_ERR_STATE;
            }
            if (m->ipc->ranks[peer].pid == (unsigned long long)getpid()) {  // several ranks of one process (one context each): the pointer itself
                m->ipc_peer_recv[s] = reinterpret_cast<void*>((uintptr_t)m->ipc->ranks[peer].recv_ptr[s ? 0 : 1]);
                m->ipc_peer_same_process[s] = 1;
            } else if (hipIpcOpenMemHandle(&m->ipc_peer_recv[s], m->ipc->ranks[peer].recv[s ? 0 : 1], hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
                ivx_set_error("ivx_slab_create: hipIpcOpenMemHandle failed (rank %d -> %d)", rank, peer);
                ivx_slab_destroy(sl);
                return IVX_ERR_HIP;
            }
            m->ipc->ranks[peer].opened.fetch_add(1, std::memory_order_release);
        }
    }
    *out = sl;
    return IVX_OK;
}

void ivx_slab_destroy(ivx_slab* sl) {
    if (!sl) return;
    (void)ivx_stream_sync(sl->comm->ctx->stream);
    if (sl->comm->comm_stream) (void)hipStreamSynchronize(sl->comm->comm_stream);
    if (sl->grid) {  // (the events are the communicator's: a grid that outlives it must not wait on them)
        sl->grid->ghost_event = sl->grid->face_ids_event = nullptr;
        sl->grid->ghost_split = 0;
    }
    (void)ivx_halo_clear(sl->grid, 0);
    (void)ivx_halo_clear(sl->grid, 1);
    for (int s = 0; s < 2; ++s) {
        if (sl->send[s]) (void)hipFree(sl->send[s]);
        if (sl->recv[s]) (void)hipFree(sl->recv[s]);
        if (sl->send2[s]) (void)hipFree(sl->send2[s]);
    }
    if (sl->record) (void)hipFree(sl->record);
    if (sl->gathered) (void)hipFree(sl->gathered);
    if (sl->host_head) (void)hipHostFree(sl->host_head);
    delete sl;
}

// One step of the protocol for the slabs of this process (RCCL: exactly one; in-process communicator: all ranks, in rank order).
// Nothing in here waits for the GPU.
int ivx_slabs_step_enqueue(ivx_slab** slabs, size_t n) {
    IVX_REQUIRE(slabs && n >= 1 && slabs[0], IVX_ERR_INVALID, "ivx_slabs_step_enqueue: null argument");
    ivx_comm* c = slabs[0]->comm;
    IVX_REQUIRE(c->rank >= 0 ? n == 1 : n == (size_t)c->nranks, IVX_ERR_INVALID, "ivx_slabs_step_enqueue: %zu slabs for a communicator of %d ranks (%s)", n,
                c->nranks, c->rank >= 0 ? "RCCL: one slab per process" : "in-process: all of them");
    for (size_t i = 0; i < n; ++i)
        IVX_REQUIRE(slabs[i] && slabs[i]->comm == c && (c->rank >= 0 || slabs[i]->rank == (int)i), IVX_ERR_INVALID, "ivx_slabs_step_enqueue: slab %zu out of order", i);
    int rc;
    // A LOCAL failure between the collectives (a stage that cannot be enqueued: capacity, a bad program) must not leave the other ranks waiting
    // inside ncclGroupEnd / ncclAllGather for a rank that has returned: the rank notes its first error, goes on through every exchange of the
    // step with whatever its buffers hold, and raises bit 3 of the flags word of its record; collect then fails on EVERY rank (the flags are
    // OR-ed over the gathered records), the failing rank with its own code. Only a failing exchange itself still returns at once.
    int local_err = IVX_OK;
    auto note = [&](int code) {
        if (code && !local_err) local_err = code;
    };
    // An arrival event that the step before handed to a grid and no sweep consumed (a slab with a local error skips its sweeps): the context's
    // stream takes the wait here, before anything of this step re-packs the send buffers or reads the receive buffers those messages used.
    for (size_t i = 0; i < n; ++i) {
        ivx_grid* g = slabs[i]->grid;
        if (g->ghost_event) (void)hipStreamWaitEvent(c->ctx->stream, static_cast<hipEvent_t>(g->ghost_event), 0);
        if (g->face_ids_event) (void)hipStreamWaitEvent(c->ctx->stream, static_cast<hipEvent_t>(g->face_ids_event), 0);
        g->ghost_event = g->face_ids_event = nullptr;
        g->ghost_split = 0;
    }
    if (c->nranks == 1) {
        // a world of one has nobody to wait for between the phases: the whole step as one enqueue, then the record (written where the gather would
        // put it, with the step's small results on the way)
        ivx_slab* sl = slabs[0];
        (void)ivx_halo_clear(sl->grid, 0);
        (void)ivx_halo_clear(sl->grid, 1);
        note(ivx_voxel_step_enqueue(sl->grid, IVX_STAGE_SAMPLE | IVX_STAGE_DERIVE | IVX_STAGE_OCCUPIED | IVX_STAGE_REGIONS | IVX_STAGE_INERTIA | IVX_STAGE_REMESH));
        if (!local_err) note(ivx_step_record_enqueue(sl->grid, sl->gathered));
        if (local_err) {
            unsigned long long head[18];
            memset(head, 0, sizeof(head));
            head[17] = 8ull;
            IVX_HIP_CHECK(ivx_memcpy_async(sl->gathered, head, sizeof(head), hipMemcpyHostToDevice, c->ctx->stream));
            IVX_HIP_CHECK(ivx_stream_sync(c->ctx->stream));  // (`head` is a local)
        }
        sl->local_err = local_err;
        sl->enqueued = 1;
        return IVX_OK;
    }
    // (the in-process transport moving its messages like RCCL does, a test mode: what a slab's receive buffers hold from the exchanges before is
    // overwritten first — a static scene sends the same bytes step after step, and a sweep that read a ghost layer ahead of its arrival would
    // go unnoticed)
    const bool scribble = c->rank < 0 && c->local_copies == 1 && c->overlap;
    if (scribble)
        for (size_t i = 0; i < n; ++i)
            for (int side = 0; side < 2; ++side) IVX_HIP_CHECK(ivx_memset_async(slabs[i]->recv[side], 0xA5, slabs[i]->halo_bytes + slabs[i]->face_bytes, c->ctx->stream));
    // 1. sample, exchange the face planes
    for (size_t i = 0; i < n; ++i) {
        ivx_slab* sl = slabs[i];
        note(ivx_voxel_step_enqueue(sl->grid, IVX_STAGE_SAMPLE));
        if (sl->has_lo || sl->has_hi) note(ivx_halo_pack_both_enqueue(sl->grid, sl->has_lo ? sl->send[0] : nullptr, sl->has_hi ? sl->send[1] : nullptr, 0));
    }
    if ((rc = exchange(slabs, n, false))) return rc;
    // 2. derived state + slab-local regions (+ moments and occupied ranges: they need nothing more from the neighbours); the planes
    // again, now with the post-demotion chunk kinds the mesher's upper-layer rule needs, and the faces' component ids behind them.
    // (Exchanges on the communicator's stream: the planes are packed and sent as soon as the derive sweep is through — they travel while the
    // region stages run —, the ids, which need the region stages, in a message of their own.)
    const uint32_t stages2 = IVX_STAGE_DERIVE | IVX_STAGE_OCCUPIED | IVX_STAGE_REGIONS | IVX_STAGE_INERTIA;
    const bool early = c->overlap && !c->ipc && (c->rank >= 0 || c->local_copies);
    for (size_t i = 0; i < n; ++i) {
        ivx_slab* sl = slabs[i];
        install_ghosts(sl);
        ivx_step_preset_ahead(sl->grid, IVX_SCRATCH_SN);  // the remesh phase below has no first kernel to host its preset
        uint8_t** pb = pack_bufs(sl, true);
        const bool faces = sl->has_lo || sl->has_hi;
        if (!early) {
            if (!local_err) note(ivx_voxel_step_enqueue(sl->grid, stages2));
            if (faces) note(ivx_halo_pack_both_enqueue(sl->grid, sl->has_lo ? pb[0] : nullptr, sl->has_hi ? pb[1] : nullptr, 1));
            continue;
        }
        if (!local_err) note(ivx_voxel_step_enqueue_part(sl->grid, stages2, 1u));
        if (faces) note(ivx_launch_halo_pack_parts(sl->grid, sl->has_lo ? pb[0] : nullptr, sl->has_hi ? pb[1] : nullptr, 1u));
        if (c->rank >= 0) {  // (one slab per process: its planes leave now; the in-process transport sends all slabs' planes below)
            if ((rc = exchange(slabs, n, true, 1u))) return rc;
            if (!local_err) note(ivx_voxel_step_enqueue_part(sl->grid, stages2, 2u));
            if (faces) note(ivx_launch_halo_pack_parts(sl->grid, sl->has_lo ? pb[0] : nullptr, sl->has_hi ? pb[1] : nullptr, 2u));
        }
    }
    if (early && c->rank < 0) {
        if (scribble)  // (every reader of the first exchange's planes is through)
            for (size_t i = 0; i < n; ++i)
                for (int side = 0; side < 2; ++side) IVX_HIP_CHECK(ivx_memset_async(slabs[i]->recv[side], 0x5A, slabs[i]->halo_bytes, c->ctx->stream));
        if ((rc = exchange(slabs, n, true, 1u))) return rc;
        for (size_t i = 0; i < n; ++i) {
            ivx_slab* sl = slabs[i];
            uint8_t** pb = pack_bufs(sl, true);
            if (!local_err) note(ivx_voxel_step_enqueue_part(sl->grid, stages2, 2u));
            if (sl->has_lo || sl->has_hi) note(ivx_launch_halo_pack_parts(sl->grid, sl->has_lo ? pb[0] : nullptr, sl->has_hi ? pb[1] : nullptr, 2u));
        }
    }
    if ((rc = exchange(slabs, n, true, early ? 2u : 3u))) return rc;
    // 3. remesh (ghost layers in place), the slab's record, the one small all-gather
    for (size_t i = 0; i < n; ++i) {
        ivx_slab* sl = slabs[i];
        install_ghosts(sl);
        // (the pass over the neighbour's face ids and the record ride in the remesh stage's launches; a world of one writes its record where the
        // gather would put it)
        unsigned long long* rec = c->nranks == 1 ? sl->gathered : sl->record;
        // (in-process: the record role also writes the record's head straight into its place in the gathered block)
        sl->grid->record_head_copy = (c->rank < 0 && !c->ipc && c->nranks > 1) ? slabs[0]->gathered + i * HEAD_WORDS : nullptr;
        sl->grid->record_head_words = (uint32_t)HEAD_WORDS;
        if (!local_err) note(ivx_slab_remesh_enqueue(sl->grid, sl->has_hi ? sl->ghost[1] + sl->halo_bytes : nullptr, rec));
        if (local_err) {  // a record that says so (words 0, 1: no components, no pairs; word 17: the flags)
            unsigned long long head[18];
            memset(head, 0, sizeof(head));
            head[17] = 8ull;
            IVX_HIP_CHECK(ivx_memcpy_async(rec, head, sizeof(head), hipMemcpyHostToDevice, c->ctx->stream));
            if (sl->grid->record_head_copy) IVX_HIP_CHECK(ivx_memcpy_async(sl->grid->record_head_copy, head, sizeof(head), hipMemcpyHostToDevice, c->ctx->stream));
            IVX_HIP_CHECK(ivx_stream_sync(c->ctx->stream));  // (`head` is a local)
        }
        sl->local_err = local_err;
    }
    if ((rc = all_gather(slabs, n, HEAD_WORDS))) return rc;
    for (size_t i = 0; i < n; ++i) slabs[i]->enqueued = 1;
    return IVX_OK;
}

// Waits once, finishes the cross-slab union-find on the host (a handful of components and pairs) and fills the global results of
// every slab of this process. The same arithmetic on every rank: all see the same records.
int ivx_slabs_step_collect(ivx_slab** slabs, size_t n, ivx_slab_result* out) {
    IVX_REQUIRE(slabs && n >= 1 && out && slabs[0] && slabs[0]->enqueued, IVX_ERR_STATE, "ivx_slabs_step_collect: nothing enqueued");
    ivx_comm* c = slabs[0]->comm;
    hipStream_t s = c->ctx->stream;
    const int world = c->nranks;
    ivx_slab* keeper = slabs[0];
    std::vector<unsigned long long>& rec = keeper->host_records;
    size_t words = HEAD_WORDS;
    rec.resize((size_t)world * keeper->rec_words);
    {
        const uint32_t n_words = (uint32_t)((size_t)world * HEAD_WORDS);
        const unsigned long long want = ++keeper->head_seq;
        IVX_KLAUNCH(k_slab_publish, dim3(1), dim3(256), 0, s, keeper->gathered, keeper->host_head_dev, n_words, want);
        IVX_HIP_CHECK(hipGetLastError());
        const volatile unsigned long long* bell = keeper->host_head + n_words;
        bool rung = false;
        const auto t0 = std::chrono::steady_clock::now();
        for (uint32_t it = 0;; ++it) {  // bounded poll (2 ms), then the runtime's blocking wait
            if (*bell == want) {
                rung = true;
                break;
            }
            __builtin_ia32_pause();
            if ((it & 255u) == 255u && std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() > 2000) break;
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        if (!rung) IVX_HIP_CHECK(ivx_stream_sync(s));
        memcpy(rec.data(), keeper->host_head, (size_t)n_words * 8);
    }
    unsigned long long max_pairs = 0;
    for (int r = 0; r < world; ++r) max_pairs = std::max(max_pairs, rec[(size_t)r * words + 1]);
    if (max_pairs > HEAD_PAIRS) {  // (the same decision on every rank) a rank lists more pairs than the head holds: the full records
        words = keeper->rec_words;
        int rc = all_gather(slabs, n, words);
        if (rc) return rc;
        IVX_HIP_CHECK(ivx_memcpy_async(rec.data(), keeper->gathered, (size_t)world * words * 8, hipMemcpyDeviceToHost, s));
        IVX_HIP_CHECK(ivx_stream_sync(s));
    }
    // error flags: decided on the gathered data every rank holds, so that all ranks fail together
    unsigned long long flags = 0;
    for (int r = 0; r < world; ++r) flags |= rec[(size_t)r * words + 17];
    std::vector<size_t> offs(world + 1, 0);
    for (int r = 0; r < world; ++r) offs[r + 1] = offs[r] + (size_t)rec[(size_t)r * words];
    std::vector<uint32_t> parent(offs[world]);
    for (size_t i = 0; i < parent.size(); ++i) parent[i] = (uint32_t)i;
    auto find = [&](uint32_t x) {
        while (parent[x] != x) {
            parent[x] = parent[parent[x]];
            x = parent[x];
        }
        return x;
    };
    bool pair_overflow = false, bad_pair = false;
    for (int r = 0; r + 1 < world; ++r) {
        const unsigned long long* q = rec.data() + (size_t)r * words;
        const size_t np = (size_t)q[1];
        if (np > IVX_MAX_FACE_PAIRS) {
            pair_overflow = true;
            continue;
        }
        for (size_t k = 0; k < np; ++k) {
            // (a pair names a component of this slab and one of the next by their slab-local ids: ids beyond the slabs' component counts can only
            // come from a message that was read before it had arrived or was damaged on the way — an error, not an index)
            if (q[28 + 2 * k] >= rec[(size_t)r * words] || q[29 + 2 * k] >= rec[(size_t)(r + 1) * words]) {
                bad_pair = true;
                continue;
            }
            const uint32_t a = find((uint32_t)(offs[r] + q[28 + 2 * k])), b = find((uint32_t)(offs[r + 1] + q[29 + 2 * k]));
            if (a != b) parent[std::max(a, b)] = std::min(a, b);
        }
    }
    // roots are minimal members, so numbering them in index order = ordering the regions by first occurrence
    std::vector<uint32_t> region_id(parent.size(), 0xFFFFFFFFu), ids(parent.size());
    uint32_t n_regions = 0;
    for (size_t i = 0; i < parent.size(); ++i) {
        const uint32_t root = find((uint32_t)i);
        if (region_id[root] == 0xFFFFFFFFu) region_id[root] = n_regions++;
        ids[i] = region_id[root];
    }
    double moments[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int r = 0; r < world; ++r)  // fixed rank order: bitwise reproducible
        for (int q = 0; q < 10; ++q) {
            double v;
            memcpy(&v, &rec[(size_t)r * words + 18 + q], 8);
            moments[q] += v;
        }
    uint32_t occ[12] = {0};
    bool any = false;
    for (int r = 0; r < world; ++r) {
        const unsigned long long* q = rec.data() + (size_t)r * words;
        if (q[3] == 0) continue;  // (hi of the chunk range along x: zero = the slab holds no voxel)
        for (int k = 0; k < 12; ++k) {
            const uint32_t v = (uint32_t)q[2 + k];
            occ[k] = !any ? v : ((k & 1) ? std::max(occ[k], v) : std::min(occ[k], v));
        }
        any = true;
    }
    unsigned long long tri_total = 0, voff = 0, ioff = 0;
    std::vector<unsigned long long> voffs(world), ioffs(world);
    for (int r = 0; r < world; ++r) {
        voffs[r] = voff;
        ioffs[r] = ioff;
        voff += rec[(size_t)r * words + 14];
        ioff += rec[(size_t)r * words + 15];
        tri_total += rec[(size_t)r * words + 15] / 3;
    }
    if (flags & 8u) {  // a rank could not enqueue its step: every rank fails here, after the step's collectives, the failing one with its own code
        int mine = IVX_OK;
        for (size_t i = 0; i < n; ++i) {
            if (slabs[i]->local_err && !mine) mine = slabs[i]->local_err;
            slabs[i]->enqueued = 0;
            slabs[i]->grid->pending_stages = 0;
        }
        IVX_HIP_CHECK(ivx_stream_sync(s));
        if (mine) return mine;  // (ivx_last_error holds the stage's own message)
        ivx_set_error("ivx_slabs_step_collect: another rank could not enqueue its step");
        return IVX_ERR_STATE;
    }
    for (size_t i = 0; i < n; ++i) {
        ivx_slab* sl = slabs[i];
        ivx_slab_result& o = out[i];
        memset(&o, 0, sizeof(o));
        ivx_step_result local;
        int rc = ivx_voxel_step_collect(sl->grid, &local);  // the stream is idle: stage timings + the mesh-buffer check
        if (rc) return rc;
        const int r = sl->rank;
        const unsigned long long* q = rec.data() + (size_t)r * words;
        o.region_count = n_regions;
        o.local_region_count = (uint32_t)q[0];
        o.first_local_component = (uint32_t)offs[r];
        memcpy(o.moments, moments, sizeof(moments));
        memcpy(o.occupied, occ, sizeof(occ));
        o.mesh.n_vertices = (uint32_t)q[14];
        o.mesh.n_indices = (uint32_t)q[15];
        o.mesh.n_submeshes = (uint32_t)q[16];
        o.vertex_offset = voffs[r];
        o.index_offset = ioffs[r];
        o.total_triangles = tri_total;
        memcpy(o.stage_ms, local.stage_ms, sizeof(o.stage_ms));
        sl->enqueued = 0;
    }
    // the map slab-local component -> global region of every rank stays with the keeper for ivx_slab_region_map
    keeper->host_records.resize((size_t)world * words + ids.size() + (size_t)world + 1);
    keeper->map_offset = (size_t)world * words;
    unsigned long long* tail = keeper->host_records.data() + (size_t)world * words;
    for (int r = 0; r <= world; ++r) tail[r] = offs[r];
    for (size_t i = 0; i < ids.size(); ++i) tail[world + 1 + i] = ids[i];
    IVX_REQUIRE((flags & 1u) == 0, IVX_ERR_CAPACITY, "ivx_slabs_step_collect: a chunk has more than 254 local regions");
    IVX_REQUIRE((flags & 4u) == 0, IVX_ERR_CAPACITY, "ivx_slabs_step_collect: a slab has 65535 or more components: its face ids do not fit the 16-bit exchange format");
    IVX_REQUIRE(!pair_overflow, IVX_ERR_CAPACITY, "ivx_slabs_step_collect: more than %d cross-slab region pairs on one face", IVX_MAX_FACE_PAIRS);
    IVX_REQUIRE(!bad_pair, IVX_ERR_STATE, "ivx_slabs_step_collect: a cross-slab region pair names a component its slab does not have (a neighbour's face ids were damaged or read before they arrived)");
    return IVX_OK;
}

// global region id of every slab-local component of rank `rank` (after ivx_slabs_step_collect on the communicator's first slab
// of this process): out[k] for k < local_region_count
int ivx_slab_region_map(ivx_slab* keeper, int rank, uint32_t* out, size_t cap, size_t* n_out) {
    IVX_REQUIRE(keeper && n_out && rank >= 0 && rank < keeper->comm->nranks, IVX_ERR_INVALID, "ivx_slab_region_map: bad argument");
    const int world = keeper->comm->nranks;
    const std::vector<unsigned long long>& h = keeper->host_records;
    const size_t base = keeper->map_offset;
    IVX_REQUIRE(base, IVX_ERR_STATE, "ivx_slab_region_map: no collected step");
    const size_t lo = (size_t)h[base + rank], hi = (size_t)h[base + rank + 1];
    *n_out = hi - lo;
    IVX_REQUIRE(hi - lo <= cap || !out, IVX_ERR_CAPACITY, "ivx_slab_region_map: %zu components exceed capacity %zu", hi - lo, cap);
    if (out)
        for (size_t k = lo; k < hi; ++k) out[k - lo] = (uint32_t)h[base + world + 1 + k];
    return IVX_OK;
}

}  // extern "C"
