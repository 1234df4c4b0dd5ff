// Many objects per launch. The reference's per-frame unit is the manager, not the object: every voxel object is
// synced each frame (impact_voxel/src/lib.rs:729-733, engine/src/tasks.rs:376-399), the fragments of an impact come into being together
// (interaction/fracturing.rs:1047-1189), its own stress scene holds a thousand small objects. One object's step, edit or sync is ~10 launches
// of a few microseconds each — for a small object all of its cost — so N objects looped are N x that, and what has to go is the launches,
// not the work.
//
// The mechanism is a RECORDER in front of the kernel launches. Between ivx_many_begin and ivx_many_flush a launch of a kernel that has a
// `_many` twin is not issued: its argument block and its block count are written down at the object's position in the chain. The per-object
// host code (stage logic, scratch bookkeeping, the submesh manager) runs unchanged, object after object. The flush then issues, chain position
// by chain position, ONE launch for all objects: the twin finds the object of a block by a search over the running block counts (scalar
// loads: the block index is uniform), fetches that object's argument block and runs the very body of the single-object kernel with the
// block index counted from the object's first block. Objects whose chains differ (a stage one of them skips) still merge wherever the same
// kernel stands at the same position; a launch or a stream operation WITHOUT a twin first flushes what has been recorded (ivx_many_break, in
// front of every such call): order on the stream is that of the recorded program in every case, merging is only ever an optimisation.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>

struct ivx_ctx;

// kernels with a twin
enum ivx_many_kernel : int {
    IVX_MK_ABSORB = 0,
    IVX_MK_DERIVE_PLANES,
    IVX_MK_DERIVE_SIGNS,
    IVX_MK_CHUNK_PRE,
    IVX_MK_LIST_REBUILD,
    IVX_MK_POST1,
    IVX_MK_POST2,
    IVX_MK_EMIT,
    IVX_MK_ASSIGN,
    IVX_MK_GATHER,
    IVX_MK_SN_EMIT_SLOTS,
    IVX_MK_SN_EMIT_GENERAL_SLOTS,
    IVX_MK_INERTIA_DENSE,
    IVX_MK_CLIP,
    IVX_MK_SVC_COUNT,  // collidable-against-voxel-object contacts: count | scan | emit (contacts.hip)
    IVX_MK_SVC_SCAN,
    IVX_MK_SVC_EMIT,
    IVX_MK_SCAN_COUNTS,  // contacts between two voxel objects: count (probes of one against the other's field) | scan | emit (collide.hip)
    IVX_MK_MUT_COUNT,
    IVX_MK_MUT_EMIT,
    IVX_MK_PROBE_SELECT_SMALL,  // collision probes of the listed chunk submeshes: select (two LDS sizes) | gather (collide.hip)
    IVX_MK_PROBE_SELECT_FULL,
    IVX_MK_PROBE_GATHER,
    IVX_MK_ZERO,    // fill a device range with one word (the twin of a hipMemsetAsync of 0x00 or 0xFF bytes)
    IVX_MK_UPLOAD,  // host words to a device range (the twin of a small hipMemcpyAsync host -> device): the words ride in the flush's one staging copy
    IVX_MK_COUNT
};

// the twin's launcher: `d_argv` n argument blocks back to back (arg_bytes each), `d_block_end` the running block counts, `total` blocks in all
typedef int (*ivx_many_launch_fn)(hipStream_t s, const void* d_argv, const uint32_t* d_block_end, uint32_t n, uint32_t total);
struct ivx_many_reg {
    ivx_many_launch_fn fn;
    uint32_t arg_bytes;
};
void ivx_many_register(int kernel, ivx_many_launch_fn fn, uint32_t arg_bytes);

// recording state of the calling thread: null = launches are issued as they come
bool ivx_many_recording();
// at the top of every entry point that enqueues work for an object: see many.cpp
struct ivx_many_other_context {
    explicit ivx_many_other_context(const ivx_ctx* c);
    ~ivx_many_other_context();
    bool suspended;
};
// true: written down (nothing was launched); false: not recording, or `c` is not the context the batch is recorded for (what has been
// recorded goes out first; the launch belongs on c's stream) — the caller launches
// `owner`: the object the launch belongs to (its grid). A batch recorded through the bare bracket (ivx_many_begin ... _flush, no
// ivx_many_object) keeps a chain per owner, in the order the owners first appear, so that its objects merge front by front as well.
bool ivx_many_capture(const ivx_ctx* c, const void* owner, int kernel, uint32_t blocks, const void* args, uint32_t arg_bytes);
// in front of every launch / stream operation that has no twin: what has been recorded goes out first. A flush that fails is remembered on
// the batch's context (ivx_many_error): the launches it dropped never reach the stream, and whoever waits for their results must say so
int ivx_many_break();
// the sticky error of a failed flush on this context (IVX_OK if none); `clear`: reported now, forget it
int ivx_many_error(ivx_ctx* c, bool clear);
// the context's recorder and its staging ring (ivx_shutdown)
void ivx_many_release(ivx_ctx* c);
// how many flushes (explicit or by a break) this thread's recorder has made: whoever recorded something and later needs it ON the stream
// compares the count at recording time with the count now instead of forcing a flush that may have happened long ago
uint64_t ivx_many_flush_count();
// the next captures belong to object `i` of the batch (chain positions count per object)
void ivx_many_object(uint32_t i);

// zero `bytes` (a multiple of 4) at d_ptr / copy `bytes` (a multiple of 4) from host memory to d_dst: recorded when a batch is being recorded
// on context `c` (true), else the caller issues the stream operation itself
bool ivx_many_zero(const ivx_ctx* c, const void* owner, void* d_ptr, size_t bytes);
bool ivx_many_fill(const ivx_ctx* c, const void* owner, void* d_ptr, uint32_t word, size_t bytes);  // (every 4-byte word of the range = `word`)
bool ivx_many_upload(const ivx_ctx* c, const void* owner, void* d_dst, const void* h_src, size_t bytes);

template <typename A>
static inline bool ivx_many_try(const ivx_ctx* c, const void* owner, int kernel, uint32_t blocks, const A& a) {
    static_assert(sizeof(A) % 8 == 0, "argument blocks are copied as 8-byte words");
    return ivx_many_capture(c, owner, kernel, blocks, &a, (uint32_t)sizeof(A));
}

// (what stands in front of every launch and asynchronous stream operation of the library that has no twin)
#define IVX_KLAUNCH(k, grid, block, shmem, stream, ...)               \
    do {                                                              \
        (void)ivx_many_break();                                       \
        hipLaunchKernelGGL(k, grid, block, shmem, stream, __VA_ARGS__); \
    } while (0)

#ifdef __HIPCC__
// first i with block_end[i] > b (uniform: scalar loads and a scalar loop)
__device__ __forceinline__ uint32_t ivx_many_find(const uint32_t* __restrict__ block_end, uint32_t n, uint32_t b) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (block_end[mid] > b) hi = mid;
        else lo = mid + 1u;
    }
    return lo;
}
// The twin of a kernel whose body is `BODY(const ARGS&, bid, nb)`. The argument block comes by 8-byte words through the scalar cache.
#define IVX_MANY_TWIN(name, ARGS, BODY, ...)                                                                                      \
    __global__ __VA_ARGS__ void name(const ARGS* __restrict__ argv, const uint32_t* __restrict__ block_end, uint32_t n) {          \
        const uint32_t i = ivx_many_find(block_end, n, blockIdx.x);                                                                \
        const uint32_t b0 = i ? block_end[i - 1u] : 0u;                                                                            \
        const ARGS a = argv[i];                                                                                                    \
        BODY(a, blockIdx.x - b0, block_end[i] - b0);                                                                               \
    }
#define IVX_MANY_LAUNCHER(fn_name, twin, ARGS, threads)                                                                            \
    static int fn_name(hipStream_t s, const void* d_argv, const uint32_t* d_block_end, uint32_t n, uint32_t total) {               \
        hipLaunchKernelGGL(twin, dim3(total), dim3(threads), 0, s, static_cast<const ARGS*>(d_argv), d_block_end, n);              \
        return hipGetLastError() == hipSuccess ? 0 : -3;                                                                           \
    }
#endif
