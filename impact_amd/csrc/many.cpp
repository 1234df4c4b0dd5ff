// The recorder behind ivx_many_begin / ivx_many_flush (many.hpp): per object of the batch the chain of captured launches; the flush merges,
// front by front, the entries of one kernel into one launch of its twin.
#include "many.hpp"
#include <execinfo.h>

#include <algorithm>
#include <new>
#include <vector>

#include "ivx_internal.hpp"

namespace {

ivx_many_reg g_regs[IVX_MK_COUNT];

struct Entry {
    int kernel;
    uint32_t blocks, off, bytes;
    uint32_t payload_off, payload_bytes;  // IVX_MK_UPLOAD: the host words, kept in the arena until the flush stages them
};
struct RangeArgs {  // IVX_MK_ZERO / IVX_MK_UPLOAD
    uint32_t* dst;
    const uint32_t* src;
    unsigned long long words;
    uint32_t fill, pad;  // (no src: the word every element gets)
};
__global__ __launch_bounds__(256) void k_range_many(const RangeArgs* __restrict__ argv, const uint32_t* __restrict__ block_end, uint32_t n) {
    const uint32_t i = ivx_many_find(block_end, n, blockIdx.x);
    const uint32_t b0 = i ? block_end[i - 1u] : 0u, nb = block_end[i] - b0, bid = blockIdx.x - b0;
    const RangeArgs a = argv[i];
    for (unsigned long long w = (unsigned long long)bid * 256u + threadIdx.x; w < a.words; w += (unsigned long long)nb * 256u) a.dst[w] = a.src ? a.src[w] : a.fill;
}
int many_range(hipStream_t s, const void* d_argv, const uint32_t* d_block_end, uint32_t n, uint32_t total) {
    hipLaunchKernelGGL(k_range_many, dim3(total), dim3(256), 0, s, static_cast<const RangeArgs*>(d_argv), d_block_end, n);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
const int s_range_registered = (ivx_many_register(IVX_MK_ZERO, many_range, sizeof(RangeArgs)), ivx_many_register(IVX_MK_UPLOAD, many_range, sizeof(RangeArgs)), 0);
constexpr int RING = 4;
struct Recorder {
    ivx_ctx* ctx = nullptr;
    bool on = false;
    uint32_t cur = 0;
    std::vector<std::vector<Entry>> chains;
    std::vector<unsigned char> arena;
    size_t n_entries = 0;
    // staging: pinned host block + device block per ring slot, an event behind the slot's last copy
    void* pinned[RING] = {nullptr, nullptr, nullptr, nullptr};
    void* dev[RING] = {nullptr, nullptr, nullptr, nullptr};
    size_t cap[RING] = {0, 0, 0, 0};
    hipEvent_t ev[RING] = {nullptr, nullptr, nullptr, nullptr};
    bool busy[RING] = {false, false, false, false};
    int slot = 0;
    std::vector<uint32_t> cursor;
    uint64_t flushes = 0, launches = 0, recorded = 0;  // (ivx_many_stats)
    // the bare bracket: chains by owner in order of first appearance (`explicit_cur`: the caller numbers its objects itself, ivx_many_object)
    bool explicit_cur = false;
    std::vector<const void*> owners;
};
// The recorder belongs to its context (ivx_ctx::many_recorder: made by the first ivx_many_begin, its staging ring allocated on the context's
// device, freed by ivx_shutdown). What the calling thread holds is only which recorder it is recording into, if any.
thread_local Recorder* t_rec = nullptr;

// the context's device for the lifetime of the object (allocations and events of the ring belong to it)
struct OnDevice {
    int prev = -1, want;
    explicit OnDevice(int device) : want(device) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != want) (void)hipSetDevice(want);
    }
    ~OnDevice() {
        if (prev >= 0 && prev != want) (void)hipSetDevice(prev);
    }
};

// whatever happens to a flush, what was recorded is gone afterwards: a failed flush must not leave entries for the next batch to issue
struct ClearOnExit {
    Recorder* r;
    ~ClearOnExit();
};

// The merged launches of a flush, in issue order: launch p takes the members [first[p], first[p + 1]) of `members` ((object, entry index)
// pairs). Flat vectors kept in the recorder: the same chains recur frame after frame and nothing is allocated after the first.
struct Plan {
    std::vector<int> kernel;
    std::vector<uint32_t> first;
    std::vector<std::pair<uint32_t, uint32_t>> members;
    std::vector<size_t> off_args, off_ends, off_payload;  // (off_payload: per member, uploads only)
    std::vector<uint32_t> totals;
    void clear() {
        kernel.clear(), first.clear(), members.clear(), off_args.clear(), off_ends.clear(), off_payload.clear(), totals.clear();
    }
};
thread_local Plan t_plan;

int flush_recorded_checked(Recorder* r) {
    if (r->n_entries == 0) return IVX_OK;
    ClearOnExit clear_{r};
    OnDevice dev_{r->ctx->device};
    r->flushes += 1;
    hipStream_t s = r->ctx->stream;
    Plan& plan = t_plan;
    plan.clear();
    const size_t n_obj = r->chains.size();
    r->cursor.assign(n_obj, 0u);
    size_t left = r->n_entries;
    // the front of the chains, position by position: the first unfinished object's next kernel, and everybody whose next entry is that kernel.
    // `lowest`: no object below it has entries left (the scan for the next kernel starts there instead of at object 0)
    size_t lowest = 0;
    while (left) {
        while (lowest < n_obj && r->cursor[lowest] >= r->chains[lowest].size()) ++lowest;
        const int k = r->chains[lowest][r->cursor[lowest]].kernel;
        plan.kernel.push_back(k);
        plan.first.push_back((uint32_t)plan.members.size());
        for (size_t i = lowest; i < n_obj; ++i) {
            // (an object's consecutive entries of this kernel merge as well)
            while (r->cursor[i] < r->chains[i].size() && r->chains[i][r->cursor[i]].kernel == k) {
                plan.members.emplace_back((uint32_t)i, r->cursor[i]);
                r->cursor[i] += 1u;
                left -= 1;
                if (k != IVX_MK_ZERO && k != IVX_MK_UPLOAD) break;  // (range operations of one object may all go together; kernels of one object stay in order)
            }
        }
    }
    const size_t n_launch = plan.kernel.size();
    plan.first.push_back((uint32_t)plan.members.size());
    // lay out argument blocks and running block counts of every launch in one staging block (+ the words of the recorded uploads)
    size_t bytes = 0;
    plan.off_args.resize(n_launch), plan.off_ends.resize(n_launch), plan.totals.resize(n_launch);
    plan.off_payload.assign(plan.members.size(), 0);
    for (size_t p = 0; p < n_launch; ++p) {
        if (plan.kernel[p] != IVX_MK_UPLOAD) continue;
        for (uint32_t m = plan.first[p]; m < plan.first[p + 1]; ++m) {
            const Entry& e = r->chains[plan.members[m].first][plan.members[m].second];
            plan.off_payload[m] = bytes;
            bytes += (e.payload_bytes + 15u) & ~15u;
        }
    }
    for (size_t p = 0; p < n_launch; ++p) {
        const uint32_t ab = g_regs[plan.kernel[p]].arg_bytes, nm = plan.first[p + 1] - plan.first[p];
        plan.off_args[p] = bytes;
        bytes += (size_t)ab * nm;
        bytes = (bytes + 15) & ~(size_t)15;
        plan.off_ends[p] = bytes;
        bytes += 4 * (size_t)nm;
        bytes = (bytes + 15) & ~(size_t)15;
    }
    const int sl = r->slot;
    r->slot = (r->slot + 1) % RING;
    if (!r->ev[sl]) IVX_HIP_CHECK(hipEventCreateWithFlags(&r->ev[sl], hipEventDisableTiming));
    if (r->busy[sl]) {  // (the slot's last flush: its copy AND the twins that read the device block, see below)
        IVX_HIP_CHECK(hipEventSynchronize(r->ev[sl]));
        r->busy[sl] = false;
    }
    if (r->cap[sl] < bytes) {
        if (r->pinned[sl]) (void)hipHostFree(r->pinned[sl]);
        if (r->dev[sl]) (void)hipFree(r->dev[sl]);
        r->pinned[sl] = r->dev[sl] = nullptr;
        r->cap[sl] = 0;
        const size_t cap = std::max<size_t>(2 * bytes, 1 << 16);
        IVX_HIP_CHECK(hipHostMalloc(&r->pinned[sl], cap, hipHostMallocDefault));
        IVX_HIP_CHECK(hipMalloc(&r->dev[sl], cap));
        r->cap[sl] = cap;
    }
    unsigned char* h = static_cast<unsigned char*>(r->pinned[sl]);
    for (size_t p = 0; p < n_launch; ++p) {
        const uint32_t ab = g_regs[plan.kernel[p]].arg_bytes;
        uint32_t run = 0;
        uint32_t* ends = reinterpret_cast<uint32_t*>(h + plan.off_ends[p]);
        for (uint32_t m = plan.first[p]; m < plan.first[p + 1]; ++m) {
            const Entry& e = r->chains[plan.members[m].first][plan.members[m].second];
            const size_t mi = m - plan.first[p];
            memcpy(h + plan.off_args[p] + mi * ab, r->arena.data() + e.off, ab);
            if (plan.kernel[p] == IVX_MK_UPLOAD) {  // the words, and where the twin finds them on the device
                memcpy(h + plan.off_payload[m], r->arena.data() + e.payload_off, e.payload_bytes);
                RangeArgs* ra = reinterpret_cast<RangeArgs*>(h + plan.off_args[p] + mi * ab);
                ra->src = reinterpret_cast<const uint32_t*>(static_cast<const unsigned char*>(r->dev[sl]) + plan.off_payload[m]);
            }
            run += e.blocks;
            ends[mi] = run;
        }
        plan.totals[p] = run;
    }
    IVX_HIP_CHECK(hipMemcpyAsync(r->dev[sl], h, bytes, hipMemcpyHostToDevice, s));
    r->busy[sl] = true;  // (from here on the slot is in use by the stream, whatever happens below: the event goes behind the last twin issued)
    const unsigned char* d = static_cast<const unsigned char*>(r->dev[sl]);
    int rc = IVX_OK;
    for (size_t p = 0; p < n_launch && rc == IVX_OK; ++p) {
        if (plan.totals[p] == 0) continue;
        const ivx_many_reg& reg = g_regs[plan.kernel[p]];
        if (!reg.fn) {
            ivx_set_error("ivx_many: kernel %d has no twin registered", plan.kernel[p]);
            rc = IVX_ERR_STATE;
        } else if (reg.fn(s, d + plan.off_args[p], reinterpret_cast<const uint32_t*>(d + plan.off_ends[p]), plan.first[p + 1] - plan.first[p], plan.totals[p]) != 0) {
            ivx_set_error("ivx_many: launch of twin %d failed", plan.kernel[p]);
            rc = IVX_ERR_HIP;
        } else {
            r->launches += 1;
        }
    }
    // the slot is free again when the twins that read its device block have run, not when the copy into it has landed
    if (hipEventRecord(r->ev[sl], s) != hipSuccess && rc == IVX_OK) {
        ivx_set_error("ivx_many: event record behind a flush failed");
        rc = IVX_ERR_HIP;
    }
    return rc;
}
// A flush that fails has dropped launches the caller believes to be on the stream (every wrapper in front of a stream operation ignores the
// break's status): the failure stays on the context until somebody who waits for results picks it up (ivx_many_error).
int flush_recorded(Recorder* r) {
    const int rc = flush_recorded_checked(r);
    if (rc != IVX_OK && r->ctx->many_error == IVX_OK) r->ctx->many_error = rc;
    return rc;
}
ClearOnExit::~ClearOnExit() {
    for (auto& c : r->chains) c.clear();
    r->arena.clear();
    r->n_entries = 0;
}

}  // namespace

void ivx_many_register(int kernel, ivx_many_launch_fn fn, uint32_t arg_bytes) {
    if (kernel < 0 || kernel >= IVX_MK_COUNT) return;
    g_regs[kernel].fn = fn;
    g_regs[kernel].arg_bytes = arg_bytes;
}

bool ivx_many_recording() { return t_rec && t_rec->on; }

// A call on an object of ANOTHER context than the one the batch is recorded for must not be recorded (its launches belong on that context's
// stream): what has been recorded goes out, and recording is off until the call returns. (Entry points that enqueue whole chains use this;
// every single capture checks its context as well, ivx_many_capture.)
ivx_many_other_context::ivx_many_other_context(const ivx_ctx* c) : suspended(false) {
    Recorder* r = t_rec;
    if (r && r->on && r->ctx != c) {
        (void)ivx_many_break();
        r->on = false;
        suspended = true;
    }
}
ivx_many_other_context::~ivx_many_other_context() {
    if (suspended && t_rec) t_rec->on = true;
}
uint64_t ivx_many_flush_count() { return t_rec ? t_rec->flushes : 0; }

int ivx_many_error(ivx_ctx* c, bool clear) {
    if (!c) return IVX_OK;
    const int rc = c->many_error;
    if (clear) c->many_error = IVX_OK;
    return rc;
}

void ivx_many_release(ivx_ctx* c) {
    if (!c || !c->many_recorder) return;
    Recorder* r = static_cast<Recorder*>(c->many_recorder);
    if (t_rec == r) t_rec = nullptr;
    OnDevice dev_{c->device};
    for (int sl = 0; sl < RING; ++sl) {
        if (r->ev[sl]) {
            if (r->busy[sl]) (void)hipEventSynchronize(r->ev[sl]);
            (void)hipEventDestroy(r->ev[sl]);
        }
        if (r->pinned[sl]) (void)hipHostFree(r->pinned[sl]);
        if (r->dev[sl]) (void)hipFree(r->dev[sl]);
    }
    delete r;
    c->many_recorder = nullptr;
}

static bool capture_range(const ivx_ctx* c, const void* owner, int kernel, void* d_dst, const void* h_src, size_t bytes, uint32_t fill = 0u) {
    Recorder* r = t_rec;
    if (!r || !r->on || (bytes & 3u) || bytes == 0 || bytes > (64u << 20)) return false;
    RangeArgs a;
    a.dst = static_cast<uint32_t*>(d_dst);
    a.src = nullptr;
    a.words = bytes / 4;
    a.fill = fill, a.pad = 0;
    const uint32_t blocks = (uint32_t)std::min<size_t>((a.words + 1023) / 1024, 64);
    if (!ivx_many_capture(c, owner, kernel, blocks, &a, (uint32_t)sizeof(a))) return false;
    Entry& e = r->chains[r->cur].back();
    e.payload_off = e.payload_bytes = 0;
    if (h_src) {
        e.payload_off = (uint32_t)r->arena.size();
        e.payload_bytes = (uint32_t)bytes;
        r->arena.resize(r->arena.size() + ((bytes + 15) & ~(size_t)15));
        memcpy(r->arena.data() + e.payload_off, h_src, bytes);
    }
    return true;
}
bool ivx_many_zero(const ivx_ctx* c, const void* owner, void* d_ptr, size_t bytes) { return capture_range(c, owner, IVX_MK_ZERO, d_ptr, nullptr, bytes); }
bool ivx_many_fill(const ivx_ctx* c, const void* owner, void* d_ptr, uint32_t word, size_t bytes) {
    return capture_range(c, owner, IVX_MK_ZERO, d_ptr, nullptr, bytes, word);
}
bool ivx_many_upload(const ivx_ctx* c, const void* owner, void* d_dst, const void* h_src, size_t bytes) {
    return capture_range(c, owner, IVX_MK_UPLOAD, d_dst, h_src, bytes);
}

bool ivx_many_capture(const ivx_ctx* c, const void* owner, int kernel, uint32_t blocks, const void* args, uint32_t arg_bytes) {
    Recorder* r = t_rec;
    if (!r || !r->on) return false;
    if (r->ctx != c) {  // an object of another context inside the bracket: its launch belongs on its own stream (and device), unrecorded
        (void)ivx_many_break();
        return false;
    }
    if (g_regs[kernel].arg_bytes != arg_bytes || !g_regs[kernel].fn) return false;  // (no twin: the caller launches — behind a break)
    if (!r->explicit_cur && (r->cur >= r->owners.size() || r->owners[r->cur] != owner)) {
        // a batch recorded through the bare bracket: the captures of an object go to that object's chain (chains in order of first appearance)
        size_t i = 0;
        while (i < r->owners.size() && r->owners[i] != owner) ++i;
        if (i == r->owners.size()) r->owners.push_back(owner);
        r->cur = (uint32_t)i;
    }
    if (r->cur >= r->chains.size()) r->chains.resize(r->cur + 1u);
    r->recorded += 1;
    Entry e;
    e.kernel = kernel;
    e.blocks = blocks;
    e.bytes = arg_bytes;
    e.payload_off = e.payload_bytes = 0;
    e.off = (uint32_t)r->arena.size();
    r->arena.resize(r->arena.size() + ((arg_bytes + 15u) & ~15u));
    memcpy(r->arena.data() + e.off, args, arg_bytes);
    r->chains[r->cur].push_back(e);
    r->n_entries += 1;
    return true;
}

int ivx_many_break() {
    Recorder* r = t_rec;
    if (!r || !r->on || r->n_entries == 0) return IVX_OK;
    // developer aid (IVX_MANY_TRACE=3): who cuts a batch short — a stream operation without a twin in the middle of a recorded phase
    static const bool where = getenv("IVX_MANY_TRACE") && atoi(getenv("IVX_MANY_TRACE")) == 3;
    if (where) {
        void* frames[12];
        const int n = backtrace(frames, 12);
        fprintf(stderr, "[ivx many] break with %u recorded launches, from:\n", (unsigned)r->n_entries);
        backtrace_symbols_fd(frames, n, 2);
    }
    return flush_recorded(r);
}

void ivx_many_object(uint32_t i) {
    if (!t_rec) return;
    t_rec->cur = i;
    t_rec->explicit_cur = true;
}

extern "C" {

int ivx_many_begin(ivx_ctx* c) {
    IVX_REQUIRE(c, IVX_ERR_INVALID, "ivx_many_begin: null context");
    if (t_rec) {
        // A bracket this thread opened and never closed (a caller that failed between begin and flush — a Python exception, an early return):
        // refusing every later bracket for good would take ivx_split_off_all and ivx_copy_polyhedra down with it. The stale bracket is closed
        // first — what it recorded goes out, nothing is dropped silently, its failure (if any) stays in the context's sticky error.
        Recorder* stale = t_rec;
        stale->on = true;
        (void)flush_recorded(stale);
        stale->on = false;
        t_rec = nullptr;
    }
    if (!c->many_recorder) c->many_recorder = new (std::nothrow) Recorder();
    Recorder* r = static_cast<Recorder*>(c->many_recorder);
    IVX_REQUIRE(r, IVX_ERR_CAPACITY, "ivx_many_begin: out of host memory");
    IVX_REQUIRE(!r->on, IVX_ERR_STATE, "ivx_many_begin: a batch is being recorded on this context already (another thread's)");
    r->ctx = c;
    r->on = true;
    r->cur = 0;
    r->explicit_cur = false;
    r->owners.clear();
    t_rec = r;
    return IVX_OK;
}

// launches recorded / merged launches issued / flushes made by this context's recorder so far (tests: do hand-recorded batches merge?)
int ivx_many_stats(ivx_ctx* c, uint64_t out[3]) {
    IVX_REQUIRE(c && out, IVX_ERR_INVALID, "ivx_many_stats: null argument");
    const Recorder* r = static_cast<const Recorder*>(c->many_recorder);
    out[0] = r ? r->recorded : 0;
    out[1] = r ? r->launches : 0;
    out[2] = r ? r->flushes : 0;
    return IVX_OK;
}

int ivx_many_flush(ivx_ctx* c) {
    IVX_REQUIRE(c && t_rec && t_rec->ctx == c, IVX_ERR_STATE, "ivx_many_flush: no batch is being recorded on this context");
    Recorder* r = t_rec;
    r->on = true;  // (a bracket suspended by a call on another context that never returned cannot be: the guard's destructor has run)
    int rc = flush_recorded(r);
    r->on = false;
    t_rec = nullptr;
    // a break inside the bracket that failed has dropped launches too: the bracket as a whole failed (reported once, here)
    const int sticky = ivx_many_error(c, true);
    if (rc == IVX_OK) rc = sticky;
    return rc;
}

}  // extern "C"
