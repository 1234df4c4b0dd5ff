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
};
__global__ __launch_bounds__(256) void k_range_many(const RangeArgs* __restrict__ argv, const uint32_t* __restrict__ block_end, uint32_t n) {
    const uint32_t i = ivx_many_find(block_end, n, blockIdx.x);
    const uint32_t b0 = i ? block_end[i - 1u] : 0u, nb = block_end[i] - b0, bid = blockIdx.x - b0;
    const RangeArgs a = argv[i];
    for (unsigned long long w = (unsigned long long)bid * 256u + threadIdx.x; w < a.words; w += (unsigned long long)nb * 256u) a.dst[w] = a.src ? a.src[w] : 0u;
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
    uint64_t flushes = 0;
};
thread_local Recorder* t_rec = nullptr;

// whatever happens to a flush, what was recorded is gone afterwards: a failed flush must not leave entries for the next batch to issue
struct ClearOnExit {
    Recorder* r;
    ~ClearOnExit();
};

int flush_recorded(Recorder* r) {
    if (r->n_entries == 0) return IVX_OK;
    ClearOnExit clear_{r};
    r->flushes += 1;
    hipStream_t s = r->ctx->stream;
    // plan the merged launches: (kernel, members...) in issue order
    struct Launch {
        int kernel;
        std::vector<std::pair<uint32_t, uint32_t>> members;  // (object, entry index)
    };
    std::vector<Launch> plan;
    const size_t n_obj = r->chains.size();
    r->cursor.assign(n_obj, 0u);
    size_t left = r->n_entries;
    while (left) {
        int k = -1;
        for (size_t i = 0; i < n_obj && k < 0; ++i)
            if (r->cursor[i] < r->chains[i].size()) k = r->chains[i][r->cursor[i]].kernel;
        Launch L;
        L.kernel = k;
        for (size_t i = 0; i < n_obj; ++i) {
            // (an object's consecutive entries of this kernel merge as well)
            while (r->cursor[i] < r->chains[i].size() && r->chains[i][r->cursor[i]].kernel == k) {
                L.members.emplace_back((uint32_t)i, r->cursor[i]);
                r->cursor[i] += 1u;
                left -= 1;
                if (k != IVX_MK_ZERO && k != IVX_MK_UPLOAD) break;  // (range operations of one object may all go together; kernels of one object stay in order)
            }
        }
        plan.push_back(std::move(L));
    }
    // lay out argument blocks and running block counts of every launch in one staging block (+ the words of the recorded uploads)
    size_t bytes = 0;
    std::vector<size_t> off_args(plan.size()), off_ends(plan.size());
    std::vector<std::vector<size_t>> off_payload(plan.size());
    for (size_t p = 0; p < plan.size(); ++p) {
        if (plan[p].kernel != IVX_MK_UPLOAD) continue;
        off_payload[p].resize(plan[p].members.size());
        for (size_t m = 0; m < plan[p].members.size(); ++m) {
            const Entry& e = r->chains[plan[p].members[m].first][plan[p].members[m].second];
            off_payload[p][m] = bytes;
            bytes += (e.payload_bytes + 15u) & ~15u;
        }
    }
    for (size_t p = 0; p < plan.size(); ++p) {
        const uint32_t ab = g_regs[plan[p].kernel].arg_bytes;
        off_args[p] = bytes;
        bytes += (size_t)ab * plan[p].members.size();
        bytes = (bytes + 15) & ~(size_t)15;
        off_ends[p] = bytes;
        bytes += 4 * plan[p].members.size();
        bytes = (bytes + 15) & ~(size_t)15;
    }
    const int sl = r->slot;
    r->slot = (r->slot + 1) % RING;
    if (!r->ev[sl]) IVX_HIP_CHECK(hipEventCreateWithFlags(&r->ev[sl], hipEventDisableTiming));
    if (r->busy[sl]) {
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
    std::vector<uint32_t> totals(plan.size());
    for (size_t p = 0; p < plan.size(); ++p) {
        const uint32_t ab = g_regs[plan[p].kernel].arg_bytes;
        uint32_t run = 0;
        uint32_t* ends = reinterpret_cast<uint32_t*>(h + off_ends[p]);
        for (size_t m = 0; m < plan[p].members.size(); ++m) {
            const Entry& e = r->chains[plan[p].members[m].first][plan[p].members[m].second];
            memcpy(h + off_args[p] + m * ab, r->arena.data() + e.off, ab);
            if (plan[p].kernel == IVX_MK_UPLOAD) {  // the words, and where the twin finds them on the device
                memcpy(h + off_payload[p][m], r->arena.data() + e.payload_off, e.payload_bytes);
                RangeArgs* ra = reinterpret_cast<RangeArgs*>(h + off_args[p] + m * ab);
                ra->src = reinterpret_cast<const uint32_t*>(static_cast<const unsigned char*>(r->dev[sl]) + off_payload[p][m]);
            }
            run += e.blocks;
            ends[m] = run;
        }
        totals[p] = run;
    }
    IVX_HIP_CHECK(hipMemcpyAsync(r->dev[sl], h, bytes, hipMemcpyHostToDevice, s));
    IVX_HIP_CHECK(hipEventRecord(r->ev[sl], s));
    r->busy[sl] = true;
    const unsigned char* d = static_cast<const unsigned char*>(r->dev[sl]);
    for (size_t p = 0; p < plan.size(); ++p) {
        if (totals[p] == 0) continue;
        const ivx_many_reg& reg = g_regs[plan[p].kernel];
        IVX_REQUIRE(reg.fn, IVX_ERR_STATE, "ivx_many: kernel %d has no twin registered", plan[p].kernel);
        const int rc = reg.fn(s, d + off_args[p], reinterpret_cast<const uint32_t*>(d + off_ends[p]), (uint32_t)plan[p].members.size(), totals[p]);
        IVX_REQUIRE(rc == 0, IVX_ERR_HIP, "ivx_many: launch of twin %d failed", plan[p].kernel);
    }
    return IVX_OK;
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
// stream): what has been recorded goes out, and recording is off until the call returns.
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

static bool capture_range(int kernel, void* d_dst, const void* h_src, size_t bytes) {
    Recorder* r = t_rec;
    if (!r || !r->on || (bytes & 3u) || bytes == 0 || bytes > (64u << 20)) return false;
    RangeArgs a;
    a.dst = static_cast<uint32_t*>(d_dst);
    a.src = nullptr;
    a.words = bytes / 4;
    const uint32_t blocks = (uint32_t)std::min<size_t>((a.words + 1023) / 1024, 64);
    if (!ivx_many_capture(kernel, blocks, &a, (uint32_t)sizeof(a))) return false;
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
bool ivx_many_zero(void* d_ptr, size_t bytes) { return capture_range(IVX_MK_ZERO, d_ptr, nullptr, bytes); }
bool ivx_many_upload(void* d_dst, const void* h_src, size_t bytes) { return capture_range(IVX_MK_UPLOAD, d_dst, h_src, bytes); }

bool ivx_many_capture(int kernel, uint32_t blocks, const void* args, uint32_t arg_bytes) {
    Recorder* r = t_rec;
    if (!r || !r->on) return false;
    if (g_regs[kernel].arg_bytes != arg_bytes || !g_regs[kernel].fn) return false;  // (no twin: the caller launches — behind a break)
    if (r->cur >= r->chains.size()) r->chains.resize(r->cur + 1u);
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
    if (t_rec) t_rec->cur = i;
}

extern "C" {

int ivx_many_begin(ivx_ctx* c) {
    IVX_REQUIRE(c, IVX_ERR_INVALID, "ivx_many_begin: null context");
    if (!t_rec) t_rec = new (std::nothrow) Recorder();
    IVX_REQUIRE(t_rec, IVX_ERR_CAPACITY, "ivx_many_begin: out of host memory");
    IVX_REQUIRE(!t_rec->on, IVX_ERR_STATE, "ivx_many_begin: a batch is being recorded on this thread already");
    t_rec->ctx = c;
    t_rec->on = true;
    t_rec->cur = 0;
    return IVX_OK;
}

int ivx_many_flush(ivx_ctx* c) {
    IVX_REQUIRE(c && t_rec && t_rec->on && t_rec->ctx == c, IVX_ERR_STATE, "ivx_many_flush: no batch is being recorded on this context");
    const int rc = flush_recorded(t_rec);
    t_rec->on = false;
    return rc;
}

}  // extern "C"
