// group_api.hip -- several GPUs behind the C ABI (BASELINE.json north_star: "the batch dimension shards
// trivially across 8 GPUs with an RCCL all-gather over xGMI only to collect per-instance optimal thrusts/costs for selection").
//
// ONE process, n devices: a brov_group owns one brov_solver per device (a contiguous shard of the batch, the first total % n
// shards one instance larger -- the partition of bluerov2_amd/distributed.py), one non-blocking stream per device and one RCCL
// communicator per device (ncclCommInitAll).  No communication during the solve.  brov_group_gather issues ONE ncclAllGather per
// device inside ncclGroupStart / ncclGroupEnd, on the devices' own streams, behind the solve:
//     BROV_GATHER_RECORDS   the 104-byte result records {u0, cost, kkt, status, qp_iter, thrusts} of every instance, to every device
//     BROV_GATHER_PACKED    one packed (cost, global index) pair per device (16 B): the local arg-min, for callers that need only
//                           the winner (SURVEY.md 8e's alternative)
// and brov_group_select_best finishes with the global arg-min of cost over the successful instances (BASELINE configs[3],
// "best-trajectory select").  This is the route for the reference's C++ callers (bluerov2_dob.cpp:270-451 runs in one process);
// bluerov2_amd/distributed.py (one process PER GPU on torch.distributed) stays as the second route.
//
// RCCL is resolved at run time (dlopen of librccl.so.1, the copy PyTorch may already have mapped): a process that never creates a
// group never loads it, and libbluerov2_nmpc.so has neither a link-time nor a build-time dependency on it (the few RCCL types this
// file needs are declared below, with the values of rccl.h 2.x).
//
// Second collective, BROV_COLLECTIVE_COPY (brov_group_create_ex / brov_group_create_rank_ex): the all-gather as device-to-device copies
// between the ranks' buffers -- every rank publishes its contribution behind an event on its own stream, every rank pulls the others'
// into its own `gathered` / `pairs` array on its own stream.  No RCCL involved, devices may repeat: W ranks on ONE GPU run exactly the
// bookkeeping of W GPUs (shard bounds by global rank, padded staging, rank-major gathered layout, packed pairs, select over W x slots,
// mailbox), which is how the 1-GPU test box executes this file with W > 1 (tests/test_gpu_group_loopback.py).  The ranks of a copy
// group must live in one process (they meet in a process-wide table keyed by the 128-byte id).
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

// ---- what this file uses of rccl.h (resolved by dlsym; values as in RCCL 2.x) ----
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclUint8 = 1, ncclDouble = 8 } ncclDataType_t;

#include "../../include/bluerov2_nmpc.h"

static thread_local std::string g_gerr;
extern "C" const char* brov_group_last_error(void) { return g_gerr.c_str(); }

namespace {

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*GetVersion)(int*) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string why;
};

Rccl& rccl() {
    static Rccl r;
    if (r.handle || !r.why.empty()) return r;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        r.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (r.handle) break;
    }
    if (!r.handle) {
        const char* e = dlerror();   // once: the call clears the message
        r.why = std::string("librccl.so.1 could not be loaded: ") + (e ? e : "?");
        return r;
    }
#define SYM(field, name) r.field = (decltype(r.field))dlsym(r.handle, name); if (!r.field) { r.why = std::string("RCCL symbol missing: ") + name; r.handle = nullptr; return r; }
    SYM(CommInitAll, "ncclCommInitAll") SYM(CommInitRank, "ncclCommInitRank") SYM(GetUniqueId, "ncclGetUniqueId") SYM(CommDestroy, "ncclCommDestroy") SYM(AllGather, "ncclAllGather")
    SYM(GroupStart, "ncclGroupStart") SYM(GroupEnd, "ncclGroupEnd") SYM(GetVersion, "ncclGetVersion") SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
    return r;
}

// ---- BROV_COLLECTIVE_COPY: where the ranks of one group meet (one table entry per 128-byte id; the brov_group objects that hold the
// ranks share it).  A rank publishes what it contributes (pointer + an event recorded on its stream behind the data) and counts the
// gathers it has entered / finished pulling; the other ranks wait for those counters on the host and for the events on their streams.
struct CopyPeer {
    bool present = false;
    int dev = -1;
    const brov_result* src_rec = nullptr;   // contribution to a BROV_GATHER_RECORDS gather: [Bmax] records (the solver's own array, or the padded staging copy)
    const double* src_pair = nullptr;       // ... to a BROV_GATHER_PACKED gather: (cost, global index)
    hipEvent_t ready = nullptr;             // recorded on the rank's stream behind its contribution
    hipEvent_t pulled = nullptr;            // ... behind its copies out of the other ranks' buffers
    long entered = 0, done = 0;             // gathers this rank has entered / has issued all pulls of
    int mode = -1;
};
struct CopyComm {
    std::mutex m;
    std::condition_variable cv;
    std::vector<CopyPeer> peer;             // by GLOBAL rank
    int joined = 0;
};
std::mutex g_copy_m;
std::map<std::string, std::shared_ptr<CopyComm>> g_copy;
std::atomic<int> kCopyWaitSeconds{60};       // a rank that does not show up within this time is reported, not waited for forever

// the calling thread's current device is the caller's business: every entry point that switches devices puts it back
struct DeviceKeeper {
    int d = -1;
    DeviceKeeper() { if (hipGetDevice(&d) != hipSuccess) { d = -1; (void)hipGetLastError(); } }
    ~DeviceKeeper() { if (d >= 0) (void)hipSetDevice(d); }
};

}  // namespace

constexpr int kSelBlocks = 64;   // blocks of group_select_kernel
struct GroupMail { brov_result rec; int32_t slot; int32_t seq; int32_t owner; int32_t pad; };   // what the select kernels deliver to the host (pinned), seq last

struct brov_group {
    // n = devices of THIS process; W = ranks of the whole group (= n for the one-process form; one process per GPU: n = 1, W = world),
    // r0 = global rank of local device 0.  lo / cnt are indexed by GLOBAL rank; everything else by local device.
    int n = 0, W = 0, r0 = 0, total = 0, Bmax = 0;
    bool even = true;                     // all shards equally large: the records are gathered straight from the solvers' own arrays
    std::vector<int> dev, lo, cnt;
    std::vector<brov_solver*> sol;
    std::vector<hipStream_t> st;
    std::vector<ncclComm_t> comm;
    std::vector<brov_result*> stage;      // [Bmax] per device (uneven shards only): the shard's records + never-selectable padding
    std::vector<brov_result*> gathered;   // [n * Bmax] per device
    std::vector<double*> pair, pairs;     // [2] local (cost, global index) and [n * 2] gathered, per device
    std::vector<int*> best;               // [2] per device: arg-min scratch (winning slot, ticket of the select kernel)
    double* sel_cost = nullptr;           // device 0: partial results of the select kernel's blocks, the winning record
    int* sel_idx = nullptr;
    brov_result* sel_rec = nullptr;
    GroupMail* mail = nullptr;            // pinned host mailbox the select kernels deliver into (device 0)
    std::vector<GroupMail*> pmail;        // per device (pinned): the local winner's record, written by the pack kernel
    int32_t mail_seq = 0;
    std::vector<hipEvent_t> ev;           // 4 per device: solve start / end = gather start / gather end / select end
    int last_mode = -1;
    int collective = BROV_COLLECTIVE_RCCL;
    std::shared_ptr<CopyComm> cc;         // BROV_COLLECTIVE_COPY: the meeting point of the group's ranks
    std::string cc_key;
    long round = 0;                       // gathers entered by this process's ranks
    bool timing = true;
    bool ev_sel = false;                  // the select-end event of the current step has been recorded
    bool ev_solve = false, ev_gather = false;   // ... the solve's pair, the gather's end
};

#define GHIP(call)                                                                          \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess) {                                                             \
            g_gerr = std::string(#call) + ": " + hipGetErrorString(e_);                     \
            (void)hipGetLastError();   /* reported here: do not leave it behind as the thread's "last error" for an unrelated call */ \
            return (e_ == hipErrorNoDevice || e_ == hipErrorInvalidDevice) ? BROV_ERR_NO_DEVICE : BROV_ERR_HIP; \
        }                                                                                   \
    } while (0)
#define GNCCL(call)                                                                         \
    do {                                                                                    \
        ncclResult_t r_ = (call);                                                           \
        if (r_ != ncclSuccess) { g_gerr = std::string(#call) + ": " + rccl().GetErrorString(r_); return BROV_ERR_HIP; } \
    } while (0)

// arg-min of cost over the successful records of a (padded) record array: slot -> (rank = slot / Bmax, i = slot % Bmax), valid while
// i < cnt[rank]; ties go to the lowest slot = lowest global index.  Up to kSelBlocks blocks scan the array (65 536 records of 104 B at
// BASELINE configs[3]: one block needs ~0.1 ms for them), the block that draws the last ticket reduces the partial results, and -- so
// that the host needs no copy command and no second round trip for the winner -- writes the winning slot AND its record into a pinned
// host mailbox, the sequence word last.  out[0] = winning slot or -1, best_dev = the record on the device.
__device__ __forceinline__ bool sel_better(double c2, int i2, double c1, int i1) { return i2 >= 0 && (i1 < 0 || c2 < c1 || (c2 == c1 && i2 < i1)); }
__global__ __launch_bounds__(256) void group_select_kernel(const brov_result* __restrict__ rec, int slots, int* __restrict__ out, double* pc, int* pi,
                                                             unsigned* ticket, brov_result* best_dev, GroupMail* mail, int seq) {
    __shared__ double sc[256];
    __shared__ int si[256];
    __shared__ bool last;
    double best = 1e300;
    int bi = -1;
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < slots; k += gridDim.x * blockDim.x) {
        const double c = rec[k].cost;
        if (rec[k].status == BROV_STATUS_SUCCESS && c == c && (bi < 0 || c < best)) { best = c; bi = k; }
    }
    auto block_reduce = [&]() {
        sc[threadIdx.x] = best; si[threadIdx.x] = bi;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o && sel_better(sc[threadIdx.x + o], si[threadIdx.x + o], sc[threadIdx.x], si[threadIdx.x])) {
                sc[threadIdx.x] = sc[threadIdx.x + o]; si[threadIdx.x] = si[threadIdx.x + o];
            }
            __syncthreads();
        }
    };
    block_reduce();
    if (threadIdx.x == 0) {
        pc[blockIdx.x] = sc[0]; pi[blockIdx.x] = si[0];
        __threadfence();
        last = atomicAdd(ticket, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (!last) return;
    __threadfence();
    best = 1e300; bi = -1;
    if (threadIdx.x < gridDim.x) { best = ((volatile double*)pc)[threadIdx.x]; bi = ((volatile int*)pi)[threadIdx.x]; }
    block_reduce();
    const int slot = si[0];
    if (slot >= 0 && threadIdx.x < sizeof(brov_result) / 8) {
        const double v = ((const double*)(rec + slot))[threadIdx.x];
        ((double*)best_dev)[threadIdx.x] = v;
        if (mail) ((double*)&mail->rec)[threadIdx.x] = v;
    }
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        out[0] = slot;
        *ticket = 0;   // the next launch on this stream
        if (mail) {
            mail->slot = slot;
            __threadfence_system();
            *(volatile int32_t*)&mail->seq = seq;
        }
    }
}
// the local arg-min as a packed pair: pair[0] = cost (+inf when no record qualifies), pair[1] = global index (exact in a double).  The
// winner's whole record goes into the device's own pinned mailbox on the way (system-scope fence ahead of the pair: whoever learns the
// pair through the all-gather finds the record in place).
__global__ void group_pack_kernel(const brov_result* __restrict__ rec, int n, int lo, double* __restrict__ pair, GroupMail* mail) {
    __shared__ double sc[256];
    __shared__ int si[256];
    double best = 1e300;
    int bi = -1;
    for (int k = threadIdx.x; k < n; k += blockDim.x) {
        const double c = rec[k].cost;
        if (rec[k].status == BROV_STATUS_SUCCESS && c == c && (bi < 0 || c < best)) { best = c; bi = k; }
    }
    sc[threadIdx.x] = best; si[threadIdx.x] = bi;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o && sel_better(sc[threadIdx.x + o], si[threadIdx.x + o], sc[threadIdx.x], si[threadIdx.x])) {
            sc[threadIdx.x] = sc[threadIdx.x + o]; si[threadIdx.x] = si[threadIdx.x + o];
        }
        __syncthreads();
    }
    const int w = si[0];
    if (w >= 0 && threadIdx.x < sizeof(brov_result) / 8) ((double*)&mail->rec)[threadIdx.x] = ((const double*)(rec + w))[threadIdx.x];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        mail->slot = w >= 0 ? lo + w : -1;
        __threadfence_system();
        pair[0] = w >= 0 ? sc[0] : __builtin_inf();
        pair[1] = w >= 0 ? (double)(lo + w) : -1.0;
    }
}
// the gathered pairs -> owner rank, global index and cost of the global winner, into the host mailbox (one wave; shards hold ascending
// index ranges, so the first minimal cost in rank order is the lowest index)
__global__ void group_pairs_kernel(const double* __restrict__ pairs, int W, GroupMail* mail, int seq) {
    if (threadIdx.x != 0) return;
    int owner = -1;
    for (int r = 0; r < W; r++)
        if (pairs[2 * r + 1] >= 0.0 && (owner < 0 || pairs[2 * r] < pairs[2 * owner])) owner = r;
    mail->owner = owner;
    mail->slot = owner >= 0 ? (int)pairs[2 * owner + 1] : -1;
    mail->rec.cost = owner >= 0 ? pairs[2 * owner] : 0.0;
    __threadfence_system();
    *(volatile int32_t*)&mail->seq = seq;
}

extern "C" {

int brov_group_rccl_version(int* version) {
    Rccl& r = rccl();
    if (!r.handle) { g_gerr = r.why; return BROV_ERR_HIP; }
    int v = 0;
    if (r.GetVersion(&v) != ncclSuccess) { g_gerr = "ncclGetVersion failed"; return BROV_ERR_HIP; }
    if (version) *version = v;
    return BROV_OK;
}

void brov_group_destroy(brov_group* g) {
    if (!g) return;
    DeviceKeeper keep;
    for (int d = 0; d < g->n; d++) {   // everything this process's ranks have enqueued is over before anything is taken apart
        if (d < (int)g->dev.size()) hipSetDevice(g->dev[d]);
        if (d < (int)g->st.size() && g->st[d]) hipStreamSynchronize(g->st[d]);
    }
    if (g->cc) {   // leave the copy collective's table (the other ranks must have finished the gathers they share with this one)
        {
            std::lock_guard<std::mutex> lk(g->cc->m);
            for (int d = 0; d < g->n; d++) {
                const int q = g->r0 + d;
                if (q >= (int)g->cc->peer.size() || !g->cc->peer[q].present || g->cc->peer[q].src_pair != (d < (int)g->pair.size() ? g->pair[d] : nullptr)) continue;
                CopyPeer& me = g->cc->peer[q];
                if (me.ready) hipEventDestroy(me.ready);
                if (me.pulled) hipEventDestroy(me.pulled);
                me = CopyPeer();
                g->cc->joined--;
            }
        }
        std::lock_guard<std::mutex> lk(g_copy_m);
        auto it = g_copy.find(g->cc_key);
        if (it != g_copy.end() && it->second == g->cc && g->cc->joined <= 0) g_copy.erase(it);
        g->cc.reset();
    }
    for (int d = 0; d < g->n; d++) {
        if (d < (int)g->dev.size()) hipSetDevice(g->dev[d]);
        if (d < (int)g->comm.size() && g->comm[d] && rccl().handle) rccl().CommDestroy(g->comm[d]);
        if (d < (int)g->sol.size()) brov_destroy(g->sol[d]);
        if (d < (int)g->stage.size() && g->stage[d]) hipFree(g->stage[d]);
        if (d < (int)g->gathered.size() && g->gathered[d]) hipFree(g->gathered[d]);
        if (d < (int)g->pair.size() && g->pair[d]) hipFree(g->pair[d]);
        if (d < (int)g->best.size() && g->best[d]) hipFree(g->best[d]);
        if (d < (int)g->pmail.size() && g->pmail[d]) hipHostFree(g->pmail[d]);
        if (d == 0) {
            if (g->sel_cost) hipFree(g->sel_cost);
            if (g->sel_idx) hipFree(g->sel_idx);
            if (g->sel_rec) hipFree(g->sel_rec);
            if (g->mail) hipHostFree(g->mail);
        }
        for (int k = 0; k < 4; k++)
            if (4 * d + k < (int)g->ev.size() && g->ev[4 * d + k]) hipEventDestroy(g->ev[4 * d + k]);
        if (d < (int)g->st.size() && g->st[d]) hipStreamDestroy(g->st[d]);
    }
    delete g;
}

// common part of the two creators: local devices dev[0..n), global ranks r0 .. r0 + n - 1 of W, instances per global rank in cnt[W]
static int group_build(brov_group** out, const int* devices, int n, int W, int r0, const std::vector<int>& cnt, const brov_opts* opts,
                       const ncclUniqueId* id, int collective) {
    if (collective != BROV_COLLECTIVE_RCCL && collective != BROV_COLLECTIVE_COPY) { g_gerr = "brov_group_create: unknown collective"; return BROV_ERR_ARG; }
    const bool copy = collective == BROV_COLLECTIVE_COPY;
    DeviceKeeper keep;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { g_gerr = "brov_group_create: no usable HIP device (this library has no CPU fallback)"; return BROV_ERR_NO_DEVICE; }
    for (int d = 0; d < n; d++) {
        if (devices[d] < 0 || devices[d] >= ndev) { g_gerr = "brov_group_create: device ordinal out of range"; return BROV_ERR_NO_DEVICE; }
        for (int e = 0; e < d && !copy; e++)
            if (devices[e] == devices[d]) { g_gerr = "brov_group_create: a device may appear once (one RCCL rank per GPU; BROV_COLLECTIVE_COPY lifts this)"; return BROV_ERR_ARG; }
    }
    Rccl* Rp = copy ? nullptr : &rccl();
    if (Rp && !Rp->handle) { g_gerr = "brov_group_create: " + Rp->why; return BROV_ERR_HIP; }
    brov_group* g = new brov_group();
    g->n = n; g->W = W; g->r0 = r0; g->collective = collective;
    g->dev.assign(devices, devices + n);
    g->cnt = cnt; g->lo.assign(W, 0);
    g->even = true;
    for (int r = 0; r < W; r++) {
        g->lo[r] = r ? g->lo[r - 1] + g->cnt[r - 1] : 0;
        if (g->cnt[r] > g->Bmax) g->Bmax = g->cnt[r];
        g->even = g->even && g->cnt[r] == g->cnt[0];
    }
    g->total = g->lo[W - 1] + g->cnt[W - 1];
    g->sol.assign(n, nullptr); g->st.assign(n, nullptr); g->comm.assign(n, nullptr);
    g->stage.assign(n, nullptr); g->gathered.assign(n, nullptr); g->pair.assign(n, nullptr); g->pairs.assign(n, nullptr);
    g->best.assign(n, nullptr); g->ev.assign(4 * (size_t)n, nullptr); g->pmail.assign(n, nullptr);
    auto fail = [&](int rc) { brov_group_destroy(g); return rc; };
    for (int d = 0; d < n; d++) {
        if (hipSetDevice(g->dev[d]) != hipSuccess) { g_gerr = "brov_group_create: hipSetDevice failed"; return fail(BROV_ERR_HIP); }
        const int rc = brov_create(&g->sol[d], g->dev[d], g->cnt[r0 + d], opts);
        if (rc != BROV_OK) { g_gerr = std::string("brov_group_create: shard ") + std::to_string(r0 + d) + ": " + brov_last_error(); return fail(rc); }
        bool ok = hipStreamCreateWithFlags(&g->st[d], hipStreamNonBlocking) == hipSuccess;
        ok = ok && hipMalloc((void**)&g->gathered[d], (size_t)W * g->Bmax * sizeof(brov_result)) == hipSuccess;
        ok = ok && hipMalloc((void**)&g->pair[d], (2 + 2 * (size_t)W) * sizeof(double)) == hipSuccess;
        ok = ok && hipMalloc((void**)&g->best[d], 2 * sizeof(int)) == hipSuccess;
        ok = ok && hipMemset(g->best[d], 0, 2 * sizeof(int)) == hipSuccess;
        ok = ok && hipHostMalloc((void**)&g->pmail[d], sizeof(GroupMail), hipHostMallocDefault) == hipSuccess;
        if (ok) std::memset(g->pmail[d], 0, sizeof(GroupMail));
        if (ok && d == 0) {   // the select runs on local device 0
            ok = hipMalloc((void**)&g->sel_cost, kSelBlocks * sizeof(double)) == hipSuccess;
            ok = ok && hipMalloc((void**)&g->sel_idx, kSelBlocks * sizeof(int)) == hipSuccess;
            ok = ok && hipMalloc((void**)&g->sel_rec, sizeof(brov_result)) == hipSuccess;
            ok = ok && hipHostMalloc((void**)&g->mail, sizeof(GroupMail), hipHostMallocDefault) == hipSuccess;
            if (ok) std::memset(g->mail, 0, sizeof(GroupMail));
        }
        if (ok && !g->even) {
            ok = hipMalloc((void**)&g->stage[d], (size_t)g->Bmax * sizeof(brov_result)) == hipSuccess;
            // padding records: status -1, cost NaN -- never selectable; only the first cnt slots are ever rewritten
            ok = ok && hipMemset(g->stage[d], 0xFF, (size_t)g->Bmax * sizeof(brov_result)) == hipSuccess;
        }
        for (int k = 0; k < 4 && ok; k++) ok = hipEventCreate(&g->ev[4 * d + k]) == hipSuccess;
        if (!ok) { g_gerr = "brov_group_create: allocation of the gather buffers failed"; return fail(BROV_ERR_ALLOC); }
        g->pairs[d] = g->pair[d] + 2;
    }
    if (copy) {   // join the table entry of this group's id (the one-process form has an id of its own)
        g->cc_key = id ? std::string(id->internal, sizeof(id->internal)) : "one-process group " + std::to_string((unsigned long long)(uintptr_t)g);
        {
            std::lock_guard<std::mutex> lk(g_copy_m);
            std::shared_ptr<CopyComm>& sp = g_copy[g->cc_key];
            if (!sp) { sp = std::make_shared<CopyComm>(); sp->peer.resize(W); }
            g->cc = sp;
        }
        std::unique_lock<std::mutex> lk(g->cc->m);
        if ((int)g->cc->peer.size() != W) { lk.unlock(); g_gerr = "brov_group_create_rank: the ranks of this id disagree about the world size"; return fail(BROV_ERR_ARG); }
        for (int d = 0; d < n; d++)
            if (g->cc->peer[r0 + d].present) { lk.unlock(); g_gerr = "brov_group_create_rank: rank " + std::to_string(r0 + d) + " of this id exists already"; return fail(BROV_ERR_ARG); }
        for (int d = 0; d < n; d++) {
            CopyPeer& me = g->cc->peer[r0 + d];
            me.dev = g->dev[d];
            me.src_rec = g->even ? brov_results_device(g->sol[d]) : g->stage[d];
            me.src_pair = g->pair[d];
            bool ok = hipSetDevice(g->dev[d]) == hipSuccess;
            ok = ok && hipEventCreateWithFlags(&me.ready, hipEventDisableTiming) == hipSuccess;
            ok = ok && hipEventCreateWithFlags(&me.pulled, hipEventDisableTiming) == hipSuccess;
            if (!ok) { lk.unlock(); g_gerr = "brov_group_create: events of the copy collective"; return fail(BROV_ERR_HIP); }
            me.present = true;
            g->cc->joined++;
        }
        g->cc->cv.notify_all();
    } else {
        Rccl& R = *Rp;
        ncclResult_t nr;
        if (id) {   // one process per GPU: this process is rank r0 of W
            if (hipSetDevice(g->dev[0]) != hipSuccess) { g_gerr = "brov_group_create_rank: hipSetDevice failed"; return fail(BROV_ERR_HIP); }
            nr = R.CommInitRank(&g->comm[0], W, *id, r0);
        } else {
            nr = R.CommInitAll(g->comm.data(), n, g->dev.data());
        }
        if (nr != ncclSuccess) { g_gerr = std::string("brov_group_create: RCCL communicator set-up: ") + R.GetErrorString(nr); return fail(BROV_ERR_HIP); }
    }
    *out = g;
    return BROV_OK;
}

int brov_group_create_ex(brov_group** out, const int* devices, int n, int total, const brov_opts* opts, int collective) {
    if (!out || !devices || !opts || n < 1 || n > 64 || total < n) { g_gerr = "brov_group_create: bad argument (1 <= n <= 64 devices, at least one instance each)"; return BROV_ERR_ARG; }
    std::vector<int> cnt(n);
    for (int d = 0; d < n; d++) cnt[d] = total / n + (d < total % n ? 1 : 0);
    return group_build(out, devices, n, n, 0, cnt, opts, nullptr, collective);
}
int brov_group_create(brov_group** out, const int* devices, int n, int total, const brov_opts* opts) {
    return brov_group_create_ex(out, devices, n, total, opts, BROV_COLLECTIVE_RCCL);
}

// ---- one process per GPU: the same group, each process holding ONE rank of it ----------------------------------------------------
int brov_group_unique_id(char id[128]) {
    Rccl& R = rccl();
    if (!R.handle) { g_gerr = R.why; return BROV_ERR_HIP; }
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    ncclUniqueId u;
    if (!id || R.GetUniqueId(&u) != ncclSuccess) { g_gerr = "ncclGetUniqueId failed"; return BROV_ERR_HIP; }
    std::memcpy(id, &u, 128);
    return BROV_OK;
}
int brov_group_create_rank_ex(brov_group** out, int device, int rank, int world, const char id[128], const int* counts, const brov_opts* opts, int collective) {
    if (!out || !id || !counts || !opts || world < 1 || rank < 0 || rank >= world) { g_gerr = "brov_group_create_rank: bad argument"; return BROV_ERR_ARG; }
    std::vector<int> cnt(counts, counts + world);
    for (int r = 0; r < world; r++)
        if (cnt[r] < 1) { g_gerr = "brov_group_create_rank: every rank needs at least one instance"; return BROV_ERR_ARG; }
    ncclUniqueId u;
    std::memcpy(&u, id, 128);
    return group_build(out, &device, 1, world, rank, cnt, opts, &u, collective);
}
int brov_group_create_rank(brov_group** out, int device, int rank, int world, const char id[128], const int* counts, const brov_opts* opts) {
    return brov_group_create_rank_ex(out, device, rank, world, id, counts, opts, BROV_COLLECTIVE_RCCL);
}
int brov_group_collective(const brov_group* g) { return g ? g->collective : -1; }
int brov_group_set_copy_wait_seconds(int seconds) { if (seconds < 1) return BROV_ERR_ARG; kCopyWaitSeconds = seconds; return BROV_OK; }

int brov_group_size(const brov_group* g) { return g ? g->n : 0; }          /* devices of this process */
int brov_group_world(const brov_group* g) { return g ? g->W : 0; }         /* ranks of the whole group */
int brov_group_first_rank(const brov_group* g) { return g ? g->r0 : 0; }
int brov_group_total(const brov_group* g) { return g ? g->total : 0; }
brov_solver* brov_group_solver(brov_group* g, int rank) { return (g && rank >= 0 && rank < g->n) ? g->sol[rank] : nullptr; }
void* brov_group_stream(brov_group* g, int rank) { return (g && rank >= 0 && rank < g->n) ? (void*)g->st[rank] : nullptr; }
int brov_group_shard(const brov_group* g, int rank, int* lo, int* hi) {
    if (!g || rank < 0 || rank >= g->W) return BROV_ERR_ARG;   /* GLOBAL rank */
    if (lo) *lo = g->lo[rank];
    if (hi) *hi = g->lo[rank] + g->cnt[rank];
    return BROV_OK;
}

// ---- whole-batch setters: global HOST arrays, sliced per shard -------------------------------------------------------------
int brov_group_set_x0_host(brov_group* g, const double* x0) {
    if (!g || !x0) return BROV_ERR_ARG;
    DeviceKeeper keep;
    for (int d = 0; d < g->n; d++)
        if (int rc = brov_set_x0_host(g->sol[d], x0 + (size_t)g->lo[g->r0 + d] * 12)) { g_gerr = brov_last_error(); return rc; }
    return BROV_OK;
}
int brov_group_set_params_host(brov_group* g, const double* p, int per_stage) {
    if (!g || !p) return BROV_ERR_ARG;
    DeviceKeeper keep;
    const size_t row = per_stage ? (size_t)(brov_horizon(g->sol[0]) + 1) * 16 : 16;
    for (int d = 0; d < g->n; d++)
        if (int rc = brov_set_params_host(g->sol[d], p + (size_t)g->lo[g->r0 + d] * row, per_stage)) { g_gerr = brov_last_error(); return rc; }
    return BROV_OK;
}
int brov_group_set_yref_host(brov_group* g, const double* yref, int shared) {
    if (!g || !yref) return BROV_ERR_ARG;
    DeviceKeeper keep;
    const size_t row = (size_t)(brov_horizon(g->sol[0]) + 1) * 16;
    for (int d = 0; d < g->n; d++)
        if (int rc = brov_set_yref_host(g->sol[d], shared ? yref : yref + (size_t)g->lo[g->r0 + d] * row, shared)) { g_gerr = brov_last_error(); return rc; }
    return BROV_OK;
}
int brov_group_set_candidate_params_host(brov_group* g, int kind, const double* p0, const double* p1, const double* phase) {
    if (!g || !p0 || !p1 || !phase) return BROV_ERR_ARG;
    DeviceKeeper keep;
    for (int d = 0; d < g->n; d++)
        if (int rc = brov_set_candidate_params_host(g->sol[d], kind, p0 + g->lo[g->r0 + d], p1 + g->lo[g->r0 + d], phase + g->lo[g->r0 + d])) { g_gerr = brov_last_error(); return rc; }
    return BROV_OK;
}
int brov_group_set_yref_candidates(brov_group* g, double t0, double dt) {   // one window kernel per device, on the device's stream
    if (!g) return BROV_ERR_ARG;
    DeviceKeeper keep;
    for (int d = 0; d < g->n; d++)
        if (int rc = brov_set_yref_candidates(g->sol[d], t0, dt, g->st[d])) { g_gerr = brov_last_error(); return rc; }
    return BROV_OK;
}

int brov_group_enable_timing(brov_group* g, int on) { if (!g) return BROV_ERR_ARG; g->timing = on != 0; return BROV_OK; }

// BROV_COLLECTIVE_COPY: before a rank rewrites what it contributed to the last gather (its records: the next solve; its staging copy /
// its pair: the next gather), every rank has pulled it -- waited for on the host until the pulls are enqueued, on the streams until
// they have run.  (RCCL needs nothing of the kind: its all-gather ends on all streams together.)
static int copy_fence(brov_group* g) {
    if (!g->cc || g->round == 0) return BROV_OK;
    CopyComm& c = *g->cc;
    {
        std::unique_lock<std::mutex> lk(c.m);
        int missing = -1;
        const bool ok = c.cv.wait_for(lk, std::chrono::seconds(kCopyWaitSeconds.load()), [&] {
            for (int r = 0; r < g->W; r++)
                if (!c.peer[r].present || c.peer[r].done < g->round) { missing = r; return false; }
            return true;
        });
        if (!ok) { g_gerr = "copy collective: rank " + std::to_string(missing) + " has not finished gather " + std::to_string(g->round); return BROV_ERR_HIP; }
    }
    for (int d = 0; d < g->n; d++) {
        GHIP(hipSetDevice(g->dev[d]));
        for (int r = 0; r < g->W; r++)
            if (r != g->r0 + d) GHIP(hipStreamWaitEvent(g->st[d], c.peer[r].pulled, 0));
    }
    return BROV_OK;
}

// the all-gather as copies (see the head of the file): publish, meet, pull
static int copy_gather(brov_group* g, int mode) {
    CopyComm& c = *g->cc;
    const size_t rec = sizeof(brov_result);
    const long round = ++g->round;
    for (int d = 0; d < g->n; d++) {
        GHIP(hipSetDevice(g->dev[d]));
        GHIP(hipEventRecord(c.peer[g->r0 + d].ready, g->st[d]));
    }
    {
        std::unique_lock<std::mutex> lk(c.m);
        for (int d = 0; d < g->n; d++) { c.peer[g->r0 + d].entered = round; c.peer[g->r0 + d].mode = mode; }
        c.cv.notify_all();
        // every rank has entered this gather: what a collective waits for on the device is waited for on the host here
        int missing = -1;
        const bool ok = c.cv.wait_for(lk, std::chrono::seconds(kCopyWaitSeconds.load()), [&] {
            for (int r = 0; r < g->W; r++)
                if (!c.peer[r].present || c.peer[r].entered < round) { missing = r; return false; }
            return true;
        });
        if (!ok) { g_gerr = "copy collective: rank " + std::to_string(missing) + " did not enter gather " + std::to_string(round); return BROV_ERR_HIP; }
        for (int r = 0; r < g->W; r++)
            if (c.peer[r].entered == round && c.peer[r].mode != mode) { g_gerr = "copy collective: the ranks disagree about the gather mode"; return BROV_ERR_ARG; }
    }
    for (int d = 0; d < g->n; d++) {
        GHIP(hipSetDevice(g->dev[d]));
        for (int r = 0; r < g->W; r++) {
            if (r != g->r0 + d) GHIP(hipStreamWaitEvent(g->st[d], c.peer[r].ready, 0));
            if (mode == BROV_GATHER_RECORDS)
                GHIP(hipMemcpyAsync(g->gathered[d] + (size_t)r * g->Bmax, c.peer[r].src_rec, (size_t)g->Bmax * rec, hipMemcpyDefault, g->st[d]));
            else
                GHIP(hipMemcpyAsync(g->pairs[d] + 2 * r, c.peer[r].src_pair, 2 * sizeof(double), hipMemcpyDefault, g->st[d]));
        }
        GHIP(hipEventRecord(c.peer[g->r0 + d].pulled, g->st[d]));
    }
    {
        std::lock_guard<std::mutex> lk(c.m);
        for (int d = 0; d < g->n; d++) c.peer[g->r0 + d].done = round;
        c.cv.notify_all();
    }
    return BROV_OK;
}

int brov_group_solve(brov_group* g) {
    if (!g) return BROV_ERR_ARG;
    DeviceKeeper keep;
    if (int rc = copy_fence(g)) return rc;
    for (int d = 0; d < g->n; d++) {
        GHIP(hipSetDevice(g->dev[d]));
        if (g->timing) GHIP(hipEventRecord(g->ev[4 * d + 0], g->st[d]));
        if (int rc = brov_solve(g->sol[d], g->st[d])) { g_gerr = brov_last_error(); return rc; }
        if (g->timing) GHIP(hipEventRecord(g->ev[4 * d + 1], g->st[d]));
    }
    g->last_mode = -1;
    g->ev_sel = false; g->ev_gather = false; g->ev_solve = g->timing;
    return BROV_OK;
}

int brov_group_gather(brov_group* g, int mode) {
    if (!g || (mode != BROV_GATHER_RECORDS && mode != BROV_GATHER_PACKED)) return BROV_ERR_ARG;
    DeviceKeeper keep;
    const size_t rec = sizeof(brov_result);
    if (int rc = copy_fence(g)) return rc;
    // what each device contributes, produced on its own stream behind the solve
    for (int d = 0; d < g->n; d++) {
        GHIP(hipSetDevice(g->dev[d]));
        // (timing: the solve's end event doubles as the gather's start)
        if (mode == BROV_GATHER_RECORDS) {
            if (!g->even) GHIP(hipMemcpyAsync(g->stage[d], brov_results_device(g->sol[d]), (size_t)g->cnt[g->r0 + d] * rec, hipMemcpyDeviceToDevice, g->st[d]));
        } else {
            hipLaunchKernelGGL(group_pack_kernel, dim3(1), dim3(256), 0, g->st[d], brov_results_device(g->sol[d]), g->cnt[g->r0 + d], g->lo[g->r0 + d], g->pair[d], g->pmail[d]);
            GHIP(hipGetLastError());
        }
    }
    if (g->cc) {
        if (int rc = copy_gather(g, mode)) return rc;
    } else {
        Rccl& R = rccl();
        GNCCL(R.GroupStart());
        for (int d = 0; d < g->n; d++) {
            ncclResult_t r;
            if (mode == BROV_GATHER_RECORDS) {
                const void* src = g->even ? (const void*)brov_results_device(g->sol[d]) : (const void*)g->stage[d];
                r = R.AllGather(src, g->gathered[d], (size_t)g->Bmax * rec, ncclUint8, g->comm[d], g->st[d]);
            } else {
                r = R.AllGather(g->pair[d], g->pairs[d], 2, ncclDouble, g->comm[d], g->st[d]);
            }
            if (r != ncclSuccess) { R.GroupEnd(); g_gerr = std::string("ncclAllGather: ") + R.GetErrorString(r); return BROV_ERR_HIP; }
        }
        GNCCL(R.GroupEnd());
    }
    if (g->timing)
        for (int d = 0; d < g->n; d++) { GHIP(hipSetDevice(g->dev[d])); GHIP(hipEventRecord(g->ev[4 * d + 2], g->st[d])); }
    g->ev_gather = g->timing && g->ev_solve;
    g->last_mode = mode;
    return BROV_OK;
}

int brov_group_synchronize(brov_group* g) {
    if (!g) return BROV_ERR_ARG;
    DeviceKeeper keep;
    for (int d = 0; d < g->n; d++) { GHIP(hipSetDevice(g->dev[d])); GHIP(hipStreamSynchronize(g->st[d])); }
    return BROV_OK;
}

static int slot_to_global(const brov_group* g, int slot) {
    if (slot < 0) return -1;
    const int r = slot / g->Bmax, i = slot % g->Bmax;
    return (r < g->W && i < g->cnt[r]) ? g->lo[r] + i : -1;
}

// poll the sequence word of device 0's mailbox; the stream is queried now and then, so that a launch that ended without delivering (a
// device fault) turns into an error instead of a hang
static int mail_wait(brov_group* g, int32_t seq) {
    volatile int32_t* pf = (volatile int32_t*)&g->mail->seq;
    for (unsigned long spin = 1; *pf != seq; spin++) {
        if ((spin & 0x3ff) == 0) {
            const hipError_t q = hipStreamQuery(g->st[0]);
            if (q == hipSuccess) {   // the launch is over: everything it wrote is visible
                if (*pf != seq) { g_gerr = "brov_group_select_best: the select kernel ended without delivering"; return BROV_ERR_HIP; }
            } else if (q != hipErrorNotReady) {
                g_gerr = std::string("brov_group_select_best: ") + hipGetErrorString(q);
                (void)hipGetLastError();
                return BROV_ERR_HIP;
            }
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    return BROV_OK;
}

int brov_group_select_best(brov_group* g, int* best_index, brov_result* best) {
    if (!g || !best_index) return BROV_ERR_ARG;
    if (g->last_mode < 0) { g_gerr = "brov_group_select_best: call brov_group_gather first"; return BROV_ERR_ARG; }
    DeviceKeeper keep;
    *best_index = -1;
    if (g->last_mode == BROV_GATHER_RECORDS) {
        // every device holds all records: device 0 selects (any would do).  The kernel delivers the winning slot and its record into
        // the pinned mailbox and the host polls the sequence word: no copy command, no stream synchronisation on the way back (the
        // other devices only finish their gather -- their streams order whatever is enqueued next behind it).
        GHIP(hipSetDevice(g->dev[0]));
        const int slots = g->W * g->Bmax;
        int nb = (slots + 1023) / 1024;
        nb = nb < 1 ? 1 : (nb > kSelBlocks ? kSelBlocks : nb);
        g->mail_seq = g->mail_seq == 0x7fffffff ? 1 : g->mail_seq + 1;
        const int32_t seq = g->mail_seq;
        hipLaunchKernelGGL(group_select_kernel, dim3(nb), dim3(256), 0, g->st[0], g->gathered[0], slots, g->best[0], g->sel_cost, g->sel_idx,
                           (unsigned*)(g->best[0] + 1), g->sel_rec, g->mail, seq);
        GHIP(hipGetLastError());
        if (g->timing) { GHIP(hipEventRecord(g->ev[3], g->st[0])); g->ev_sel = true; }
        if (int rc = mail_wait(g, seq)) return rc;
        const int slot = g->mail->slot;
        *best_index = slot_to_global(g, slot);
        if (best && slot >= 0) std::memcpy(best, &g->mail->rec, sizeof(brov_result));
    } else {
        // the gathered pairs are reduced on device 0 and the winner's (owner rank, global index, cost) comes back through the mailbox;
        // its whole record is already in the owner's own mailbox (group_pack_kernel) when the owner is a device of this process
        GHIP(hipSetDevice(g->dev[0]));
        g->mail_seq = g->mail_seq == 0x7fffffff ? 1 : g->mail_seq + 1;
        const int32_t seq = g->mail_seq;
        hipLaunchKernelGGL(group_pairs_kernel, dim3(1), dim3(64), 0, g->st[0], g->pairs[0], g->W, g->mail, seq);
        GHIP(hipGetLastError());
        if (g->timing) { GHIP(hipEventRecord(g->ev[3], g->st[0])); g->ev_sel = true; }
        if (int rc = mail_wait(g, seq)) return rc;
        const int owner = g->mail->owner;
        if (owner >= 0) {
            *best_index = g->mail->slot;
            if (best) {
                std::memset(best, 0, sizeof(*best));
                const int d = owner - g->r0;
                if (d >= 0 && d < g->n) {   // the winner lives in this process: its whole record
                    std::memcpy(best, &g->pmail[d]->rec, sizeof(brov_result));
                } else {                    // ... in another process (one process per GPU): the pair carries its cost only
                    best->cost = g->mail->rec.cost; best->status = BROV_STATUS_SUCCESS;
                }
            }
        }
    }
    return BROV_OK;
}

int brov_group_get_results_host(brov_group* g, brov_result* res) {
    if (!g || !res) return BROV_ERR_ARG;
    if (g->last_mode != BROV_GATHER_RECORDS) { g_gerr = "brov_group_get_results_host: needs a BROV_GATHER_RECORDS gather"; return BROV_ERR_ARG; }
    if (int rc = brov_group_synchronize(g)) return rc;
    DeviceKeeper keep;
    GHIP(hipSetDevice(g->dev[0]));
    for (int r = 0; r < g->W; r++)   // strip the padding slots of uneven shards
        GHIP(hipMemcpy(res + g->lo[r], g->gathered[0] + (size_t)r * g->Bmax, (size_t)g->cnt[r] * sizeof(brov_result), hipMemcpyDeviceToHost));
    return BROV_OK;
}
const brov_result* brov_group_gathered_device(const brov_group* g, int rank) { return (g && rank >= 0 && rank < g->n) ? g->gathered[rank] : nullptr; }
int brov_group_slots_per_rank(const brov_group* g) { return g ? g->Bmax : 0; }

int brov_group_last_seconds(brov_group* g, double* solve, double* gather, double* select) {
    if (!g || !g->timing || !g->ev_solve) { g_gerr = "brov_group_last_seconds: no timed solve yet (timing must be on before brov_group_solve)"; return BROV_ERR_ARG; }
    double ts = 0, tg = 0, tsel = 0;
    DeviceKeeper keep;
    for (int d = 0; d < g->n; d++) {
        GHIP(hipSetDevice(g->dev[d]));
        float a = 0, b = 0;
        GHIP(hipEventSynchronize(g->ev[4 * d + 1]));
        GHIP(hipEventElapsedTime(&a, g->ev[4 * d + 0], g->ev[4 * d + 1]));
        if (g->last_mode >= 0 && g->ev_gather) { GHIP(hipEventSynchronize(g->ev[4 * d + 2])); GHIP(hipEventElapsedTime(&b, g->ev[4 * d + 1], g->ev[4 * d + 2])); }
        if (a * 1e-3 > ts) ts = a * 1e-3;
        if (b * 1e-3 > tg) tg = b * 1e-3;
    }
    if (g->last_mode >= 0 && g->ev_sel && g->ev_gather) {   // (an event that was never recorded makes hipEventElapsedTime fail, and the failure would stay
        float c = 0;                        // behind as the thread's "last error" for the next hipGetLastError of an unrelated call)
        GHIP(hipSetDevice(g->dev[0]));
        GHIP(hipEventSynchronize(g->ev[3]));
        GHIP(hipEventElapsedTime(&c, g->ev[2], g->ev[3]));
        tsel = c * 1e-3;
    }
    if (solve) *solve = ts;
    if (gather) *gather = tg;
    if (select) *select = tsel;
    return BROV_OK;
}

}  // extern "C"
