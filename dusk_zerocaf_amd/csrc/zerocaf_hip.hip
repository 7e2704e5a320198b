// zerocaf_hip.hip -- host side of libzerocaf_hip.so: context, residency detection,
// staging and kernel dispatch behind the C ABI of include/zerocaf_hip.h.
// gfx950 only; there is no CPU fallback anywhere in this library.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <dlfcn.h>
#include <rccl/rccl.h>      // types only: librccl is opened with dlopen when a communicator is requested

#include "../../include/zerocaf_hip.h"
#include "zc_kernels.hip.h"
#include "zc_msm.hip.h"

using zc::u64;

namespace {

thread_local std::string g_last_error;

int fail(int code, const char* what, hipError_t e = hipSuccess)
{
    g_last_error = what;
    if (e != hipSuccess) {
        g_last_error += ": ";
        g_last_error += hipGetErrorString(e);
    }
    return code;
}

#define HIP_TRY(expr)                                              \
    do {                                                           \
        hipError_t e_ = (expr);                                    \
        if (e_ != hipSuccess) return fail(ZC_ERR_HIP, #expr, e_);  \
    } while (0)

constexpr int MAX_ARGS = 6;

// Tuning knobs.  Read from the environment ONCE, when a context is created, and kept
// with the context: a process can hold contexts with different settings side by side, and no call path touches
// getenv afterwards.  0 / -1 = "not set": the library's own choice applies.
//   * the first block is what a caller may have a reason to choose; every build reads it;
//   * the second block are PATH FORCERS of the GPU test tier, read only by libzerocaf_hip_test.so (-DZC_TEST_HOOKS): they select,
//     at test sizes, the paths the product takes at other sizes (two-word sort records and 8192-key tiles of shards beyond
//     2^22 pairs, the in-line normalisation of shards beyond 2^23, run and segment lengths of other list lengths);
//   * variants that were measured and lost (chains in line, one lane per fold addition, packed 96-byte records, other
//     launch sizes ...) are compile-time macros below: `python -m dusk_zerocaf_amd.build --variant NAME MACRO=V` builds an
//     A/B library, the product never carries the switch.
struct Tuning {
    long host_chunks = 0;            // ZC_HOST_CHUNKS=k: host batches move in k chunks
    bool sched_block = false;        // ZC_SCHED=block: one workgroup per 256 elements instead of persistent waves
    unsigned ring_slots = 0;         // ZC_RING_SLOTS=k (1..512): wave slots per XCD of the windowed core's table ring
    bool ristretto_strict = false;   // ZC_RISTRETTO_STRICT=1: config-4 round trip on the reference's formula sequence
    long inv_chunk = 0;              // ZC_INV_CHUNK=c (1..64): elements per lane sharing one inversion
    int jacobi_rounds = -1;          // ZC_JACOBI_ROUNDS=r (0..200): rounds before legendre_symbol falls back to the power
    int msm_window = 0;              // ZC_MSM_WINDOW=c
    int msm_affine = -1;             // ZC_MSM_AFFINE=0/1: projective 128-byte records / affine 112-byte records whatever the shard size
    int msm_groups[4] = {0, 0, 0, 0};   // ZC_MSM_GROUPS="a,b[,c[,d]]": windows per group, top group first ("1" = one group)
    int msm_ngroups = 0;
    // ---- test-hooks build only
    int msm_sort_packed = -1;        // ZC_MSM_SORT_PACKED=0/1
    int msm_sort_big = -1;           // ZC_MSM_SORT_BIG=0/1
    long msm_sort_g = 0;             // ZC_MSM_SORT_G=g (1..64)
    int msm_run = 0;                 // ZC_MSM_RUN=T (4..4096)
    int msm_run_edges = 0;           // ZC_MSM_RUN_EDGES=T (4..4096, even)
    int msm_fork = -1;               // ZC_MSM_FORK=0/1
    int msm_affine_chunk = 0;        // ZC_MSM_AFFINE_CHUNK=c (1..64)
    int msm_seg = 0;                 // ZC_MSM_SEG=s (power of two, 2..256)
    long test_stream_min = 0;        // ZC_TEST_STREAM_MIN_BYTES=b: the 40-byte element ops take their LDS-staged kernels from b bytes per call on
    bool test_ring_poison = false;   // ZC_TEST_RING_POISON: pretend a wave of every windowed-core launch gave up
    unsigned test_ring_spins = 0;    // ZC_TEST_RING_SPINS=b: waves give up after 2^b polls (default 22, about 4 s)
};
// (the compile-time variants of the MSM pipeline: zc_msm.hip.h, ZC_MSM_* macros)
inline long env_long(const char* name, long lo, long hi, long unset)
{
    const char* e = getenv(name);
    if (!e || !*e) return unset;
    const long v = atol(e);
    return v >= lo && v <= hi ? v : unset;
}
Tuning tuning_from_env()
{
    Tuning t;
    t.host_chunks = env_long("ZC_HOST_CHUNKS", 1, 1 << 20, 0);
    if (const char* e = getenv("ZC_SCHED")) t.sched_block = std::string(e) == "block";
    t.ring_slots = (unsigned)env_long("ZC_RING_SLOTS", 1, 512, 0);
    t.ristretto_strict = env_long("ZC_RISTRETTO_STRICT", 0, 1 << 30, 0) != 0;
    t.inv_chunk = env_long("ZC_INV_CHUNK", 1, 64, 0);
    t.jacobi_rounds = (int)env_long("ZC_JACOBI_ROUNDS", 0, 200, -1);
    t.msm_window = (int)env_long("ZC_MSM_WINDOW", 1, 64, 0);
    t.msm_affine = (int)env_long("ZC_MSM_AFFINE", 0, 1 << 30, -1);
    if (const char* e = getenv("ZC_MSM_GROUPS")) {
        for (const char* q = e; *q && t.msm_ngroups < 4;) {
            char* end = nullptr;
            const long x = strtol(q, &end, 10);
            if (end == q || x < 1 || x > 64) break;
            t.msm_groups[t.msm_ngroups++] = (int)x;
            if (*end != ',') break;
            q = end + 1;
        }
    }
#ifdef ZC_TEST_HOOKS
    t.msm_sort_packed = (int)env_long("ZC_MSM_SORT_PACKED", 0, 1 << 30, -1);
    t.msm_sort_big = (int)env_long("ZC_MSM_SORT_BIG", 0, 1 << 30, -1);
    t.msm_sort_g = env_long("ZC_MSM_SORT_G", 1, 64, 0);
    t.msm_run = (int)env_long("ZC_MSM_RUN", 4, 4096, 0);
    t.msm_run_edges = (int)env_long("ZC_MSM_RUN_EDGES", 4, 4096, 0);
    t.msm_fork = (int)env_long("ZC_MSM_FORK", 0, 1 << 30, -1);
    t.msm_affine_chunk = (int)env_long("ZC_MSM_AFFINE_CHUNK", 1, 64, 0);
    {
        const long f = env_long("ZC_MSM_SEG", 2, 256, 0);
        if (f && (f & (f - 1)) == 0) t.msm_seg = (int)f;
    }
    t.test_stream_min = env_long("ZC_TEST_STREAM_MIN_BYTES", 1, 1l << 40, 0);
    t.test_ring_poison = getenv("ZC_TEST_RING_POISON") != nullptr;
    t.test_ring_spins = (unsigned)env_long("ZC_TEST_RING_SPINS", 1, 30, 0);
#endif
    return t;
}

struct DevState {
    int device = 0;
    Tuning tune;                        // the context's knobs (a copy per device slot)
    int cus = 256;                      // compute units (multiProcessorCount)
    hipStream_t stream = nullptr;       // owned
    hipStream_t borrowed = nullptr;     // set by zc_ctx_set_stream (device 0 only)
    bool use_borrowed = false;
    hipStream_t copy_in = nullptr;      // host batches: upload / download streams of the chunk pipeline
    hipStream_t copy_out = nullptr;
    std::vector<hipEvent_t> ev;         // 2 per chunk: inputs landed, kernel done
    void* scratch[MAX_ARGS] = {};
    size_t scratch_bytes[MAX_ARGS] = {};
    void* tmp[2] = {};                  // zc_msm partials
    size_t tmp_bytes[2] = {};
    void* bal = nullptr;                // lane balancing: 1024 u32 bins + n u32 indices
    size_t bal_bytes = 0;
    void* msm = nullptr;                // bucket-method workspace (zc_msm)
    size_t msm_bytes = 0;
    void* fast = nullptr;               // windowed-core tables: ring of wave slots, 256 MB (zc_kernels.hip.h)
    size_t fast_bytes = 0;
    void* ring = nullptr;               // tickets and slot flags of the table ring (+ the device address of the error word)
    size_t ring_bytes = 0;
    volatile zc::u32* ring_err = nullptr;   // the ring's error word: pinned host memory, written by a wave that gave up
    void* base_table = nullptr;         // comb table of the basepoint: 33 x 128 cached affine points
    size_t base_bytes = 0;
    void* odd_table = nullptr;          // (2j - 1) B, j = 1..125, cached affine: the w-NAF's odd multiples
    size_t odd_bytes = 0;
    void* part = nullptr;               // MSM exchange: gathered per-rank / per-device partials + the folded result
    size_t part_bytes = 0;
    hipEvent_t ev_order = nullptr;      // orders work across a stream switch / across devices
    hipStream_t aux = nullptr;          // MSM: the point normalisation runs beside the key sort (other priority than `stream`:
                                        // two streams of one priority share a hardware queue here and run one after the other)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipStream_t grp = nullptr;          // MSM window groups: the chains of the groups above the lowest one, one after the other (highest priority)
    hipEvent_t ev_grp_go[3] = {}, ev_grp_done[3] = {};   // group g's bucket sums are enqueued / its chain is through
    unsigned long long staged_launches = 0;   // element-wise / point launches that took the LDS-staged kernel (read by the test build's zc_test_staged_launches)
    hipStream_t s() const { return use_borrowed ? borrowed : stream; }
};

// librccl entry points (resolved at zc_comm_init; the library has no link-time dependency on RCCL)
struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

}  // namespace

struct zc_ctx {
    std::vector<DevState> devs;
    std::mutex mu;
    ncclComm_t comm = nullptr;          // zc_comm_init: one rank per process, device 0 of the context
    int rank = 0, world = 1;
};

namespace {

int ring_check(struct DevState& D);     // windowed-core table ring: error word of the last launches (defined with fast_ring)

// One buffer argument of a batched call.
struct Arg {
    const void* ptr;     // caller pointer (host or device), may be null when optional
    size_t elt_bytes;    // bytes per element
    bool is_out;
};

enum Residency { RES_HOST = 0, RES_DEVICE = 1 };

int residency_of(const void* p, Residency* res, int* device)
{
    hipPointerAttribute_t attr;
    hipError_t e = hipPointerGetAttributes(&attr, p);
    if (e != hipSuccess) {
        (void)hipGetLastError();                    // plain malloc memory: not known to HIP
        *res = RES_HOST;
        return ZC_OK;
    }
    if (attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged) {
        *res = RES_DEVICE;
        *device = attr.device;
    } else {
        *res = RES_HOST;                            // pinned / registered / unregistered host
    }
    return ZC_OK;
}

int ensure(void** buf, size_t* have, size_t need)
{
    if (*have >= need) return ZC_OK;
    if (*buf) HIP_TRY(hipFree(*buf));
    *buf = nullptr;
    *have = 0;
    size_t want = std::max(need, (size_t)1 << 20);
    hipError_t e = hipMalloc(buf, want);
    if (e != hipSuccess) return fail(ZC_ERR_NOMEM, "hipMalloc(scratch)", e);
    *have = want;
    return ZC_OK;
}

inline unsigned grid_for(size_t n) { return (unsigned)((n + zc::ZC_BLOCK - 1) / zc::ZC_BLOCK); }

// Host batches of the long-running kernels (scalar multiplications: >= 10 ms per 2^20 elements)
// move through the device in chunks so that only the first upload and the last download are
// exposed. A chunk is a whole number of full-chip rounds of workgroups (2^17 lanes = 512 blocks),
// otherwise every chunk pays a partially filled tail round: measured on 2^20 strict
// scalar-muls, 4 x 2^18 takes 26.1 ms, 3 chunks 30.2 ms, 1 chunk 32.0 ms (kernel alone 22.5 ms).
// Copy-bound calls (field / point element-wise ops) stay in one piece: pageable copies block the
// calling thread, so chunking them buys no overlap. ZC_HOST_CHUNKS=k forces k chunks.
constexpr size_t CHUNK_ROUND = (size_t)1 << 17;
constexpr size_t MAX_CHUNKS = 4096;
inline size_t host_chunk_elems(size_t cnt, bool heavy, long forced)
{
    size_t chunk = cnt;
    if (forced > 0)
        chunk = (cnt + forced - 1) / forced;
    else if (heavy && cnt >= 4 * CHUNK_ROUND)
        // eight chunks, but none below 2^18 elements: the exposed ends (first upload, last download) shrink with
        // the chunk, the per-chunk launches lose efficiency below 2^18 (2^22 strict scalar-muls: 16 chunks of
        // 2^18 89.4 ms, 8 of 2^19 83.1 ms, 4 of 2^20 85.7 ms; 2^20: 4 chunks of 2^18 23.8 ms, 8 of 2^17 34.8 ms)
        chunk = std::max(2 * CHUNK_ROUND, (cnt / 8 + CHUNK_ROUND - 1) / CHUNK_ROUND * CHUNK_ROUND);
    else if (heavy && cnt >= 2 * CHUNK_ROUND)
        chunk = CHUNK_ROUND;
    chunk = (chunk + 1023) / 1024 * 1024;
    return std::max(chunk, (cnt + MAX_CHUNKS - 1) / MAX_CHUNKS);
}
// Launch functor: receives device pointers in argument order, element count, device state.
template <class Launch>
int run_batched(zc_ctx* ctx, Arg* args, int nargs, size_t n, Launch&& launch, bool heavy = false)
{
    if (!ctx) return fail(ZC_ERR_BAD_ARG, "null context");
    if (nargs > MAX_ARGS) return fail(ZC_ERR_BAD_ARG, "too many arguments");
    if (n == 0) return ZC_OK;
    std::lock_guard<std::mutex> lock(ctx->mu);

    int ndevptr = 0, nhostptr = 0, dev_of_ptrs = -1;
    for (int a = 0; a < nargs; a++) {
        if (!args[a].ptr) continue;
        Residency r;
        int d = -1;
        residency_of(args[a].ptr, &r, &d);
        if (r == RES_DEVICE) {
            ndevptr++;
            if (dev_of_ptrs >= 0 && d != dev_of_ptrs) return fail(ZC_ERR_MIXED_MEM, "buffers on different devices");
            dev_of_ptrs = d;
        } else {
            nhostptr++;
        }
    }
    if (ndevptr && nhostptr) return fail(ZC_ERR_MIXED_MEM, "host and device buffers mixed in one call");

    if (ndevptr) {
        // in-place on the device that owns the buffers, asynchronous on the context stream
        DevState* ds = nullptr;
        for (auto& d : ctx->devs)
            if (d.device == dev_of_ptrs) ds = &d;
        if (!ds) return fail(ZC_ERR_MIXED_MEM, "device buffers do not belong to a device of this context");
        if (int rc = ring_check(*ds)) return rc;            // an asynchronous failure of an earlier call surfaces here
        HIP_TRY(hipSetDevice(ds->device));
        void* dptr[MAX_ARGS];
        for (int a = 0; a < nargs; a++) dptr[a] = const_cast<void*>(args[a].ptr);
        launch(dptr, n, *ds);
        HIP_TRY(hipGetLastError());
        return ZC_OK;
    }

    // host buffers: shard into contiguous ranges, one per device (no exchange step); each range
    // moves through its device in chunks so uploads, kernels and downloads overlap.  Copies from /
    // to pageable memory block the issuing thread, so every device gets its own worker thread:
    // the devices' uploads proceed side by side instead of one after the other (buffers pinned
    // with zc_host_register copy asynchronously in any case).
    const size_t ndev = ctx->devs.size();
    const size_t per = (n + ndev - 1) / ndev;

    auto device_job = [&](size_t di, std::string* err) -> int {
        const size_t lo = di * per, hi = std::min(n, lo + per);
        if (lo >= hi) return ZC_OK;
        const size_t total = hi - lo;
        const size_t chunk = host_chunk_elems(total, heavy, ctx->devs[di].tune.host_chunks);
        const size_t nchunks = (total + chunk - 1) / chunk;
        DevState& ds = ctx->devs[di];
        auto body = [&]() -> int {
            if (int rc0 = ring_check(ds)) return rc0;       // an asynchronous failure of an earlier call surfaces here
            HIP_TRY(hipSetDevice(ds.device));
            void* base[MAX_ARGS] = {};
            for (int a = 0; a < nargs; a++) {
                if (!args[a].ptr) continue;
                int rc = ensure(&ds.scratch[a], &ds.scratch_bytes[a], args[a].elt_bytes * total);
                if (rc) return rc;
                base[a] = ds.scratch[a];
            }
            while (ds.ev.size() < 2 * nchunks) {
                hipEvent_t e;
                HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
                ds.ev.push_back(e);
            }
            // chunk j: upload on copy_in, kernel on the context stream, download on copy_out
            auto upload_and_launch = [&](size_t j) -> int {
                if (j >= nchunks) return ZC_OK;
                const size_t off = j * chunk, cnt = std::min(chunk, total - off);
                void* dptr[MAX_ARGS];
                for (int a = 0; a < nargs; a++) {
                    dptr[a] = base[a] ? (char*)base[a] + args[a].elt_bytes * off : nullptr;
                    if (!args[a].ptr || args[a].is_out) continue;
                    const char* src = (const char*)args[a].ptr + args[a].elt_bytes * (lo + off);
                    HIP_TRY(hipMemcpyAsync(dptr[a], src, args[a].elt_bytes * cnt, hipMemcpyHostToDevice, ds.copy_in));
                }
                HIP_TRY(hipEventRecord(ds.ev[2 * j], ds.copy_in));
                HIP_TRY(hipStreamWaitEvent(ds.s(), ds.ev[2 * j], 0));
                launch(dptr, cnt, ds);
                HIP_TRY(hipGetLastError());
                HIP_TRY(hipEventRecord(ds.ev[2 * j + 1], ds.s()));
                return ZC_OK;
            };
            auto download = [&](size_t j) -> int {
                const size_t off = j * chunk, cnt = std::min(chunk, total - off);
                HIP_TRY(hipStreamWaitEvent(ds.copy_out, ds.ev[2 * j + 1], 0));
                for (int a = 0; a < nargs; a++) {
                    if (!args[a].ptr || !args[a].is_out) continue;
                    char* dst = (char*)const_cast<void*>(args[a].ptr) + args[a].elt_bytes * (lo + off);
                    HIP_TRY(hipMemcpyAsync(dst, (char*)base[a] + args[a].elt_bytes * off, args[a].elt_bytes * cnt, hipMemcpyDeviceToHost, ds.copy_out));
                }
                return ZC_OK;
            };
            // keep LOOKAHEAD kernels queued before the thread blocks on a download
            constexpr size_t LOOKAHEAD = 2;
            int rc = ZC_OK;
            for (size_t j = 0; j < std::min(LOOKAHEAD, nchunks) && !rc; j++) rc = upload_and_launch(j);
            for (size_t j = 0; j < nchunks && !rc; j++) {
                rc = download(j);
                if (!rc) rc = upload_and_launch(j + LOOKAHEAD);
            }
            HIP_TRY(hipStreamSynchronize(ds.copy_in));
            HIP_TRY(hipStreamSynchronize(ds.s()));
            HIP_TRY(hipStreamSynchronize(ds.copy_out));
            if (!rc) rc = ring_check(ds);
            return rc;
        };
        const int rc = body();
        if (rc && err) *err = g_last_error;                 // thread-local: hand the message to the caller's thread
        return rc;
    };

    if (ndev == 1 || n <= per) {
        std::string err;
        return device_job(0, &err);
    }
    std::vector<int> rcs(ndev, ZC_OK);
    std::vector<std::string> errs(ndev);
    std::vector<std::thread> workers;
    for (size_t di = 1; di < ndev; di++) workers.emplace_back([&, di] { rcs[di] = device_job(di, &errs[di]); });
    rcs[0] = device_job(0, &errs[0]);
    for (auto& t : workers) t.join();
    (void)hipSetDevice(ctx->devs[0].device);
    for (size_t di = 0; di < ndev; di++)
        if (rcs[di]) {
            g_last_error = errs[di];
            return rcs[di];
        }
    return ZC_OK;
}

inline Arg in_arg(const void* p, size_t b) { return Arg{p, b, false}; }
inline Arg out_arg(void* p, size_t b) { return Arg{p, b, true}; }

#define REQUIRE(p) \
    if (!(p)) return fail(ZC_ERR_BAD_ARG, "null pointer: " #p)

typedef void (*kbin_t)(const u64*, const u64*, u64*, size_t);
typedef void (*kun_t)(const u64*, u64*, size_t);

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// Streams larger than this (bytes over all arrays of the call) use the LDS-staged kernel
// `k_stream` when one is given: it wins only for the compute-free two-input ops beyond the
// 256 MB Infinity Cache (zc_kernels.hip.h, "LDS-staged element I/O").
constexpr size_t STREAM_BYTES = (size_t)256 << 20;

// `stream_min`: the call's bytes from which the staged kernel wins (STREAM_BYTES for the 40-byte element ops; the point ops
// -- 160-byte records, far beyond what a lane reads well on its own -- from the first full launch on).
// The test build can lower the 40-byte ops' threshold (ZC_TEST_STREAM_MIN_BYTES) so that small batches reach the staged kernels.
inline size_t staged_min(const DevState& D, size_t elt, size_t stream_min)
{
#ifdef ZC_TEST_HOOKS
    if (elt == 40 && D.tune.test_stream_min > 0) return (size_t)D.tune.test_stream_min;
#endif
    (void)D; (void)elt;
    return stream_min;
}
int binop(zc_ctx* ctx, kbin_t k, kbin_t k_stream, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n, size_t elt, size_t stream_min = STREAM_BYTES)
{
    REQUIRE(a); REQUIRE(b); REQUIRE(out);
    // elt == 0: (point, scalar) -> point
    Arg args[3] = {in_arg(a, elt ? elt : 160), in_arg(b, elt ? elt : 40), out_arg(out, elt ? elt : 160)};
    return run_batched(ctx, args, 3, n, [&](void** d, size_t cnt, DevState& D) {
        const bool stream = k_stream && cnt * elt * 3 > staged_min(D, elt, stream_min) && aligned16(d[0]) && aligned16(d[1]) && aligned16(d[2]);
        const unsigned blk = stream && elt == 160 ? (unsigned)zc::ED_STAGED_BLOCK : (unsigned)zc::ZC_BLOCK;     // the staged point kernels have their own workgroup size
        D.staged_launches += stream ? 1 : 0;
        hipLaunchKernelGGL(stream ? k_stream : k, dim3((unsigned)((cnt + blk - 1) / blk)), dim3(blk), 0, D.s(), (const u64*)d[0], (const u64*)d[1], (u64*)d[2], cnt);
    }, elt == 0);
}
int unop(zc_ctx* ctx, kun_t k, const uint64_t* a, uint64_t* out, size_t n, size_t elt, kun_t k_stream = nullptr, size_t stream_min = STREAM_BYTES)
{
    REQUIRE(a); REQUIRE(out);
    Arg args[2] = {in_arg(a, elt), out_arg(out, elt)};
    return run_batched(ctx, args, 2, n, [&](void** d, size_t cnt, DevState& D) {
        const bool stream = k_stream && cnt * elt * 2 > staged_min(D, elt, stream_min) && aligned16(d[0]) && aligned16(d[1]);
        const unsigned blk = stream && elt == 160 ? (unsigned)zc::ED_STAGED_BLOCK : (unsigned)zc::ZC_BLOCK;
        D.staged_launches += stream ? 1 : 0;
        hipLaunchKernelGGL(stream ? k_stream : k, dim3((unsigned)((cnt + blk - 1) / blk)), dim3(blk), 0, D.s(), (const u64*)d[0], (u64*)d[1], cnt);
    });
}

// Cost-sorted permutation for the unified-step kernels (see zc_kernels.hip.h "lane balancing").
// Returns nullptr (natural order) for small batches or when scratch cannot be had.
// The persistent-wave kernel walks it; the block-shaped kernels (small batches, ZC_SCHED=block) rank their 256 scalars in
// LDS instead, which keeps HBM traffic algorithmic.
constexpr size_t BALANCE_MIN_N = 1 << 14;
constexpr size_t BAL_COUNTERS = 64;                       // u32 work counters of the persistent kernels, after the bins
const zc::u32* balance_index(DevState& D, const u64* k, size_t cnt, zc::u32** counter)
{
    if (cnt < BALANCE_MIN_N || cnt > 0xFFFFFFFFull) return nullptr;
    const size_t need = (zc::ZC_COST_BINS + BAL_COUNTERS + cnt) * sizeof(zc::u32);
    if (ensure(&D.bal, &D.bal_bytes, need) != ZC_OK) return nullptr;
    zc::u32* hist = (zc::u32*)D.bal;
    zc::u32* idx = hist + zc::ZC_COST_BINS + BAL_COUNTERS;
    if (hipMemsetAsync(hist, 0, (zc::ZC_COST_BINS + BAL_COUNTERS) * sizeof(zc::u32), D.s()) != hipSuccess) return nullptr;
    hipLaunchKernelGGL(zc::k_sm_cost_hist, dim3(grid_for(cnt)), dim3(zc::ZC_BLOCK), 0, D.s(), k, hist, cnt);
    hipLaunchKernelGGL(zc::k_sm_cost_scan, dim3(1), dim3(zc::ZC_BLOCK), 0, D.s(), hist);
    hipLaunchKernelGGL(zc::k_sm_cost_scatter, dim3(grid_for(cnt)), dim3(zc::ZC_BLOCK), 0, D.s(), k, hist, idx, cnt);
    if (counter) *counter = hist + zc::ZC_COST_BINS;
    return idx;
}
// Strict scalar-mul batches from this size on run on persistent waves over the cost-sorted
// permutation (k_ed_scalar_mul_pw); ZC_SCHED=block keeps one workgroup per 256 elements.
constexpr size_t PW_MIN_ELEMS = (size_t)1 << 17;

// Launches of at most one workgroup per CU keep a single wave on every SIMD; a lone wave cannot
// hide the latency of the column-ordered multiplier's serial chain, so those launches run the
// variant with independent column chains (2^16 units: 2.09 -> 1.79 ms; from 384 workgroups on the
// default kernel is faster again).
constexpr unsigned SMALL_LAUNCH_BLOCKS = 256;
constexpr size_t QUAD_LAUNCH_ELEMS = (size_t)1 << 14;     // 4 lanes per element still leave one wave per SIMD
typedef void (*strict_kernel_t)(const u64*, const u64*, size_t, u64*, size_t);
inline strict_kernel_t strict_kernel_for(size_t cnt)
{
    return grid_for(cnt) <= SMALL_LAUNCH_BLOCKS ? zc::k_ed_scalar_mul_small : zc::k_ed_scalar_mul;
}
// The windowed-core kernels keep 1 KB of table scratch per lane in a ring of wave slots per XCD
// (zc_kernels.hip.h: ring_acquire / ring_release): 256 MB of tables plus 17 KB of tickets and flags, zeroed
// on the stream before every launch.  One launch covers the batch (the kernels index with 32 bits:
// beyond 2^31 elements the batch goes in pieces, one after the other on the stream).
// launch(table, ring_state, slots_per_xcd, offset, count).  ZC_RING_SLOTS=k (1..512) shrinks the ring so that
// waves really wait for one another (tests; the default leaves more slots than waves fit an XCD).
constexpr size_t FAST_MAX_LAUNCH = (size_t)1 << 31;
template <class L>
int fast_ring(DevState& D, size_t cnt, L&& launch)
{
    int rc = ensure(&D.fast, &D.fast_bytes, zc::RING_TABLE_BYTES);
    if (rc) return rc;
    if (!D.ring) {
        // The error word lives in pinned HOST memory the device can write (a wave that gives up stores through the
        // address parked behind the ring state): the host reads it at every entry point without any synchronisation.
        if (!D.ring_err) {
            void* h = nullptr;
            // coherent (fine-grained) so that the host sees the store while kernels run whatever HIP_HOST_COHERENT says;
            // portable so that every device slot of a multi-device context may map it
            hipError_t e = hipHostMalloc(&h, 64, hipHostMallocMapped | hipHostMallocCoherent | hipHostMallocPortable);
            if (e != hipSuccess) return fail(ZC_ERR_NOMEM, "hipHostMalloc(ring error word)", e);
            D.ring_err = (volatile zc::u32*)h;
            *D.ring_err = 0;
        }
        // the ring state is published (D.ring) only once the error word's address sits behind it: a failure on the way
        // leaves D.ring null and the next call starts over -- a wave never reads an unset address
        void* ring = nullptr;
        HIP_TRY(hipMalloc(&ring, zc::RING_ALLOC_WORDS * sizeof(zc::u32)));
        void* dev_view = nullptr;
        hipError_t e = hipHostGetDevicePointer(&dev_view, (void*)D.ring_err, 0);
        const u64 addr = (u64)(uintptr_t)dev_view;
        if (e == hipSuccess) e = hipMemcpy((zc::u32*)ring + zc::RING_ERR_WORD, &addr, sizeof addr, hipMemcpyHostToDevice);   // once per device slot
        if (e != hipSuccess) {
            (void)hipFree(ring);
            return fail(ZC_ERR_HIP, "windowed core: ring state setup", e);
        }
        D.ring = ring;
        D.ring_bytes = zc::RING_ALLOC_WORDS * sizeof(zc::u32);
    }
    // a launch hands out fewer than 2^19 generations of its slots (the 19-bit field of the word ring_acquire parks)
    const zc::u32 slots = D.tune.ring_slots ? (zc::u32)D.tune.ring_slots : zc::RING_SLOTS;
    zc::u32 slots_arg = slots;
#ifdef ZC_TEST_HOOKS
    slots_arg |= (zc::u32)D.tune.test_ring_spins << 16;      // test build: the kernels take their spin limit from the upper half
#endif
    const size_t max_launch = std::min(FAST_MAX_LAUNCH, (size_t)slots << 24);
    for (size_t off = 0; off < cnt; off += max_launch) {
        HIP_TRY(hipMemsetAsync(D.ring, 0, zc::RING_STATE_WORDS * sizeof(zc::u32), D.s()));
        launch((zc::u32*)D.fast, (zc::u32*)D.ring, slots_arg, off, std::min(max_launch, cnt - off));
    }
#ifdef ZC_TEST_HOOKS
    if (D.tune.test_ring_poison) *D.ring_err = 1;            // pretend a wave gave up (exercises the report-and-recover path)
#endif
    return ZC_OK;
}
// Did a wave of an earlier windowed-core launch on this device give up waiting for its table slot (zc_kernels.hip.h:
// ring_acquire)?  Such a wave writes poison outputs (limbs / bytes of all ones, ok = 0) and sets the error word in
// host memory; every entry point that touches the device looks at it first, and so does everything that synchronises.
// Reported once (ZC_ERR_HIP), then cleared: the context stays usable.
int ring_check(DevState& D)
{
    if (!D.ring_err || *D.ring_err == 0) return ZC_OK;
    *D.ring_err = 0;
    return fail(ZC_ERR_HIP, "windowed core: a wave timed out waiting for its table slot; the rows it owned hold poison (all ones) -- "
                            "the outputs of the last windowed-core calls on this device are not valid");
}
int scalar_mul_impl(zc_ctx* ctx, const uint64_t* p, const uint64_t* k, uint64_t* out, size_t n)
{
    REQUIRE(p); REQUIRE(k); REQUIRE(out);
    Arg args[3] = {in_arg(p, 160), in_arg(k, 40), out_arg(out, 160)};
    return run_batched(ctx, args, 3, n, [&](void** d, size_t cnt, DevState& D) {
        if (cnt >= PW_MIN_ELEMS && !D.tune.sched_block) {
            zc::u32* counter = nullptr;
            if (const zc::u32* perm = balance_index(D, (const u64*)d[1], cnt, &counter)) {
                hipLaunchKernelGGL(zc::k_ed_scalar_mul_pw, dim3((unsigned)(3 * D.cus)), dim3(zc::ZC_BLOCK), 0, D.s(), (const u64*)d[0], (const u64*)d[1],
                                   (u64*)d[2], perm, counter, (zc::u32)cnt);
                return;
            }
        }
        if (cnt <= QUAD_LAUNCH_ELEMS) {
            // four lanes per element: the batch cannot fill the chip anyway, so buy latency with lanes
            hipLaunchKernelGGL(zc::k_ed_scalar_mul_quad, dim3((unsigned)((cnt + 63) / 64)), dim3(zc::ZC_BLOCK), 0, D.s(), (const u64*)d[0],
                               (const u64*)d[1], (u64*)d[2], cnt);
            return;
        }
        hipLaunchKernelGGL(strict_kernel_for(cnt), dim3(grid_for(cnt)),
                           dim3(zc::ZC_BLOCK), 0, D.s(), (const u64*)d[0], (const u64*)d[1], (size_t)5, (u64*)d[2], cnt);
    }, true);
}
// the same scalar for every point, handed to the kernel by value
int scalar_mul_bcast(zc_ctx* ctx, const uint64_t* p, const uint64_t (&k)[5], uint64_t* out, size_t n)
{
    REQUIRE(p); REQUIRE(out);
    zc::scalar_arg ka;
    for (int j = 0; j < 5; j++) ka.l[j] = k[j];
    Arg args[2] = {in_arg(p, 160), out_arg(out, 160)};
    return run_batched(ctx, args, 2, n, [&](void** d, size_t cnt, DevState& D) {
        hipLaunchKernelGGL(zc::k_ed_scalar_mul_bcast, dim3(grid_for(cnt)), dim3(zc::ZC_BLOCK), 0, D.s(), (const u64*)d[0], ka, (u64*)d[1], cnt);
    }, true);
}

// ---------------------------------------------------------------- MSM device pipeline
constexpr size_t MSM_BUCKET_MIN_N = 1 << 12;

// pairwise folds until one point is left; returns the buffer holding it
const u64* fold_all(DevState& D, u64* a, u64* b, size_t cnt)
{
    u64* cur = a;
    u64* nxt = b;
    while (cnt > 1) {
        hipLaunchKernelGGL(zc::k_ed_fold_pairs, dim3(grid_for((cnt + 1) / 2)), dim3(zc::ZC_BLOCK), 0, D.s(), (const u64*)cur, nxt, cnt);
        cnt = (cnt + 1) / 2;
        std::swap(cur, nxt);
    }
    return cur;
}

struct Carver {
    char* base;
    size_t off = 0;
    template <class T> T* take(size_t count)
    {
        T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
        off += (count * sizeof(T) + 255) & ~(size_t)255;
        return p;
    }
};

// Window width: signed digits put 2^(c-1) buckets in a window; c = log2(n) - 4 keeps about 32
// points per bucket, where the bucket reduction (~3.7 additions per bucket) stays well below the
// bucket sums (1 addition per point and window); measured flat within 3 % for c +- 1 up to 2^18 and at 2^21,
// and for c = 18..21 at 2^24 (tools/quick_bench.py msmsweep).  ZC_MSM_WINDOW=c overrides (tests, tuning; read at context creation like every knob).
int msm_window_bits(size_t cnt, const Tuning& tune)
{
    int c = 0;
    while (((size_t)1 << (c + 1)) <= cnt) c++;
    c -= 4;
    if (c == 15 || c == 16) c = 17;                       // 2^19, 2^20 pairs: 16 windows of 17 bits beat 18 of 15 / 17 of 16 (measured -4 %)
    if (c < zc::MSM_MIN_C) c = zc::MSM_MIN_C;
    if (c > 18) c = 18;                                   // beyond: no faster (2^24 pairs: c = 18 / 19 / 20: 21.4 / 21.9 / 22.9 ms), bucket memory doubles per step
    if (tune.msm_window >= zc::MSM_MIN_C && tune.msm_window <= zc::MSM_MAX_C) c = tune.msm_window;
    return c;
}

// The key sort of the MSM (zc_sort.hip.h): `passes` stable counting-sort passes over the c - 1 digit bits of
// every window, at most 9 bits each.  A table column = G tiles of 4096 keys walked by one workgroup of the
// scatter kernel; G grows with the batch so that every window keeps about 128 columns (2^21 pairs per window:
// G = 4; 2^24: G = 16), which bounds the table (windows x bins x columns words) at a few MB.
struct MsmSortPlan {
    int passes = 0;
    bool packed = false;                                      // two passes with the one-word intermediate (zc_sort.hip.h):
                                                              // sign | high digit bits + the zero-digit flag | point index fit 32 bits
    bool big = false;                                         // tiles of 8192 keys
    zc::msm_sort_pass pass[4];
    size_t table_words = 0;                                   // largest table, padded to whole scan blocks
};
MsmSortPlan msm_sort_plan(size_t n, int c, int W, const Tuning& tune)
{
    MsmSortPlan pl;
    const int B = c - 1;
    pl.passes = (B + zc::MSM_SORT_PASS_BITS - 1) / zc::MSM_SORT_PASS_BITS;
    // two-word records of large batches: tiles of 8192 keys (ZC_MSM_SORT_BIG=0/1 forces the choice)
    int idx_bits = 1;
    while (((size_t)1 << idx_bits) < n) idx_bits++;
    pl.packed = pl.passes == 2 && 1 + (B - (B + 1) / 2) + 1 + idx_bits <= 32 && tune.msm_sort_packed != 0;
    pl.big = !pl.packed && n >= ((size_t)1 << 22);
    if (tune.msm_sort_big >= 0) pl.big = !pl.packed && tune.msm_sort_big != 0;
    const size_t tile = (size_t)zc::ZC_BLOCK * (pl.big ? zc::MSM_SORT_KPT_BIG : zc::MSM_SORT_KPT);
    const size_t ntiles = (n + tile - 1) / tile;
    size_t G = ntiles / 128;
    G = std::max<size_t>(1, std::min<size_t>(16, G));
    if (tune.msm_sort_g) G = (size_t)tune.msm_sort_g;
    const size_t ncols = (ntiles + G - 1) / G;
    int shift = 0;
    for (int i = 0; i < pl.passes; i++) {
        const int bits = B / pl.passes + (i < B % pl.passes ? 1 : 0);
        zc::msm_sort_pass& p = pl.pass[i];
        p.n = (zc::u32)n;
        p.W = (zc::u32)W;
        p.tile = (zc::u32)tile;
        p.G = (zc::u32)G;
        p.ncols = (zc::u32)ncols;
        p.shift = (zc::u32)shift;
        p.bits = (zc::u32)bits;
        p.last = i + 1 == pl.passes ? 1u : 0u;
        p.c = (zc::u32)c;
        p.idx_bits = pl.packed ? (zc::u32)idx_bits : 0;
        p.w0 = 0;
        shift += bits;
        const size_t words = ((size_t)W * ((size_t)1 << bits) + (p.last ? (size_t)W : 0)) * ncols;
        pl.table_words = std::max(pl.table_words, (words + zc::SCAN_BLOCK_ELEMS - 1) / zc::SCAN_BLOCK_ELEMS * zc::SCAN_BLOCK_ELEMS);
    }
    return pl;
}
// table words of one pass over `nw` windows, padded to whole scan blocks
inline size_t msm_sort_table_words(const zc::msm_sort_pass& p, size_t nw)
{
    const size_t words = (nw * ((size_t)1 << p.bits) + (p.last ? nw : 0)) * p.ncols;
    return (words + zc::SCAN_BLOCK_ELEMS - 1) / zc::SCAN_BLOCK_ELEMS * zc::SCAN_BLOCK_ELEMS;
}
// Sorts the windows [w0, w0 + nw) of the window-major digit words on stream `st`: pairs ordered by bucket in buf_a, in the
// windows' own part of the arrays ([w0 n, (w0 + nw) n): buckets first, the zero digits of these windows behind them).  With
// w0 = 0, nw = W that is the whole list with every zero digit at its end; a pipeline that takes the windows in groups sorts
// every group on its own, so that the bucket sums of the top group need not wait for the keys of the others.
// buf_b holds m pairs, or m words when the plan is packed; `tables` = two tables of `table_words` words (this call's own);
// the last pass's scanned table stays in tables + ((passes - 1) & 1) * table_words (positions relative to w0 n).
int msm_sort(DevState& D, hipStream_t st, const MsmSortPlan& pl, int w0, int nw, const zc::u32* digits, uint2* buf_a, void* buf_b, zc::u32* tables, size_t table_words,
             zc::u32* sums)
{
    const size_t base = (size_t)w0 * pl.pass[0].n;            // the group's first entry in every array
    const zc::u32* in = digits + base;
    uint2* a = buf_a + base;
    void* b = !buf_b ? nullptr : pl.packed ? (void*)((zc::u32*)buf_b + base) : (void*)((uint2*)buf_b + base);
    // the last pass writes buf_a; the passes before it alternate so that no pass reads what it writes
    void* out = (pl.passes & 1) ? (void*)a : b;
    for (int i = 0; i < pl.passes; i++) {
        zc::msm_sort_pass p = pl.pass[i];
        p.W = (zc::u32)nw;
        p.w0 = (zc::u32)w0;
        zc::u32* table = tables + (size_t)(i & 1) * table_words;
        const size_t words = ((size_t)p.W * ((size_t)1 << p.bits) + (p.last ? (size_t)p.W : 0)) * p.ncols;
        const size_t padded = msm_sort_table_words(p, (size_t)nw);
        const unsigned nblk = (unsigned)(padded / zc::SCAN_BLOCK_ELEMS), grid = (unsigned)(p.W * p.ncols);
        if (padded > words) HIP_TRY(hipMemsetAsync(table + words, 0, (padded - words) * sizeof(zc::u32), st));
        hipLaunchKernelGGL(!i ? zc::k_msm_sort_hist : pl.packed ? zc::k_msm_sort_hist_packed : zc::k_msm_sort_hist_pairs, dim3(grid), dim3(zc::ZC_BLOCK), 0, st, in, table, p);
        hipLaunchKernelGGL(zc::k_scan_reduce, dim3(nblk), dim3(zc::ZC_BLOCK), 0, st, (const zc::u32*)table, sums);
        hipLaunchKernelGGL(zc::k_scan_sums, dim3(1), dim3(zc::ZC_BLOCK), 0, st, sums, (zc::u32)nblk);
        hipLaunchKernelGGL(zc::k_scan_apply, dim3(nblk), dim3(zc::ZC_BLOCK), 0, st, table, (const zc::u32*)sums);
        if (pl.packed && i == 0)
            hipLaunchKernelGGL(zc::k_msm_sort_scatter_pack, dim3(grid), dim3(zc::ZC_BLOCK), 0, st, in, (zc::u32*)out, (const zc::u32*)table, p);
        else if (pl.packed)
            hipLaunchKernelGGL(zc::k_msm_sort_scatter_unpack, dim3(grid), dim3(zc::ZC_BLOCK), 0, st, in, (uint2*)out, (const zc::u32*)table,
                               (const zc::u32*)(tables + (size_t)((i - 1) & 1) * table_words), p);
        else
            hipLaunchKernelGGL(pl.big ? (i ? zc::k_msm_sort_scatter_pairs_big : zc::k_msm_sort_scatter_big) : (i ? zc::k_msm_sort_scatter_pairs : zc::k_msm_sort_scatter),
                               dim3(grid), dim3(zc::ZC_BLOCK), 0, st, in, (uint2*)out, (const zc::u32*)table, p);
        in = reinterpret_cast<const zc::u32*>(out);
        out = out == (void*)a ? b : (void*)a;
    }
    return ZC_OK;
}

// Affine cached records (7-multiplication bucket additions, 112-byte gathers) from this many points on: the
// normalisation costs one division-step inversion per lane, which small batches cannot amortise.
// ZC_MSM_AFFINE=0/1 forces the choice (tests, A/B).
constexpr size_t MSM_AFFINE_MIN_N = (size_t)1 << 17;
inline bool msm_affine(size_t cnt, const Tuning& tune)
{
    if (tune.msm_affine >= 0) return tune.msm_affine != 0;
    return cnt >= MSM_AFFINE_MIN_N;
}

// Everything the pipeline derives from the shard size and the knobs, in one place (also what zc_msm_plan reports).
struct MsmPlan {
    bool buckets = false;          // false: below MSM_BUCKET_MIN_N -- n scalar multiplications + pairwise folds
    int c = 0, W = 0;              // window bits, windows
    bool affine = false;           // affine records (27 limb words) + 7-multiplication additions (else 128-byte projective, 8)
    int T = 0, TE = 0;             // run lengths of the segmented reduction: level 0, deeper levels
    int seg = 0;                   // buckets per reduction segment
    size_t m = 0, nb = 0, nseg = 0;   // list entries (n W), buckets, segments
    int rec_bytes = 128;           // stride of the cached records (affine: 96 packed or 128 = one per cache line; projective: 128)
    int G = 1;                     // window groups, top windows first: gw[g] windows, run length gT[g]
    int gw[4] = {0, 0, 0, 0}, gT[4] = {0, 0, 0, 0};
    int gseg[4] = {0, 0, 0, 0};    // buckets per reduction segment, per group (the lowest group's chain is exposed: shorter segments)
    int bad_groups = 0;            // ZC_MSM_GROUPS was given and adds up to this many windows instead of W: the call fails
    MsmSortPlan sort;
};
// Run length of the bucket-sum kernel for a list of m entries: 128 entries per lane, fewer when the list is short (keep
// >= 2^17 lanes = two waves per SIMD busy); longer runs leave fewer edges (2 per run) for the deeper levels.
// Measured (tools/quick_bench.py, ZC_MSM_RUN / ZC_MSM_RUN_EDGES): 2^20 pairs T = 32 / 128 / 256: 2.90 / 2.83 / 3.12 ms;
// 2^21: 4.67 / 4.48 / 4.52; edge runs of 8 / 16 / 32: 2^21 4.44 / 4.53 / 4.63 ms.  Round 3, 2^24 pairs (2^27.9
// entries): T = 128 / 256: 21.94 / 21.53 ms -- half the edges for the deeper levels.
inline int msm_run_length(size_t m, const Tuning& tune, int lanes_log2 = 17)
{
    if (tune.msm_run) return tune.msm_run;                // T >= 4: every level shortens the list (2 ceil(len / T) < len)
    return m >= ((size_t)1 << 27) ? 256 : (int)std::min<size_t>(128, std::max<size_t>(8, m >> lanes_log2));
}
MsmPlan msm_plan(size_t cnt, bool points_aligned16, const Tuning& tune)
{
    MsmPlan p;
    if (cnt < MSM_BUCKET_MIN_N) return p;
    p.buckets = true;
    p.c = msm_window_bits(cnt, tune);
    p.W = (zc::MSM_SCALAR_BITS + p.c - 1) / p.c;          // any 260-bit pattern + the recoding carry
    p.m = cnt * (size_t)p.W;
    p.nb = (size_t)p.W << (p.c - 1);                      // buckets (digit magnitudes 1 .. 2^(c-1) per window)
    p.seg = tune.msm_seg ? tune.msm_seg : zc::msm_segment_buckets(p.nb);
    while (p.seg > (1 << (p.c - 1))) p.seg >>= 1;        // a segment never spans windows (ZC_MSM_SEG beside a narrow ZC_MSM_WINDOW)
    p.nseg = p.nb / (size_t)p.seg;
    p.sort = msm_sort_plan(cnt, p.c, p.W, tune);
    p.affine = msm_affine(cnt, tune) && points_aligned16;  // the normalisation moves the point records with 16-byte loads
    // affine records: 108 bytes of payload (rounds 3-5: 96) at a 128-byte stride -- one record per cache line.  Packed (96-byte stride) three records
    // of four straddle two lines: measured (rocprofv3 TCC_EA0_RDREQ of k_msm_runs_affine, profiles/r04_msm_record_stride.md)
    // 34.6 -> 25.1 read requests per pair at 2^21 pairs, 34.3 -> 27.7 at 2^24; 2^21: 3.50 -> 3.50 ms, 2^22: 6.30 -> 6.13, 2^24: 21.18 -> 20.22.
    p.rec_bytes = p.affine ? ZC_MSM_REC_STRIDE : 128;
    p.T = msm_run_length(p.m, tune);
    p.TE = 8;                                             // deeper levels: short lists, short runs (even: see k_msm_runs_edges)
    if (tune.msm_run_edges) p.TE = tune.msm_run_edges & ~1;
    // Window groups (msm_on_device): ZC_MSM_GROUPS="a,b,.." = windows per group, top group first; must add up to W.
    p.G = 1;
    p.gw[0] = p.W;
    {
        int sum = 0;
        for (int g = 0; g < tune.msm_ngroups; g++) sum += tune.msm_groups[g];
        if (tune.msm_ngroups >= 2 && sum == p.W) {
            p.G = tune.msm_ngroups;
            for (int g = 0; g < p.G; g++) p.gw[g] = tune.msm_groups[g];
        } else if (tune.msm_ngroups >= 2) {
            p.bad_groups = sum;                           // fail closed: a split for another window count is not silently replaced by one group
        } else if (tune.msm_ngroups == 0 && p.W >= 8 && cnt >= ((size_t)1 << 21) && cnt < ((size_t)1 << 22)) {
            // default for config-5-sized shards (2^21 pairs: 16 windows as 9 + 4 + 3): three groups, the lowest (exposed) one the
            // smallest.  Measured on one box, 2^21 pairs (tools/msm_groups_sweep.py): one group 3.68 ms, 13+3 3.50, 12+4 3.51,
            // 10+6 3.74, 7+6+3 3.55, 8+5+3 3.55, 9+4+3 3.44, 6+6+4 3.50, four groups 3.8 - 4.1.  Below 2^21 and from 2^22 on
            // the groups gain nothing (2^20: 2.39 -> 2.59 ms; 2^22: 6.27 -> 6.24; 2^24: 21.2 -> 21.6): one group.
            p.G = 3;
            p.gw[2] = std::max(1, (3 * p.W + 8) / 16);
            p.gw[1] = std::max(1, (4 * p.W + 8) / 16);
            p.gw[0] = p.W - p.gw[1] - p.gw[2];
        }
    }
    // a group's launch keeps 2^17 lanes busy like the whole list (ZC_MSM_GROUP_LANES=16 / 17 / 18 / 19 at 2^21 pairs in three groups:
    // 3.71 / 3.48 / 3.61 / 4.30 ms: shorter runs cut more buckets, and every cut is an edge for the levels behind)
    for (int g = 0; g < p.G; g++)
        p.gT[g] = p.G == 1 ? p.T : msm_run_length(cnt * (size_t)p.gw[g], tune, ZC_MSM_GROUP_LANES);
    // Segment length per group.  (ZC_MSM_LOW_SEG_HALF: half the length for the lowest group, whose chain is on the call's critical
    // path -- 39 instead of 54 dependent additions; measured in round 6 and not taken, the segments are not pure latency.)
    for (int g = 0; g < p.G; g++) {
        p.gseg[g] = p.seg;
        const size_t nsegg2 = 2 * (size_t)p.gw[g] * (((size_t)1 << (p.c - 1)) / (size_t)p.seg);
        if (ZC_MSM_LOW_SEG_HALF && p.G > 1 && g == p.G - 1 && !tune.msm_seg && p.seg >= 4 && nsegg2 <= 2 * (size_t)ZC_MSM_SEG_QUAD) p.gseg[g] = p.seg / 2;
    }
    p.nseg = 0;
    for (int g = 0; g < p.G; g++) p.nseg += (size_t)p.gw[g] * (((size_t)1 << (p.c - 1)) / (size_t)p.gseg[g]);
    return p;
}

// sum_i k_i P_i of one device's shard, enqueued on D.s() without any host synchronisation;
// *result points at the 160-byte sum in D's memory (valid until the next MSM on this device).
int msm_on_device(DevState& D, const u64* dP, const u64* dK, size_t cnt, const u64** result)
{
    if (cnt < MSM_BUCKET_MIN_N) {
        // small shard: n scalar-muls, then pairwise folds
        int rc = ensure(&D.tmp[0], &D.tmp_bytes[0], cnt * 160);
        if (rc) return rc;
        rc = ensure(&D.tmp[1], &D.tmp_bytes[1], ((cnt + 1) / 2) * 160 + 256);
        if (rc) return rc;
        hipLaunchKernelGGL(strict_kernel_for(cnt), dim3(grid_for(cnt)), dim3(zc::ZC_BLOCK), 0, D.s(), dP, dK, (size_t)5, (u64*)D.tmp[0], cnt);
        *result = fold_all(D, (u64*)D.tmp[0], (u64*)D.tmp[1], cnt);
        HIP_TRY(hipGetLastError());
        return ZC_OK;
    }
    if (cnt > 0x7FFFFFFFull) return fail(ZC_ERR_BAD_ARG, "zc_msm: shard too large for 31-bit point indices");
    const Tuning& tune = D.tune;
    const MsmPlan mp = msm_plan(cnt, aligned16(dP), tune);
    if (mp.bad_groups) {
        char msg[160];
        snprintf(msg, sizeof msg, "zc_msm: ZC_MSM_GROUPS adds up to %d windows, a shard of %zu pairs has %d (%d-bit windows)", mp.bad_groups, cnt, mp.W, mp.c);
        return fail(ZC_ERR_BAD_ARG, msg);
    }
    const int c = mp.c, W = mp.W, TE = mp.TE;
    const size_t m = mp.m, nb = mp.nb, nseg = mp.nseg;
    if (m > 0xFFFFFFFFull) return fail(ZC_ERR_BAD_ARG, "zc_msm: shard too large for 32-bit pair indices");
    const MsmSortPlan& plan = mp.sort;
    const bool affine = mp.affine;
    // ---- window groups.  Everything behind the bucket sums -- the deeper levels of the segmented reduction, the bucket
    // reduction, Horner's rule -- is a chain of dependent point operations on few waves: a third of a 2^21 shard during
    // which the chip idles.  The sorted list is ordered by window, so the bucket-sum kernel is launched over the windows
    // in GROUPS, top windows first (each launch takes its part of the list from the sort's own scan table, on the device;
    // the launches follow one another on the caller's stream); the chain of group g -- edges, segments, folds, its stretch
    // of Horner's rule -- runs on a side stream BESIDE the bucket sums of the groups below, with raised wave priority
    // (zc_msm.hip.h: msm_tail_priority).  The chains of the upper groups follow one another on one side stream; only the
    // lowest group's chain, which has the fewest buckets and the shortest stretch of Horner's rule, is exposed.  One
    // sort, one bucket array, one workspace; G = 1 is the pipeline of rounds 2-3.
    // (One launch for all groups with a per-group count of finished waves and a waiting kernel on the side stream was
    // built and measured: the release fence every wave then needs writes the XCD's whole L2 back -- the launch took
    // 2.2 - 3.3 ms instead of 1.8 -- and at 128 entries per run the launch is a single round of resident workgroups
    // anyway, so its groups all end together.)
    const int G = mp.G;
    struct Group {
        int w0 = 0, nw = 0, T = 0;
        size_t nl0 = 0, slot0 = 0;                         // level-0 lanes (upper bound: the list part's length is known on the device only), first lane in the edge arrays
        hipStream_t st = nullptr;
    } grp[4];
    size_t lanes_total = 0;
    {
        int top = W;
        for (int g = 0; g < G; g++) {
            grp[g].nw = mp.gw[g];
            top -= mp.gw[g];
            grp[g].w0 = top;
            grp[g].T = mp.gT[g];
            grp[g].nl0 = (cnt * (size_t)grp[g].nw + grp[g].T - 1) / grp[g].T;
            grp[g].slot0 = lanes_total;
            lanes_total += grp[g].nl0;
            grp[g].st = g == G - 1 ? D.s() : D.grp;          // ONE side stream: streams of one priority share a hardware queue here anyway
        }
    }
    size_t seg_off[5] = {0, 0, 0, 0, 0};                        // group g's part of the segment arrays (segments per window: a power of two per group)
    for (int g = 0; g < G; g++) seg_off[g + 1] = seg_off[g] + (size_t)grp[g].nw * (((size_t)1 << (c - 1)) / (size_t)mp.gseg[g]);
    const zc::u32 rec_words = affine ? (zc::u32)(mp.rec_bytes / 4) : 32u;
    for (int pass = 0; pass < 2; pass++) {
        Carver cv{pass ? (char*)D.msm : nullptr};
        zc::u32* digits = cv.take<zc::u32>(m);
        uint2* pairs_a = cv.take<uint2>(m);
        void* pairs_b = plan.passes == 1 ? nullptr : plan.packed ? (void*)cv.take<zc::u32>(m) : (void*)cv.take<uint2>(m);
        // one sort for all windows (a sort per window group under the bucket sums of the group above was built and measured in
        // round 5 -- not taken; the patch: tools/debug/probes/msm_sort_per_group.patch)
        zc::u32* sort_table = cv.take<zc::u32>(2 * plan.table_words);
        zc::u32* sort_sums = cv.take<zc::u32>(plan.table_words / zc::SCAN_BLOCK_ELEMS + 1);
        zc::u32* cached = cv.take<zc::u32>(cnt * 32);
        zc::u32* buckets = cv.take<zc::u32>(nb * zc::MSM_RAW_WORDS);
        uint8_t* present = cv.take<uint8_t>(nb);
        zc::u32* ekeys[2] = {cv.take<zc::u32>(2 * lanes_total), cv.take<zc::u32>(2 * lanes_total)};       // edge lists, ping-pong
        zc::u32* erecs[2] = {cv.take<zc::u32>(2 * lanes_total * zc::MSM_RAW_WORDS), cv.take<zc::u32>(2 * lanes_total * zc::MSM_RAW_WORDS)};
        u64* seg_out = cv.take<u64>(nseg * 20);
        u64* fold_b = cv.take<u64>(nseg * 20);
        u64* grp_out = cv.take<u64>((size_t)(G + 1) * 20);  // Horner's rule after every group; the lowest group's is the result
        if (!pass) {
            int rc = ensure(&D.msm, &D.msm_bytes, cv.off);
            if (rc) return rc;
            continue;
        }
        // the point normalisation (an inversion-heavy, half compute-bound pass) runs on a second stream beside the key
        // sort (latency- and bandwidth-bound): they share no buffer, and the bucket sums wait for both.  Measured, 2^21 pairs:
        // 3.89 -> 3.80 ms (the sort's kernels slow down beside it, the pair still ends 60-90 us earlier); at 2^24 both sides are
        // bandwidth-bound for milliseconds and the pair ends no earlier (21.5 vs 21.7 ms), so large shards stay in line.
        // ZC_MSM_FORK=0/1 forces the choice.
        const bool fork = tune.msm_fork >= 0 ? tune.msm_fork != 0 : cnt < ((size_t)1 << 23);
        hipStream_t ps = D.s();
        if (fork && D.aux) {
            HIP_TRY(hipEventRecord(D.ev_fork, D.s()));
            HIP_TRY(hipStreamWaitEvent(D.aux, D.ev_fork, 0));
            ps = D.aux;
        }
        if (affine) {
            // points per lane of the normalisation: a CU holds twelve of its one-wave workgroups (LDS), so 2^17 lanes = 2048 waves are
            // one round of resident waves with room left for the key sort beside them, and 8 - 16 points amortise the lane's
            // inversion (round 6, prefetching kernel: 2^20 pairs 4 -> 8 per lane 2.30 -> 2.26 ms, 2^21 8 -> 16 3.34 -> 3.31, flat
            // from 10 to 16; ZC_MSM_AFFINE_CHUNK overrides)
            int ac = (int)std::min<size_t>(16, std::max<size_t>(1, cnt >> 17));
            if (tune.msm_affine_chunk) ac = tune.msm_affine_chunk;
            const size_t lanes = (cnt + ac - 1) / ac;            // lane g owns points g, g + stride, ...: stride = the launch's lanes
            hipLaunchKernelGGL(zc::k_msm_prepare_affine, dim3((unsigned)((lanes + zc::MSM_PREP_BLOCK - 1) / zc::MSM_PREP_BLOCK)), dim3(zc::MSM_PREP_BLOCK), 0, ps, dP, cached, cnt, ac, rec_words);
        } else {
            hipLaunchKernelGGL(aligned16(dP) ? zc::k_msm_prepare : zc::k_msm_prepare_lane, dim3(grid_for(cnt)), dim3(zc::ZC_BLOCK), 0, ps, dP, cached, cnt);
        }
        if (ps != D.s()) HIP_TRY(hipEventRecord(D.ev_join, ps));
        hipLaunchKernelGGL(zc::k_msm_digits, dim3(grid_for(cnt)), dim3(zc::ZC_BLOCK), 0, D.s(), dK, digits, cnt, c, W);
        const uint2* sorted = pairs_a;
        // The key sort
        HIP_TRY(hipMemsetAsync(present, 0, nb, D.s()));       // one flag per bucket: record written (else: empty = identity)
        if (int rc = msm_sort(D, D.s(), plan, 0, W, digits, pairs_a, pairs_b, sort_table, plan.table_words, sort_sums)) return rc;
        if (ps != D.s()) HIP_TRY(hipStreamWaitEvent(D.s(), D.ev_join, 0));
        // where window w's part of the sorted list starts: the last pass's scanned table at (window w, bin 0, column 0); the row
        // behind the last window is the zero digits' = the end of the buckets (zc_sort.hip.h: msm_sort_slot).
        const zc::msm_sort_pass& lastp = plan.pass[plan.passes - 1];
        auto window_start = [&](int w) {
            const zc::u32* last_table = sort_table + (size_t)((plan.passes - 1) & 1) * plan.table_words;
            return last_table + ((size_t)w << lastp.bits) * lastp.ncols;
        };
        for (int g = 0; g < G; g++) {
            Group& gr = grp[g];
            const size_t b0 = (size_t)gr.w0 << (c - 1);                  // the group's first bucket
            const int seg = mp.gseg[g];
            const size_t nsegg = seg_off[g + 1] - seg_off[g];
            // A launch that runs beside the chain of the group above it leaves that chain room: its workgroups are padded with
            // dynamic LDS so that only `wgs` of them fit a CU (three: one wave slot per SIMD, 200 VGPRs and 39 KB of LDS stay free;
            // a chain kernel that finds every slot taken waits for a bucket-sum workgroup to retire).
            size_t pad = 0;
            if (g > 0) {
                const long wgs = ZC_MSM_GROUP_WGS;
                const size_t own = (affine ? zc::MSM_AFF_PIECES : 8) * 16 * (size_t)zc::MSM_RUN_BLOCK;       // the kernel's static staging area
                if (wgs > 0 && (size_t)(wgs + 1) * own <= 163840) {
                    const size_t per = 163840 / (size_t)(wgs + 1) + 512;                     // wgs + 1 of these do not fit 160 KB
                    pad = per > own ? std::min<size_t>(per - own, 65536 - own) : 0;
                }
            }
            hipLaunchKernelGGL(affine ? zc::k_msm_runs_affine : zc::k_msm_runs, dim3((unsigned)((gr.nl0 + zc::MSM_RUN_BLOCK - 1) / zc::MSM_RUN_BLOCK)), dim3(zc::MSM_RUN_BLOCK), pad,
                               D.s(), sorted, (const zc::u32*)cached, (zc::u32)m, (zc::u32)gr.T, (zc::u32)nb, buckets, present, ekeys[0], erecs[0],
                               G == 1 ? (const zc::u32*)nullptr : window_start(gr.w0), G == 1 ? (const zc::u32*)nullptr : window_start(gr.w0 + gr.nw),
                               (zc::u32)gr.nl0, (zc::u32)gr.slot0, rec_words);
            hipStream_t st = ZC_MSM_TAIL_SIDE ? gr.st : D.s();
            if (st != D.s()) {
                HIP_TRY(hipEventRecord(D.ev_grp_go[g], D.s()));
                HIP_TRY(hipStreamWaitEvent(st, D.ev_grp_go[g], 0));
            }
            // deeper levels of the segmented reduction: the edge list of the level above, level by level, in short runs (a bucket
            // cut once closes at level 1: nearly every edge of a uniform batch; the levels behind it find sentinel keys only and
            // take 5 us each.  Runs of 64 there -- 5 launches instead of 9 -- were measured: a lane then walks 64 sentinel keys
            // one dependent load after the other, 340 us per level instead of 5).
            {
                const zc::u32* lk = ekeys[0] + 2 * gr.slot0;
                const zc::u32* lr = erecs[0] + 2 * gr.slot0 * zc::MSM_RAW_WORDS;
                size_t len = 2 * gr.nl0;
                for (int level = 1; gr.nl0 > 1; level++) {
                    // runs shifted by one entry, [jT+1, (j+1)T+1), run 0 one longer
                    const size_t t = (size_t)TE;
                    const size_t nl = len <= t + 1 ? 1 : (len - 1 + t - 1) / t;
                    zc::u32* nk = ekeys[level & 1] + 2 * gr.slot0;
                    zc::u32* nr = erecs[level & 1] + 2 * gr.slot0 * zc::MSM_RAW_WORDS;
                    if (nl <= (size_t)ZC_MSM_EDGES_QUAD)       // four lanes per run: the level is a few dependent additions on a fraction of the chip
                        hipLaunchKernelGGL(zc::k_msm_runs_edges_quad, dim3(grid_for(4 * nl)), dim3(zc::ZC_BLOCK), 0, st, lk, lr, (zc::u32)len, (zc::u32)t, (zc::u32)nb,
                                           buckets, present, nk, nr);
                    else
                        hipLaunchKernelGGL(zc::k_msm_runs_edges, dim3(grid_for(nl)), dim3(zc::ZC_BLOCK), 0, st, lk, lr, (zc::u32)len, (zc::u32)t, (zc::u32)nb,
                                           buckets, present, nk, nr);
                    if (nl <= 1) break;                    // one lane saw the whole list: nothing is left open
                    if (level > 40) return fail(ZC_ERR_HIP, "zc_msm: segmented reduction did not converge");
                    lk = nk;
                    lr = nr;
                    len = 2 * nl;
                }
            }
            // bucket reduction: one lane per segment -> sum_j (first' + j + 1) B_(first + j), the product by first' included
            u64* cur = seg_out + 20 * seg_off[g];
            u64* nxt = fold_b + 20 * seg_off[g];
            // few segments (the lowest group, small shards): four lanes per segment, three multiplication latencies per addition
            const size_t quad_max = (size_t)ZC_MSM_SEG_QUAD * (G > 1 && g == G - 1 ? 2 : 1);     // (the exposed chain: lanes for latency)
            if (nsegg <= quad_max)
                hipLaunchKernelGGL(zc::k_msm_segments_quad, dim3((unsigned)((nsegg + 63) / 64)), dim3(zc::ZC_BLOCK), 0, st, (const zc::u32*)(buckets + b0 * zc::MSM_RAW_WORDS),
                                   (const uint8_t*)(present + b0), cur, nsegg, c, seg);
            else
                hipLaunchKernelGGL(zc::k_msm_segments, dim3(grid_for(nsegg)), dim3(zc::ZC_BLOCK), 0, st, (const zc::u32*)(buckets + b0 * zc::MSM_RAW_WORDS),
                                   (const uint8_t*)(present + b0), cur, nsegg, c, seg);
            // fold every window's segment sums (a power of two per window) to one point per window:
            // one workgroup per group of up to 128 points (four lanes per addition) or 512, two launches
            size_t left = nsegg;
            while (left > (size_t)gr.nw) {
#if ZC_MSM_FOLD_QUAD
                const size_t fg = std::min<size_t>(128, left / (size_t)gr.nw);
                hipLaunchKernelGGL(zc::k_msm_fold_groups_quad, dim3((unsigned)(left / fg)), dim3(zc::ZC_BLOCK), 0, st, (const u64*)cur, nxt, (zc::u32)fg);
#else
                const size_t fg = std::min<size_t>(512, left / (size_t)gr.nw);
                hipLaunchKernelGGL(zc::k_msm_fold_groups, dim3((unsigned)(left / fg)), dim3(zc::ZC_BLOCK), 0, st, (const u64*)cur, nxt, (zc::u32)fg);
#endif
                left /= fg;
                std::swap(cur, nxt);
            }
            // Horner's rule, top window first across the groups: this group continues from the result of the group above it
            // (same stream, or -- the lowest group -- behind that stream's event)
            if (g == G - 1 && G > 1 && ZC_MSM_TAIL_SIDE) HIP_TRY(hipStreamWaitEvent(st, D.ev_grp_done[G - 2], 0));
            // The lowest group's stretch of the rule is on the call's critical path: it takes the result of the groups above ALREADY
            // multiplied by 2^(c nw) -- k_msm_shift runs behind the second-lowest group's stretch on the side stream, beside the
            // lowest group's bucket sums (slot G of grp_out) -- and adds it last.
            const bool preshift = ZC_MSM_CARRY_PRESHIFT && G > 1 && ZC_MSM_TAIL_SIDE;
            const bool lowest = g == G - 1;
            const u64* carry = g == 0 ? nullptr : (lowest && preshift) ? grp_out + 20 * (size_t)G : grp_out + 20 * (size_t)(g - 1);
            hipLaunchKernelGGL(zc::k_msm_window_combine, dim3(1), dim3(64), 0, st, (const u64*)cur, grp_out + 20 * (size_t)g, gr.nw, c, carry, lowest && preshift ? 1 : 0);
            if (preshift && g == G - 2)
                hipLaunchKernelGGL(zc::k_msm_shift, dim3(1), dim3(64), 0, st, (const u64*)(grp_out + 20 * (size_t)g), grp_out + 20 * (size_t)G, c * grp[G - 1].nw);
            if (st != D.s()) HIP_TRY(hipEventRecord(D.ev_grp_done[g], st));
        }
        *result = grp_out + 20 * (size_t)(G - 1);
        HIP_TRY(hipGetLastError());
    }
    return ZC_OK;
}

// ---------------------------------------------------------------- RCCL (opened on demand)
RcclApi g_rccl;
std::mutex g_rccl_mu;

int rccl_load()
{
    std::lock_guard<std::mutex> lock(g_rccl_mu);
    if (g_rccl.handle) return ZC_OK;
    // an RCCL already mapped by the host application (PyTorch ships one) is shared, not duplicated
    const char* names[] = {getenv("ZC_RCCL_PATH"), "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* nm : names)
        if (nm && !h) h = dlopen(nm, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
    for (const char* nm : names)
        if (nm && !h) h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
    if (!h) return fail(ZC_ERR_HIP, "librccl.so not found (set ZC_RCCL_PATH)");
    RcclApi api;
    api.handle = h;
    api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    api.CommInitRank = (decltype(api.CommInitRank))dlsym(h, "ncclCommInitRank");
    api.CommDestroy = (decltype(api.CommDestroy))dlsym(h, "ncclCommDestroy");
    api.CommCount = (decltype(api.CommCount))dlsym(h, "ncclCommCount");
    api.AllGather = (decltype(api.AllGather))dlsym(h, "ncclAllGather");
    api.GetErrorString = (decltype(api.GetErrorString))dlsym(h, "ncclGetErrorString");
    if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.CommCount || !api.AllGather || !api.GetErrorString)
        return fail(ZC_ERR_HIP, "librccl.so lacks an expected symbol");
    g_rccl = api;
    return ZC_OK;
}
int rccl_fail(const char* what, ncclResult_t r)
{
    g_last_error = std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "RCCL error");
    return ZC_ERR_HIP;
}
#define RCCL_TRY(expr)                                         \
    do {                                                       \
        ncclResult_t r_ = (expr);                              \
        if (r_ != ncclSuccess) return rccl_fail(#expr, r_);    \
    } while (0)

DevState* dev_state_of(zc_ctx* ctx, int device)
{
    for (auto& d : ctx->devs)
        if (d.device == device) return &d;
    return nullptr;
}

// One device's shard of an MSM, inputs host (staged) or device (in place); the 160-byte sum stays
// in device memory (*result), everything enqueued on ds.s().
int msm_shard(DevState& ds, const uint64_t* points, const uint64_t* scalars, size_t cnt, bool on_device, const u64** result)
{
    if (int rc = ring_check(ds)) return rc;                  // an asynchronous failure of an earlier windowed-core call surfaces here too
    HIP_TRY(hipSetDevice(ds.device));
    const u64 *dP = points, *dK = scalars;
    if (!on_device) {
        int rc = ensure(&ds.scratch[0], &ds.scratch_bytes[0], cnt * 160);
        if (rc) return rc;
        rc = ensure(&ds.scratch[1], &ds.scratch_bytes[1], cnt * 40);
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(ds.scratch[0], points, cnt * 160, hipMemcpyHostToDevice, ds.s()));
        HIP_TRY(hipMemcpyAsync(ds.scratch[1], scalars, cnt * 40, hipMemcpyHostToDevice, ds.s()));
        dP = (const u64*)ds.scratch[0];
        dK = (const u64*)ds.scratch[1];
    }
    return msm_on_device(ds, dP, dK, cnt, result);
}

const uint64_t IDENT_POINT[20] = {0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0};

}  // namespace

// =============================================================================== C ABI
extern "C" {

// ZC_SRC_HASH: sha256 of the kernel sources + the ABI header, passed in by dusk_zerocaf_amd/build.py;
// it ties a profile (profiles/roofline_inputs.json) to the build it was taken on
#ifndef ZC_SRC_HASH
#define ZC_SRC_HASH "unknown"
#endif
const char* zc_version(void) { return "zerocaf_hip 0.5 (gfx950, radix-2^29 Montgomery R=2^261) src:" ZC_SRC_HASH; }
const char* zc_last_error(void) { return g_last_error.c_str(); }

int zc_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

int zc_ctx_create(const int* devices, int ndev, zc_ctx** out)
{
    if (!out) return fail(ZC_ERR_BAD_ARG, "null out");
    *out = nullptr;
    int avail = zc_device_count();
    if (avail <= 0) return fail(ZC_ERR_NO_DEVICE, "no HIP device visible (this library has no CPU fallback)");
    std::vector<int> ids;
    if (!devices || ndev <= 0) {
        int cur = 0;
        HIP_TRY(hipGetDevice(&cur));
        ids.push_back(cur);
    } else {
        for (int i = 0; i < ndev; i++) {
            if (devices[i] < 0 || devices[i] >= avail) return fail(ZC_ERR_BAD_ARG, "device index out of range");
            ids.push_back(devices[i]);
        }
    }
    zc_ctx* ctx = new zc_ctx();
    const Tuning tune = tuning_from_env();                   // the only place the library reads its knobs
    for (int id : ids) {
        DevState ds;
        ds.device = id;
        ds.tune = tune;
        hipError_t e = hipSetDevice(id);
        if (e == hipSuccess) e = hipDeviceGetAttribute(&ds.cus, hipDeviceAttributeMultiprocessorCount, id);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&ds.stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&ds.copy_in, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&ds.copy_out, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&ds.ev_order, hipEventDisableTiming);
        if (e == hipSuccess) {
            int lo_p = 0, hi_p = 0;
            (void)hipDeviceGetStreamPriorityRange(&lo_p, &hi_p);         // lowest, highest (numerically smaller = higher)
            e = hipStreamCreateWithPriority(&ds.aux, hipStreamNonBlocking, lo_p);
        }
        if (e == hipSuccess) e = hipEventCreateWithFlags(&ds.ev_fork, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&ds.ev_join, hipEventDisableTiming);
        if (e == hipSuccess) {
            int lo_p = 0, hi_p = 0;
            (void)hipDeviceGetStreamPriorityRange(&lo_p, &hi_p);
            e = hipStreamCreateWithPriority(&ds.grp, hipStreamNonBlocking, ZC_MSM_TAIL_PRIO ? hi_p : 0);
        }
        for (int g = 0; g < 3 && e == hipSuccess; g++) {
            e = hipEventCreateWithFlags(&ds.ev_grp_go[g], hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&ds.ev_grp_done[g], hipEventDisableTiming);
        }
        if (e != hipSuccess) {
            delete ctx;
            return fail(ZC_ERR_HIP, "stream creation", e);
        }
        ctx->devs.push_back(ds);
    }
    (void)hipSetDevice(ids[0]);
    *out = ctx;
    return ZC_OK;
}

int zc_ctx_device(zc_ctx* ctx, int slot)
{
    if (!ctx || slot < 0 || slot >= (int)ctx->devs.size()) return fail(ZC_ERR_BAD_ARG, "zc_ctx_device: no such device slot");
    return ctx->devs[(size_t)slot].device;
}
int zc_ctx_device_count(zc_ctx* ctx) { return ctx ? (int)ctx->devs.size() : fail(ZC_ERR_BAD_ARG, "null context"); }

int zc_ctx_destroy(zc_ctx* ctx)
{
    if (!ctx) return ZC_OK;
    if (ctx->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(ctx->comm);
    for (auto& ds : ctx->devs) {
        (void)hipSetDevice(ds.device);
        (void)hipStreamSynchronize(ds.s());
        if (ds.part) (void)hipFree(ds.part);
        if (ds.ev_order) (void)hipEventDestroy(ds.ev_order);
        if (ds.ev_fork) (void)hipEventDestroy(ds.ev_fork);
        if (ds.ev_join) (void)hipEventDestroy(ds.ev_join);
        if (ds.aux) (void)hipStreamDestroy(ds.aux);
        if (ds.grp) (void)hipStreamSynchronize(ds.grp), (void)hipStreamDestroy(ds.grp);
        for (int g = 0; g < 3; g++) {
            if (ds.ev_grp_go[g]) (void)hipEventDestroy(ds.ev_grp_go[g]);
            if (ds.ev_grp_done[g]) (void)hipEventDestroy(ds.ev_grp_done[g]);
        }
        for (int a = 0; a < MAX_ARGS; a++)
            if (ds.scratch[a]) (void)hipFree(ds.scratch[a]);
        for (int a = 0; a < 2; a++)
            if (ds.tmp[a]) (void)hipFree(ds.tmp[a]);
        if (ds.bal) (void)hipFree(ds.bal);
        if (ds.msm) (void)hipFree(ds.msm);
        if (ds.fast) (void)hipFree(ds.fast);
        if (ds.ring) (void)hipFree(ds.ring);
        if (ds.ring_err) (void)hipHostFree((void*)ds.ring_err);
        if (ds.base_table) (void)hipFree(ds.base_table);
        if (ds.odd_table) (void)hipFree(ds.odd_table);
        for (hipEvent_t e : ds.ev) (void)hipEventDestroy(e);
        if (ds.copy_in) (void)hipStreamDestroy(ds.copy_in);
        if (ds.copy_out) (void)hipStreamDestroy(ds.copy_out);
        if (ds.stream) (void)hipStreamDestroy(ds.stream);
    }
    delete ctx;
    return ZC_OK;
}

// Switching the launch stream of a device slot: everything already enqueued on the old stream
// (including the producers of shared scratch: window tables, MSM workspace, the basepoint table)
// is ordered before later work on the new one with an event, without a host synchronisation.
int zc_ctx_set_stream_dev(zc_ctx* ctx, int slot, void* hip_stream, int external)
{
    if (!ctx) return fail(ZC_ERR_BAD_ARG, "null context");
    if (slot < 0 || (size_t)slot >= ctx->devs.size()) return fail(ZC_ERR_BAD_ARG, "device slot out of range");
    std::lock_guard<std::mutex> lock(ctx->mu);
    DevState& ds = ctx->devs[slot];
    hipStream_t next = external ? (hipStream_t)hip_stream : ds.stream;
    if (ds.s() == next && ds.use_borrowed == (external != 0)) return ZC_OK;
    HIP_TRY(hipSetDevice(ds.device));
    HIP_TRY(hipEventRecord(ds.ev_order, ds.s()));
    HIP_TRY(hipStreamWaitEvent(next, ds.ev_order, 0));
    ds.borrowed = external ? (hipStream_t)hip_stream : nullptr;
    ds.use_borrowed = external != 0;
    return ZC_OK;
}
int zc_ctx_set_stream(zc_ctx* ctx, void* hip_stream, int external) { return zc_ctx_set_stream_dev(ctx, 0, hip_stream, external); }

// Pin a caller-owned host buffer (hipHostRegister): copies from / to it are truly asynchronous
// and skip the runtime's bounce buffers.  Worth it for buffers reused across calls.
int zc_host_register(void* ptr, size_t bytes)
{
    if (!ptr || !bytes) return fail(ZC_ERR_BAD_ARG, "zc_host_register: null buffer");
    HIP_TRY(hipHostRegister(ptr, bytes, hipHostRegisterDefault));
    return ZC_OK;
}
int zc_host_unregister(void* ptr)
{
    if (!ptr) return fail(ZC_ERR_BAD_ARG, "zc_host_unregister: null buffer");
    HIP_TRY(hipHostUnregister(ptr));
    return ZC_OK;
}

int zc_ctx_synchronize(zc_ctx* ctx)
{
    if (!ctx) return fail(ZC_ERR_BAD_ARG, "null context");
    for (auto& ds : ctx->devs) {
        HIP_TRY(hipSetDevice(ds.device));
        HIP_TRY(hipStreamSynchronize(ds.s()));
        const int rc = ring_check(ds);
        if (rc) return rc;
    }
    return ZC_OK;
}

// ---- FieldElement
int zc_fe_add(zc_ctx* c, const uint64_t* a, const uint64_t* b, uint64_t* o, size_t n) { return binop(c, zc::k_fe_add, zc::k_fe_add_stream, a, b, o, n, 40); }
int zc_fe_sub(zc_ctx* c, const uint64_t* a, const uint64_t* b, uint64_t* o, size_t n) { return binop(c, zc::k_fe_sub, zc::k_fe_sub_stream, a, b, o, n, 40); }
// Mul / Square: LDS-staged beyond the Infinity Cache since round 5 -- with the one-pass product (151 / 115 multiply-adds instead of
// 270 / 234) the staged kernels' coalesced traffic pays: 2^24 elements, same box, mul 0.411 -> 0.376 ms, square 0.273 -> 0.248
// (with the two Montgomery passes of rounds 1-4 the staged kernels were the slower ones: 0.435 against 0.360 ms).
#ifndef ZC_MULSQ_STAGED
#define ZC_MULSQ_STAGED 1            // 0: A/B build with per-lane accesses at every size
#endif
#ifndef ZC_MULSQ_STAGED_MIN_BYTES
#define ZC_MULSQ_STAGED_MIN_BYTES STREAM_BYTES
#endif
// Neg: one input array, no arithmetic to speak of.  Round 6 A/B at 2^24 / 2^26 decides whether the staged form ships
// (profiles/r06_experiments/neg_staged_ab.md); a kernel without a launch site is not kept.
#ifndef ZC_NEG_STAGED
#define ZC_NEG_STAGED 1
#endif
int zc_fe_mul(zc_ctx* c, const uint64_t* a, const uint64_t* b, uint64_t* o, size_t n) { return binop(c, zc::k_fe_mul, ZC_MULSQ_STAGED ? zc::k_fe_mul_stream : nullptr, a, b, o, n, 40, ZC_MULSQ_STAGED_MIN_BYTES); }
int zc_fe_neg(zc_ctx* c, const uint64_t* a, uint64_t* o, size_t n) { return unop(c, zc::k_fe_neg, a, o, n, 40, ZC_NEG_STAGED ? zc::k_fe_neg_stream : nullptr); }
int zc_fe_square(zc_ctx* c, const uint64_t* a, uint64_t* o, size_t n) { return unop(c, zc::k_fe_square, a, o, n, 40, ZC_MULSQ_STAGED ? zc::k_fe_square_stream : nullptr, ZC_MULSQ_STAGED_MIN_BYTES); }

// Montgomery's trick shares one inversion among the c consecutive elements of a lane (3 multiplications per
// element + one inversion per lane).  c = cnt / INV_LANES_TARGET keeps that many lanes busy, capped at 64;
// below 2 the kernels take one element per lane.  ZC_INV_CHUNK=c overrides (tuning, tests).
constexpr size_t INV_LANES_TARGET = 65536;
inline size_t inv_chunk(size_t cnt, const Tuning& tune)
{
    if (tune.inv_chunk) return (size_t)tune.inv_chunk;
    size_t c = cnt / INV_LANES_TARGET;
    return c > 32 ? 32 : c;
}

int zc_fe_invert(zc_ctx* ctx, const uint64_t* a, uint64_t* out, uint8_t* ok, size_t n)
{
    REQUIRE(a); REQUIRE(out);
    Arg args[3] = {in_arg(a, 40), out_arg(out, 40), out_arg(ok, 1)};
    return run_batched(ctx, args, 3, n, [&](void** d, size_t cnt, DevState& D) {
        const size_t c = inv_chunk(cnt, D.tune);
        if (c < 2 || d[0] == d[1]) {                       // tiny batch or in-place: one element per lane
            hipLaunchKernelGGL(zc::k_fe_invert, dim3(grid_for(cnt)), dim3(zc::ZC_BLOCK), 0, D.s(), (const u64*)d[0], (u64*)d[1], (uint8_t*)d[2], cnt);
        } else {
            const size_t lanes = (cnt + c - 1) / c;
            // at most one wave per SIMD (2^20 elements: BASELINE configs[1]): the independent-chain multiplier, -7 %
            hipLaunchKernelGGL(lanes <= (size_t)D.cus * 256 ? zc::k_fe_invert_chunked_lone : zc::k_fe_invert_chunked, dim3(grid_for(lanes)), dim3(zc::ZC_BLOCK), 0, D.s(),
                               (const u64*)d[0], (u64*)d[1], (uint8_t*)d[2], cnt, (int)c);
        }
    });
}
int zc_fe_div(zc_ctx* ctx, const uint64_t* a, const uint64_t* b, uint64_t* out, uint8_t* ok, size_t n)
{
    REQUIRE(a); REQUIRE(b); REQUIRE(out);
    Arg args[4] = {in_arg(a, 40), in_arg(b, 40), out_arg(out, 40), out_arg(ok, 1)};
    return run_batched(ctx, args, 4, n, [&](void** d, size_t cnt, DevState& D) {
        const size_t c = inv_chunk(cnt, D.tune);
        if (c < 2 || d[2] == d[0] || d[2] == d[1]) {
            hipLaunchKernelGGL(zc::k_fe_div, dim3(grid_for(cnt)), dim3(zc::ZC_BLOCK), 0, D.s(), (const u64*)d[0], (const u64*)d[1], (u64*)d[2], (uint8_t*)d[3], cnt);
        } else {
            const size_t lanes = (cnt + c - 1) / c;
            hipLaunchKernelGGL(lanes <= (size_t)D.cus * 256 ? zc::k_fe_div_chunked_lone : zc::k_fe_div_chunked, dim3(grid_for(lanes)), dim3(zc::ZC_BLOCK), 0, D.s(),
                               (const u64*)d[0], (const u64*)d[1], (u64*)d[2], (uint8_t*)d[3], cnt, (int)c);
        }
    });
}
int zc_fe_half(zc_ctx* c, const uint64_t* a, uint64_t* o, size_t n) { return unop(c, zc::k_fe_half, a, o, n, 40); }
int zc_fe_pow(zc_ctx* c, const uint64_t* a, const uint64_t* e, uint64_t* o, size_t n) { return binop(c, zc::k_fe_pow, nullptr, a, e, o, n, 40); }
static int fe_flag_op(zc_ctx* ctx, void (*k)(const u64*, uint8_t*, size_t), const uint64_t* a, uint8_t* out, size_t n)
{
    REQUIRE(a); REQUIRE(out);
    Arg args[2] = {in_arg(a, 40), out_arg(out, 1)};
    return run_batched(ctx, args, 2, n, [&](void** d, size_t cnt, DevState& D) {
        hipLaunchKernelGGL(k, dim3(grid_for(cnt)), dim3(zc::ZC_BLOCK), 0, D.s(), (const u64*)d[0], (uint8_t*)d[1], cnt);
    });
}
// ZC_JACOBI_ROUNDS=r (tests): rounds of 30 positive division steps before a lane falls back to the exponentiation
int zc_fe_legendre_symbol(zc_ctx* ctx, const uint64_t* a, uint8_t* out, size_t n)
{
    REQUIRE(a); REQUIRE(out);
    Arg args[2] = {in_arg(a, 40), out_arg(out, 1)};
    return run_batched(ctx, args, 2, n, [&](void** d, size_t cnt, DevState& D) {
        hipLaunchKernelGGL(zc::k_fe_legendre, dim3(grid_for(cnt)), dim3(zc::ZC_BLOCK), 0, D.s(), (const u64*)d[0], (uint8_t*)d[1], cnt, D.tune.jacobi_rounds >= 0 ? D.tune.jacobi_rounds : zc::JACOBI_MAX_ROUNDS);
    });
}
int zc_fe_is_positive(zc_ctx* c, const uint64_t* a, uint8_t* o, size_t n) { return fe_flag_op(c, zc::k_fe_is_positive, a, o, n); }
int zc_fe_mod_sqrt(zc_ctx* ctx, const uint64_t* a, int sign, uint64_t* out, uint8_t* ok, size_t n)
{
    REQUIRE(a); REQUIRE(out);
    Arg args[3] = {in_arg(a, 40), out_arg(out, 40), out_arg(ok, 1)};
    return run_batched(ctx, args, 3, n, [&](void** d, size_t cnt, DevState& D) {
        hipLaunchKernelGGL(zc::k_fe_mod_sqrt, dim3(grid_for(cnt)), dim3(zc::ZC_BLOCK), 0, D.s(), (const u64*)d[0], sign, (u64*)d[1], (uint8_t*)d[2], cnt);
    });
}
int zc_fe_from_bytes(zc_ctx* ctx, const uint8_t* in32, uint64_t* out, size_t n)
{
    REQUIRE(in32); REQUIRE(out);
    Arg args[2] = {in_arg(in32, 32), out_arg(out, 40)};
    return run_batched(ctx, args, 2, n, [&](void** d, size_t cnt, DevState& D) {
        hipLaunchKernelGGL(zc::k_from_bytes, dim3(grid_for(cnt)), dim3(zc::ZC_BLOCK), 0, D.s(), (const uint8_t*)d[0], (u64*)d[1], (uint8_t*)nullptr, 0, cnt);
    });
}
int zc_fe_to_bytes(zc_ctx* ctx, const uint64_t* in, uint8_t* out32, size_t n)
{
    REQUIRE(in); REQUIRE(out32);
    Arg args[2] = {in_arg(in, 40), out_arg(out32, 32)};
    return run_batched(ctx, args, 2, n, [&](void** d, size_t cnt, DevState& D) {
        hipLaunchKernelGGL(zc::k_to_bytes, dim3(grid_for(cnt)), dim3(zc::ZC_BLOCK), 0, D.s(), (const u64*)d[0], (uint8_t*)d[1], cnt);
    });
}
int zc_fe_sqrt_ratio_i(zc_ctx* ctx, const uint64_t* u, const uint64_t* v, uint64_t* out, uint8_t* was_square, size_t n)
{
    REQUIRE(u); REQUIRE(v); REQUIRE(out);
    Arg args[4] = {in_arg(u, 40), in_arg(v, 40), out_arg(out, 40), out_arg(was_square, 1)};
    return run_batched(ctx, args, 4, n, [&](void** d, size_t cnt, DevState& D) {
        hipLaunchKernelGGL(zc::k_fe_sqrt_ratio_i, dim3(grid_for(cnt)), dim3(zc::ZC_BLOCK), 0, D.s(), (const u64*)d[0], (const u64*)d[1], (u64*)d[2], (uint8_t*)d[3], cnt);
    });
}
int zc_fe_inv_sqrt(zc_ctx* ctx, const uint64_t* a, uint64_t* out, uint8_t* was_square, size_t n)
{
    REQUIRE(a); REQUIRE(out);
    Arg args[3] = {in_arg(a, 40), out_arg(out, 40), out_arg(was_square, 1)};
    return run_batched(ctx, args, 3, n, [&](void** d, size_t cnt, DevState& D) {
        hipLaunchKernelGGL(zc::k_fe_inv_sqrt, dim3(grid_for(cnt)), dim3(zc::ZC_BLOCK), 0, D.s(), (const u64*)d[0], (u64*)d[1], (uint8_t*)d[2], cnt);
    });
}

// ---- Scalar
int zc_sc_add(zc_ctx* c, const uint64_t* a, const uint64_t* b, uint64_t* o, size_t n) { return binop(c, zc::k_sc_add, zc::k_sc_add_stream, a, b, o, n, 40); }
int zc_sc_sub(zc_ctx* c, const uint64_t* a, const uint64_t* b, uint64_t* o, size_t n) { return binop(c, zc::k_sc_sub, zc::k_sc_sub_stream, a, b, o, n, 40); }
int zc_sc_mul(zc_ctx* c, const uint64_t* a, const uint64_t* b, uint64_t* o, size_t n) { return binop(c, zc::k_sc_mul, ZC_MULSQ_STAGED ? zc::k_sc_mul_stream : nullptr, a, b, o, n, 40, ZC_MULSQ_STAGED_MIN_BYTES); }
int zc_sc_neg(zc_ctx* c, const uint64_t* a, uint64_t* o, size_t n) { return unop(c, zc::k_sc_neg, a, o, n, 40, ZC_NEG_STAGED ? zc::k_sc_neg_stream : nullptr); }
int zc_sc_square(zc_ctx* c, const uint64_t* a, uint64_t* o, size_t n) { return unop(c, zc::k_sc_square, a, o, n, 40, ZC_MULSQ_STAGED ? zc::k_sc_square_stream : nullptr, ZC_MULSQ_STAGED_MIN_BYTES); }
// S-x rows: the Scalar operations beside the default scalar-mul path
int zc_sc_half(zc_ctx* c, const uint64_t* a, uint64_t* o, size_t n) { return unop(c, zc::k_sc_half, a, o, n, 40); }
int zc_sc_pow(zc_ctx* c, const uint64_t* a, const uint64_t* e, uint64_t* o, size_t n) { return binop(c, zc::k_sc_pow, nullptr, a, e, o, n, 40); }
int zc_sc_shr(zc_ctx* ctx, const uint64_t* a, unsigned shift, uint64_t* out, size_t n)
{
    REQUIRE(a); REQUIRE(out);
    if (shift > 255) return fail(ZC_ERR_BAD_ARG, "zc_sc_shr: the reference shifts by a u8");
    Arg args[2] = {in_arg(a, 40), out_arg(out, 40)};
    return run_batched(ctx, args, 2, n, [&](void** d, size_t cnt, DevState& D) {
        hipLaunchKernelGGL(zc::k_sc_shr, dim3(grid_for(cnt)), dim3(zc::ZC_BLOCK), 0, D.s(), (const u64*)d[0], (zc::u32)shift, (u64*)d[1], cnt);
    });
}
int zc_sc_into_bits(zc_ctx* ctx, const uint64_t* a, uint8_t* bits256, size_t n)
{
    REQUIRE(a); REQUIRE(bits256);
    Arg args[2] = {in_arg(a, 40), out_arg(bits256, 256)};
    return run_batched(ctx, args, 2, n, [&](void** d, size_t cnt, DevState& D) {
        hipLaunchKernelGGL(zc::k_sc_into_bits, dim3(grid_for(cnt)), dim3(zc::ZC_BLOCK), 0, D.s(), (const u64*)d[0], (uint8_t*)d[1], cnt);
    });
}
int zc_sc_compute_naf(zc_ctx* ctx, const uint64_t* a, unsigned width, int8_t* naf256, size_t n)
{
    REQUIRE(a); REQUIRE(naf256);
    if (width == 1 || width > 7) return fail(ZC_ERR_BAD_ARG, "zc_sc_compute_naf: width 0 (compute_NAF) or 2..7 (compute_window_NAF: digits are i8)");
    Arg args[2] = {in_arg(a, 40), out_arg(naf256, 256)};
    return run_batched(ctx, args, 2, n, [&](void** d, size_t cnt, DevState& D) {
        hipLaunchKernelGGL(zc::k_sc_compute_naf, dim3(grid_for(cnt)), dim3(zc::ZC_BLOCK), 0, D.s(), (const u64*)d[0], (zc::u32)width, (int8_t*)d[1], cnt);
    });
}
int zc_sc_from_bytes(zc_ctx* ctx, const uint8_t* in32, uint64_t* out, uint8_t* ok, size_t n)
{
    REQUIRE(in32); REQUIRE(out);
    Arg args[3] = {in_arg(in32, 32), out_arg(out, 40), out_arg(ok, 1)};
    return run_batched(ctx, args, 3, n, [&](void** d, size_t cnt, DevState& D) {
        hipLaunchKernelGGL(zc::k_from_bytes, dim3(grid_for(cnt)), dim3(zc::ZC_BLOCK), 0, D.s(), (const uint8_t*)d[0], (u64*)d[1], (uint8_t*)d[2], 1, cnt);
    });
}
int zc_sc_to_bytes(zc_ctx* ctx, const uint64_t* in, uint8_t* out32, size_t n) { return zc_fe_to_bytes(ctx, in, out32, n); }

// ---- EdwardsPoint
// staged records from 2^12 points on (below, a launch is a handful of workgroups and the barriers only cost)
constexpr size_t ED_STAGED_MIN_BYTES = (size_t)160 * 3 << 12;     // binop compares cnt * 160 * 3 against it: above 2^12 points
constexpr size_t ED_STAGED_MIN_BYTES_1 = (size_t)160 * 2 << 12;   // unop compares cnt * 160 * 2: the same 2^12 points for double / neg
int zc_ed_add(zc_ctx* c, const uint64_t* p, const uint64_t* q, uint64_t* o, size_t n) { return binop(c, zc::k_ed_add, zc::k_ed_add_staged, p, q, o, n, 160, ED_STAGED_MIN_BYTES); }
int zc_ed_sub(zc_ctx* c, const uint64_t* p, const uint64_t* q, uint64_t* o, size_t n) { return binop(c, zc::k_ed_sub, zc::k_ed_sub_staged, p, q, o, n, 160, ED_STAGED_MIN_BYTES); }
int zc_ed_double(zc_ctx* c, const uint64_t* p, uint64_t* o, size_t n) { return unop(c, zc::k_ed_double, p, o, n, 160, zc::k_ed_double_staged, ED_STAGED_MIN_BYTES_1); }
int zc_ed_neg(zc_ctx* c, const uint64_t* p, uint64_t* o, size_t n) { return unop(c, zc::k_ed_neg, p, o, n, 160, zc::k_ed_neg_staged, ED_STAGED_MIN_BYTES_1); }

int zc_ed_scalar_mul(zc_ctx* ctx, const uint64_t* p, const uint64_t* k, uint64_t* out, size_t n, unsigned flags)
{
    if (flags == ZC_SCALAR_MUL_STRICT) return scalar_mul_impl(ctx, p, k, out, n);
    if (flags == ZC_SCALAR_MUL_FAST) {
        REQUIRE(p); REQUIRE(k); REQUIRE(out);
        Arg args[3] = {in_arg(p, 160), in_arg(k, 40), out_arg(out, 160)};
        int inner = ZC_OK;
        int rc = run_batched(ctx, args, 3, n, [&](void** d, size_t cnt, DevState& D) {
            inner = fast_ring(D, cnt, [&](zc::u32* table, zc::u32* ring, zc::u32 slots, size_t off, size_t c) {
                hipLaunchKernelGGL(zc::k_ed_scalar_mul_fast, dim3(grid_for(c)), dim3(zc::ZC_BLOCK), 0, D.s(), (const u64*)d[0] + 20 * off,
                                   (const u64*)d[1] + 5 * off, (zc::u32)5, (u64*)d[2] + 20 * off, table, ring, slots, (zc::u32)c);
            });
        }, true);
        return rc ? rc : inner;
    }
    if (flags == ZC_SCALAR_MUL_LTR_BIN) return binop(ctx, zc::k_ed_scalar_mul_ltr_bin, nullptr, p, k, out, n, 0);
    if (flags == ZC_SCALAR_MUL_BINARY_NAF) return binop(ctx, zc::k_ed_scalar_mul_naf, nullptr, p, k, out, n, 0);
    return fail(ZC_ERR_BAD_ARG, "unknown scalar_mul flags");
}
int zc_ed_mul_by_pow_2(zc_ctx* ctx, const uint64_t* p, uint64_t kexp, uint64_t* out, size_t n)
{
    if (kexp >= 250) return fail(ZC_ERR_BAD_ARG, "Exponent can't be greater than the sub-group order");   // scalar.rs:531
    uint64_t k[5] = {0, 0, 0, 0, 0};
    k[kexp / 52] = 1ull << (kexp % 52);                  // Scalar::two_pow_k, scalar.rs:525-552
    return scalar_mul_bcast(ctx, p, k, out, n);
}
int zc_ed_mul_by_cofactor(zc_ctx* ctx, const uint64_t* p, uint64_t* out, size_t n)
{
    return zc_ed_mul_by_pow_2(ctx, p, 3, out, n);         // Scalar::from(8u8), edwards.rs:174-179
}
int zc_ed_to_affine(zc_ctx* ctx, const uint64_t* p, uint64_t* xy, uint8_t* ok, size_t n)
{
    REQUIRE(p); REQUIRE(xy);
    Arg args[3] = {in_arg(p, 160), out_arg(xy, 80), out_arg(ok, 1)};
    return run_batched(ctx, args, 3, n, [&](void** d, size_t cnt, DevState& D) {
        const size_t c = inv_chunk(cnt, D.tune);
        if (c < 2) {
            hipLaunchKernelGGL(zc::k_ed_to_affine, dim3(grid_for(cnt)), dim3(zc::ZC_BLOCK), 0, D.s(), (const u64*)d[0], (u64*)d[1], (uint8_t*)d[2], cnt);
        } else {
            const size_t lanes = (cnt + c - 1) / c;
            hipLaunchKernelGGL(zc::k_ed_to_affine_chunked, dim3(grid_for(lanes)), dim3(zc::ZC_BLOCK), 0, D.s(), (const u64*)d[0], (u64*)d[1], (uint8_t*)d[2], cnt, (int)c);
        }
    });
}
int zc_ed_eq(zc_ctx* ctx, const uint64_t* p, const uint64_t* q, uint8_t* eq, size_t n)
{
    REQUIRE(p); REQUIRE(q); REQUIRE(eq);
    Arg args[3] = {in_arg(p, 160), in_arg(q, 160), out_arg(eq, 1)};
    return run_batched(ctx, args, 3, n, [&](void** d, size_t cnt, DevState& D) {
        hipLaunchKernelGGL(zc::k_ed_eq, dim3(grid_for(cnt)), dim3(zc::ZC_BLOCK), 0, D.s(), (const u64*)d[0], (const u64*)d[1], (uint8_t*)d[2], cnt);
    });
}
int zc_ed_compress(zc_ctx* ctx, const uint64_t* p, uint8_t* out32, uint8_t* ok, size_t n)
{
    REQUIRE(p); REQUIRE(out32);
    Arg args[3] = {in_arg(p, 160), out_arg(out32, 32), out_arg(ok, 1)};
    return run_batched(ctx, args, 3, n, [&](void** d, size_t cnt, DevState& D) {
        hipLaunchKernelGGL(zc::k_ed_compress, dim3(grid_for(cnt)), dim3(zc::ZC_BLOCK), 0, D.s(), (const u64*)d[0], (uint8_t*)d[1], (uint8_t*)d[2], cnt);
    });
}
int zc_ed_decompress(zc_ctx* ctx, const uint8_t* in32, uint64_t* out, uint8_t* ok, size_t n)
{
    REQUIRE(in32); REQUIRE(out);
    Arg args[3] = {in_arg(in32, 32), out_arg(out, 160), out_arg(ok, 1)};
    return run_batched(ctx, args, 3, n, [&](void** d, size_t cnt, DevState& D) {
        hipLaunchKernelGGL(zc::k_ed_decompress, dim3(grid_for(cnt)), dim3(zc::ZC_BLOCK), 0, D.s(), (const uint8_t*)d[0], (u64*)d[1], (uint8_t*)d[2], cnt);
    });
}

// ---- Ristretto
int zc_ris_compress(zc_ctx* ctx, const uint64_t* p, uint8_t* out32, size_t n)
{
    REQUIRE(p); REQUIRE(out32);
    Arg args[2] = {in_arg(p, 160), out_arg(out32, 32)};
    return run_batched(ctx, args, 2, n, [&](void** d, size_t cnt, DevState& D) {
        hipLaunchKernelGGL(zc::k_ris_compress, dim3(grid_for(cnt)), dim3(zc::ZC_BLOCK), 0, D.s(), (const u64*)d[0], (uint8_t*)d[1], cnt);
    });
}
int zc_ris_decompress(zc_ctx* ctx, const uint8_t* in32, uint64_t* out, uint8_t* ok, size_t n)
{
    REQUIRE(in32); REQUIRE(out);
    Arg args[3] = {in_arg(in32, 32), out_arg(out, 160), out_arg(ok, 1)};
    return run_batched(ctx, args, 3, n, [&](void** d, size_t cnt, DevState& D) {
        hipLaunchKernelGGL(zc::k_ris_decompress, dim3(grid_for(cnt)), dim3(zc::ZC_BLOCK), 0, D.s(), (const uint8_t*)d[0], (u64*)d[1], (uint8_t*)d[2], cnt);
    });
}
int zc_ris_eq(zc_ctx* ctx, const uint64_t* p, const uint64_t* q, uint8_t* eq, size_t n)
{
    REQUIRE(p); REQUIRE(q); REQUIRE(eq);
    Arg args[3] = {in_arg(p, 160), in_arg(q, 160), out_arg(eq, 1)};
    return run_batched(ctx, args, 3, n, [&](void** d, size_t cnt, DevState& D) {
        hipLaunchKernelGGL(zc::k_ris_eq, dim3(grid_for(cnt)), dim3(zc::ZC_BLOCK), 0, D.s(), (const u64*)d[0], (const u64*)d[1], (uint8_t*)d[2], cnt);
    });
}
int zc_ris_roundtrip_mul(zc_ctx* ctx, const uint8_t* in32, const uint64_t* k, uint8_t* out32, uint8_t* ok, size_t n)
{
    REQUIRE(in32); REQUIRE(k); REQUIRE(out32);
    Arg args[4] = {in_arg(in32, 32), in_arg(k, 40), out_arg(out32, 32), out_arg(ok, 1)};
    // The boundary is encodings in / encodings out, which depend only on the group element, so
    // the fast scalar-mul core is used (ZC_RISTRETTO_STRICT=1 runs the reference formula sequence).
    int inner = ZC_OK;
    int rc = run_batched(ctx, args, 4, n, [&](void** d, size_t cnt, DevState& D) {
        if (D.tune.ristretto_strict) {
            hipLaunchKernelGGL(zc::k_ris_roundtrip_mul, dim3(grid_for(cnt)), dim3(zc::ZC_BLOCK), 0, D.s(), (const uint8_t*)d[0], (const u64*)d[1], (uint8_t*)d[2], (uint8_t*)d[3], cnt);
            return;
        }
        inner = fast_ring(D, cnt, [&](zc::u32* table, zc::u32* ring, zc::u32 slots, size_t off, size_t c) {
            hipLaunchKernelGGL(zc::k_ris_roundtrip_mul_fast, dim3(grid_for(c)), dim3(zc::ZC_BLOCK), 0, D.s(), (const uint8_t*)d[0] + 32 * off,
                               (const u64*)d[1] + 5 * off, (uint8_t*)d[2] + 32 * off, d[3] ? (uint8_t*)d[3] + off : (uint8_t*)nullptr, table, ring, slots, (zc::u32)c);
        });
    }, true);
    return rc ? rc : inner;
}

// ---- "next" rows (N3, N4)
int zc_ed_is_valid(zc_ctx* ctx, const uint64_t* p, uint8_t* valid, size_t n)
{
    REQUIRE(p); REQUIRE(valid);
    Arg args[2] = {in_arg(p, 160), out_arg(valid, 1)};
    return run_batched(ctx, args, 2, n, [&](void** d, size_t cnt, DevState& D) {
        hipLaunchKernelGGL(zc::k_ed_is_valid, dim3(grid_for(cnt)), dim3(zc::ZC_BLOCK), 0, D.s(), (const u64*)d[0], (uint8_t*)d[1], cnt);
    });
}
int zc_ris_is_valid(zc_ctx* ctx, const uint64_t* p, uint8_t* valid, size_t n)
{
    REQUIRE(p); REQUIRE(valid);
    Arg args[2] = {in_arg(p, 160), out_arg(valid, 1)};
    return run_batched(ctx, args, 2, n, [&](void** d, size_t cnt, DevState& D) {
        hipLaunchKernelGGL(zc::k_ris_is_valid, dim3(grid_for(cnt)), dim3(zc::ZC_BLOCK), 0, D.s(), (const u64*)d[0], (uint8_t*)d[1], cnt);
    });
}
int zc_ris_elligator(zc_ctx* ctx, const uint64_t* r0, uint64_t* out, size_t n)
{
    REQUIRE(r0); REQUIRE(out);
    Arg args[2] = {in_arg(r0, 40), out_arg(out, 160)};
    return run_batched(ctx, args, 2, n, [&](void** d, size_t cnt, DevState& D) {
        hipLaunchKernelGGL(zc::k_ris_elligator, dim3(grid_for(cnt)), dim3(zc::ZC_BLOCK), 0, D.s(), (const u64*)d[0], (u64*)d[1], cnt);
    });
}
int zc_ris_from_uniform_bytes(zc_ctx* ctx, const uint8_t* in64, uint64_t* out, size_t n)
{
    REQUIRE(in64); REQUIRE(out);
    Arg args[2] = {in_arg(in64, 64), out_arg(out, 160)};
    return run_batched(ctx, args, 2, n, [&](void** d, size_t cnt, DevState& D) {
        hipLaunchKernelGGL(zc::k_ris_from_uniform_bytes, dim3(grid_for(cnt)), dim3(zc::ZC_BLOCK), 0, D.s(), (const uint8_t*)d[0], (u64*)d[1], cnt);
    });
}
int zc_proj_add(zc_ctx* c, const uint64_t* p, const uint64_t* q, uint64_t* o, size_t n) { return binop(c, zc::k_proj_add, nullptr, p, q, o, n, 120); }
int zc_proj_double(zc_ctx* c, const uint64_t* p, uint64_t* o, size_t n) { return unop(c, zc::k_proj_double, p, o, n, 120); }
int zc_proj_to_extended(zc_ctx* ctx, const uint64_t* p, uint64_t* out, size_t n)
{
    REQUIRE(p); REQUIRE(out);
    Arg args[2] = {in_arg(p, 120), out_arg(out, 160)};
    return run_batched(ctx, args, 2, n, [&](void** d, size_t cnt, DevState& D) {
        hipLaunchKernelGGL(zc::k_proj_to_extended, dim3(grid_for(cnt)), dim3(zc::ZC_BLOCK), 0, D.s(), (const u64*)d[0], (u64*)d[1], cnt);
    });
}
// E-x rows: coset4 and the remaining ProjectivePoint operations
int zc_ed_coset4(zc_ctx* ctx, const uint64_t* p, uint64_t* out4, size_t n)
{
    REQUIRE(p); REQUIRE(out4);
    Arg args[2] = {in_arg(p, 160), out_arg(out4, 640)};
    return run_batched(ctx, args, 2, n, [&](void** d, size_t cnt, DevState& D) {
        hipLaunchKernelGGL(zc::k_ed_coset4, dim3(grid_for(cnt)), dim3(zc::ZC_BLOCK), 0, D.s(), (const u64*)d[0], (u64*)d[1], cnt);
    });
}
int zc_proj_neg(zc_ctx* c, const uint64_t* p, uint64_t* o, size_t n) { return unop(c, zc::k_proj_neg, p, o, n, 120); }
int zc_proj_sub(zc_ctx* c, const uint64_t* p, const uint64_t* q, uint64_t* o, size_t n) { return binop(c, zc::k_proj_sub, nullptr, p, q, o, n, 120); }
int zc_proj_eq(zc_ctx* ctx, const uint64_t* p, const uint64_t* q, uint8_t* eq, size_t n)
{
    REQUIRE(p); REQUIRE(q); REQUIRE(eq);
    Arg args[3] = {in_arg(p, 120), in_arg(q, 120), out_arg(eq, 1)};
    return run_batched(ctx, args, 3, n, [&](void** d, size_t cnt, DevState& D) {
        hipLaunchKernelGGL(zc::k_proj_eq, dim3(grid_for(cnt)), dim3(zc::ZC_BLOCK), 0, D.s(), (const u64*)d[0], (const u64*)d[1], (uint8_t*)d[2], cnt);
    });
}
int zc_proj_is_valid(zc_ctx* ctx, const uint64_t* p, uint8_t* valid, size_t n)
{
    REQUIRE(p); REQUIRE(valid);
    Arg args[2] = {in_arg(p, 120), out_arg(valid, 1)};
    return run_batched(ctx, args, 2, n, [&](void** d, size_t cnt, DevState& D) {
        hipLaunchKernelGGL(zc::k_proj_is_valid, dim3(grid_for(cnt)), dim3(zc::ZC_BLOCK), 0, D.s(), (const u64*)d[0], (uint8_t*)d[1], cnt);
    });
}
int zc_proj_scalar_mul(zc_ctx* ctx, const uint64_t* p, const uint64_t* k, uint64_t* out, size_t n)
{
    REQUIRE(p); REQUIRE(k); REQUIRE(out);
    Arg args[3] = {in_arg(p, 120), in_arg(k, 40), out_arg(out, 120)};
    return run_batched(ctx, args, 3, n, [&](void** d, size_t cnt, DevState& D) {
        hipLaunchKernelGGL(zc::k_proj_scalar_mul, dim3(grid_for(cnt)), dim3(zc::ZC_BLOCK), 0, D.s(), (const u64*)d[0], (const u64*)d[1], (u64*)d[2], cnt);
    });
}

// ---- fixed-base multiplication of the basepoint
static int base_table(DevState& D, const zc::u32** table)
{
    if (!D.base_table) {
        int rc = ensure(&D.base_table, &D.base_bytes, (size_t)zc::ZC_BASE_WINDOWS * zc::ZC_BASE_ENTRIES * 128);
        if (rc) return rc;
        hipLaunchKernelGGL(zc::k_base_table_build, dim3(1), dim3(zc::ZC_BASE_ENTRIES), 0, D.s(), (zc::u32*)D.base_table);
        HIP_TRY(hipGetLastError());
    }
    *table = (const zc::u32*)D.base_table;
    return ZC_OK;
}
int zc_ed_mul_base(zc_ctx* ctx, const uint64_t* k, uint64_t* out, size_t n)
{
    REQUIRE(k); REQUIRE(out);
    Arg args[2] = {in_arg(k, 40), out_arg(out, 160)};
    int inner = ZC_OK;
    int rc = run_batched(ctx, args, 2, n, [&](void** d, size_t cnt, DevState& D) {
        const zc::u32* t = nullptr;
        if ((inner = base_table(D, &t)) != ZC_OK) return;
        hipLaunchKernelGGL(zc::k_ed_mul_base, dim3(grid_for(cnt)), dim3(zc::ZC_BLOCK), 0, D.s(), (const u64*)d[0], (u64*)d[1], t, cnt);
    });
    return rc ? rc : inner;
}
int zc_ris_mul_base_compress(zc_ctx* ctx, const uint64_t* k, uint8_t* out32, size_t n)
{
    REQUIRE(k); REQUIRE(out32);
    Arg args[2] = {in_arg(k, 40), out_arg(out32, 32)};
    int inner = ZC_OK;
    int rc = run_batched(ctx, args, 2, n, [&](void** d, size_t cnt, DevState& D) {
        const zc::u32* t = nullptr;
        if ((inner = base_table(D, &t)) != ZC_OK) return;
        hipLaunchKernelGGL(zc::k_ris_mul_base_compress, dim3(grid_for(cnt)), dim3(zc::ZC_BLOCK), 0, D.s(), (const u64*)d[0], (uint8_t*)d[1], t, cnt);
    });
    return rc ? rc : inner;
}

// window_naf_mul (src/edwards.rs:155-171) with its table indexed correctly: see k_ed_mul_base_wnaf
int zc_ed_mul_base_wnaf(zc_ctx* ctx, const uint64_t* k, unsigned width, uint64_t* out, size_t n)
{
    REQUIRE(k); REQUIRE(out);
    if (width < 2 || width > 7) return fail(ZC_ERR_BAD_ARG, "zc_ed_mul_base_wnaf: window width 2..7 (compute_window_NAF's digits are i8)");
    Arg args[2] = {in_arg(k, 40), out_arg(out, 160)};
    int inner = ZC_OK;
    int rc = run_batched(ctx, args, 2, n, [&](void** d, size_t cnt, DevState& D) {
        if (!D.odd_table) {
            if ((inner = ensure(&D.odd_table, &D.odd_bytes, (size_t)zc::ZC_ODD_ENTRIES * 128)) != ZC_OK) return;
            hipLaunchKernelGGL(zc::k_odd_table_build, dim3(1), dim3(128), 0, D.s(), (zc::u32*)D.odd_table);
        }
        hipLaunchKernelGGL(zc::k_ed_mul_base_wnaf, dim3(grid_for(cnt)), dim3(zc::ZC_BLOCK), 0, D.s(), (const u64*)d[0], (zc::u32)width, (u64*)d[1],
                           (const zc::u32*)D.odd_table, cnt);
    });
    return rc ? rc : inner;
}

// ---- MSM: sum_i k_i * P_i (not in the reference; specified as the reference's own
// sum of `&P_i * &k_i`, src/edwards.rs:547-561 + :465-489).  Per GPU: bucket method (zc_msm.hip.h)
// for shards of >= MSM_BUCKET_MIN_N pairs, otherwise batched scalar-mul + pairwise folds.
// The exchange step lives here: per-device partial sums are gathered INTO DEVICE MEMORY
// (hipMemcpyPeerAsync inside one process, ncclAllGather between processes) and folded in
// device / rank order by ONE kernel (k_ed_fold_ordered), so every rank ends with identical limbs.

// partial sums of all device slots -> slot 0's `part` buffer, folded there; result at part[nparts]
static int gather_and_fold(zc_ctx* ctx, const std::vector<DevState*>& used, const std::vector<const u64*>& partial_ptr, const u64** result)
{
    DevState* d0 = used[0];
    const size_t np = used.size();
    if (np == 1) {
        *result = partial_ptr[0];
        return ZC_OK;
    }
    HIP_TRY(hipSetDevice(d0->device));
    int rc = ensure(&d0->part, &d0->part_bytes, (np + 1) * 160);
    if (rc) return rc;
    u64* part = (u64*)d0->part;
    for (size_t ui = 0; ui < np; ui++) {
        DevState* ds = used[ui];
        HIP_TRY(hipSetDevice(ds->device));
        if (ds->device == d0->device) HIP_TRY(hipMemcpyAsync(part + 20 * ui, partial_ptr[ui], 160, hipMemcpyDeviceToDevice, ds->s()));
        else HIP_TRY(hipMemcpyPeerAsync(part + 20 * ui, d0->device, partial_ptr[ui], ds->device, 160, ds->s()));
        if (ds != d0) {
            HIP_TRY(hipEventRecord(ds->ev_order, ds->s()));
            HIP_TRY(hipStreamWaitEvent(d0->s(), ds->ev_order, 0));
        }
    }
    HIP_TRY(hipSetDevice(d0->device));
    hipLaunchKernelGGL(zc::k_ed_fold_ordered, dim3(1), dim3(64), 0, d0->s(), (const u64*)part, np, (const u64*)nullptr, part + 20 * np);
    HIP_TRY(hipGetLastError());
    *result = part + 20 * np;
    return ZC_OK;
}

// local part of an MSM over every device slot of the context (host inputs: contiguous shards, one
// worker thread per slot; device inputs: the owning slot); *result in device memory of *owner
static int msm_local(zc_ctx* ctx, const uint64_t* points, const uint64_t* scalars, size_t n, DevState** owner, const u64** result)
{
    Residency rp, rk;
    int dp = -1, dk = -1;
    residency_of(points, &rp, &dp);
    residency_of(scalars, &rk, &dk);
    if (rp != rk || (rp == RES_DEVICE && dp != dk)) return fail(ZC_ERR_MIXED_MEM, "points/scalars residency differs");
    std::vector<DevState*> used;
    std::vector<const u64*> partial_ptr;
    if (rp == RES_DEVICE) {
        DevState* ds = dev_state_of(ctx, dp);
        if (!ds) return fail(ZC_ERR_MIXED_MEM, "device buffers do not belong to a device of this context");
        const u64* part = nullptr;
        int rc = msm_shard(*ds, points, scalars, n, true, &part);
        if (rc) return rc;
        *owner = ds;
        *result = part;
        return ZC_OK;
    }
    const size_t ndev = ctx->devs.size();
    const size_t per = (n + ndev - 1) / ndev;
    size_t nshards = 0;
    while (nshards < ndev && nshards * per < n) nshards++;
    std::vector<int> rcs(nshards, ZC_OK);
    std::vector<std::string> errs(nshards);
    partial_ptr.assign(nshards, nullptr);
    auto job = [&](size_t di) {
        const size_t lo = di * per, hi = std::min(n, lo + per);
        rcs[di] = msm_shard(ctx->devs[di], points + 20 * lo, scalars + 5 * lo, hi - lo, false, &partial_ptr[di]);
        if (rcs[di]) errs[di] = g_last_error;
    };
    std::vector<std::thread> workers;                       // pageable uploads block their thread: one per device
    for (size_t di = 1; di < nshards; di++) workers.emplace_back(job, di);
    job(0);
    for (auto& t : workers) t.join();
    for (size_t di = 0; di < nshards; di++) {
        if (rcs[di]) {
            g_last_error = errs[di];
            return rcs[di];
        }
        used.push_back(&ctx->devs[di]);
    }
    *owner = used[0];
    return gather_and_fold(ctx, used, partial_ptr, result);
}

int zc_msm(zc_ctx* ctx, const uint64_t* points, const uint64_t* scalars, size_t n, uint64_t* out_point)
{
    if (!ctx) return fail(ZC_ERR_BAD_ARG, "null context");
    REQUIRE(points); REQUIRE(scalars); REQUIRE(out_point);
    if (n == 0) {
        memcpy(out_point, IDENT_POINT, sizeof IDENT_POINT);
        return ZC_OK;
    }
    std::lock_guard<std::mutex> lock(ctx->mu);
    DevState* owner = nullptr;
    const u64* res = nullptr;
    int rc = msm_local(ctx, points, scalars, n, &owner, &res);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(owner->device));
    HIP_TRY(hipMemcpyAsync(out_point, res, 160, hipMemcpyDeviceToHost, owner->s()));
    HIP_TRY(hipStreamSynchronize(owner->s()));
    return ZC_OK;
}

#ifdef ZC_TEST_HOOKS
// Test hook, compiled only into libzerocaf_hip_test.so (-DZC_TEST_HOOKS), NOT part of the ABI (not in include/zerocaf_hip.h,
// not mirrored): the MSM's digit + sort stage alone.
// `scalars` (n x 5 u64) and `out_pairs` (n * ceil(261 / c) pairs of u32: bucket key, point index | sign << 31)
// are DEVICE buffers of ctx's device slot 0; synchronises before returning.
int zc_test_msm_sort(zc_ctx* ctx, const uint64_t* scalars, size_t n, int c, uint32_t* out_pairs)
{
    if (!ctx) return fail(ZC_ERR_BAD_ARG, "null context");
    REQUIRE(scalars); REQUIRE(out_pairs);
    if (c < zc::MSM_MIN_C || c > zc::MSM_MAX_C || n == 0) return fail(ZC_ERR_BAD_ARG, "zc_test_msm_sort: bad window width / empty batch");
    std::lock_guard<std::mutex> lock(ctx->mu);
    DevState& D = ctx->devs[0];
    HIP_TRY(hipSetDevice(D.device));
    const int W = (zc::MSM_SCALAR_BITS + c - 1) / c;
    const size_t m = n * (size_t)W;
    if (m > 0xFFFFFFFFull) return fail(ZC_ERR_BAD_ARG, "zc_test_msm_sort: too many pairs");
    const MsmSortPlan plan = msm_sort_plan(n, c, W, D.tune);
    for (int pass = 0; pass < 2; pass++) {
        Carver cv{pass ? (char*)D.msm : nullptr};
        zc::u32* digits = cv.take<zc::u32>(m);
        uint2* pairs_a = cv.take<uint2>(m);
        void* pairs_b = plan.passes == 1 ? nullptr : plan.packed ? (void*)cv.take<zc::u32>(m) : (void*)cv.take<uint2>(m);
        zc::u32* table = cv.take<zc::u32>(2 * plan.table_words);
        zc::u32* sums = cv.take<zc::u32>(plan.table_words / zc::SCAN_BLOCK_ELEMS + 1);
        if (!pass) {
            int rc = ensure(&D.msm, &D.msm_bytes, cv.off);
            if (rc) return rc;
            continue;
        }
        hipLaunchKernelGGL(zc::k_msm_digits, dim3(grid_for(n)), dim3(zc::ZC_BLOCK), 0, D.s(), (const u64*)scalars, digits, n, c, W);
        // ZC_MSM_GROUPS that adds up to the windows: every group sorted on its own, top group first (as msm_on_device does)
        int gsum = 0;
        for (int g = 0; g < D.tune.msm_ngroups; g++) gsum += D.tune.msm_groups[g];
        if (D.tune.msm_ngroups >= 2 && gsum == W) {
            int top = W;
            for (int g = 0; g < D.tune.msm_ngroups; g++) {
                top -= D.tune.msm_groups[g];
                if (int rc = msm_sort(D, D.s(), plan, top, D.tune.msm_groups[g], digits, pairs_a, pairs_b, table, plan.table_words, sums)) return rc;
            }
        } else if (int rc = msm_sort(D, D.s(), plan, 0, W, digits, pairs_a, pairs_b, table, plan.table_words, sums)) {
            return rc;
        }
        HIP_TRY(hipMemcpyAsync(out_pairs, pairs_a, m * sizeof(uint2), hipMemcpyDeviceToDevice, D.s()));
        HIP_TRY(hipStreamSynchronize(D.s()));
        HIP_TRY(hipGetLastError());
    }
    return ZC_OK;
}
// Test hook: the w-NAF's odd-multiples table as the device built it, as 125 points in DEVICE memory of slot 0.
int zc_test_odd_table(zc_ctx* ctx, uint64_t* out_dev_points)
{
    if (!ctx) return fail(ZC_ERR_BAD_ARG, "null context");
    REQUIRE(out_dev_points);
    std::lock_guard<std::mutex> lock(ctx->mu);
    DevState& D = ctx->devs[0];
    HIP_TRY(hipSetDevice(D.device));
    if (!D.odd_table) {
        int rc = ensure(&D.odd_table, &D.odd_bytes, (size_t)zc::ZC_ODD_ENTRIES * 128);
        if (rc) return rc;
        hipLaunchKernelGGL(zc::k_odd_table_build, dim3(1), dim3(128), 0, D.s(), (zc::u32*)D.odd_table);
    }
    hipLaunchKernelGGL(zc::k_test_odd_table_dump, dim3(1), dim3(128), 0, D.s(), (const zc::u32*)D.odd_table, (u64*)out_dev_points);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(D.s()));
    return ZC_OK;
}
// Test hook: how many element-wise / point launches of this context took their LDS-staged kernel so far (all device slots).
// Lets a parity test assert that the staged kernel -- not the per-lane one -- produced the output it compared.
long long zc_test_staged_launches(zc_ctx* ctx)
{
    if (!ctx) return -1;
    std::lock_guard<std::mutex> lock(ctx->mu);
    unsigned long long t = 0;
    for (auto& d : ctx->devs) t += d.staged_launches;
    return (long long)t;
}
#endif  // ZC_TEST_HOOKS

// What the bucket method would do for a shard of n pairs on this context (its knobs included) -- a query, no device
// work.  A measurement aid: a roofline record counts the useful multiplications from c, W and the addition formula.
// Writes min(nout, 17) entries (nout >= 8): [0] window bits c (0: below the bucket threshold, n scalar multiplications + folds),
// [1] windows W, [2] 1 = affine records / 7-multiplication additions, 0 = projective / 8, [3] PAYLOAD bytes of a gathered
// record (112 / 128), [4] run length of the bucket-sum kernel (window groups: the top group's), [5] buckets per reduction
// segment, [6] sort passes, [7] window groups G, [8] record STRIDE in bytes (what a gather touches: one 128-byte line),
// [9..12] windows per group (top group first), [13..16] run length per group.
int zc_msm_plan(zc_ctx* ctx, size_t n, int points_aligned16, int32_t* out, int nout)
{
    if (!ctx) return fail(ZC_ERR_BAD_ARG, "null context");
    REQUIRE(out);
    if (nout < 8) return fail(ZC_ERR_BAD_ARG, "zc_msm_plan: nout < 8");
    const MsmPlan p = msm_plan(n, points_aligned16 != 0, ctx->devs[0].tune);
    if (p.bad_groups) return fail(ZC_ERR_BAD_ARG, "zc_msm_plan: ZC_MSM_GROUPS does not add up to this shard's window count");
    const bool b = p.buckets;
    const int32_t v[17] = {p.c, p.W, p.affine ? 1 : 0, b ? (p.affine ? zc::MSM_AFF_WORDS * 4 : 128) : 0, b ? p.gT[0] : 0, p.seg, p.sort.passes, b ? p.G : 0, b ? p.rec_bytes : 0,
                           p.gw[0], p.gw[1], p.gw[2], p.gw[3], p.gT[0], p.gT[1], p.gT[2], p.gT[3]};
    memcpy(out, v, sizeof(int32_t) * (size_t)std::min(nout, 17));
    return ZC_OK;
}

// The same with the sum left in DEVICE memory (out_dev_point: 160 bytes on the device that owns
// the inputs, or on device slot 0 for host inputs); asynchronous on the context stream.
int zc_msm_partial(zc_ctx* ctx, const uint64_t* points, const uint64_t* scalars, size_t n, uint64_t* out_dev_point)
{
    if (!ctx) return fail(ZC_ERR_BAD_ARG, "null context");
    REQUIRE(out_dev_point);
    std::lock_guard<std::mutex> lock(ctx->mu);
    Residency ro;
    int dvo = -1;
    residency_of(out_dev_point, &ro, &dvo);
    if (ro != RES_DEVICE) return fail(ZC_ERR_MIXED_MEM, "zc_msm_partial: out_dev_point must be device memory");
    DevState* od = dev_state_of(ctx, dvo);
    if (!od) return fail(ZC_ERR_MIXED_MEM, "device buffers do not belong to a device of this context");
    if (n == 0) {
        HIP_TRY(hipSetDevice(od->device));
        HIP_TRY(hipMemcpyAsync(out_dev_point, IDENT_POINT, 160, hipMemcpyHostToDevice, od->s()));
        HIP_TRY(hipStreamSynchronize(od->s()));           // the source is host constant memory
        return ZC_OK;
    }
    REQUIRE(points); REQUIRE(scalars);
    DevState* owner = nullptr;
    const u64* res = nullptr;
    int rc = msm_local(ctx, points, scalars, n, &owner, &res);
    if (rc) return rc;
    if (owner != od) return fail(ZC_ERR_MIXED_MEM, "zc_msm_partial: output lives on another device than the sum");
    HIP_TRY(hipSetDevice(owner->device));
    HIP_TRY(hipMemcpyAsync(out_dev_point, res, 160, hipMemcpyDeviceToDevice, owner->s()));
    return ZC_OK;
}

// ((p_0 + p_1) + p_2) + ... + p_(count-1), unified addition (src/edwards.rs:465-489), ONE launch;
// host or device pointers as everywhere.  The exchange step of a sharded MSM after an all-gather.
int zc_ed_fold_ordered(zc_ctx* ctx, const uint64_t* parts, size_t count, uint64_t* out)
{
    REQUIRE(parts); REQUIRE(out);
    if (!ctx) return fail(ZC_ERR_BAD_ARG, "null context");
    if (count == 0) return fail(ZC_ERR_BAD_ARG, "zc_ed_fold_ordered: empty list");
    Residency rp, ro;
    int dp = -1, dvo = -1;
    residency_of(parts, &rp, &dp);
    residency_of(out, &ro, &dvo);
    if (rp != ro || (rp == RES_DEVICE && dp != dvo)) return fail(ZC_ERR_MIXED_MEM, "host and device buffers mixed in one call");
    std::lock_guard<std::mutex> lock(ctx->mu);
    DevState* ds = rp == RES_DEVICE ? dev_state_of(ctx, dp) : &ctx->devs[0];
    if (!ds) return fail(ZC_ERR_MIXED_MEM, "device buffers do not belong to a device of this context");
    if (int rc = ring_check(*ds)) return rc;
    HIP_TRY(hipSetDevice(ds->device));
    if (rp == RES_DEVICE) {
        hipLaunchKernelGGL(zc::k_ed_fold_ordered, dim3(1), dim3(64), 0, ds->s(), (const u64*)parts, count, (const u64*)nullptr, (u64*)out);
        HIP_TRY(hipGetLastError());
        return ZC_OK;
    }
    int rc = ensure(&ds->part, &ds->part_bytes, (count + 1) * 160);
    if (rc) return rc;
    u64* part = (u64*)ds->part;
    HIP_TRY(hipMemcpyAsync(part, parts, count * 160, hipMemcpyHostToDevice, ds->s()));
    hipLaunchKernelGGL(zc::k_ed_fold_ordered, dim3(1), dim3(64), 0, ds->s(), (const u64*)part, count, (const u64*)nullptr, part + 20 * count);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, part + 20 * count, 160, hipMemcpyDeviceToHost, ds->s()));
    HIP_TRY(hipStreamSynchronize(ds->s()));
    return ZC_OK;
}

// ---- one process per GPU: RCCL communicator owned by the context ------------------------------
int zc_comm_unique_id(uint8_t* id_out128)
{
    REQUIRE(id_out128);
    int rc = rccl_load();
    if (rc) return rc;
    ncclUniqueId id;
    RCCL_TRY(g_rccl.GetUniqueId(&id));
    static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
    memcpy(id_out128, &id, sizeof id);
    return ZC_OK;
}
int zc_comm_init(zc_ctx* ctx, const uint8_t* id128, int rank, int world)
{
    if (!ctx) return fail(ZC_ERR_BAD_ARG, "null context");
    REQUIRE(id128);
    if (world < 1 || rank < 0 || rank >= world) return fail(ZC_ERR_BAD_ARG, "zc_comm_init: bad rank / world size");
    int rc = rccl_load();
    if (rc) return rc;
    std::lock_guard<std::mutex> lock(ctx->mu);
    if (ctx->comm) return fail(ZC_ERR_BAD_ARG, "zc_comm_init: the context already has a communicator");
    HIP_TRY(hipSetDevice(ctx->devs[0].device));
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    RCCL_TRY(g_rccl.CommInitRank(&ctx->comm, world, id, rank));
    ctx->rank = rank;
    ctx->world = world;
    return ZC_OK;
}
int zc_comm_destroy(zc_ctx* ctx)
{
    if (!ctx) return fail(ZC_ERR_BAD_ARG, "null context");
    std::lock_guard<std::mutex> lock(ctx->mu);
    if (ctx->comm) {
        (void)zc_ctx_synchronize(ctx);
        RCCL_TRY(g_rccl.CommDestroy(ctx->comm));
        ctx->comm = nullptr;
        ctx->world = 1;
        ctx->rank = 0;
    }
    return ZC_OK;
}

// The number of ranks RCCL itself reports for the context's communicator (ncclCommCount); 0 without one.
// What a scaling record quotes to show that the exchange really ran over N ranks.
int zc_comm_size(zc_ctx* ctx, int* ranks)
{
    if (!ctx) return fail(ZC_ERR_BAD_ARG, "null context");
    REQUIRE(ranks);
    std::lock_guard<std::mutex> lock(ctx->mu);
    *ranks = 0;
    if (ctx->comm) RCCL_TRY(g_rccl.CommCount(ctx->comm, ranks));
    return ZC_OK;
}

// BASELINE configs[4]: this rank's shard of a global MSM.  Local bucket method -> ncclAllGather of
// the 160-byte partial sums over xGMI (20 x ncclUint64 per rank, on the context stream) -> ordered
// fold in one kernel -> every rank returns the same point (identical limbs).  Point addition is not
// an ncclRedOp_t, hence all-gather + fold rather than ncclAllReduce.
// A rank whose local part fails (bad arguments, no memory, a HIP error) still joins the collective -- with a
// poison record no point can equal (limbs of all ones) -- so that no other rank is left waiting in it; every
// rank then sees the poison among the gathered rows and ALL of them return an error.
int zc_msm_sharded(zc_ctx* ctx, const uint64_t* points, const uint64_t* scalars, size_t n_local, uint64_t* out_point)
{
    if (!ctx) return fail(ZC_ERR_BAD_ARG, "null context");
    REQUIRE(out_point);
    std::lock_guard<std::mutex> lock(ctx->mu);
    if (!ctx->comm) return fail(ZC_ERR_BAD_ARG, "zc_msm_sharded: call zc_comm_init first");
    DevState* d0 = &ctx->devs[0];
    const size_t world = (size_t)ctx->world;
    HIP_TRY(hipSetDevice(d0->device));
    const size_t front = ctx->devs.size() + 1;               // gather_and_fold's region (multi-slot host inputs)
    int rc = ensure(&d0->part, &d0->part_bytes, (front + world + 2) * 160);
    if (rc) return rc;                                       // nothing to send from: the one failure that cannot join
    u64* gathered = (u64*)d0->part + 20 * front;
    u64* mine = gathered + 20 * world;                       // mine, then the folded result
    int local_rc = ZC_OK;
    std::string local_err;
    if (n_local == 0) {
        HIP_TRY(hipMemcpyAsync(mine, IDENT_POINT, 160, hipMemcpyHostToDevice, d0->s()));
    } else {
        DevState* owner = nullptr;
        const u64* res = nullptr;
        if (!points || !scalars) local_rc = fail(ZC_ERR_BAD_ARG, "zc_msm_sharded: null points / scalars");
        if (!local_rc) local_rc = msm_local(ctx, points, scalars, n_local, &owner, &res);
        if (!local_rc && owner != d0) local_rc = fail(ZC_ERR_MIXED_MEM, "zc_msm_sharded: inputs must live on device slot 0 (or on the host)");
        if (local_rc) local_err = g_last_error;
        HIP_TRY(hipSetDevice(d0->device));
        if (local_rc)
            HIP_TRY(hipMemsetAsync(mine, 0xFF, 160, d0->s()));
        else
            HIP_TRY(hipMemcpyAsync(mine, res, 160, hipMemcpyDeviceToDevice, d0->s()));
    }
    RCCL_TRY(g_rccl.AllGather(mine, gathered, 20, ncclUint64, ctx->comm, d0->s()));
    hipLaunchKernelGGL(zc::k_ed_fold_ordered, dim3(1), dim3(64), 0, d0->s(), (const u64*)gathered, world, (const u64*)nullptr, mine + 20);
    HIP_TRY(hipGetLastError());
    std::vector<u64> host(20 * (world + 2));                 // the gathered rows ride along: one small copy
    HIP_TRY(hipMemcpyAsync(host.data(), gathered, host.size() * sizeof(u64), hipMemcpyDeviceToHost, d0->s()));
    HIP_TRY(hipStreamSynchronize(d0->s()));
    if (local_rc) {
        g_last_error = local_err;
        return local_rc;
    }
    for (size_t r = 0; r < world; r++)
        if (host[20 * r] == ~(u64)0) return fail(ZC_ERR_HIP, ("zc_msm_sharded: rank " + std::to_string(r) + " failed its local part").c_str());
    memcpy(out_point, host.data() + 20 * (world + 1), 160);
    return ZC_OK;
}

}  // extern "C"
