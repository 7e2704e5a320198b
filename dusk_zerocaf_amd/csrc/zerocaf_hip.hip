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
#include <vector>

#include "../../include/zerocaf_hip.h"
#include "zc_kernels.cuh"
#include "zc_msm.cuh"

#include <rocprim/rocprim.hpp>

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

struct DevState {
    int device = 0;
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
    void* fast = nullptr;               // fast scalar-mul window tables: 1 KB per lane
    size_t fast_bytes = 0;
    void* base_table = nullptr;         // comb table of the basepoint: 66 x 8 cached points
    size_t base_bytes = 0;
    hipStream_t s() const { return use_borrowed ? borrowed : stream; }
};

}  // namespace

struct zc_ctx {
    std::vector<DevState> devs;
    std::mutex mu;
};

namespace {

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
inline size_t host_chunk_elems(size_t cnt, bool heavy)
{
    const char* e = getenv("ZC_HOST_CHUNKS");
    const long forced = e ? atol(e) : 0L;
    size_t chunk = cnt;
    if (forced > 0)
        chunk = (cnt + forced - 1) / forced;
    else if (heavy && cnt >= 4 * CHUNK_ROUND)
        chunk = 2 * CHUNK_ROUND;
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
        HIP_TRY(hipSetDevice(ds->device));
        void* dptr[MAX_ARGS];
        for (int a = 0; a < nargs; a++) dptr[a] = const_cast<void*>(args[a].ptr);
        launch(dptr, n, *ds);
        HIP_TRY(hipGetLastError());
        return ZC_OK;
    }

    // host buffers: shard into contiguous ranges, one per device (no exchange step); each range
    // moves through the device in chunks so uploads, kernels and downloads overlap
    const size_t ndev = ctx->devs.size();
    const size_t per = (n + ndev - 1) / ndev;

    struct Plan {
        size_t lo = 0, cnt = 0, chunk = 0, nchunks = 0;
        void* base[MAX_ARGS] = {};
    };
    std::vector<Plan> plans(ndev);
    size_t max_chunks = 0;
    for (size_t di = 0; di < ndev; di++) {
        Plan& pl = plans[di];
        pl.lo = di * per;
        const size_t hi = std::min(n, pl.lo + per);
        if (pl.lo >= hi) break;
        pl.cnt = hi - pl.lo;
        pl.chunk = host_chunk_elems(pl.cnt, heavy);
        pl.nchunks = (pl.cnt + pl.chunk - 1) / pl.chunk;
        max_chunks = std::max(max_chunks, pl.nchunks);
        DevState& ds = ctx->devs[di];
        HIP_TRY(hipSetDevice(ds.device));
        for (int a = 0; a < nargs; a++) {
            if (!args[a].ptr) continue;
            int rc = ensure(&ds.scratch[a], &ds.scratch_bytes[a], args[a].elt_bytes * pl.cnt);
            if (rc) return rc;
            pl.base[a] = ds.scratch[a];
        }
        while (ds.ev.size() < 2 * pl.nchunks) {
            hipEvent_t e;
            HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            ds.ev.push_back(e);
        }
    }
    // chunk j of device di: upload on copy_in, kernel on the context stream, download on copy_out
    auto upload_and_launch = [&](size_t di, size_t j) -> int {
        Plan& pl = plans[di];
        if (j >= pl.nchunks) return ZC_OK;
        DevState& ds = ctx->devs[di];
        HIP_TRY(hipSetDevice(ds.device));
        const size_t off = j * pl.chunk, cnt = std::min(pl.chunk, pl.cnt - off);
        void* dptr[MAX_ARGS];
        for (int a = 0; a < nargs; a++) {
            dptr[a] = pl.base[a] ? (char*)pl.base[a] + args[a].elt_bytes * off : nullptr;
            if (!args[a].ptr || args[a].is_out) continue;
            const char* src = (const char*)args[a].ptr + args[a].elt_bytes * (pl.lo + off);
            HIP_TRY(hipMemcpyAsync(dptr[a], src, args[a].elt_bytes * cnt, hipMemcpyHostToDevice, ds.copy_in));
        }
        HIP_TRY(hipEventRecord(ds.ev[2 * j], ds.copy_in));
        HIP_TRY(hipStreamWaitEvent(ds.s(), ds.ev[2 * j], 0));
        launch(dptr, cnt, ds);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(ds.ev[2 * j + 1], ds.s()));
        return ZC_OK;
    };
    auto download = [&](size_t di, size_t j) -> int {
        Plan& pl = plans[di];
        if (j >= pl.nchunks) return ZC_OK;
        DevState& ds = ctx->devs[di];
        HIP_TRY(hipSetDevice(ds.device));
        const size_t off = j * pl.chunk, cnt = std::min(pl.chunk, pl.cnt - off);
        HIP_TRY(hipStreamWaitEvent(ds.copy_out, ds.ev[2 * j + 1], 0));
        for (int a = 0; a < nargs; a++) {
            if (!args[a].ptr || !args[a].is_out) continue;
            char* dst = (char*)const_cast<void*>(args[a].ptr) + args[a].elt_bytes * (pl.lo + off);
            HIP_TRY(hipMemcpyAsync(dst, (char*)pl.base[a] + args[a].elt_bytes * off, args[a].elt_bytes * cnt, hipMemcpyDeviceToHost, ds.copy_out));
        }
        return ZC_OK;
    };
    // Copies from/to pageable memory block the calling thread, so the issue order keeps
    // LOOKAHEAD kernels queued on every device before the thread waits on a download.
    constexpr size_t LOOKAHEAD = 2;
    int rc = ZC_OK;
    for (size_t j = 0; j < std::min(LOOKAHEAD, max_chunks) && !rc; j++)
        for (size_t di = 0; di < ndev && !rc; di++) rc = upload_and_launch(di, j);
    for (size_t j = 0; j < max_chunks && !rc; j++) {
        for (size_t di = 0; di < ndev && !rc; di++) rc = download(di, j);
        for (size_t di = 0; di < ndev && !rc; di++) rc = upload_and_launch(di, j + LOOKAHEAD);
    }
    for (size_t di = 0; di < ndev; di++) {
        if (!plans[di].cnt) continue;
        DevState& ds = ctx->devs[di];
        HIP_TRY(hipSetDevice(ds.device));
        HIP_TRY(hipStreamSynchronize(ds.copy_in));
        HIP_TRY(hipStreamSynchronize(ds.s()));
        HIP_TRY(hipStreamSynchronize(ds.copy_out));
    }
    return rc;
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
// 256 MB Infinity Cache (zc_kernels.cuh, "LDS-staged element I/O").
constexpr size_t STREAM_BYTES = (size_t)256 << 20;

int binop(zc_ctx* ctx, kbin_t k, kbin_t k_stream, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n, size_t elt)
{
    REQUIRE(a); REQUIRE(b); REQUIRE(out);
    // elt == 0: (point, scalar) -> point
    Arg args[3] = {in_arg(a, elt ? elt : 160), in_arg(b, elt ? elt : 40), out_arg(out, elt ? elt : 160)};
    return run_batched(ctx, args, 3, n, [&](void** d, size_t cnt, DevState& D) {
        const bool stream = k_stream && cnt * elt * 3 > STREAM_BYTES && aligned16(d[0]) && aligned16(d[1]) && aligned16(d[2]);
        hipLaunchKernelGGL(stream ? k_stream : k, dim3(grid_for(cnt)), dim3(zc::ZC_BLOCK), 0, D.s(), (const u64*)d[0], (const u64*)d[1], (u64*)d[2], cnt);
    }, elt == 0);
}
int unop(zc_ctx* ctx, kun_t k, const uint64_t* a, uint64_t* out, size_t n, size_t elt)
{
    REQUIRE(a); REQUIRE(out);
    Arg args[2] = {in_arg(a, elt), out_arg(out, elt)};
    return run_batched(ctx, args, 2, n, [&](void** d, size_t cnt, DevState& D) {
        hipLaunchKernelGGL(k, dim3(grid_for(cnt)), dim3(zc::ZC_BLOCK), 0, D.s(), (const u64*)d[0], (u64*)d[1], cnt);
    });
}

// Cost-sorted permutation for the unified-step kernels (see zc_kernels.cuh "lane balancing").
// Returns nullptr (natural order) for small batches or when scratch cannot be had.
// ZC_BALANCE=global selects it; the default is the in-kernel block-local ranking, which keeps
// HBM traffic algorithmic (a batch-wide permutation turns record reads into cache-line gathers).
constexpr size_t BALANCE_MIN_N = 1 << 14;
inline bool global_balance()
{
    static const bool g = [] { const char* e = getenv("ZC_BALANCE"); return e && std::string(e) == "global"; }();
    return g;
}
const zc::u32* balance_index(DevState& D, const u64* k, size_t cnt)
{
    if (!global_balance() || cnt < BALANCE_MIN_N || cnt > 0xFFFFFFFFull) return nullptr;
    const size_t need = zc::ZC_COST_BINS * sizeof(zc::u32) + cnt * sizeof(zc::u32);
    if (ensure(&D.bal, &D.bal_bytes, need) != ZC_OK) return nullptr;
    zc::u32* hist = (zc::u32*)D.bal;
    zc::u32* idx = hist + zc::ZC_COST_BINS;
    if (hipMemsetAsync(hist, 0, zc::ZC_COST_BINS * sizeof(zc::u32), D.s()) != hipSuccess) return nullptr;
    hipLaunchKernelGGL(zc::k_sm_cost_hist, dim3(grid_for(cnt)), dim3(zc::ZC_BLOCK), 0, D.s(), k, hist, cnt);
    hipLaunchKernelGGL(zc::k_sm_cost_scan, dim3(1), dim3(zc::ZC_BLOCK), 0, D.s(), hist);
    hipLaunchKernelGGL(zc::k_sm_cost_scatter, dim3(grid_for(cnt)), dim3(zc::ZC_BLOCK), 0, D.s(), k, hist, idx, cnt);
    return idx;
}

// Launches of at most one workgroup per CU keep a single wave on every SIMD; a lone wave cannot
// hide the latency of the column-ordered multiplier's serial chain, so those launches run the
// variant with independent column chains (2^16 units: 2.09 -> 1.79 ms; from 384 workgroups on the
// default kernel is faster again).
constexpr unsigned SMALL_LAUNCH_BLOCKS = 256;
constexpr size_t QUAD_LAUNCH_ELEMS = (size_t)1 << 14;     // 4 lanes per element still leave one wave per SIMD
typedef void (*strict_kernel_t)(const u64*, const u64*, size_t, u64*, const zc::u32*, size_t);
inline strict_kernel_t strict_kernel_for(size_t cnt)
{
    return grid_for(cnt) <= SMALL_LAUNCH_BLOCKS ? zc::k_ed_scalar_mul_small : zc::k_ed_scalar_mul;
}
int scalar_mul_impl(zc_ctx* ctx, const uint64_t* p, const uint64_t* k, uint64_t* out, size_t n)
{
    REQUIRE(p); REQUIRE(k); REQUIRE(out);
    Arg args[3] = {in_arg(p, 160), in_arg(k, 40), out_arg(out, 160)};
    return run_batched(ctx, args, 3, n, [&](void** d, size_t cnt, DevState& D) {
        const zc::u32* idx = balance_index(D, (const u64*)d[1], cnt);
        if (cnt <= QUAD_LAUNCH_ELEMS && !idx) {
            // four lanes per element: the batch cannot fill the chip anyway, so buy latency with lanes
            hipLaunchKernelGGL(zc::k_ed_scalar_mul_quad, dim3((unsigned)((cnt + 63) / 64)), dim3(zc::ZC_BLOCK), 0, D.s(), (const u64*)d[0],
                               (const u64*)d[1], (u64*)d[2], cnt);
            return;
        }
        hipLaunchKernelGGL(strict_kernel_for(cnt), dim3(grid_for(cnt)),
                           dim3(zc::ZC_BLOCK), 0, D.s(), (const u64*)d[0], (const u64*)d[1], (size_t)5, (u64*)d[2], idx, cnt);
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

int msm_on_device(DevState& D, const u64* dP, const u64* dK, size_t cnt, const u64** result)
{
    if (cnt < MSM_BUCKET_MIN_N) {
        // small shard: n scalar-muls, then pairwise folds
        int rc = ensure(&D.tmp[0], &D.tmp_bytes[0], cnt * 160);
        if (rc) return rc;
        rc = ensure(&D.tmp[1], &D.tmp_bytes[1], ((cnt + 1) / 2) * 160 + 256);
        if (rc) return rc;
        hipLaunchKernelGGL(strict_kernel_for(cnt), dim3(grid_for(cnt)), dim3(zc::ZC_BLOCK), 0, D.s(), dP, dK, (size_t)5, (u64*)D.tmp[0],
                           (const zc::u32*)nullptr, cnt);
        *result = fold_all(D, (u64*)D.tmp[0], (u64*)D.tmp[1], cnt);
        HIP_TRY(hipGetLastError());
        return ZC_OK;
    }
    // window width: ~16-32 points per bucket, 4 <= c <= 16
    int c = 0;
    while (((size_t)1 << (c + 1)) <= cnt) c++;
    c -= 4;
    if (c < 4) c = 4;
    if (c > 16) c = 16;
    if (const char* e = getenv("ZC_MSM_WINDOW")) {         // test hook: force the window width
        const int f = atoi(e);
        if (f >= 4 && f <= 16) c = f;
    }
    // number of windows from the longest scalar actually present (canonical scalars: 250-253 bits)
    int maxbits = 0;
    {
        int rc = ensure(&D.tmp[1], &D.tmp_bytes[1], 256);
        if (rc) return rc;
        HIP_TRY(hipMemsetAsync(D.tmp[1], 0, sizeof(int), D.s()));
        hipLaunchKernelGGL(zc::k_msm_maxbits, dim3(grid_for(cnt)), dim3(zc::ZC_BLOCK), 0, D.s(), dK, (int*)D.tmp[1], cnt);
        HIP_TRY(hipMemcpyAsync(&maxbits, D.tmp[1], sizeof(int), hipMemcpyDeviceToHost, D.s()));
        HIP_TRY(hipStreamSynchronize(D.s()));
    }
    if (maxbits == 0) {                                   // all scalars zero: the sum is the identity
        static const uint64_t ident[20] = {0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        int rc = ensure(&D.tmp[0], &D.tmp_bytes[0], 256);
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(D.tmp[0], ident, sizeof ident, hipMemcpyHostToDevice, D.s()));
        HIP_TRY(hipStreamSynchronize(D.s()));
        *result = (const u64*)D.tmp[0];
        return ZC_OK;
    }
    const int W = (maxbits + c - 1) / c;
    const size_t m = cnt * (size_t)W;
    if (m > 0xFFFFFFFFull) return fail(ZC_ERR_BAD_ARG, "zc_msm: shard too large for 32-bit pair indices");
    const size_t nb = (size_t)W << c;                     // buckets
    const size_t nseg = nb / zc::MSM_SEG;
    int keybits = c;
    while ((1 << (keybits - c)) < W) keybits++;

    size_t sort_tmp = 0;
    {
        rocprim::double_buffer<zc::u32> kq(nullptr, nullptr), vq(nullptr, nullptr);
        HIP_TRY(rocprim::radix_sort_pairs(nullptr, sort_tmp, kq, vq, m, 0, (unsigned)keybits, D.s()));
        size_t t2 = 0;
        HIP_TRY(rocprim::radix_sort_pairs_desc(nullptr, t2, kq, vq, nb, 0, 32, D.s()));
        if (t2 > sort_tmp) sort_tmp = t2;
    }
    for (int pass = 0; pass < 2; pass++) {
        Carver cv{pass ? (char*)D.msm : nullptr};
        zc::u32* keys0 = cv.take<zc::u32>(m);
        zc::u32* keys1 = cv.take<zc::u32>(m);
        zc::u32* vals0 = cv.take<zc::u32>(m);
        zc::u32* vals1 = cv.take<zc::u32>(m);
        char* tmp = cv.take<char>(sort_tmp);
        zc::u32* start = cv.take<zc::u32>(nb);
        zc::u32* end = cv.take<zc::u32>(nb);
        zc::u32* bcnt0 = cv.take<zc::u32>(nb);
        zc::u32* bcnt1 = cv.take<zc::u32>(nb);
        zc::u32* bid0 = cv.take<zc::u32>(nb);
        zc::u32* bid1 = cv.take<zc::u32>(nb);
        zc::u32* cached = cv.take<zc::u32>(cnt * 32);
        u64* buckets = cv.take<u64>(nb * 20);
        u64* seg_sum = cv.take<u64>(nseg * 20);
        u64* seg_acc = cv.take<u64>(nseg * 20);
        u64* seg_k = cv.take<u64>(nseg * 5);
        u64* fold_b = cv.take<u64>((nseg / 2 + 1) * 20);
        if (!pass) {
            int rc = ensure(&D.msm, &D.msm_bytes, cv.off);
            if (rc) return rc;
            continue;
        }
        hipLaunchKernelGGL(zc::k_msm_digits, dim3(grid_for(cnt)), dim3(zc::ZC_BLOCK), 0, D.s(), dK, keys0, vals0, cnt, c, W);
        rocprim::double_buffer<zc::u32> kb(keys0, keys1), vb(vals0, vals1);
        size_t st = sort_tmp;
        HIP_TRY(rocprim::radix_sort_pairs(tmp, st, kb, vb, m, 0, (unsigned)keybits, D.s()));
        HIP_TRY(hipMemsetAsync(start, 0, nb * sizeof(zc::u32), D.s()));
        HIP_TRY(hipMemsetAsync(end, 0, nb * sizeof(zc::u32), D.s()));
        hipLaunchKernelGGL(zc::k_msm_bounds, dim3(grid_for(m)), dim3(zc::ZC_BLOCK), 0, D.s(), (const zc::u32*)kb.current(), start, end, m);
        hipLaunchKernelGGL(zc::k_msm_prepare, dim3(grid_for(cnt)), dim3(zc::ZC_BLOCK), 0, D.s(), dP, cached, cnt);
        hipLaunchKernelGGL(zc::k_msm_counts, dim3(grid_for(nb)), dim3(zc::ZC_BLOCK), 0, D.s(), (const zc::u32*)start, (const zc::u32*)end, bcnt0, bid0, nb, c);
        rocprim::double_buffer<zc::u32> cb(bcnt0, bcnt1), ib(bid0, bid1);
        st = sort_tmp;
        HIP_TRY(rocprim::radix_sort_pairs_desc(tmp, st, cb, ib, nb, 0, 32, D.s()));
        hipLaunchKernelGGL(zc::k_msm_accumulate, dim3(grid_for(nb)), dim3(zc::ZC_BLOCK), 0, D.s(), (const zc::u32*)cached, (const zc::u32*)vb.current(),
                           (const zc::u32*)start, (const zc::u32*)end, (const zc::u32*)ib.current(), buckets, nb, c);
        hipLaunchKernelGGL(zc::k_msm_segments, dim3(grid_for(nseg)), dim3(zc::ZC_BLOCK), 0, D.s(), (const u64*)buckets, seg_sum, seg_acc, seg_k, nseg, c);
        // seg_acc <- (lo - 1) * seg_acc ; seg_sum <- seg_sum + seg_acc
        hipLaunchKernelGGL(strict_kernel_for(nseg), dim3(grid_for(nseg)), dim3(zc::ZC_BLOCK), 0, D.s(), (const u64*)seg_acc, (const u64*)seg_k, (size_t)5,
                           seg_acc, (const zc::u32*)nullptr, nseg);
        hipLaunchKernelGGL(zc::k_ed_add, dim3(grid_for(nseg)), dim3(zc::ZC_BLOCK), 0, D.s(), (const u64*)seg_sum, (const u64*)seg_acc, seg_sum, nseg);
        // fold every window's nseg/W segment sums (power of two per window: pairs never straddle windows)
        size_t left = nseg;
        u64* cur = seg_sum;
        u64* nxt = fold_b;
        while (left > (size_t)W) {
            hipLaunchKernelGGL(zc::k_ed_fold_pairs, dim3(grid_for(left / 2)), dim3(zc::ZC_BLOCK), 0, D.s(), (const u64*)cur, nxt, left);
            left /= 2;
            std::swap(cur, nxt);
        }
        // sum_w 2^(c w) S_w
        hipLaunchKernelGGL(zc::k_msm_window_combine, dim3(1), dim3(64), 0, D.s(), (const u64*)cur, nxt, W, c);
        *result = nxt;
        HIP_TRY(hipGetLastError());
    }
    return ZC_OK;
}

}  // namespace

// =============================================================================== C ABI
extern "C" {

const char* zc_version(void) { return "zerocaf_hip 0.1 (gfx950, radix-2^29 Montgomery R=2^261)"; }
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
    for (int id : ids) {
        DevState ds;
        ds.device = id;
        hipError_t e = hipSetDevice(id);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&ds.stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&ds.copy_in, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&ds.copy_out, hipStreamNonBlocking);
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

int zc_ctx_destroy(zc_ctx* ctx)
{
    if (!ctx) return ZC_OK;
    for (auto& ds : ctx->devs) {
        (void)hipSetDevice(ds.device);
        (void)hipStreamSynchronize(ds.s());
        for (int a = 0; a < MAX_ARGS; a++)
            if (ds.scratch[a]) (void)hipFree(ds.scratch[a]);
        for (int a = 0; a < 2; a++)
            if (ds.tmp[a]) (void)hipFree(ds.tmp[a]);
        if (ds.bal) (void)hipFree(ds.bal);
        if (ds.msm) (void)hipFree(ds.msm);
        if (ds.fast) (void)hipFree(ds.fast);
        if (ds.base_table) (void)hipFree(ds.base_table);
        for (hipEvent_t e : ds.ev) (void)hipEventDestroy(e);
        if (ds.copy_in) (void)hipStreamDestroy(ds.copy_in);
        if (ds.copy_out) (void)hipStreamDestroy(ds.copy_out);
        if (ds.stream) (void)hipStreamDestroy(ds.stream);
    }
    delete ctx;
    return ZC_OK;
}

int zc_ctx_set_stream(zc_ctx* ctx, void* hip_stream, int external)
{
    if (!ctx) return fail(ZC_ERR_BAD_ARG, "null context");
    std::lock_guard<std::mutex> lock(ctx->mu);
    ctx->devs[0].borrowed = external ? (hipStream_t)hip_stream : nullptr;
    ctx->devs[0].use_borrowed = external != 0;
    return ZC_OK;
}

int zc_ctx_synchronize(zc_ctx* ctx)
{
    if (!ctx) return fail(ZC_ERR_BAD_ARG, "null context");
    for (auto& ds : ctx->devs) {
        HIP_TRY(hipSetDevice(ds.device));
        HIP_TRY(hipStreamSynchronize(ds.s()));
    }
    return ZC_OK;
}

// ---- FieldElement
int zc_fe_add(zc_ctx* c, const uint64_t* a, const uint64_t* b, uint64_t* o, size_t n) { return binop(c, zc::k_fe_add, zc::k_fe_add_stream, a, b, o, n, 40); }
int zc_fe_sub(zc_ctx* c, const uint64_t* a, const uint64_t* b, uint64_t* o, size_t n) { return binop(c, zc::k_fe_sub, zc::k_fe_sub_stream, a, b, o, n, 40); }
int zc_fe_mul(zc_ctx* c, const uint64_t* a, const uint64_t* b, uint64_t* o, size_t n) { return binop(c, zc::k_fe_mul, nullptr, a, b, o, n, 40); }
int zc_fe_neg(zc_ctx* c, const uint64_t* a, uint64_t* o, size_t n) { return unop(c, zc::k_fe_neg, a, o, n, 40); }
int zc_fe_square(zc_ctx* c, const uint64_t* a, uint64_t* o, size_t n) { return unop(c, zc::k_fe_square, a, o, n, 40); }

int zc_fe_invert(zc_ctx* ctx, const uint64_t* a, uint64_t* out, uint8_t* ok, size_t n)
{
    REQUIRE(a); REQUIRE(out);
    Arg args[3] = {in_arg(a, 40), out_arg(out, 40), out_arg(ok, 1)};
    return run_batched(ctx, args, 3, n, [&](void** d, size_t cnt, DevState& D) {
        // chunk length: keep >= 2 waves per SIMD busy (256 CUs x 4 SIMDs x 2 x 64 lanes), cap at 64
        size_t c = cnt / 131072;
        if (c > 64) c = 64;
        if (c < 2 || d[0] == d[1]) {                       // tiny batch or in-place: one element per lane
            hipLaunchKernelGGL(zc::k_fe_invert, dim3(grid_for(cnt)), dim3(zc::ZC_BLOCK), 0, D.s(), (const u64*)d[0], (u64*)d[1], (uint8_t*)d[2], cnt);
        } else {
            const size_t lanes = (cnt + c - 1) / c;
            hipLaunchKernelGGL(zc::k_fe_invert_chunked, dim3(grid_for(lanes)), dim3(zc::ZC_BLOCK), 0, D.s(), (const u64*)d[0], (u64*)d[1], (uint8_t*)d[2], cnt, (int)c);
        }
    });
}
int zc_fe_div(zc_ctx* ctx, const uint64_t* a, const uint64_t* b, uint64_t* out, uint8_t* ok, size_t n)
{
    REQUIRE(a); REQUIRE(b); REQUIRE(out);
    Arg args[4] = {in_arg(a, 40), in_arg(b, 40), out_arg(out, 40), out_arg(ok, 1)};
    return run_batched(ctx, args, 4, n, [&](void** d, size_t cnt, DevState& D) {
        size_t c = cnt / 131072;                           // as zc_fe_invert
        if (c > 64) c = 64;
        if (c < 2 || d[2] == d[0] || d[2] == d[1]) {
            hipLaunchKernelGGL(zc::k_fe_div, dim3(grid_for(cnt)), dim3(zc::ZC_BLOCK), 0, D.s(), (const u64*)d[0], (const u64*)d[1], (u64*)d[2], (uint8_t*)d[3], cnt);
        } else {
            const size_t lanes = (cnt + c - 1) / c;
            hipLaunchKernelGGL(zc::k_fe_div_chunked, dim3(grid_for(lanes)), dim3(zc::ZC_BLOCK), 0, D.s(), (const u64*)d[0], (const u64*)d[1], (u64*)d[2], (uint8_t*)d[3], cnt, (int)c);
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
int zc_fe_legendre_symbol(zc_ctx* c, const uint64_t* a, uint8_t* o, size_t n) { return fe_flag_op(c, zc::k_fe_legendre, a, o, n); }
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

// ---- Scalar
int zc_sc_add(zc_ctx* c, const uint64_t* a, const uint64_t* b, uint64_t* o, size_t n) { return binop(c, zc::k_sc_add, zc::k_sc_add_stream, a, b, o, n, 40); }
int zc_sc_sub(zc_ctx* c, const uint64_t* a, const uint64_t* b, uint64_t* o, size_t n) { return binop(c, zc::k_sc_sub, zc::k_sc_sub_stream, a, b, o, n, 40); }
int zc_sc_mul(zc_ctx* c, const uint64_t* a, const uint64_t* b, uint64_t* o, size_t n) { return binop(c, zc::k_sc_mul, nullptr, a, b, o, n, 40); }
int zc_sc_neg(zc_ctx* c, const uint64_t* a, uint64_t* o, size_t n) { return unop(c, zc::k_sc_neg, a, o, n, 40); }
int zc_sc_square(zc_ctx* c, const uint64_t* a, uint64_t* o, size_t n) { return unop(c, zc::k_sc_square, a, o, n, 40); }
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
int zc_ed_add(zc_ctx* c, const uint64_t* p, const uint64_t* q, uint64_t* o, size_t n) { return binop(c, zc::k_ed_add, nullptr, p, q, o, n, 160); }
int zc_ed_sub(zc_ctx* c, const uint64_t* p, const uint64_t* q, uint64_t* o, size_t n) { return binop(c, zc::k_ed_sub, nullptr, p, q, o, n, 160); }
int zc_ed_double(zc_ctx* c, const uint64_t* p, uint64_t* o, size_t n) { return unop(c, zc::k_ed_double, p, o, n, 160); }
int zc_ed_neg(zc_ctx* c, const uint64_t* p, uint64_t* o, size_t n) { return unop(c, zc::k_ed_neg, p, o, n, 160); }

int zc_ed_scalar_mul(zc_ctx* ctx, const uint64_t* p, const uint64_t* k, uint64_t* out, size_t n, unsigned flags)
{
    if (flags == ZC_SCALAR_MUL_STRICT) return scalar_mul_impl(ctx, p, k, out, n);
    if (flags == ZC_SCALAR_MUL_FAST) {
        REQUIRE(p); REQUIRE(k); REQUIRE(out);
        Arg args[3] = {in_arg(p, 160), in_arg(k, 40), out_arg(out, 160)};
        int inner = ZC_OK;
        int rc = run_batched(ctx, args, 3, n, [&](void** d, size_t cnt, DevState& D) {
            const size_t lanes = (size_t)grid_for(cnt) * zc::ZC_BLOCK;
            if ((inner = ensure(&D.fast, &D.fast_bytes, lanes * 1024)) != ZC_OK) return;
            hipLaunchKernelGGL(zc::k_ed_scalar_mul_fast, dim3(grid_for(cnt)), dim3(zc::ZC_BLOCK), 0, D.s(), (const u64*)d[0],
                               (const u64*)d[1], (size_t)5, (u64*)d[2], (zc::u32*)D.fast, cnt);
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
        size_t c = cnt / 131072;                           // as zc_fe_invert: >= 2 waves per SIMD stay busy
        if (c > 64) c = 64;
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
    static const bool strict = [] { const char* e = getenv("ZC_RISTRETTO_STRICT"); return e && atoi(e) != 0; }();
    int inner = ZC_OK;
    int rc = run_batched(ctx, args, 4, n, [&](void** d, size_t cnt, DevState& D) {
        if (strict) {
            const zc::u32* idx = balance_index(D, (const u64*)d[1], cnt);
            hipLaunchKernelGGL(zc::k_ris_roundtrip_mul, dim3(grid_for(cnt)), dim3(zc::ZC_BLOCK), 0, D.s(), (const uint8_t*)d[0], (const u64*)d[1], (uint8_t*)d[2], (uint8_t*)d[3], idx, cnt);
            return;
        }
        const size_t lanes = (size_t)grid_for(cnt) * zc::ZC_BLOCK;
        if ((inner = ensure(&D.fast, &D.fast_bytes, lanes * 1024)) != ZC_OK) return;
        hipLaunchKernelGGL(zc::k_ris_roundtrip_mul_fast, dim3(grid_for(cnt)), dim3(zc::ZC_BLOCK), 0, D.s(), (const uint8_t*)d[0], (const u64*)d[1], (uint8_t*)d[2], (uint8_t*)d[3], (zc::u32*)D.fast, cnt);
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

// ---- fixed-base multiplication of the basepoint
static int base_table(DevState& D, const zc::u32** table)
{
    if (!D.base_table) {
        int rc = ensure(&D.base_table, &D.base_bytes, (size_t)zc::ZC_BASE_WINDOWS * 8 * 128);
        if (rc) return rc;
        hipLaunchKernelGGL(zc::k_base_table_build, dim3(1), dim3(64), 0, D.s(), (zc::u32*)D.base_table);
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

// ---- MSM: sum_i k_i * P_i (not in the reference).  Per GPU: bucket method (zc_msm.cuh) for
// shards of >= MSM_BUCKET_MIN_N pairs, otherwise batched scalar-mul + pairwise folds.  The
// per-device partial points are folded in device order on the first device.
int zc_msm(zc_ctx* ctx, const uint64_t* points, const uint64_t* scalars, size_t n, uint64_t* out_point)
{
    if (!ctx) return fail(ZC_ERR_BAD_ARG, "null context");
    REQUIRE(points); REQUIRE(scalars); REQUIRE(out_point);
    static const uint64_t ident[20] = {0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (n == 0) {
        memcpy(out_point, ident, sizeof ident);
        return ZC_OK;
    }
    Residency rp, rk;
    int dp = -1, dk = -1;
    residency_of(points, &rp, &dp);
    residency_of(scalars, &rk, &dk);
    if (rp != rk || (rp == RES_DEVICE && dp != dk)) return fail(ZC_ERR_MIXED_MEM, "points/scalars residency differs");
    std::lock_guard<std::mutex> lock(ctx->mu);

    const size_t ndev = (rp == RES_DEVICE) ? 1 : ctx->devs.size();
    const size_t per = (n + ndev - 1) / ndev;
    std::vector<DevState*> used;
    std::vector<const u64*> partial_ptr;
    for (size_t di = 0; di < ndev; di++) {
        const size_t lo = di * per, hi = std::min(n, lo + per);
        if (lo >= hi) break;
        const size_t cnt = hi - lo;
        DevState* ds = nullptr;
        if (rp == RES_DEVICE) {
            for (auto& x : ctx->devs)
                if (x.device == dp) ds = &x;
            if (!ds) return fail(ZC_ERR_MIXED_MEM, "device buffers do not belong to a device of this context");
        } else {
            ds = &ctx->devs[di];
        }
        HIP_TRY(hipSetDevice(ds->device));
        const u64 *dP, *dK;
        if (rp == RES_DEVICE) {
            dP = points;
            dK = scalars;
        } else {
            int rc = ensure(&ds->scratch[0], &ds->scratch_bytes[0], cnt * 160);
            if (rc) return rc;
            rc = ensure(&ds->scratch[1], &ds->scratch_bytes[1], cnt * 40);
            if (rc) return rc;
            HIP_TRY(hipMemcpyAsync(ds->scratch[0], points + 20 * lo, cnt * 160, hipMemcpyHostToDevice, ds->s()));
            HIP_TRY(hipMemcpyAsync(ds->scratch[1], scalars + 5 * lo, cnt * 40, hipMemcpyHostToDevice, ds->s()));
            dP = (const u64*)ds->scratch[0];
            dK = (const u64*)ds->scratch[1];
        }
        const u64* part = nullptr;
        int rc = msm_on_device(*ds, dP, dK, cnt, &part);
        if (rc) return rc;
        used.push_back(ds);
        partial_ptr.push_back(part);
    }
    std::vector<uint64_t> partials(used.size() * 20);
    for (size_t ui = 0; ui < used.size(); ui++) {
        HIP_TRY(hipSetDevice(used[ui]->device));
        HIP_TRY(hipMemcpyAsync(partials.data() + 20 * ui, partial_ptr[ui], 160, hipMemcpyDeviceToHost, used[ui]->s()));
    }
    for (DevState* ds : used) {
        HIP_TRY(hipSetDevice(ds->device));
        HIP_TRY(hipStreamSynchronize(ds->s()));
    }
    // fold the per-device partials in device order on the first device
    size_t cnt = used.size();
    if (cnt > 1) {
        DevState* ds = used[0];
        HIP_TRY(hipSetDevice(ds->device));
        int rc = ensure(&ds->tmp[0], &ds->tmp_bytes[0], cnt * 160);
        if (rc) return rc;
        rc = ensure(&ds->tmp[1], &ds->tmp_bytes[1], cnt * 160);
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(ds->tmp[0], partials.data(), cnt * 160, hipMemcpyHostToDevice, ds->s()));
        const u64* res = fold_all(*ds, (u64*)ds->tmp[0], (u64*)ds->tmp[1], cnt);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(partials.data(), res, 160, hipMemcpyDeviceToHost, ds->s()));
        HIP_TRY(hipStreamSynchronize(ds->s()));
    }
    memcpy(out_point, partials.data(), 160);
    return ZC_OK;
}

}  // extern "C"
