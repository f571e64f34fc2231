// Exclusive prefix sums on device (n+1 outputs).  256 threads x 16 items per block, two-level
// (recursive on block partials).  Bandwidth-bound helper; not on the critical path.
#include "dbg_internal.hpp"

namespace {
constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 16;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;

template <class T>
__device__ __forceinline__ T block_exclusive_scan(T v, T* total_out) {
    __shared__ T wave_sums[SCAN_THREADS / DBG_WAVE];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    T incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        T o = __shfl_up(incl, d);
        if (lane >= d) incl += o;
    }
    if (lane == 63) wave_sums[wave] = incl;
    __syncthreads();
    T base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < SCAN_THREADS / DBG_WAVE; w++) {
        T s = wave_sums[w];
        if (w < wave) base += s;
        tot += s;
    }
    *total_out = tot;
    __syncthreads();
    return base + incl - v;
}

template <class TIn, class TOut>
__global__ void __launch_bounds__(SCAN_THREADS) scan_partials_kernel(const TIn* __restrict__ in, uint64_t n,
                                                                     TOut* __restrict__ partial) {
    uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE;
    TOut s = 0;
    for (int i = 0; i < SCAN_ITEMS; i++) {
        uint64_t e = base + (uint64_t)i * SCAN_THREADS + threadIdx.x;
        if (e < n) s += (TOut)in[e];
    }
    TOut tot;
    (void)block_exclusive_scan<TOut>(s, &tot);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

template <class TIn, class TOut>
__global__ void __launch_bounds__(SCAN_THREADS) scan_apply_kernel(const TIn* __restrict__ in, uint64_t n,
                                                                  const TOut* __restrict__ partial_scanned,
                                                                  TOut* __restrict__ out) {
    uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * SCAN_ITEMS;
    TOut v[SCAN_ITEMS];
    TOut s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
        uint64_t e = base + i;
        v[i] = e < n ? (TOut)in[e] : (TOut)0;
        s += v[i];
    }
    TOut tot;
    TOut ex = block_exclusive_scan<TOut>(s, &tot);
    TOut off = (partial_scanned ? partial_scanned[blockIdx.x] : (TOut)0) + ex;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
        uint64_t e = base + i;
        if (e < n) out[e] = off;
        off += v[i];
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == SCAN_THREADS - 1) out[n] = off;   // grand total
}

template <class TIn, class TOut>
int scan_impl(dbg_ctx* ctx, const TIn* in, TOut* out, uint64_t n) {
    if (n == 0) {
        HIP_TRY(ctx, hipMemsetAsync(out, 0, sizeof(TOut), ctx->stream));
        return 0;
    }
    uint32_t nb = cdiv(n, SCAN_TILE);
    if (nb == 1) {
        scan_apply_kernel<TIn, TOut><<<1, SCAN_THREADS, 0, ctx->stream>>>(in, n, (const TOut*)nullptr, out);
        LAUNCH_CHECK(ctx, "scan_apply");
        return 0;
    }
    DBuf<TOut> partial, partial_scanned;
    ALLOC_OR_FAIL(ctx, partial, nb);
    ALLOC_OR_FAIL(ctx, partial_scanned, (size_t)nb + 1);
    scan_partials_kernel<TIn, TOut><<<nb, SCAN_THREADS, 0, ctx->stream>>>(in, n, partial.p);
    LAUNCH_CHECK(ctx, "scan_partials");
    DBG_TRY((scan_impl<TOut, TOut>(ctx, partial.p, partial_scanned.p, nb)));
    scan_apply_kernel<TIn, TOut><<<nb, SCAN_THREADS, 0, ctx->stream>>>(in, n, partial_scanned.p, out);
    LAUNCH_CHECK(ctx, "scan_apply");
    return 0;
}
}  // namespace

int scan_exclusive_u32(dbg_ctx* ctx, const uint32_t* in, uint32_t* out, uint64_t n) {
    return scan_impl<uint32_t, uint32_t>(ctx, in, out, n);
}
int scan_exclusive_u32_u64(dbg_ctx* ctx, const uint32_t* in, uint64_t* out, uint64_t n) {
    return scan_impl<uint32_t, uint64_t>(ctx, in, out, n);
}
int scan_exclusive_u64(dbg_ctx* ctx, const uint64_t* in, uint64_t* out, uint64_t n) {
    return scan_impl<uint64_t, uint64_t>(ctx, in, out, n);
}
