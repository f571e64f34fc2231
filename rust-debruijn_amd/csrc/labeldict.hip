// Full-width D1 labels (round 6).  CountFilterSet<D: Ord> takes any u32 (src/filter.rs:68-101): hashes, barcodes, 32-bit ids.  The
// device routes pack a label next to other fields (24 bits in the generic path's payload word and in the label lists' sort keys), so
// labels of 2^24 and more used to be refused.  Now a call whose largest label does not fit is served through an order-preserving
// dictionary: the distinct labels of the call are ranked (label -> its index among the sorted distinct labels), every route runs on the
// ranks -- sorted, de-duplicated rank lists are sorted, de-duplicated label lists (filter.rs:97-98) -- and the output's set_val is
// translated back.  At most 2^24 distinct labels per call.
//   presence bitmap over the 2^32 values (512 MB) -> popcounts per block of 1024 values -> exclusive scan = rank of the block's first
//   value; rank(v) = block base + the set bits of the block below v; dictionary[rank] = v from one sweep over the bitmap.
#include "dbg_internal.hpp"

namespace {
constexpr uint32_t LD_BLOCK_WORDS = 16;                      // 64-bit words per block: 1024 values
constexpr uint64_t LD_WORDS = 1ull << 26;                    // 2^32 bits
constexpr uint32_t LD_BLOCKS = (uint32_t)(LD_WORDS / LD_BLOCK_WORDS);

__global__ void __launch_bounds__(256) ld_mark_kernel(const uint32_t* __restrict__ lab, uint64_t n, unsigned long long* __restrict__ bm) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint32_t v = lab[i];
        const unsigned long long bit = 1ull << (v & 63u);
        if (!(bm[v >> 6] & bit)) atomicOr(&bm[v >> 6], bit);      // (a call has few distinct labels next to its reads: most bits are set already)
    }
}
__global__ void __launch_bounds__(256) ld_block_count_kernel(const unsigned long long* __restrict__ bm, uint32_t* __restrict__ cnt) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= LD_BLOCKS) return;
    const ulonglong2* w = reinterpret_cast<const ulonglong2*>(bm + (uint64_t)b * LD_BLOCK_WORDS);
    uint32_t s = 0;
#pragma unroll
    for (uint32_t q = 0; q < LD_BLOCK_WORDS / 2; q++) { const ulonglong2 x = w[q]; s += (uint32_t)__popcll(x.x) + (uint32_t)__popcll(x.y); }
    cnt[b] = s;
}
__global__ void __launch_bounds__(256) ld_fill_kernel(const unsigned long long* __restrict__ bm, const uint32_t* __restrict__ base, uint32_t* __restrict__ dict) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= LD_BLOCKS) return;
    uint32_t r = base[b];
    if (base[b + 1] == r) return;                                   // nothing in this block
    for (uint32_t q = 0; q < LD_BLOCK_WORDS; q++) {
        unsigned long long x = bm[(uint64_t)b * LD_BLOCK_WORDS + q];
        while (x) {
            const uint32_t bit = (uint32_t)__ffsll((long long)x) - 1u;
            dict[r++] = (b * LD_BLOCK_WORDS + q) * 64u + bit;
            x &= x - 1;
        }
    }
}
__global__ void __launch_bounds__(256) ld_rank_kernel(const uint32_t* __restrict__ lab, uint64_t n, const unsigned long long* __restrict__ bm,
                                                      const uint32_t* __restrict__ base, uint32_t* __restrict__ rank) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint32_t v = lab[i], b = v >> 10, wq = (v >> 6) & (LD_BLOCK_WORDS - 1);
        uint32_t r = base[b];
        for (uint32_t q = 0; q < wq; q++) r += (uint32_t)__popcll(bm[(uint64_t)b * LD_BLOCK_WORDS + q]);
        r += (uint32_t)__popcll(bm[(uint64_t)b * LD_BLOCK_WORDS + wq] & ((1ull << (v & 63u)) - 1ull));
        rank[i] = r;
    }
}
__global__ void __launch_bounds__(256) ld_translate_kernel(uint32_t* __restrict__ val, uint64_t n, const uint32_t* __restrict__ dict, uint32_t n_dict,
                                                           uint32_t* __restrict__ bad) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint32_t r = val[i];
        if (r < n_dict) val[i] = dict[r]; else *bad = 1u;
    }
}
}  // namespace

// the dictionary of a label array: presence bitmap, block bases, rank -> label table
int label_dict_make(dbg_ctx* c, const uint32_t* labels, uint64_t n, LabelDict* d) {
    DBuf<uint32_t> cnt;
    ALLOC_OR_FAIL(c, d->bm, LD_WORDS);
    ALLOC_OR_FAIL(c, cnt, LD_BLOCKS);
    ALLOC_OR_FAIL(c, d->base, (size_t)LD_BLOCKS + 1);
    HIP_TRY(c, hipMemsetAsync(d->bm.p, 0, LD_WORDS * 8, c->stream));
    c->t_begin("label_dict", n);
    const uint32_t grid = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(cdiv(n, 256), 256ull * 32));
    if (n) { ld_mark_kernel<<<grid, 256, 0, c->stream>>>(labels, n, d->bm.p); LAUNCH_CHECK(c, "ld_mark"); }
    ld_block_count_kernel<<<cdiv(LD_BLOCKS, 256), 256, 0, c->stream>>>(d->bm.p, cnt.p);
    LAUNCH_CHECK(c, "ld_block_count");
    DBG_TRY(scan_exclusive_u32(c, cnt.p, d->base.p, LD_BLOCKS));
    uint32_t nd = 0;
    HIP_TRY(c, hipMemcpyAsync(&nd, d->base.p + LD_BLOCKS, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (nd >= (1u << 24)) { c->t_end(); return c->fail(17, "CountFilterSet: more than 2^24 - 1 distinct D1 values in one call"); }
    ALLOC_OR_FAIL(c, d->dict, std::max<uint32_t>(nd, 1));
    ld_fill_kernel<<<cdiv(LD_BLOCKS, 256), 256, 0, c->stream>>>(d->bm.p, d->base.p, d->dict.p);
    c->t_end();
    LAUNCH_CHECK(c, "ld_fill");
    HIP_TRY(c, hipStreamSynchronize(c->stream));                   // (cnt goes back to the pool)
    d->n = nd;
    return 0;
}

// rank_out[i] = index of labels[i] among the dictionary's labels (every label must be in the dictionary)
int label_dict_rank(dbg_ctx* c, const LabelDict& d, const uint32_t* labels, uint64_t n, DBuf<uint32_t>* rank_out) {
    ALLOC_OR_FAIL(c, *rank_out, std::max<uint64_t>(n, 1));
    if (!n) return 0;
    c->t_begin("label_dict_rank", n);
    ld_rank_kernel<<<(uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(cdiv(n, 256), 256ull * 32)), 256, 0, c->stream>>>(labels, n, d.bm.p, d.base.p, rank_out->p);
    c->t_end();
    LAUNCH_CHECK(c, "ld_rank");
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return 0;
}

// ranks of the n labels (u32) among the distinct labels of the call, and the dictionary rank -> label
int label_dict_build(dbg_ctx* c, const uint32_t* labels, uint64_t n, DBuf<uint32_t>* rank_out, DBuf<uint32_t>* dict_out, uint32_t* n_distinct) {
    LabelDict d;
    DBG_TRY(label_dict_make(c, labels, n, &d));
    DBG_TRY(label_dict_rank(c, d, labels, n, rank_out));
    *dict_out = std::move(d.dict);
    *n_distinct = d.n;
    return 0;
}

// set_val[i] = dict[set_val[i]]
int label_dict_translate(dbg_ctx* c, uint32_t* set_val, uint64_t n, const uint32_t* dict, uint32_t n_dict) {
    if (!n) return 0;
    DBuf<uint32_t> bad;
    ALLOC_OR_FAIL(c, bad, 1);
    HIP_TRY(c, hipMemsetAsync(bad.p, 0, 4, c->stream));
    c->t_begin("label_dict_back", n);
    ld_translate_kernel<<<(uint32_t)std::min<uint64_t>(cdiv(n, 256), 256ull * 64), 256, 0, c->stream>>>(set_val, n, dict, n_dict, bad.p);
    c->t_end();
    LAUNCH_CHECK(c, "ld_translate");
    uint32_t hb = 0;
    HIP_TRY(c, hipMemcpyAsync(&hb, bad.p, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (hb) return c->fail(165, "label dictionary: a label rank outside the dictionary came back from the counting stage");
    return 0;
}
