// group_by(key) + KmerSummarizer::summarize over key-sorted records in HBM (generic path):
//   CountFilter    (filter.rs:53-62):  u16 saturating count, OR of Exts, valid iff count >= min
//   CountFilterSet (filter.rs:85-100): sorted de-duplicated D1 list (CSR), OR of Exts, valid iff nobs >= min
// followed by compaction of the valid groups into the ascending-key output table
// (the vectors handed to BoomHashMap2::new, filter.rs:213-217, :227-230).
#include "dbg_internal.hpp"

namespace {

template <bool HAS_HI>
__global__ void mark_heads_kernel(RecArrays r, uint32_t n, uint32_t* __restrict__ head) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    bool h = i == 0;
    if (!h) {
        h = r.lo[i] != r.lo[i - 1];
        if (HAS_HI) h = h || (r.hi[i] != r.hi[i - 1]);
    }
    head[i] = h ? 1u : 0u;
}

// one thread per group head walks its run
template <bool HAS_HI, bool IS_SET>
__global__ void group_reduce_kernel(RecArrays r, uint32_t n, const uint32_t* __restrict__ head,
                                    const uint32_t* __restrict__ gid, uint64_t min_obs,
                                    uint64_t* __restrict__ gkey_hi, uint64_t* __restrict__ gkey_lo,
                                    uint8_t* __restrict__ gexts, uint32_t* __restrict__ gnobs,
                                    uint32_t* __restrict__ gsetn, uint32_t* __restrict__ gstart,
                                    uint32_t* __restrict__ gvalid) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !head[i]) return;
    uint32_t g = gid[i];
    uint32_t nobs = 0, ex = 0, setn = 0, prev = 0xffffffffu;
    uint32_t j = i;
    do {
        uint32_t p = r.pay[j];
        ex |= p & 0xffu;
        nobs++;
        if (IS_SET) {
            uint32_t d = p >> 8;
            if (d != prev) { setn++; prev = d; }       // records arrive sorted by (key, D1)
        }
        j++;
    } while (j < n && !head[j]);
    if (HAS_HI) gkey_hi[g] = r.hi[i];
    gkey_lo[g] = r.lo[i];
    gexts[g] = (uint8_t)ex;
    gnobs[g] = nobs;
    gstart[g] = i;
    bool valid;
    if (IS_SET) { gsetn[g] = setn; valid = (uint64_t)nobs >= min_obs; }
    else { uint32_t c = nobs > 65535u ? 65535u : nobs; valid = (uint64_t)c >= min_obs; }   // saturating u16 (filter.rs:57,61)
    gvalid[g] = valid ? 1u : 0u;
    if (IS_SET) gsetn[g] = valid ? setn : 0u;
}

template <bool HAS_HI, bool IS_SET>
__global__ void compact_kernel(uint32_t n_groups, RecArrays r, uint32_t n,
                               const uint64_t* __restrict__ gkey_hi, const uint64_t* __restrict__ gkey_lo,
                               const uint8_t* __restrict__ gexts, const uint32_t* __restrict__ gnobs,
                               const uint32_t* __restrict__ gstart, const uint32_t* __restrict__ gvalid,
                               const uint32_t* __restrict__ vpos, const uint64_t* __restrict__ setpos,
                               uint64_t* __restrict__ o_hi, uint64_t* __restrict__ o_lo, uint8_t* __restrict__ o_exts,
                               uint16_t* __restrict__ o_count, uint64_t* __restrict__ o_set_off,
                               uint32_t* __restrict__ o_set_val) {
    uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_groups || !gvalid[g]) return;
    uint32_t o = vpos[g];
    if (HAS_HI) o_hi[o] = gkey_hi[g]; else o_hi[o] = 0;
    o_lo[o] = gkey_lo[g];
    o_exts[o] = gexts[g];
    if (!IS_SET) {
        uint32_t c = gnobs[g];
        o_count[o] = (uint16_t)(c > 65535u ? 65535u : c);
    } else {
        uint64_t so = setpos[g];
        o_set_off[o] = so;
        uint32_t prev = 0xffffffffu;
        uint32_t j = gstart[g], e = j + gnobs[g];
        for (; j < e; j++) {
            uint32_t d = r.pay[j] >> 8;
            if (d != prev) { o_set_val[so++] = d; prev = d; }
        }
    }
}

__global__ void set_u64_kernel(uint64_t* p, const uint64_t* src) { *p = *src; }
}  // namespace

int reduce_sorted_records(dbg_ctx* ctx, uint64_t n64, RecArrays r, bool has_hi, int summarizer, uint64_t min_obs,
                          bool report_all, ReduceOut* out) {
    *out = ReduceOut();
    const bool is_set = summarizer == DBG_COUNT_FILTER_SET;
    const uint32_t n = (uint32_t)n64;
    DBuf<uint64_t> o_hi, o_lo, o_set_off, all_hi, all_lo;
    DBuf<uint8_t> o_exts;
    DBuf<uint16_t> o_count;
    DBuf<uint32_t> o_set_val;
    uint64_t n_valid = 0, n_groups = 0, n_setval = 0;
    if (n > 0) {
        DBuf<uint32_t> head, gid;
        ALLOC_OR_FAIL(ctx, head, n);
        ALLOC_OR_FAIL(ctx, gid, (size_t)n + 1);
        ctx->t_begin("reduce_mark_heads", n);
        if (has_hi) mark_heads_kernel<true><<<cdiv(n, 256), 256, 0, ctx->stream>>>(r, n, head.p);
        else        mark_heads_kernel<false><<<cdiv(n, 256), 256, 0, ctx->stream>>>(r, n, head.p);
        ctx->t_end();
        LAUNCH_CHECK(ctx, "mark_heads");
        DBG_TRY(scan_exclusive_u32(ctx, head.p, gid.p, n));
        uint32_t ng32 = 0;
        HIP_TRY(ctx, hipMemcpyAsync(&ng32, gid.p + n, 4, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        n_groups = ng32;
        DBuf<uint64_t> gkey_hi, gkey_lo;
        DBuf<uint8_t> gexts;
        DBuf<uint32_t> gnobs, gsetn, gstart, gvalid, vpos;
        DBuf<uint64_t> setpos;
        if (has_hi) ALLOC_OR_FAIL(ctx, gkey_hi, n_groups);
        ALLOC_OR_FAIL(ctx, gkey_lo, n_groups);
        ALLOC_OR_FAIL(ctx, gexts, n_groups);
        ALLOC_OR_FAIL(ctx, gnobs, n_groups);
        ALLOC_OR_FAIL(ctx, gstart, n_groups);
        ALLOC_OR_FAIL(ctx, gvalid, n_groups);
        ALLOC_OR_FAIL(ctx, vpos, n_groups + 1);
        if (is_set) { ALLOC_OR_FAIL(ctx, gsetn, n_groups); ALLOC_OR_FAIL(ctx, setpos, n_groups + 1); }
        ctx->t_begin("reduce_groups", n);
#define GO(HH, SS) group_reduce_kernel<HH, SS><<<cdiv(n, 256), 256, 0, ctx->stream>>>(                     \
        r, n, head.p, gid.p, min_obs, gkey_hi.p, gkey_lo.p, gexts.p, gnobs.p, gsetn.p, gstart.p, gvalid.p)
        if (has_hi) { if (is_set) GO(true, true); else GO(true, false); }
        else        { if (is_set) GO(false, true); else GO(false, false); }
#undef GO
        ctx->t_end();
        LAUNCH_CHECK(ctx, "group_reduce");
        DBG_TRY(scan_exclusive_u32(ctx, gvalid.p, vpos.p, n_groups));
        if (is_set) DBG_TRY(scan_exclusive_u32_u64(ctx, gsetn.p, setpos.p, n_groups));
        uint32_t nv32 = 0;
        HIP_TRY(ctx, hipMemcpyAsync(&nv32, vpos.p + n_groups, 4, hipMemcpyDeviceToHost, ctx->stream));
        if (is_set) HIP_TRY(ctx, hipMemcpyAsync(&n_setval, setpos.p + n_groups, 8, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        n_valid = nv32;
        ALLOC_OR_FAIL(ctx, o_hi, n_valid);
        ALLOC_OR_FAIL(ctx, o_lo, n_valid);
        ALLOC_OR_FAIL(ctx, o_exts, n_valid);
        if (is_set) { ALLOC_OR_FAIL(ctx, o_set_off, n_valid + 1); ALLOC_OR_FAIL(ctx, o_set_val, n_setval); }
        else ALLOC_OR_FAIL(ctx, o_count, n_valid);
        ctx->t_begin("reduce_compact", n_groups);
#define GO(HH, SS) compact_kernel<HH, SS><<<cdiv(n_groups, 256), 256, 0, ctx->stream>>>(                    \
        (uint32_t)n_groups, r, n, gkey_hi.p, gkey_lo.p, gexts.p, gnobs.p, gstart.p, gvalid.p, vpos.p, setpos.p, \
        o_hi.p, o_lo.p, o_exts.p, o_count.p, o_set_off.p, o_set_val.p)
        if (n_groups) {
            if (has_hi) { if (is_set) GO(true, true); else GO(true, false); }
            else        { if (is_set) GO(false, true); else GO(false, false); }
        }
#undef GO
        ctx->t_end();
        LAUNCH_CHECK(ctx, "compact");
        if (is_set) {
            set_u64_kernel<<<1, 1, 0, ctx->stream>>>(o_set_off.p + n_valid, setpos.p + n_groups);
            LAUNCH_CHECK(ctx, "set_u64");
        }
        if (report_all) {
            if (has_hi) all_hi = std::move(gkey_hi);
            else { ALLOC_OR_FAIL(ctx, all_hi, n_groups); HIP_TRY(ctx, hipMemsetAsync(all_hi.p, 0, n_groups * 8, ctx->stream)); }
            all_lo = std::move(gkey_lo);
        }
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    } else {
        ALLOC_OR_FAIL(ctx, o_hi, 0); ALLOC_OR_FAIL(ctx, o_lo, 0); ALLOC_OR_FAIL(ctx, o_exts, 0);
        if (is_set) { ALLOC_OR_FAIL(ctx, o_set_off, 1); HIP_TRY(ctx, hipMemsetAsync(o_set_off.p, 0, 8, ctx->stream)); ALLOC_OR_FAIL(ctx, o_set_val, 0); }
        else ALLOC_OR_FAIL(ctx, o_count, 0);
    }
    out->n_valid = n_valid;
    out->n_all = report_all ? n_groups : 0;
    out->n_set_val = n_setval;
    out->key_hi = o_hi.take(); out->key_lo = o_lo.take(); out->exts = o_exts.take();
    out->count = o_count.take(); out->set_off = o_set_off.take(); out->set_val = o_set_val.take();
    out->all_hi = all_hi.take(); out->all_lo = all_lo.take();
    return 0;
}
