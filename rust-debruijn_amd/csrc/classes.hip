// CountFilterSet payload -> ScmapCompress data (BASELINE config 5: "ScmapCompress with per-kmer colour payload").
//
// In the reference the index handed to compress_kmers_with_hash after filter_kmers(CountFilterSet) is a
// BoomHashMap2<K, Exts, Vec<D1>> (src/filter.rs:68-101), and ScmapCompress<Vec<D1>>::join_test is `d1 == d2` on those
// label lists (src/compression.rs:68-98); reduce returns the common list.  Only equality of lists is ever used, so a
// dense class id per distinct list carries exactly the same information: equal lists <=> equal ids, and the data of a
// node is the id of the list all its k-mers share.  The class table (id -> list, CSR) goes back to the caller so that a
// node's Vec<D1> can be materialised.
//
// Device formulation: one 64-bit key per k-mer -- the exact colour bitmask when every label is < 64, otherwise a hash of
// (length, labels) that is verified exactly below -- sorted with the k-mer index as payload; heads of runs of equal keys
// are classes; ids are ranks in ascending key order (for masks: ascending mask value).
#include "dbg_internal.hpp"
#include <algorithm>

namespace {

__device__ __forceinline__ uint64_t mix64(uint64_t h) {
    h ^= h >> 33; h *= 0xff51afd7ed558ccdull; h ^= h >> 33; h *= 0xc4ceb9fe1a85ec53ull; h ^= h >> 33;
    return h;
}

__global__ void __launch_bounds__(256) set_max_kernel(const uint32_t* __restrict__ v, uint64_t n, uint32_t* __restrict__ out) {
    uint32_t m = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) m = max(m, v[i]);
    for (int d = 32; d; d >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, d));
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}

__global__ void __launch_bounds__(256) set_key_kernel(const uint64_t* __restrict__ set_off, const uint32_t* __restrict__ set_val, uint32_t n,
                                                      int exact, uint64_t* __restrict__ key, uint32_t* __restrict__ idx) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t a = set_off[i], b = set_off[i + 1];
    uint64_t h;
    if (exact) {
        h = 0;
        for (uint64_t j = a; j < b; j++) h |= 1ull << set_val[j];
    } else {
        h = mix64(b - a);
        for (uint64_t j = a; j < b; j++) h = mix64(h ^ ((uint64_t)set_val[j] + 0x9E3779B97F4A7C15ull));
    }
    key[i] = h;
    idx[i] = i;
}

// heads of the runs of equal keys; with hashed keys, neighbours that share a key must hold the same list
__global__ void __launch_bounds__(256) class_heads_kernel(const uint64_t* __restrict__ key, const uint32_t* __restrict__ idx, uint32_t n, int exact,
                                                          const uint64_t* __restrict__ set_off, const uint32_t* __restrict__ set_val,
                                                          uint32_t* __restrict__ head, uint32_t* __restrict__ flag) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const bool h = j == 0 || key[j] != key[j - 1];
    head[j] = h ? 1u : 0u;
    if (!h && !exact) {
        const uint32_t x = idx[j], y = idx[j - 1];
        const uint64_t ax = set_off[x], bx = set_off[x + 1], ay = set_off[y], by = set_off[y + 1];
        bool same = bx - ax == by - ay;
        for (uint64_t q = 0; same && q < bx - ax; q++) same = set_val[ax + q] == set_val[ay + q];
        if (!same) atomicOr(flag, 1u);
    }
}

__global__ void __launch_bounds__(256) class_assign_kernel(const uint32_t* __restrict__ idx, const uint32_t* __restrict__ head,
                                                           const uint32_t* __restrict__ head_excl, uint32_t n,
                                                           const uint64_t* __restrict__ set_off, uint32_t* __restrict__ cls_out,
                                                           uint32_t* __restrict__ rep, uint32_t* __restrict__ rep_len) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const uint32_t c = head_excl[j] + head[j] - 1u, i = idx[j];
    cls_out[i] = c;
    if (head[j]) { rep[c] = i; rep_len[c] = (uint32_t)(set_off[i + 1] - set_off[i]); }
}

__global__ void __launch_bounds__(256) class_table_kernel(const uint32_t* __restrict__ rep, const uint64_t* __restrict__ cls_off, uint32_t n_cls,
                                                          const uint64_t* __restrict__ set_off, const uint32_t* __restrict__ set_val,
                                                          uint32_t* __restrict__ out_val) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_cls) return;
    const uint64_t a = set_off[rep[c]], o = cls_off[c], len = cls_off[c + 1] - o;
    for (uint64_t q = 0; q < len; q++) out_val[o + q] = set_val[a + q];
}

}  // namespace

// class_dev[n] <- dense class id of every k-mer's label list; classes (host CSR, may be null) <- id -> list
int label_classes_device(dbg_ctx* c, uint64_t n64, const uint64_t* set_off_dev, const uint32_t* set_val_dev, uint64_t n_set_val,
                         uint32_t* class_dev, dbg_label_classes* classes) {
    if (classes) memset(classes, 0, sizeof(*classes));
    if (n64 >= (1ull << 32)) return c->fail(160, "label classes: more than 2^32-1 k-mers in one table");
    const uint32_t n = (uint32_t)n64;
    if (!n) return 0;
    DBuf<uint32_t> d_max, d_flag;
    ALLOC_OR_FAIL(c, d_max, 1); ALLOC_OR_FAIL(c, d_flag, 1);
    HIP_TRY(c, hipMemsetAsync(d_max.p, 0, 4, c->stream));
    HIP_TRY(c, hipMemsetAsync(d_flag.p, 0, 4, c->stream));
    if (n_set_val) {
        set_max_kernel<<<(uint32_t)std::min<uint64_t>(cdiv(n_set_val, 256), 4096), 256, 0, c->stream>>>(set_val_dev, n_set_val, d_max.p);
        LAUNCH_CHECK(c, "set_max");
    }
    uint32_t mx = 0;
    HIP_TRY(c, hipMemcpyAsync(&mx, d_max.p, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    const int exact = mx < 64 ? 1 : 0;
    DBuf<uint64_t> ka, kb;
    DBuf<uint32_t> ia, ib, head, head_sc, rep, rep_len;
    ALLOC_OR_FAIL(c, ka, n); ALLOC_OR_FAIL(c, kb, n); ALLOC_OR_FAIL(c, ia, n); ALLOC_OR_FAIL(c, ib, n);
    c->t_begin("label_classes", n);
    set_key_kernel<<<cdiv(n, 256), 256, 0, c->stream>>>(set_off_dev, set_val_dev, n, exact, ka.p, ia.p);
    LAUNCH_CHECK(c, "set_key");
    bool in_b = false;
    // masks of labels < 8/16/32 need only the low key bits sorted
    const int key_bits = exact ? (mx < 8 ? 8 : (mx < 16 ? 16 : (mx < 32 ? 32 : 64))) : 64;
    DBG_TRY(radix_sort_records(c, n, RecArrays{nullptr, ka.p, ia.p}, RecArrays{nullptr, kb.p, ib.p}, key_bits, 0, 0, &in_b));
    const uint64_t* keys = in_b ? kb.p : ka.p;
    const uint32_t* idx = in_b ? ib.p : ia.p;
    ALLOC_OR_FAIL(c, head, n); ALLOC_OR_FAIL(c, head_sc, (size_t)n + 1);
    class_heads_kernel<<<cdiv(n, 256), 256, 0, c->stream>>>(keys, idx, n, exact, set_off_dev, set_val_dev, head.p, d_flag.p);
    LAUNCH_CHECK(c, "class_heads");
    DBG_TRY(scan_exclusive_u32(c, head.p, head_sc.p, n));
    uint32_t n_cls = 0, fl = 0;
    HIP_TRY(c, hipMemcpyAsync(&n_cls, head_sc.p + n, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(&fl, d_flag.p, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (fl) return c->fail(161, "label classes: two different label lists share a 64-bit hash");
    ALLOC_OR_FAIL(c, rep, n_cls); ALLOC_OR_FAIL(c, rep_len, n_cls);
    class_assign_kernel<<<cdiv(n, 256), 256, 0, c->stream>>>(idx, head.p, head_sc.p, n, set_off_dev, class_dev, rep.p, rep_len.p);
    c->t_end();
    LAUNCH_CHECK(c, "class_assign");
    if (classes) {
        DBuf<uint64_t> cls_off;
        DBuf<uint32_t> cls_val;
        ALLOC_OR_FAIL(c, cls_off, (size_t)n_cls + 1);
        DBG_TRY(scan_exclusive_u32_u64(c, rep_len.p, cls_off.p, n_cls));
        uint64_t tot = 0;
        HIP_TRY(c, hipMemcpyAsync(&tot, cls_off.p + n_cls, 8, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        ALLOC_OR_FAIL(c, cls_val, std::max<uint64_t>(tot, 1));
        class_table_kernel<<<cdiv(n_cls, 256), 256, 0, c->stream>>>(rep.p, cls_off.p, n_cls, set_off_dev, set_val_dev, cls_val.p);
        LAUNCH_CHECK(c, "class_table");
        classes->n_classes = n_cls;
        classes->n_set_val = tot;
        classes->set_off = (uint64_t*)malloc(((size_t)n_cls + 1) * 8);
        classes->set_val = (uint32_t*)malloc(std::max<size_t>((size_t)tot * 4, 4));
        if (!classes->set_off || !classes->set_val) { dbg_free_label_classes(classes); return c->fail(101, "out of host memory"); }
        hipError_t e = hipMemcpyAsync(classes->set_off, cls_off.p, ((size_t)n_cls + 1) * 8, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess && tot) e = hipMemcpyAsync(classes->set_val, cls_val.p, (size_t)tot * 4, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) { dbg_free_label_classes(classes); return c->fail(100, std::string("HIP error in label classes: ") + hipGetErrorString(e)); }
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" void dbg_free_label_classes(dbg_label_classes* cl) {
    if (!cl) return;
    free(cl->set_off); free(cl->set_val);
    memset(cl, 0, sizeof(*cl));
}

extern "C" int dbg_label_classes_dev(dbg_ctx* c, uint64_t n, const uint64_t* set_off_dev, const uint32_t* set_val_dev, uint64_t n_set_val,
                                     uint32_t* class_dev, dbg_label_classes* classes) {
    if (n && (!set_off_dev || !class_dev || (n_set_val && !set_val_dev))) return c->fail(10, "null argument");
    HIP_TRY(c, hipSetDevice(c->device));
    c->t_clear();
    return label_classes_device(c, n, set_off_dev, set_val_dev, n_set_val, class_dev, classes);
}

// compress_kmers_with_hash on the table filter_kmers left in HBM: D = the count column (CountFilter) or the class of the
// label list (CountFilterSet)
extern "C" int dbg_compress_table_dev(dbg_ctx* c, uint32_t k, int stranded, int spec, const dbg_kmer_table* t, dbg_graph* out,
                                      dbg_label_classes* classes) {
    if (!t || !out) return c->fail(10, "null argument");
    if (!t->on_device) return c->fail(162, "dbg_compress_table_dev needs a device-resident table");
    if (classes) memset(classes, 0, sizeof(*classes));
    HIP_TRY(c, hipSetDevice(c->device));
    if (!t->set_off) return dbg_compress_kmers_with_hash_dev(c, k, stranded, spec, t->n, t->key_hi, t->key_lo, t->exts, nullptr, t->count, out);
    DBuf<uint32_t> cls;
    ALLOC_OR_FAIL(c, cls, std::max<uint64_t>(t->n, 1));
    DBG_TRY(label_classes_device(c, t->n, t->set_off, t->set_val, t->n_set_val, cls.p, classes));
    const int r = dbg_compress_kmers_with_hash_dev(c, k, stranded, spec, t->n, t->key_hi, t->key_lo, t->exts, cls.p, nullptr, out);
    if (r && classes) dbg_free_label_classes(classes);
    return r;
}
