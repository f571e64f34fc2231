// Device-side building blocks shared by every kernel: 2-bit k-mers as (hi, lo) u64 pairs,
// Exts bytes, packed DnaString access.  gfx950 only (wave64).
//
// Semantics follow the reference crate (paths under /root/reference/src):
//   key layout  : kmer.rs:429-437  (right-aligned, base 0 = most significant pair)
//   rc          : kmer.rs:620-634  (!reverse_by_twos >> 2*(cap-k))
//   min_rc_flip : lib.rs:224-231   (ties return (rc, true))
//   Exts        : lib.rs:569-749
//   DnaString   : dna_string.rs:383-399 (base i -> word i/32, bits 63-2(i%32)..62-2(i%32))
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DBG_WAVE 64

struct K128 {
    uint64_t hi, lo;
};

__host__ __device__ __forceinline__ bool k128_lt(K128 a, K128 b) { return a.hi < b.hi || (a.hi == b.hi && a.lo < b.lo); }
__host__ __device__ __forceinline__ bool k128_eq(K128 a, K128 b) { return a.hi == b.hi && a.lo == b.lo; }

// logical shifts of a 128-bit value by 0..128
__host__ __device__ __forceinline__ K128 k128_shr(K128 a, int s) {
    if (s == 0) return a;
    if (s >= 128) return K128{0, 0};
    if (s >= 64) return K128{0, a.hi >> (s - 64)};
    return K128{a.hi >> s, (a.lo >> s) | (a.hi << (64 - s))};
}
__host__ __device__ __forceinline__ K128 k128_shl(K128 a, int s) {
    if (s == 0) return a;
    if (s >= 128) return K128{0, 0};
    if (s >= 64) return K128{a.lo << (s - 64), 0};
    return K128{(a.hi << s) | (a.lo >> (64 - s)), a.lo << s};
}
// mask of the low 2k bits
__host__ __device__ __forceinline__ K128 k128_mask(int k) {
    int bits = 2 * k;
    if (bits >= 128) return K128{~0ull, ~0ull};
    if (bits > 64) return K128{(1ull << (bits - 64)) - 1, ~0ull};
    if (bits == 64) return K128{0, ~0ull};
    return K128{0, (1ull << bits) - 1};
}

// reverse the order of the 32 two-bit units of a u64 (kmer.rs:141-160), via bit reverse + pair fix
__device__ __forceinline__ uint64_t rev2_64(uint64_t x) {
    x = __brevll(x);
    return ((x & 0x5555555555555555ull) << 1) | ((x >> 1) & 0x5555555555555555ull);
}
__host__ inline uint64_t rev2_64_host(uint64_t s) {
    uint64_t r = ((s & 0x3333333333333333ull) << 2) | ((s >> 2) & 0x3333333333333333ull);
    r = ((r & 0x0F0F0F0F0F0F0F0Full) << 4) | ((r >> 4) & 0x0F0F0F0F0F0F0F0Full);
    r = ((r & 0x00FF00FF00FF00FFull) << 8) | ((r >> 8) & 0x00FF00FF00FF00FFull);
    r = ((r & 0x0000FFFF0000FFFFull) << 16) | ((r >> 16) & 0x0000FFFF0000FFFFull);
    r = ((r & 0x00000000FFFFFFFFull) << 32) | ((r >> 32) & 0x00000000FFFFFFFFull);
    return r;
}

// reverse complement of a right-aligned k-mer (kmer.rs:620-634)
__device__ __forceinline__ K128 kmer_rc(K128 a, int k) {
    K128 r{~rev2_64(a.lo), ~rev2_64(a.hi)};
    return k128_shr(r, 128 - 2 * k);
}
__host__ inline K128 kmer_rc_host(K128 a, int k) {
    K128 r{~rev2_64_host(a.lo), ~rev2_64_host(a.hi)};
    return k128_shr(r, 128 - 2 * k);
}

// extend_right (kmer.rs:479-487) / extend_left (kmer.rs:469-477)
__host__ __device__ __forceinline__ K128 kmer_extend_right(K128 a, int k, uint32_t b) {
    K128 s = k128_shl(a, 2);
    K128 m = k128_mask(k);
    s.hi &= m.hi; s.lo &= m.lo;
    s.lo |= (uint64_t)b;
    return s;
}
__host__ __device__ __forceinline__ K128 kmer_extend_left(K128 a, int k, uint32_t b) {
    K128 s = k128_shr(a, 2);
    K128 t = k128_shl(K128{0, (uint64_t)b}, 2 * (k - 1));
    s.hi |= t.hi; s.lo |= t.lo;
    return s;
}
// base at position pos (kmer.rs:574-577)
__host__ __device__ __forceinline__ uint32_t kmer_get(K128 a, int k, int pos) {
    return (uint32_t)(k128_shr(a, 2 * (k - 1 - pos)).lo & 3ull);
}

// Exts::rc (lib.rs:729-748): swap nibbles, then reverse the 4 bits inside each nibble
__host__ __device__ __forceinline__ uint32_t exts_rc(uint32_t v) {
    uint32_t r = ((v & 0xf) << 4) | (v >> 4);
    r = ((r & 0x55u) << 1) | ((r >> 1) & 0x55u);
    r = ((r & 0x33u) << 2) | ((r >> 2) & 0x33u);
    return r & 0xff;
}
__host__ __device__ __forceinline__ uint32_t exts_complement(uint32_t v) {    // lib.rs:729-738
    uint32_t r = ((v & 0x55u) << 1) | ((v >> 1) & 0x55u);
    r = ((r & 0x33u) << 2) | ((r >> 2) & 0x33u);
    return r & 0xff;
}

// 2-bit base at absolute base offset o of a packed word stream (dna_string.rs:383-399)
__host__ __device__ __forceinline__ uint32_t packed_get(const uint64_t* __restrict__ w, uint64_t o) {
    return (uint32_t)((w[o >> 5] >> (62 - 2 * (o & 31))) & 3ull);
}

// k-mer starting at absolute base offset o: the analogue of DnaString::get_kmer's per-word
// set_slice_mut loop (dna_string.rs:123-153): three funnel-shifted words, right-aligned.
// Host form: words beyond the last one the k-mer touches are never dereferenced.
__host__ inline K128 packed_get_kmer(const uint64_t* w, uint64_t o, int k) {
    uint64_t wi = o >> 5;
    int s = (int)(o & 31) * 2;
    int need = s + 2 * k;                       // bits consumed starting at word wi
    uint64_t w0 = w[wi];
    uint64_t w1 = need > 64 ? w[wi + 1] : 0;
    uint64_t w2 = need > 128 ? w[wi + 2] : 0;
    K128 top;
    if (s) { top.hi = (w0 << s) | (w1 >> (64 - s)); top.lo = (w1 << s) | (w2 >> (64 - s)); }
    else   { top.hi = w0; top.lo = w1; }
    return k128_shr(top, 128 - 2 * k);
}
// Device form: all three words are loaded UNCONDITIONALLY, with indices clamped to last_word (the index of the buffer's last
// word; a clamped word never reaches the result).  The form with the second and third word fetched only by the lanes that need
// them (the host form above) returned wrong words for whole wavefronts now and then -- seen first in the dense path (0.7 % of the
// two-word k-mers of a 10^8-read run), then as a 1 % failure rate of compress_graph when several processes shared the GPU;
// tools/micro/kmer_fetch.hip reproduces it in isolation (same inputs, same stream order, only the fetch differs: hundreds of
// mismatches per 2*10^4 launches with four processes on the GPU, none with unconditional loads).  The ISA of the conditional
// form waits for its loads on every path; what the hardware does with its partially masked loads under those conditions is
// not understood.  No device code fetches words conditionally any more.
__device__ __forceinline__ K128 packed_get_kmer(const uint64_t* __restrict__ w, uint64_t o, int k, uint64_t last_word) {
    const uint64_t wi = o >> 5;
    const int s = (int)(o & 31) * 2;
    const uint64_t i1 = wi + 1 < last_word ? wi + 1 : last_word, i2 = wi + 2 < last_word ? wi + 2 : last_word;
    const uint64_t w0 = w[wi], w1 = w[i1], w2 = w[i2];
    K128 top;
    if (s) { top.hi = (w0 << s) | (w1 >> (64 - s)); top.lo = (w1 << s) | (w2 >> (64 - s)); }
    else   { top.hi = w0; top.lo = w1; }
    return k128_shr(top, 128 - 2 * k);
}
// p <= 32 bases (one or two words), same rules
__device__ __forceinline__ uint64_t packed_get_pmer64(const uint64_t* __restrict__ w, uint64_t o, int p, uint64_t last_word) {
    const uint64_t wi = o >> 5;
    const int s = (int)(o & 31) * 2;
    const uint64_t i1 = wi + 1 < last_word ? wi + 1 : last_word;
    const uint64_t w0 = w[wi], w1 = w[i1];
    const uint64_t v = s ? (w0 << s) | (w1 >> (64 - s)) : w0;
    return v >> (64 - 2 * p);
}

// ---- wave helpers (wave64) ---------------------------------------------------------------
__device__ __forceinline__ uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
__device__ __forceinline__ uint64_t lanemask_lt() { uint32_t l = lane_id(); return l ? (~0ull >> (64 - l)) : 0ull; }

// payload word carried with every k-mer record: bits 0-7 Exts, bits 8-31 D1
__host__ __device__ __forceinline__ uint32_t pay_make(uint32_t exts, uint32_t data) { return (exts & 0xff) | (data << 8); }
