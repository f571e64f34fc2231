// Data formats either side of the hot path.
//
// Near side (ingestion): DnaString::from_acgt_bytes (src/dna_string.rs:222-250) with its AVX2 helpers
// convert_bases / pack_32_bases (src/bitops_avx2.rs:9-132): ASCII bases -> 2-bit packed words, 32 bases per
// u64, first base in the two most significant bits, anything outside [aAcCgGtT] -> A (base_to_bits,
// src/lib.rs:65-73).  On the GPU this is a pure streaming kernel: 32 B in, 8 B out per word, HBM bound.
//
// Far side: DnaString::to_ascii_vec / Display (bits_to_ascii, src/lib.rs:53-61) for the S lines of the GFA
// writer (graph.rs:537-611, host formatting in graph.hip).
#include "dbg_internal.hpp"

namespace {

// four ASCII bytes (little endian in w) -> 8 bits, first byte in the top two bits; *bad += number of bytes
// outside [aAcCgGtT]
__device__ __forceinline__ uint32_t pack4(uint32_t w, uint32_t* bad) {
    // (c >> 1) & 3 maps A,C,G,T (either case) to 0,1,3,2; x ^= x >> 1 fixes the order of G and T
    uint32_t x = (w >> 1) & 0x03030303u;
    x ^= (x >> 1) & 0x01010101u;
    // exact per-byte zero test of (u ^ t): ~(((z & 0x7f..) + 0x7f..) | z | 0x7f..) has 0x80 in zero bytes
    const uint32_t u = w | 0x20202020u;
    uint32_t ok = 0;
    const uint32_t targets[4] = {0x61616161u, 0x63636363u, 0x67676767u, 0x74747474u};
#pragma unroll
    for (int t = 0; t < 4; t++) {
        uint32_t z = u ^ targets[t];
        ok |= ~(((z & 0x7f7f7f7fu) + 0x7f7f7f7fu) | z | 0x7f7f7f7fu);
    }
    *bad += 4 - __popc(ok);
    x &= (ok >> 7) * 3u;                              // invalid characters become A
    return (x * 0x40100401u) >> 24;                   // b0<<6 | b1<<4 | b2<<2 | b3
}

// bases [base, base + 32) of the buffer -> one word; bytes past n are absent (zero bits, not counted as invalid)
__device__ __forceinline__ uint64_t pack_word(const uint8_t* __restrict__ ascii, uint64_t n, uint64_t base, uint32_t* bad) {
    uint64_t out = 0;
    if (base + 32 <= n && ((uintptr_t)(ascii + base) & 15) == 0) {
        const uint4* p = (const uint4*)(ascii + base);
        uint4 a = p[0], b = p[1];
        out = ((uint64_t)pack4(a.x, bad) << 56) | ((uint64_t)pack4(a.y, bad) << 48) | ((uint64_t)pack4(a.z, bad) << 40) |
              ((uint64_t)pack4(a.w, bad) << 32) | ((uint64_t)pack4(b.x, bad) << 24) | ((uint64_t)pack4(b.y, bad) << 16) |
              ((uint64_t)pack4(b.z, bad) << 8) | (uint64_t)pack4(b.w, bad);
    } else {
        const uint32_t m = (uint32_t)((n - base) < 32 ? (n - base) : 32);
        for (uint32_t g = 0; g < 8; g++) {
            uint32_t v = 0;
            for (uint32_t j = 0; j < 4; j++) {
                uint32_t i = g * 4 + j;
                if (i < m) v |= (uint32_t)ascii[base + i] << (8 * j);
                else v |= 0x41u << (8 * j);       // pad with 'A': zero bits, valid
            }
            out |= (uint64_t)pack4(v, bad) << (56 - 8 * g);
        }
    }
    return out;
}

// one thread per output word; the last word may be partial
__global__ void pack_acgt_kernel(const uint8_t* __restrict__ ascii, uint64_t n, uint64_t* __restrict__ words,
                                 unsigned long long* __restrict__ n_invalid) {
    const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t n_words = (n + 31) >> 5;
    uint32_t bad = 0;
    if (w < n_words) words[w] = pack_word(ascii, n, w << 5, &bad);
    if (n_invalid) {
        // wave-level reduction, one atomic per wave
        for (int o = 32; o > 0; o >>= 1) bad += __shfl_down(bad, o, 64);
        if ((threadIdx.x & 63) == 0 && bad) atomicAdd(n_invalid, (unsigned long long)bad);
    }
}

// SipHash-1-3 with the all-zero key, fed byte-wise: Rust's DefaultHasher::new() (std, not part of the crate)
struct Sip13 {
    uint64_t v0, v1, v2, v3, buf;
    uint32_t nbuf, total;
    __device__ void init() {
        v0 = 0x736f6d6570736575ull; v1 = 0x646f72616e646f6dull; v2 = 0x6c7967656e657261ull; v3 = 0x7465646279746573ull;
        buf = 0; nbuf = 0; total = 0;
    }
    __device__ static uint64_t rotl(uint64_t x, int b) { return (x << b) | (x >> (64 - b)); }
    __device__ void round() {
        v0 += v1; v1 = rotl(v1, 13); v1 ^= v0; v0 = rotl(v0, 32);
        v2 += v3; v3 = rotl(v3, 16); v3 ^= v2;
        v0 += v3; v3 = rotl(v3, 21); v3 ^= v0;
        v2 += v1; v1 = rotl(v1, 17); v1 ^= v2; v2 = rotl(v2, 32);
    }
    __device__ void block(uint64_t m) { v3 ^= m; round(); v0 ^= m; }
    __device__ void byte(uint8_t b) {
        buf |= (uint64_t)b << (8 * nbuf);
        total++;
        if (++nbuf == 8) { block(buf); buf = 0; nbuf = 0; }
    }
    __device__ void u64(uint64_t x) {                 // 8 bytes, little endian, at any alignment of the stream
        if (nbuf == 0) { block(x); total += 8; return; }
        const uint32_t s = 8 * nbuf;
        block(buf | (x << s));
        buf = x >> (64 - s);
        total += 8;
    }
    __device__ uint64_t finish() {
        const uint64_t b = ((uint64_t)(total & 0xff) << 56) | buf;
        block(b);
        v2 ^= 0xff;
        round(); round(); round();
        return v0 ^ v1 ^ v2 ^ v3;
    }
};

// DnaString::from_acgt_bytes_hashn (dna_string.rs:255-278) over a batch: a character outside [aAcCgGtT] at
// position pos of read r becomes DefaultHasher(name_r, pos).finish() % 4, i.e. SipHash-1-3 of
// (len(name_r) as u64 LE || name_r || pos as u64 LE).  Words with 32 valid characters (almost all) take the
// same path as pack_acgt_kernel; the others find their read by binary search in seq_off.
__global__ void pack_acgt_hashn_kernel(const uint8_t* __restrict__ ascii, uint64_t n, const uint64_t* __restrict__ seq_off,
                                       uint64_t n_seqs, const uint8_t* __restrict__ names, const uint64_t* __restrict__ name_off,
                                       uint64_t* __restrict__ words, unsigned long long* __restrict__ n_replaced) {
    const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t n_words = (n + 31) >> 5;
    uint32_t bad = 0;
    if (w < n_words) {
        const uint64_t base = w << 5;
        uint64_t out = pack_word(ascii, n, base, &bad);
        if (bad) {
            const uint32_t m = (uint32_t)((n - base) < 32 ? (n - base) : 32);
            uint64_t lo = 0, hi = n_seqs;             // last read with seq_off[r] <= base
            while (hi - lo > 1) {
                const uint64_t mid = (lo + hi) >> 1;
                if (seq_off[mid] <= base) lo = mid; else hi = mid;
            }
            uint64_t r = lo, r_end = seq_off[r + 1];
            Sip13 h0;
            bool have = false;
            for (uint32_t i = 0; i < m; i++) {
                const uint64_t g = base + i;
                while (g >= r_end) { r++; r_end = seq_off[r + 1]; have = false; }   // empty reads are skipped too
                const uint32_t c = ascii[g] | 0x20u;
                if (c == 'a' || c == 'c' || c == 'g' || c == 't') continue;
                if (!have) {                          // hasher state after read_name.hash(): cloned per position (:268)
                    const uint64_t a = name_off[r], b = name_off[r + 1];
                    h0.init();
                    h0.u64(b - a);
                    for (uint64_t j = a; j < b; j++) h0.byte(names[j]);
                    have = true;
                }
                Sip13 h = h0;
                h.u64(g - seq_off[r]);
                out |= (h.finish() & 3ull) << (62 - 2 * i);
            }
        }
        words[w] = out;
    }
    if (n_replaced) {
        for (int o = 32; o > 0; o >>= 1) bad += __shfl_down(bad, o, 64);
        if ((threadIdx.x & 63) == 0 && bad) atomicAdd(n_replaced, (unsigned long long)bad);
    }
}

// 16 bits = 8 bases (first base in the top two bits) -> 8 ASCII characters, first base in the lowest byte
__device__ __forceinline__ uint64_t expand8(uint32_t v16) {
    uint64_t x = v16;
    x = (x | (x << 24)) & 0x000000FF000000FFull;
    x = (x | (x << 12)) & 0x000F000F000F000Full;
    x = (x | (x << 6)) & 0x0303030303030303ull;
    x = __builtin_bswap64(x);
    // 0,1,2,3 -> 'A','C','G','T' = 65 + 2*b0 + 6*b1 + 11*(b0 & b1), bytewise without carries
    const uint64_t b0 = x & 0x0101010101010101ull, b1 = (x >> 1) & 0x0101010101010101ull;
    return 0x4141414141414141ull + b0 * 2 + b1 * 6 + (b0 & b1) * 11;
}

// one thread per 32 output characters: one (funnel-shifted) packed word in, 32 B out
__global__ void unpack_acgt_kernel(const uint64_t* __restrict__ words, uint64_t first_base, uint64_t n,
                                   uint8_t* __restrict__ ascii) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t o = t * 32;
    if (o >= n) return;
    const uint64_t b = first_base + o;
    const uint64_t wi = b >> 5, last = (first_base + n - 1) >> 5;
    const uint32_t s = 2 * (uint32_t)(b & 31);
    // (both words are loaded unconditionally, the second index clamped: see packed_get_kmer in dbg_device.hpp)
    const uint64_t w0 = words[wi], w1 = words[wi < last ? wi + 1 : last];
    uint64_t w = w0 << s;
    if (s && wi < last) w |= w1 >> (64 - s);
    const uint64_t c0 = expand8((uint32_t)(w >> 48)), c1 = expand8((uint32_t)(w >> 32) & 0xffffu),
                   c2 = expand8((uint32_t)(w >> 16) & 0xffffu), c3 = expand8((uint32_t)w & 0xffffu);
    if (o + 32 <= n && ((uintptr_t)(ascii + o) & 15) == 0) {
        uint4* q = (uint4*)(ascii + o);
        q[0] = make_uint4((uint32_t)c0, (uint32_t)(c0 >> 32), (uint32_t)c1, (uint32_t)(c1 >> 32));
        q[1] = make_uint4((uint32_t)c2, (uint32_t)(c2 >> 32), (uint32_t)c3, (uint32_t)(c3 >> 32));
    } else {
        const uint64_t cs[4] = {c0, c1, c2, c3};
        const uint32_t m = (uint32_t)((n - o) < 32 ? (n - o) : 32);
        for (uint32_t j = 0; j < m; j++) ascii[o + j] = (uint8_t)(cs[j >> 3] >> (8 * (j & 7)));
    }
}

}  // namespace

int unpack_acgt_stage(dbg_ctx* c, const uint64_t* words_dev, uint64_t first_base, uint64_t n, uint8_t* ascii_dev) {
    if (!n) return 0;
    const uint64_t nt = (n + 31) / 32;
    c->t_begin("unpack_acgt", n);
    unpack_acgt_kernel<<<(unsigned)cdiv(nt, 256), 256, 0, c->stream>>>(words_dev, first_base, n, ascii_dev);
    c->t_end();
    LAUNCH_CHECK(c, "unpack_acgt");
    return 0;
}

// replaces DnaString::from_acgt_bytes (dna_string.rs:222-250) over one concatenated buffer: words_dev receives
// ceil(n/32) words in DnaString / PackedDnaStringSet layout; *n_invalid (host, may be null) = number of
// characters outside [aAcCgGtT] (the `valid` flag of convert_bases, bitops_avx2.rs:52-54, as a count)
extern "C" int dbg_pack_acgt_dev(dbg_ctx* c, const uint8_t* ascii_dev, uint64_t n, uint64_t* words_dev, uint64_t* n_invalid) {
    HIP_TRY(c, hipSetDevice(c->device));
    if (n_invalid) *n_invalid = 0;
    if (!n) return 0;
    if (n >= (1ull << 37)) return c->fail(60, "pack_acgt: at most 2^37-1 bases per call");
    DBuf<unsigned long long> cnt;
    if (n_invalid) {
        ALLOC_OR_FAIL(c, cnt, 1);
        HIP_TRY(c, hipMemsetAsync(cnt.p, 0, 8, c->stream));
    }
    const uint64_t n_words = (n + 31) >> 5;
    c->t_begin("pack_acgt", n);
    pack_acgt_kernel<<<(unsigned)cdiv(n_words, 256), 256, 0, c->stream>>>(ascii_dev, n, words_dev, n_invalid ? cnt.p : nullptr);
    c->t_end();
    LAUNCH_CHECK(c, "pack_acgt");
    if (n_invalid) {
        unsigned long long h = 0;
        HIP_TRY(c, hipMemcpyAsync(&h, cnt.p, 8, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        *n_invalid = h;
    }
    return 0;
}

extern "C" int dbg_pack_acgt(dbg_ctx* c, const uint8_t* ascii, uint64_t n, uint64_t* words, uint64_t* n_invalid) {
    HIP_TRY(c, hipSetDevice(c->device));
    if (n_invalid) *n_invalid = 0;
    if (!n) return 0;
    const uint64_t n_words = (n + 31) >> 5;
    DBuf<uint8_t> d_in;
    DBuf<uint64_t> d_out;
    ALLOC_OR_FAIL(c, d_in, n);
    ALLOC_OR_FAIL(c, d_out, n_words);
    HIP_TRY(c, hipMemcpyAsync(d_in.p, ascii, n, hipMemcpyHostToDevice, c->stream));
    DBG_TRY(dbg_pack_acgt_dev(c, d_in.p, n, d_out.p, n_invalid));
    HIP_TRY(c, hipMemcpyAsync(words, d_out.p, n_words * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return 0;
}

// replaces DnaString::from_acgt_bytes_hashn (dna_string.rs:255-278) for a batch of reads in one concatenated
// buffer: read i = ascii[seq_off[i], seq_off[i+1]) with seq_off[0] = 0 and seq_off[n_seqs] = n, its name =
// names[name_off[i], name_off[i+1]); output as dbg_pack_acgt_dev.  *n_replaced (host, may be null) = number of
// characters that were replaced by a hashed base.
extern "C" int dbg_pack_acgt_hashn_dev(dbg_ctx* c, const uint8_t* ascii_dev, uint64_t n, const uint64_t* seq_off_dev,
                                       uint64_t n_seqs, const uint8_t* names_dev, const uint64_t* name_off_dev,
                                       uint64_t* words_dev, uint64_t* n_replaced) {
    HIP_TRY(c, hipSetDevice(c->device));
    if (n_replaced) *n_replaced = 0;
    if (!n) return 0;
    if (!n_seqs) return c->fail(61, "pack_acgt_hashn: bases without reads");
    if (n >= (1ull << 37)) return c->fail(60, "pack_acgt: at most 2^37-1 bases per call");
    DBuf<unsigned long long> cnt;
    if (n_replaced) {
        ALLOC_OR_FAIL(c, cnt, 1);
        HIP_TRY(c, hipMemsetAsync(cnt.p, 0, 8, c->stream));
    }
    const uint64_t n_words = (n + 31) >> 5;
    c->t_begin("pack_acgt_hashn", n);
    pack_acgt_hashn_kernel<<<(unsigned)cdiv(n_words, 256), 256, 0, c->stream>>>(ascii_dev, n, seq_off_dev, n_seqs, names_dev,
                                                                                  name_off_dev, words_dev, n_replaced ? cnt.p : nullptr);
    c->t_end();
    LAUNCH_CHECK(c, "pack_acgt_hashn");
    if (n_replaced) {
        unsigned long long h = 0;
        HIP_TRY(c, hipMemcpyAsync(&h, cnt.p, 8, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        *n_replaced = h;
    }
    return 0;
}

extern "C" int dbg_pack_acgt_hashn(dbg_ctx* c, const uint8_t* ascii, const uint64_t* seq_off, uint64_t n_seqs,
                                   const uint8_t* names, const uint64_t* name_off, uint64_t* words, uint64_t* n_replaced) {
    HIP_TRY(c, hipSetDevice(c->device));
    if (n_replaced) *n_replaced = 0;
    if (!n_seqs) return 0;
    if (seq_off[0] != 0 || name_off[0] != 0) return c->fail(62, "pack_acgt_hashn: offsets must start at 0");
    for (uint64_t i = 0; i < n_seqs; i++)
        if (seq_off[i + 1] < seq_off[i] || name_off[i + 1] < name_off[i]) return c->fail(63, "pack_acgt_hashn: offsets must not decrease");
    const uint64_t n = seq_off[n_seqs], nn = name_off[n_seqs];
    if (!n) return 0;
    const uint64_t n_words = (n + 31) >> 5;
    DBuf<uint8_t> d_in, d_names;
    DBuf<uint64_t> d_so, d_no, d_out;
    ALLOC_OR_FAIL(c, d_in, n);
    ALLOC_OR_FAIL(c, d_names, nn ? nn : 1);
    ALLOC_OR_FAIL(c, d_so, n_seqs + 1);
    ALLOC_OR_FAIL(c, d_no, n_seqs + 1);
    ALLOC_OR_FAIL(c, d_out, n_words);
    HIP_TRY(c, hipMemcpyAsync(d_in.p, ascii, n, hipMemcpyHostToDevice, c->stream));
    if (nn) HIP_TRY(c, hipMemcpyAsync(d_names.p, names, nn, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(d_so.p, seq_off, (n_seqs + 1) * 8, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(d_no.p, name_off, (n_seqs + 1) * 8, hipMemcpyHostToDevice, c->stream));
    DBG_TRY(dbg_pack_acgt_hashn_dev(c, d_in.p, n, d_so.p, n_seqs, d_names.p, d_no.p, d_out.p, n_replaced));
    HIP_TRY(c, hipMemcpyAsync(words, d_out.p, n_words * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return 0;
}

// DnaString::to_ascii_vec (dna_string.rs:297-299) of bases [first_base, first_base + n) of a packed stream
extern "C" int dbg_unpack_acgt_dev(dbg_ctx* c, const uint64_t* words_dev, uint64_t first_base, uint64_t n, uint8_t* ascii_dev) {
    HIP_TRY(c, hipSetDevice(c->device));
    return unpack_acgt_stage(c, words_dev, first_base, n, ascii_dev);
}

extern "C" int dbg_unpack_acgt(dbg_ctx* c, const uint64_t* words, uint64_t first_base, uint64_t n, uint8_t* ascii) {
    HIP_TRY(c, hipSetDevice(c->device));
    if (!n) return 0;
    const uint64_t w0 = first_base >> 5, w1 = (first_base + n + 31) >> 5;
    DBuf<uint64_t> d_in;
    DBuf<uint8_t> d_out;
    ALLOC_OR_FAIL(c, d_in, w1 - w0);
    ALLOC_OR_FAIL(c, d_out, n);
    HIP_TRY(c, hipMemcpyAsync(d_in.p, words + w0, (w1 - w0) * 8, hipMemcpyHostToDevice, c->stream));
    DBG_TRY(unpack_acgt_stage(c, d_in.p, first_base & 31, n, d_out.p));
    HIP_TRY(c, hipMemcpyAsync(ascii, d_out.p, n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return 0;
}
