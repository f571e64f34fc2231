// =============================================================================
// dbg_oracle.cpp  --  TEST INFRASTRUCTURE ONLY (see dbg_oracle.hpp header).
// CPU restatement of the reference hot path; each function cites the
// reference file:line (relative to /root/reference/src) it follows.
// =============================================================================
#include "dbg_oracle.hpp"
#include <set>
#include <algorithm>
#include <cstring>
#include <cassert>

namespace orc {

// ---------------------------------------------------------------------------
// kmer.rs:104-228 IntHelp::reverse_by_twos for u128 / u64
// ---------------------------------------------------------------------------
static inline u128 mk128(uint64_t hi, uint64_t lo) { return ((u128)hi << 64) | lo; }

u128 reverse_by_twos(u128 x, int tbits) {
    if (tbits == 64) {                                  // kmer.rs:141-160
        uint64_t s = (uint64_t)x;
        uint64_t r = ((s & 0x3333333333333333ull) << 2) | ((s >> 2) & 0x3333333333333333ull);
        r = ((r & 0x0F0F0F0F0F0F0F0Full) << 4) | ((r >> 4) & 0x0F0F0F0F0F0F0F0Full);
        r = ((r & 0x00FF00FF00FF00FFull) << 8) | ((r >> 8) & 0x00FF00FF00FF00FFull);
        r = ((r & 0x0000FFFF0000FFFFull) << 16) | ((r >> 16) & 0x0000FFFF0000FFFFull);
        r = ((r & 0x00000000FFFFFFFFull) << 32) | ((r >> 32) & 0x00000000FFFFFFFFull);
        return (u128)r;
    }
    // kmer.rs:104-132
    const u128 m2 = mk128(0x3333333333333333ull, 0x3333333333333333ull);
    const u128 m4 = mk128(0x0F0F0F0F0F0F0F0Full, 0x0F0F0F0F0F0F0F0Full);
    const u128 m8 = mk128(0x00FF00FF00FF00FFull, 0x00FF00FF00FF00FFull);
    const u128 m16 = mk128(0x0000FFFF0000FFFFull, 0x0000FFFF0000FFFFull);
    const u128 m32 = mk128(0x00000000FFFFFFFFull, 0x00000000FFFFFFFFull);
    const u128 m64 = mk128(0, 0xFFFFFFFFFFFFFFFFull);
    u128 r = ((x & m2) << 2) | ((x >> 2) & m2);
    r = ((r & m4) << 4) | ((r >> 4) & m4);
    r = ((r & m8) << 8) | ((r >> 8) & m8);
    r = ((r & m16) << 16) | ((r >> 16) & m16);
    r = ((r & m32) << 32) | ((r >> 32) & m32);
    r = ((r & m64) << 64) | ((r >> 64) & m64);
    return r;
}

static inline u128 tmask(int tbits) { return tbits == 128 ? ~(u128)0 : (((u128)1 << tbits) - 1); }

// kmer.rs:538-549
u128 top_mask(const KmerSpec& s, int n_bases) {
    int unused_bits = s.tbits - 2 * s.k;
    int mask_bits = n_bases * 2 + unused_bits;
    if (mask_bits > 0) {
        u128 ones = mask_bits >= 128 ? ~(u128)0 : (((u128)1 << mask_bits) - 1);
        return (ones << (s.tbits - mask_bits)) & tmask(s.tbits);
    }
    return 0;
}
// kmer.rs:552-560
u128 bottom_mask(int n_bases) {
    if (n_bases > 0) return n_bases * 2 >= 128 ? ~(u128)0 : (((u128)1 << (n_bases * 2)) - 1);
    return 0;
}

// kmer.rs:574-577
uint8_t kmer_get(const KmerSpec& s, Kmer km, int pos) {
    int bit = (s.k - 1 - pos) * 2;           // addr(), kmer.rs:514-517
    return (uint8_t)((km.storage >> bit) & 3u);
}
// kmer.rs:579-584
void kmer_set_mut(const KmerSpec& s, Kmer& km, int pos, uint8_t v) {
    int bit = (s.k - 1 - pos) * 2;
    u128 mask = ~((u128)3 << bit);
    km.storage = (km.storage & mask) | ((u128)v << bit);
}
// kmer.rs:590-617
void kmer_set_slice_mut(const KmerSpec& s, Kmer& km, int pos, int n_bases, uint64_t value) {
    assert(pos + n_bases <= s.k);
    // "Move bases up to the top of this type": T is u64 or u128 here
    u128 v = s.tbits > 64 ? ((u128)value << (s.tbits - 64)) : (u128)value;
    u128 tm = top_mask(s, pos);
    u128 bm = bottom_mask(s.k - (pos + n_bases));
    u128 mask = tm | bm;
    int shift = 2 * pos + (s.tbits - 2 * s.k);
    u128 value_slide = shift >= 128 ? 0 : (v >> shift);
    km.storage = (km.storage & mask) | (value_slide & ~mask & tmask(s.tbits));
}
// kmer.rs:469-477
Kmer kmer_extend_left(const KmerSpec& s, Kmer km, uint8_t v) {
    Kmer n(km.storage >> 2);
    kmer_set_mut(s, n, 0, v);
    return n;
}
// kmer.rs:479-487
Kmer kmer_extend_right(const KmerSpec& s, Kmer km, uint8_t v) {
    Kmer n(((km.storage << 2) & tmask(s.tbits)) & ~top_mask(s, 0));
    kmer_set_mut(s, n, s.k - 1, v);
    return n;
}
// lib.rs:210-215
Kmer kmer_extend(const KmerSpec& s, Kmer km, uint8_t v, Dir d) {
    return d == Left ? kmer_extend_left(s, km, v) : kmer_extend_right(s, km, v);
}
// kmer.rs:620-634
Kmer kmer_rc(const KmerSpec& s, Kmer km) {
    u128 n = (~reverse_by_twos(km.storage, s.tbits)) & tmask(s.tbits);
    if (s.k < s.tbits / 2) {
        int up_shift = 2 * (s.tbits / 2 - s.k);
        n = n >> up_shift;
    }
    return Kmer(n);
}
// lib.rs:224-231 -- NOTE: equality returns (rc, true)
void kmer_min_rc_flip(const KmerSpec& s, Kmer km, Kmer& out, bool& flip) {
    Kmer rc = kmer_rc(s, km);
    if (km < rc) { out = km; flip = false; }
    else { out = rc; flip = true; }
}
// lib.rs:234-241
Kmer kmer_min_rc(const KmerSpec& s, Kmer km) {
    Kmer rc = kmer_rc(s, km);
    return km < rc ? km : rc;
}
// lib.rs:244-246
bool kmer_is_palindrome(const KmerSpec& s, Kmer km) {
    return s.k % 2 == 0 && km == kmer_rc(s, km);
}
// kmer.rs:457-459
uint64_t kmer_to_u64(Kmer km) {
    assert((km.storage >> 64) == 0);
    return (uint64_t)km.storage;
}
// lib.rs:249-261
Kmer kmer_from_bytes(const KmerSpec& s, const uint8_t* b) {
    Kmer k0;
    for (int i = 0; i < s.k; i++) kmer_set_mut(s, k0, i, b[i]);
    return k0;
}
// lib.rs:279-285
std::string kmer_to_string(const KmerSpec& s, Kmer km) {
    std::string out;
    for (int i = 0; i < s.k; i++) out.push_back(bits_to_base(kmer_get(s, km, i)));
    return out;
}

// ---------------------------------------------------------------------------
// DnaString (dna_string.rs)
// ---------------------------------------------------------------------------
uint8_t DnaString::get(size_t i) const {              // :90-93, :383-386, :395-399
    size_t kbit = i * 2;
    size_t block = kbit / 64, bit = kbit % 64;
    return (uint8_t)((storage[block] >> (62 - bit)) & 3u);
}
void DnaString::push(uint8_t value) {                 // :303-310 (+ set_by_addr :388-393)
    size_t kbit = len * 2;
    size_t block = kbit / 64, bit = kbit % 64;
    if (bit == 0 && block >= storage.size()) storage.push_back(0);
    uint64_t mask = 3ull << (62 - bit);
    storage[block] |= mask;
    storage[block] ^= mask;
    storage[block] |= ((uint64_t)value & 3ull) << (62 - bit);
    len += 1;
}
void DnaString::extend(const uint8_t* b, size_t n) {  // :312-343
    size_t i = 0;
    while (len % 32 != 0) {
        if (i < n) push(b[i++]); else return;
    }
    while (i < n) {
        uint64_t val = 0;
        int offset = 62;
        size_t n_added = 0;
        for (int j = 0; j < 32; j++) {
            if (i < n) {
                assert(b[i] < 4);
                val |= (uint64_t)b[i] << offset;
                offset -= 2;
                n_added++;
                i++;
            } else break;
        }
        storage.push_back(val);
        len += n_added;
    }
}
DnaString DnaString::from_bytes(const uint8_t* b, size_t n) {
    DnaString d; d.extend(b, n); return d;
}
DnaString DnaString::from_acgt_bytes(const uint8_t* b, size_t n) {     // dna_string.rs:222-250
    // the AVX2 branch (:229-244; bitops_avx2.rs:9-132) produces the same words as the scalar one (:247-249):
    // convert_bases maps [aAcCgGtT] like base_to_bits and everything else to 0, pack_32_bases puts the first
    // byte in the top two bits.  Restated here in the scalar form.
    DnaString d;
    for (size_t i = 0; i < n; i++) d.push(base_to_bits(b[i]));
    return d;
}
uint64_t siphash(int c_rounds, int d_rounds, uint64_t k0, uint64_t k1, const uint8_t* msg, size_t len) {
    uint64_t v0 = k0 ^ 0x736f6d6570736575ull, v1 = k1 ^ 0x646f72616e646f6dull;
    uint64_t v2 = k0 ^ 0x6c7967656e657261ull, v3 = k1 ^ 0x7465646279746573ull;
    auto rotl = [](uint64_t x, int b) { return (x << b) | (x >> (64 - b)); };
    auto round = [&]() {
        v0 += v1; v1 = rotl(v1, 13); v1 ^= v0; v0 = rotl(v0, 32);
        v2 += v3; v3 = rotl(v3, 16); v3 ^= v2;
        v0 += v3; v3 = rotl(v3, 21); v3 ^= v0;
        v2 += v1; v1 = rotl(v1, 17); v1 ^= v2; v2 = rotl(v2, 32);
    };
    size_t full = len & ~(size_t)7;
    for (size_t i = 0; i < full; i += 8) {
        uint64_t m = 0;
        for (int j = 0; j < 8; j++) m |= (uint64_t)msg[i + j] << (8 * j);
        v3 ^= m;
        for (int r = 0; r < c_rounds; r++) round();
        v0 ^= m;
    }
    uint64_t b = (uint64_t)(len & 0xff) << 56;
    for (size_t j = full; j < len; j++) b |= (uint64_t)msg[j] << (8 * (j - full));
    v3 ^= b;
    for (int r = 0; r < c_rounds; r++) round();
    v0 ^= b;
    v2 ^= 0xff;
    for (int r = 0; r < d_rounds; r++) round();
    return v0 ^ v1 ^ v2 ^ v3;
}
DnaString DnaString::from_acgt_bytes_hashn(const uint8_t* b, size_t n, const uint8_t* read_name, size_t name_len) {
    // dna_string.rs:255-278.  `read_name.hash(&mut hasher)` on a &[u8] feeds the length as a usize (8 bytes, little
    // endian) and then the bytes (impl Hash for [T], core/src/hash/mod.rs); `pos.hash(..)` feeds pos as a usize;
    // the streaming hasher equals SipHash-1-3 of the concatenation.
    std::vector<uint8_t> msg(8 + name_len + 8);
    for (int j = 0; j < 8; j++) msg[j] = (uint8_t)((uint64_t)name_len >> (8 * j));
    for (size_t j = 0; j < name_len; j++) msg[8 + j] = read_name[j];
    DnaString d;
    for (size_t pos = 0; pos < n; pos++) {
        uint8_t v;
        switch (b[pos]) {                                           // :262-266
            case 'A': case 'a': v = 0; break;
            case 'C': case 'c': v = 1; break;
            case 'G': case 'g': v = 2; break;
            case 'T': case 't': v = 3; break;
            default: {                                              // :267-271
                for (int j = 0; j < 8; j++) msg[8 + name_len + j] = (uint8_t)((uint64_t)pos >> (8 * j));
                v = (uint8_t)(siphash(1, 3, 0, 0, msg.data(), msg.size()) % 4);
            }
        }
        d.push(v);
    }
    return d;
}
std::vector<uint8_t> DnaString::to_ascii_vec() const {                  // dna_string.rs:297-299, bits_to_ascii lib.rs:53-61
    std::vector<uint8_t> v(len);
    for (size_t i = 0; i < len; i++) v[i] = (uint8_t)"ACGT"[get(i)];
    return v;
}
DnaString DnaString::from_dna_string(const char* s) { // :187-195
    DnaString d;
    std::vector<uint8_t> b;
    for (const char* c = s; *c; ++c) b.push_back(base_to_bits((uint8_t)*c));
    d.extend(b.data(), b.size());
    return d;
}
std::string DnaString::to_string() const {
    std::string out;
    for (size_t i = 0; i < len; i++) out.push_back(bits_to_base(get(i)));
    return out;
}

// dna_string.rs:123-153 -- get_kmer by whole-word set_slice_mut
Kmer seq_get_kmer(const KmerSpec& s, const SeqView& v, size_t pos) {
    assert(v.length - pos >= (size_t)s.k);
    size_t gp = v.start + pos;          // DnaStringSlice::get_kmer :575-580 forwards start+pos
    size_t block = gp / 32;
    int kmer_pos = 0;
    int block_pos = (int)(gp % 32);
    Kmer kmer;
    while (kmer_pos < s.k) {
        int nb = std::min(s.k - kmer_pos, 32 - block_pos);
        uint64_t w = v.words[block];
        uint64_t val = w << (2 * block_pos);
        kmer_set_slice_mut(s, kmer, kmer_pos, nb, val);
        block += 1;
        kmer_pos += nb;
        block_pos = 0;
    }
    return kmer;
}

// ---------------------------------------------------------------------------
// filter_kmers (filter.rs:139-231)
// ---------------------------------------------------------------------------
// filter.rs:18-23
static inline size_t filter_bucket(const KmerSpec& s, Kmer km) {
    return ((size_t)kmer_get(s, km, 0) << 6) | ((size_t)kmer_get(s, km, 1) << 4) |
           ((size_t)kmer_get(s, km, 2) << 2) | (size_t)kmer_get(s, km, 3);
}

struct Obs { Kmer kmer; Exts exts; uint32_t d; };

// size_of::<(K, D1)>() as rustc (>=1.77: u128 align 16) lays it out; only the pass count depends on it
static size_t sizeof_tuple_k_d1(const KmerSpec& s, size_t sizeof_d1) {
    size_t ksz = s.tbits / 8, al = ksz;          // u64 -> 8/8, u128 -> 16/16
    size_t raw = ksz + sizeof_d1;
    return (raw + al - 1) / al * al;
}

int filter_kmers(const SeqSet& seqs, int k, Summarizer summarizer, size_t min_kmer_obs,
                 bool stranded, bool report_all_kmers, size_t memory_size,
                 KmerTable& out, std::string& err) {
    if (k < 4 || k > 64) { err = "k out of range (bucket() needs k >= 4)"; return 1; }
    if (memory_size == 0) { err = "attempt to divide by zero (memory_size = 0, filter.rs:158)"; return 2; }
    KmerSpec ks(k);
    bool rc_norm = !stranded;                                       // :149

    size_t input_kmers = 0;                                         // :152-155
    for (size_t i = 0; i < seqs.n_seqs; i++) {
        size_t len = seqs.length[i];
        input_kmers += len >= (size_t)(k - 1) ? len - (k - 1) : 0;  // saturating_sub
    }
    size_t kmer_mem = input_kmers * sizeof_tuple_k_d1(ks, seqs.sizeof_d1);   // :156
    size_t max_mem = memory_size * 1000000000ull;                   // :157
    size_t slices = kmer_mem / max_mem + 1;                         // :158
    size_t sz = 256 / slices + 1;                                   // :159

    std::vector<std::pair<size_t, size_t>> bucket_ranges;           // :161-166
    size_t start = 0;
    while (start < 256) { bucket_ranges.push_back({start, start + sz}); start += sz; }
    if (!(bucket_ranges.back().second >= 256)) { err = "assert bucket_ranges (filter.rs:167)"; return 3; }
    out = KmerTable();
    out.n_passes = (uint32_t)bucket_ranges.size();
    if (summarizer == COUNT_FILTER_SET) out.set_off.push_back(0);

    for (auto& br : bucket_ranges) {                                // :183
        std::vector<std::vector<Obs>> kmer_buckets(256);            // :186
        for (size_t si = 0; si < seqs.n_seqs; si++) {               // :188
            SeqView seq{seqs.words, (size_t)seqs.start[si], (size_t)seqs.length[si]};
            Exts seq_exts(seqs.exts ? seqs.exts[si] : 0);
            uint32_t d = seqs.data ? seqs.data[si] : 0;
            // Vmer::iter_kmer_exts (lib.rs:408-422) + KmerExtsIter::next (lib.rs:812-841)
            size_t len = seq.length;
            Kmer kmer = len >= (size_t)k ? seq_get_kmer(ks, seq, 0) : Kmer();
            size_t pos = (size_t)k;
            while (pos <= len) {                                    // lib.rs:813
                uint8_t next_base = pos < len ? seq.get(pos) : 0;   // :814-818
                Exts cur_left = pos == (size_t)k ? seq_exts : Exts::mk_left(seq.get(pos - k - 1));   // :820-824
                Exts cur_right = pos < len ? Exts::mk_right(next_base) : seq_exts;                  // :826-830
                Exts cur_exts = Exts::merge(cur_left, cur_right);   // :832
                Kmer retval = kmer;
                kmer = kmer_extend_right(ks, kmer, next_base);      // :835
                pos += 1;
                // filter.rs:190-201
                Kmer min_kmer = retval; Exts flip_exts = cur_exts;
                if (rc_norm) {
                    bool flip; kmer_min_rc_flip(ks, retval, min_kmer, flip);
                    flip_exts = flip ? cur_exts.rc() : cur_exts;
                }
                size_t b = filter_bucket(ks, min_kmer);
                if (b >= br.first && b < br.second) kmer_buckets[b].push_back(Obs{min_kmer, flip_exts, d});
            }
        }
        for (auto& kmer_vec : kmer_buckets) {                       // :205
            std::stable_sort(kmer_vec.begin(), kmer_vec.end(),      // :206 sort_by_key is stable
                             [](const Obs& a, const Obs& b) { return a.kmer < b.kmer; });
            size_t i = 0;
            while (i < kmer_vec.size()) {                           // :208 group_by
                size_t j = i;
                while (j < kmer_vec.size() && kmer_vec[j].kmer == kmer_vec[i].kmer) j++;
                Kmer kmer = kmer_vec[i].kmer;
                bool is_valid; Exts all_exts = Exts::empty();
                if (summarizer == COUNT_FILTER) {                   // filter.rs:53-62
                    uint16_t count = 0;
                    for (size_t t = i; t < j; t++) {
                        count = count == 0xFFFF ? 0xFFFF : (uint16_t)(count + 1);   // saturating_add
                        all_exts = all_exts.add(kmer_vec[t].exts);
                    }
                    is_valid = (size_t)count >= min_kmer_obs;
                    if (report_all_kmers) out.all_kmers.push_back(kmer);            // :210-212
                    if (is_valid) { out.keys.push_back(kmer); out.exts.push_back(all_exts); out.count.push_back(count); }
                } else {                                            // filter.rs:85-100
                    std::vector<uint32_t> out_data;
                    size_t nobs = 0;
                    for (size_t t = i; t < j; t++) {
                        out_data.push_back(kmer_vec[t].d);
                        all_exts = all_exts.add(kmer_vec[t].exts);
                        nobs++;
                    }
                    std::sort(out_data.begin(), out_data.end());
                    out_data.erase(std::unique(out_data.begin(), out_data.end()), out_data.end());
                    is_valid = nobs >= min_kmer_obs;
                    if (report_all_kmers) out.all_kmers.push_back(kmer);
                    if (is_valid) {
                        out.keys.push_back(kmer); out.exts.push_back(all_exts);
                        out.set_val.insert(out.set_val.end(), out_data.begin(), out_data.end());
                        out.set_off.push_back(out.set_val.size());
                    }
                }
                i = j;
            }
        }
    }
    return 0;
}

static bool sorted_contains(const std::vector<Kmer>& v, Kmer k) {
    return std::binary_search(v.begin(), v.end(), k);
}
// filter.rs:238-276
void remove_censored_exts_sharded(const KmerSpec& s, bool stranded, std::vector<Kmer>& keys,
                                  std::vector<Exts>& exts, const std::vector<Kmer>& all_kmers) {
    for (size_t idx = 0; idx < keys.size(); idx++) {
        Exts new_exts = Exts::empty();
        Kmer kmer = keys[idx];
        Exts e = exts[idx];
        for (int d = 0; d < 2; d++) {
            Dir dir = (Dir)d;
            for (uint8_t i = 0; i < 4; i++) {
                if (e.has_ext(dir, i)) {
                    Kmer ek = kmer_extend(s, kmer, i, dir);
                    if (!stranded) ek = kmer_min_rc(s, ek);
                    bool censored = sorted_contains(keys, ek) ? false : sorted_contains(all_kmers, ek);
                    if (!censored) new_exts = new_exts.set(dir, i);
                }
            }
        }
        exts[idx] = new_exts;
    }
}
// filter.rs:280-306
void remove_censored_exts(const KmerSpec& s, bool stranded, std::vector<Kmer>& keys, std::vector<Exts>& exts) {
    for (size_t idx = 0; idx < keys.size(); idx++) {
        Exts new_exts = Exts::empty();
        Kmer kmer = keys[idx];
        Exts e = exts[idx];
        for (int d = 0; d < 2; d++) {
            Dir dir = (Dir)d;
            for (uint8_t i = 0; i < 4; i++) {
                if (e.has_ext(dir, i)) {
                    Kmer ek = kmer_extend(s, kmer, i, dir);
                    if (!stranded) ek = kmer_min_rc(s, ek);
                    if (sorted_contains(keys, ek)) new_exts = new_exts.set(dir, i);
                }
            }
        }
        exts[idx] = new_exts;
    }
}

// ---------------------------------------------------------------------------
// MSP (msp.rs)
// ---------------------------------------------------------------------------
struct MinPos { size_t val; size_t pos; Kmer kmer; };   // msp.rs:120-125
// msp.rs:127-141: smaller val is Less; on equal val, LARGER pos is Less
static inline bool minpos_less(const MinPos& a, const MinPos& b) {
    if (a.val != b.val) return a.val < b.val;
    return a.pos > b.pos;
}
// std::cmp::min(a, b): returns a unless b < a
static inline MinPos minpos_min(const MinPos& a, const MinPos& b) { return minpos_less(b, a) ? b : a; }

static inline uint32_t at_count(const KmerSpec& ps, Kmer km) {      // lib.rs:140-149
    uint32_t c = 0;
    for (int i = 0; i < ps.k; i++) { uint8_t b = kmer_get(ps, km, i); if (b == 0 || b == 3) c++; }
    return c;
}

int msp_scan(const uint8_t* seq, size_t m, int k, int p, const uint64_t* perm, bool rc,
             ScoreMode mode, std::vector<MspInterval>& out, std::string& err) {
    out.clear();
    if (!(m >= (size_t)k)) { err = "assert seq.len() >= k (msp.rs:209)"; return 1; }
    if (!(m < (1ull << 32))) { err = "assert seq.len() < 1<<32 (msp.rs:210)"; return 2; }
    if (p > 32 || p < 1 || p > k) { err = "p out of range"; return 3; }
    KmerSpec ps(p);
    auto score = [&](Kmer pi) -> size_t {                           // msp.rs:305-311 (72-81)
        if (mode == SCORE_AT_COUNT) return at_count(ps, pi);
        uint64_t a = kmer_to_u64(pi);
        size_t sa = perm ? (size_t)perm[a] : (size_t)a;
        if (rc) {
            uint64_t b = kmer_to_u64(kmer_rc(ps, pi));
            size_t sb = perm ? (size_t)perm[b] : (size_t)b;
            return std::min(sa, sb);
        }
        return sa;
    };
    auto mp = [&](size_t pos) -> MinPos {                           // msp.rs:194-198
        Kmer kmer = kmer_from_bytes(ps, seq + pos);                 // DnaSlice::get_kmer lib.rs:530-532
        return MinPos{score(kmer), pos, kmer};
    };
    auto incr = [&](const MinPos& mpv) -> MinPos {                  // msp.rs:200-205
        size_t pos = mpv.pos + 1;
        Kmer kmer = kmer_extend_right(ps, mpv.kmer, seq[pos + p - 1]);
        return MinPos{score(kmer), pos, kmer};
    };
    auto find_min = [&](size_t start, size_t stop) -> MinPos {      // msp.rs:218-228
        MinPos min_pos = mp(start);
        MinPos current = min_pos;
        while (current.pos < stop) {
            current = incr(current);
            min_pos = minpos_min(min_pos, current);
        }
        return min_pos;
    };
    std::vector<std::pair<size_t, MinPos>> min_positions;
    MinPos min_pos = find_min(0, k - p);                            // :232
    MinPos end_pos = mp(k - p);                                     // :233
    min_positions.push_back({0, min_pos});                          // :235
    for (size_t i = 1; i < m - k + 1; i++) {                        // :237
        end_pos = incr(end_pos);                                    // :239
        if (i > min_pos.pos) {                                      // :241
            min_pos = find_min(i, i + k - p);
            min_positions.push_back({i, min_pos});
        } else if (end_pos.val < min_pos.val) {                     // :244
            min_pos = end_pos;
            min_positions.push_back({i, min_pos});
        }
    }
    for (size_t q = 0; q + 1 < min_positions.size(); q++) {         // :253-264
        size_t start_pos = min_positions[q].first;
        const MinPos& mpq = min_positions[q].second;
        size_t next_pos = min_positions[q + 1].first;
        out.push_back(MspInterval{kmer_to_u64(mpq.kmer), (uint32_t)start_pos,
                                  (uint16_t)(next_pos + k - 1 - start_pos), (uint32_t)mpq.pos});
    }
    {                                                               // :266-273
        size_t last_pos = min_positions.back().first;
        const MinPos& mpq = min_positions.back().second;
        out.push_back(MspInterval{kmer_to_u64(mpq.kmer), (uint32_t)last_pos,
                                  (uint16_t)(m - last_pos), (uint32_t)mpq.pos});
    }
    return 0;
}

int msp_sequence(const uint8_t* seq, size_t m, int k, int p, const uint64_t* perm, bool rc,
                 size_t vmer_max_len, std::vector<MspPiece>& out, std::string& err) {
    out.clear();
    if (!(vmer_max_len >= (size_t)(2 * k - p))) { err = "assert V::max_len() >= 2k-p (msp.rs:292)"; return 4; }
    if (m < (size_t)k) return 0;                                    // msp.rs:294-296
    std::vector<MspInterval> iv;
    int r = msp_scan(seq, m, k, p, perm, rc, SCORE_PERM, iv, err);
    if (r) return r;
    KmerSpec ps(p);
    for (auto& msp : iv) {                                          // :315-323
        Exts e = Exts::from_slice_bounds(seq, m, msp.start, msp.len);
        uint64_t bucket = kmer_to_u64(kmer_min_rc(ps, Kmer(msp.minimizer)));   // :115-117
        out.push_back(MspPiece{(uint32_t)bucket, e, msp.start, msp.len});
    }
    return 0;
}

// vmer.rs:125-134 (new: len in low 8 bits of last word) + lib.rs:356-363 (from_slice) + vmer.rs:58-64 (set_mut)
void lmer_from_slice(const uint8_t* seq, size_t len, int n_words, uint64_t* w) {
    for (int i = 0; i < n_words; i++) w[i] = 0;
    w[n_words - 1] = (uint64_t)len & 0xff;
    for (size_t i = 0; i < len; i++) {
        size_t block = i / 32, off = i % 32;
        int offset = (int)(31 - off) * 2;                           // vmer.rs:16-21 block_set
        uint64_t mask = ~(3ull << offset);
        w[block] = (w[block] & mask) | ((uint64_t)seq[i] << offset);
    }
}

// ---------------------------------------------------------------------------
// CompressionSpec instances (compression.rs:34-98)
// ---------------------------------------------------------------------------
uint32_t spec_reduce(Spec spec, uint32_t a, uint32_t b, bool& panic) {
    switch (spec) {
        case SPEC_SIMPLE_SAT_ADD_U16: { uint32_t s = a + b; return s > 65535 ? 65535 : s; }   // test.rs:383
        case SPEC_SIMPLE_ADD_MOD_U16: return (a + b) % 65535;                                  // test.rs:247
        case SPEC_SIMPLE_MAX_U16: return std::max(a, b);                                       // test.rs:469
        case SPEC_SCMAP_EQ: if (a != b) panic = true; return a;                                // compression.rs:88-93
        case SPEC_SIMPLE_WRAP_ADD_U16: return (a + b) & 0xFFFF;                                // test.rs:265 (release build)
    }
    return a;
}
bool spec_join_test(Spec spec, uint32_t a, uint32_t b) {
    return spec == SPEC_SCMAP_EQ ? a == b : true;                   // compression.rs:62-64, :95-97
}

// ---------------------------------------------------------------------------
// CompressFromHash (compression.rs:355-584)
// ---------------------------------------------------------------------------
namespace {
struct ExtMode { bool unique; Kmer kmer; Dir dir; Exts exts; };     // compression.rs:17-21

struct CompressFromHash {
    KmerSpec ks;
    bool stranded;
    Spec spec;
    const std::vector<Kmer>& keys;
    const std::vector<Exts>& exts;
    const std::vector<uint32_t>& data;
    std::vector<uint8_t> available;
    // key -> id lookup standing in for BoomHashMap2::get_key_id: sorted (key,id) pairs
    std::vector<std::pair<Kmer, uint32_t>> idx;
    std::string* err;
    bool failed = false;

    CompressFromHash(int k, bool st, Spec sp, const std::vector<Kmer>& ke, const std::vector<Exts>& ex,
                     const std::vector<uint32_t>& da, std::string* e)
        : ks(k), stranded(st), spec(sp), keys(ke), exts(ex), data(da), err(e) {
        available.assign(keys.size(), 1);                           // :550-554
        idx.reserve(keys.size());
        for (size_t i = 0; i < keys.size(); i++) idx.push_back({keys[i], (uint32_t)i});
        std::sort(idx.begin(), idx.end(), [](auto& a, auto& b) { return a.first < b.first; });
    }
    long get_kmer_id(Kmer k) const {                                // :373-375
        auto it = std::lower_bound(idx.begin(), idx.end(), k, [](auto& a, Kmer b) { return a.first < b; });
        if (it != idx.end() && it->first == k) return it->second;
        return -1;
    }
    void fail(const std::string& m) { if (!failed) { failed = true; *err = m; } }

    ExtMode try_extend_kmer(Kmer kmer, Dir dir) {                   // :382-444
        long kid = get_kmer_id(kmer);
        if (kid < 0) { fail("couldn't find kmer (compression.rs:369)"); return ExtMode{false, Kmer(), dir, Exts()}; }
        Exts e = exts[kid]; uint32_t kmer_data = data[kid];         // :384
        if (e.num_ext_dir(dir) != 1 || (!stranded && kmer_is_palindrome(ks, kmer))) {   // :386
            return ExtMode{false, Kmer(), dir, e.single_dir(dir)};
        }
        int ext_base = e.get_unique_extension(dir);                 // :390
        Kmer next_kmer = kmer_extend(ks, kmer, (uint8_t)ext_base, dir);   // :392
        bool do_flip = false;
        if (!stranded) {                                            // :396-400
            Kmer mn; kmer_min_rc_flip(ks, next_kmer, mn, do_flip);
            next_kmer = mn;
        }
        Dir next_dir = dir_cond_flip(dir, do_flip);                 // :402
        bool is_palindrome = !stranded && kmer_is_palindrome(ks, next_kmer);   // :403
        long nid = get_kmer_id(next_kmer);                          // :410-415
        if (!(nid >= 0 && available[nid])) return ExtMode{false, Kmer(), dir, e.single_dir(dir)};
        Dir new_incoming_dir = dir_cond_flip(dir_flip(dir), do_flip);          // :419
        Exts next_exts = exts[nid]; uint32_t next_data = data[nid];
        uint8_t incoming_count = next_exts.num_ext_dir(new_incoming_dir);      // :422
        Exts outgoing_exts = next_exts.single_dir(dir_flip(new_incoming_dir)); // :423
        bool can_join = spec_join_test(spec, kmer_data, next_data);            // :426
        if (incoming_count == 0 && !is_palindrome) {                // :428-434
            fail("unreachable (compression.rs:434)");
            return ExtMode{false, Kmer(), dir, e.single_dir(dir)};
        } else if (can_join && incoming_count == 1 && !is_palindrome) {        // :435-437
            return ExtMode{true, next_kmer, next_dir, outgoing_exts};
        }
        return ExtMode{false, Kmer(), dir, e.single_dir(dir)};      // :438-442
    }

    Exts extend_kmer(Kmer kmer, Dir start_dir, std::vector<std::pair<Kmer, Dir>>& path) {   // :450-479
        Dir current_dir = start_dir;
        Kmer current_kmer = kmer;
        path.clear();
        long id = get_kmer_id(kmer);
        if (id < 0) { fail("should have this kmer (compression.rs:457)"); return Exts(); }
        available[id] = 0;                                          // :458
        while (!failed) {
            ExtMode r = try_extend_kmer(current_kmer, current_dir);
            if (r.unique) {
                path.push_back({r.kmer, r.dir});
                long next_id = get_kmer_id(r.kmer);
                available[next_id] = 0;                             // :467
                current_kmer = r.kmer;
                current_dir = r.dir;
            } else {
                return r.exts;
            }
        }
        return Exts();
    }

    void build_node(size_t seed_id, std::vector<std::pair<Kmer, Dir>>& path, std::deque<uint8_t>& edge_seq,
                    Exts& node_exts, uint32_t& node_data_out) {     // :483-541
        Kmer seed = keys[seed_id];                                  // :489
        edge_seq.clear();
        for (int i = 0; i < ks.k; i++) edge_seq.push_back(kmer_get(ks, seed, i));   // :491-493
        uint32_t node_data = data[seed_id];                         // :495
        bool panic = false;
        Exts l_ext = extend_kmer(seed, Left, path);                 // :497
        for (auto& pr : path) {                                     // :500-511
            Kmer kmer = pr.second == Left ? pr.first : kmer_rc(ks, pr.first);
            edge_seq.push_front(kmer_get(ks, kmer, 0));
            node_data = spec_reduce(spec, node_data, data[get_kmer_id(pr.first)], panic);
        }
        Exts left_extend = l_ext;                                   // :513-517
        if (!path.empty() && path.back().second == Right) left_extend = l_ext.complement();
        Exts r_ext = extend_kmer(seed, Right, path);                // :519
        for (auto& pr : path) {                                     // :522-532
            Kmer kmer = pr.second == Left ? kmer_rc(ks, pr.first) : pr.first;
            edge_seq.push_back(kmer_get(ks, kmer, ks.k - 1));
            node_data = spec_reduce(spec, node_data, data[get_kmer_id(pr.first)], panic);
        }
        Exts right_extend = r_ext;                                  // :534-538
        if (!path.empty() && path.back().second == Left) right_extend = r_ext.complement();
        if (panic) fail("Should not happen: ScmapCompress reduce on unequal data (compression.rs:90)");
        node_exts = Exts::from_single_dirs(left_extend, right_extend);   // :540
        node_data_out = node_data;
    }
};
}  // namespace

// compress_kmers_no_exts (compression.rs:619-659)
int compress_kmers_no_exts(int k, bool stranded, Spec spec, const std::vector<Kmer>& keys_in, const std::vector<uint32_t>& data_in,
                           const uint64_t* seed_order, BaseGraph& out, std::vector<Exts>* exts_out, std::string& err) {
    KmerSpec ks(k);
    std::set<Kmer> kmer_set(keys_in.begin(), keys_in.end());        // :624
    std::vector<Kmer> keys; std::vector<Exts> exts; std::vector<uint32_t> data;
    for (size_t i = 0; i < keys_in.size(); i++) {                    // :631
        const Kmer kmer = keys_in[i];
        Exts e;                                                      // Exts::empty() :632
        for (uint8_t l = 0; l < 4; l++) {                            // :634-640
            Kmer nw = kmer_min_rc(ks, kmer_extend_left(ks, kmer, l));       // `can` is min_rc whatever `stranded` is (:626)
            if (kmer_set.count(nw)) e = e.set(Left, l);
        }
        for (uint8_t r = 0; r < 4; r++) {                            // :642-648
            Kmer nw = kmer_min_rc(ks, kmer_extend_right(ks, kmer, r));
            if (kmer_set.count(nw)) e = e.set(Right, r);
        }
        keys.push_back(kmer); data.push_back(data_in[i]); exts.push_back(e);   // :650-652
    }
    if (kmer_set.size() != keys.size()) { err = "assertion failed: kmer_set.len() == keys.len() (compression.rs:655)"; return 1; }
    if (exts_out) *exts_out = exts;
    return compress_kmers_with_hash(k, stranded, spec, keys, exts, data, seed_order, out, err);   // :657-658
}

int compress_kmers_with_hash(int k, bool stranded, Spec spec,
                             const std::vector<Kmer>& keys, const std::vector<Exts>& exts,
                             const std::vector<uint32_t>& data, const uint64_t* seed_order,
                             BaseGraph& out, std::string& err) {
    CompressFromHash comp(k, stranded, spec, keys, exts, data, &err);
    out = BaseGraph();
    out.stranded = stranded;                                        // :566
    std::vector<std::pair<Kmer, Dir>> path_buf;
    std::deque<uint8_t> edge_seq_buf;
    size_t n = keys.size();
    for (size_t c = 0; c < n; c++) {                                // :574
        size_t kmer_counter = seed_order ? (size_t)seed_order[c] : c;
        if (comp.available[kmer_counter]) {                         // :575
            Exts ne; uint32_t nd;
            comp.build_node(kmer_counter, path_buf, edge_seq_buf, ne, nd);
            if (comp.failed) return 1;
            out.sequences.add(edge_seq_buf.begin(), edge_seq_buf.end());   // graph.rs:104-113
            out.exts.push_back(ne);
            out.data.push_back(nd);
        }
    }
    return 0;
}

// ---------------------------------------------------------------------------
// BaseGraph::combine / finish, DebruijnGraph::find_link etc.
// ---------------------------------------------------------------------------
int graph_combine(const std::vector<BaseGraph>& graphs, BaseGraph& out, std::string& err) {   // graph.rs:71-100
    out = BaseGraph();
    bool all_s = true, none_s = true;
    for (auto& g : graphs) {
        for (size_t s = 0; s < g.sequences.len(); s++) {
            SeqView v = g.sequences.get(s);
            std::vector<uint8_t> b(v.length);
            for (size_t i = 0; i < v.length; i++) b[i] = v.get(i);
            out.sequences.add(b.begin(), b.end());
        }
        out.exts.insert(out.exts.end(), g.exts.begin(), g.exts.end());
        out.data.insert(out.data.end(), g.data.begin(), g.data.end());
        all_s = all_s && g.stranded;
        none_s = none_s && !g.stranded;
    }
    if (!all_s && !none_s) { err = "attempted to combine stranded and unstranded graphs (graph.rs:90)"; return 1; }
    out.stranded = all_s;
    return 0;
}

void graph_finish(int k, BaseGraph&& base, DebruijnGraph& out) {    // graph.rs:116-142
    out.base = std::move(base);
    out.k = k;
    KmerSpec ks(k);
    out.left_order.clear(); out.right_order.clear();
    for (size_t i = 0; i < out.base.len(); i++) {
        SeqView v = out.base.sequences.get(i);
        out.left_order.push_back({seq_get_kmer(ks, v, 0), (uint32_t)i});              // first_kmer
        out.right_order.push_back({seq_get_kmer(ks, v, v.length - k), (uint32_t)i});  // last_kmer
    }
    auto cmp = [](auto& a, auto& b) { return a.first < b.first; };
    std::stable_sort(out.left_order.begin(), out.left_order.end(), cmp);
    std::stable_sort(out.right_order.begin(), out.right_order.end(), cmp);
}

static long search_kmer(const DebruijnGraph& g, Kmer kmer, Dir side) {   // graph.rs:243-249
    auto& v = side == Left ? g.left_order : g.right_order;
    auto it = std::lower_bound(v.begin(), v.end(), kmer, [](auto& a, Kmer b) { return a.first < b; });
    if (it != v.end() && it->first == kmer) return it->second;
    return -1;
}

bool graph_find_link(const DebruijnGraph& g, Kmer kmer, Dir dir, size_t& node, Dir& ndir, bool& flip) {   // graph.rs:252-291
    KmerSpec ks(g.k);
    Kmer rc = kmer_rc(ks, kmer);
    long idx;
    if (dir == Left) {
        if ((idx = search_kmer(g, kmer, Right)) >= 0) { node = idx; ndir = Right; flip = false; return true; }
        if (!g.base.stranded && (idx = search_kmer(g, rc, Left)) >= 0) { node = idx; ndir = Left; flip = true; return true; }
    } else {
        if ((idx = search_kmer(g, kmer, Left)) >= 0) { node = idx; ndir = Left; flip = false; return true; }
        if (!g.base.stranded && (idx = search_kmer(g, rc, Right)) >= 0) { node = idx; ndir = Right; flip = true; return true; }
    }
    return false;
}

struct Edge { size_t node; Dir dir; bool flip; };
static Kmer term_kmer(const DebruijnGraph& g, size_t node_id, Dir dir) {   // lib.rs:384-389
    KmerSpec ks(g.k);
    SeqView v = g.base.sequences.get(node_id);
    return dir == Left ? seq_get_kmer(ks, v, 0) : seq_get_kmer(ks, v, v.length - g.k);
}
static std::vector<Edge> find_edges(const DebruijnGraph& g, size_t node_id, Dir dir) {   // graph.rs:223-241
    KmerSpec ks(g.k);
    Exts e = g.base.exts[node_id];
    Kmer kmer = term_kmer(g, node_id, dir);
    std::vector<Edge> edges;
    for (uint8_t i = 0; i < 4; i++) {
        if (e.has_ext(dir, i)) {
            size_t n; Dir nd; bool f;
            if (graph_find_link(g, kmer_extend(ks, kmer, i, dir), dir, n, nd, f)) edges.push_back(Edge{n, nd, f});
        }
    }
    return edges;
}

// node_to_gfa + write_gfa (graph.rs:537-611) without tags
std::string graph_write_gfa(const DebruijnGraph& g) {
    std::string out = "H\tVN:Z:debruijn-rs\n";                                          // :604
    for (size_t i = 0; i < g.base.len(); i++) {                                         // :609-612
        SeqView v = g.base.sequences.get(i);
        out += "S\t" + std::to_string(i) + "\t";                                        // :557-562
        for (size_t p = 0; p < v.length; p++) out += "ACGT"[v.get(p)];                  // Display, dna_string.rs:466-473
        out += "\n";
        for (auto& e : find_edges(g, i, Left)) {                                        // :565-579
            if (e.node >= i)
                out += "L\t" + std::to_string(i) + "\t-\t" + std::to_string(e.node) + "\t" + (e.dir == Left ? "+" : "-") + "\t" +
                       std::to_string(g.k - 1) + "M\n";
        }
        for (auto& e : find_edges(g, i, Right)) {                                       // :581-595
            if (e.node > i)
                out += "L\t" + std::to_string(i) + "\t+\t" + std::to_string(e.node) + "\t" + (e.dir == Left ? "+" : "-") + "\t" +
                       std::to_string(g.k - 1) + "M\n";
        }
    }
    return out;
}

bool graph_is_compressed(const DebruijnGraph& g, Spec spec, size_t& a, size_t& b) {   // graph.rs:296-334
    KmerSpec ks(g.k);
    for (size_t i = 0; i < g.base.len(); i++) {
        for (int d = 0; d < 2; d++) {
            Dir dir = (Dir)d;
            auto dir_edges = find_edges(g, i, dir);
            if (dir_edges.size() == 1) {
                size_t next_id = dir_edges[0].node; Dir return_dir = dir_edges[0].dir;
                auto ret_edges = find_edges(g, next_id, return_dir);
                if (ret_edges.size() == 1) {
                    SeqView n = g.base.sequences.get(i), nx = g.base.sequences.get(next_id);
                    if (n.length == (size_t)g.k && kmer_is_palindrome(ks, seq_get_kmer(ks, n, 0))) continue;
                    if (nx.length == (size_t)g.k && kmer_is_palindrome(ks, seq_get_kmer(ks, nx, 0))) continue;
                    if (i == next_id) continue;
                    if (spec_join_test(spec, g.base.data[i], g.base.data[next_id])) { a = i; b = next_id; return false; }
                }
            }
        }
    }
    return true;
}

static Exts get_valid_exts(const DebruijnGraph& g, size_t node_id, const std::vector<uint8_t>* valid) {   // graph.rs:344-377
    KmerSpec ks(g.k);
    Exts new_exts = Exts::empty();
    Exts e = g.base.exts[node_id];
    Kmer l_kmer = term_kmer(g, node_id, Left), r_kmer = term_kmer(g, node_id, Right);
    for (uint8_t i = 0; i < 4; i++) {
        size_t t; Dir nd; bool f;
        if (e.has_ext(Left, i)) {
            if (graph_find_link(g, kmer_extend_left(ks, l_kmer, i), Left, t, nd, f) && (!valid || (*valid)[t]))
                new_exts = new_exts.set(Left, i);
        }
        if (e.has_ext(Right, i)) {
            if (graph_find_link(g, kmer_extend_right(ks, r_kmer, i), Right, t, nd, f) && (!valid || (*valid)[t]))
                new_exts = new_exts.set(Right, i);
        }
    }
    return new_exts;
}
static void fix_exts(DebruijnGraph& g, const std::vector<uint8_t>* valid) {   // graph.rs:337-342
    for (size_t i = 0; i < g.base.len(); i++) g.base.exts[i] = get_valid_exts(g, i, valid);
}

// CompressFromGraph (compression.rs:100-335)
namespace {
struct ExtModeNode { bool unique; size_t node; Dir dir; Exts exts; };
struct CompressFromGraph {
    bool stranded; Spec spec; std::vector<uint8_t> available; const DebruijnGraph& graph; std::string* err; bool failed = false;
    KmerSpec ks;
    CompressFromGraph(bool st, Spec sp, std::vector<uint8_t> av, const DebruijnGraph& g, std::string* e)
        : stranded(st), spec(sp), available(std::move(av)), graph(g), err(e), ks(g.k) {}
    void fail(const std::string& m) { if (!failed) { failed = true; *err = m; } }

    ExtModeNode try_extend_node(size_t node, Dir dir) {             // :115-205
        SeqView bases = graph.base.sequences.get(node);
        Exts e = graph.base.exts[node];
        if (e.num_ext_dir(dir) != 1 ||
            (!stranded && bases.length == (size_t)ks.k && kmer_is_palindrome(ks, seq_get_kmer(ks, bases, 0)))) {   // :120-123
            return ExtModeNode{false, 0, dir, e.single_dir(dir)};
        }
        int ext_base = e.get_unique_extension(dir);
        Kmer end_kmer = term_kmer(graph, node, dir);                // :127
        Kmer next_kmer = kmer_extend(ks, end_kmer, (uint8_t)ext_base, dir);
        size_t next_node_id; Dir next_side_incoming; bool rc;
        if (!graph_find_link(graph, next_kmer, dir, next_node_id, next_side_incoming, rc)) {   // :130-140
            fail("No kmer (compression.rs:138)");
            return ExtModeNode{false, 0, dir, e.single_dir(dir)};
        }
        SeqView next_node = graph.base.sequences.get(next_node_id);
        Exts next_exts = graph.base.exts[next_node_id];
        bool consistent = next_node.length == (size_t)ks.k ||       // :145-165
                          (dir == Left && next_side_incoming == Right && !rc) ||
                          (dir == Left && next_side_incoming == Left && rc) ||
                          (dir == Right && next_side_incoming == Left && !rc) ||
                          (dir == Right && next_side_incoming == Right && rc);
        if (!consistent) { fail("assert consistent (compression.rs:165)"); return ExtModeNode{false, 0, dir, e.single_dir(dir)}; }
        if (!available[next_node_id] || (!stranded && kmer_is_palindrome(ks, next_kmer)) ||
            !spec_join_test(spec, graph.base.data[node], graph.base.data[next_node_id])) {   // :173-182
            return ExtModeNode{false, 0, dir, e.single_dir(dir)};
        }
        Dir next_side_outgoing = dir_flip(next_side_incoming);      // :185
        uint8_t incoming_count = next_exts.num_ext_dir(next_side_incoming);
        Exts outgoing_exts = next_exts.single_dir(next_side_outgoing);
        if (incoming_count == 0) { fail("unreachable (compression.rs:195)"); return ExtModeNode{false, 0, dir, e.single_dir(dir)}; }
        else if (incoming_count == 1) return ExtModeNode{true, next_node_id, next_side_outgoing, outgoing_exts};
        return ExtModeNode{false, 0, dir, e.single_dir(dir)};
    }
    Exts extend_node(size_t start_node, Dir start_dir, std::vector<std::pair<size_t, Dir>>& path) {   // :208-235
        Dir current_dir = start_dir; size_t current_node = start_node;
        path.clear();
        available[start_node] = 0;
        while (!failed) {
            ExtModeNode r = try_extend_node(current_node, current_dir);
            if (r.unique) {
                Dir next_dir_incoming = dir_flip(r.dir);
                path.push_back({r.node, next_dir_incoming});
                available[r.node] = 0;
                current_node = r.node;
                current_dir = r.dir;
            } else return r.exts;
        }
        return Exts();
    }
    void build_node(size_t seed_node, std::vector<uint8_t>& seq, Exts& ne, uint32_t& nd) {   // :240-287
        std::vector<std::pair<size_t, Dir>> l_path, r_path;
        Exts l_ext = extend_node(seed_node, Left, l_path);
        Exts r_ext = extend_node(seed_node, Right, r_path);
        std::deque<std::pair<size_t, Dir>> node_path;
        uint32_t node_data = graph.base.data[seed_node];
        bool panic = false;
        node_path.push_back({seed_node, Left});
        for (auto& pr : l_path) {                                   // :251-256
            node_path.push_front({pr.first, dir_flip(pr.second)});
            node_data = spec_reduce(spec, node_data, graph.base.data[pr.first], panic);
        }
        for (auto& pr : r_path) {                                   // :259-264
            node_path.push_back({pr.first, pr.second});
            node_data = spec_reduce(spec, node_data, graph.base.data[pr.first], panic);
        }
        Exts left_extend = l_ext;                                   // :266-270
        if (!l_path.empty() && l_path.back().second == Left) left_extend = l_ext.complement();
        Exts right_extend = r_ext;                                  // :272-276
        if (!r_path.empty() && r_path.back().second == Right) right_extend = r_ext.complement();
        if (panic) fail("Should not happen: ScmapCompress reduce on unequal data (compression.rs:90)");
        // sequence_of_path (graph.rs:471-491)
        seq.clear();
        size_t idx = 0;
        for (auto& pr : node_path) {
            SeqView v = graph.base.sequences.get(pr.first);
            size_t start = idx == 0 ? 0 : (size_t)ks.k - 1;
            for (size_t p = start; p < v.length; p++) {
                uint8_t b = pr.second == Left ? v.get(p) : complement(v.get(v.length - 1 - p));   // DnaStringSlice::rc get :572-578
                seq.push_back(b);
            }
            idx++;
        }
        ne = Exts::from_single_dirs(left_extend, right_extend);
        nd = node_data;
    }
};
}  // namespace

int compress_graph(bool stranded, Spec spec, DebruijnGraph&& old_graph,
                   const std::vector<size_t>* censor_nodes, DebruijnGraph& out, std::string& err) {   // compression.rs:291-334
    size_t n_nodes = old_graph.base.len();
    std::vector<uint8_t> available(n_nodes, 1);
    if (censor_nodes) for (size_t c : *censor_nodes) available[c] = 0;
    fix_exts(old_graph, &available);                                // :309
    CompressFromGraph comp(stranded, spec, available, old_graph, &err);
    BaseGraph graph; graph.stranded = stranded;                     // :320
    std::vector<uint8_t> seq;
    for (size_t node_counter = 0; node_counter < n_nodes; node_counter++) {   // :322-327
        if (comp.available[node_counter]) {
            Exts ne; uint32_t nd;
            comp.build_node(node_counter, seq, ne, nd);
            if (comp.failed) return 1;
            graph.sequences.add(seq.begin(), seq.end());
            graph.exts.push_back(ne);
            graph.data.push_back(nd);
        }
    }
    graph_finish(old_graph.k, std::move(graph), out);               // :330
    fix_exts(out, nullptr);                                         // :331
    return 0;
}

}  // namespace orc
