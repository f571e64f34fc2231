// Device-side unitig construction for compress_kmers_with_hash (src/compression.rs:355-583).
//
// The reference grows each unitig greedily from a seed (compression.rs:450-541) in seed order
// (compression.rs:574).  Because the "Unique" relation of try_extend_kmer (compression.rs:382-444) is
// mutual -- A extends to B through side d exactly when B extends back to A through its facing side -- the
// k-mers form disjoint simple chains (and rare simple cycles), and the greedy result is a function of the
// chains and the seed ranks alone:
//   * a chain becomes one node; its seed is the member with the smallest seed rank; the node's sequence is
//     read in the seed's stored orientation; nodes are emitted in increasing seed rank;
//   * an isolated cycle is cut at the right side of its seed (the left walk consumes the whole cycle first).
// That makes the construction data-parallel.  Three routes, fastest first (compress_links_device picks):
//   1. chain route (k-mer tables): every chain end walks its chain (chain_scan_kernel: length, seed, which end is the
//      unitig's left end); after two prefix sums one lane per node walks it again and writes it front to back
//      (chain_emit_kernel).  No per-state table; one cache line per step when link_kernel left NodeRec records.
//   2. end walk into a table (compress_graph, or route 1 switched off): walk_ends_kernel leaves in every state "(element,
//      side I leave through)" its distance to the chain end it faces, that end and the minimum rank on the way; then
//      per-element arithmetic, two prefix sums and atomic ORs into the packed output (emit_kernel / emit_nodes_kernel).
//   3. doubling (jump_kernel): the same table by pointer jumping -- for cycles (cut at their seed, then redone) and for
//      chains too long to walk (WALK_CAP).
// If the links are not mutual (inconsistent Exts, which also makes the reference panic or depend on visiting order)
// the caller falls back to the literal host walk.
#include "dbg_internal.hpp"
#include <algorithm>

namespace {
constexpr uint32_t U_TERM = 0xFFFFFFFFu;
constexpr uint32_t U_PANIC = 0x80000000u;
constexpr uint32_t ST_NONE = 0xFFFFFFFFu;
constexpr uint32_t R_INF = 0xFFFFFFFFu;

struct __attribute__((aligned(16))) Jump {
    uint32_t nxt;       // state reached after `dist` steps, ST_NONE once the walk has hit a terminal side
    uint32_t dist;      // k-mers passed so far (excluding the start)
    uint32_t minr;      // minimum seed rank among them
    uint32_t endst;     // valid when nxt == ST_NONE: the terminal state (k-mer, side) the walk stopped at
};

__device__ __forceinline__ bool link_valid(uint32_t L, uint32_t self) {
    return L != U_TERM && !(L & U_PANIC) && ((L & 0x7FFFFFFFu) >> 1) != self;     // self links are terminal (compression.rs:410-415)
}

// flags: bit0 = a PANIC link exists, bit1 = a non-mutual link exists
// avail (compress_graph's censor list, compression.rs:297-307): 0 = the element may neither be entered nor emitted; a
// link into it is terminal for the walk that meets it (compression.rs:173-182).  null = everything available.
__global__ void check_links_kernel(const uint32_t* __restrict__ link, const uint8_t* __restrict__ avail, uint32_t n, uint32_t* __restrict__ flags) {
    uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= 2 * n) return;
    uint32_t i = s >> 1, p = s & 1;
    if (avail && !avail[i]) return;
    uint32_t L = link[(uint64_t)p * n + i];
    if (L == U_TERM) return;
    if (avail && !avail[(L & 0x7FFFFFFFu) >> 1]) return;                // terminal before the panic test is reached (compression.rs:173-195)
    if (L & U_PANIC) { atomicOr(flags, 1u); return; }
    uint32_t j = L >> 1, nd = L & 1;
    if (j == i) return;
    uint32_t q = 1 - nd;                                           // side of j that faces i
    uint32_t back = link[(uint64_t)q * n + j];
    if (back != ((i << 1) | (1 - p))) atomicOr(flags, 2u);
}

// weight[i] = k-mers element i stands for (null: every element is one k-mer); distances are counted in k-mers
__global__ void init_states_kernel(const uint32_t* __restrict__ link, const uint32_t* __restrict__ rank, const uint32_t* __restrict__ weight,
                                   const uint8_t* __restrict__ avail, uint32_t n, Jump* __restrict__ J) {
    uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= 2 * n) return;
    uint32_t i = s >> 1, p = s & 1;
    uint32_t L = link[(uint64_t)p * n + i];
    Jump o;
    const bool usable = link_valid(L, i) && (!avail || (avail[i] && avail[(L & 0x7FFFFFFFu) >> 1]));
    if (usable) { uint32_t j = L >> 1, nd = L & 1; o.nxt = 2 * j + nd; o.dist = weight ? weight[j] : 1u; o.minr = rank ? rank[j] : j; o.endst = ST_NONE; }
    else { o.nxt = ST_NONE; o.dist = 0; o.minr = R_INF; o.endst = s; }
    J[s] = o;
}

// One doubling round over the LIVE states only (most chains are short: after r rounds only the states more than 2^r
// steps from their chain end are still walking).  A state that reaches its end in this round is kept for one more round,
// in which it only copies its final value into the other buffer -- so a finished state reads the same from both buffers
// and no buffer is written while it is being read.
__global__ void jump_kernel(const Jump* __restrict__ in, Jump* __restrict__ out, const uint32_t* __restrict__ live_in, uint32_t n_live,
                            uint32_t* __restrict__ live_out, uint32_t* __restrict__ counters /* [0] next live count, [1] still walking */) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    bool keep = false, walking = false;
    uint32_t s = 0;
    if (t < n_live) {
        s = live_in ? live_in[t] : t;
        Jump a = in[s];
        if (a.nxt != ST_NONE) {
            Jump b = in[a.nxt];
            a.dist += b.dist;
            a.minr = a.minr < b.minr ? a.minr : b.minr;
            a.nxt = b.nxt;
            if (b.nxt == ST_NONE) a.endst = b.endst;
            walking = a.nxt != ST_NONE;
            keep = true;                                          // walking on, or finalising next round
        }
        out[s] = a;
    }
    // ordered append: one global atomic per 1024-thread block (a hot single address otherwise), waves in order
    __shared__ uint32_t s_cnt[16], s_base;
    __shared__ uint32_t s_walk;
    const uint64_t km = __ballot(keep);
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_walk = 0;
    if (lane == 0) s_cnt[wave] = (uint32_t)__popcll(km);
    __syncthreads();
    if (__any(walking) && lane == 0) s_walk = 1;
    if (threadIdx.x == 0) {
        uint32_t tot = 0;
        for (uint32_t w = 0; w < blockDim.x / 64; w++) { uint32_t x = s_cnt[w]; s_cnt[w] = tot; tot += x; }
        s_base = tot ? atomicAdd(&counters[0], tot) : 0u;
    }
    __syncthreads();
    if (keep) live_out[s_base + s_cnt[wave] + (uint32_t)__popcll(km & ((1ull << lane) - 1ull))] = s;
    if (threadIdx.x == 0 && s_walk) atomicOr(&counters[1], 1u);
}

// ---- fast route: every chain end walks its chain once -------------------------------------------------------------
// Chains are short on sequencing data (a few dozen k-mers between branch points), so the doubling above -- rounds over
// all 2n states -- does far more memory traffic than the chains have links.  Here every terminal state T = (k-mer, side
// it cannot be left through) walks inwards and leaves, in the state of each k-mer it passes that faces back towards T,
// the finished Jump value of that state: {ST_NONE, k-mers between it and T, their minimum rank, T}.  The two ends of a
// chain fill the two states of each of its k-mers between them; states on cycles are never reached, which the count of
// written states shows (the caller then runs the doubling, which also serves chains longer than WALK_CAP).
constexpr uint32_t WALK_CAP = 1u << 14;

__device__ __forceinline__ bool state_usable(const uint32_t* __restrict__ link, const uint8_t* __restrict__ avail, uint32_t n, uint32_t i, uint32_t p,
                                             uint32_t* L_out) {
    const uint32_t L = link[(uint64_t)p * n + i];
    *L_out = L;
    return link_valid(L, i) && (!avail || (avail[i] && avail[(L & 0x7FFFFFFFu) >> 1]));
}

// appends the terminal states to `ends` (any order).  A block covers 8192 states and reserves its share with one global
// atomic: one atomic per 1024 states on a single address (10^6 of them at config 3) was what this kernel waited for.
constexpr uint32_t ENDS_ITEMS = 8;
__global__ void __launch_bounds__(1024) collect_ends_kernel(const uint32_t* __restrict__ link, const uint8_t* __restrict__ avail, uint32_t n,
                                                            uint32_t* __restrict__ ends, uint32_t* __restrict__ n_ends,
                                                            uint32_t* __restrict__ link_flags = nullptr /* |= 1 when a panic link exists */) {
    const uint64_t base = (uint64_t)blockIdx.x * (1024 * ENDS_ITEMS);
    uint32_t mask = 0;
#pragma unroll
    for (uint32_t j = 0; j < ENDS_ITEMS; j++) {
        const uint64_t s = base + j * 1024 + threadIdx.x;
        uint32_t L;
        if (s < 2ull * n && !state_usable(link, avail, n, (uint32_t)(s >> 1), (uint32_t)(s & 1), &L)) {
            mask |= 1u << j;
            if (link_flags && L != U_TERM && (L & U_PANIC)) atomicOr(link_flags, 1u);
        }
    }
    const uint32_t cnt = (uint32_t)__popc(mask);
    __shared__ uint32_t s_cnt[16], s_base;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(incl, d); if (lane >= (uint32_t)d) incl += o; }
    if (lane == 63) s_cnt[wave] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tot = 0;
        for (uint32_t w = 0; w < 16; w++) { uint32_t x = s_cnt[w]; s_cnt[w] = tot; tot += x; }
        s_base = tot ? atomicAdd(n_ends, tot) : 0u;
    }
    __syncthreads();
    uint32_t o = s_base + s_cnt[wave] + incl - cnt;
#pragma unroll
    for (uint32_t j = 0; j < ENDS_ITEMS; j++)
        if (mask & (1u << j)) ends[o++] = (uint32_t)(base + j * 1024 + threadIdx.x);
}

// Persistent wavefronts: chain lengths differ widely, so a lane that finishes its chain takes the next chain end at once
// (each wave reserves WALK_GRAB ends per global atomic and deals them to its idle lanes by ballot) -- every lane keeps
// one dependent random read in flight instead of idling until the longest chain of its wave is done.
constexpr uint32_t WALK_GRAB = 256;
__global__ void __launch_bounds__(256) walk_ends_kernel(const uint32_t* __restrict__ link, const uint32_t* __restrict__ rank,
                                                        const uint32_t* __restrict__ weight, const uint8_t* __restrict__ avail, uint32_t n,
                                                        const uint32_t* __restrict__ ends, uint32_t n_ends, Jump* __restrict__ J,
                                                        uint32_t* __restrict__ next, unsigned long long* __restrict__ written,
                                                        uint32_t* __restrict__ capped, uint32_t walk_cap) {
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t lt = (1ull << lane) - 1ull;
    uint32_t wpos = 0, wend = 0;                                   // this wave's reserved range of `ends` (wave-uniform)
    bool exhausted = false, active = false;
    uint32_t total = 0, steps = 0, cur = 0, side = 0;
    Jump o; o.nxt = ST_NONE; o.dist = 0; o.minr = R_INF; o.endst = 0;
    for (;;) {
        const uint64_t idle = __ballot(!active);
        if (idle && !(exhausted && wpos == wend)) {
            if (wpos == wend) {
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(next, WALK_GRAB);
                base = __shfl(base, 0);
                wpos = base < n_ends ? base : n_ends;
                wend = base + WALK_GRAB < n_ends ? base + WALK_GRAB : n_ends;
                if (base + WALK_GRAB >= n_ends) exhausted = true;
            }
            const uint32_t have = wend - wpos, want = (uint32_t)__popcll(idle), mine = (uint32_t)__popcll(idle & lt);
            if (!active && mine < have) {
                const uint32_t T = ends[wpos + mine];
                o.nxt = ST_NONE; o.dist = 0; o.minr = R_INF; o.endst = T;
                J[T] = o;                                          // what init_states_kernel gives a terminal state
                total++;
                steps = 1;
                cur = T >> 1; side = 1u - (T & 1u);                // leave the end k-mer through its other side
                active = true;
            }
            wpos += want < have ? want : have;
        }
        if (!__any(active)) {
            if (exhausted && wpos == wend) break;
            continue;
        }
        if (active) {
            uint32_t L;
            if (!state_usable(link, avail, n, cur, side, &L)) active = false;   // (cur, side) is the far terminal state: its own walker writes it
            else {
                o.dist += weight ? weight[cur] : 1u;               // cur is now behind the walker
                const uint32_t r = rank ? rank[cur] : cur;
                o.minr = r < o.minr ? r : o.minr;
                const uint32_t j = L >> 1, nd = L & 1u;            // arrive at j, to be left through nd
                J[2 * j + (1u - nd)] = o;                          // j's state that faces back
                total++;
                cur = j; side = nd;
                if (++steps > walk_cap) { atomicOr(capped, 1u); active = false; }
            }
        }
    }
    for (int d = 32; d > 0; d >>= 1) total += __shfl_down(total, d, 64);
    if (lane == 0 && total) atomicAdd(written, (unsigned long long)total);
}

// Chains longer than WALK_CAP (the censored graphs real callers compress: 10^5 unitigs of kilobases, the longest 10^5 k-mers) leave the
// middle of the chain unwritten: the table was filled with UNWRITTEN before the walk, the states the walkers reached hold their final
// values, and only the others -- a few per cent of the states -- go through the doubling, starting from their one-step links.  (The
// doubling used to start over from all 2n states: 465 of the 670 ms of a censored config-3 graph.)
constexpr uint32_t J_UNWRITTEN = 0xFFFFFFFFu;                      // Jump::dist of a state no walker reached (a chain has fewer than 2^32 - 1 k-mers)
__global__ void __launch_bounds__(1024) init_unwritten_kernel(const uint32_t* __restrict__ link, const uint32_t* __restrict__ rank, const uint32_t* __restrict__ weight,
                                                              const uint8_t* __restrict__ avail, uint32_t n, Jump* __restrict__ J,
                                                              uint32_t* __restrict__ live_out, uint32_t* __restrict__ n_live) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    bool todo = false;
    if (s < 2 * n && J[s].dist == J_UNWRITTEN) {
        const uint32_t i = s >> 1, p = s & 1;
        const uint32_t L = link[(uint64_t)p * n + i];
        Jump o;
        const bool usable = link_valid(L, i) && (!avail || (avail[i] && avail[(L & 0x7FFFFFFFu) >> 1]));
        if (usable) { const uint32_t j = L >> 1, nd = L & 1; o.nxt = 2 * j + nd; o.dist = weight ? weight[j] : 1u; o.minr = rank ? rank[j] : j; o.endst = ST_NONE; todo = true; }
        else { o.nxt = ST_NONE; o.dist = 0; o.minr = R_INF; o.endst = s; }           // (a terminal state of an unavailable element: nobody walks from it)
        J[s] = o;
    }
    __shared__ uint32_t s_cnt[16], s_base;
    const uint64_t km = __ballot(todo);
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) s_cnt[wave] = (uint32_t)__popcll(km);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tot = 0;
        for (uint32_t w = 0; w < blockDim.x / 64; w++) { uint32_t x = s_cnt[w]; s_cnt[w] = tot; tot += x; }
        s_base = tot ? atomicAdd(n_live, tot) : 0u;
    }
    __syncthreads();
    if (todo) live_out[s_base + s_cnt[wave] + (uint32_t)__popcll(km & ((1ull << lane) - 1ull))] = s;
}

// after the doubling has covered 2n steps, any state still walking sits on a cycle; cut it at its seed's right side
__global__ void cut_cycles_kernel(const Jump* __restrict__ J, const uint32_t* __restrict__ rank, uint32_t* __restrict__ link, uint32_t n,
                                  uint32_t* __restrict__ seed_list = nullptr, uint32_t* __restrict__ seed_count = nullptr, uint32_t seed_cap = 0) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Jump a = J[2 * i];
    if (a.nxt == ST_NONE) return;
    uint32_t r = rank ? rank[i] : i;
    if (r <= a.minr) {                                             // I hold the smallest rank of my cycle
        uint32_t L = link[(uint64_t)n + i];                        // my right link
        uint32_t j = L >> 1, nd = L & 1;
        link[(uint64_t)n + i] = U_TERM;
        link[(uint64_t)(1 - nd) * n + j] = U_TERM;
        if (seed_count) { const uint32_t at = atomicAdd(seed_count, 1u); if (at < seed_cap) seed_list[at] = i; }
    }
}

struct NodeInfo { uint32_t seedrank, pos, m; bool toA_is_left; };
__device__ __forceinline__ NodeInfo node_info(const Jump* __restrict__ J, const uint32_t* __restrict__ rank, uint32_t i, uint32_t wi = 1u) {
    Jump a = J[2 * i], b = J[2 * i + 1];
    NodeInfo o;
    uint32_t r = rank ? rank[i] : i;
    uint32_t mr = a.minr < b.minr ? a.minr : b.minr;
    o.seedrank = r < mr ? r : mr;
    o.m = a.dist + b.dist + wi;                                    // k-mers of the whole chain
    o.toA_is_left = a.endst < b.endst;                             // chain end A = the smaller terminal state
    o.pos = o.toA_is_left ? a.dist : b.dist;                       // distance from end A
    return o;
}

__global__ void mark_seeds_kernel(const Jump* __restrict__ J, const uint32_t* __restrict__ rank, const uint32_t* __restrict__ weight,
                                  const uint8_t* __restrict__ avail, uint32_t n, int k,
                                  uint32_t* __restrict__ flag_by_rank, uint32_t* __restrict__ len_by_rank, uint8_t* __restrict__ rev_by_rank) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    NodeInfo f = node_info(J, rank, i, weight ? weight[i] : 1u);
    uint32_t r = rank ? rank[i] : i;
    if (r == f.seedrank && (!avail || avail[i])) {
        flag_by_rank[r] = 1;
        len_by_rank[r] = f.m + (uint32_t)k - 1;
        rev_by_rank[r] = f.toA_is_left ? 0 : 1;                    // seed's left side must face the unitig's left end
    }
}

__global__ void gather_lens_kernel(const uint32_t* __restrict__ flag_by_rank, const uint32_t* __restrict__ uidx_by_rank,
                                   const uint32_t* __restrict__ len_by_rank, uint32_t n, uint32_t* __restrict__ ulen) {
    uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n && flag_by_rank[r]) ulen[uidx_by_rank[r]] = len_by_rank[r];
}

__device__ __forceinline__ void or_bits(uint64_t* __restrict__ words, uint64_t base_off, K128 v, int nbases) {
    // OR `nbases` bases (right-aligned in v, base 0 most significant) into the packed stream at base offset base_off
    K128 top = k128_shl(v, 128 - 2 * nbases);                      // left-align
    uint64_t w = base_off >> 5;
    int s = (int)(base_off & 31) * 2;
    uint64_t p0 = s ? top.hi >> s : top.hi;
    uint64_t p1 = s ? (top.hi << (64 - s)) | (top.lo >> s) : top.lo;
    uint64_t p2 = s ? top.lo << (64 - s) : 0;
    if (p0) atomicOr((unsigned long long*)&words[w], (unsigned long long)p0);
    if (p1) atomicOr((unsigned long long*)&words[w + 1], (unsigned long long)p1);
    if (p2) atomicOr((unsigned long long*)&words[w + 2], (unsigned long long)p2);
}

// Folding an element's data into its node's accumulator.  A long node (clean sequence: chains of 10^8 k-mers) would receive
// every lane's atomic on one address (88 per microsecond): the lanes of a wavefront that feed the same node fold first and
// issue one atomic -- for the (up to four) most frequent nodes of the wave; whoever is left (short nodes: all different)
// uses its own atomic.  Must be reached by all 64 lanes (no early returns before it).  ucnt: element counter or null.
__device__ __forceinline__ void fold_into_node(bool valid, uint32_t ui, uint32_t d, int spec, bool is_seed,
                                               unsigned long long* __restrict__ uacc, uint32_t* __restrict__ ucnt) {
    if (spec == DBG_SPEC_SCMAP_EQ) {                               // the seed's value, stored by the seed
        if (valid && is_seed) uacc[ui] = d;
        if (!ucnt) return;
    }
    bool todo = valid;
    for (int round = 0; round < 4; round++) {
        const uint64_t vm = __ballot(todo);
        if (__popcll(vm) < 2) break;
        const int lead = __ffsll((long long)vm) - 1;
        const uint32_t ui0 = __shfl(ui, lead);
        const bool mine = todo && ui == ui0;
        if (__popcll(__ballot(mine)) < 8) break;                   // short nodes: folding would cost more than it saves
        unsigned long long sum = mine ? d : 0ull;
        uint32_t mx = mine ? d : 0u, cn = mine ? 1u : 0u;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            sum += __shfl_xor(sum, o);
            const uint32_t m2 = __shfl_xor(mx, o);
            mx = m2 > mx ? m2 : mx;
            cn += __shfl_xor(cn, o);
        }
        if ((int)(threadIdx.x & 63) == lead) {
            if (spec == DBG_SPEC_SIMPLE_MAX_U16) __hip_atomic_fetch_max((uint32_t*)&uacc[ui0], mx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else if (spec != DBG_SPEC_SCMAP_EQ) atomicAdd(&uacc[ui0], sum);
            if (ucnt) atomicAdd(&ucnt[ui0], cn);
        }
        todo = todo && !mine;
    }
    if (!todo) return;
    if (spec == DBG_SPEC_SIMPLE_MAX_U16) __hip_atomic_fetch_max((uint32_t*)&uacc[ui], d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // low word of the zeroed accumulator
    else if (spec != DBG_SPEC_SCMAP_EQ) atomicAdd(&uacc[ui], (unsigned long long)d);
    if (ucnt) atomicAdd(&ucnt[ui], 1u);
}

__global__ void emit_kernel(const Jump* __restrict__ J, const uint32_t* __restrict__ rank, uint32_t n, int k,
                            const uint64_t* __restrict__ key_hi, const uint64_t* __restrict__ key_lo, const uint8_t* __restrict__ exts,
                            const uint32_t* __restrict__ data, int spec, const uint32_t* __restrict__ uidx_by_rank,
                            const uint8_t* __restrict__ rev_by_rank, const uint64_t* __restrict__ ustart,
                            uint64_t* __restrict__ words, uint32_t* __restrict__ uexts, unsigned long long* __restrict__ uacc) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = i < n;
    uint32_t ui = 0, d = 0;
    bool is_seed = false;
    if (valid) {
        NodeInfo f = node_info(J, rank, i);
        ui = uidx_by_rank[f.seedrank];
        const bool rev = rev_by_rank[f.seedrank] != 0;
        const uint32_t u = rev ? f.m - 1 - f.pos : f.pos;              // position in the unitig
        const bool fwd = f.toA_is_left != rev;                         // stored orientation == unitig orientation?
        K128 km{key_hi ? key_hi[i] : 0ull, key_lo[i]};
        uint32_t e = exts[i];
        if (!fwd) { km = kmer_rc(km, k); e = exts_rc(e); }
        const uint64_t st = ustart[ui];
        if (u == 0) or_bits(words, st, km, k);                         // the first k-mer contributes all k bases
        else or_bits(words, st + u + (uint32_t)k - 1, K128{0, km.lo & 3ull}, 1);   // every other one its last base
        uint32_t eo = 0;
        if (u == 0) eo |= e & 0x0fu;                                   // left end keeps its outward (hanging) exts
        if (u == f.m - 1) eo |= e & 0xf0u;
        if (eo) atomicOr(&uexts[ui], eo);
        d = data ? data[i] : 0u;
        is_seed = (rank ? rank[i] : i) == f.seedrank;
    }
    fold_into_node(valid, ui, d, spec, is_seed, uacc, nullptr);
}

// reverse complement of nb (<= 32) bases held right-aligned in a u64
__device__ __forceinline__ uint64_t rc_bases64(uint64_t x, uint32_t nb) {
    uint64_t r = __brevll(x);
    r = ((r & 0x5555555555555555ull) << 1) | ((r >> 1) & 0x5555555555555555ull);
    return (~r) >> (64 - 2 * nb);
}

// Same construction for compress_graph (compression.rs:100-349): the elements are graph nodes (weight = k-mers of the
// node).  A node writes its (oriented) sequence -- everything when it opens the path, otherwise what follows the
// (k-1)-base overlap (sequence_of_path, graph.rs:471-491) -- 32 bases at a time.
__global__ void emit_nodes_kernel(const Jump* __restrict__ J, uint32_t n, int k, const uint32_t* __restrict__ weight, const uint8_t* __restrict__ avail,
                                  const uint64_t* __restrict__ nwords, const uint64_t* __restrict__ nstart, const uint32_t* __restrict__ nlen,
                                  const uint8_t* __restrict__ exts, const uint32_t* __restrict__ data, int spec,
                                  const uint32_t* __restrict__ uidx_by_rank, const uint8_t* __restrict__ rev_by_rank,
                                  const uint64_t* __restrict__ ustart, uint64_t* __restrict__ words, uint32_t* __restrict__ uexts,
                                  unsigned long long* __restrict__ uacc, uint32_t* __restrict__ ucnt) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = i < n && !(avail && !avail[i]);
    uint32_t ui = 0, d = 0;
    bool is_seed = false;
    if (valid) {
        const uint32_t wi = weight[i];
        NodeInfo f = node_info(J, nullptr, i, wi);
        ui = uidx_by_rank[f.seedrank];
        const bool rev = rev_by_rank[f.seedrank] != 0;
        const uint32_t u = rev ? f.m - f.pos - wi : f.pos;            // k-mer offset of the node's first k-mer in the path
        const bool fwd = f.toA_is_left != rev;                         // stored orientation == path orientation?
        uint32_t e = exts[i];
        if (!fwd) e = exts_rc(e);
        const uint32_t L = nlen[i];
        const uint64_t ns = nstart[i];
        const uint32_t skip = u == 0 ? 0u : (uint32_t)k - 1;           // bases already written by the previous node
        const uint64_t dst = ustart[ui] + u + skip;
        for (uint32_t o = skip; o < L; o += 32) {
            const uint32_t nb = L - o < 32 ? L - o : 32;
            uint64_t chunk;
            // (the node words carry two words of slack: graph.hip's dev_graph_build)
            if (fwd) chunk = packed_get_pmer64(nwords, ns + o, (int)nb, ~0ull);
            else chunk = rc_bases64(packed_get_pmer64(nwords, ns + (L - o - nb), (int)nb, ~0ull), nb);   // DnaStringSlice::rc (dna_string.rs:572-578)
            or_bits(words, dst + (o - skip), K128{0, chunk}, (int)nb);
        }
        uint32_t eo = 0;
        if (u == 0) eo |= e & 0x0fu;
        if (u + wi == f.m) eo |= e & 0xf0u;
        if (eo) atomicOr(&uexts[ui], eo);
        d = data ? data[i] : 0u;
        is_seed = i == f.seedrank;
    }
    fold_into_node(valid, ui, d, spec, is_seed, uacc, ucnt);
}

// ucnt (compress_graph only): elements joined into the node; null = one element per k-mer of the node
__global__ void finish_nodes_kernel(uint32_t n_nodes, int spec, int k, const uint32_t* __restrict__ ulen, const uint32_t* __restrict__ uexts,
                                    const unsigned long long* __restrict__ uacc, const uint32_t* __restrict__ ucnt,
                                    uint8_t* __restrict__ o_exts, uint32_t* __restrict__ o_data) {
    uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= n_nodes) return;
    o_exts[u] = (uint8_t)uexts[u];
    unsigned long long a = uacc[u];
    uint32_t m = ucnt ? ucnt[u] : ulen[u] - (uint32_t)k + 1;
    // (an if-chain: a switch with a `default` arm was miscompiled by hipcc 7.2 for gfx950 here)
    uint32_t d = (uint32_t)a;                                                              // max / ScmapCompress: the value itself
    if (spec == DBG_SPEC_SIMPLE_SAT_ADD_U16) d = a > 65535ull ? 65535u : (uint32_t)a;
    else if (spec == DBG_SPEC_SIMPLE_ADD_MOD_U16) { if (m != 1) d = (uint32_t)(a % 65535ull); }   // a lone k-mer is never reduced
    else if (spec == DBG_SPEC_SIMPLE_WRAP_ADD_U16) { if (m != 1) d = (uint32_t)(a & 0xFFFFull); }
    o_data[u] = d;
}

// ScmapCompress::reduce panics on unequal data (compression.rs:88-93); with join_test = equality a chain can
// only hold equal data, so nothing to check.  For SimpleCompress specs join_test is always true.
}  // namespace

namespace {
// ---- chain route for compress_kmers_with_hash (elements = k-mers) -----------------------------------------------------
// The same end walk, taken one step further: no per-state table at all.  Pass 1 (chain_scan_kernel): every chain end T
// walks to the far end and learns the chain's length, its seed (minimum rank) and whether the seed's left side faces T.
// Both ends of a chain see the same seed, and the seed's left side faces exactly one of them: that end is the unitig's
// left end ("nodes are read in the seed's stored orientation") and records {length, T} under the seed's rank.  After the
// scans that turn seed ranks into node numbers and lengths into offsets, pass 2 (chain_emit_kernel) lets one lane per
// node walk its chain from the left end and write the sequence front to back (first k-mer whole, then one base per
// k-mer, gathered into 64-bit words), the Exts of the two ends and the data fold.  Per link one random 4-byte read in
// each pass plus one 8-byte key read and one 4-byte data read in pass 2 -- no table of 2n x 16 bytes, no random atomics.
// Chains on cycles have no ends: the k-mers of all recorded chains then do not add up to n and the caller takes the
// general route.
struct WaveFeed {                                  // hands out items [0, n_items) to the idle lanes of persistent waves
    uint32_t wpos = 0, wend = 0;
    bool exhausted = false;
    // returns true in lanes that received item *out
    __device__ __forceinline__ bool feed(bool idle_lane, uint32_t n_items, uint32_t* __restrict__ next, uint32_t* out) {
        const uint32_t lane = threadIdx.x & 63;
        const uint64_t idle = __ballot(idle_lane);
        if (!idle || (exhausted && wpos == wend)) return false;
        if (wpos == wend) {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(next, WALK_GRAB);
            base = __shfl(base, 0);
            wpos = base < n_items ? base : n_items;
            wend = base + WALK_GRAB < n_items ? base + WALK_GRAB : n_items;
            if (base + WALK_GRAB >= n_items) exhausted = true;
        }
        const uint32_t have = wend - wpos, want = (uint32_t)__popcll(idle), mine = (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
        const bool got = idle_lane && mine < have;
        if (got) *out = wpos + mine;
        wpos += want < have ? want : have;
        return got;
    }
    __device__ __forceinline__ bool done() const { return exhausted && wpos == wend; }
};

__global__ void __launch_bounds__(256) chain_scan_kernel(const uint32_t* __restrict__ link, const uint32_t* __restrict__ rank, uint32_t n,
                                                         const uint32_t* __restrict__ ends, uint32_t n_ends, int k,
                                                         uint32_t* __restrict__ flag_by_rank, uint32_t* __restrict__ len_by_rank,
                                                         uint32_t* __restrict__ start_by_rank, uint32_t* __restrict__ seed_by_rank,
                                                         uint32_t* __restrict__ next, unsigned long long* __restrict__ kmers_seen,
                                                         uint32_t* __restrict__ capped,
                                                         const NodeRec* __restrict__ nrec /* or null */, uint32_t* __restrict__ link_flags,
                                                         uint32_t* __restrict__ done_bits /* one bit per state: its chain has been walked from the other end; or null */,
                                                         uint32_t chain_cap, uint32_t* __restrict__ visited = nullptr /* one bit per k-mer, set as the walk passes (or null) */) {
    // With node records both links of a k-mer arrive in the one line a step reads, so the walk itself verifies that every link
    // it takes is answered by the facing link of its target (check_links_kernel's test; every link of an open chain is taken
    // by one of the chain's two walkers): *link_flags |= 2 on a mismatch.
    WaveFeed wf;
    bool active = false;
    uint32_t T = 0, cur = 0, face = 0, m = 0, best = R_INF, seed = 0, expect_back = U_TERM;
    bool seed_left_faces_T = false;
    unsigned long long total = 0;
    for (;;) {
        uint32_t item;
        if (wf.feed(!active, n_ends, next, &item)) {
            T = ends[item];
            cur = T >> 1; face = T & 1u;                          // `face` = the side of cur that points towards T
            m = 0; best = R_INF; expect_back = U_TERM;
            // ONE walk per chain (round 4): the walker that reaches a chain's far end marks that end, and an end that is marked when
            // its turn comes is skipped.  A stale or late mark only costs a second walk (both then find the same seed, length and left
            // end; the claim on flag_by_rank lets one of them record).  With the ends dealt in list order to ~5*10^5 walkers at a
            // time, ~97 % of the chains are walked once instead of twice.
            active = !(done_bits && ((__hip_atomic_load(&done_bits[T >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> (T & 31u)) & 1u));
        }
        if (!__any(active)) {
            if (wf.done()) break;
            continue;
        }
        if (active) {
            const uint32_t r = rank ? rank[cur] : cur;
            if (r < best) { best = r; seed = cur; seed_left_faces_T = face == 0; }
            m++;
            if (visited) atomicOr(&visited[cur >> 5], 1u << (cur & 31u));
            uint32_t L;
            bool more;
            if (nrec) {
                const uint4 b = reinterpret_cast<const uint4*>(nrec + cur)[1];           // {link[0], link[1], data, exts}
                L = face == 0 ? b.y : b.x;
                more = link_valid(L, cur);
                if (expect_back != U_TERM && (face == 0 ? b.x : b.y) != expect_back) atomicOr(link_flags, 2u);
                expect_back = (cur << 1) | face;                   // what the next k-mer's facing link must say
            } else more = state_usable(link, nullptr, n, cur, 1u - face, &L);
            if (!more) {                                           // far end reached
                const uint32_t F = (cur << 1) | (1u - face);       // the far end as a state: (k-mer, its terminal side)
                if (done_bits) {
                    // the left end of the unitig (in the seed's stored orientation) is T when the seed's left side faces T, else F
                    if (atomicCAS(&flag_by_rank[best], 0u, 1u) == 0u) {
                        len_by_rank[best] = m + (uint32_t)k - 1;
                        start_by_rank[best] = seed_left_faces_T ? T : F;
                        seed_by_rank[best] = seed;
                        total += m;
                    }
                    if (F != T) atomicOr(&done_bits[F >> 5], 1u << (F & 31u));
                } else if (seed_left_faces_T) {                    // two walks per chain: the one that started at the left end records
                    flag_by_rank[best] = 1;
                    len_by_rank[best] = m + (uint32_t)k - 1;
                    start_by_rank[best] = T;
                    seed_by_rank[best] = seed;
                    total += m;
                }
                active = false;
            } else {
                cur = L >> 1; face = 1u - (L & 1u);
                if (m > chain_cap) { atomicOr(capped, 1u); active = false; }
            }
        }
    }
    for (int d = 32; d > 0; d >>= 1) total += __shfl_down(total, d, 64);
    if ((threadIdx.x & 63) == 0 && total) atomicAdd(kmers_seen, total);
}

__global__ void gather_chains_kernel(const uint32_t* __restrict__ flag_by_rank, const uint32_t* __restrict__ uidx_by_rank,
                                     const uint32_t* __restrict__ len_by_rank, const uint32_t* __restrict__ start_by_rank,
                                     const uint32_t* __restrict__ seed_by_rank, uint32_t n, uint32_t* __restrict__ ulen,
                                     uint32_t* __restrict__ ufirst, uint32_t* __restrict__ useed) {
    uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n && flag_by_rank[r]) { const uint32_t u = uidx_by_rank[r]; ulen[u] = len_by_rank[r]; ufirst[u] = start_by_rank[r]; useed[u] = seed_by_rank[r]; }
}

__global__ void __launch_bounds__(256) chain_emit_kernel(const uint32_t* __restrict__ link, uint32_t n, int k, const uint64_t* __restrict__ key_hi,
                                                         const uint64_t* __restrict__ key_lo, const uint8_t* __restrict__ exts,
                                                         const uint32_t* __restrict__ data, int spec, uint32_t node_lo, uint32_t n_nodes,
                                                         const uint32_t* __restrict__ ufirst, const uint32_t* __restrict__ useed,
                                                         const uint64_t* __restrict__ ustart, uint64_t* __restrict__ words,
                                                         uint32_t* __restrict__ uexts, unsigned long long* __restrict__ uacc, uint32_t* __restrict__ next,
                                                         const NodeRec* __restrict__ nrec /* or null: read the separate arrays */) {
    WaveFeed wf;
    bool active = false;
    uint32_t ui = 0, cur = 0, face = 0, u = 0, seed = 0, eo = 0;
    uint64_t pos = 0, wacc = 0, widx = 0;
    unsigned long long acc = 0;
    const int top_shift = 2 * (k - 1) - (k > 32 ? 64 : 0);          // position of a key's first base inside hi (k > 32) or lo
    for (;;) {
        uint32_t item;
        if (wf.feed(!active, n_nodes, next, &item)) {
            ui = node_lo + item;
            const uint32_t T = ufirst[ui];
            cur = T >> 1; face = T & 1u;
            seed = useed[ui];
            pos = ustart[ui];
            u = 0; acc = 0; wacc = 0; widx = 0; eo = 0;
            active = true;
        }
        if (!__any(active)) {
            if (wf.done()) break;
            continue;
        }
        if (active) {
            const bool fwd = face == 0;                            // the k-mer's left side points to the unitig's left end
            // everything this step needs about the k-mer: one 32-byte record, or the separate arrays
            uint64_t klo, khi = 0;
            uint32_t e, d, L;
            if (nrec) {
                const ulonglong2 a = reinterpret_cast<const ulonglong2*>(nrec + cur)[0];
                const uint4 b = reinterpret_cast<const uint4*>(nrec + cur)[1];
                klo = a.x; khi = a.y; L = face == 0 ? b.y : b.x; d = b.z; e = b.w;       // leave through side 1 - face
            } else {
                klo = (u == 0 || fwd || k <= 32) ? key_lo[cur] : 0ull;
                if (key_hi && (u == 0 || !fwd)) khi = key_hi[cur];
                L = link[(uint64_t)(1u - face) * n + cur];
                d = data ? data[cur] : 0u;
                e = 0;
            }
            const bool more = link_valid(L, cur);
            if (!nrec && (u == 0 || !more)) e = exts[cur];
            if (!fwd) e = exts_rc(e);
            if (u == 0) {                                          // the first k-mer contributes all k bases
                K128 km{khi, klo};
                if (!fwd) km = kmer_rc(km, k);
                or_bits(words, pos, km, k);
                eo = e & 0x0fu;                                    // the left end keeps its outward (hanging) exts
                pos += (uint32_t)k;
                widx = pos >> 5;
            } else {                                               // every other one its last base
                uint64_t b;
                if (fwd) b = klo & 3ull;
                else b = 3ull - (((k > 32 ? khi : klo) >> top_shift) & 3ull);
                if ((pos >> 5) != widx) {
                    if (wacc) atomicOr((unsigned long long*)&words[widx], (unsigned long long)wacc);
                    wacc = 0; widx = pos >> 5;
                }
                wacc |= b << (62 - 2 * (uint32_t)(pos & 31));
                pos++;
            }
            if (spec == DBG_SPEC_SIMPLE_MAX_U16) acc = d > acc ? d : acc;
            else if (spec == DBG_SPEC_SCMAP_EQ) { if (cur == seed) acc = d; }
            else acc += d;
            u++;
            if (!more) {                                           // the right end
                eo |= e & 0xf0u;
                if (wacc) atomicOr((unsigned long long*)&words[widx], (unsigned long long)wacc);
                uexts[ui] = eo;
                uacc[ui] = acc;
                active = false;
            } else {
                cur = L >> 1; face = 1u - (L & 1u);
            }
        }
    }
}

// ---- segment route: long chains in two levels -----------------------------------------------------------------------------
// A censored graph (what real callers compress) has 10^5 chains of thousands of k-mers, the longest 10^5: one walker per chain end keeps
// 2.6e5 lanes busy for as long as the longest walk (92 ms of walking + 33 ms of doubling + a 16 GB table at config-3 size).  Here every
// `every`-th k-mer (by id: ids are in key order, unrelated to the position in a chain) cuts the link on its right side.  The pieces
// (~`every` k-mers, the longest ~20 x that) go through the chain route like any short chains, and the resulting BaseGraph of
// segments is joined by compress_graph's device route (graph.hip), whose rules at a cut -- one extension each way, mutual, same
// join_test -- are the k-mer rules, and at a real chain end see the same Exts the k-mer walk saw.  Seeds carry over: segments are
// numbered by their seed's rank and stored in their seed's orientation, the unitig's seed segment holds the unitig's seed.  What does
// not carry over is the cut of a CYCLE (at its seed k-mer's right side): every open chain has two ends, so unitigs != ends / 2 means
// a cycle somewhere and the caller takes the table route on the untouched links.
__global__ void cut_links_kernel(const uint32_t* __restrict__ link_in, uint32_t* __restrict__ link, NodeRec* __restrict__ nrec, uint32_t n, uint32_t every,
                                 uint32_t* __restrict__ flags) {
    const uint64_t i64 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * every;
    if (i64 >= n) return;
    const uint32_t i = (uint32_t)i64;
    const uint32_t L = link_in[(uint64_t)n + i];                  // side 1 of k-mer i
    if (!link_valid(L, i)) return;
    const uint32_t j = L >> 1, q = 1u - (L & 1u);                 // the side of j that faces i (check_links_kernel)
    if (link_in[(uint64_t)q * n + j] != (i << 1)) { atomicOr(flags, 2u); return; }   // not mutual: the walks below check every link they take, this one they no longer see
    link[(uint64_t)n + i] = U_TERM;
    link[(uint64_t)q * n + j] = U_TERM;
    if (nrec) { nrec[i].link[1] = U_TERM; nrec[j].link[q] = U_TERM; }
}

// A cycle NO sampled cut fell into (short ones: 100 k-mers escape every 128th with probability 0.46) has no end, so no walker of the
// first level passes it: its k-mers are the ones left unmarked after a marking walk.  Every such k-mer walks its cycle once (they are
// short: 2000 k-mers escape with probability 1.5e-7) and the one that holds the smallest rank cuts the link on its right side, in the
// call's own links -- the reference's cut.  `gave_up`: a walk that did not come back within the cap (the table route then).
__global__ void unvisited_cycle_seeds_kernel(const uint32_t* __restrict__ link, const uint32_t* __restrict__ rank, uint32_t n, const uint32_t* __restrict__ visited,
                                             uint32_t* __restrict__ outer_link, NodeRec* __restrict__ nrec, uint32_t* __restrict__ n_cut, uint32_t* __restrict__ gave_up) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || ((visited[i >> 5] >> (i & 31u)) & 1u)) return;
    const uint32_t mine = rank ? rank[i] : i;
    uint32_t cur = i, face = 0;                                    // leave through the side opposite `face`
    for (uint32_t steps = 0;; steps++) {
        const uint32_t L = link[(uint64_t)(1u - face) * n + cur];
        if (!link_valid(L, cur) || steps > (1u << 16)) { atomicOr(gave_up, 1u); return; }
        cur = L >> 1; face = 1u - (L & 1u);
        if (cur == i) break;                                       // (either way round: the minimum has been seen over the whole cycle)
        if ((rank ? rank[cur] : cur) < mine) return;               // not the seed
    }
    const uint32_t L = outer_link[(uint64_t)n + i];
    if (!link_valid(L, i)) return;
    const uint32_t j = L >> 1, q = 1u - (L & 1u);
    outer_link[(uint64_t)n + i] = U_TERM;
    outer_link[(uint64_t)q * n + j] = U_TERM;
    if (nrec) { nrec[i].link[1] = U_TERM; nrec[j].link[q] = U_TERM; }
    atomicAdd(n_cut, 1u);
}

// A cycle with exactly ONE cut is a single piece whose right side leads to its own left side: a self link, terminal for the walk
// (compression.rs:410-415 at the node level), so no cycle of pieces shows -- it is reported here.
__global__ void self_loop_seeds_kernel(const uint32_t* __restrict__ link, uint32_t n, uint32_t* __restrict__ seed_list, uint32_t* __restrict__ seed_count, uint32_t seed_cap) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (link[(uint64_t)n + i] == ((i << 1) | 1u)) { const uint32_t at = atomicAdd(seed_count, 1u); if (at < seed_cap) seed_list[at] = i; }
}

// A cycle that holds a cut shows up as a cycle of pieces in the second level, which names its seed piece (cut_cycles_kernel's list);
// the cycle's seed k-mer is that piece's seed.  The link on its right side is removed from the call's links for good -- exactly the
// cut the reference makes ("an isolated cycle is cut at the right side of its seed") -- and the route runs again on open chains only.
__global__ void cut_at_piece_seeds_kernel(const uint32_t* __restrict__ seed_pieces, uint32_t n_seeds, const uint32_t* __restrict__ piece_seed,
                                          uint32_t* __restrict__ link, NodeRec* __restrict__ nrec, uint32_t n, uint32_t* __restrict__ n_cut) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_seeds) return;
    const uint32_t i = piece_seed[seed_pieces[t]];
    const uint32_t L = link[(uint64_t)n + i];
    if (!link_valid(L, i)) return;
    const uint32_t j = L >> 1, q = 1u - (L & 1u);
    link[(uint64_t)n + i] = U_TERM;
    link[(uint64_t)q * n + j] = U_TERM;
    if (nrec) { nrec[i].link[1] = U_TERM; nrec[j].link[q] = U_TERM; }
    atomicAdd(n_cut, 1u);
}

}  // namespace

// Builds the BaseGraph on the device from the neighbour links.  rank_dev: seed rank of every (sorted) k-mer id,
// or null for the identity.  link_dev is modified when cycles are cut.  *done = false (nothing produced)
// when the links are not mutual or contain a panic marker: the caller then runs the literal host walk.
int compress_links_device(dbg_ctx* c, int k, uint32_t n, const uint64_t* key_hi, const uint64_t* key_lo, const uint8_t* exts,
                          const uint32_t* data, uint32_t* link_dev, const uint32_t* rank_dev, int spec, int stranded,
                          dbg_graph* out, bool* done, const UnitigNodes* nodes, const NodeRec* nrec) {
    *done = false;
    const uint32_t* weight = nodes ? nodes->weight : nullptr;
    const uint8_t* avail = nodes ? nodes->avail : nullptr;
    if (n == 0 || n >= (1u << 30)) return 0;
    const uint32_t n2 = 2 * n;
    DBuf<uint32_t> flags;
    ALLOC_OR_FAIL(c, flags, 2);
    HIP_TRY(c, hipMemsetAsync(flags.p, 0, 8, c->stream));
    const bool try_chains = !nodes && !c->opt("DBG_UNITIG_NO_CHAINS") && !c->opt("DBG_UNITIG_NO_WALK");
    // the chain route with node records checks the links while it walks them; everything else checks them first
    bool links_checked = false;
    auto check_links = [&]() -> int {
        c->t_begin("unitig_check_links", n);
        check_links_kernel<<<cdiv(n2, 256), 256, 0, c->stream>>>(link_dev, avail, n, flags.p);
        c->t_end();
        LAUNCH_CHECK(c, "check_links");
        links_checked = true;
        return 0;
    };
    auto links_bad = [&](bool* bad) -> int {
        uint32_t fl[2] = {0, 0};
        HIP_TRY(c, hipMemcpyAsync(fl, flags.p, 8, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        *bad = fl[0] != 0;
        return 0;
    };
    if (!(try_chains && nrec)) {
        DBG_TRY(check_links());
        bool bad = false;
        DBG_TRY(links_bad(&bad));
        if (bad) return 0;
    }

    if (nodes && c->cycle_seed_count) {                             // (second level of the segment route)
        self_loop_seeds_kernel<<<cdiv(n, 256), 256, 0, c->stream>>>(link_dev, n, c->cycle_seed_list, c->cycle_seed_count, c->cycle_seed_cap);
        LAUNCH_CHECK(c, "self_loop_seeds");
    }

    DBuf<uint32_t> LA, counters;
    ALLOC_OR_FAIL(c, LA, n2);
    ALLOC_OR_FAIL(c, counters, 8);
    // results of either route
    DBuf<uint32_t> flag_by_rank, len_by_rank, uidx_by_rank, ulen, uexts, o_data, useed;
    DBuf<uint64_t> ustart, words;
    DBuf<unsigned long long> uacc;
    DBuf<uint8_t> o_exts;
    uint32_t n_nodes = 0;
    uint64_t total_bases = 0, n_words = 0;
    bool emitted = false, early_nodes = false;
    uint64_t early_words = 0;
    dbg_graph out_early{};                                         // host arrays the chain route fills while its kernels run

    // ---- chain route (k-mers only): two walks per chain, no per-state table ----
    if (try_chains) {
        DBuf<uint32_t> start_by_rank, seed_by_rank, ufirst;
        ALLOC_OR_FAIL(c, flag_by_rank, n); ALLOC_OR_FAIL(c, len_by_rank, n); ALLOC_OR_FAIL(c, uidx_by_rank, (size_t)n + 1);
        ALLOC_OR_FAIL(c, start_by_rank, n); ALLOC_OR_FAIL(c, seed_by_rank, n);
        HIP_TRY(c, hipMemsetAsync(flag_by_rank.p, 0, (size_t)n * 4, c->stream));
        HIP_TRY(c, hipMemsetAsync(counters.p, 0, 32, c->stream));
        c->t_begin("unitig_chain_scan", n);
        collect_ends_kernel<<<cdiv(n2, 1024 * ENDS_ITEMS), 1024, 0, c->stream>>>(link_dev, nullptr, n, LA.p, counters.p, links_checked ? nullptr : flags.p);
        LAUNCH_CHECK(c, "collect_ends");
        uint32_t n_ends = 0;
        HIP_TRY(c, hipMemcpyAsync(&n_ends, counters.p, 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        // long chains (a censored graph: thousands of k-mers per chain) are not for this route -- its two walks per chain are as long
        // as the longest chain, and they give up at WALK_CAP: straight to the table routes
        uint32_t chain_cap = WALK_CAP;
        if (const char* e = c->opt("DBG_CHAIN_CAP")) chain_cap = (uint32_t)std::max(16, atoi(e));     // measurement / tests
        // (DBG_SEGMENTS_FORCE, tests: every graph takes the segment route below, whatever the length of its chains)
        const bool force_segments = c->opt("DBG_SEGMENTS_FORCE") && c->segment_depth == 0 && n_ends;
        const bool long_chains = (n_ends && (uint64_t)n2 / n_ends > 512u && !c->opt("DBG_CHAIN_CAP")) || force_segments;
        DBuf<uint32_t> done_bits;
        const bool one_walk = !(c->opt("DBG_CHAIN_WALKS") && !strcmp(c->opt("DBG_CHAIN_WALKS"), "2"));
        if (n_ends && one_walk) {
            ALLOC_OR_FAIL(c, done_bits, (size_t)(n2 + 31) / 32 + 1);
            HIP_TRY(c, hipMemsetAsync(done_bits.p, 0, ((size_t)(n2 + 31) / 32 + 1) * 4, c->stream));
        }
        if (n_ends && !long_chains) {
            chain_scan_kernel<<<std::min<uint32_t>(cdiv(n_ends, 256), 2048), 256, 0, c->stream>>>(
                link_dev, rank_dev, n, LA.p, n_ends, k, flag_by_rank.p, len_by_rank.p, start_by_rank.p, seed_by_rank.p,
                counters.p + 4, (unsigned long long*)(counters.p + 2), counters.p + 1, links_checked ? nullptr : nrec, flags.p, done_bits.p, chain_cap);
            LAUNCH_CHECK(c, "chain_scan");
        }
        uint32_t res[4] = {0, 0, 0, 0};
        HIP_TRY(c, hipMemcpyAsync(res, counters.p, 16, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        c->t_end();
        const uint64_t seen = (uint64_t)res[2] | ((uint64_t)res[3] << 32);
        if (!links_checked) {                                      // the walk checked the links of the chains it covered
            bool bad = false;
            DBG_TRY(links_bad(&bad));
            if (bad) return 0;
        }
        const bool ok = !long_chains && res[1] == 0 && seen == n;  // no walker gave up, every k-mer sits on an open chain
        if (c->opt("DBG_DEBUG")) fprintf(stderr, "[unitig] %u chain ends, chains hold %llu of %u k-mers%s\n", n_ends, (unsigned long long)seen, n,
                                         ok ? "" : " -> general route");
        if (!ok && c->segment_depth > 0) {
            // first level of the segment route: k-mers no walker passed = cycles without a cut.  Found by a marking walk, cut at their seeds
            // in the caller's links; the caller runs the route again (or, if nothing could be cut, the table route).
            c->segment_new_cuts = 0;
            if (res[1] == 0 && n_ends && c->segment_outer_links) {
                DBuf<uint32_t> visited, cc;
                const size_t vw = ((size_t)n + 31) / 32;
                ALLOC_OR_FAIL(c, visited, vw); ALLOC_OR_FAIL(c, cc, 2);
                HIP_TRY(c, hipMemsetAsync(visited.p, 0, vw * 4, c->stream));
                HIP_TRY(c, hipMemsetAsync(cc.p, 0, 8, c->stream));
                HIP_TRY(c, hipMemsetAsync(counters.p + 1, 0, 20, c->stream));       // the walk's own counters: capped, k-mers seen, next end
                c->t_begin("unitig_chain_scan", n);
                chain_scan_kernel<<<std::min<uint32_t>(cdiv(n_ends, 256), 2048), 256, 0, c->stream>>>(
                    link_dev, rank_dev, n, LA.p, n_ends, k, flag_by_rank.p, len_by_rank.p, start_by_rank.p, seed_by_rank.p,
                    counters.p + 4, (unsigned long long*)(counters.p + 2), counters.p + 1, nullptr, flags.p, nullptr, chain_cap, visited.p);
                unvisited_cycle_seeds_kernel<<<cdiv(n, 256), 256, 0, c->stream>>>(link_dev, rank_dev, n, visited.p, c->segment_outer_links, const_cast<NodeRec*>(nrec), cc.p, cc.p + 1);
                c->t_end();
                LAUNCH_CHECK(c, "unvisited_cycle_seeds");
                uint32_t h[2] = {0, 0};
                HIP_TRY(c, hipMemcpyAsync(h, cc.p, 8, hipMemcpyDeviceToHost, c->stream));
                HIP_TRY(c, hipStreamSynchronize(c->stream));
                if (!h[1]) c->segment_new_cuts = h[0];
                if (c->opt("DBG_DEBUG")) fprintf(stderr, "[unitig] segment route: %llu k-mers on cycles without a cut, %u cycles cut at their seeds%s\n", (unsigned long long)(n - seen), h[0], h[1] ? " (a walk gave up)" : "");
            }
            return 0;
        }
        // ---- segment route: long chains are cut into pieces, the pieces are compressed, the pieces are joined ----
        // (also when the chain route met a chain beyond its cap among short ones: a few very long chains in an ordinary graph)
        if (!ok && (long_chains || res[1] != 0) && !(c->opt("DBG_SEGMENTS") && atoi(c->opt("DBG_SEGMENTS")) == 0)) {
            // (links: with node records the pieces' walks check every link they take, as the chain route does, and the cutter checks the
            //  links it removes; without records check_links_kernel has run above)
            const uint32_t every = c->opt("DBG_SEGMENTS") ? (uint32_t)std::max(2, atoi(c->opt("DBG_SEGMENTS"))) : 128u;   // 32 / 64 / 128 / 256: 112 / 102 / 98 / 103 ms at config-3 size
            // this level's chain-route buffers go back to the pool: the first level below asks for the same sizes
            flag_by_rank.release(); len_by_rank.release(); uidx_by_rank.release(); start_by_rank.release(); seed_by_rank.release(); done_bits.release();
            DBuf<uint32_t> link_cut, cyc_list, cyc_ctl;
            constexpr uint32_t CYC_CAP = 1u << 16;
            ALLOC_OR_FAIL(c, cyc_list, CYC_CAP); ALLOC_OR_FAIL(c, cyc_ctl, 2);
            GraphDev* outer_sink = c->graph_sink;
            uint64_t ends_now = n_ends;                            // terminal states of the call's links (two more per cycle cut open below)
            // A cycle cannot be joined the way an open chain is (its cut belongs at its seed k-mer, not at a sampled one): the second
            // level reports the seed pieces of the cycles it met, their seed k-mers are cut for good, and the route runs once more.
            for (int attempt = 0; attempt < 3 && !emitted; attempt++) {
                ALLOC_OR_FAIL(c, link_cut, n2);
                c->t_begin("unitig_segments", n);
                HIP_TRY(c, hipMemcpyAsync(link_cut.p, link_dev, (size_t)n2 * 4, hipMemcpyDeviceToDevice, c->stream));
                cut_links_kernel<<<cdiv(cdiv(n, every), 256), 256, 0, c->stream>>>(link_dev, link_cut.p, const_cast<NodeRec*>(nrec), n, every, flags.p);   // (the records are this call's own)
                c->t_end();
                LAUNCH_CHECK(c, "cut_links");
                GraphDev pieces, joined;
                c->graph_sink = &pieces;
                c->segment_outer_links = link_dev;
                c->segment_new_cuts = 0;
                c->segment_depth++;
                dbg_graph sizes_only;
                bool pieces_done = false;
                int r = compress_links_device(c, k, n, key_hi, key_lo, exts, data, link_cut.p, rank_dev, spec, stranded, &sizes_only, &pieces_done, nullptr, nrec);
                c->segment_depth--;
                c->graph_sink = outer_sink;
                c->segment_outer_links = nullptr;
                if (r) return r;
                link_cut.release();
                {
                    bool bad = false;                              // a cut link that was not mutual
                    DBG_TRY(links_bad(&bad));
                    if (bad) return 0;
                }
                if (!pieces_done) {                                // cycles without a cut among the pieces (or links the walks rejected)
                    if (c->segment_new_cuts && attempt < 2) { ends_now += 2ull * c->segment_new_cuts; continue; }   // cut at their seeds below: once more
                    if (c->opt("DBG_DEBUG")) fprintf(stderr, "[unitig] segment route: the pieces hold a cycle that could not be cut -> table route\n");
                    break;
                }
                const uint64_t n_pieces = pieces.n_nodes;
                HIP_TRY(c, hipMemsetAsync(cyc_ctl.p, 0, 8, c->stream));
                c->graph_sink = nullptr;                           // (graph_dev_compress installs its own)
                c->cycle_seed_list = cyc_list.p; c->cycle_seed_count = cyc_ctl.p; c->cycle_seed_cap = CYC_CAP;
                r = graph_dev_join_segments(c, k, stranded, spec, &pieces, &joined);
                c->cycle_seed_list = nullptr; c->cycle_seed_count = nullptr; c->cycle_seed_cap = 0;
                c->graph_sink = outer_sink;
                if (r && r != 48) return r;                        // 48: node links not mutual -- the table route decides what that means
                const bool joined_ok = r == 0 && joined.filled && joined.n_nodes * 2 == ends_now;
                uint32_t ctl[2] = {0, 0};
                HIP_TRY(c, hipMemcpyAsync(ctl, cyc_ctl.p, 8, hipMemcpyDeviceToHost, c->stream));
                HIP_TRY(c, hipStreamSynchronize(c->stream));
                if (c->opt("DBG_DEBUG")) fprintf(stderr, "[unitig] segment route: %llu pieces (every %u) -> %llu unitigs, %llu chain ends, %u cycles of pieces%s\n",
                                                 (unsigned long long)n_pieces, every, (unsigned long long)joined.n_nodes, (unsigned long long)ends_now, ctl[0],
                                                 joined_ok ? "" : (r == 0 && ctl[0] && ctl[0] <= CYC_CAP && pieces.seed.p && attempt < 2 ? " -> cycles cut at their seeds, once more" : " -> table route"));
                if (joined_ok) {
                    words = std::move(joined.words); ustart = std::move(joined.start); ulen = std::move(joined.length);
                    o_exts = std::move(joined.exts); o_data = std::move(joined.data);
                    n_nodes = (uint32_t)joined.n_nodes; n_words = joined.n_words; total_bases = joined.n_bases;
                    emitted = true;
                    c->t_begin("unitig_segments_joined", n_nodes); // (a marker in the timing list: the route ran to its end)
                    c->t_end();
                    break;
                }
                if (r || !ctl[0] || ctl[0] > CYC_CAP || !pieces.seed.p) break;
                cut_at_piece_seeds_kernel<<<cdiv(ctl[0], 256), 256, 0, c->stream>>>(cyc_list.p, ctl[0], pieces.seed.p, link_dev, const_cast<NodeRec*>(nrec), n, cyc_ctl.p + 1);
                LAUNCH_CHECK(c, "cut_at_piece_seeds");
                HIP_TRY(c, hipMemcpyAsync(ctl, cyc_ctl.p, 8, hipMemcpyDeviceToHost, c->stream));
                HIP_TRY(c, hipStreamSynchronize(c->stream));
                if (!ctl[1]) break;
                ends_now += 2ull * ctl[1];
            }
        }
        if (ok) {
            DBG_TRY(scan_exclusive_u32(c, flag_by_rank.p, uidx_by_rank.p, n));
            HIP_TRY(c, hipMemcpyAsync(&n_nodes, uidx_by_rank.p + n, 4, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            const uint32_t nn = std::max<uint32_t>(n_nodes, 1);
            ALLOC_OR_FAIL(c, ulen, nn); ALLOC_OR_FAIL(c, ufirst, nn); ALLOC_OR_FAIL(c, useed, nn);
            ALLOC_OR_FAIL(c, ustart, (size_t)n_nodes + 1);
            ALLOC_OR_FAIL(c, uexts, nn); ALLOC_OR_FAIL(c, uacc, nn); ALLOC_OR_FAIL(c, o_exts, nn); ALLOC_OR_FAIL(c, o_data, nn);
            gather_chains_kernel<<<cdiv(n, 256), 256, 0, c->stream>>>(flag_by_rank.p, uidx_by_rank.p, len_by_rank.p, start_by_rank.p, seed_by_rank.p, n,
                                                                      ulen.p, ufirst.p, useed.p);
            LAUNCH_CHECK(c, "gather_chains");
            DBG_TRY(scan_exclusive_u32_u64(c, ulen.p, ustart.p, n_nodes));
            HIP_TRY(c, hipMemcpyAsync(&total_bases, ustart.p + n_nodes, 8, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            n_words = (total_bases + 31) / 32;
            ALLOC_OR_FAIL(c, words, n_words + 3);
            HIP_TRY(c, hipMemsetAsync(words.p, 0, (n_words + 3) * 8, c->stream));
            HIP_TRY(c, hipMemsetAsync(counters.p + 6, 0, 4, c->stream));
            // The nodes are emitted in two halves so that results can leave while kernels still run: offsets and lengths are
            // final already, the first half of the sequence words is final when the second half starts.  (A copy into pageable
            // host memory blocks the host thread, not the device: it is issued on a second stream after the kernels it overlaps.)
            if (!c->graph_sink) {
                // (pinned blocks from the ctx's result pool: a copy into pageable memory runs at a third of the link's rate and
                //  takes a page fault per 4 KB of a fresh array -- 680 MB of graph at config-3 size cost ~40 ms that way)
                out_early.seq_words = (uint64_t*)ctx_halloc(c, std::max<uint64_t>(n_words, 1) * 8);
                out_early.start = (uint64_t*)ctx_halloc(c, (size_t)nn * 8);
                out_early.length = (uint32_t*)ctx_halloc(c, (size_t)nn * 4);
            }
            hipStream_t cs = c->graph_sink ? nullptr : c->get_copy_stream();
            c->t_begin("unitig_emit", n);
            if (n_nodes) {
                const uint32_t half = n_nodes / 2;
                uint64_t first_half_bases = 0;
                if (half) HIP_TRY(c, hipMemcpyAsync(&first_half_bases, ustart.p + half, 8, hipMemcpyDeviceToHost, c->stream));
                HIP_TRY(c, hipMemsetAsync(counters.p + 6, 0, 8, c->stream));
                HIP_TRY(c, hipStreamSynchronize(c->stream));
                hipEvent_t ev = c->get_event();
                if (half) {
                    chain_emit_kernel<<<std::min<uint32_t>(cdiv(half, 256), 2048), 256, 0, c->stream>>>(
                        link_dev, n, k, key_hi, key_lo, exts, data, spec, 0, half, ufirst.p, useed.p, ustart.p, words.p, uexts.p, uacc.p, counters.p + 6, nrec);
                    LAUNCH_CHECK(c, "chain_emit");
                }
                HIP_TRY(c, hipEventRecord(ev, c->stream));
                chain_emit_kernel<<<std::min<uint32_t>(cdiv(n_nodes - half, 256), 2048), 256, 0, c->stream>>>(
                    link_dev, n, k, key_hi, key_lo, exts, data, spec, half, n_nodes - half, ufirst.p, useed.p, ustart.p, words.p, uexts.p, uacc.p, counters.p + 7, nrec);
                LAUNCH_CHECK(c, "chain_emit");
                finish_nodes_kernel<<<cdiv(n_nodes, 256), 256, 0, c->stream>>>(n_nodes, spec, k, ulen.p, uexts.p, uacc.p, nullptr, o_exts.p, o_data.p);
                LAUNCH_CHECK(c, "finish_nodes");
                if (cs) {
                    HIP_TRY(c, hipMemcpyAsync(out_early.start, ustart.p, (size_t)n_nodes * 8, hipMemcpyDeviceToHost, cs));
                    HIP_TRY(c, hipMemcpyAsync(out_early.length, ulen.p, (size_t)n_nodes * 4, hipMemcpyDeviceToHost, cs));
                    early_nodes = true;
                    // words wholly inside the first half: the word that holds the boundary base may still receive bits
                    early_words = first_half_bases / 32;
                    if (early_words) {
                        HIP_TRY(c, hipStreamWaitEvent(cs, ev, 0));
                        HIP_TRY(c, hipMemcpyAsync(out_early.seq_words, words.p, early_words * 8, hipMemcpyDeviceToHost, cs));
                    }
                    HIP_TRY(c, hipStreamSynchronize(cs));
                }
                c->event_pool.push_back(ev);
            }
            c->t_end();
            emitted = true;
        }
    }

    if (!emitted) {
    if (!links_checked) {                                          // links off the walked chains (cycles, long chains) are unchecked so far
        DBG_TRY(check_links());
        bool bad = false;
        DBG_TRY(links_bad(&bad));
        if (bad) return 0;
    }
    DBuf<Jump> JA, JB;
    DBuf<uint32_t> LB;
    ALLOC_OR_FAIL(c, JA, n2);
    Jump* cur = nullptr;
    bool walked = false;
    bool partial = false;                                          // the walk filled a part of the table: the doubling takes the rest only
    if (!c->opt("DBG_UNITIG_NO_WALK")) {
        HIP_TRY(c, hipMemsetAsync(counters.p, 0, 24, c->stream));
        c->t_begin("unitig_table_init", n);
        HIP_TRY(c, hipMemsetAsync(JA.p, 0xFF, (size_t)n2 * sizeof(Jump), c->stream));      // every state UNWRITTEN
        c->t_end();
        c->t_begin("unitig_walk_ends", n);
        collect_ends_kernel<<<cdiv(n2, 1024 * ENDS_ITEMS), 1024, 0, c->stream>>>(link_dev, avail, n, LA.p, counters.p);
        LAUNCH_CHECK(c, "collect_ends");
        uint32_t n_ends = 0;
        HIP_TRY(c, hipMemcpyAsync(&n_ends, counters.p, 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        // a handful of ends in front of millions of states (an error-free genome: one chain; its pieces in the segment route's second
        // level): the walkers would cover n_ends x 2^15 states at one dependent read per microsecond each -- 30 ms for 4 ends -- and
        // leave nearly everything to the doubling anyway
        const bool few_ends = n_ends && (uint64_t)n2 / n_ends > 65536u && !c->opt("DBG_WALK_CAP");
        if (n_ends && !few_ends) {
            const uint32_t wblocks = std::min<uint32_t>(cdiv(n_ends, 256), 2048);        // 8 resident blocks per CU
            // the table walk goes twice as far as the chain route's: at config-3 size (censored: 1.3e5 chains of ~3900 k-mers, the longest 1e5)
            // 2^14 left 16 % of the states to the doubling (156 ms), 2^15 leaves 3 % (30 ms) for 10 ms more of walking, 2^16 nothing for 36 more
            uint32_t walk_cap = 2 * WALK_CAP;
            if (const char* e = c->opt("DBG_WALK_CAP")) walk_cap = (uint32_t)std::max(16, atoi(e));       // measurement / tests
            walk_ends_kernel<<<wblocks, 256, 0, c->stream>>>(link_dev, rank_dev, weight, avail, n, LA.p, n_ends, JA.p, counters.p + 4,
                                                             (unsigned long long*)(counters.p + 2), counters.p + 1, walk_cap);
            LAUNCH_CHECK(c, "walk_ends");
        }
        uint32_t res[4] = {0, 0, 0, 0};
        HIP_TRY(c, hipMemcpyAsync(res, counters.p, 16, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        c->t_end();
        const uint64_t written = (uint64_t)res[2] | ((uint64_t)res[3] << 32);
        walked = !few_ends && res[1] == 0 && written == n2;         // no walker gave up, no state on a cycle
        if (c->opt("DBG_DEBUG")) fprintf(stderr, "[unitig] %u chain ends wrote %llu of %u states%s\n", n_ends, (unsigned long long)written, n2,
                                         walked ? "" : " -> doubling");
        if (walked) cur = JA.p;
        else partial = !c->opt("DBG_UNITIG_FULL_DOUBLING");
    }
    if (!walked) {
        ALLOC_OR_FAIL(c, JB, n2);
        ALLOC_OR_FAIL(c, LB, n2);
    }
    for (int phase = 0; phase < 2 && !walked; phase++) {
        Jump *a = JA.p, *b = JB.p;
        uint32_t *la = nullptr, *lb = LB.p;                         // round 0 visits every state
        uint32_t n_live = n2;
        if (phase == 0 && partial) {
            // the states the walkers did not reach start from their one-step links; everything else is final already, in both buffers
            HIP_TRY(c, hipMemsetAsync(counters.p, 0, 4, c->stream));
            c->t_begin("unitig_table_init", n);
            init_unwritten_kernel<<<cdiv(n2, 1024), 1024, 0, c->stream>>>(link_dev, rank_dev, weight, avail, n, JA.p, LA.p, counters.p);
            LAUNCH_CHECK(c, "init_unwritten");
            HIP_TRY(c, hipMemcpyAsync(JB.p, JA.p, (size_t)n2 * sizeof(Jump), hipMemcpyDeviceToDevice, c->stream));
            c->t_end();
            HIP_TRY(c, hipMemcpyAsync(&n_live, counters.p, 4, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            la = LA.p;
            if (c->opt("DBG_DEBUG")) fprintf(stderr, "[unitig] doubling over the %u states the walkers left (%.1f %% of %u)\n", n_live, 100.0 * n_live / n2, n2);
        } else {
            init_states_kernel<<<cdiv(n2, 256), 256, 0, c->stream>>>(link_dev, rank_dev, weight, avail, n, JA.p);
            LAUNCH_CHECK(c, "init_states");
        }
        bool walking = true;
        int rounds = 0;
        const int max_rounds = 34;                                 // 2^33 steps > any chain (+ the finalising round)
        c->t_begin("unitig_pointer_jump", n);
        while (n_live && rounds < max_rounds) {
            HIP_TRY(c, hipMemsetAsync(counters.p, 0, 8, c->stream));
            jump_kernel<<<cdiv(n_live, 1024), 1024, 0, c->stream>>>(a, b, la, n_live, lb, counters.p);
            LAUNCH_CHECK(c, "jump");
            uint32_t cnt[2] = {0, 0};
            HIP_TRY(c, hipMemcpyAsync(cnt, counters.p, 8, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            std::swap(a, b);
            la = lb; lb = (lb == LB.p) ? LA.p : LB.p;
            n_live = cnt[0];
            walking = cnt[1] != 0;
            rounds++;
            if (c->opt("DBG_DEBUG") && partial) fprintf(stderr, "[unitig]   round %d: %u states live\n", rounds, n_live);
        }
        c->t_end();
        // every state's final value is in the buffer written last; states that finished earlier were copied into both
        cur = a;
        if (!n_live) break;
        (void)walking;
        if (phase == 1) return c->fail(150, "unitig construction: cycle cutting did not terminate");
        // cycles: cut each at its seed's right side and redo the doubling
        cut_cycles_kernel<<<cdiv(n, 256), 256, 0, c->stream>>>(cur, rank_dev, link_dev, n, c->cycle_seed_list, c->cycle_seed_count, c->cycle_seed_cap);
        LAUNCH_CHECK(c, "cut_cycles");
    }
    // ---- seeds -> node order -> offsets ----
    DBuf<uint8_t> rev_by_rank;
    ALLOC_OR_FAIL(c, flag_by_rank, n); ALLOC_OR_FAIL(c, len_by_rank, n); ALLOC_OR_FAIL(c, uidx_by_rank, (size_t)n + 1);
    ALLOC_OR_FAIL(c, rev_by_rank, n);
    c->t_begin("unitig_seeds", n);                                  // seeds, node order, offsets, output buffers cleared
    HIP_TRY(c, hipMemsetAsync(flag_by_rank.p, 0, (size_t)n * 4, c->stream));
    mark_seeds_kernel<<<cdiv(n, 256), 256, 0, c->stream>>>(cur, rank_dev, weight, avail, n, k, flag_by_rank.p, len_by_rank.p, rev_by_rank.p);
    LAUNCH_CHECK(c, "mark_seeds");
    DBG_TRY(scan_exclusive_u32(c, flag_by_rank.p, uidx_by_rank.p, n));
    HIP_TRY(c, hipMemcpyAsync(&n_nodes, uidx_by_rank.p + n, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    ALLOC_OR_FAIL(c, ulen, std::max<uint32_t>(n_nodes, 1));
    ALLOC_OR_FAIL(c, ustart, (size_t)n_nodes + 1);
    ALLOC_OR_FAIL(c, uexts, std::max<uint32_t>(n_nodes, 1));
    gather_lens_kernel<<<cdiv(n, 256), 256, 0, c->stream>>>(flag_by_rank.p, uidx_by_rank.p, len_by_rank.p, n, ulen.p);
    LAUNCH_CHECK(c, "gather_lens");
    DBG_TRY(scan_exclusive_u32_u64(c, ulen.p, ustart.p, n_nodes));
    HIP_TRY(c, hipMemcpyAsync(&total_bases, ustart.p + n_nodes, 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    n_words = (total_bases + 31) / 32;
    ALLOC_OR_FAIL(c, words, n_words + 3);
    ALLOC_OR_FAIL(c, uacc, std::max<uint32_t>(n_nodes, 1));
    ALLOC_OR_FAIL(c, o_exts, std::max<uint32_t>(n_nodes, 1));
    ALLOC_OR_FAIL(c, o_data, std::max<uint32_t>(n_nodes, 1));
    HIP_TRY(c, hipMemsetAsync(words.p, 0, (n_words + 3) * 8, c->stream));
    HIP_TRY(c, hipMemsetAsync(uacc.p, 0, (size_t)std::max<uint32_t>(n_nodes, 1) * 8, c->stream));
    HIP_TRY(c, hipMemsetAsync(uexts.p, 0, (size_t)std::max<uint32_t>(n_nodes, 1) * 4, c->stream));
    DBuf<uint32_t> ucnt;
    c->t_end();
    c->t_begin("unitig_emit", n);
    if (nodes) {
        ALLOC_OR_FAIL(c, ucnt, std::max<uint32_t>(n_nodes, 1));
        HIP_TRY(c, hipMemsetAsync(ucnt.p, 0, (size_t)std::max<uint32_t>(n_nodes, 1) * 4, c->stream));
        emit_nodes_kernel<<<cdiv(n, 256), 256, 0, c->stream>>>(cur, n, k, weight, avail, nodes->words, nodes->start, nodes->length, exts, data, spec,
                                                               uidx_by_rank.p, rev_by_rank.p, ustart.p, words.p, uexts.p, uacc.p, ucnt.p);
    } else {
        emit_kernel<<<cdiv(n, 256), 256, 0, c->stream>>>(cur, rank_dev, n, k, key_hi, key_lo, exts, data, spec, uidx_by_rank.p, rev_by_rank.p,
                                                         ustart.p, words.p, uexts.p, uacc.p);
    }
    c->t_end();
    LAUNCH_CHECK(c, "emit");
    if (n_nodes) {
        finish_nodes_kernel<<<cdiv(n_nodes, 256), 256, 0, c->stream>>>(n_nodes, spec, k, ulen.p, uexts.p, uacc.p, nodes ? ucnt.p : nullptr, o_exts.p, o_data.p);
        LAUNCH_CHECK(c, "finish_nodes");
    }
    if (c->opt("DBG_DEBUG") && n_nodes) {
        unsigned long long a0 = 0; uint32_t d0 = 0, e0 = 0;
        (void)hipMemcpy(&a0, uacc.p, 8, hipMemcpyDeviceToHost); (void)hipMemcpy(&d0, o_data.p, 4, hipMemcpyDeviceToHost);
        (void)hipMemcpy(&e0, uexts.p, 4, hipMemcpyDeviceToHost);
        fprintf(stderr, "[unitig] nodes=%u spec=%d uacc[0]=%llx o_data[0]=%x uexts[0]=%x uacc.p=%p o_data.p=%p uexts.p=%p\n", n_nodes, spec, a0, d0, e0,
                (void*)uacc.p, (void*)o_data.p, (void*)uexts.p);
    }
    }   // general route
    memset(out, 0, sizeof(*out));
    out->stranded = stranded ? 1 : 0;
    out->n_nodes = n_nodes; out->n_seq_words = n_words; out->seq_len_bases = total_bases;
    if (GraphDev* sink = c->graph_sink) {
        // the rank-spanning second stage keeps the graph in HBM: the buffers change hands, nothing is copied (out carries the
        // sizes only, its arrays stay null)
        sink->words = std::move(words); sink->start = std::move(ustart); sink->length = std::move(ulen);
        sink->exts = std::move(o_exts); sink->data = std::move(o_data);
        if (c->segment_depth > 0) sink->seed = std::move(useed);   // (chain route: the pieces' seed k-mers, for cycles found one level up)
        sink->n_nodes = n_nodes; sink->n_words = n_words; sink->n_bases = total_bases; sink->stranded = stranded ? 1 : 0;
        sink->filled = true;
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        *done = true;
        return 0;
    }
    // ---- to the host BaseGraph ----
    out->seq_words = out_early.seq_words ? out_early.seq_words : (uint64_t*)ctx_halloc(c, std::max<uint64_t>(n_words, 1) * 8);
    out->start = out_early.start ? out_early.start : (uint64_t*)ctx_halloc(c, std::max<uint32_t>(n_nodes, 1) * 8ull);
    out->length = out_early.length ? out_early.length : (uint32_t*)ctx_halloc(c, std::max<uint32_t>(n_nodes, 1) * 4ull);
    out->exts = (uint8_t*)ctx_halloc(c, std::max<uint32_t>(n_nodes, 1));
    out->data = (uint32_t*)ctx_halloc(c, std::max<uint32_t>(n_nodes, 1) * 4ull);
    c->t_begin("graph_to_host", n_nodes);
    if (n_words > early_words)
        HIP_TRY(c, hipMemcpyAsync(out->seq_words + early_words, words.p + early_words, (n_words - early_words) * 8, hipMemcpyDeviceToHost, c->stream));
    if (n_nodes) {
        if (!early_nodes) {
            HIP_TRY(c, hipMemcpyAsync(out->start, ustart.p, (size_t)n_nodes * 8, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipMemcpyAsync(out->length, ulen.p, (size_t)n_nodes * 4, hipMemcpyDeviceToHost, c->stream));
        }
        HIP_TRY(c, hipMemcpyAsync(out->exts, o_exts.p, (size_t)n_nodes, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipMemcpyAsync(out->data, o_data.p, (size_t)n_nodes * 4, hipMemcpyDeviceToHost, c->stream));
    }
    c->t_end();
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    *done = true;
    return 0;
}
