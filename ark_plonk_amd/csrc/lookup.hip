// Round 2 of the prover on the device (proof_system/prover.rs:228-317): the compressed query column and the "sorted
// concatenation" of the table and query multisets, split into h_1 / h_2 -- the O(n) CPU work between the iffts of round 1
// and the iffts / commitments of f, h_1, h_2.
//
//  * zk_lookup_query_dev        prover.rs:244-279: row i of the query is the wire row compressed with zeta
//                               (MultiSet::compress = util.rs lc(): w_l + zeta w_r + zeta^2 w_o + zeta^3 w_4) where q_lookup[i] != 0
//                               and the first element of the compressed table where it is 0 (rows past q_lookup's length count as 0).
//  * zk_lookup_combine_split_dev  lookup/multiset.rs:131-176 `MultiSet::combine_split`: the reference counts t and f into an
//                               IndexMap keyed by value (insertion order = first appearance in t), fails with ElementNotIndexed
//                               when f holds a value t lacks, then walks the buckets in order writing count/2 copies to each half
//                               and the odd one alternately to evens / odds.
//    Here: an open-addressing table over the 32-byte values of t (slot = first claim by CAS, then atomicMin so that the slot names
//    the FIRST index of its value in t, whatever order the lanes arrive in), counts by atomicAdd (one atomic per distinct slot in a
//    wavefront: the reference pads every non-lookup row with one value, so one bucket can hold most of f), exclusive scans over
//    t's positions (odd-bucket parity; evens and odds offsets, packed in one 64-bit scan) and an output-parallel fill that finds its bucket by binary search in the offsets.  No sort, like the reference; results are the
//    reference's exactly because the emission order is a function of (first index in t, count) only.
#include "ctx.h"
#include "fr_io.cuh"

namespace {

constexpr uint32_t EMPTY = 0xFFFFFFFFu;

struct El {
    uint4 a, b;
};
ZK_D El ld_el(const void* base, uint64_t i) {
    const uint4* q = reinterpret_cast<const uint4*>(base) + 2 * i;
    El e;
    e.a = q[0];
    e.b = q[1];
    return e;
}
ZK_D bool el_eq(const El& x, const El& y) {
    return x.a.x == y.a.x && x.a.y == y.a.y && x.a.z == y.a.z && x.a.w == y.a.w && x.b.x == y.b.x && x.b.y == y.b.y && x.b.z == y.b.z &&
           x.b.w == y.b.w;
}
ZK_D uint32_t el_hash(const El& e) {
    uint64_t h = ((uint64_t)e.a.y << 32 | e.a.x) * 0x9E3779B97F4A7C15ull;
    h ^= ((uint64_t)e.a.w << 32 | e.a.z) * 0xC2B2AE3D27D4EB4Full;
    h ^= ((uint64_t)e.b.y << 32 | e.b.x) * 0x165667B19E3779F9ull;
    h ^= ((uint64_t)e.b.w << 32 | e.b.z) * 0xD6E8FEB86659FD93ull;
    h ^= h >> 33;
    h *= 0xFF51AFD7ED558CCDull;
    h ^= h >> 29;
    return (uint32_t)h;
}

// One atomic per DISTINCT slot among the live lanes of a wavefront (the reference pads every non-lookup row of the query and
// every padding row of the table with one value, so one slot can take most of the traffic: ~9 ns per atomic on one address,
// 12 ms per proof before this).  add != 0: cnt[slot] += lanes in the group; add == 0: slots[slot] = min(slots[slot], the group's
// smallest index) -- lane order is index order, so that is the group's first lane.
ZK_D void wave_group_update(uint32_t* arr, uint32_t s, uint32_t i, bool live, bool add) {
    uint64_t todo = __ballot(live);
    const int lane = threadIdx.x & 63;
    while (todo) {
        const int lead = __ffsll((long long)todo) - 1;
        const uint32_t s0 = __shfl(s, lead, 64);
        const uint64_t same = __ballot(live && s == s0) & todo;
        if (lane == lead) {
            if (add) atomicAdd(&arr[s0], (uint32_t)__popcll(same));
            else atomicMin(&arr[s0], i);
        }
        todo &= ~same;
    }
}

// Lanes of a wavefront that hold the same 32-byte value elect their first lane (returns its id; the own id for a value no other
// lane holds).  Without it every lane of the first few thousand wavefronts CASes the one slot of the padding value at once:
// ls_insert_table took 1.36 ms at n = 2^20 with 3/4 of the table padded, 78 % of the whole function.
ZK_D int wave_value_leader(const El& v, uint32_t h, bool live) {
    const int lane = threadIdx.x & 63;
    int leader = lane;
    uint64_t todo = __ballot(live);
    while (todo) {
        const int lead = __ffsll((long long)todo) - 1;
        const uint32_t h0 = __shfl(h, lead, 64);
        uint64_t same = __ballot(live && h == h0) & todo;
        if (same & (same - 1)) {                  // someone shares the leader's hash: compare the values (wave-uniform branch)
            bool eq = h == h0;
            eq = eq && v.a.x == __shfl(v.a.x, lead, 64);
            eq = eq && v.a.y == __shfl(v.a.y, lead, 64);
            eq = eq && v.a.z == __shfl(v.a.z, lead, 64);
            eq = eq && v.a.w == __shfl(v.a.w, lead, 64);
            eq = eq && v.b.x == __shfl(v.b.x, lead, 64);
            eq = eq && v.b.y == __shfl(v.b.y, lead, 64);
            eq = eq && v.b.z == __shfl(v.b.z, lead, 64);
            eq = eq && v.b.w == __shfl(v.b.w, lead, 64);
            same = (__ballot(live && eq) & todo) | (1ull << lead);
        }
        if ((same >> lane) & 1) leader = lead;
        todo &= ~same;
    }
    return leader;
}

// t -> table.  slots[s] = smallest index i of the value stored at s; slot_of[i] = the slot of t[i]; cnt[s] += 1 per element
__global__ void __launch_bounds__(256) ls_insert_table(const void* t, uint32_t n_t, uint32_t* slots, uint32_t mask, uint32_t* slot_of,
                                                      uint32_t* cnt) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < n_t;
    bool need_min = false;
    uint32_t s = 0, h = 0;
    El v = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};
    if (live) {
        v = ld_el(t, i);
        h = el_hash(v);
    }
    const int leader = wave_value_leader(v, h, live);
    if (live && leader == (int)(threadIdx.x & 63)) {          // one probe per distinct value of the wavefront, by its smallest index
        s = h & mask;
        for (;;) {
            uint32_t cur = slots[s];
            if (cur == EMPTY) {
                cur = atomicCAS(&slots[s], EMPTY, i);
                if (cur == EMPTY) break;
            }
            if (el_eq(ld_el(t, cur), v)) {       // any index stored here carries this slot's value
                need_min = cur > i;
                break;
            }
            s = (s + 1) & mask;
        }
    }
    s = __shfl(s, leader, 64);
    if (live) slot_of[i] = s;
    wave_group_update(slots, s, i, need_min, false);
    wave_group_update(cnt, s, i, live, true);
}

// f -> counts; err[0] = 1 when some f[i] is not in t (Error::ElementNotIndexed), err[1] = the smallest such i
__global__ void __launch_bounds__(256) ls_count_queries(const void* f, uint32_t n_f, const void* t, const uint32_t* slots, uint32_t mask,
                                                       uint32_t* cnt, uint32_t* err) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    bool live = i < n_f;
    uint32_t s = EMPTY, h = 0;
    El v = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};
    if (live) {
        v = ld_el(f, i);
        h = el_hash(v);
    }
    const int leader = wave_value_leader(v, h, live);
    if (live && leader == (int)(threadIdx.x & 63)) {
        s = h & mask;
        for (;;) {
            const uint32_t cur = slots[s];
            if (cur == EMPTY) {
                atomicOr(&err[0], 1u);
                atomicMin(&err[1], i);
                s = EMPTY;
                break;
            }
            if (el_eq(ld_el(t, cur), v)) break;
            s = (s + 1) & mask;
        }
    }
    s = __shfl(s, leader, 64);
    live = live && s != EMPTY;
    wave_group_update(cnt, s, i, live, true);
}

// position i of t owns its bucket iff it is the first index of its value; odd[i] = owner with an odd count
__global__ void ls_owner_odd(uint32_t n_t, const uint32_t* slots, const uint32_t* slot_of, const uint32_t* cnt, uint32_t* odd) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_t) return;
    const uint32_t s = slot_of[i];
    odd[i] = (slots[s] == i) ? (cnt[s] & 1u) : 0u;
}

// eo[i] = (copies to odds) << 32 | (copies to evens) for an owner, 0 otherwise.  par[i] = odd-count owners before i
// (multiset.rs:153-171: parity starts at 0 -> evens takes the first odd one)
__global__ void ls_emit_counts(uint32_t n_t, const uint32_t* slots, const uint32_t* slot_of, const uint32_t* cnt, const uint32_t* par,
                               uint64_t* eo) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_t) return;
    const uint32_t s = slot_of[i];
    uint64_t v = 0;
    if (slots[s] == i) {
        const uint32_t c = cnt[s], half = c >> 1;
        uint32_t e = half, o = half;
        if (c & 1u) {
            if (par[i] & 1u) o += 1;
            else e += 1;
        }
        v = (uint64_t)o << 32 | e;
    }
    eo[i] = v;
}

// ---- exclusive scan of n values (n <= SCAN_BLOCK^2), out has n + 1 entries (out[n] = total) --------------------------------
constexpr uint32_t SCAN_T = 1024, SCAN_PER = 4, SCAN_BLOCK = SCAN_T * SCAN_PER;

template <class T>
ZK_D T block_excl_scan(T v, T* sh, T& total) {       // exclusive scan of one value per thread over the workgroup
    const uint32_t u = threadIdx.x;
    sh[u] = v;
    __syncthreads();
    for (uint32_t d = 1; d < SCAN_T; d <<= 1) {
        T o = u >= d ? sh[u - d] : T(0);
        __syncthreads();
        sh[u] += o;
        __syncthreads();
    }
    total = sh[SCAN_T - 1];
    return sh[u] - v;
}
template <class T>
__global__ void __launch_bounds__(SCAN_T) scan_blocks(const T* in, uint32_t n, T* out, T* block_tot) {
    __shared__ T sh[SCAN_T];
    const uint32_t base = blockIdx.x * SCAN_BLOCK + threadIdx.x * SCAN_PER;
    T v[SCAN_PER], sum = 0;
#pragma unroll
    for (uint32_t k = 0; k < SCAN_PER; ++k) {
        v[k] = base + k < n ? in[base + k] : T(0);
        sum += v[k];
    }
    T tot;
    T run = block_excl_scan<T>(sum, sh, tot);
#pragma unroll
    for (uint32_t k = 0; k < SCAN_PER; ++k) {
        if (base + k < n) out[base + k] = run;
        run += v[k];
    }
    if (threadIdx.x == 0) block_tot[blockIdx.x] = tot;
}
template <class T>
__global__ void __launch_bounds__(SCAN_T) scan_totals(T* block_tot, uint32_t n_blocks, T* grand) {
    __shared__ T sh[SCAN_T];
    const uint32_t base = threadIdx.x * SCAN_PER;
    T v[SCAN_PER], sum = 0;
#pragma unroll
    for (uint32_t k = 0; k < SCAN_PER; ++k) {
        v[k] = base + k < n_blocks ? block_tot[base + k] : T(0);
        sum += v[k];
    }
    T tot;
    T run = block_excl_scan<T>(sum, sh, tot);
#pragma unroll
    for (uint32_t k = 0; k < SCAN_PER; ++k) {
        if (base + k < n_blocks) block_tot[base + k] = run;
        run += v[k];
    }
    if (threadIdx.x == 0) *grand = tot;
}
template <class T>
__global__ void __launch_bounds__(SCAN_T) scan_add(T* out, uint32_t n, const T* block_tot) {
    const uint32_t base = blockIdx.x * SCAN_BLOCK + threadIdx.x * SCAN_PER;
    const T off = block_tot[blockIdx.x];
#pragma unroll
    for (uint32_t k = 0; k < SCAN_PER; ++k)
        if (base + k < n) out[base + k] += off;
}
template <class T>
int scan_excl(const T* in, uint32_t n, T* out /* n + 1 */, T* block_tot /* >= ceil(n / SCAN_BLOCK) */, hipStream_t st) {
    const uint32_t nb = (n + SCAN_BLOCK - 1) / SCAN_BLOCK;
    if (nb > SCAN_BLOCK) return ZK_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(scan_blocks<T>, dim3(nb), dim3(SCAN_T), 0, st, in, n, out, block_tot);
    hipLaunchKernelGGL(scan_totals<T>, dim3(1), dim3(SCAN_T), 0, st, block_tot, nb, out + n);
    hipLaunchKernelGGL(scan_add<T>, dim3(nb), dim3(SCAN_T), 0, st, out, n, block_tot);
    return ZK_OK;
}

// output position p of one half takes the value of the last position i of t whose offset is <= p (offsets are
// non-decreasing; the positions after an owner that emit nothing repeat the NEXT offset, which is > p)
__global__ void ls_fill(const void* t, uint32_t n_t, const uint64_t* off /* n_t + 1 */, uint32_t hi_half, void* out) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t sh = hi_half ? 32 : 0;
    const uint32_t total = (uint32_t)(off[n_t] >> sh);
    if (p >= total) return;
    uint32_t lo = 0, hi = n_t;                 // invariant: off[lo] <= p < off[hi]
    while (hi - lo > 1) {
        const uint32_t mid = lo + (hi - lo) / 2;
        if ((uint32_t)(off[mid] >> sh) <= p) lo = mid;
        else hi = mid;
    }
    const El v = ld_el(t, lo);
    uint4* q = reinterpret_cast<uint4*>(out) + 2 * (uint64_t)p;
    q[0] = v.a;
    q[1] = v.b;
}

struct QueryArgs {
    const void* q_lookup;
    uint64_t q_len;
    const void* w[4];
    Packed zeta[3];       // zeta, zeta^2, zeta^3 in the R' form
    const void* table;
};
template <class FU>
__global__ void lookup_query(QueryArgs a, uint64_t n, void* out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    bool sel = false;
    if (i < a.q_len) {
        const El q = ld_el(a.q_lookup, i);
        sel = (q.a.x | q.a.y | q.a.z | q.a.w | q.b.x | q.b.y | q.b.z | q.b.w) != 0;
    }
    uint4* o = reinterpret_cast<uint4*>(out) + 2 * i;
    if (!sel) {                                   // prover.rs:262-264: the first row of the compressed table, then zeros
        const El d = ld_el(a.table, 0);
        o[0] = d.a;
        o[1] = d.b;
        return;
    }
    FU acc = ld_u<FU>(a.w[0], i);                 // < r
#pragma unroll
    for (int k = 0; k < 3; ++k) acc = FU::add(acc, FU::mul(ld_u<FU>(a.w[k + 1], i), unpack<FU>(a.zeta[k])));   // + 3 terms < 2r
    st_u<FU>(out, i, FU::mul(acc, FU::one()));
}

template <class C>
int query_run(zk_ctx* c, size_t n, const void* d_q, size_t q_len, const void* const* d_w, const uint64_t* zeta_mont, const void* d_table, void* d_out) {
    typedef typename C::Fr Fr;
    typedef typename C::FrU FU;
    Fr z;
    memcpy(z.v, zeta_mont, 32);
    if (Fr::reduce_once(z) != z) return ZK_ERR_BAD_ARG;
    Fr to_rp;
    FU::one().pack_words(to_rp.v);
    QueryArgs a;
    memset(&a, 0, sizeof a);
    a.q_lookup = d_q;
    a.q_len = q_len < n ? q_len : n;
    Fr pw = z;
    for (int k = 0; k < 3; ++k) {
        const Fr r = Fr::mul(pw, to_rp);
        memcpy(a.zeta[k].w, r.v, 32);
        pw = Fr::mul(pw, z);
    }
    for (int k = 0; k < 4; ++k) a.w[k] = d_w[k];
    a.table = d_table;
    ProfScope ps(c, "lookup_query");
    const int T = 256;
    hipLaunchKernelGGL(lookup_query<FU>, dim3((unsigned)((n + T - 1) / T)), dim3(T), 0, c->stream, a, (uint64_t)n, d_out);
    ZK_HIP_TRY(hipGetLastError());
    return ZK_OK;
}

}  // namespace

int lookup_query_dev(zk_ctx* c, int curve, size_t n, const void* d_q, size_t q_len, const void* const* d_w, const uint64_t* zeta_mont,
                     const void* d_table, void* d_out) {
    if (n == 0) return ZK_OK;
    if (curve == ZK_CURVE_BLS12_381) return query_run<CurveBls>(c, n, d_q, q_len, d_w, zeta_mont, d_table, d_out);
    if (curve == ZK_CURVE_BN254) return query_run<CurveBn>(c, n, d_q, q_len, d_w, zeta_mont, d_table, d_out);
    return ZK_ERR_BAD_ARG;
}

int lookup_combine_split_dev(zk_ctx* c, const void* d_t, size_t n_t, const void* d_f, size_t n_f, void* d_h1, void* d_h2, size_t* len_h1,
                             size_t* len_h2) {
    *len_h1 = *len_h2 = 0;
    if (n_t >= (1ull << 24) || n_f >= (1ull << 31)) return ZK_ERR_UNSUPPORTED;
    if (n_t == 0) return n_f ? ZK_ERR_NOT_INDEXED : ZK_OK;
    uint32_t M = 1024;
    while (M < 2 * n_t) M <<= 1;
    const uint32_t nb_scan = (uint32_t)((n_t + SCAN_BLOCK - 1) / SCAN_BLOCK);
    // scratch: slots[M] | cnt[M] | slot_of[n_t] | odd[n_t] | par[n_t + 1] | err[2] | pad | eo[n_t] | off[n_t + 1] | block totals
    size_t w32 = (size_t)2 * M + 3 * n_t + 1 + 2;
    w32 = (w32 + 1) & ~(size_t)1;
    const size_t bytes = w32 * 4 + ((size_t)2 * n_t + 1 + nb_scan + 1) * 8;
    int rc = c->io_b.ensure(bytes);
    if (rc) return rc;
    uint32_t* slots = (uint32_t*)c->io_b.p;
    uint32_t* cnt = slots + M;
    uint32_t* slot_of = cnt + M;
    uint32_t* odd = slot_of + n_t;
    uint32_t* par = odd + n_t;
    uint32_t* err = par + n_t + 1;
    uint64_t* eo = (uint64_t*)((uint32_t*)c->io_b.p + w32);
    uint64_t* off = eo + n_t;
    uint64_t* btot = off + n_t + 1;
    hipStream_t st = c->stream;
    const int T = 256;
    const unsigned gt = (unsigned)((n_t + T - 1) / T);
    {
        ProfScope ps(c, "lookup_combine_split");
        ZK_HIP_TRY(hipMemsetAsync(slots, 0xFF, (size_t)M * 4, st));
        ZK_HIP_TRY(hipMemsetAsync(cnt, 0, (size_t)M * 4, st));
        const uint32_t err0[2] = {0u, 0xFFFFFFFFu};
        ZK_HIP_TRY(hipMemcpyAsync(err, err0, 8, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(ls_insert_table, dim3(gt), dim3(T), 0, st, d_t, (uint32_t)n_t, slots, M - 1, slot_of, cnt);
        if (n_f)
            hipLaunchKernelGGL(ls_count_queries, dim3((unsigned)((n_f + T - 1) / T)), dim3(T), 0, st, d_f, (uint32_t)n_f, d_t, (const uint32_t*)slots,
                               M - 1, cnt, err);
        hipLaunchKernelGGL(ls_owner_odd, dim3(gt), dim3(T), 0, st, (uint32_t)n_t, (const uint32_t*)slots, (const uint32_t*)slot_of,
                           (const uint32_t*)cnt, odd);
        if ((rc = scan_excl<uint32_t>(odd, (uint32_t)n_t, par, (uint32_t*)btot, st))) return rc;
        hipLaunchKernelGGL(ls_emit_counts, dim3(gt), dim3(T), 0, st, (uint32_t)n_t, (const uint32_t*)slots, (const uint32_t*)slot_of,
                           (const uint32_t*)cnt, (const uint32_t*)par, eo);
        if ((rc = scan_excl<uint64_t>(eo, (uint32_t)n_t, off, btot, st))) return rc;
        ZK_HIP_TRY(hipGetLastError());
    }
    // the totals decide the launch sizes of the fill (and ElementNotIndexed must surface before anything is written)
    struct {
        uint32_t err[2];
        uint64_t tot;
    } h;
    ZK_HIP_TRY(hipMemcpyAsync(h.err, err, 8, hipMemcpyDeviceToHost, st));
    ZK_HIP_TRY(hipMemcpyAsync(&h.tot, off + n_t, 8, hipMemcpyDeviceToHost, st));
    ZK_HIP_TRY(hipStreamSynchronize(st));
    c->d2h_bytes += 16;
    if (h.err[0]) return ZK_ERR_NOT_INDEXED;
    const uint32_t ne = (uint32_t)h.tot, no = (uint32_t)(h.tot >> 32);
    if ((uint64_t)ne + no != (uint64_t)n_t + n_f) return ZK_ERR_HIP;       // cannot happen: every element was counted once
    {
        ProfScope ps(c, "lookup_combine_split");
        if (ne) hipLaunchKernelGGL(ls_fill, dim3((ne + T - 1) / T), dim3(T), 0, st, d_t, (uint32_t)n_t, (const uint64_t*)off, 0u, d_h1);
        if (no) hipLaunchKernelGGL(ls_fill, dim3((no + T - 1) / T), dim3(T), 0, st, d_t, (uint32_t)n_t, (const uint64_t*)off, 1u, d_h2);
        ZK_HIP_TRY(hipGetLastError());
    }
    *len_h1 = ne;
    *len_h2 = no;
    return ZK_OK;
}
