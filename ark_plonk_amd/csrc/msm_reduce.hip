// MSM unit 3 of 4 (msm_common.cuh): from the accumulation's buckets and chunk-edge partials to window sums.
//   msm_combine*      joins the chunk-edge partials of each bucket: lane(s) / wavefront (shuffle tree) / workgroup per bucket by size class
//   msm_seg_reduce    segmented running-sum reduction (sum_j j * B_j), level 1
//   msm_win_finish    LDS suffix-scan + tree reduction per (virtual) window -> window sums: arkworks layout straight into pinned host
//                     memory, or -- the multi-GPU exchange -- the internal form into a device buffer (the collective's send buffer)
//   *_q               quad-cooperative forms (ecq.cuh): four lanes per dependent chain where the launch is latency-shaped
//   g1_sum_winsums_q  the ranks' window sums added element-wise after the all-gather
// host: the few window sums are combined (table path: sum of 64 virtual windows; per-window path: Horner with W * c doublings) and
// normalised to affine in msm_plan.hip.
#include "msm_common.cuh"
#include "ecq.cuh"

namespace {


ZK_D uint64_t partial_slot(uint32_t t, uint32_t ta, uint32_t s, uint32_t L) {
    return (t == ta && (s % L) != 0) ? 2ull * t + 1 : 2ull * t;
}

// wave reduction: lane 0 ends with the sum of all 64 lanes (order irrelevant: abelian group)
template <class F>
ZK_D XYZZu<F> wave_sum(XYZZu<F> acc) {
#pragma unroll 1
    for (int d = 32; d >= 1; d >>= 1) {
        XYZZu<F> o;
#pragma unroll
        for (int i = 0; i < F::NL; ++i) {
            o.x.v[i] = __shfl_down(acc.x.v[i], d, 64);
            o.y.v[i] = __shfl_down(acc.y.v[i], d, 64);
            o.zz.v[i] = __shfl_down(acc.zz.v[i], d, 64);
            o.zzz.v[i] = __shfl_down(acc.zzz.v[i], d, 64);
        }
        acc = XYZZu<F>::add(acc, o);
    }
    return acc;
}

// One lane per bucket: a bucket whose entries span p >= 2 chunks has exactly p partials at slots
// known from the offsets (see msm_accumulate).  Small p is summed here; larger p is queued.
// queues: q[0] = medium count, q[1] = large count, q[2 ..] medium ids (grow up), q[.. 2+nb) large ids (grow down)
// COMBINE_SG lanes cooperate on one small bucket: 4 shortens the dependent chain when the launch is
// latency-bound (1-2 jobs: 0.25 -> 0.19 ms); with more jobs the launch is throughput-bound and the idle
// lanes of the shuffle tree cost more than they save (7 jobs: 1.0 ms at 4 lanes), so 1 is used there.
template <class F, uint32_t COMBINE_SG>
__global__ void __launch_bounds__(128) msm_combine(RJobs jobs, uint32_t nb) {
    const void* part_pt = jobs.part_pt[blockIdx.y];
    const uint32_t* offsets = jobs.offsets[blockIdx.y];
    void* buckets = jobs.buckets[blockIdx.y];
    uint32_t* q = jobs.q[blockIdx.y];
    const uint32_t L = chunk_len(offsets[jobs.nbk[blockIdx.y]], jobs.lanes[blockIdx.y], jobs.L[blockIdx.y]);
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t b = id / COMBINE_SG, sub = id % COMBINE_SG;
    // every lane stays to the end: the sub-group sums below are wave shuffles
    XYZZu<F> acc = XYZZu<F>::infinity();
    bool mine = false;
    if (b < nb) {
        const uint32_t s = offsets[b], e = offsets[b + 1];
        if (e != s) {
            const uint32_t ta = s / L, tb = (e - 1) / L;
            const uint32_t p = tb - ta + 1;   // p == 1: whole bucket inside one chunk, already complete
            if (p > COMBINE_MEDIUM) {
                if (sub == 0) q[2 + nb - 1 - atomicAdd(&q[1], 1u)] = b;
            } else if (p > COMBINE_SMALL) {
                if (sub == 0) q[2 + atomicAdd(&q[0], 1u)] = b;
            } else if (p > 1) {
                mine = true;
#pragma unroll 1
                for (uint32_t t = ta + sub; t <= tb; t += COMBINE_SG)
                    acc = XYZZu<F>::add(acc, ld_xyzz<F>(part_pt, partial_slot(t, ta, s, L)));
            }
        }
    }
#pragma unroll 1
    for (int d = COMBINE_SG / 2; d >= 1; d >>= 1) {
        XYZZu<F> o;
#pragma unroll
        for (int i = 0; i < F::NL; ++i) {
            o.x.v[i] = __shfl_down(acc.x.v[i], d, 64);
            o.y.v[i] = __shfl_down(acc.y.v[i], d, 64);
            o.zz.v[i] = __shfl_down(acc.zz.v[i], d, 64);
            o.zzz.v[i] = __shfl_down(acc.zzz.v[i], d, 64);
        }
        acc = XYZZu<F>::add(acc, o);
    }
    if (mine && sub == 0) st_xyzz<F>(buckets, b, acc);
}

// medium buckets: one wavefront per bucket, lanes stride over its partials, shuffle tree
template <class F>
__global__ void __launch_bounds__(256) msm_combine_wave(RJobs jobs) {
    const void* part_pt = jobs.part_pt[blockIdx.y];
    const uint32_t* offsets = jobs.offsets[blockIdx.y];
    void* buckets = jobs.buckets[blockIdx.y];
    const uint32_t* q = jobs.q[blockIdx.y];
    const uint32_t L = chunk_len(offsets[jobs.nbk[blockIdx.y]], jobs.lanes[blockIdx.y], jobs.L[blockIdx.y]);
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t n_waves = (gridDim.x * blockDim.x) >> 6;
    const uint32_t nm = q[0];
    for (uint32_t h = wave; h < nm; h += n_waves) {
        const uint32_t b = q[2 + h];
        const uint32_t s = offsets[b], e = offsets[b + 1];
        const uint32_t ta = s / L, tb = (e - 1) / L;
        XYZZu<F> acc = XYZZu<F>::infinity();
        for (uint32_t t = ta + lane; t <= tb; t += 64) acc = XYZZu<F>::add(acc, ld_xyzz<F>(part_pt, partial_slot(t, ta, s, L)));
        acc = wave_sum<F>(acc);
        if (lane == 0) st_xyzz<F>(buckets, b, acc);
    }
}

// large buckets (heavily skewed scalars): one 256-lane workgroup per bucket
template <class F>
__global__ void __launch_bounds__(256) msm_combine_block(RJobs jobs, uint32_t nb) {
    extern __shared__ uint4 sh[];
    const void* part_pt = jobs.part_pt[blockIdx.y];
    const uint32_t* offsets = jobs.offsets[blockIdx.y];
    void* buckets = jobs.buckets[blockIdx.y];
    const uint32_t* q = jobs.q[blockIdx.y];
    const uint32_t L = chunk_len(offsets[jobs.nbk[blockIdx.y]], jobs.lanes[blockIdx.y], jobs.L[blockIdx.y]);
    const uint32_t u = threadIdx.x;
    const uint32_t nl = q[1];
    for (uint32_t h = blockIdx.x; h < nl; h += gridDim.x) {
        const uint32_t b = q[2 + nb - 1 - h];
        const uint32_t s = offsets[b], e = offsets[b + 1];
        const uint32_t ta = s / L, tb = (e - 1) / L;
        XYZZu<F> acc = XYZZu<F>::infinity();
        for (uint32_t t = ta + u; t <= tb; t += 256) acc = XYZZu<F>::add(acc, ld_xyzz<F>(part_pt, partial_slot(t, ta, s, L)));
        acc = wave_sum<F>(acc);
        __syncthreads();
        if ((u & 63) == 0) st_xyzz<F>(sh, u >> 6, acc);
        __syncthreads();
        if (u == 0) {
            for (uint32_t w = 1; w < 4; ++w) acc = XYZZu<F>::add(acc, ld_xyzz<F>(sh, w));
            st_xyzz<F>(buckets, b, acc);
        }
    }
}

// level 1 of the per-window reduction: segment s of window w covers buckets [s*G, (s+1)*G)
//   run = sum B_i ; acc = sum (i+1) * B_i   (i local index)
template <class F>
__global__ void __launch_bounds__(128) msm_seg_reduce(RJobs jobs, MsmGeom g) {
    const void* buckets = jobs.buckets[blockIdx.y];
    const uint32_t* offsets = jobs.offsets[blockIdx.y];
    void* seg_run = jobs.seg_run[blockIdx.y];
    void* seg_acc = jobs.seg_acc[blockIdx.y];
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= g.W * g.ns) return;
    const uint32_t w = id / g.ns, s = id % g.ns;
    const uint32_t G = 1u << g.logG;
    XYZZu<F> run = XYZZu<F>::infinity(), acc = XYZZu<F>::infinity();
    for (int i = (int)G - 1; i >= 0; --i) {
        const uint32_t bi = w * g.B + s * G + (uint32_t)i;
        if (!offsets || offsets[bi + 1] != offsets[bi]) run = XYZZu<F>::add(run, ld_xyzz<F>(buckets, bi));
        acc = XYZZu<F>::add(acc, run);
    }
    st_xyzz<F>(seg_run, id, run);
    st_xyzz<F>(seg_acc, id, acc);
}

// level 2: one 256-lane workgroup per window.
//   S_w = sum_s acc_s + G * sum_s s * run_s
// The window sum leaves the device in the arkworks layout (XYZZ of 4 x SAT words, canonical).
template <class F>
// tot_out (optional): sum of all buckets of the window, same layout (used when one real window is
// reduced as several "virtual" windows to shorten the dependent-addition chain).
__global__ void __launch_bounds__(256) msm_win_finish(RJobs jobs, MsmGeom g) {
    extern __shared__ uint4 sh[];
    const void* seg_run = jobs.seg_run[blockIdx.y];
    const void* seg_acc = jobs.seg_acc[blockIdx.y];
    uint32_t* win_out = jobs.win_s[blockIdx.y];
    uint32_t* tot_out = jobs.win_t[blockIdx.y];
    const uint32_t w = blockIdx.x, u = threadIdx.x;
    const uint32_t q = 1u << g.logq;
    typedef XYZZu<F> P;
    P A = P::infinity(), V = P::infinity(), tsum = P::infinity(), R = P::infinity();
    for (int v = (int)q - 1; v >= 0; --v) {
        const uint32_t s = u * q + (uint32_t)v;
        P x = P::infinity();
        if (s < g.ns) {
            x = ld_xyzz<F>(seg_run, (uint64_t)w * g.ns + s);
            A = P::add(A, ld_xyzz<F>(seg_acc, (uint64_t)w * g.ns + s));
        }
        if (v >= 1) {
            tsum = P::add(tsum, x);
            V = P::add(V, tsum);
        } else {
            R = P::add(tsum, x);
        }
    }
    // Y = A + G * V
    for (uint32_t k = 0; k < g.logG; ++k) V = P::dbl(V);
    P Y = P::add(A, V);
    // suffix sums Q_u = sum_{u' >= u} R_u'  (Hillis-Steele in LDS)
    st_xyzz<F>(sh, u, R);
    for (uint32_t d = 1; d < 256; d <<= 1) {
        __syncthreads();
        P o = P::infinity();
        if (u + d < 256) o = ld_xyzz<F>(sh, u + d);
        __syncthreads();
        R = P::add(R, o);
        st_xyzz<F>(sh, u, R);
    }
    if (tot_out && u == 0) {
        uint32_t* o = tot_out + (size_t)w * 4 * F::SAT;
        if (R.is_inf()) {
            for (int i = 0; i < 4 * F::SAT; ++i) o[i] = 0;
        } else {
            R.x.to_sat(o);
            R.y.to_sat(o + F::SAT);
            R.zz.to_sat(o + 2 * F::SAT);
            R.zzz.to_sat(o + 3 * F::SAT);
        }
    }
    // Z = Y + (G*q) * Q_u   (u >= 1)
    P Z = Y;
    if (u >= 1) {
        P Qm = R;
        for (uint32_t k = 0; k < g.logG + g.logq; ++k) Qm = P::dbl(Qm);
        Z = P::add(Z, Qm);
    }
    __syncthreads();
    st_xyzz<F>(sh, u, Z);
    for (uint32_t d = 128; d >= 1; d >>= 1) {
        __syncthreads();
        if (u < d) {
            Z = P::add(Z, ld_xyzz<F>(sh, u + d));
            st_xyzz<F>(sh, u, Z);
        }
    }
    if (u == 0) {
        uint32_t* o = win_out + (size_t)w * 4 * F::SAT;
        if (Z.is_inf()) {
            for (int i = 0; i < 4 * F::SAT; ++i) o[i] = 0;
        } else {
            Z.x.to_sat(o);
            Z.y.to_sat(o + F::SAT);
            Z.zz.to_sat(o + 2 * F::SAT);
            Z.zzz.to_sat(o + 3 * F::SAT);
        }
    }
}

// ---- quad-cooperative forms of the three reduction kernels (ecq.cuh): four lanes per dependent chain,
// an addition in 4.5 product-times instead of 13.5.  Same inputs, outputs and arithmetic results.
template <class F>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) msm_combine_q(RJobs jobs, uint32_t nb) {
    const void* part_pt = jobs.part_pt[blockIdx.y];
    const uint32_t* offsets = jobs.offsets[blockIdx.y];
    void* buckets = jobs.buckets[blockIdx.y];
    uint32_t* q = jobs.q[blockIdx.y];
    const uint32_t L = chunk_len(offsets[jobs.nbk[blockIdx.y]], jobs.lanes[blockIdx.y], jobs.L[blockIdx.y]);
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t b = id >> 2, role = id & 3;
    uint32_t s = 0, ta = 0, np = 0;     // np = partials this quad sums itself (0: nothing to do)
    if (b < nb) {
        s = offsets[b];
        const uint32_t e = offsets[b + 1];
        if (e != s) {
            ta = s / L;
            const uint32_t p = (e - 1) / L - ta + 1;
            if (p > COMBINE_MEDIUM) {
                if (role == 0) q[2 + nb - 1 - atomicAdd(&q[1], 1u)] = b;
            } else if (p > COMBINE_SMALL) {
                if (role == 0) q[2 + atomicAdd(&q[0], 1u)] = b;
            } else if (p > 1) {
                np = p;
            }
        }
    }
    // the quads of a wavefront run in lock step to the longest bucket among them; shorter ones add infinity
    uint32_t steps = np;
#pragma unroll
    for (int d = 32; d >= 4; d >>= 1) {
        const uint32_t o = __shfl_xor(steps, d, 64);
        steps = o > steps ? o : steps;
    }
    F acc = F::zero();
#pragma unroll 1
    for (uint32_t k = 0; k < steps; ++k) {
        F v = F::zero();
        if (k < np) v = ld_coord<F>(part_pt, partial_slot(ta + k, ta, s, L), role);
        acc = qadd<F>(acc, v, role);
    }
    if (np) st_coord<F>(buckets, b, role, acc);
}

template <class F>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) msm_seg_reduce_q(RJobs jobs, MsmGeom g) {
    const void* buckets = jobs.buckets[blockIdx.y];
    const uint32_t* offsets = jobs.offsets[blockIdx.y];
    void* seg_run = jobs.seg_run[blockIdx.y];
    void* seg_acc = jobs.seg_acc[blockIdx.y];
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t id = tid >> 2, role = tid & 3;
    const bool live = id < g.W * g.ns;              // no early exit: wave shuffles inside qadd
    const uint32_t w = live ? id / g.ns : 0, sg = live ? id % g.ns : 0;
    const uint32_t G = 1u << g.logG;
    F run = F::zero(), acc = F::zero();
    // the bucket of the NEXT step (its two offsets, then its coordinate: two dependent round trips) is requested before the two
    // additions of the current one; at one to two wavefronts per SIMD nothing else would hide them
    auto bucket = [&](int i) -> F {
        const uint32_t bi = w * g.B + sg * G + (uint32_t)i;
        F v = F::zero();
        if (live && (!offsets || offsets[bi + 1] != offsets[bi])) v = ld_coord<F>(buckets, bi, role);
        return v;
    };
    F nv = bucket((int)G - 1);
#pragma unroll 1
    for (int i = (int)G - 1; i >= 0; --i) {
        const F v = nv;
        if (i > 0) nv = bucket(i - 1);
        run = qadd<F>(run, v, role);
        acc = qadd<F>(acc, run, role);
    }
    if (live) {
        st_coord<F>(seg_run, id, role, run);
        st_coord<F>(seg_acc, id, role, acc);
    }
}

// msm_win_finish for logq == 0 (one segment per chain): 4 * ns lanes per workgroup, ns <= 256
// raw != 0: the two sums stay on the device in the internal point form (the "buckets" of the next reduction level);
// raw == 0: arkworks layout for the host, as msm_win_finish.
// MAXT = lanes per workgroup the instance is built for.  512 (up to 128 chains: every launch of the window-table path) leaves the
// register file to two wavefronts per SIMD and nothing spills; built for 1024 lanes -- four wavefronts per SIMD, 128 registers --
// the same code spilled 162 registers to 332 bytes of scratch per lane (the form every launch used before round 4; only the
// per-window path's 256-chain geometry still needs it).
template <class F, int MAXT>
__global__ void __launch_bounds__(MAXT) __attribute__((amdgpu_waves_per_eu(MAXT / 256, MAXT / 256))) msm_win_finish_q(RJobs jobs, MsmGeom g, uint32_t raw) {
    extern __shared__ uint4 sh[];
    const void* seg_run = jobs.seg_run[blockIdx.y];
    const void* seg_acc = jobs.seg_acc[blockIdx.y];
    uint32_t* win_out = jobs.win_s[blockIdx.y];
    uint32_t* tot_out = jobs.win_t[blockIdx.y];
    const uint32_t w = blockIdx.x, u = threadIdx.x >> 2, role = threadIdx.x & 3;
    const uint32_t T = blockDim.x >> 2;             // chains = power of two >= ns
    F R = F::zero(), Y = F::zero();
    if (u < g.ns) {
        R = ld_coord<F>(seg_run, (uint64_t)w * g.ns + u, role);
        Y = ld_coord<F>(seg_acc, (uint64_t)w * g.ns + u, role);
    }
    // suffix sums Q_u = sum_{u' >= u} R_u'  (Hillis-Steele in LDS)
    st_coord<F>(sh, u, role, R);
    for (uint32_t d = 1; d < T; d <<= 1) {
        __syncthreads();
        F o = F::zero();
        if (u + d < T) o = ld_coord<F>(sh, u + d, role);
        __syncthreads();
        R = qadd<F>(R, o, role);
        st_coord<F>(sh, u, role, R);
    }
    const bool r_inf = quad_is_inf(R, role);
    if (tot_out && u == 0) {
        if (raw) {
            st_coord<F>(tot_out, w, role, r_inf ? F::zero() : R);
        } else {
            uint32_t* o = tot_out + (size_t)w * 4 * F::SAT + role * F::SAT;
            if (r_inf) {
                for (int i = 0; i < F::SAT; ++i) o[i] = 0;
            } else {
                R.to_sat(o);
            }
        }
    }
    // Z = Y + G * Q_u   (u >= 1)
    F Qm = R;
    for (uint32_t k = 0; k < g.logG; ++k) Qm = qdbl<F>(Qm, role);
    F Z = qadd<F>(Y, u >= 1 ? Qm : F::zero(), role);
    __syncthreads();
    st_coord<F>(sh, u, role, Z);
    for (uint32_t d = T / 2; d >= 1; d >>= 1) {
        __syncthreads();
        if (u < d) {                                 // quad-uniform: all four lanes of a chain agree
            Z = qadd<F>(Z, ld_coord<F>(sh, u + d, role), role);
            st_coord<F>(sh, u, role, Z);
        }
    }
    const bool z_inf = quad_is_inf(Z, role);
    if (u == 0) {
        if (raw) {
            st_coord<F>(win_out, w, role, z_inf ? F::zero() : Z);
        } else {
            uint32_t* o = win_out + (size_t)w * 4 * F::SAT + role * F::SAT;
            if (z_inf) {
                for (int i = 0; i < F::SAT; ++i) o[i] = 0;
            } else {
                Z.to_sat(o);
            }
        }
    }
}

// Work-efficient level of the wide reduction: a node (run, acc) stands for m = 2^logm consecutive buckets,
//   run = sum B_i,  acc = sum (i + 1) B_i   (i local to the node);
// K = 2^logk neighbouring nodes make one node of K*m buckets:  run' = sum_j run_j,  acc' = sum_j acc_j + m * sum_j j * run_j
// -- 3 additions per child instead of the log2(chains) of the Hillis-Steele scan in msm_win_finish_q, which is what made
// 2^17 level-1 nodes per job (window tables with c = 20) cost more than the level below them.  One quad per output node.
// in: jobs.seg_run / seg_acc (n_out * K nodes);  out: jobs.win_s (run') / jobs.win_t (acc'), internal point form.
template <class F>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) msm_node_reduce_q(RJobs jobs, uint32_t n_out, uint32_t logk, uint32_t logm) {
    const void* in_run = jobs.seg_run[blockIdx.y];
    const void* in_acc = jobs.seg_acc[blockIdx.y];
    void* out_run = jobs.win_s[blockIdx.y];
    void* out_acc = jobs.win_t[blockIdx.y];
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t id = tid >> 2, role = tid & 3;
    const bool live = id < n_out;                  // no early exit: wave shuffles inside qadd
    const uint32_t K = 1u << logk;
    const uint64_t first = (uint64_t)(live ? id : 0) * K;
    F run = F::zero(), wsum = F::zero(), asum = F::zero();
#pragma unroll 1
    for (int j = (int)K - 1; j >= 1; --j) {
        run = qadd<F>(run, live ? ld_coord<F>(in_run, first + (uint32_t)j, role) : F::zero(), role);
        wsum = qadd<F>(wsum, run, role);           // after the loop: sum_j j * run_j
        asum = qadd<F>(asum, live ? ld_coord<F>(in_acc, first + (uint32_t)j, role) : F::zero(), role);
    }
    run = qadd<F>(run, live ? ld_coord<F>(in_run, first, role) : F::zero(), role);
    asum = qadd<F>(asum, live ? ld_coord<F>(in_acc, first, role) : F::zero(), role);
    for (uint32_t t = 0; t < logm; ++t) wsum = qdbl<F>(wsum, role);
    asum = qadd<F>(asum, wsum, role);
    if (live) {
        st_coord<F>(out_run, id, role, run);
        st_coord<F>(out_acc, id, role, asum);
    }
}

// arkworks-layout affine (x||y Montgomery words) -> internal points.  Accepted encodings of the point at infinity:
// the flag, x = y = 0, and GroupAffine::zero() = (0, 1) (Montgomery one) -- what this library itself emits for an
// infinite result and what an arkworks caller holds; (0, 1) lies on neither supported curve (b = 4 / b = 3).
// element-wise sum over the ranks of every job's 2 VW virtual-window sums (ranks x n_jobs x 2 VW points as the all-gather leaves
// them) -> n_jobs x 2 VW points in the arkworks layout in pinned host memory, where the single-GPU path's last reduction kernel
// puts them.  Q = 2^logq quads share one sum: quad j adds the ranks j, j + Q, ... (ranks / Q - 1 dependent additions), an LDS tree
// adds the Q partial sums (logq more): 3 dependent additions for 8 ranks instead of 7 -- the launch is latency-shaped (2 VW n_jobs
// points, at most a few thousand quads), so the chain is what it costs.
template <class F>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) g1_sum_winsums_q(const void* all, uint32_t ranks, uint32_t n_pts /* n_jobs * 2 VW */,
                                                                                                     uint32_t logq, uint32_t* out_sat) {
    extern __shared__ uint4 sh[];
    const uint32_t Q = 1u << logq;
    const uint32_t quad = threadIdx.x >> 2, role = threadIdx.x & 3;
    const uint32_t per_block = (blockDim.x >> 2) >> logq;                      // points per workgroup
    const uint32_t k = blockIdx.x * per_block + (quad >> logq), j = quad & (Q - 1);
    const bool live = k < n_pts;                   // no early exit: wave shuffles inside qadd, barriers below
    F acc = F::zero();
#pragma unroll 1
    for (uint32_t r = j; r < ranks; r += Q) acc = qadd<F>(acc, live ? ld_coord<F>(all, (uint64_t)r * n_pts + k, role) : F::zero(), role);
    for (uint32_t d = Q >> 1; d >= 1; d >>= 1) {
        st_coord<F>(sh, quad, role, acc);
        __syncthreads();
        const F o = j < d ? ld_coord<F>(sh, quad + d, role) : F::zero();
        __syncthreads();
        acc = qadd<F>(acc, o, role);               // quads with j >= d add the point at infinity: uniform control flow
    }
    const bool inf = quad_is_inf(acc, role);
    if (live && j == 0) {
        uint32_t* o = out_sat + (size_t)k * 4 * F::SAT + role * F::SAT;
        if (inf) {
            for (int i = 0; i < F::SAT; ++i) o[i] = 0;
        } else {
            acc.to_sat(o);
        }
    }
}

// ---------------------------------------------------------------------------------------- host side
// msm_win_finish_q with `chains` (a power of two <= 256) chains of four lanes per workgroup
template <class F>
int launch_win_finish_q(dim3 grid, uint32_t chains, hipStream_t st, const RJobs& jobs, const MsmGeom& g, uint32_t raw) {
    constexpr size_t PT = (size_t)4 * Store<F>::WORDS * 4;
    const size_t shmem = (size_t)chains * PT;
    if (chains <= 128) {
        if (shmem > 48 * 1024)
            ZK_HIP_TRY(hipFuncSetAttribute((const void*)msm_win_finish_q<F, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
        hipLaunchKernelGGL((msm_win_finish_q<F, 512>), grid, dim3(4 * chains), shmem, st, jobs, g, raw);
    } else if (chains <= 256) {
        if (shmem > 48 * 1024)
            ZK_HIP_TRY(hipFuncSetAttribute((const void*)msm_win_finish_q<F, 1024>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
        hipLaunchKernelGGL((msm_win_finish_q<F, 1024>), grid, dim3(4 * chains), shmem, st, jobs, g, raw);
    } else {
        return ZK_ERR_UNSUPPORTED;
    }
    return ZK_OK;
}

// combine + segmented reduction of n_jobs MSMs that share the geometry (nb buckets, reduction geometry gr)
template <class F>
int queue_reduce(zk_ctx* c, const RJobs& jobs, uint32_t n_jobs, uint32_t nb, const MsmGeom& gr, hipStream_t st, bool queues_cleared = false,
                 uint32_t raw = 0 /* 1: the window sums stay in the internal point form (device buffers), quad geometry only */) {
    constexpr size_t PT = (size_t)4 * Store<F>::WORDS * 4;
    ProfScope ps(c, "msm_reduce", st);
    const int T = 128;
    int rc;
    if (!queues_cleared)
        for (uint32_t k = 0; k < n_jobs; ++k) ZK_HIP_TRY(hipMemsetAsync(jobs.q[k], 0, 8, st));
    // quad-cooperative kernels where the geometry allows (one segment per chain, <= 256 chains per window)
    const bool quad = gr.logq == 0 && gr.ns <= 256;
    if (n_jobs <= 2) {
        if (quad) {
            unsigned blocks = (unsigned)(((uint64_t)nb * 4 + 255) / 256);
            hipLaunchKernelGGL(msm_combine_q<F>, dim3(blocks, n_jobs), dim3(256), 0, st, jobs, nb);
        } else {
            unsigned blocks = (unsigned)(((uint64_t)nb * 4 + T - 1) / T);
            hipLaunchKernelGGL((msm_combine<F, 4>), dim3(blocks, n_jobs), dim3(T), 0, st, jobs, nb);
        }
    } else {
        // lanes per small bucket when the launch has many jobs.  Option "combine_sg": tuning hook (profiles/r03/r03_notes.md)
        const int sg = c->tune.combine_sg ? c->tune.combine_sg : 1;
        if (sg == 4) {
            unsigned blocks = (unsigned)(((uint64_t)nb * 4 + T - 1) / T);
            hipLaunchKernelGGL((msm_combine<F, 4>), dim3(blocks, n_jobs), dim3(T), 0, st, jobs, nb);
        } else if (sg == 2) {
            unsigned blocks = (unsigned)(((uint64_t)nb * 2 + T - 1) / T);
            hipLaunchKernelGGL((msm_combine<F, 2>), dim3(blocks, n_jobs), dim3(T), 0, st, jobs, nb);
        } else {
            unsigned blocks = (unsigned)(((uint64_t)nb + T - 1) / T);
            hipLaunchKernelGGL((msm_combine<F, 1>), dim3(blocks, n_jobs), dim3(T), 0, st, jobs, nb);
        }
    }
    hipLaunchKernelGGL(msm_combine_wave<F>, dim3(256, n_jobs), dim3(256), 0, st, jobs);
    hipLaunchKernelGGL(msm_combine_block<F>, dim3(64, n_jobs), dim3(256), 4 * PT, st, jobs, nb);
    if (quad) {
        unsigned sblocks = (unsigned)(((uint64_t)gr.W * gr.ns * 4 + 255) / 256);
        hipLaunchKernelGGL(msm_seg_reduce_q<F>, dim3(sblocks, n_jobs), dim3(256), 0, st, jobs, gr);
        uint32_t chains = 1;
        while (chains < gr.ns) chains <<= 1;
        size_t shmem = (size_t)chains * PT;
        if ((rc = launch_win_finish_q<F>(dim3(gr.W, n_jobs), chains, st, jobs, gr, raw))) return rc;
    } else {
        if (raw) return ZK_ERR_UNSUPPORTED;
        unsigned sblocks = (gr.W * gr.ns + T - 1) / T;
        hipLaunchKernelGGL(msm_seg_reduce<F>, dim3(sblocks, n_jobs), dim3(T), 0, st, jobs, gr);
        size_t shmem = 256 * PT;
        if (shmem > 48 * 1024)
            ZK_HIP_TRY(hipFuncSetAttribute((const void*)msm_win_finish<F>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
        hipLaunchKernelGGL(msm_win_finish<F>, dim3(gr.W, n_jobs), dim3(256), shmem, st, jobs, gr);
    }
    ZK_HIP_TRY(hipGetLastError());
    return ZK_OK;
}

// Reduction for window tables with c > 16 (nb = 2^(c-1) >= 2^16 shared buckets), all on the device:
//   level 1  msm_seg_reduce      one LANE per node of 4 buckets: with >= 16 Ki nodes per job the launch is throughput-bound,
//                                where the single-lane group law costs 13.5 product-times per addition against the quad form's 18
//   level 2  msm_node_reduce_q   4 nodes -> one node of 16 buckets, 3 additions per child (quads)
//   level 3  msm_win_finish_q    VW = nb / 2048 virtual windows of 128 nodes -> S_v = sum_l (l+1) B_{v,l},  T_v = sum_l B_{v,l}, kept on
//                                the device in the internal point form
//   level 4  msm_seg_reduce_q + msm_win_finish_q over the two arrays S and T of every job (2 n_jobs "jobs", every element present):
//                                sum_v S_v,  K = sum_v (v+1) T_v,  sum_v T_v  -> pinned host memory, arkworks layout
//   host     total = sum_v S_v + 2048 * (K - sum_v T_v)      (bucket j = 2048 v + l has weight j + 1)
template <class F>
int queue_reduce_wide(zk_ctx* c, const RJobs& jobs, uint32_t n_jobs, uint32_t nb, void* const* d_vw, void* const* d_seg3, void* const* d_seg2,
                      char* h_out, size_t h_stride, hipStream_t st) {
    constexpr size_t PT = (size_t)4 * Store<F>::WORDS * 4;
    ProfScope ps(c, "msm_reduce", st);
    int rc;
    const uint32_t VW = nb / WIDE_VB;
    if (VW == 0 || VW > 2048) return ZK_ERR_UNSUPPORTED;
    {   // chunk-edge partials -> buckets (queues cleared by the job's sort)
        const int T = 128;
        if (n_jobs <= 2) {
            unsigned blocks = (unsigned)(((uint64_t)nb * 4 + 255) / 256);
            hipLaunchKernelGGL(msm_combine_q<F>, dim3(blocks, n_jobs), dim3(256), 0, st, jobs, nb);
        } else {
            unsigned blocks = (unsigned)(((uint64_t)nb + T - 1) / T);
            hipLaunchKernelGGL((msm_combine<F, 1>), dim3(blocks, n_jobs), dim3(T), 0, st, jobs, nb);
        }
        hipLaunchKernelGGL(msm_combine_wave<F>, dim3(256, n_jobs), dim3(256), 0, st, jobs);
        hipLaunchKernelGGL(msm_combine_block<F>, dim3(64, n_jobs), dim3(256), 4 * PT, st, jobs, nb);
    }
    const uint32_t n1 = nb >> WIDE_LOGG1, n2 = n1 >> WIDE_LOGK2;
    {   // level 1: flat over all buckets (one "window" of nb buckets, nodes of 4)
        MsmGeom g1;
        memset(&g1, 0, sizeof g1);
        g1.W = 1;
        g1.B = nb;
        g1.nb = nb;
        g1.logG = WIDE_LOGG1;
        g1.ns = n1;
        const int T = 128;
        unsigned sblocks = (unsigned)(((uint64_t)n1 + T - 1) / T);
        hipLaunchKernelGGL(msm_seg_reduce<F>, dim3(sblocks, n_jobs), dim3(T), 0, st, jobs, g1);
    }
    RJobs j2 = jobs;      // level 2: seg_run / seg_acc (n1 nodes) -> d_seg2 (n2 nodes: run | acc)
    for (uint32_t k = 0; k < n_jobs; ++k) {
        j2.win_s[k] = (uint32_t*)d_seg2[k];
        j2.win_t[k] = (uint32_t*)((char*)d_seg2[k] + (size_t)n2 * PT);
    }
    {
        unsigned blocks = (unsigned)(((uint64_t)n2 * 4 + 255) / 256);
        hipLaunchKernelGGL(msm_node_reduce_q<F>, dim3(blocks, n_jobs), dim3(256), 0, st, j2, n2, WIDE_LOGK2, WIDE_LOGG1);
    }
    RJobs j3 = jobs;      // level 3: virtual windows of 128 level-2 nodes (16 buckets each)
    MsmGeom gv;
    memset(&gv, 0, sizeof gv);
    gv.W = VW;
    gv.B = WIDE_VB;
    gv.nb = nb;
    gv.logG = WIDE_LOGG1 + WIDE_LOGK2;
    gv.ns = WIDE_CHAINS;
    for (uint32_t k = 0; k < n_jobs; ++k) {
        j3.seg_run[k] = d_seg2[k];
        j3.seg_acc[k] = (char*)d_seg2[k] + (size_t)n2 * PT;
        j3.win_s[k] = (uint32_t*)d_vw[k];
        j3.win_t[k] = (uint32_t*)((char*)d_vw[k] + (size_t)VW * PT);
    }
    {
        if ((rc = launch_win_finish_q<F>(dim3(VW, n_jobs), WIDE_CHAINS, st, j3, gv, 1u))) return rc;
    }
    MsmGeom g4;           // level 4: the VW pairs (S_v, T_v) of every job
    memset(&g4, 0, sizeof g4);
    g4.W = 1;
    g4.B = VW;
    g4.nb = VW;
    g4.logG = VW <= 4 ? 0 : VW <= 1024 ? 2 : 3;
    g4.ns = VW >> g4.logG;
    if (g4.ns == 0 || g4.ns > 256) return ZK_ERR_UNSUPPORTED;
    RJobs j4;
    memset(&j4, 0, sizeof j4);
    const size_t PHB = h_stride / 4;      // bytes of one host point
    for (uint32_t k = 0; k < n_jobs; ++k)
        for (uint32_t a = 0; a < 2; ++a) {
            const uint32_t j = 2 * k + a;
            j4.buckets[j] = (char*)d_vw[k] + (size_t)a * VW * PT;
            j4.offsets[j] = nullptr;
            j4.seg_run[j] = (char*)d_seg3[k] + (size_t)a * 2 * g4.ns * PT;
            j4.seg_acc[j] = (char*)d_seg3[k] + ((size_t)a * 2 + 1) * g4.ns * PT;
            j4.win_s[j] = (uint32_t*)(h_out + (size_t)k * h_stride + (size_t)a * 2 * PHB);
            j4.win_t[j] = (uint32_t*)(h_out + (size_t)k * h_stride + ((size_t)a * 2 + 1) * PHB);
        }
    {
        unsigned sblocks = (unsigned)(((uint64_t)g4.ns * 4 + 255) / 256);
        hipLaunchKernelGGL(msm_seg_reduce_q<F>, dim3(sblocks, 2 * n_jobs), dim3(256), 0, st, j4, g4);
        uint32_t chains = 1;
        while (chains < g4.ns) chains <<= 1;
        if ((rc = launch_win_finish_q<F>(dim3(1, 2 * n_jobs), chains, st, j4, g4, 0u))) return rc;
    }
    ZK_HIP_TRY(hipGetLastError());
    return ZK_OK;
}

// fused reduction of the jobs mbs[0..n_jobs) (same geometry); their virtual-window sums land in h_win (n_jobs x win_bytes, pinned host
// memory, arkworks layout) or, with d_winsums, stay on the device in the internal form at d_winsums[k]
template <class Cv>
int pre_queue_reduce(zk_ctx* c, const PrePlan* pls, MsmBufs* const* mbs, uint32_t n_jobs, void* h_win, hipStream_t st, void* const* d_winsums) {
    typedef typename Cv::FqU F;
    constexpr size_t PT = (size_t)4 * Store<F>::WORDS * 4;
    RJobs jobs;
    memset(&jobs, 0, sizeof jobs);
    const PrePlan& p0 = pls[0];
    if (d_winsums && !pre_partial_dev_ok(p0)) return ZK_ERR_UNSUPPORTED;     // checked by the callers before anything is queued
    for (uint32_t k = 0; k < n_jobs; ++k) {
        MsmBufs& mb = *mbs[k];
        jobs.part_pt[k] = mb.part_pt.p;
        jobs.offsets[k] = (const uint32_t*)mb.offsets.p;
        jobs.buckets[k] = mb.buckets.p;
        jobs.q[k] = (uint32_t*)mb.part_key.p + PRE_Q_OFF;
        jobs.seg_run[k] = mb.seg.p;
        jobs.seg_acc[k] = (char*)mb.seg.p + (p0.wide_red ? (size_t)(p0.g.B >> WIDE_LOGG1) : (size_t)p0.gv.W * p0.gv.ns) * PT;
        // the window sums (a few KiB per job) are written by the last kernel straight into the pinned host
        // buffer (hipHostMalloc memory is device-visible): no copy launches at the tail of the call
        if (d_winsums) {
            jobs.win_s[k] = (uint32_t*)d_winsums[k];                                        // S_v | T_v, internal form, straight into the
            jobs.win_t[k] = (uint32_t*)((char*)d_winsums[k] + (size_t)p0.gv.W * PT);        // caller's buffer (the collective's send buffer)
        } else {
            jobs.win_s[k] = (uint32_t*)((char*)h_win + (size_t)k * p0.win_bytes);
            jobs.win_t[k] = jobs.win_s[k] + (size_t)p0.gv.W * 4 * F::SAT;
        }
        jobs.L[k] = mb.acc_chunk_l;            // the plan the accumulation really ran with (pre_queue_accumulate), not a re-derived one
        jobs.lanes[k] = mb.acc_n_lanes;
        jobs.nbk[k] = p0.g1.nb;
    }
    // the queue counters were cleared by the job's sort (psort_hist / the memset of the fallback sort)
    if (p0.wide_red) {
        void* d_vw[MAX_JOBS];
        void* d_seg3[MAX_JOBS];
        void* d_seg2[MAX_JOBS];
        for (uint32_t k = 0; k < n_jobs; ++k) {
            d_vw[k] = mbs[k]->win.p;
            d_seg3[k] = mbs[k]->seg3.p;
            d_seg2[k] = mbs[k]->seg2.p;
        }
        return queue_reduce_wide<F>(c, jobs, n_jobs, p0.g1.nb, d_vw, d_seg3, d_seg2, (char*)h_win, p0.win_bytes, st);
    }
    // on the device the window sums ARE the result: zk_g1_sum_winsums_dev adds the ranks' and the host combines
    return queue_reduce<F>(c, jobs, n_jobs, p0.g1.nb, p0.gv, st, true, d_winsums ? 1u : 0u);
}

}  // namespace

int ZK_SYM(queue_reduce)(zk_ctx* c, const RJobs& jobs, uint32_t n_jobs, uint32_t nb, const MsmGeom& gr, hipStream_t st, bool queues_cleared, uint32_t raw) {
    return queue_reduce<CurveSel::FqU>(c, jobs, n_jobs, nb, gr, st, queues_cleared, raw);
}
int ZK_SYM(pre_queue_reduce)(zk_ctx* c, const PrePlan* pls, MsmBufs* const* mbs, uint32_t n_jobs, void* h_win, hipStream_t st, void* const* d_winsums) {
    return pre_queue_reduce<CurveSel>(c, pls, mbs, n_jobs, h_win, st, d_winsums);
}
int ZK_SYM(queue_sum_winsums)(zk_ctx* c, const void* d_all, uint32_t ranks, uint32_t n_pts, void* h_out, hipStream_t st) {
    typedef CurveSel::FqU F;
    ProfScope ps(c, "msm_sum_winsums", st);
    constexpr size_t PT = (size_t)4 * Store<F>::WORDS * 4;
    uint32_t logq = 0;                                    // quads per sum: half the ranks (rounded up to a power of two), at most 16
    while (logq < 4 && (2u << logq) < ranks) ++logq;
    const uint32_t per_block = 64u >> logq;
    hipLaunchKernelGGL(g1_sum_winsums_q<F>, dim3((n_pts + per_block - 1) / per_block), dim3(256), 64 * PT, st, d_all, ranks, n_pts, logq, (uint32_t*)h_out);
    ZK_HIP_TRY(hipGetLastError());
    return ZK_OK;
}
