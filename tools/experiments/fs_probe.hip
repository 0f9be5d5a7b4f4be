// Prototype of the base-field product on SIGNED radix-2^30 limbs (13 per BLS12-381 Fq element, Montgomery radix 2^390) next to the
// library's unsigned 29-bit form (14 limbs, fieldu.cuh): the same XYZZ mixed-addition chain on both, timed, plus raw products printed
// for an external check (tools/experiments/fs_check.py).  NOT product code: no bound analysis beyond |inputs| < 4q, no conversions.
// build: hipcc --offload-arch=gfx950 -O3 -I ark_plonk_amd/csrc -I include tools/experiments/fs_probe.hip -o tools/bin/fs_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "curve_params.h"
#include "field.cuh"
#include "fieldu.cuh"
#include "fs_consts.h"
typedef Fu<FqBls12_381UParams> FqU;

struct Fs {
    int32_t v[13];
    __device__ static Fs add(const Fs& a, const Fs& b) {
        Fs r;
#pragma unroll
        for (int i = 0; i < 13; ++i) r.v[i] = a.v[i] + b.v[i];
        return r;
    }
    __device__ static Fs sub(const Fs& a, const Fs& b) {
        Fs r;
#pragma unroll
        for (int i = 0; i < 13; ++i) r.v[i] = a.v[i] - b.v[i];
        return r;
    }
    __device__ static void normalize(Fs& t) {          // centred digits, top limb keeps the excess
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            const int32_t c = (t.v[i] + (1 << 29)) >> 30;
            t.v[i] -= c << 30;
            t.v[i + 1] += c;
        }
    }
    // (a*b + c*d) / 2^390 mod q; c, d may be null (plain product)
    template <bool DOT>
    __device__ static Fs mul_(const Fs& a, const Fs& b, const Fs& c, const Fs& d) {
        int32_t m[13];
        Fs r;
        int64_t carry = 0;
#pragma unroll
        for (int k = 0; k < 13; ++k) {
            int64_t a0 = 0, a1 = 0, am = 0;
#pragma unroll
            for (int i = 0; i <= k; ++i) {
                if (DOT) {
                    a0 += (int64_t)a.v[i] * b.v[k - i];
                    a1 += (int64_t)c.v[i] * d.v[k - i];
                } else if (i & 1) a1 += (int64_t)a.v[i] * b.v[k - i];
                else a0 += (int64_t)a.v[i] * b.v[k - i];
            }
#pragma unroll
            for (int i = 0; i < k; ++i) am += (int64_t)m[i] * FS_P[k - i];
            int64_t t = a0 + a1 + am + carry;
            m[k] = (int32_t)(((uint32_t)t * FS_PINV) << 2) >> 2;        // centred 30-bit quotient digit
            t += (int64_t)m[k] * FS_P[0];
            carry = t >> 30;                                              // exact: the low 30 bits are zero
        }
#pragma unroll
        for (int k = 13; k < 25; ++k) {
            int64_t a0 = 0, a1 = 0, am = 0;
#pragma unroll
            for (int i = k - 12; i < 13; ++i) {
                if (DOT) {
                    a0 += (int64_t)a.v[i] * b.v[k - i];
                    a1 += (int64_t)c.v[i] * d.v[k - i];
                } else if (i & 1) a1 += (int64_t)a.v[i] * b.v[k - i];
                else a0 += (int64_t)a.v[i] * b.v[k - i];
                am += (int64_t)m[i] * FS_P[k - i];
            }
            const int64_t t = a0 + a1 + am + carry;
            carry = (t + (1 << 29)) >> 30;
            r.v[k - 13] = (int32_t)((uint32_t)t << 2) >> 2;
        }
        r.v[12] = (int32_t)carry;
        return r;
    }
    __device__ static Fs mul(const Fs& a, const Fs& b) { return mul_<false>(a, b, a, b); }
    __device__ static Fs dot2(const Fs& a, const Fs& b, const Fs& c, const Fs& d) { return mul_<true>(a, b, c, d); }
    __device__ static Fs sqr(const Fs& a) {
        int32_t m[13], a2[13];
        Fs r;
#pragma unroll
        for (int i = 0; i < 13; ++i) a2[i] = a.v[i] * 2;
        int64_t carry = 0;
#pragma unroll
        for (int k = 0; k < 13; ++k) {
            int64_t aa = 0, am = 0;
#pragma unroll
            for (int i = 0; 2 * i < k; ++i) aa += (int64_t)a2[i] * a.v[k - i];
            if ((k & 1) == 0) aa += (int64_t)a.v[k / 2] * a.v[k / 2];
#pragma unroll
            for (int i = 0; i < k; ++i) am += (int64_t)m[i] * FS_P[k - i];
            int64_t t = aa + am + carry;
            m[k] = (int32_t)(((uint32_t)t * FS_PINV) << 2) >> 2;
            t += (int64_t)m[k] * FS_P[0];
            carry = t >> 30;
        }
#pragma unroll
        for (int k = 13; k < 25; ++k) {
            int64_t aa = 0, am = 0;
#pragma unroll
            for (int i = k - 12; 2 * i < k; ++i) aa += (int64_t)a2[i] * a.v[k - i];
            if ((k & 1) == 0) aa += (int64_t)a.v[k / 2] * a.v[k / 2];
#pragma unroll
            for (int i = k - 12; i < 13; ++i) am += (int64_t)m[i] * FS_P[k - i];
            const int64_t t = aa + am + carry;
            carry = (t + (1 << 29)) >> 30;
            r.v[k - 13] = (int32_t)((uint32_t)t << 2) >> 2;
        }
        r.v[12] = (int32_t)carry;
        return r;
    }
};

// the mixed addition of ecu.cuh (madd-2008-s), exceptional cases left out: 7 products, 2 squarings, 1 two-product sum
template <class F>
struct Xy { F x, y; };
template <class F>
struct Xyzz { F x, y, zz, zzz; };

__device__ void madd_s(Xyzz<Fs>& p, const Xy<Fs>& q) {
    Fs s2 = Fs::mul(q.y, p.zzz), u2 = Fs::mul(q.x, p.zz);
    Fs r_ = Fs::sub(s2, p.y), pp_ = Fs::sub(u2, p.x);
    Fs::normalize(r_);
    Fs::normalize(pp_);
    Fs pp = Fs::sqr(pp_), rr = Fs::sqr(r_);
    Fs ppp = Fs::mul(pp_, pp), qq = Fs::mul(p.x, pp);
    p.zz = Fs::mul(p.zz, pp);
    p.zzz = Fs::mul(p.zzz, ppp);
    Fs x3 = Fs::sub(Fs::sub(rr, ppp), Fs::add(qq, qq));
    Fs::normalize(x3);
    Fs dq = Fs::sub(qq, x3), ny = Fs::sub(Fs{{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}}, p.y);
    Fs::normalize(dq);
    p.y = Fs::dot2(r_, dq, ny, ppp);
    p.x = x3;
}
__device__ void madd_u(Xyzz<FqU>& p, const Xy<FqU>& q) {
    typedef FqU F;
    F s2 = F::mul(q.y, p.zzz), u2 = F::mul(q.x, p.zz);
    F r_ = F::sub16(s2, p.y), pp_ = F::sub16(u2, p.x);
    F pp = F::sqr(pp_), rr = F::sqr(r_);
    F ppp = F::mul(pp_, pp), qq = F::mul(p.x, pp);
    p.zz = F::mul(p.zz, pp);
    p.zzz = F::mul(p.zzz, ppp);
    p.x = F::sub8(rr, F::add3(ppp, qq, qq));
    p.y = F::dot2(r_, F::sub16(qq, p.x), F::neg16(p.y), ppp);
}

template <int SIGNED>
__global__ void __launch_bounds__(128) chain(const uint32_t* pts, uint32_t n_pts, uint32_t iters, uint32_t* out) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (SIGNED) {
        Xyzz<Fs> p;
        Xy<Fs> q;
#pragma unroll
        for (int i = 0; i < 13; ++i) {
            p.x.v[i] = (int32_t)(pts[(t * 4 + 0) % n_pts * 16 + i] & 0x1fffffff);
            p.y.v[i] = (int32_t)(pts[(t * 4 + 1) % n_pts * 16 + i] & 0x1fffffff);
            p.zz.v[i] = (int32_t)(pts[(t * 4 + 2) % n_pts * 16 + i] & 0x1fffffff);
            p.zzz.v[i] = (int32_t)(pts[(t * 4 + 3) % n_pts * 16 + i] & 0x1fffffff);
        }
        p.x.v[12] &= 0x3ff; p.y.v[12] &= 0x3ff; p.zz.v[12] &= 0x3ff; p.zzz.v[12] &= 0x3ff;
        for (uint32_t k = 0; k < iters; ++k) {
            const uint32_t j = (t * 7 + k * 13) % n_pts;
#pragma unroll
            for (int i = 0; i < 13; ++i) {
                q.x.v[i] = (int32_t)(pts[j * 16 + i] & 0x1fffffff) - (1 << 28);
                q.y.v[i] = (int32_t)(pts[((j + 1) % n_pts) * 16 + i] & 0x1fffffff) - (1 << 28);
            }
            q.x.v[12] &= 0xff; q.y.v[12] &= 0xff;
            madd_s(p, q);
        }
        uint32_t h = 0;
#pragma unroll
        for (int i = 0; i < 13; ++i) h ^= (uint32_t)(p.x.v[i] + p.y.v[i] * 3 + p.zz.v[i] * 5 + p.zzz.v[i] * 7);
        out[t] = h;
    } else {
        Xyzz<FqU> p;
        Xy<FqU> q;
#pragma unroll
        for (int i = 0; i < 14; ++i) {
            p.x.v[i] = pts[(t * 4 + 0) % n_pts * 16 + i] & 0x1fffffff;
            p.y.v[i] = pts[(t * 4 + 1) % n_pts * 16 + i] & 0x1fffffff;
            p.zz.v[i] = pts[(t * 4 + 2) % n_pts * 16 + i] & 0x1fffffff;
            p.zzz.v[i] = pts[(t * 4 + 3) % n_pts * 16 + i] & 0x1fffffff;
        }
        p.x.v[13] &= 0xf; p.y.v[13] &= 0xf; p.zz.v[13] &= 0xf; p.zzz.v[13] &= 0xf;
        for (uint32_t k = 0; k < iters; ++k) {
            const uint32_t j = (t * 7 + k * 13) % n_pts;
#pragma unroll
            for (int i = 0; i < 14; ++i) {
                q.x.v[i] = pts[j * 16 + i] & 0x1fffffff;
                q.y.v[i] = pts[((j + 1) % n_pts) * 16 + i] & 0x1fffffff;
            }
            q.x.v[13] &= 0xf; q.y.v[13] &= 0xf;
            madd_u(p, q);
        }
        uint32_t h = 0;
#pragma unroll
        for (int i = 0; i < 14; ++i) h ^= p.x.v[i] + p.y.v[i] * 3 + p.zz.v[i] * 5 + p.zzz.v[i] * 7;
        out[t] = h;
    }
}

// raw products for the external check: out = [a (13) | b (13) | a*b/2^390 (13)] per lane
__global__ void products(const uint32_t* pts, uint32_t n_pts, int32_t* out) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    Fs a, b;
#pragma unroll
    for (int i = 0; i < 13; ++i) {
        a.v[i] = (int32_t)(pts[(2 * t) % n_pts * 16 + i] & 0x3fffffff) - (1 << 29);
        b.v[i] = (int32_t)(pts[(2 * t + 1) % n_pts * 16 + i] & 0x3fffffff) - (1 << 29);
    }
    a.v[12] = (a.v[12] >> 18);       // |value| below ~ 2^11 * 2^360 ~ 4q
    b.v[12] = (b.v[12] >> 18);
    const Fs r = Fs::mul(a, b), s = Fs::sqr(a), d = Fs::dot2(a, b, b, a);
#pragma unroll
    for (int i = 0; i < 13; ++i) {
        out[t * 65 + i] = a.v[i];
        out[t * 65 + 13 + i] = b.v[i];
        out[t * 65 + 26 + i] = r.v[i];
        out[t * 65 + 39 + i] = s.v[i];
        out[t * 65 + 52 + i] = d.v[i];
    }
}

int main() {
    const uint32_t n_pts = 4096, lanes = 256 * 4 * 2 * 64, iters = 256;
    std::vector<uint32_t> h(n_pts * 16);
    uint64_t s = 88172645463325252ull;
    for (auto& x : h) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        x = (uint32_t)s;
    }
    uint32_t *d_pts, *d_out;
    int32_t* d_prod;
    (void)hipMalloc(&d_pts, h.size() * 4);
    (void)hipMalloc(&d_out, lanes * 4);
    (void)hipMalloc(&d_prod, 256 * 65 * 4);
    (void)hipMemcpy(d_pts, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int sg = 0; sg < 2; ++sg) {
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0);
            if (sg) chain<1><<<lanes / 128, 128>>>(d_pts, n_pts, iters, d_out);
            else chain<0><<<lanes / 128, 128>>>(d_pts, n_pts, iters, d_out);
            (void)hipEventRecord(e1);
            (void)hipDeviceSynchronize();
            float ms = 0;
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("%s: %.3f ms for %u lanes x %u mixed additions -> %.3e additions/s\n", sg ? "signed 13 x 30-bit" : "unsigned 14 x 29-bit", ms,
                            lanes, iters, (double)lanes * iters / (ms * 1e-3));
        }
    }
    products<<<1, 256>>>(d_pts, n_pts, d_prod);
    std::vector<int32_t> hp(256 * 65);
    (void)hipMemcpy(hp.data(), d_prod, hp.size() * 4, hipMemcpyDeviceToHost);
    FILE* f = fopen("gpurun_out/fs_products.txt", "w");
    if (f) {
        for (int t = 0; t < 256; ++t) {
            for (int i = 0; i < 65; ++i) fprintf(f, "%d ", hp[t * 65 + i]);
            fprintf(f, "\n");
        }
        fclose(f);
    }
    printf("hipGetLastError: %s\n", hipGetErrorString(hipGetLastError()));
    return 0;
}
