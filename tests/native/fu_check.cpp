// Host-side harness for the unsaturated field / lazy XYZZ code (fieldu.cuh, ecu.cuh): the same
// templates the HIP kernels instantiate, compiled with g++ so tests/test_fieldu.py can check them
// against the big-int oracle without a GPU.
#include <cstdint>
#include <cstring>
#include "../../ark_plonk_amd/csrc/curve_params.h"
#include "../../ark_plonk_amd/csrc/ecu.cuh"
#include "../../ark_plonk_amd/csrc/fields.cuh"

template <class F>
static void fu_binop(int op, const uint32_t* a, const uint32_t* b, uint32_t* out) {
    F x = F::from_sat(a), y = F::from_sat(b), r;
    switch (op) {
    case 0: r = F::mul(x, y); break;
    case 1: r = F::add(x, y); break;
    case 2: r = F::sub8(x, y); break;
    case 3: r = F::sub16(x, y); break;
    case 4: r = F::sqr(x); break;
    case 5: r = F::inverse(x); break;
    case 6: r = F::neg_canonical(x); break;
    case 7: r = F::dbl(x); break;
    case 8: {  // a long lazy chain: ((x - y + 16p) * (x + x + y) - x*y + 8p)^2
        F t = F::mul(F::sub16(x, y), F::add3(x, x, y));
        r = F::sqr(F::sub8(t, F::mul(x, y)));
        break;
    }
    case 9:  // x*y + (x - y)(2x + y) with one reduction
        r = F::dot2(x, y, F::sub16(x, y), F::add3(x, x, y));
        break;
    case 10:  // x*y - y*y through the negated operand
        r = F::dot2(x, y, F::neg16(y), y);
        break;
    default: r = F::zero();
    }
    r.to_sat(out);
}

// acc (affine or infinity) + sequence of affine points with signs -> affine result
template <class F>
static int xyzz_chain(const uint32_t* pts_xy, const uint8_t* flags /* bit0 negate, bit1 use full add, bit2 double acc first */, int n,
                      uint32_t* out_xy) {
    constexpr int W = F::SAT;
    XYZZu<F> acc = XYZZu<F>::infinity();
    for (int i = 0; i < n; ++i) {
        AffineU<F> p;
        p.x = F::canonical_lt2p(F::from_sat(pts_xy + (2 * i) * W));
        p.y = F::canonical_lt2p(F::from_sat(pts_xy + (2 * i + 1) * W));
        bool null = true;
        for (int k = 0; k < 2 * W; ++k) null = null && pts_xy[2 * i * W + k] == 0;
        if (null) continue;
        if (flags[i] & 1) p.y = F::neg_canonical(p.y);
        if (flags[i] & 4) acc = XYZZu<F>::dbl(acc);
        if (flags[i] & 2) acc = XYZZu<F>::add(acc, XYZZu<F>::from_affine(p));
        else acc = XYZZu<F>::madd(acc, p);
    }
    AffineU<F> a;
    bool fin = acc.to_affine(a);
    a.x.to_sat(out_xy);
    a.y.to_sat(out_xy + W);
    return fin ? 0 : 1;
}

typedef Fu<FqBls12_381UParams> FqB;
typedef Fu<FrBls12_381UParams> FrB;
typedef Fu<FqBn254UParams> FqN;
typedef Fu<FrBn254UParams> FrN;
typedef Fs<FqBls12_381SParams> FqBs;     // signed 30-bit limbs (fields.cuh)
typedef Fs<FqBn254SParams> FqNs;

extern "C" {
// field: 0 Fq-BLS, 1 Fr-BLS, 2 Fq-BN, 3 Fr-BN (29-bit limbs); 4 Fq-BLS, 5 Fq-BN (signed 30-bit limbs)
void fu_op(int field, int op, const uint32_t* a, const uint32_t* b, uint32_t* out) {
    switch (field) {
    case 0: fu_binop<FqB>(op, a, b, out); break;
    case 1: fu_binop<FrB>(op, a, b, out); break;
    case 2: fu_binop<FqN>(op, a, b, out); break;
    case 3: fu_binop<FrN>(op, a, b, out); break;
    case 4: fu_binop<FqBs>(op, a, b, out); break;
    case 5: fu_binop<FqNs>(op, a, b, out); break;
    }
}
int fu_xyzz_chain(int curve, const uint32_t* pts_xy, const uint8_t* flags, int n, uint32_t* out_xy) {
    switch (curve) {
    case 0: return xyzz_chain<FqB>(pts_xy, flags, n, out_xy);
    case 1: return xyzz_chain<FqN>(pts_xy, flags, n, out_xy);
    case 2: return xyzz_chain<FqBs>(pts_xy, flags, n, out_xy);
    default: return xyzz_chain<FqNs>(pts_xy, flags, n, out_xy);
    }
}
}
