// Host-side harness for the unsaturated fields / lazy XYZZ code (fieldu.cuh, fields.cuh, ecu.cuh): the same
// templates the HIP kernels instantiate, compiled with g++ so tests/test_fieldu.py can check them
// against the big-int oracle without a GPU.
#include <cstdint>
#include <cstring>
#include "../../ark_plonk_amd/csrc/curve_params.h"
#include "../../ark_plonk_amd/csrc/fieldu.cuh"
#include "../../ark_plonk_amd/csrc/ecu.cuh"

template <class F>
static void fs_dot2(int op, const uint32_t* a, const uint32_t* b, uint32_t* out) {
    F x = F::from_sat(a), y = F::from_sat(b), r;
    if (op == 9) r = F::dot2(x, y, F::sub16(x, y), F::add3(x, x, y));       // x*y + (x - y)(2x + y) with one reduction
    else if (op == 10) r = F::dot2(x, y, F::neg16(y), y);                   // x*y - y*y through the negated operand
    else r = F::sub_sum3(F::mul(x, y), F::sqr(x), F::sqr(y), F::mul(x, x));  // xy - 2x^2 - y^2, one carry step
    r.to_sat(out);
}

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

typedef Fu<FrBls12_381UParams> FrB;      // 29-bit limbs (fieldu.cuh): the scalar fields
typedef Fu<FrBn254UParams> FrN;
typedef Fs<FqBls12_381SParams> FqBs;     // signed 30-bit limbs (fields.cuh): the base fields
typedef Fs<FqBn254SParams> FqNs;

// raw signed-limb operands (no conversion on the way in): the worst-case limb patterns of the column bounds of fields.cuh.
// op 0: mul(a, b)   1: sqr(a)   2: dot2(a, b, c, d)   3: sub_sum3(a, b, c, d) (strict operands)   4: add3 / sub chains
template <class F>
static void fs_raw(int op, const int32_t* a, const int32_t* b, const int32_t* c, const int32_t* d, int32_t* out) {
    F x, y, z, w, r;
    for (int i = 0; i < F::NL; ++i) {
        x.v[i] = (uint32_t)a[i];
        y.v[i] = (uint32_t)b[i];
        z.v[i] = (uint32_t)c[i];
        w.v[i] = (uint32_t)d[i];
    }
    switch (op) {
    case 0: r = F::mul(x, y); break;
    case 1: r = F::sqr(x); break;
    case 2: r = F::dot2(x, y, z, w); break;
    case 3: r = F::sub_sum3(x, y, z, w); break;
    default: r = F::sub16(F::add3(x, y, z), w); break;
    }
    for (int i = 0; i < F::NL; ++i) out[i] = (int32_t)r.v[i];
}

extern "C" {
int fs_limbs(int field) { return field == 0 ? FqBs::NL : FqNs::NL; }
void fs_raw_op(int field, int op, const int32_t* a, const int32_t* b, const int32_t* c, const int32_t* d, int32_t* out) {
    if (field == 0) fs_raw<FqBs>(op, a, b, c, d, out);
    else fs_raw<FqNs>(op, a, b, c, d, out);
}
// field: 0 Fq-BLS, 2 Fq-BN (signed 30-bit limbs); 1 Fr-BLS, 3 Fr-BN (29-bit limbs)
void fu_op(int field, int op, const uint32_t* a, const uint32_t* b, uint32_t* out) {
    switch (field) {
    case 0: op >= 9 ? fs_dot2<FqBs>(op, a, b, out) : fu_binop<FqBs>(op, a, b, out); break;
    case 1: fu_binop<FrB>(op, a, b, out); break;
    case 2: op >= 9 ? fs_dot2<FqNs>(op, a, b, out) : fu_binop<FqNs>(op, a, b, out); break;
    case 3: fu_binop<FrN>(op, a, b, out); break;
    }
}
int fu_xyzz_chain(int curve, const uint32_t* pts_xy, const uint8_t* flags, int n, uint32_t* out_xy) {
    return curve == 0 ? xyzz_chain<FqBs>(pts_xy, flags, n, out_xy) : xyzz_chain<FqNs>(pts_xy, flags, n, out_xy);
}
}
