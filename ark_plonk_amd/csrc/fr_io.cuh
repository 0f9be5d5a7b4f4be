// Loads / stores of scalar-field elements for the kernels that work on the 29-bit-limb Fr type (kzg.hip, lookup.hip):
// 32-byte arkworks elements <-> limbs with no domain change, lazily reduced limb vectors (48 bytes), and the kernel-argument
// form of a constant multiplier.
#pragma once
#include "zk_common.h"

namespace {

struct Packed {            // a field element as 8 little-endian words (kernel argument form)
    uint32_t w[8];
};

template <class FU>
ZK_D FU ld_u(const void* base, uint64_t idx) {       // 32-byte element -> limbs (no domain change)
    const uint4* q = reinterpret_cast<const uint4*>(base) + 2 * idx;
    uint4 a = q[0], b = q[1];
    uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    return FU::split_words(w);
}
template <class FU>
ZK_D void st_u(void* base, uint64_t idx, const FU& x) {   // value < 2r -> canonical 32-byte element
    uint32_t w[8];
    FU::canonical_lt2p(x).pack_words(w);
    uint4* q = reinterpret_cast<uint4*>(base) + 2 * idx;
    q[0] = make_uint4(w[0], w[1], w[2], w[3]);
    q[1] = make_uint4(w[4], w[5], w[6], w[7]);
}
// lazily reduced limb vectors: 9 limbs in 12 words
template <class FU>
ZK_D FU ld_l(const void* base, uint64_t idx) {
    const uint4* q = reinterpret_cast<const uint4*>(base) + 3 * idx;
    uint4 a = q[0], b = q[1], c = q[2];
    const uint32_t w[12] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w};
    FU r;
#pragma unroll
    for (int i = 0; i < FU::NL; ++i) r.v[i] = w[i];
    return r;
}
template <class FU>
ZK_D void st_l(void* base, uint64_t idx, const FU& x) {
    uint32_t w[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < FU::NL; ++i) w[i] = x.v[i];
    uint4* q = reinterpret_cast<uint4*>(base) + 3 * idx;
    q[0] = make_uint4(w[0], w[1], w[2], w[3]);
    q[1] = make_uint4(w[4], w[5], w[6], w[7]);
    q[2] = make_uint4(w[8], w[9], w[10], w[11]);
}
template <class FU>
ZK_D FU unpack(const Packed& p) { return FU::split_words(p.w); }

}  // namespace
