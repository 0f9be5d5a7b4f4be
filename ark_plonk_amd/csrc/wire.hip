// N4: canonical wire formats and the transcript, host side only (no device code).
//
//  * ark-serialize 0.3 encodings the reference uses wherever it serialises (transcript.rs:27-33 `append`, the derive on
//    proof.rs:41-103 `Proof`, widget/mod.rs:252-278 verifier key):
//      Fp           little-endian canonical integer, ceil((MODULUS_BITS + flag bits) / 8) bytes
//                   (Fr: 32 B; Fq as the x of a compressed point: 48 B BLS12-381 / 32 B BN254)
//      G1Affine     compressed = x with SWFlags in the two top bits of the LAST byte: bit 7 = "y is the larger of
//                   (y, -y)" (PositiveY), bit 6 = infinity (x written as 0); uncompressed = x, then y carrying the
//                   infinity flag
//  * merlin 3.0 `Transcript` (STROBE-128 over Keccak-f[1600], operations meta-AD / AD / PRF only) and the reference's
//    TranscriptProtocol on top of it (transcript.rs:27-49): append = serialize + append_message, challenge_scalar =
//    size_in_bits/8 = 31 challenge bytes read as a little-endian integer (ark-ff `from_random_bytes`),
//    circuit_domain_sep = ("dom-sep","circuit_size") + append_u64("n").
// merlin, ark-serialize and ark-ff are crates.io dependencies absent from /root/reference (plonk-core/Cargo.toml:51-65):
// this file restates their published behaviour; tests pin it on merlin's published conformance vector.
#include "ctx.h"

namespace {

// ------------------------------------------------------------------------------------------ Keccak-f[1600]
inline uint64_t rol(uint64_t x, int s) { return s ? (x << s) | (x >> (64 - s)) : x; }

void keccak_f1600(uint64_t st[25]) {
    static const uint64_t RC[24] = {0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808aull, 0x8000000080008000ull,
                                    0x000000000000808bull, 0x0000000080000001ull, 0x8000000080008081ull, 0x8000000000008009ull,
                                    0x000000000000008aull, 0x0000000000000088ull, 0x0000000080008009ull, 0x000000008000000aull,
                                    0x000000008000808bull, 0x800000000000008bull, 0x8000000000008089ull, 0x8000000000008003ull,
                                    0x8000000000008002ull, 0x8000000000000080ull, 0x000000000000800aull, 0x800000008000000aull,
                                    0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};
    static const int ROT[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
    for (int round = 0; round < 24; ++round) {
        uint64_t c[5], d[5], b[25];
        for (int x = 0; x < 5; ++x) c[x] = st[x] ^ st[x + 5] ^ st[x + 10] ^ st[x + 15] ^ st[x + 20];
        for (int x = 0; x < 5; ++x) d[x] = c[(x + 4) % 5] ^ rol(c[(x + 1) % 5], 1);
        for (int i = 0; i < 25; ++i) st[i] ^= d[i % 5];
        // rho + pi: lane (x, y) -> (y, 2x + 3y)
        for (int x = 0; x < 5; ++x)
            for (int y = 0; y < 5; ++y) b[y + 5 * ((2 * x + 3 * y) % 5)] = rol(st[x + 5 * y], ROT[x + 5 * y]);
        for (int y = 0; y < 5; ++y)
            for (int x = 0; x < 5; ++x) st[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);
        st[0] ^= RC[round];
    }
}

// ------------------------------------------------------------------------------------------ STROBE-128 (merlin's subset)
struct Strobe128 {
    static constexpr uint8_t R = 166;
    static constexpr uint8_t FLAG_I = 1, FLAG_A = 2, FLAG_C = 4, FLAG_T = 8, FLAG_M = 16, FLAG_K = 32;
    uint8_t state[200];
    uint8_t pos = 0, pos_begin = 0, cur_flags = 0;

    void permute() {
        uint64_t w[25];
        for (int i = 0; i < 25; ++i) {
            w[i] = 0;
            for (int b = 0; b < 8; ++b) w[i] |= (uint64_t)state[8 * i + b] << (8 * b);
        }
        keccak_f1600(w);
        for (int i = 0; i < 25; ++i)
            for (int b = 0; b < 8; ++b) state[8 * i + b] = (uint8_t)(w[i] >> (8 * b));
    }
    explicit Strobe128(const uint8_t* label, size_t len) {
        memset(state, 0, sizeof state);
        const uint8_t head[6] = {1, (uint8_t)(R + 2), 1, 0, 1, 96};
        memcpy(state, head, 6);
        memcpy(state + 6, "STROBEv1.0.2", 12);
        permute();
        meta_ad(label, len, false);
    }
    void run_f() {
        state[pos] ^= pos_begin;
        state[pos + 1] ^= 0x04;
        state[R + 1] ^= 0x80;
        permute();
        pos = 0;
        pos_begin = 0;
    }
    void absorb(const uint8_t* d, size_t n) {
        for (size_t i = 0; i < n; ++i) {
            state[pos] ^= d[i];
            if (++pos == R) run_f();
        }
    }
    void squeeze(uint8_t* d, size_t n) {
        for (size_t i = 0; i < n; ++i) {
            d[i] = state[pos];
            state[pos] = 0;
            if (++pos == R) run_f();
        }
    }
    void begin_op(uint8_t flags, bool more) {
        if (more) return;      // continuation of the running operation (same flags by construction here)
        const uint8_t old_begin = pos_begin;
        pos_begin = (uint8_t)(pos + 1);
        cur_flags = flags;
        const uint8_t hdr[2] = {old_begin, flags};
        absorb(hdr, 2);
        if ((flags & (FLAG_C | FLAG_K)) && pos != 0) run_f();
    }
    void meta_ad(const uint8_t* d, size_t n, bool more) {
        begin_op(FLAG_M | FLAG_A, more);
        absorb(d, n);
    }
    void ad(const uint8_t* d, size_t n, bool more) {
        begin_op(FLAG_A, more);
        absorb(d, n);
    }
    void prf(uint8_t* d, size_t n, bool more) {
        begin_op(FLAG_I | FLAG_A | FLAG_C, more);
        squeeze(d, n);
    }
};

// ------------------------------------------------------------------------------------------ field / curve helpers
template <class F>
void fp_to_le_bytes(const F& mont, uint8_t* out, size_t nbytes) {
    F c = F::from_mont(mont);
    for (size_t i = 0; i < nbytes; ++i) out[i] = i / 4 < (size_t)F::N ? (uint8_t)(c.v[i / 4] >> (8 * (i % 4))) : 0;
}
// canonical little-endian bytes -> Montgomery; false when the integer is >= p
template <class F, class P>
bool fp_from_le_bytes(const uint8_t* in, size_t nbytes, F& out) {
    F c = F::zero();
    for (size_t i = 0; i < nbytes; ++i) {
        if (i / 4 >= (size_t)F::N) {
            if (in[i]) return false;
            continue;
        }
        c.v[i / 4] |= (uint32_t)in[i] << (8 * (i % 4));
    }
    for (int i = F::N - 1; i >= 0; --i) {      // c < p ?
        if (c.v[i] < P::MOD(i)) break;
        if (c.v[i] > P::MOD(i)) return false;
        if (i == 0) return false;               // c == p
    }
    out = F::to_mont(c);
    return true;
}
// a > b as canonical integers (ark's Ord on Fp compares into_repr())
template <class F>
bool fp_gt(const F& a_mont, const F& b_mont) {
    F a = F::from_mont(a_mont), b = F::from_mont(b_mont);
    for (int i = F::N - 1; i >= 0; --i) {
        if (a.v[i] != b.v[i]) return a.v[i] > b.v[i];
    }
    return false;
}

template <class Cv>
struct Wire {
    typedef typename Cv::Fq Fq;
    typedef typename Cv::Fr Fr;
    typedef typename Cv::FqP FqP;
    typedef typename Cv::FrP FrP;
    static constexpr size_t FQ_BYTES = (FqP::BITS + 2 + 7) / 8;   // serialize_with_flags::<SWFlags>: 2 flag bits
    static constexpr size_t FR_BYTES = (FrP::BITS + 7) / 8;

    static int fr_ser(const uint64_t* mont, uint8_t* out) {
        Fr x;
        memcpy(x.v, mont, 32);
        fp_to_le_bytes<Fr>(x, out, FR_BYTES);
        return ZK_OK;
    }
    static int fr_de(const uint8_t* in, uint64_t* mont) {
        Fr x;
        if (!fp_from_le_bytes<Fr, FrP>(in, FR_BYTES, x)) return ZK_ERR_BAD_ARG;
        memcpy(mont, x.v, 32);
        return ZK_OK;
    }
    static void load_xy(const uint64_t* xy, Fq& x, Fq& y) {
        memcpy(x.v, xy, sizeof x.v);
        memcpy(y.v, xy + Fq::N / 2, sizeof y.v);
    }
    static bool is_inf(const uint64_t* xy, uint8_t inf) {
        if (inf) return true;
        Fq x, y;
        load_xy(xy, x, y);
        return x.is_zero() && (y.is_zero() || y == Fq::one());
    }
    static int g1_ser_c(const uint64_t* xy, uint8_t inf, uint8_t* out) {
        if (is_inf(xy, inf)) {
            memset(out, 0, FQ_BYTES);
            out[FQ_BYTES - 1] |= 1u << 6;
            return ZK_OK;
        }
        Fq x, y;
        load_xy(xy, x, y);
        fp_to_le_bytes<Fq>(x, out, FQ_BYTES);
        if (fp_gt<Fq>(y, Fq::neg(y))) out[FQ_BYTES - 1] |= 1u << 7;
        return ZK_OK;
    }
    static int g1_ser_u(const uint64_t* xy, uint8_t inf, uint8_t* out) {
        if (is_inf(xy, inf)) {
            memset(out, 0, 2 * FQ_BYTES);
            out[2 * FQ_BYTES - 1] |= 1u << 6;
            return ZK_OK;
        }
        Fq x, y;
        load_xy(xy, x, y);
        fp_to_le_bytes<Fq>(x, out, FQ_BYTES);
        fp_to_le_bytes<Fq>(y, out + FQ_BYTES, FQ_BYTES);
        return ZK_OK;
    }
    static Fq curve_rhs(const Fq& x) { return Fq::add(Fq::mul(Fq::sqr(x), x), Fq::from_u32(FqP::COEFF_B)); }
    // r * P == O ?  (ark's is_in_correct_subgroup_assuming_on_curve)
    static bool in_subgroup(const Fq& x, const Fq& y) {
        typedef XYZZ<Fq> PH;
        Affine<Fq> a;
        a.x = x;
        a.y = y;
        PH acc = PH::infinity();
        for (int i = Fr::N - 1; i >= 0; --i)
            for (int b = 31; b >= 0; --b) {
                acc = PH::dbl(acc);
                if ((FrP::MOD(i) >> b) & 1u) acc = PH::madd(acc, a);
            }
        return acc.is_inf();
    }
    static void store_xy(uint64_t* xy, uint8_t* inf, const Fq& x, const Fq& y, bool infinity) {
        memcpy(xy, x.v, sizeof x.v);
        memcpy(xy + Fq::N / 2, y.v, sizeof y.v);
        if (inf) *inf = infinity ? 1 : 0;
    }
    static int g1_de_c(const uint8_t* in, uint64_t* xy, uint8_t* inf) {
        uint8_t buf[FQ_BYTES];
        memcpy(buf, in, FQ_BYTES);
        const uint8_t flags = buf[FQ_BYTES - 1] & 0xC0;
        buf[FQ_BYTES - 1] &= 0x3F;
        if (flags == 0xC0) return ZK_ERR_BAD_ARG;           // SWFlags::from_u8: both bits set is invalid
        Fq x;
        if (!fp_from_le_bytes<Fq, FqP>(buf, FQ_BYTES, x)) return ZK_ERR_BAD_ARG;
        if (flags & 0x40) {                                  // infinity: GroupAffine::zero() = (0, 1)
            store_xy(xy, inf, Fq::zero(), Fq::one(), true);
            return ZK_OK;
        }
        // both base fields are 3 mod 4: sqrt(a) = a^((p+1)/4)
        uint32_t e[Fq::N + 1];
        uint64_t carry = 1;
        for (int i = 0; i < Fq::N; ++i) {
            uint64_t t = (uint64_t)FqP::MOD(i) + carry;
            e[i] = (uint32_t)t;
            carry = t >> 32;
        }
        e[Fq::N] = (uint32_t)carry;
        for (int i = 0; i < Fq::N; ++i) e[i] = (e[i] >> 2) | (e[i + 1] << 30);
        const Fq rhs = curve_rhs(x);
        Fq y = Fq::pow_limbs(rhs, e, Fq::N);
        if (Fq::sqr(y) != rhs) return ZK_ERR_BAD_ARG;        // x is not on the curve
        const Fq ny = Fq::neg(y);
        const bool want_greatest = (flags & 0x80) != 0;
        if (fp_gt<Fq>(y, ny) != want_greatest) y = ny;
        if (!in_subgroup(x, y)) return ZK_ERR_BAD_ARG;
        store_xy(xy, inf, x, y, false);
        return ZK_OK;
    }
    static int g1_de_u(const uint8_t* in, uint64_t* xy, uint8_t* inf) {
        uint8_t buf[FQ_BYTES];
        memcpy(buf, in + FQ_BYTES, FQ_BYTES);
        const uint8_t flags = buf[FQ_BYTES - 1] & 0xC0;
        if (flags == 0xC0) return ZK_ERR_BAD_ARG;           // SWFlags::from_u8: both bits set is invalid (as g1_de_c)
        buf[FQ_BYTES - 1] &= 0x3F;
        Fq x, y;
        if (!fp_from_le_bytes<Fq, FqP>(in, FQ_BYTES, x) || !fp_from_le_bytes<Fq, FqP>(buf, FQ_BYTES, y)) return ZK_ERR_BAD_ARG;
        if (flags & 0x40) {
            store_xy(xy, inf, Fq::zero(), Fq::one(), true);
            return ZK_OK;
        }
        if (Fq::sqr(y) != curve_rhs(x) || !in_subgroup(x, y)) return ZK_ERR_BAD_ARG;
        store_xy(xy, inf, x, y, false);
        return ZK_OK;
    }
    // transcript.rs:35-46: size = size_in_bits / 8 challenge bytes -> F::from_random_bytes
    static int challenge(Strobe128& st, const uint8_t* label, size_t len, uint64_t* out_mont);
};

void transcript_append(Strobe128& st, const uint8_t* label, size_t label_len, const uint8_t* msg, size_t msg_len) {
    const uint32_t n = (uint32_t)msg_len;
    const uint8_t le[4] = {(uint8_t)n, (uint8_t)(n >> 8), (uint8_t)(n >> 16), (uint8_t)(n >> 24)};
    st.meta_ad(label, label_len, false);
    st.meta_ad(le, 4, true);
    st.ad(msg, msg_len, false);
}
void transcript_challenge(Strobe128& st, const uint8_t* label, size_t label_len, uint8_t* out, size_t out_len) {
    const uint32_t n = (uint32_t)out_len;
    const uint8_t le[4] = {(uint8_t)n, (uint8_t)(n >> 8), (uint8_t)(n >> 16), (uint8_t)(n >> 24)};
    st.meta_ad(label, label_len, false);
    st.meta_ad(le, 4, true);
    st.prf(out, out_len, false);
}

template <class Cv>
int Wire<Cv>::challenge(Strobe128& st, const uint8_t* label, size_t len, uint64_t* out_mont) {
    constexpr size_t SZ = FrP::BITS / 8;      // 31 for both curves: always below the modulus
    uint8_t buf[32] = {0};
    transcript_challenge(st, label, len, buf, SZ);
    Fr x;
    if (!fp_from_le_bytes<Fr, FrP>(buf, 32, x)) return ZK_ERR_BAD_ARG;
    memcpy(out_mont, x.v, 32);
    return ZK_OK;
}

}  // namespace

struct zk_transcript {
    Strobe128 st;
    explicit zk_transcript(const uint8_t* l, size_t n) : st((const uint8_t*)"Merlin v1.0", 11) { transcript_append(st, (const uint8_t*)"dom-sep", 7, l, n); }
};

#define WIRE_DISPATCH(curve_id, expr_bls, expr_bn)          \
    do {                                                    \
        if ((curve_id) == ZK_CURVE_BLS12_381) return expr_bls; \
        if ((curve_id) == ZK_CURVE_BN254) return expr_bn;   \
        return ZK_ERR_BAD_ARG;                              \
    } while (0)

extern "C" {

size_t zk_fr_serialized_size(int curve_id) {
    return curve_id == ZK_CURVE_BLS12_381 ? Wire<CurveBls>::FR_BYTES : curve_id == ZK_CURVE_BN254 ? Wire<CurveBn>::FR_BYTES : 0;
}
size_t zk_g1_compressed_size(int curve_id) {
    return curve_id == ZK_CURVE_BLS12_381 ? Wire<CurveBls>::FQ_BYTES : curve_id == ZK_CURVE_BN254 ? Wire<CurveBn>::FQ_BYTES : 0;
}
int zk_fr_serialize(int curve_id, const uint64_t* fr_mont, uint8_t* out) {
    if (!fr_mont || !out) return ZK_ERR_BAD_ARG;
    WIRE_DISPATCH(curve_id, Wire<CurveBls>::fr_ser(fr_mont, out), Wire<CurveBn>::fr_ser(fr_mont, out));
}
int zk_fr_deserialize(int curve_id, const uint8_t* in, uint64_t* fr_mont) {
    if (!fr_mont || !in) return ZK_ERR_BAD_ARG;
    WIRE_DISPATCH(curve_id, Wire<CurveBls>::fr_de(in, fr_mont), Wire<CurveBn>::fr_de(in, fr_mont));
}
int zk_g1_serialize_compressed(int curve_id, const uint64_t* xy_mont, uint8_t inf, uint8_t* out) {
    if (!xy_mont || !out) return ZK_ERR_BAD_ARG;
    WIRE_DISPATCH(curve_id, Wire<CurveBls>::g1_ser_c(xy_mont, inf, out), Wire<CurveBn>::g1_ser_c(xy_mont, inf, out));
}
int zk_g1_deserialize_compressed(int curve_id, const uint8_t* in, uint64_t* xy_mont, uint8_t* inf) {
    if (!xy_mont || !in) return ZK_ERR_BAD_ARG;
    WIRE_DISPATCH(curve_id, Wire<CurveBls>::g1_de_c(in, xy_mont, inf), Wire<CurveBn>::g1_de_c(in, xy_mont, inf));
}
int zk_g1_serialize_uncompressed(int curve_id, const uint64_t* xy_mont, uint8_t inf, uint8_t* out) {
    if (!xy_mont || !out) return ZK_ERR_BAD_ARG;
    WIRE_DISPATCH(curve_id, Wire<CurveBls>::g1_ser_u(xy_mont, inf, out), Wire<CurveBn>::g1_ser_u(xy_mont, inf, out));
}
int zk_g1_deserialize_uncompressed(int curve_id, const uint8_t* in, uint64_t* xy_mont, uint8_t* inf) {
    if (!xy_mont || !in) return ZK_ERR_BAD_ARG;
    WIRE_DISPATCH(curve_id, Wire<CurveBls>::g1_de_u(in, xy_mont, inf), Wire<CurveBn>::g1_de_u(in, xy_mont, inf));
}

zk_transcript* zk_transcript_new(const uint8_t* label, size_t label_len) {
    if (!label && label_len) return nullptr;
    return new zk_transcript(label, label_len);
}
zk_transcript* zk_transcript_clone(const zk_transcript* t) { return t ? new zk_transcript(*t) : nullptr; }
void zk_transcript_free(zk_transcript* t) { delete t; }
int zk_transcript_append_message(zk_transcript* t, const uint8_t* label, size_t label_len, const uint8_t* msg, size_t msg_len) {
    if (!t || (!label && label_len) || (!msg && msg_len)) return ZK_ERR_BAD_ARG;
    transcript_append(t->st, label, label_len, msg, msg_len);
    return ZK_OK;
}
int zk_transcript_append_u64(zk_transcript* t, const uint8_t* label, size_t label_len, uint64_t v) {
    uint8_t le[8];
    for (int i = 0; i < 8; ++i) le[i] = (uint8_t)(v >> (8 * i));
    return zk_transcript_append_message(t, label, label_len, le, 8);
}
int zk_transcript_challenge_bytes(zk_transcript* t, const uint8_t* label, size_t label_len, uint8_t* out, size_t out_len) {
    if (!t || (!label && label_len) || (!out && out_len)) return ZK_ERR_BAD_ARG;
    transcript_challenge(t->st, label, label_len, out, out_len);
    return ZK_OK;
}
int zk_transcript_append_fr(zk_transcript* t, int curve_id, const uint8_t* label, size_t label_len, const uint64_t* fr_mont) {
    uint8_t buf[32];
    int rc = zk_fr_serialize(curve_id, fr_mont, buf);
    if (rc) return rc;
    return zk_transcript_append_message(t, label, label_len, buf, zk_fr_serialized_size(curve_id));
}
int zk_transcript_append_g1(zk_transcript* t, int curve_id, const uint8_t* label, size_t label_len, const uint64_t* xy_mont, uint8_t inf) {
    uint8_t buf[48];
    int rc = zk_g1_serialize_compressed(curve_id, xy_mont, inf, buf);
    if (rc) return rc;
    return zk_transcript_append_message(t, label, label_len, buf, zk_g1_compressed_size(curve_id));
}
int zk_transcript_challenge_scalar(zk_transcript* t, int curve_id, const uint8_t* label, size_t label_len, uint64_t* fr_mont) {
    if (!t || !fr_mont || (!label && label_len)) return ZK_ERR_BAD_ARG;
    WIRE_DISPATCH(curve_id, Wire<CurveBls>::challenge(t->st, label, label_len, fr_mont), Wire<CurveBn>::challenge(t->st, label, label_len, fr_mont));
}
// `PublicInputs { values: BTreeMap<usize, F> }` (pi.rs:28-36) under `label` (prover.rs:182 uses b"pi"):
// u64 count, then (u64 position, Fr) in ascending position order.  `PublicInputs::insert` (pi.rs:56-65) drops zero values
// ("zeros are the implicit value of empty positions"), so the reference's map never holds one -- neither on the prover's nor on
// the verifier's side: a zero entry handed in here is skipped the same way, it is not part of the message.
int zk_transcript_append_public_inputs(zk_transcript* t, int curve_id, const uint8_t* label, size_t label_len, const uint64_t* positions,
                                       const uint64_t* values_mont, size_t n) {
    if (!t || (n && (!positions || !values_mont))) return ZK_ERR_BAD_ARG;
    const size_t f = zk_fr_serialized_size(curve_id);
    if (!f) return ZK_ERR_BAD_ARG;
    for (size_t i = 1; i < n; ++i)
        if (positions[i] <= positions[i - 1]) return ZK_ERR_BAD_ARG;     // a BTreeMap iterates in strictly ascending key order
    auto is_zero = [&](size_t i) { return (values_mont[4 * i] | values_mont[4 * i + 1] | values_mont[4 * i + 2] | values_mont[4 * i + 3]) == 0; };
    size_t kept = 0;
    for (size_t i = 0; i < n; ++i) kept += is_zero(i) ? 0 : 1;
    std::vector<uint8_t> buf(8 + kept * (8 + f));
    uint8_t* w = buf.data();
    for (int b = 0; b < 8; ++b) *w++ = (uint8_t)((uint64_t)kept >> (8 * b));
    for (size_t i = 0; i < n; ++i) {
        if (is_zero(i)) continue;
        for (int b = 0; b < 8; ++b) *w++ = (uint8_t)(positions[i] >> (8 * b));
        int rc = zk_fr_serialize(curve_id, values_mont + 4 * i, w);
        if (rc) return rc;
        w += f;
    }
    return zk_transcript_append_message(t, label, label_len, buf.data(), buf.size());
}

int zk_transcript_circuit_domain_sep(zk_transcript* t, uint64_t n) {
    int rc = zk_transcript_append_message(t, (const uint8_t*)"dom-sep", 7, (const uint8_t*)"circuit_size", 12);
    if (rc) return rc;
    return zk_transcript_append_u64(t, (const uint8_t*)"n", 1, n);
}

// proof.rs:41-103 + proof.rs ProofEvaluations: the derive serialises the fields in declaration order, every commitment
// and opening proof as a compressed G1 (kzg10::Commitment(G1Affine); kzg10::Proof { w: G1Affine, random_v: Option<Fr> }
// with random_v = None -> one 0 byte), every evaluation as an Fr
size_t zk_proof_serialized_size(int curve_id, uint32_t n_custom_evals, const uint32_t* label_lens) {
    const size_t g = zk_g1_compressed_size(curve_id), f = zk_fr_serialized_size(curve_id);
    if (!g) return 0;
    size_t sz = 13 * g + 2 * (g + 1) + (ZK_PROOF_N_EVALS)*f + 8;   // 13 commitments, 2 openings, fixed evaluations, Vec length
    for (uint32_t i = 0; i < n_custom_evals; ++i) sz += 8 + (label_lens ? label_lens[i] : 0) + f;   // (String, F): u64 length + bytes + Fr
    return sz;
}

int zk_proof_serialize(int curve_id, const zk_proof* p, uint8_t* out, size_t cap, size_t* written) {
    if (!p || !out) return ZK_ERR_BAD_ARG;
    if (!p->commitments || !p->commitment_inf || !p->openings || !p->opening_inf || !p->evals) return ZK_ERR_BAD_ARG;
    if (p->n_custom_evals && (!p->custom_labels || !p->custom_evals)) return ZK_ERR_BAD_ARG;
    const size_t g = zk_g1_compressed_size(curve_id), f = zk_fr_serialized_size(curve_id);
    if (!g) return ZK_ERR_BAD_ARG;
    std::vector<uint32_t> ll(p->n_custom_evals);
    for (uint32_t i = 0; i < p->n_custom_evals; ++i) {
        if (!p->custom_labels[i]) return ZK_ERR_BAD_ARG;
        ll[i] = (uint32_t)strlen(p->custom_labels[i]);
    }
    const size_t need = zk_proof_serialized_size(curve_id, p->n_custom_evals, ll.data());
    if (cap < need) return ZK_ERR_BAD_ARG;
    const int L = curve_id == ZK_CURVE_BLS12_381 ? 6 : 4;
    uint8_t* w = out;
    int rc;
    for (int i = 0; i < 13; ++i) {
        if ((rc = zk_g1_serialize_compressed(curve_id, p->commitments + (size_t)i * 2 * L, p->commitment_inf[i], w))) return rc;
        w += g;
    }
    for (int i = 0; i < 2; ++i) {
        if ((rc = zk_g1_serialize_compressed(curve_id, p->openings + (size_t)i * 2 * L, p->opening_inf[i], w))) return rc;
        w += g;
        *w++ = 0;     // Option<Fr>::None
    }
    for (int i = 0; i < ZK_PROOF_N_EVALS; ++i) {
        if ((rc = zk_fr_serialize(curve_id, p->evals + (size_t)i * 4, w))) return rc;
        w += f;
    }
    const uint64_t nc = p->n_custom_evals;
    for (int b = 0; b < 8; ++b) *w++ = (uint8_t)(nc >> (8 * b));
    for (uint32_t i = 0; i < p->n_custom_evals; ++i) {
        const uint64_t sl = ll[i];
        for (int b = 0; b < 8; ++b) *w++ = (uint8_t)(sl >> (8 * b));
        memcpy(w, p->custom_labels[i], sl);
        w += sl;
        if ((rc = zk_fr_serialize(curve_id, p->custom_evals + (size_t)i * 4, w))) return rc;
        w += f;
    }
    if (written) *written = (size_t)(w - out);
    return ZK_OK;
}

}  // extern "C"
