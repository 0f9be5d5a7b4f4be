// Host-side sanitizer run of the N4 code (ark_plonk_amd/csrc/wire.hip is host-only): built by tests/test_sanitize.py with
// -fsanitize=address,undefined on the CPU (GPU sanitizers are not available on the pool) and driven with random and
// adversarial byte strings through every decoder, plus encode/decode round trips and transcript calls of odd sizes.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "ark_plonk_amd.h"

static uint64_t s = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() {
    s ^= s << 13;
    s ^= s >> 7;
    s ^= s << 17;
    return s;
}

int main() {
    long decoded_fr = 0, decoded_g1 = 0, rejected = 0, roundtrips = 0;
    for (int curve = 0; curve < 2; ++curve) {
        const size_t fr_b = zk_fr_serialized_size(curve), g1_b = zk_g1_compressed_size(curve);
        const size_t L = g1_b / 8;                                  // Fq limbs
        std::vector<uint8_t> buf(2 * g1_b), enc(2 * g1_b);
        std::vector<uint64_t> fr(4), xy(2 * L), xy2(2 * L);
        uint8_t inf = 0, inf2 = 0;
        for (int it = 0; it < 20000; ++it) {
            for (auto& b : buf) b = (uint8_t)rnd();
            if (it % 7 == 0) memset(buf.data(), 0xff, buf.size());           // above the modulus
            if (it % 11 == 0) memset(buf.data(), 0, buf.size());
            if (it % 5 == 0) buf[g1_b - 1] &= 0x1f;                          // plausible x, no flags
            if (it % 13 == 0) buf[g1_b - 1] = 0x40;                          // infinity flag with junk x
            if (zk_fr_deserialize(curve, buf.data(), fr.data()) == ZK_OK) {
                ++decoded_fr;
                if (zk_fr_serialize(curve, fr.data(), enc.data()) != ZK_OK || memcmp(enc.data(), buf.data(), fr_b)) return 10;
                ++roundtrips;
            } else {
                ++rejected;
            }
            if (zk_g1_deserialize_compressed(curve, buf.data(), xy.data(), &inf) == ZK_OK) {
                ++decoded_g1;
                if (zk_g1_serialize_compressed(curve, xy.data(), inf, enc.data()) != ZK_OK) return 11;
                // ark-ec accepts an infinity flag over any canonical x (GroupAffine::deserialize returns zero() before looking at x);
                // every finite point has exactly one encoding
                if (!inf && memcmp(enc.data(), buf.data(), g1_b)) return 14;
                if (inf && (enc[g1_b - 1] != 0x40 || enc[0] != 0)) return 15;
                if (zk_g1_serialize_uncompressed(curve, xy.data(), inf, enc.data()) != ZK_OK) return 12;
                if (zk_g1_deserialize_uncompressed(curve, enc.data(), xy2.data(), &inf2) != ZK_OK || inf != inf2 || (!inf && xy != xy2)) return 13;
                // both SWFlags bits set is no encoding at all (SWFlags::from_u8 returns None), in either form
                enc[2 * g1_b - 1] |= 0xC0;
                if (zk_g1_deserialize_uncompressed(curve, enc.data(), xy2.data(), &inf2) == ZK_OK) return 16;
                ++roundtrips;
            } else {
                ++rejected;
            }
            (void)zk_g1_deserialize_uncompressed(curve, buf.data(), xy2.data(), &inf2);
        }
        // transcript: odd sizes around the STROBE rate (166), clones, large challenge reads
        zk_transcript* t = zk_transcript_new((const uint8_t*)"fuzz", 4);
        std::vector<uint8_t> msg(1000), out(1000);
        for (size_t len : {0u, 1u, 165u, 166u, 167u, 331u, 332u, 333u, 1000u}) {
            for (auto& b : msg) b = (uint8_t)rnd();
            if (zk_transcript_append_message(t, msg.data(), len % 17, msg.data(), len) != ZK_OK) return 20;
            zk_transcript* c = zk_transcript_clone(t);
            if (zk_transcript_challenge_bytes(c, (const uint8_t*)"c", 1, out.data(), len) != ZK_OK) return 21;
            if (zk_transcript_challenge_scalar(c, curve, (const uint8_t*)"s", 1, fr.data()) != ZK_OK) return 22;
            if (zk_transcript_append_fr(c, curve, (const uint8_t*)"f", 1, fr.data()) != ZK_OK) return 23;
            zk_transcript_free(c);
        }
        if (zk_transcript_circuit_domain_sep(t, 1u << 20) != ZK_OK) return 24;
        zk_transcript_free(t);
        // a proof with every commitment at infinity and zero evaluations: sizes and bounds
        zk_proof p;
        memset(&p, 0, sizeof p);
        std::vector<uint64_t> zero_xy(2 * L, 0);
        size_t need = zk_proof_serialized_size(curve, 0, nullptr), written = 0;
        std::vector<uint8_t> pb(need);
        (void)zk_proof_serialize(curve, &p, pb.data(), need, &written);          // null members -> error code, no crash
        (void)zk_proof_serialize(curve, &p, pb.data(), 0, &written);
    }
    // null arguments are errors, never crashes
    if (zk_fr_deserialize(0, nullptr, nullptr) == ZK_OK || zk_transcript_append_message(nullptr, nullptr, 0, nullptr, 0) == ZK_OK) return 30;
    if (zk_fr_serialize(7, nullptr, nullptr) == ZK_OK) return 31;
    std::printf("fuzz_wire ok decoded_fr=%ld decoded_g1=%ld rejected=%ld roundtrips=%ld\n", decoded_fr, decoded_g1, rejected, roundtrips);
    return 0;
}
