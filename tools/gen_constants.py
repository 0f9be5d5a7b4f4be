#!/usr/bin/env python3
"""Generate ark_plonk_amd/csrc/curve_params.h from the field moduli.

Every derived constant (Montgomery R, R^2, -p^-1 mod 2^32, 2-adic root of unity, coset
generator, curve coefficient b, G1 generator in Montgomery form) is computed here from the
moduli / small generators, never hand-typed.  The moduli themselves are
the published BLS12-381 / BN254 parameters (ark-bls12-381 0.3 src/fields/{fr,fq}.rs).

Run:  python3 tools/gen_constants.py > ark_plonk_amd/csrc/curve_params.h
"""
import sys

CURVES = [
    dict(
        name="Bls12_381", cid=0,
        r=0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001,
        q=0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab,
        fr_gen=7, b=4,
        gx=0x17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb,
        gy=0x08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1,
    ),
    dict(
        name="Bn254", cid=1,
        r=21888242871839275222246405745257275088548364400416034343698204186575808495617,
        q=21888242871839275222246405745257275088696311157297823662689037894645226208583,
        fr_gen=5, b=3, gx=1, gy=2,
    ),
]


def limbs32(x, n):
    return [(x >> (32 * i)) & 0xFFFFFFFF for i in range(n)]


def arr(x, n):
    return ", ".join("0x%08xu" % l for l in limbs32(x, n))


def two_adicity(p):
    s, t = 0, p - 1
    while t % 2 == 0:
        t //= 2
        s += 1
    return s


def field_struct(sname, p, extra=""):
    n = (p.bit_length() + 31) // 32
    n += n % 2  # 64-bit limb multiple (host layout is u64 limbs)
    R = (1 << (32 * n)) % p
    R2 = R * R % p
    inv32 = (-pow(p, -1, 1 << 32)) % (1 << 32)
    assert 2 * p < (1 << (32 * n)), "need one spare bit"
    out = []
    out.append(f"struct {sname} {{")
    out.append(f"    static constexpr int N = {n};            // 32-bit limbs")
    out.append(f"    static constexpr int BITS = {p.bit_length()};")
    out.append(f"    static constexpr uint32_t INV32 = 0x{inv32:08x}u;   // -p^-1 mod 2^32")
    inv64 = (-pow(p, -1, 1 << 64)) % (1 << 64)
    out.append(f"    static constexpr uint64_t INV64 = 0x{inv64:016x}ull;   // -p^-1 mod 2^64 (host path)")
    for cname, val in (("MOD", p), ("R", R), ("R2", R2)):
        out.append(f"    ZK_HD static constexpr uint32_t {cname}(int i) {{")
        out.append(f"        constexpr uint32_t t[{n}] = {{{arr(val, n)}}};")
        out.append("        return t[i];")
        out.append("    }")
    out.append(extra)
    out.append("};")
    return "\n".join(out), n, R


LB = 29  # unsaturated limb width used by the device hot paths


def limbs29(x, n):
    return [(x >> (LB * i)) & ((1 << LB) - 1) for i in range(n)]


def arr29(vals):
    return ", ".join("0x%08xu" % l for l in vals)


def borrow_proof(x, n):
    """limbs z_i of x with z_i in [2^29, 2^30) for i < n-1 (so a_i + z_i - b_i never borrows for
    29-bit b_i); sum z_i 2^(29 i) == x."""
    ls = limbs29(x, n - 1) + [x >> (LB * (n - 1))]
    out = []
    for i in range(n - 1):
        out.append(ls[i] + (1 << LB))
        ls[i + 1] -= 1
    out.append(ls[n - 1])
    assert out[-1] >= 0 and sum(v << (LB * i) for i, v in enumerate(out)) == x
    assert all((1 << LB) <= v < (1 << (LB + 1)) for v in out[:-1])
    return out


def unsat_struct(sname, p, nl, sat_words):
    """Parameters of the 29-bit-limb Montgomery representation (R' = 2^(29*nl))."""
    Rp = 1 << (LB * nl)
    Rs = 1 << (32 * sat_words)
    slack = LB * nl - p.bit_length()
    assert 16 * p < (1 << (LB * nl))
    pinv = (-pow(p, -1, 1 << LB)) % (1 << LB)
    out = [f"struct {sname} {{"]
    out.append(f"    static constexpr int NL = {nl};                 // 29-bit limbs")
    out.append(f"    static constexpr int SAT_WORDS = {sat_words};          // 32-bit words of the arkworks layout")
    out.append(f"    static constexpr int SLACK_BITS = {slack};")
    out.append(f"    static constexpr uint32_t PINV = 0x{pinv:08x}u;     // -p^-1 mod 2^29")
    consts = {
        "MOD": limbs29(p, nl),
        "ONE": limbs29(Rp % p, nl),                       # Montgomery one (R' mod p)
        "C_IN": limbs29(Rp * Rp * pow(Rs, -1, p) % p, nl),  # x*R (arkworks form) -> x*R' : montmul by R'^2/R
        "C_OUT": limbs29(Rs % p, nl),                     # x*R' -> x*R : montmul by R
        # multiples of p added before a subtraction (limbs are signed during carry propagation)
        "ZP": limbs29(p, nl),
        "Z2": limbs29(2 * p, nl),
        "Z8": limbs29(8 * p, nl),
        "Z16": limbs29(16 * p, nl),
        # the same multiples with every limb below the top one in [2^29, 2^30): a_i + z_i - b_i never goes below zero for a
        # NORMALISED b (b_i < 2^29), so a difference needs no carry sweep of its own (ntt_pass.cuh: lazily normalised butterflies)
        # (3p rather than 2p for the subtrahend below 2p: the TOP limb of a borrow-proof form is one less than the plain one, and
        # must still cover the subtrahend's)
        "ZB3": borrow_proof(3 * p, nl),
        "ZB8": borrow_proof(8 * p, nl),
        "ZB16": borrow_proof(16 * p, nl),
    }
    # the borrow-proof multiples: the top limb (one less than the plain form's) must cover the subtrahend's -- below 2p, 7p, 15p
    for k, below in ((3, 2), (8, 7), (16, 15)):
        assert consts[f"ZB{k}"][-1] >= (below * p) >> (LB * (nl - 1)), (sname, k)
    for cname, vals in consts.items():
        out.append(f"    ZK_HD static constexpr uint32_t {cname}(int i) {{")
        out.append(f"        constexpr uint32_t t[{nl}] = {{{arr29(vals)}}};")
        out.append("        return t[i];")
        out.append("    }")
    out.append("};")
    return "\n".join(out)


SB = 30  # signed (balanced) limb width of the MSM base-field representation (fields.cuh)


def balanced(x, n):
    """digits d_i of the integer x (any sign): d_i in [-2^29, 2^29) for i < n-1, the top digit takes the rest"""
    out = []
    for _ in range(n - 1):
        d = x & ((1 << SB) - 1)
        if d >= 1 << (SB - 1):
            d -= 1 << SB
        out.append(d)
        x = (x - d) >> SB
    out.append(x)
    return out


def signed_struct(sname, p, nl, sat_words):
    """Parameters of the signed 30-bit-limb Montgomery representation (R' = 2^(30*nl)), fields.cuh.
    Checks the column bound of Fs::mul / Fs::dot2 with the digits of THIS modulus: every product-scanning column must
    stay inside a signed 64-bit accumulator."""
    Rp = 1 << (SB * nl)
    Rs = 1 << (32 * sat_words)
    pd = balanced(p, nl)
    assert sum(d << (SB * i) for i, d in enumerate(pd)) == p
    # operands: "almost balanced" digits |d| <= 2^29 + 4, |value| < 12 p; m digits in [-2^29, 2^29)
    A = (1 << 29) + 4
    a = [A] * (nl - 1) + [(12 * p >> (SB * (nl - 1))) + 2]
    m = [1 << 29] * nl
    for k in range(2 * nl - 1):
        ab = sum(a[i] * a[k - i] for i in range(nl) if 0 <= k - i < nl)
        mp = sum(m[i] * abs(pd[k - i]) for i in range(nl) if 0 <= k - i < nl)
        assert 2 * ab + mp + (1 << 36) < (1 << 63), (sname, k)      # dot2: two products + the reduction + carry + bias
    assert Rp // p >= 512          # |a||b| / R' stays far below p/2 for |a|, |b| < 12 p: products come out in (-p, p)
    pinv = (-pow(p, -1, 1 << SB)) % (1 << SB)
    out = [f"struct {sname} {{"]
    out.append(f"    static constexpr int NL = {nl};                 // signed 30-bit limbs")
    out.append(f"    static constexpr int SAT_WORDS = {sat_words};          // 32-bit words of the arkworks layout")
    out.append(f"    static constexpr uint32_t PINV = 0x{pinv:08x}u;     // -p^-1 mod 2^30")
    consts = {
        "MOD": pd,                                               # p
        "NMOD": balanced(-p, nl),                                # -p
        "ONE": balanced(Rp % p, nl),                             # Montgomery one (R' mod p)
        "C_IN": balanced(Rp * Rp * pow(Rs, -1, p) % p, nl),      # x*R (arkworks form) -> x*R' : montmul by R'^2/R
        "C_OUT": balanced(Rs % p, nl),                           # x*R' -> x*R : montmul by R
    }
    for cname, vals in consts.items():
        out.append(f"    ZK_HD static constexpr int32_t {cname}(int i) {{")
        out.append(f"        constexpr int32_t t[{nl}] = {{{', '.join(str(v) for v in vals)}}};")
        out.append("        return t[i];")
        out.append("    }")
    out.append("    ZK_HD static constexpr uint32_t MODW(int i) {          // p as 32-bit words")
    out.append(f"        constexpr uint32_t t[{sat_words}] = {{{arr(p, sat_words)}}};")
    out.append("        return t[i];")
    out.append("    }")
    out.append("};")
    return "\n".join(out)


def main():
    o = []
    o.append("// GENERATED by tools/gen_constants.py -- do not edit. All values derived from the moduli.")
    o.append("#pragma once")
    o.append("#include <stdint.h>")
    o.append("#include \"zk_common.h\"")
    o.append("")
    for c in CURVES:
        r, q = c["r"], c["q"]
        s = two_adicity(r)
        root = pow(c["fr_gen"], (r - 1) >> s, r)
        assert pow(root, 1 << s, r) == 1 and pow(root, 1 << (s - 1), r) != 1
        nr = (r.bit_length() + 63) // 64 * 2
        Rr = (1 << (32 * nr)) % r
        extra = []
        extra.append(f"    static constexpr int TWO_ADICITY = {s};")
        extra.append("    // 2^TWO_ADICITY-th primitive root of unity = GENERATOR^((r-1)/2^s), Montgomery form")
        extra.append("    ZK_HD static constexpr uint32_t ROOT(int i) {")
        extra.append(f"        constexpr uint32_t t[{nr}] = {{{arr(root * Rr % r, nr)}}};")
        extra.append("        return t[i];")
        extra.append("    }")
        extra.append(f"    static constexpr uint32_t GENERATOR = {c['fr_gen']};   // coset shift (FftParameters::GENERATOR)")
        fr_s, _, _ = field_struct(f"Fr{c['name']}Params", r, "\n".join(extra))
        o.append(fr_s)
        o.append("")
        nq = (q.bit_length() + 63) // 64 * 2
        Rq = (1 << (32 * nq)) % q
        assert (c["gy"] ** 2 - c["gx"] ** 3 - c["b"]) % q == 0
        extra = []
        for cname, val in (("GX", c["gx"] * Rq % q), ("GY", c["gy"] * Rq % q)):
            extra.append(f"    ZK_HD static constexpr uint32_t {cname}(int i) {{")
            extra.append(f"        constexpr uint32_t t[{nq}] = {{{arr(val, nq)}}};")
            extra.append("        return t[i];")
            extra.append("    }")
        extra.append(f"    static constexpr uint32_t COEFF_B = {c['b']};")
        fq_s, _, _ = field_struct(f"Fq{c['name']}Params", q, "\n".join(extra))
        o.append(fq_s)
        o.append("")
        # unsaturated 29-bit limbs for the scalar field (NTT and polynomial kernels)
        nl_r = -(-r.bit_length() // LB)
        o.append(unsat_struct(f"Fr{c['name']}UParams", r, nl_r, nr))
        o.append("")
        # signed 30-bit limbs for the MSM base field
        o.append(signed_struct(f"Fq{c['name']}SParams", q, -(-(q.bit_length() + 5) // SB), nq))
        o.append("")
    sys.stdout.write("\n".join(o) + "\n")


if __name__ == "__main__":
    main()
