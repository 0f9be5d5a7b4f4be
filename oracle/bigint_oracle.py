"""TEST INFRASTRUCTURE ONLY -- definitional big-integer oracle for the NTT + MSM hot path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
The product path (ark_plonk_amd/) never imports anything under oracle/.

PARITY UNPINNED BY THE REFERENCE: heliaxdev/ark-plonk holds no golden vectors, KATs or
fixtures for this path (every reference test draws from OsRng) and
the arithmetic lives in crates.io dependencies that are absent from /root/reference:
ark-poly 0.3.0, ark-ec 0.3.0, ark-ff 0.3.0, ark-bls12-381 0.3.0, ark-poly-commit 0.3.0
(plonk-core/Cargo.toml:51-58; no Cargo.lock).  This file restates the *published
mathematical definitions* those crates implement; because the outputs at the kernel
boundary are canonical (a reduced Montgomery residue / a normalised affine point has one
representation), any correct implementation is bit-identical with arkworks.

What is restated, and the reference call site each function stands behind:
  * ntt(kind, ...)    -- ark_poly::EvaluationDomain::{fft,ifft,coset_fft,coset_ifft}
                         as called from plonk-core/src/proof_system/prover.rs:196-203,
                         quotient_poly.rs:72-120,175-177, permutation/mod.rs:671-674,751.
  * msm(...)          -- ark_ec::msm::VariableBaseMSM::multi_scalar_mul as called from
                         plonk-core/src/commitment.rs:45 and every PC::commit
                         (prover.rs:213,289-291,...).
  * kzg_commit/open   -- ark_poly_commit::kzg10::KZG10::{commit,open} semantics
                         (leading-zero stripping, into_repr, witness polynomial division),
                         called from prover.rs:213 and prover.rs:582-591.
  * perm_product / lookup_product -- the serial loops of Permutation::compute_permutation_poly and
                         ::compute_lookup_permutation_poly (plonk-core/src/permutation/mod.rs:652-752,
                         754-822) up to their final domain.ifft; this code IS in
                         the reference, and is restated line by line (row ratio, running product,
                         (n+1)-th value dropped).
  * quotient_at / quotient_evals -- the pointwise quotient of proof_system/quotient_poly.rs:34-178 with
                         every widget it sums (arithmetic.rs:51-63, range.rs:47-74, logic.rs:65-133,
                         ecc/fixed_base_scalar_mul.rs:88-156, ecc/curve_addition.rs:62-97,
                         proof_system/permutation.rs:62-153, widget/lookup.rs:97-151) and the division by
                         the vanishing polynomial over the coset (preprocess.rs:429-452); SURVEY.md 8f row N1.
External anchors used to pin constants (tests/test_oracle.py):
  * TWO_ADIC_ROOT_OF_UNITY of ark-bls12-381 Fr (decimal constant quoted in SURVEY.md 8a).
  * [2]G1 x-coordinate of BLS12-381 (the published compressed encoding a572cbea...f0f4e).
  * tests/published_points.py: EIP-2537's G1 + G1, EIP-196's [2] / [3] / [9] (1, 2) and "chfast1" scalar multiplication on
    alt_bn128, c-kzg-4844's SCALE2_ROOT_OF_UNITY[2..4], circom's 2^28-th root of unity of BN254's scalar field.
"""
from __future__ import annotations

from dataclasses import dataclass


@dataclass(frozen=True)
class Curve:
    name: str
    curve_id: int
    r: int            # scalar field modulus
    q: int            # base field modulus
    fr_limbs: int     # 64-bit limbs of Fr
    fq_limbs: int     # 64-bit limbs of Fq
    two_adicity: int
    fr_generator: int  # multiplicative generator of Fr (coset shift, ark FftParameters::GENERATOR)
    b: int            # y^2 = x^3 + b
    gx: int
    gy: int

    @property
    def fr_R(self):
        return 1 << (64 * self.fr_limbs)

    @property
    def fq_R(self):
        return 1 << (64 * self.fq_limbs)

    def root_of_unity(self, log_n: int) -> int:
        """ark FftParameters::TWO_ADIC_ROOT_OF_UNITY ^ (2^(two_adicity-log_n))."""
        if log_n > self.two_adicity:
            raise ValueError("log_n exceeds two-adicity")
        root = pow(self.fr_generator, (self.r - 1) >> self.two_adicity, self.r)
        return pow(root, 1 << (self.two_adicity - log_n), self.r)


BLS12_381 = Curve(
    name="bls12_381", curve_id=0,
    r=0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001,
    q=0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab,
    fr_limbs=4, fq_limbs=6, two_adicity=32, fr_generator=7, b=4,
    gx=0x17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb,
    gy=0x08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1,
)

BN254 = Curve(
    name="bn254", curve_id=1,
    r=21888242871839275222246405745257275088548364400416034343698204186575808495617,
    q=21888242871839275222246405745257275088696311157297823662689037894645226208583,
    fr_limbs=4, fq_limbs=4, two_adicity=28, fr_generator=5, b=3, gx=1, gy=2,
)

CURVES = {0: BLS12_381, 1: BN254, "bls12_381": BLS12_381, "bn254": BN254}

KIND_FFT, KIND_IFFT, KIND_COSET_FFT, KIND_COSET_IFFT = 0, 1, 2, 3


# ----------------------------------------------------------------------------- field helpers
def to_mont(x: int, p: int, R: int) -> int:
    return (x * R) % p


def from_mont(x: int, p: int, R: int) -> int:
    return (x * pow(R, -1, p)) % p


def int_to_limbs(x: int, n: int) -> list:
    return [(x >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(n)]


def limbs_to_int(limbs) -> int:
    v = 0
    for i, l in enumerate(limbs):
        v |= int(l) << (64 * i)
    return v


# ----------------------------------------------------------------------------- NTT (definition)
def _dft_pow2(a: list, w: int, p: int) -> list:
    """Recursive radix-2 DFT: out[i] = sum_j a[j] w^(ij) mod p. len(a) a power of two."""
    n = len(a)
    if n == 1:
        return [a[0] % p]
    w2 = w * w % p
    ev = _dft_pow2(a[0::2], w2, p)
    od = _dft_pow2(a[1::2], w2, p)
    out = [0] * n
    t = 1
    h = n // 2
    for i in range(h):
        x = od[i] * t % p
        out[i] = (ev[i] + x) % p
        out[i + h] = (ev[i] - x) % p
        t = t * w % p
    return out


def dft_naive(a: list, w: int, p: int) -> list:
    """O(N^2) DFT straight from the definition (used to pin _dft_pow2 at tiny sizes)."""
    n = len(a)
    return [sum(a[j] * pow(w, i * j, p) for j in range(n)) % p for i in range(n)]


def ntt(curve: Curve, kind: int, log_n: int, values: list) -> list:
    """Canonical-integer (non-Montgomery) semantics of ark_poly 0.3 Radix2EvaluationDomain.

    fft        : e[i] = sum_j a[j] w^(ij)                (input zero-extended to N)
    ifft       : a[j] = N^-1 sum_i e[i] w^(-ij)
    coset_fft  : e[i] = sum_j a[j] (g w^i)^j,  g = Fr::multiplicative_generator
    coset_ifft : a[j] = g^-j N^-1 sum_i e[i] w^(-ij)
    """
    p = curve.r
    n = 1 << log_n
    if len(values) > n:
        raise ValueError("input longer than domain")
    a = [v % p for v in values] + [0] * (n - len(values))
    w = curve.root_of_unity(log_n)
    g = curve.fr_generator
    if kind == KIND_FFT:
        return _dft_pow2(a, w, p)
    if kind == KIND_COSET_FFT:
        gj = 1
        b = []
        for v in a:
            b.append(v * gj % p)
            gj = gj * g % p
        return _dft_pow2(b, w, p)
    winv = pow(w, -1, p)
    ninv = pow(n, -1, p)
    out = [v * ninv % p for v in _dft_pow2(a, winv, p)]
    if kind == KIND_IFFT:
        return out
    if kind == KIND_COSET_IFFT:
        ginv = pow(g, -1, p)
        gj = 1
        res = []
        for v in out:
            res.append(v * gj % p)
            gj = gj * ginv % p
        return res
    raise ValueError("bad kind")


def horner(coeffs: list, x: int, p: int) -> int:
    acc = 0
    for c in reversed(coeffs):
        acc = (acc * x + c) % p
    return acc


# ----------------------------------------------------------------------------- G1 (affine, definition)
INF = None  # point at infinity


def ec_add(curve: Curve, P, Q):
    q = curve.q
    if P is INF:
        return Q
    if Q is INF:
        return P
    x1, y1 = P
    x2, y2 = Q
    if x1 == x2:
        if (y1 + y2) % q == 0:
            return INF
        lam = 3 * x1 * x1 * pow(2 * y1, -1, q) % q
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, q) % q
    x3 = (lam * lam - x1 - x2) % q
    y3 = (lam * (x1 - x3) - y1) % q
    return (x3, y3)


def ec_neg(curve: Curve, P):
    if P is INF:
        return INF
    return (P[0], (-P[1]) % curve.q)


def ec_mul(curve: Curve, k: int, P):
    """double-and-add, MSB first."""
    R = INF
    if k < 0:
        k, P = -k, ec_neg(curve, P)
    for bit in bin(k)[2:] if k else "":
        R = ec_add(curve, R, R)
        if bit == "1":
            R = ec_add(curve, R, P)
    return R


def on_curve(curve: Curve, P) -> bool:
    if P is INF:
        return True
    x, y = P
    return (y * y - x * x * x - curve.b) % curve.q == 0


def msm(curve: Curve, bases: list, scalars: list):
    """sum_i scalars[i] * bases[i] over min(len) pairs (ark truncates to the shorter slice)."""
    acc = INF
    for P, s in zip(bases, scalars):
        acc = ec_add(curve, acc, ec_mul(curve, s % curve.r if s >= curve.r else s, P))
    return acc


def srs_powers(curve: Curve, tau: int, n: int) -> list:
    """powers_of_g[i] = tau^i * G  (KZG10 setup shape, ark-poly-commit 0.3 kzg10::setup)."""
    out = []
    G = (curve.gx, curve.gy)
    t = 1
    for _ in range(n):
        out.append(ec_mul(curve, t, G))
        t = t * tau % curve.r
    return out


# ----------------------------------------------------------------------------- KZG10 commit / open semantics
def kzg_commit(curve: Curve, powers: list, coeffs: list):
    """ark-poly-commit 0.3 KZG10::commit with hiding_bound=None:
    skip leading (low-degree) zero coefficients, canonical scalars, MSM over powers[lz..]."""
    lz = 0
    while lz < len(coeffs) and coeffs[lz] % curve.r == 0:
        lz += 1
    return msm(curve, powers[lz:], [c % curve.r for c in coeffs[lz:]])


def kzg_witness_poly(curve: Curve, coeffs: list, z: int) -> list:
    """(p(X) - p(z)) / (X - z) by synthetic division (KZG10::compute_witness_polynomial)."""
    p = curve.r
    n = len(coeffs)
    if n <= 1:
        return []
    w = [0] * (n - 1)
    acc = 0
    for i in range(n - 1, 0, -1):
        acc = (coeffs[i] + acc * z) % p
        w[i - 1] = acc
    return w


def kzg_open(curve: Curve, powers: list, polys: list, z: int, challenge: int):
    """PC::open for SonicKZG10 without degree bounds / hiding: p = sum_k chi^k p_k;
    witness = (p - p(z))/(X - z); proof.w = commit(witness)."""
    p = curve.r
    m = max(len(c) for c in polys)
    comb = [0] * m
    chi = 1
    for c in polys:
        for i, v in enumerate(c):
            comb[i] = (comb[i] + chi * v) % p
        chi = chi * challenge % p
    return kzg_commit(curve, powers, kzg_witness_poly(curve, comb, z))


# ----------------------------------------------------------------------------- window rule (for add counting)
# ----------------------------------------------------------------------------- grand products (N2)
PERM_K = (1, 7, 13, 17)   # plonk-core/src/permutation/constants.rs:12-22 (K1, K2, K3; the first coset is H itself)


def perm_product(curve: Curve, log_n: int, wires, sigmas, beta: int, gamma: int):
    """Evaluations of z over the domain, permutation/mod.rs:652-752 before the ifft.
    wires, sigmas: 4 lists of n canonical integers (sigmas = domain.fft of the sigma polynomials, mod.rs:671-676).
    Returns (z[0..n), dropped (n+1)-th value)."""
    p, n = curve.r, 1 << log_n
    w = curve.root_of_unity(log_n)
    state, root, z = 1, 1, []
    for i in range(n):
        z.append(state)
        num = den = 1
        for k in range(4):
            num = num * ((wires[k][i] + beta * PERM_K[k] * root + gamma) % p) % p      # numerator_irreducible, mod.rs:626-636
            den = den * ((wires[k][i] + beta * sigmas[k][i] + gamma) % p) % p          # denominator_irreducible, mod.rs:638-647
        if den == 0:
            raise ZeroDivisionError("zero denominator (the reference panics: inverse().unwrap(), mod.rs:727)")
        state = state * num % p * pow(den, -1, p) % p
        root = root * w % p
    return z, state


def lookup_product(curve: Curve, f, t, h1, h2, delta: int, epsilon: int):
    """Evaluations of z2, permutation/mod.rs:754-822 before the ifft.  Returns (p[0..n), dropped value)."""
    p, n = curve.r, len(f)
    opd = (1 + delta) % p
    e1d = epsilon * opd % p
    state, out = 1, []
    for i in range(n):
        out.append(state)
        nx = (i + 1) % n                                                                # t_next / h_1_next, mod.rs:771-772
        num = opd * ((epsilon + f[i]) % p) % p * ((e1d + t[i] + delta * t[nx]) % p) % p
        den = ((e1d + h1[i] + h2[i] * delta) % p) * ((e1d + h2[i] + h1[nx] * delta) % p) % p
        if den == 0:
            raise ZeroDivisionError("zero denominator (lookup_ratio: inverse().unwrap(), mod.rs:820)")
        state = state * num % p * pow(den, -1, p) % p
    return out, state


# ----------------------------------------------------------------------------- quotient numerator / Z_H (N1)
QUOTIENT_COLS = ("w_l", "w_r", "w_o", "w_4", "z", "z2", "f", "table", "h1", "h2", "pi", "l1",
                 "q_m", "q_l", "q_r", "q_o", "q_4", "q_c", "q_arith", "q_range", "q_logic", "q_fixed", "q_var", "q_lookup",
                 "sigma0", "sigma1", "sigma2", "sigma3")
QUOTIENT_CHALLENGES = ("alpha", "beta", "gamma", "delta", "epsilon", "zeta", "range", "logic", "fixed", "var", "lookup",
                       "coeff_a", "coeff_d")


def _delta4(f, p):
    """f(f-1)(f-2)(f-3): widget/range.rs:66-74, widget/logic.rs:94-102."""
    return f * (f - 1) % p * (f - 2) % p * (f - 3) % p


def range_constraint(p, s, a, b, c, d, d_n):
    """`Range::constraints` (widget/range.rs:47-63): the factor of the range selector, at one point or at the evaluations."""
    k = s * s % p
    return (_delta4((c - 4 * d) % p, p) + _delta4((b - 4 * c) % p, p) * k + _delta4((a - 4 * b) % p, p) * k * k
            + _delta4((d_n - 4 * a) % p, p) * k * k * k) % p * s % p


def logic_constraint(p, s, a, b, c, d, a_n, b_n, d_n, q_c):
    """`Logic::constraints` + `delta_xor_and` (widget/logic.rs:65-133)."""
    k = s * s % p
    la, lb, ld, w = (a_n - 4 * a) % p, (b_n - 4 * b) % p, (d_n - 4 * d) % p, c
    F = w * (w * (4 * w - 18 * (la + lb) + 81) + 18 * (la * la + lb * lb) - 81 * (la + lb) + 83) % p
    E = (3 * (la + lb + ld) - 2 * F) % p
    B = q_c * (9 * ld - 3 * (la + lb)) % p
    return (_delta4(la, p) + _delta4(lb, p) * k + _delta4(ld, p) * k * k + (w - la * lb) * k * k * k + (B + E) * k * k * k * k) % p * s % p


def fixed_base_constraint(p, s, a, b, c, d, a_n, b_n, d_n, q_l, q_r, q_c, ca, cd):
    """`FixedBaseScalarMul::constraints` (widget/ecc/fixed_base_scalar_mul.rs:88-156)."""
    k = s * s % p
    bit = (d_n - d - d) % p
    bit_cons = bit * (bit - 1) % p * (bit + 1) % p
    y_alpha = (bit * bit * (q_r - 1) + 1) % p
    x_alpha = q_l * bit % p
    xy_cons = (bit * q_c - c) * k % p
    x_acc = ((a_n + a_n * c * a * b * cd) - (x_alpha * b + y_alpha * a)) * k * k % p
    y_acc = ((b_n - b_n * c * a * b * cd) - (y_alpha * b - ca * x_alpha * a)) * k * k * k % p
    return (bit_cons + x_acc + y_acc + xy_cons) % p * s % p


def curve_add_constraint(p, s, a, b, c, d, a_n, b_n, d_n, ca, cd):
    """`CurveAddition::constraints` (widget/ecc/curve_addition.rs:62-97)."""
    k = s * s % p
    x1, x3, y1, y3, x2, y2, x1y2 = a, a_n, b, b_n, c, d, d_n
    y1x2, y1y2, x1x2 = y1 * x2 % p, y1 * y2 % p, x1 * x2 % p
    xy = (x1 * y2 - x1y2) % p
    x3c = ((x1y2 + y1x2) - (x3 + x3 * cd * x1y2 * y1x2)) * k % p
    y3c = ((y1y2 - ca * x1x2) - (y3 - y3 * cd * x1y2 * y1x2)) * k * k % p
    return (xy + x3c + y3c) % p * s % p


def quotient_at(curve: Curve, log_n: int, i: int, col, ch) -> int:
    """One evaluation of the quotient over the 4n coset: plonk-core/src/proof_system/quotient_poly.rs:34-178
    (`compute`) at index i, i.e. (gate_constraints[i] + permutation[i] + lookup[i]) / v_h_coset_4n[i].
    col: dict name -> list of 4n canonical integers (coset evaluations; QUOTIENT_COLS); "next" = index i+4 cyclic
    (the reference appends e[0..4], quotient_poly.rs:75-118).  ch: dict of challenges / curve coefficients."""
    p = curve.r
    n, n4 = 1 << log_n, 4 << log_n
    nx = (i + 4) % n4
    a, b, c, d = col["w_l"][i], col["w_r"][i], col["w_o"][i], col["w_4"][i]
    a_n, b_n, d_n = col["w_l"][nx], col["w_r"][nx], col["w_4"][nx]
    q_l, q_r, q_c = col["q_l"][i], col["q_r"][i], col["q_c"][i]
    # -- gate constraints, quotient_poly.rs:182-268
    arith = (a * b * col["q_m"][i] + a * q_l + b * q_r + c * col["q_o"][i] + d * col["q_4"][i] + q_c) % p * col["q_arith"][i] % p  # arithmetic.rs:51-63
    ca, cd = ch["coeff_a"], ch["coeff_d"]
    rng = range_constraint(p, ch["range"], a, b, c, d, d_n) * col["q_range"][i] % p
    logic = logic_constraint(p, ch["logic"], a, b, c, d, a_n, b_n, d_n, q_c) * col["q_logic"][i] % p
    fixed = fixed_base_constraint(p, ch["fixed"], a, b, c, d, a_n, b_n, d_n, q_l, q_r, q_c, ca, cd) * col["q_fixed"][i] % p
    var = curve_add_constraint(p, ch["var"], a, b, c, d, a_n, b_n, d_n, ca, cd) * col["q_var"][i] % p
    gate = (arith + col["pi"][i] + rng + logic + fixed + var) % p                                        # quotient_poly.rs:262-266
    # -- permutation, proof_system/permutation.rs:62-153
    al, be, ga = ch["alpha"], ch["beta"], ch["gamma"]
    x = curve.fr_generator * pow(curve.root_of_unity(log_n + 2), i, p) % p                              # linear_evaluations, preprocess.rs:209-212
    z_i, z_n = col["z"][i], col["z"][nx]
    ident = (a + be * x + ga) * (b + be * PERM_K[1] * x + ga) % p * (c + be * PERM_K[2] * x + ga) % p * (d + be * PERM_K[3] * x + ga) % p * z_i % p * al % p
    copy = (a + be * col["sigma0"][i] + ga) * (b + be * col["sigma1"][i] + ga) % p * (c + be * col["sigma2"][i] + ga) % p \
        * (d + be * col["sigma3"][i] + ga) % p * z_n % p * al % p
    l1a = al * al % p * col["l1"][i] % p                                                                # coset_fft(alpha^2 * L1), quotient_poly.rs:294
    perm = (ident - copy + (z_i - 1) * l1a) % p
    # -- lookup, widget/lookup.rs:97-151
    de, ep, ze, ls = ch["delta"], ch["epsilon"], ch["zeta"], ch["lookup"]
    opd = (1 + de) % p
    e1d = ep * opd % p
    tuple_ = (a + ze * (b + ze * (c + ze * d))) % p                                                     # util.rs lc()
    t_i, t_n, h1_i, h1_n, h2_i = col["table"][i], col["table"][nx], col["h1"][i], col["h1"][nx], col["h2"][i]
    z2_i, z2_n = col["z2"][i], col["z2"][nx]
    la_ = col["q_lookup"][i] * (tuple_ - col["f"][i]) % p * ls % p
    lb_ = z2_i * opd % p * (ep + col["f"][i]) % p * (e1d + t_i + de * t_n) % p * ls * ls % p
    lc_ = -z2_n * (e1d + h1_i + de * h2_i) % p * (e1d + h2_i + de * h1_n) % p * ls * ls % p
    ld_ = (z2_i - 1) * col["l1"][i] % p * ls * ls * ls % p
    look = (la_ + lb_ + lc_ + ld_) % p
    # -- / Z_H over the coset, preprocess.rs:429-452: v_h[i] = g^n * (w_4n^n)^i - 1
    vh = (pow(curve.fr_generator, n, p) * pow(curve.root_of_unity(log_n + 2), n * i, p) - 1) % p
    return (gate + perm + look) % p * pow(vh, -1, p) % p


def quotient_evals(curve: Curve, log_n: int, col, ch) -> list:
    return [quotient_at(curve, log_n, i, col, ch) for i in range(4 << log_n)]


LIN_POLYS = ("w_l", "w_r", "w_o", "w_4", "t_1", "t_2", "t_3", "t_4", "z", "z2", "f", "h1", "h2", "table")
LIN_KEY = ("q_m", "q_l", "q_r", "q_o", "q_4", "q_c", "q_arith", "q_range", "q_logic", "q_fixed", "q_var", "q_lookup",
           "sigma0", "sigma1", "sigma2", "sigma3")


def poly_scale(poly, s, p):
    return [c * s % p for c in poly]


def poly_add(*polys, p):
    m = max(len(q) for q in polys)
    return [sum(q[i] for q in polys if i < len(q)) % p for i in range(m)]


def linearisation(curve: Curve, log_n: int, key, polys, ch):
    """plonk-core/src/proof_system/linearisation_poly.rs:164-350 (`compute`) on canonical integers.
    key: name -> coefficient list for LIN_KEY (the prover key's polynomials); polys: name -> coefficient list for LIN_POLYS;
    ch: the challenge dict of quotient_at plus "z" (z_challenge).
    Returns (coefficients of the linearisation polynomial, evaluations: name -> value with the reference's field names)."""
    p = curve.r
    n = 1 << log_n
    z = ch["z"]
    zw = z * curve.root_of_unity(log_n) % p                                                   # :200-201
    ev = lambda name, x: horner((key[name] if name in key else polys[name]), x, p)            # noqa: E731  DensePolynomial::evaluate
    a, b, c, d = ev("w_l", z), ev("w_r", z), ev("w_o", z), ev("w_4", z)                       # :203-206
    s1, s2, s3 = ev("sigma0", z), ev("sigma1", z), ev("sigma2", z)                            # :214-219
    z_next = ev("z", zw)                                                                      # :220
    q_arith, q_lookup = ev("q_arith", z), ev("q_lookup", z)                                   # :230-233
    q_c, q_l, q_r = ev("q_c", z), ev("q_l", z), ev("q_r", z)                                  # :236-238
    a_n, b_n, d_n = ev("w_l", zw), ev("w_r", zw), ev("w_4", zw)                               # :239-241
    z2_next, h1_e, h1_next, h2_e = ev("z2", zw), ev("h1", z), ev("h1", zw), ev("h2", z)       # :255-258
    f_e, t_e, t_next = ev("f", z), ev("table", z), ev("table", zw)                            # :259-261
    vanishing = (pow(z, n, p) - 1) % p                                                        # :267-269
    z_n = (vanishing + 1) % p
    l1 = vanishing * pow(n * (z - 1) % p, -1, p) % p                                          # proof.rs:622-633
    ca, cd = ch["coeff_a"], ch["coeff_d"]
    # gate constraints, :353-411
    arith = poly_scale(poly_add(poly_scale(key["q_m"], a * b % p, p), poly_scale(key["q_l"], a, p), poly_scale(key["q_r"], b, p),
                                poly_scale(key["q_o"], c, p), poly_scale(key["q_4"], d, p), key["q_c"], p=p), q_arith, p)   # arithmetic.rs:66-82
    rng = poly_scale(key["q_range"], range_constraint(p, ch["range"], a, b, c, d, d_n), p)                                   # widget/mod.rs:96-104
    logic = poly_scale(key["q_logic"], logic_constraint(p, ch["logic"], a, b, c, d, a_n, b_n, d_n, q_c), p)
    fixed = poly_scale(key["q_fixed"], fixed_base_constraint(p, ch["fixed"], a, b, c, d, a_n, b_n, d_n, q_l, q_r, q_c, ca, cd), p)
    var = poly_scale(key["q_var"], curve_add_constraint(p, ch["var"], a, b, c, d, a_n, b_n, d_n, ca, cd), p)
    gate = poly_add(arith, rng, logic, fixed, var, p=p)
    # permutation, proof_system/permutation.rs:156-291
    al, be, ga = ch["alpha"], ch["beta"], ch["gamma"]
    bz = be * z % p
    ident = (a + bz + ga) * (b + PERM_K[1] * bz + ga) % p * (c + PERM_K[2] * bz + ga) % p * (d + PERM_K[3] * bz + ga) % p * al % p
    copy = (a + be * s1 + ga) * (b + be * s2 + ga) % p * (c + be * s3 + ga) % p * (be * z_next % p) % p * al % p
    perm = poly_add(poly_scale(polys["z"], ident, p), poly_scale(key["sigma3"], -copy % p, p),
                    poly_scale(polys["z"], l1 * al % p * al % p, p), p=p)
    # lookup, widget/lookup.rs:154-203
    de, ep, ze, ls = ch["delta"], ch["epsilon"], ch["zeta"], ch["lookup"]
    opd = (1 + de) % p
    e1d = ep * opd % p
    tuple_ = (a + ze * (b + ze * (c + ze * d))) % p
    la = poly_scale(key["q_lookup"], (tuple_ - f_e) * ls % p, p)
    lb = poly_scale(polys["z2"], (opd * (ep + f_e) % p * (e1d + t_e + de * t_next) % p * ls % p * ls + l1 * ls * ls * ls) % p, p)
    lc_ = poly_scale(polys["h1"], (-z2_next * ls * ls) % p * ((e1d + h2_e + de * h1_next) % p) % p, p)
    look = poly_add(la, lb, lc_, p=p)
    # :322-331
    qt = poly_add(poly_scale(polys["t_4"], z_n, p), polys["t_3"], p=p)
    qt = poly_add(poly_scale(qt, z_n, p), polys["t_2"], p=p)
    qt = poly_add(poly_scale(qt, z_n, p), polys["t_1"], p=p)
    neg_q = poly_scale(qt, -vanishing % p, p)
    lin = poly_add(gate, perm, look, neg_q, p=p)
    evals = {"a_eval": a, "b_eval": b, "c_eval": c, "d_eval": d, "left_sigma_eval": s1, "right_sigma_eval": s2, "out_sigma_eval": s3,
             "permutation_eval": z_next, "q_lookup_eval": q_lookup, "z2_next_eval": z2_next, "h1_eval": h1_e, "h1_next_eval": h1_next,
             "h2_eval": h2_e, "f_eval": f_e, "table_eval": t_e, "table_next_eval": t_next,
             "q_arith_eval": q_arith, "q_c_eval": q_c, "q_l_eval": q_l, "q_r_eval": q_r, "a_next_eval": a_n, "b_next_eval": b_n, "d_next_eval": d_n}
    return lin, evals


def multiset_compress(columns, alpha, p):
    """`MultiSet::compress` (lookup/multiset.rs:207-213) = util.rs `lc`: v_0 + alpha v_1 + ... + alpha^k v_k, row by row."""
    return [sum(col[i] * pow(alpha, k, p) for k, col in enumerate(columns)) % p for i in range(len(columns[0]))]


def combine_split(t, f):
    """`MultiSet::combine_split` (lookup/multiset.rs:131-176).  A Python dict keeps insertion order, as the reference's IndexMap
    does.  Returns (evens, odds) or raises KeyError for Error::ElementNotIndexed."""
    counters = {}
    for e in t:
        counters[e] = counters.get(e, 0) + 1
    for e in f:
        if e not in counters:
            raise KeyError("ElementNotIndexed")
        counters[e] += 1
    evens, odds, parity = [], [], 0
    for e, count in counters.items():
        evens += [e] * (count // 2)
        odds += [e] * (count // 2)
        if count % 2 == 1:
            if parity == 1:
                odds.append(e)
                parity = 0
            else:
                evens.append(e)
                parity = 1
    return evens, odds


def lookup_round2(curve: Curve, n: int, table_cols, q_lookup, wires, zeta):
    """prover.rs:228-317 up to the iffts: (compressed table, compressed query f, h_1, h_2) as evaluation vectors.
    table_cols: 4 lists of n; q_lookup: list of <= n (zero-padded, :252-254); wires: 4 lists of n (already padded, :188-192)."""
    p = curve.r
    t = multiset_compress(table_cols, zeta, p)                                    # :229-237
    q = list(q_lookup) + [0] * (n - len(q_lookup))
    f_cols = [[], [], [], []]
    for i in range(n):                                                            # :259-273
        if q[i] == 0:
            f_cols[0].append(t[0])
            for k in (1, 2, 3):
                f_cols[k].append(0)
        else:
            for k in range(4):
                f_cols[k].append(wires[k][i])
    f = multiset_compress(f_cols, zeta, p)                                        # :276
    h1, h2 = combine_split(t, f)                                                  # :295-297
    return t, f, h1, h2


def ark_window_size(n: int) -> int:
    """ark-ec 0.3 variable_base.rs: c = 3 if n < 32 else ln_without_floats(n) + 2,
    ln_without_floats(a) = log2(a) * 69 / 100 with log2 = ceil(log2)."""
    if n < 32:
        return 3
    lg = (n - 1).bit_length()
    return lg * 69 // 100 + 2


def ark_msm_adds(n: int, scalar_bits: int = 255) -> int:
    """Reference-equivalent G1 additions: W*N + 2*W*(2^c - 1)."""
    c = ark_window_size(n)
    w = -(-scalar_bits // c)
    return w * n + 2 * w * ((1 << c) - 1)


# ----------------------------------------------------------------------------- seeded inputs
def splitmix64(state: int):
    state = (state + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
    z = state
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return state, z ^ (z >> 31)


def seeded_scalars(curve: Curve, seed: int, n: int) -> list:
    """n field elements: 4 splitmix64 words (little-endian limbs) reduced mod r."""
    st = seed & 0xFFFFFFFFFFFFFFFF
    out = []
    for _ in range(n):
        v = 0
        for k in range(4):
            st, z = splitmix64(st)
            v |= z << (64 * k)
        out.append(v % curve.r)
    return out
