#!/usr/bin/env python3
"""Generates celo-bls-snark-rs_amd/csrc/fp_consts.h: reduced-radix constants for the device/host
field library (see fp.h).  Everything is derived from the two moduli; nothing is copied.

Representation: an Fq element x is held as L limbs of W bits (in 32-bit words), value = x * 2^(W*L) mod p
("device Montgomery form", radix R_d = 2^(W*L)), loosely reduced (value < ~16p, limbs < ~3*2^W).
arkworks hands over / expects x * 2^(64*N) mod p in 64-bit limbs; conversion is one Montgomery
multiplication by a constant each way plus a bit repack:
   in :  dev = montmul_d(repack(ark), C_IN )   with C_IN  = 2^(2*W*L - 64*N) mod p   (raw limbs)
   out:  ark = repack(montmul_d(dev, C_OUT))   with C_OUT = 2^(64*N) mod p             (raw limbs)
"""
import os

Q377 = 0x01AE3A4617C510EAC63B05C06CA1493B1A22D9F300F5138F1EF3622FBA094800170B5D44300000008508C00000000001
Q761 = 0x122E824FB83CE0AD187C94004FAFF3EB926186A81D14688528275EF8087BE41707BA638E584E91903CEBAFF25B423048689C8ED12F9FD9071DCD3DC73EBFF2E98A116C25667A8F8160CF8AEEAF0A437E6913E6870000082F49D00000000008B
R377 = 0x12AB655E9A2CA55660B44D1E5C37B00159AA76FED00000010A11800000000001


def limbs(v, W, L):
    out = []
    for _ in range(L):
        out.append(v & ((1 << W) - 1))
        v >>= W
    assert v == 0
    return out


def redundant(v, W, L, m):
    """limbs r_i of v with r_i >= m*(2^W - 1) for i < L-1 (so that r - b has no negative limb when
    every limb of b is <= m*(2^W-1)), and the top limb reduced by the borrowed amount."""
    n = limbs(v, W, L)
    r = list(n)
    r[0] += m << W
    for i in range(1, L - 1):
        r[i] += (m << W) - m
    r[L - 1] -= m
    assert r[L - 1] > 0
    assert sum(x << (W * i) for i, x in enumerate(r)) == v
    assert all(x < (1 << 32) for x in r)
    return r


def arr(name, vals, ty="uint32_t"):
    return f"  static constexpr {ty} {name}[{len(vals)}] = {{" + ", ".join(hex(x) for x in vals) + "};\n"


def emit(name, p, W, L, N64, subs):
    Rd = 1 << (W * L)
    s = f"struct {name} {{\n"
    s += f"  static constexpr int W = {W};\n  static constexpr int L = {L};\n  static constexpr int N64 = {N64};\n"
    s += f"  static constexpr int BITS = {p.bit_length()};\n"
    s += f"  static constexpr uint32_t MASK = {hex((1 << W) - 1)};\n"
    s += f"  static constexpr uint32_t INV = {hex((-pow(p, -1, 1 << W)) % (1 << W))};  // -p^-1 mod 2^W\n"
    s += arr("P", limbs(p, W, L))
    s += arr("ONE", limbs(Rd % p, W, L))  # 1 in device Montgomery form
    s += arr("C_IN", limbs(pow(2, 2 * W * L - 64 * N64, p), W, L))
    s += arr("C_OUT", limbs(pow(2, 64 * N64, p), W, L))
    s += arr("RAW_ONE", limbs(1, W, L))
    s += arr("R2", limbs(pow(Rd, 2, p), W, L))  # canonical int -> device form: montmul(x, R2)
    for (K, m) in subs:
        s += arr(f"KP{K}_M{m}", redundant(K * p, W, L, m))
    for K in (1, 2, 4, 8, 16, 32, 64):
        s += arr(f"NP{K}", limbs(K * p, W, L))  # normalised multiples for canonical reduction
    # p in 64-bit limbs (host-side canonical compare / serialization)
    s += arr("P64", [(p >> (64 * i)) & ((1 << 64) - 1) for i in range(N64)], "uint64_t")
    s += arr("PM1_HALF64", [(((p - 1) // 2) >> (64 * i)) & ((1 << 64) - 1) for i in range(N64)], "uint64_t")
    s += "};\n\n"
    return s


def fq2_mul(a, b, p):
    return ((a[0] * b[0] - 5 * a[1] * b[1]) % p, (a[0] * b[1] + a[1] * b[0]) % p)


def fq2_pow(a, e, p):
    r = (1, 0)
    while e:
        if e & 1:
            r = fq2_mul(r, a, p)
        a = fq2_mul(a, a, p)
        e >>= 1
    return r


def emit_tower377():
    """Constants of the BLS12-377 pairing tower in DEVICE Montgomery form (x * 2^392 mod p, 28-bit limbs):
    Frobenius coefficients g_i = xi^((q^i - 1)/6), xi = u, i = 1..3 (and their powers are formed on the fly),
    1/2, and the twist coefficient B' = 1/u = (0, -1/5)."""
    p, W, L = Q377, 28, 14
    Rd = 1 << (W * L)
    dev = lambda v: limbs(v * Rd % p, W, L)
    s = "struct T377 {\n"
    for i in (1, 2, 3):
        g = fq2_pow((0, 1), (p ** i - 1) // 6, p)
        pw = (1, 0)
        for k in range(1, 6):
            pw = fq2_mul(pw, g, p)
            s += arr(f"FROB{i}_{k}_C0", dev(pw[0]))
            s += arr(f"FROB{i}_{k}_C1", dev(pw[1]))
    G2 = ((233578398248691099356572568220835526895379068987715365179118596935057653620464273615301663571204657964920925606294,
           140913150380207355837477652521042157274541796891053068589147167627541651775299824604154852141315666357241556069118),
          (63160294768292073209381361943935198908131692476676907196754037919244929611450776219210369229519898517858833747423,
           149157405641012693445398062341192467754805999074082136895788947234480009303640899064710353187729182149407503257491))
    G1 = (81937999373150964239938255573465948239988671502647976594219695644855304257327692006745978603320413799295628339695,
          241266749859715473739788878240585681733927191168601896383759122102112907357779751001206799952863815012735208165030)
    # generators (G2 recovered from crates/epoch-snark/src/epoch_block.rs:246), device Montgomery form
    s += arr("G1_GEN_X", dev(G1[0])); s += arr("G1_GEN_Y", dev(G1[1]))
    s += arr("G2_GEN_X0", dev(G2[0][0])); s += arr("G2_GEN_X1", dev(G2[0][1]))
    s += arr("G2_GEN_Y0", dev(G2[1][0])); s += arr("G2_GEN_Y1", dev(G2[1][1]))
    # psi = twist^-1 o Frobenius o twist on G2: psi(x, y) = (PSI_X conj(x), PSI_Y conj(y)), PSI_X = (-5)^((q-1)/6), PSI_Y = (-5)^((q-1)/4) in Fq;
    # psi acts on the r-torsion of E'(Fq2) as multiplication by the curve parameter x (wire.h proves it): the GLS split of the batched G2
    # MSM (msm.h k_gls_expand) uses psi^j, j = 1..3: (PSI_X^j conj^j(x), PSI_Y^j conj^j(y))
    psx, psy = pow(-5 % p, (p - 1) // 6, p), pow(-5 % p, (p - 1) // 4, p)
    for j in (1, 2, 3):
        s += arr(f"PSI_X{j}", dev(pow(psx, j, p)))
        s += arr(f"PSI_Y{j}", dev(pow(psy, j, p)))
    # beta = PSI_X^4, the cube root of unity with (beta x, y) = -[x^2](x, y) on the prime-order subgroup of E(Fq) (wire.h proves it; there it
    # is the G1 subgroup test): the GLV split of the single G1 MSM for subgroup points (msm.h k_glv_expand) uses [x^2]P = (beta x, -y)
    s += arr("BETA_GLV", dev(pow(psx, 4, p)))
    s += arr("TWO_INV", dev(pow(2, -1, p)))
    s += arr("TWIST_B_C1", dev((-pow(5, -1, p)) % p))
    s += "  static constexpr uint64_t X = 0x8508c00000000001ULL;  // BLS12-377 seed\n"
    s += "};\n\n"
    return s


def emit_tower761():
    """BW6-761 pairing constants (device Montgomery form, 28 x 28-bit limbs): Frobenius scalars h^k, h = (-4)^((q-1)/6);
    the twist coefficient is the small integer 4 (applied with additions); Miller-loop digit tables and the
    hard-part exponents R0(x), R1(x) of eprint 2020/351 Alg. 6 as used by ark-ec's bw6 engine."""
    p, W, L = Q761, 28, 28
    Rd = 1 << (W * L)
    dev = lambda v: limbs(v * Rd % p, W, L)
    x = 0x8508C00000000001
    s = "struct T761 {\n"
    h = pow(-4 % p, (p - 1) // 6, p)
    for k in range(1, 6):
        s += arr(f"FROB1_{k}", dev(pow(h, k, p)))
    s += "  static constexpr uint64_t LOOP1 = 0x8508c00000000002ULL;  // x + 1\n"
    n = x ** 3 - x ** 2 - x
    naf = []
    while n:
        if n & 1:
            d = 2 - (n & 3)
            n -= d
        else:
            d = 0
        naf.append(d)
        n >>= 1
    s += f"  static constexpr int LOOP2_LEN = {len(naf)};\n"
    s += f"  static constexpr int8_t LOOP2_NAF[{len(naf)}] = {{" + ", ".join(str(d) for d in naf) + "};  // x^3-x^2-x, little-endian signed digits\n"
    R0 = -103 * x**7 + 70 * x**6 + 269 * x**5 - 197 * x**4 - 314 * x**3 - 73 * x**2 - 263 * x - 220
    R1 = 103 * x**9 - 276 * x**8 + 77 * x**7 + 492 * x**6 - 445 * x**5 - 65 * x**4 + 452 * x**3 - 181 * x**2 + 34 * x + 229
    for nm, v in (("R0", R0), ("R1", R1)):
        mag = abs(v)
        n64 = (mag.bit_length() + 63) // 64
        s += f"  static constexpr int {nm}_BITS = {mag.bit_length()};\n  static constexpr bool {nm}_NEG = {'true' if v < 0 else 'false'};\n"
        s += arr(f"{nm}_MAG", [(mag >> (64 * i)) & ((1 << 64) - 1) for i in range(n64)], "uint64_t")
    s += "};\n\n"
    return s


def emit_sqrt_chain():
    """The exponent of the Fq square root, (t - 1) / 2 with q - 1 = 2^46 t, as a left-to-right sliding-window program over the odd
    powers a, a^3, a^5, a^7: FIRST = the leading window's digit, then one byte per squaring - 0 = square only, d = square and then
    multiply by a^d.  The exponent is a constant, so the program is the same in every lane: the four table entries live in registers
    and every branch is scalar (wire.h: wire_pow_root_exponent)."""
    q = Q377
    t = (q - 1) >> 46
    assert t & 1 and (q - 1) == t << 46
    e = (t - 1) // 2
    bits = bin(e)[2:]
    i, prog, first = 0, [], None
    while i < len(bits):
        if bits[i] == "0":
            prog.append(0)
            i += 1
            continue
        w = min(3, len(bits) - i)
        while bits[i + w - 1] == "0":
            w -= 1
        d = int(bits[i:i + w], 2)
        if first is None:
            first = d
        else:
            prog += [0] * (w - 1) + [d]
        i += w
    # check the program
    r = first
    for d in prog:
        r = 2 * r + 0
        if d:
            r += d
    # (r above tracks the exponent: squaring doubles it, a multiplication by a^d adds d)
    assert r == e, "sliding-window program does not rebuild the exponent"
    s = "struct SqrtChain377 {   // exponent (t - 1) / 2 of the Fq square root, q - 1 = 2^46 t\n"
    s += f"  static constexpr int FIRST = {first};\n  static constexpr int LEN = {len(prog)};\n"
    s += f"  static constexpr uint8_t STEP[{len(prog)}] = {{" + ", ".join(str(d) for d in prog) + "};\n};\n\n"
    return s


def main():
    out = "// GENERATED by tools/gen_consts.py — do not edit.\n#pragma once\n#include <cstdint>\n\nnamespace celo {\n\n"
    subs = [(4, 1), (8, 1), (16, 1), (32, 1), (64, 1), (8, 3), (16, 3), (32, 3)]
    out += emit("P377", Q377, 28, 14, 6, subs)
    out += emit("P761", Q761, 28, 28, 12, subs)
    # Fr of BLS12-377 (253 bits, 2-adicity 47): the field of the hash-helper proof's witness map (crates/epoch-snark/src/api/prover.rs:83-118)
    out += emit("P253", R377, 28, 10, 4, subs)
    out += emit_sqrt_chain()
    out += emit_tower377()
    out += emit_tower761()
    out += "}  // namespace celo\n"
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "celo-bls-snark-rs_amd", "csrc", "fp_consts.h")
    open(path, "w").write(out)
    print("wrote", os.path.normpath(path))


if __name__ == "__main__":
    main()
