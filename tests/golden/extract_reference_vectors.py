#!/usr/bin/env python3
"""Extracts the reference's own golden VECTORS (hex data literals held by its tests)
into tests/golden/reference_vectors.json.  Run in the build container only
(/root/reference does not exist on the GPU box); the JSON it writes is committed.

Only data is taken — hex strings and the scalar parameters of the test that
holds them — never source text.  Sources (paths relative to /root/reference):
  crates/bls-crypto/src/hash_to_curve/mod.rs:412-513   hash-to-G1/G2 outputs (compressed points)
  crates/bls-snark-sys/src/snark/mod.rs:52-119         Groth16/BW6-761 accept vector
  crates/epoch-snark/src/epoch_block.rs:243-246        epoch encodings (embed the G2 generator)
  crates/bls-crypto/src/hashers/composite.rs:105-190   CompositeHasher crh / xof / hash outputs (XorShift-seeded inputs)
"""
import json, os, re, sys

REF = "/root/reference/crates"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_vectors.json")


def hex_lists_by_fn(path):
    """map test-fn name -> list of hex string literals inside its expected_hashes vec."""
    txt = open(path).read()
    out = {}
    for m in re.finditer(r"fn (test_hash_to_curve\w*)\(\)", txt):
        body = txt[m.end():]
        end = body.find("].into_iter()")
        out[m.group(1)] = re.findall(r'"([0-9a-f]{96,})"', body[:end])
    return out


def consts(path):
    txt = open(path).read()
    return {m.group(1): m.group(2) for m in re.finditer(r'(?:const|static) (\w+): &str\s*=\s*"([0-9a-f]*)"', txt)}


def main():
    h = hex_lists_by_fn(f"{REF}/bls-crypto/src/hash_to_curve/mod.rs")
    # the file holds two functions named test_hash_to_curve_g1 (compat / non-compat modules);
    # regex dict keeps the last; re-scan in order to keep both
    txt = open(f"{REF}/bls-crypto/src/hash_to_curve/mod.rs").read()
    ordered = []
    for m in re.finditer(r"fn (test_hash_to_curve\w*)\(\)", txt):
        body = txt[m.end():]
        end = body.find("].into_iter()")
        ordered.append((m.group(1), re.findall(r'"([0-9a-f]{96,})"', body[:end])))
    names = ["g1_compat_before_donut_dup", "g1_compat", "g1_compat_cip22", "g1_noncompat", "g2_noncompat"]
    vec = {}
    k = 0
    for name, lst in ordered:
        if not lst:
            continue
        vec[names[k] if k < len(names) else f"extra{k}"] = {"fn": name, "points": lst}
        k += 1
    g = consts(f"{REF}/bls-snark-sys/src/snark/mod.rs")
    e = consts(f"{REF}/epoch-snark/src/epoch_block.rs")
    # DirectHasher vectors with fully specified inputs (crates/bls-crypto/src/hashers/direct.rs:88-96, 149-172)
    dtxt = open(f"{REF}/bls-crypto/src/hashers/direct.rs").read()
    crh_empty = re.search(r'fn test_crh_empty.*?"([0-9a-f]{64})"', dtxt, re.S).group(1)
    tv_block = dtxt[dtxt.index("fn test_blake2s_test_vectors"):]
    strs = re.findall(r'"([0-9a-fA-F]+)"', tv_block)
    tv = list(zip(strs[0::2], strs[1::2]))
    # CompositeHasher vectors (crates/bls-crypto/src/hashers/composite.rs:105-190): per test fn, the first byte of the XorShift
    # seed (the other 15 are shared with RNG_SEED), the message length, the output length and the expected hex
    ctxt = open(f"{REF}/bls-crypto/src/hashers/composite.rs").read()
    comp = {}
    for m in re.finditer(r"fn (test_(?:crh|xof|hash)_\w+)\(\)", ctxt):
        body = ctxt[m.end():]
        body = body[:body.index("assert_eq!") + 4000]
        exp = re.search(r'assert_eq!\(hex::encode\(\w+\),\s*"([0-9a-f]+)"', body).group(1)
        seed = re.search(r"from_seed\(\[\s*0x([0-9a-f]{2})", body[:body.index("assert_eq!")])
        mlen = re.search(r"vec!\[0; ([0-9 */]+)\]", body[:body.index("assert_eq!")])
        nb = re.search(r"\.(?:xof|hash)\(b\"ULforxof\", &\w+, (\d+)\)", body[:body.index("assert_eq!")])
        comp[m.group(1)] = {"seed0": int(seed.group(1), 16) if seed else None, "msg_len": eval(mlen.group(1).replace("/", "//")) if mlen else 0,
                            "out_bytes": int(nb.group(1)) if nb else None, "expected": exp}
    # DirectHasher vectors with XorShift-seeded inputs (crates/bls-crypto/src/hashers/direct.rs:99-147), same shape as the composite ones
    direct_x = {}
    for m in re.finditer(r"fn (test_(?:crh_random|xof_random_96|hash_random))\(\)", dtxt):
        body = dtxt[m.end():]
        head = body[:body.index("assert_eq!")]
        exp = re.search(r'"([0-9a-f]{64,})"', body[body.index("assert_eq!"):]).group(1)
        seed = re.search(r"from_seed\(\[\s*0x([0-9a-f]{2})", head)
        mlen = re.search(r"vec!\[0; ([0-9 */]+)\]", head)
        nb = re.search(r"\.(?:xof|hash)\(b\"ULforxof\", &\w+, (\d+)\)", head)
        direct_x[m.group(1)] = {"seed0": int(seed.group(1), 16), "msg_len": eval(mlen.group(1).replace("/", "//")),
                                "out_bytes": int(nb.group(1)) if nb else None, "expected": exp}
    out = {
        "direct_hasher_random": direct_x,
        "composite_hasher": comp,
        "direct_hasher": {"crh_empty_xof96": crh_empty, "blake2x_hash_vectors": [{"input": a, "output": b} for a, b in tv]},
        "_source": "celo-org/celo-bls-snark-rs test vectors (data literals only); see extract_reference_vectors.py",
        "hash_to_curve": vec,
        "groth16_bw6_761": {
            "vk": g["ENTROPY_VK"], "proof": g["ENTROPY_PROOF"],
            "first_pubkeys": g["ENTROPY_FIRST_PUBKEYS"], "last_pubkeys": g["ENTROPY_LAST_PUBKEYS"],
            "first_epoch_entropy": g["FIRST_EPOCH_ENTROPY"], "first_parent_entropy": g["FIRST_PARENT_ENTROPY"],
            "last_epoch_entropy": g["LAST_EPOCH_ENTROPY"], "last_parent_entropy": g["LAST_PARENT_ENTROPY"],
            "first": {"index": 0, "round": 0, "maximum_non_signers": 1, "pubkeys_num": 4, "maximum_validators": 4},
            "last": {"index": 2, "round": 0, "maximum_non_signers": 1, "pubkeys_num": 4, "maximum_validators": 4},
            "expected": True,
        },
        "epoch_encoding": {k: v for k, v in e.items() if k.startswith("EXPECTED_")},
    }
    json.dump(out, open(OUT, "w"), indent=1)
    print("wrote", OUT, {k: len(v["points"]) for k, v in vec.items()})


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("reference not present (build container only)")
    main()
