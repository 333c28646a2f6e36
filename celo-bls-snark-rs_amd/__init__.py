"""celo-bls-snark-rs_amd — MI355X-native MSM / pairing hot path behind the celo-bls-snark-rs API.

Layout:
  csrc/      hand-written HIP (gfx950) kernels + the extern "C" boundary (include/celo_bls_amd.h)
  ffi.py     ctypes binding of build/libcelo_bls_amd.so (fails loudly if the library is missing)
  codec.py   arkworks wire/limb formats (host plumbing)
  bls.py     host-side mirror of bls-crypto's PublicKey / Signature / Batch over the C ABI
"""
__all__ = ["ffi", "codec"]
