"""jubjub_schnorr_amd: batch Schnorr-on-JubJub verification on MI355X (gfx950).

Host-side mirror of the reference crate's verify interface (`PublicKey::verify`,
`PublicKeyDouble::verify`, `PublicKeyVarGen::verify`) over the C ABI of include/jjs_gpu.h.
"""
from .api import (Engine, Error, InvalidPoint, InvalidSignature, Malformed, PublicKey, PublicKeyDouble,  # noqa: F401
                  PublicKeyVarGen, Signature, SignatureDouble, SignatureVarGen, STATUS_NAMES, engine)
from ._ffi import JjsError, LIB_PATH  # noqa: F401
