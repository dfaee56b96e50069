"""Host-side ingest of the reference's serde representation.

In the reference every key and signature type serialises to a JSON *string*: the base58 (Bitcoin alphabet)
text of its `to_bytes()` (reference src/serde_support.rs:21-46 for `PublicKey`; the other types repeat the
pattern), and deserialisation is base58 -> exactly `SIZE` bytes -> `from_bytes`.  This module does the first
two steps on the host and hands the bytes to the wire entry points, where the device does `from_bytes`
(canonical checks + point decompression) and the verification: an item whose bytes the reference's
`from_bytes` would reject comes back with status 3 (Malformed) instead of a deserialisation error.

Sizes (reference `Serializable<N>` impls): PublicKey 32, Signature 64, PublicKeyDouble 64, SignatureDouble 96,
PublicKeyVarGen 64, SignatureVarGen 64.
"""
from __future__ import annotations

import json
from typing import Iterable, Sequence

import numpy as np

ALPHABET = "123456789ABCDEFGHJKLMNPQRSTUVWXYZabcdefghijkmnopqrstuvwxyz"
_INDEX = {c: i for i, c in enumerate(ALPHABET)}

SIZES = {"PublicKey": 32, "Signature": 64, "PublicKeyDouble": 64, "SignatureDouble": 96,
         "PublicKeyVarGen": 64, "SignatureVarGen": 64}
SCHEME_TYPES = {"single": ("Signature", "PublicKey"), "double": ("SignatureDouble", "PublicKeyDouble"),
                "vargen": ("SignatureVarGen", "PublicKeyVarGen")}


class SerdeError(ValueError):
    """What the reference's `Deserialize` impls report as `SerdeError::custom` / `invalid_length`."""

    def __init__(self, index: int, reason: str):
        super().__init__(f"item {index}: {reason}")
        self.index, self.reason = index, reason


def b58decode(text: str) -> bytes:
    """Base58 text -> bytes (leading '1' characters are leading zero bytes)."""
    value = 0
    for ch in text:
        digit = _INDEX.get(ch)
        if digit is None:
            raise ValueError(f"invalid base58 character {ch!r}")
        value = value * 58 + digit
    zeros = len(text) - len(text.lstrip("1"))
    body = value.to_bytes((value.bit_length() + 7) // 8, "big") if value else b""
    return b"\x00" * zeros + body


def b58encode(data: bytes) -> str:
    data = bytes(data)
    value = int.from_bytes(data, "big")
    out = []
    while value:
        value, rem = divmod(value, 58)
        out.append(ALPHABET[rem])
    zeros = len(data) - len(data.lstrip(b"\x00"))
    return "1" * zeros + "".join(reversed(out))


def decode_column(strings: Iterable[str], type_name: str) -> np.ndarray:
    """Base58 strings of one reference type -> (n, SIZE) uint8 array; a bad character or a wrong decoded
    length raises SerdeError naming the item, as the reference's deserialiser fails on it."""
    size = SIZES[type_name]
    rows = []
    for i, s in enumerate(strings):
        try:
            raw = b58decode(s)
        except ValueError as e:
            raise SerdeError(i, str(e)) from None
        if len(raw) != size:
            raise SerdeError(i, f"invalid length {len(raw)}, expected {size}")
        rows.append(np.frombuffer(raw, np.uint8))
    return np.stack(rows) if rows else np.zeros((0, size), np.uint8)


def encode_column(rows: np.ndarray) -> list:
    return [b58encode(bytes(r)) for r in np.asarray(rows, np.uint8)]


def verify_strings(engine, scheme: str, signatures: Sequence[str], public_keys: Sequence[str], messages):
    """Batch verify from the serde strings.  `messages`: (n, 32) uint8, canonical little-endian BlsScalars.
    Returns (status, tally) as numpy arrays (blocking host-buffer call)."""
    sig_type, pk_type = SCHEME_TYPES[scheme]
    sig = decode_column(signatures, sig_type)
    pk = decode_column(public_keys, pk_type)
    m = np.ascontiguousarray(np.asarray(messages, np.uint8).reshape(-1, 32))
    if not (len(sig) == len(pk) == len(m)):
        raise ValueError("signatures, public keys and messages must have the same length")
    return engine.verify_wire(scheme, sig, pk, m)


def verify_json(engine, scheme: str, document: str):
    """`document`: JSON array of objects {"signature": str, "public_key": str, "message": hex of 32 LE bytes}."""
    items = json.loads(document)
    msgs = np.stack([np.frombuffer(bytes.fromhex(it["message"]), np.uint8) for it in items]) if items else np.zeros((0, 32), np.uint8)
    return verify_strings(engine, scheme, [it["signature"] for it in items], [it["public_key"] for it in items], msgs)
