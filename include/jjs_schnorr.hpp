// Host-side mirror (C++17, header only) of the reference crate's verify interface, over the C ABI of
// jjs_gpu.h.  The reference is Rust and this image has no Rust toolchain, so this header plays the role
// of the shim in INTEGRATION.md: same type names, same method names, same argument meaning and the same
// error classes as
//   PublicKey::verify        /root/reference/src/keys/public.rs:114-135
//   PublicKeyDouble::verify  src/keys/public/double.rs:86-117
//   PublicKeyVarGen::verify  src/keys/public/var_gen.rs:107-133
//   Error                    src/error.rs:13-26
//   to_bytes / from_bytes    src/signatures.rs:101-119, src/keys/public.rs:80-94 (wire entry points)
// plus the batch entry points a GPU-backed crate would add (`verify_batch`, `verify_batch_bytes`).
// Points are held as affine canonical bytes (u || v): what `to_hash_inputs()` yields; `verify_batch_extended` takes
// them as the Rust `JubJubExtended` holds them (U || V || Z, normalised on the device: no host arithmetic).
#pragma once
#include <array>
#include <cstdint>
#include <cstring>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

#include "jjs_gpu.h"

namespace jjs {

using Scalar = std::array<uint8_t, 32>;     // JubJubScalar / BlsScalar: canonical little-endian
using AffinePoint = std::array<uint8_t, 64>;  // u || v
using ExtendedPoint = std::array<uint8_t, 96>;  // U || V || Z, affine point (U/Z, V/Z): get_u / get_v / get_z of JubJubExtended
using BlsScalar = Scalar;
using JubJubScalar = Scalar;

// reference src/error.rs:13-26 (the variants reachable from verify) + the engine's own failures
enum class Error { InvalidSignature, InvalidPoint, BytesError, Engine };

inline const char* to_string(Error e) {
    switch (e) {
    case Error::InvalidSignature: return "Invalid Signature";   // src/error.rs:40-42
    case Error::InvalidPoint: return "Invalid Point";           // src/error.rs:43-45
    case Error::BytesError: return "InvalidData";
    default: return "engine error";
    }
}

// Result<(), Error>: empty optional == Ok(())
using VerifyResult = std::optional<Error>;

inline VerifyResult from_status(uint8_t s) {
    switch (s) {
    case JJS_STATUS_OK: return std::nullopt;
    case JJS_STATUS_INVALID_POINT: return Error::InvalidPoint;
    case JJS_STATUS_INVALID_SIGNATURE: return Error::InvalidSignature;
    default: return Error::BytesError;   // 3: an encoding the Rust from_bytes would have rejected
    }
}

struct EngineError : std::runtime_error {
    int code;
    EngineError(int c, const char* what) : std::runtime_error(std::string(what) + ": " + jjs_last_error()), code(c) {}
};

// RAII handle on the process-wide engine: device_count 1 = the current HIP device, k = devices 0..k-1,
// 0 = every visible device (batches of the host-buffer calls are then sharded across them).
class Engine {
  public:
    explicit Engine(int device_count = 1) { int rc = jjs_init(device_count); if (rc != JJS_OK) throw EngineError(rc, "jjs_init"); }
    int device_count() const { return jjs_device_count(); }
    // Pre-sizes the engine for verify_batch calls of at most `max_items` items of one scheme (JJS_SCHEME_*) and input format
    // (JJS_FORMAT_*): afterwards no such call allocates (include/jjs_gpu.h jjs_reserve).  The batch methods below are the
    // blocking host-buffer calls, hence host_buffers = 1.
    void reserve(int scheme, int format, size_t max_items) {
        int rc = jjs_reserve(scheme, format, max_items, 1);
        if (rc != JJS_OK) throw EngineError(rc, "jjs_reserve");
    }
    // Waits for the device, then gives back what growth has retired and the key-table pools (jjs_trim).
    void trim() { int rc = jjs_trim(); if (rc != JJS_OK) throw EngineError(rc, "jjs_trim"); }
    ~Engine() { jjs_shutdown(); }
    Engine(const Engine&) = delete;
    Engine& operator=(const Engine&) = delete;
};

namespace detail {
// 16-byte aligned SoA staging buffer
class Soa {
  public:
    Soa(size_t items, size_t width) : width_(width), store_((items * width + 15) / 16 + 1) {}
    uint8_t* data() { return reinterpret_cast<uint8_t*>(store_.data()); }
    const uint8_t* data() const { return reinterpret_cast<const uint8_t*>(store_.data()); }
    template <size_t N>
    void put(size_t i, size_t off, const std::array<uint8_t, N>& v) { std::memcpy(data() + i * width_ + off, v.data(), N); }
  private:
    struct alignas(16) Block { uint8_t b[16]; };
    size_t width_;
    std::vector<Block> store_;
};
inline std::vector<VerifyResult> results(const std::vector<uint8_t>& status) {
    std::vector<VerifyResult> out;
    out.reserve(status.size());
    for (uint8_t s : status) out.push_back(from_status(s));
    return out;
}
}  // namespace detail

// The reference's serde form (src/serde_support.rs:21-46 and the same pattern for every type): a JSON
// string holding the base58 (Bitcoin alphabet) text of `to_bytes()`.  Decoding fails, like the reference's
// deserialiser, on a character outside the alphabet or a decoded length other than N.
namespace serde {
inline const char* alphabet() { return "123456789ABCDEFGHJKLMNPQRSTUVWXYZabcdefghijkmnopqrstuvwxyz"; }
template <size_t N>
inline std::array<uint8_t, N> from_base58(const std::string& text) {
    std::vector<uint8_t> num;                       // big-endian base-256 digits, most significant first
    size_t zeros = 0;
    while (zeros < text.size() && text[zeros] == '1') ++zeros;
    for (char ch : text) {
        const char* pos = std::strchr(alphabet(), ch);
        if (!pos || ch == 0) throw std::invalid_argument("invalid base58 character");
        unsigned carry = (unsigned)(pos - alphabet());
        for (size_t i = num.size(); i-- > 0;) { carry += 58u * num[i]; num[i] = (uint8_t)carry; carry >>= 8; }
        while (carry) { num.insert(num.begin(), (uint8_t)carry); carry >>= 8; }
    }
    size_t lead = 0;
    while (lead < num.size() && num[lead] == 0) ++lead;
    if (zeros + (num.size() - lead) != N) throw std::invalid_argument("invalid length");
    std::array<uint8_t, N> out{};
    std::copy(num.begin() + lead, num.end(), out.begin() + zeros);
    return out;
}
template <size_t N>
inline std::string to_base58(const std::array<uint8_t, N>& bytes) {
    std::vector<uint8_t> digits;                    // base-58 digits, least significant first
    size_t zeros = 0;
    while (zeros < N && bytes[zeros] == 0) ++zeros;
    for (uint8_t b : bytes) {
        unsigned carry = b;
        for (auto& d : digits) { carry += 256u * d; d = (uint8_t)(carry % 58u); carry /= 58u; }
        while (carry) { digits.push_back((uint8_t)(carry % 58u)); carry /= 58u; }
    }
    std::string out(zeros, '1');
    for (size_t i = digits.size(); i-- > 0;) out.push_back(alphabet()[digits[i]]);
    return out;
}
}  // namespace serde

// `Signature { u, R }` (reference src/signatures.rs:62-65)
struct Signature {
    JubJubScalar u;
    AffinePoint R;
};
// `SignatureDouble { u, R, R_prime }` (reference src/signatures/double.rs:66-70)
struct SignatureDouble {
    JubJubScalar u;
    AffinePoint R, R_prime;
};
// `SignatureVarGen { u, R }` (reference src/signatures/var_gen.rs)
struct SignatureVarGen {
    JubJubScalar u;
    AffinePoint R;
};

// `PublicKey(JubJubExtended)` (reference src/keys/public.rs:52)
class PublicKey {
  public:
    explicit PublicKey(const AffinePoint& p) : point_(p) {}
    const AffinePoint& as_ref() const { return point_; }

    struct Item { AffinePoint pk; Signature sig; BlsScalar message; };   // (key point, signature, message)
    // PublicKey::verify (reference src/keys/public.rs:114-135)
    VerifyResult verify(const Signature& sig, const BlsScalar& message) const {
        return verify_batch({Item{point_, sig, message}})[0];
    }
    static std::vector<VerifyResult> verify_batch(const std::vector<Item>& items, uint64_t tally[4] = nullptr) {
        const size_t n = items.size();
        detail::Soa u(n, 32), r(n, 64), pk(n, 64), m(n, 32);
        for (size_t i = 0; i < n; ++i) {
            u.put(i, 0, items[i].sig.u); r.put(i, 0, items[i].sig.R);
            pk.put(i, 0, items[i].pk); m.put(i, 0, items[i].message);
        }
        std::vector<uint8_t> status(n);
        uint64_t t[4];
        int rc = jjs_verify_single(u.data(), r.data(), pk.data(), m.data(), n, status.data(), t);
        if (rc != JJS_OK) throw EngineError(rc, "jjs_verify_single");
        if (tally) std::memcpy(tally, t, sizeof(t));
        return detail::results(status);
    }
    // The same with every point in extended coordinates: what `PublicKey::verify(&self, &Signature, BlsScalar)` holds
    // (reference src/keys/public.rs:114-118); the engine normalises on the device (jjs_verify_single_ext).
    struct ItemExtended { ExtendedPoint pk; JubJubScalar u; ExtendedPoint R; BlsScalar message; };
    static std::vector<VerifyResult> verify_batch_extended(const std::vector<ItemExtended>& items) {
        const size_t n = items.size();
        detail::Soa u(n, 32), r(n, 96), pk(n, 96), m(n, 32);
        for (size_t i = 0; i < n; ++i) {
            u.put(i, 0, items[i].u); r.put(i, 0, items[i].R); pk.put(i, 0, items[i].pk); m.put(i, 0, items[i].message);
        }
        std::vector<uint8_t> status(n);
        int rc = jjs_verify_single_ext(u.data(), r.data(), pk.data(), m.data(), n, status.data(), nullptr);
        if (rc != JJS_OK) throw EngineError(rc, "jjs_verify_single_ext");
        return detail::results(status);
    }
    // Batch verify straight from the reference's wire formats (`Signature::to_bytes` 64 B = u || R,
    // `PublicKey::to_bytes` 32 B; reference src/signatures.rs:101-119, src/keys/public.rs:80-94): points are
    // decompressed on the device; an undecodable item yields Error::BytesError, as `from_bytes` would.
    struct ItemBytes { std::array<uint8_t, 32> pk; std::array<uint8_t, 64> sig; BlsScalar message; };
    static std::vector<VerifyResult> verify_batch_bytes(const std::vector<ItemBytes>& items) {
        const size_t n = items.size();
        detail::Soa sig(n, 64), pk(n, 32), m(n, 32);
        for (size_t i = 0; i < n; ++i) { sig.put(i, 0, items[i].sig); pk.put(i, 0, items[i].pk); m.put(i, 0, items[i].message); }
        std::vector<uint8_t> status(n);
        int rc = jjs_verify_single_wire(sig.data(), pk.data(), m.data(), n, status.data(), nullptr);
        if (rc != JJS_OK) throw EngineError(rc, "jjs_verify_single_wire");
        return detail::results(status);
    }
  private:
    AffinePoint point_;
};

// `PublicKeyDouble(pk, pk_prime)` (reference src/keys/public/double.rs:45)
class PublicKeyDouble {
  public:
    PublicKeyDouble(const AffinePoint& pk, const AffinePoint& pk_prime) : pk_(pk), pk_prime_(pk_prime) {}
    const AffinePoint& pk() const { return pk_; }
    const AffinePoint& pk_prime() const { return pk_prime_; }

    struct Item { AffinePoint pk, pk_prime; SignatureDouble sig; BlsScalar message; };
    // PublicKeyDouble::verify (reference src/keys/public/double.rs:86-117)
    VerifyResult verify(const SignatureDouble& sig, const BlsScalar& message) const {
        return verify_batch({Item{pk_, pk_prime_, sig, message}})[0];
    }
    static std::vector<VerifyResult> verify_batch(const std::vector<Item>& items) {
        const size_t n = items.size();
        detail::Soa u(n, 32), r(n, 64), rp(n, 64), pk(n, 64), pkp(n, 64), m(n, 32);
        for (size_t i = 0; i < n; ++i) {
            u.put(i, 0, items[i].sig.u); r.put(i, 0, items[i].sig.R); rp.put(i, 0, items[i].sig.R_prime);
            pk.put(i, 0, items[i].pk); pkp.put(i, 0, items[i].pk_prime); m.put(i, 0, items[i].message);
        }
        std::vector<uint8_t> status(n);
        int rc = jjs_verify_double(u.data(), r.data(), rp.data(), pk.data(), pkp.data(), m.data(), n, status.data(), nullptr);
        if (rc != JJS_OK) throw EngineError(rc, "jjs_verify_double");
        return detail::results(status);
    }
  private:
    AffinePoint pk_, pk_prime_;
};

// `PublicKeyVarGen { pk, generator }` (reference src/keys/public/var_gen.rs:40-43)
class PublicKeyVarGen {
  public:
    PublicKeyVarGen(const AffinePoint& pk, const AffinePoint& generator) : pk_(pk), generator_(generator) {}
    const AffinePoint& public_key() const { return pk_; }
    const AffinePoint& generator() const { return generator_; }

    struct Item { AffinePoint pk, generator; SignatureVarGen sig; BlsScalar message; };
    // PublicKeyVarGen::verify (reference src/keys/public/var_gen.rs:107-133)
    VerifyResult verify(const SignatureVarGen& sig, const BlsScalar& message) const {
        return verify_batch({Item{pk_, generator_, sig, message}})[0];
    }
    static std::vector<VerifyResult> verify_batch(const std::vector<Item>& items) {
        const size_t n = items.size();
        detail::Soa u(n, 32), r(n, 64), pk(n, 64), gen(n, 64), m(n, 32);
        for (size_t i = 0; i < n; ++i) {
            u.put(i, 0, items[i].sig.u); r.put(i, 0, items[i].sig.R);
            pk.put(i, 0, items[i].pk); gen.put(i, 0, items[i].generator); m.put(i, 0, items[i].message);
        }
        std::vector<uint8_t> status(n);
        int rc = jjs_verify_vargen(u.data(), r.data(), pk.data(), gen.data(), m.data(), n, status.data(), nullptr);
        if (rc != JJS_OK) throw EngineError(rc, "jjs_verify_vargen");
        return detail::results(status);
    }
  private:
    AffinePoint pk_, generator_;
};

}  // namespace jjs
