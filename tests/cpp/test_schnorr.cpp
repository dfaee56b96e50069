// C++ host-side mirror exercised the way the reference's own tests exercise the crate
// (reference tests/schnorr.rs:16-66, tests/schnorr_double.rs:18-82, tests/schnorr_var_generator.rs:19-124):
// a valid signature verifies, a wrong key gives InvalidSignature, an identity key gives InvalidPoint.
// Inputs: a text file of golden vectors (one per line: scheme name expected_status hex-fields...) written by
// the pytest driver from tests/golden/verify_vectors.json.  Exit code 0 = all expectations met.
#include <cstdio>
#include <fstream>
#include <iostream>
#include <sstream>

#include "jjs_schnorr.hpp"

template <size_t N>
static std::array<uint8_t, N> unhex(const std::string& s) {
    std::array<uint8_t, N> out{};
    if (s.size() != 2 * N) throw std::runtime_error("bad hex length");
    for (size_t i = 0; i < N; ++i) out[i] = (uint8_t)std::stoul(s.substr(2 * i, 2), nullptr, 16);
    return out;
}
static int expect(const jjs::VerifyResult& got, int want, const std::string& name) {
    int g = !got ? 0 : (*got == jjs::Error::InvalidPoint ? 1 : (*got == jjs::Error::InvalidSignature ? 2 : 3));
    if (g != want) { std::printf("FAIL %s: got %d (%s), want %d\n", name.c_str(), g, got ? jjs::to_string(*got) : "Ok", want); return 1; }
    std::printf("ok   %s -> %s\n", name.c_str(), got ? jjs::to_string(*got) : "Ok(())");
    return 0;
}

int main(int argc, char** argv) {
    if (argc < 2) { std::puts("usage: test_schnorr vectors.txt"); return 2; }
    jjs::Engine engine;
    engine.reserve(JJS_SCHEME_SINGLE, JJS_FORMAT_AFFINE, 4096);      // a service pre-sizes at start-up: no later call allocates
    std::ifstream in(argv[1]);
    std::string line;
    int failures = 0, n = 0;
    std::vector<jjs::PublicKey::Item> singles;
    std::vector<int> singles_want;
    while (std::getline(in, line)) {
        std::istringstream ss(line);
        std::string scheme, name; int want;
        ss >> scheme;
        if (scheme == "wire" || scheme == "serde") continue;
        ss >> name >> want;
        std::vector<std::string> f; std::string tok;
        while (ss >> tok) f.push_back(tok);
        ++n;
        if (scheme == "single") {            // u R PK m
            jjs::PublicKey pk(unhex<64>(f[2]));
            jjs::Signature sig{unhex<32>(f[0]), unhex<64>(f[1])};
            failures += expect(pk.verify(sig, unhex<32>(f[3])), want, name);
            singles.push_back({pk.as_ref(), sig, unhex<32>(f[3])}); singles_want.push_back(want);
        } else if (scheme == "double") {     // u R Rp PK PKp m
            jjs::PublicKeyDouble pk(unhex<64>(f[3]), unhex<64>(f[4]));
            jjs::SignatureDouble sig{unhex<32>(f[0]), unhex<64>(f[1]), unhex<64>(f[2])};
            failures += expect(pk.verify(sig, unhex<32>(f[5])), want, name);
        } else if (scheme == "vargen") {     // u R PK Gen m
            jjs::PublicKeyVarGen pk(unhex<64>(f[2]), unhex<64>(f[3]));
            jjs::SignatureVarGen sig{unhex<32>(f[0]), unhex<64>(f[1])};
            failures += expect(pk.verify(sig, unhex<32>(f[4])), want, name);
        }
    }
    // the batch entry point returns the same results, and the tally counts them
    uint64_t tally[4];
    auto res = jjs::PublicKey::verify_batch(singles, tally);
    uint64_t want_tally[4] = {0, 0, 0, 0};
    for (size_t i = 0; i < res.size(); ++i) { failures += expect(res[i], singles_want[i], "batch[" + std::to_string(i) + "]"); ++want_tally[singles_want[i]]; }
    for (int k = 0; k < 4; ++k) if (tally[k] != want_tally[k]) { std::printf("FAIL tally[%d]\n", k); ++failures; }
    // extended-coordinate entry point: the same items as (u, v, 1) -- the engine divides by Z on the device
    {
        std::vector<jjs::PublicKey::ItemExtended> ext;
        for (const auto& it : singles) {
            jjs::PublicKey::ItemExtended e{};
            std::memcpy(e.pk.data(), it.pk.data(), 64); e.pk[64] = 1;
            std::memcpy(e.R.data(), it.sig.R.data(), 64); e.R[64] = 1;
            e.u = it.sig.u; e.message = it.message;
            ext.push_back(e);
        }
        auto eres = jjs::PublicKey::verify_batch_extended(ext);
        for (size_t i = 0; i < eres.size(); ++i) failures += expect(eres[i], singles_want[i], "extended[" + std::to_string(i) + "]");
    }
    // wire entry point: the reference's own serialised bytes (line "wire single <name> <status> sig pk m")
    {
        std::ifstream in2(argv[1]);
        std::vector<jjs::PublicKey::ItemBytes> wire; std::vector<int> wire_want;
        while (std::getline(in2, line)) {
            std::istringstream ss(line);
            std::string tag, scheme, name; int want;
            ss >> tag;
            if (tag != "wire") continue;
            ss >> scheme >> name >> want;
            std::string a, b, c; ss >> a >> b >> c;
            wire.push_back({unhex<32>(b), unhex<64>(a), unhex<32>(c)}); wire_want.push_back(want);
        }
        auto wres = jjs::PublicKey::verify_batch_bytes(wire);
        for (size_t i = 0; i < wres.size(); ++i) failures += expect(wres[i], wire_want[i], "wire[" + std::to_string(i) + "]");
        n += (int)wire.size();
    }
    // serde strings (line "serde single <name> <status> sig58 pk58 m-hex"): base58 -> bytes -> wire entry point,
    // and the text round-trips
    {
        std::ifstream in3(argv[1]);
        while (std::getline(in3, line)) {
            std::istringstream ss(line);
            std::string tag, scheme, name; int want;
            ss >> tag;
            if (tag != "serde") continue;
            ss >> scheme >> name >> want;
            std::string a, b, c; ss >> a >> b >> c;
            auto sig = jjs::serde::from_base58<64>(a);
            auto pk = jjs::serde::from_base58<32>(b);
            if (jjs::serde::to_base58(sig) != a || jjs::serde::to_base58(pk) != b) { std::printf("FAIL base58 round trip %s\n", name.c_str()); ++failures; }
            failures += expect(jjs::PublicKey::verify_batch_bytes({{pk, sig, unhex<32>(c)}})[0], want, "serde " + name);
            ++n;
        }
        bool threw = false;
        try { jjs::serde::from_base58<32>("0OIl"); } catch (const std::invalid_argument&) { threw = true; }
        if (!threw) { std::puts("FAIL base58 accepts characters outside the alphabet"); ++failures; }
    }
    if (jjs::PublicKey::verify_batch({}).size() != 0) { std::puts("FAIL empty batch"); ++failures; }
    std::printf("%d vectors, %d failures\n", n, failures);
    return failures ? 1 : 0;
}
