// Host-only code of the prover library (rng and setup sampling, sponge challenge, pairing, proof (de)serialization, transcript) in one
// executable, built with AddressSanitizer and UndefinedBehaviorSanitizer by tests/test_host_logic.py.  The two device entry points that
// setup.cpp references are stubbed: nothing here touches a GPU.
#include "../../plonk_amd/host/plonk_host.hpp"
#include <stdio.h>
#include <string.h>
#include <fstream>
#include <sstream>
using namespace plonk;
extern "C" {
int32_t plk_srs_setup_cyclic(plk_ctx*, const uint64_t*, const uint64_t*, const uint64_t*, uint32_t, uint32_t, uint32_t, uint64_t) { return 1; }
const char* plk_last_error(plk_ctx*) { return "stub"; }
}
int main(int argc, char** argv) {
    StdRng rng = test_rng();
    printf("%llu\n", (unsigned long long)rng.next_u64());
    for (int i = 0; i < 200; ++i) { rng.next_u32(); rng.next_u64(); }
    Fr a = fr_rand(rng), b = fr_rand(rng);
    G1Affine g = g1_rand(rng); G2Affine h = g2_rand(rng);
    (void)g; (void)h;
    Fr c = test_sponge_challenge();
    printf("chal %016llx\n", (unsigned long long)c.v[0]);
    printf("pairing %d\n", (int)pairing_selftest(a, b));
    // proof bytes in hex from argv[1]
    if (argc > 1) {
        std::string hex = argv[1];
        std::vector<uint8_t> bytes(hex.size() / 2);
        for (size_t i = 0; i < bytes.size(); ++i) bytes[i] = (uint8_t)strtol(hex.substr(2 * i, 2).c_str(), nullptr, 16);
        PlonkProof p = deserialize_proof(bytes.data(), bytes.size());
        std::vector<uint8_t> again = serialize_proof(p);
        printf("roundtrip %d\n", again == bytes);
        try { bytes[5] ^= 0xff; bytes[31] = 0xff; deserialize_proof(bytes.data(), bytes.size()); printf("no throw\n"); } catch (const Error& e) { printf("rejected: %s\n", e.what()); }
        try { deserialize_proof(bytes.data(), 100); } catch (const Error& e) { printf("rejected: %s\n", e.what()); }
    }
    PlonkTranscript t("plonk_proof"); Fr beta = t.challenge_beta(); printf("beta %016llx\n", (unsigned long long)beta.v[0]);
    return 0;
}
