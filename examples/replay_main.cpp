// The reference's driver (src/main.rs:24-240) on the MI355X backend, line for line where it matters: the 4-gate example circuit,
// sigma, the permutation and selector polynomials, PCS::setup from test_rng (deterministic branch), the proof, the verification.
// Prints the 1170 proof bytes in hex and the verdict.   usage: replay_main [device]
#include <stdio.h>
#include <stdlib.h>
#include "../plonk_amd/host/circuit.hpp"
#include "../plonk_amd/host/plonk_host.hpp"

using namespace plonk;

int main(int argc, char** argv) {
    try {
        StdRng rng = test_rng();                                             // :25
        const uint64_t max_degree = 32;                                      // :26
        CircuitBuilder builder;                                              // :29-39
        const Variable a = builder.new_variable(3);
        const Variable b = builder.new_variable(4);
        const Variable c = builder.add_gate(GateType::Add, a, b);
        const Variable d = builder.new_variable(2);
        const Variable e = builder.add_gate(GateType::Mul, c, d);
        const Variable f = builder.new_variable(5);
        const Variable g = builder.add_gate(GateType::Add, e, f);
        builder.add_gate(GateType::Mul, g, a);
        const Domain domain = Domain::with_at_least(builder.gates.size());   // :43-45
        const Domain permutation_domain = Domain::with_at_least(3 * domain.size());   // :46-47
        const Circuit circuit = Circuit::from_builder(std::move(builder), domain);     // :50

        std::vector<uint64_t> sigma_for_grand_product = circuit.permutation.compute_sigma_mapping(domain.size());   // :61-64
        std::vector<uint64_t> sigma_for_interpolation = sigma_for_grand_product;                                      // :79-82
        while (sigma_for_interpolation.size() < permutation_domain.size()) sigma_for_interpolation.push_back(sigma_for_interpolation.size());

        KZGProverKey pk(argc > 1 ? atoi(argv[1]) : 0);
        const auto perm = interpolate_permutation_polynomials(pk, sigma_for_interpolation, permutation_domain);       // :84
        const std::vector<Fr> witness_flat = circuit.witness.flatten();                                               // :96-100

        for (size_t row = 0; row < witness_flat.size() / 3; ++row)           // :126-137, placeholders drawn and dropped
            for (int wire = 0; wire < 3; ++wire) { fr_rand(rng); fr_rand(rng); }
        fr_rand(rng); fr_rand(rng); fr_rand(rng);                            // :174-176

        const KZGVerifierKey vk = pk.setup(max_degree, rng);                 // :182-183
        pk.opening_challenge = test_sponge_challenge();                      // what PCS::open squeezes, src/prover.rs:285-303

        const DensePolynomial a_poly = DensePolynomial::from_coefficients_vec(circuit.witness.a_col);               // :187-191
        const DensePolynomial b_poly = DensePolynomial::from_coefficients_vec(circuit.witness.b_col);
        const DensePolynomial c_poly = DensePolynomial::from_coefficients_vec(circuit.witness.c_col);
        const DensePolynomial q_add_poly = interpolate_selector(pk, circuit.witness.q_add);
        const DensePolynomial q_mul_poly = interpolate_selector(pk, circuit.witness.q_mul);
        std::vector<Fr> public_inputs;                                       // :203-206
        for (const Variable& v : circuit.builder.public_inputs) public_inputs.push_back(circuit.builder.variables[v.index]);

        Rng draw = as_rng(rng);
        const PlonkProof proof = create_plonk_proof_with_transcript(pk, a_poly, b_poly, c_poly, q_add_poly, q_mul_poly, perm.first, perm.second, public_inputs,
                                                                    witness_flat, sigma_for_grand_product, domain, draw);   // :211-221
        const bool is_valid = verify_plonk_proof_with_transcript(vk, proof, public_inputs, pk.opening_challenge);          // :226-230

        const std::vector<uint8_t> bytes = serialize_proof(proof);
        printf("proof ");
        for (uint8_t x : bytes) printf("%02x", x);
        printf("\nCircuit gates: %zu\nDomain size: %llu\nPublic inputs: %zu\nProof verified: %s\n", circuit.builder.gates.size(),
               (unsigned long long)domain.size(), public_inputs.size(), is_valid ? "true" : "false");
        return is_valid ? 0 : 1;
    } catch (const Error& err) {
        fprintf(stderr, "error %d: %s\n", err.status, err.what());
        return 2;
    }
}
