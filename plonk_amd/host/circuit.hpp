// Circuit construction of the reference (src/circuit.rs:29-308) on the host, in C++: the data that
// create_plonk_proof_with_transcript is called with comes from here (src/main.rs:29-100).  Plain host code, no GPU involved:
// these are O(gates) table builders; the transforms and interpolations that follow them run through KZGProverKey.
#pragma once
#include <stddef.h>
#include <map>
#include <vector>
#include "plonk_host.hpp"

namespace plonk {

struct Variable { size_t index; };                  // src/circuit.rs:31
enum class GateType { Add, Mul };                   // :35-38
struct Gate {                                       // :42-48
    GateType gate_type;
    Variable inputs[2];
    Variable output;
    size_t selector_row;
};
enum class WireColumn { A = 0, B = 1, C = 2 };      // :59-63
struct WirePosition { WireColumn col; size_t row; };  // :67-70

// src/circuit.rs:74-113.  std::map instead of HashMap: the cycles of different variables are disjoint, so the iteration order
// of the map does not show in sigma.
struct PermutationLayout {
    std::map<size_t, std::vector<WirePosition>> positions;
    // flat index 3 * row + column; identity outside the cycles
    std::vector<uint64_t> compute_sigma_mapping(size_t num_rows) const;
};

// src/circuit.rs:122-142
struct WitnessTable {
    std::vector<Fr> a_col, b_col, c_col, q_add, q_mul;
    std::vector<Fr> flatten() const;
};

// src/circuit.rs:151-278.  Values are Option<F> in the reference; add_gate unwraps its inputs, so a variable without a value
// raises the same "called `Option::unwrap()` on a `None` value" here.
class CircuitBuilder {
public:
    Variable new_variable(const Fr* value);            // nullptr = None
    Variable new_variable(uint64_t value);             // Some(Fr::from(value))
    Variable add_gate(GateType gate_type, Variable a, Variable b);
    void mark_public(Variable v) { public_inputs.push_back(v); }
    WitnessTable generate_witness_table(size_t domain_size) const;
    PermutationLayout compute_permutation_layout() const;

    std::vector<Fr> variables;
    std::vector<bool> has_value;
    std::vector<Variable> public_inputs;
    std::vector<Gate> gates;
};

// src/circuit.rs:293-308 (Circuit::from_builder); the permutation argument of the reference's struct is filled by the prover
struct Circuit {
    CircuitBuilder builder;
    WitnessTable witness;
    PermutationLayout permutation;
    Domain domain;
    static Circuit from_builder(CircuitBuilder builder, Domain domain);
};

Fr fr_from_u64(uint64_t x);
Fr fr_add(const Fr& a, const Fr& b);
Fr fr_mul(const Fr& a, const Fr& b);

}  // namespace plonk
