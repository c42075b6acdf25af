#include "circuit.hpp"
#include "../csrc/host_math.hpp"
#include <string.h>

namespace plonk {

namespace hm = plk::host;

Fr fr_from_u64(uint64_t x) { const hm::el e = hm::from_u64(hm::FR, x); Fr r; memcpy(r.v, e.v, 32); return r; }
Fr fr_add(const Fr& a, const Fr& b) { const hm::el e = hm::add(hm::FR, hm::el_load(a.v), hm::el_load(b.v)); Fr r; memcpy(r.v, e.v, 32); return r; }
Fr fr_mul(const Fr& a, const Fr& b) { const hm::el e = hm::mul(hm::FR, hm::el_load(a.v), hm::el_load(b.v)); Fr r; memcpy(r.v, e.v, 32); return r; }

Variable CircuitBuilder::new_variable(const Fr* value) {
    const Variable var{variables.size()};
    variables.push_back(value ? *value : Fr{});
    has_value.push_back(value != nullptr);
    return var;
}
Variable CircuitBuilder::new_variable(uint64_t value) {
    const Fr v = fr_from_u64(value);
    return new_variable(&v);
}
Variable CircuitBuilder::add_gate(GateType gate_type, Variable a, Variable b) {
    if (a.index >= variables.size() || b.index >= variables.size()) throw Error(PLK_ERR_INDEX_OUT_OF_RANGE, "index out of bounds: no such variable");
    if (!has_value[a.index] || !has_value[b.index]) throw Error(PLK_ERR_INVALID_ARG, "called `Option::unwrap()` on a `None` value");
    const Fr out_value = gate_type == GateType::Add ? fr_add(variables[a.index], variables[b.index]) : fr_mul(variables[a.index], variables[b.index]);
    const Variable out = new_variable(&out_value);
    gates.push_back(Gate{gate_type, {a, b}, out, gates.size()});
    return out;
}
WitnessTable CircuitBuilder::generate_witness_table(size_t domain_size) const {
    WitnessTable w;
    const Fr one = fr_from_u64(1), zero{};
    for (const Gate& g : gates) {
        for (size_t v : {g.inputs[0].index, g.inputs[1].index, g.output.index})
            if (!has_value[v]) throw Error(PLK_ERR_INVALID_ARG, "called `Option::unwrap()` on a `None` value");
        w.a_col.push_back(variables[g.inputs[0].index]);
        w.b_col.push_back(variables[g.inputs[1].index]);
        w.c_col.push_back(variables[g.output.index]);
        w.q_add.push_back(g.gate_type == GateType::Add ? one : zero);
        w.q_mul.push_back(g.gate_type == GateType::Mul ? one : zero);
    }
    while (w.a_col.size() < domain_size) {       // src/circuit.rs:250-256; more gates than rows are kept, as in the reference
        w.a_col.push_back(zero); w.b_col.push_back(zero); w.c_col.push_back(zero); w.q_add.push_back(zero); w.q_mul.push_back(zero);
    }
    return w;
}
PermutationLayout CircuitBuilder::compute_permutation_layout() const {
    PermutationLayout layout;
    for (size_t row = 0; row < gates.size(); ++row) {
        const Gate& g = gates[row];
        layout.positions[g.inputs[0].index].push_back(WirePosition{WireColumn::A, row});
        layout.positions[g.inputs[1].index].push_back(WirePosition{WireColumn::B, row});
        layout.positions[g.output.index].push_back(WirePosition{WireColumn::C, row});
    }
    return layout;
}
std::vector<uint64_t> PermutationLayout::compute_sigma_mapping(size_t num_rows) const {
    std::vector<uint64_t> sigma(3 * num_rows);
    for (size_t i = 0; i < sigma.size(); ++i) sigma[i] = i;
    auto flat = [&](const WirePosition& p) {
        if (p.row >= num_rows) throw Error(PLK_ERR_INDEX_OUT_OF_RANGE, "index out of bounds: a wire position beyond num_rows");   // the HashMap lookup of :104 panics
        return 3 * p.row + (size_t)p.col;
    };
    for (const auto& kv : positions) {
        const std::vector<WirePosition>& uses = kv.second;
        for (size_t i = 0; i < uses.size(); ++i) sigma[flat(uses[i])] = flat(uses[(i + 1) % uses.size()]);
    }
    return sigma;
}
std::vector<Fr> WitnessTable::flatten() const {
    std::vector<Fr> flat;
    flat.reserve(3 * a_col.size());
    for (size_t i = 0; i < a_col.size(); ++i) { flat.push_back(a_col[i]); flat.push_back(b_col[i]); flat.push_back(c_col[i]); }
    return flat;
}
Circuit Circuit::from_builder(CircuitBuilder builder, Domain domain) {
    Circuit c{std::move(builder), {}, {}, domain};
    c.witness = c.builder.generate_witness_table(domain.size());
    c.permutation = c.builder.compute_permutation_layout();
    return c;
}

}  // namespace plonk
