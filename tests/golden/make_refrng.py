"""Writes tests/golden/refrng.json: the draws of the reference's driver (src/main.rs:25-221 with DETERMINISTIC_TEST_RNG=1) as the CPU
restatement oracle/refrng.py produces them.  Run from the repository root: python tests/golden/make_refrng.py

The two external known answers in the file (`rfc8439_block`, `rand08_first_u64`) are published values, not outputs of this code;
everything else is this build's own restatement and guards against regressions only."""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from oracle import refrng as R  # noqa: E402

H = lambda x: hex(x)  # noqa: E731


def main():
    rng = R.test_rng()
    words = [rng.next_u64() for _ in range(40)]          # crosses the first 64-word buffer
    d = R.main_draws()
    s = d["srs"]
    out = {
        "rfc8439_block": {"key": bytes(range(32)).hex(), "counter": 1 | (0x09000000 << 32), "stream": 0x4A000000, "rounds": 20,
                          "first_word": "0xe4e7f110", "last_word": "0x4e3c50a2"},
        "rand08_first_u64": 10719222850664546238,
        "next_u64": [H(w) for w in words],
        "discarded": [H(x) for x in d["discarded"]],
        "beta": H(s["beta"]),
        "g": [H(c) for c in s["g"]],
        "gamma_g": [H(c) for c in s["gamma_g"]],
        "h": [[H(c) for c in s["h"][0]], [H(c) for c in s["h"][1]]],
        "beta_h": [[H(c) for c in s["beta_h"][0]], [H(c) for c in s["beta_h"][1]]],
        "powers_of_g_last": [H(c) for c in s["powers_of_g"][32]],
        "powers_of_gamma_g": [[H(c) for c in p] for p in s["powers_of_gamma_g"]],
        "blinds": [H(x) for x in d["blinds"]],
        "opening_challenge": H(d["challenge"]),
    }
    # the proof of the example circuit that the replay of src/main.rs produces from these draws (CPU oracle)
    import numpy as np
    from oracle import oracle as orc
    from oracle import pymodel as pm
    from plonk_amd import synth
    inp = pm.prepare_inputs(*pm.example_circuit())
    tables = synth.circuit_tables(*synth.example_circuit())
    polys = [synth.trim(orc.fr_from_ints(inp[k])) for k in ("a", "b", "c", "q_add", "q_mul", "s_id", "s_sigma")]
    rc, proof, _, _ = orc.prove(orc.points_from_ints(s["powers_of_g"]), orc.points_from_ints(s["powers_of_gamma_g"]), polys, tables["witness_flat"], tables["sigma"],
                                tables["log_n"], orc.fr_from_ints(d["blinds"]), orc.fr_from_ints([d["challenge"]])[0])
    assert rc == 0
    out["replay_proof_hex"] = proof.hex()
    path = os.path.join(os.path.dirname(__file__), "refrng.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
