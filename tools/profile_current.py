#!/usr/bin/env python3
"""Writes profiles/current.json: the profile-derived figures bench.py prints beside its live measurements, tied to the kernel library
they were measured on.

    python3 tools/profile_current.py <bench20_pmc_FETCH_SIZE.txt> <bench20_pmc_WRITE_SIZE.txt> <msm20_pmc_sq.txt> <bench20.json> <libplonk_hip.so> <tag> <proofs> <isolated launches>

Inputs are the per-kernel averages tools/pmc_summary.py prints (rocprofv3 --pmc passes, one counter block per pass) and the bench
line of the same command (pairs per proof).  FETCH_SIZE / WRITE_SIZE are in kilobytes per launch and are summed over all launches
of the counter pass (<proofs> proofs: warm-up, steps and the two proofs of the kernel breakdown, then bench.py's isolated launches); the SQ pass is taken over standalone 2^20-point MSMs (tools/sq_pass.sh), 2^20 x 16 additions per launch."""
import hashlib, json, re, sys


def counters(path, prefix):
    """{kernel name: ({counter: average per launch}, launches)} for the kernels whose name starts with prefix"""
    out = {}
    for line in open(path):   # "<kernel name, possibly with template arguments> COUNTER=value(n=launches) ..."
        m = re.match(r"(.*?) ((?:\w+=[0-9.e+]+\(n=\d+\) ?)+)$", line.strip())
        if m and m.group(1).replace(" ", "").startswith(prefix):
            vals = {k: (float(v), int(n)) for k, v, n in re.findall(r"(\w+)=([0-9.e+]+)\(n=(\d+)\)", m.group(2))}
            out[m.group(1).replace(" ", "")] = vals
    if not out:
        raise SystemExit(f"{prefix} not in {path}")
    return out


fetch_txt, write_txt, sq_txt, bench_json, lib, tag, proofs, iso = sys.argv[1:9]
proofs, iso = int(proofs), int(iso)     # proofs the counter passes ran (warm-up + steps + the two of the breakdown), isolated launches behind them
k = "k_msm_accumulate"
# all builds of the kernel together (one per window plan): bytes over all their launches, entries of all the proofs
bytes_total = 0.0
for path, ctr in ((fetch_txt, "FETCH_SIZE"), (write_txt, "WRITE_SIZE")):
    for vals in counters(path, k).values():
        bytes_total += vals[ctr][0] * 1e3 * vals[ctr][1]
bench = json.load(open(bench_json))
log_n = bench["config"]["log_gates"]
iso_windows = bench["roofline"]["isolated"]["windows"]
entries_total = proofs * bench["config"]["msm_additions_per_proof"] + iso * ((1 << log_n) + 3) * iso_windows
wide = counters(sq_txt, k + "<%d>" % (1 << (254 // iso_windows + (1 if 254 % iso_windows else 0) - 1)))
sq = {c: v[0] for c, v in next(iter(wide.values())).items()}
additions_sq = (1 << 20) * iso_windows      # the standalone launch of the counter pass: 2^20 scalars
out = {
    "tag": tag,
    "library_sha16": hashlib.sha256(open(lib, "rb").read()).hexdigest()[:16],
    "k_msm_accumulate": {
        "hbm_bytes_all_launches": round(bytes_total), "entries_all_launches": int(entries_total), "proofs": proofs,
        "traffic_bytes_per_entry": round(bytes_total / entries_total, 2),
        "valu_instructions_per_addition": round(sq["SQ_INSTS_VALU"] * 64 / additions_sq, 1),
        # the VALU pipes issue one wave instruction per four cycles; 1024 SIMDs; SQ_BUSY_CYCLES counts per shader engine (32)
        "valu_busy": round(sq["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / (sq["SQ_BUSY_CYCLES"] / 32), 3),
    },
    "sources": {"fetch": fetch_txt, "write": write_txt, "sq": sq_txt, "bench": bench_json},
}
print(json.dumps(out, indent=1))
