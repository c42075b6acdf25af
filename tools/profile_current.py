#!/usr/bin/env python3
"""Writes profiles/current.json: the profile-derived figures bench.py prints beside its live measurements, tied to the kernel library
they were measured on.

    python3 tools/profile_current.py <bench20_pmc_FETCH_SIZE.txt> <bench20_pmc_WRITE_SIZE.txt> <msm20_pmc_sq.txt> <bench20.json> <libplonk_hip.so> <tag>

Inputs are the per-kernel averages tools/pmc_summary.py prints (rocprofv3 --pmc passes, one counter block per pass) and the bench
line of the same command (for the average number of entries per k_msm_accumulate launch).  FETCH_SIZE / WRITE_SIZE are in
kilobytes per launch; the SQ pass is taken over standalone 2^20-point MSMs (tools/sq_pass.sh), 2^20 x 16 additions per launch."""
import hashlib, json, re, sys


def counters(path, kernel):
    for line in open(path):
        name, _, rest = line.partition(" ")
        if name == kernel:
            return {k: float(v) for k, v in re.findall(r"(\w+)=([0-9.e+]+)\(", rest)}
    raise SystemExit(f"{kernel} not in {path}")


fetch_txt, write_txt, sq_txt, bench_json, lib, tag = sys.argv[1:7]
k = "k_msm_accumulate"
fetch_kb = counters(fetch_txt, k)["FETCH_SIZE"]
write_kb = counters(write_txt, k)["WRITE_SIZE"]
sq = counters(sq_txt, k)
bench = json.load(open(bench_json))
windows = bench["roofline"].get("windows", 16)
entries = bench["roofline"]["per_launch"]["pairs_per_launch_avg"] * windows if "per_launch" in bench["roofline"] else bench["roofline"]["pairs_per_launch_avg"] * windows
additions_sq = (1 << 20) * 16
out = {
    "tag": tag,
    "library_sha16": hashlib.sha256(open(lib, "rb").read()).hexdigest()[:16],
    "k_msm_accumulate": {
        "fetch_bytes_per_launch": round(fetch_kb * 1e3), "write_bytes_per_launch": round(write_kb * 1e3),
        "entries_per_launch_avg": int(entries),
        "traffic_bytes_per_entry": round((fetch_kb + write_kb) * 1e3 / entries, 2),
        "valu_instructions_per_addition": round(sq["SQ_INSTS_VALU"] * 64 / additions_sq, 1),
        # the VALU pipes issue one wave instruction per four cycles; 1024 SIMDs; SQ_BUSY_CYCLES counts per shader engine (32)
        "valu_busy": round(sq["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / (sq["SQ_BUSY_CYCLES"] / 32), 3),
    },
    "sources": {"fetch": fetch_txt, "write": write_txt, "sq": sq_txt, "bench": bench_json},
}
print(json.dumps(out, indent=1))
