#!/usr/bin/env python3
"""Writes profiles/r04_pmc.json, the committed PMC record bench.py reads for `roofline.traffic` and `roofline.valu`, from the
summaries of tools/pmc_r04.sh (one directory per kernel and workload):

    python tools/pmc_record_r04.py gpurun_out/pmc4_base_runs_1920x1080 gpurun_out/pmc4_base_tiles_1920x1080 ...

The record carries the SHA-256 of the kernel sources it was measured on; bench.py reports whether that still matches the
tree (`pmc_fresh`) and tests/test_bench_record.py fails on a stale record, so a kernel change cannot leave old counters in
the bench line unnoticed."""
import hashlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SOURCES = ["ec504_imageencoder_amd/csrc/m1v_kernels.hip", "ec504_imageencoder_amd/csrc/m1v_tiles.h", "ec504_imageencoder_amd/csrc/m1v_assemble.h",
           "ec504_imageencoder_amd/csrc/fdct_f32.h"]


def source_sha256():
    """bench.py's own hash (over the sources without comments and whitespace differences), so that both sides agree."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod._pmc_sources_sha256(SOURCES)


def parse(path):
    out, cur = {}, None
    for line in open(path):
        if not line.startswith(" "):
            cur = line.strip()
            out[cur] = {}
        else:
            m = re.match(r"\s+(\S+)\s+n=\s*\d+ mean=(\S+)", line)
            if m and cur:
                out[cur][m.group(1)] = float(m.group(2))
    return out


def main():
    recs = []
    for d in sys.argv[1:]:
        m = re.search(r"pmc4_(\w+?)_(runs|tiles)_(\d+)x(\d+)(?:_n(\d+))?", os.path.basename(d.rstrip("/")))
        if not m:
            raise SystemExit(f"{d}: not a tools/pmc_r04.sh directory")
        kernel = "k_encode_tiles" if m.group(2) == "tiles" else "k_encode_dense"
        c = parse(os.path.join(d, "summary.txt")).get(kernel)
        if not c:
            raise SystemExit(f"{d}: no counters of {kernel}")
        W, H = int(m.group(3)), int(m.group(4))
        rec = {"kernel": kernel, "width": W, "height": H, "frames": 300, "fetch_size_kib": c["FETCH_SIZE"], "write_size_kib": c["WRITE_SIZE"],
               "kernel_us_under_profiler": round(c["DURATION_NS"] / 1e3, 1), "l1_to_l2_read_requests": int(c["TCP_TCC_READ_REQ_sum"]),
               "pixel_lines_128B": W * H * 3 * 300 // 128, "source_dir": os.path.basename(d.rstrip("/")),
               "valu": {"insts_per_launch": int(c["SQ_INSTS_VALU"]), "simds": 1024,
                        "clock_ghz": round(c["GRBM_GUI_ACTIVE"] / 8 / (c["DURATION_NS"] * 1e-9) / 1e9, 3),
                        "source": "rocprofv3 --pmc SQ_INSTS_VALU (tools/pmc_r04.sh)"},
               "waves": int(c["SQ_WAVES"]), "wave_quad_cycles": c["SQ_WAVE_CYCLES"], "wait_any_quad_cycles": c["SQ_WAIT_ANY"],
               "ta_addr_stalled_by_tc_cycles": c["TA_ADDR_STALLED_BY_TC_CYCLES_sum"], "l2_hits": c["TCC_HIT_sum"], "l2_misses": c["TCC_MISS_sum"]}
        asm = parse(os.path.join(d, "summary.txt")).get("k_assemble")
        if asm:
            rec["assemble"] = {"kernel": "k_assemble", "kernel_us_under_profiler": round(asm["DURATION_NS"] / 1e3, 1),
                               "fetch_size_kib": asm.get("FETCH_SIZE"), "write_size_kib": asm.get("WRITE_SIZE"),
                               "insts_valu": int(asm["SQ_INSTS_VALU"]), "insts_salu": int(asm["SQ_INSTS_SALU"]), "waves": int(asm["SQ_WAVES"])}
        recs.append(rec)
    out = os.path.join(ROOT, "profiles", "r04_pmc.json")
    json.dump({"source_sha256": source_sha256(), "sources": SOURCES, "workloads": recs}, open(out, "w"), indent=1)
    print(open(out).read())


if __name__ == "__main__":
    main()
