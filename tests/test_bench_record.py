"""The bench line's roofline record is assembled from committed profiler output (profiles/r02_pmc.json): check on the CPU
that the record is there for both full-size workloads, that its figures are consistent with the algorithmic bytes, and that
the helper functions of bench.py accept it — a malformed record would only show on the GPU box, at the end of a round."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)   # bench.py only runs under __main__
    return mod


def test_committed_pmc_record_feeds_the_roofline_fields():
    b = _bench()
    for (W, H, n), out_per_frame in (((1920, 1080, 300), 114552.3), ((3840, 2160, 300), 460259.7)):
        rec = b._committed_pmc(W, H, n)
        assert rec is not None, (W, H)
        algorithmic = (3 * W * H + out_per_frame) * n
        traffic = b.pmc_traffic(rec)
        # HBM traffic can only exceed the algorithmic bytes, and wasted re-reads stay below 30 % (1.04x at 1080p, 1.22x at 4K)
        assert algorithmic * 0.99 < traffic < algorithmic * 1.3, (traffic, algorithmic)
        v = b.valu_roofline(rec, 0.6 if W == 1920 else 2.5)
        assert v["bound"] == "valu-issue" and 0.5 < v["frac"] < 1.0, v
        assert v["insts_per_launch"] > 1e8 and 0.5 < v["ns_per_inst_per_simd"] < 3.0
    assert b._committed_pmc(640, 480, 7) is None and b.pmc_traffic(None) is None and b.valu_roofline(None, 1.0) is None


def test_saved_bench_lines_carry_the_contract_fields():
    """The bench lines committed under profiles/ (what the judge reads beside BENCH_rNN.json) have every field of the contract."""
    need = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline"}
    for name in ("r02_1080p_bench.json", "r02_4k_bench.json"):
        line = next(l for l in open(os.path.join(ROOT, "profiles", name)) if l.startswith("{"))
        d = json.loads(line)
        assert need <= set(d), need - set(d)
        r = d["roofline"]
        assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
        assert r["traffic"] and r["valu"] and r["launches_timed"] == d["steps"]
        assert "workload" in d["config"] and "model" not in d["config"]
