"""The bench line's roofline record is assembled from committed profiler output (profiles/r04_pmc.json): check on the CPU
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
    for kernel, ratio_1080p, ratio_4k in (("k_encode_dense", 1.10, 1.30), ("k_encode_tiles", 1.10, 1.10)):
        for (W, H, n), out_per_frame, worst in (((1920, 1080, 300), 114552.3, ratio_1080p), ((3840, 2160, 300), 460259.7, ratio_4k)):
            rec = b._committed_pmc(W, H, n, kernel)
            assert rec is not None, (W, H, kernel)
            algorithmic = (3 * W * H + out_per_frame) * n
            traffic = b.pmc_traffic(rec)
            # HBM traffic can only exceed the algorithmic bytes; the tile kernel keeps the chroma re-reads in L2 at 4K too
            assert algorithmic * 0.99 < traffic < algorithmic * worst, (kernel, traffic / algorithmic)
            v = b.valu_roofline(rec, 0.62 if W == 1920 else 2.5, sclk_mhz=2150.0)
            assert v["bound"] == "valu-issue" and 0.4 < v["frac"] < 1.0, v
            assert v["insts_per_launch"] > 1e8 and v["cycles_per_inst"] == 2.0 and v["sclk_ghz"] == 2.15
            assert b.valu_roofline(rec, 0.62 if W == 1920 else 2.5)["sclk_source"].startswith("GRBM")
            # lines fetched into an L1 per 128 bytes of pixels: the tile kernel's reason to exist
            lines = rec["l1_to_l2_read_requests"] / rec["pixel_lines_128B"]
            assert (lines < 2.0) if kernel == "k_encode_tiles" else (lines > 3.0), (kernel, lines)
    assert b._committed_pmc(640, 480, 7) is None and b.pmc_traffic(None) is None and b.valu_roofline(None, 1.0) is None


def test_committed_pmc_record_was_measured_on_this_tree():
    """roofline.traffic / roofline.valu are counters replayed from profiles/r04_pmc.json: the record names the kernel sources
    it was measured on, and a change to any of them must be followed by tools/pmc_r04.sh + tools/pmc_record_r04.py."""
    b = _bench()
    rec = b._committed_pmc(1920, 1080, 300, "k_encode_dense")
    assert rec is not None and rec["fresh"], "kernel sources changed since profiles/r04_pmc.json was recorded: re-run tools/pmc_r04.sh"


def test_saved_bench_lines_carry_the_contract_fields():
    """The bench lines committed under profiles/ (what the judge reads beside BENCH_rNN.json) have every field of the contract."""
    need = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline"}
    for name in ("r04_1080p_bench.json", "r04_4k_bench.json"):
        line = next(l for l in open(os.path.join(ROOT, "profiles", name)) if l.startswith("{"))
        d = json.loads(line)
        assert need <= set(d), need - set(d)
        r = d["roofline"]
        assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
        assert abs(r["step_frac"] - r["step_achieved"] / r["peak"]) < 1e-4 and r["step_frac"] < r["frac"]
        assert r["traffic"] and r["valu"] and r["launches_timed"] == d["steps"] and "binding" not in r
        assert "workload" in d["config"] and "model" not in d["config"]
    d = json.loads(next(l for l in open(os.path.join(ROOT, "profiles", "r04_1080p_bench.json")) if l.startswith("{")))
    assert d["config4"]["workload"].startswith("300 x 3840x2160") and d["config4"]["frac"] > 0.3 and d["sustained"]["power"]["power_w"] > 100


def test_bench_launches_its_own_ranks_command():
    """`python bench.py --gpus N` without WORLD_SIZE starts torch.distributed.run as a child (never exec, never touches the GPU
    in the parent): one rank per GPU, rendezvous on 127.0.0.1, the same arguments."""
    b = _bench()
    cmd = b.launch_command(8, ["--gpus", "8", "--steps", "20", "--warmup", "5"], 29555, script="/x/bench.py")
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29555"
    assert cmd[-7:] == ["/x/bench.py", "--gpus", "8", "--steps", "20", "--warmup", "5"]
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "os.exec" not in src and "execv" not in src
