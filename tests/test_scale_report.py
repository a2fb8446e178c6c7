"""CPU test of tools/scale_report.py: the N = 1/2/4/8 lines of `bench.py --gpus N` -> rays/s, efficiency vs N = 1 and the rank-0
communication shares, for the weak (sn64) and the strong (one DTU image) workload."""
import io
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import scale_report  # noqa: E402


def _line(n, value, ms, strong=None, one=None):
    d = {"metric": "rays/sec (64 coarse + 128 fine samples)", "value": value, "unit": "rays/s", "n_gpus": n, "ms_per_step": ms,
         "scaling": "weak", "dtype": "f16x3", "config": {"workload": "sn64 NMR 64x64, 1 input view, 64+128; 65536 rays per rank", "bcast": "tree"}}
    if n > 1:
        d["comm"] = {"bcast_ms_rank0": 0.5, "gather_ms_rank0": 0.25, "grid_bytes": 2 << 20}
        d["extra"] = {"strong_dtu": dict(rays_per_s=strong, ms_per_image=120000.0 / strong * 1e3, bcast_ms_rank0=1.5, gather_ms_rank0=0.3,
                                         bcast_algo="tree")}
    elif one:
        d["extra"] = {"configs": {"dtu": {"f16x3": {"rays_per_s": one, "ms_per_call": 120000.0 / one * 1e3}}}}
    return d


def test_scale_report_reads_driver_files_and_plain_lines(tmp_path):
    runs = [_line(1, 320e3, 204.8, one=133e3), _line(2, 636e3, 206.0, 262e3), _line(4, 1.26e6, 208.0, 510e3), _line(8, 2.48e6, 211.4, 960e3)]
    nested = tmp_path / "SCALE.json"
    nested.write_text(json.dumps({"runs": [{"n": r["n_gpus"], "parsed": r} for r in runs]}))
    plain = tmp_path / "n8.log"
    plain.write_text("some stderr noise\n" + json.dumps(runs[3]) + "\n")
    t = scale_report.rows(scale_report.collect([str(nested)]))
    weak = [k for k in t if k.startswith("weak")][0]
    assert [r[0] for r in sorted(t[weak])] == [1, 2, 4, 8]
    strong = t["strong: one DTU 400x300 image (extra.strong_dtu)"]
    assert sorted(r[0] for r in strong) == [1, 2, 4, 8]
    buf = io.StringIO()
    scale_report.report(t, out=buf)
    text = buf.getvalue()
    assert "7.75x at N = 8" in text and "7.22x at N = 8" in text       # 2.48e6 / 320e3, 960e3 / 133e3
    assert "96.9 %" in text                                             # weak efficiency at N = 8
    assert "0.500 ( 0.24 %)" in text                                    # the broadcast's share of a 211.4 ms step
    assert len(scale_report.collect([str(plain)])) == 1
