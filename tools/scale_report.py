#!/usr/bin/env python3
"""Scaling report from the JSON lines of `bench.py --gpus N` (N = 1, 2, 4, 8; both workloads).

    python tools/scale_report.py SCALE_r06.json                 # the driver's file: a list / dict of per-N bench lines
    python tools/scale_report.py n1.json n2.json n4.json n8.json  # or one file per run (the last `{...}` line of each is read)
    python bench.py --gpus 1 | python tools/scale_report.py -     # or stdin

Prints, per workload found (the weak-scaling sn64 headline `value`, and `extra.strong_dtu` = ONE 120 000-ray DTU image sharded over
the ranks): rays/s, speed-up and efficiency against N = 1, the step / image time, and the rank-0 communication
(`comm.bcast_ms_rank0`, `comm.gather_ms_rank0`: the one grid broadcast and the one gather per render) as a fraction of
the step -- with the broadcast algorithm (`tree` = RCCL's broadcast, `flat` = 1 -> N-1 point-to-point fan-out over one xGMI link
each) named, so that the first real 8-GPU run yields the curve and the tree-vs-flat decision in one command
(reference: src/render/nerf.py:354-371 bind_parallel / DataParallel(dim=1))."""
import json
import sys


def _lines_of(text):
    out = []
    text = text.strip()
    if not text:
        return out
    try:
        d = json.loads(text)
        stack = [d]
        while stack:
            x = stack.pop()
            if isinstance(x, dict):
                if "metric" in x and "n_gpus" in x:
                    out.append(x)
                else:
                    stack.extend(x.values())
            elif isinstance(x, list):
                stack.extend(x)
        if out:
            return out
    except json.JSONDecodeError:
        pass
    for ln in text.splitlines():
        ln = ln.strip()
        if ln.startswith("{"):
            try:
                d = json.loads(ln)
            except json.JSONDecodeError:
                continue
            if "metric" in d and "n_gpus" in d:
                out.append(d)
    return out


def collect(paths):
    runs = []
    for p in paths:
        text = sys.stdin.read() if p == "-" else open(p).read()
        runs.extend(_lines_of(text))
    return runs


def rows(runs):
    """-> {workload name: [(n, rays/s, ms per step, bcast ms, gather ms, bcast algo)]}"""
    table = {}
    for d in runs:
        n = int(d["n_gpus"])
        comm = d.get("comm") or {}
        wl = (d.get("config") or {}).get("workload", "?")
        key = "weak: %s per rank" % wl.split(";")[0][:60] if d.get("scaling") == "weak" else "strong: %s" % wl[:60]
        table.setdefault(key, []).append((n, float(d["value"]), float(d.get("ms_per_step", 0.0)), comm.get("bcast_ms_rank0"),
                                          comm.get("gather_ms_rank0"), (d.get("config") or {}).get("bcast", d.get("bcast_algo", ""))))
        one = ((d.get("extra") or {}).get("configs") or {}).get("dtu", {}).get(d.get("dtype", "f16x3"))
        if n == 1 and one and "rays_per_s" in one:  # the N = 1 point of the strong-scaling curve: the same image on one GPU (extra.configs.dtu)
            table.setdefault("strong: one DTU 400x300 image (extra.strong_dtu)", []).append(
                (1, float(one["rays_per_s"]), float(one["ms_per_call"]), None, None, ""))
        st = (d.get("extra") or {}).get("strong_dtu")
        if st and "rays_per_s" in st:
            table.setdefault("strong: one DTU 400x300 image (extra.strong_dtu)", []).append(
                (n, float(st["rays_per_s"]), float(st["ms_per_image"]), st.get("bcast_ms_rank0"), st.get("gather_ms_rank0"), st.get("bcast_algo", "")))
    return table


def report(table, out=sys.stdout):
    for key in sorted(table):
        rs = sorted(table[key])
        base = next((r for r in rs if r[0] == 1), None)
        print(key, file=out)
        print("    N        rays/s   speed-up  efficiency    ms/step   bcast ms (share)   gather ms (share)  bcast", file=out)
        for n, v, ms, b, g, algo in rs:
            sp = v / base[1] if base else float("nan")
            eff = sp / n if base else float("nan")
            share = lambda t: "      -        " if t is None or not ms else "%7.3f (%5.2f %%)" % (t, 100.0 * t / ms)  # noqa: E731
            print("  %3d  %12.0f  %8.2fx  %9.1f %%  %9.2f   %s   %s  %s" % (n, v, sp, 100.0 * eff, ms, share(b), share(g), algo or ""), file=out)
        if base is None:
            print("    (no N = 1 line: speed-up and efficiency need one)", file=out)
        big = [r for r in rs if r[0] == max(x[0] for x in rs)][0]
        if base and big[0] > 1:
            print("    => %.2fx at N = %d (north_star: >= 6x at 8 over 1)" % (big[1] / base[1], big[0]), file=out)
    if not table:
        print("no bench lines found", file=out)


def main():
    paths = sys.argv[1:] or ["-"]
    report(rows(collect(paths)))


if __name__ == "__main__":
    main()
