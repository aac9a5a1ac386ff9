"""Per-template-instantiation roofline of the inference forward from a rocprofv3 kernel trace.

    # on the GPU box, right after `rocprofv3 --kernel-trace --stats -d DIR -o NAME -- python bench.py ...`:
    python tools/per_kernel_roofline.py summarize DIR/.../NAME_results.db gpurun_out/trace_cfg2.json --depth-maps N
    # anywhere (the summaries are small JSON files):
    python tools/per_kernel_roofline.py report gpurun_out/trace_cfg2.json profiles/r04_per_kernel_roofline \
        --config cfg2 [--pmc-fetch profiles/r04_pmc_fetch.json --pmc-write profiles/r04_pmc_write.json]

`summarize` folds the trace into {kernel name: calls, total us} (the .db is too large to ship back).  `report` joins
that with an ANALYTIC model of every hand-written kernel instantiation at the named BASELINE configuration -- FLOPs and
algorithmic HBM bytes per depth map, the convention of SURVEY.md section 8(d) / DESIGN.md section 4 -- and, when given,
with the PMC traffic of the same instantiation (FETCH_SIZE / WRITE_SIZE in KiB, the guide's x2 FETCH correction for the
16-byte streaming readers), and writes <out>.md and <out>.json: per instantiation launches / depth map, average
microseconds, TFLOP/s and fraction of the 157.3 TF f32 MFMA peak, algorithmic GB/s and fraction of the 8 TB/s HBM peak,
measured traffic and its ratio to the algorithmic bytes.  bench.py's ``roofline.frac`` for the kernel it names is
reproducible from these files.
"""
import argparse
import json
import os
import re
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

MFMA_PEAK_TF, HBM_PEAK_GBS = 157.3, 8000.0
WIDE_READERS = ("channel_stats_kernel", "channel_bn_apply_kernel", "channel_affine_kernel", "pointwise_gemm_",
                "frustum_variance_cl_kernel")


def summarize(db, out, depth_maps):
    con = sqlite3.connect(db)
    rows = con.execute("select name, count(*), sum(duration) / 1000.0 from kernels group by name").fetchall()
    if not depth_maps:                                # one soft-argmin launch per forward
        depth_maps = float(sum(c for n, c, _ in rows if "softargmin_prob_kernel" in n))
    json.dump({"depth_maps": depth_maps, "kernels": {n: {"calls": c, "total_us": t} for n, c, t in rows}},
              open(out, "w"), indent=0)
    print("wrote", out, len(rows), "kernels")


def _targs(name):
    m = re.search(r"<([^<>]*)>\(", name)
    return [a.strip() for a in m.group(1).split(",")] if m else []


def model(cfg):
    """{regex on the kernel name: f(template args) -> (flops per depth map, algorithmic bytes per depth map, what)} for
    one depth map of BASELINE configuration ``cfg`` through the fused inference pipeline (both towers per launch)."""
    from pointmvsnet_amd import synthetic
    H, W, V, D, _, scales, _ = synthetic.CONFIGS[cfg]
    FH, FW = H // 8, W // 8
    vol = D * FH * FW
    res = {1: (H, W), 2: (H // 2, W // 2), 4: (H // 4, W // 4), 8: (FH, FW)}
    layer_in = {(3, 16): 1, (8, 8): 1, (8, 16): 1, (16, 16): 2, (16, 32): 2, (32, 32): 4, (32, 64): 4, (64, 64): 8}
    layer_count = {(16, 16): 2, (32, 32): 2, (64, 64): 2}
    # points of the PointFlow iterations and their sub-grid structure (test mode)
    its = []
    for s in scales:
        h, w = int(H * s), int(W * s)
        its.append((5 * h * w, h, w))
    Ntot = sum(n for n, _, _ in its)

    def conv2d(a):
        K, S, Cin, Cout = int(a[0]), int(a[1]), int(a[2]), int(a[3])
        hi, wi = res[layer_in[(Cin, Cout)]]
        ho, wo = hi // S, wi // S
        samples = V if Cin == 3 else 2 * V
        cin_real, launches = (3 if Cin == 3 else Cin), layer_count.get((Cin, Cout), 1)
        fl = 2.0 * K * K * cin_real * Cout * ho * wo * samples * launches
        by = 4.0 * samples * (cin_real * hi * wi + Cout * ho * wo) * launches
        return fl, by, "tower %d->%d %dx%d/%d" % (cin_real, Cout, K, K, S)

    def conv3d(a):
        NT, S = int(a[0]), int(a[1])
        cin, cout, v_in = {(1, 2): (64, 16, vol), (2, 2): (16, 32, vol // 8), (1, 1): (16, 16, vol // 8),
                           (2, 1): (32, 32, vol // 64)}[(NT, S)]
        v_out = v_in // (S ** 3)
        return 2.0 * 27 * cin * cout * v_out, 4.0 * (cin * v_in + cout * v_out), "VolumeConv %d->%d /%d" % (cin, cout, S)

    def bottom(a):
        stride = int(a[0])
        cin = int(a[1])
        v_in = vol // 64 if stride == 2 else vol // 512
        return 2.0 * 27 * cin * 64 * (vol // 512), 4.0 * (cin * v_in + 64 * (vol // 512)), "VolumeConv %d->64 /%d" % (cin, stride)

    def _dc_group(cout, cells):                      # pf_deconv3d_k3s2_f32's channel-group rule (csrc/deconv3d.hip)
        blocks = (cells + 127) // 128
        if cout % 4 == 0 and blocks * (cout // 4) >= 512:
            return 4
        return 2 if (cout % 2 == 0 and blocks * (cout // 2) >= 512) else 1

    dc_layers = [(32, 16, vol // 64), (16, 8, vol // 8)]           # conv5_0, conv6_0: (Cin, Cout, input cells)

    def deconv(a):
        mine = [l for l in dc_layers if _dc_group(l[1], l[2]) == int(a[0])] or dc_layers
        fl = sum(2.0 * 27 * ci * co * cells for ci, co, cells in mine)
        by = sum(4.0 * cells * (2 * ci + 8 * co) for ci, co, cells in mine)
        return fl, by, "VolumeConv deconv " + " + ".join("%d->%d" % (ci, co) for ci, co, _ in mine)

    def gemm(a):
        kj, nt = int(a[0]), int(a[1])
        K, Nc, what = {(17, 2): (136, 64, "EdgeConvNoC 136->[32|32]"), (4, 2): (32, 64, "EdgeConv 32->[32|32]"),
                       (8, 4): (64, 128, "EdgeConv 64->[64|64]"), (28, 2): (224, 64, "MLP 224->64"),
                       (8, 2): (64, 64, "MLP 64->64"), (8, 1): (64, 16, "MLP 64->16")}[(kj, nt)]
        return 2.0 * Ntot * K * Nc, 4.0 * Ntot * (K + Nc), what

    def edge(passes_out):
        def f(a):
            C = int(a[0])
            launches = 2 if C == 32 else 1                     # E0 and E1 have 32 output channels, E2 has 64
            by = Ntot * launches * (4.0 * C + 16.0 + 4.0 * C * 16 + (4.0 * C * (2 if passes_out else 0)))
            return 0.0, by, "EdgeConv C=%d %s pass" % (C, "apply" if passes_out else "statistics")
        return f

    return [
        (r"conv2d_wide16_kernel<|conv2d_wide_kernel<", conv2d),
        (r"conv3d_k3_pair_kernel<", lambda a: (2.0 * 27 * 64 * 8 * vol, 4.0 * (64 + 8) * vol, "VolumeConv conv0_1 64->8")),
        (r"::conv3d_k3_kernel<", conv3d),
        (r"::conv3d_bottom_kernel<", bottom),
        (r"::deconv3d_bottom_kernel<", lambda a: (2.0 * 27 * 64 * 32 * (vol // 512), 4.0 * (vol // 512) * (64 + 8 * 32),
                                                  "VolumeConv deconv 64->32")),
        (r"deconv3d_k3s2_kernel<", deconv),
        (r"conv3d_k3_few_kernel<", lambda a: (2.0 * 27 * 8 * vol, 4.0 * 9 * vol, "VolumeConv 8->1")),
        (r"pointwise_gemm_direct_kernel<", gemm),
        (r"edge_stats_kernel<", edge(False)),
        (r"edge_apply_kernel<", edge(True)),
        (r"frustum_variance_cl_kernel<", lambda a: (0.0, 4.0 * (V * 64 * FH * FW + 64 * vol + 3 * vol), "coarse warp + variance")),
        (r"flow_features_hyp_kernel<", lambda a: (0.0, sum(4.0 * (V * 112 * h * w + 139 * n + h * w) for n, h, w in its),
                                                  "flow feature assembly")),
        (r"pyramid_resize_kernel", lambda a: (0.0, sum(4.0 * V * (16 * (H // 2) * (W // 2) + 32 * (H // 4) * (W // 4)
                                                                   + 64 * FH * FW + 112 * h * w) for _, h, w in its),
                                              "pyramid resize (+ pending BatchNorm)")),
        (r"knn_net_kernel<", lambda a: (0.0, 28.0 * Ntot / len(its), "lattice kNN (window codes; one iteration per instantiation)")),
        (r"flow_head_kernel<", lambda a: (0.0, 4.0 * Ntot * 16 + sum(28.0 * h * w for _, h, w in its), "flow head")),
        (r"softargmin_prob_kernel", lambda a: (0.0, 4.0 * FH * FW * (D + 2), "soft-argmin + probability")),
    ]


def report(trace, out, cfg, fetch, write):
    t = json.load(open(trace))
    n = float(t["depth_maps"])
    rules = model(cfg)
    pf = json.load(open(fetch))["kernels"] if fetch else {}
    pw = json.load(open(write))["kernels"] if write else {}
    rows, other_us = [], 0.0
    for name, k in t["kernels"].items():
        us_map = k["total_us"] / n
        hit = None
        for pat, fn in rules:
            if re.search(pat, name):
                try:
                    hit = fn(_targs(name))
                except (KeyError, ValueError, IndexError):
                    hit = None
                break
        if hit is None:
            other_us += us_map
            continue
        fl, by, what = hit
        short = re.sub(r"\(.*", "", name.replace("void ", "").replace("(anonymous namespace)::", ""))
        rec = {"kernel": short, "what": what, "launches_per_depth_map": k["calls"] / n, "us_per_depth_map": us_map,
               "avg_us": k["total_us"] / k["calls"], "flops_per_depth_map": fl, "algorithmic_bytes_per_depth_map": by,
               "TFLOPs": fl / us_map / 1e6 if fl else None, "frac_mfma_peak": fl / us_map / 1e6 / MFMA_PEAK_TF if fl else None,
               "algo_GBps": by / us_map / 1e3, "frac_hbm_peak": by / us_map / 1e3 / HBM_PEAK_GBS}
        f, w = pf.get(name, {}).get("FETCH_SIZE"), pw.get(name, {}).get("WRITE_SIZE")
        if f and w:
            corr = 2.0 if any(s in name for s in WIDE_READERS) else 1.0
            per_map = (corr * f["sum"] / f["dispatches"] + w["sum"] / w["dispatches"]) * 1024.0 * k["calls"] / n
            rec["hbm_traffic_bytes_per_depth_map"] = per_map
            rec["traffic_over_algorithmic"] = per_map / by if by else None
        rows.append(rec)
    rows.sort(key=lambda r: -r["us_per_depth_map"])
    total = sum(r["us_per_depth_map"] for r in rows)
    json.dump({"config": cfg, "depth_maps": n, "modelled_kernel_us_per_depth_map": total,
               "other_kernel_us_per_depth_map": other_us, "kernels": rows}, open(out + ".json", "w"), indent=1)
    with open(out + ".md", "w") as f:
        f.write("# per-kernel roofline, %s (%d depth maps in the trace)\n\n" % (cfg, n))
        f.write("Kernel durations: rocprofv3 `--kernel-trace` of the timed execution mode; FLOPs / bytes: analytic per "
                "template instantiation (tools/per_kernel_roofline.py); peaks: 157.3 TF f32 MFMA, 8.0 TB/s HBM.\n\n")
        f.write("Modelled kernels: %.1f us per depth map; everything else (ATen glue, copies, BatchNorm finalizes): %.1f us.\n\n"
                % (total, other_us))
        f.write("| kernel | what | launches | avg us | us / map | TFLOP/s | of MFMA peak | algo GB/s | of HBM peak | traffic / algo |\n")
        f.write("|---|---|---|---|---|---|---|---|---|---|\n")
        for r in rows:
            f.write("| `%s` | %s | %.1f | %.1f | %.1f | %s | %s | %.0f | %.3f | %s |\n" % (
                r["kernel"], r["what"], r["launches_per_depth_map"], r["avg_us"], r["us_per_depth_map"],
                "%.1f" % r["TFLOPs"] if r["TFLOPs"] else "-", "%.3f" % r["frac_mfma_peak"] if r["TFLOPs"] else "-",
                r["algo_GBps"], r["frac_hbm_peak"],
                "x%.2f" % r["traffic_over_algorithmic"] if r.get("traffic_over_algorithmic") else "-"))
    print(open(out + ".md").read()[:4000])


def main():
    ap = argparse.ArgumentParser()
    sub = ap.add_subparsers(dest="cmd", required=True)
    a = sub.add_parser("summarize")
    a.add_argument("db")
    a.add_argument("out")
    a.add_argument("--depth-maps", type=float, default=0.0,
                   help="forwards in the trace (default: the number of soft-argmin launches, one per forward)")
    b = sub.add_parser("report")
    b.add_argument("trace")
    b.add_argument("out")
    b.add_argument("--config", default="cfg2")
    b.add_argument("--pmc-fetch")
    b.add_argument("--pmc-write")
    args = ap.parse_args()
    if args.cmd == "summarize":
        summarize(args.db, args.out, args.depth_maps)
    else:
        report(args.trace, args.out, args.config, args.pmc_fetch, args.pmc_write)


if __name__ == "__main__":
    main()
