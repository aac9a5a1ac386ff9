"""Fold the two PMC summaries (tools/pmc_summary.py on a FETCH_SIZE run and a WRITE_SIZE run) into
profiles/pmc_traffic.json: measured HBM bytes per launch for every C-ABI entry point.

    python tools/pmc_to_traffic.py gpurun_out/pmc_fetch.json gpurun_out/pmc_write.json profiles/pmc_traffic.json

Units / corrections (/opt/skills/guides/MI355X_MICROARCH.md, section HBM): rocprofv3 reports FETCH_SIZE and
WRITE_SIZE in kilobytes (x1024 -> bytes).  The guide's gfx950 caveat -- FETCH_SIZE reads exactly half of a
wide (16 B/lane) coalesced streaming read -- is applied to the kernels whose read side is a float4 stream
(WIDE_READERS); the calibration on a known byte count that the guide asks for: channel_bn_apply reads and
writes the same tensor and reports FETCH = 0.51 x WRITE.  Kernels that read 4 B/lane are taken as reported,
which known byte counts confirm (fetch_variance: 6.98 MB reported vs 6.9 MB of maps + points; its WRITE_SIZE
62.9 MB vs the 62.9 MB cost volume).  The 16-byte gathers of the EdgeConv passes are uncalibrated and taken
as reported.
"""
import json
import sys

ENTRY_KERNELS = {
    "pf_conv3d_k3_f32": "conv3d_k3_kernel", "pf_conv3d_k3_few_f32": "conv3d_k3_few_kernel",
    "pf_pointwise_gemm_f32": "pointwise_gemm_",
    "pf_edge_apply_f32": "edge_apply_kernel", "pf_edge_stats_f32": "edge_stats_kernel",
    "pf_flow_features_f32": "flow_features_", "pf_knn_lattice_f32": "knn_",
    # single-chain forward (what a scene lane captures): the towers' first layer is ONE stacked 3 -> 8 + 8 convolution
    # through pf_conv2d_wide_f32, the other ten layers go through pf_conv2d_wide_sets_f32 (round 4: the two entries no
    # longer share one averaged figure; per template instantiation: the "instantiations" table below)
    "pf_conv2d_wide_f32": "conv2d_wide16_kernel<3, 1, 3, 16", "pf_conv2d_wide_sets_f32": "conv2d_wide",
    "pf_conv3d_k3_pair_f32": "conv3d_k3_pair_kernel",
    "pf_conv3d_bottom_f32": "::conv3d_bottom_kernel", "pf_deconv3d_bottom_f32": "::deconv3d_bottom_kernel",
    "pf_fetch_variance_f32": "fetch_variance_kernel", "pf_frustum_variance_f32": "fetch_variance_kernel",
    "pf_frustum_variance_cl_f32": "frustum_variance_cl_kernel", "pf_nchw_to_nhwc_f32": "nchw_to_nhwc_kernel",
    "pf_flow_pyramid_f32": "pyramid_resize_kernel", "pf_deconv3d_k3s2_f32": "deconv3d_k3s2_kernel",
    "pf_channel_bn_fused_f32": "channel_bn_fused_kernel", "pf_channel_bn_apply_f32": "channel_bn_apply_kernel",
    "pf_channel_stats_f32": "channel_stats_kernel", "pf_bn_finalize_f32": "bn_finalize_kernel", "pf_bn_finalize_jobs_f32": "bn_finalize_kernel",
    "pf_resize_bilinear_f32": "resize_bilinear_kernel", "pf_softargmin_prob_f32": "softargmin_prob_kernel",
    "pf_flow_head_f32": "flow_head_kernel", "pf_channel_affine_f32": "channel_affine_kernel",
}
ENTRY_EXCLUDE = {"pf_conv2d_wide_sets_f32": "conv2d_wide16_kernel<3, 1, 3, 16"}
WIDE_READERS = ("channel_stats_kernel", "channel_bn_apply_kernel", "channel_affine_kernel", "pointwise_gemm_",
                "frustum_variance_cl_kernel")      # 16-byte-per-lane streaming reads (the guide's x2)


def main():
    fetch = json.load(open(sys.argv[1]))["kernels"]
    write = json.load(open(sys.argv[2]))["kernels"]
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), bench.py --eager cfg2",
           "entries": {}}
    for entry, sub in ENTRY_KERNELS.items():
        f_sum = f_n = w_sum = w_n = 0.0
        skip = ENTRY_EXCLUDE.get(entry)
        for name, ctrs in fetch.items():
            if skip and skip in name:
                continue
            if sub in name and "FETCH_SIZE" in ctrs:
                f_sum += ctrs["FETCH_SIZE"]["sum"]
                f_n += ctrs["FETCH_SIZE"]["dispatches"]
        for name, ctrs in write.items():
            if skip and skip in name:
                continue
            if sub in name and "WRITE_SIZE" in ctrs:
                w_sum += ctrs["WRITE_SIZE"]["sum"]
                w_n += ctrs["WRITE_SIZE"]["dispatches"]
        if f_n == 0 and w_n == 0:
            continue
        corr = 2.0 if sub in WIDE_READERS else 1.0
        fb = corr * f_sum * 1024.0 / max(f_n, 1.0)
        wb = w_sum * 1024.0 / max(w_n, 1.0)
        out["entries"][entry] = {"fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb,
                                 "bytes_per_launch": fb + wb, "fetch_correction": corr,
                                 "dispatches": int(max(f_n, w_n))}
    # per template instantiation (the kernel name as rocprofv3 prints it, arguments stripped)
    inst = {}
    for name in sorted(set(fetch) & set(write)):
        f, w = fetch[name].get("FETCH_SIZE"), write[name].get("WRITE_SIZE")
        if not f or not w or "(anonymous namespace)" not in name:
            continue
        corr = 2.0 if any(s in name for s in WIDE_READERS) else 1.0
        short = name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
        fb = corr * f["sum"] * 1024.0 / max(f["dispatches"], 1)
        wb = w["sum"] * 1024.0 / max(w["dispatches"], 1)
        inst[short] = {"fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb, "bytes_per_launch": fb + wb,
                       "fetch_correction": corr, "dispatches": int(max(f["dispatches"], w["dispatches"]))}
    out["instantiations"] = inst
    json.dump(out, open(sys.argv[3], "w"), indent=1, sort_keys=True)
    print("wrote", sys.argv[3], len(out["entries"]), "entries")


if __name__ == "__main__":
    main()
