#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE cannot share a pass:
MI355X_MICROARCH.md, TCC counter budget).

    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_fetch -o f -- python bench.py --steps 1 --warmup 1 --no-cpu --streams 1
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_write -o w -- python bench.py --steps 1 --warmup 1 --no-cpu --streams 1
    python profiles/pmc_summarize.py gpurun_out/pmc_fetch/f_results.db gpurun_out/pmc_write/w_results.db \
           profiles/r01_pmc_hbm_traffic.csv profiles/pmc_traffic.json

Both counters are in KiB.  gfx950 correction (guide, "HBM"): FETCH_SIZE tallies 128-B requests at 64 B,
so it is doubled; calibrated here on temporal_vec4_kernel (reads 300*10000*200*4 B = 2 343 750 KiB,
FETCH_SIZE says 1 171 904; WRITE_SIZE says 2 343 750 = exact).  hbm_bytes = (2*FETCH + WRITE) * 1024,
averaged per launch.  bench.py reads the json to fill roofline.traffic.
"""
import json
import sqlite3
import sys

STAGES = [("iou_bits_sym_kernel", "iou_bits"), ("iou_bits_kernel", "iou_bits_general"), ("adj_build_kernel", "adj_build"), ("adj_rows_kernel", "adj_build"), ("strip_scan_kernel", "adj_prepass"), ("strip_totals_kernel", "adj_prepass"),
          ("sort_kernel", "sort"), ("walk_kernel", "walk"), ("volume_pass_kernel", "temporal"), ("temporal_both_vec4_kernel", "temporal"), ("temporal_vec4_kernel", "temporal"),
          ("transpose_keys_kernel", "transpose_keys"), ("track_pick_kernel", "track_pick"),
          ("track_link_kernel", "track_link"), ("track_suppress_kernel", "track_suppress"),
          ("rescore_spatial_kernel", "rescore_spatial"), ("rescore_series_kernel", "rescore_series"),
          ("rescore_series_wave_kernel", "rescore_series"), ("rescore_adj_kernel", "rescore_adj"), ("track_loop_kernel", "track_loop"),
          ("track_link_memo_kernel", "track_link"), ("track_warm_anchors_kernel", "track_warm"), ("binsort_kernel", "sort"),
          ("sort_list_kernel", "sort_fallback"), ("bucket_kernel", "sort"), ("link_fill_kernel", "track_fill"),
          ("det_nms_kernel", "det_nms"), ("check_order_kernel", "check_order")]


def per_kernel(db, counter):
    cur = sqlite3.connect(db).cursor()
    acc = {}
    for name, val in cur.execute("select kernel_name, value from counters_collection where counter_name=?", (counter,)):
        if "vdet::" not in name:
            continue
        short = name.split("(")[0].replace("void ", "")
        a = acc.setdefault(short, [0, 0.0])
        a[0] += 1
        a[1] += val
    return acc


def main():
    fdb, wdb, out_csv, out_json = sys.argv[1:5]
    fe, wr = per_kernel(fdb, "FETCH_SIZE"), per_kernel(wdb, "WRITE_SIZE")
    rows, js = [], {}
    for k in sorted(set(fe) | set(wr)):
        nf, sf = fe.get(k, [0, 0.0])
        nw, sw = wr.get(k, [0, 0.0])
        f = sf / nf if nf else 0.0
        w = sw / nw if nw else 0.0
        hbm = (2.0 * f + w) * 1024.0
        rows.append((k, max(nf, nw), f, w, hbm))
        for pat, stage in STAGES:
            if pat in k:
                # (several kernels can share a stage -- the counting sort and the LSD kernel's small x1 sorts: the one that
                #  moves more bytes over the run stands for the stage)
                if stage not in js or js[stage]["hbm_bytes_per_launch"] * js[stage]["launches_sampled"] < hbm * max(nf, nw):
                    js[stage] = {"hbm_bytes_per_launch": hbm, "fetch_KiB_raw": f, "write_KiB": w, "kernel": k,
                                 "launches_sampled": max(nf, nw)}
                break
    with open(out_csv, "w") as fo:
        fo.write("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes, --kernel-trace only);"
                 " python bench.py --steps 1 --warmup 1 --no-cpu --streams 1\n")
        fo.write("# counters in KiB; FETCH_SIZE doubled (gfx950 tallies 128-B requests at 64 B; calibrated on"
                 " temporal_vec4_kernel); hbm_bytes = (2*FETCH + WRITE) * 1024, mean per launch\n")
        fo.write("kernel,launches_sampled,fetch_KiB_per_launch_raw,write_KiB_per_launch,hbm_bytes_per_launch_corrected\n")
        for k, n, f, w, hbm in sorted(rows, key=lambda r: -r[4] * r[1]):
            fo.write('"%s",%d,%.1f,%.1f,%.0f\n' % (k, n, f, w, hbm))
    # everything one video step moves: all launches of all vdet kernels / number of videos in the run (= launches of the
    # volume pass, one per step)
    nvid = max([n for k, n, f, w, hbm in rows if "volume_pass_kernel" in k] or [1])
    total = sum(n * hbm for k, n, f, w, hbm in rows) / nvid
    js["_per_video"] = {"hbm_bytes": total, "videos_sampled": nvid,
                        "by_kernel_bytes": {k: n * hbm / nvid for k, n, f, w, hbm in sorted(rows, key=lambda r: -r[4] * r[1])},
                        "algorithmic_bytes": 3216 * 300 * 10000, "note": "sum over all launches of one step (PMC FETCH_SIZE x 2 + WRITE_SIZE); "
                        "algorithmic = 3216 B/box x 3 M boxes (config 2)"}
    with open(out_csv, "a") as fo:
        fo.write("# per video step: %.3f GB over %d videos sampled\n" % (total / 1e9, nvid))
    json.dump(js, open(out_json, "w"), indent=1)
    print(open(out_csv).read())


if __name__ == "__main__":
    main()
