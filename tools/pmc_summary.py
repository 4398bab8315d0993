"""Fold rocprofv3 --pmc counter_collection.csv files (one per pass) into per-kernel averages.
    python tools/pmc_summary.py out.json pass1_dir pass2_dir ...
FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB.  MI355X_MICROARCH.md (HBM section): on
gfx950 FETCH_SIZE tallies 128-B requests at 64 B, i.e. reads exactly half the bytes of a wide
coalesced stream, so `fetch_bytes_corrected` = 2 x FETCH_SIZE x 1024; WRITE_SIZE is exact for 16-B
streaming stores and float atomics."""
import collections, csv, glob, json, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

SIMDS, CUS, XCDS = 1024, 256, 8      # MI355X: 256 CUs x 4 SIMD-32, 8 XCDs (GRBM_GUI_ACTIVE is summed over them)

SHORT = [("blend_bwd_quad", "blend_backward"), ("blend_fwd_quad", "blend_forward"), ("param_bwd", "param_backward"),
         ("preprocess_batch_kernel", "preprocess"), ("preprocess_kernel", "preprocess"), ("image_loss_kernel", "image_loss"),
         # the tile sort's kernels are told from the depth sort's by their template arguments: the tile sort moves
         # (tile, gaussian) pairs (scatter MODE 1 / 2, histogram PAIRS = true or the tile-histogram kernel)
         ("tile_hist_kernel", "tile_hist"), ("radix_hist_kernel<256, true>", "tile_hist2"), ("radix_hist_kernel<512, true>", "tile_hist2"),
         ("radix_hist_kernel<256, (bool)1>", "tile_hist2"), ("radix_hist_kernel<512, (bool)1>", "tile_hist2"),
         ("radix_hist", "depth_hist"), ("ranges_from_counts", "bin_ranges"),
         ("radix_scan", "radix_scan"), ("duplicate_kernel", "bin_duplicate"), ("pack_kernel", "blend_pack"),
         # modes 0 / 2: count (by gaussian id / in depth order), 1 / 3: emit (gathering / from the depth-ordered summaries)
         ("culled_pairs_kernel<0>", "bin_count_culled"), ("culled_pairs_kernel<2>", "bin_count_culled"),
         ("culled_pairs_kernel<1>", "bin_duplicate_culled"), ("culled_pairs_kernel<3>", "bin_duplicate_culled"),
         ("culled_pairs_kernel<(int)0>", "bin_count_culled"), ("culled_pairs_kernel<(int)2>", "bin_count_culled"),
         ("culled_pairs_kernel<(int)1>", "bin_duplicate_culled"), ("culled_pairs_kernel<(int)3>", "bin_duplicate_culled"),
         ("count_sorted_kernel", "bin_count"),
         ("count_sum_kernel", "bin_count_sum"), ("count_kernel", "bin_count"), ("ranges_kernel", "bin_ranges"), ("sorted_offsets", "bin_offsets"),
         ("sorted_block_sums", "bin_block_sums")]


def short(name):
    m = re.search(r"radix_scatter_kernel<(\d+), (\d+), (\d+)>", name)
    if m:
        return "depth_scatter" if m.group(3) == "0" else "tile_scatter"
    for pat, s in SHORT:
        if pat in name:
            return s
    return None


def main():
    out_path, dirs = sys.argv[1], sys.argv[2:]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in dirs:
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                if k is None:
                    continue
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
                acc[k]["_dur_ns_" + d].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
                acc[k]["_vgpr"].append(float(r["VGPR_Count"]))
                acc[k]["_lds"].append(float(r["LDS_Block_Size"]))
    res = {}
    for k, cs in acc.items():
        e = {c: sum(v) / len(v) for c, v in cs.items() if not c.startswith("_dur_ns_")}
        durs = [sum(v) / len(v) for c, v in cs.items() if c.startswith("_dur_ns_")]
        e["avg_duration_us_under_pmc"] = sum(durs) / len(durs) / 1e3
        e["launches_per_pass"] = min(len(v) for c, v in cs.items() if not c.startswith("_"))   # a counter of ONE pass
        e["launches_per_view"] = e["launches_per_pass"] / float(os.environ.get("GS_VIEWS", 8))
        if "FETCH_SIZE" in e:
            e["fetch_bytes_raw"] = e["FETCH_SIZE"] * 1024
            e["fetch_bytes_corrected"] = 2 * e["FETCH_SIZE"] * 1024
        if "WRITE_SIZE" in e:
            e["write_bytes"] = e["WRITE_SIZE"] * 1024
        if "fetch_bytes_corrected" in e and "write_bytes" in e:
            e["hbm_traffic_bytes"] = e["fetch_bytes_corrected"] + e["write_bytes"]
        # issue utilisation from the counters: chip cycles of the launch = GRBM_GUI_ACTIVE / 8 (the counter is
        # summed over the XCDs); a wave64 vector instruction occupies its SIMD-32 for 2 cycles (transcendentals,
        # DPP, compares and lane swaps for more: a lower bound), the one scalar unit of a CU issues one SALU
        # instruction per cycle
        if "GRBM_GUI_ACTIVE" in e and "SQ_INSTS_VALU" in e:
            cycles = e["GRBM_GUI_ACTIVE"] / XCDS
            e["clock_ghz"] = round(cycles / (e["avg_duration_us_under_pmc"] * 1e3), 3)
            e["valu_issue_frac"] = round(e["SQ_INSTS_VALU"] * 2.0 / (SIMDS * cycles), 4)
            if "SQ_INSTS_SALU" in e:
                e["salu_issue_frac"] = round(e["SQ_INSTS_SALU"] / (CUS * cycles), 4)
            # LDS: cycles its array was active, summed over the CUs, against CU cycles (one array per CU; the
            # instruction table of MI355X_MICROARCH prices each ds_* form in those cycles)
            if "SQ_LDS_IDX_ACTIVE" in e:
                e["lds_busy_frac"] = round(e["SQ_LDS_IDX_ACTIVE"] / (CUS * cycles), 4)
            if "SQ_WAVE_CYCLES" in e and "SQ_WAIT_INST_ANY" in e:
                e["wave_time_waiting_to_issue"] = round(e["SQ_WAIT_INST_ANY"] / e["SQ_WAVE_CYCLES"], 4)
            if "SQ_WAVE_CYCLES" in e and "SQ_WAIT_ANY" in e:
                e["wave_time_in_waitcnt"] = round(e["SQ_WAIT_ANY"] / e["SQ_WAVE_CYCLES"], 4)
        res[k] = e
    import bench
    res["_meta"] = {"source_sha16": bench.source_sha16(), "passes": dirs,
                    "gaussians": int(os.environ.get("GS_N", 1_000_000)), "width": int(os.environ.get("GS_W", 1920)),
                    "height": int(os.environ.get("GS_H", 1080)),
                    "log_scale": [float(os.environ.get("GS_LS0", 1.0)), float(os.environ.get("GS_LS1", 3.0))],
                    "lists": os.environ.get("GS_LISTS", "exact"), "views": int(os.environ.get("GS_VIEWS", 8)),
                    "workload": "tools/one_view.py: bench scene, GS_VIEWS views fwd+bwd on ONE stream, one batched K1 and K8"}
    json.dump(res, open(out_path, "w"), indent=1, sort_keys=True)
    res.pop("_meta")
    for k, e in sorted(res.items(), key=lambda kv: -kv[1]["avg_duration_us_under_pmc"]):
        print(f"{k:16s} {e['avg_duration_us_under_pmc']:8.1f} us  fetch(corr) {e.get('fetch_bytes_corrected', 0) / 1e6:8.1f} MB  "
              f"write {e.get('write_bytes', 0) / 1e6:8.1f} MB  atomics {e.get('TCC_EA0_ATOMIC_sum', 0) * 64 / 1e6:7.1f} MB")


if __name__ == "__main__":
    main()
