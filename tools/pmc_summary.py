"""Fold rocprofv3 --pmc counter_collection.csv files (one per pass) into per-kernel averages.
    python tools/pmc_summary.py out.json pass1_dir pass2_dir ...
FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB.  MI355X_MICROARCH.md (HBM section): on
gfx950 FETCH_SIZE tallies 128-B requests at 64 B, i.e. reads exactly half the bytes of a wide
coalesced stream, so `fetch_bytes_corrected` = 2 x FETCH_SIZE x 1024; WRITE_SIZE is exact for 16-B
streaming stores and float atomics."""
import collections, csv, glob, json, re, sys

SHORT = [("blend_bwd_quad", "blend_backward"), ("blend_fwd_quad", "blend_forward"), ("param_bwd", "param_backward"),
         ("preprocess_batch_kernel", "preprocess"), ("preprocess_kernel", "preprocess"), ("image_loss_kernel", "image_loss"), ("radix_scatter", "radix_scatter"), ("radix_hist", "radix_hist"),
         ("radix_scan", "radix_scan"), ("duplicate_kernel", "bin_duplicate"), ("pack_kernel", "blend_pack"),
         ("culled_pairs_kernel<false>", "bin_count_culled"), ("culled_pairs_kernel<true>", "bin_duplicate_culled"),
         ("culled_pairs_kernel<(bool)0>", "bin_count_culled"), ("culled_pairs_kernel<(bool)1>", "bin_duplicate_culled"),
         ("count_sum_kernel", "bin_count_sum"), ("count_kernel", "bin_count"), ("ranges_kernel", "bin_ranges"), ("sorted_offsets", "bin_offsets"),
         ("sorted_block_sums", "bin_block_sums")]


def short(name):
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
        e["launches_per_pass"] = max(len(v) for c, v in cs.items() if not c.startswith("_"))
        if "FETCH_SIZE" in e:
            e["fetch_bytes_raw"] = e["FETCH_SIZE"] * 1024
            e["fetch_bytes_corrected"] = 2 * e["FETCH_SIZE"] * 1024
        if "WRITE_SIZE" in e:
            e["write_bytes"] = e["WRITE_SIZE"] * 1024
        if "fetch_bytes_corrected" in e and "write_bytes" in e:
            e["hbm_traffic_bytes"] = e["fetch_bytes_corrected"] + e["write_bytes"]
        res[k] = e
    json.dump(res, open(out_path, "w"), indent=1, sort_keys=True)
    for k, e in sorted(res.items(), key=lambda kv: -kv[1]["avg_duration_us_under_pmc"]):
        print(f"{k:16s} {e['avg_duration_us_under_pmc']:8.1f} us  fetch(corr) {e.get('fetch_bytes_corrected', 0) / 1e6:8.1f} MB  "
              f"write {e.get('write_bytes', 0) / 1e6:8.1f} MB  atomics {e.get('TCC_EA0_ATOMIC_sum', 0) * 64 / 1e6:7.1f} MB")


if __name__ == "__main__":
    main()
