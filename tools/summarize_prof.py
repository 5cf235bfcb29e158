#!/usr/bin/env python3
"""Condense gpurun_out/{prof,pmc_fetch,pmc_write} (rocprofv3 CSV) into tracked files under profiles/.

usage: tools/summarize_prof.py <round-tag>      e.g. r01
Writes profiles/<tag>_kernel_stats.csv (the rocprofv3 --kernel-trace --stats summary, verbatim),
profiles/<tag>_pmc_hbm.csv (per-kernel mean FETCH_SIZE / WRITE_SIZE) and profiles/hbm_traffic.json
(per-launch HBM bytes, corrected as MI355X_MICROARCH.md section HBM prescribes: FETCH_SIZE counts 64 B per
128-B request on gfx950 for wide coalesced streaming reads -> x2; the factor is re-calibrated here on
k_stream_read, which reads exactly 8*n bytes).
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
out = os.path.join(ROOT, "profiles")
os.makedirs(out, exist_ok=True)


def one(pattern):
    g = sorted(glob.glob(os.path.join(ROOT, "gpurun_out", pattern)), key=os.path.getmtime)
    return g[-1] if g else None  # newest: gpurun merges every call's files into the same local directory


ks = one("prof/*/*_kernel_stats.csv")
if ks:
    shutil.copy(ks, os.path.join(out, f"{tag}_kernel_stats.csv"))


def short(name):
    # longest names first: k_stream_read2 must not be averaged into k_stream_read (it reads twice the bytes)
    for k in ("k_stream_read2", "k_stream_read", "k_exsum_segmented", "k_exsum", "k_exdot", "k_finalize", "k_gen",
              "k_gemv_finish", "k_gemvN_fpe_sx", "k_gemvN_fpe", "k_gemvN_sa", "k_gemvT", "k_scale_x", "k_gemm_mfma", "k_gemm_crt", "k_crt_finish", "k_crt_residues", "k_crt_decide", "k_gemm_i8g",
              "k_gemm_i8", "k_i8_slice_contig",
              "k_i8_slice_strided", "k_i8_finish", "k_i8_zero_w", "k_i8_decide", "k_gemm", "k_trsv", "k_dtrsv",
              "k_scan"):
        if k + "<" in name or k + "(" in name:
            return k
    return None


PREDICATED = {"k_gemm_i8", "k_gemm", "k_i8_finish", "k_i8_zero_w", "k_crt_finish"}
PER_CALL = {"k_gemm_crt": "k_crt_decide"}
means = collections.defaultdict(dict)
for which, counter in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    f = one(f"{which}/*/*_counter_collection.csv")
    if not f:
        continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if k and r["Counter_Name"] == counter:
            acc[k].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        # Some kernels are launched with very different amounts of work in one bench run: k_gemm_i8 once per (digit
        # block, k block) pass, exiting at once when the device-side decision does not need that pass; k_exsum on the
        # 2^28-element vectors AND on the 64 MiB chunks of the host-pointer call.  For those the figure wanted is the
        # full-size launch: the mean over the launches within 5 % of the largest.
        if k in PER_CALL and acc.get(PER_CALL[k]):
            # a kernel launched several times per library call (k_gemm_crt: a few moduli per launch, launches beyond
            # the data's modulus count exit at once): the figure wanted is the sum over one CALL = total / number of
            # launches of the once-per-call kernel named in PER_CALL
            means[k][counter] = sum(v) / len(acc[PER_CALL[k]])
            continue
        if k in PREDICATED or k in ("k_exsum", "k_exdot"):
            top = max(v)
            v = [t for t in v if t >= 0.95 * top]
        means[k][counter] = sum(v) / len(v)

# SQ pass: per-kernel means of every counter collected (kernel names kept in full, template arguments included)
f = one("pmc_sq/*/*_counter_collection.csv")
if f:
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "exb::" in name:
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    counters = sorted({c for d in acc.values() for c in d})
    with open(os.path.join(out, f"{tag}_pmc_sq.csv"), "w") as fh:
        fh.write("kernel,launches," + ",".join(counters) + "\n")
        for k, d in sorted(acc.items()):
            def full_size(vals):
                # see the note at PREDICATED: only the launches that did the full-size work (k_gemm_crt: the launches
                # with all their moduli in use)
                if short(k + "(") in PREDICATED or short(k + "(") in ("k_exsum", "k_exdot", "k_gemm_crt"):
                    top = max(vals)
                    vals = [t for t in vals if t >= 0.95 * top] if top > 0 else vals
                return vals
            nl = max(len(full_size(v)) for v in d.values())
            fh.write(k.replace(",", ";") + f",{nl}," + ",".join(f"{sum(full_size(d[c])) / len(full_size(d[c])):.6g}" if d.get(c) else "" for c in counters) + "\n")

if means:
    with open(os.path.join(out, f"{tag}_pmc_hbm.csv"), "w") as fh:
        fh.write("kernel,FETCH_SIZE_KB_mean,WRITE_SIZE_KB_mean\n")
        for k, d in sorted(means.items()):
            fh.write(f"{k},{d.get('FETCH_SIZE', '')},{d.get('WRITE_SIZE', '')}\n")
    n = 1 << 28
    calib = 2.0
    if "k_stream_read" in means and means["k_stream_read"].get("FETCH_SIZE"):
        calib = (8.0 * n) / (means["k_stream_read"]["FETCH_SIZE"] * 1024.0)
    traffic = {"_note": f"HBM bytes per launch = FETCH_SIZE[KB]*1024*{calib:.4f} + WRITE_SIZE[KB]*1024; "
                        f"read factor calibrated on k_stream_read (exactly {8 * n} bytes), round {tag}"}
    for k, d in means.items():
        traffic[k] = d.get("FETCH_SIZE", 0.0) * 1024.0 * calib + d.get("WRITE_SIZE", 0.0) * 1024.0
    json.dump(traffic, open(os.path.join(out, "hbm_traffic.json"), "w"), indent=1)
    print(json.dumps(traffic, indent=1))
