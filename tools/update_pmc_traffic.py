"""profiles/pmc_traffic.json from the per-pass counter CSVs of tools/gpu_profile_round.sh (copied into profiles/ first).
usage: python tools/update_pmc_traffic.py <tag>   (tag = r06: REQUIRED, the round whose profiles/<tag>_* files are read; every
value the CSVs of that round provide is rewritten and the notes are re-stamped with the tag, so the file never cites another round)
traffic bytes per launch = FETCH_SIZE[KB] * 2 * 1024 + WRITE_SIZE[KB] * 1024 (the x2 is the gfx950 FETCH_SIZE correction of
MI355X_MICROARCH.md for 16-B-per-lane reads, which all of these kernels issue)."""
import csv, json, os, sys
if len(sys.argv) < 2:
    sys.exit("usage: python tools/update_pmc_traffic.py <tag>   (e.g. r06)")
TAG = sys.argv[1]
P = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles") + "/"


def avg(path, pat, counter):
    if not os.path.exists(P + path):
        return None
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(P + path)) if pat in r["Kernel_Name"] and r["Counter_Name"] == counter]
    return sum(v) / len(v) if v else None


t = json.load(open(P + "pmc_traffic.json"))


def traffic(tag, pat, extra=None):
    f, w = avg("%s_pmc_%s_FET.csv" % (TAG, tag), pat, "FETCH_SIZE"), avg("%s_pmc_%s_WRI.csv" % (TAG, tag), pat, "WRITE_SIZE")
    if f is None or w is None:
        return None
    tot = f * 2 * 1024 + w * 1024
    if extra:
        fr, wr = avg("%s_pmc_%s_FET.csv" % (TAG, tag), extra, "FETCH_SIZE") or 0, avg("%s_pmc_%s_WRI.csv" % (TAG, tag), extra, "WRITE_SIZE") or 0
        tot += fr * 2 * 1024 + wr * 1024
    return int(tot)


for key, tag, pat, extra in (
        ("sph3d_depthwise_conv3d[16, 8192, 8192, 33, 128, 2, 64]", "fwd", "dwconv_fwd_multi", None),
        ("sph3d_depthwise_conv3d_grad_t[16, 8192, 8192, 33, 128, 2]", "bwd", "dwconv_bwd_t_vec<2, 4, 17", None),
        ("sph3d_pointwise_gemm[131072, 256, 128, 0, 0]", "gemmnn", "_mfma<", None),
        ("sph3d_pointwise_gemm_bnstats[131072, 256, 128]", "gemmnn", "_mfma<", None),
        ("sph3d_pointwise_gemm_tn[32768, 1024, 128]", "gemmtn", "_mfma<", "gemm_reduce_splits"),
        ("sph3d_pointwise_gemm_tn[131072, 256, 128]", "gemmtn0", "_mfma<", "gemm_reduce_splits")):
    v = traffic(tag, pat, extra)
    if v is not None:
        t[key] = v


def busy(tag):
    b = avg("%s_pmc_mfma_%s.csv" % (TAG, tag), "_mfma<", "SQ_VALU_MFMA_BUSY_CYCLES")
    g = avg("%s_pmc_mfma_%s.csv" % (TAG, tag), "_mfma<", "GRBM_GUI_ACTIVE")
    return round(b / (1024 * g / 8), 4) if b and g else None


mb = t.setdefault("mfma_pipe_busy", {})
mb["_note"] = ("SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE/8) per launch: the fraction of the time the matrix pipe of a SIMD is busy; "
               "v_mfma_f32_32x32x16_bf16 (split products, the default) = 32 busy cycles each, v_mfma_f32_32x32x2_f32 (SPH3D_GEMM_SPLIT=0) = 64")
for key, tag in (("sph3d_pointwise_gemm[131072, 256, 128, 0, 0]", "nn"), ("sph3d_pointwise_gemm_bnstats[131072, 256, 128]", "nn"),
                 ("sph3d_pointwise_gemm[131072, 128, 256, 0, 1]", "nt"), ("sph3d_pointwise_gemm_tn[32768, 1024, 128]", "tn"),
                 ("sph3d_pointwise_gemm_tn[131072, 256, 128]", "tn0")):
    v = busy(tag)
    if v is not None:
        mb[key] = v

rows = list(csv.DictReader(open(P + "%s_kernel_trace_by_launch_shape.csv" % TAG)))


def tr(pat, col="mean_us", grid=None):
    for r in rows:
        if pat in r["kernel"] and (grid is None or r["grid_x"] == str(grid)):
            return float(r[col])


tu = t.setdefault("trace_us", {})
tu["sph3d_depthwise_conv3d[16, 8192, 8192, 33, 128, 2, 64]"] = tr("dwconv_fwd_multi<2, 32, 4>")
tu["sph3d_depthwise_conv3d[16, 8192, 8192, 33, 64, 2, 64]"] = tr("dwconv_fwd_multi<2, 16, 4>")
tu["sph3d_depthwise_conv3d_grad_t[16, 8192, 8192, 33, 64, 2]"] = tr("dwconv_bwd_t_vec<2, 4, 17, 2, true>")
tu["sph3d_depthwise_conv3d_grad_t[16, 8192, 8192, 33, 128, 2]"] = tr("dwconv_bwd_t_vec<2, 4, 17, 1, true>", "max_us", 262144)


def dur_us(path, pat):
    """mean dispatch duration of a kernel in a counter pass (one row per counter: take one counter's rows)"""
    if not os.path.exists(P + path):
        return None
    v = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(P + path))
         if pat in r["Kernel_Name"] and r["Counter_Name"] == "SQ_INSTS_MFMA"]
    return sum(v) / len(v) if v else None


for key, tag in (("sph3d_pointwise_gemm_tn[131072, 256, 128]", "tn0"), ("sph3d_pointwise_gemm_tn[32768, 1024, 128]", "tn"),
                 ("sph3d_pointwise_gemm[131072, 256, 128, 0, 0]", "nn")):
    a, b = dur_us("%s_pmc_mfma_%s.csv" % (TAG, tag), "_mfma<"), dur_us("%s_pmc_mfma_%s.csv" % (TAG, tag), "gemm_reduce_splits")
    if a is not None:
        tu[key] = round(a + (b or 0.0), 1)          # product kernel (+ slab sum) in the isolated rocprofv3 pass of that call
t["_round"] = TAG
missing = [k for k in ("sph3d_depthwise_conv3d[16, 8192, 8192, 33, 128, 2, 64]", "sph3d_pointwise_gemm_tn[131072, 256, 128]") if traffic(*{
    "sph3d_depthwise_conv3d[16, 8192, 8192, 33, 128, 2, 64]": ("fwd", "dwconv_fwd_multi"),
    "sph3d_pointwise_gemm_tn[131072, 256, 128]": ("gemmtn0", "_mfma<", "gemm_reduce_splits")}[k]) is None]
if missing:
    sys.exit("profiles/%s_pmc_* counter CSVs are missing for %s: copy the round's CSVs into profiles/ first" % (TAG, missing))


# notes that quote another round's counters move under _history[<that round>]; the fresh note names this round's files only
prev = t.get("_round_of_notes", "r04")
if prev != TAG:
    hist = t.setdefault("_history", {}).setdefault(prev, {})
    for k in [k for k in t if k.startswith("_note") and k != "_note_gather_roofline"]:
        hist[k] = t.pop(k)
    for sub in ("mfma_pipe_busy", "trace_us"):
        for k in [k for k in t.get(sub, {}) if k.startswith("_") and k != "_note"]:
            hist[sub + "." + k] = t[sub].pop(k)
t["_round_of_notes"] = TAG
t["_note"] = ("Round %d (tools/gpu_profile_round.sh %s; raw per-kernel CSVs: profiles/%s_pmc_{fwd,bwd}_{FET,WRI,TCC}.csv, "
              "profiles/%s_pmc_sq{1,2}_{fwd,bwd}.csv, profiles/%s_pmc_mfma_{nn,nt,tn,tn0}.csv, profiles/%s_pmc_gemm{nn,tn,tn0}_{FET,WRI}.csv, "
              "profiles/%s_pmc_nnquery_{after,chain}.csv).  Separate --pmc passes per counter group.  traffic bytes per launch = "
              "FETCH_SIZE[KB]*2*1024 + WRITE_SIZE[KB]*1024 (the x2 is the gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md for "
              "16-B-per-lane reads, which is what all of these kernels issue)." % ((int(TAG[1:]),) + (TAG,) * 6))
t["trace_us"]["_note"] = ("rocprofv3 --kernel-trace of `python bench.py --steps 10` (profiles/%s_kernel_trace_by_launch_shape.csv): mean "
                          "duration of the launch shape; the 128-channel gradient shares its launch shape with the smaller levels: its entry "
                          "is the shape's MAX duration; the GEMM entries are the product kernel (+ the slab sum) in the isolated rocprofv3 "
                          "counter pass of that call (profiles/%s_pmc_mfma_*.csv)" % (TAG, TAG))
json.dump(t, open(P + "pmc_traffic.json", "w"), indent=1)
print({k: v for k, v in t.items() if not k.startswith("_") and not isinstance(v, dict)})
print(mb)
