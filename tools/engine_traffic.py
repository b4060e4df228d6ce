#!/usr/bin/env python3
"""Per-kernel-class HBM traffic of the decode step from two rocprofv3 --pmc csv directories (FETCH_SIZE, WRITE_SIZE).
FETCH_SIZE / WRITE_SIZE count KiB; on gfx950 a wide coalesced read stream is tallied at half its bytes
(/opt/skills/guides/MI355X_MICROARCH.md, HBM section): reads are doubled.  Writes traffic.json next to the inputs.
A third directory (optional): a plain --kernel-trace csv run of the same command -> mean kernel duration per class (begin -> end of a dispatch).
usage: engine_traffic.py <fetch_dir> <write_dir> [<kernel_trace_dir>]"""
import collections, csv, glob, json, os, sys

import re
CLASSES = [(r"gemm_wide_kernel", "gemm_quant"), (r"gemm_fullk64_kernel|gemm_splitk64_kernel|gemm_fullk_kernel", "gemm_quant"),
           (r"gemm_wq_kernel(ILi|<)(4|8)[E,]", "gemm_quant"), (r"gemm_smallm", "gemm_quant"),
           (r"gemm_prefill", "gemm_quant"), (r"gemm_wq_kernel(ILi|<)16[E,]", "gemm_lmhead"), (r"paged_attn|attn_reduce", "attn"),
           (r"rope_kv", "rope_kv"), (r"add_rmsnorm", "norm"), (r"reduce_epilogue", "gemm_quant_reduce")]


def klass(name):
    for pat, c in CLASSES:
        if re.search(pat, name):
            return c
    return None


def load(d):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            c = klass(r.get("Kernel_Name", ""))
            if c:
                a = acc[c]
                a[0] += float(r["Counter_Value"]); a[1] += 1
    return acc


def load_times(d):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            c = klass(r.get("Kernel_Name", ""))
            if c:
                a = acc[c]
                a[0] += (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3; a[1] += 1
    return acc


fetch, write = load(sys.argv[1]), load(sys.argv[2])
times = load_times(sys.argv[3]) if len(sys.argv) > 3 else {}
out = {}
print(f"{'class':20s} {'launches':>9s} {'read MB/launch':>15s} {'write MB/launch':>16s}")
for c in sorted(set(fetch) | set(write)):
    rd = 2.0 * fetch[c][0] * 1024 / max(1, fetch[c][1])       # KiB -> bytes, x2 gfx950 correction
    wr = write[c][0] * 1024 / max(1, write[c][1])
    out[c] = {"launches": fetch[c][1], "read_bytes_per_launch": rd, "write_bytes_per_launch": wr}
    if c in times and times[c][1]:
        out[c]["kernel_us_per_launch"] = times[c][0] / times[c][1]
    print(f"{c:20s} {fetch[c][1]:9d} {rd / 1e6:15.2f} {wr / 1e6:16.2f}" + (f"   kernel {out[c]['kernel_us_per_launch']:.2f} us/launch" if "kernel_us_per_launch" in out[c] else ""))
g = out.get("gemm_quant")
if g:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    js = {"qwen2-7b-w4a16": {"batch": 64, "gemm_sources_sha": bench.gemm_sources_sha(), "gemm_sources_sha_over": list(bench.TRAFFIC_KERNEL_SOURCES),
                             "gemm_quant_read_bytes_per_launch": g["read_bytes_per_launch"],
                             "gemm_quant_write_bytes_per_launch": g["write_bytes_per_launch"],
                             "gemm_quant_bytes_per_launch": g["read_bytes_per_launch"] + g["write_bytes_per_launch"],
                             "gemm_quant_kernel_us_per_launch": g.get("kernel_us_per_launch"),
                             "per_class": out,
                             "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over `bench.py --no-graph`: the engine's own "
                                       "launches of the four quantised linears (FETCH_SIZE x2 gfx950 correction); static, collected by "
                                       "tools/engine_traffic.sh"}}
    json.dump(js, open(os.path.join(os.path.dirname(os.path.abspath(sys.argv[1])), "traffic.json"), "w"), indent=1)
