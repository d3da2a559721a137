"""HBM bytes per launch of the K-contiguous bf16 Dense GEMM kernels from two rocprofv3 --pmc passes.

    rocprofv3 --pmc FETCH_SIZE  --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline
    rocprofv3 --pmc WRITE_SIZE  --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline
    python tools/hbm_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r01_gemm_hbm_traffic.json

Units and corrections as in /opt/skills/guides/MI355X_MICROARCH.md: both counters are in KiB;
on gfx950 FETCH_SIZE reports half of the bytes of wide coalesced reads (doubled here), WRITE_SIZE is exact.
"""
import collections
import csv
import glob
import json
import sys


def is_kc_gemm(name):
    # the ping-pong kernels (per-tile and persistent form) and gemm_ring_kernel<bf16, A K-contig, B K-contig, *>
    return "gemm_pp_kernel" in name or "gemm_ppp_kernel" in name or "gemm_ring_kernelIDF16bLb0ELb0E" in name


def per_kernel(dirname, counter):
    vals = collections.defaultdict(list)
    for f in glob.glob(dirname + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and is_kc_gemm(r["Kernel_Name"]):
                vals[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return vals


def main():
    fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
    nf, nw = sum(len(v) for v in fetch.values()), sum(len(v) for v in write.values())
    assert nf and nf == nw, (nf, nw)
    f_kib = sum(sum(v) for v in fetch.values()) / nf
    w_kib = sum(sum(v) for v in write.values()) / nw
    out = {
        "kernel": "gemm_pp_kernel<256 | 192 wide, *>, its persistent form gemm_ppp_kernel (and gemm_ring_kernel<bf16, K-contig, K-contig, *> where chosen): forward Dense and dX = dY.W^T",
        "launches_counted": nf,
        "per_kernel": {k[:90]: {"launches": len(v), "FETCH_SIZE_KiB": sum(v) / len(v),
                                "WRITE_SIZE_KiB": sum(write[k]) / len(write[k])} for k, v in fetch.items()},
        "FETCH_SIZE_KiB_mean": f_kib, "WRITE_SIZE_KiB_mean": w_kib,
        "correction": "gfx950: FETCH_SIZE x2 (wide coalesced reads are reported at half size); WRITE_SIZE exact; KiB",
        "bytes_per_launch": (2.0 * f_kib + w_kib) * 1024.0,
        "algorithmic_bytes_per_launch": 141.25e6,
        "algorithmic_note": "A + B read once, C (and the GELU pre-activation) written once, residual / aux read once: "
                            "104, 76, 230, 155 MB forward (QKV, out, FFN1, FFN2), 129, 51, 230, 155 MB for their dX",
    }
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(json.dumps({k: out[k] for k in ("launches_counted", "FETCH_SIZE_KiB_mean", "WRITE_SIZE_KiB_mean", "bytes_per_launch")}))


if __name__ == "__main__":
    main()
